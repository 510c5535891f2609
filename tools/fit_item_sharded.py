"""Item-row-sharded fit across the ranks of one node (or rehearsed on one card), timed through the public API.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      tools/fit_item_sharded.py [--windows-per-rank q] [--users M --items N --rank r --samples S --nnz NNZ --epochs E]
  TMF_BENCH_REHEARSE=1: every rank on card 0 with a gloo group (host-staged collectives; timings mean nothing).

Every rank draws its own M users (weak scaling, like bench.py) and passes them with model.local_users; the item table
is owned in sub-blocks of every window (model.local_items: a rank initialises only the rows dist.owned_item_rows names).  Rank 0 prints one JSON line: ms per epoch
(max over ranks), interactions/s of the whole job, bytes each walk moves per rank."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from teamoflow_amd import _lib, _windowed  # noqa: E402
from teamoflow_amd import dist as tdist  # noqa: E402
from teamoflow_amd.mf.initializer_graphs import FixedInitializer, Initializer  # noqa: E402
from teamoflow_amd.mf.loss_graphs import MSELoss, WMRBLoss  # noqa: E402
from teamoflow_amd.mf.matrix_factorization import MatrixFactorization  # noqa: E402
from teamoflow_amd.mf.sparse import SparseInteractions, eye  # noqa: E402
from teamoflow_amd.mf.utils import random_sampler_device  # noqa: E402


class OwnedRows(Initializer):
    """The initial rows of the items THIS rank owns (model.local_items): asked for the catalog, returns the block."""

    def __init__(self, rows):
        self.rows = rows

    def initialize_weights(self, n_features, n_components):
        return self.rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--windows-per-rank', type=int, default=2)
    ap.add_argument('--users', type=int, default=1_000_000)
    ap.add_argument('--items', type=int, default=100_000)
    ap.add_argument('--rank', type=int, default=128, dest='r')
    ap.add_argument('--samples', type=int, default=1024)
    ap.add_argument('--nnz', type=int, default=100_000_000)
    ap.add_argument('--epochs', type=int, default=7)
    ap.add_argument('--loss', choices=['wmrb', 'mse'], default='wmrb')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32')
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    rehearse = os.environ.get('TMF_BENCH_REHEARSE') == '1'
    local = 0 if rehearse else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('gloo' if rehearse else 'nccl', **({} if rehearse else {'device_id': dev}))
    _lib.get()
    m, n, r, S, q = args.users, args.items, args.r, args.samples, args.windows_per_rank
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    idx, val = bench.gen_interactions(m, n, args.nnz, 'zipf', rank, dev)
    ld = _lib.padded_ld(r, dtype)
    rows, _, _ = _windowed.window_geometry(n, world * q, ld, 2 if dtype is torch.bfloat16 else 4, world)
    V_all = bench.init_table(n, r, 7, dev)                    # same seed everywhere; only the owned rows are kept
    owned = torch.cat([V_all[g0:g0 + cnt] for _, g0, cnt in tdist.owned_blocks(rows, world * q, world, rank, n)])
    kw = dict(user_weight_graph=FixedInitializer(bench.init_table(m, r, 11 + rank, dev)),
              item_weight_graph=OwnedRows(owned))
    del V_all
    if args.loss == 'wmrb':
        kw.update(loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S)
    else:
        kw.update(loss_graph=MSELoss())
    model = MatrixFactorization(r, **kw)
    model.verbose, model.shard_items, model.factor_dtype = False, q, dtype
    model.local_users, model.local_items = (rank * m, (rank + 1) * m), True
    if args.loss == 'wmrb':
        model.random_ind = random_sampler_device(n, m, S, seed=100 + rank, device=dev)
    t0 = time.perf_counter()
    # user / item feature matrices only give the shapes on the indicator path: the item one is the CATALOG, the user one this block
    model.fit(args.epochs, eye(m), eye(n), SparseInteractions(idx, val, (m, n)), lr=0.1)
    wall = time.perf_counter() - t0
    red = 'cpu' if rehearse else dev
    t = torch.tensor([model.fit_seconds_, float(val.numel())], dtype=torch.float64, device=red)
    if world > 1:
        mx = t[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        tot = t[1:].clone()
        dist.all_reduce(tot)
        t = torch.cat([mx, tot])
    if rank == 0:
        sec = float(t[0]) / args.epochs
        esz = 2 if dtype is torch.bfloat16 else 4
        print(json.dumps(dict(metric='train_interactions_per_sec', value=float(t[1]) / sec, unit='interactions/s', n_gpus=world,
                              ms_per_epoch=sec * 1e3, epochs=args.epochs, windows_per_rank=q, window_rows=rows,
                              plan_seconds=model.plan_seconds_, wall_seconds=wall, loss_first_last=[model.loss_history_[0], model.loss_history_[-1]],
                              item_rows_rank0=int(model.item_rows.numel()),
                              per_rank_per_epoch=dict(all_gather_bytes_received=(2 if args.loss == 'wmrb' else 1) * (world - 1) * q * rows * ld * esz,
                                                      reduce_scatter_bytes_sent=(world - 1) * q * rows * ld * 4),
                              data='synthetic' + (' (REHEARSAL on one card: timings invalid)' if rehearse else ''),
                              config=dict(workload=f'{m} users x {n} items per rank, r={r}, {args.loss.upper()}, item rows sharded over {world} rank(s)'))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
