"""Rank of a multi-rank rehearsal of the data-parallel fit on ONE card: every rank uses cuda:0 and a gloo
group (RCCL refuses duplicate devices), so dist.py stages the collectives through the host while the local
passes run on the HIP engine - the same code path bench.py --gpus N takes, minus the wire.
Rank 0 also runs the single-process fit and writes the comparison as JSON.

With a third argument q >= 1 the ranks run the item-row-sharded fit instead (model.shard_items = q windows per rank:
dist.ItemShardedEpoch - window broadcasts, per-window reduce into the owner), compared with the same single-process fit.

env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT; TMF_REHEARSE_NCCL=1: one GPU per rank and an nccl (RCCL) group instead - no longer
a rehearsal.  usage: python tools/dp_rehearsal.py OUT.json [loss] [q]"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow.mf.initializer_graphs import FixedInitializer  # noqa: E402
from teamoflow.mf.loss_graphs import MSELoss, WMRBLoss  # noqa: E402
from teamoflow.mf.matrix_factorization import MatrixFactorization  # noqa: E402
from teamoflow.mf.sparse import SparseInteractions, eye  # noqa: E402
from teamoflow_amd import dist as tdist  # noqa: E402


def main():
    out, loss = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'wmrb')
    shard = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if os.environ.get('TMF_REHEARSE_NCCL') == '1':
        # the real thing on a box with one GPU per rank: RCCL collectives, nothing staged through the host (the asynchronous
        # all-gather / reduce-scatter branch of dist.ItemShardedEpoch, the in-place all-gather of dist.DataParallelEpoch)
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    m, n, r, S, lr, epochs = 3001, 701, 64, 48, 0.05, 3      # n not a multiple of world: V is padded
    deg = np.minimum(rng.zipf(1.6, m), n // 2)                 # skewed users: the partition is by cost, not by count
    rows = np.repeat(np.arange(m), deg)
    cols = np.concatenate([rng.choice(n, d, replace=False) for d in deg])
    order = np.lexsort((cols, rows))
    idx = np.stack([rows, cols], 1)[order]
    val = rng.integers(1, 6, len(idx)).astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, S, replace=False) for _ in range(m)])

    def run(parallel, epochs=epochs):
        kw = dict(user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
        if loss == 'wmrb':
            kw.update(loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S)
        else:
            kw.update(loss_graph=MSELoss())
        model = MatrixFactorization(r, **kw)
        model.verbose, model.data_parallel = False, parallel and not shard
        if os.environ.get('TMF_REHEARSE_DTYPE') == 'bf16':     # config 5's storage: bf16 rows cross the wire, fp32 gradients
            model.factor_dtype = torch.bfloat16
        model.shard_items = shard if parallel else 0
        if loss == 'wmrb':
            model.random_ind = torch.as_tensor(R)
        model.fit(epochs, eye(m), eye(n), SparseInteractions(idx, val, (m, n)), lr=lr)
        if parallel and shard:   # the catalog fits here: assemble it for the comparison
            model.top10_sharded = model._top_items(10, True)     # ranked over the windows (a collective), before the table is assembled
            model.item_embedding = tdist.gather_item_embedding(model, n)
        return model

    dp1 = run(True, 1)                                         # one epoch: U must not depend on the partition at all
    U_dp1 = tdist.gather_user_embedding(dp1, m)
    dp = run(True)
    U_dp = tdist.gather_user_embedding(dp, m)
    topk_equal = None
    if shard:
        # ranking over the sharded catalog == ranking the assembled tables with the resident fused kernel, exactly
        from teamoflow_amd import _ops
        mine = _ops.predict_topk(dp.user_embedding, dp.item_embedding.to(dp.user_embedding.dtype), 10, clamp_negatives=True)
        same = [None] * world
        dist.all_gather_object(same, bool(torch.equal(mine, dp.top10_sharded)))
        topk_equal = all(same)
    # mean recall@10 over all users: every rank ranks its own users, two doubles are all-reduced
    b_, e_ = dp.user_block
    sel = (idx[:, 0] >= b_) & (idx[:, 0] < e_)
    loc = idx[sel].copy()
    loc[:, 0] -= b_
    saved = dp.item_embedding
    if shard:
        dp.item_embedding = None         # a sharded model must not touch a table here: it ranks over the windows
    recall_all = tdist.mean_recall_at_k(dp, SparseInteractions(loc, val[sel], (e_ - b_, n)))
    dp.item_embedding = saved
    blocks = [None] * world
    dist.all_gather_object(blocks, dp.user_block)
    item_rows = [None] * world
    dist.all_gather_object(item_rows, dp.item_rows.cpu().tolist() if shard else None)
    if rank == 0:
        one, one1 = run(False), run(False, 1)
        f32 = lambda m: (m.user_embedding.float(), m.item_embedding.float())   # noqa: E731
        (one.user_embedding, one.item_embedding), (one1.user_embedding, one1.item_embedding) = f32(one), f32(one1)
        dp.item_embedding, dp1.item_embedding = dp.item_embedding.float(), dp1.item_embedding.float()
        # the same number from the assembled tables in one process
        whole = run(False, 0)
        whole.user_embedding, whole.item_embedding = U_dp.to(dp.user_embedding.dtype), dp.item_embedding.to(dp.user_embedding.dtype)
        recall_ref = float(whole.recall_at_k(SparseInteractions(idx, val, (m, n))).double().mean())
        res = {'recall_all_ranks': recall_all, 'recall_assembled_tables': recall_ref,
               'U1_equal': bool(torch.equal(U_dp1, one1.user_embedding)),
               'U1_frac_close': float(((U_dp1 - one1.user_embedding).abs() <= 1e-6).float().mean()),
               'U1_max_abs_diff': float((U_dp1 - one1.user_embedding).abs().max()),
               'item_rows_per_rank': [len(x) for x in item_rows] if shard else None,
               'item_rows_partition_the_catalog': sorted(sum(item_rows, [])) == list(range(n)) if shard else None, 'shard_items': shard, 'sharded_top10_equals_resident': topk_equal,
               'V1_frac_close': float(((dp1.item_embedding - one1.item_embedding).abs() <= 1e-6).float().mean()),
               'V1_max_abs_diff': float((dp1.item_embedding - one1.item_embedding).abs().max()),
               'world': world, 'loss': loss, 'blocks': blocks,
               'loss_dp': dp.loss_history_, 'loss_one': one.loss_history_,
               'U_max_abs_diff': float((U_dp - one.user_embedding).abs().max()),
               'V_max_abs_diff': float((dp.item_embedding - one.item_embedding).abs().max()),
               'U_frac_equal': float((U_dp == one.user_embedding).float().mean()),
               'V_frac_equal': float((dp.item_embedding == one.item_embedding).float().mean()),
               'recall_dp_local': float(dp.recall_at_k(SparseInteractions(idx, val, (m, n))).mean())
               if world == 1 else None}
        with open(out, 'w') as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
