#!/usr/bin/env python3
"""Benchmark of the matrix-factorization training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full-batch training epoch (user pass + item pass + fresh-Adam updates + loss), the
unit the reference times at matrix_factorization.py:129-177.  Workload = BASELINE.json's metric
configuration: 1M users x 100K items, r = 128, WMRB with S = 1024 static negatives, ~1e8 interactions
("C4"), fp32, synthetic data generated on the device.  With N > 1 every rank holds its
own 1M users (weak scaling), V is replicated and its gradient is exchanged by RCCL reduce-scatter /
all-gather (teamoflow_amd/dist.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from teamoflow_amd import _engine, _lib, _ops  # noqa: E402
from teamoflow_amd import dist as tdist  # noqa: E402
from teamoflow_amd.mf.utils import random_sampler_device  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def gen_interactions(m, n, nnz_target, items, seed, dev):
    """Synthetic interactions of SURVEY.md §8d: user degrees lognormal(4.2, 0.8) rescaled to the
    target; item ids from a power law (alpha = 1) over a fixed permutation ('zipf') or uniform;
    values in {1..5}; unique row-major pairs."""
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(4242))
    inflate = 1.0
    for attempt in range(4):
        # duplicate (user, item) draws collapse (19 % of them at C4 with the power law), so the draw is repeated with
        # an inflated target until the number of UNIQUE pairs is within 1 % of the requested nnz
        g = torch.Generator(device=dev).manual_seed(1000 + seed)
        deg = torch.exp(4.2 + 0.8 * torch.randn(m, device=dev, generator=g))
        deg = torch.clamp(torch.round(deg * (nnz_target * inflate / float(deg.sum()))), 1, n // 4).to(torch.int64)
        total = int(deg.sum())
        u = torch.repeat_interleave(torch.arange(m, device=dev), deg, output_size=total)
        if items == 'zipf':
            x = torch.rand(total, device=dev, generator=g)
            ranks = torch.clamp(torch.pow(float(n + 1), x).to(torch.int64) - 1, 0, n - 1)
            j = perm[ranks]
            del x, ranks
        else:
            j = torch.randint(0, n, (total,), device=dev, generator=g)
        key = torch.unique(u * n + j)
        del u, j
        if key.numel() >= 0.99 * nnz_target:
            break
        inflate *= nnz_target / key.numel()
    u, j = key // n, key % n
    vals = torch.randint(1, 6, (key.numel(),), device=dev, generator=g).to(torch.float32)
    return torch.stack([u, j], dim=1), vals


def init_table(rows, r, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(rows, r, device=dev, generator=g)
    return x * torch.rsqrt(torch.clamp((x * x).sum(), min=1e-12))


def wmrb_bytes(m, n, S, P, r, s=4):
    """Algorithmic bytes per epoch split by kernel."""
    user = m * S * (2 * r * s + 8) + P * (r * s + 12) + m * 2 * r * s
    item = m * S * (r * s + 8) + P * (r * s + 12) + n * 2 * r * s
    return user, item


def mse_bytes(m, n, nnz, r, s=4):
    return nnz * (r * s + 12) + m * 2 * r * s, nnz * (r * s + 12) + n * 2 * r * s


def cpu_baseline(loss, idx, val, R, U0, V0, n, S, lr, users=None, epochs=3):
    """The C/OpenMP closed-form epoch (oracle/sparse_ref.c - "sparse CPU restatement, not the reference
    formulation") on the first `users` users of the same workload with every host core this
    process may use.  A reported baseline, never the measured path."""
    from oracle import sparse_c
    cores = host_cores()
    sparse_c.set_threads(cores)
    users = min(users or (131072 if loss == 'wmrb' else 262144), int(U0.shape[0]))   # ~10-20 s of CPU work at C4 on 16 cores
    rows = idx[:, 0] < users
    sidx, sval = idx[rows].cpu().numpy(), val[rows].cpu().numpy()
    Us, Vs = U0[:users].cpu().numpy(), V0.cpu().numpy()
    plan = sparse_c.Plan(sidx, sval, users, n, R[:users].cpu().numpy() if loss == 'wmrb' else None)

    def epoch(Uc, Vc):
        if loss == 'wmrb':
            return sparse_c.wmrb_epoch(Uc, Vc, plan, n, S, lr, want_grads=False)[:2]
        return sparse_c.mse_epoch(Uc, Vc, plan, lr, want_grads=False)[:2]

    Uc, Vc = epoch(Us, Vs)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    for _ in range(epochs):
        Uc, Vc = epoch(Uc, Vc)
    dt = (time.perf_counter() - t0) / epochs
    return dict(value=len(sval) / dt, unit='interactions/s', cores=cores, kind='port',
                sample=f'{epochs} {loss.upper()} epochs of oracle/sparse_ref.c (C, OpenMP, {cores} threads) on the first {users} '
                       f'users of the same workload against all {n} items ({len(sval)} interactions'
                       + (f', S={S}' if loss == 'wmrb' else '') + f', r={Us.shape[1]}), {dt:.2f} s per epoch; '
                       'sparse closed-form restatement, not the reference\'s dense formulation (that one: reference_formulation_cpu)')


def recall_parity(dev):
    """recall@10 of the engine vs the oracle on the C1 golden case (BASELINE 'recall@10 parity')."""
    from oracle import dense_ref
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'c1_mse.npz')))
    model = MatrixFactorization(5)
    model.user_embedding = torch.tensor(g['U_450']).to(dev)
    model.item_embedding = torch.tensor(g['V_450']).to(dev)
    got = float(model.recall_at_k(torch.tensor(g['A'])).mean())
    want = float(dense_ref.recall_at_k_dense(g['U_450'], g['V_450'], g['A'], 10).mean())
    return got, want


PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_c4_latest.json')
PMC_KERNELS = {'wmrb_user_pass': 'tmf::k_wmrb_user<32, 1, float, false, false>',
               'wmrb_item_pass': 'tmf::k_wsum_pass<32, 1, float>'}


def pmc_traffic(kname):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate runs, bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, the
    gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect counters itself; the workload is
    seeded, so the profile applies to this run.  None when no profile is committed."""
    try:
        d = json.load(open(PMC_FILE))
        return float(d[PMC_KERNELS[kname]]['hbm_traffic_bytes_per_launch_corrected']), os.path.relpath(PMC_FILE, ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box
    shows all 256 host CPUs but grants a 16-core share; spinning up 256 OpenMP threads there stalls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, 16) if n > 64 else n


def small_configs(dev, quick=False):
    """BASELINE configs 1-3 (CPU-runnable shapes): fit() wall time on the GPU (hipGraph-replayed epochs)
    next to the dense-faithful CPU restatement of the reference formulation (oracle/dense_ref.py: identity
    feature matmuls, [m, n] scores, autograd, fresh Adam) timed over the same region the reference times."""
    from oracle import datagen as G
    from oracle import dense_ref
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import WMRBLoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    out = {}
    cases = [('C1', 100, 50, 5, 0.05, 'mse', 1e-2, 450, 450, None),
             ('C2', 943, 1682, 32, 100000 / (0.9 * 943 * 1682), 'mse', 1e-3, 100, 100, None),
             ('C3', 6040, 3706, 64, 1000209 / (0.9 * 6040 * 3706), 'wmrb', 0.1, 100, 2, 3706 // 2)]
    if quick:  # default bench run: C2 in full, one dense CPU epoch of C3 (~10 s of CPU work in all)
        cases = [cases[1], cases[2][:8] + (1,) + cases[2][9:]]
    for name, m, n, r, density, loss, lr, epochs, cpu_epochs, S in cases:
        np.random.seed(0)
        idx, val, shape, A = G.generate_random_interaction(m, n, density=density)
        U0 = G.normal_init(m, r, 1) if loss == 'mse' else G.uniform_init(m, r, 1)
        V0 = G.normal_init(n, r, 2) if loss == 'mse' else G.uniform_init(n, r, 2)
        kw = dict(user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
        R = None
        if loss == 'wmrb':
            R = np.stack([np.random.choice(n, S, replace=False) for _ in range(m)])
            kw.update(loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S)
        model = MatrixFactorization(r, **kw)
        model.verbose = False
        if R is not None:
            model.random_ind = torch.as_tensor(R)
        inter = SparseInteractions(idx, val, shape)
        model.fit(epochs, eye(m), eye(n), inter, lr=lr)      # includes graph capture
        first = model.fit_seconds_
        model.fit(epochs, eye(m), eye(n), inter, lr=lr)
        gpu_s = model.fit_seconds_
        torch.set_num_threads(host_cores())
        ref = dense_ref.fit_dense(U0, V0, idx, val, loss, cpu_epochs, lr, random_ind=R, n_items=n, n_samples=S)
        cpu_per_epoch = ref['seconds'] / cpu_epochs
        k = min(len(ref['loss']), len(model.loss_history_))
        out[name] = dict(shape=[m, n], r=r, loss=loss, nnz=int(len(val)), epochs=epochs,
                         gpu_fit_seconds=gpu_s, gpu_fit_seconds_first_call=first,
                         gpu_interactions_per_sec=len(val) * epochs / gpu_s,
                         cpu_dense_seconds_per_epoch=cpu_per_epoch, cpu_epochs_timed=cpu_epochs,
                         cpu_interactions_per_sec=len(val) / cpu_per_epoch, cpu_threads=torch.get_num_threads(),
                         loss_rel_diff_first_epochs=float(np.abs(np.array(model.loss_history_[:k]) - ref['loss'][:k]).max()
                                                          / np.abs(ref['loss'][:k]).max()),
                         recall_at_10=float(model.recall_at_k(torch.tensor(A)).mean()))
        if cpu_epochs == epochs:  # the CPU restatement ran the whole fit: end-to-end recall@10 parity (BASELINE: within 1e-3)
            want = float(dense_ref.recall_at_k_dense(ref['U'], ref['V'], A, 10).mean())
            out[name].update(recall_at_10_cpu_restatement=want, recall_at_10_abs_diff=abs(out[name]['recall_at_10'] - want))
        if loss == 'wmrb' and not quick:
            # end-to-end WMRB parity: 20 epochs by the engine and by the C/OpenMP closed-form restatement from the same start
            from oracle import sparse_c
            sparse_c.set_threads(host_cores())
            e2e = 20
            plan_c = sparse_c.Plan(idx, val, m, n, R)
            Uc, Vc, closs = U0, V0, []
            for _ in range(e2e):
                Uc, Vc, mean, _t = sparse_c.wmrb_epoch(Uc, Vc, plan_c, n, S, lr, want_grads=False)
                closs.append(mean)
            model.fit(e2e, eye(m), eye(n), inter, lr=lr)
            got = float(model.recall_at_k(torch.tensor(A)).mean())
            want = float(dense_ref.recall_at_k_dense(Uc, Vc, A, 10).mean())
            out[name]['end_to_end_20_epochs'] = dict(
                recall_at_10_engine=got, recall_at_10_c_restatement=want, abs_diff=abs(got - want),
                loss_rel_diff=float(np.abs(np.array(model.loss_history_) - np.array(closs)).max() / np.abs(closs).max()))
        log(f'[bench] {name}: {out[name]}')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--users', type=int, default=1_000_000)
    ap.add_argument('--items', type=int, default=100_000)
    ap.add_argument('--rank', type=int, default=128, dest='r')
    ap.add_argument('--nnz', type=int, default=100_000_000)
    ap.add_argument('--samples', type=int, default=1024)
    ap.add_argument('--loss', choices=['wmrb', 'mse'], default='wmrb')
    ap.add_argument('--item-dist', choices=['zipf', 'uniform'], default='zipf')
    ap.add_argument('--lr', type=float, default=0.1)
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32', help='factor storage (arithmetic is fp32 either way)')
    ap.add_argument('--small-configs', action='store_true', help='also time BASELINE configs 1-3 (GPU fit vs dense CPU restatement)')
    ap.add_argument('--no-extras', action='store_true', help='skip cpu baseline / predict / mse side measurements')
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner at communicator
    # creation) write to file descriptor 1 as well, so keep a private copy of stdout for the JSON line and point
    # fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # TMF_BENCH_REHEARSE=1: every rank on card 0 with a gloo group (dist.py stages the collectives through the host) -
    # a functional rehearsal of the N>1 path on a one-GPU box; its timings mean nothing
    rehearse = os.environ.get('TMF_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dp_mode = world > 1 or os.environ.get('TMF_BENCH_FORCE_DP') == '1'  # the env knob rehearses the N>1 path on one GPU
    if dp_mode:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29577')
            torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        elif rehearse:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=dev)
    red_dev = 'cpu' if rehearse else dev   # where the two scalar reductions of the report live
    _lib.get()

    m, n, r, S = args.users, args.items, args.r, args.samples
    t_prep = time.perf_counter()
    idx, val = gen_interactions(m, n, args.nnz, args.item_dist, rank, dev)
    nnz = int(val.numel())
    n_pad = tdist.padded_rows(n, world)
    U0 = init_table(m, r, 11 + rank, dev)
    V0 = torch.zeros(n_pad, r, device=dev)
    V0[:n] = init_table(n, r, 7, dev)  # identical on every rank
    plan = _engine.InteractionPlan(idx, val, m, n_pad, user_chunks=_engine.mse_user_chunks() if args.loss == 'mse' else 1,
                                   csc=args.loss == 'mse')
    wplan, R = None, None
    if args.loss == 'wmrb':
        R = random_sampler_device(n, m, S, seed=100 + rank, device=dev)
        wplan = _engine.WmrbPlan(plan, R, user_chunks=_engine.default_user_chunks(m, _lib.padded_ld(r), n_items=n),
                                 item_slices=_engine.default_item_slices(n, _lib.padded_ld(r)), n_components=r)
    tdtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    sbytes = 2 if args.dtype == 'bf16' else 4
    st = _engine.TrainState(U0, V0, plan, r, wplan, dtype=tdtype)
    adam = _engine.adam_constants(args.lr)
    c = n / S
    torch.cuda.synchronize()
    if rank == 0:
        log(f'[bench] prepared {nnz} interactions, m={m} n={n} r={r} S={S} in {time.perf_counter() - t_prep:.1f} s; '
            f'{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak')

    prof = _engine.KernelTimer()
    loss_buf = torch.zeros(args.steps + args.warmup + 1, dtype=torch.float64, device=dev)
    if dp_mode:
        backend = tdist.HipBackend(st, args.loss, c, adam, prof=None)
        dp = tdist.DataParallelEpoch(backend, plan.n_pos if args.loss == 'wmrb' else nnz)

    def step(i, p):
        if dp_mode:
            backend.prof = p
            loss_buf[i] = dp.step()
        else:
            if args.loss == 'wmrb':
                _engine.epoch_wmrb(st, adam, c, loss_buf[i:i + 1], prof=p)
            else:
                _engine.epoch_mse(st, adam, loss_buf[i:i + 1], prof=p)
            st.swap()

    def fence():
        torch.cuda.synchronize()
        if dp_mode:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, None)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, prof)
    fence()
    elapsed = time.perf_counter() - t0
    if dp_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
        tot = torch.tensor([float(nnz)], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(tot)
        nnz_total = float(tot)
    else:
        nnz_total = float(nnz)
    ms_per_step = elapsed / args.steps * 1e3

    # dominant kernel + roofline (HIP events on the launch stream, inside the timed region)
    if args.loss == 'wmrb':
        ub, ib = wmrb_bytes(m, n, S, plan.n_pos, r, sbytes)
        ums, ims = prof.mean_ms('wmrb_user_pass'), prof.mean_ms('wmrb_item_pass')
        sliced = wplan.sliced
        # dominant single kernel: the fused user pass, or (sliced user pass = 4 kernels) the item gather-sum
        if not sliced and ums >= ims:
            kname, kbytes, kms = 'wmrb_user_pass', ub, ums
        else:
            kname, kbytes, kms = 'wmrb_item_pass', ib, ims
        other = {'wmrb_user_pass_ms': ums, 'wmrb_user_pass_alg_bytes': ub, 'wmrb_item_pass_ms': ims,
                 'wmrb_item_pass_alg_bytes': ib,
                 'wmrb_user_pass_form': (f'sliced: scores + hinge + gradU + finish kernels over {wplan.n_slices} item slices'
                                         if sliced else 'fused single kernel'),
                 'wmrb_item_lists_user_blocks': wplan.user_chunks}
    else:
        ub, ib = mse_bytes(m, n, nnz, r, sbytes)
        kname, kbytes = 'mse_item_pass', ib
        kms = prof.mean_ms('mse_item_pass')
        other = {'mse_user_pass_ms': prof.mean_ms('mse_user_pass'), 'mse_user_pass_alg_bytes': ub}
    achieved = kbytes / (kms * 1e-3) / 1e9
    # the committed counters were collected on the default workload: only that run may quote them
    default_workload = (m, n, r, S, args.nnz, args.loss, args.item_dist, args.dtype) == \
        (1_000_000, 100_000, 128, 1024, 100_000_000, 'wmrb', 'zipf', 'f32')
    traffic, traffic_src = pmc_traffic(kname) if default_workload else (None, None)
    roofline = dict(bound='hbm', kernel=kname, achieved=achieved, peak=HBM_PEAK / 1e9, unit='GB/s',
                    frac=achieved / (HBM_PEAK / 1e9), traffic=traffic, traffic_source=traffic_src,
                    traffic_rate_frac=(traffic / (kms * 1e-3) / HBM_PEAK) if traffic else None, kernel_ms=kms,
                    alg_bytes_per_launch=kbytes,
                    note='achieved counts ALGORITHMIC bytes (SURVEY 8d); gathers served by L2 / Infinity Cache let it exceed the HBM peak; '
                         'traffic_rate_frac = measured fabric bytes (PMC) / kernel time / peak',
                    epoch_alg_bytes=ub + ib, epoch_frac=(ub + ib) / (ms_per_step * 1e-3) / HBM_PEAK, **other)

    out = dict(metric='train_interactions_per_sec', value=nnz_total / (elapsed / args.steps), unit='interactions/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
               scaling='weak', vs_baseline=None, dtype='f32' if args.dtype == 'f32' else 'bf16 storage / f32 arithmetic',
               data='synthetic' + (' (REHEARSAL: all ranks on one card, host-staged gloo collectives - timings invalid)' if rehearse else ''),
               config=dict(workload=f'C4: {m} users x {n} items per GPU, r={r}, {args.loss.upper()}'
                                    + (f' S={S}' if args.loss == 'wmrb' else '') + f', item ids {args.item_dist}, '
                                    f'lognormal user degrees', interactions_per_gpu=nnz, positives_per_gpu=plan.n_pos,
                           parallelism=f'user-partition dp{world}', lr=args.lr),
               roofline=roofline)

    if rank == 0 and not args.no_extras and world == 1:
        losses = loss_buf[:args.steps + args.warmup].cpu().numpy() / (plan.n_pos if args.loss == 'wmrb' else nnz)
        out['loss_first_last'] = [float(losses[0]), float(losses[-1])]
        out['cpu_baseline'] = cpu_baseline(args.loss, idx, val, R, U0, V0[:n], n, S, args.lr)
        # predict rows/s: stable top-10 over the full catalog, fused GEMM + top-k (no [m, n] matrix)
        Ue, Ve = st.U[:, :r], st.V[:n, :r]
        rows = min(m, 262144)
        _ops.predict_topk(Ue[:rows], Ve, 10, clamp_negatives=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _ops.predict_topk(Ue[:rows], Ve, 10, clamp_negatives=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        out['predict_rows_per_sec'] = rows / dt
        out['predict_tflops'] = 2.0 * rows * n * r / dt / 1e12
        out['predict_note'] = (f'stable top-10 of U.V^T over all {n} items for {rows} users, fused MFMA GEMM + top-k '
                               + ('(tmf_predict_topk_bf16: bf16 MFMA, fp32 accumulate; dense bf16 peak ~2500 TF)' if args.dtype == 'bf16'
                                  else '(tmf_predict_topk_f32: exact-fp32 MFMA, peak 157.3 TF)'))
        got, want = recall_parity(dev)
        out['recall_at_10'] = dict(engine=got, oracle=want, abs_diff=abs(got - want),
                                   case='C1 golden fixture: ranking of the oracle-trained tables (450 epochs)')
    if rank == 0 and world == 1 and (args.small_configs or not args.no_extras):
        # the reference's own (dense, full-batch) formulation on the host cores next to the engine, BASELINE configs 1-3
        out['reference_formulation_cpu'] = small_configs(dev, quick=not args.small_configs)
        c2 = out['reference_formulation_cpu'].get('C2', {})
        if 'recall_at_10_abs_diff' in c2 and 'recall_at_10' in out:
            out['recall_at_10']['end_to_end_C2'] = dict(engine=c2['recall_at_10'], oracle=c2['recall_at_10_cpu_restatement'],
                                                        abs_diff=c2['recall_at_10_abs_diff'],
                                                        case='C2 (943 x 1682, r=32, MSE): 100 epochs trained by each side from the same start')
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if dp_mode:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
