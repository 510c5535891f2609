"""tmf_predict_topk_split_f32 against the fp32 MFMA kernel and an fp64 reference (small), then timed at the C4 predict shape.
python tools/split_check.py [quick]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
dev = 'cuda'
torch.manual_seed(0)
ok = True
for (m, n, r, k, clamp) in ((1000, 3000, 128, 10, False), (513, 129, 5, 3, False), (256, 128, 32, 32, True), (700, 5000, 64, 16, False),
                            (300, 1000, 100, 10, False), (4096, 20000, 128, 10, False), (64, 1, 7, 1, False), (2048, 40000, 128, 10, False),
                            (1500, 50001, 64, 16, True), (1024, 33000, 32, 5, False), (600, 70000, 96, 1, False)):
    U = torch.randn(m, r, device=dev) * 0.05
    V = torch.randn(n, r, device=dev) * 0.05
    ref = U.double() @ V.double().T
    if clamp:
        ref = ref.clamp_min(0)
    v32, i32 = _ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='fp32')
    vs, ix = _ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='split')
    vh, ih = _ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='half2')
    eh = (vh.double() - torch.gather(ref, 1, ih.long())).abs().max().item() / ref.abs().max().item()
    sameh = (ih.long() == torch.topk(ref, k, dim=1)[1]).all(1).float().mean().item()
    norm = ref.abs().max().item()
    e32 = (v32.double() - torch.gather(ref, 1, i32.long())).abs().max().item() / norm
    es = (vs.double() - torch.gather(ref, 1, ix.long())).abs().max().item() / norm
    # the lists must be sorted (value desc, index asc) and every value must be the score of its index
    sorted_ok = bool(((vs[:, :-1] > vs[:, 1:]) | ((vs[:, :-1] == vs[:, 1:]) & (ix[:, :-1] < ix[:, 1:]))).all()) if k > 1 else True
    rv, ri = torch.topk(ref, k, dim=1)
    same64 = (ix.long() == ri).all(1).float().mean().item()
    same32 = (ix == i32).all(1).float().mean().item()
    kth_ok = bool((vs[:, -1].double() >= rv[:, -1] - 1e-6 * norm).all())
    good = es < 1e-6 and eh < 1.5e-6 and sorted_ok and kth_ok and (clamp or (same64 > 0.99 and sameh > 0.99))
    ok &= good
    print(f'm={m} n={n} r={r} k={k} clamp={clamp}: err fp32 {e32:.2e} split {es:.2e} half2 {eh:.2e} (rows == fp64 {sameh:.4f}); rows == fp64 top-k {same64:.4f}, == fp32 kernel {same32:.4f}; '
          f'sorted {sorted_ok} kth {kth_ok} -> {"ok" if good else "FAIL"}', flush=True)
print('ALL OK' if ok else 'FAILED')
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    sys.exit(0 if ok else 1)
m, n = 262144, 100000
for r in (128, 64, 32):
    U = torch.randn(m, r, device=dev) * 0.05
    V = torch.randn(n, r, device=dev) * 0.05
    for k in (10, 32):
        for arith in ('fp32', 'split', 'half2'):
            for _ in range(2):
                _ops.predict_topk(U, V, k, arithmetic=arith)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                _ops.predict_topk(U, V, k, arithmetic=arith)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f'r={r:3d} k={k:2d} {arith:5s}: {dt * 1e3:8.2f} ms   {2 * m * n * r / dt / 1e12:6.1f} TF fp32-equivalent   {m / dt / 1e6:6.2f} M users/s', flush=True)
sys.exit(0 if ok else 1)
