import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
m, n, r, k = 262144, 100000, 256, 10
U = torch.randn(m, r, device='cuda') * 0.05; V = torch.randn(n, r, device='cuda') * 0.05
ref = U[:1024].double() @ V.double().T
for arith in ('fp32', 'split', 'half2'):
    v, i = _ops.predict_topk(U[:1024], V, k, return_values=True, arithmetic=arith)
    err = float((v.double() - torch.gather(ref, 1, i.long())).abs().max() / ref.abs().max())
    same = float((i.long() == torch.topk(ref, k, dim=1)[1]).all(1).float().mean())
    for _ in range(2): _ops.predict_topk(U, V, k, arithmetic=arith)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): _ops.predict_topk(U, V, k, arithmetic=arith)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'r=256 {arith}: {dt*1e3:.2f} ms {2*m*n*r/dt/1e12:.1f} TF-equivalent  err {err:.2e} rows==fp64 {same:.4f}', flush=True)
