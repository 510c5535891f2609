import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
m, n, r = 1 << 20, 1_000_000, 256
U = (torch.randn(m, r, device='cuda') * 0.1).to(torch.bfloat16)
V = (torch.randn(n, r, device='cuda') * 0.1).to(torch.bfloat16)
_ops.predict_topk(U[:65536], V, 10, clamp_negatives=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
_ops.predict_topk(U, V, 10, clamp_negatives=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'bf16 fused top-10: {m} users x {n} items r={r}: {dt*1e3:.1f} ms, {m/dt:.3e} users/s, {2*m*n*r/dt/1e15:.3f} PFLOP/s')
