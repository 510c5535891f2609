import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
dev = 'cuda'
m, n = 262144, 100000
shapes = [(128, 10)] if len(sys.argv) < 2 else [tuple(map(int, a.split(','))) for a in sys.argv[1:]]
for r, k in shapes:
    U = torch.randn(m, r, device=dev) * 0.05
    V = torch.randn(n, r, device=dev) * 0.05
    for _ in range(2):
        _ops.predict_topk(U, V, k, arithmetic=os.environ.get('ARITH', 'split'))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        _ops.predict_topk(U, V, k, arithmetic=os.environ.get('ARITH', 'split'))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'{os.environ.get("TMF_LIB", "default"):32s} r={r:3d} k={k:2d}: {dt * 1e3:8.2f} ms   {2 * m * n * r / dt / 1e12:6.1f} TF fp32-equivalent', flush=True)
