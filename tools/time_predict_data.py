"""Fused fp32 predict top-10 timed on different operand DATA (same shapes): if zeros / constant rows run faster than
random data, the gap is clock (power) give-back, not instruction stalls.  usage: python tools/time_predict_data.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
dev = 'cuda'
m, n, r = 262144, 100000, 128
for name, mk in (('random', lambda a, b: torch.randn(a, b, device=dev) * 0.1),
                 ('zeros', lambda a, b: torch.zeros(a, b, device=dev)),
                 ('constant rows', lambda a, b: (torch.randn(1, b, device=dev) * 0.1).expand(a, b).contiguous()),
                 ('random', lambda a, b: torch.randn(a, b, device=dev) * 0.1)):
    U, V = mk(m, r), mk(n, r)
    _ops.predict_topk(U, V, 10, clamp_negatives=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        _ops.predict_topk(U, V, 10, clamp_negatives=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'{name:14s} {dt * 1e3:7.2f} ms  {2 * m * n * r / dt / 1e12:6.1f} TF')
