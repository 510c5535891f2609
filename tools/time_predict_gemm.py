"""Materialised predict (tmf_predict_gemm_f32): TFLOP/s and the output write rate.  usage: python tools/time_predict_gemm.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
for m, n, r in ((8192, 8192, 128), (32768, 32768, 128), (6040, 3706, 64), (65536, 100000, 128)):
    U = torch.randn(m, r, device='cuda') * 0.1
    V = torch.randn(n, r, device='cuda') * 0.1
    _ops.predict_gemm(U, V)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        out = _ops.predict_gemm(U, V)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'{m} x {n} x {r}: {dt * 1e3:.2f} ms  {2 * m * n * r / dt / 1e12:.1f} TF  output {4 * m * n / dt / 1e12:.2f} TB/s')
    del out
