"""Fused top-k predict on SHORT catalogs (1, 4, 16, 64 tiles of 128 items): what the first tiles - where almost every score is a
candidate - cost.  python tools/time_predict_small.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
dev = 'cuda'
m, r = 262144, 128
U = torch.randn(m, r, device=dev) * 0.1
for n in (128, 512, 2048, 8192):
    V = torch.randn(n, r, device=dev) * 0.1
    for k in (1, 10, 16, 17):
        for _ in range(2):
            _ops.predict_topk(U, V, k, clamp_negatives=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            _ops.predict_topk(U, V, k, clamp_negatives=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'n={n:5d} k={k:2d}: {dt * 1e3:8.3f} ms   {2 * m * n * r / dt / 1e12:6.1f} TF', flush=True)
