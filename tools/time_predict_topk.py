"""Fused top-k predict throughput: python tools/time_predict_topk.py [r ...]   (fp32 tables, 262144 users x 100000 items;
TMF_TIME_K=10 restricts the k values)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops  # noqa: E402

dev = 'cuda'
for r in [int(a) for a in sys.argv[1:]] or [128]:
    for k in [int(x) for x in os.environ.get('TMF_TIME_K', '10,64').split(',')]:
        m, n = 262144, 100000
        U = torch.randn(m, r, device=dev) * 0.1
        V = torch.randn(n, r, device=dev) * 0.1
        for _ in range(2):
            _ops.predict_topk(U, V, k, clamp_negatives=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _ops.predict_topk(U, V, k, clamp_negatives=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f'r={r} k={k}: {m / dt:.4g} rows/s, {2 * m * n * r / dt / 1e12:.1f} TF')
