"""Prints what the HIP runtime says about the resident workgroups per CU of the fused predict kernel
(hipOccupancyMaxActiveBlocksPerMultiprocessor through a tiny hipcc-built probe would need a binary; instead the
timing of 1 vs 2 workgroups' worth of LDS tells the same).  usage: python tools/occupancy_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
dev = 'cuda'
n, r = 100000, 128
V = torch.randn(n, r, device=dev) * 0.1
for wgs_per_cu in (1, 2, 3, 4, 8):
    m = 256 * wgs_per_cu * 128          # exactly wgs_per_cu workgroups per CU
    U = torch.randn(m, r, device=dev) * 0.1
    _ops.predict_topk(U, V, 10, clamp_negatives=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        _ops.predict_topk(U, V, 10, clamp_negatives=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'{wgs_per_cu} WG/CU: {dt * 1e3:.2f} ms  {2 * m * n * r / dt / 1e12:.1f} TF')
