"""Does the residue of the gathered row ids (mod 4 = which 4 of the 16 L2 channels a 512-byte row touches) matter for
tmf_wmrb_scores3?  C4 shape, negatives only."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teamoflow_amd import _engine, _lib

dev = torch.device('cuda', 0)
m, n, r, S, ns = 1_000_000, 100_000, 128, 1024, 13
lib = _lib.get()
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn(m, r, device=dev, generator=g) * 0.1
V = torch.randn(n, r, device=dev, generator=g) * 0.1
i32 = ctypes.c_int32
idx = torch.zeros(0, 2, dtype=torch.int64, device=dev); val = torch.zeros(0, device=dev)
plan = _engine.InteractionPlan(idx, val, m, n, csc=False)
sp = torch.zeros(m, S, device=dev); pk = torch.zeros(1, device=dev)

def sorted_multiples(step):
    """[m, S] ascending distinct multiples of `step` below n (approximately uniform)."""
    q = n // step
    out = torch.empty(m, S, dtype=torch.int32, device=dev)
    for b in range(0, m, 65536):
        rows = min(65536, m - b)
        keys = torch.rand(rows, q, device=dev, generator=g)
        out[b:b + rows] = (torch.sort(torch.topk(keys, S, dim=1)[1], dim=1)[0] * step).to(torch.int32)
    return out

def timeit(Rs, tag):
    off = torch.empty(m, ns + 1, dtype=torch.int32, device=dev)
    poff = torch.zeros(m, ns + 1, dtype=torch.int32, device=dev)
    _lib.check(lib.tmf_slice_offsets(_lib.ptr(Rs), None, S, i32(m), i32(n), i32(ns), _lib.ptr(off), _lib.stream_ptr()), lib)
    lists = _lib.SliceLists(Rs.data_ptr(), off.data_ptr(), plan.rowptr_u.data_ptr(), plan.col_u.data_ptr(), poff.data_ptr(), m, S, ns, 0, 0, 0, 0)
    run = lambda: _lib.check(lib.tmf_wmrb_scores3_f32(ctypes.byref(lists), _lib.ptr(U), _lib.ptr(V), _lib.ptr(sp), _lib.ptr(pk), r, _lib.stream_ptr()), lib)
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): run()
    b.record(); torch.cuda.synchronize()
    res = torch.bincount((Rs[:4096].flatten() % 4).to(torch.int64), minlength=4).tolist()
    print(f'{tag:55s} {a.elapsed_time(b) / 3:7.2f} ms   residues of the first rows {res}', flush=True)

R1 = sorted_multiples(1)
timeit(R1, 'uniform ids')
R4 = sorted_multiples(4)
timeit(R4, 'all ids = 0 mod 4 (one channel group)')
pos = torch.arange(S, device=dev, dtype=torch.int32)[None, :] % 4
timeit(R4 + pos, 'ids cycle through residues 0,1,2,3 along every list')
R16 = sorted_multiples(16)
timeit(R16, 'all ids = 0 mod 16')
timeit(R16 + (torch.arange(S, device=dev, dtype=torch.int32)[None, :] % 16), 'ids cycle through residues 0..15')
