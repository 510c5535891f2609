"""Times the LDS-tile prototype against tmf_wmrb_scores3 on the C4 shape (negatives only) and checks it."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teamoflow_amd import _engine, _lib
from teamoflow_amd.mf.utils import random_sampler_device

dev = torch.device('cuda', 0)
m, n, r, S = int(os.environ.get('M', 1_000_000)), 100_000, 128, 1024
lib = _lib.get()
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libproto.so'))
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn(m, r, device=dev, generator=g) * 0.1
V = torch.randn(n, r, device=dev, generator=g) * 0.1
R = random_sampler_device(n, m, S, seed=5, device=dev)
Rs_flat = torch.cat([torch.sort(R, dim=1)[0].flatten(), torch.zeros(64, dtype=torch.int32, device=dev)])   # padded: the kernels read a few ids ahead
Rs = Rs_flat[:m * S].view(m, S)
sp = torch.zeros(m, S, device=dev)
ptr = lambda t: ctypes.c_void_p(t.data_ptr())

def run_proto(T, threads, blocks, mode=0):
    rc = P.proto_scores_lds(ptr(U), ptr(V), ptr(Rs), ptr(sp), m, n, S, T, threads, blocks, mode, None)
    assert rc == 0, rc

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

# reference: existing sliced kernel, negatives only (no interactions)
ns = 13
idx = torch.zeros(0, 2, dtype=torch.int64, device=dev); val = torch.zeros(0, device=dev)
plan = _engine.InteractionPlan(idx, val, m, n, csc=False)
off = torch.empty(m, ns + 1, dtype=torch.int32, device=dev)
poff = torch.zeros(m, ns + 1, dtype=torch.int32, device=dev)
i32 = ctypes.c_int32
_lib.check(lib.tmf_slice_offsets(_lib.ptr(Rs), None, S, i32(m), i32(n), i32(ns), _lib.ptr(off), _lib.stream_ptr()), lib)
lists = _lib.SliceLists(Rs.data_ptr(), off.data_ptr(), plan.rowptr_u.data_ptr(), plan.col_u.data_ptr(), poff.data_ptr(), m, S, ns, 0, 0, 0, 0)
sp_ref = torch.zeros(m, S, device=dev); pk = torch.zeros(1, device=dev)
def run_ref():
    _lib.check(lib.tmf_wmrb_scores3_f32(ctypes.byref(lists), _lib.ptr(U), _lib.ptr(V), _lib.ptr(sp_ref), _lib.ptr(pk), r, _lib.stream_ptr()), lib)
print('scores3 (13 slices, negatives only): %.2f ms' % timeit(run_ref), flush=True)
for threads, T, blocks in [(512, 296, 256)]:
    sp.zero_()
    try:
        t = timeit(lambda: run_proto(T, threads, blocks))
    except AssertionError as e:
        print('proto', threads, T, blocks, 'failed', e); continue
    err = float((sp - sp_ref).abs().max()); ref = float(sp_ref.abs().max())
    print('proto threads=%d T=%d blocks=%d: %.2f ms   max|diff| %.3g (max|ref| %.3g)' % (threads, T, blocks, t, err, ref), flush=True)

def run_proto2(T, threads, blocks, mode=0):
    rc = P.proto_scores_lds2(ptr(U), ptr(V), ptr(Rs), ptr(sp), m, n, S, T, threads, blocks, mode, None)
    assert rc == 0, rc
for threads, T, blocks in [(512, 296, 256)]:
    sp.zero_()
    t = timeit(lambda: run_proto2(T, threads, blocks))
    err = float((sp - sp_ref).abs().max())
    print('proto2 threads=%d T=%d blocks=%d: %.2f ms   max|diff| %.3g' % (threads, T, blocks, t, err), flush=True)
for mode in (1, 2, 3):
    print('proto2 mode', mode, '(1 = no stores, 2 = synthetic ids) %.2f ms' % timeit(lambda: run_proto2(296, 512, 256, mode)), flush=True)
for mode, name in ((1, 'no tile loads'), (2, 'no compute'), (3, 'no per-entry store'), (4, 'no store, synthetic ids (no id loads)')):
    print(name, '%.2f ms' % timeit(lambda: run_proto(296, 512, 256, mode)), flush=True)
