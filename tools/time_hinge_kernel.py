import sys, os, torch, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from teamoflow_amd import _engine, _lib
from teamoflow_amd.mf.utils import random_sampler_device
dev = torch.device('cuda', 0)
m, n, r, S = 1_000_000, 100_000, 128, 1024
idx, val = bench.gen_interactions(m, n, 100_000_000, 'zipf', 0, dev)
U0, V0 = bench.init_table(m, r, 11, dev), bench.init_table(n, r, 7, dev)
plan = _engine.InteractionPlan(idx, val, m, n)
R = random_sampler_device(n, m, S, seed=100, device=dev)
wplan = _engine.WmrbPlan(plan, R, user_chunks=123, item_slices=13)
st = _engine.TrainState(U0, V0, plan, r, wplan)
adam = _engine.adam_constants(0.1)
loss = torch.zeros(1, dtype=torch.float64, device=dev)
lib = _lib.get()
import ctypes
i32 = ctypes.c_int32
p, w = plan, wplan
for _ in range(2):
    _engine.epoch_wmrb(st, adam, n / S, loss)
torch.cuda.synchronize()
# the kernel with fewer samples per user than the tables hold (same buffers, smaller rows): what is left at S -> 0 is the
# per-positive work (row gathers of V[j_k], dot products, the gpos accumulation) and the per-user overhead
for S_eff in (S, 512, 256, 64, 16):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        _lib.check(lib.tmf_wmrb_hinge_f32(_lib.ptr(p.rowptr_u), _lib.ptr(p.col_u), _lib.ptr(p.val_u), _lib.ptr(st.sp), i32(m), i32(S_eff), n / S,
                                          _lib.ptr(st.U), _lib.ptr(st.V), _lib.ptr(st.gpos), _lib.ptr(w.delta), _lib.ptr(w.D), _lib.ptr(st.loss_part), r, _lib.stream_ptr()), lib)
    ev[1].record(); torch.cuda.synchronize()
    print(os.environ.get('TMF_LIB'), f'hinge S={S_eff}: {ev[0].elapsed_time(ev[1]) / 5:.2f} ms')
