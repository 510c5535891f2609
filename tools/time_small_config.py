"""Per-kernel timing (HIP events) of one WMRB or MSE epoch at a BASELINE small-config shape.
usage: python tools/time_small_config.py c3|c2"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _engine, _lib
from teamoflow_amd.mf.utils import random_sampler_device
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
if cfg == 'c3':
    m, n, r, S, nnz, loss = 6040, 3706, 64, 1853, 1_000_209, 'wmrb'
else:
    m, n, r, S, nnz, loss = 943, 1682, 32, 0, 100_000, 'mse'
key = torch.unique(torch.randint(0, m * n, (int(nnz * 1.03),), device=dev, generator=g))[:nnz]
idx = torch.stack([key // n, key % n], 1)
val = torch.randint(1, 6, (idx.shape[0],), device=dev, generator=g).float()
ld = _lib.padded_ld(r)
plan = _engine.InteractionPlan(idx, val, m, n)
wplan = None
if loss == 'wmrb':
    R = random_sampler_device(n, m, S, seed=1, device=dev)
    ns, sliced = _engine.choose_wmrb_user_pass(m, n, ld, S, plan.n_pos, r)
    wplan = _engine.WmrbPlan(plan, R, user_chunks=_engine.default_user_chunks(m, ld, n_items=n), item_slices=ns, n_components=r, sliced=sliced)
U0 = torch.rand(m, r, device=dev) * 0.01
V0 = torch.rand(n, r, device=dev) * 0.01
st = _engine.TrainState(U0, V0, plan, r, wplan)
adam = _engine.adam_constants(0.1)
out = torch.zeros(1, dtype=torch.float64, device=dev)
prof = _engine.KernelTimer()
for i in range(30):
    p = prof if i >= 10 else None
    if loss == 'wmrb':
        _engine.epoch_wmrb(st, adam, n / S, out, prof=p)
    else:
        _engine.epoch_mse(st, adam, out, prof=p)
    st.swap()
torch.cuda.synchronize()
for k in prof.spans:
    print(cfg, k, f'{prof.mean_ms(k) * 1e3:.1f} us')
