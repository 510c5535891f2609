"""Diagnosis of one seed of tests/test_gpu_fuzz_forms.py: python3 tools/fuzz_forms_diag.py SEED - the drawn forms and every single-form
variation of them, error of the raw gradients against the fp64 closed form."""
import os, sys
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
from test_gpu_fuzz_forms import draw, FORM_KEYS
from teamoflow_amd import _engine as eng, _lib
from oracle import sparse_ref as SR
seed = int(sys.argv[1])
m, n, r, S, idx, val, R, U, V, dtype, env = draw(seed)
print('seed', seed, dict(m=m, n=n, r=r, S=S, nnz=len(val), dtype=dtype), env, flush=True)

def run(env):
    for k in FORM_KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    plan = eng.InteractionPlan(torch.tensor(idx, device='cuda'), torch.tensor(val, device='cuda'), m, n)
    wplan = eng.wmrb_plan_for(plan, torch.tensor(R, device='cuda'), r, dtype)
    st = eng.TrainState(torch.tensor(U, device='cuda'), torch.tensor(V, device='cuda'), plan, r, wplan, dtype=dtype)
    loss = torch.zeros(1, dtype=torch.float64, device='cuda')
    gU = torch.full((m, st.ld), 7.0, device='cuda'); gV = torch.full((n, st.ld), 7.0, device='cuda')
    eng.epoch_wmrb(st, eng.adam_constants(0.05), n / S, loss, item_epi=_lib.EPI_GRAD, item_out=gV, user_epi=_lib.EPI_GRAD, user_out=gU)
    torch.cuda.synchronize()
    return st, gU[:, :r].double().cpu().numpy(), gV[:, :r].double().cpu().numpy(), float(loss)

st, gU, gV, loss = run(env)
U64, V64 = st.U[:, :r].double().cpu().numpy(), st.V[:, :r].double().cpu().numpy()
_, _, mean, t = SR.wmrb_epoch(U64, V64, idx, val.astype(np.float64), R.astype(np.int64), n, S, 0.05)
sl = SR.wmrb_slack(U64, V64, idx, val.astype(np.float64), R.astype(np.int64), n, S)
def report(tag, gU, gV):
    for name, g, ref, s in (('gU', gU, t['gU'], sl['gU']), ('gV', gV, t['gV'], sl['gV'])):
        d = np.abs(g - ref) - 1.0001 * s
        i = np.unravel_index(np.argmax(d), d.shape)
        print(f'{tag:40s} {name}: max excess {d.max():.3e} at {i}: got {g[i]!r} ref {ref[i]!r} slack {s[i]:.2e}; row max |ref| {np.abs(ref[i[0]]).max():.3e}, '
              f'global max {np.abs(ref).max():.3e}', flush=True)
report('as drawn', gU, gV)
for k in ('TMF_ROWS4', 'TMF_SCORES6', 'TMF_ROW_STATIONARY'):
    e = dict(env); e[k] = '0' if env.get(k) == '1' else '1'
    _, a, b, _ = run(e)
    report(f'{k}={e[k]}', a, b)
e = dict(env); e['TMF_ITEM_SLICES'] = '1'
_, a, b, _ = run(e); report('one slice', a, b)
i = np.unravel_index(np.argmax(np.abs(gU - t['gU']) - 1.0001 * sl['gU']), gU.shape)
u = i[0]
pos = idx[(idx[:, 0] == u) & (val > 0)]
print('user', u, 'positives', len(pos), 'interactions', int((idx[:, 0] == u).sum()), 'negatives', R[u])
sc = V64[R[u]] @ U64[u]; sp = V64[pos[:, 1]] @ U64[u]
print('neg scores', sc, 'pos scores', sp[:10])
