"""Diagnosis of one seed of test_random_forms_over_several_epochs_teacher_forced: python3 tools/fuzz_epochs_diag.py SEED EPOCH"""
import os, sys
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
from test_gpu_fuzz_forms import draw, FORM_KEYS
from teamoflow_amd import _engine as eng, _lib
from oracle import sparse_ref as SR
seed, upto = int(sys.argv[1]), int(sys.argv[2])
m, n, r, S, idx, val, R, U, V, dtype, env = draw(5000 + seed)
dtype = torch.float32
for k in FORM_KEYS: os.environ.pop(k, None)
os.environ.update(env)
print(dict(m=m, n=n, r=r, S=S, nnz=len(val)), env)
plan = eng.InteractionPlan(torch.tensor(idx, device='cuda'), torch.tensor(val, device='cuda'), m, n)
wplan = eng.wmrb_plan_for(plan, torch.tensor(R, device='cuda'), r, dtype)
st = eng.TrainState(torch.tensor(U, device='cuda'), torch.tensor(V, device='cuda'), plan, r, wplan, dtype=dtype)
adam = eng.adam_constants(0.05)
loss = torch.zeros(1, dtype=torch.float64, device='cuda')
for e in range(upto):
    eng.epoch_wmrb(st, adam, n / S, loss.zero_()); st.swap()
gU = torch.full((m, st.ld), 7.0, device='cuda'); gV = torch.full((n, st.ld), 7.0, device='cuda')
U64, V64 = st.U[:, :r].double().cpu().numpy(), st.V[:, :r].double().cpu().numpy()
eng.epoch_wmrb(st, adam, n / S, loss.zero_(), item_epi=_lib.EPI_GRAD, item_out=gV, user_epi=_lib.EPI_GRAD, user_out=gU)
torch.cuda.synchronize()
_, _, mean, t = SR.wmrb_epoch(U64, V64, idx, val.astype(np.float64), R.astype(np.int64), n, S, 0.05)
g = gU[:, :r].double().cpu().numpy()
print('loss gpu', float(loss), 'mean ref', mean, 'max |g_ref|', np.abs(t['gU']).max(), 'max |g gpu|', np.abs(g).max())
bad = np.where((np.abs(g).max(1) > 0) & (np.abs(t['gU']).max(1) == 0))[0]
print('users with a gradient on the GPU and none in fp64:', bad)
D = wplan.D_in_model_order().cpu().numpy(); delta = wplan.delta.cpu().numpy()
sp = st.sp.view(m, S).cpu().numpy(); pk = st.pk.cpu().numpy()
uo = plan.user_of.cpu().numpy(); cu = plan.col_u.cpu().numpy(); vu = plan.val_u.cpu().numpy()
Rs = wplan.R.cpu().numpy()
for u in bad[:3]:
    ks = np.where(uo == u)[0]
    print('user', u, 'max|g|', np.abs(g[u]).max(), 'D row nonzero:', D[u][D[u] != 0], 'interactions', ks, 'items', cu[ks], 'val', vu[ks], 'delta', delta[ks], 'p', pk[ks])
    print('   sorted-neg scores (gpu) max', sp[u].max(), 'min', sp[u].min())
    for k in ks:
        if vu[k] > 0:
            arg = 1.0 - pk[k] + sp[u]
            a64 = 1.0 - V64[cu[k]] @ U64[u] + V64[Rs[u]] @ U64[u]
            print('   positive', k, 'hinge args fp32: max', arg.max(), ' fp64: max', a64.max(), 'n active fp32', int((arg > 0).sum()))
