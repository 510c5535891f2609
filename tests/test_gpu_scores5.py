"""tmf_wmrb_scores5 - the row-stationary form of the sliced pass's scores kernel (workgroups own 256 users, rows in LDS, one flat
stream of (user, item) pairs ordered by item; csrc/tmf_wmrb.hip, _engine.Scores5Plan) - against tmf_wmrb_scores3 and the
oracle.  Both compute sp[u, s] = <U[u], V[R[u, s]]> and p[k] = <U[u_k], V[j_k]> (matrix_factorization.py:153-154,
utils.py:94-105); they differ in the order of the fp32 sum inside a dot product only: equal to rounding on real data, bit for bit
on dyadic data - and then the whole epoch (hinge, gradients, fresh Adam) is bit-identical too."""
import numpy as np
import pytest
import torch

from conftest import assert_step, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from teamoflow_amd import _engine, _lib
    _lib.get()
    return _engine


def problem(m, n, r, S, nnz, seed, dyadic=False, dev='cuda'):
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, m, (nnz,), generator=g)
    u[u % 7 == 3] = (u[u % 7 == 3] + 1) % m                       # some users without interactions, some heavy
    j = torch.randint(0, n, (nnz,), generator=g)
    key = torch.unique(u * n + j)
    idx = torch.stack([key // n, key % n], 1)
    val = torch.randint(-1, 6, (key.numel(),), generator=g).float()   # a few non-positive stored values
    R = torch.stack([torch.randperm(n, generator=g)[:S] for _ in range(m)]).to(torch.int32)
    if dyadic:   # multiples of 1/8 in [-1, 1]: every product and every partial sum is exact in fp32 (and in bf16 storage)
        U = torch.randint(-8, 9, (m, r), generator=g).float() / 8
        V = torch.randint(-8, 9, (n, r), generator=g).float() / 8
    else:
        U = torch.randn(m, r, generator=g) * 0.3
        V = torch.randn(n, r, generator=g) * 0.3
    return idx.to(dev), val.to(dev), R.to(dev), U.to(dev), V.to(dev)


def epoch(eng, monkeypatch, s5, idx, val, R, U, V, m, n, r, S, dtype, slices):
    monkeypatch.setenv('TMF_SCORES5', '1' if s5 else '0')
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.WmrbPlan(plan, R, item_slices=slices, n_components=r, sliced=True)
    st = eng.TrainState(U, V, plan, r, wplan, dtype=dtype)
    assert (wplan.s5 is not None) == s5
    loss = torch.zeros(1, dtype=torch.float64, device=idx.device)
    eng.epoch_wmrb(st, eng.adam_constants(0.05), n / S, loss)
    torch.cuda.synchronize()
    return dict(sp=st.sp.clone(), pk=st.pk.clone(), U=st.U_nxt.float().clone(), V=st.V_nxt.float().clone(), loss=float(loss),
                D=wplan.D.clone(), delta=wplan.delta.clone(), plan=plan, wplan=wplan, st=st)


CASES = [  # m, n, r, S, nnz, dtype, slices
    (1000, 3000, 128, 64, 20000, torch.float32, 3),
    (257, 500, 100, 33, 3000, torch.float32, 2),          # one user beyond a workgroup of 256, ragged width
    (255, 70000, 128, 128, 9000, torch.float32, 7),
    (5000, 20000, 256, 96, 100000, torch.bfloat16, 5),
    (300, 999, 200, 17, 2500, torch.bfloat16, 1),
    (2048, 4096, 65, 40, 1, torch.float32, 2),            # a single interaction
]


@pytest.mark.parametrize('m,n,r,S,nnz,dtype,slices', CASES)
def test_scores5_equals_scores3_to_rounding(eng, monkeypatch, m, n, r, S, nnz, dtype, slices):
    idx, val, R, U, V = problem(m, n, r, S, nnz, seed=m + n)
    a = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, slices)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices)
    scale = float(a['sp'].abs().max())
    assert float((a['sp'] - b['sp']).abs().max()) <= 2e-6 * scale
    assert float((a['pk'] - b['pk']).abs().max()) <= 2e-6 * scale
    assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(a['loss'])
    # against an fp64 product of the stored rows (the tables as the kernels read them)
    st = b['st']
    U64, V64 = st.U[:, :r].double(), st.V[:, :r].double()
    Rs = b['wplan'].R.long()
    want = torch.einsum('ur,usr->us', U64, V64[Rs])
    assert float((b['sp'].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    pl = b['plan']
    wantp = (U64[pl.user_of.long()] * V64[pl.col_u.long()]).sum(1)
    assert float((b['pk'].double() - wantp).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize('m,n,r,S,nnz,dtype,slices', [CASES[0], CASES[1], CASES[3]])
def test_scores5_epoch_is_bit_identical_on_dyadic_tables(eng, monkeypatch, m, n, r, S, nnz, dtype, slices):
    """Dyadic factors: every dot product is exact whatever its summation tree, so the two scores kernels must agree bit for bit -
    and with them everything downstream: D, delta, loss, both updated tables."""
    idx, val, R, U, V = problem(m, n, r, S, nnz, seed=5 * m + n, dyadic=True)
    a = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, slices)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices)
    for k in ('sp', 'pk', 'D', 'delta', 'U', 'V'):
        assert torch.equal(a[k], b[k]), k
    assert a['loss'] == b['loss']
    c = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices)   # and a rerun reproduces itself
    assert torch.equal(b['sp'], c['sp']) and torch.equal(b['U'], c['U'])


def test_scores5_one_step_against_the_fp64_closed_form(eng, monkeypatch):
    """One epoch through the row-stationary scores kernel against oracle.sparse_ref (fp64 closed forms): mean loss to 1e-5, both
    tables inside the step interval (conftest.step_bounds) with the hinge-boundary slack of the oracle."""
    from oracle import sparse_ref as SR
    m, n, r, S, lr = 600, 900, 128, 48, 0.05
    idx, val, R, U, V = problem(m, n, r, S, 12000, seed=77)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, torch.float32, 2)
    U64, V64 = U.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.float64)
    i_np, v_np, R_np = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.int64)
    _, _, mean, t = SR.wmrb_epoch(U64, V64, i_np, v_np, R_np, n, S, lr)
    sl = SR.wmrb_slack(U64, V64, i_np, v_np, R_np, n, S)
    n_pos = int((val > 0).sum())
    assert abs(b['loss'] / n_pos - mean) <= 1e-5 * abs(mean)
    assert_step(b['U'][:, :r].cpu().numpy(), U64, t['gU'], lr, what='scores5 U', slack=sl['gU'])
    assert_step(b['V'][:, :r].cpu().numpy(), V64, t['gV'], lr, what='scores5 V', slack=sl['gV'])


def test_scores5_is_opt_in_only(eng, monkeypatch):
    """Measured slower than scores3 on the shape it was built for (profiles/r04_scores5_experiment.txt): never chosen unless
    TMF_SCORES5=1, and then only for the geometries the kernel has."""
    class P:
        pass
    for n_items, r, dtype, has_kernel in ((1_000_000, 256, torch.bfloat16, True), (100_000, 128, torch.float32, True),
                                          (1_000_000, 64, torch.float32, False), (20_000_000, 128, torch.float32, False)):
        plan, w = P(), P()
        plan.n_items, plan.nnz, plan.n_users, plan.col_u = n_items, 100 * 1000, 1000, torch.zeros(1, device='cuda')
        w.sliced, w.S, w.R = True, 1024, torch.zeros(1000, 1024)
        monkeypatch.delenv('TMF_SCORES5', raising=False)
        assert not eng.scores5_wanted(plan, w, r, dtype)
        monkeypatch.setenv('TMF_SCORES5', '1')
        assert eng.scores5_wanted(plan, w, r, dtype) == has_kernel, (n_items, r)


@pytest.mark.parametrize('lag', [0, 2])
def test_scores5_pacing_changes_nothing(eng, monkeypatch, lag):
    """The pacer (16th wave, per-XCD progress counters, LDS gate) is speed only: tiny windows and lag 0 make the workgroups
    really wait for each other (20 workgroups, up to 3 per XCD lane) - the scores must be those of the free-running kernel,
    bit for bit, and so must a second paced run."""
    m, n, r, S = 5000, 20000, 128, 64
    idx, val, R, U, V = problem(m, n, r, S, 60000, seed=11)
    monkeypatch.setenv('TMF_S5_PACE', '0')
    free = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, torch.float32, 3)
    assert not free['wplan'].s5.paced
    monkeypatch.setenv('TMF_S5_PACE', '1')
    monkeypatch.setenv('TMF_S5_SLICE_BYTES', str(64 * 512))    # slices of 64 rows: 313 of them
    monkeypatch.setenv('TMF_S5_PACE_EVERY', '2')               # a rendezvous every second slice
    monkeypatch.setenv('TMF_S5_LAG', str(lag))
    a = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, torch.float32, 3)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, torch.float32, 3)
    s5 = a['wplan'].s5
    assert s5.paced and s5.n_slices > 300 and s5.n_windows == (s5.n_slices + 1) // 2 + 1 and s5.lag == lag
    ws = s5.wstart.cpu().numpy()
    assert (np.diff(ws, axis=1) >= 0).all() and (ws[:, :2] == 0).all()      # window 0 is the (empty) start line
    assert np.array_equal(ws[:, -1], np.diff(s5.wg_ptr.cpu().numpy()) // 8)
    for k in ('sp', 'pk', 'U', 'V'):
        assert torch.equal(a[k], free[k]) and torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('dtype,r', [(torch.float32, 128), (torch.float32, 70), (torch.bfloat16, 256)])
def test_lean_scores_walk_equals_the_general_form(eng, monkeypatch, dtype, r):
    """tmf_wmrb_scores3 walks a (user, slice) range in a leaner form when tmf_slice_lists.n_items says V is addressable with 32-bit
    offsets (rows of 32 lanes): one instruction per row address, packed dot products, four scores reduced together.  Against the
    general form (TMF_LEAN=0): bit for bit on dyadic tables - and then the whole epoch is - and to rounding on Gaussian ones."""
    m, n, S, nnz = 3000, 9000, 100, 60000
    for dyadic in (True, False):
        idx, val, R, U, V = problem(m, n, r, S, nnz, seed=31 + r, dyadic=dyadic)
        monkeypatch.setenv('TMF_LEAN', '0')
        a = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, 5)
        monkeypatch.setenv('TMF_LEAN', '1')
        b = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, 5)
        if dyadic:
            for k in ('sp', 'pk', 'D', 'delta', 'U', 'V'):
                assert torch.equal(a[k], b[k]), (k, r)
            assert a['loss'] == b['loss']
        else:
            scale = float(a['sp'].abs().max())
            assert float((a['sp'] - b['sp']).abs().max()) <= 2e-6 * scale and float((a['pk'] - b['pk']).abs().max()) <= 2e-6 * scale
            assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(a['loss'])
            assert not torch.equal(a['sp'], b['sp'])    # another summation tree: the lean form really ran
