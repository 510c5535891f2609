"""Randomised selection of the WMRB epoch's kernel FORMS on random problems: scores3 / scores6 (flat streams), gradu3 + finish /
gradu4 (row-stationary), item pass as lists + slab / rows4 / rows5 (virtual rows), any slice and user-block counts, fp32 and bf16
storage.  Every form computes the same epoch (matrix_factorization.py:150-183 with loss_graphs.py:74-88): the raw gradients of
the sum of the losses with respect to U and V against the fp64 closed form of the oracle on the tables as stored, the mean loss,
and - fp32 - the updated tables inside the fresh-Adam step interval.  TMF_FUZZ_SEEDS=300 for a soak run."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_step

pytestmark = pytest.mark.gpu

FORM_KEYS = ('TMF_ROWS4', 'TMF_ROWS5', 'TMF_ROWS5_TARGET', 'TMF_SCORES5', 'TMF_SCORES6', 'TMF_S6_SLICE_BYTES', 'TMF_ROW_STATIONARY',
             'TMF_ITEM_SLICES', 'TMF_USER_CHUNKS', 'TMF_FORCE_SLICED', 'TMF_SLICE_XCD', 'TMF_G4_USERS', 'TMF_ROWS4_PER_LAUNCH')


@pytest.fixture(scope='module')
def eng():
    from teamoflow_amd import _engine, _lib
    _lib.get()
    return _engine


def draw(seed):
    rng = np.random.default_rng(77000 + seed)
    bf16 = bool(seed % 3 == 2)
    r = int(rng.choice([128, 256] if bf16 and seed % 2 else [16, 24, 32, 64, 100, 128, 160, 256]))
    m, n = int(rng.integers(1, 700)), int(rng.integers(3, 900))
    S = int(rng.integers(1, min(n, 48) + 1))
    n = max(3, min(n, 32 * S))   # c = n / S multiplies every rounding of a score (see the tolerance below); the reference's own
                                 # configurations have c = 2 ... 98
    S = min(S, n)
    nnz = int(rng.integers(1, 6 * m + 2))
    u, j = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
    if seed % 4 == 1:                                            # a popular item (a long list) and a heavy user
        u = np.concatenate([u, np.arange(m), np.zeros(min(n, 300), np.int64)])
        j = np.concatenate([j, np.full(m, int(rng.integers(0, n))), np.arange(min(n, 300))])
    key = np.unique(u.astype(np.int64) * n + j)
    if seed % 5 == 0:
        key = rng.permutation(key)                               # the engine must not rely on row-major order
    idx = np.stack([key // n, key % n], 1)
    val = rng.integers(-1, 6, len(key)).astype(np.float32)
    if not (val > 0).any():
        val[0] = 1.0
    R = np.stack([rng.choice(n, S, replace=False) for _ in range(m)]).astype(np.int32)
    if seed % 4 == 3:
        R[::2, 0] = n - 1                                        # a popular NEGATIVE (may repeat an id within a row: allowed)
    U = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    env = {'TMF_FORCE_SLICED': '1', 'TMF_SCORES5': '0'}
    env['TMF_ITEM_SLICES'] = str(int(rng.integers(1, 10)))
    env['TMF_USER_CHUNKS'] = str(int(rng.integers(1, 7)))
    env['TMF_ROWS4'] = str(int(rng.integers(0, 2)))
    env['TMF_ROWS5'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env['TMF_ROWS5_TARGET'] = str(int(rng.choice([1, 2, 5, 20, 1000])))
    env['TMF_SCORES6'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env['TMF_S6_SLICE_BYTES'] = str(int(rng.choice([512, 4096, 50000, 1 << 20])))
    env['TMF_ROW_STATIONARY'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 3) == 0:
        env['TMF_SLICE_XCD'] = str(int(rng.integers(0, 2)))
    return m, n, r, S, idx, val, R, U, V, torch.bfloat16 if bf16 else torch.float32, env


def self_pair_floor(c, U64, V64):
    """A positive that is also one of the user's sampled negatives: the reference adds both contributions in one cell of the dense
    d loss / d predictions (w - w = 0 when that pair is the positive's only active term), the kernels add D[u, s] V[j] and
    delta_k V[j] as two FMAs of one sum, which leaves the rounding of one product: <= 2^-24 c |V[j]| per element.
    Visible only where the rest of the gradient is exactly zero."""
    return 1e-7 * c * np.abs(V64).max(), 1e-7 * c * np.abs(U64).max()


@pytest.mark.parametrize('seed', range(int(os.environ.get('TMF_FUZZ_SEEDS', '24'))))
def test_random_forms_against_the_closed_form(eng, monkeypatch, seed):
    from oracle import sparse_ref as SR
    from teamoflow_amd import _lib
    m, n, r, S, idx, val, R, U, V, dtype, env = draw(seed)
    for k in FORM_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = 'cuda'
    plan = eng.InteractionPlan(torch.tensor(idx, device=dev), torch.tensor(val, device=dev), m, n)
    wplan = eng.wmrb_plan_for(plan, torch.tensor(R, device=dev), r, dtype)
    st = eng.TrainState(torch.tensor(U, device=dev), torch.tensor(V, device=dev), plan, r, wplan, dtype=dtype)
    what = dict(env, m=m, n=n, r=r, S=S, nnz=len(val), dtype=str(dtype), rows4=wplan.rows4, vrows=wplan.vrows is not None,
                s6=wplan.s6 is not None, rs=st.row_stationary, slices=wplan.n_slices)
    assert wplan.sliced and wplan.s5 is None, what
    lr, adam = 0.05, eng.adam_constants(0.05)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    gU = torch.full((m, st.ld), 7.0, device=dev)
    gV = torch.full((n, st.ld), 7.0, device=dev)
    eng.epoch_wmrb(st, adam, n / S, loss, item_epi=_lib.EPI_GRAD, item_out=gV, user_epi=_lib.EPI_GRAD, user_out=gU)
    loss2 = torch.zeros(1, dtype=torch.float64, device=dev)
    eng.epoch_wmrb(st, adam, n / S, loss2)
    torch.cuda.synchronize()
    U64, V64 = st.U[:, :r].double().cpu().numpy(), st.V[:, :r].double().cpu().numpy()   # the tables as stored (bf16: rounded)
    v64, R64 = val.astype(np.float64), R.astype(np.int64)
    _, _, mean, t = SR.wmrb_epoch(U64, V64, idx, v64, R64, n, S, lr)
    sl = SR.wmrb_slack(U64, V64, idx, v64, R64, n, S)
    n_pos = int((val > 0).sum())
    # A term's weight is c / (1 + c * sum of hinges): where the sum is small, an fp32 rounding ds of one score moves the weight by
    # c * ds relative - the conditioning of the loss, the same for every form (tools/fuzz_forms_diag.py: seeds 75 / 14 of the first
    # draws, c = 374 / 32 with a hinge argument of 0.002 / 0.003, all forms 1.8e-5 / 1.5e-5 off together).  ds ~ 5e-7 at these widths.
    rtol = 1e-5 + 5e-7 * (n / S)
    assert float(loss) == float(loss2), what                                             # the two epochs read the same tables
    assert abs(float(loss) / n_pos - mean) <= 1e-5 * abs(mean) + 1e-12, (float(loss) / n_pos, mean, what)
    fU, fV = self_pair_floor(n / S, U64, V64)
    for name, g, ref, s in (('gU', gU, t['gU'], sl['gU'] + fU), ('gV', gV, t['gV'], sl['gV'] + fV)):
        d = np.abs(g[:, :r].double().cpu().numpy() - ref) - 1.0001 * s
        assert d.max() <= rtol * max(np.abs(ref).max(), 1e-30), (name, float(d.max()), float(np.abs(ref).max()), what)
    if dtype is torch.float32:
        assert_step(st.U_nxt[:, :r].cpu().numpy(), U64, t['gU'], lr, rtol=rtol, what=f'U {what}', slack=sl['gU'] + fU)
        assert_step(st.V_nxt[:, :r].cpu().numpy(), V64, t['gV'], lr, rtol=rtol, what=f'V {what}', slack=sl['gV'] + fV)
    # the pad columns of the raw gradients stay what the kernels define them to be: never NaN
    assert torch.isfinite(gU).all() and torch.isfinite(gV).all(), what


def draw_medium(seed):
    """Thousands of users and items: several launches of the row-stationary kernels (TMF_G4_USERS, TMF_ROWS4_PER_LAUNCH), hundreds
    of workgroups per launch, chunks of every length, cut rows with many parts - on DYADIC tables (multiples of 1/8), where every
    score is exact in fp32 whatever the order of its sum."""
    rng = np.random.default_rng(99000 + seed)
    bf16 = bool(seed % 2)
    r = int(rng.choice([128, 256]))
    m, n = int(rng.integers(3000, 50000)), int(rng.integers(2000, 40000))
    S = int(rng.integers(8, 129))
    nnz = int(m * rng.integers(2, 12))
    g = torch.Generator().manual_seed(99000 + seed)
    u = torch.randint(0, m, (nnz,), generator=g)
    j = (torch.rand(nnz, generator=g) ** 3 * n).long().clamp_(0, n - 1)          # popular items: lists of very different length
    key = torch.unique(u * n + j)
    idx = torch.stack([key // n, key % n], 1)
    val = torch.randint(-1, 6, (key.numel(),), generator=g).float()
    R = torch.stack([torch.randperm(n, generator=g)[:S] for _ in range(min(m, 512))]).to(torch.int32)
    R = (R[torch.arange(m) % R.shape[0]] + torch.randint(0, n, (m, 1), generator=g).to(torch.int32)) % n   # distinct ids per row, cheap to draw
    U = torch.randint(-8, 9, (m, r), generator=g).float() / 8
    V = torch.randint(-8, 9, (n, r), generator=g).float() / 8
    env = {'TMF_FORCE_SLICED': '1', 'TMF_SCORES5': '0'}
    env['TMF_ITEM_SLICES'] = str(int(rng.integers(1, 24)))
    env['TMF_USER_CHUNKS'] = str(int(rng.integers(1, 12)))
    env['TMF_ROWS4'] = str(int(rng.integers(0, 2)))
    env['TMF_ROWS5'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env['TMF_ROWS5_TARGET'] = str(int(rng.choice([8, 40, 200, 5000])))
    env['TMF_SCORES6'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env['TMF_S6_SLICE_BYTES'] = str(int(rng.choice([20000, 200000, 1 << 20])))
    env['TMF_ROW_STATIONARY'] = str(int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env['TMF_G4_USERS'] = str(int(rng.choice([2048, 8192])))
    if rng.integers(0, 2):
        env['TMF_ROWS4_PER_LAUNCH'] = str(int(rng.choice([64, 300])))
    return m, n, r, S, idx, val, R, U, V, torch.bfloat16 if bf16 else torch.float32, env


PLAIN = {'TMF_FORCE_SLICED': '1', 'TMF_SCORES5': '0', 'TMF_SCORES6': '0', 'TMF_ROWS4': '0', 'TMF_ROW_STATIONARY': '0'}


def one_epoch(eng, monkeypatch, env, m, n, r, S, idx, val, R, U, V, dtype):
    from teamoflow_amd import _lib
    for k in FORM_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.wmrb_plan_for(plan, R, r, dtype)
    st = eng.TrainState(U, V, plan, r, wplan, dtype=dtype)
    adam = eng.adam_constants(0.05)
    loss = torch.zeros(1, dtype=torch.float64, device='cuda')
    gU = torch.full((m, st.ld), 7.0, device='cuda')
    gV = torch.full((n, st.ld), 7.0, device='cuda')
    eng.epoch_wmrb(st, adam, n / S, loss, item_epi=_lib.EPI_GRAD, item_out=gV, user_epi=_lib.EPI_GRAD, user_out=gU)
    eng.epoch_wmrb(st, adam, n / S, loss.zero_())
    torch.cuda.synchronize()
    forms = dict(rows4=wplan.rows4, vrows=wplan.vrows is not None, s6=wplan.s6 is not None, rs=st.row_stationary,
                 slices=wplan.n_slices, chunks=wplan.user_chunks)
    return dict(D=wplan.D_in_model_order().clone(), delta=wplan.delta.clone(), gU=gU[:, :r].clone(), gV=gV[:, :r].clone(),
                U=st.U_nxt[:, :r].float().clone(), V=st.V_nxt[:, :r].float().clone(), loss=float(loss), forms=forms)


@pytest.mark.parametrize('seed', range(max(6, int(os.environ.get('TMF_FUZZ_SEEDS', '24')) // 4)))
def test_random_forms_at_medium_size_against_the_plain_forms(eng, monkeypatch, seed):
    m, n, r, S, idx, val, R, U, V, dtype, env = draw_medium(seed)
    args = (m, n, r, S, idx.cuda(), val.cuda(), R.cuda(), U.cuda(), V.cuda(), dtype)
    a = one_epoch(eng, monkeypatch, PLAIN, *args)
    b = one_epoch(eng, monkeypatch, env, *args)
    what = dict(env, m=m, n=n, r=r, S=S, nnz=len(val), dtype=str(dtype), **b['forms'])
    assert not (a['forms']['rows4'] or a['forms']['s6'] or a['forms']['rs']), a['forms']
    # exact scores -> the same hinge weights, bit for bit, whatever computed the scores and in whatever slices
    assert torch.equal(a['D'], b['D']) and torch.equal(a['delta'], b['delta']) and a['loss'] == b['loss'], what
    for k in ('gU', 'gV'):     # the same fp32 products added in another order (lists of thousands of terms: seed 19 differs by 4.3e-6)
        scale = float(a[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 1e-5 * scale, (k, float((a[k] - b[k]).abs().max()), scale, what)
    c = one_epoch(eng, monkeypatch, env, *args)     # and every form is reproducible bit for bit
    for k in ('gU', 'gV', 'U', 'V', 'D'):
        assert torch.equal(b[k], c[k]), (k, what)


@pytest.mark.parametrize('forms', [dict(), dict(TMF_SCORES6='1'), dict(TMF_ROW_STATIONARY='1'), dict(TMF_ROWS4='1'),
                                   dict(TMF_ROWS4='1', TMF_ROWS5='0'), dict(TMF_SCORES6='1', TMF_ROW_STATIONARY='1', TMF_ROWS4='1')])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_no_interactions_at_all(eng, monkeypatch, forms, dtype):
    """An interaction table without a single entry (found by the round-5 soak on the windowed path: the per-interaction arrays are
    then NULL): every loss term is gone, both gradients are zero and the fresh-Adam step leaves the tables as they are
    (matrix_factorization.py:160-183: the mean of no losses is NaN, which the caller reports; the sum here is 0)."""
    from teamoflow_amd import _lib
    m, n, r, S = 300, 500, 128, 20
    g = torch.Generator().manual_seed(1)
    idx, val = torch.zeros(0, 2, dtype=torch.int64, device='cuda'), torch.zeros(0, device='cuda')
    R = torch.stack([torch.randperm(n, generator=g)[:S] for _ in range(m)]).to(torch.int32).cuda()
    U, V = (torch.randn(m, r, generator=g) * 0.3).cuda(), (torch.randn(n, r, generator=g) * 0.3).cuda()
    for k in FORM_KEYS:
        monkeypatch.delenv(k, raising=False)
    env = dict(PLAIN, TMF_ITEM_SLICES='3', TMF_USER_CHUNKS='2')
    env.update(forms)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.wmrb_plan_for(plan, R, r, dtype)
    st = eng.TrainState(U, V, plan, r, wplan, dtype=dtype)
    adam = eng.adam_constants(0.05)
    loss = torch.ones(1, dtype=torch.float64, device='cuda')
    gU = torch.full((m, st.ld), 7.0, device='cuda')
    gV = torch.full((n, st.ld), 7.0, device='cuda')
    eng.epoch_wmrb(st, adam, n / S, loss, item_epi=_lib.EPI_GRAD, item_out=gV, user_epi=_lib.EPI_GRAD, user_out=gU)
    eng.epoch_wmrb(st, adam, n / S, loss)
    torch.cuda.synchronize()
    assert float(loss) == 0.0 and not gU[:, :r].any() and not gV[:, :r].any()
    assert torch.equal(st.U_nxt[:, :r], st.U[:, :r]) and torch.equal(st.V_nxt[:, :r], st.V[:, :r])
    assert not wplan.D.any()


@pytest.mark.parametrize('seed', range(max(6, int(os.environ.get('TMF_FUZZ_SEEDS', '24')) // 4)))
def test_random_forms_over_several_epochs_teacher_forced(eng, monkeypatch, seed):
    """Three epochs in a row on one TrainState (the buffers swap, the rendezvous counters and the partial-row layers are used
    again): after every epoch the tables must lie in the fresh-Adam step interval of the fp64 closed form evaluated at the
    tables the epoch STARTED from - nothing stale from the epoch before."""
    from oracle import sparse_ref as SR
    m, n, r, S, idx, val, R, U, V, dtype, env = draw(5000 + seed)
    if dtype is not torch.float32:
        dtype = torch.float32                      # the interval test is an fp32 statement; bf16 storage is covered above
    for k in FORM_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = 'cuda'
    plan = eng.InteractionPlan(torch.tensor(idx, device=dev), torch.tensor(val, device=dev), m, n)
    wplan = eng.wmrb_plan_for(plan, torch.tensor(R, device=dev), r, dtype)
    st = eng.TrainState(torch.tensor(U, device=dev), torch.tensor(V, device=dev), plan, r, wplan, dtype=dtype)
    what = dict(env, m=m, n=n, r=r, S=S, nnz=len(val), rows4=wplan.rows4, vrows=wplan.vrows is not None, s6=wplan.s6 is not None,
                rs=st.row_stationary)
    lr, adam = 0.05, eng.adam_constants(0.05)
    v64, R64 = val.astype(np.float64), R.astype(np.int64)
    rtol = 1e-5 + 5e-7 * (n / S)
    n_pos = int((val > 0).sum())
    loss = torch.zeros(3, dtype=torch.float64, device=dev)
    for e in range(3):
        U64, V64 = st.U[:, :r].double().cpu().numpy(), st.V[:, :r].double().cpu().numpy()
        eng.epoch_wmrb(st, adam, n / S, loss[e:e + 1])
        st.swap()
        torch.cuda.synchronize()
        _, _, mean, t = SR.wmrb_epoch(U64, V64, idx, v64, R64, n, S, lr)
        sl = SR.wmrb_slack(U64, V64, idx, v64, R64, n, S)
        fU, fV = self_pair_floor(n / S, U64, V64)
        assert abs(float(loss[e]) / n_pos - mean) <= rtol * abs(mean) + 1e-12, (e, what)
        assert_step(st.U[:, :r].cpu().numpy(), U64, t['gU'], lr, rtol=rtol, what=f'epoch {e} U {what}', slack=sl['gU'] + fU)
        assert_step(st.V[:, :r].cpu().numpy(), V64, t['gV'], lr, rtol=rtol, what=f'epoch {e} V {what}', slack=sl['gV'] + fV)


@pytest.mark.parametrize('seed', range(max(4, int(os.environ.get('TMF_FUZZ_SEEDS', '24')) // 6)))
def test_captured_epochs_equal_launched_epochs_whatever_the_forms(monkeypatch, seed):
    """fit() captures the epochs of small problems into a hipGraph (matrix_factorization.py here: GRAPH_EPOCHS): replaying it
    must give the bits that launching the same kernels one by one gives - with every form of every pass, including the ones that
    keep counters in a workspace between launches."""
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import WMRBLoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    m, n, r, S, idx, val, R, U, V, dtype, env = draw(9000 + seed)
    for k in FORM_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)

    def fit(no_graph):
        if no_graph:
            monkeypatch.setenv('TMF_NO_GRAPH', '1')
        else:
            monkeypatch.delenv('TMF_NO_GRAPH', raising=False)
        model = MatrixFactorization(r, loss_graph=WMRBLoss(), user_weight_graph=FixedInitializer(U), item_weight_graph=FixedInitializer(V),
                                    n_users=m, n_items=n, n_samples=S)
        model.random_ind = torch.as_tensor(R)
        model.verbose, model.factor_dtype = False, dtype
        model.fit(7, eye(m), eye(n), SparseInteractions(idx, val, (m, n)), lr=0.05)       # six captured epochs + one launched
        return model
    a, b = fit(False), fit(True)
    assert a.loss_history_ == b.loss_history_, (env, a.loss_history_, b.loss_history_)
    assert torch.equal(a.user_embedding, b.user_embedding) and torch.equal(a.item_embedding, b.item_embedding), env
