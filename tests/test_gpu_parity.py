"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
committed golden fixtures.  Tolerances: indices / top-k bit-exact; loss, gradients and predictions
1e-5 relative (norm-wise, fp32); factor tables after a step per conftest.step_bounds (the interval the
reference's update spans for a gradient within 1e-5 of the oracle's fp64 closed form)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_step, rel_err

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)


@pytest.fixture(scope='module')
def tm():
    from teamoflow_amd import _engine, _lib, _ops
    from teamoflow_amd.mf import initializer_graphs, loss_graphs, matrix_factorization, sparse, utils
    _lib.get()

    class NS:
        pass
    ns = NS()
    ns.engine, ns.lib, ns.ops = _engine, _lib, _ops
    ns.MF = matrix_factorization.MatrixFactorization
    ns.Fixed = initializer_graphs.FixedInitializer
    ns.WMRB = loss_graphs.WMRBLoss
    ns.Sparse = sparse.SparseInteractions
    ns.eye = sparse.eye
    ns.utils = utils
    return ns


def fit_model(tm, U0, V0, idx, val, shape, epochs, lr, loss='mse', R=None, n_items=None, n_samples=None):
    m, n = shape
    kw = dict(user_weight_graph=tm.Fixed(U0), item_weight_graph=tm.Fixed(V0))
    if loss == 'wmrb':
        kw.update(loss_graph=tm.WMRB(), n_users=m, n_items=n_items, n_samples=n_samples)
    model = tm.MF(U0.shape[1], **kw)
    if loss == 'wmrb':
        model.random_ind = torch.as_tensor(R)
    model.verbose = False
    model.fit(epochs, tm.eye(m), tm.eye(n), tm.Sparse(idx, val, shape), lr=lr)
    return model


def check_one_step(tm, U0, V0, idx, val, shape, lr, loss='mse', R=None, n_items=None, n_samples=None, fixture=None,
                   boundary_slack=True):
    """One epoch on the GPU from (U0, V0) against the fp64 closed form: mean loss to 1e-5, both tables
    inside the step interval.  ``fixture`` = (U_1, V_1) from the committed golden file (dense fp32 oracle)
    must lie in the same interval.  WMRB: hinge terms whose argument is within 1e-5 of the kink may be active or
    not in a valid fp32 evaluation; what they can move is added to the interval (oracle.sparse_ref.wmrb_slack) unless
    ``boundary_slack=False`` (exact-tie tests).  Returns (model, oracle terms; terms['slack'] for WMRB)."""
    from oracle import sparse_ref as S
    model = fit_model(tm, U0, V0, idx, val, shape, 1, lr, loss, R, n_items, n_samples)
    U64, V64 = np.asarray(U0, np.float64), np.asarray(V0, np.float64)
    idx = np.asarray(idx, np.int64)
    if loss == 'mse':
        _, _, mean, t = S.mse_epoch(U64, V64, idx, np.asarray(val, np.float64), lr)
    else:
        _, _, mean, t = S.wmrb_epoch(U64, V64, idx, np.asarray(val, np.float64), np.asarray(R), n_items, n_samples, lr)
    sU = sV = None
    if loss == 'wmrb':
        t['slack'] = S.wmrb_slack(U64, V64, idx, np.asarray(val, np.float64), np.asarray(R), n_items, n_samples)
        if boundary_slack:
            sU, sV = t['slack']['gU'], t['slack']['gV']
    if np.isfinite(mean):
        assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean), (model.loss_history_[0], mean)
    assert_step(model.user_embedding.cpu().numpy(), U0, t['gU'], lr, what=f'{loss} U', slack=sU)
    assert_step(model.item_embedding.cpu().numpy(), V0, t['gV'], lr, what=f'{loss} V', slack=sV)
    if fixture is not None:
        assert_step(fixture[0], U0, t['gU'], lr, what=f'{loss} fixture U', slack=sU)
        assert_step(fixture[1], V0, t['gV'], lr, what=f'{loss} fixture V', slack=sV)
    return model, t


def test_gather_matrix_indices_known_answer(tm, golden):
    g = golden('gather_known_answer')
    out = tm.utils.gather_matrix_indices(torch.tensor(g['input']), torch.tensor(g['index']))
    assert np.array_equal(out.cpu().numpy(), g['expected'])


def test_c1_mse_trajectory_and_steps(tm, golden):
    g = golden('c1_mse')
    lr, shape = float(g['lr']), g['A'].shape
    model = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], shape, 450, lr)
    assert rel_err(model.loss_history_[:25], g['loss'][:25]) < 1e-5
    assert rel_err(model.loss_history_, g['loss']) < 1e-3  # 450 near-sign Adam steps amplify fp32 reordering
    # single steps from the initial state and (teacher-forced) from the oracle's own later states
    for e in (0, 1, 25, 450):
        U, V = (g['U0'], g['V0']) if e == 0 else (g[f'U_{e}'], g[f'V_{e}'])
        check_one_step(tm, U, V, g['indices'], g['values'], shape, lr)
    check_one_step(tm, g['U0'], g['V0'], g['indices'], g['values'], shape, lr, fixture=(g['U_1'], g['V_1']))


def test_c1_full_run_predict_topk_recall(tm, golden):
    g = golden('c1_mse')
    m, n = g['A'].shape
    model = tm.MF(5)
    model.user_embedding = torch.tensor(g['U_450']).cuda()
    model.item_embedding = torch.tensor(g['V_450']).cuda()
    pred = model.predict().cpu().numpy()
    assert rel_err(pred, g['predictions']) < 1e-5
    assert np.array_equal(model.retrieve_user_recs(k=10), g['top10'])
    assert model.retrieve_user_recs(k=10).dtype == np.int32
    full = model.retrieve_user_recs()
    assert full.shape == (m, n) and np.array_equal(full[:, :10], g['top10'])
    assert np.array_equal(model.retrieve_user_recs(user=3, k=7), g['top10'][3, :7])
    A = torch.tensor(g['A'])
    assert np.array_equal(model.recall_at_k(A).cpu().numpy(), g['recall10'])
    assert np.array_equal(model.recall_at_k(A, preserve_rows=True).cpu().numpy(), g['recall10_rows'])
    assert np.array_equal(model.precision_at_k(A).cpu().numpy(), g['precision10'])
    sp = tm.Sparse(g['indices'], g['values'], (m, n))
    assert np.array_equal(model.recall_at_k(sp).cpu().numpy(), g['recall10'])
    allp, unobs = model.predict(A)
    assert unobs.shape[0] == int((g['A'] == 0).sum())


def test_remaining_metrics_match_dense_oracle(tm, golden):
    """precision / f1 / dcg / idcg / ndcg / predict_ranks / predict(A) / save_model."""
    from oracle import dense_ref as D
    for name, e in (('c1_mse', 450), ('wmrb_mixed', 10)):
        g = golden(name)
        U, V, A = g[f'U_{e}'], g[f'V_{e}'], g['A']
        model = tm.MF(U.shape[1], n_users=A.shape[0], n_items=A.shape[1])
        model.user_embedding, model.item_embedding = torch.tensor(U).cuda(), torch.tensor(V).cuda()
        At = torch.tensor(A)
        for k in (1, 10):
            assert np.array_equal(model.precision_at_k(At, k).cpu().numpy(), D.precision_at_k_dense(U, V, A, k))
            assert np.array_equal(model.precision_at_k(At, k, preserve_rows=True).cpu().numpy(),
                                  D.precision_at_k_dense(U, V, A, k, preserve_rows=True))
            assert abs(float(model.f1_at_k(At, k)) - float(D.f1_at_k_dense(U, V, A, k))) < 1e-6
            assert rel_err(model.dcg_at_k(At, k).cpu().numpy(), D.dcg_at_k_dense(U, V, A, k)) < 1e-5
            assert rel_err(model.idcg_at_k(At, k).cpu().numpy(), D.dcg_at_k_dense(U, V, A, k, ideal=True)) < 1e-5
            got, want = model.ndcg_at_k(At, k).cpu().numpy(), D.ndcg_at_k_dense(U, V, A, k)
            assert got.shape == want.shape and np.allclose(got, want, rtol=1e-5, atol=1e-7, equal_nan=True)
            got = model.ndcg_at_k(At, k, preserve_rows=True).cpu().numpy()
            assert np.allclose(got, D.ndcg_at_k_dense(U, V, A, k, preserve_rows=True), rtol=1e-5, atol=1e-7)
        allp, unobs = model.predict(At)
        want_all, want_un = D.predict_dense(U, V, A)
        assert rel_err(unobs.cpu().numpy(), want_un) < 1e-5 and unobs.shape[0] == want_un.shape[0]
        ranks = model.predict_ranks(At).cpu().numpy()
        # ranking of fp32 scores: compare through the values they select (ties / 1-ulp differences aside)
        assert np.array_equal(np.sort(ranks), np.arange(len(want_un)))
        assert rel_err(unobs.cpu().numpy()[ranks], want_un[D.predict_ranks_dense(U, V, A)]) < 1e-5
        cfg, res = model.save_model()
        assert cfg['Latent Dimension'] == U.shape[1] and res['User Embedding'] is model.user_embedding
        again = tm.MF.from_saved(cfg)
        assert again.n_components == U.shape[1] and again.n_items == A.shape[1]


def test_c2_mse(tm, golden):
    g = golden('c2_mse')
    idx, val, A, U0, V0 = MG.c2_inputs()
    lr = float(g['lr'])
    model = fit_model(tm, U0, V0, idx, val, A.shape, 100, lr)
    assert rel_err(model.loss_history_[:10], g['loss'][:10]) < 1e-5
    assert rel_err(model.loss_history_, g['loss']) < 1e-4  # 100 near-sign Adam steps amplify rounding
    check_one_step(tm, U0, V0, idx, val, A.shape, lr, fixture=(g['U_1'], g['V_1']))
    rec = float(model.recall_at_k(torch.tensor(A)).mean())
    assert abs(rec - float(g['recall10_mean'])) <= 1e-3


@pytest.mark.parametrize('name', ['wmrb_small', 'wmrb_mixed'])
def test_wmrb_fixtures(tm, golden, name):
    g = golden(name)
    lr, n_items, n_samples = float(g['lr']), int(g['n_items']), int(g['n_samples'])
    E = len(g['loss'])
    model = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, E, lr, 'wmrb', g['R'], n_items,
                      n_samples)
    assert rel_err(model.loss_history_[:3], g['loss'][:3]) < 1e-5
    assert rel_err(model.loss_history_, g['loss']) < 2e-3
    one, _ = check_one_step(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, lr, 'wmrb', g['R'], n_items,
                            n_samples, fixture=(g['U_1'], g['V_1']))
    check_one_step(tm, g[f'U_{E}'], g[f'V_{E}'], g['indices'], g['values'], g['A'].shape, lr, 'wmrb', g['R'], n_items, n_samples)
    if name == 'wmrb_small':
        st = one._state
        assert rel_err(st.wplan.D_in_model_order().cpu().numpy(), g['D_first']) < 1e-5
        pos = g['values'] > 0
        assert rel_err(st.wplan.delta.cpu().numpy()[pos], g['delta_first']) < 1e-5


def test_c3r_wmrb(tm, golden):
    g = golden('c3r_wmrb')
    idx, val, A, R, U0, V0 = MG.c3r_inputs()
    m, n = A.shape
    lr = float(g['lr'])
    model = fit_model(tm, U0, V0, idx, val, A.shape, 3, lr, 'wmrb', R, n, n // 2)
    assert rel_err(model.loss_history_, g['loss'][:3]) < 1e-5
    check_one_step(tm, U0, V0, idx, val, A.shape, lr, 'wmrb', R, n, n // 2, fixture=(g['U_1'], g['V_1']))


@pytest.mark.parametrize('r', [1, 3, 4, 7, 16, 33, 64, 100, 128, 200, 256, 300, 512])
def test_every_rank_geometry_mse_and_wmrb(tm, r):
    rng = np.random.default_rng(r)
    m, n, S_ = 37, 29, 11
    A = (rng.random((m, n)) < 0.2) * rng.integers(-2, 6, (m, n))
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    lr = 0.01
    check_one_step(tm, U0, V0, idx, val, (m, n), lr)
    w, t = check_one_step(tm, U0, V0, idx, val, (m, n), lr, 'wmrb', R, n, S_)
    assert rel_err(w._state.wplan.D_in_model_order().cpu().numpy(), t['D']) < 1e-5


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('r', [33, 64, 100, 128, 200, 256, 300, 512])
def test_item_pass_lane_group_per_list_every_geometry(tm, monkeypatch, r, dtype):
    """k_wsum_pass_pg (one lane group per list segment) is chosen only when a launch has >= 16384 waves of segments - i.e. by the
    full-size tests at r = 128 / 256.  Forced here (TMF_WSUM_PER_GROUP=1) for every row geometry of 16 lanes and more, both
    storage types, with user-blocked lists and a heavy item cut into several segments."""
    monkeypatch.setenv('TMF_WSUM_PER_GROUP', '1')
    monkeypatch.setenv('TMF_USER_CHUNKS', '3')
    rng = np.random.default_rng(r)
    m, n, S_ = 1400, 23, 9                       # ~550 entries per (user block, item) list; item 0: > 1024 entries in a block
    A = (rng.random((m, n)) < 0.2) * rng.integers(1, 6, (m, n))
    A[:, 0] = 1
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([np.concatenate([[0], rng.choice(np.arange(1, n), S_ - 1, replace=False)]) for _ in range(m)])   # everyone samples item 0
    lr = 0.01
    if dtype == 'f32':
        check_one_step(tm, U0, V0, idx, val, (m, n), lr, 'wmrb', R, n, S_)
        return
    from oracle import sparse_ref as S
    U0, V0 = _bf16(U0), _bf16(V0)
    kw = dict(user_weight_graph=tm.Fixed(U0), item_weight_graph=tm.Fixed(V0), loss_graph=tm.WMRB(), n_users=m, n_items=n, n_samples=S_)
    model = tm.MF(r, **kw)
    model.random_ind, model.verbose, model.factor_dtype = torch.as_tensor(R), False, torch.bfloat16
    model.fit(1, tm.eye(m), tm.eye(n), tm.Sparse(idx, val, (m, n)), lr=lr)
    U1, V1, mean, _ = S.wmrb_epoch(U0.astype(np.float64), V0.astype(np.float64), idx, val.astype(np.float64), R, n, S_, lr)
    assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
    dV = np.abs(model.item_embedding.float().cpu().numpy() - V1)
    assert (dV <= 2 ** -8 * np.abs(V1) + 1e-6).mean() > 0.99 and dV.max() <= 2 * lr + 0.01


def test_wmrb_user_chunked_item_lists(tm, golden, monkeypatch):
    """TMF_USER_CHUNKS > 1: item lists split by user block, every segment through slab + combine."""
    g = golden('wmrb_small')
    for chunks in ('2', '7'):
        monkeypatch.setenv('TMF_USER_CHUNKS', chunks)
        model, _ = check_one_step(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, float(g['lr']), 'wmrb',
                                  g['R'], 100, 50, fixture=(g['U_1'], g['V_1']))
        assert model._state.wplan.user_chunks == int(chunks)
    monkeypatch.delenv('TMF_USER_CHUNKS')
    base = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 3, 0.1, 'wmrb', g['R'], 100, 50)
    monkeypatch.setenv('TMF_USER_CHUNKS', '4')
    chunked = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 3, 0.1, 'wmrb', g['R'], 100, 50)
    assert rel_err(chunked.loss_history_, base.loss_history_) < 1e-6


@pytest.mark.parametrize('slices,xcd', [('2', '0'), ('5', '0'), ('64', '0'), ('5', '1'), ('13', '1'), ('64', '1')])
def test_wmrb_sliced_user_pass(tm, golden, monkeypatch, slices, xcd):
    """TMF_ITEM_SLICES > 1: the sliced user pass (scores / hinge / gradU / finish) instead of the fused kernel;
    also with per-slice gradU launches, and in XCD-major block order (every XCD its own slice; gradU then in rounds of
    eight slices, one layer each, when the layers do not fit)."""
    monkeypatch.setenv('TMF_ITEM_SLICES', slices)
    monkeypatch.setenv('TMF_SLICE_XCD', xcd)
    if slices in ('5', '13'):
        monkeypatch.setattr(tm.engine, 'PART_BUDGET', 0)  # memory-light gradU: one launch per slice / per round of eight
    for name in ('wmrb_small', 'wmrb_mixed'):
        g = golden(name)
        model, t = check_one_step(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, float(g['lr']), 'wmrb',
                                  g['R'], int(g['n_items']), int(g['n_samples']), fixture=(g['U_1'], g['V_1']))
        assert model._state.wplan.n_slices == int(slices) and model._state.wplan.xcd_major == (xcd == '1')
        if slices in ('5', '13'):
            assert model._state.gradu_launches == (3 if xcd == '1' else 1)
        assert rel_err(model._state.wplan.D_in_model_order().cpu().numpy(), t['D']) < 1e-5
    g = golden('wmrb_small')
    sliced = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 25, 0.1, 'wmrb', g['R'], 100, 50)
    assert rel_err(sliced.loss_history_[:3], g['loss'][:3]) < 1e-5 and rel_err(sliced.loss_history_, g['loss']) < 2e-3
    monkeypatch.setenv('TMF_USER_CHUNKS', '3')
    # rows of 32 lanes: also through the row-stationary gradU (tmf_wmrb_gradu4: lane groups own users, walk the slices themselves)
    monkeypatch.setenv('TMF_ROW_STATIONARY', '1' if slices in ('5', '64') else '0')
    monkeypatch.setenv('TMF_ROWS4', '1' if slices in ('2', '64') else '0')   # and the row-stationary item pass (tmf_wsum_rows4)
    rng = np.random.default_rng(int(slices))
    m, n, r, S_ = 41, 67, 100, 23
    A = (rng.random((m, n)) < 0.15) * rng.integers(-1, 6, (m, n))
    idx, val = np.argwhere(A != 0), A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    model, _ = check_one_step(tm, U0, V0, idx, val, (m, n), 0.01, 'wmrb', R, n, S_)
    assert model._state.row_stationary == (slices in ('5', '64')) and model._state.wplan.rows4 == (slices in ('2', '64'))


@pytest.mark.parametrize('forced_slices', [None, '3'])
def test_wmrb_n_samples_beyond_lds_takes_the_sliced_pass(tm, monkeypatch, forced_slices):
    """S = 30000 negatives per user: 240 KB of scores + D per user do not fit the 160 KB of LDS the one-kernel user pass
    needs, so the plan switches to the sliced pass (which has no limit on S) - with one slice for this small V table,
    or with the forced number."""
    if forced_slices:
        monkeypatch.setenv('TMF_ITEM_SLICES', forced_slices)
    rng = np.random.default_rng(3)
    m, n, r, S_ = 5, 40000, 8, 30000
    idx = np.stack([rng.integers(0, m, 60), rng.integers(0, n, 60)], axis=1)
    idx = np.unique(idx, axis=0)
    val = rng.integers(-1, 6, len(idx)).astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    model, t = check_one_step(tm, U0, V0, idx, val, (m, n), 0.01, 'wmrb', R, n, S_)
    w = model._state.wplan
    assert w.sliced and w.n_slices == (int(forced_slices) if forced_slices else 1)
    assert not tm.engine.fused_user_pass_fits(S_, r) and tm.engine.fused_user_pass_fits(1024, 128)
    assert rel_err(w.D_in_model_order().cpu().numpy(), t['D']) < 1e-5


def test_mse_user_blocked_item_pass(tm, golden, monkeypatch):
    """TMF_USER_CHUNKS > 1 with MSE: the item pass walks (user block, item) lists; every segment through slab + combine."""
    monkeypatch.setenv('TMF_USER_CHUNKS', '4')
    for name in ('c1_mse', 'wmrb_mixed'):
        g = golden(name)
        model, _ = check_one_step(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, float(g['lr']))
        assert model._state.plan.user_chunks == 4 and model._state.plan.seg_i.row_mod == g['A'].shape[1]
    g = golden('c1_mse')
    blocked = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 25, float(g['lr']))
    assert rel_err(blocked.loss_history_, g['loss'][:25]) < 1e-5


def test_heavy_rows_are_segmented_and_combined(tm):
    """Rows longer than the segment length go through the slab + combine path."""
    rng = np.random.default_rng(7)
    m, n, r = 6, 5000, 32
    A = np.zeros((m, n), np.float32)
    A[0, :] = rng.integers(1, 6, n)           # one user with 5000 interactions (5 segments)
    A[1, :1024] = 3                            # exactly one full segment
    A[2, :1025] = 2                            # one entry into the second segment
    A[4, ::7] = rng.integers(1, 6, len(A[4, ::7]))
    idx = np.argwhere(A != 0)
    val = A[A != 0]
    U0 = (rng.standard_normal((m, r)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.1).astype(np.float32)
    model, _ = check_one_step(tm, U0, V0, idx, val, (m, n), 0.01)
    assert model._state.plan.seg_u.n_long == 2 and model._state.plan.seg_u.n_slab == 5 + 2
    assert np.array_equal(model.user_embedding.cpu().numpy()[3], U0[3])  # user without interactions: untouched


def test_unsorted_and_duplicate_interactions(tm):
    rng = np.random.default_rng(11)
    m, n, r = 20, 30, 8
    idx = np.stack([rng.integers(0, m, 200), rng.integers(0, n, 200)], axis=1)  # unsorted, with duplicates
    val = rng.integers(1, 6, 200).astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.2).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.2).astype(np.float32)
    check_one_step(tm, U0, V0, idx, val, (m, n), 0.01)


def test_empty_interactions_leave_tables_unchanged(tm):
    U0 = np.random.default_rng(0).standard_normal((5, 4)).astype(np.float32)
    V0 = np.random.default_rng(1).standard_normal((6, 4)).astype(np.float32)
    model = fit_model(tm, U0, V0, np.zeros((0, 2), np.int64), np.zeros(0, np.float32), (5, 6), 2, 0.01)
    assert np.array_equal(model.user_embedding.cpu().numpy(), U0)
    assert np.array_equal(model.item_embedding.cpu().numpy(), V0)
    assert all(np.isnan(x) for x in model.loss_history_)


@pytest.mark.parametrize('force', [None, 'TMF_FORCE_SLICED', 'TMF_FORCE_FUSED'])
def test_empty_interactions_leave_tables_unchanged_wmrb(tm, monkeypatch, force):
    """The same with the WMRB loss, on the one-kernel and on the sliced user pass: no positive, no loss term, no gradient."""
    for k in ('TMF_FORCE_SLICED', 'TMF_FORCE_FUSED', 'TMF_ITEM_SLICES'):
        monkeypatch.delenv(k, raising=False)
    if force:
        monkeypatch.setenv(force, '1')
    rng = np.random.default_rng(3)
    m, n, r, S = 40, 60, 16, 7
    U0 = rng.standard_normal((m, r)).astype(np.float32)
    V0 = rng.standard_normal((n, r)).astype(np.float32)
    R = np.stack([rng.choice(n, S, replace=False) for _ in range(m)])
    for idx, val in ((np.zeros((0, 2), np.int64), np.zeros(0, np.float32)),
                     (np.array([[3, 5], [7, 1]]), np.array([0.0, -2.0], np.float32))):      # stored values, none of them positive
        model = fit_model(tm, U0, V0, idx, val, (m, n), 2, 0.01, 'wmrb', R, n, S)
        assert np.array_equal(model.user_embedding.cpu().numpy(), U0)
        assert np.array_equal(model.item_embedding.cpu().numpy(), V0)
        assert all(np.isnan(x) for x in model.loss_history_)


def test_topk_tie_rule_and_clamp(tm):
    x = torch.tensor([[0., 1, 1, 0, 1], [-1., -2, -3, -4, -5]])
    assert tm.ops.topk_stable(x, 3).cpu().tolist() == [[1, 2, 4], [0, 1, 2]]
    assert tm.ops.topk_stable(x, 3, clamp_negatives=True).cpu().tolist() == [[1, 2, 4], [0, 1, 2]]
    z = torch.zeros(3, 1000)
    assert tm.ops.topk_stable(z, 10).cpu().tolist() == [list(range(10))] * 3
    rng = np.random.default_rng(5)
    for cols in (1, 7, 64, 257, 1000, 4099, 16385, 100_003):   # the last two: wider than the in-LDS sort holds
        y = torch.tensor(rng.integers(-3, 4, (17 if cols < 50_000 else 5, cols)).astype(np.float32))  # many ties
        if cols > 1000:
            y[1, ::3] = -0.0                                         # -0.0 ties with +0.0, lower index first
        for k in sorted({1, min(10, cols), min(64, cols), min(100, cols), min(1000, cols), cols}):
            want = torch.sort(y, dim=1, descending=True, stable=True)[1][:, :k]
            got = tm.ops.topk_stable(y, k).cpu()
            assert torch.equal(got.to(torch.int64), want), (cols, k)
            wc = torch.sort(torch.clamp(y, min=0), dim=1, descending=True, stable=True)[1][:, :k]
            assert torch.equal(tm.ops.topk_stable(y, k, clamp_negatives=True).cpu().to(torch.int64), wc), (cols, k)
    with pytest.raises(ValueError):
        tm.ops.topk_stable(x, 6)
    # rows split over several calls of the wide path (workspace bounded), values returned too
    big = torch.tensor(rng.standard_normal((6, 40_000)).astype(np.float32)).cuda()
    old = tm.ops.SORT_MAX_ELEMS
    tm.ops.SORT_MAX_ELEMS = 100_000
    try:
        vals, idx = tm.ops.topk_stable(big, 500, return_values=True)
    finally:
        tm.ops.SORT_MAX_ELEMS = old
    sv, si = torch.sort(big, dim=1, descending=True, stable=True)
    assert torch.equal(idx.to(torch.int64), si[:, :500]) and torch.equal(vals, sv[:, :500])


def test_full_ranking_and_large_k_on_a_real_catalog(tm):
    """retrieve_user_recs(user, k=None), k = 100 for every user, and the dcg / ndcg family at catalog widths the in-LDS
    sort cannot hold (matrix_factorization.py:424-438, :332-413): bit-equal to the oracle's stable ranking."""
    from oracle import dense_ref as D
    from oracle import sparse_ref as S
    rng = np.random.default_rng(31)
    m, n, r = 40, 100_000, 16
    U = rng.integers(-2, 3, (m, r)).astype(np.float32)      # small integers: many exact score ties
    V = rng.integers(-2, 3, (n, r)).astype(np.float32)
    model = tm.MF(r, n_users=m, n_items=n)
    model.user_embedding, model.item_embedding = torch.tensor(U).cuda(), torch.tensor(V).cuda()
    sc = U @ V.T
    full = model.retrieve_user_recs(user=3)
    assert full.dtype == np.int32 and np.array_equal(full, S.topk_stable(sc[3:4], n)[0])
    assert np.array_equal(model.retrieve_user_recs(k=100), S.topk_stable(sc, 100))       # k > 64: block-wise scores + sort
    assert np.array_equal(model.retrieve_user_recs(k=64), S.topk_stable(sc, 64))         # fused kernel, largest k
    assert np.array_equal(model.retrieve_user_recs(), S.topk_stable(sc, n))              # every user's full ranking
    n2 = 20_000
    A = ((rng.random((m, n2)) < 0.01) * rng.integers(1, 6, (m, n2))).astype(np.float32)
    model.item_embedding = model.item_embedding[:n2]
    Un = U / 8
    model.user_embedding = torch.tensor(Un).cuda()
    for k in (10, 100):
        got = model.ndcg_at_k(torch.tensor(A), k, preserve_rows=True).cpu().numpy()
        assert np.allclose(got, D.ndcg_at_k_dense(Un, V[:n2], A, k, preserve_rows=True), rtol=1e-5, atol=1e-7)
        assert rel_err(model.dcg_at_k(torch.tensor(A), k).cpu().numpy(), D.dcg_at_k_dense(Un, V[:n2], A, k)) < 1e-5
        rec = model.recall_at_k(torch.tensor(A), k).cpu().numpy()
        assert np.array_equal(rec, D.recall_at_k_dense(Un, V[:n2], A, k))


def test_fused_predict_topk_matches_materialised_and_oracle(tm):
    from oracle import sparse_ref as S
    rng = np.random.default_rng(9)
    for m, n, r in [(1, 5, 3), (100, 50, 5), (130, 257, 32), (257, 1000, 64), (300, 4099, 128), (64, 128, 100), (200, 1500, 256),
                    (70, 900, 130)]:
        U = rng.standard_normal((m, r)).astype(np.float32)
        V = rng.standard_normal((n, r)).astype(np.float32)
        if n > 200:
            V[50:60] = V[40]          # duplicate item rows -> exact score ties across tiles
            V[n - 3:] = V[7]
        Ut, Vt = torch.tensor(U), torch.tensor(V)
        scores = tm.ops.predict_gemm(Ut, Vt)
        for k in sorted({1, min(10, n), min(32, n), min(33, n), min(64, n)}):
            for clamp in (False, True):
                want = tm.ops.topk_stable(scores, k, clamp_negatives=clamp).cpu()
                got_v, got = tm.ops.predict_topk(Ut, Vt, k, clamp_negatives=clamp, return_values=True)
                assert torch.equal(got.cpu(), want), (m, n, r, k, clamp)
                sc = scores.cpu().numpy()
                ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
                assert np.array_equal(got.cpu().numpy(), ref), (m, n, r, k, clamp)
    # all-equal scores (everything clamped to 0): must return 0..k-1
    Z = torch.zeros(200, 16)
    assert tm.ops.predict_topk(Z, torch.ones(5000, 16), 10, clamp_negatives=True).cpu().tolist() == [list(range(10))] * 200
    # scores increasing with the item index: every tile overflows the pending buffer (slow path)
    inc = torch.arange(3000, dtype=torch.float32)[:, None] * torch.ones(1, 4)
    got = tm.ops.predict_topk(torch.ones(7, 4), inc, 5).cpu().tolist()
    assert got == [[2999, 2998, 2997, 2996, 2995]] * 7
    with pytest.raises(Exception):
        tm.ops.predict_topk(torch.ones(3, 4), torch.ones(10, 4), 33 if False else 11)


def test_fused_predict_topk_bf16(tm):
    """bf16 tables on the bf16 MFMA: with small-integer factors every product and partial sum is exact, so the
    ranking must equal the fp32 path bit for bit (ties included); random factors are compared by value."""
    from oracle import sparse_ref as S
    rng = np.random.default_rng(17)
    for m, n, r in [(1, 9, 3), (300, 515, 40), (257, 1000, 64), (513, 2051, 128), (100, 700, 256), (64, 300, 200)]:
        U = rng.integers(-2, 3, (m, r)).astype(np.float32)
        V = rng.integers(-2, 3, (n, r)).astype(np.float32)
        Ub, Vb = torch.tensor(U).to(torch.bfloat16).cuda(), torch.tensor(V).to(torch.bfloat16).cuda()
        sc = U @ V.T
        for k in sorted({1, min(10, n), min(32, n)}):
            for clamp in (False, True):
                vals, got = tm.ops.predict_topk(Ub, Vb, k, clamp_negatives=clamp, return_values=True)
                ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
                assert np.array_equal(got.cpu().numpy(), ref), (m, n, r, k, clamp)
                want = np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, 1)
                assert np.array_equal(vals.cpu().numpy(), want)
    U = torch.tensor(rng.standard_normal((700, 256)).astype(np.float32)).to(torch.bfloat16).cuda()
    V = torch.tensor(rng.standard_normal((5000, 256)).astype(np.float32)).to(torch.bfloat16).cuda()
    vals, idx = tm.ops.predict_topk(U, V, 10, return_values=True)
    sc = U.float() @ V.float().T
    want = torch.sort(sc, dim=1, descending=True)[0][:, :10]
    assert rel_err(vals.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert rel_err(torch.gather(sc, 1, idx.to(torch.int64)).cpu().numpy(), vals.cpu().numpy()) < 1e-5
    model = tm.MF(256)
    model.user_embedding, model.item_embedding = U, V
    assert np.array_equal(model.retrieve_user_recs(k=10), idx.cpu().numpy())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fused_predict_topk_deferred_merges(tm, dtype):
    """Rows whose running top-k is beaten every 20..41 items: a few candidates per 128-item tile, so the pending
    buffers fill over several tiles, rows of one wave reach the merge threshold at different times, and the final
    flush matters (the last records sit in the last tiles).  Ragged last tile; exact expected ranking."""
    from oracle import sparse_ref as S
    n, r, m = 3001, 8, 300
    period = 20 + 3 * np.arange(r)
    j = np.arange(n)
    V = np.where(j[:, None] % period[None, :] == 0, (j // 16 + 1)[:, None], -1).astype(np.float32)
    U = np.eye(r, dtype=np.float32)[np.arange(m) % r]
    U[7] = 0                                            # a row of all-equal (zero) scores: lowest indices win
    sc = U @ V.T
    Ut, Vt = torch.tensor(U), torch.tensor(V)
    if dtype == 'bf16':
        Ut, Vt = Ut.to(torch.bfloat16).cuda(), Vt.to(torch.bfloat16).cuda()
    for k in (1, 10, 32):
        for clamp in (False, True):
            ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
            vals, got = tm.ops.predict_topk(Ut, Vt, k, clamp_negatives=clamp, return_values=True)
            assert np.array_equal(got.cpu().numpy(), ref), (dtype, k, clamp)
            assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, 1))


def test_predict_gemm_shapes(tm):
    rng = np.random.default_rng(3)
    for m, n, r in [(1, 1, 1), (100, 50, 5), (129, 257, 32), (300, 1000, 128), (64, 64, 7)]:
        U = torch.tensor(rng.standard_normal((m, r)).astype(np.float32))
        V = torch.tensor(rng.standard_normal((n, r)).astype(np.float32))
        got = tm.ops.predict_gemm(U, V).cpu().numpy()
        want = (U.double() @ V.double().T).numpy()
        assert rel_err(got, want) < 1e-5, (m, n, r)


def test_graph_replay_equals_eager(tm, golden, monkeypatch):
    """fit() replays hipGraph-captured epochs on small problems; results must equal the eager launches bit for bit."""
    g = golden('wmrb_small')
    args = (g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape)
    for epochs in (10, 13, 120):
        a = fit_model(tm, *args, epochs, 0.1, 'wmrb', g['R'], 100, 50)
        m = fit_model(tm, *args, epochs, 0.01)
        monkeypatch.setenv('TMF_NO_GRAPH', '1')
        b = fit_model(tm, *args, epochs, 0.1, 'wmrb', g['R'], 100, 50)
        n = fit_model(tm, *args, epochs, 0.01)
        monkeypatch.delenv('TMF_NO_GRAPH')
        assert a.loss_history_ == b.loss_history_ and torch.equal(a.user_embedding, b.user_embedding)
        assert torch.equal(a.item_embedding, b.item_embedding)
        assert m.loss_history_ == n.loss_history_ and torch.equal(m.item_embedding, n.item_embedding)


def _bf16(x):
    return torch.tensor(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize('r', [5, 32, 100, 256, 300, 600])
def test_bf16_storage_one_step(tm, r):
    """bf16 factor storage / fp32 arithmetic (BASELINE config 5, reduced size): against the fp64 closed form
    evaluated on the bf16-rounded tables; the new rows must lie in the step interval rounded to bf16."""
    from conftest import step_bounds
    from oracle import sparse_ref as S
    rng = np.random.default_rng(r)
    m, n, S_, lr = 33, 47, 9, 0.05
    A = (rng.random((m, n)) < 0.2) * rng.integers(-1, 6, (m, n))
    idx, val = np.argwhere(A != 0), A[A != 0].astype(np.float32)
    U0 = _bf16(rng.standard_normal((m, r)) * 0.3)
    V0 = _bf16(rng.standard_normal((n, r)) * 0.3)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    for loss in ('mse', 'wmrb'):
        kw = dict(user_weight_graph=tm.Fixed(U0), item_weight_graph=tm.Fixed(V0))
        if loss == 'wmrb':
            kw.update(loss_graph=tm.WMRB(), n_users=m, n_items=n, n_samples=S_)
        model = tm.MF(r, **kw)
        model.factor_dtype, model.verbose = torch.bfloat16, False
        if loss == 'wmrb':
            model.random_ind = torch.as_tensor(R)
        model.fit(1, tm.eye(m), tm.eye(n), tm.Sparse(idx, val, (m, n)), lr=lr)
        assert model.user_embedding.dtype == torch.bfloat16
        U64, V64 = U0.astype(np.float64), V0.astype(np.float64)
        if loss == 'mse':
            _, _, mean, t = S.mse_epoch(U64, V64, idx, val.astype(np.float64), lr)
        else:
            _, _, mean, t = S.wmrb_epoch(U64, V64, idx, val.astype(np.float64), R, n, S_, lr)
        assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
        for got, W0, g in ((model.user_embedding, U0, t['gU']), (model.item_embedding, V0, t['gV'])):
            lo, hi = step_bounds(W0, g, lr)
            got = got.to(torch.float32).cpu().numpy().astype(np.float64)
            assert (got >= _bf16(lo) - 1e-12).all() and (got <= _bf16(hi) + 1e-12).all(), (loss, r)


def test_bf16_sliced_user_pass(tm, monkeypatch):
    """bf16 tables through the staged sliced pass must give the same bits as through the fused kernel
    (same arithmetic, different gather order is not involved: per-user sums keep their order)."""
    monkeypatch.setenv('TMF_USER_CHUNKS', '3')
    rng = np.random.default_rng(21)
    m, n, r, S_ = 70, 90, 64, 31
    A = (rng.random((m, n)) < 0.15) * rng.integers(-1, 6, (m, n))
    idx, val = np.argwhere(A != 0), A[A != 0].astype(np.float32)
    U0, V0 = _bf16(rng.standard_normal((m, r)) * 0.3), _bf16(rng.standard_normal((n, r)) * 0.3)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    out = []
    for slices in (None, '4'):
        if slices:
            monkeypatch.setenv('TMF_ITEM_SLICES', slices)
        model = tm.MF(r, loss_graph=tm.WMRB(), n_users=m, n_items=n, n_samples=S_, user_weight_graph=tm.Fixed(U0),
                      item_weight_graph=tm.Fixed(V0))
        model.factor_dtype, model.verbose, model.random_ind = torch.bfloat16, False, torch.as_tensor(R)
        model.fit(3, tm.eye(m), tm.eye(n), tm.Sparse(idx, val, (m, n)), lr=0.05)
        out.append(model)
    assert out[1]._state.wplan.n_slices == 4
    assert rel_err(out[1].loss_history_, out[0].loss_history_) < 1e-5
    diff = (out[0].user_embedding.float() - out[1].user_embedding.float()).abs().max()
    assert float(diff) <= 0.05 * 2 + 1e-6  # a step of lr either way on elements whose gradient is ~0


def test_bf16_storage_trajectory(tm, golden):
    from oracle import sparse_ref as S
    g = golden('wmrb_small')
    U, V = _bf16(g['U0']), _bf16(g['V0'])
    model = tm.MF(3, loss_graph=tm.WMRB(), n_users=50, n_items=100, n_samples=50, user_weight_graph=tm.Fixed(U),
                  item_weight_graph=tm.Fixed(V))
    model.factor_dtype, model.verbose, model.random_ind = torch.bfloat16, False, torch.as_tensor(g['R'])
    model.fit(12, tm.eye(50), tm.eye(100), tm.Sparse(g['indices'], g['values'], (50, 100)), lr=0.1)
    ref = []
    for _ in range(12):  # oracle with the tables rounded to bf16 after every step
        U, V, mean, _ = S.wmrb_epoch(U, V, g['indices'], g['values'], g['R'], 100, 50, 0.1)
        U, V = _bf16(U), _bf16(V)
        ref.append(mean)
    assert rel_err(model.loss_history_[:2], ref[:2]) < 1e-5
    assert rel_err(model.loss_history_, ref) < 2e-2  # bf16 rounding ties flip under fp32 reordering
    assert float(model.recall_at_k(torch.tensor(g['A'])).mean()) > 0


def test_data_parallel_fit_world_size_one_equals_single_process(tm, golden):
    """The N>1 code path (gradient epilogue -> reduce-scatter -> Adam on the shard -> all-gather) run with a
    1-rank RCCL group on this GPU must give the same bits as the single-process fit."""
    import torch.distributed as dist
    from teamoflow_amd import dist as tdist
    if not dist.is_initialized():
        import os
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        for name, loss in (('wmrb_small', 'wmrb'), ('c1_mse', 'mse')):
            g = golden(name)
            m, n = g['A'].shape
            args = dict(R=g.get('R'), n_items=n, n_samples=int(g['n_samples']) if loss == 'wmrb' else None)
            base = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], (m, n), 4, float(g['lr']), loss, **args)
            kw = dict(user_weight_graph=tm.Fixed(g['U0']), item_weight_graph=tm.Fixed(g['V0']))
            if loss == 'wmrb':
                kw.update(loss_graph=tm.WMRB(), n_users=m, n_items=n, n_samples=int(g['n_samples']))
            model = tm.MF(g['U0'].shape[1], **kw)
            model.verbose, model.data_parallel = False, 'force'
            if loss == 'wmrb':
                model.random_ind = torch.as_tensor(g['R'])
            model.fit(4, tm.eye(m), tm.eye(n), tm.Sparse(g['indices'], g['values'], (m, n)), lr=float(g['lr']))
            assert model.user_block == (0, m)
            assert rel_err(model.loss_history_, base.loss_history_) < 1e-12
            assert torch.equal(model.item_embedding, base.item_embedding)
            assert torch.equal(tdist.gather_user_embedding(model, m), base.user_embedding)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('slices', [1, 5])
def test_midsize_against_c_oracle(tm, monkeypatch, slices):
    """20K x 5K, r=128, S=512, ~1M power-law interactions: too big for the NumPy oracle's [P, S] temporaries in a
    test, seconds for the C/OpenMP restatement (oracle/sparse_ref.c).  Whole tables against the step interval, D
    and delta to 1e-5, for the fused and the sliced user pass."""
    from oracle import sparse_c as C
    rng = np.random.default_rng(11)
    m, n, r, S_, lr = 20000, 5000, 128, 512, 0.1
    deg = np.clip(np.round(np.exp(3.3 + 0.8 * rng.standard_normal(m))), 1, n // 4).astype(np.int64)
    rows = np.repeat(np.arange(m), deg)
    cols = np.minimum((np.power(n + 1.0, rng.random(len(rows))) - 1).astype(np.int64), n - 1)
    key = np.unique(rows * n + cols)
    idx = np.stack([key // n, key % n], 1)
    val = rng.integers(-1, 6, len(idx)).astype(np.float32)      # some zero / negative stored values
    U0 = (rng.standard_normal((m, r)) / np.sqrt(r)).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) / np.sqrt(r)).astype(np.float32)
    R = np.argsort(rng.random((m, n)), axis=1)[:, :S_].astype(np.int32)
    monkeypatch.setenv('TMF_ITEM_SLICES', str(slices))
    plan = C.Plan(idx, val, m, n, R)
    Uc, Vc, mean, t = C.wmrb_epoch(U0, V0, plan, n, S_, lr)
    model = fit_model(tm, U0, V0, idx, val, (m, n), 1, lr, 'wmrb', R, n, S_)
    assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
    assert rel_err(model._state.wplan.D_in_model_order().cpu().numpy(), t['D']) < 1e-5
    assert_step(model.user_embedding.cpu().numpy(), U0, t['gU'], lr, what='wmrb U')
    assert_step(model.item_embedding.cpu().numpy(), V0, t['gV'], lr, what='wmrb V')
    if slices == 1:
        Uc, Vc, mean, t = C.mse_epoch(U0, V0, plan, 1e-2)
        model = fit_model(tm, U0, V0, idx, val, (m, n), 1, 1e-2)
        assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
        assert_step(model.user_embedding.cpu().numpy(), U0, t['gU'], 1e-2, what='mse U')
        assert_step(model.item_embedding.cpu().numpy(), V0, t['gV'], 1e-2, what='mse V')


@pytest.mark.parametrize('loss,knobs', [('mse', {}), ('wmrb', {}), ('wmrb', {'TMF_ITEM_SLICES': '3', 'TMF_USER_CHUNKS': '2'}),
                                        ('mse', {'TMF_USER_CHUNKS': '3'})])
def test_two_ranks_on_one_card_match_single_process(tmp_path, loss, knobs):
    """Two ranks, both on cuda:0, gloo group with host-staged collectives (tools/dp_rehearsal.py): the skewed
    user partition, the padded V, the reduce-scatter -> Adam-on-shard -> all-gather exchange and the loss
    all-reduce on the HIP engine reproduce the single-process fit.  The item gradient is summed in a different
    order (two partials), so later epochs are compared at trajectory tolerance, not bitwise."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out = tmp_path / 'dp.json'
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **knobs)  # knobs: sliced user pass / user-blocked lists on every rank
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tools', 'dp_rehearsal.py'), str(out), loss],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    res = json.loads(out.read_text())
    (b0, e0), (b1, e1) = res['blocks']
    assert b0 == 0 and e0 == b1 and e1 == 3001 and 0 < e0 < 3001
    assert abs(res['loss_dp'][0] - res['loss_one'][0]) <= 1e-9 * abs(res['loss_one'][0])   # same weights, same data
    assert rel_err(res['loss_dp'], res['loss_one']) < 1e-5
    # mean recall@10 over all users = 2-double all-reduce of per-rank (sum, count); equals ranking the assembled tables
    assert abs(res['recall_all_ranks'] - res['recall_assembled_tables']) <= 1e-12 and 0 < res['recall_all_ranks'] < 1
    assert res['U1_equal']                       # after one epoch the user table is bit-identical
    assert res['V1_frac_close'] > 0.99           # item rows: two partial sums instead of one; |diff| <= 1e-6 except where g ~ 0
    assert res['V1_max_abs_diff'] <= 2.0 * 0.05 + 1e-6


@pytest.mark.parametrize('seed', range(int(os.environ.get('TMF_FUZZ_SEEDS', '12'))))   # TMF_FUZZ_SEEDS=300 for a soak run
def test_randomized_shapes_against_oracle(tm, monkeypatch, seed):
    """Random small problems: ragged / empty rows, mixed-sign values, every path selector (fused or sliced
    user pass, user-blocked item lists, fp32 or bf16-free) - one step each against the fp64 closed form."""
    rng = np.random.default_rng(1000 + seed)
    m, n = int(rng.integers(1, 60)), int(rng.integers(2, 80))
    r = int(rng.choice([1, 2, 5, 8, 17, 32, 48, 64, 96, 128, 160, 257]))
    S_ = int(rng.integers(1, n + 1))
    density = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
    A = (rng.random((m, n)) < density) * rng.integers(-2, 6, (m, n))
    if seed % 4 == 1:
        A[rng.integers(0, m)] = rng.integers(1, 6, n)      # one user interacted with everything
    idx, val = np.argwhere(A != 0), A[A != 0].astype(np.float32)
    if seed % 3 == 0 and len(val):                          # shuffle: the engine must not rely on row-major order
        p = rng.permutation(len(val))
        idx, val = idx[p], val[p]
    U0 = (rng.standard_normal((m, r)) * 0.4).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.4).astype(np.float32)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    lr = float(rng.choice([1e-3, 1e-2, 0.1]))
    if seed % 2:
        monkeypatch.setenv('TMF_ITEM_SLICES', str(int(rng.integers(2, 9))))
    if seed % 3 == 2:
        monkeypatch.setenv('TMF_USER_CHUNKS', str(int(rng.integers(2, 6))))
    if len(val):
        check_one_step(tm, U0, V0, idx, val, (m, n), lr)
    if (val > 0).any():
        model, t = check_one_step(tm, U0, V0, idx, val, (m, n), lr, 'wmrb', R, n, S_)
        assert rel_err(model._state.wplan.D_in_model_order().cpu().numpy(), t['D']) < 1e-5
        top = model.retrieve_user_recs(k=min(5, n))
        from oracle import sparse_ref as S
        sc = (model.user_embedding.float() @ model.item_embedding.float().T).cpu().numpy()
        pred = tm.ops.predict_gemm(model.user_embedding, model.item_embedding).cpu().numpy()
        assert np.array_equal(top, S.topk_stable(pred, min(5, n)))
        assert rel_err(pred, sc) < 1e-5


def test_native_index_prep_equals_torch_prep(tm):
    """tmf_csr_build / tmf_stable_order_i32 (device path of the plans) against the torch ops the CPU path uses."""
    rng = np.random.default_rng(8)
    for m, n, nnz, S_, chunks in [(40, 30, 500, 7, 1), (1000, 777, 50000, 16, 5), (5, 9, 0, 3, 2), (300, 4000, 20000, 64, 3)]:
        idx = np.stack([rng.integers(0, m, nnz), rng.integers(0, n, nnz)], axis=1).reshape(-1, 2)
        val = rng.integers(-1, 5, nnz).astype(np.float32)
        R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)]).astype(np.int32)
        plans = []
        for dev in ('cpu', 'cuda'):
            p = tm.engine.InteractionPlan(torch.tensor(idx, device=dev), torch.tensor(val, device=dev), m, n, chunk=64,
                                          user_chunks=chunks)
            w = tm.engine.WmrbPlan(p, torch.tensor(R, device=dev), chunk=64, user_chunks=chunks, item_slices=3)
            plans.append((p, w))
        (pc, wc), (pg, wg) = plans
        for name in ('rowptr_u', 'col_u', 'val_u', 'rowptr_i', 'row_i', 'val_i', 'user_of'):
            assert torch.equal(getattr(pc, name), getattr(pg, name).cpu()), (name, m, n, nnz)
        for name in ('ent_row', 'ent_w', 'rowptr_e', 'R', 'slice_off', 'pos_off'):
            assert torch.equal(getattr(wc, name), getattr(wg, name).cpu()), (name, m, n, nnz)
        wc1 = tm.engine.WmrbPlan(pc, torch.tensor(R), chunk=64, user_chunks=chunks)          # one-kernel user pass: ent_w
        wg1 = tm.engine.WmrbPlan(pg, torch.tensor(R, device='cuda'), chunk=64, user_chunks=chunks)
        for name in ('ent_row', 'ent_w', 'rowptr_e', 'R'):
            assert torch.equal(getattr(wc1, name), getattr(wg1, name).cpu()), (name, m, n, nnz)
        for seg_c, seg_g in ((pc.seg_u, pg.seg_u), (pc.seg_i, pg.seg_i), (wc.seg_e, wg.seg_e)):
            for name in ('seg_row', 'seg_chunk', 'seg_slab', 'long_rows', 'long_slab_beg'):
                assert torch.equal(getattr(seg_c, name), getattr(seg_g, name).cpu()), name


def test_c_abi_error_codes_and_messages(tm):
    """Bad arguments come back as negative codes with a message, never as a launch on bad pointers."""
    import ctypes
    lib = tm.lib.get()
    adam = lib.tmf_adam_fresh(0.01)
    s = tm.lib.stream_ptr()
    x = torch.zeros(8, 8, device='cuda')
    out = torch.zeros(8, 8, device='cuda', dtype=torch.int32)
    rc = lib.tmf_topk_stable_f32(tm.lib.ptr(x), 8, 8, 8, 9, 0, tm.lib.ptr(out), None, None, 0, s)        # k > cols
    assert rc == -1 and b'k=9' in lib.tmf_last_error()
    wide = torch.zeros(2, 20000, device='cuda')
    wout = torch.zeros(2, 100, device='cuda', dtype=torch.int32)
    assert lib.tmf_topk_workspace_bytes(2, 20000, 64) == 0 and lib.tmf_topk_workspace_bytes(2, 20000, 100) > 0
    rc = lib.tmf_topk_stable_f32(tm.lib.ptr(wide), 2, 20000, 20000, 100, 0, tm.lib.ptr(wout), None, None, 0, s)   # no workspace
    assert rc == -1 and b'workspace' in lib.tmf_last_error()
    rc = lib.tmf_predict_topk_f32(tm.lib.ptr(x), tm.lib.ptr(x), 8, 8, 8, 8, 8, 33, 0, tm.lib.ptr(out), None, s)  # k > 32
    assert rc == -1 or rc == -3
    big = torch.zeros(4, 260, device='cuda')
    rc = lib.tmf_predict_topk_f32(tm.lib.ptr(big), tm.lib.ptr(big), 4, 4, 258, 260, 260, 2, 0, tm.lib.ptr(out), None, s)
    assert rc == -3 and b'n_components <= 256' in lib.tmf_last_error()                            # unsupported width
    rc = lib.tmf_predict_gemm_f32(None, tm.lib.ptr(x), tm.lib.ptr(x), 8, 8, 8, 8, 8, 8, s)
    assert rc == -1
    seg = tm.lib.Segments(0, 0, 0, 0, 5, 1024, 0)                                                 # null arrays, nseg = 5
    rc = lib.tmf_mse_pass_f32(ctypes.byref(seg), None, None, tm.lib.ptr(x), tm.lib.ptr(x), tm.lib.ptr(x), None, None, 8, 0,
                              adam, s)
    assert rc == -1 and b'segments' in lib.tmf_last_error()
    rc = lib.tmf_adam_fresh_rows_f32(tm.lib.ptr(x), tm.lib.ptr(x), 8, 5000, adam, s)              # rank out of range
    assert rc == -1
    assert lib.tmf_padded_ld(5000) == 0 and lib.tmf_wmrb_user_pass_fits(100, 8) == 1
    assert lib.tmf_wmrb_user_pass_fits(40000, 8) == 0 and lib.tmf_wmrb_user_pass_fits(100, 5000) == 0
    d = torch.zeros(8, 40000, device='cuda')
    rp = torch.zeros(9, dtype=torch.int64, device='cuda')
    Ri = torch.zeros(8, 40000, dtype=torch.int32, device='cuda')
    rc = lib.tmf_wmrb_user_pass_f32(tm.lib.ptr(rp), None, None, tm.lib.ptr(Ri), 8, 40000, 1.0, tm.lib.ptr(x), tm.lib.ptr(x),
                                    tm.lib.ptr(x), None, tm.lib.ptr(d), None, None, 8, 0, adam, s)
    assert rc == -3 and b'sliced' in lib.tmf_last_error()                                         # scores do not fit LDS
    with pytest.raises(tm.lib.EngineError):
        tm.lib.check(rc, lib)
    torch.cuda.synchronize()


def test_example_script_runs_both_branches(tm, capsys):
    """examples/toydata.py = the reference's toy benchmark call sequence through the `teamoflow` alias package."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('toydata', os.path.join(os.path.dirname(GOLDEN), '..', 'examples', 'toydata.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    np.random.seed(5)
    m = mod.main('mse', 100, 50, 5)
    assert len(m.loss_history_) == 450 and m.loss_history_[-1] < m.loss_history_[0]
    w = mod.main('wmrb', 60, 40, 3)
    assert len(w.loss_history_) == 100 and tuple(w.random_ind.shape) == (60, 20)
    out = capsys.readouterr().out
    assert 'Epoch 450 Complete | Loss' in out and 'Recall @ 10 w/ WMRB' in out and 'NDCG @ 10' in out


def test_wmrb_without_sample_table_raises(tm):
    model = tm.MF(3, loss_graph=tm.WMRB(), n_users=5, n_items=6)
    with pytest.raises(AttributeError):
        model.fit(1, tm.eye(5), tm.eye(6), tm.Sparse(np.array([[0, 1]]), np.array([1.0]), (5, 6)))


def test_runs_are_bit_reproducible(tm, golden):
    g = golden('wmrb_small')
    a = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 5, 0.1, 'wmrb', g['R'], 100, 50)
    b = fit_model(tm, g['U0'], g['V0'], g['indices'], g['values'], g['A'].shape, 5, 0.1, 'wmrb', g['R'], 100, 50)
    assert torch.equal(a.user_embedding, b.user_embedding) and torch.equal(a.item_embedding, b.item_embedding)
    assert a.loss_history_ == b.loss_history_


@pytest.mark.parametrize('seed', range(int(os.environ.get('TMF_FUZZ_SEEDS', '12'))))
def test_randomized_fused_topk(tm, seed):
    """Random (users, items, rank, k) incl. ranks that take every staging mode (K % 4 != 0, K % 32 != 0), small-integer
    factors (many exact ties across tiles), clamping, fp32 and bf16: fused top-k == stable top-k of the materialised scores."""
    from oracle import sparse_ref as S
    rng = np.random.default_rng(7000 + seed)
    m, n = int(rng.integers(1, 400)), int(rng.integers(1, 3000))
    r = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 31, 32, 33, 48, 64, 65, 96, 100, 127, 128, 129, 200, 256]))
    k = int(min(n, rng.choice([1, 2, 5, 10, 17, 32, 50, 64])))
    span = int(rng.choice([1, 2, 4]))
    U = rng.integers(-span, span + 1, (m, r)).astype(np.float32)
    V = rng.integers(-span, span + 1, (n, r)).astype(np.float32)
    sc = U @ V.T                                        # exact in fp32 and in bf16 products / fp32 sums
    clamp = bool(rng.integers(0, 2))
    ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
    got = tm.ops.predict_topk(torch.tensor(U), torch.tensor(V), k, clamp_negatives=clamp).cpu().numpy()
    assert np.array_equal(got, ref), (m, n, r, k, clamp, 'f32')
    if k <= tm.ops.FUSED_MAX_K:   # bf16 tables: the bf16 kernel up to k = 32, beyond it the fp32 kernel on exact fp32 copies (_ops.predict_topk)
        gb = tm.ops.predict_topk(torch.tensor(U).to(torch.bfloat16).cuda(), torch.tensor(V).to(torch.bfloat16).cuda(), k,
                                 clamp_negatives=clamp).cpu().numpy()
        assert np.array_equal(gb, ref), (m, n, r, k, clamp, 'bf16')


def test_bf16_tables_beyond_k32_windowed_when_the_fp32_copy_does_not_fit(tm, monkeypatch):
    """32 < k <= 64 on bf16 tables ranks exact fp32 copies of the rows - with no users that raised IndexError, and
    the fp32 copy of the whole item table had no out-of-memory path.  Now: empty in -> empty out; when the copy does not fit the
    catalog is ranked in windows and the lists merged - same ids, ties in catalog order included."""
    from oracle import sparse_ref as S
    rng = np.random.default_rng(99)
    m, n, r, k = 70, 1900, 40, 48
    U = rng.integers(-2, 3, (m, r)).astype(np.float32)
    V = rng.integers(-2, 3, (n, r)).astype(np.float32)           # small integers: many exact ties, also across windows
    Ub, Vb = torch.tensor(U).to(torch.bfloat16).cuda(), torch.tensor(V).to(torch.bfloat16).cuda()
    for clamp in (False, True):
        sc = U @ V.T
        ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
        whole = tm.ops.predict_topk(Ub, Vb, k, clamp_negatives=clamp)

        def no_room(B):
            raise torch.OutOfMemoryError('forced by the test')
        with monkeypatch.context() as mp:
            mp.setattr(tm.ops, '_upcast_table', no_room)
            vals, windowed = tm.ops.predict_topk(Ub, Vb, k, clamp_negatives=clamp, return_values=True)
        assert np.array_equal(whole.cpu().numpy(), ref) and np.array_equal(windowed.cpu().numpy(), ref), clamp
        want = np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, axis=1)
        assert np.array_equal(vals.cpu().numpy(), want)
    empty = tm.ops.predict_topk(Ub[:0], Vb, k)
    assert tuple(empty.shape) == (0, k) and empty.dtype == torch.int32
    v0, i0 = tm.ops.predict_topk(Ub[:0], Vb, k, return_values=True)
    assert tuple(v0.shape) == (0, k) and tuple(i0.shape) == (0, k)
