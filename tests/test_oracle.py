"""CPU tests of the oracle itself: the two independent restatements agree, the reference's own
known-answer vector holds, and the committed fixtures are what the oracle produces today."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_step, rel_err
from oracle import datagen as G
from oracle import dense_ref as D
from oracle import sparse_ref as S

spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_gather_matrix_indices_reference_known_answer(golden):
    g = golden('gather_known_answer')  # /root/reference/test/test_utils.py:47-57
    out = D.gather_matrix_indices(torch.tensor(g['input']), torch.tensor(g['index']))
    assert np.array_equal(out.numpy(), g['expected'])


def test_tf_top_k_tie_rule():
    v, i = D.tf_top_k(torch.tensor([0., 1, 1, 0, 1]), 3)
    assert i.tolist() == [1, 2, 4]
    assert S.topk_stable(np.array([[0., 1, 1, 0, 1]]), 3).tolist() == [[1, 2, 4]]
    v, i = D.tf_top_k(torch.zeros(2, 7), 4)
    assert i.tolist() == [[0, 1, 2, 3]] * 2


def test_tf_maximum_subgradient_at_tie():
    x = torch.tensor([-1.0, 0.0, 2.0], requires_grad=True)
    D.tf_maximum(x, 0.0).sum().backward()
    assert x.grad.tolist() == [0.0, 1.0, 1.0]


def test_adam_fresh_is_near_sign():
    w = torch.zeros(4)
    g = torch.tensor([1.0, -3.0, 0.0, 1e-6])
    out = D.adam_fresh_step(w, g, 0.01).numpy()
    ref = -0.01 * g.numpy() / (np.abs(g.numpy()) + 3.1623e-6)
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-9)
    assert out[2] == 0.0
    assert np.array_equal(out, S.adam_fresh(np.zeros(4, np.float32), g.numpy(), 0.01))


def test_generators_reproducible_from_seed(golden):
    idx, val, A, U0, V0 = MG.c1_inputs()
    g = golden('c1_mse')
    assert np.array_equal(idx, g['indices']) and np.array_equal(val, g['values'])
    assert len(val) == 225 and set(val.tolist()) <= {1., 2., 3., 4., 5.}
    # row-major sorted unique pairs, like tf.sparse.SparseTensor built from scipy CSR
    keys = idx[:, 0] * 50 + idx[:, 1]
    assert np.all(np.diff(keys) > 0)
    R = MG.wmrb_small_inputs()[3]
    assert R.shape == (50, 50) and all(len(set(r)) == 50 for r in R.tolist())


def test_dense_vs_sparse_mse(golden):
    g = golden('c1_mse')
    s = S.fit_sparse(g['U0'], g['V0'], g['indices'], g['values'], 'mse', 25, float(g['lr']), record_epochs=(1, 25))
    assert rel_err(s['loss'], g['loss'][:25]) < 1e-6
    assert rel_err(s['snapshots'][1][0], g['U_1']) < 1e-6 and rel_err(s['snapshots'][1][1], g['V_1']) < 1e-6
    assert rel_err(s['snapshots'][25][0], g['U_25']) < 1e-4


def test_dense_vs_sparse_wmrb(golden):
    for name in ('wmrb_small', 'wmrb_mixed'):
        g = golden(name)
        s = S.fit_sparse(g['U0'], g['V0'], g['indices'], g['values'], 'wmrb', 2, float(g['lr']), random_ind=g['R'],
                         n_items=int(g['n_items']), n_samples=int(g['n_samples']), record_epochs=(1,))
        assert rel_err(s['loss'], g['loss'][:2]) < 1e-5, name
        assert rel_err(s['snapshots'][1][0], g['U_1']) < 1e-5, name
        assert rel_err(s['snapshots'][1][1], g['V_1']) < 1e-5, name


def test_wmrb_ignores_non_positive_entries(golden):
    g = golden('wmrb_mixed')
    assert (g['values'] <= 0).any() and (g['values'] > 0).any()
    keep = g['values'] > 0
    a = S.wmrb_epoch(g['U0'], g['V0'], g['indices'], g['values'], g['R'], 60, 12, 0.05)
    b = S.wmrb_epoch(g['U0'], g['V0'], g['indices'][keep], g['values'][keep], g['R'], 60, 12, 0.05)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_fixture_c1_is_current_oracle_output(golden):
    g = golden('c1_mse')
    out = D.fit_dense(g['U0'], g['V0'], g['indices'], g['values'], 'mse', 25, float(g['lr']), record_epochs=(25,))
    assert rel_err(out['loss'], g['loss'][:25]) < 1e-6
    assert rel_err(out['snapshots'][25][0], g['U_25']) < 1e-5


def test_fixture_c2_inputs_regenerate(golden):
    g = golden('c2_mse')
    idx, val, A, U0, V0 = MG.c2_inputs()
    assert len(val) == int(g['nnz']) and 95000 < len(val) < 105000
    assert sha(idx) == str(g['indices_sha']) and sha(val) == str(g['values_sha'])
    assert sha(U0) == str(g['U0_sha']) and sha(V0) == str(g['V0_sha'])
    s = S.mse_epoch(U0, V0, idx, val, float(g['lr']))
    assert abs(s[2] - g['loss'][0]) / g['loss'][0] < 1e-6
    s64 = S.mse_epoch(U0.astype(np.float64), V0.astype(np.float64), idx, val.astype(np.float64), float(g['lr']))
    for got in (s[0], g['U_1']):
        assert_step(got, U0, s64[3]['gU'], float(g['lr']))
    for got in (s[1], g['V_1']):
        assert_step(got, V0, s64[3]['gV'], float(g['lr']))


def test_recall_dense_vs_sparse(golden):
    for name in ('c1_mse', 'wmrb_mixed'):
        g = golden(name)
        e = 450 if name == 'c1_mse' else 10
        U, V = g[f'U_{e}'], g[f'V_{e}']
        assert np.array_equal(D.recall_at_k_dense(U, V, g['A'], 10), g['recall10'])
        assert np.array_equal(S.recall_at_k_sparse(U, V, g['indices'], g['values'], 10), g['recall10'])
        assert np.array_equal(S.recall_at_k_sparse(U, V, g['indices'], g['values'], 10, True), g['recall10_rows'])


# ---- the C / OpenMP restatement (oracle/sparse_ref.c): third independently written oracle + CPU baseline ----
def _c_oracle():
    from oracle import sparse_c
    sparse_c.build()
    return sparse_c


def test_c_oracle_matches_golden_fixtures(golden):
    """Loss trajectories and first-epoch tables of the dense-oracle fixtures, from the C restatement."""
    C = _c_oracle()
    g = golden('c1_mse')
    m, n = g['A'].shape
    plan = C.Plan(g['indices'], g['values'], m, n)
    U, V, losses = g['U0'], g['V0'], []
    for epoch in range(25):
        U, V, mean, t = C.mse_epoch(U, V, plan, float(g['lr']))
        losses.append(mean)
        if epoch == 0:
            ref = S.mse_epoch(g['U0'].astype(np.float64), g['V0'].astype(np.float64), g['indices'], g['values'].astype(np.float64),
                              float(g['lr']))[3]
            assert_step(U, g['U0'], ref['gU'], float(g['lr']))
            assert_step(V, g['V0'], ref['gV'], float(g['lr']))
            assert_step(g['U_1'], g['U0'], ref['gU'], float(g['lr']))   # the fixture itself lies in the same interval
    assert rel_err(losses, g['loss'][:25]) < 1e-6
    for name in ('wmrb_small', 'wmrb_mixed'):
        g = golden(name)
        m, n = g['A'].shape
        S_ = int(g['n_samples'])
        plan = C.Plan(g['indices'], g['values'], m, n, g['R'])
        U1, V1, mean, t = C.wmrb_epoch(g['U0'], g['V0'], plan, n, S_, float(g['lr']))
        ref = S.wmrb_epoch(g['U0'].astype(np.float64), g['V0'].astype(np.float64), g['indices'], g['values'].astype(np.float64),
                           g['R'], n, S_, float(g['lr']))[3]
        assert abs(mean - float(g['loss'][0])) <= 1e-6 * abs(float(g['loss'][0])), name
        assert rel_err(t['D'], ref['D']) < 1e-6 and rel_err(t['gU'], ref['gU']) < 1e-6 and rel_err(t['gV'], ref['gV']) < 1e-6
        if 'D_first' in g:
            assert rel_err(t['D'], g['D_first']) < 1e-6
        assert_step(U1, g['U0'], ref['gU'], float(g['lr']))
        assert_step(V1, g['V0'], ref['gV'], float(g['lr']))
        U2, V2, mean2, _ = C.wmrb_epoch(U1, V1, plan, n, S_, float(g['lr']))
        assert abs(mean2 - float(g['loss'][1])) <= 1e-5 * abs(float(g['loss'][1])), name


def test_c_oracle_random_shapes_and_thread_counts():
    """Ragged / empty rows, shuffled input order, non-positive values; the result does not depend on the
    number of OpenMP threads (every row is reduced by one thread in a fixed order)."""
    C = _c_oracle()
    rng = np.random.default_rng(5)
    for trial in range(6):
        m, n, r = int(rng.integers(1, 70)), int(rng.integers(2, 60)), int(rng.choice([1, 3, 8, 17, 64]))
        S_ = int(rng.integers(1, n + 1))
        A = (rng.random((m, n)) < rng.choice([0.0, 0.05, 0.4])) * rng.integers(-2, 6, (m, n))
        idx, val = np.argwhere(A != 0), A[A != 0].astype(np.float32)
        p = rng.permutation(len(val))
        idx, val = idx[p], val[p]
        U = (rng.standard_normal((m, r)) * 0.4).astype(np.float32)
        V = (rng.standard_normal((n, r)) * 0.4).astype(np.float32)
        R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
        plan = C.Plan(idx, val, m, n, R)
        outs = []
        for nt in (1, 3):
            C.set_threads(nt)
            outs.append((C.mse_epoch(U, V, plan, 0.01), C.wmrb_epoch(U, V, plan, n, S_, 0.1)))
        C.set_threads(os.cpu_count() or 1)
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.array_equal(a[3]['delta'], b[3]['delta'])
        (mU, mV, mmean, mt), (wU, wV, wmean, wt) = outs[0]
        U64, V64, v64 = U.astype(np.float64), V.astype(np.float64), val.astype(np.float64)
        if len(val):
            ref = S.mse_epoch(U64, V64, idx, v64, 0.01)
            assert abs(mmean - ref[2]) <= 1e-6 * abs(ref[2])
            assert rel_err(mt['delta'], ref[3]['delta']) < 1e-6
            assert_step(mU, U, ref[3]['gU'], 0.01)
            assert_step(mV, V, ref[3]['gV'], 0.01)
        else:
            assert np.array_equal(mU, U) and np.array_equal(mV, V)
        ref = S.wmrb_epoch(U64, V64, idx, v64, R, n, S_, 0.1)
        if (val > 0).any():
            assert abs(wmean - ref[2]) <= 1e-6 * abs(ref[2])
            assert wt['n_pos'] == int((val > 0).sum())
            assert rel_err(wt['D'], ref[3]['D']) < 1e-5
            assert_step(wU, U, ref[3]['gU'], 0.1)
            assert_step(wV, V, ref[3]['gV'], 0.1)
        else:
            assert np.array_equal(wU, U) and np.array_equal(wV, V)


def test_boundary_slack_c_equals_numpy_and_is_zero_away_from_the_kink(golden):
    """oracle_wmrb_boundary_slack (C) == sparse_ref.wmrb_slack (NumPy) on a trained state; no boundary terms at the
    start of training (every hinge argument is ~1) and only a small share of the terms later."""
    from oracle import sparse_c as C
    g = golden('wmrb_mixed')
    n, S_ = int(g['n_items']), int(g['n_samples'])
    idx, val, R = g['indices'], g['values'], g['R']
    plan = C.Plan(idx, val, g['A'].shape[0], n, R)
    assert C.wmrb_boundary_slack(g['U0'], g['V0'], plan, n, S_)['pairs'] == 0
    for tol in (1e-5, 3e-2):   # 3e-2: enough boundary terms on this small problem to compare every output
        U, V = g['U_10'], g['V_10']
        a = C.wmrb_boundary_slack(U, V, plan, n, S_, tol_rel=tol)
        b = S.wmrb_slack(U.astype(np.float64), V.astype(np.float64), idx, val.astype(np.float64), R, n, S_, tol_rel=tol)
        pos = val > 0
        assert abs(a['pairs'] - b['pairs']) <= max(2, 0.02 * b['pairs'])     # fp32 vs fp64 arguments right at the tolerance edge
        if tol > 1e-3:
            assert b['pairs'] > 20 and b['pairs'] < 0.2 * pos.sum() * S_
            assert rel_err(a['D'], b['D']) < 0.1 and rel_err(a['gU'], b['gU']) < 0.1 and rel_err(a['gV'], b['gV']) < 0.1
            assert rel_err(a['delta'][pos], b['delta']) < 0.1


def _loss_sum_by_definition(U, V, idx, val, loss, R=None, n_items=None, n_samples=None):
    """Sum of the per-interaction losses written out from the definitions (loss_graphs.py:47-52 / :74-88) with plain loops -
    no closed-form gradient, no vectorised helper of the oracle involved."""
    import math
    total = 0.0
    for (u, j), a in zip(idx.tolist(), val.tolist()):
        p = float(np.dot(U[u], V[j]))
        if loss == 'mse':
            total += (a - p) ** 2
        elif a > 0:
            hinge = sum(max(0.0, 1.0 - p + float(np.dot(U[u], V[s]))) for s in R[u].tolist())
            total += math.log(1.0 + (n_items / n_samples) * hinge)
    return total


@pytest.mark.parametrize('loss', ['mse', 'wmrb'])
def test_closed_form_gradients_match_finite_differences(loss):
    """A line of evidence that does not pass through autograd or through the hand derivation: central differences of the loss
    written out from its definition against the closed-form gradients the sparse oracle (and with it every kernel test) uses.
    What `tape.gradient` differentiates is the SUM of the per-interaction losses (matrix_factorization.py:170-171)."""
    rng = np.random.default_rng(5)
    m, n, r, Sn = 9, 11, 4, 5
    A = (rng.random((m, n)) < 0.4) * rng.integers(1, 6, (m, n))
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float64)
    val[::5] = -1.0 if loss == 'mse' else 0.0          # MSE fits negative values too; WMRB skips stored non-positives
    U = rng.standard_normal((m, r)) * 0.4
    V = rng.standard_normal((n, r)) * 0.4
    R = np.stack([rng.choice(n, Sn, replace=False) for _ in range(m)])
    if loss == 'mse':
        t = S.mse_epoch(U, V, idx, val, 0.01)[3]
    else:
        t = S.wmrb_epoch(U, V, idx, val, R, n, Sn, 0.01)[3]
        x = 1.0 - t['p'][:, None] + t['sp'][idx[t['pos'], 0]]
        assert np.abs(x).min() > 1e-4                   # no hinge argument near the kink: the loss is smooth around (U, V)
    h = 1e-6
    for W, g, name in ((U, t['gU'], 'U'), (V, t['gV'], 'V')):
        for i, c in zip(rng.integers(0, W.shape[0], 12), rng.integers(0, r, 12)):
            keep = W[i, c]
            W[i, c] = keep + h
            up = _loss_sum_by_definition(U, V, idx, val, loss, R, n, Sn)
            W[i, c] = keep - h
            dn = _loss_sum_by_definition(U, V, idx, val, loss, R, n, Sn)
            W[i, c] = keep
            fd = (up - dn) / (2 * h)
            assert abs(fd - g[i, c]) <= 1e-6 * max(1.0, np.abs(g).max()), (name, i, c, fd, g[i, c])


# ---------------------------------------------------------------------------------------------------------------------
# the plug-ins beyond Linear + MSE / WMRB: oracle.dense_ref.fit_dense_plugins
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', sorted(MG.PLUGIN_CASES))
def test_plugin_fixtures_are_current_oracle_output(golden, name):
    g, now = golden(name), MG.plugin_case(name)
    assert set(g) == set(now)
    for k in g:
        assert np.array_equal(g[k], np.asarray(now[k])), k
    assert np.isfinite(g['loss']).all() and g['loss'][-1] < g['loss'][0]


def test_tfp_normal_cdf_restatement_against_scipy():
    from scipy.special import ndtr
    z = np.concatenate([np.linspace(-9, 9, 181), [-0.70710678, 0.70710678, 0.0]])
    assert np.abs(D.tfp_ndtr(torch.tensor(z)).numpy() - ndtr(z)).max() < 1e-15
    assert rel_err(D.tfp_ndtr(torch.tensor(z, dtype=torch.float32)).numpy(), ndtr(z)) < 1e-6


def _plugin_loss_by_definition(case, Fu, Fv, idx, val, uv, iv):
    """Sum of the loss of one epoch written out from the definitions (embedding_graphs.py:38 / :58 / :85-87, loss_graphs.py:52 /
    :109-122) in plain NumPy with SciPy's normal CDF - neither torch nor any helper of the oracle."""
    from scipy.special import ndtr

    def emb(kind, F, vs):
        if kind == 'linear':
            return F @ vs[0]
        if kind == 'biased':
            return F @ vs[0] + vs[1]
        return np.maximum(F @ vs[1] + vs[2], 0.0) @ vs[0]
    P = emb(case.get('user_embedding', 'linear'), Fu, uv) @ emb(case.get('item_embedding', 'linear'), Fv, iv).T
    p = P[idx[:, 0], idx[:, 1]]
    if case['loss'] == 'mse':
        return float(((val - p) ** 2).sum())
    pos, neg = p[val > 0], p[val <= 0]
    loc = neg.mean() - pos.mean()
    scale = np.sqrt(((pos - pos.mean()) ** 2).mean() + ((neg - neg.mean()) ** 2).mean())
    return float(1.0 - ndtr((0.0 - loc) / scale))


@pytest.mark.parametrize('name', sorted(MG.PLUGIN_CASES))
def test_plugin_oracle_step_matches_finite_differences(name):
    """The plug-in oracle differentiates by autograd; this line of evidence does not: central differences (fp64) of the epoch's
    loss written out from its definition give the gradient of EVERY trainable, and the oracle's first fresh-Adam step (fp64 run)
    must lie in the step interval of that gradient."""
    case = dict(MG.PLUGIN_CASES[name])
    rng = np.random.default_rng(11)
    m, n, r = 9, 7, 2
    A = ((rng.random((m, n)) < 0.5) * rng.integers(-2, 6, (m, n))).astype(np.float64)
    idx, val = np.argwhere(A != 0), A[A != 0]
    Fu, Fv = np.eye(m) + 0.1 * rng.random((m, m)), np.eye(n) + 0.1 * rng.random((n, n))
    relu = case.get('user_embedding') == 'relu'
    U0 = rng.standard_normal((5 * r if relu else m, r)) * 0.4
    V0 = rng.standard_normal((n, r)) * 0.4
    relu_w0 = rng.standard_normal((m, 5 * r)) * 0.3 if relu else None
    lr = 0.05
    out = D.fit_dense_plugins(U0, V0, idx, val, case['loss'], 1, lr, Fu, Fv, user_embedding=case.get('user_embedding', 'linear'),
                              item_embedding=case.get('item_embedding', 'linear'), user_relu_weight0=relu_w0, dtype=torch.float64)
    uv = [U0] + ([np.zeros((1, r))] if case.get('user_embedding') == 'biased' else []) + ([relu_w0, np.zeros((1, 5 * r))] if relu else [])
    iv = [V0] + ([np.zeros((1, r))] if case.get('item_embedding') == 'biased' else [])
    assert rel_err(out['loss'][0] * (len(val) if case['loss'] == 'mse' else 1), _plugin_loss_by_definition(case, Fu, Fv, idx, val, uv, iv)) < 1e-12
    if relu:   # away from the kink of the ReLU, so that the loss is smooth where it is differenced
        assert np.abs(Fu @ relu_w0).min() > 1e-4
    h = 1e-6
    for vs, new in ((uv, out['user_vars']), (iv, out['item_vars'])):
        for W, W1 in zip(vs, new):
            g = np.zeros_like(W)
            for pos in np.ndindex(*W.shape):
                keep = W[pos]
                W[pos] = keep + h
                up = _plugin_loss_by_definition(case, Fu, Fv, idx, val, uv, iv)
                W[pos] = keep - h
                dn = _plugin_loss_by_definition(case, Fu, Fv, idx, val, uv, iv)
                W[pos] = keep
                g[pos] = (up - dn) / (2 * h)
            assert np.abs(g).max() > 0
            assert_step(W1, W, g, lr, rtol=1e-5, what=name)
