"""CPU tests of the oracle itself: the two independent restatements agree, the reference's own
known-answer vector holds, and the committed fixtures are what the oracle produces today."""
import hashlib
import importlib.util
import os

import numpy as np
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
