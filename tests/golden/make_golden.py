"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference itself cannot be imported here (it needs TensorFlow, which is not in the image -
SURVEY.md §8c), so the expected outputs come from ``oracle.dense_ref`` (the dense-faithful,
autograd-differentiated restatement).  The one vector that comes from the reference's own tests
is ``gather_known_answer`` (/root/reference/test/test_utils.py:47-57, also the docstring example
at src/teamoflow/mf/utils.py:68-84) - it is DATA copied from that test, not code.

Inputs use the reference's own generator recipe (``utils.generate_random_interaction`` /
``utils.random_sampler`` restated in ``oracle.datagen``) under a fixed ``np.random.seed``.
Initial weights come from ``oracle.datagen.{normal,uniform}_init`` (TF's RNG stream cannot be
reproduced) and are stored in the fixtures.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))

from oracle import datagen as G  # noqa: E402
from oracle import dense_ref as D  # noqa: E402
from oracle import sparse_ref as S  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **kw):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **kw)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def gather_known_answer():
    save('gather_known_answer',
         input=np.array([[1, 4, 2], [5, 7, 8], [6, 2, 1]], dtype=np.float32),
         index=np.array([[0, 2, 0], [2, 2, 2], [2, 1, 0]], dtype=np.int64),
         expected=np.array([[1, 2, 1], [8, 8, 8], [1, 2, 6]], dtype=np.float32))


def c1_inputs():
    """BASELINE config 1: 100x50, density .05, r=5, MSE, lr 1e-2 (examples/benchmark_toydata.py:40 call order)."""
    np.random.seed(0)
    idx, val, shape, A = G.generate_random_interaction(100, 50, density=0.05)
    return idx, val, A, G.normal_init(100, 5, 101), G.normal_init(50, 5, 102)


def c1_mse():
    idx, val, A, U0, V0 = c1_inputs()
    epochs = 450
    out = D.fit_dense(U0, V0, idx, val, 'mse', epochs, 1e-2, record_epochs=(1, 2, 25, epochs))
    U, V = out['U'], out['V']
    kw = dict(indices=idx, values=val, A=A, U0=U0, V0=V0, lr=1e-2, loss=out['loss'],
              predictions=D.predict_dense(U, V), top10=D.retrieve_user_recs_dense(U, V, k=10),
              recall10=D.recall_at_k_dense(U, V, A, 10),
              recall10_rows=D.recall_at_k_dense(U, V, A, 10, preserve_rows=True),
              precision10=D.precision_at_k_dense(U, V, A, 10))
    for e, (u, v) in out['snapshots'].items():
        kw[f'U_{e}'], kw[f'V_{e}'] = u, v
    save('c1_mse', **kw)


def wmrb_small_inputs():
    """Shape of the reference's own WMRB smoke test (test/test_loss.py:14-21,50-57):
    50x100, density .05, r=3, S = n_items // 2, lr .1, 25 epochs."""
    np.random.seed(1)
    idx, val, shape, A = G.generate_random_interaction(50, 100, density=0.05)
    R = G.random_sampler(100, 50, 50)
    return idx, val, A, R, G.uniform_init(50, 3, 201), G.uniform_init(100, 3, 202)


def wmrb_small():
    idx, val, A, R, U0, V0 = wmrb_small_inputs()
    out = D.fit_dense(U0, V0, idx, val, 'wmrb', 25, 0.1, random_ind=R, n_items=100, n_samples=50,
                      record_epochs=(1, 2, 25))
    t = S.wmrb_terms(U0, V0, idx, val, R, 100, 50)
    kw = dict(indices=idx, values=val, A=A, R=R, U0=U0, V0=V0, lr=0.1, n_items=100, n_samples=50,
              loss=out['loss'], D_first=t['D'], delta_first=t['delta'], M_first=t['M'],
              recall10=D.recall_at_k_dense(out['U'], out['V'], A, 10),
              top10=D.retrieve_user_recs_dense(out['U'], out['V'], k=10))
    for e, (u, v) in out['snapshots'].items():
        kw[f'U_{e}'], kw[f'V_{e}'] = u, v
    save('wmrb_small', **kw)


def wmrb_mixed_inputs():
    """Mixed-sign interactions (test/test_loss.py:20-21 recipe: min_val=-5): non-positive entries
    must contribute nothing to WMRB but count as hits in recall_at_k."""
    np.random.seed(2)
    idx, val, shape, A = G.generate_random_interaction(40, 60, min_val=-5.0, max_val=5.0, density=0.08)
    R = G.random_sampler(60, 40, 12)
    return idx, val, A, R, G.normal_init(40, 8, 301), G.normal_init(60, 8, 302)


def wmrb_mixed():
    idx, val, A, R, U0, V0 = wmrb_mixed_inputs()
    out = D.fit_dense(U0, V0, idx, val, 'wmrb', 10, 0.05, random_ind=R, n_items=60, n_samples=12,
                      record_epochs=(1, 10))
    outm = D.fit_dense(U0, V0, idx, val, 'mse', 10, 0.05, record_epochs=(1, 10))
    kw = dict(indices=idx, values=val, A=A, R=R, U0=U0, V0=V0, lr=0.05, n_items=60, n_samples=12,
              loss=out['loss'], loss_mse=outm['loss'],
              recall10=D.recall_at_k_dense(out['U'], out['V'], A, 10),
              recall10_rows=D.recall_at_k_dense(out['U'], out['V'], A, 10, preserve_rows=True))
    for e, (u, v) in out['snapshots'].items():
        kw[f'U_{e}'], kw[f'V_{e}'] = u, v
    for e, (u, v) in outm['snapshots'].items():
        kw[f'Um_{e}'], kw[f'Vm_{e}'] = u, v
    save('wmrb_mixed', **kw)


def c2_inputs():
    """BASELINE config 2: 943x1682 (MovieLens-100K shape), r=32, MSE, lr 1e-3, target nnz 100,000
    (density = target / (0.9 m n): the recipe drops the ~10% of entries that round to 0)."""
    m, n = 943, 1682
    np.random.seed(0)
    idx, val, shape, A = G.generate_random_interaction(m, n, density=100000 / (0.9 * m * n))
    return idx, val, A, G.normal_init(m, 32, 401), G.normal_init(n, 32, 402)


def c2_mse():
    idx, val, A, U0, V0 = c2_inputs()
    out = D.fit_dense(U0, V0, idx, val, 'mse', 100, 1e-3, record_epochs=(1, 100))
    U1, V1 = out['snapshots'][1]
    U, V = out['U'], out['V']
    save('c2_mse', nnz=len(val), indices_sha=sha(idx), values_sha=sha(val), U0_sha=sha(U0), V0_sha=sha(V0),
         lr=1e-3, loss=out['loss'], U_1=U1, V_1=V1, U_100_head=U[:64], V_100_head=V[:64],
         recall10_mean=float(D.recall_at_k_dense(U, V, A, 10).mean()),
         top10_head=D.retrieve_user_recs_dense(U, V, k=10)[:64])


def c3r_inputs():
    """Reduced BASELINE config 3 (1/10 linear scale of 6040x3706, r=64, WMRB, S = n // 2, lr .1)."""
    m, n = 604, 371
    np.random.seed(3)
    idx, val, shape, A = G.generate_random_interaction(m, n, density=10000 / (0.9 * m * n))
    R = G.random_sampler(n, m, n // 2)
    return idx, val, A, R, G.uniform_init(m, 64, 501), G.uniform_init(n, 64, 502)


def c3r_wmrb():
    idx, val, A, R, U0, V0 = c3r_inputs()
    m, n = A.shape
    out = D.fit_dense(U0, V0, idx, val, 'wmrb', 20, 0.1, random_ind=R, n_items=n, n_samples=n // 2,
                      record_epochs=(1, 20))
    U1, V1 = out['snapshots'][1]
    save('c3r_wmrb', nnz=len(val), indices_sha=sha(idx), values_sha=sha(val), R_sha=sha(R), U0_sha=sha(U0),
         V0_sha=sha(V0), lr=0.1, loss=out['loss'], U_1=U1, V_1=V1,
         recall10_mean=float(D.recall_at_k_dense(out['U'], out['V'], A, 10).mean()))


def plugin_inputs(seed=7, m=40, n=30, r=6):
    """Small mixed-sign problem with dense NON-identity features for the plug-ins beyond Linear + MSE / WMRB:
    BiasedLinearEmbedding / ReLUEmbedding (embedding_graphs.py:41-87) and KLDivergenceLoss (loss_graphs.py:91-122)."""
    rng = np.random.default_rng(seed)
    A = ((rng.random((m, n)) < 0.3) * rng.integers(-2, 6, (m, n))).astype(np.float32)
    idx, val = np.argwhere(A != 0).astype(np.int64), A[A != 0]
    return dict(A=A, indices=idx, values=val,
                U0=(rng.standard_normal((m, r)) * 0.3).astype(np.float32), V0=(rng.standard_normal((n, r)) * 0.3).astype(np.float32),
                Fu=(np.eye(m) + 0.05 * rng.random((m, m))).astype(np.float32), Fv=(np.eye(n) + 0.05 * rng.random((n, n))).astype(np.float32),
                U0_relu=(rng.standard_normal((5 * r, r)) * 0.3).astype(np.float32),        # ReLUEmbedding: weights are [aux_dim = 5 r, r]
                relu_w0=(rng.standard_normal((m, 5 * r)) * 0.2).astype(np.float32))        # the matrix TF would draw at first use


PLUGIN_CASES = {'plugin_biased': dict(loss='mse', user_embedding='biased', item_embedding='biased'),
                'plugin_relu': dict(loss='mse', user_embedding='relu'),
                'plugin_kl': dict(loss='kl')}
PLUGIN_EPOCHS, PLUGIN_LR = 12, 0.02


def plugin_case(name):
    p = plugin_inputs()
    kw = dict(PLUGIN_CASES[name])
    relu = kw.get('user_embedding') == 'relu'
    out = D.fit_dense_plugins(p['U0_relu'] if relu else p['U0'], p['V0'], p['indices'], p['values'], kw.pop('loss'), PLUGIN_EPOCHS,
                              PLUGIN_LR, p['Fu'], p['Fv'], user_relu_weight0=p['relu_w0'] if relu else None, record_epochs=(1,), **kw)
    fx = dict(p, lr=PLUGIN_LR, epochs=PLUGIN_EPOCHS, loss=out['loss'], user_embedding=out['user_embedding'],
              item_embedding=out['item_embedding'])
    for side, (first, last) in dict(user=(out['snapshots'][1][0], out['user_vars']), item=(out['snapshots'][1][1], out['item_vars'])).items():
        for i, (a, b) in enumerate(zip(first, last)):
            fx[f'{side}_var{i}_1'], fx[f'{side}_var{i}_{PLUGIN_EPOCHS}'] = a, b
    return fx


def plugins():
    for name in PLUGIN_CASES:
        save(name, **plugin_case(name))


if __name__ == '__main__':
    plugins()
    gather_known_answer()
    c1_mse()
    wmrb_small()
    wmrb_mixed()
    c2_mse()
    c3r_wmrb()
