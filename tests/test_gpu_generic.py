"""The generic plug-in path (dense non-identity features, BiasedLinearEmbedding, ReLUEmbedding,
KLDivergenceLoss - /root/reference/src/teamoflow/mf/embedding_graphs.py:41-87, loss_graphs.py:91-122) ON THE MI355X:
`_fit_generic` runs the reference's dense loop with torch ops on the device, predict / ranking then go through the HIP
kernels.  Dense-feature models are checked against oracle.dense_ref.fit_dense(user_features=...), the other built-in plug-ins against oracle.dense_ref.fit_dense_plugins
through the committed fixtures tests/golden/plugin_*.npz (no comparison with a host run of the same code)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ns():
    from teamoflow_amd import _lib
    from teamoflow_amd.mf import embedding_graphs as E
    from teamoflow_amd.mf import initializer_graphs as I
    from teamoflow_amd.mf import loss_graphs as L
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions
    _lib.get()

    class NS:
        pass
    o = NS()
    o.E, o.I, o.L, o.MF, o.Sparse = E, I, L, MatrixFactorization, SparseInteractions
    return o


def problem(seed, m=40, n=30, r=6, S_=8):
    rng = np.random.default_rng(seed)
    A = ((rng.random((m, n)) < 0.3) * rng.integers(-2, 6, (m, n))).astype(np.float32)
    idx, val = np.argwhere(A != 0), A[A != 0]
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, S_, replace=False) for _ in range(m)])
    Fu = (np.eye(m) + 0.05 * rng.random((m, m))).astype(np.float32)      # dense, non-identity features
    Fv = (np.eye(n) + 0.05 * rng.random((n, n))).astype(np.float32)
    return A, idx, val, U0, V0, R, Fu, Fv


@pytest.mark.parametrize('loss', ['mse', 'wmrb'])
def test_dense_features_match_the_dense_oracle(ns, loss):
    from oracle import dense_ref as D
    A, idx, val, U0, V0, R, Fu, Fv = problem(1)
    m, n = A.shape
    kw = dict(user_weight_graph=ns.I.FixedInitializer(U0), item_weight_graph=ns.I.FixedInitializer(V0))
    if loss == 'wmrb':
        kw.update(loss_graph=ns.L.WMRBLoss(), n_users=m, n_items=n, n_samples=R.shape[1])
    model = ns.MF(U0.shape[1], **kw)
    model.verbose = False
    if loss == 'wmrb':
        model.random_ind = torch.as_tensor(R)
    model.fit(8, torch.tensor(Fu), torch.tensor(Fv), ns.Sparse(idx, val, (m, n)), lr=0.05)
    assert model.user_embedding.is_cuda and not model._on_fast_path(torch.tensor(Fu), torch.tensor(Fv))
    ref = D.fit_dense(U0, V0, idx, val, loss, 8, 0.05, random_ind=R, n_items=n, n_samples=R.shape[1], user_features=Fu,
                      item_features=Fv)
    assert rel_err(model.loss_history_[:3], ref['loss'][:3]) < 1e-5 and rel_err(model.loss_history_, ref['loss']) < 1e-3
    assert rel_err(model.user_trainable[0].detach().cpu().numpy(), ref['U']) < 0.05 * 8   # near-sign steps: lr per epoch at most
    # the embeddings the fit leaves behind are features @ weights, and ranking them runs on the HIP kernels
    want_ue = Fu @ model.user_trainable[0].detach().cpu().numpy()
    assert rel_err(model.user_embedding.detach().cpu().numpy(), want_ue) < 1e-5
    pred = model.predict().cpu().numpy()
    assert rel_err(pred, model.user_embedding.detach().cpu().numpy() @ model.item_embedding.detach().cpu().numpy().T) < 1e-5
    assert np.array_equal(model.retrieve_user_recs(k=5), np.argsort(-pred, axis=1, kind='stable')[:, :5])
    assert np.isfinite(float(model.recall_at_k(torch.tensor(A)).mean()))


@pytest.mark.parametrize('variant', ['biased', 'relu', 'kl'])
def test_other_plugins_match_the_plugin_oracle(ns, variant, golden):
    """BiasedLinearEmbedding / ReLUEmbedding (embedding_graphs.py:41-87) and KLDivergenceLoss (loss_graphs.py:91-122) trained on
    the MI355X by `_fit_generic` against oracle.dense_ref.fit_dense_plugins - an independent restatement of the reference loop
    (pinned on the CPU by central differences of the loss written from its definition, tests/test_oracle.py) - through the
    committed fixtures tests/golden/plugin_*.npz: the loss trajectory, every trainable after ONE fresh-Adam step (step interval
    of the oracle's fp64 gradient) and the embeddings the fit leaves behind."""
    from conftest import assert_step
    from oracle import dense_ref as D
    g = golden('plugin_' + variant)
    A, idx, val, V0, Fu, Fv = g['A'], g['indices'], g['values'], g['V0'], g['Fu'], g['Fv']
    m, n = A.shape
    r = V0.shape[1]
    U0 = g['U0_relu'] if variant == 'relu' else g['U0']
    lr, epochs = float(g['lr']), int(g['epochs'])
    okw = dict(user_embedding={'biased': 'biased', 'relu': 'relu'}.get(variant, 'linear'),
               item_embedding='biased' if variant == 'biased' else 'linear',
               user_relu_weight0=g['relu_w0'] if variant == 'relu' else None)
    loss = 'kl' if variant == 'kl' else 'mse'

    def model_for(n_epochs):
        kw = dict(user_weight_graph=ns.I.FixedInitializer(U0), item_weight_graph=ns.I.FixedInitializer(V0))
        if variant == 'biased':
            kw.update(user_repr_graph=ns.E.BiasedLinearEmbedding(), item_repr_graph=ns.E.BiasedLinearEmbedding())
        elif variant == 'relu':
            kw.update(user_repr_graph=ns.E.ReLUEmbedding())
        else:
            kw.update(loss_graph=ns.L.KLDivergenceLoss())
        model = ns.MF(r, **kw)
        model.verbose = False
        if variant == 'relu':   # the reference draws this matrix from TF's RNG at first use (:80-81): pinned to the fixture's
            model.user_relu_weight = torch.tensor(g['relu_w0'], device='cuda').requires_grad_(True)
        model.fit(n_epochs, torch.tensor(Fu, device='cuda'), torch.tensor(Fv, device='cuda'), ns.Sparse(idx, val, (m, n), device='cuda'), lr=lr)
        return model

    # ---- one step: every trainable inside the step interval of the oracle's fp64 gradient, and at the fp32 oracle's own step ----
    one = model_for(1)
    ref64 = D.fit_dense_plugins(U0.astype(np.float64), V0.astype(np.float64), idx, val.astype(np.float64), loss, 1, lr, Fu.astype(np.float64),
                                Fv.astype(np.float64), dtype=torch.float64, **okw)
    n_vars = {'biased': 2, 'relu': 3, 'kl': 1}[variant]
    assert len(one.user_trainable) == n_vars and len(one.item_trainable) == (2 if variant == 'biased' else 1)
    starts = dict(user=[U0] + ([np.zeros((1, r), np.float32)] if variant == 'biased' else [])
                  + ([g['relu_w0'], np.zeros((1, 5 * r), np.float32)] if variant == 'relu' else []),
                  item=[V0] + ([np.zeros((1, r), np.float32)] if variant == 'biased' else []))
    for side, got_vars, grads in (('user', one.user_trainable, ref64['first_grads'][0]), ('item', one.item_trainable, ref64['first_grads'][1])):
        for i, (w, w0, gr) in enumerate(zip(got_vars, starts[side], grads)):
            assert w.is_cuda
            assert_step(w.detach().cpu().numpy(), w0, gr, lr, what=f'{variant} {side} variable {i}')
            assert_step(g[f'{side}_var{i}_1'], w0, gr, lr, what=f'fixture {variant} {side} variable {i}')   # the fp32 oracle lies there too
    assert rel_err(one.loss_history_[0], g['loss'][0]) < 1e-5
    # ---- the whole fit: loss trajectory and what it leaves behind ----
    model = model_for(epochs)
    assert model.user_embedding.is_cuda and model.user_embedding.shape == (m, r)
    assert rel_err(model.loss_history_[:3], g['loss'][:3]) < 1e-5
    assert rel_err(model.loss_history_, g['loss']) < 1e-3       # near-sign steps amplify rounding over the epochs
    assert np.abs(model.user_embedding.detach().cpu().numpy() - g['user_embedding']).max() <= lr * epochs * 0.5
    # the HIP ranking of the device model's embeddings equals the stable ranking of their product
    pred = model.predict().cpu().numpy()
    assert np.array_equal(model.retrieve_user_recs(k=7), np.argsort(-pred, axis=1, kind='stable')[:, :7])


@pytest.mark.parametrize('loss', ['mse', 'wmrb'])
def test_opt_in_persistent_adam(ns, loss, monkeypatch):
    """optimizer='adam' (extension, off by default - the reference rebuilds its optimizer every epoch,
    matrix_factorization.py:176): the first step equals the default's bit for bit, later steps follow Keras Adam with
    moments carried over, evaluated here in NumPy from the oracle's gradients."""
    from oracle import sparse_ref as S
    from teamoflow_amd.mf.sparse import eye
    A, idx, val, U0, V0, R, Fu, Fv = problem(3, m=50, n=35, r=8, S_=9)
    m, n = A.shape
    Sn = R.shape[1]

    def model(opt, epochs):
        kw = dict(user_weight_graph=ns.I.FixedInitializer(U0), item_weight_graph=ns.I.FixedInitializer(V0))
        if loss == 'wmrb':
            kw.update(loss_graph=ns.L.WMRBLoss(), n_users=m, n_items=n, n_samples=Sn)
        mf = ns.MF(U0.shape[1], **kw)
        mf.verbose, mf.optimizer = False, opt
        if loss == 'wmrb':
            mf.random_ind = torch.as_tensor(R)
        mf.fit(epochs, eye(m), eye(n), ns.Sparse(idx, val, (m, n)), lr=0.02)
        return mf
    assert ns.MF(3).optimizer == 'fresh_adam'
    a1, f1 = model('adam', 1), model('fresh_adam', 1)
    assert torch.equal(a1.user_embedding, f1.user_embedding) and torch.equal(a1.item_embedding, f1.item_embedding)
    for slices in (None, '3'):
        if slices:
            monkeypatch.setenv('TMF_ITEM_SLICES', slices)
        got = model('adam', 8)
        U, V = U0.astype(np.float64), V0.astype(np.float64)
        mU, vU, mV, vV = (np.zeros_like(x) for x in (U, U, V, V))
        ref = []
        for t in range(1, 9):
            if loss == 'mse':
                _, _, mean, tt = S.mse_epoch(U, V, idx, val.astype(np.float64), 0.02)
            else:
                _, _, mean, tt = S.wmrb_epoch(U, V, idx, val.astype(np.float64), R, n, Sn, 0.02)
            ref.append(mean)
            alpha = 0.02 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            mU += (tt['gU'] - mU) * 0.1
            vU += (tt['gU'] ** 2 - vU) * 0.001
            mV += (tt['gV'] - mV) * 0.1
            vV += (tt['gV'] ** 2 - vV) * 0.001
            U = U - alpha * mU / (np.sqrt(vU) + 1e-7)
            V = V - alpha * mV / (np.sqrt(vV) + 1e-7)
        assert rel_err(got.loss_history_[:3], ref[:3]) < 1e-5 and rel_err(got.loss_history_, ref) < 1e-3
        assert rel_err(got.user_embedding.cpu().numpy(), U) < 2e-2 and rel_err(got.item_embedding.cpu().numpy(), V) < 2e-2
    with pytest.raises(ValueError):
        model('sgd', 1)


@pytest.mark.parametrize('loss', ['mse', 'wmrb'])
def test_minibatch_over_user_batches(loss):
    """Opt-in extension (model.batch_users): an epoch = one fresh-Adam step per batch of users on that batch's part of the
    loss, the next batch seeing the updated item table.  Oracle: the same sequence composed from the full-batch closed-form
    epoch applied to each batch's sub-problem.  batch_users >= n_users is the reference's full-batch fit, bit for bit."""
    import numpy as np
    from oracle import sparse_ref as S
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import WMRBLoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    rng = np.random.default_rng(3)
    m, n, r, Sn, lr, B = 103, 57, 12, 9, 0.02, 40          # batches of 40, 40, 23 users
    A = (rng.random((m, n)) < 0.15) * rng.integers(1, 6, (m, n))
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, Sn, replace=False) for _ in range(m)])

    def fit(batch, epochs):
        kw = dict(user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
        if loss == 'wmrb':
            kw.update(loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=Sn)
        model = MatrixFactorization(r, **kw)
        model.verbose, model.batch_users = False, batch
        if loss == 'wmrb':
            model.random_ind = torch.as_tensor(R)
        model.fit(epochs, eye(m), eye(n), SparseInteractions(idx, val, (m, n)), lr=lr)
        return model

    model = fit(B, 2)
    U, V = U0.copy(), V0.copy()
    ref = []
    for _ in range(2):
        tot, cnt = 0.0, 0
        for b0 in range(0, m, B):
            b1 = min(b0 + B, m)
            keep = (idx[:, 0] >= b0) & (idx[:, 0] < b1)
            sub = idx[keep].copy()
            sub[:, 0] -= b0
            if loss == 'mse':
                U[b0:b1], V, _, t = S.mse_epoch(U[b0:b1], V, sub, val[keep], lr)
                cnt += int(keep.sum())
            else:
                U[b0:b1], V, _, t = S.wmrb_epoch(U[b0:b1], V, sub, val[keep], R[b0:b1], n, Sn, lr)
                cnt += int((val[keep] > 0).sum())
            tot += float(t['loss'].astype(np.float64).sum())
        ref.append(tot / cnt)
    assert np.allclose(model.loss_history_, ref, rtol=2e-5), (model.loss_history_, ref)
    dU = np.abs(model.user_embedding.cpu().numpy() - U)
    dV = np.abs(model.item_embedding.cpu().numpy() - V)
    assert (dU < 1e-4).mean() > 0.98 and dU.max() <= 2 * 2 * lr            # near-sign steps: elements with g ~ 0 may differ by a step
    assert (dV < 1e-4).mean() > 0.95 and dV.max() <= 2 * 6 * lr
    full, one = fit(0, 2), fit(m + 5, 2)
    assert one.loss_history_ == full.loss_history_ and torch.equal(one.item_embedding, full.item_embedding)
    assert torch.equal(one.user_embedding, full.user_embedding)
