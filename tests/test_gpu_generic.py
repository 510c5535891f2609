"""The generic plug-in path (dense non-identity features, BiasedLinearEmbedding, ReLUEmbedding,
KLDivergenceLoss - /root/reference/src/teamoflow/mf/embedding_graphs.py:41-87, loss_graphs.py:91-122) ON THE MI355X:
`_fit_generic` runs the reference's dense loop with torch ops on the device, predict / ranking then go through the HIP
kernels.  Dense-feature models are checked against oracle.dense_ref.fit_dense(user_features=...); for the plug-ins the
oracle does not restate, the device run must reproduce the same loop run on the host from the same initial state."""
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

    class HostFixed(I.Initializer):
        """Weights that stay on the host: the generic loop then runs on CPU torch (the comparison run)."""

        def __init__(self, w):
            self.w = w

        def initialize_weights(self, n_features, n_components):
            return torch.tensor(np.asarray(self.w, np.float32)).clone().requires_grad_(True)

    class NS:
        pass
    o = NS()
    o.E, o.I, o.L, o.MF, o.Sparse, o.HostFixed = E, I, L, MatrixFactorization, SparseInteractions, HostFixed
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
def test_other_plugins_device_run_equals_host_run(ns, variant):
    A, idx, val, U0, V0, R, Fu, Fv = problem(2)
    m, n = A.shape
    r = U0.shape[1]
    rng = np.random.default_rng(5)
    relu_w = (rng.standard_normal((m, 5 * r)) * 0.2).astype(np.float32)
    Urelu = (rng.standard_normal((5 * r, r)) * 0.3).astype(np.float32)    # ReLUEmbedding: weights are [aux_dim, r]

    def run(on_device):
        Fix = ns.I.FixedInitializer if on_device else ns.HostFixed
        dev = 'cuda' if on_device else 'cpu'
        kw = dict(user_weight_graph=Fix(Urelu if variant == 'relu' else U0), item_weight_graph=Fix(V0))
        if variant == 'biased':
            kw.update(user_repr_graph=ns.E.BiasedLinearEmbedding(), item_repr_graph=ns.E.BiasedLinearEmbedding())
        elif variant == 'relu':
            kw.update(user_repr_graph=ns.E.ReLUEmbedding())
        else:
            kw.update(loss_graph=ns.L.KLDivergenceLoss())
        model = ns.MF(r, **kw)
        model.verbose = False
        if variant == 'relu':   # the reference draws this matrix at first use; pin it so both runs start equal
            model.user_relu_weight = torch.tensor(relu_w, device=dev).requires_grad_(True)
        model.fit(12, torch.tensor(Fu, device=dev), torch.tensor(Fv, device=dev), ns.Sparse(idx, val, (m, n), device=dev), lr=0.02)
        return model

    dev_model, host_model = run(True), run(False)
    assert dev_model.user_embedding.is_cuda and not host_model.user_embedding.is_cuda
    assert rel_err(dev_model.loss_history_[:3], host_model.loss_history_[:3]) < 1e-5
    assert rel_err(dev_model.loss_history_, host_model.loss_history_) < 1e-3
    assert np.isfinite(dev_model.loss_history_).all()
    if variant != 'kl':
        assert dev_model.loss_history_[-1] < dev_model.loss_history_[0]
    n_vars = {'biased': 2, 'relu': 3, 'kl': 1}[variant]
    assert len(dev_model.user_trainable) == n_vars and dev_model.user_embedding.shape == (m, r)
    # the HIP ranking of the device model's embeddings equals the stable ranking of their product
    pred = dev_model.predict().cpu().numpy()
    assert np.array_equal(dev_model.retrieve_user_recs(k=7), np.argsort(-pred, axis=1, kind='stable')[:, :7])


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
