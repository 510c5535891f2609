"""Sparse closed-form CPU restatement of one training epoch (the algorithm the HIP kernels implement).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) - parity unpinned except
``gather_matrix_indices``.

With indicator (identity) features and ``LinearEmbedding`` the reference's dense step
(/root/reference/src/teamoflow/mf/matrix_factorization.py:130-176) reduces to sums over the
observed interactions only.  These are SURVEY.md Appendix A.2 / A.3; every function here is
checked against ``dense_ref`` (autograd over the dense formulation) in ``tests/test_oracle.py``.

MSE  (loss_graphs.py:47-52):
    p_k = <U[u_k], V[j_k]>; l_k = (a_k - p_k)^2; d_k = -2 (a_k - p_k)
    gU[i] = sum_{k: u_k = i} d_k V[j_k];  gV[j] = sum_{k: j_k = j} d_k U[u_k]
WMRB (loss_graphs.py:74-88, matrix_factorization.py:153-154,167):
    c = n_items / n_samples (constructor ints); positives P = {k: a_k > 0}
    sp[u, s] = <U[u], V[R[u, s]]>;  x_ks = (1 - p_k) + sp[u_k, s]
    M_k = c * sum_s max(x_ks, 0); l_k = log(1 + M_k); w_k = c / (1 + M_k); a_ks = [x_ks >= 0]
    d_k = -w_k * sum_s a_ks;  D[u, s] = sum_{k in P, u_k = u} w_k a_ks
    gU[i] = sum_{k in P: u_k = i} d_k V[j_k] + sum_s D[i, s] V[R[i, s]]
    gV[j] = sum_{k in P: j_k = j} d_k U[u_k] + sum_{(u, s): R[u, s] = j} D[u, s] U[u]
Update (matrix_factorization.py:176, fresh Keras Adam => t = 1): see ``adam_fresh``.
All gradients use the PRE-update U and V; both tables are updated simultaneously.
"""
import numpy as np


def adam_fresh(w, g, lr):
    """Fresh-Adam step in the dtype of ``w`` (unsimplified op sequence)."""
    f = w.dtype.type
    one, b1, b2, eps = f(1.0), f(0.9), f(0.999), f(1e-7)
    alpha = f(f(lr) * np.sqrt(f(one - b2)) / f(one - b1))
    m = (g - f(0)) * f(one - b1)
    v = (g * g - f(0)) * f(one - b2)
    return (w - (m * alpha) / (np.sqrt(v) + eps)).astype(w.dtype)


def mse_epoch(U, V, indices, values, lr):
    """Returns (U_new, V_new, mean_loss, terms) with terms = dict(loss, delta, gU, gV)."""
    u, j = indices[:, 0], indices[:, 1]
    a = values.astype(U.dtype)
    p = np.einsum('kc,kc->k', U[u], V[j]).astype(U.dtype)
    e = a - p
    loss = e * e
    d = (U.dtype.type(-2.0) * e)
    gU = np.zeros_like(U)
    gV = np.zeros_like(V)
    np.add.at(gU, u, d[:, None] * V[j])
    np.add.at(gV, j, d[:, None] * U[u])
    mean = float(loss.astype(np.float64).mean()) if len(loss) else float('nan')
    return adam_fresh(U, gU, lr), adam_fresh(V, gV, lr), mean, dict(loss=loss, delta=d, gU=gU, gV=gV)


def wmrb_terms(U, V, indices, values, R, n_items, n_samples):
    """Forward quantities of the WMRB loss: dict with pos (index array into interactions),
    p [P], M [P], loss [P], w [P], cnt [P], delta [P], D [m, S], sp [m, S]."""
    f = U.dtype.type
    c = f(n_items / n_samples)
    pos = np.nonzero(values > 0)[0]
    u, j = indices[pos, 0], indices[pos, 1]
    p = np.einsum('kc,kc->k', U[u], V[j]).astype(U.dtype)
    sp = np.einsum('uc,usc->us', U, V[R]).astype(U.dtype)
    x = (f(1.0) - p)[:, None] + sp[u]
    M = c * np.maximum(x, f(0)).sum(axis=1, dtype=U.dtype)
    act = (x >= 0)
    w = c / (f(1.0) + M)
    cnt = act.sum(axis=1).astype(U.dtype)
    delta = -w * cnt
    D = np.zeros(sp.shape, dtype=U.dtype)
    np.add.at(D, u, w[:, None] * act)
    return dict(pos=pos, p=p, M=M, loss=np.log(f(1.0) + M), w=w, cnt=cnt, delta=delta, D=D, sp=sp)


def wmrb_slack(U, V, indices, values, R, n_items, n_samples, tol_rel=None):
    """How far D, delta (over the positives), gU and gV may move when the hinge terms whose argument x_ks lies within
    the tolerance the predictions are compared at - |x_ks| <= tol_rel (1 + sum_c |U V[j_k]| + sum_c |U V[R[u,s]]|),
    tol_rel = max(1e-5, 2 r 2^-24): an fp32 dot product is only determined up to r 2^-24 times the sum of the magnitudes
    of its terms - switch between active and inactive.  The hinge has a kink at 0 (loss_graphs.py:83-84): two valid fp32 evaluations of the reference
    (different summation order of the dot products) disagree on exactly those terms, and one switch moves D[u,s] by
    w_k, delta_k by w_k, gU[u] by w_k (V[R[u,s]] - V[j_k]) and the two item rows by w_k U[u].  Same quantity as
    oracle_wmrb_boundary_slack in sparse_ref.c.  -> dict(D, delta, gU, gV, pairs)."""
    t = wmrb_terms(U, V, indices, values, R, n_items, n_samples)
    pos = t['pos']
    u, j = indices[pos, 0], indices[pos, 1]
    if tol_rel is None:
        tol_rel = max(1e-5, 2 * U.shape[1] * 2.0 ** -24)
    x = (1.0 - t['p'])[:, None] + t['sp'][u]
    ap = np.einsum('kc,kc->k', np.abs(U[u]), np.abs(V[j]))
    asp = np.einsum('uc,usc->us', np.abs(U), np.abs(V[R]))
    near = np.abs(x) <= tol_rel * (1.0 + ap[:, None] + asp[u])
    D = np.zeros(t['sp'].shape)
    gU, gV = np.zeros(U.shape), np.zeros(V.shape)
    kk, ss = np.nonzero(near)
    w = t['w'][kk]
    np.add.at(D, (u[kk], ss), w)
    delta = np.zeros(len(pos))
    np.add.at(delta, kk, w)
    js = R[u[kk], ss]
    np.add.at(gU, u[kk], w[:, None] * np.abs(V[js] - V[j[kk]]))
    np.add.at(gV, js, w[:, None] * np.abs(U[u[kk]]))
    np.add.at(gV, j[kk], w[:, None] * np.abs(U[u[kk]]))
    return dict(D=D, delta=delta, gU=gU, gV=gV, pairs=int(len(kk)))


def wmrb_epoch(U, V, indices, values, R, n_items, n_samples, lr):
    """Returns (U_new, V_new, mean_loss over positives, terms dict)."""
    t = wmrb_terms(U, V, indices, values, R, n_items, n_samples)
    pos = t['pos']
    u, j = indices[pos, 0], indices[pos, 1]
    gU = np.zeros_like(U)
    gV = np.zeros_like(V)
    np.add.at(gU, u, t['delta'][:, None] * V[j])
    np.add.at(gV, j, t['delta'][:, None] * U[u])
    gU += np.einsum('us,usc->uc', t['D'], V[R]).astype(U.dtype)
    m, S = R.shape
    np.add.at(gV, R.reshape(-1), (t['D'][:, :, None] * U[:, None, :]).reshape(m * S, -1))
    mean = float(t['loss'].astype(np.float64).mean()) if len(pos) else float('nan')
    t['gU'], t['gV'] = gU, gV
    return adam_fresh(U, gU, lr), adam_fresh(V, gV, lr), mean, t


def fit_sparse(U0, V0, indices, values, loss, epochs, lr=1e-2, random_ind=None, n_items=None,
               n_samples=None, dtype=np.float32, record_epochs=()):
    U = np.asarray(U0, dtype=dtype).copy()
    V = np.asarray(V0, dtype=dtype).copy()
    indices = np.asarray(indices, dtype=np.int64)
    values = np.asarray(values, dtype=dtype)
    losses, snaps = [], {}
    for epoch in range(epochs):
        if loss == 'mse':
            U, V, mean, _ = mse_epoch(U, V, indices, values, lr)
        elif loss == 'wmrb':
            U, V, mean, _ = wmrb_epoch(U, V, indices, values, np.asarray(random_ind), n_items, n_samples, lr)
        else:
            raise ValueError(loss)
        losses.append(mean)
        if (epoch + 1) in record_epochs:
            snaps[epoch + 1] = (U.copy(), V.copy())
    return dict(loss=np.asarray(losses), U=U, V=V, snapshots=snaps)


def topk_stable(x, k):
    """(value desc, index asc) top-k indices of each row - tf.math.top_k ordering."""
    idx = np.argsort(-x, axis=-1, kind='stable')
    return idx[..., :k]


def recall_at_k_sparse(Ue, Ve, indices, values, k=10, preserve_rows=False):
    """recall_at_k (matrix_factorization.py:236-269) evaluated from COO interactions instead of a
    dense A: hits = top-k items with a NON-ZERO stored value, relevant = #entries with value > 0."""
    m, n = Ue.shape[0], Ve.shape[0]
    pred = Ue @ Ve.T
    top = topk_stable(np.where(pred > 0, pred, 0).astype(pred.dtype), k)
    nz = values != 0
    keys = set((indices[nz, 0] * n + indices[nz, 1]).tolist())
    hits = np.array([[int(i * n + c) in keys for c in top[i]] for i in range(m)]).sum(axis=1).astype(np.float32)
    relevant = np.bincount(indices[values > 0, 0], minlength=m).astype(np.float32)
    if not preserve_rows:
        mask = relevant != 0
        return hits[mask] / relevant[mask]
    with np.errstate(invalid='ignore', divide='ignore'):
        r = hits / relevant
    return np.where(np.isnan(r), 0, r).astype(np.float32)
