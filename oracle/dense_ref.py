"""Dense-faithful CPU restatement of ``MatrixFactorization.fit/predict/recall_at_k``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) - parity unpinned except
``gather_matrix_indices``.

This file follows the reference op for op in torch-CPU, with ``torch.autograd``
standing in for ``tf.GradientTape``.  It deliberately keeps the reference's dense,
full-batch formulation (identity-feature matmuls, the [m, n] score matrix, the
[P, S] hinge tensor) so that it is independent of the hand-derived closed forms
the HIP kernels implement (those are restated separately in ``sparse_ref.py``).

Reference lines followed (all under /root/reference/src/teamoflow/mf/):
  matrix_factorization.py:96-187  fit loop (timing region :129-:177)
  matrix_factorization.py:189-201 predict
  matrix_factorization.py:218-269 recall_at_k
  matrix_factorization.py:416-438 retrieve_user_recs
  loss_graphs.py:36-52            MSELoss.get_loss
  loss_graphs.py:62-88            WMRBLoss.get_loss
  embedding_graphs.py:30-38       LinearEmbedding.get_repr
  embedding_graphs.py:41-58       BiasedLinearEmbedding.get_repr   (fit_dense_plugins)
  embedding_graphs.py:61-87       ReLUEmbedding.get_repr           (fit_dense_plugins)
  loss_graphs.py:91-122           KLDivergenceLoss.get_loss        (kl_loss; normal CDF as tensorflow-probability's ndtr)
  initializer_graphs.py:27-52     Normal/UniformInitializer (global l2_normalize)
  utils.py:62-105                 gather_matrix_indices
TensorFlow semantics restated (TF >= 2.9, unpinned in the reference):
  keras Adam   alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
               w -= (m*alpha)/(sqrt(v)+eps), b1=.9 b2=.999 eps=1e-7, all in fp32
  tf.maximum   gradient goes to the first argument where x >= y
  tf.math.top_k  sorted descending, ties -> lower index first
  tape.gradient(vector_target) = gradient of the sum of the vector
"""
import timeit

import numpy as np
import torch


# --------------------------------------------------------------------------------------
# TensorFlow op semantics
# --------------------------------------------------------------------------------------
class _TFMaximum(torch.autograd.Function):
    """tf.maximum(x, y): forward max, backward routes to x where x >= y (else to y)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x >= y)
        return torch.maximum(x, y)

    @staticmethod
    def backward(ctx, g):
        (xmask,) = ctx.saved_tensors
        gx = torch.where(xmask, g, torch.zeros_like(g))
        gy = torch.where(xmask, torch.zeros_like(g), g)
        return gx, gy


def tf_maximum(x, y):
    y = torch.as_tensor(y, dtype=x.dtype).expand_as(x)
    return _TFMaximum.apply(x, y)


def tf_top_k(x, k):
    """tf.math.top_k(x, k): values/indices sorted descending; equal values keep the
    lower index first.  Works on the last axis."""
    vals, idx = torch.sort(x, dim=-1, descending=True, stable=True)
    return vals[..., :k], idx[..., :k]


def adam_fresh_constants(lr, dtype=np.float32):
    """Scalars of a Keras Adam step at iteration 1, computed in ``dtype`` like TF does
    (matrix_factorization.py:176 builds a NEW optimizer every epoch => t = 1, m = v = 0)."""
    f = dtype
    one, b1, b2, eps = f(1.0), f(0.9), f(0.999), f(1e-7)
    lr = f(lr)
    alpha = f(lr * f(np.sqrt(f(one - b2))) / f(one - b1))
    return alpha, f(one - b1), f(one - b2), eps


def adam_fresh_step(w, g, lr):
    """w, g: torch tensors of the same dtype.  Returns the updated weights."""
    npdt = np.float32 if w.dtype == torch.float32 else np.float64
    alpha, omb1, omb2, eps = (float(x) for x in adam_fresh_constants(lr, npdt))
    m = (g - 0.0) * omb1
    v = (g * g - 0.0) * omb2
    return w - (m * alpha) / (torch.sqrt(v) + eps)


def l2_normalize_global(x, eps=1e-12):
    """tf.math.l2_normalize(x) with axis=None: x * rsqrt(max(sum(x^2), eps))
    (initializer_graphs.py:34, :51)."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(), min=eps))


# --------------------------------------------------------------------------------------
# utils.py:62-105
# --------------------------------------------------------------------------------------
def gather_matrix_indices(input_arr, index_arr):
    """out[i, c] = input_arr[i, index_arr[i, c]] built the way the reference builds it:
    an explicit [rows, cols, 2] (row, col) index tensor followed by gather_nd."""
    rows, cols = index_arr.shape
    row_ind = torch.arange(rows, dtype=torch.int64)[:, None].repeat(1, cols)
    return input_arr[row_ind, index_arr.to(torch.int64)]


# --------------------------------------------------------------------------------------
# loss_graphs.py
# --------------------------------------------------------------------------------------
def mse_loss(indices, values, predictions):
    """loss_graphs.py:47-52."""
    p = predictions[indices[:, 0], indices[:, 1]]
    return torch.square(values - p)


def wmrb_loss(indices, values, sample_predictions, prediction_serial, n_items, n_samples):
    """loss_graphs.py:74-88."""
    mask = values > 0.0
    pos_idx = indices[mask]
    pos_pred = prediction_serial[mask]
    mapped = sample_predictions[pos_idx[:, 0]]  # [P, S]
    summation = tf_maximum(1.0 - pos_pred[:, None] + mapped, 0.0)
    rank = (n_items / n_samples) * summation.sum(dim=1)
    return torch.log(1.0 + rank)


def tfp_ndtr(z):
    """Normal CDF the way tensorflow-probability >= 0.17 evaluates it (special_math._ndtr; the reference's
    `tp.distributions.Normal(...).cdf`, loss_graphs.py:120-122, is ndtr((x - loc) / scale)):
        w = z / sqrt(2);  |w| < sqrt(1/2): (1 + erf(w)) / 2;  w > 0: 1 - erfc(w) / 2;  else erfc(-w) / 2."""
    half_sqrt_2 = 0.5 * np.sqrt(2.0)
    w = z * half_sqrt_2
    a = torch.abs(w)
    y = torch.where(a < half_sqrt_2, 1.0 + torch.erf(w), torch.where(w > 0.0, 2.0 - torch.erfc(a), torch.erfc(a)))
    return 0.5 * y


def tf_moments(x):
    """tf.nn.moments(x, axes=[0]): mean and POPULATION variance mean((x - mean)^2)."""
    mean = x.mean()
    return mean, torch.square(x - mean).mean()


def kl_loss(values, prediction_serial):
    """loss_graphs.py:109-122: 1 - CDF_{N(mean_neg - mean_pos, sqrt(var_pos + var_neg))}(0) - ONE scalar for the whole epoch;
    positives are the stored values > 0, negatives the stored values <= 0."""
    pos_mask = values > 0.0
    neg_mask = values <= 0.0
    pos_mean, pos_var = tf_moments(prediction_serial[pos_mask])
    neg_mean, neg_var = tf_moments(prediction_serial[neg_mask])
    loc, scale = neg_mean - pos_mean, torch.sqrt(pos_var + neg_var)
    return 1.0 - tfp_ndtr((0.0 - loc) / scale)


def embed(kind, features, weights, relu_weight=None, relu_bias=None, linear_bias=None):
    """embedding_graphs.py:30-87 -> (embedding, [trainables]) for kind in {'linear', 'biased', 'relu'}."""
    if kind == 'linear':
        return features @ weights, [weights]                                   # :38
    if kind == 'biased':
        return features @ weights + linear_bias, [weights, linear_bias]        # :58 (bias [1, r], broadcast over the rows)
    if kind == 'relu':
        hidden = torch.relu(features @ relu_weight + relu_bias)                # :85
        return hidden @ weights, [weights, relu_weight, relu_bias]             # :87
    raise ValueError(kind)


def fit_dense_plugins(U0, V0, indices, values, loss, epochs, lr, user_features, item_features, user_embedding='linear',
                      item_embedding='linear', user_relu_weight0=None, item_relu_weight0=None, random_ind=None,
                      n_items=None, n_samples=None, dtype=torch.float32, record_epochs=()):
    """The reference training loop (matrix_factorization.py:96-187) for ANY built-in plug-in combination:
    embeddings 'linear' | 'biased' | 'relu' per side, loss 'mse' | 'wmrb' | 'kl'.

    What the reference does that this follows:
      * biased: the [1, r] bias is created as zeros inside get_repr at the first epoch (:53-55), kept on the model (:139-146) and
        fed back in; ReLU: weights are [aux_dim = 5 r, r] (:117, :122), relu_bias zeros [1, aux_dim] (:82-83); the [n_features,
        aux_dim] relu_weight is drawn from TF's RNG at first use (:80-81) - not reproducible, so it is an INPUT here;
      * every trainable of both sides takes the same fresh Keras-Adam step (:173-176);
      * KL: tf_prediction_serial = gather_nd(predictions, indices) (:158-160), the loss is ONE scalar, reduce_mean of it is itself.
    Returns dict(loss=[epochs], user_vars=[...], item_vars=[...] (final, numpy), snapshots={epoch: (user_vars, item_vars)},
    first_grads=(user, item) gradients of the first epoch, user_embedding / item_embedding of the final weights (:186-187))."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype).clone()   # noqa: E731
    idx = torch.as_tensor(np.asarray(indices), dtype=torch.int64)
    val = t(values)
    Fu, Fv = t(user_features), t(item_features)
    R = None if random_ind is None else torch.as_tensor(np.asarray(random_ind), dtype=torch.int64)
    r = np.asarray(U0).shape[1]

    def variables(kind, W0, relu_w0, n_features):
        if kind == 'linear':
            return [t(W0)]
        if kind == 'biased':
            return [t(W0), torch.zeros(1, r, dtype=dtype)]
        aux = 5 * r
        assert np.asarray(W0).shape == (aux, r) and np.asarray(relu_w0).shape == (n_features, aux)
        return [t(W0), t(relu_w0), torch.zeros(1, aux, dtype=dtype)]

    def run(kind, feats, vs):
        if kind == 'linear':
            return embed(kind, feats, vs[0])
        if kind == 'biased':
            return embed(kind, feats, vs[0], linear_bias=vs[1])
        return embed(kind, feats, vs[0], relu_weight=vs[1], relu_bias=vs[2])

    uv = variables(user_embedding, U0, user_relu_weight0, Fu.shape[1])
    iv = variables(item_embedding, V0, item_relu_weight0, Fv.shape[1])
    losses, snaps, first_grads = [], {}, None
    for epoch in range(epochs):
        for v in uv + iv:
            v.requires_grad_(True)
        ue, utr = run(user_embedding, Fu, uv)
        ie, itr = run(item_embedding, Fv, iv)
        predictions = ue @ ie.T                                                 # :149
        if loss == 'wmrb':
            loss_fn = wmrb_loss(idx, val, gather_matrix_indices(predictions, R), predictions[idx[:, 0], idx[:, 1]], n_items, n_samples)
        elif loss == 'mse':
            loss_fn = mse_loss(idx, val, predictions)
        elif loss == 'kl':
            loss_fn = kl_loss(val, predictions[idx[:, 0], idx[:, 1]])           # :158-160
        else:
            raise ValueError(loss)
        grads = torch.autograd.grad(loss_fn.sum(), utr + itr)                   # :170-171 (gradient of the sum)
        if first_grads is None:
            first_grads = ([g.numpy().copy() for g in grads[:len(uv)]], [g.numpy().copy() for g in grads[len(uv):]])
        with torch.no_grad():
            new = [adam_fresh_step(v.detach(), g, lr) for v, g in zip(utr + itr, grads)]   # :176
        uv, iv = new[:len(uv)], new[len(uv):]
        losses.append(float(loss_fn.detach().mean()))                           # :179
        if (epoch + 1) in record_epochs:
            snaps[epoch + 1] = ([v.numpy().copy() for v in uv], [v.numpy().copy() for v in iv])
    with torch.no_grad():
        ue, _ = run(user_embedding, Fu, uv)
        ie, _ = run(item_embedding, Fv, iv)
    return dict(loss=np.asarray(losses, dtype=np.float64), user_vars=[v.numpy() for v in uv], item_vars=[v.numpy() for v in iv],
                snapshots=snaps, first_grads=first_grads, user_embedding=ue.numpy(), item_embedding=ie.numpy())


# --------------------------------------------------------------------------------------
# matrix_factorization.py:96-187
# --------------------------------------------------------------------------------------
def fit_dense(U0, V0, indices, values, loss, epochs, lr=1e-2, random_ind=None, n_items=None,
              n_samples=None, user_features=None, item_features=None, dtype=torch.float32,
              record_epochs=(), timer=None):
    """One full run of the reference training loop.

    U0/V0: initial weights [n_user_features, r] / [n_item_features, r].
    indices [nnz, 2] int64, values [nnz]; loss in {'mse', 'wmrb'}.
    n_items / n_samples are the CONSTRUCTOR ints (matrix_factorization.py:167).
    user_features / item_features default to identity (indicator features).
    Returns dict(loss=[epochs] mean loss per epoch (pre-update weights), U, V,
    snapshots={epoch: (U, V)} for epochs in record_epochs (1-based, state AFTER that epoch),
    seconds=sum of the per-epoch timed regions).
    """
    U = torch.as_tensor(np.asarray(U0), dtype=dtype).clone()
    V = torch.as_tensor(np.asarray(V0), dtype=dtype).clone()
    idx = torch.as_tensor(np.asarray(indices), dtype=torch.int64)
    val = torch.as_tensor(np.asarray(values), dtype=dtype)
    Fu = torch.eye(U.shape[0], dtype=dtype) if user_features is None else torch.as_tensor(user_features, dtype=dtype)
    Fv = torch.eye(V.shape[0], dtype=dtype) if item_features is None else torch.as_tensor(item_features, dtype=dtype)
    R = None if random_ind is None else torch.as_tensor(np.asarray(random_ind), dtype=torch.int64)
    losses, snaps, seconds = [], {}, 0.0
    for epoch in range(epochs):
        t0 = timeit.default_timer()
        U.requires_grad_(True)
        V.requires_grad_(True)
        user_embedding = Fu @ U  # embedding_graphs.py:38
        item_embedding = Fv @ V
        predictions = user_embedding @ item_embedding.T  # :149
        if loss == 'wmrb':
            sample_predictions = gather_matrix_indices(predictions, R)  # :153
            prediction_serial = predictions[idx[:, 0], idx[:, 1]]  # :154
            loss_fn = wmrb_loss(idx, val, sample_predictions, prediction_serial, n_items, n_samples)
        elif loss == 'mse':
            loss_fn = mse_loss(idx, val, predictions)
        else:
            raise ValueError(loss)
        gU, gV = torch.autograd.grad(loss_fn.sum(), [U, V])  # :170-171
        with torch.no_grad():
            U = adam_fresh_step(U.detach(), gU, lr)  # :176
            V = adam_fresh_step(V.detach(), gV, lr)
        seconds += timeit.default_timer() - t0
        losses.append(float(loss_fn.detach().mean()))  # :179
        if (epoch + 1) in record_epochs:
            snaps[epoch + 1] = (U.numpy().copy(), V.numpy().copy())
    return dict(loss=np.asarray(losses, dtype=np.float64), U=U.numpy(), V=V.numpy(), snapshots=snaps,
                seconds=seconds)


def predict_dense(user_embedding, item_embedding, A=None):
    """matrix_factorization.py:189-201."""
    Ue = torch.as_tensor(np.asarray(user_embedding))
    Ve = torch.as_tensor(np.asarray(item_embedding))
    allp = Ue @ Ve.T
    if A is None:
        return allp.numpy()
    A = torch.as_tensor(np.asarray(A))
    where = torch.nonzero(A == 0)  # row-major order, like tf.where
    return allp.numpy(), allp[where[:, 0], where[:, 1]].numpy()


def recall_at_k_dense(user_embedding, item_embedding, A, k=10, preserve_rows=False):
    """matrix_factorization.py:236-269."""
    predictions = torch.as_tensor(predict_dense(user_embedding, item_embedding))
    A = torch.as_tensor(np.asarray(A), dtype=predictions.dtype)
    positive_predictions = torch.where(predictions > 0.0, predictions, torch.zeros_like(predictions))
    known_positives = torch.where(A > 0.0, A, torch.zeros_like(A))
    _, top_k_items_user = tf_top_k(positive_predictions, k)
    res_top_k = gather_matrix_indices(A, top_k_items_user)
    relevant = torch.count_nonzero(known_positives, dim=1).to(torch.float32)
    hits = torch.count_nonzero(res_top_k, dim=1).to(torch.float32)
    if not preserve_rows:
        mask = relevant != 0.0
        return (hits[mask] / relevant[mask]).numpy()
    recall = hits / relevant
    return torch.where(torch.isnan(recall), torch.zeros_like(recall), recall).numpy()


def precision_at_k_dense(user_embedding, item_embedding, A, k=10, preserve_rows=False):
    """matrix_factorization.py:285-304."""
    predictions = torch.as_tensor(predict_dense(user_embedding, item_embedding))
    A = torch.as_tensor(np.asarray(A), dtype=predictions.dtype)
    positive_predictions = torch.where(predictions > 0.0, predictions, torch.zeros_like(predictions))
    _, top_k_items_user = tf_top_k(positive_predictions, k)
    hits = torch.count_nonzero(gather_matrix_indices(A, top_k_items_user), dim=1).to(torch.float32)
    if not preserve_rows:
        relevant = torch.count_nonzero(torch.where(A > 0.0, A, torch.zeros_like(A)), dim=1)
        return (hits[relevant != 0] / k).numpy()
    return (hits / k).numpy()


def retrieve_user_recs_dense(user_embedding, item_embedding, user=None, k=None):
    """matrix_factorization.py:424-438 (tf.math.top_k indices are int32)."""
    p = torch.as_tensor(predict_dense(user_embedding, item_embedding))
    n = p.shape[1]
    if user is not None:
        p = p[user]
    return tf_top_k(p, n if k is None else k)[1].numpy().astype(np.int32)


def dcg_at_k_dense(user_embedding, item_embedding, A, k=10, ideal=False):
    """matrix_factorization.py:332-351 / :363-384."""
    p = torch.as_tensor(predict_dense(user_embedding, item_embedding))
    A = torch.as_tensor(np.asarray(A), dtype=p.dtype)
    m, n = p.shape
    _, ranks = tf_top_k(p, n)
    numerator = torch.pow(2.0, gather_matrix_indices(A, ranks)) - 1.0
    if ideal:
        numerator, _ = tf_top_k(numerator, n)
    denom = torch.log1p(torch.arange(1, n + 1, dtype=p.dtype)) / np.log(np.float32(2.0))
    return (numerator / denom[None, :])[:, :k].sum(dim=1).numpy()


def ndcg_at_k_dense(user_embedding, item_embedding, A, k=10, preserve_rows=False):
    """matrix_factorization.py:397-413."""
    dcg = dcg_at_k_dense(user_embedding, item_embedding, A, k)
    idcg = dcg_at_k_dense(user_embedding, item_embedding, A, k, ideal=True)
    with np.errstate(invalid='ignore', divide='ignore'):
        ndcg = dcg / idcg
    if not preserve_rows:
        return ndcg[np.count_nonzero(np.asarray(A), axis=1) > 0]
    return np.where(~np.isnan(ndcg), ndcg, 0.0).astype(np.float32)


def f1_at_k_dense(user_embedding, item_embedding, A, k=10, beta=1.0):
    """matrix_factorization.py:314-318 (denominator beta^2 (p + r), as the reference has it)."""
    prec = precision_at_k_dense(user_embedding, item_embedding, A, k).mean()
    rec = recall_at_k_dense(user_embedding, item_embedding, A, k).mean()
    return ((1 + beta ** 2) * prec * rec) / (beta ** 2 * (prec + rec))


def predict_ranks_dense(user_embedding, item_embedding, A):
    """matrix_factorization.py:209-216: global descending ranking of the unobserved predictions."""
    _, unobserved = predict_dense(user_embedding, item_embedding, A)
    return tf_top_k(torch.as_tensor(unobserved), len(unobserved))[1].numpy()
