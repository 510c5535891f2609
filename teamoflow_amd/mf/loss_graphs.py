"""Loss graphs (plug-in interface of /root/reference/src/teamoflow/mf/loss_graphs.py).

``MSELoss`` and ``WMRBLoss`` with ``LinearEmbedding`` over indicator features are recognised by
``MatrixFactorization.fit`` (same isinstance dispatch as matrix_factorization.py:152-162) and run as
fused HIP kernels; ``get_loss`` below is the generic differentiable definition used when the model is
built from other plug-ins (dense features, custom embeddings), always called by keyword like
matrix_factorization.py:165-167 does.
"""
from abc import ABC, abstractmethod

import torch


class _TFMaximum(torch.autograd.Function):
    """tf.maximum semantics: the gradient goes to x where x >= y (ties included)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x >= y)
        return torch.maximum(x, y)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        zero = torch.zeros_like(g)
        return torch.where(m, g, zero), torch.where(m, zero, g)


class LossGraph(ABC):
    """loss_graphs.py:8-28: returns the 1-D per-interaction loss vector."""

    @abstractmethod
    def get_loss(self, tf_interactions, tf_sample_predictions, tf_prediction_serial, predictions, n_items, n_samples):
        pass


class MSELoss(LossGraph):
    """loss_graphs.py:31-52: squared error on every stored interaction."""

    def get_loss(self, tf_interactions, predictions, tf_sample_predictions=None, tf_prediction_serial=None,
                 n_items=None, n_samples=None):
        idx = tf_interactions.indices
        return torch.square(tf_interactions.values - predictions[idx[:, 0], idx[:, 1]])


class WMRBLoss(LossGraph):
    """loss_graphs.py:55-88: log(1 + (n_items / n_samples) * sum_s max(1 - p_k + sp[u_k, s], 0)) for
    the positive interactions only."""

    def get_loss(self, tf_interactions, tf_sample_predictions, tf_prediction_serial, n_items, n_samples,
                 predictions=None):
        mask = tf_interactions.values > 0.0
        users = tf_interactions.indices[:, 0][mask]
        pos = tf_prediction_serial[mask]
        x = 1.0 - pos[:, None] + tf_sample_predictions[users]
        hinge = _TFMaximum.apply(x, torch.zeros_like(x))
        return torch.log(1.0 + (n_items / n_samples) * hinge.sum(dim=1))


class KLDivergenceLoss(LossGraph):
    """loss_graphs.py:91-122: 1 - CDF_{N(mu_neg - mu_pos, sqrt(var_pos + var_neg))}(0); a single scalar.
    The normal CDF is written with erf (the reference uses tensorflow-probability)."""

    def get_loss(self, tf_prediction_serial, tf_interactions, tf_sample_predictions=None, predictions=None,
                 n_items=None, n_samples=None):
        pos_mask = tf_interactions.values > 0.0
        pos, neg = tf_prediction_serial[pos_mask], tf_prediction_serial[~pos_mask]
        loc = neg.mean() - pos.mean()
        scale = torch.sqrt(pos.var(unbiased=False) + neg.var(unbiased=False))
        cdf0 = 0.5 * (1.0 + torch.erf((0.0 - loc) / (scale * 2.0 ** 0.5)))
        return 1.0 - cdf0
