"""Feature -> embedding maps (plug-in interface of /root/reference/src/teamoflow/mf/embedding_graphs.py).

``LinearEmbedding`` over indicator features is the hot path: the embedding IS the weight table, so
``MatrixFactorization.fit`` never calls ``get_repr`` there and runs the sparse HIP engine instead.
``get_repr`` itself is the generic definition (torch ops on whatever device the tensors live on) used
for dense, non-identity features and for the biased / ReLU variants.
"""
from abc import ABC, abstractmethod

import torch

from .sparse import IndicatorFeatures


def _dense(features, like):
    if isinstance(features, IndicatorFeatures):
        return features.to_dense(device=like.device)
    return torch.as_tensor(features, dtype=like.dtype, device=like.device)


class Embeddings(ABC):
    """embedding_graphs.py:7-22: get_repr returns (embedding, [trainable tensors...])."""

    @abstractmethod
    def get_repr(self, features, weights, aux_dim=None, relu_weight=None, relu_bias=None, linear_bias=None):
        pass


class LinearEmbedding(Embeddings):
    """embedding_graphs.py:30-38: features @ weights."""

    def get_repr(self, features, weights, aux_dim=None, relu_weight=None, relu_bias=None, linear_bias=None):
        if isinstance(features, IndicatorFeatures):
            return weights, [weights]  # I @ W == W
        return _dense(features, weights) @ weights, [weights]


class BiasedLinearEmbedding(Embeddings):
    """embedding_graphs.py:41-58: features @ weights + a trainable [1, r] bias (zeros at first use)."""

    def get_repr(self, features, weights, aux_dim=None, relu_weight=None, relu_bias=None, linear_bias=None):
        r = weights.shape[1]
        if linear_bias is None:
            linear_bias = torch.zeros(1, r, dtype=weights.dtype, device=weights.device, requires_grad=True)
        return _dense(features, weights) @ weights + linear_bias, [weights, linear_bias]


class ReLUEmbedding(Embeddings):
    """embedding_graphs.py:61-87: relu(features @ relu_weight + relu_bias) @ weights, aux width 5r."""

    def get_repr(self, features, weights, aux_dim=None, relu_weight=None, relu_bias=None, linear_bias=None):
        f = _dense(features, weights)
        n_features = f.shape[1]
        if aux_dim is None:
            aux_dim = 5 * weights.shape[1]
        if relu_weight is None:
            relu_weight = torch.randn(n_features, aux_dim, dtype=weights.dtype, device=weights.device).requires_grad_(True)
        if relu_bias is None:
            relu_bias = torch.zeros(1, aux_dim, dtype=weights.dtype, device=weights.device, requires_grad=True)
        hidden = torch.relu(f @ relu_weight + relu_bias)
        return hidden @ weights, [weights, relu_weight, relu_bias]
