"""Interaction containers that stand in for the TensorFlow types the reference consumes.

``SparseInteractions`` exposes exactly the three attributes the reference reads from a
``tf.sparse.SparseTensor`` (loss_graphs.py:47,52,74,76; matrix_factorization.py:154):
``indices`` [nnz, 2] int64, ``values`` [nnz] float32, ``dense_shape``.
``IndicatorFeatures(n)`` stands in for ``tf.eye(n)`` (README.md:125-127) without the O(n^2) memory.
"""
import numpy as np
import torch


def default_device():
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


class SparseInteractions:
    def __init__(self, indices, values, dense_shape, device=None):
        device = default_device() if device is None else torch.device(device)
        self.indices = torch.as_tensor(np.asarray(indices) if not torch.is_tensor(indices) else indices)
        self.indices = self.indices.to(device=device, dtype=torch.int64).reshape(-1, 2).contiguous()
        self.values = torch.as_tensor(np.asarray(values) if not torch.is_tensor(values) else values)
        self.values = self.values.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        self.dense_shape = (int(dense_shape[0]), int(dense_shape[1]))
        if self.indices.shape[0] != self.values.shape[0]:
            raise ValueError('indices and values disagree on the number of interactions')

    @property
    def shape(self):
        return self.dense_shape

    @property
    def device(self):
        return self.values.device

    @property
    def nnz(self):
        return int(self.values.shape[0])

    def to(self, device):
        return SparseInteractions(self.indices, self.values, self.dense_shape, device=device)

    def to_dense(self):
        A = torch.zeros(self.dense_shape, dtype=torch.float32, device=self.device)
        A.index_put_((self.indices[:, 0], self.indices[:, 1]), self.values, accumulate=True)
        return A

    @classmethod
    def from_dense(cls, A, device=None):
        A = torch.as_tensor(np.asarray(A) if not torch.is_tensor(A) else A)
        idx = torch.nonzero(A)  # row-major order, like scipy CSR .nonzero()
        return cls(idx, A[idx[:, 0], idx[:, 1]], A.shape, device=device)

    @classmethod
    def from_scipy(cls, mat, device=None):
        mat = mat.tocsr()
        row, col = mat.nonzero()
        return cls(np.stack([row, col], axis=1), mat.data, mat.shape, device=device)

    def __repr__(self):
        return f'SparseInteractions(nnz={self.nnz}, dense_shape={self.dense_shape}, device={self.device})'


class IndicatorFeatures:
    """Identity feature matrix of size n (every user / item is its own feature)."""

    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n, self.n)

    def to_dense(self, device=None):
        return torch.eye(self.n, dtype=torch.float32, device=default_device() if device is None else device)

    def __repr__(self):
        return f'IndicatorFeatures({self.n})'


def eye(n):
    """Drop-in for ``tf.eye(n)`` as the reference's examples use it for indicator features."""
    return IndicatorFeatures(n)


def is_indicator(features):
    """True for IndicatorFeatures and for a dense square 0/1 matrix equal to the identity."""
    if isinstance(features, IndicatorFeatures):
        return True
    if torch.is_tensor(features) or isinstance(features, np.ndarray):
        f = torch.as_tensor(features)
        if f.dim() == 2 and f.shape[0] == f.shape[1]:
            n = f.shape[0]
            return bool((f.diagonal() == 1).all()) and int(torch.count_nonzero(f)) == n
    return False
