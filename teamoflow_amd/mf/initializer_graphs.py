"""Weight factories (plug-in interface of /root/reference/src/teamoflow/mf/initializer_graphs.py).

Both built-ins draw a random matrix and divide it by its WHOLE-MATRIX L2 norm
(tf.math.l2_normalize with axis=None, eps 1e-12 - initializer_graphs.py:34,51).
The random stream is torch's, not TensorFlow's Philox stream, which cannot be reproduced; parity
tests inject fixed weights through a custom ``Initializer`` (that is what the plug-in point is for).
"""
from abc import ABC, abstractmethod

import torch

from .sparse import default_device


def _l2_normalize_all(x, eps=1e-12):
    return x * torch.rsqrt(torch.clamp((x * x).sum(), min=eps))


class Initializer(ABC):
    """initializer_graphs.py:7-19."""

    @abstractmethod
    def initialize_weights(self, n_features, n_components):
        pass


class NormalInitializer(Initializer):
    """N(0, 1) sample, globally L2-normalised (initializer_graphs.py:22-35)."""

    def __init__(self, seed=None):
        self.seed = seed

    def _generator(self, device):
        if self.seed is None:
            return None
        return torch.Generator(device=device).manual_seed(int(self.seed))

    def initialize_weights(self, n_features, n_components):
        dev = default_device()
        x = torch.randn(n_features, n_components, dtype=torch.float32, device=dev, generator=self._generator(dev))
        return _l2_normalize_all(x).requires_grad_(True)


class UniformInitializer(NormalInitializer):
    """U[0, 1) sample, globally L2-normalised (initializer_graphs.py:38-52)."""

    def initialize_weights(self, n_features, n_components):
        dev = default_device()
        x = torch.rand(n_features, n_components, dtype=torch.float32, device=dev, generator=self._generator(dev))
        return _l2_normalize_all(x).requires_grad_(True)


class FixedInitializer(Initializer):
    """Returns the given matrix (extension: how tests and benchmarks pin U0 / V0)."""

    def __init__(self, weights):
        self.weights = weights

    def initialize_weights(self, n_features, n_components):
        w = torch.as_tensor(self.weights, dtype=torch.float32).detach().clone().to(default_device())
        if tuple(w.shape) != (n_features, n_components):
            raise ValueError(f'FixedInitializer holds {tuple(w.shape)}, asked for {(n_features, n_components)}')
        return w.requires_grad_(True)
