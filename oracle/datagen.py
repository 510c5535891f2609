"""CPU restatement of the reference's synthetic-input helpers (NumPy/SciPy cores only).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

  generate_random_interaction  /root/reference/src/teamoflow/mf/utils.py:37-59
  random_sampler               /root/reference/src/teamoflow/mf/utils.py:20-22
Both consume the GLOBAL NumPy RNG exactly like the reference, so the same
``np.random.seed`` yields the same arrays the reference would wrap in TF tensors.
"""
import numpy as np
from scipy import sparse


def generate_random_interaction(n_users, n_items, min_val=0.0, max_val=5.0, density=0.50):
    """Returns (indices [nnz, 2] int64 row-major, values [nnz] fp32, dense_shape, A dense fp32)."""
    p = sparse.random(n_users, n_items, density=density)  # utils.py:37
    p = (max_val - min_val) * p + min_val * p.ceil()  # :39
    random_arr = np.round(p.toarray())  # :41
    csr = sparse.csr_matrix(random_arr)  # :43 (drops the entries rounded to 0)
    row, col = csr.nonzero()  # :53
    indices = np.stack([row, col], axis=1).astype(np.int64)  # :55
    return indices, csr.data.astype(np.float32), (n_users, n_items), random_arr.astype(np.float32)


def random_sampler(n_items, n_users, n_samples, replace=False):
    """utils.py:20-22: one np.random.choice per user, global RNG, int64 [n_users, n_samples]."""
    items_per_user = [np.random.choice(a=n_items, size=n_samples, replace=replace) for _ in range(n_users)]
    return np.array(items_per_user).astype(np.int64)


def normal_init(rows, r, seed):
    """Deterministic stand-in for NormalInitializer (TF's RNG stream is not reproducible here):
    N(0,1) from torch.Generator(seed) then the same whole-matrix L2 normalisation
    (initializer_graphs.py:34)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, r, generator=g, dtype=torch.float32)
    return (x * torch.rsqrt(torch.clamp((x * x).sum(), min=1e-12))).numpy()


def uniform_init(rows, r, seed):
    """Stand-in for UniformInitializer (initializer_graphs.py:51)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(rows, r, generator=g, dtype=torch.float32)
    return (x * torch.rsqrt(torch.clamp((x * x).sum(), min=1e-12))).numpy()
