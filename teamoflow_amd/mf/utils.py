"""Helpers of /root/reference/src/teamoflow/mf/utils.py on torch tensors."""
import numpy as np
import torch
from scipy import sparse as sp

from .. import _ops
from .sparse import SparseInteractions, default_device


def random_sampler(n_items, n_users, n_samples, replace=False):
    """utils.py:8-22.  One ``np.random.choice`` per user from the GLOBAL NumPy RNG, so the same
    ``np.random.seed`` gives the table the reference would draw.  int64 [n_users, n_samples]."""
    items_per_user = [np.random.choice(a=n_items, size=n_samples, replace=replace) for _ in range(n_users)]
    return torch.as_tensor(np.array(items_per_user), dtype=torch.int64).to(default_device())


def random_sampler_device(n_items, n_users, n_samples, seed=0, device=None, rows_per_block=65536):
    """Extension for tables too large for the host loop above (1M users x 1024 samples): distinct
    items per user drawn on the device.  Not the NumPy stream - use ``random_sampler`` for parity."""
    if n_samples > n_items:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    device = default_device() if device is None else torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty(n_users, n_samples, dtype=torch.int32, device=device)
    if 2 * n_samples > n_items:
        # most of the catalog per row: rejection would take coupon-collector time - the first n_samples entries of a random
        # permutation instead (argsort of uniform keys), in blocks of at most 2^26 keys
        step = max(1, (1 << 26) // n_items)
        for r0 in range(0, n_users, step):
            rows = min(step, n_users - r0)
            keys = torch.rand(rows, n_items, device=device, generator=gen)
            out[r0:r0 + rows] = torch.argsort(keys, dim=1)[:, :n_samples].to(torch.int32)
        return out
    for r0 in range(0, n_users, rows_per_block):
        rows = min(rows_per_block, n_users - r0)
        blk = torch.randint(0, n_items, (rows, n_samples), device=device, generator=gen, dtype=torch.int32)
        for _ in range(64):
            srt, perm = torch.sort(blk, dim=1)
            dup = torch.zeros_like(srt, dtype=torch.bool)
            dup[:, 1:] = srt[:, 1:] == srt[:, :-1]
            ndup = int(dup.sum())
            if ndup == 0:
                break
            srt[dup] = torch.randint(0, n_items, (ndup,), device=device, generator=gen, dtype=torch.int32)
            blk = srt
        else:
            raise RuntimeError('could not draw distinct samples')
        shuffle = torch.argsort(torch.rand(rows, n_samples, device=device, generator=gen), dim=1)
        out[r0:r0 + rows] = torch.gather(blk, 1, shuffle)
    return out


def generate_random_interaction(n_users, n_items, min_val=0.0, max_val=5.0, density=0.50):
    """utils.py:25-59: scipy.sparse.random -> affine rescale -> round -> (sparse, dense) pair."""
    p = sp.random(n_users, n_items, density=density)
    p = (max_val - min_val) * p + min_val * p.ceil()
    random_arr = np.round(p.toarray())
    interactions = SparseInteractions.from_scipy(sp.csr_matrix(random_arr))
    A = torch.as_tensor(random_arr, dtype=torch.float32).to(default_device())
    return interactions, A


def gather_matrix_indices(input_arr, index_arr):
    """utils.py:62-105: out[i, c] = input_arr[i, index_arr[i, c]] (torch.gather along dim 1)."""
    return _ops.gather_rows_cols(input_arr, index_arr)
