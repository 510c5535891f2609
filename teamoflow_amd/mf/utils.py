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


def _i64(x):
    """Python int -> the int64 value with the same low 64 bits (torch integer arithmetic wraps)."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _srl(x, s):
    """Logical right shift of an int64 tensor (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix64(x):
    """splitmix64's finaliser on an int64 tensor: a bijection of the 64-bit words with full avalanche."""
    x = (x ^ _srl(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _srl(x, 27)) * _i64(0x94D049BB133111EB)
    return x ^ _srl(x, 31)


def _affine(x, mul, add):
    """x * mul + add in wrapping 64-bit arithmetic, for an int or an int64 tensor x."""
    if torch.is_tensor(x):
        return x * _i64(mul) + _i64(add)
    return torch.tensor(_i64(int(x) * mul + add), dtype=torch.int64)


def counter_hash(seed, a, b=0, c=0):
    """Counter-based random words: a 64-bit hash of (seed, a, b, c), each an int or a broadcastable int64 tensor.  Stateless,
    so the word of (a, b, c) does not depend on which other words are drawn with it - a rank that draws only its own
    users' rows gets exactly the rows of the whole table."""
    h = _mix64(_affine(a, 1, (2 * int(seed) + 1) * 0x9E3779B97F4A7C15))
    k = _affine(b, 0xD6E8FEB86659FD93, 0xA0761D6478BD642F)
    h = _mix64(h ^ k.to(h.device))
    k = _affine(c, 0xE7037ED1A0B428DB, 0x8EBC6AF09C88C6E3)
    return _mix64(h ^ k.to(h.device))


def hash_below(h, n):
    """Random words -> integers in [0, n) (53 random bits; the modulo bias is below n / 2^53)."""
    return _srl(h, 11) % int(n)


def hash_unit(h):
    """Random words -> float64 in (0, 1)."""
    return (_srl(h, 11).to(torch.float64) + 0.5) * (2.0 ** -53)


def random_sampler_device(n_items, n_users, n_samples, seed=0, device=None, rows_per_block=32768, user_offset=0):
    """Extension for tables too large for the host loop above (1M users x 1024 samples): distinct
    items per user drawn on the device.  Not the NumPy stream - use ``random_sampler`` for parity.
    The row of user u is a function of (seed, u) alone (counter_hash): ``user_offset=b`` returns rows b .. b + n_users of
    the table, so the ranks of a user-partitioned job each draw their own block of ONE table."""
    if n_samples > n_items:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    device = default_device() if device is None else torch.device(device)
    out = torch.empty(n_users, n_samples, dtype=torch.int32, device=device)
    if 2 * n_samples > n_items:
        # most of the catalog per row: rejection would take coupon-collector time - the first n_samples entries of a random
        # permutation instead (argsort of per-(user, item) keys), in blocks of at most 2^25 keys
        step = max(1, (1 << 25) // n_items)
        items = torch.arange(n_items, device=device, dtype=torch.int64)[None, :]
        for r0 in range(0, n_users, step):
            rows = min(step, n_users - r0)
            u = torch.arange(user_offset + r0, user_offset + r0 + rows, device=device, dtype=torch.int64)[:, None]
            keys = _srl(counter_hash(seed, u, items, 1), 1)
            out[r0:r0 + rows] = torch.argsort(keys, dim=1, stable=True)[:, :n_samples].to(torch.int32)
        return out
    rows_per_block = max(1, min(rows_per_block, (1 << 25) // n_samples))
    pos = torch.arange(n_samples, device=device, dtype=torch.int64)[None, :]
    for r0 in range(0, n_users, rows_per_block):
        rows = min(rows_per_block, n_users - r0)
        u = torch.arange(user_offset + r0, user_offset + r0 + rows, device=device, dtype=torch.int64)[:, None]
        # every row is kept sorted between rounds and a repeated item is redrawn from (user, position, round): what happens
        # to a row never depends on the other rows of the block (a row without repeats is a fixed point of a round)
        blk = torch.sort(hash_below(counter_hash(seed, u, pos, 2), n_items), dim=1)[0]
        for rnd in range(3, 67):
            dup = torch.zeros_like(blk, dtype=torch.bool)
            dup[:, 1:] = blk[:, 1:] == blk[:, :-1]
            if not bool(dup.any()):
                break
            blk = torch.sort(torch.where(dup, hash_below(counter_hash(seed, u, pos, rnd), n_items), blk), dim=1)[0]
        else:
            raise RuntimeError('could not draw distinct samples')
        shuffle = torch.argsort(_srl(counter_hash(seed, u, pos, 0), 1), dim=1, stable=True)
        out[r0:r0 + rows] = torch.gather(blk, 1, shuffle).to(torch.int32)
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
