"""Input converters of /root/reference/src/teamoflow/mf/input_utils.py (host-side data preparation,
not on the timed path): everything lands in ``SparseInteractions`` (row-major COO int64 + fp32)."""
import numpy as np
import torch
from scipy import sparse as sp

from .sparse import SparseInteractions, default_device


def df_to_interaction_triples(df, user_col, item_col, rating_col):
    """input_utils.py:10-23: map raw user / item ids to dense 0..n-1 ids; returns (rows, cols, vals, maps)."""
    users, u_inv = np.unique(df[user_col].to_numpy(), return_inverse=True)
    items, i_inv = np.unique(df[item_col].to_numpy(), return_inverse=True)
    return u_inv, i_inv, df[rating_col].to_numpy(dtype=np.float32), (users, items)


def mask_train_test_split(rows, cols, vals, shape, test_size=0.2, seed=None):
    """input_utils.py:26-79: random mask over the observed entries -> two CSR matrices of the full shape."""
    rng = np.random.default_rng(seed)
    mask = rng.random(len(vals)) < test_size
    train = sp.csr_matrix((vals[~mask], (rows[~mask], cols[~mask])), shape=shape)
    test = sp.csr_matrix((vals[mask], (rows[mask], cols[mask])), shape=shape)
    return train, test


def convert_to_sparse(data, device=None):
    """input_utils.py:133-220 (convert_*_to_tf_sparse family): numpy / nested list / DataFrame /
    torch dense / scipy sparse -> SparseInteractions."""
    if isinstance(data, SparseInteractions):
        return data if device is None else data.to(device)
    if sp.issparse(data):
        return SparseInteractions.from_scipy(data, device=device)
    if hasattr(data, 'to_numpy'):
        data = data.to_numpy()
    if torch.is_tensor(data):
        if data.is_sparse:
            c = data.coalesce()
            return SparseInteractions(c.indices().T, c.values(), c.shape, device=device)
        return SparseInteractions.from_dense(data, device=device)
    return SparseInteractions.from_dense(np.asarray(data, dtype=np.float32), device=device)


convert_to_tf_sparse = convert_to_sparse  # reference name


def convert_to_tensor_constant(data, device=None):
    """input_utils.py:223-237."""
    return torch.as_tensor(np.asarray(data), dtype=torch.float32).to(default_device() if device is None else device)


def convert_to_tensor_trainable(data, device=None):
    """input_utils.py:240-253."""
    return convert_to_tensor_constant(data, device).requires_grad_(True)
