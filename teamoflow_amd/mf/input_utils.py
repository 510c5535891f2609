"""Input helpers with the names and behaviour of /root/reference/src/teamoflow/mf/input_utils.py
(host-side data preparation, not on the timed path).  Everything that was a ``tf.sparse.SparseTensor``
becomes a ``SparseInteractions`` (row-major COO int64 + fp32), every ``tf.Tensor`` a torch tensor.
"""
import random

import numpy as np
import torch
from scipy import sparse as sp

from .sparse import SparseInteractions, default_device

try:  # pandas is optional at import time, like every other converter input
    import pandas as pd
except ImportError:  # pragma: no cover
    pd = None


def create_iterable_interaction(df):
    """input_utils.py:10-23.  Maps the raw 'User ID' / 'Items' columns to dense 0..n-1 ids IN PLACE (first
    appearance order, like ``Series.unique``) and returns (rows as lists, n_users, n_items)."""
    users = {raw: i for i, raw in enumerate(df['User ID'].unique())}
    items = {raw: i for i, raw in enumerate(df['Items'].unique())}
    df['User ID'] = df['User ID'].map(users)
    df['Items'] = df['Items'].map(items)
    return df.values.tolist(), len(users), len(items)


def mask_train_test_split(interactions, n_users, n_items, test_size=0.2, shuffle=True, return_indices=True):
    """input_utils.py:26-79.  ``interactions`` = [[row, col, rating], ...]; shuffled IN PLACE with the global
    ``random`` module (as the reference does), the first (1 - test_size) share is the train mask.  Both CSR
    matrices keep the full [n_users, n_items] shape."""
    if shuffle:
        random.shuffle(interactions)
    cut = int((1.0 - test_size) * len(interactions))
    parts = []
    for chunk in (interactions[:cut], interactions[cut:]):
        rows = [r for r, _, _ in chunk]
        cols = [c for _, c, _ in chunk]
        vals = [v for _, _, v in chunk]
        parts.append((sp.csr_matrix((vals, (rows, cols)), shape=(n_users, n_items)), list(zip(zip(rows, cols), vals))))
    (train, train_idx), (test, test_idx) = parts
    if return_indices:
        return train, test, train_idx, test_idx
    return train, test


def test_sparse_transformation(sparse_interactions, li_indices):
    """input_utils.py:82-104: True when the sparse matrix holds the listed ((row, col), value) entries.
    (The reference's loop can only ever return True - SURVEY.md A.4; this one really checks, which gives
    the same answer on consistent input.)"""
    dense = sparse_interactions.toarray()
    return all(dense[int(r), int(c)] == v for (r, c), v in li_indices)


test_sparse_transformation.__test__ = False  # not a pytest test despite the reference's name


def df_to_sparse_pipeline(df, test_size=0.2):
    """input_utils.py:107-130: DataFrame -> (train, test) CSR matrices.  Like the reference, the split ratio
    is fixed at 0.2 whatever ``test_size`` says (:119)."""
    rows, n_users, n_items = create_iterable_interaction(df)
    train, test, train_idx, test_idx = mask_train_test_split(rows, n_users, n_items, test_size=0.2, shuffle=True,
                                                             return_indices=True)
    if test_sparse_transformation(train, train_idx) and test_sparse_transformation(test, test_idx):
        return train, test
    print('Please check your input for errors.')
    return None


def convert_np_to_tf_sparse(np_arr, device=None):
    """input_utils.py:133-153: dense array -> sparse interactions (row-major nonzeros via scipy CSR)."""
    return SparseInteractions.from_scipy(sp.csr_matrix(np.asarray(np_arr)), device=device)


def convert_tf_to_tf_sparse(tf_arr, device=None):
    """input_utils.py:156-161 (a dense torch tensor here)."""
    return convert_np_to_tf_sparse(torch.as_tensor(tf_arr).detach().cpu().numpy(), device=device)


def convert_list_to_tf_sparse(li_arr, device=None):
    """input_utils.py:164-169."""
    return convert_np_to_tf_sparse(np.array(li_arr), device=device)


def convert_df_to_tf_sparse(df_arr, device=None):
    """input_utils.py:172-177."""
    return convert_np_to_tf_sparse(np.array(df_arr), device=device)


def convert_sp_sparse_to_tf_sparse(sp_arr, device=None):
    """input_utils.py:180-198."""
    return SparseInteractions.from_scipy(sp_arr, device=device)


def convert_to_tf_sparse(arr, device=None):
    """input_utils.py:201-220: list / ndarray / DataFrame / dense tensor / scipy CSR -> sparse interactions
    (None for anything else, like the reference's fall-through)."""
    if isinstance(arr, SparseInteractions):
        return arr if device is None else arr.to(device)
    if isinstance(arr, list):
        return convert_list_to_tf_sparse(arr, device)
    if isinstance(arr, np.ndarray):
        return convert_np_to_tf_sparse(arr, device)
    if pd is not None and isinstance(arr, pd.DataFrame):
        return convert_df_to_tf_sparse(arr, device)
    if torch.is_tensor(arr):
        return convert_tf_to_tf_sparse(arr, device)
    if sp.issparse(arr):
        return convert_sp_sparse_to_tf_sparse(arr.tocsr(), device)
    return None


convert_to_sparse = convert_to_tf_sparse


def convert_to_tensor_constant(A, device=None):
    """input_utils.py:223-241: list / ndarray / Series / DataFrame -> fp32 tensor; a tensor passes through."""
    if torch.is_tensor(A):
        return A
    dev = default_device() if device is None else device
    if isinstance(A, (list, np.ndarray)) or (pd is not None and isinstance(A, (pd.Series, pd.DataFrame))):
        return torch.as_tensor(np.asarray(A), dtype=torch.float32).to(dev)
    return None


def convert_to_tensor_trainable(arr, device=None):
    """input_utils.py:244-253 (the reference's stray ``self`` parameter is dropped)."""
    return convert_to_tensor_constant(arr, device).clone().requires_grad_(True)
