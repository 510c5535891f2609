"""CPU tests of the parts of the class surface that are NOT the HIP hot path:
the input pipeline and the generic autograd loop behind non-default plug-ins."""
import random

import numpy as np
import pandas as pd
import pytest
import torch
from scipy import sparse as sp

from teamoflow_amd.mf import input_utils as iu
from teamoflow_amd.mf.embedding_graphs import BiasedLinearEmbedding, LinearEmbedding, ReLUEmbedding
from teamoflow_amd.mf.initializer_graphs import FixedInitializer, NormalInitializer, UniformInitializer
from teamoflow_amd.mf.loss_graphs import KLDivergenceLoss, MSELoss, WMRBLoss
from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
from teamoflow_amd.mf.sparse import SparseInteractions, eye, is_indicator

cpu_only = pytest.mark.skipif(torch.cuda.is_available(), reason='generic path exercised on CPU torch here')


def test_input_pipeline_from_dataframe():
    rng = np.random.default_rng(0)
    df = pd.DataFrame({'User ID': rng.integers(100, 120, 300), 'Items': rng.integers(7000, 7040, 300),
                       'Ratings': rng.integers(1, 6, 300).astype(float)}).drop_duplicates(['User ID', 'Items'])
    n_rows = len(df)
    random.seed(0)
    train, test = iu.df_to_sparse_pipeline(df.copy())
    assert train.shape == test.shape == (df['User ID'].nunique(), df['Items'].nunique())
    assert train.nnz + test.nnz == n_rows and train.nnz == int(0.8 * n_rows)
    assert (train.multiply(test)).nnz == 0  # disjoint masks
    rows, nu, ni = iu.create_iterable_interaction(df.copy())
    tr, te, tri, tei = iu.mask_train_test_split(rows, nu, ni, test_size=0.25, shuffle=False)
    assert tr.nnz == int(0.75 * n_rows) and iu.test_sparse_transformation(tr, tri) and iu.test_sparse_transformation(te, tei)
    assert not iu.test_sparse_transformation(tr, [((0, 0), -123.0)])


def test_converters_agree():
    A = np.array([[0, 2.0, 0], [1.5, 0, 0], [0, 0, -3.0]], dtype=np.float32)
    want_idx, want_val = [[0, 1], [1, 0], [2, 2]], [2.0, 1.5, -3.0]
    for src in (A, A.tolist(), pd.DataFrame(A), torch.tensor(A), sp.csr_matrix(A)):
        s = iu.convert_to_tf_sparse(src, device='cpu')
        assert isinstance(s, SparseInteractions) and s.dense_shape == (3, 3)
        assert s.indices.tolist() == want_idx and s.values.tolist() == want_val and s.indices.dtype == torch.int64
        assert torch.equal(s.to_dense(), torch.tensor(A))
    assert iu.convert_to_tf_sparse('nope') is None
    c = iu.convert_to_tensor_constant([[1, 2], [3, 4]], device='cpu')
    assert c.dtype == torch.float32 and iu.convert_to_tensor_trainable(c).requires_grad


def test_indicator_detection():
    assert is_indicator(eye(5)) and is_indicator(torch.eye(4)) and is_indicator(np.eye(3))
    assert not is_indicator(torch.ones(3, 3)) and not is_indicator(torch.eye(3)[:, :2])
    f = torch.eye(3)
    f[0, 1] = 1
    assert not is_indicator(f)


@cpu_only
@pytest.mark.parametrize('emb', [BiasedLinearEmbedding, ReLUEmbedding])
def test_generic_path_with_other_embeddings(emb):
    torch.manual_seed(0)
    m, n, r = 12, 9, 4
    A = (torch.rand(m, n) < 0.4) * torch.randint(1, 6, (m, n)).float()
    inter = SparseInteractions.from_dense(A)
    model = MatrixFactorization(r, user_repr_graph=emb(), item_repr_graph=LinearEmbedding(),
                                user_weight_graph=NormalInitializer(seed=1), item_weight_graph=UniformInitializer(seed=2))
    model.verbose = False
    model.fit(30, torch.eye(m), torch.eye(n), inter, lr=0.05)
    assert model.loss_history_[-1] < model.loss_history_[0]
    assert model.user_embedding.shape == (m, r) and model.item_embedding.shape == (n, r)
    assert len(model.user_trainable) == (2 if emb is BiasedLinearEmbedding else 3)
    if emb is ReLUEmbedding:
        assert model.user_relu_weight.shape == (m, 5 * r) and model.user_aux_dim == 5 * r


@cpu_only
def test_generic_path_kl_and_dense_features_wmrb(golden):
    torch.manual_seed(1)
    m, n, r = 15, 11, 3
    A = ((torch.rand(m, n) < 0.5) * torch.randint(-5, 6, (m, n))).float()
    inter = SparseInteractions.from_dense(A)
    model = MatrixFactorization(r, loss_graph=KLDivergenceLoss())
    model.verbose = False
    model.fit(5, torch.eye(m), torch.eye(n), inter, lr=1e-3)
    assert len(model.loss_history_) == 5 and np.isfinite(model.loss_history_).all()
    # WMRB over dense NON-identity features goes through the generic loop and equals the dense oracle
    from oracle import dense_ref as D
    g = golden('wmrb_small')
    Fu = torch.eye(50) + 0.01 * torch.rand(50, 50)
    model = MatrixFactorization(3, loss_graph=WMRBLoss(), n_users=50, n_items=100, n_samples=50,
                                user_weight_graph=FixedInitializer(g['U0']), item_weight_graph=FixedInitializer(g['V0']))
    model.random_ind, model.verbose = torch.as_tensor(g['R']), False
    model.fit(3, Fu, torch.eye(100), SparseInteractions(g['indices'], g['values'], (50, 100)), lr=0.1)
    ref = D.fit_dense(g['U0'], g['V0'], g['indices'], g['values'], 'wmrb', 3, 0.1, random_ind=g['R'], n_items=100,
                      n_samples=50, user_features=Fu.numpy())
    assert np.abs(np.array(model.loss_history_) - ref['loss']).max() / ref['loss'].max() < 1e-5


def test_initializers_normalise_globally():
    for init in (NormalInitializer(seed=3), UniformInitializer(seed=3)):
        w = init.initialize_weights(40, 7)
        assert tuple(w.shape) == (40, 7) and w.requires_grad
        assert abs(float((w.detach() ** 2).sum()) - 1.0) < 1e-5  # whole-matrix L2 norm 1 (initializer_graphs.py:34,51)
    with pytest.raises(ValueError):
        FixedInitializer(np.zeros((3, 2))).initialize_weights(4, 2)


def test_save_and_load_roundtrip(tmp_path):
    """Extension: one-file persistence.  Tables, plug-in objects, the negative table and the loss history survive."""
    import numpy as np
    import torch
    from teamoflow.mf.initializer_graphs import UniformInitializer
    from teamoflow.mf.loss_graphs import WMRBLoss
    from teamoflow.mf.matrix_factorization import MatrixFactorization
    np.random.seed(3)
    model = MatrixFactorization(4, loss_graph=WMRBLoss(), user_weight_graph=UniformInitializer(), n_users=6, n_items=9,
                                n_samples=5, generate_sample=True)
    model.user_embedding = torch.arange(24, dtype=torch.float32).reshape(6, 4)
    model.item_embedding = torch.arange(36, dtype=torch.float32).reshape(9, 4) / 7
    model.loss_history_ = [3.0, 2.5]
    path = tmp_path / 'model.pt'
    model.save(path)
    back = MatrixFactorization.load(path, device='cpu')
    assert isinstance(back.loss_graph, WMRBLoss) and isinstance(back.user_weight_graph, UniformInitializer)
    assert (back.n_components, back.n_users, back.n_items, back.n_samples, back.generate_sample) == (4, 6, 9, 5, True)
    assert torch.equal(back.user_embedding, model.user_embedding) and torch.equal(back.item_embedding, model.item_embedding)
    assert torch.equal(torch.as_tensor(back.random_ind).cpu(), torch.as_tensor(model.random_ind).cpu())
    assert back.loss_history_ == [3.0, 2.5]
    model.save(path, include_samples=False)
    assert MatrixFactorization.load(path, device='cpu').random_ind is None
    # the file holds plain data only: it loads with weights_only=True (no code runs on load) ...
    blob = torch.load(path, map_location='cpu', weights_only=True)
    assert blob['config']['Loss'] == {'plugin': 'WMRBLoss', 'state': {}}
    # ... a FixedInitializer keeps its matrix, and a user-defined plug-in needs the explicit pickle opt-in on both sides
    from teamoflow.mf.initializer_graphs import FixedInitializer, Initializer
    model.item_weight_graph = FixedInitializer(np.arange(6, dtype=np.float32).reshape(3, 2))
    model.save(path)
    again = MatrixFactorization.load(path, device='cpu')
    assert torch.equal(torch.as_tensor(again.item_weight_graph.weights), torch.arange(6.).reshape(3, 2))
    model.user_weight_graph = _CustomInit()
    with pytest.raises(TypeError):
        model.save(path)
    model.save(path, allow_pickle=True)
    with pytest.raises(Exception):
        MatrixFactorization.load(path, device='cpu')
    assert isinstance(MatrixFactorization.load(path, device='cpu', allow_pickle=True).user_weight_graph, _CustomInit)


class _CustomInit:
    def initialize_weights(self, n_features, n_components):
        return torch.zeros(n_features, n_components)
