"""The reference's own test suite (/root/reference/test/*.py), scenario by scenario, against this package on the GPU.

The reference's tests are smoke tests wrapped in bare `except:` blocks (they print on failure and pass regardless), plus one
known-answer test.  Here every scenario is the same call sequence with the same shapes and arguments, through the same import
root (`src.teamoflow...`), WITHOUT the except blocks - an exception fails the test - and with the checks the scenario allows:
finite, decreasing loss trajectories, result types and shapes, and the dense oracle's trajectory where the oracle implements the
scenario (the oracle needs the start the model drew, so those run from a FixedInitializer copy of it).

  test_loss.py:29-72         -> test_mseloss / test_wmrbloss / test_kl_divergence
  test_embedding.py:26-66    -> test_linear_repr / test_biased_linear_repr / test_ReLU_repr
  test_initializer.py:26-53  -> test_normalinitializer / test_uniforminitializer
  test_predict.py:26-38      -> test_dot_product_prediction
  test_utils.py:18-81        -> test_random_sampler / test_gather_matrix_indices / test_generate_random_interactions
The only substitution: tf.eye(n) -> torch.eye(n) (a dense identity, detected and treated as indicator features).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

# the reference's import lines (test_loss.py:5-7, test_embedding.py, test_initializer.py, test_predict.py, test_utils.py:5)
from src.teamoflow.mf.embedding_graphs import *      # noqa: E402,F401,F403
from src.teamoflow.mf.initializer_graphs import *    # noqa: E402,F401,F403
from src.teamoflow.mf.loss_graphs import *           # noqa: E402,F401,F403
from src.teamoflow.mf.predict_graphs import *        # noqa: E402,F401,F403
from src.teamoflow.mf.utils import *                 # noqa: E402,F401,F403
from src.teamoflow.mf.utils import generate_random_interaction  # noqa: E402
from src.teamoflow.mf.matrix_factorization import MatrixFactorization  # noqa: E402
from src.teamoflow.mf.sparse import SparseInteractions  # noqa: E402


@pytest.fixture(scope='module')
def data():
    # "load the data outside of the class" (test_loss.py:13-22)
    np.random.seed(0)
    n_users, n_items = 50, 100
    sparse_interaction, dense_interaction = generate_random_interaction(n_users=n_users, n_items=n_items, density=0.05)
    sparse_mixed, dense_mixed = generate_random_interaction(n_users=n_users, n_items=n_items, min_val=-5.0, max_val=5.0, density=0.01)
    return dict(n_users=n_users, n_items=n_items, n_samples=n_items // 2, sparse=sparse_interaction, dense=dense_interaction,
                sparse_mixed=sparse_mixed, user_features=torch.eye(n_users), item_features=torch.eye(n_items))


def fitted(model, d, inter, **kw):
    model.verbose = False
    model.fit(epochs=25, user_features=d['user_features'], item_features=d['item_features'], tf_interactions=inter, **kw)
    h = np.array(model.loss_history_)
    assert h.shape == (25,) and np.isfinite(h).all()
    assert tuple(model.user_embedding.shape) == (d['n_users'], 3) and tuple(model.item_embedding.shape) == (d['n_items'], 3)
    return h


def test_mseloss(data):
    h = fitted(MatrixFactorization(3), data, data['sparse'])
    assert h[-1] < h[0]


def test_wmrbloss(data):
    from oracle import dense_ref as D
    from src.teamoflow.mf.initializer_graphs import FixedInitializer
    np.random.seed(1)
    mf_model = MatrixFactorization(3, loss_graph=WMRBLoss(), n_users=data['n_users'], n_items=data['n_items'], generate_sample=True)  # noqa: F405
    assert tuple(mf_model.random_ind.shape) == (data['n_users'], data['n_items'] // 2)      # class default n_items // 2 (:68-69)
    h = fitted(mf_model, data, data['sparse'], lr=0.1)
    assert h[-1] < h[0]
    # the same scenario from a pinned start against the dense oracle (same negative table)
    U0 = (np.random.default_rng(0).random((data['n_users'], 3)) / 10).astype(np.float32)
    V0 = (np.random.default_rng(1).random((data['n_items'], 3)) / 10).astype(np.float32)
    pinned = MatrixFactorization(3, loss_graph=WMRBLoss(), n_users=data['n_users'], n_items=data['n_items'],  # noqa: F405
                                 user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
    pinned.random_ind = mf_model.random_ind
    hp = fitted(pinned, data, data['sparse'], lr=0.1)
    idx, val = data['sparse'].indices.cpu().numpy(), data['sparse'].values.cpu().numpy()
    ref = D.fit_dense(U0, V0, idx, val, 'wmrb', 25, 0.1, random_ind=mf_model.random_ind.cpu().numpy(), n_items=data['n_items'],
                      n_samples=data['n_items'] // 2)
    assert rel_err(hp[:5], ref['loss'][:5]) < 1e-5 and rel_err(hp, ref['loss']) < 5e-3


def test_kl_divergence(data):
    h = fitted(MatrixFactorization(3, loss_graph=KLDivergenceLoss()), data, data['sparse_mixed'], lr=0.1)  # noqa: F405
    assert h[-1] <= h[0]


@pytest.mark.parametrize('graph', ['LinearEmbedding', 'BiasedLinearEmbedding', 'ReLUEmbedding'])
def test_user_repr_graphs(data, graph):
    """test_linear_repr / test_biased_linear_repr / test_ReLU_repr: only the USER embedding graph is changed."""
    h = fitted(MatrixFactorization(3, user_repr_graph=globals()[graph]()), data, data['sparse'])
    assert h[-1] < h[0]


@pytest.mark.parametrize('init', ['NormalInitializer', 'UniformInitializer'])
def test_user_weight_graphs(data, init):
    """test_normalinitializer / test_uniforminitializer: only the USER initializer is changed."""
    h = fitted(MatrixFactorization(3, user_weight_graph=globals()[init]()), data, data['sparse'])
    assert h[-1] < h[0]


def test_dot_product_prediction(data):
    from oracle import dense_ref as D
    mf_model = MatrixFactorization(3)
    fitted(mf_model, data, data['sparse'])
    prediction = mf_model.predict()
    assert prediction is not None and tuple(prediction.shape) == (data['n_users'], data['n_items'])
    want = D.predict_dense(mf_model.user_embedding.cpu().numpy(), mf_model.item_embedding.cpu().numpy())
    assert rel_err(prediction.cpu().numpy(), want) < 1e-5
    both = mf_model.predict(data['dense'])                      # :195-201: (all, the unobserved ones)
    assert isinstance(both, tuple) and both[1].numel() == int((data['dense'] == 0).sum())


def test_random_sampler():
    n_users, n_items = 50, 100
    n_samples = n_items // 2
    arr = random_sampler(n_items, n_users, n_samples)  # noqa: F405
    assert torch.is_tensor(arr) and tuple(arr.shape) == (n_users, n_samples) and arr.dtype == torch.int64
    assert all(len(set(row.tolist())) == n_samples for row in arr.cpu())


def test_gather_matrix_indices():
    input_arr = torch.tensor([[1, 4, 2], [5, 7, 8], [6, 2, 1]], dtype=torch.float32)
    index_arr = torch.tensor([[0, 2, 0], [2, 2, 2], [2, 1, 0]], dtype=torch.int64)
    result = torch.tensor([[1, 2, 1], [8, 8, 8], [1, 2, 6]], dtype=torch.float32)
    test_arr = gather_matrix_indices(input_arr, index_arr)  # noqa: F405
    assert torch.equal(test_arr.cpu(), result)


def test_generate_random_interactions():
    n_users, n_items = 50, 100
    sparse_int, dense_int = generate_random_interaction(n_users, n_items, density=0.05)
    assert isinstance(sparse_int, SparseInteractions) and torch.is_tensor(dense_int)
    assert tuple(sparse_int.dense_shape) == tuple(dense_int.shape)
    assert torch.equal(sparse_int.to_dense().cpu(), dense_int.cpu())     # "consistent with one another"
