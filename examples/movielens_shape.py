"""MovieLens-shaped benchmark in the call order of the reference's examples/benchmarking_ML.py (:35-175), written against this
package.  The data set itself cannot be downloaded here, so the ratings DataFrame is synthetic with the same columns the
reference builds from ratings.csv ('User ID', 'Items', rating) - raw ids are sparse and unordered like MovieLens ids, ratings
are 0.5 .. 5.0 in half steps.  From there on every call is the reference's:

    df_to_sparse_pipeline(df)                          train / test CSR masks            (input_utils.py:107-130)
    .multiply(ratings >= 4.0)                          the "4plus" views                 (benchmarking_ML.py:40-41)
    convert_to_tf_sparse(train.toarray())              interactions for fit()            (:54, :63)
    MatrixFactorization(n_components)                  MSE rating model, lr 1e-3         (:73, :100)
    MatrixFactorization(..., UniformInitializer(), WMRBLoss(), n_samples = n_items // 5, generate_sample=True), lr 0.1   (:76-78, :102)
    recall_at_k(A, k) for k = 10 / 30 / 50 on train, test and their >= 4 views          (:109-175)

    python examples/movielens_shape.py [100k | 1m | n_users n_items n_ratings] [--components 5] [--epochs 100] [--biased]

`--biased` also trains the reference's third model (BiasedLinearEmbedding, :81-86), which takes the generic autograd path.
"""
import argparse
import os
import random
import sys

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from teamoflow.mf.embedding_graphs import BiasedLinearEmbedding              # noqa: E402
from teamoflow.mf.initializer_graphs import NormalInitializer, UniformInitializer  # noqa: E402
from teamoflow.mf.input_utils import convert_to_tf_sparse, df_to_sparse_pipeline   # noqa: E402
from teamoflow.mf.loss_graphs import WMRBLoss                                # noqa: E402
from teamoflow.mf.matrix_factorization import MatrixFactorization            # noqa: E402
from teamoflow.mf.sparse import eye                                          # noqa: E402

SHAPES = {'100k': (943, 1682, 100_000), '1m': (6040, 3706, 1_000_209)}   # BASELINE configs 2 and 3


def synthetic_ratings(n_users, n_items, n_ratings, seed=0):
    """A ratings frame of the MovieLens form: unique (user, item) pairs, popular items and active users over-represented,
    raw ids that are neither dense nor ordered.  Every user and every item appears at least once."""
    rng = np.random.default_rng(seed)
    n_ratings = min(n_ratings, n_users * n_items // 2)
    pu = rng.lognormal(0.0, 0.8, n_users)
    pi = 1.0 / np.arange(1, n_items + 1) ** 0.8
    pairs = np.stack([np.arange(n_users * n_items) // n_items, np.arange(n_users * n_items) % n_items], axis=1) \
        if n_users * n_items <= 4_000_000 else None
    if pairs is not None:
        w = (pu[:, None] * pi[None, :]).ravel()
        pick = rng.choice(n_users * n_items, n_ratings, replace=False, p=w / w.sum())
        u, j = pairs[pick, 0], pairs[pick, 1]
    else:
        u = rng.choice(n_users, int(n_ratings * 1.6), p=pu / pu.sum())
        j = rng.choice(n_items, int(n_ratings * 1.6), p=pi / pi.sum())
        key = np.unique(u.astype(np.int64) * n_items + j)
        key = rng.permutation(key)[:n_ratings]
        u, j = key // n_items, key % n_items
    # every user / item at least once (df_to_sparse_pipeline sizes the matrices by the ids it has seen)
    miss_u = np.setdiff1d(np.arange(n_users), u)
    miss_j = np.setdiff1d(np.arange(n_items), j)
    u = np.concatenate([u, miss_u, rng.integers(0, n_users, len(miss_j))])
    j = np.concatenate([j, rng.integers(0, n_items, len(miss_u)), miss_j])
    key = np.unique(u.astype(np.int64) * n_items + j)
    u, j = key // n_items, key % n_items
    order = rng.permutation(len(u))
    raw_u = rng.permutation(n_users * 3)[:n_users] + 1         # sparse raw ids, like userId / movieId
    raw_j = rng.permutation(n_items * 50)[:n_items] + 1
    ratings = rng.integers(1, 11, len(u)) * 0.5
    return pd.DataFrame({'User ID': raw_u[u][order], 'Items': raw_j[j][order], 'rating': ratings[order]})


def run(n_users, n_items, n_ratings, n_components=5, epochs=100, seed=0, biased=False, initializers=None, verbose=True,
        ks=(10, 30, 50)):
    """The reference's benchmark on a synthetic frame.  `initializers` = {'mse': (user, item), 'wmrb': (user, item)} replaces
    the random initialisers (tests inject the oracle's start); the split uses the global `random` stream and the negative
    table the global NumPy stream, exactly like the reference, so seeding both reproduces a run."""
    random.seed(seed)
    np.random.seed(seed)
    df = synthetic_ratings(n_users, n_items, n_ratings, seed)
    sparse_train_ratings, sparse_test_ratings = df_to_sparse_pipeline(df, test_size=0.25)
    sparse_train_ratings_4plus = sparse_train_ratings.multiply(sparse_train_ratings >= 4.0)
    sparse_test_ratings_4plus = sparse_test_ratings.multiply(sparse_test_ratings >= 4.0)

    train_np, test_np = sparse_train_ratings.toarray(), sparse_test_ratings.toarray()
    train_np_4plus, test_np_4plus = sparse_train_ratings_4plus.toarray(), sparse_test_ratings_4plus.toarray()
    tf_train = convert_to_tf_sparse(train_np)
    tf_train_4plus = convert_to_tf_sparse(train_np_4plus)
    A = {'train': torch.tensor(train_np, dtype=torch.float32), 'test': torch.tensor(test_np, dtype=torch.float32),
         'train_4plus': torch.tensor(train_np_4plus, dtype=torch.float32), 'test_4plus': torch.tensor(test_np_4plus, dtype=torch.float32)}

    n_users, n_items = train_np.shape
    n_sampled_items = n_items // 5
    init = initializers or {}
    mse_u, mse_i = init.get('mse', (NormalInitializer(), NormalInitializer()))
    wm_u, wm_i = init.get('wmrb', (UniformInitializer(), UniformInitializer()))
    models = {'mse': MatrixFactorization(n_components, user_weight_graph=mse_u, item_weight_graph=mse_i),
              'wmrb': MatrixFactorization(n_components, user_weight_graph=wm_u, item_weight_graph=wm_i, loss_graph=WMRBLoss(),
                                          n_users=n_users, n_items=n_items, n_samples=n_sampled_items, generate_sample=True)}
    if biased:
        models['wmrb_biased'] = MatrixFactorization(n_components, user_weight_graph=UniformInitializer(), item_weight_graph=UniformInitializer(),
                                                    loss_graph=WMRBLoss(), user_repr_graph=BiasedLinearEmbedding(),
                                                    item_repr_graph=BiasedLinearEmbedding(), n_users=n_users, n_items=n_items,
                                                    n_samples=n_sampled_items, generate_sample=True)
    user_features, item_features = eye(n_users), eye(n_items)
    for model in models.values():
        model.verbose = verbose
    models['mse'].fit(epochs, user_features, item_features, tf_train, lr=1e-3)
    models['wmrb'].fit(epochs, user_features, item_features, tf_train_4plus, lr=0.1)
    if biased:
        models['wmrb_biased'].fit(epochs, torch.eye(n_users), torch.eye(n_items), tf_train_4plus, lr=0.1)

    recalls = {}
    for name, model in models.items():
        for split, mat in A.items():
            for k in (ks if split == 'test_4plus' else ks[:1]):
                recalls[(name, split, k)] = float(model.recall_at_k(mat, k).mean())
    if verbose:
        for split, label in (('train', 'training set'), ('test', 'testing set'), ('train_4plus', 'training set (ratings >= 4)'),
                             ('test_4plus', 'testing set (ratings >= 4)')):
            for k in (ks if split == 'test_4plus' else ks[:1]):
                for name in models:
                    print(f'Recall @ {k} on {label} w/ {name.upper()}: {recalls[(name, split, k)]}')
            print()
    return dict(models=models, recalls=recalls, A=A, train=tf_train, train_4plus=tf_train_4plus, n_samples=n_sampled_items)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('shape', nargs='*', default=['100k'])
    ap.add_argument('--components', type=int, default=5)
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--biased', action='store_true')
    a = ap.parse_args()
    dims = SHAPES[a.shape[0]] if a.shape[0] in SHAPES else tuple(int(x) for x in a.shape[:3])
    run(*dims, n_components=a.components, epochs=a.epochs, seed=a.seed, biased=a.biased)
