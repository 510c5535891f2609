"""Toy-data walkthrough in the call order of the reference's examples/benchmark_toydata.py (its MSE and WMRB
branches), written against this package: the only changes a user of the reference makes are the import root
and `sparse.eye(n)` in place of `tf.eye(n)` (a dense torch.eye(n) also works).

    python examples/toydata.py [mse|wmrb] [n_users n_items n_components]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from teamoflow.mf.loss_graphs import WMRBLoss                      # noqa: E402
from teamoflow.mf.matrix_factorization import MatrixFactorization  # noqa: E402
from teamoflow.mf.sparse import eye                                # noqa: E402
from teamoflow.mf.utils import generate_random_interaction         # noqa: E402


def main(loss='mse', n_users=300, n_items=1000, n_components=5):
    interactions, A = generate_random_interaction(n_users, n_items, min_val=0.0, max_val=5.0, density=0.01)
    user_features, item_features = eye(n_users), eye(n_items)
    if loss == 'mse':
        model = MatrixFactorization(n_components)
        model.fit(450, user_features, item_features, interactions)
        _, unobserved = model.predict(A)
        print(f'{unobserved.numel()} unobserved predictions')
        print(f'Recall at 10: {model.recall_at_k(A).mean()}')
        print(f'Precision at 10: {model.precision_at_k(A).mean()}')
        print(f'F1 at 10: {model.f1_at_k(A)}')
    else:
        model = MatrixFactorization(n_components=n_components, n_users=n_users, n_items=n_items,
                                    n_samples=n_items // 2, generate_sample=True, loss_graph=WMRBLoss())
        model.fit(100, user_features, item_features, interactions, lr=0.1)
        print(f'Recall @ 10 w/ WMRB: {model.recall_at_k(A, preserve_rows=True).mean()}')
        print(f'Precision @ 10 w/ WMRB: {model.precision_at_k(A, preserve_rows=True).mean()}')
        print(f'f1 @ 10 w/ WMRB: {model.f1_at_k(A)}')
        print(f'NDCG @ 10 w/ WMRB: {model.ndcg_at_k(A).mean()}')
    print('top-5 items of user 0:', model.retrieve_user_recs(user=0, k=5))
    return model


if __name__ == '__main__':
    args = sys.argv[1:]
    main(args[0] if args else 'mse', *(int(a) for a in args[1:4]))
