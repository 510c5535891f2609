"""End-to-end recall@10 on C1 (100 x 50, r=5, MSE, 450 epochs): engine-trained vs oracle-trained tables from the SAME start,
for the golden fixture's start and for a few other seeds.  With 225 interactions and near-sign Adam steps the two
trajectories drift apart (loss agrees to ~1e-3 after 450 epochs); this prints how far recall@10 moves.
(a checker script, not collected by pytest - it lives under tests/ because only tests may use oracle/)
usage: python tests/check_c1_recall_e2e.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import datagen as G, dense_ref as D, sparse_ref as S
from teamoflow.mf.initializer_graphs import FixedInitializer
from teamoflow.mf.matrix_factorization import MatrixFactorization
from teamoflow.mf.sparse import SparseInteractions, eye

def gpu_fit(U0, V0, idx, val, shape, epochs, lr):
    model = MatrixFactorization(U0.shape[1], user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
    model.verbose = False
    model.fit(epochs, eye(shape[0]), eye(shape[1]), SparseInteractions(idx, val, shape), lr=lr)
    return model

g = dict(np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'c1_mse.npz')))
A, idx, val = g['A'], g['indices'], g['values']
for name, U0, V0 in [('fixture', g['U0'], g['V0'])] + [(f'seed {s}', G.normal_init(100, 5, s), G.normal_init(50, 5, s + 100)) for s in (1, 2, 3, 4)]:
    model = gpu_fit(U0, V0, idx, val, A.shape, 450, 1e-2)
    ref = D.fit_dense(U0, V0, idx, val, 'mse', 450, 1e-2)
    s64 = S.fit_sparse(U0, V0, idx, val, 'mse', 450, 1e-2, dtype=np.float64)
    r_gpu = float(model.recall_at_k(torch.tensor(A)).mean())
    r_cpu = float(D.recall_at_k_dense(ref['U'], ref['V'], A, 10).mean())
    r_64 = float(D.recall_at_k_dense(s64['U'].astype(np.float32), s64['V'].astype(np.float32), A, 10).mean())
    print(f'{name:8s} recall@10: engine {r_gpu:.6f} | dense fp32 oracle {r_cpu:.6f} | closed form fp64 {r_64:.6f} | '
          f'final loss {model.loss_history_[-1]:.6e} / {ref["loss"][-1]:.6e} / {s64["loss"][-1]:.6e}')
