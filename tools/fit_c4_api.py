"""End-to-end check of the PUBLIC API at the C4 shape: MatrixFactorization(...generate_sample=True).fit(...)
then recall_at_k on the sparse interactions.  usage: python tools/fit_c4_api.py [epochs]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from teamoflow.mf.loss_graphs import WMRBLoss
from teamoflow.mf.matrix_factorization import MatrixFactorization
from teamoflow.mf.sparse import SparseInteractions, eye
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
m, n, r, S = 1_000_000, 100_000, 128, 1024
idx, val = bench.gen_interactions(m, n, 100_000_000, 'zipf', 0, dev)
inter = SparseInteractions(idx, val, (m, n))
t0 = time.perf_counter()
model = MatrixFactorization(r, loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S, generate_sample=True)
t1 = time.perf_counter()
model.verbose = False
model.fit(epochs, eye(m), eye(n), inter, lr=0.1)
t2 = time.perf_counter()
rec = float(model.recall_at_k(inter).mean())
t3 = time.perf_counter()
print(f'ctor {t1 - t0:.2f} s | fit({epochs}) {t2 - t1:.2f} s of which epochs {model.fit_seconds_:.3f} s | recall@10 {rec:.5f} in {t3 - t2:.2f} s')
print('loss', model.loss_history_[0], '->', model.loss_history_[-1])
