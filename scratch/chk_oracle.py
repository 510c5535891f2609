import numpy as np, torch, time
from oracle import dense_ref as D, sparse_ref as S, datagen as G
np.random.seed(0)
idx, val, shape, A = G.generate_random_interaction(100, 50, density=0.05)
print('nnz', len(val), sorted(set(val.tolist())))
U0 = G.normal_init(100, 5, 1); V0 = G.normal_init(50, 5, 2)
d = D.fit_dense(U0, V0, idx, val, 'mse', 30, 1e-2, record_epochs=(1,30))
s = S.fit_sparse(U0, V0, idx, val, 'mse', 30, 1e-2, record_epochs=(1,30))
s64 = S.fit_sparse(U0, V0, idx, val, 'mse', 30, 1e-2, dtype=np.float64, record_epochs=(1,30))
print('mse loss rel', np.abs(d['loss']-s['loss']).max()/d['loss'].max(), np.abs(d['loss']-s64['loss']).max()/d['loss'].max())
for e in (1,30):
    print(e, 'U', np.abs(d['snapshots'][e][0]-s['snapshots'][e][0]).max(), np.abs(d['snapshots'][e][0]-s64['snapshots'][e][0]).max(), np.abs(d['snapshots'][e][0]).max())
# wmrb
np.random.seed(1)
idx, val, shape, A = G.generate_random_interaction(50, 100, density=0.05)
R = G.random_sampler(100, 50, 50)
U0 = G.uniform_init(50, 3, 1); V0 = G.uniform_init(100, 3, 2)
d = D.fit_dense(U0, V0, idx, val, 'wmrb', 25, 0.1, random_ind=R, n_items=100, n_samples=50, record_epochs=(1,25))
s = S.fit_sparse(U0, V0, idx, val, 'wmrb', 25, 0.1, random_ind=R, n_items=100, n_samples=50, record_epochs=(1,25))
s64 = S.fit_sparse(U0, V0, idx, val, 'wmrb', 25, 0.1, random_ind=R, n_items=100, n_samples=50, dtype=np.float64, record_epochs=(1,25))
print('wmrb loss', d['loss'][:3], d['loss'][-1])
print('wmrb loss rel', np.abs(d['loss']-s['loss']).max()/d['loss'].max(), np.abs(d['loss']-s64['loss']).max()/d['loss'].max())
for e in (1,25):
    print(e, 'U', np.abs(d['snapshots'][e][0]-s['snapshots'][e][0]).max(), np.abs(d['snapshots'][e][0]-s64['snapshots'][e][0]).max(), np.abs(d['snapshots'][e][0]).max())
    print(e, 'V', np.abs(d['snapshots'][e][1]-s['snapshots'][e][1]).max(), np.abs(d['snapshots'][e][1]-s64['snapshots'][e][1]).max())
# gather known answer
inp = torch.tensor([[1,4,2],[5,7,8],[6,2,1]], dtype=torch.float32); ind = torch.tensor([[0,2,0],[2,2,2],[2,1,0]])
print(D.gather_matrix_indices(inp, ind))
print(D.tf_top_k(torch.tensor([0.,1,1,0,1]),3))
r = D.recall_at_k_dense(d['U'], d['V'], A); print('recall', r.mean(), len(r))
r2 = S.recall_at_k_sparse(d['U'], d['V'], idx, val); print('recall sparse', r2.mean())
