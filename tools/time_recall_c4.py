"""recall_at_k through the class surface at the C4 shape (1M x 100K, r = 128, 1e8 interactions): the fused top-k and the hit count."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
from teamoflow_amd.mf.sparse import SparseInteractions
dev = 'cuda'
m, n, r, nnz = 1_000_000, 100_000, 128, 100_000_000
g = torch.Generator(device=dev).manual_seed(1)
keys = torch.unique(torch.randint(0, m * n, (int(nnz * 1.001),), device=dev, generator=g))[:nnz]
idx = torch.stack([keys // n, keys % n], 1)
val = torch.randint(1, 6, (keys.numel(),), device=dev, generator=g).float()
inter = SparseInteractions(idx, val, (m, n))
model = MatrixFactorization(r)
model.user_embedding = torch.randn(m, r, device=dev, generator=g) * 0.05
model.item_embedding = torch.randn(n, r, device=dev, generator=g) * 0.05
for what, fn in (('retrieve_user_recs(k=10) [top-k only, device tensor]', lambda: model._top_items(10, clamp=True)),
                 ('recall_at_k(sparse interactions, k=10)', lambda: model.recall_at_k(inter, 10))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{what}: {dt * 1e3:.1f} ms', flush=True)
