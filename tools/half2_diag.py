import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teamoflow_amd import _ops
from oracle import sparse_ref as S
rng = np.random.default_rng(1)
m, n = 300, 33001
for r in (128, 129, 256):
    U = rng.integers(-2, 3, (m, r)).astype(np.float32); V = rng.integers(-2, 3, (n, r)).astype(np.float32)
    sc = U @ V.T
    for k, clamp in ((10, False), (10, True), (22, False), (23, False), (30, False), (30, True), (32, False)):
        for waves in ('',):
            ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
            vals, got = _ops.predict_topk(torch.tensor(U), torch.tensor(V), k, clamp_negatives=clamp, return_values=True, arithmetic='half2')
            got = got.cpu().numpy(); bad = (got != ref).any(1)
            msg = ''
            if bad.any():
                i = int(np.argmax(bad)); j = int(np.argmax(got[i] != ref[i]))
                msg = f' first bad row {i} pos {j}: got {got[i, j]} (score {sc[i, got[i, j]] if 0 <= got[i, j] < n else None}) want {ref[i, j]} (score {sc[i, ref[i, j]]}); vals {vals[i, max(0,j-1):j+2].tolist()}'
            print(f'r={r} k={k} clamp={clamp} waves={waves or "auto"}: bad rows {int(bad.sum())}/{m}{msg}', flush=True)
