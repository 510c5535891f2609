"""C4 (1M x 100K, r=128, WMRB S=1024) for E epochs through the public API, resident and with the catalog in 4 windows
(model.shard_items = 4, one rank): loss trajectories side by side.  usage: python tools/windowed_vs_resident_c4.py [epochs]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from teamoflow_amd.mf.initializer_graphs import FixedInitializer  # noqa: E402
from teamoflow_amd.mf.loss_graphs import WMRBLoss  # noqa: E402
from teamoflow_amd.mf.matrix_factorization import MatrixFactorization  # noqa: E402
from teamoflow_amd.mf.sparse import SparseInteractions, eye  # noqa: E402
from teamoflow_amd.mf.utils import random_sampler_device  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda', 0)
m, n, r, S = 1_000_000, 100_000, 128, 1024
idx, val = bench.gen_interactions(m, n, 100_000_000, 'zipf', 0, dev)
U0, V0 = bench.init_table(m, r, 11, dev), bench.init_table(n, r, 7, dev)
R = random_sampler_device(n, m, S, seed=100, device=dev)
inter = SparseInteractions(idx, val, (m, n))
out = {}
for name, shard in (('resident', 0), ('windowed_4', 4)):
    model = MatrixFactorization(r, loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S,
                                user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
    model.verbose, model.shard_items, model.random_ind = False, shard, R
    model.fit(epochs, eye(m), eye(n), inter, lr=0.1)
    out[name] = dict(loss=model.loss_history_, ms_per_epoch=model.fit_seconds_ / epochs * 1e3,
                     recall_at_10=float(model.recall_at_k(inter).mean()))
    del model
    torch.cuda.empty_cache()
a, b = out['resident']['loss'], out['windowed_4']['loss']
rel = [abs(x - y) / abs(x) for x, y in zip(a, b)]
print(json.dumps(dict(epochs=epochs, ms_per_epoch={k: v['ms_per_epoch'] for k, v in out.items()},
                      recall_at_10={k: v['recall_at_10'] for k, v in out.items()},
                      loss_first={k: v['loss'][0] for k, v in out.items()}, loss_last={k: v['loss'][-1] for k, v in out.items()},
                      max_rel_loss_diff=max(rel), rel_loss_diff_at=[rel[i] for i in (0, 1, 9, min(49, epochs - 1), epochs - 1)])))
