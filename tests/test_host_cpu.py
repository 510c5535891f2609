"""CPU tests: the C-ABI library loads and exports every symbol include/tmf.h declares, the host-side
index builders agree with a NumPy restatement, the engine refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from teamoflow_amd import _lib
    import __graft_entry__ as ge
    ge.build()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, 'include', 'tmf.h')).read()
    declared = set(re.findall(r'\b(tmf_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations found'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tmf_version() == 203 == _lib.MIN_LIB_VERSION


def test_padded_ld_and_adam_constants_match_host_mirror():
    from teamoflow_amd import _lib
    from oracle.dense_ref import adam_fresh_constants
    lib = _lib.load_library()
    for r in list(range(1, 70)) + [100, 128, 129, 255, 256, 257, 512, 513, 768, 1024]:
        assert lib.tmf_padded_ld(r) == _lib.padded_ld(r) >= r
        assert _lib.padded_ld(r) % 4 == 0
    assert lib.tmf_padded_ld(0) == 0 and lib.tmf_padded_ld(1025) == 0
    for r in list(range(1, 70)) + [100, 128, 129, 255, 256, 257, 512, 513, 768, 1024]:
        assert lib.tmf_padded_ld_bf16(r) == _lib.padded_ld(r, torch.bfloat16) >= r
        assert _lib.padded_ld(r, torch.bfloat16) % 8 == 0
    for lr in (1e-2, 1e-3, 0.1, 0.05):
        a = lib.tmf_adam_fresh(lr)
        alpha, omb1, omb2, eps = adam_fresh_constants(lr)
        assert (np.float32(a.alpha), np.float32(a.one_minus_b1), np.float32(a.one_minus_b2), np.float32(a.eps)) == \
            (alpha, omb1, omb2, eps)


def test_engine_refuses_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from teamoflow_amd import _lib
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    with pytest.raises(_lib.EngineUnavailable):
        _lib.get()
    model = MatrixFactorization(3)
    with pytest.raises(_lib.EngineUnavailable):
        model.fit(1, eye(4), eye(5), SparseInteractions(np.array([[0, 1]]), np.array([1.0]), (4, 5)))


def test_segment_table_and_plans_against_numpy():
    from teamoflow_amd._engine import InteractionPlan, SegmentTable, WmrbPlan
    rng = np.random.default_rng(0)
    m, n, S, chunk = 13, 9, 4, 5
    idx = np.stack([rng.integers(0, m, 120), rng.integers(0, n, 120)], axis=1)
    val = rng.integers(-1, 4, 120).astype(np.float32)
    plan = InteractionPlan(torch.tensor(idx), torch.tensor(val), m, n, chunk=chunk)
    order = np.argsort(idx[:, 0] * n + idx[:, 1], kind='stable')   # row-major, duplicates in input order
    u, j, v = idx[order, 0], idx[order, 1], val[order]
    assert np.array_equal(plan.col_u.numpy(), j) and np.array_equal(plan.val_u.numpy(), v)
    assert np.array_equal(plan.rowptr_u.numpy(), np.concatenate([[0], np.cumsum(np.bincount(u, minlength=m))]))
    oc = np.argsort(j, kind='stable')
    assert np.array_equal(plan.row_i.numpy(), u[oc]) and np.array_equal(plan.val_i.numpy(), v[oc])
    # user-blocked CSC (MSE item pass): list row = block * n + item, same multiset of entries per item
    pb = InteractionPlan(torch.tensor(idx), torch.tensor(val), m, n, chunk=chunk, user_chunks=3)
    upc = -(-m // 3)
    rpb = pb.rowptr_i.numpy()
    assert len(rpb) == 3 * n + 1 and pb.seg_i.row_mod == n and (pb.seg_i.seg_slab.numpy() >= 0).all()
    for item in range(n):
        got = []
        for blk in range(3):
            rows = pb.row_i.numpy()[rpb[blk * n + item]:rpb[blk * n + item + 1]]
            assert ((rows // upc) == blk).all()
            got += list(rows)
        assert sorted(got) == sorted(u[j == item])
    # segments: every entry covered exactly once, slab slots consecutive per long row
    seg = plan.seg_u
    covered = np.zeros(len(v), int)
    rp = seg.rowptr.numpy()
    for s in range(seg.nseg):
        row, ch, slot = int(seg.seg_row[s]), int(seg.seg_chunk[s]), int(seg.seg_slab[s])
        beg = rp[row] + ch * chunk
        end = min(beg + chunk, rp[row + 1])
        covered[beg:end] += 1
        nch = max(1, -(-(rp[row + 1] - rp[row]) // chunk))
        assert (slot == -1) == (nch == 1)
    assert (covered == 1).all()
    lb = seg.long_slab_beg.numpy()
    for i, row in enumerate(seg.long_rows.numpy()):
        slots = seg.seg_slab.numpy()[seg.seg_row.numpy() == row]
        assert list(slots) == list(range(lb[i], lb[i + 1]))
    assert seg.n_slab == lb[-1]
    # WMRB entry lists: per item, positives (ascending user) then (user, slot) pairs
    R = np.stack([rng.choice(n, S, replace=False) for _ in range(m)]).astype(np.int32)
    w = WmrbPlan(plan, torch.tensor(R), chunk=chunk)
    assert not w.sliced
    rpe = w.rowptr_e.numpy()
    assert len(rpe) == n + 1 and rpe[-1] == (v > 0).sum() + m * S    # stored values <= 0 are in no list
    for item in range(n):
        rows = w.ent_row.numpy()[rpe[item]:rpe[item + 1]]
        ws = w.ent_w.numpy()[rpe[item]:rpe[item + 1]]
        pos_k = np.nonzero((j == item) & (v > 0))[0]
        us, ss = np.nonzero(R == item)
        assert list(ws) == list(pos_k) + list(len(v) + us * S + ss)
        assert list(rows) == list(u[pos_k]) + list(us)
    assert w.delta.numel() == len(v) and tuple(w.D.shape) == (m, S)
    # the sliced pass keeps every user's negatives sorted by item: same lists, entry ids of the sorted slots
    ws_ = WmrbPlan(plan, torch.tensor(R), chunk=chunk, item_slices=2)
    assert ws_.sliced and np.array_equal(ws_.rowptr_e.numpy(), rpe)
    Rs = np.sort(R, axis=1)
    for item in range(n):
        pos_k = np.nonzero((j == item) & (v > 0))[0]
        us, ss = np.nonzero(Rs == item)
        assert list(ws_.ent_w.numpy()[rpe[item]:rpe[item + 1]]) == list(pos_k) + list(len(v) + us * S + ss)
        assert list(ws_.ent_row.numpy()[rpe[item]:rpe[item + 1]]) == list(u[pos_k]) + list(us)
    # user-chunked lists: list row = block * n + item; per item the union over blocks is the same entry set,
    # every segment owns a slab slot and the slots of an item are consecutive
    C = 3
    wc = WmrbPlan(plan, torch.tensor(R), chunk=chunk, user_chunks=C)
    upc = -(-m // C)
    rpc = wc.rowptr_e.numpy()
    assert len(rpc) == C * n + 1
    for item in range(n):
        got = []
        for blk in range(C):
            b, e = rpc[blk * n + item], rpc[blk * n + item + 1]
            rows = wc.ent_row.numpy()[b:e]
            assert ((rows // upc) == blk).all()
            got += list(wc.ent_w.numpy()[b:e])
        want = w.ent_w.numpy()[rpe[item]:rpe[item + 1]]
        assert sorted(got) == sorted(want)
    sg = wc.seg_e
    assert sg.n_long == n and sg.n_slab == sg.nseg and (sg.seg_slab.numpy() >= 0).all()
    assert sorted(sg.seg_slab.numpy()) == list(range(sg.nseg))
    lb = sg.long_slab_beg.numpy()
    item_of_seg = sg.seg_row.numpy() % n
    for item in range(n):
        assert sorted(sg.seg_slab.numpy()[item_of_seg == item]) == list(range(lb[item], lb[item + 1]))


def test_reference_style_imports_and_surface():
    import teamoflow
    from teamoflow.mf import matrix_factorization, loss_graphs, embedding_graphs, initializer_graphs, predict_graphs, utils  # noqa
    from teamoflow.mf.matrix_factorization import MatrixFactorization
    import inspect
    sig = inspect.signature(MatrixFactorization.__init__)
    assert list(sig.parameters)[1:] == ['n_components', 'user_repr_graph', 'item_repr_graph', 'loss_graph',
                                        'user_weight_graph', 'item_weight_graph', 'n_users', 'n_items', 'n_samples',
                                        'generate_sample']
    assert list(inspect.signature(MatrixFactorization.fit).parameters)[1:] == ['epochs', 'user_features',
                                                                                 'item_features', 'tf_interactions', 'lr']
    for name in ('predict', 'predict_ranks', 'recall_at_k', 'precision_at_k', 'f1_at_k', 'dcg_at_k', 'idcg_at_k',
                 'ndcg_at_k', 'retrieve_user_recs', 'save_model', 'from_saved'):
        assert hasattr(MatrixFactorization, name)
    mf = MatrixFactorization(4, n_users=3, n_items=10)
    assert mf.n_samples == 5 and mf.random_ind is None
    cfg, res = None, None
    m2 = MatrixFactorization.from_saved({'n_components': 7})
    assert m2.n_components == 7


def test_import_root_of_the_reference_tests():
    """The import lines of the reference's own tests (/root/reference/test/test_loss.py:5-7, test_utils.py:5,
    test_predict.py:5-8, test_embedding.py, test_initializer.py) resolve to this package, star-imports included."""
    ns = {}
    for line in ('from src.teamoflow.mf.loss_graphs import *',
                 'from src.teamoflow.mf.utils import *',
                 'from src.teamoflow.mf.predict_graphs import *',
                 'from src.teamoflow.mf.embedding_graphs import *',
                 'from src.teamoflow.mf.initializer_graphs import *',
                 'from src.teamoflow.mf.utils import generate_random_interaction',
                 'from src.teamoflow.mf.matrix_factorization import MatrixFactorization',
                 'import src.teamoflow.mf.input_utils'):
        exec(line, ns)
    import teamoflow_amd.mf.matrix_factorization as impl
    assert ns['MatrixFactorization'] is impl.MatrixFactorization
    for name in ('MSELoss', 'WMRBLoss', 'KLDivergenceLoss', 'LinearEmbedding', 'BiasedLinearEmbedding', 'ReLUEmbedding',
                 'NormalInitializer', 'UniformInitializer', 'random_sampler', 'gather_matrix_indices', 'generate_random_interaction'):
        assert name in ns, name


def test_generic_path_matches_oracle_on_cpu(golden):
    """Non-fast-path plug-ins (here: a dense non-identity feature matrix) train through the generic
    autograd loop; with identity features given densely AND a subclassed loss it must equal the oracle."""
    if torch.cuda.is_available():
        pytest.skip('covered on the GPU box by the fast path')
    from oracle import dense_ref as D
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import MSELoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions

    class MyMSE(MSELoss):
        pass
    g = golden('c1_mse')
    m, n = g['A'].shape
    model = MatrixFactorization(5, loss_graph=MyMSE(), user_weight_graph=FixedInitializer(g['U0']),
                                item_weight_graph=FixedInitializer(g['V0']))
    model.verbose = False
    model.fit(5, torch.eye(m), torch.eye(n), SparseInteractions(g['indices'], g['values'], (m, n)), lr=float(g['lr']))
    assert np.abs(np.array(model.loss_history_) - g['loss'][:5]).max() / g['loss'][0] < 1e-6


def test_item_slices_and_defaults_on_cpu(monkeypatch):
    from teamoflow_amd import _engine
    from teamoflow_amd._engine import InteractionPlan, WmrbPlan
    rng = np.random.default_rng(4)
    m, n, S, NS = 11, 50, 7, 4
    idx = np.stack([rng.integers(0, m, 40), rng.integers(0, n, 40)], axis=1)
    val = rng.integers(1, 4, 40).astype(np.float32)
    plan = InteractionPlan(torch.tensor(idx), torch.tensor(val), m, n)
    R = np.stack([rng.choice(n, S, replace=False) for _ in range(m)]).astype(np.int32)
    w = WmrbPlan(plan, torch.tensor(R), user_chunks=2, item_slices=NS)
    Rs = w.R.numpy()
    assert np.array_equal(Rs, np.sort(R, axis=1))                      # negatives kept sorted by item
    width = -(-n // NS)
    off, poff = w.slice_off.numpy(), w.pos_off.numpy()
    assert off.shape == (m, NS + 1) and (off[:, 0] == 0).all() and (off[:, -1] == S).all()
    rp, col = plan.rowptr_u.numpy(), plan.col_u.numpy()
    assert poff.shape == (m, NS + 1) and (poff[:, 0] == 0).all() and np.array_equal(poff[:, -1], np.diff(rp))
    for u in range(m):
        assert (np.diff(col[rp[u]:rp[u + 1]]) >= 0).all()              # interactions of a user: ascending item
        for sl in range(NS):
            seg = Rs[u, off[u, sl]:off[u, sl + 1]]
            assert ((seg >= sl * width) & (seg < (sl + 1) * width)).all()
            seg = col[rp[u] + poff[u, sl]:rp[u] + poff[u, sl + 1]]
            assert ((seg >= sl * width) & (seg < (sl + 1) * width)).all()
    # D written in sorted order maps back to the model's order
    w.D.copy_(torch.tensor(Rs.astype(np.float32)))                      # pretend D[u, s] = item id of the sorted slot
    assert np.array_equal(w.D_in_model_order().numpy(), R.astype(np.float32))
    # defaults: small catalogs keep the fused pass / a single user block; C4 gets 13 slices and 163 blocks
    monkeypatch.setenv('TMF_SLAB_BUDGET', str(8 << 30))
    assert _engine.default_item_slices(1682, 32) == 1 and _engine.default_user_chunks(943, 32, n_items=1682) == 1
    assert _engine.default_item_slices(100_000, 128) == 13
    assert _engine.default_user_chunks(1_000_000, 128, n_items=100_000) == 163   # 3 MB blocks
    assert _engine.default_user_chunks(1_250_000, 256, n_items=1_000_000) == 8   # an 8 GB slab budget bounds it
    monkeypatch.setenv('TMF_SLAB_BUDGET', str(64 << 30))
    assert _engine.default_user_chunks(1_250_000, 256, n_items=1_000_000) == 67


def test_half2_range_guard_on_cpu():
    """The 'auto' rule of the fused top-k takes the two-plane fp16 kernel only while the item rows span <= 2^12 in magnitude
    (one power-of-two scale for the whole item table); pure tensor logic, no GPU needed."""
    import torch
    from teamoflow_amd import _ops
    g = torch.Generator().manual_seed(0)
    V = torch.randn(500, 8, generator=g)
    assert _ops.half2_range_ok(V) and _ops.half2_range_ok(torch.zeros(3, 4))
    V[7] *= 2.0 ** 14
    assert not _ops.half2_range_ok(V)
    V[7] = 0                                   # all-zero rows do not count
    assert _ops.half2_range_ok(V)
    V[9, 0] = float('inf')
    assert not _ops.half2_range_ok(V)
    assert _ops.split_topk_supported(256, 32) and not _ops.split_topk_supported(257, 1) and _ops.split_topk_supported(64, 40) and not _ops.split_topk_supported(64, 41)
    assert _ops.half2_topk_supported(256, 32) and not _ops.half2_topk_supported(257, 1) and not _ops.half2_topk_supported(8, 33)


def test_virtual_rows_of_the_balanced_item_pass():
    """_engine.VirtualRows (work units of tmf_wsum_rows5): every output row is cut into ceil(entries / target) parts, the virtual
    rows are listed in (row, part) order with a sentinel behind them, whole rows have slot -1, the parts of a cut row consecutive
    slab slots in part order - what tmf_combine_rows sums."""
    import torch
    from teamoflow_amd import _engine as E
    C, n = 3, 7
    lens = torch.tensor([[0, 5, 1, 40, 2, 0, 9], [1, 4, 0, 35, 2, 1, 8], [0, 6, 2, 30, 1, 0, 7]])
    rowptr = E._excl_cumsum(lens.reshape(-1))
    v = E.VirtualRows(rowptr, C, n, target=10)
    tot = lens.sum(0)
    parts = torch.clamp((tot + 9) // 10, min=1)
    assert v.n_vrows == int(parts.sum()) and v.item.numel() == v.n_vrows + 1 and v.slot.numel() == v.n_vrows
    assert v.item[-1] == n and v.part[-1] == 0 and v.nparts[-1] == 1                     # the sentinel: the end of a block's lists
    for j in range(n):
        rows = (v.item[:-1] == j).nonzero().flatten()
        assert rows.numel() == int(parts[j]) and v.part[rows].tolist() == list(range(int(parts[j])))
        assert (v.nparts[rows] == int(parts[j])).all() and (rows[1:] - rows[:-1] == 1).all()   # consecutive virtual rows
        if parts[j] == 1:
            assert v.slot[rows].tolist() == [-1]
    cut = (parts > 1).nonzero().flatten()
    assert v.long_rows.tolist() == cut.tolist() and v.n_long == cut.numel() and v.n_slab == int(parts[cut].sum())
    for i, j in enumerate(cut.tolist()):
        rows = (v.item[:-1] == j).nonzero().flatten()
        b, e = int(v.long_slab_beg[i]), int(v.long_slab_beg[i + 1])
        assert v.slot[rows].tolist() == list(range(b, e))
    # the default target: just above the bulk (1.15 x the median, at least the mean) - ordinary rows stay whole
    d = E.VirtualRows(rowptr, C, n)
    assert d.target == int(max(1.15 * float(tot.double().median()), float(tot.double().mean()))) + 1
    assert d.max_parts == int(-(-int(tot.max()) // d.target))
    # in block t part p of P takes [b + p L / P, b + (p + 1) L / P): the parts tile the list exactly (the kernel's arithmetic)
    L = 37
    for P in (1, 2, 5, 11, 40):
        cuts = [p * L // P for p in range(P + 1)]
        assert cuts[0] == 0 and cuts[-1] == L and all(a <= b for a, b in zip(cuts, cuts[1:]))


def test_scores6_entry_streams_on_the_host():
    """_engine.Scores6Plan (the streams tmf_wmrb_scores6 walks), built on CPU tensors: every score the epoch needs - negative (u, pos)
    of the item-sorted table, interaction k of the CSR - appears exactly once, in the chunk of its (item slice, user group), with
    the right packed id and place; inside a chunk the interactions come first, then the negatives, each by user and item; chunks
    are padded to steps of 8 with the PAD place."""
    import torch
    from teamoflow_amd import _engine as E, _lib
    g = torch.Generator().manual_seed(5)
    m, n, r, S = 150, 900, 128, 24
    key = torch.unique(torch.randint(0, m, (2500,), generator=g) * n + torch.randint(0, n, (2500,), generator=g))
    idx = torch.stack([key // n, key % n], 1)
    val = torch.randint(-1, 6, (key.numel(),), generator=g).float()
    R = torch.stack([torch.randperm(n, generator=g)[:S] for _ in range(m)]).to(torch.int32)
    plan = E.InteractionPlan(idx, val, m, n)
    wplan = E.WmrbPlan(plan, R, item_slices=3, n_components=r, sliced=True)
    s6 = E.Scores6Plan(plan, wplan, r, torch.float32, slice_bytes=100 * 512)      # slices of 100 items: 9 of them
    UG = _lib.load_library().tmf_wmrb_scores6_users_per_group()
    ng, ns = s6.n_groups, s6.n_slices
    assert ng == -(-m // UG) and ns == 9 and s6.chunk_ptr.numel() == ns * ng + 1
    width = -(-n // ns)
    ptr, ids, outs = s6.chunk_ptr.tolist(), s6.ids.tolist(), s6.outs.tolist()
    assert all(p % 8 == 0 for p in ptr) and ptr[-1] == s6.n_padded and len(ids) == s6.n_padded + 8
    Rs, user_of, col = wplan.R.tolist(), plan.user_of.tolist(), plan.col_u.tolist()
    seen_neg, seen_pos = set(), set()
    for c in range(ns * ng):
        sl, grp = divmod(c, ng)
        real = [(ids[e], outs[e]) for e in range(ptr[c], ptr[c + 1]) if outs[e] != E.Scores6Plan.PAD]
        pads = [e for e in range(ptr[c], ptr[c + 1]) if outs[e] == E.Scores6Plan.PAD]
        assert len(pads) < 8 and all(e >= ptr[c] + len(real) for e in pads)        # padding sits behind the chunk's entries
        order = []
        for pid, out in real:
            u, item = grp * UG + ((pid >> 24) & 0xff), pid & 0xffffff
            assert item // width == sl and u < m
            if out >= 0:
                assert Rs[out // S][out % S] == item and out // S == u and out not in seen_neg
                seen_neg.add(out)
                order.append((1, u, item))
            else:
                k = ~out
                assert user_of[k] == u and col[k] == item and k not in seen_pos
                seen_pos.add(k)
                order.append((0, u, item))
        assert order == sorted(order)                                              # interactions, then negatives; each by user and item
        for e in pads:                                                             # a padding entry repeats a valid id of the chunk
            assert (ids[e] & 0xffffff) // width == sl
    assert len(seen_neg) == m * S and len(seen_pos) == plan.nnz
