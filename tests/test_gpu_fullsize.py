"""Full-size (BASELINE C4: 1M users x 100K items, r=128, WMRB S=1024, ~8e7 interactions) parity through
size-independent properties: the oracle cannot run the whole epoch in seconds, but
  * a user's new row, its D[u, :], delta_k and loss depend only on V and the user's own data -> exact
    oracle check on a random sample of users (fp64 closed form, step-interval criterion);
  * an item's gradient is a weighted sum over its entry list -> fp64 re-summation on the CPU for a few
    items, including the heaviest one (~1M entries, ~1000 segments through the slab + combine path);
  * loss sum = sum of per-user partials; two runs are bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close_with_slack, assert_step, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def c4():
    sys.path.insert(0, ROOT)
    import bench
    from teamoflow_amd import _engine, _lib
    from teamoflow_amd.mf.utils import random_sampler_device
    _lib.get()
    dev = torch.device('cuda', 0)
    m, n, r, S, lr = 1_000_000, 100_000, 128, 1024, 0.1
    idx, val = bench.gen_interactions(m, n, 100_000_000, 'zipf', 0, dev)
    # entries ~ N(0, 0.3^2): scores ~ N(0, 1), so about a fifth of the hinge terms are inactive and every bucket of the
    # hinge step (none / some / all positives active for a sample) is in use
    U0 = bench.init_table(m, r, 11, dev) * (0.3 * (m * r) ** 0.5)
    V0 = bench.init_table(n, r, 7, dev) * (0.3 * (n * r) ** 0.5)
    plan = _engine.InteractionPlan(idx, val, m, n)
    R = random_sampler_device(n, m, S, seed=100, device=dev)
    wplan = _engine.WmrbPlan(plan, R, user_chunks=_engine.default_user_chunks(m, _lib.padded_ld(r), n_items=n),
                             item_slices=_engine.default_item_slices(n, _lib.padded_ld(r)), n_components=r)
    assert wplan.user_chunks > 1 and wplan.n_slices > 1
    Dm = None
    st = _engine.TrainState(U0, V0, plan, r, wplan)
    adam = _engine.adam_constants(lr)
    loss = torch.zeros(2, dtype=torch.float64, device=dev)
    _engine.epoch_wmrb(st, adam, n / S, loss[0:1])
    torch.cuda.synchronize()
    return dict(m=m, n=n, r=r, S=S, lr=lr, idx=idx, val=val, U0=U0, V0=V0, plan=plan, R=R, wplan=wplan, st=st,
                D_model=wplan.D_in_model_order(),
                adam=adam, loss=loss, engine=_engine)


def check_user_against_oracle(c4, u, V64=None):
    """D[u, :], delta_k, loss and the new row of one user against the fp64 closed form (oracle.sparse_ref)."""
    from oracle import sparse_ref as S
    st, plan, w = c4['st'], c4['plan'], c4['wplan']
    b, e = int(plan.rowptr_u[u]), int(plan.rowptr_u[u + 1])
    cols = plan.col_u[b:e].to(torch.int64)
    Ru = c4['R'][u].to(torch.int64)
    # only the rows this user touches (positives + negatives), remapped to a compact table
    items, inv = torch.unique(torch.cat([cols, Ru]), return_inverse=True)
    Vc = c4['V0'][items].cpu().numpy().astype(np.float64)
    idx = np.stack([np.zeros(e - b, np.int64), inv[:e - b].cpu().numpy()], axis=1)
    val = plan.val_u[b:e].cpu().numpy().astype(np.float64)
    U64 = c4['U0'][u:u + 1].cpu().numpy().astype(np.float64)
    Rc = inv[e - b:].cpu().numpy()[None]
    t = S.wmrb_terms(U64, Vc, idx, val, Rc, c4['n'], c4['S'])
    sl = S.wmrb_slack(U64, Vc, idx, val, Rc, c4['n'], c4['S'])   # what hinge terms sitting on the kink may move
    assert_close_with_slack(c4['D_model'][u].cpu().numpy(), t['D'][0], sl['D'][0], what=f'D of user {u}')
    pos = val > 0
    assert_close_with_slack(w.delta[b:e].cpu().numpy()[pos], t['delta'], sl['delta'], what=f'delta of user {u}')
    assert float(w.delta[b:e][~torch.as_tensor(pos, device=w.delta.device)].abs().sum()) == 0.0
    assert abs(float(st.loss_part[u]) - t['loss'].sum()) <= 1e-5 * t['loss'].sum(), u
    gU = (t['delta'][:, None] * Vc[idx[pos, 1]]).sum(0) + t['D'][0] @ Vc[Rc[0]]
    assert_step(st.U_nxt[u, :c4['r']].cpu().numpy()[None], U64, gU[None], c4['lr'], what=f'user {u}', slack=sl['gU'])
    return t


def test_sampled_users_match_oracle(c4):
    rng = np.random.default_rng(0)
    plan = c4['plan']
    deg = (plan.rowptr_u[1:] - plan.rowptr_u[:-1]).cpu().numpy()
    users = list(rng.integers(0, c4['m'], 24)) + [int(deg.argmax()), int(deg.argmin())]
    inactive = total = 0
    for u in users:
        t = check_user_against_oracle(c4, int(u))
        inactive += int((t['cnt'] < c4['S']).sum())
        total += len(t['cnt'])
    assert inactive > 0.5 * total   # the sampled users really exercise partly inactive hinges


def independent_item_gradient(j, R_model, D_model, plan, delta, U, r):
    """fp64 gradient of item j from its entry set built WITHOUT the engine's entry lists: the (user, slot) pairs
    with R[u, s] == j straight from the model-order negative table, the positives straight from the CSR arrays.
    Returns (g [r] numpy fp64, number of entries)."""
    us = (R_model == j).nonzero()                                 # [E_neg, 2] (user, slot)
    w_neg = D_model[us[:, 0], us[:, 1]].to(torch.float64)
    g = (w_neg[:, None] * U[us[:, 0], :r].to(torch.float64)).sum(0)
    k = ((plan.col_u == j) & (plan.val_u > 0)).nonzero().flatten()  # CSR positions of the item's positives
    g = g + (delta[k].to(torch.float64)[:, None] * U[plan.user_of[k].to(torch.int64), :r].to(torch.float64)).sum(0)
    return g.cpu().numpy(), int(us.shape[0] + k.numel())


def test_sampled_items_match_fp64_resummation(c4):
    """Item gradients against an entry set that does not come from WmrbPlan (ent_row / ent_w / rowptr_e are not read for
    the sums): a (u, s) pair dropped, duplicated or put into the wrong user block by the plan shows up here.  The plan's
    own list lengths must equal the independent entry counts."""
    st, plan, w = c4['st'], c4['plan'], c4['wplan']
    n, C = c4['n'], w.user_chunks
    rp = w.rowptr_e.cpu().numpy()
    lens = np.diff(rp).reshape(C, n).sum(0)  # list row = user block * n + item
    rng = np.random.default_rng(1)
    items = [int(lens.argmax()), int(lens.argmin())] + list(rng.integers(0, n, 6))
    assert lens.max() > 500_000  # the zipf head really is a ~1000-segment row
    for j in items:
        g, n_ent = independent_item_gradient(j, c4['R'], c4['D_model'], plan, w.delta, st.U, c4['r'])
        assert n_ent == lens[j], (j, n_ent, lens[j])
        assert_step(st.V_nxt[j, :c4['r']].cpu().numpy()[None], c4['V0'][j:j + 1].cpu().numpy(), g[None], c4['lr'],
                    what=f'item {j} ({lens[j]} entries)')
        # the weights used above are the engine's D / delta: pin a few of them to the oracle through their users
        us = (c4['R'] == j).nonzero()
        for u in us[rng.integers(0, len(us), 2), 0].tolist() if len(us) else []:
            check_user_against_oracle(c4, u)


def test_loss_is_sum_of_user_partials_and_runs_are_bit_identical(c4):
    st, eng = c4['st'], c4['engine']
    total = float(c4['loss'][0])
    assert abs(total - float(st.loss_part[:c4['m']].to(torch.float64).sum())) <= 1e-9 * total
    U1, V1, D1 = st.U_nxt.clone(), st.V_nxt.clone(), c4['wplan'].D.clone()
    eng.epoch_wmrb(st, c4['adam'], c4['n'] / c4['S'], c4['loss'][1:2])
    torch.cuda.synchronize()
    assert float(c4['loss'][1]) == total
    assert torch.equal(U1, st.U_nxt) and torch.equal(V1, st.V_nxt) and torch.equal(D1, c4['wplan'].D)
    # rows of users / items without any entry are untouched; pad columns stay zero
    assert float(st.U_nxt[:, c4['r']:].abs().sum()) == 0.0 if st.ld > c4['r'] else True


def test_full_size_mse_sampled_rows(c4):
    """MSE epoch at the same size: sampled user rows and item rows (incl. the heaviest item) against fp64."""
    st, plan, eng = c4['st'], c4['plan'], c4['engine']
    loss = torch.zeros(1, dtype=torch.float64, device=st.U.device)
    eng.epoch_mse(st, c4['adam'], loss)
    torch.cuda.synchronize()
    r = c4['r']
    rng = np.random.default_rng(2)
    rp = plan.rowptr_u.cpu().numpy()
    tot = 0.0
    for u in rng.integers(0, c4['m'], 16):
        b, e = int(rp[u]), int(rp[u + 1])
        Vr = st.V[plan.col_u[b:e].to(torch.int64), :r].to(torch.float64)
        x = st.U[u, :r].to(torch.float64)
        err = plan.val_u[b:e].to(torch.float64) - Vr @ x
        g = ((-2.0 * err)[:, None] * Vr).sum(0).cpu().numpy()
        assert_step(st.U_nxt[u, :r].cpu().numpy()[None], st.U[u, :r].cpu().numpy()[None], g[None], c4['lr'], what=f'mse user {u}')
    rpi = plan.rowptr_i.cpu().numpy()
    lens = np.diff(rpi)
    for j in [int(lens.argmax())] + list(rng.integers(0, c4['n'], 5)):
        b, e = int(rpi[j]), int(rpi[j + 1])
        Ur = st.U[plan.row_i[b:e].to(torch.int64), :r].to(torch.float64)
        y = st.V[j, :r].to(torch.float64)
        err = plan.val_i[b:e].to(torch.float64) - Ur @ y
        g = ((-2.0 * err)[:, None] * Ur).sum(0).cpu().numpy()
        assert_step(st.V_nxt[j, :r].cpu().numpy()[None], st.V[j, :r].cpu().numpy()[None], g[None], c4['lr'], what=f'mse item {j}')
    # loss: fp64 recomputation over ALL interactions with torch (chunked)
    ref = 0.0
    u_ids = plan.user_of
    for b in range(0, plan.nnz, 1 << 24):
        e = min(b + (1 << 24), plan.nnz)
        p = (st.U[u_ids[b:e].to(torch.int64), :r].to(torch.float64) * st.V[plan.col_u[b:e].to(torch.int64), :r].to(torch.float64)).sum(1)
        ref += float(((plan.val_u[b:e].to(torch.float64) - p) ** 2).sum())
    assert abs(float(loss[0]) - ref) <= 1e-5 * ref


def oracle_topk(scores_cpu, k):
    """oracle.dense_ref.tf_top_k (tf.math.top_k: descending, equal values lower index first) of every row of a wide fp32 CPU
    matrix without sorting the million columns: the candidates of a row are the columns >= its (k+1)-th largest value, in
    ascending column order, and tf_top_k ranks those."""
    from oracle import dense_ref as D
    kth = torch.topk(scores_cpu, k + 1, dim=1).values[:, -1]
    vals, idx = [], []
    for i in range(scores_cpu.shape[0]):
        cand = torch.nonzero(scores_cpu[i] >= kth[i]).flatten()          # ascending column order
        v, p = D.tf_top_k(scores_cpu[i, cand], k)
        vals.append(v)
        idx.append(cand[p])
    return torch.stack(vals), torch.stack(idx)


def check_topk_against_cpu_oracle(U, V, vals, idx, k, min_identical_rows=0.97):
    """Fused top-k of the GPU against the oracle's ranking of a CPU fp32 matmul of the same rows.  The CPU BLAS sums the
    r products in another order than the MFMA chain, so scores differ in the last bits: rows whose top k+1 oracle values are
    separated by more than 1e-5 (relative) MUST match index for index; in the others (near-ties) the engine's picks must
    carry the oracle's values, and over all rows at least `min_identical_rows` are index-identical anyway."""
    S = U.float().cpu() @ V.float().cpu().T
    want_v, want_i = oracle_topk(S, k + 1)
    gaps = (want_v[:, :-1] - want_v[:, 1:]) / want_v[:, :1].abs().clamp_min(1e-30)
    clear = (gaps > 1e-5).all(dim=1)
    got_i = idx.cpu().to(torch.int64)
    assert bool(clear.any())
    assert torch.equal(got_i[clear], want_i[clear, :k])
    identical = float((got_i == want_i[:, :k]).all(dim=1).float().mean())
    print(f'[top-k vs CPU oracle] rows identical index for index: {identical:.4f}; rows with clear gaps: {float(clear.float().mean()):.4f}')
    assert identical >= min_identical_rows, (identical, float(clear.float().mean()))
    picked = torch.gather(S, 1, got_i)
    assert rel_err(picked.numpy(), want_v[:, :k].numpy()) < 1e-5
    assert rel_err(vals.cpu().numpy(), picked.numpy()) < 1e-5
    return S


def test_default_ranking_on_c4_trained_tables(c4):
    """retrieve_user_recs(k=10) as the class surface runs it at C4 (nothing selected: the three-plane kernel, 24 bits of every
    factor) on the tables one C4 epoch leaves: index for index oracle.dense_ref.tf_top_k of a CPU fp32 matmul of the same rows
    wherever the oracle's values are separated (check_topk_against_cpu_oracle), 4,096 users spread over the table - and the
    same lists from the fp32 MFMA kernel.  matrix_factorization.py:424-438."""
    from teamoflow_amd import _ops
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    st, r, m = c4['st'], c4['r'], c4['m']
    users = torch.arange(0, m, m // 4096, device=st.U.device)[:4096]
    model = MatrixFactorization(r)
    assert model.predict_arithmetic is None and _ops.PREDICT_ARITHMETIC == 'auto'
    model.user_embedding, model.item_embedding = st.U_nxt[users, :r].contiguous(), st.V_nxt[:c4['n'], :r]
    assert users.numel() * c4['n'] >= _ops.SPLIT_MIN_SCORES and _ops.split_topk_supported(r, 10)
    got = torch.as_tensor(np.asarray(model.retrieve_user_recs(k=10)))
    vals, idx = _ops.predict_topk(model.user_embedding, model.item_embedding, 10, return_values=True)
    assert torch.equal(idx.cpu().long(), got.long())
    check_topk_against_cpu_oracle(model.user_embedding, model.item_embedding, vals, idx, 10, min_identical_rows=0.9)
    v32, i32 = _ops.predict_topk(model.user_embedding, model.item_embedding, 10, return_values=True, arithmetic='fp32')
    same = (i32 == idx).all(1)
    assert float(same.float().mean()) > 0.999
    assert float((v32 - vals).abs().max()) <= 2e-6 * float(v32.abs().max())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_full_catalog_fused_topk_matches_the_cpu_oracle(dtype):
    """predict at catalog scale (1,000,003 items - a ragged last tile, 512 MB of V, byte offsets up to 2^29 in the
    kernel's buffer loads): the fused top-k of 300 users against oracle.dense_ref.tf_top_k on a CPU fp32 matmul of the same
    rows (not against another kernel of this library), and the first / last items win when they should (planted maxima at
    both ends of the catalog, an exact tie in the ragged tile)."""
    from teamoflow_amd import _ops
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(5)
    n, m, r = 1_000_003, 300, (128 if dtype == 'f32' else 256)
    V = torch.randn(n, r, device=dev, generator=g) * 0.1
    U = torch.randn(m, r, device=dev, generator=g) * 0.1
    V[0] = U[0] * 50                 # user 0's best item is the very first ...
    V[n - 1] = U[1] * 50             # ... user 1's the very last (inside the ragged tile)
    V[n - 2] = V[n - 1]              # and an exact tie right before it: the lower index must come first
    if dtype == 'bf16':
        U, V = U.to(torch.bfloat16), V.to(torch.bfloat16)
    vals, idx = _ops.predict_topk(U, V, 10, return_values=True)
    check_topk_against_cpu_oracle(U, V, vals, idx, 10)
    assert int(idx[0, 0]) == 0 and idx[1, :2].tolist() == [n - 2, n - 1]
    if dtype == 'f32':   # and the library's own two-kernel path agrees bit for bit (same fmaf chain)
        assert torch.equal(idx, _ops.topk_stable(_ops.predict_gemm(U, V), 10))
