"""BASELINE configs at their stated sizes and the reference semantics the fixtures do not force:

  * C3 (6040 x 3706, MovieLens-1M shape, r=64, WMRB, S = n // 2 = 1853 as the class default and S = n // 5 = 741 as
    /root/reference/examples/benchmarking_ML.py:65 uses) against the C/OpenMP closed-form oracle: one step (loss, D,
    delta, both tables in the step interval), a 20-epoch trajectory, teacher-forced steps from the oracle's later
    state, recall@10.
  * WMRB hinge exactly at zero: tf.maximum routes the gradient to its first argument when x >= y
    (/root/reference/src/teamoflow/mf/loss_graphs.py:83-84), so a term with 1 - p_k + sp[u, s] == 0 counts as active.
"""
import numpy as np
import pytest
import torch

from conftest import assert_close_with_slack, assert_step, rel_err, report_slack
from test_gpu_parity import check_one_step, fit_model, tm  # noqa: F401  (tm is a fixture)

pytestmark = pytest.mark.gpu


def c3_inputs(S_):
    from oracle import datagen as G
    m, n, r = 6040, 3706, 64
    np.random.seed(0)
    idx, val, shape, A = G.generate_random_interaction(m, n, density=1000209 / (0.9 * m * n))
    R = np.stack([np.random.choice(n, S_, replace=False) for _ in range(m)]).astype(np.int32)  # utils.py:20
    return idx, val, A, R, G.uniform_init(m, r, 1), G.uniform_init(n, r, 2)


@pytest.mark.parametrize('S_', [1853, 741])
def test_c3_full_size(tm, S_):  # noqa: F811
    from oracle import dense_ref as D
    from oracle import sparse_c as C
    idx, val, A, R, U0, V0 = c3_inputs(S_)
    m, n = A.shape
    lr, epochs = 0.1, 20
    assert len(val) > 990_000
    plan = C.Plan(idx, val, m, n, R)
    Uc, Vc, losses, first = U0, V0, [], None
    for e in range(epochs):
        Uc, Vc, mean, t = C.wmrb_epoch(Uc, Vc, plan, n, S_, lr, want_grads=(e == 0))
        losses.append(mean)
        if e == 0:
            first = t
    # one step from the initial tables: loss, D, delta, both tables.  "Equal" = within 1e-5 plus what the hinge terms
    # sitting on the kink may move (C.wmrb_boundary_slack; none at the start, thousands of (k, s) pairs after training)
    def check_step(model, Ub, Vb, mean, t, tag):
        sl = C.wmrb_boundary_slack(Ub, Vb, plan, n, S_)
        assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
        sD = assert_close_with_slack(model._state.wplan.D_in_model_order().cpu().numpy(), t['D'], sl['D'], what=f'D {tag}')
        # a switch moves cnt_k by 1, i.e. delta_k by w_k
        sd = assert_close_with_slack(model._state.wplan.delta.cpu().numpy(), t['delta'], sl['delta'], what=f'delta {tag}')
        # the slack is computed by the oracle that is being trusted: report how many terms sat on the kink and what the GPU
        # result actually consumed of it, and bound both - every switched term moves ONE element of D and ONE of delta, so no
        # more elements than boundary pairs may need slack at all
        report_slack(test=f'c3_full_size S={S_} {tag}', hinge_terms=int(len(val)) * S_, boundary_pairs=int(sl['pairs']),
                     D_elements_needing_slack=sD['n_over'], D_max_slack_consumed=sD['max_consumed'],
                     delta_elements_needing_slack=sd['n_over'], delta_max_slack_consumed=sd['max_consumed'])
        assert sD['n_over'] <= sl['pairs'] and sd['n_over'] <= sl['pairs'], (sD, sd, sl['pairs'])
        assert sD['max_consumed'] <= 1.0 and sd['max_consumed'] <= 1.0
        assert_step(model.user_embedding.cpu().numpy(), Ub, t['gU'], lr, what=f'C3 U {tag}', slack=sl['gU'])
        assert_step(model.item_embedding.cpu().numpy(), Vb, t['gV'], lr, what=f'C3 V {tag}', slack=sl['gV'])
        return sl['pairs']
    one = fit_model(tm, U0, V0, idx, val, (m, n), 1, lr, 'wmrb', R, n, S_)
    assert check_step(one, U0, V0, losses[0], first, 'epoch 1') < 100
    # 20 epochs: the near-sign Adam step amplifies fp32 reordering, so the tail is compared at trajectory tolerance
    model = fit_model(tm, U0, V0, idx, val, (m, n), epochs, lr, 'wmrb', R, n, S_)
    assert rel_err(model.loss_history_[:5], losses[:5]) < 1e-5
    assert rel_err(model.loss_history_, losses) < 1e-3
    assert model.loss_history_[-1] < model.loss_history_[0]
    # teacher-forced step from the oracle's state after 20 epochs (hinges partly inactive by then)
    Un, Vn, mean, t = C.wmrb_epoch(Uc, Vc, plan, n, S_, lr)
    late = fit_model(tm, Uc, Vc, idx, val, (m, n), 1, lr, 'wmrb', R, n, S_)
    pairs = check_step(late, Uc, Vc, mean, t, 'epoch 21')
    # the criterion stays sharp: the boundary terms are a vanishing share of the hinge terms and of the rows
    assert pairs < 2e-4 * len(val) * S_, pairs
    # recall@10 of the engine-trained model against the oracle-trained one (BASELINE: within 1e-3)
    got = float(model.recall_at_k(torch.tensor(A)).mean())
    want = float(D.recall_at_k_dense(Uc, Vc, A, 10).mean())
    assert abs(got - want) <= 1e-3, (got, want)
    # ranking of the ORACLE's tables by the engine: bit-exact indices
    model.user_embedding, model.item_embedding = torch.tensor(Uc).cuda(), torch.tensor(Vc).cuda()
    assert np.array_equal(model.retrieve_user_recs(k=10), D.retrieve_user_recs_dense(Uc, Vc, k=10))


def tie_problem():
    """Dyadic factors: user u = [a_u, b_u, 0, 0], items 0..7 = [x_j, 0, 0, 0] (the positives), items 8..15 =
    [0, y_j, 0, 0] (the negatives), so p_k = a_u x_j and sp[u, s] = b_u y_s exactly and 1 - p_k + sp hits 0 exactly
    for the pairs on the diagonal (and for more once a_u, b_u scale them), with neighbours one ulp to either side."""
    e23, e24 = 2.0 ** -23, 2.0 ** -24
    x = np.array([2, 1.5, 1 + e23, 1, 0.5, 3, 1.25, 1 - e24], np.float32)
    y = np.array([1, 0.5, e23, 0, -0.5, 2, 0.25, -e24], np.float32)
    n, r = 16, 4
    V = np.zeros((n, r), np.float32)
    V[:8, 0], V[8:, 1] = x, y
    ab = [(1, 1), (1, 1), (2, 2), (0.5, 0.5), (1, 2), (2, 1), (1, 0.5), (1, 1), (4, 1), (1, 1)]
    m = len(ab)
    U = np.zeros((m, r), np.float32)
    U[:, 0], U[:, 1] = [p[0] for p in ab], [p[1] for p in ab]
    rng = np.random.default_rng(0)
    A = np.zeros((m, n), np.float32)
    for u in range(m):
        A[u, rng.choice(8, 6 if u else 8, replace=False)] = rng.integers(1, 6, 6 if u else 8)
    A[7, 9] = 3        # a positive that is also one of the "negative" items
    A[3, 2] = -2       # a non-positive stored value: contributes nothing
    S_ = 9
    R = np.stack([np.concatenate([rng.permutation(8)[:8] + 8, [rng.integers(0, 8)]]) for _ in range(m)])  # 8 negatives + 1 positive-type item
    R[0] = np.concatenate([np.arange(8) + 8, [0]])
    idx = np.argwhere(A != 0)
    return U, V, idx, A[A != 0], A, R.astype(np.int64), S_


@pytest.mark.parametrize('slices', [None, '3'])
def test_wmrb_subgradient_at_exact_ties(tm, monkeypatch, slices):  # noqa: F811
    from oracle import dense_ref as D
    from oracle import sparse_ref as S
    U0, V0, idx, val, A, R, S_ = tie_problem()
    m, n = A.shape
    if slices:
        monkeypatch.setenv('TMF_ITEM_SLICES', slices)
        monkeypatch.setenv('TMF_USER_CHUNKS', '2')
    # the data really holds exact ties and one-ulp neighbours on both sides (fp32 arithmetic as the reference does it)
    pos = val > 0
    p = np.einsum('kc,kc->k', U0[idx[pos, 0]], V0[idx[pos, 1]])
    sp = np.einsum('uc,usc->us', U0, V0[R])
    xks = (np.float32(1) - p)[:, None] + sp[idx[pos, 0]]
    assert (xks == 0).sum() >= 12 and ((xks > 0) & (xks < 1e-6)).sum() >= 2 and ((xks < 0) & (xks > -1e-6)).sum() >= 2
    lr = 0.05
    model, t = check_one_step(tm, U0, V0, idx, val, (m, n), lr, 'wmrb', R, n, S_, boundary_slack=False)
    assert model._state.wplan.n_slices == (int(slices) if slices else 1)
    # fp64 closed form with the `>=` rule
    assert rel_err(model._state.wplan.D_in_model_order().cpu().numpy(), t['D']) < 1e-6
    assert rel_err(model._state.wplan.delta.cpu().numpy()[pos], t['delta']) < 1e-6
    # every tie counted: cnt_k = -delta_k / w_k is an integer and equals the oracle's count of x >= 0
    w = (n / S_) / (1.0 + t['M'])
    cnt = -model._state.wplan.delta.cpu().numpy()[pos].astype(np.float64) / w
    assert np.array_equal(np.rint(cnt), (xks >= 0).sum(1)) and np.abs(cnt - np.rint(cnt)).max() < 1e-4
    # the dense autograd restatement (tf.maximum sub-gradient rule written out) agrees after one step
    ref = D.fit_dense(U0, V0, idx, val, 'wmrb', 1, lr, random_ind=R, n_items=n, n_samples=S_)
    assert abs(model.loss_history_[0] - ref['loss'][0]) <= 1e-5 * ref['loss'][0]
    assert_step(ref['U'], U0, t['gU'], lr, what='dense oracle U')
    assert_step(ref['V'], V0, t['gV'], lr, what='dense oracle V')
    # flipping the rule to a strict `>` would change the answer visibly: this test can fail
    strict = S.wmrb_terms(U0.astype(np.float64), V0.astype(np.float64), idx, val.astype(np.float64), R, n, S_)
    act_strict = (((1.0 - strict['p'])[:, None] + strict['sp'][idx[pos, 0]]) > 0).sum(1)
    assert (act_strict != (xks >= 0).sum(1)).any()


def hinge_reference(rowptr, val, p, sp, c):
    """loss_graphs.py:80-88 for given scores, in fp64 with the fp32 hinge arguments the reference forms:
    x = fl(fl(1 - p_k) + sp[u, s]); M = c sum max(x, 0); w = c / (1 + M); delta = -w #{x >= 0}; D[u, s] = sum_k w [x >= 0]."""
    m, S = sp.shape
    delta = np.zeros(len(p))
    D = np.zeros((m, S))
    loss = np.zeros(m)
    for u in range(m):
        for k in range(rowptr[u], rowptr[u + 1]):
            if not val[k] > 0:
                continue
            x = (np.float32(1) - np.float32(p[k])) + sp[u].astype(np.float32)
            act = x >= 0
            M = c * np.maximum(x.astype(np.float64), 0).sum()
            w = c / (1 + M)
            loss[u] += np.log1p(M)
            delta[k] = -w * act.sum()
            D[u] += w * act
    return delta, D, loss


@pytest.mark.parametrize('case', ['chunks', 'ties', 'wide', 'tiny', 'degenerate'])
def test_hinge_kernel_alone(tm, case):  # noqa: F811
    """tmf_wmrb_hinge2 through the C ABI on scores made up here: chunk boundaries (255 / 256 / 511 interactions of a
    user), sample counts that are no multiple of 64 and go beyond the two register-held tiles, many equal scores and
    thresholds (every bucket of the rank histogram, atomics landing on one address), users without positives, empty users,
    scores of very different magnitude (the fixed-point scale).  Against an fp64 evaluation of the reference's formula."""
    import ctypes
    rng = np.random.default_rng({'chunks': 1, 'ties': 2, 'wide': 3, 'tiny': 4, 'degenerate': 5}[case])
    lib = tm.lib.get()
    if case == 'chunks':
        degs, S_ = [0, 1, 254, 255, 256, 257, 510, 511, 700, 3], 300
    elif case == 'ties':
        degs, S_ = [40, 300, 5, 64, 65], 1024
    elif case == 'wide':
        degs, S_ = [10, 270, 33], 2500            # 5 tiles of 512: ranks are searched twice
    elif case == 'tiny':
        degs, S_ = [3, 1, 2, 70], 1
    else:
        degs, S_ = [20, 20, 20, 20, 0, 6], 130
    m = len(degs)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    nnz = int(rowptr[-1])
    val = rng.integers(-1, 5, nnz).astype(np.float32)           # ~1/3 of the entries are not positives
    p = rng.standard_normal(nnz).astype(np.float32)
    sp = rng.standard_normal((m, S_)).astype(np.float32)
    if case == 'ties':      # few distinct values on both sides: exact ties everywhere, buckets with hundreds of samples
        p = rng.integers(-2, 3, nnz).astype(np.float32) * 0.5
        sp = rng.integers(-3, 3, (m, S_)).astype(np.float32) * 0.5
    if case == 'degenerate':
        sp[0] = 5.0                                   # every sample active for every positive ("all" bucket only)
        sp[1] = -50.0                                 # none active: M = 0, w = c, delta = 0, D = 0
        val[rowptr[2]:rowptr[3]] = -1.0               # a user without positives
        p[rowptr[3]:rowptr[4]] = 0.25                 # all thresholds equal
        sp[3, ::2] *= 1e-6                            # magnitudes 1e-6 .. 1 in one row
        sp[5] = rng.standard_normal(S_).astype(np.float32) * 1e4
        p[rowptr[5]:rowptr[6]] = rng.standard_normal(6).astype(np.float32) * 1e4
    c = 37.5
    t = lambda a, dt: torch.tensor(a, dtype=dt, device='cuda')  # noqa: E731
    d_rowptr, d_val, d_p, d_sp = t(rowptr, torch.int64), t(val, torch.float32), t(p, torch.float32), t(sp, torch.float32)
    delta = torch.full((max(nnz, 1),), 7.0, device='cuda')
    D = torch.full((m, S_), 7.0, device='cuda')
    loss = torch.full((m,), 7.0, device='cuda')
    for _ in range(2):      # twice: the second launch must reproduce the bits of the first
        tm.lib.check(lib.tmf_wmrb_hinge2(tm.lib.ptr(d_rowptr), tm.lib.ptr(d_val), tm.lib.ptr(d_p), tm.lib.ptr(d_sp),
                                         ctypes.c_int32(m), ctypes.c_int32(S_), c, tm.lib.ptr(delta), tm.lib.ptr(D),
                                         tm.lib.ptr(loss), tm.lib.stream_ptr()), lib)
        torch.cuda.synchronize()
        got = (delta[:nnz].cpu().numpy().copy(), D.cpu().numpy().copy(), loss.cpu().numpy().copy())
        if _ == 0:
            first = got
    assert all(np.array_equal(a, b) for a, b in zip(first, got))
    r_delta, r_D, r_loss = hinge_reference(rowptr, val, p, sp, c)
    assert rel_err(got[0], r_delta) < 1e-5 and rel_err(got[1], r_D) < 1e-5
    assert np.abs(got[2] - r_loss).max() <= 1e-5 * max(np.abs(r_loss).max(), 1e-30)
    assert (got[0][~(val > 0)] == 0).all()
    # the ordered launch (heavy users first: _engine.hinge_user_order) computes every user by itself: same bits
    order = tm.engine.hinge_user_order(d_rowptr)
    if case in ('chunks', 'ties', 'wide'):
        assert order is not None and sorted(order.cpu().tolist()) == list(range(m))
        heavy_first = np.asarray(degs)[order.cpu().numpy()]
        assert (np.diff(-((heavy_first + 254) // 255)) >= 0).all()
    for o in (order, torch.randperm(m, device='cuda').to(torch.int32)):
        delta.fill_(7.0), D.fill_(7.0), loss.fill_(7.0)
        tm.lib.check(lib.tmf_wmrb_hinge2_ordered(tm.lib.ptr(d_rowptr), tm.lib.ptr(d_val), tm.lib.ptr(d_p), tm.lib.ptr(d_sp),
                                                 ctypes.c_int32(m), ctypes.c_int32(S_), c, tm.lib.ptr(delta), tm.lib.ptr(D),
                                                 tm.lib.ptr(loss), tm.lib.ptr(o), tm.lib.stream_ptr()), lib)
        torch.cuda.synchronize()
        again = (delta[:nnz].cpu().numpy(), D.cpu().numpy(), loss.cpu().numpy())
        assert all(np.array_equal(a, b) for a, b in zip(first, again))


@pytest.mark.parametrize('npos', [7, 15, 31, 63, 100, 255])
def test_hinge_kernel_with_an_infinite_score(tm, npos):  # noqa: F811
    """the binary search skips the steps a short chunk cannot need, so a sample of +inf (an overflowed score) no longer
    ran to slot 255 but stopped at 15 / 31 / 63 - and was counted or dropped depending on whether that happened to equal the
    number of positives.  It is active for EVERY positive now, as in the reference (loss_graphs.py:80-88: 1 - p + inf = inf, M = inf,
    loss = inf, w = c / (1 + inf) = 0): the user's loss is +inf, its delta and D are 0; the other users are untouched."""
    import ctypes
    rng = np.random.default_rng(npos)
    lib = tm.lib.get()
    S_, degs = 200, [npos, 9, npos]
    m = len(degs)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    nnz = int(rowptr[-1])
    val = np.ones(nnz, np.float32)
    p = rng.standard_normal(nnz).astype(np.float32)
    sp = rng.standard_normal((m, S_)).astype(np.float32)
    sp[0, 17] = np.inf
    sp[2, 3] = -np.inf          # never active: contributes nothing, like any very negative score
    c = 12.5
    t = lambda a, dt: torch.tensor(a, dtype=dt, device='cuda')  # noqa: E731
    d_rowptr, d_val, d_p, d_sp = t(rowptr, torch.int64), t(val, torch.float32), t(p, torch.float32), t(sp, torch.float32)
    delta, D, loss = torch.full((nnz,), 7.0, device='cuda'), torch.full((m, S_), 7.0, device='cuda'), torch.full((m,), 7.0, device='cuda')
    tm.lib.check(lib.tmf_wmrb_hinge2(tm.lib.ptr(d_rowptr), tm.lib.ptr(d_val), tm.lib.ptr(d_p), tm.lib.ptr(d_sp), ctypes.c_int32(m),
                                     ctypes.c_int32(S_), c, tm.lib.ptr(delta), tm.lib.ptr(D), tm.lib.ptr(loss), tm.lib.stream_ptr()), lib)
    torch.cuda.synchronize()
    delta, D, loss = delta.cpu().numpy(), D.cpu().numpy(), loss.cpu().numpy()
    with np.errstate(invalid='ignore', over='ignore'):
        r_delta, r_D, r_loss = hinge_reference(rowptr, val, p, sp, c)
    assert np.isposinf(r_loss[0]) and (r_delta[:npos] == 0).all() and (r_D[0] == 0).all()      # what the formula gives
    assert np.isposinf(loss[0]) and (delta[:npos] == 0).all() and (D[0] == 0).all()
    assert rel_err(delta[npos:], r_delta[npos:]) < 1e-5 and rel_err(D[1:], r_D[1:]) < 1e-5
    assert np.abs(loss[1:] - r_loss[1:]).max() <= 1e-5 * np.abs(r_loss[1:]).max()


def test_slice_grids_beyond_one_launch(tm, monkeypatch):  # noqa: F811
    """A HIP launch carries < 2^32 work-items.  1.1M users x 512 slices is 34375 user groups x 512 slices x 256 threads =
    4.5e9 for the slice-major kernels: they have to go out in several launches of whole slices (found in round 3, when such a
    grid was truncated silently and most (user, slice) ranges were skipped).  One epoch against the C oracle, in both block
    orders, plus the row-stationary gradU."""
    from oracle import sparse_c as C
    rng = np.random.default_rng(11)
    m, n, r, S_ = 1_100_000, 1024, 32, 16   # c = n / S = 64: a larger ratio amplifies the rounding of barely active hinge terms
    u = rng.integers(0, m, 2_400_000)
    j = rng.integers(0, n, 2_400_000)
    key = np.unique(u.astype(np.int64) * n + j)
    idx = np.stack([key // n, key % n], axis=1)
    val = rng.integers(-1, 6, len(key)).astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.permutation(n)[:S_] for _ in range(4096)])
    R = R[rng.integers(0, 4096, m)].astype(np.int32)       # distinct per row; rows drawn from 4096 patterns (a fast generator)
    lr = 0.05
    plan = C.Plan(idx, val, m, n, R)
    C.set_threads(16)
    Uc, Vc, mean, t = C.wmrb_epoch(U0, V0, plan, n, S_, lr)
    sl = C.wmrb_boundary_slack(U0, V0, plan, n, S_)
    monkeypatch.setenv('TMF_ITEM_SLICES', '512')
    for xcd, rs in (('0', '0'), ('1', '0'), ('0', '1')):
        monkeypatch.setenv('TMF_SLICE_XCD', xcd)
        monkeypatch.setenv('TMF_ROW_STATIONARY', rs)
        model = fit_model(tm, U0, V0, idx, val, (m, n), 1, lr, 'wmrb', R, n, S_)
        w = model._state.wplan
        assert w.sliced and w.n_slices == 512 and model._state.row_stationary == (rs == '1')
        assert abs(model.loss_history_[0] - mean) <= 1e-5 * abs(mean)
        assert_close_with_slack(w.D_in_model_order().cpu().numpy(), t['D'], sl['D'], what=f'D xcd={xcd} rs={rs}')
        assert_step(model.user_embedding.cpu().numpy(), U0, t['gU'], lr, what=f'U xcd={xcd} rs={rs}', slack=sl['gU'])
        assert_step(model.item_embedding.cpu().numpy(), V0, t['gV'], lr, what=f'V xcd={xcd} rs={rs}', slack=sl['gV'])
        del model
        torch.cuda.empty_cache()


def test_segment_lists_beyond_one_launch(tm):  # noqa: F811
    """The row passes launch one wave per list segment, two waves per workgroup: beyond 2^32 / 64 = 67.1M segments a single launch
    would exceed 2^32 work-items (the config-5 shard reaches that with more than ~130 user blocks x 1M items).  68M users with one
    interaction each: the MSE user pass goes out in two pieces; loss and sampled rows - first, last, and around the seam -
    against fp64 on the GPU."""
    dev = torch.device('cuda', 0)
    m, n, r, lr = 68_000_000, 1000, 4, 0.01
    g = torch.Generator(device=dev).manual_seed(3)
    u = torch.arange(m, device=dev)
    idx = torch.stack([u, u % n], dim=1)
    val = torch.randint(1, 6, (m,), device=dev, generator=g).to(torch.float32)
    U0 = torch.randn(m, r, device=dev, generator=g) * 0.3
    V0 = torch.randn(n, r, device=dev, generator=g) * 0.3
    plan = tm.engine.InteractionPlan(idx, val, m, n, csc=True)
    assert plan.seg_u.nseg == m > (1 << 32) // 64
    st = tm.engine.TrainState(U0, V0, plan, r)
    adam = tm.engine.adam_constants(lr)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    tm.engine.epoch_mse(st, adam, loss)
    torch.cuda.synchronize()
    p = (U0.double() * V0.double()[u % n]).sum(1)
    err = val.double() - p
    assert abs(float(loss[0]) - float((err * err).sum())) <= 1e-6 * float((err * err).sum())
    seam = (((1 << 32) // 128) - 1) * 2          # first segment of the second launch piece
    rows = torch.tensor([0, 1, 12345, seam - 2, seam - 1, seam, seam + 1, m - 2, m - 1], device=dev)
    gU = (-2.0 * err[rows])[:, None] * V0.double()[rows % n]
    assert_step(st.U_nxt[rows, :r].cpu().numpy(), U0[rows].cpu().numpy(), gU.cpu().numpy(), lr, what='users around the launch seam')
    # the item side (1000 rows of 68000 entries each = 67 segments per row + combine) sees every user once
    j = 7
    sel = torch.arange(j, m, n, device=dev)
    gV = ((-2.0 * err[sel])[:, None] * U0.double()[sel]).sum(0)
    assert_step(st.V_nxt[j:j + 1, :r].cpu().numpy(), V0[j:j + 1].cpu().numpy(), gV[None].cpu().numpy(), lr, what='item row')
