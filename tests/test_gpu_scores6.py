"""tmf_wmrb_scores6 - flat entry streams on the slice-major grid (one workgroup per (item slice, group of 32 users) chunk, the
group's rows in LDS; csrc/tmf_wmrb.hip, _engine.Scores6Plan) - against tmf_wmrb_scores3 and an fp64 product.  Both compute
sp[u, s] = <U[u], V[R[u, s]]> and p[k] = <U[u_k], V[j_k]> (matrix_factorization.py:153-154, utils.py:94-105); they differ in the
order of the fp32 sum inside a dot product only: equal to rounding on real data, bit for bit on dyadic data - and then the whole
epoch (hinge, gradients, fresh Adam) is bit-identical too."""
import numpy as np
import pytest
import torch

from conftest import assert_step
from test_gpu_scores5 import CASES, problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from teamoflow_amd import _engine, _lib
    _lib.get()
    return _engine


def epoch(eng, monkeypatch, s6, idx, val, R, U, V, m, n, r, S, dtype, slices, slice_bytes=None):
    monkeypatch.setenv('TMF_SCORES5', '0')
    monkeypatch.setenv('TMF_SCORES6', '1' if s6 else '0')
    if slice_bytes:
        monkeypatch.setenv('TMF_S6_SLICE_BYTES', str(slice_bytes))
    else:
        monkeypatch.delenv('TMF_S6_SLICE_BYTES', raising=False)
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.WmrbPlan(plan, R, item_slices=slices, n_components=r, sliced=True)
    st = eng.TrainState(U, V, plan, r, wplan, dtype=dtype)
    assert (wplan.s6 is not None) == s6 and wplan.s5 is None
    loss = torch.zeros(1, dtype=torch.float64, device=idx.device)
    eng.epoch_wmrb(st, eng.adam_constants(0.05), n / S, loss)
    torch.cuda.synchronize()
    return dict(sp=st.sp.clone(), pk=st.pk.clone(), U=st.U_nxt.float().clone(), V=st.V_nxt.float().clone(), loss=float(loss),
                D=wplan.D.clone(), delta=wplan.delta.clone(), plan=plan, wplan=wplan, st=st)


@pytest.mark.parametrize('m,n,r,S,nnz,dtype,slices', CASES + [(33, 5000, 128, 700, 4000, torch.float32, 2)])
def test_scores6_equals_scores3_to_rounding(eng, monkeypatch, m, n, r, S, nnz, dtype, slices):
    idx, val, R, U, V = problem(m, n, r, S, nnz, seed=m + n)
    row_bytes = 512
    a = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, slices)
    # slices of 37 rows: many chunks per group, most of them a few entries long, some empty
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices, slice_bytes=37 * row_bytes)
    s6 = b['wplan'].s6
    ug = int(eng._lib.load_library().tmf_wmrb_scores6_users_per_group())
    assert s6.n_slices >= n // 37 - 1 and s6.n_groups == -(-m // ug) and s6.n_padded >= s6.n_entries
    cp = s6.chunk_ptr.cpu().numpy()
    assert (np.diff(cp) % 8 == 0).all() and cp[-1] == s6.n_padded
    scale = float(a['sp'].abs().max())
    assert float((a['sp'] - b['sp']).abs().max()) <= 2e-6 * scale
    assert float((a['pk'] - b['pk']).abs().max()) <= 2e-6 * scale
    assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(a['loss'])
    st = b['st']
    U64, V64 = st.U[:, :r].double(), st.V[:, :r].double()
    want = torch.einsum('ur,usr->us', U64, V64[b['wplan'].R.long()])
    assert float((b['sp'].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    pl = b['plan']
    wantp = (U64[pl.user_of.long()] * V64[pl.col_u.long()]).sum(1)
    assert float((b['pk'].double() - wantp).abs().max()) <= 1e-5 * float(want.abs().max())
    # the default slice width (one 4 MB slice for these small catalogs) gives the same numbers as the tiny slices, bit for bit:
    # a score does not depend on the chunk it is computed in
    c = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices)
    assert torch.equal(b['sp'], c['sp']) and torch.equal(b['pk'], c['pk'])


@pytest.mark.parametrize('m,n,r,S,nnz,dtype,slices', [CASES[0], CASES[1], CASES[3]])
def test_scores6_epoch_is_bit_identical_on_dyadic_tables(eng, monkeypatch, m, n, r, S, nnz, dtype, slices):
    idx, val, R, U, V = problem(m, n, r, S, nnz, seed=5 * m + n, dyadic=True)
    a = epoch(eng, monkeypatch, False, idx, val, R, U, V, m, n, r, S, dtype, slices)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, dtype, slices, slice_bytes=100 * 512)
    for k in ('sp', 'pk', 'D', 'delta', 'U', 'V'):
        assert torch.equal(a[k], b[k]), k
    assert a['loss'] == b['loss']


def test_scores6_one_step_against_the_fp64_closed_form(eng, monkeypatch):
    from oracle import sparse_ref as SR
    m, n, r, S, lr = 600, 900, 128, 48, 0.05
    idx, val, R, U, V = problem(m, n, r, S, 12000, seed=77)
    b = epoch(eng, monkeypatch, True, idx, val, R, U, V, m, n, r, S, torch.float32, 2, slice_bytes=64 * 512)
    U64, V64 = U.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.float64)
    i_np, v_np, R_np = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.int64)
    _, _, mean, t = SR.wmrb_epoch(U64, V64, i_np, v_np, R_np, n, S, lr)
    sl = SR.wmrb_slack(U64, V64, i_np, v_np, R_np, n, S)
    n_pos = int((val > 0).sum())
    assert abs(b['loss'] / n_pos - mean) <= 1e-5 * abs(mean)
    assert_step(b['U'][:, :r].cpu().numpy(), U64, t['gU'], lr, what='scores6 U', slack=sl['gU'])
    assert_step(b['V'][:, :r].cpu().numpy(), V64, t['gV'], lr, what='scores6 V', slack=sl['gV'])


def test_scores6_geometries(eng, monkeypatch):
    """The kernel exists for rows of 32 lanes, fewer than 2^24 items and tables below 4 GB; everything else stays with scores3,
    whatever TMF_SCORES6 says."""
    class P:
        pass
    for n_items, r, dtype, has_kernel in ((1_000_000, 256, torch.bfloat16, True), (100_000, 128, torch.float32, True),
                                          (1_000_000, 64, torch.float32, False), (20_000_000, 128, torch.float32, False)):
        plan, w = P(), P()
        plan.n_items, plan.nnz, plan.n_users, plan.col_u = n_items, 100 * 1000, 1000, torch.zeros(1, device='cuda')
        w.sliced, w.R = True, torch.zeros(1000, 1024)
        monkeypatch.setenv('TMF_SCORES5', '0')
        monkeypatch.setenv('TMF_SCORES6', '0')
        assert not eng.scores6_wanted(plan, w, r, dtype)
        monkeypatch.setenv('TMF_SCORES6', '1')
        assert eng.scores6_wanted(plan, w, r, dtype) == has_kernel, (n_items, r)


def test_stated_catalog_size_is_checked_on_request(eng, monkeypatch):
    """tmf_slice_lists.n_items switches the 32-bit-offset walk of tmf_wmrb_scores3 on - it counts only with
    TMF_SLICE_N_ITEMS_STATED in `flags` (a caller built against the older, shorter struct passes indeterminate bytes there), and
    TMF_CHECK_IDS=1 makes the call verify that every id of the lists is below it."""
    import ctypes
    from teamoflow_amd import _lib
    lib = _lib.get()
    m, n, r, S = 300, 5000, 128, 40
    idx, val, R, U, V = problem(m, n, r, S, 4000, seed=3)
    monkeypatch.setenv('TMF_SCORES6', '0')
    monkeypatch.setenv('TMF_SCORES5', '0')
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.WmrbPlan(plan, R, item_slices=3, n_components=r, sliced=True)
    st = eng.TrainState(U, V, plan, r, wplan)

    def lists(flags, n_items):
        return _lib.SliceLists(wplan.R.data_ptr(), wplan.slice_off.data_ptr(), plan.rowptr_u.data_ptr(), plan.col_u.data_ptr(),
                               wplan.pos_off.data_ptr(), m, S, 3, 0, 0, 0, flags, n_items)

    def scores(sl):
        return lib.tmf_wmrb_scores3_f32(ctypes.byref(sl), _lib.ptr(st.U), _lib.ptr(st.V), _lib.ptr(st.sp), _lib.ptr(st.pk), r, _lib.stream_ptr())
    monkeypatch.setenv('TMF_CHECK_IDS', '1')
    assert scores(lists(_lib.SLICE_N_ITEMS_STATED, n)) == 0                 # the true size: fine
    want = st.sp.clone()
    assert scores(lists(_lib.SLICE_N_ITEMS_STATED, n // 2)) != 0            # ids beyond the stated size: refused, with a count
    assert b'outside [0, n_items' in lib.tmf_last_error()
    # without the flag the field is ignored (garbage in it cannot switch the lean walk on, nor trip the check): same scores
    st.sp.zero_()
    assert scores(lists(0, 7)) == 0
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert float((st.sp - want).abs().max()) <= 2e-6 * scale
