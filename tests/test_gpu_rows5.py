"""tmf_wsum_rows5 - the BALANCED row-stationary item pass (csrc/tmf_train.hip k_wsum_rows4 with virtual rows, _engine.VirtualRows):
lane groups own (item, part) virtual rows, keep their sums in registers and walk the user blocks; a popular item is cut into parts
of about the mean size, its partial sums are added by tmf_combine_rows in part order.  It computes the gradient of the sum of the
WMRB losses with respect to V (matrix_factorization.py:170-171 through loss_graphs.py:80-88) like tmf_wsum_pass + tmf_combine_rows:
checked against that form, against the fp64 closed form of the oracle and for bit-reproducibility."""
import numpy as np
import pytest
import torch

from conftest import assert_step, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from teamoflow_amd import _engine, _lib
    _lib.get()
    return _engine


def problem(m, n, r, S, nnz, seed, heavy=3, dev='cuda'):
    """Random interactions + `heavy` items that (nearly) every user has - as a positive or as a negative: the popular items of a
    power-law catalog, whose lists are hundreds of times the mean."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, m, (nnz,), generator=g)
    j = torch.randint(0, n, (nnz,), generator=g)
    hu = torch.arange(m).repeat(heavy)
    hj = torch.arange(heavy).repeat_interleave(m)
    keep = torch.rand(hu.numel(), generator=g) < 0.8
    key = torch.unique(torch.cat([u * n + j, (hu * n + hj)[keep]]))
    idx = torch.stack([key // n, key % n], 1)
    val = torch.randint(-1, 6, (key.numel(),), generator=g).float()
    R = torch.stack([torch.randperm(n, generator=g)[:S] for _ in range(m)]).to(torch.int32)
    R[::2, 0] = n - 1                                                  # a heavy NEGATIVE: item n-1 sampled by every second user
    U = torch.randn(m, r, generator=g) * 0.3
    V = torch.randn(n, r, generator=g) * 0.3
    return idx.to(dev), val.to(dev), R.to(dev), U.to(dev), V.to(dev)


def item_gradient(eng, monkeypatch, form, idx, val, R, U, V, m, n, r, S, dtype, chunks, target=None, slices=2):
    """(raw fp32 gradient of V, updated V, the plan) of one epoch with the item pass in the given form."""
    from teamoflow_amd import _lib
    monkeypatch.setenv('TMF_ROWS4', '0' if form == 'slab' else '1')
    monkeypatch.setenv('TMF_ROWS5', '0' if form == 'rows4' else '1')
    monkeypatch.setenv('TMF_USER_CHUNKS', str(chunks))
    if target is None:
        monkeypatch.delenv('TMF_ROWS5_TARGET', raising=False)
    else:
        monkeypatch.setenv('TMF_ROWS5_TARGET', str(target))
    monkeypatch.setenv('TMF_FORCE_SLICED', '1')
    monkeypatch.setenv('TMF_ITEM_SLICES', str(slices))
    plan = eng.InteractionPlan(idx, val, m, n)
    wplan = eng.wmrb_plan_for(plan, R, r, dtype)
    assert wplan.rows4 == (form != 'slab') and (wplan.vrows is not None) == (form == 'rows5') and wplan.user_chunks == chunks
    st = eng.TrainState(U, V, plan, r, wplan, dtype=dtype)
    loss = torch.zeros(1, dtype=torch.float64, device=idx.device)
    gV = torch.full((n, st.ld), 7.0, device=idx.device)
    adam = eng.adam_constants(0.05)
    eng.epoch_wmrb(st, adam, n / S, loss, item_epi=_lib.EPI_GRAD, item_out=gV)
    eng.epoch_wmrb(st, adam, n / S, loss)
    torch.cuda.synchronize()
    return gV[:, :r].clone(), st.V_nxt[:, :r].float().clone(), wplan, st


CASES = [  # m, n, r, S, nnz, dtype, user blocks, target
    (900, 700, 128, 40, 9000, torch.float32, 5, None),
    (900, 700, 128, 40, 9000, torch.float32, 5, 3),          # nearly every row cut, hundreds of parts for the heavy ones
    (300, 257, 100, 17, 2500, torch.float32, 1, 16),         # one user block, ragged width, a last row group that is not full
    (2000, 1500, 256, 64, 30000, torch.bfloat16, 7, None),
    (2000, 1500, 256, 64, 30000, torch.bfloat16, 7, 50),
    (500, 90, 64, 12, 1, torch.float32, 3, 4),               # a single interaction
]


@pytest.mark.parametrize('m,n,r,S,nnz,dtype,chunks,target', CASES)
def test_rows5_equals_the_slab_form_and_the_closed_form(eng, monkeypatch, m, n, r, S, nnz, dtype, chunks, target):
    from oracle import sparse_ref as SR
    idx, val, R, U, V = problem(m, n, r, S, nnz, seed=m + n + (target or 0))
    ga, Va, wa, _ = item_gradient(eng, monkeypatch, 'slab', idx, val, R, U, V, m, n, r, S, dtype, chunks)
    gb, Vb, wb, st = item_gradient(eng, monkeypatch, 'rows5', idx, val, R, U, V, m, n, r, S, dtype, chunks, target)
    v = wb.vrows
    if target is not None and nnz > 1:
        assert v.n_long >= 2 and v.max_parts >= 8          # the heavy items really are cut
    scale = float(ga.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-6 * scale    # another order of the same fp32 additions
    # the fp64 closed form on the tables as stored
    U64, V64 = st.U[:, :r].double().cpu().numpy(), st.V[:, :r].double().cpu().numpy()
    i_np, v_np, R_np = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.int64)
    _, _, _, t = SR.wmrb_epoch(U64, V64, i_np, v_np, R_np, n, S, 0.05)
    sl = SR.wmrb_slack(U64, V64, i_np, v_np, R_np, n, S)
    d = np.abs(gb.double().cpu().numpy() - t['gV']) - 1.0001 * sl['gV']
    assert d.max() <= 1e-5 * np.abs(t['gV']).max()
    if dtype is torch.float32:
        assert_step(Vb.cpu().numpy(), V64, t['gV'], 0.05, what='rows5 V', slack=sl['gV'])
    # bit-reproducible: a second run of the same form gives the same bits
    gc, Vc, _, _ = item_gradient(eng, monkeypatch, 'rows5', idx, val, R, U, V, m, n, r, S, dtype, chunks, target)
    assert torch.equal(gb, gc) and torch.equal(Vb, Vc)


def test_plain_rows4_is_still_there_and_agrees(eng, monkeypatch):
    m, n, r, S = 700, 400, 128, 30
    idx, val, R, U, V = problem(m, n, r, S, 6000, seed=5)
    ga, _, _, _ = item_gradient(eng, monkeypatch, 'slab', idx, val, R, U, V, m, n, r, S, torch.float32, 4)
    gb, _, _, _ = item_gradient(eng, monkeypatch, 'rows4', idx, val, R, U, V, m, n, r, S, torch.float32, 4)
    gc, _, w, _ = item_gradient(eng, monkeypatch, 'rows5', idx, val, R, U, V, m, n, r, S, torch.float32, 4, target=10 ** 9)
    assert float((ga - gb).abs().max()) <= 2e-6 * float(ga.abs().max())
    assert w.vrows.n_long == 0 and w.vrows.n_vrows == n and torch.equal(gb, gc)   # no row cut: rows5 IS rows4, bit for bit


def test_the_item_pass_form_follows_the_shape(eng, monkeypatch):
    """Row-stationary exactly where the slab form cannot block the users for the L2s (its slab would exceed the budget at ~4 MB
    blocks): the config-5 shard, not C4 and not the small shapes.  TMF_ROWS4 = 0 | 1 forces either."""
    monkeypatch.delenv('TMF_ROWS4', raising=False)
    monkeypatch.delenv('TMF_SLAB_BUDGET', raising=False)
    bf, f32 = torch.bfloat16, torch.float32
    assert eng.rows4_wanted(256, bf, n_users=1_250_000, n_items=1_000_000)          # one GPU's share of config 5
    assert not eng.rows4_wanted(128, f32, n_users=1_000_000, n_items=100_000)       # C4: 163 blocks, 8.3 GB of slab
    assert not eng.rows4_wanted(64, f32, n_users=6040, n_items=3706)                # C3
    assert not eng.rows4_wanted(8, f32, n_users=10_000_000, n_items=10_000_000)     # rows of fewer than 16 lanes: no such kernel
    monkeypatch.setenv('TMF_ROWS4', '1')
    assert eng.rows4_wanted(128, f32, n_users=1000, n_items=1000)
    monkeypatch.setenv('TMF_ROWS4', '0')
    assert not eng.rows4_wanted(256, bf, n_users=1_250_000, n_items=1_000_000)
    assert eng.rows5_user_chunks(1_250_000, 256, bf) == 153      # 4 MB of bf16 rows per block


def test_the_user_pass_forms_follow_the_shape_too(eng, monkeypatch):
    """Short (user, 4 MB slice) visits - the config-5 shard: 9 rows - take the flat-stream scores kernel and the row-stationary
    gradU with 3.2 MB slices; C4 (88 rows per visit) and the small configurations keep scores3 / gradu3."""
    for k in ('TMF_ROW_STATIONARY', 'TMF_SCORES6', 'TMF_SCORES5', 'TMF_ITEM_SLICES'):
        monkeypatch.delenv(k, raising=False)
    bf, f32 = torch.bfloat16, torch.float32
    assert eng.short_visits(1_250_000, 1_000_000, 1024, 125_000_000, 256, bf)
    assert not eng.short_visits(1_000_000, 100_000, 1024, 100_000_000, 128, f32)
    assert not eng.short_visits(6040, 3706, 1853, 1_000_209, 64, f32) and not eng.short_visits(50, 100, 5, 200, 128, f32)
    assert eng.row_stationary_wanted(1_250_000, 1_000_000, 1024, 125_000_000, 256, bf)
    assert not eng.row_stationary_wanted(1_000_000, 100_000, 1024, 100_000_000, 128, f32)
    monkeypatch.setenv('TMF_ROW_STATIONARY', '0')
    assert not eng.row_stationary_wanted(1_250_000, 1_000_000, 1024, 125_000_000, 256, bf)
