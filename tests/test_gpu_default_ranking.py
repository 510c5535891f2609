"""The DEFAULT arithmetic of the ranking path (retrieve_user_recs / recall_at_k with model.predict_arithmetic unset) keeps the
reference's fp32 operands whole: the fp32 MFMA kernel, or - from 2^26 scores on - the three-bf16-plane split (24 bits of every
factor, exact plane products).  The 22-bit two-plane fp16 form is opt-in only.  Checked index for index against
oracle.dense_ref.tf_top_k (tf.math.top_k of U V^T: /root/reference/src/teamoflow/mf/matrix_factorization.py:236-248, 424-438)
on the golden tables of C1 / C2 / C3 and, in tests/test_gpu_fullsize.py, on C4-trained tables."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)


@pytest.fixture(scope='module')
def ops():
    from teamoflow_amd import _lib, _ops
    _lib.get()
    return _ops


def golden_tables():
    out = {}
    g = dict(np.load(os.path.join(GOLDEN, 'c1_mse.npz')))
    out['C1 after 450 epochs'] = (g['U_450'], g['V_450'], g['top10'])
    out['C1 after 1 epoch'] = (g['U_1'], g['V_1'], None)
    g = dict(np.load(os.path.join(GOLDEN, 'c2_mse.npz')))
    out['C2 after 1 epoch'] = (g['U_1'], g['V_1'], None)
    g = dict(np.load(os.path.join(GOLDEN, 'c3r_wmrb.npz')))
    out['C3 (reduced) after 1 epoch'] = (g['U_1'], g['V_1'], None)
    g = dict(np.load(os.path.join(GOLDEN, 'wmrb_small.npz')))
    out['WMRB small after 25 epochs'] = (g['U_25'], g['V_25'], g['top10'])
    return out


@pytest.mark.parametrize('force_planes', [False, True])
def test_default_ranking_equals_tf_top_k_on_the_golden_tables(ops, monkeypatch, force_planes):
    """retrieve_user_recs(k=10) with nothing selected, on every golden table: index for index the oracle's tf_top_k of the
    dense fp32 product wherever the oracle's top 11 values are separated by more than 1e-5 of the row's best (a CPU matmul and
    an MFMA chain sum r products in different orders), the fixture's own top-10 where it holds one, and by value in near-ties.
    force_planes: the threshold from which 'auto' takes the three-plane kernel set to 0, so that kernel ranks these small
    tables too (at their real size they go to the fp32 MFMA)."""
    from oracle import dense_ref as D
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    if force_planes:
        monkeypatch.setattr(ops, 'SPLIT_MIN_SCORES', 0)
    for name, (U, V, top10) in golden_tables().items():
        model = MatrixFactorization(U.shape[1])
        assert model.predict_arithmetic is None
        model.user_embedding, model.item_embedding = torch.tensor(U).cuda(), torch.tensor(V).cuda()
        got = np.asarray(model.retrieve_user_recs(k=10)).astype(np.int64)
        S = torch.tensor(D.predict_dense(U, V))
        k1 = min(11, S.shape[1])
        wv, wi = D.tf_top_k(S, k1)
        gaps = (wv[:, :-1] - wv[:, 1:]) / wv[:, :1].abs().clamp_min(1e-30)
        clear = (gaps > 1e-5).all(dim=1).numpy()
        assert clear.mean() > 0.9, (name, clear.mean())
        assert np.array_equal(got[clear], wi.numpy()[clear, :10]), name
        picked = np.take_along_axis(S.numpy(), got, 1)
        assert np.abs(picked - wv.numpy()[:, :10]).max() <= 1e-5 * np.abs(wv.numpy()).max(), name
        if top10 is not None and not force_planes:   # the fixture's list, bit for bit (fp32 MFMA = the fmaf chain of the oracle)
            assert np.array_equal(got, top10), name


def test_auto_never_takes_the_two_plane_form(ops, monkeypatch):
    """From 2^26 scores on 'auto' runs the three-plane kernel (width <= 256, k <= 40) or the fp32 MFMA - never half2, whatever
    the range of the item rows; half2 still answers when asked for by name."""
    from teamoflow_amd import _lib
    lib = _lib.get()
    calls = []
    real = {n: getattr(lib, n) for n in ('tmf_predict_topk_half2_f32', 'tmf_predict_topk_split_f32', 'tmf_predict_topk_f32')}
    for n, fn in real.items():
        monkeypatch.setattr(lib, n, (lambda n_, fn_: (lambda *a: (calls.append(n_), fn_(*a))[1]))(n, fn))
    g = torch.Generator().manual_seed(3)
    m, n, r = 2048, 40_000, 128
    assert m * n >= ops.SPLIT_MIN_SCORES
    U, V = (torch.randn(m, r, generator=g) * 0.05).cuda(), (torch.randn(n, r, generator=g) * 0.05).cuda()
    want = torch.topk(U.double() @ V.double().T, 10, dim=1)[1]
    idx = ops.predict_topk(U, V, 10)
    assert calls == ['tmf_predict_topk_split_f32']
    assert float((idx.long() == want).all(1).float().mean()) > 0.995
    calls.clear()
    ops.predict_topk(U, V, 41)                                     # beyond the plane kernels' k: the fp32 MFMA
    ops.predict_topk(U[:100], V, 10)                               # a small job: the fp32 MFMA
    ops.predict_topk(U, V, 40)                                     # the plane kernel's last k (round 5: 32 -> 40)
    assert calls == ['tmf_predict_topk_f32', 'tmf_predict_topk_f32', 'tmf_predict_topk_split_f32']
    calls.clear()
    ops.predict_topk(U[:, :32].contiguous(), V[:, :32].contiguous(), 40)   # narrow tables beyond k = 32: the fp32 kernel is the faster one
    ops.predict_topk(U[:, :32].contiguous(), V[:, :32].contiguous(), 32)
    assert calls == ['tmf_predict_topk_f32', 'tmf_predict_topk_split_f32']
    calls.clear()
    ops.predict_topk(U, V, 10, arithmetic='half2')
    assert calls == ['tmf_predict_topk_half2_f32']


@pytest.mark.parametrize('arith', ['split', 'fp32'])
def test_rows_with_k_or_more_infinite_scores_keep_valid_ids(ops, arith):
    """a row with >= k scores of +inf in the warm-up tiles (overflowed / diverged factors) made the warm-up
    threshold inf - inf = NaN, every candidate of the scan was rejected and the row came back with ids 0x7fffffff.  The
    catalog is long enough for the warm-up pass (>= 256 tiles of 128)."""
    n, r, k, m = 40_000, 64, 10, 300
    g = torch.Generator().manual_seed(9)
    U, V = torch.randn(m, r, generator=g) * 0.1, torch.randn(n, r, generator=g) * 0.1
    hot = torch.arange(0, 400, 13)                      # 31 items in the first 1/64 of the catalog, on 31 different lanes ...
    # 2^127 is one bf16 plane (the other two are 0), so the plane kernel's products overflow to +inf like the fp32 kernel's; a value
    # with a NEGATIVE low plane (3.0e38 rounds up in bf16) gives +inf - inf = NaN there: such scores are not ranked (_ops.predict_topk)
    V[hot, 0] = 2.0 ** 127
    U[:50, 0] = 2.0 ** 127                              # ... whose score with users 0..49 overflows to +inf
    U[:50, 1:] = 0
    V[hot, 1:] = 0
    vals, idx = ops.predict_topk(U, V, k, return_values=True, arithmetic=arith)
    idx, vals = idx.cpu().long(), vals.cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    assert torch.equal(idx[:50], hot[:k].expand(50, k)) and bool(torch.isinf(vals[:50]).all())   # ties at +inf: lowest ids first
    from oracle import dense_ref as D
    rest = D.tf_top_k(U[50:].double() @ V.double().T, k)[1]      # the hot items tie exactly for a user: lower id first
    assert float((idx[50:] == rest).all(1).float().mean()) > 0.99


def test_tiny_tables_keep_their_values_in_the_two_plane_form(ops):
    """1 / (row scale * table scale) overflowed to inf -> 0 for tables below ~2^-97 and every returned value was 0."""
    g = torch.Generator().manual_seed(1)
    U, V = torch.randn(64, 16, generator=g) * 2.0 ** -60, torch.randn(500, 16, generator=g) * 2.0 ** -50
    vals, idx = ops.predict_topk(U, V, 5, return_values=True, arithmetic='half2')
    ref = U.double() @ V.double().T
    want = torch.gather(ref, 1, idx.cpu().long())
    assert float(vals.abs().min()) > 0 and float(((vals.cpu().double() - want).abs() / ref.abs().amax(1, keepdim=True)).max()) < 1e-5


@pytest.mark.parametrize('cand', ['0', '1', '2'])
@pytest.mark.parametrize('k', [1, 10, 28, 33, 64])
def test_every_candidate_path_keeps_the_tie_rule(ops, monkeypatch, cand, k):
    """tf.math.top_k: value descending, equal values lower index first (matrix_factorization.py:245, 429-438).  The three ways
    candidates reach a row's list in the fp32 fused kernel - pending buffer + one-lane merges, 64-lane insertion, pending
    buffer + wave-wide bitonic merges (round 4, the default from k = 28) - on a catalog where ties are everywhere: a few
    distinct score levels, thousands of items on each, and a clamp that ties every negative score at 0."""
    from oracle import sparse_ref as S
    monkeypatch.setenv('TMF_PREDICT_CAND', cand)
    g = torch.Generator().manual_seed(k)
    m, n, r = 300, 9001, 16
    U = torch.zeros(m, r)
    V = torch.zeros(n, r)
    U[:, 0] = torch.randint(1, 4, (m,), generator=g).float()
    V[:, 0] = torch.randint(-2, 3, (n,), generator=g).float()            # five score levels per user, ~1800 items on each
    U[:, 1] = 1.0
    V[::1000, 1] = 0.5                                                    # and a few items lifted off their level
    sc = (U @ V.T).numpy()
    for clamp in (False, True):
        vals, idx = ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='fp32')
        ref = S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)
        assert np.array_equal(idx.cpu().numpy(), ref), (cand, k, clamp)
        assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, 1))
