"""tmf_predict_topk_split_f32 (fp32 factors on the bf16 matrix cores, three exact bf16 planes, six products): the same
contract as the fp32 MFMA kernel - tf.math.top_k order (value desc, index asc), values fp32-accurate - checked against the
CPU oracle (oracle/sparse_ref.topk_stable, oracle/dense_ref.tf_top_k) and an fp64 product.  Reference semantics:
matrix_factorization.py:236-248 (recall_at_k) and :424-438 (retrieve_user_recs): tf.math.top_k(U V^T, k)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from teamoflow_amd import _lib, _ops
    _lib.get()
    return _ops


def expected(sc, k, clamp):
    from oracle import sparse_ref as S
    return S.topk_stable(np.where(sc > 0, sc, 0) if clamp else sc, k)


@pytest.mark.parametrize('arith', ['split', 'half2'])
@pytest.mark.parametrize('seed', range(int(os.environ.get('TMF_FUZZ_SEEDS', '10'))))
def test_exact_on_small_integer_factors(ops, seed, arith):
    """Small-integer factors: every plane product and every sum is exact, so the split kernel must return the oracle's
    ranking bit for bit - with exact score ties across tiles, clamping, ragged last tiles, every table width class
    (<= 32, <= 64, <= 128) and catalogs long enough for the warm-up pass (>= 256 tiles)."""
    rng = np.random.default_rng(4100 + seed)
    m = int(rng.integers(1, 600))
    n = int(rng.choice([1, 7, 128, 129, 1000, 4097, 16384, 20011, 33000, 40000]))
    r = int(rng.choice([1, 3, 8, 31, 32, 33, 50, 64, 65, 100, 127, 128, 129, 192, 255, 256]))
    k = int(min(n, rng.choice([1, 2, 5, 10, 16, 17, 32] if arith == 'half2' else [1, 2, 5, 10, 16, 17, 23, 32, 33, 39, 40])))
    span = int(rng.choice([1, 2, 4]))
    U = rng.integers(-span, span + 1, (m, r)).astype(np.float32)
    V = rng.integers(-span, span + 1, (n, r)).astype(np.float32)
    clamp = bool(rng.integers(0, 2))
    sc = U @ V.T
    vals, got = ops.predict_topk(torch.tensor(U), torch.tensor(V), k, clamp_negatives=clamp, return_values=True, arithmetic=arith)
    ref = expected(sc, k, clamp)
    assert np.array_equal(got.cpu().numpy(), ref), (m, n, r, k, clamp)
    assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, 1))


@pytest.mark.parametrize('m,n,r,k,clamp', [(700, 3000, 128, 10, False), (300, 129, 5, 3, False), (256, 128, 32, 32, True),
                                            (513, 5000, 64, 16, False), (300, 1000, 100, 10, True), (1024, 40000, 128, 10, False),
                                            (640, 33001, 64, 5, False), (333, 70000, 24, 1, False)])
@pytest.mark.parametrize('arith', ['split', 'half2'])
def test_values_are_fp32_accurate(ops, m, n, r, k, clamp, arith):
    """Gaussian factors at the scale of normalised tables: values against an fp64 product (tolerance 1e-6 of the largest score -
    the 1e-5 gate of the predictions with a decade to spare - and no worse than twice the fp32 MFMA kernel's own error), the
    ranking against the oracle's top_k of the fp64 scores wherever the k-th and (k+1)-th scores are further apart than that."""
    from oracle import dense_ref as D
    g = torch.Generator().manual_seed(m * 31 + n)
    U = (torch.randn(m, r, generator=g) * 0.05).float()
    V = (torch.randn(n, r, generator=g) * 0.05).float()
    ref = U.double() @ V.double().T
    if clamp:
        ref = ref.clamp_min(0)
    norm = float(ref.abs().max())
    v32, i32 = ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='fp32')
    if arith == 'half2':   # rows of very different magnitude: every user row has its own scale, the item table one
        U = U * torch.exp2(torch.randint(-20, 20, (m, 1), generator=g).float())
        V = V * torch.exp2(torch.randint(-5, 6, (n, 1), generator=g).float())
        ref = U.double() @ V.double().T
        if clamp:
            ref = ref.clamp_min(0)
        v32, i32 = ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic='fp32')
    vs, ix = ops.predict_topk(U, V, k, clamp_negatives=clamp, return_values=True, arithmetic=arith)
    vs, ix, v32, i32 = vs.cpu(), ix.cpu().long(), v32.cpu(), i32.cpu().long()
    if arith == 'half2':   # judged row by row: a user's scores share the user's scale
        norm = ref.abs().amax(1, keepdim=True).clamp_min(1e-300)
        e32 = float(((v32.double() - torch.gather(ref, 1, i32)).abs() / norm).max())
        es = float(((vs.double() - torch.gather(ref, 1, ix)).abs() / norm).max())
        assert es < 1.5e-6 and es <= 2 * e32 + 2e-7, (es, e32)
        kk = min(k + 1, n)
        wv, wi = D.tf_top_k(ref, kk)
        gaps = wv[:, :-1] - wv[:, 1:]
        clear = (gaps.min(1).values > 6e-6 * norm[:, 0]) if gaps.numel() else torch.ones(m, dtype=torch.bool)
        assert torch.equal(ix[clear], wi[clear, :k])
        return
    e32 = float((v32.double() - torch.gather(ref, 1, i32)).abs().max()) / norm
    es = float((vs.double() - torch.gather(ref, 1, ix)).abs().max()) / norm
    assert es < 1e-6 and es <= 2 * e32 + 1e-7, (es, e32)
    if k > 1:   # sorted under the total order
        assert bool(((vs[:, :-1] > vs[:, 1:]) | ((vs[:, :-1] == vs[:, 1:]) & (ix[:, :-1] < ix[:, 1:]))).all())
    kk = min(k + 1, n)
    wv, wi = D.tf_top_k(ref, kk)
    gaps = (wv[:, :-1] - wv[:, 1:]) if kk > k else (wv[:, :-1] - wv[:, 1:])
    clear = (gaps.min(1).values > 4e-6 * norm) if gaps.numel() else torch.ones(m, dtype=torch.bool)
    assert clear.float().mean() > 0.5 or clamp
    assert torch.equal(ix[clear], wi[clear, :k])


def test_split_is_exact_per_value(ops):
    """One user per value x, one item [1.0]: the score is x3 + x2 + x1 summed in fp32 and must give x back exactly - the three
    planes lose nothing, for values across the exponent range, values that round up to the next binade in bf16, and a value
    that would round to inf in bf16 (truncated instead)."""
    xs = np.array([0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 0.1, -0.3333333, 3.14159274, 255.99998, 1e-20, -7.7e-12, 6.5e4 + 1,
                   1.9999999, 3.3e38, -3.3e38, 1e-30, 123456.789, 2 ** -100 * 1.2345678], dtype=np.float32)
    rng = np.random.default_rng(3)
    xs = np.concatenate([xs, (rng.standard_normal(4000) * np.exp(rng.uniform(-30, 30, 4000))).astype(np.float32)])
    U = torch.tensor(xs[:, None].copy())
    vals, idx = ops.predict_topk(U, torch.ones(1, 1), 1, return_values=True, arithmetic='split')
    assert np.array_equal(vals.cpu().numpy()[:, 0], xs) and int(idx.abs().max()) == 0


def test_half2_keeps_22_bits_per_value(ops):
    """One user per value x, one item [1.0]: two fp16 planes under the row's power-of-two scale give x back to 22 bits."""
    rng = np.random.default_rng(4)
    xs = (rng.standard_normal(4000) * np.exp(rng.uniform(-30, 30, 4000))).astype(np.float32)
    xs = np.concatenate([np.array([0.0, 1.0, -1.0, 0.1, 3.14159274, 65504.0, 1e-20, -7.7e-12, 123456.789], dtype=np.float32), xs])
    vals, _ = ops.predict_topk(torch.tensor(xs[:, None].copy()), torch.ones(1, 1), 1, return_values=True, arithmetic='half2')
    got = vals.cpu().numpy()[:, 0]
    assert np.all(np.abs(got - xs) <= 2.0 ** -21 * np.abs(xs))


@pytest.mark.parametrize('arith', ['half2', 'split'])
@pytest.mark.parametrize('r', [129, 200, 256])
def test_wide_tables(ops, r, arith):
    """Both plane forms reach width 256 (fp16: 128 A registers per lane; bf16 x 3 since round 5: 192, eight waves per workgroup at two
    per SIMD): exact on integer factors, fp32-accurate on Gaussian ones, with clamping and both workgroup shapes (k = 10: 4 waves,
    k = 30: 8 waves)."""
    rng = np.random.default_rng(r)
    m, n = 300, 33001
    for k, clamp in ((10, False), (30, True)):
        U = rng.integers(-2, 3, (m, r)).astype(np.float32)
        V = rng.integers(-2, 3, (n, r)).astype(np.float32)
        sc = U @ V.T
        vals, got = ops.predict_topk(torch.tensor(U), torch.tensor(V), k, clamp_negatives=clamp, return_values=True, arithmetic=arith)
        ref = expected(sc, k, clamp)
        assert np.array_equal(got.cpu().numpy(), ref), (r, k, arith)
        assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(np.where(sc > 0, sc, 0) if clamp else sc, ref, 1))
    k = 10
    g = torch.Generator().manual_seed(r)
    Ug, Vg = torch.randn(m, r, generator=g) * 0.05, torch.randn(n, r, generator=g) * 0.05
    want = Ug.double() @ Vg.double().T
    vh, ih = ops.predict_topk(Ug, Vg, k, return_values=True, arithmetic=arith)
    v32, i32 = ops.predict_topk(Ug, Vg, k, return_values=True, arithmetic='fp32')
    norm = float(want.abs().max())
    eh = float((vh.cpu().double() - torch.gather(want, 1, ih.cpu().long())).abs().max()) / norm
    e32 = float((v32.cpu().double() - torch.gather(want, 1, i32.cpu().long())).abs().max()) / norm
    assert eh < 1.5e-6 and eh <= 2 * e32 + 2e-7, (eh, e32)
    lib = __import__('teamoflow_amd._lib', fromlist=['get']).get()
    assert lib.tmf_predict_topk_half2_supported(256, 32) == 1 and lib.tmf_predict_topk_half2_supported(257, 1) == 0
    assert lib.tmf_predict_topk_split_supported(256, 32) == 1 and lib.tmf_predict_topk_split_supported(257, 1) == 0


def test_half2_range_guard(ops):
    """'auto' takes the fp16 planes only while the item rows span <= 2^12 in magnitude (one scale for the whole table)."""
    V = torch.randn(1000, 16)
    assert ops.half2_range_ok(V.cuda())
    V[3] *= 1e6
    assert not ops.half2_range_ok(V.cuda())
    V[3] = 0
    assert ops.half2_range_ok(V.cuda()) and ops.half2_range_ok(torch.zeros(4, 4, device='cuda'))


@pytest.mark.parametrize('arith', ['split', 'half2'])
def test_deferred_merges_and_overflow(ops, arith):
    """Records every 20..41 items (pending buffers fill over several tiles, rows merge at different times, the last records sit
    in the last tiles), and scores increasing with the item index (every tile overflows the pending buffer)."""
    n, r, m = 3001, 8, 300
    period = 20 + 3 * np.arange(r)
    j = np.arange(n)
    V = np.where(j[:, None] % period[None, :] == 0, (j // 16 + 1)[:, None], -1).astype(np.float32)
    U = np.eye(r, dtype=np.float32)[np.arange(m) % r]
    U[7] = 0
    sc = U @ V.T
    for k in (1, 10, 32) + ((33, 40) if arith == 'split' else ()):
        for clamp in (False, True):
            vals, got = ops.predict_topk(torch.tensor(U), torch.tensor(V), k, clamp_negatives=clamp, return_values=True, arithmetic=arith)
            ref = expected(sc, k, clamp)
            assert np.array_equal(got.cpu().numpy(), ref), (k, clamp)
    inc = torch.arange(40000, dtype=torch.float32)[:, None] * torch.ones(1, 4)      # long enough for the warm-up pass
    got = ops.predict_topk(torch.ones(7, 4), inc, 5, arithmetic=arith).cpu().tolist()
    assert got == [[39999, 39998, 39997, 39996, 39995]] * 7
    Z = torch.zeros(200, 16)                                                         # everything clamped to 0: 0..k-1
    assert ops.predict_topk(Z, torch.ones(40000, 16), 10, clamp_negatives=True, arithmetic=arith).cpu().tolist() == [list(range(10))] * 200
    dec = -inc                                                                       # best items first: the warm-up bound is tight
    assert ops.predict_topk(torch.ones(3, 4), dec, 4, arithmetic=arith).cpu().tolist() == [[0, 1, 2, 3]] * 3


def test_limits_and_errors(ops):
    from teamoflow_amd import _lib
    lib = _lib.get()
    assert lib.tmf_predict_topk_split_supported(256, 32) == 1 and lib.tmf_predict_topk_split_supported(257, 10) == 0
    assert lib.tmf_predict_topk_split_supported(64, 40) == 1 and lib.tmf_predict_topk_split_supported(64, 41) == 0
    assert lib.tmf_predict_topk_half2_supported(64, 33) == 0 and lib.tmf_predict_topk_split_workspace_bytes(1000, 257) == 0
    assert lib.tmf_predict_topk_split_workspace_bytes(1000, 129) == 3 * 1024 * 256 * 2
    assert lib.tmf_predict_topk_split_workspace_bytes(1000, 100) == 3 * 1024 * 128 * 2
    with pytest.raises(ValueError):
        ops.predict_topk(torch.ones(4, 300), torch.ones(9, 300), 2, arithmetic='split')
    with pytest.raises(ValueError):
        ops.predict_topk(torch.ones(4, 8), torch.ones(99, 8), 41, arithmetic='split')
    with pytest.raises(ValueError):
        ops.predict_topk(torch.ones(4, 8), torch.ones(99, 8), 33, arithmetic='half2')
    x = torch.ones(8, 8, device='cuda')
    out = torch.empty(8, 2, dtype=torch.int32, device='cuda')
    ws = torch.empty(64, dtype=torch.uint8, device='cuda')
    rc = lib.tmf_predict_topk_split_f32(_lib.ptr(x), _lib.ptr(x), 8, 8, 8, 8, 8, 2, 0, _lib.ptr(out), None, _lib.ptr(ws), 64,
                                        _lib.stream_ptr())
    assert rc != 0 and b'workspace' in lib.tmf_last_error()
    big = torch.ones(4, 260, device='cuda')
    rc = lib.tmf_predict_topk_split_f32(_lib.ptr(big), _lib.ptr(big), 4, 4, 258, 260, 260, 2, 0, _lib.ptr(out), None, _lib.ptr(ws), 64,
                                        _lib.stream_ptr())
    assert rc != 0 and b'supports' in lib.tmf_last_error()
    assert lib.tmf_predict_topk_split_f32(None, None, 0, 5, 4, 4, 4, 1, 0, None, None, None, ctypes.c_size_t(0), _lib.stream_ptr()) == 0


def test_model_level_switch(ops):
    """retrieve_user_recs / recall_at_k through the class surface with the split arithmetic selected: same lists as the fp32 kernel
    on a trained-table-like input (matrix_factorization.py:424-438)."""
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    g = torch.Generator().manual_seed(5)
    U, V = torch.randn(500, 64, generator=g) * 0.1, torch.randn(20000, 64, generator=g) * 0.1
    model = MatrixFactorization(64)
    model.user_embedding, model.item_embedding = U.cuda(), V.cuda()
    model.predict_arithmetic = 'fp32'
    a = model.retrieve_user_recs(k=10)
    model.predict_arithmetic = 'split'
    b = model.retrieve_user_recs(k=10)
    assert a.dtype == np.int32 and (a == b).all(1).mean() > 0.999


def test_auto_rule_at_catalog_scale(ops, monkeypatch):
    """arithmetic='auto' from 2^26 scores on: the fp16 planes while the item rows span <= 2^12 in magnitude, the bf16 planes beyond
    (one item row a million times larger than the rest) - either way the fp64 ranking on rows with clear gaps, and the explicit
    'half2' call on the wide-range table still inside the norm-wise bound."""
    g = torch.Generator().manual_seed(77)
    m, n, r, k = 2048, 33000, 64, 10
    assert m * n >= ops.SPLIT_MIN_SCORES
    U = (torch.randn(m, r, generator=g) * 0.05).cuda()
    V = (torch.randn(n, r, generator=g) * 0.05).cuda()
    for big_row in (False, True):
        if big_row:
            V = V.clone()
            V[123] *= 1e6
        assert ops.half2_range_ok(V) == (not big_row)
        ref = U.double() @ V.double().T
        wv, wi = torch.topk(ref, k + 1, dim=1)
        vals, idx = ops.predict_topk(U, V, k, return_values=True)            # auto
        norm = ref.abs().amax(1, keepdim=True)
        # gaps judged against the scores' own size: the kernel 'auto' picked resolves every factor to fp32 precision, so the
        # ranking among ordinary items holds even beside an item whose scores are a million times larger
        scale = wv[:, 1:].abs().amax(1)
        clear = ((wv[:, :-1] - wv[:, 1:]).min(1).values > 2e-5 * scale)
        assert clear.float().mean() > 0.5
        assert torch.equal(idx.long()[clear], wi[clear, :k])
        got = torch.gather(ref, 1, idx.long())
        ordinary = (idx.long() != 123) if big_row else torch.ones_like(idx, dtype=torch.bool)   # the big item's scores: norm-wise, below
        err = (((vals.double() - got).abs() / scale[:, None]) * ordinary).max()
        assert float(err) < 2e-6
        vh, ih = ops.predict_topk(U, V, k, return_values=True, arithmetic='half2')
        errh = (vh.double() - torch.gather(ref, 1, ih.long())).abs().max() / ref.abs().max()
        assert float(errh) < 2e-6
