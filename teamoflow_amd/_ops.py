"""Tensor-level wrappers over the predict / ranking entry points of libtmf.so."""
import os

import torch

from . import _lib


def _cuda(t, dtype=None):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    _lib.get()  # raises EngineUnavailable without a GPU: there is no CPU path
    if not t.is_cuda:
        t = t.cuda()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def _gemm_operand(t):
    """[rows, r] fp32 tensor -> (tensor kept alive, rows, r, ld) with 16-byte aligned rows."""
    t = _cuda(t, torch.float32).detach()
    rows, r = t.shape
    ok = t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= r and t.data_ptr() % 16 == 0
    if not ok:
        ld = (r + 3) // 4 * 4
        p = torch.zeros(rows, ld, dtype=torch.float32, device=t.device)
        p[:, :r] = t
        t = p[:, :r]
    return t, rows, r, t.stride(0)


def predict_gemm(user_embedding, item_embedding, out=None):
    """user_embedding [m, r] @ item_embedding [n, r]^T -> [m, n] fp32 (exact-fp32 MFMA)."""
    lib = _lib.get()
    A, m, r, lda = _gemm_operand(user_embedding)
    B, n, rb, ldb = _gemm_operand(item_embedding)
    if r != rb:
        raise ValueError(f'embedding widths differ: {r} vs {rb}')
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=A.device)
    _lib.check(lib.tmf_predict_gemm_f32(_lib.ptr(A), _lib.ptr(B), _lib.ptr(out), m, n, r, lda, ldb, out.stride(0),
                                        _lib.stream_ptr()), lib)
    return out


FUSED_MAX_K, FUSED_MAX_K_BF16, FUSED_MAX_R, FUSED_MAX_R_BF16 = 64, 32, 256, 256
SPLIT_MIN_SCORES = 1 << 26   # arithmetic='auto' takes the three-plane bf16 split from this many scores (m * n) on: below, the pass that
                             # splits the item table and its workspace are not worth it and the fp32 MFMA kernel answers
SORT_MAX_ELEMS = 1 << 29   # elements ranked per call of the wide-row path (2 GB of keys + 2 GB of ids, twice)


def _bf16_operand(t):
    """[rows, r] bf16 tensor -> (tensor, rows, r, ld) with 16-byte aligned rows (ld % 8 == 0)."""
    t = t.detach()
    rows, r = t.shape
    ok = t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.stride(0) >= r and t.data_ptr() % 16 == 0
    if not ok:
        ld = (r + 7) // 8 * 8
        p = torch.zeros(rows, ld, dtype=torch.bfloat16, device=t.device)
        p[:, :r] = t
        t = p[:, :r]
    return t, rows, r, t.stride(0)


def fused_topk_supported(user_embedding, item_embedding, k):
    """Whether predict_topk ranks these tables without materialising scores: k <= 64, width <= 256 (bf16 tables beyond the bf16 kernel's
    k <= 32 go through the fp32 fused kernel on exact fp32 copies: predict_topk)."""
    bf16 = user_embedding.dtype == torch.bfloat16 and item_embedding.dtype == torch.bfloat16
    return k <= FUSED_MAX_K and user_embedding.shape[1] <= (FUSED_MAX_R_BF16 if bf16 else FUSED_MAX_R)


BF16_UPCAST_WINDOWS = 8       # item windows of the same path when the fp32 copy of the whole item table does not fit
BF16_UPCAST_USERS = 1 << 18   # users per call when bf16 tables are ranked through the fp32 kernel (bounds the fp32 copy of their rows)


PREDICT_ARITHMETIC = os.environ.get('TMF_PREDICT_ARITHMETIC', 'auto')   # 'auto' | 'fp32' | 'split' | 'half2'


SPLIT_MAX_K = 40    # two 4-wave workgroups' lists fit a CU's LDS beside their pending buffers up to k = 40 (tmf_predict_split.hip)
SPLIT_MAX_R = 256   # the three-plane kernel: 96 A registers per lane at r = 128, 192 at r = 256 (eight waves per workgroup, two per SIMD)


def split_topk_supported(r, k):
    return 1 <= r <= SPLIT_MAX_R and 1 <= k <= SPLIT_MAX_K


def half2_topk_supported(r, k):
    return 1 <= r <= 256 and 1 <= k <= 32


HALF2_MAX_ROW_RANGE = 2.0 ** 12   # largest / smallest item-row magnitude up to which the opt-in two-plane fp16 kernel keeps 22 bits


def half2_range_ok(item_rows):
    """The fp16 planes of the item table share ONE power-of-two scale: rows whose largest magnitude is within 2^12 of the
    table's keep 22 bits of their dominant factors (the second plane stays in fp16's normal range); beyond that the three
    bf16 planes ('split': no range limit) are the safe choice.  One small reduction and a host read."""
    row_max = item_rows.abs().amax(dim=1)
    top = row_max.max()
    low = torch.where(row_max > 0, row_max, top).min()
    top, low = float(top), float(low)
    return top == 0.0 or (top < float('inf') and low > 0 and top / low <= HALF2_MAX_ROW_RANGE)


def _upcast_table(B):
    """Exact fp32 copy of a bf16 item table (twice its size; torch.OutOfMemoryError when it does not fit)."""
    return B.float()


def predict_topk(user_embedding, item_embedding, k, clamp_negatives=False, return_values=False, arithmetic=None):
    """Top-k item ids (int32) of user_embedding @ item_embedding^T per user, fused (no [m, n] matrix).
    fp32 tables: 'fp32' = fp32 MFMA (k <= 64, width <= 256, bit-equal to an fmaf chain); 'split' = the bf16 matrix cores with ALL
    24 significand bits of every factor (three exact bf16 planes per factor, six plane products each exact in the fp32
    accumulator; ~1.9x the rate of the fp32 kernel, errors against fp64 at or below its; width <= 256 - 1.55x the fp32 kernel at 256: 171 against 110 TF -, k <= 40).
    'auto' (the default) takes 'split' where it applies and the job has SPLIT_MIN_SCORES scores or more, else 'fp32' - both keep
    the reference's fp32 operands whole (tf.matmul on fp32, matrix_factorization.py:236-248, 424-438).
    'half2' is an OPT-IN approximation, never chosen by 'auto': two fp16 planes under power-of-two scales = 22 bits of every
    factor, three products, ~2.9x; one scale for the whole item table, so check half2_range_ok(item_embedding) first.
    Scores BEYOND the fp32 range: the fp32 kernel returns +-inf like tf.matmul; the plane kernels may form +inf - inf = NaN between
    plane products of opposite sign, and a NaN score is never ranked (the ids returned are those of the finite scores).
    bf16 tables (both operands): bf16 MFMA with fp32 accumulation, k <= 32, width <= 256; 32 < k <= 64: the fp32 kernel on exact fp32 copies.
    See topk_stable(predict_gemm(...)) for the general case."""
    lib = _lib.get()
    arithmetic = arithmetic or PREDICT_ARITHMETIC
    if arithmetic not in ('auto', 'fp32', 'split', 'half2'):
        raise ValueError(f"arithmetic={arithmetic!r}: expected 'auto', 'fp32', 'split' or 'half2'")
    if torch.is_tensor(user_embedding) and torch.is_tensor(item_embedding) and \
            user_embedding.dtype == torch.bfloat16 and item_embedding.dtype == torch.bfloat16:
        A, m, r, lda = _bf16_operand(_cuda(user_embedding))
        B, n, rb, ldb = _bf16_operand(_cuda(item_embedding))
        if r != rb:
            raise ValueError(f'embedding widths differ: {r} vs {rb}')
        k = int(k)
        if not 1 <= k <= n:
            raise ValueError(f'k={k} must be in [1, {n}]')
        if FUSED_MAX_K_BF16 < k <= FUSED_MAX_K:
            # The bf16 kernel keeps 256 users' lists in LDS: k <= 32.  Beyond it the fp32 fused kernel (k <= 64) ranks exact fp32
            # copies of the rows - a bf16 x bf16 product is exact in fp32 either way, the fp32 sums differ in order only.
            # Costs a transient fp32 copy of the item table (twice its size); when that does not fit the catalog is ranked in windows
            # of rows (a copy of one window at a time) and the per-window lists are merged by one stable top-k - windows in catalog
            # order, so equal scores keep ascending item order.
            if m == 0:
                idx = torch.empty(0, k, dtype=torch.int32, device=A.device)
                return (torch.empty(0, k, dtype=torch.float32, device=A.device), idx) if return_values else idx
            try:
                windows = [(0, _upcast_table(B))]
            except torch.OutOfMemoryError:
                step = max(k, -(-n // BF16_UPCAST_WINDOWS))
                windows = [(c0, None) for c0 in range(0, n, step)]
                if n - windows[-1][0] < k:   # a last window narrower than k joins the one before it
                    windows.pop()
            out_i, out_v = [], []
            for b in range(0, m, BF16_UPCAST_USERS):
                Au = A[b:b + BF16_UPCAST_USERS].float()
                cand_v, cand_i = [], []
                for w, (c0, Bf) in enumerate(windows):
                    c1 = windows[w + 1][0] if w + 1 < len(windows) else n
                    v_, i_ = predict_topk(Au, Bf if Bf is not None else B[c0:c1].float(), k, clamp_negatives=clamp_negatives,
                                          return_values=True, arithmetic='fp32')
                    cand_v.append(v_)
                    cand_i.append(i_ + c0)
                if len(windows) > 1:
                    cv, ci = torch.cat(cand_v, dim=1), torch.cat(cand_i, dim=1)
                    v_, pos = topk_stable(cv, k, return_values=True)
                    cand_v, cand_i = [v_], [torch.gather(ci, 1, pos.long())]
                out_v.append(cand_v[0])
                out_i.append(cand_i[0])
            idx = torch.cat(out_i) if len(out_i) > 1 else out_i[0]
            return ((torch.cat(out_v) if len(out_v) > 1 else out_v[0]), idx) if return_values else idx
        idx = torch.empty(m, k, dtype=torch.int32, device=A.device)
        vals = torch.empty(m, k, dtype=torch.float32, device=A.device) if return_values else None
        _lib.check(lib.tmf_predict_topk_bf16(_lib.ptr(A), _lib.ptr(B), m, n, r, lda, ldb, k, int(bool(clamp_negatives)),
                                             _lib.ptr(idx), _lib.ptr(vals), _lib.stream_ptr()), lib)
        return (vals, idx) if return_values else idx
    A, m, r, lda = _gemm_operand(user_embedding)
    B, n, rb, ldb = _gemm_operand(item_embedding)
    if r != rb:
        raise ValueError(f'embedding widths differ: {r} vs {rb}')
    k = int(k)
    if not 1 <= k <= n:
        raise ValueError(f'k={k} must be in [1, {n}]')
    idx = torch.empty(m, k, dtype=torch.int32, device=A.device)
    vals = torch.empty(m, k, dtype=torch.float32, device=A.device) if return_values else None
    if (arithmetic == 'split' and not split_topk_supported(r, k)) or (arithmetic == 'half2' and not half2_topk_supported(r, k)):
        raise ValueError(f"the plane kernels support widths <= 256 and k <= {SPLIT_MAX_K} ('split') / 32 ('half2') (got {r}, {k})")
    if arithmetic == 'auto':
        # 32 < k <= 40 on tables of width <= 32: the fp32 kernel is the faster one (262144 x 100000, r = 32, k = 40: 43.9 against 38.1 TF;
        # r = 64: 65 against 69, r = 96: 65 against 86, r = 128: 86 against 113, r = 256: 100 against 127)
        planes = split_topk_supported(r, k) and not (k > 32 and r <= 32)
        arithmetic = 'split' if m * n >= SPLIT_MIN_SCORES and planes else 'fp32'
        planes_optional = True    # 'auto' may fall back to the fp32 kernel (it needs no workspace) when memory is short
    else:
        planes_optional = False
    if arithmetic in ('half2', 'split'):
        size = lib.tmf_predict_topk_half2_workspace_bytes if arithmetic == 'half2' else lib.tmf_predict_topk_split_workspace_bytes
        run = lib.tmf_predict_topk_half2_f32 if arithmetic == 'half2' else lib.tmf_predict_topk_split_f32
        need = size(n, r)
        try:
            ws = torch.empty(need, dtype=torch.uint8, device=A.device)
        except torch.OutOfMemoryError:
            if not planes_optional:
                raise
            ws = None   # the planes of the item table (1.5x its size) do not fit: the fp32 MFMA kernel ranks without them
        if ws is not None:
            _lib.check(run(_lib.ptr(A), _lib.ptr(B), m, n, r, lda, ldb, k, int(bool(clamp_negatives)),
                           _lib.ptr(idx), _lib.ptr(vals), _lib.ptr(ws), need, _lib.stream_ptr()), lib)
            return (vals, idx) if return_values else idx
    _lib.check(lib.tmf_predict_topk_f32(_lib.ptr(A), _lib.ptr(B), m, n, r, lda, ldb, k, int(bool(clamp_negatives)),
                                        _lib.ptr(idx), _lib.ptr(vals), _lib.stream_ptr()), lib)
    return (vals, idx) if return_values else idx


def topk_stable(x, k, clamp_negatives=False, return_values=False):
    """Row-wise top-k indices (int32) ordered like tf.math.top_k: value desc, ties -> lower index."""
    lib = _lib.get()
    x = _cuda(x, torch.float32)
    squeeze = x.dim() == 1
    if squeeze:
        x = x[None, :]
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    k = int(k)
    if not 1 <= k <= cols:
        raise ValueError(f'k={k} must be in [1, {cols}]')  # tf.math.top_k raises for k > last dim
    idx = torch.empty(rows, k, dtype=torch.int32, device=x.device)
    vals = torch.empty(rows, k, dtype=torch.float32, device=x.device) if return_values else None
    # large k over wide rows goes through a segmented radix sort with a workspace: a bounded number of rows per call
    step = rows if lib.tmf_topk_workspace_bytes(1, cols, k) == 0 else max(1, SORT_MAX_ELEMS // cols)
    ws = None
    for b in range(0, rows, step):
        e = min(b + step, rows)
        need = lib.tmf_topk_workspace_bytes(e - b, cols, k)
        if need and (ws is None or ws.numel() < need):
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.tmf_topk_stable_f32(_lib.ptr(x[b:e]), e - b, cols, x.stride(0), k, int(bool(clamp_negatives)),
                                           _lib.ptr(idx[b:e]), _lib.ptr(vals[b:e]) if return_values else None, _lib.ptr(ws),
                                           ws.numel() if ws is not None else 0, _lib.stream_ptr()), lib)
    if squeeze:
        idx = idx[0]
        vals = vals[0] if return_values else None
    return (vals, idx) if return_values else idx


def gather_rows_cols(x, idx):
    """out[i, c] = x[i, idx[i, c]]."""
    lib = _lib.get()
    x = _cuda(x, torch.float32).contiguous()
    idx = _cuda(idx, torch.int64).contiguous()
    rows, cols = x.shape
    if idx.shape[0] != rows:
        raise ValueError('index_arr must have the same number of rows as input_arr')
    k = idx.shape[1]
    if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= cols):
        raise IndexError('column index out of range')
    out = torch.empty(rows, k, dtype=torch.float32, device=x.device)
    _lib.check(lib.tmf_gather_rows_cols_f32(_lib.ptr(x), _lib.ptr(idx), _lib.ptr(out), rows, cols, k,
                                            _lib.stream_ptr()), lib)
    return out
