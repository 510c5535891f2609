"""Local compute of the item-row-sharded fit (SURVEY.md §8e "when V no longer fits"; teamoflow_amd/dist.py holds the
exchange): the item table is never resident as a whole.  The catalog is cut into T windows of ``rows`` item rows; a pass
over the catalog is a loop over the windows, each call getting the window's rows ``Vwin [rows, ld]`` from whoever owns
them.  Everything numeric is the same libtmf.so kernels as the resident path, launched on one window at a time:

  WMRB (matrix_factorization.py:153-176 + loss_graphs.py:74-88): the loss couples all items of a user, so the catalog is
    walked twice - scores of the window's negatives / interactions (tmf_wmrb_scores3 on the window's slices), the hinge step
    once all scores exist (tmf_wmrb_hinge2), then per window the user-gradient partial (tmf_wmrb_gradu3, one [users, ld]
    layer summed in window order) and the item gradient of the window's rows (tmf_wsum_pass over the window's entry lists).
  MSE (loss_graphs.py:47-52): every interaction touches one item, so one walk - a sub-problem per window (its interactions,
    item ids relative to the window): user-gradient partials summed in window order, item gradient of the window's rows.

U is updated by the reference's fresh Adam when the last window is done; the item gradient of every window is handed
back raw (fp32) because the rows' owner has to sum it over ranks first."""
import ctypes

import torch

from . import _engine, _lib

WINDOW_SLICE_BYTES = 4 << 20   # V rows per slice inside a window (the resident path's slice size)


def window_geometry(n_items, n_windows, ld, elem_size, world=1):
    """(rows per window, slices per window, padded catalog rows): a window holds a whole number of ~4 MB slices and of the
    ``world`` ownership sub-blocks, the padded catalog a whole number of windows (rows >= n_items have no interactions and
    stay zero)."""
    import math
    rows = max(1, -(-n_items // n_windows))
    # at most 64 slices over the whole catalog, like the resident pass (_engine.default_item_slices): shorter (user, slice)
    # ranges than that are bound by the walk, not by where the rows come from (config-5 shard: 128 slices 288 ms, 64 255 ms)
    k = int(min(max(1, -(-rows * ld * elem_size // WINDOW_SLICE_BYTES)), max(1, 64 // n_windows)))
    step = k * world // math.gcd(k, world)
    rows = -(-rows // step) * step
    return rows, k, rows * n_windows


def pad_table(W, ld, dtype, dev, r):
    W = torch.as_tensor(W).detach().to(device=dev, dtype=torch.float32)
    out = torch.zeros(W.shape[0], ld, dtype=dtype, device=dev)
    out[:, :r] = W
    return out


class WindowedHipBackend:
    """One rank's users against a windowed catalog.  ``V_own`` are the item rows this rank owns (of every window its
    sub-block, window-major; updated in place by ``adam_rows``); they take part in the passes only through the ``Vwin``
    arguments.  ``world`` only enters the window geometry (a window must split into that many equal sub-blocks)."""

    def __init__(self, U_blk, V_own, indices, values, R, n_users, n_items, n_windows, n_components, loss, c, lr,
                 dtype=torch.float32, user_chunks=None, world=1):
        lib = _lib.get()
        dev = indices.device
        self.loss, self.c, self.r, self.T = loss, float(c), int(n_components), int(n_windows)
        self.two_phase = loss == 'wmrb'
        self.dtype = dtype
        self.sfx = '_bf16' if dtype is torch.bfloat16 else '_f32'
        self.ld = ld = _lib.padded_ld(self.r, dtype)
        esz = 2 if dtype is torch.bfloat16 else 4
        self.rows, self.k, self.n_pad = window_geometry(n_items, self.T, ld, esz, world)
        self.m = m = int(n_users)
        self.adam = lib.tmf_adam_fresh(float(lr))
        self.U = pad_table(U_blk, ld, dtype, dev, self.r)
        self.U_nxt = torch.empty_like(self.U)
        self.V_shard = V_own   # [rows owned, ld] table dtype, or None on a rank that owns nothing
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self.part = torch.empty(2 * max(m, 1), ld, dtype=torch.float32, device=dev)   # layer 0 sums, layer 1 = one window (MSE)
        if loss == 'wmrb':
            self.plan = _engine.InteractionPlan(indices, values, m, self.n_pad, csc=False)
            # user blocks as the resident pass would choose them for the whole catalog: more blocks would fit the slab of one
            # window but leave a handful of entries per (block, item) list
            C = user_chunks or _engine.default_user_chunks(m, ld, n_items=self.n_pad)
            self.wplan = w = _engine.WmrbPlan(self.plan, R, user_chunks=C, item_slices=self.T * self.k,
                                              n_components=self.r, sliced=True, item_lists=False)
            self.lists = [w.window_lists(self.plan, t * self.k, self.k, t * self.rows) for t in range(self.T)]
            blocks = torch.arange(C, device=dev, dtype=torch.int64)[:, None] * self.n_pad
            self.seg = []
            for t in range(self.T):
                items = torch.arange(t * self.rows, (t + 1) * self.rows, device=dev, dtype=torch.int64)
                rows = (blocks + items[None, :]).reshape(-1)          # user blocks outermost, like the resident item pass
                out_row = (items - t * self.rows).repeat(C)
                self.seg.append(_engine.SegmentTable.of_rows(w.rowptr_e, rows, out_row, self.rows))
            self.sp = torch.empty(m, w.S, dtype=torch.float32, device=dev)
            self.pk = torch.empty(max(self.plan.nnz, 1), dtype=torch.float32, device=dev)
            self.loss_part = torch.zeros(max(m, 1), dtype=torch.float32, device=dev)
            n_slab = max([s.n_slab for s in self.seg] + [1])
            self.n_loss = self.plan.n_pos
        else:
            item = indices[:, 1]
            self.sub = []
            for t in range(self.T):
                keep = (item >= t * self.rows) & (item < (t + 1) * self.rows)
                idx_t, val_t = _engine.take_interactions(indices, values, keep, item_offset=t * self.rows)
                self.sub.append(_engine.InteractionPlan(idx_t, val_t, m, self.rows, csc=True))
            n_slab = max([max(p.seg_u.n_slab, p.seg_i.n_slab) for p in self.sub] + [1])
            self.loss_part = torch.zeros(max([p.seg_u.nseg for p in self.sub] + [1]), dtype=torch.float32, device=dev)
            self.loss_w = torch.zeros(self.T, dtype=torch.float64, device=dev)
            self.n_loss = int(indices.shape[0])
        self.slab = torch.empty(n_slab, ld, dtype=torch.float32, device=dev)

    # ---- the interface ItemShardedEpoch drives ----
    def V_own(self):
        return self.V_shard

    def scores_window(self, t, Vwin):
        """WMRB walk 1: sp / p of the window's negatives and interactions."""
        lib = _lib.get()
        if self.m == 0:
            return
        _lib.check(getattr(lib, 'tmf_wmrb_scores3' + self.sfx)(ctypes.byref(self.lists[t]), _lib.ptr(self.U), _lib.ptr(Vwin),
                                                               _lib.ptr(self.sp), _lib.ptr(self.pk), self.r, _lib.stream_ptr()), lib)

    def between(self):
        """WMRB: hinge step over the complete scores -> delta, D, per-user loss."""
        lib, p, w = _lib.get(), self.plan, self.wplan
        i32 = ctypes.c_int32
        _lib.check(lib.tmf_wmrb_hinge2_ordered(_lib.ptr(p.rowptr_u), _lib.ptr(p.val_u), _lib.ptr(self.pk), _lib.ptr(self.sp), i32(self.m),
                                               i32(w.S), self.c, _lib.ptr(w.delta), _lib.ptr(w.D), _lib.ptr(self.loss_part),
                                               _lib.ptr(getattr(w, 'hinge_order', None)), _lib.stream_ptr()), lib)

    def grads_window(self, t, Vwin, out):
        """Adds the window's part of the user gradient to layer 0 of ``part`` and writes the raw item gradient of the
        window's rows (this rank's users only) into ``out`` [rows, ld] fp32."""
        lib, s, r = _lib.get(), _lib.stream_ptr(), self.r
        i32 = ctypes.c_int32
        if self.loss == 'wmrb':
            w = self.wplan
            if self.m:
                _lib.check(getattr(lib, 'tmf_wmrb_gradu3' + self.sfx)(ctypes.byref(self.lists[t]), _lib.ptr(w.D), _lib.ptr(w.delta),
                                                                      _lib.ptr(Vwin), _lib.ptr(self.part), 1 if t == 0 else 2, r, s), lib)
            seg = self.seg[t]
            _lib.check(getattr(lib, 'tmf_wsum_pass' + self.sfx)(seg.cstruct(), _lib.ptr(w.ent_row), _lib.ptr(w.ent_w), _lib.ptr(w.wbuf),
                                                                _lib.ptr(self.U), None, _lib.ptr(out), _lib.ptr(self.slab), r,
                                                                _lib.EPI_GRAD, self.adam, s), lib)
            _engine._row_pass_finish(lib, seg, self.slab, None, out, r, _lib.EPI_GRAD, self.adam, s, self.sfx)
            return
        p = self.sub[t]
        mse_pass = getattr(lib, 'tmf_mse_pass' + self.sfx)
        layer = self.part[:self.m] if t == 0 else self.part[self.m:2 * self.m]
        if self.m:
            _lib.check(mse_pass(p.seg_u.cstruct(), _lib.ptr(p.col_u), _lib.ptr(p.val_u), _lib.ptr(self.U), _lib.ptr(Vwin),
                                _lib.ptr(layer), _lib.ptr(self.slab), _lib.ptr(self.loss_part), r, _lib.EPI_GRAD, self.adam, s), lib)
            _engine._row_pass_finish(lib, p.seg_u, self.slab, self.U, layer, r, _lib.EPI_GRAD, self.adam, s, self.sfx)
            if t:   # layer 0 += layer 1 (the finish kernel sums layers row by row; reading and writing row u in one thread)
                _lib.check(getattr(lib, 'tmf_wmrb_finish' + self.sfx)(_lib.ptr(self.part), i32(2), i32(self.m), None,
                                                                      _lib.ptr(self.part), r, _lib.EPI_GRAD, self.adam, s), lib)
        _lib.check(lib.tmf_sum_f32(_lib.ptr(self.loss_part), p.seg_u.nseg, _lib.ptr(self.loss_w[t:t + 1]), s), lib)
        _lib.check(mse_pass(p.seg_i.cstruct(), _lib.ptr(p.row_i), _lib.ptr(p.val_i), _lib.ptr(Vwin), _lib.ptr(self.U),
                            _lib.ptr(out), _lib.ptr(self.slab), None, r, _lib.EPI_GRAD, self.adam, s), lib)
        _engine._row_pass_finish(lib, p.seg_i, self.slab, Vwin, out, r, _lib.EPI_GRAD, self.adam, s, self.sfx)

    def finish_users(self):
        """U <- fresh-Adam(U, summed user gradient); returns this rank's loss sum (1-element fp64 device tensor)."""
        lib, s = _lib.get(), _lib.stream_ptr()
        i32 = ctypes.c_int32
        if self.m:
            _lib.check(getattr(lib, 'tmf_wmrb_finish' + self.sfx)(_lib.ptr(self.part), i32(1), i32(self.m), _lib.ptr(self.U),
                                                                  _lib.ptr(self.U_nxt), self.r, _lib.EPI_ADAM, self.adam, s), lib)
        if self.loss == 'wmrb':
            _lib.check(lib.tmf_sum_f32(_lib.ptr(self.loss_part), self.m, _lib.ptr(self.loss_out), s), lib)
        else:
            self.loss_out[0] = self.loss_w.sum()
        self.U, self.U_nxt = self.U_nxt, self.U
        return self.loss_out

    def adam_rows(self, W_rows, G_rows):
        lib = _lib.get()
        if W_rows.shape[0]:
            _lib.check(getattr(lib, 'tmf_adam_fresh_rows' + self.sfx)(_lib.ptr(W_rows), _lib.ptr(G_rows), W_rows.shape[0], self.r,
                                                                      self.adam, _lib.stream_ptr()), lib)
