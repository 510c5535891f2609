"""Host side of the sparse training path: index structures built once per ``fit`` and the per-epoch
launch sequence.  Everything numeric runs in libtmf.so (HIP); torch is used for device memory,
streams and the one-off index preparation (sort / bincount / cumsum).

What replaces what (reference paths under /root/reference/src/teamoflow/mf/):
  epoch_mse   <- matrix_factorization.py:130-176 with MSELoss  (loss_graphs.py:47-52)
  epoch_wmrb  <- matrix_factorization.py:130-176 with WMRBLoss (loss_graphs.py:74-88, utils.py:94-105)
"""
import ctypes
import os

import torch

from . import _lib

DEFAULT_CHUNK = 1024  # list entries per segment (heavy rows are cut into several segments)
# bytes of gradU slice partials kept as one layer per slice (one launch); above it a single layer is summed in place by
# per-slice launches (C4: 6.6 GB of layers, 0.8 ms faster than in place; the config-5 shard would need 82 GB and sums in place)
PART_BUDGET = int(os.environ.get('TMF_PART_BUDGET', 8 << 30))


def _excl_cumsum(x):
    out = torch.zeros(x.numel() + 1, dtype=torch.int64, device=x.device)
    torch.cumsum(x, 0, out=out[1:])
    return out


def take_interactions(indices, values, keep, user_offset=0, item_offset=0):
    """(indices[keep] - offsets, values[keep]) of a COO list, selected COLUMN BY COLUMN.  Row-indexing a [nnz, 2] int64 device
    tensor with a mask (or index_select) returns wrong rows beyond ~6e7 rows on this PyTorch-ROCm build
    (tools/torch_row_index_probe.py: wrong from 7e7 rows on, 1-D masked selects stay correct) - and C4 has 1e8."""
    u = indices[:, 0][keep] - user_offset
    j = indices[:, 1][keep] - item_offset
    return torch.stack([u, j], dim=1), values[keep]


def stable_order(keys, n_rows):
    """(perm, rowptr): stable order of int keys in [0, n_rows) and the row pointers of the sorted list.
    Device tensors go through libtmf (tmf_stable_order_i32: rocPRIM radix sort + binary-search row pointers);
    CPU tensors (host-logic tests) through torch."""
    n = keys.numel()
    if keys.is_cuda and n < 2 ** 31 and n_rows < 2 ** 31:
        lib = _lib.get()
        k32 = keys.to(torch.int32).contiguous()
        perm = torch.empty(n, dtype=torch.int64, device=keys.device)
        rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=keys.device)
        ws = torch.empty(lib.tmf_stable_order_workspace_bytes(n), dtype=torch.uint8, device=keys.device)
        _lib.check(lib.tmf_stable_order_i32(_lib.ptr(k32), n, n_rows, _lib.ptr(perm), None, _lib.ptr(rowptr), _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr()), lib)
        return perm, rowptr
    perm = torch.sort(keys, stable=True)[1]
    return perm, _excl_cumsum(torch.bincount(keys.to(torch.int64), minlength=n_rows))


class SegmentTable:
    """Rows of one side cut into segments of <= chunk entries (see include/tmf.h, tmf_segments).

    ``out_row`` (optional, int64 [rows]) maps every list row to the table row its partial sums belong
    to (several list rows per table row: the user-chunked WMRB item lists).  Then EVERY segment gets a
    slab slot, slots are numbered table-row-major, and all ``n_out`` table rows are finished by
    tmf_combine_rows."""

    def __init__(self, rowptr, chunk=DEFAULT_CHUNK, out_row=None, n_out=None, row_mod=0):
        dev = rowptr.device
        rows = rowptr.numel() - 1
        lens = rowptr[1:] - rowptr[:-1]
        nch = torch.clamp((lens + (chunk - 1)) // chunk, min=1)
        nseg = int(nch.sum())
        seg_first = _excl_cumsum(nch)
        ar = torch.arange(rows, device=dev)
        seg_row = torch.repeat_interleave(ar, nch, output_size=nseg)
        seg_chunk = torch.arange(nseg, device=dev) - seg_first[seg_row]
        if out_row is None:
            multi = nch > 1
            slab_beg = _excl_cumsum(nch * multi)
            seg_slab = torch.where(multi[seg_row], slab_beg[seg_row] + seg_chunk, torch.full_like(seg_chunk, -1))
            long_rows = torch.nonzero(multi).flatten()
            self.n_slab = int(slab_beg[-1])
            self.long_slab_beg = torch.cat([slab_beg[long_rows], slab_beg[-1:]]).contiguous()
        else:
            seg_out = out_row[seg_row]
            order = torch.sort(seg_out, stable=True)[1]  # table-row-major, list order kept inside a row
            seg_slab = torch.empty(nseg, dtype=torch.int64, device=dev)
            seg_slab[order] = torch.arange(nseg, device=dev)
            long_rows = torch.arange(n_out, device=dev)
            self.n_slab = nseg
            self.long_slab_beg = _excl_cumsum(torch.bincount(seg_out, minlength=n_out))
        self.rows, self.chunk, self.nseg, self.row_mod = rows, int(chunk), nseg, int(row_mod)
        self.rowptr = rowptr.contiguous()
        self.seg_row = seg_row.to(torch.int32)
        self.seg_chunk = seg_chunk.to(torch.int32)
        self.seg_slab = seg_slab.to(torch.int32)
        self.long_rows = long_rows.to(torch.int32)
        self.n_long = int(long_rows.numel())
        self._c = None

    @classmethod
    def of_rows(cls, rowptr, rows, out_row, n_out, chunk=DEFAULT_CHUNK):
        """Segments of a SUBSET of the list rows of ``rowptr`` (int64 ids ``rows``, any order), every one with a slab slot;
        ``out_row[i]`` in [0, n_out) is the output row the partial sums of list row ``rows[i]`` belong to.  One window of
        an item-row-sharded pass: the lists of the window's items, output rows relative to the window."""
        self = cls.__new__(cls)
        dev = rowptr.device
        lens = rowptr[rows + 1] - rowptr[rows]
        nch = torch.clamp((lens + (chunk - 1)) // chunk, min=1)
        nseg = int(nch.sum())
        seg_first = _excl_cumsum(nch)
        sel = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), nch, output_size=nseg)
        seg_chunk = torch.arange(nseg, device=dev) - seg_first[sel]
        seg_out = out_row[sel]
        order = torch.sort(seg_out, stable=True)[1]
        seg_slab = torch.empty(nseg, dtype=torch.int64, device=dev)
        seg_slab[order] = torch.arange(nseg, device=dev)
        self.n_slab = nseg
        self.long_slab_beg = _excl_cumsum(torch.bincount(seg_out, minlength=n_out))
        self.rows, self.chunk, self.nseg, self.row_mod = int(rows.numel()), int(chunk), nseg, 0
        self.rowptr = rowptr
        self.seg_row = rows[sel].to(torch.int32)
        self.seg_chunk = seg_chunk.to(torch.int32)
        self.seg_slab = seg_slab.to(torch.int32)
        self.long_rows = torch.arange(n_out, device=dev, dtype=torch.int32)
        self.n_long = int(n_out)
        self._c = None
        return self

    def cstruct(self):
        if self._c is None:
            self._c = _lib.Segments(self.rowptr.data_ptr(), self.seg_row.data_ptr(), self.seg_chunk.data_ptr(),
                                    self.seg_slab.data_ptr(), self.nseg, self.chunk, self.row_mod)
        return ctypes.byref(self._c)


class InteractionPlan:
    """CSR-by-user and CSC-by-item views of the interactions (values duplicated in both orders).  Inside a user the
    interactions are in ascending item order - the row-major order tf.sparse.SparseTensor is specified in - whatever
    order the caller's COO list has (duplicates keep their input order)."""

    def __init__(self, indices, values, n_users, n_items, chunk=DEFAULT_CHUNK, user_chunks=1, csc=True):
        dev = indices.device
        u = indices[:, 0].contiguous()
        j = indices[:, 1].contiguous()
        nnz = u.numel()
        if nnz:
            lo_u, hi_u, lo_j, hi_j = int(u.min()), int(u.max()), int(j.min()), int(j.max())
            if lo_u < 0 or hi_u >= n_users or lo_j < 0 or hi_j >= n_items:
                raise IndexError(f'interaction indices outside dense_shape: users [{lo_u}, {hi_u}] of {n_users}, '
                                 f'items [{lo_j}, {hi_j}] of {n_items}')
        values = values.to(torch.float32).contiguous()
        self.nnz, self.n_users, self.n_items = nnz, n_users, n_items
        if indices.is_cuda and nnz < 2 ** 31:
            # native: one stable sort on user * n_items + item, row pointers, column / value gather (tmf_csr_build)
            lib = _lib.get()
            idx64 = indices.to(torch.int64).contiguous()
            self.rowptr_u = torch.empty(n_users + 1, dtype=torch.int64, device=dev)
            self.col_u = torch.empty(nnz, dtype=torch.int32, device=dev)
            self.val_u = torch.empty(nnz, dtype=torch.float32, device=dev)
            self.user_of = torch.empty(nnz, dtype=torch.int32, device=dev)
            ws = torch.empty(lib.tmf_csr_build_workspace_bytes(nnz), dtype=torch.uint8, device=dev)
            _lib.check(lib.tmf_csr_build(_lib.ptr(idx64), _lib.ptr(values), nnz, n_users, n_items, _lib.ptr(self.rowptr_u),
                                         _lib.ptr(self.col_u), _lib.ptr(self.val_u), _lib.ptr(self.user_of), _lib.ptr(ws),
                                         ws.numel(), _lib.stream_ptr()), lib)
            del ws, idx64
        else:
            key = u.to(torch.int64) * n_items + j.to(torch.int64)
            if nnz > 1 and not bool((key[1:] >= key[:-1]).all()):
                perm = torch.sort(key, stable=True)[1]
                u, j, values = u[perm], j[perm], values[perm]
            self.rowptr_u = _excl_cumsum(torch.bincount(u, minlength=n_users))
            self.col_u = j.to(torch.int32)
            self.val_u = values.contiguous()
            self.user_of = u.to(torch.int32)
        self.seg_u = SegmentTable(self.rowptr_u, chunk)
        # CSC by item (only the MSE item pass reads it); with user_chunks = C > 1 by (user block, item), blocks
        # outermost, so that the U rows gathered at any time come from one cache-sized block of users
        C = max(1, int(user_chunks))
        self.user_chunks = C
        self.seg_i = self.rowptr_i = self.row_i = self.val_i = None
        if csc:
            if C > 1:
                upc = -(-n_users // C)
                key = (self.user_of.to(torch.int64) // upc) * n_items + self.col_u.to(torch.int64)
            else:
                key = self.col_u
            perm_c, self.rowptr_i = stable_order(key, C * n_items)
            self.row_i = self.user_of[perm_c].contiguous()
            self.val_i = self.val_u[perm_c].contiguous()
            if C > 1:
                out_row = torch.arange(C * n_items, device=dev) % n_items
                self.seg_i = SegmentTable(self.rowptr_i, chunk, out_row=out_row, n_out=n_items, row_mod=n_items)
            else:
                self.seg_i = SegmentTable(self.rowptr_i, chunk)
        self.n_pos = int((self.val_u > 0).sum())

    @property
    def user_ids(self):
        """int64 user of every CSR entry (tests, multi-GPU bookkeeping)."""
        return self.user_of.to(torch.int64)


def _slab_budget():
    """Bytes of per-(user block, item) partial rows the item pass may use: TMF_SLAB_BUDGET, else a quarter of the card's
    memory capped at 64 GB (config-5 shard, 1M items: 8 blocks 118 ms, 33 -> 106, 67 -> 90; C4 needs 8.3 GB either way)."""
    env = os.environ.get('TMF_SLAB_BUDGET')
    if env:
        return int(env)
    total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory if torch.cuda.is_available() else 32 << 30
    return int(min(64 << 30, total // 4))




def default_user_chunks(n_users, ld, target_bytes=3 << 20, n_items=None):
    """Number of user blocks the WMRB item lists are cut into so that the U rows the lists of one block
    gather (n_users / chunks * ld * 4 bytes, most of one XCD L2) stay cache-resident while that block is processed.
    Measured at C4 (1M users, 512-byte rows), item pass ms: 48 blocks 69.7, 64 -> 60.4, 96 -> 45.2, 123 -> 35.8,
    160 -> 33.8, 256 -> 33.7 (profiles/r01_user_chunk_sweep.txt; with non-temporal partial-row stores)."""
    env = os.environ.get('TMF_USER_CHUNKS')
    if env:
        return max(1, int(env))
    c = -(-n_users * ld * 4 // target_bytes)
    if n_items:  # every (block, item) list owns at least one fp32 partial row: keep that slab within budget
        c = min(c, max(1, _slab_budget() // (n_items * ld * 4)))
    return int(min(max(c, 1), 256)) if c > 1 else 1


def mse_user_chunks():
    """User blocks for the MSE item pass (CSC by (user block, item)).  Off by default: MSE lists are short (81M
    entries over 100K items at C4 against 1.1e9 for WMRB), so splitting them buys nothing - item pass 5.73 ms with
    1 block, 5.49 with 32, 8.57 with 123.  TMF_USER_CHUNKS forces a value (tests)."""
    env = os.environ.get('TMF_USER_CHUNKS')
    return max(1, int(env)) if env else 1


def default_item_slices(n_items, ld, n_samples=None, target_bytes=4 << 20, elem_size=4):
    """Number of item slices of the sliced WMRB user pass; 1 = the fused single-kernel pass.

    The sliced pass keeps every user's negatives sorted by item and walks the catalog in ~4 MB slices
    (slice-major grid) so that V rows are gathered from the XCD L2s instead of the Infinity Cache; the ids
    (and D) of a (user, slice) range are staged in LDS so a row gather depends on an LDS read only.
    At C4 (profiles/r01_sliced_user_pass.txt): scores 33.6 ms + hinge 25.5 ms + gradU 24.9 ms + finish = 86 ms
    against 134 ms for the fused kernel, which is bound by the Infinity-Cache gather rate.  Used when the V table
    is larger than two slices; small catalogs are L2-resident anyway and keep the fused kernel."""
    env = os.environ.get('TMF_ITEM_SLICES')
    if env:
        return max(1, int(env))
    # slices of ~4 MB of V rows in the table's OWN storage type (bf16 rows are half as long: C4 bf16 7 slices 65.4 ms, 13 68.0)
    if n_items * ld * elem_size <= 2 * target_bytes:
        return 1
    return int(min(-(-n_items * ld * elem_size // target_bytes), 64))


def slices_xcd_major(n_slices):
    """Block order of the slice kernels (tmf_slice_lists.xcd_major): every XCD its own slice (eight resident at a time) or all
    of them the same one.  TMF_SLICE_XCD = 0 | 1 forces it (the library reads the same variable)."""
    env = os.environ.get('TMF_SLICE_XCD')
    if env in ('0', '1'):
        return env == '1'
    return XCD_MAJOR_DEFAULT and n_slices >= 8


XCD_MAJOR_DEFAULT = False
SLICED_MIN_HINGE_TERMS = 1 << 16   # positives x negatives per user from which the O((S + P) log P) hinge step pays
SLICED_MIN_SCORES = 1 << 21        # n_users x n_samples below which the epoch is launch-bound and one fused kernel wins


def choose_wmrb_user_pass(n_users, n_items, ld, n_samples, n_positives, n_components, elem_size=4):
    """(item_slices, sliced) for a WMRB fit.  Large catalogs: the sliced pass with ~4 MB slices (default_item_slices).
    Small catalogs (V in the L2s, scores in LDS) default to the one-kernel pass, whose hinge step costs P_u x S terms per
    user - at the MovieLens-1M shape (C3: 6040 x 3706, S = 1853, 165 positives per user) that is 3e5 terms per user and
    the sliced pass with its O((S + P) log P) hinge kernel is faster although the catalog needs no slicing: 0.37 ms
    against 0.61 ms per user pass, item pass 0.20 against 0.26 (sorted negatives) - profiles/r02_hinge_rewrite.txt item 11.
    The slice count then only provides workgroups (the kernels also shrink their user groups, tmf_wmrb.hip)."""
    ns = default_item_slices(n_items, ld, elem_size=elem_size)
    if ns > 1 or not fused_user_pass_fits(n_samples, n_components) or os.environ.get('TMF_FORCE_SLICED') == '1':
        return ns, True
    if os.environ.get('TMF_ITEM_SLICES') or os.environ.get('TMF_FORCE_FUSED') == '1':
        return ns, False
    if (n_positives / max(n_users, 1)) * n_samples >= SLICED_MIN_HINGE_TERMS and n_users * n_samples >= SLICED_MIN_SCORES:
        groups = -(-n_users // 16)
        # slices: enough workgroups for a small user count, and no slice larger than one XCD L2 once the pass is sliced anyway
        return int(max(min(max(round(1400 / groups), 1), 8), -(-n_items * ld * elem_size // (4 << 20)))), True
    return 1, False


def rows4_wanted(n_components, dtype=torch.float32, plan=None, R=None, n_users=None, n_items=None):
    """Which form the WMRB item pass takes: False = lists -> partial rows -> combine (tmf_wsum_pass + tmf_combine_rows: every
    (user block, item) list writes one fp32 partial row into the slab), True = the ROW-STATIONARY form (lane groups own output rows,
    keep their sums in registers and walk the user blocks; no slab, no combine - tmf_wsum_rows5, balanced over virtual rows so that
    popular items are cut into parts, VirtualRows below).  Needs rows of at least 16 lanes.

    Default: row-stationary exactly when the slab form cannot block the users for the L2s - its slab (user blocks x items x fp32
    row) would exceed the budget at ~4 MB blocks, so the blocks grow (config-5 shard: 64 blocks of 10 MB, L2 hit rate 0.43, 128 GB
    of slab traffic per epoch; profiles/r05_c5_item_pass.txt).  Where the slab fits (C4: 163 blocks, 8.3 GB) the slab form stays:
    register-resident rows halve the waves per CU there (68 ms against 33, profiles/r03_c5_experiments.txt item 5).
    TMF_ROWS4 = 1 | 0 forces the choice."""
    env = os.environ.get('TMF_ROWS4')
    if env == '0':
        return False
    if not _lib.load_library().tmf_wsum_rows4_rows_per_group(int(n_components), int(dtype is torch.bfloat16)):
        return False
    if env == '1':
        return True
    if plan is not None:
        n_users = plan.n_users if n_users is None else n_users
        n_items = plan.n_items if n_items is None else n_items
    if not n_users or not n_items:
        return False
    ld = _lib.padded_ld(n_components, dtype)
    row_bytes = ld * (2 if dtype is torch.bfloat16 else 4)
    blocks_wanted = -(-int(n_users) * row_bytes // ROWS5_BLOCK_BYTES)          # user blocks of ~4 MB of rows as stored
    blocks_slab_allows = max(1, _slab_budget() // (int(n_items) * ld * 4))     # one fp32 partial row per (block, item)
    return blocks_wanted > 2 * blocks_slab_allows


ROWS5_TARGET_OVER_MEDIAN = 1.15
ROWS5_BLOCK_BYTES = 4 << 20   # user blocks of the row-stationary item pass (config-5 shard: 160 blocks of 4 MB 80.6 ms, 305 blocks 86.5)


def rows5_user_chunks(n_users, n_components, dtype=torch.float32):
    """User blocks of the row-stationary item pass: ~4 MB of U rows as stored, at most 256 (TMF_USER_CHUNKS overrides)."""
    env = os.environ.get('TMF_USER_CHUNKS')
    if env:
        return max(1, int(env))
    row_bytes = _lib.padded_ld(n_components, dtype) * (2 if dtype is torch.bfloat16 else 4)
    return int(min(max(1, -(-int(n_users) * row_bytes // ROWS5_BLOCK_BYTES)), 256))


class VirtualRows:
    """Work units of tmf_wsum_rows5 (include/tmf.h): output row i with cnt_i list entries (all user blocks together) is cut into
    P_i = ceil(cnt_i / target) parts, target = 1.15 x the median, at least the mean (TMF_ROWS5_TARGET overrides): the ordinary rows
    stay whole, the popular ones become parts of about the bulk's size, and a lane group's K consecutive virtual rows carry about K x mean entries
    whatever the popularity of its items.  Cut rows get consecutive slab slots (part order) and are finished by tmf_combine_rows."""

    def __init__(self, rowptr_e, n_blocks, n_rows, target=None):
        dev = rowptr_e.device
        lens = (rowptr_e[1:] - rowptr_e[:-1]).view(n_blocks, n_rows).sum(0)                       # entries of every output row
        mean = float(lens.to(torch.float64).mean()) if n_rows else 0.0
        median = float(lens.to(torch.float64).median()) if n_rows else 0.0
        env = os.environ.get('TMF_ROWS5_TARGET')
        # just above the bulk of the rows: the ordinary rows stay whole (a cut row costs a slab slot and a second pass over it), whatever
        # lies above is cut to about the bulk's size.  Config-5 shard (median 1,300, mean 1,405 entries per item), item pass ms by
        # target: 500: 103   700: 96   1000: 95   1200: 100 (every row cut in two)   1500: 81   2100: 93   4000: 170
        self.target = int(env) if env else (int(target) if target else max(1, int(max(ROWS5_TARGET_OVER_MEDIAN * median, mean)) + 1))
        parts = torch.clamp((lens + (self.target - 1)) // self.target, min=1)
        first = _excl_cumsum(parts)                                                               # first virtual row of every row
        nv = int(first[-1])
        item = torch.repeat_interleave(torch.arange(n_rows, device=dev), parts, output_size=nv)
        part = torch.arange(nv, device=dev) - first[item]
        multi = parts > 1
        slab_beg = _excl_cumsum(parts * multi)
        slot = torch.where(multi[item], slab_beg[item] + part, torch.full_like(part, -1))
        i32 = torch.int32
        self.n_vrows, self.n_slab = nv, int(slab_beg[-1])
        self.item = torch.cat([item, torch.tensor([n_rows], device=dev)]).to(i32).contiguous()   # + the sentinel (n_rows, 0, 1)
        self.part = torch.cat([part, torch.zeros(1, dtype=part.dtype, device=dev)]).to(i32).contiguous()
        self.nparts = torch.cat([parts[item], torch.ones(1, dtype=parts.dtype, device=dev)]).to(i32).contiguous()
        self.slot = slot.to(i32).contiguous()
        long_rows = torch.nonzero(multi).flatten()
        self.long_rows = long_rows.to(i32).contiguous()
        self.long_slab_beg = torch.cat([slab_beg[long_rows], slab_beg[-1:]]).contiguous()
        self.n_long = int(long_rows.numel())
        self.max_parts = int(parts.max()) if n_rows else 0


def fused_user_pass_fits(n_samples, n_components):
    """The one-kernel user pass keeps a user's scores and D in LDS; above ~13K negatives they do not fit and the sliced
    pass (no such limit) takes over whatever the catalog size."""
    return bool(_lib.load_library().tmf_wmrb_user_pass_fits(int(n_samples), min(int(n_components), 1024)))


HINGE_CHUNK = 255   # interactions the hinge kernel ranks per pass (csrc/tmf_hinge.hip: HCHUNK)


def hinge_user_order(rowptr_u):
    """Order in which the hinge kernel's waves take the users: by passes over their interactions (ceil(degree / 255)),
    most first, users of equal cost in their natural order (the bulk - one pass - keeps streaming its rows in sequence).
    None when no user needs a second pass.  Speed only (tmf_wmrb_hinge2_ordered)."""
    if os.environ.get('TMF_HINGE_ORDER', '1') == '0':
        return None
    deg = rowptr_u[1:] - rowptr_u[:-1]
    passes = (deg + (HINGE_CHUNK - 1)) // HINGE_CHUNK
    if passes.numel() == 0 or int(passes.max()) <= 1:
        return None
    return torch.argsort(passes, descending=True, stable=True).to(torch.int32)


class WmrbPlan:
    """Index structures of a WMRB fit, built once from the interactions and the static negative table.

    Item side: per-(user block, item) entry lists - the item's positives followed by the (user, sample) pairs whose
    negative is the item; list row = block * n_items + item, blocks of ceil(n_users / user_chunks) users outermost, so
    the waves running at any moment gather U rows of ONE cache-sized block of users.  Entry ids: k < nnz = interaction k
    of the CSR, nnz + u * S + s = negative s of user u; ``ent_row[e]`` = user of list entry e.
    Weights: ``wbuf = [delta (nnz) | D (m * S)]`` indexed by entry id; the user pass writes it and the item pass gathers
    through ``ent_w`` (entry id of every list entry).
    Sliced pass (``sliced``): ``R`` holds every user's negatives in ascending item order, ``slice_off`` / ``pos_off`` the first
    negative / interaction of every item slice."""

    def __init__(self, plan, R, chunk=DEFAULT_CHUNK, user_chunks=1, item_slices=1, n_components=128, sliced=None, item_lists=True,
                 xcd_major=None, rows4=False):
        dev = R.device
        m, S = R.shape
        nnz, n = plan.nnz, plan.n_items
        self.S, self.n_slices = S, max(1, int(item_slices))
        self.xcd_major = slices_xcd_major(self.n_slices) if xcd_major is None else bool(xcd_major)
        self.sliced = bool(sliced) if sliced is not None else (self.n_slices > 1 or not fused_user_pass_fits(S, n_components))
        C = self.user_chunks = max(1, int(user_chunks))
        E = nnz + m * S
        if E >= 2 ** 31:
            raise ValueError(f'interactions + n_users * n_samples = {E} >= 2^31 on one GPU: split the users over more GPUs')
        self.R_model = R  # the caller's table (model order): only D_in_model_order() reads it
        native = R.is_cuda
        lib = _lib.get() if native else None
        i32 = ctypes.c_int32
        self.slice_off = self.pos_off = None
        if self.sliced:
            # every user's negatives in ascending item order (the order of s is immaterial to the loss); D is
            # produced in this order - D_in_model_order() maps it back
            ns = self.n_slices
            if native:
                Rs = torch.empty_like(R)
                ws = torch.empty(lib.tmf_sort_samples_workspace_bytes(m, S), dtype=torch.uint8, device=dev)
                _lib.check(lib.tmf_sort_samples(_lib.ptr(R), i32(m), i32(S), i32(n), _lib.ptr(Rs), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr()), lib)
                del ws
                self.slice_off = torch.empty(m, ns + 1, dtype=torch.int32, device=dev)
                self.pos_off = torch.empty(m, ns + 1, dtype=torch.int32, device=dev)
                _lib.check(lib.tmf_slice_offsets(_lib.ptr(Rs), None, S, i32(m), i32(n), i32(ns), _lib.ptr(self.slice_off),
                                                 _lib.stream_ptr()), lib)
                _lib.check(lib.tmf_slice_offsets(_lib.ptr(plan.col_u), _lib.ptr(plan.rowptr_u), 0, i32(m), i32(n), i32(ns),
                                                 _lib.ptr(self.pos_off), _lib.stream_ptr()), lib)
            else:
                Rs = torch.sort(R.to(torch.int64), dim=1, stable=True)[0].to(torch.int32).contiguous()
                width = -(-n // ns)
                bnd = torch.arange(ns + 1, dtype=torch.int64) * width
                self.slice_off = torch.searchsorted(Rs.to(torch.int64), bnd[None, :].expand(m, -1).contiguous()).to(torch.int32)
                self.slice_off[:, -1] = S
                key = plan.user_of.to(torch.int64) * (ns * width + 1) + plan.col_u.to(torch.int64)  # ascending (CSR order)
                q = torch.arange(m, dtype=torch.int64)[:, None] * (ns * width + 1) + bnd[None, :]
                self.pos_off = (torch.searchsorted(key, q.reshape(-1)).reshape(m, ns + 1) - plan.rowptr_u[:-1, None]).to(torch.int32)
                self.pos_off[:, -1] = (plan.rowptr_u[1:] - plan.rowptr_u[:-1]).to(torch.int32)
            R = Rs
        self.R = R.contiguous()
        # ---- item-side entry lists ----
        self.ent_row = torch.empty(E, dtype=torch.int32, device=dev)
        rowptr = torch.empty(C * n + 2, dtype=torch.int64, device=dev)
        if native:
            ent_id = torch.empty(E, dtype=torch.int32, device=dev)
            ws = torch.empty(lib.tmf_wmrb_entry_lists_workspace_bytes(nnz, m, S), dtype=torch.uint8, device=dev)
            _lib.check(lib.tmf_wmrb_entry_lists(_lib.ptr(plan.user_of), _lib.ptr(plan.col_u), _lib.ptr(plan.val_u), nnz,
                                                _lib.ptr(self.R), i32(m), i32(S), i32(n), i32(C), _lib.ptr(self.ent_row),
                                                _lib.ptr(ent_id), _lib.ptr(rowptr), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr()), lib)
            del ws
        else:
            upc = max(1, -(-m // C))
            ku = plan.user_of.to(torch.int64)
            kp = torch.where(plan.val_u > 0, (ku // upc) * n + plan.col_u.to(torch.int64), torch.full_like(ku, C * n))
            ks = (torch.arange(m, dtype=torch.int64) // upc).repeat_interleave(S) * n + self.R.reshape(-1).to(torch.int64)
            keys = torch.cat([kp, ks])
            ent_id = torch.sort(keys, stable=True)[1]
            rowptr = _excl_cumsum(torch.bincount(keys, minlength=C * n + 1))
            owner = torch.cat([ku, torch.arange(m, dtype=torch.int64).repeat_interleave(S)])  # user of every entry id
            self.ent_row = owner[ent_id].to(torch.int32)
            ent_id = ent_id.to(torch.int32)
        self.rowptr_e = rowptr[:C * n + 1].contiguous()  # the row behind them holds the stored values <= 0: never read
        self.ent_w = ent_id
        self.rows4 = bool(rows4)
        # the row-stationary item pass walks VIRTUAL rows (popular items cut into parts); TMF_ROWS5=0: the plain form (tmf_wsum_rows4)
        self.vrows = VirtualRows(self.rowptr_e, C, n) if self.rows4 and os.environ.get('TMF_ROWS5', '1') != '0' else None
        if not item_lists or self.rows4:
            self.seg_e = None   # the caller cuts the lists into windows itself (SegmentTable.of_rows) / row-stationary item pass
        elif C > 1:
            out_row = torch.arange(C * n, device=dev) % n
            self.seg_e = SegmentTable(self.rowptr_e, chunk, out_row=out_row, n_out=n)
        else:
            self.seg_e = SegmentTable(self.rowptr_e, chunk)
        self.wbuf = torch.zeros(E, dtype=torch.float32, device=dev)
        self.delta = self.wbuf[:nnz]
        self.hinge_order = hinge_user_order(plan.rowptr_u)
        self.D = self.wbuf[nnz:].view(m, S)
        self._lists = None
        self.s5 = None   # Scores5Plan, built by the TrainState that wants the row-stationary scores kernel (scores5_wanted)

    def lists(self, plan):
        """tmf_slice_lists of the sliced pass (kept alive with the plan)."""
        if self._lists is None:
            m, S = self.R.shape
            self._lists = _lib.SliceLists(self.R.data_ptr(), self.slice_off.data_ptr(), plan.rowptr_u.data_ptr(),
                                          plan.col_u.data_ptr(), self.pos_off.data_ptr(), m, S, self.n_slices, 0, 0, 0,
                                          (_lib.SLICE_XCD_MAJOR if self.xcd_major else 0) | _lib.SLICE_N_ITEMS_STATED, int(plan.n_items))
        return ctypes.byref(self._lists)

    def window_lists(self, plan, slice_begin, slice_count, item_base):
        """tmf_slice_lists restricted to slices [slice_begin, +slice_count); the V pointer of the call then holds the rows
        from item_base on (item-row-sharded V).  The caller keeps the returned struct alive."""
        m, S = self.R.shape
        return _lib.SliceLists(self.R.data_ptr(), self.slice_off.data_ptr(), plan.rowptr_u.data_ptr(), plan.col_u.data_ptr(),
                               self.pos_off.data_ptr(), m, S, self.n_slices, slice_begin, slice_count, item_base,
                               _lib.SLICE_N_ITEMS_STATED, int(plan.n_items))

    def D_in_model_order(self):
        """D[u, s] indexed like the model's random_ind (the sliced pass keeps every user's negatives sorted by item)."""
        if not self.sliced:
            return self.D
        perm = torch.sort(self.R_model.to(torch.int64), dim=1, stable=True)[1]
        out = torch.empty_like(self.D)
        out.scatter_(1, perm, self.D)
        return out


def row_stationary_wanted(n_users, n_items, n_samples, nnz, n_components, dtype=torch.float32, sliced=True):
    """Whether gradU runs row-stationary (tmf_wmrb_gradu4: lane groups own users, keep their gradient rows in registers and walk
    all slices; no partial rows, no finish kernel).  TMF_ROW_STATIONARY = 1 | 0 forces it; default: where the visits are short
    (short_visits) and the kernel exists for the geometry."""
    env = os.environ.get('TMF_ROW_STATIONARY')
    if env == '0' or not sliced:
        return False
    if not _lib.load_library().tmf_wmrb_gradu4_supported(int(n_components), int(dtype is torch.bfloat16)):
        return False
    return env == '1' or short_visits(n_users, n_items, n_samples, nnz, n_components, dtype)


def wmrb_plan_for(plan, R, n_components, dtype=torch.float32, n_items=None):
    """The WmrbPlan a fit builds for (plan, R): the user-pass form (choose_wmrb_user_pass), the item-pass form (rows4_wanted) and the
    user blocks that go with it - ONE place for MatrixFactorization._fit_sparse, dist.fit_data_parallel and bench.py.
    n_items: the catalog size for the slice geometry when the plan's table is padded (multi-GPU: rows padded to the world size)."""
    m = plan.n_users
    n = plan.n_items if n_items is None else n_items
    bf16 = dtype is torch.bfloat16
    ld_store = _lib.padded_ld(n_components, dtype)
    ns, sliced = choose_wmrb_user_pass(m, n, ld_store, int(R.shape[1]), plan.n_pos, n_components, elem_size=2 if bf16 else 4)
    # short visits (config-5 shard): the row-stationary gradU with its own slice size, unless the environment decides
    # (TMF_ROW_STATIONARY = 0 | 1, TMF_ITEM_SLICES)
    rs = row_stationary_wanted(m, n, int(R.shape[1]), plan.nnz, n_components, dtype, sliced)
    if rs and not os.environ.get('TMF_ITEM_SLICES'):
        ns = int(max(1, -(-n * ld_store * (2 if bf16 else 4) // GRADU4_SLICE_BYTES)))
    rows4 = rows4_wanted(n_components, dtype, plan, R)
    C = rows5_user_chunks(m, n_components, dtype) if rows4 else default_user_chunks(m, _lib.padded_ld(n_components), n_items=plan.n_items)
    return WmrbPlan(plan, R, user_chunks=C, item_slices=ns, n_components=n_components, sliced=sliced, rows4=rows4)


def scores6_wanted(plan, wplan, n_components, dtype=torch.float32):
    """Whether the sliced user pass computes its scores with flat streams on the slice-major grid (tmf_wmrb_scores6, Scores6Plan)
    instead of per-(user, slice) visits (tmf_wmrb_scores3).  TMF_SCORES6 = 1 | 0 forces the choice; see SCORES6_MAX_VISIT for the
    default.  Needs rows of 32 lanes, fewer than 2^24 items, a table below 4 GB and int32 places (m S, nnz < 2^31)."""
    env = os.environ.get('TMF_SCORES6')
    if env == '0' or wplan is None or not wplan.sliced or not plan.col_u.is_cuda or os.environ.get('TMF_SCORES5') == '1':
        return False
    m, S = wplan.R.shape
    if m * S >= 2 ** 31 or plan.nnz >= 2 ** 31:
        return False
    if not _lib.load_library().tmf_wmrb_scores6_supported(int(n_components), int(dtype is torch.bfloat16), int(plan.n_items)):
        return False
    if env == '1':
        return True
    return SCORES6_DEFAULT and short_visits(m, plan.n_items, S, plan.nnz, n_components, dtype)


def short_visits(n_users, n_items, n_samples, nnz, n_components, dtype=torch.float32):
    """Whether a (user, ~4 MB slice of V) visit of the sliced user pass is only a handful of rows - S x slice bytes / table bytes
    (+ the user's interactions in the slice): 9 at the config-5 shard (1M items x 512 bytes), 88 at C4.  Short visits are what
    tmf_wmrb_scores3 / gradu3 pay their per-visit round trips, their row of U and their partial row for; such shapes take the flat
    streams (tmf_wmrb_scores6) and the row-stationary gradU (tmf_wmrb_gradu4) instead."""
    row_bytes = _lib.padded_ld(n_components, dtype) * (2 if dtype is torch.bfloat16 else 4)
    table = n_items * row_bytes
    if table < 8 * VISIT_SLICE_BYTES:   # short because the catalog is cut fine, not because a user has few negatives (tiny S)
        return False
    return (n_samples + nnz / max(n_users, 1)) * VISIT_SLICE_BYTES / table <= SCORES6_MAX_VISIT


SCORES6_SLICE_BYTES = 6 << 20   # slices of the flat-stream scores kernel, whatever the slice count of the other kernels.  Config-5 shard, scores ms
                                # by slice size: 2 MB 70.3   3 MB 59.6   4 MB 55.4   5 MB 53.9   6 MB 52.9   7 MB 53.8   8 MB 56.9 (scores3, 8 MB: 59.5)
VISIT_SLICE_BYTES = 4 << 20     # the slice size a visit's length is judged at (short_visits)
GRADU4_SLICE_BYTES = 3200000    # 3.2 MB slices for the row-stationary gradU: config-5 shard 160 slices 74.8 ms (128: 77.4, 192: 75.8, 256: 82.2;
                                # gradu3 + finish at 64 slices: 82.3)
SCORES6_MAX_VISIT = 24          # rows of a (user, 4 MB slice) visit up to which visits are too short for scores3 (config-5 shard: 9; C4: 86)
SCORES6_DEFAULT = True          # config-5 shard: 55.4 ms against 59.5 for scores3, fabric traffic 183 GB against 446 (profiles/r05_c5_shard.txt)


class Scores6Plan:
    """Entry stream of tmf_wmrb_scores6 (include/tmf.h), built once per fit from the sliced plan: every (user, item) pair whose
    score the epoch needs - interaction k of the CSR (score -> p[k]) and negative (u, pos) of the item-sorted table (score ->
    sp[u, pos]) - keyed by chunk = slice * n_groups + user group and put in that order by ONE stable radix sort: inside a chunk
    the interactions come first, then the negatives, each by user and item (so the scores of a user's visit are neighbours in the
    stream AND in sp / p).  Every chunk is padded to whole steps of 8 entries with its last id and the PAD place."""

    PAD = -2 ** 31

    def __init__(self, plan, wplan, n_components, dtype=torch.float32, slice_bytes=None):
        lib = _lib.load_library()   # the builder itself needs no GPU (tests/test_host_cpu.py runs it on CPU tensors)
        dev = plan.col_u.device
        self.key = (int(n_components), dtype)
        UG = int(lib.tmf_wmrb_scores6_users_per_group())
        m, S = wplan.R.shape
        nnz, n = plan.nnz, plan.n_items
        E = nnz + m * S
        self.n_groups = ng = -(-m // UG)
        row_bytes = _lib.padded_ld(n_components, dtype) * (2 if dtype is torch.bfloat16 else 4)
        slice_bytes = int(os.environ.get('TMF_S6_SLICE_BYTES', slice_bytes or SCORES6_SLICE_BYTES))
        ns = int(min(max(1, -(-n * row_bytes // slice_bytes)), max(1, (2 ** 31 - 1) // max(ng, 1))))
        width = -(-n // ns)
        self.n_slices = ns = -(-n // width)
        i32 = torch.int32
        uo = plan.user_of                                                  # int32 [nnz]
        rows = torch.arange(m, device=dev, dtype=i32)
        keys = torch.empty(E, dtype=i32, device=dev)
        keys[:nnz] = (plan.col_u // width) * ng + uo // UG
        keys[nnz:].view(m, S).copy_((wplan.R // width) * ng + (rows // UG)[:, None])
        perm, rowptr = stable_order(keys, ns * ng)
        del keys
        packed = torch.empty(E, dtype=i32, device=dev)
        packed[:nnz] = ((uo % UG) << 24) | plan.col_u
        packed[nnz:].view(m, S).copy_(((rows % UG) << 24)[:, None] | wplan.R)
        ids_sorted = packed[perm]
        del packed
        outs = torch.empty(E, dtype=i32, device=dev)
        outs[:nnz] = -1 - torch.arange(nnz, device=dev, dtype=i32)         # ~k: the score of interaction k goes to p[k]
        outs[nnz:] = torch.arange(m * S, device=dev, dtype=i32)            # u * S + pos: to sp[u, pos]
        outs_sorted = outs[perm]
        del outs, perm
        cnt = rowptr[1:] - rowptr[:-1]
        ptr8 = _excl_cumsum((cnt + 7) // 8 * 8)
        E8 = int(ptr8[-1])
        self.ids = torch.empty(E8 + 8, dtype=i32, device=dev)
        self.outs = torch.full((E8 + 8,), self.PAD, dtype=i32, device=dev)
        self.ids[E8:] = 0
        step = 1 << 27
        for q0 in range(0, E8, step):
            q = torch.arange(q0, min(q0 + step, E8), device=dev, dtype=torch.int64)
            ch = torch.searchsorted(ptr8, q, right=True) - 1
            r = q - ptr8[ch]
            c = cnt[ch]
            src = rowptr[ch] + torch.minimum(r, c - 1)
            self.ids[q0:q0 + q.numel()] = ids_sorted[src]
            self.outs[q0:q0 + q.numel()] = torch.where(r < c, outs_sorted[src], torch.full_like(src, self.PAD).to(i32))
            del q, ch, r, c, src
        self.chunk_ptr = ptr8
        self.n_entries, self.n_padded = E, E8


def scores5_wanted(plan, wplan, n_components, dtype=torch.float32):
    """Whether the sliced user pass computes its scores with the row-stationary kernel (tmf_wmrb_scores5: workgroups own 256
    users, keep their rows in LDS and walk one flat stream of (user, item) pairs ordered by item) instead of tmf_wmrb_scores3.
    OPT-IN (TMF_SCORES5=1), never the default: built in round 4 for the config-5 shard (1M items x 512 bytes, 9 rows per
    (user, slice) visit of scores3) and measured there at 68 - 124 ms against 63 ms for scores3, whatever the pacing
    (profiles/r04_scores5_experiment.txt).  Needs rows of 32 lanes, fewer than 2^24 items and a table below 4 GB."""
    if os.environ.get('TMF_SCORES5') != '1' or wplan is None or not wplan.sliced or not plan.col_u.is_cuda:
        return False
    m, S = wplan.R.shape
    if m * S >= 2 ** 31 or plan.nnz >= 2 ** 31:   # the stream's int32 places (sp index / ~p index): such shapes stay with scores3
        return False
    return bool(_lib.load_library().tmf_wmrb_scores5_supported(int(n_components), int(dtype is torch.bfloat16), int(plan.n_items)))


class Scores5Plan:
    """Entry streams of tmf_wmrb_scores5 (include/tmf.h), built once per fit from the sliced plan: every (user, item) pair whose
    score the epoch needs - interaction k of the CSR (score -> p[k]) and negative (u, pos) of the item-sorted table (score ->
    sp[u, pos]) - keyed by (workgroup = u // 256, slice = item // width) and put in that order by ONE stable radix sort
    (stable_order): inside a (workgroup, slice) chunk the interactions come first, then the negatives, each by user and item.
    The slices only order the stream (the kernel never sees them), so they are fine: ~512 KB of V rows each."""

    PAD = -2 ** 31

    def __init__(self, plan, wplan, n_components, dtype=torch.float32, slice_bytes=None):
        lib = _lib.get()
        dev = plan.col_u.device
        self.key = (int(n_components), dtype)   # what the streams were built for (the slice width follows the row bytes)
        UB = int(lib.tmf_wmrb_scores5_users_per_workgroup())
        m, S = wplan.R.shape
        nnz, n = plan.nnz, plan.n_items
        E = nnz + m * S
        self.n_wg = n_wg = -(-m // UB)
        row_bytes = _lib.padded_ld(n_components, dtype) * (2 if dtype is torch.bfloat16 else 4)
        slice_bytes = int(os.environ.get('TMF_S5_SLICE_BYTES', slice_bytes or (512 << 10)))
        ns = int(min(max(1, -(-n * row_bytes // slice_bytes)), 4096, max(1, (2 ** 31 - 1) // max(n_wg, 1))))
        width = -(-n // ns)
        self.n_slices = ns
        i32 = torch.int32
        uo = plan.user_of                                                  # int32 [nnz]
        rows = torch.arange(m, device=dev, dtype=i32)
        keys = torch.empty(E, dtype=i32, device=dev)
        keys[:nnz] = (uo // UB) * ns + plan.col_u // width
        keys[nnz:].view(m, S).copy_(((rows // UB) * ns)[:, None] + wplan.R // width)
        perm, rowptr = stable_order(keys, n_wg * ns)
        del keys
        packed = torch.empty(E, dtype=i32, device=dev)
        packed[:nnz] = ((uo % UB) << 24) | plan.col_u
        packed[nnz:].view(m, S).copy_(((rows % UB) << 24)[:, None] | wplan.R)
        ids_sorted = packed[perm]
        del packed
        outs = torch.empty(E, dtype=i32, device=dev)
        outs[:nnz] = -1 - torch.arange(nnz, device=dev, dtype=i32)         # ~k: the score of interaction k goes to p[k]
        outs[nnz:] = torch.arange(m * S, device=dev, dtype=i32)            # u * S + pos: to sp[u, pos]
        outs_sorted = outs[perm]
        del outs, perm
        # every workgroup's stream padded to whole steps of 8 with its last entry's id (a valid user, a resident row) and PAD
        wg_ptr = rowptr[::ns].contiguous()                                  # [n_wg + 1]
        cnt = wg_ptr[1:] - wg_ptr[:-1]
        wg_ptr8 = _excl_cumsum((cnt + 7) // 8 * 8)
        E8 = int(wg_ptr8[-1])
        self.ids = torch.empty(E8 + 8, dtype=i32, device=dev)
        self.outs = torch.full((E8 + 8,), self.PAD, dtype=i32, device=dev)
        self.ids[E8:] = 0
        step = 1 << 27
        for q0 in range(0, E8, step):
            q = torch.arange(q0, min(q0 + step, E8), device=dev, dtype=torch.int64)
            wg = torch.searchsorted(wg_ptr8, q, right=True) - 1
            r = q - wg_ptr8[wg]
            c = cnt[wg]
            src = wg_ptr[wg] + torch.minimum(r, c - 1)
            self.ids[q0:q0 + q.numel()] = ids_sorted[src]
            self.outs[q0:q0 + q.numel()] = torch.where(r < c, outs_sorted[src], torch.full_like(src, self.PAD).to(i32))
            del q, wg, r, c, src
        self.wg_ptr = wg_ptr8
        self.n_entries, self.n_padded = E, E8
        # pacing: the workgroups of a launch meet every `pace_every` slices (tmf.h: window w may start when the peers have
        # completed window w - lag - 1; lag 0 = a barrier per window) and run freely in between - equal work per slice keeps them
        # within a slice or two of each other over such a stretch, a rendezvous per slice would cost more than a slice takes.
        # Window 0 is empty: passing it is the start line (every workgroup has its rows in LDS).  wstart = first step (8 entries)
        # of every window in every workgroup's stream; the padding sits behind a workgroup's last entry, so offsets inside a
        # workgroup are those of the sorted list
        # at most kS5MaxWindows (4096) windows: their first steps sit in LDS - a smaller request is widened, never an error
        every = max(1, int(os.environ.get('TMF_S5_PACE_EVERY', 32)), -(-ns // 4000))
        firsts = list(range(0, ns, every))
        off = rowptr[:n_wg * ns].view(n_wg, ns)[:, firsts] - wg_ptr[:-1, None]
        self.n_windows = nwin = len(firsts) + 1
        self.wstart = torch.zeros(n_wg, nwin + 1, dtype=i32, device=dev)
        self.wstart[:, 1:nwin] = ((off + 7) // 8).to(i32)
        self.wstart[:, nwin] = ((wg_ptr8[1:] - wg_ptr8[:-1]) // 8).to(i32)
        self.lag = int(os.environ.get('TMF_S5_LAG', 0))
        self.wgs_per_launch = int(os.environ.get('TMF_S5_WGS', 0))
        self.paced = os.environ.get('TMF_S5_PACE', '1') != '0' and ns > 1
        nb = lib.tmf_wmrb_scores5_workspace_bytes(n_wg, nwin, self.wgs_per_launch) if self.paced else 0
        self.sync = torch.zeros(max(nb // 4, 1), dtype=i32, device=dev)


class TrainState:
    """Double-buffered factor tables [rows, ld] and the scratch the passes need.
    ``V_tables`` = (V, V_nxt) already padded: several states (the user batches of a mini-batch fit) step the SAME item table;
    ``scratch`` = a dict shared by such states: the per-step buffers are allocated once at the largest size any of them needs
    (``share_scratch``) instead of once per state."""

    def __init__(self, U0, V0, plan, n_components, wplan=None, dtype=torch.float32, V_tables=None, scratch=None):
        dev = plan.col_u.device
        self.r = int(n_components)
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('factor tables are stored as float32 or bfloat16')
        self.dtype = dtype
        self.sfx = '_bf16' if dtype is torch.bfloat16 else '_f32'
        self.ld = _lib.padded_ld(self.r, dtype)
        self.U = self._pad(U0, dev)
        self.U_nxt = torch.empty_like(self.U)
        if V_tables is None:
            self.V = self._pad(V0, dev)
            self.V_nxt = torch.empty_like(self.V)
        else:
            self.V, self.V_nxt = V_tables
        self.plan, self.wplan = plan, wplan
        need = dict(slab=max(plan.seg_u.n_slab, plan.seg_i.n_slab if plan.seg_i else 0,
                             wplan.seg_e.n_slab if wplan is not None and wplan.seg_e is not None else 0,
                             wplan.vrows.n_slab if wplan is not None and getattr(wplan, 'vrows', None) is not None else 0, 1) * self.ld,
                    loss_part=max(plan.seg_u.nseg, plan.n_users, 1))
        self.row_stationary = False
        if wplan is not None and wplan.sliced:
            m, S = wplan.R.shape
            self.row_stationary = row_stationary_wanted(m, plan.n_items, S, plan.nnz, self.r, dtype)
            self.users_per_launch = int(os.environ.get('TMF_G4_USERS', 32768))
            if self.row_stationary:
                nb = _lib.load_library().tmf_wmrb_gradu4_workspace_bytes(m, wplan.n_slices, self.users_per_launch)
                self.g4_sync = torch.zeros(max(nb // 4, 1), dtype=torch.int32, device=dev)
            # gradU partials: one layer per slice, or (above PART_BUDGET bytes) a single layer summed by per-slice launches -
            # eight layers summed by per-round launches when the slices are walked XCD-major
            ns = wplan.n_slices
            if ns * m * self.ld * 4 <= PART_BUDGET:
                self.part_layers, self.gradu_launches = ns, 0          # one launch, a layer per slice
            elif wplan.xcd_major:
                self.part_layers, self.gradu_launches = min(8, ns), 3  # a launch per round of eight slices
            else:
                self.part_layers, self.gradu_launches = 1, 1           # a launch per slice
            need.update(sp=m * S, pk=max(plan.nnz, 1), part=self.part_layers * max(m, 1) * self.ld)
            wplan.s6 = getattr(wplan, 's6', None)
            if scores6_wanted(plan, wplan, self.r, dtype):
                if wplan.s6 is None or wplan.s6.key != (self.r, dtype):
                    wplan.s6 = Scores6Plan(plan, wplan, self.r, dtype)
            else:
                wplan.s6 = None
            if wplan.seg_e is not None and scores5_wanted(plan, wplan, self.r, dtype):
                if getattr(wplan, 's5', None) is None or wplan.s5.key != (self.r, dtype):
                    wplan.s5 = Scores5Plan(plan, wplan, self.r, dtype)
            else:
                wplan.s5 = None
        if wplan is not None and wplan.rows4:
            L = _lib.load_library()
            per_group = L.tmf_wsum_rows4_rows_per_group(self.r, int(dtype is torch.bfloat16))
            if not per_group:
                raise ValueError('the row-stationary item pass needs rows of at least 16 lanes')
            cus = torch.cuda.get_device_properties(dev).multi_processor_count if torch.cuda.is_available() else 256
            self.rows4_per_launch = int(os.environ.get('TMF_ROWS4_PER_LAUNCH', 2 * cus * per_group))   # two workgroups per CU resident
            n_work = wplan.vrows.n_vrows if wplan.vrows is not None else plan.n_items
            nb = L.tmf_wsum_rows4_workspace_bytes(n_work, wplan.user_chunks, self.rows4_per_launch)
            self.rows4_sync = torch.zeros(max(nb // 4, 1), dtype=torch.int32, device=dev)
        self._need = need
        if scratch is None:
            self._bind({k: torch.zeros(v, dtype=torch.float32, device=dev) if k == 'loss_part'
                        else torch.empty(v, dtype=torch.float32, device=dev) for k, v in need.items()})
        else:
            scratch.setdefault('states', []).append(self)

    def _bind(self, bufs):
        """Views of the (possibly shared, larger) flat fp32 scratch buffers in this state's shapes."""
        need = self._need
        self.slab = bufs['slab'][:need['slab']].view(-1, self.ld)
        self.loss_part = bufs['loss_part'][:need['loss_part']]
        if 'sp' in need:
            m, S = self.wplan.R.shape
            self.sp = bufs['sp'][:need['sp']].view(m, S)               # sampled scores, R-sorted order
            self.pk = bufs['pk'][:need['pk']]                          # scores of the interactions, CSR order
            self.part = bufs['part'][:need['part']].view(-1, self.ld)

    def _pad(self, W, dev):
        W = torch.as_tensor(W).detach().to(device=dev, dtype=torch.float32)
        out = torch.zeros(W.shape[0], self.ld, dtype=self.dtype, device=dev)
        out[:, :self.r] = W  # bf16: round-to-nearest-even, like the kernels' stores
        return out

    def swap(self):
        self.U, self.U_nxt = self.U_nxt, self.U
        self.V, self.V_nxt = self.V_nxt, self.V


def share_scratch(scratch, dev):
    """Allocate the per-step buffers of the states registered in ``scratch`` once, at the largest size any of them needs."""
    states = scratch.get('states', [])
    sizes = {}
    for st in states:
        for k, v in st._need.items():
            sizes[k] = max(sizes.get(k, 0), v)
    bufs = {k: torch.zeros(v, dtype=torch.float32, device=dev) for k, v in sizes.items()}
    for st in states:
        st._bind(bufs)
    return bufs


class KernelTimer:
    """HIP-event brackets around named launches on the current stream (bench.py uses it to get the
    dominant kernel's own duration inside the timed region)."""

    def __init__(self):
        self.spans = {}

    def start(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.spans.setdefault(name, []).append([ev, None])

    def stop(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.spans[name][-1][1] = ev

    def mean_ms(self, name):
        v = [a.elapsed_time(b) for a, b in self.spans.get(name, []) if b is not None]
        return sum(v) / len(v) if v else float('nan')


def _row_pass_finish(lib, seg, slab, X_old, X_out, r, epi, adam, stream, sfx='_f32'):
    if seg.n_long:
        _lib.check(getattr(lib, 'tmf_combine_rows' + sfx)(_lib.ptr(seg.long_rows), _lib.ptr(seg.long_slab_beg), seg.n_long,
                                            _lib.ptr(slab), _lib.ptr(X_old), _lib.ptr(X_out), r, epi, adam, stream), lib)


def epoch_mse(st, adam, loss_out, item_epi=_lib.EPI_ADAM, item_out=None, prof=None, user_epi=_lib.EPI_ADAM, user_out=None):
    """One MSE epoch: user pass (+loss), item pass; both read the pre-update tables.
    loss_out: 1-element fp64 device tensor receiving sum_k (a_k - p_k)^2.
    item_epi / user_epi = EPI_GRAD write the raw fp32 gradient into item_out / user_out instead of the updated table
    (multi-GPU; optimisers other than the reference's)."""
    lib, p, r = _lib.get(), st.plan, st.r
    s = _lib.stream_ptr()
    U_out = st.U_nxt if user_out is None else user_out
    if prof:
        prof.start('mse_user_pass')
    mse_pass = getattr(lib, 'tmf_mse_pass' + st.sfx)
    _lib.check(mse_pass(p.seg_u.cstruct(), _lib.ptr(p.col_u), _lib.ptr(p.val_u), _lib.ptr(st.U),
                                    _lib.ptr(st.V), _lib.ptr(U_out), _lib.ptr(st.slab), _lib.ptr(st.loss_part),
                                    r, user_epi, adam, s), lib)
    if prof:
        prof.stop('mse_user_pass')
    _row_pass_finish(lib, p.seg_u, st.slab, st.U, U_out, r, user_epi, adam, s, st.sfx)
    _lib.check(lib.tmf_sum_f32(_lib.ptr(st.loss_part), p.seg_u.nseg, _lib.ptr(loss_out), s), lib)
    V_out = st.V_nxt if item_out is None else item_out
    if prof:
        prof.start('mse_item_pass')
    _lib.check(mse_pass(p.seg_i.cstruct(), _lib.ptr(p.row_i), _lib.ptr(p.val_i), _lib.ptr(st.V),
                                    _lib.ptr(st.U), _lib.ptr(V_out), _lib.ptr(st.slab), None, r, item_epi, adam, s), lib)
    if prof:
        prof.stop('mse_item_pass')
    _row_pass_finish(lib, p.seg_i, st.slab, st.V, V_out, r, item_epi, adam, s, st.sfx)


def _wmrb_user_pass_sliced(lib, st, adam, c, prof=None, user_epi=_lib.EPI_ADAM, U_out=None):
    """Sliced user pass: scores -> hinge -> gradU (+ weights into entry order) -> finish (csrc/tmf_wmrb.hip, tmf_hinge.hip)."""
    p, w, r = st.plan, st.wplan, st.r
    i32 = ctypes.c_int32
    m, S, ns = p.n_users, w.S, w.n_slices
    s = _lib.stream_ptr()
    lists = w.lists(p)

    def timed(name, rc_fn):
        if prof:
            prof.start(name)
        _lib.check(rc_fn(), lib)
        if prof:
            prof.stop(name)
    s5, s6 = getattr(w, 's5', None), getattr(w, 's6', None)
    if s6 is not None:
        # flat streams on the slice-major grid: one workgroup per (slice, group of 32 users) chunk, the group's rows in LDS
        timed('wmrb_scores', lambda: getattr(lib, 'tmf_wmrb_scores6' + st.sfx)(
            _lib.ptr(s6.ids), _lib.ptr(s6.outs), _lib.ptr(s6.chunk_ptr), s6.n_groups, i32(s6.n_slices), m, p.n_items, _lib.ptr(st.U),
            _lib.ptr(st.V), _lib.ptr(st.sp), _lib.ptr(st.pk), r, s))
    elif s5 is not None:
        # row-stationary scores: workgroups own 256 users (rows in LDS) and walk one flat stream of (user, item) pairs
        timed('wmrb_scores', lambda: getattr(lib, 'tmf_wmrb_scores5' + st.sfx)(
            _lib.ptr(s5.ids), _lib.ptr(s5.outs), _lib.ptr(s5.wg_ptr), s5.n_wg, m, p.n_items, _lib.ptr(st.U), _lib.ptr(st.V),
            _lib.ptr(st.sp), _lib.ptr(st.pk), r, s5.wgs_per_launch, _lib.ptr(s5.wstart) if s5.paced else None, i32(s5.n_windows),
            s5.lag, _lib.ptr(s5.sync) if s5.paced else None, s5.sync.numel() * 4, s))
    else:
        timed('wmrb_scores', lambda: getattr(lib, 'tmf_wmrb_scores3' + st.sfx)(lists, _lib.ptr(st.U), _lib.ptr(st.V), _lib.ptr(st.sp),
                                                                                _lib.ptr(st.pk), r, s))
    timed('wmrb_hinge', lambda: lib.tmf_wmrb_hinge2_ordered(_lib.ptr(p.rowptr_u), _lib.ptr(p.val_u), _lib.ptr(st.pk), _lib.ptr(st.sp),
                                                            i32(m), i32(S), c, _lib.ptr(w.delta), _lib.ptr(w.D), _lib.ptr(st.loss_part),
                                                            _lib.ptr(w.hinge_order), s))
    if st.row_stationary:
        # gradU + finish in one row-stationary kernel (lane groups own users and walk the slices; no partial rows)
        timed('wmrb_gradu', lambda: getattr(lib, 'tmf_wmrb_gradu4' + st.sfx)(
            lists, _lib.ptr(w.D), _lib.ptr(w.delta), _lib.ptr(st.V), _lib.ptr(st.U), _lib.ptr(st.U_nxt if U_out is None else U_out),
            r, user_epi, adam, i32(st.users_per_launch), _lib.ptr(st.g4_sync), st.g4_sync.numel() * 4, s))
        return
    timed('wmrb_gradu', lambda: getattr(lib, 'tmf_wmrb_gradu3' + st.sfx)(
        lists, _lib.ptr(w.D), _lib.ptr(w.delta), _lib.ptr(st.V), _lib.ptr(st.part),
        st.gradu_launches, r, s))
    timed('wmrb_finish', lambda: getattr(lib, 'tmf_wmrb_finish' + st.sfx)(_lib.ptr(st.part), i32(st.part_layers), i32(m),
                                                                          _lib.ptr(st.U), _lib.ptr(st.U_nxt if U_out is None else U_out),
                                                                          r, user_epi, adam, s))


def epoch_wmrb(st, adam, c, loss_out, item_epi=_lib.EPI_ADAM, item_out=None, prof=None, user_epi=_lib.EPI_ADAM, user_out=None):
    """One WMRB epoch.  loss_out receives sum over positives of log(1 + M_k).  item_epi / user_epi as in epoch_mse."""
    lib, p, w, r = _lib.get(), st.plan, st.wplan, st.r
    s = _lib.stream_ptr()
    if prof:
        prof.start('wmrb_user_pass')
    if w.sliced:
        _wmrb_user_pass_sliced(lib, st, adam, c, prof, user_epi, user_out)
    else:
        _lib.check(getattr(lib, 'tmf_wmrb_user_pass' + st.sfx)(_lib.ptr(p.rowptr_u), _lib.ptr(p.col_u), _lib.ptr(p.val_u), _lib.ptr(w.R),
                                              p.n_users, w.S, c, _lib.ptr(st.U), _lib.ptr(st.V),
                                              _lib.ptr(st.U_nxt if user_out is None else user_out),
                                              _lib.ptr(w.delta), _lib.ptr(w.D), _lib.ptr(st.loss_part), None,
                                              r, user_epi, adam, s), lib)
    if prof:
        prof.stop('wmrb_user_pass')
    _lib.check(lib.tmf_sum_f32(_lib.ptr(st.loss_part), p.n_users, _lib.ptr(loss_out), s), lib)
    V_out = st.V_nxt if item_out is None else item_out
    if w.rows4:
        # row-stationary item pass: lane groups own (virtual) rows and walk the user blocks; no slab but for the parts of cut rows
        if prof:
            prof.start('wmrb_item_pass')
        i32 = ctypes.c_int32
        if w.vrows is not None:
            v = w.vrows
            _lib.check(getattr(lib, 'tmf_wsum_rows5' + st.sfx)(_lib.ptr(w.rowptr_e), i32(p.n_items), i32(w.user_chunks), _lib.ptr(w.ent_row),
                                                               _lib.ptr(w.ent_w), _lib.ptr(w.wbuf), _lib.ptr(st.U), _lib.ptr(st.V), _lib.ptr(V_out),
                                                               _lib.ptr(st.slab), _lib.ptr(v.item), _lib.ptr(v.part), _lib.ptr(v.nparts),
                                                               _lib.ptr(v.slot), i32(v.n_vrows), r, item_epi, adam, i32(st.rows4_per_launch),
                                                               _lib.ptr(st.rows4_sync), st.rows4_sync.numel() * 4, s), lib)
        else:
            _lib.check(getattr(lib, 'tmf_wsum_rows4' + st.sfx)(_lib.ptr(w.rowptr_e), i32(p.n_items), i32(w.user_chunks),
                                                               _lib.ptr(w.ent_row), _lib.ptr(w.ent_w), _lib.ptr(w.wbuf), _lib.ptr(st.U),
                                                               _lib.ptr(st.V), _lib.ptr(V_out), r, item_epi, adam,
                                                               i32(st.rows4_per_launch), _lib.ptr(st.rows4_sync),
                                                               st.rows4_sync.numel() * 4, s), lib)
        if prof:
            prof.stop('wmrb_item_pass')
        if w.vrows is not None and w.vrows.n_long:
            if prof:
                prof.start('wmrb_combine')
            v = w.vrows
            _lib.check(getattr(lib, 'tmf_combine_rows' + st.sfx)(_lib.ptr(v.long_rows), _lib.ptr(v.long_slab_beg), v.n_long, _lib.ptr(st.slab),
                                                                 _lib.ptr(st.V), _lib.ptr(V_out), r, item_epi, adam, s), lib)
            if prof:
                prof.stop('wmrb_combine')
        return
    if prof:
        prof.start('wmrb_item_pass')
    _lib.check(getattr(lib, 'tmf_wsum_pass' + st.sfx)(w.seg_e.cstruct(), _lib.ptr(w.ent_row), _lib.ptr(w.ent_w), _lib.ptr(w.wbuf),
                                     _lib.ptr(st.U), _lib.ptr(st.V), _lib.ptr(V_out), _lib.ptr(st.slab), r, item_epi,
                                     adam, s), lib)
    if prof:
        prof.stop('wmrb_item_pass')
    if prof:
        prof.start('wmrb_combine')
    _row_pass_finish(lib, w.seg_e, st.slab, st.V, V_out, r, item_epi, adam, s, st.sfx)
    if prof:
        prof.stop('wmrb_combine')


def adam_constants(lr):
    return _lib.load_library().tmf_adam_fresh(float(lr))
