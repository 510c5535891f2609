"""User-partitioned data parallelism over the GPUs of one node.

The reference is single-process; this is the one exchange step its algorithm has when users are
split across ranks.  Rank g owns a contiguous block of users: its rows of U, its interactions, its
rows of the negative table R and of D.  U is never communicated.  V is replicated for compute and
row-sharded for the optimiser:

    user pass            local (HIP)                         U_blk <- fresh-Adam(U_blk, gU_blk)
    item pass            local (HIP, gradient epilogue)      gV_partial [n_pad, ld]
    reduce-scatter(sum)  RCCL over xGMI                      gV_shard   [n_pad / G, ld]
    fresh-Adam           local (HIP) on the owned V rows     the step is non-linear in g, so the sum
                                                             must complete before any update
    all-gather           RCCL                                V [n_pad, ld] replicated again
    all-reduce           2 doubles                           (sum of losses, count)

The compute is injected (``backend``) so the choreography can be exercised with gloo on CPU
(tests/test_dist_cpu.py drives it with the NumPy oracle); on the GPU the backend is ``HipBackend``.
"""
import collections

import torch
import torch.distributed as dist

from . import _engine, _lib


def partition_users(rowptr, world_size, per_user_cost=0):
    """Contiguous user blocks with balanced cost; cost(u) = #interactions(u) + per_user_cost
    (per_user_cost = n_samples for WMRB).  Returns world_size + 1 boundaries."""
    rowptr = torch.as_tensor(rowptr).to(torch.int64).cpu()
    m = rowptr.numel() - 1
    cost = (rowptr[1:] - rowptr[:-1]) + int(per_user_cost)
    cum = torch.cumsum(cost, 0)
    total = int(cum[-1]) if m else 0
    bounds = [0]
    for g in range(1, world_size):
        target = total * g // world_size
        b = int(torch.searchsorted(cum, torch.tensor(target), right=False)) + 1 if m else 0
        bounds.append(min(max(b, bounds[-1]), m))
    bounds.append(m)
    return bounds


def padded_rows(n, world_size):
    return (n + world_size - 1) // world_size * world_size


class HipBackend:
    """Local compute of one rank on the HIP engine.  ``st`` is an _engine.TrainState whose V tables
    have n_pad rows (rows >= n_items have no interactions and stay zero)."""

    def __init__(self, st, loss, c, adam, prof=None):
        self.st, self.loss, self.c, self.adam, self.prof = st, loss, c, adam, prof
        dev = st.V.device
        self.gV = torch.empty(st.V.shape, dtype=torch.float32, device=dev)  # raw gradient is fp32 whatever the table dtype
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)

    def local_passes(self):
        """U block updated into st.U_nxt; raw item gradient of this rank's users into self.gV."""
        if self.loss == 'wmrb':
            _engine.epoch_wmrb(self.st, self.adam, self.c, self.loss_out, _lib.EPI_GRAD, self.gV, self.prof)
        else:
            _engine.epoch_mse(self.st, self.adam, self.loss_out, _lib.EPI_GRAD, self.gV, self.prof)
        return self.gV, self.loss_out

    def adam_rows(self, W_rows, G_rows):
        lib = _lib.get()
        _lib.check(getattr(lib, 'tmf_adam_fresh_rows' + self.st.sfx)(_lib.ptr(W_rows), _lib.ptr(G_rows), W_rows.shape[0], self.st.r,
                                               self.adam, _lib.stream_ptr()), lib)

    def V(self):
        return self.st.V

    def V_next(self):
        """Where the all-gather assembles the updated table: the other half of the double buffer (no copy afterwards)."""
        return self.st.V_nxt

    def finish(self):
        self.st.swap()


def _staged(group, t):
    """gloo has no device collectives for every op used here: with a gloo group and device tensors the
    collective runs on host copies (rehearsal of the N>1 path with several ranks on ONE card, where RCCL
    refuses duplicate devices).  With RCCL - the production path - nothing is staged."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def reduce_scatter_sum(out, inp, group=None):
    if _staged(group, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(o, inp.cpu(), op=dist.ReduceOp.SUM, group=group)
        out.copy_(o)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


def all_gather_rows(out, inp, group=None):
    if _staged(group, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp.contiguous(), group=group)


def all_reduce_sum(t, group=None):
    if _staged(group, t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class DataParallelEpoch:
    """One epoch across the process group.  Collectives run on torch.distributed's default group
    ('nccl' = RCCL on the GPU box, 'gloo' in the CPU tests)."""

    def __init__(self, backend, local_count, group=None):
        self.b = backend
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        V = backend.V()
        if V.shape[0] % self.world:
            raise ValueError(f'V has {V.shape[0]} rows, not a multiple of world_size={self.world}')
        self.rows_per_rank = V.shape[0] // self.world
        self.g_shard = torch.empty(self.rows_per_rank, V.shape[1], dtype=torch.float32, device=V.device)
        self.stats = torch.zeros(2, dtype=torch.float64, device=V.device)
        # all-gather target: the backend's other V buffer when it has one (it becomes the table of the next epoch, no
        # copy), else a buffer of our own that is copied back; never the buffer the send rows live in (no aliasing)
        self.V_gather = None if hasattr(backend, 'V_next') else torch.empty_like(V)
        self.local_count = float(local_count)
        self.bytes = dict(reduce_scatter=V.numel() * 4, all_gather=V.numel() * V.element_size())
        # the first span (communicator set-up) and the latest 32: a long fit must not accumulate events without bound
        self._spans = {'reduce_scatter': collections.deque(maxlen=33), 'all_gather': collections.deque(maxlen=33)}
        self._first_span_kept = {'reduce_scatter': True, 'all_gather': True}

    def _timed(self, name, fn, t):
        if not t.is_cuda:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        q = self._spans[name]
        if len(q) == q.maxlen:
            self._first_span_kept[name] = False   # the set-up call has rotated out
        q.append((a, b))

    def step(self):
        """Returns the global mean loss as a 0-d fp64 tensor (no host sync)."""
        gV, loss_sum = self.b.local_passes()
        self._timed('reduce_scatter', lambda: reduce_scatter_sum(self.g_shard, gV, self.group), gV)
        V = self.b.V()
        mine = V[self.rank * self.rows_per_rank:(self.rank + 1) * self.rows_per_rank]
        self.b.adam_rows(mine, self.g_shard)
        if self.V_gather is None:
            out = self.b.V_next()
            self._timed('all_gather', lambda: all_gather_rows(out, mine, self.group), mine)
        else:
            all_gather_rows(self.V_gather, mine, self.group)
            V.copy_(self.V_gather)
        self.stats[0] = loss_sum.reshape(())
        self.stats[1] = self.local_count
        all_reduce_sum(self.stats, self.group)
        self.b.finish()
        return self.stats[0] / self.stats[1]

    def comm_report(self):
        """Mean milliseconds of the two collectives on this rank (events on the launch stream, so queueing behind the
        local passes is not included), their payloads, and the process group they ran on."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()   # elapsed_time raises on events that have not completed

        def mean(name):
            spans = list(self._spans[name])
            skip = 1 if self._first_span_kept[name] and len(spans) > 1 else 0   # the first call builds the communicator
            v = [a.elapsed_time(b) for a, b in spans[skip:]]
            return sum(v) / len(v) if v else None
        return dict(backend=dist.get_backend(self.group), ranks=self.world,
                    reduce_scatter_ms=mean('reduce_scatter'), reduce_scatter_bytes=self.bytes['reduce_scatter'],
                    all_gather_ms=mean('all_gather'), all_gather_bytes=self.bytes['all_gather'],
                    note='per rank and epoch: reduce-scatter(sum) of the fp32 item gradient, all-gather of the updated item rows; '
                         'not overlapped with compute (a few ms against ~100 ms of local passes at C4)')


# ------------------------------------------------------------------------------------------------
# Item-row-sharded V: the table is never resident as a whole (north star: tables beyond one GPU's HBM).
# ------------------------------------------------------------------------------------------------
def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0   # no process group: one rank that owns every row (single-GPU streaming of V, tests)


class _Done:
    def wait(self):
        pass


def _global_rank(group, r):
    return r if group is None else dist.get_global_rank(group, r)


def all_gather_rows_async(out, inp, group=None):
    """``out`` [world * rows, ld] <- the ranks' ``inp`` [rows, ld] in rank order; returns a handle whose wait() orders the
    CURRENT stream after it (RCCL: the collective runs on the communicator's stream, which itself first waits for the work
    already queued on the current stream - so a staging buffer is not overwritten while an earlier kernel still reads it)."""
    if _staged(group, inp):
        all_gather_rows(out, inp, group)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)


def reduce_scatter_sum_async(out, inp, group=None):
    """``out`` [rows, ld] <- sum over ranks of block `rank` of their ``inp`` [world * rows, ld]."""
    if _staged(group, inp):
        reduce_scatter_sum(out, inp, group)
        return _Done()
    return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=True)


def owned_blocks(rows_per_window, n_windows, world, rank, n_items):
    """[(local_begin, global_begin, count)]: the runs of VALID catalog rows (< n_items) that ``rank`` owns - one per window,
    ``count`` rows from global row ``global_begin`` stored from local row ``local_begin`` on.  Tables are filled and read back
    with these slices rather than with index gathers (torch's row gather miscomputes offsets on multi-GB tensors here,
    tools/torch_row_index_probe.py)."""
    sub = rows_per_window // world
    out = []
    for t in range(n_windows):
        g0 = t * rows_per_window + rank * sub
        cnt = max(0, min(g0 + sub, n_items) - g0)
        if cnt:
            out.append((t * sub, g0, cnt))
    return out


def owned_item_rows(rows_per_window, n_windows, world, rank, device=None):
    """Global row ids (int64, ascending) of the padded catalog that ``rank`` owns: of every window its ``rank``-th
    sub-block of rows_per_window / world rows."""
    sub = rows_per_window // world
    t = torch.arange(n_windows, device=device, dtype=torch.int64)[:, None] * rows_per_window
    return (t + rank * sub + torch.arange(sub, device=device, dtype=torch.int64)[None, :]).reshape(-1)


class ItemShardedEpoch:
    """One epoch with item-row-sharded V.  The padded catalog is cut into T windows of equal row count and every window into
    ``world`` sub-blocks: rank g owns sub-block g of EVERY window - those rows and their optimiser step - so a window is
    assembled by an all-gather to which every rank contributes 1/world, and its gradient is returned by a reduce-scatter.
    No rank ever holds more than its own rows (1/world of the table) plus two staging windows.  Users stay partitioned as in
    DataParallelEpoch; U is never communicated.

        walk 1 (WMRB only)   for every window t, in order:   all-gather(V rows of t)  ->  scores of t
        hinge                local
        walk 2               for every window t, in order:   all-gather(V rows of t)  ->  user-gradient partial of t,
                                                             raw item gradient of t   ->  reduce-scatter(sum) to the owners
        U <- fresh-Adam      local;      owned V rows <- fresh-Adam, after every window went out for the last time
        all-reduce           2 doubles   (sum of losses, count)

    The all-gather of window t+1 and the reduce-scatter of window t-1 are in flight while window t computes (two staging
    buffers each; the collectives are asynchronous on the RCCL stream).  Why interleaved ownership and not one owner per
    window: xGMI is point-to-point (7 links x ~153 GB/s per GPU).  With one owner a window of w bytes leaves through that GPU's
    links only - every receiver gets it over ONE link, w / 153 GB/s; with every rank holding 1/world of the window all links
    of all GPUs carry w / world each, world times faster - the same reason the replicated path uses reduce-scatter + all-gather
    rather than reduce + broadcast.  The table crosses the fabric once per walk, overlapped with that walk's compute.
    STATUS - design intent, UNMEASURED: the asynchronous branch has only run on a 1-rank nccl group (no peer, no real overlap)
    and, host-staged and synchronous, under gloo; no run over RCCL with world > 1 has happened, so neither the overlap nor
    the link arithmetic above has been observed on hardware.
    ``backend``: V_own(), two_phase, scores_window(t, Vwin), between(), grads_window(t, Vwin, out), finish_users(),
    adam_rows(W, G) - teamoflow_amd._windowed.WindowedHipBackend on the GPU, the NumPy oracle in tests/test_dist_cpu.py."""

    def __init__(self, backend, local_count, n_windows, group=None, always_collective=False):
        self.b, self.group = backend, group
        self.world, self.rank = _world(group)
        # always_collective: run the all-gather / reduce-scatter even with one rank (a test of the asynchronous RCCL path on
        # one GPU: staging buffers, stream ordering, dtypes); otherwise one rank reads and writes its own rows directly
        self.collective = self.world > 1 or (bool(always_collective) and dist.is_available() and dist.is_initialized())
        self.T = int(n_windows)
        own = backend.V_own()
        if own.shape[0] % self.T:
            raise ValueError(f'{own.shape[0]} owned rows are not a multiple of the {self.T} windows')
        self.sub = own.shape[0] // self.T           # owned rows per window
        self.rows = self.sub * self.world           # rows of a window
        dev, ld = own.device, own.shape[1]
        multi = self.collective
        self.stage = [torch.empty(self.rows, ld, dtype=own.dtype, device=dev) for _ in range(2)] if multi else []
        self.gbuf = [torch.empty(self.rows, ld, dtype=torch.float32, device=dev) for _ in range(2)] if multi else []
        self.g_own = torch.empty(own.shape[0], ld, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(2, dtype=torch.float64, device=dev)
        self.local_count = float(local_count)
        self.bytes = dict(all_gather_per_walk=self.T * self.rows * ld * own.element_size(),
                          reduce_scatter=self.T * self.rows * ld * 4, walks=2 if backend.two_phase else 1)

    def _fetch(self, t):
        mine = self.b.V_own()[t * self.sub:(t + 1) * self.sub]
        if not self.collective:
            return mine, _Done()      # one rank: the window is its own rows
        buf = self.stage[t % 2]
        return buf, all_gather_rows_async(buf, mine, self.group)

    def _windows(self):
        nxt = self._fetch(0)
        for t in range(self.T):
            cur, handle = nxt
            handle.wait()
            if t + 1 < self.T:
                nxt = self._fetch(t + 1)
            yield t, cur

    def step(self):
        """Returns the global mean loss as a 0-d fp64 tensor."""
        b = self.b
        if b.two_phase:
            for t, Vwin in self._windows():
                b.scores_window(t, Vwin)
            b.between()
        pending = [_Done(), _Done()]
        for t, Vwin in self._windows():
            mine = self.g_own[t * self.sub:(t + 1) * self.sub]
            if not self.collective:
                b.grads_window(t, Vwin, mine)
                continue
            pending[t % 2].wait()           # the reduce-scatter that read this staging buffer two windows ago
            out = self.gbuf[t % 2]
            b.grads_window(t, Vwin, out)
            pending[t % 2] = reduce_scatter_sum_async(mine, out, self.group)
        for h in pending:
            h.wait()
        loss_sum = b.finish_users()
        b.adam_rows(b.V_own(), self.g_own)   # every window of this epoch has been gathered: the rows may change now
        self.stats[0] = loss_sum.reshape(())
        self.stats[1] = self.local_count
        if self.world > 1:
            all_reduce_sum(self.stats, self.group)
        return self.stats[0] / self.stats[1]


def fit_item_sharded(model, epochs, n_users, n_items, interactions, lr, U0, V0, windows_per_rank=1, group=None):
    """``MatrixFactorization.fit`` with item-row-sharded V (``model.shard_items = q``: world * q windows): users are partitioned
    over the ranks as in ``fit_data_parallel``; of every catalog window each rank owns one sub-block of rows, and the windows
    only pass through the ranks one at a time (``ItemShardedEpoch``).  Without a process group it is one rank streaming its
    own table window by window.  Every rank passes the same global inputs (or, with ``model.local_users``, its own users and
    - ``model.local_items = True`` - only the initial rows ``owned_item_rows`` < n_items names, in that order).  On return
    ``user_embedding`` is this rank's user block (``user_block``), ``item_embedding`` its item rows (``item_rows``: their
    global ids); ``gather_item_embedding`` assembles the catalog where it fits."""
    import timeit
    from ._windowed import WindowedHipBackend, window_geometry
    from .mf.loss_graphs import WMRBLoss
    t_plan = timeit.default_timer()
    if getattr(model, 'optimizer', 'fresh_adam') != 'fresh_adam':
        raise ValueError("the multi-GPU fits implement the reference's optimiser only (optimizer='fresh_adam')")
    world, rank = _world(group)
    dev = interactions.device
    wmrb = isinstance(model.loss_graph, WMRBLoss)
    T = world * max(1, int(windows_per_rank))
    r, dtype = model.n_components, model.factor_dtype
    ld = _lib.padded_ld(r, dtype)
    rows, _, n_pad = window_geometry(n_items, T, ld, 2 if dtype is torch.bfloat16 else 4, world)
    local = getattr(model, 'local_users', None)
    if local is not None:
        b, e = int(local[0]), int(local[1])
        idx, val = interactions.indices, interactions.values
    else:
        u = interactions.indices[:, 0]
        rowptr = _engine._excl_cumsum(torch.bincount(u, minlength=n_users))
        S = int(model.random_ind.shape[1]) if wmrb else 0
        bounds = partition_users(rowptr, world, per_user_cost=S)
        b, e = bounds[rank], bounds[rank + 1]
        keep = (u >= b) & (u < e)
        idx, val = _engine.take_interactions(interactions.indices, interactions.values, keep, user_offset=b)
    R, c = None, 0.0
    if wmrb:
        Rall = torch.as_tensor(model.random_ind)
        want_rows = e - b if local is not None else n_users
        if Rall.dim() != 2 or Rall.shape[0] != want_rows:
            raise ValueError(f'random_ind has shape {tuple(Rall.shape)}, expected [{want_rows}, n_samples]')
        R = (Rall if local is not None else Rall[b:e]).to(device=dev, dtype=torch.int32).contiguous()
        if R.numel() and (int(R.min()) < 0 or int(R.max()) >= n_items):
            raise IndexError('random_ind holds item ids outside [0, n_items)')
        c = model.n_items / model.n_samples
    mine = owned_item_rows(rows, T, world, rank, dev)          # global ids of the owned rows of the padded catalog
    blocks = owned_blocks(rows, T, world, rank, n_items)       # the same as slices: (local row, global row, count) per window
    n_valid = sum(cnt for _, _, cnt in blocks)
    V0 = torch.as_tensor(V0).detach()
    V_own = torch.zeros(mine.numel(), ld, dtype=dtype, device=dev)
    local_items = bool(getattr(model, 'local_items', False))
    if local_items and V0.shape[0] != n_valid:
        raise ValueError(f'local_items: the item initialiser must return the {n_valid} rows this rank owns')
    given = 0
    for l0, g0, cnt in blocks:
        src = V0[given:given + cnt] if local_items else V0[g0:g0 + cnt]
        V_own[l0:l0 + cnt, :r] = src.to(device=dev, dtype=torch.float32)      # rounds to the table dtype on assignment
        given += cnt
    U_blk = torch.as_tensor(U0).detach()
    if local is not None and U_blk.shape[0] != e - b:
        raise ValueError(f'local_users = {local}: the user initialiser must return the {e - b} rows of this block')
    U_blk = U_blk if local is not None else U_blk[b:e]
    backend = WindowedHipBackend(U_blk, V_own, idx, val, R, e - b, n_items, T, r, 'wmrb' if wmrb else 'mse', c, lr, dtype=dtype,
                                 world=world)
    ep = ItemShardedEpoch(backend, backend.n_loss, T, group=group, always_collective=getattr(model, 'shard_always_collective', False))
    losses = torch.zeros(max(epochs, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    t0 = timeit.default_timer()
    model.plan_seconds_ = t0 - t_plan
    for epoch in range(epochs):
        losses[epoch] = ep.step()
    torch.cuda.synchronize(dev)
    model.fit_seconds_ = timeit.default_timer() - t0
    model.loss_history_ = losses[:epochs].cpu().tolist()
    model._state, model.user_block, model.item_rows = backend, (b, e), mine[mine < n_items]
    model._sharded_epoch, model._n_items_fit = ep, n_items
    model.user_embedding = backend.U[:, :r]
    # one rank: the owned rows are the catalog in order (a view); several: the valid owned rows, in item_rows order (a copy)
    model.item_embedding = (V_own[:n_items, :r] if world == 1 else
                            torch.cat([V_own[l0:l0 + cnt, :r] for l0, _, cnt in blocks]) if blocks else V_own[:0, :r])
    model.user_trainable, model.item_trainable = [model.user_embedding], [model.item_embedding]


def sharded_top_items(model, k, clamp_negatives=False, users=None):
    """Top-k item ids (int32, global) of THIS rank's users over the whole item-row-sharded catalog - what ``recall_at_k`` /
    ``retrieve_user_recs`` (matrix_factorization.py:236-248, :424-438) rank with.  A collective: every rank calls it (with its own
    users); the windows are broadcast once more and each is ranked by the fused GEMM + top-k kernel, the per-window lists
    (value desc, index asc - already in catalog order among equal values, because windows are visited in item order) are merged
    by one more stable top-k over the <= T k candidates.  Scores are the same MFMA dot products as on the resident path, so the
    result is identical to ranking the assembled table."""
    from . import _ops
    ep, be = model._sharded_epoch, model._state
    r, n_items = model.n_components, model._n_items_fit
    U = be.U[:, :r] if users is None else be.U[users:users + 1, :r]
    vals, ids = [], []
    for t, Vwin in ep._windows():
        valid = min(be.rows, n_items - t * be.rows)        # rows beyond the catalog are padding: never candidates
        if valid <= 0 or U.shape[0] == 0:
            continue
        kt = min(int(k), valid)
        W = Vwin[:valid, :r]
        if _ops.fused_topk_supported(U, W, kt):
            v, i = _ops.predict_topk(U, W, kt, clamp_negatives=clamp_negatives, return_values=True)
        else:
            v, i = _ops.topk_stable(_ops.predict_gemm(U.float(), W.float()), kt, clamp_negatives=clamp_negatives, return_values=True)
        vals.append(v)
        ids.append(i + t * be.rows)
    if not vals:
        return torch.zeros(U.shape[0], 0, dtype=torch.int32, device=be.U.device)
    cv, ci = torch.cat(vals, dim=1), torch.cat(ids, dim=1)
    pos = _ops.topk_stable(cv, min(int(k), cv.shape[1]), clamp_negatives=clamp_negatives)
    return torch.gather(ci, 1, pos.to(torch.int64))


def gather_item_embedding(model, n_items, group=None):
    """Full [n_items, r] item table from the owned sub-blocks (for catalogs that fit after all: tests, export)."""
    world, rank = _world(group)
    ep = model._sharded_epoch
    mine = model._state.V_shard[:, :model.n_components].float().contiguous()
    if world == 1:
        return mine[:n_items]
    out = torch.empty(world * mine.shape[0], mine.shape[1], dtype=torch.float32, device=mine.device)
    all_gather_rows(out, mine, group)                       # [rank][window][sub-block row]
    out = out.view(world, ep.T, ep.sub, -1).permute(1, 0, 2, 3).reshape(ep.T * ep.rows, -1)   # catalog order
    return out[:n_items]


def fit_data_parallel(model, epochs, n_users, n_items, interactions, lr, U0, V0, group=None):
    """``MatrixFactorization.fit`` across the ranks of an initialised process group (one process per GPU).

    Every rank passes the SAME global inputs (interactions, model.random_ind, initial weights); it keeps the
    users of its block (``partition_users``: balanced by interactions + negatives), trains them against the
    replicated V and exchanges the item gradient once per epoch (``DataParallelEpoch``).  A rank that only holds its
    own users sets ``model.local_users = (begin, end)`` first: ``interactions`` (user ids relative to ``begin``),
    ``model.random_ind`` and the user initialiser's rows are then this block's and nothing global is touched.
    On return the model holds ``item_embedding`` (replicated), ``user_embedding`` = this rank's block,
    ``user_block`` = (begin, end); ``gather_user_embedding(model)`` assembles the full table."""
    from .mf.loss_graphs import WMRBLoss
    if getattr(model, 'optimizer', 'fresh_adam') != 'fresh_adam':
        raise ValueError("the multi-GPU fits implement the reference's optimiser only (optimizer='fresh_adam')")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = interactions.device
    wmrb = isinstance(model.loss_graph, WMRBLoss)
    local = getattr(model, 'local_users', None)
    if local is not None:
        b, e = int(local[0]), int(local[1])
        idx, val = interactions.indices, interactions.values
        if tuple(torch.as_tensor(U0).shape)[0] != e - b:
            raise ValueError(f'local_users = {local}: the user initialiser must return the {e - b} rows of this block')
    else:
        u = interactions.indices[:, 0]
        rowptr = _engine._excl_cumsum(torch.bincount(u, minlength=n_users))
        S = int(model.random_ind.shape[1]) if wmrb else 0
        bounds = partition_users(rowptr, world, per_user_cost=S)
        b, e = bounds[rank], bounds[rank + 1]
        keep = (u >= b) & (u < e)
        idx, val = _engine.take_interactions(interactions.indices, interactions.values, keep, user_offset=b)
    n_pad = padded_rows(n_items, world)
    plan = _engine.InteractionPlan(idx, val, e - b, n_pad, user_chunks=1 if wmrb else _engine.mse_user_chunks(), csc=not wmrb)
    ld = _lib.padded_ld(model.n_components, model.factor_dtype)
    wplan, c = None, 0.0
    if wmrb:
        Rall = torch.as_tensor(model.random_ind)
        want_rows = e - b if local is not None else n_users
        if Rall.dim() != 2 or Rall.shape[0] != want_rows:
            raise ValueError(f'random_ind has shape {tuple(Rall.shape)}, expected [{want_rows}, n_samples]')
        R = (Rall if local is not None else Rall[b:e]).to(device=dev, dtype=torch.int32).contiguous()
        if R.numel() and (int(R.min()) < 0 or int(R.max()) >= n_items):
            raise IndexError('random_ind holds item ids outside [0, n_items)')
        c = model.n_items / model.n_samples
        wplan = _engine.wmrb_plan_for(plan, R, model.n_components, model.factor_dtype)
    V0p = torch.zeros(n_pad, model.n_components, dtype=torch.float32, device=dev)
    V0p[:n_items] = torch.as_tensor(V0).detach().to(device=dev, dtype=torch.float32)
    U_blk = torch.as_tensor(U0).detach() if local is not None else torch.as_tensor(U0).detach()[b:e]
    st = _engine.TrainState(U_blk, V0p, plan, model.n_components, wplan, dtype=model.factor_dtype)
    adam = _engine.adam_constants(lr)
    backend = HipBackend(st, 'wmrb' if wmrb else 'mse', c, adam)
    dp = DataParallelEpoch(backend, plan.n_pos if wmrb else plan.nnz, group=group)
    losses = torch.zeros(max(epochs, 1), dtype=torch.float64, device=dev)
    for epoch in range(epochs):
        losses[epoch] = dp.step()
    torch.cuda.synchronize(dev)
    model.loss_history_ = losses[:epochs].cpu().tolist()
    model._state, model.user_block = st, (b, e)
    r = model.n_components
    model.user_embedding, model.item_embedding = st.U[:, :r], st.V[:n_items, :r]
    model.user_trainable, model.item_trainable = [model.user_embedding], [model.item_embedding]


def mean_recall_at_k(model, local_interactions, k=10, group=None):
    """Mean recall@k over ALL users of a multi-GPU fit (SURVEY.md §8e "predict / recall": user-partitioned, the final mean is a
    2-float all-reduce).  ``local_interactions``: this rank's users (ids relative to ``model.user_block[0]``) against the whole
    catalog.  Each rank ranks its own users - against the replicated table (``fit_data_parallel``) or window by window
    (``fit_item_sharded``, where the call is a collective anyway) - and only (sum of recalls, number of users with
    positives) crosses the wire.  Returns a Python float, the same on every rank."""
    rec = model.recall_at_k(local_interactions, k=k)             # users with at least one positive (matrix_factorization.py:255-258)
    stats = torch.stack([rec.double().sum(), torch.tensor(float(rec.numel()), dtype=torch.float64, device=rec.device)])
    if _world(group)[0] > 1:
        all_reduce_sum(stats, group)
    return float(stats[0] / stats[1]) if float(stats[1]) else float('nan')


def gather_user_embedding(model, n_users, group=None):
    """Full [n_users, r] user table from the per-rank blocks (blocks are padded to the largest one)."""
    world = dist.get_world_size(group)
    b, e = model.user_block
    sizes = torch.zeros(world, dtype=torch.int64, device=model.user_embedding.device)
    sizes[dist.get_rank(group)] = e - b
    all_reduce_sum(sizes, group)
    mx = int(sizes.max())
    mine = torch.zeros(mx, model.n_components, dtype=torch.float32, device=sizes.device)
    mine[:e - b] = model.user_embedding.float()
    out = torch.empty(world * mx, model.n_components, dtype=torch.float32, device=sizes.device)
    all_gather_rows(out, mine, group)
    return torch.cat([out[g * mx:g * mx + int(sizes[g])] for g in range(world)])[:n_users]
