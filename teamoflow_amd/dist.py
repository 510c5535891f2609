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
        self._spans = {'reduce_scatter': [], 'all_gather': []}

    def _timed(self, name, fn, t):
        if not t.is_cuda:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self._spans[name].append((a, b))

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
        def mean(spans):
            v = [a.elapsed_time(b) for a, b in spans[1:]] or [a.elapsed_time(b) for a, b in spans]   # the first call builds the communicator
            return sum(v) / len(v) if v else None
        return dict(backend=dist.get_backend(self.group), ranks=self.world,
                    reduce_scatter_ms=mean(self._spans['reduce_scatter']), reduce_scatter_bytes=self.bytes['reduce_scatter'],
                    all_gather_ms=mean(self._spans['all_gather']), all_gather_bytes=self.bytes['all_gather'],
                    note='per rank and epoch: reduce-scatter(sum) of the fp32 item gradient, all-gather of the updated item rows; '
                         'not overlapped with compute (a few ms against ~100 ms of local passes at C4)')


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
        idx = interactions.indices[keep].clone()
        idx[:, 0] -= b
        val = interactions.values[keep]
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
        wplan = _engine.WmrbPlan(plan, R, user_chunks=_engine.default_user_chunks(e - b, ld, n_items=n_pad),
                                 item_slices=_engine.default_item_slices(n_pad, ld), n_components=model.n_components)
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
