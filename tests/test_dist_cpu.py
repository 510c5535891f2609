"""world_size-2 gloo test of the data-parallel epoch (teamoflow_amd/dist.py) on CPU.

The HIP engine cannot run here, so the local compute is injected: a backend that evaluates the
oracle's closed forms for this rank's user block.  What is under test is the N>1 choreography -
user partition, raw item gradient -> reduce-scatter -> fresh-Adam on the owned rows -> in-place
all-gather -> loss all-reduce - which must reproduce the single-process oracle epoch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


class OracleBackend:
    def __init__(self, U_blk, V_pad, idx_local, val, R_blk, n_items, n_samples, lr, loss):
        self.U, self.Vt = U_blk.copy(), torch.tensor(V_pad)
        self.idx, self.val, self.R = idx_local, val, R_blk
        self.n_items, self.n_samples, self.lr, self.loss = n_items, n_samples, lr, loss

    def V(self):
        return self.Vt

    def local_passes(self):
        from oracle import sparse_ref as S
        V = self.Vt.numpy().copy()
        if self.loss == 'mse':
            U_new, _, _, t = S.mse_epoch(self.U, V, self.idx, self.val, self.lr)
            loss_sum = float(t['loss'].astype(np.float64).sum())
        else:
            U_new, _, _, t = S.wmrb_epoch(self.U, V, self.idx, self.val, self.R, self.n_items, self.n_samples, self.lr)
            loss_sum = float(t['loss'].astype(np.float64).sum())
        self.U_new = U_new
        return torch.tensor(t['gV']), torch.tensor([loss_sum], dtype=torch.float64)

    def adam_rows(self, W_rows, G_rows):
        from oracle import sparse_ref as S
        W_rows.copy_(torch.tensor(S.adam_fresh(W_rows.numpy().copy(), G_rows.numpy(), self.lr)))

    def finish(self):
        self.U = self.U_new


def _worker(rank, world, port, loss, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import sparse_ref as S
    from teamoflow_amd import dist as tdist
    rng = np.random.default_rng(0)
    m, n, r, Sn, lr = 23, 17, 6, 5, 0.05
    A = (rng.random((m, n)) < 0.3) * rng.integers(1, 6, (m, n))
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, Sn, replace=False) for _ in range(m)])
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(idx[:, 0], minlength=m))])
    bounds = tdist.partition_users(rowptr, world, per_user_cost=Sn if loss == 'wmrb' else 0)
    b, e = bounds[rank], bounds[rank + 1]
    sel = (idx[:, 0] >= b) & (idx[:, 0] < e)
    idx_l = idx[sel].copy()
    idx_l[:, 0] -= b
    n_pad = tdist.padded_rows(n, world)
    V_pad = np.zeros((n_pad, r), np.float32)
    V_pad[:n] = V0
    backend = OracleBackend(U0[b:e], V_pad, idx_l, val[sel], R[b:e], n, Sn, lr, loss)
    dp = tdist.DataParallelEpoch(backend, local_count=int(sel.sum()))
    losses = [float(dp.step()) for _ in range(3)]
    # single-process oracle
    U, V = U0.copy(), V0.copy()
    ref_losses = []
    for _ in range(3):
        if loss == 'mse':
            U, V, l, _ = S.mse_epoch(U, V, idx, val, lr)
        else:
            U, V, l, _ = S.wmrb_epoch(U, V, idx, val, R, n, Sn, lr)
        ref_losses.append(l)
    ok = (np.allclose(losses, ref_losses, rtol=1e-5)
          and np.abs(backend.U - U[b:e]).max() < 5e-3 * lr + 1e-6
          and np.abs(backend.V().numpy()[:n] - V).max() < 5e-2 * lr + 1e-6
          and np.all(backend.V().numpy()[n:] == 0))
    # V must be identical on every rank after the all-gather
    Vall = [torch.zeros_like(backend.V()) for _ in range(world)]
    dist.all_gather(Vall, backend.V())
    ok = ok and all(torch.equal(Vall[0], v) for v in Vall)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('loss,world', [('mse', 2), ('wmrb', 2), ('wmrb', 3)])
def test_data_parallel_epoch_matches_single_process(loss, world):
    port = 29600 + (os.getpid() % 200) + (0 if loss == 'mse' else 1) + 2 * world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, loss, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)


def test_partition_users_balances_cost():
    from teamoflow_amd.dist import padded_rows, partition_users
    rowptr = np.concatenate([[0], np.cumsum([100, 1, 1, 1, 100, 1, 1, 95])])
    assert partition_users(rowptr, 2) == [0, 4, 8] or partition_users(rowptr, 2) == [0, 5, 8]
    b = partition_users(rowptr, 3, per_user_cost=10)
    assert b[0] == 0 and b[-1] == 8 and all(x <= y for x, y in zip(b, b[1:]))
    assert partition_users(np.array([0]), 4) == [0, 0, 0, 0, 0]
    assert padded_rows(100000, 8) == 100000 and padded_rows(10, 4) == 12


def test_partition_with_more_ranks_than_users_and_empty_plans():
    """Skewed costs or world > n_users give empty user blocks: the boundaries stay monotone and a rank with no users
    still builds valid (empty) plans - its passes are no-ops and it only takes part in the collectives."""
    from teamoflow_amd._engine import InteractionPlan, WmrbPlan
    from teamoflow_amd.dist import partition_users
    rowptr = np.concatenate([[0], np.cumsum([1000, 1, 1])])
    for world in (2, 4, 8):
        b = partition_users(rowptr, world, per_user_cost=3)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 3 and all(x <= y for x, y in zip(b, b[1:]))
    assert any(x == y for x, y in zip(b, b[1:]))          # at 8 ranks some block is empty
    n, S = 7, 3
    plan = InteractionPlan(torch.zeros(0, 2, dtype=torch.int64), torch.zeros(0), 0, n)
    assert plan.nnz == 0 and plan.rowptr_u.tolist() == [0] and plan.seg_u.nseg == 0 and plan.n_pos == 0
    for slices in (1, 3):
        w = WmrbPlan(plan, torch.zeros(0, S, dtype=torch.int32), user_chunks=2, item_slices=slices)
        assert w.ent_row.numel() == 0 and w.rowptr_e.tolist() == [0] * (2 * n + 1) and tuple(w.D.shape) == (0, S)


# ------------------------------------------------------------------------------------------------
# Item-row-sharded V (dist.ItemShardedEpoch): no rank holds the table; every rank owns one sub-block of every window, a
# window is assembled by all-gather, its item gradient goes back by reduce-scatter and each rank updates its own rows.
# ------------------------------------------------------------------------------------------------
class WindowedOracleBackend:
    """Oracle closed forms behind the window interface.  MSE is evaluated window by window from the rows the epoch hands
    over (it is separable by item).  WMRB couples the items of a user: the windows received in walk 1 are assembled, the
    closed form is evaluated on the assembled table, and walk 2 checks that every window arrives again unchanged and hands
    out the matching rows of the item gradient."""

    def __init__(self, U_blk, V_own, idx_local, val, R_blk, n_items, n_samples, lr, loss, rows, T):
        self.U, self.own = U_blk.copy(), torch.tensor(V_own)
        self.idx, self.val, self.R = idx_local, val, R_blk
        self.n_items, self.n_samples, self.lr, self.loss, self.rows, self.T = n_items, n_samples, lr, loss, rows, T
        self.two_phase = loss == 'wmrb'
        self.seen = np.zeros((rows * T, U_blk.shape[1]), np.float32)
        self.log = []

    def V_own(self):
        return self.own

    def scores_window(self, t, Vwin):
        self.log.append(('scores', t))
        self.seen[t * self.rows:(t + 1) * self.rows] = Vwin.numpy()

    def between(self):
        from oracle import sparse_ref as S
        self.log.append(('hinge',))
        self.U_new, _, _, self.terms = S.wmrb_epoch(self.U, self.seen.copy(), self.idx, self.val, self.R, self.n_items,
                                                    self.n_samples, self.lr)
        self.loss_sum = float(self.terms['loss'].astype(np.float64).sum())

    def grads_window(self, t, Vwin, out):
        from oracle import sparse_ref as S
        self.log.append(('grads', t))
        lo = t * self.rows
        if self.loss == 'wmrb':
            assert np.array_equal(Vwin.numpy(), self.seen[lo:lo + self.rows])     # the owner has not stepped yet
            out.copy_(torch.tensor(self.terms['gV'][lo:lo + self.rows]))
            return
        if t == 0:
            self.gU, self.loss_sum = np.zeros(self.U.shape, np.float64), 0.0
        keep = (self.idx[:, 1] >= lo) & (self.idx[:, 1] < lo + self.rows)
        sub = self.idx[keep].copy()
        sub[:, 1] -= lo
        _, _, _, tm = S.mse_epoch(self.U, Vwin.numpy().copy(), sub, self.val[keep], self.lr)
        self.gU += tm['gU']
        self.loss_sum += float(tm['loss'].astype(np.float64).sum())
        out.copy_(torch.tensor(tm['gV'].astype(np.float32)))

    def finish_users(self):
        from oracle import sparse_ref as S
        self.U = self.U_new if self.loss == 'wmrb' else S.adam_fresh(self.U, self.gU.astype(np.float32), self.lr)
        return torch.tensor([self.loss_sum], dtype=torch.float64)

    def adam_rows(self, W_rows, G_rows):
        from oracle import sparse_ref as S
        W_rows.copy_(torch.tensor(S.adam_fresh(W_rows.numpy().copy(), G_rows.numpy(), self.lr)))


def _sharded_worker(rank, world, port, loss, q, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import sparse_ref as S
    from teamoflow_amd import dist as tdist
    rng = np.random.default_rng(1)
    m, n, r, Sn, lr = 19, 21, 5, 6, 0.05
    A = (rng.random((m, n)) < 0.3) * rng.integers(1, 6, (m, n))
    idx = np.argwhere(A != 0)
    val = A[A != 0].astype(np.float32)
    U0 = (rng.standard_normal((m, r)) * 0.3).astype(np.float32)
    V0 = (rng.standard_normal((n, r)) * 0.3).astype(np.float32)
    R = np.stack([rng.choice(n, Sn, replace=False) for _ in range(m)])
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(idx[:, 0], minlength=m))])
    bounds = tdist.partition_users(rowptr, world, per_user_cost=Sn if loss == 'wmrb' else 0)
    b, e = bounds[rank], bounds[rank + 1]
    sel = (idx[:, 0] >= b) & (idx[:, 0] < e)
    idx_l = idx[sel].copy()
    idx_l[:, 0] -= b
    T = world * q
    rows = -(-(-(-n // T)) // world) * world                 # a window splits into `world` equal sub-blocks
    V_pad = np.zeros((rows * T, r), np.float32)
    V_pad[:n] = V0
    mine = tdist.owned_item_rows(rows, T, world, rank).numpy()
    own = V_pad[mine]
    backend = WindowedOracleBackend(U0[b:e], own, idx_l, val[sel], R[b:e], n, Sn, lr, loss, rows, T)
    ep = tdist.ItemShardedEpoch(backend, local_count=int(sel.sum()), n_windows=T)
    losses = [float(ep.step()) for _ in range(3)]
    U, V = U0.copy(), V0.copy()
    ref_losses = []
    for _ in range(3):
        if loss == 'mse':
            U, V, l, _ = S.mse_epoch(U, V, idx, val, lr)
        else:
            U, V, l, _ = S.wmrb_epoch(U, V, idx, val, R, n, Sn, lr)
        ref_losses.append(l)
    V_ref = np.zeros_like(V_pad)
    V_ref[:n] = V
    got = backend.V_own().numpy()
    walk = [('scores', t) for t in range(T)] + [('hinge',)] if loss == 'wmrb' else []
    ok = (np.allclose(losses, ref_losses, rtol=1e-5)
          and np.abs(backend.U - U[b:e]).max() < 5e-3 * lr + 1e-6
          and np.abs(got - V_ref[mine]).max() < 5e-2 * lr + 1e-6
          and np.all(got[mine >= n] == 0)                                         # padding rows stay zero
          and backend.log[:len(walk) + T] == walk + [('grads', t) for t in range(T)]   # every window once per walk, in order
          and ep.stage[0].shape[0] == rows and len(ep.stage) == 2                 # two staging windows, never the table
          and got.shape[0] == rows * T // world)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('loss,q,world', [('mse', 1, 2), ('wmrb', 1, 2), ('mse', 3, 2), ('wmrb', 2, 2), ('wmrb', 2, 3), ('mse', 1, 4)])
def test_item_sharded_epoch_matches_single_process(loss, q, world):
    port = 29850 + (os.getpid() % 100) + 2 * q + (0 if loss == 'mse' else 1) + 10 * world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, loss, q, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)
