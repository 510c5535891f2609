"""OPT-IN EXTENSION, not the reference's behaviour: mini-batch training over user batches.

The reference is full-batch: one optimiser step per epoch on the gradient of the loss summed over ALL interactions
(matrix_factorization.py:130-176).  With ``model.batch_users = B`` an epoch is instead a sweep over contiguous batches of B
users, each one a complete step of the same kind on that batch's part of the loss: its users' rows of U and the whole item
table V are updated by the reference's fresh Adam from the gradient of  sum over the batch's interactions  (MSE) /
positives (WMRB; the loss of a positive only involves its own user's scores, so it is separable by user).  The next batch
sees the updated V.  Every batch is the resident engine's epoch (``_engine.epoch_mse`` / ``epoch_wmrb``) on the batch's own
plans; the batches share the double-buffered item table and one set of per-step scratch buffers."""
import timeit

import numpy as np
import torch

from . import _engine, _lib


def fit_minibatch(model, epochs, n_users, n_items, interactions, lr, U0, V0, batch_users):
    from .mf.loss_graphs import WMRBLoss
    dev = interactions.device
    wmrb = isinstance(model.loss_graph, WMRBLoss)
    if getattr(model, 'optimizer', 'fresh_adam') != 'fresh_adam':
        raise ValueError("batch_users needs optimizer='fresh_adam' (every batch step is the reference's first Adam step)")
    B = int(batch_users)
    if B <= 0:
        raise ValueError(f'batch_users={batch_users}')
    t_plan = timeit.default_timer()
    r, dtype = model.n_components, model.factor_dtype
    ld = _lib.padded_ld(r, dtype)
    u = interactions.indices[:, 0]
    R = None
    if wmrb:
        R = torch.as_tensor(model.random_ind).to(device=dev, dtype=torch.int32).contiguous()
        if R.dim() != 2 or R.shape[0] != n_users:
            raise ValueError(f'random_ind has shape {tuple(R.shape)}, expected [{n_users}, n_samples]')
        if R.numel() and (int(R.min()) < 0 or int(R.max()) >= n_items):
            raise IndexError('random_ind holds item ids outside [0, n_items)')
    c = model.n_items / model.n_samples if wmrb else 0.0
    U0 = torch.as_tensor(U0).detach()
    V = torch.zeros(n_items, ld, dtype=dtype, device=dev)
    V[:, :r] = torch.as_tensor(V0).detach().to(device=dev, dtype=torch.float32)
    tables = [V, torch.empty_like(V)]
    scratch, states, bounds = {}, [], list(range(0, n_users, B)) + [n_users]
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        keep = (u >= b0) & (u < b1)
        idx, val = _engine.take_interactions(interactions.indices, interactions.values, keep, user_offset=b0)
        plan = _engine.InteractionPlan(idx, val, b1 - b0, n_items,
                                       user_chunks=1 if wmrb else _engine.mse_user_chunks(), csc=not wmrb)
        wplan = None
        if wmrb:
            ns, sliced = _engine.choose_wmrb_user_pass(b1 - b0, n_items, ld, int(R.shape[1]), plan.n_pos, r,
                                                       elem_size=2 if dtype is torch.bfloat16 else 4)
            wplan = _engine.WmrbPlan(plan, R[b0:b1].contiguous(), user_chunks=_engine.default_user_chunks(b1 - b0, ld, n_items=n_items),
                                     item_slices=ns, n_components=r, sliced=sliced)
        states.append(_engine.TrainState(U0[b0:b1], None, plan, r, wplan, dtype=dtype, V_tables=tuple(tables), scratch=scratch))
    _engine.share_scratch(scratch, dev)
    adam = _engine.adam_constants(lr)
    nb = len(states)
    sums = torch.zeros(max(epochs, 1), nb, dtype=torch.float64, device=dev)
    denom = sum(st.plan.n_pos if wmrb else st.plan.nnz for st in states)
    torch.cuda.synchronize(dev)
    t0 = timeit.default_timer()
    model.plan_seconds_ = t0 - t_plan
    for epoch in range(epochs):
        for i, st in enumerate(states):
            st.V, st.V_nxt = tables
            if wmrb:
                _engine.epoch_wmrb(st, adam, c, sums[epoch, i:i + 1])
            else:
                _engine.epoch_mse(st, adam, sums[epoch, i:i + 1])
            st.U, st.U_nxt = st.U_nxt, st.U
            tables.reverse()                      # the item table this batch wrote is the one the next batch reads
        if model.verbose and (epoch + 1) % 25 == 0:
            model._report(epoch, float(sums[epoch].sum()) / denom if denom else float('nan'), timeit.default_timer() - t0)
    torch.cuda.synchronize(dev)
    model.fit_seconds_ = timeit.default_timer() - t0
    tot = sums[:epochs].sum(dim=1).cpu().numpy()
    # mean over the epoch of the batch losses, each evaluated with the tables its step started from
    model.loss_history_ = (tot / denom if denom else np.full(epochs, np.nan)).tolist()
    model._state = states
    model.user_embedding = torch.cat([st.U[:, :r] for st in states]) if states else torch.zeros(0, r, device=dev)
    model.item_embedding = tables[0][:, :r]
    model.user_trainable, model.item_trainable = [model.user_embedding], [model.item_embedding]
