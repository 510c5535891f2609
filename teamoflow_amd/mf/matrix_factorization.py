"""``MatrixFactorization`` with the class surface of
/root/reference/src/teamoflow/mf/matrix_factorization.py:23-475, computed sparsely on MI355X.

Dispatch (same isinstance test the reference uses at :115,:136-162): a model made of
``LinearEmbedding`` x indicator features x (``MSELoss`` | ``WMRBLoss``) trains on the HIP engine
(``_engine.py`` -> libtmf.so) and needs a GPU - there is no CPU fallback for it.  Any other
combination of plug-ins (dense features, biased / ReLU embeddings, KL loss, user subclasses) trains
through ``_fit_generic``: the reference's dense loop written with torch autograd around the plug-ins'
own ``get_repr`` / ``get_loss``.
"""
import os
import timeit
import warnings

import numpy as np
import torch

from .. import _engine, _lib, _ops
from .embedding_graphs import BiasedLinearEmbedding, Embeddings, LinearEmbedding, ReLUEmbedding
from .initializer_graphs import NormalInitializer
from .loss_graphs import KLDivergenceLoss, LossGraph, MSELoss, WMRBLoss
from .sparse import IndicatorFeatures, SparseInteractions, default_device, is_indicator
from .utils import gather_matrix_indices, random_sampler, random_sampler_device

PREDICT_CHUNK_BYTES = 2 << 30  # users are scored in blocks of at most this many bytes of scores
HOST_SAMPLER_MAX_WORK = 2_000_000_000  # n_users * n_items above which generate_sample=True samples on the device
GRAPH_EPOCHS = 50              # epochs captured per hipGraph on launch-bound problems
GRAPH_MAX_WORK = 20_000_000    # interactions + sampled scores per epoch below which fit() uses graphs


class SampleTableMissing(AttributeError):
    """WMRB needs the static negative table: build the model with generate_sample=True
    (the reference fails with AttributeError inside gather_matrix_indices, matrix_factorization.py:153)."""


def _as_interactions(x):
    if isinstance(x, SparseInteractions):
        return x
    if hasattr(x, 'indices') and hasattr(x, 'values') and hasattr(x, 'dense_shape'):
        return SparseInteractions(x.indices, x.values, tuple(int(d) for d in x.dense_shape))
    from .input_utils import convert_to_sparse
    return convert_to_sparse(x)


class MatrixFactorization:
    """Standard matrix factorization with pluggable embedding / loss / initializer graphs."""

    def __init__(self, n_components, user_repr_graph=LinearEmbedding(), item_repr_graph=LinearEmbedding(),
                 loss_graph=MSELoss(), user_weight_graph=NormalInitializer(), item_weight_graph=NormalInitializer(),
                 n_users=None, n_items=None, n_samples=None, generate_sample=False):
        self.n_components = n_components
        self.user_repr_graph = user_repr_graph
        self.item_repr_graph = item_repr_graph
        self.loss_graph = loss_graph
        self.user_weight_graph = user_weight_graph
        self.item_weight_graph = item_weight_graph

        self.n_users = n_users
        self.n_items = n_items
        self.n_samples = n_samples
        self.random_ind = None
        self.generate_sample = generate_sample
        if n_samples is None and n_items is not None:  # :68-69
            self.n_samples = n_items // 2
        if generate_sample == True:  # noqa: E712  (:72-73; the table is drawn once and never resampled)
            if n_users * n_items > HOST_SAMPLER_MAX_WORK and torch.cuda.is_available():
                # the reference's host loop (one O(n_items) np.random.choice per user) would take minutes here:
                # draw the table on the device instead - same distribution, not the NumPy stream
                warnings.warn(f'random_sampler: {n_users} users x {n_items} items is too large for the host loop; '
                              'drawing the negative table on the device (utils.random_sampler_device)')
                self.random_ind = random_sampler_device(n_items, n_users, self.n_samples)
            else:
                self.random_ind = random_sampler(n_items, n_users, self.n_samples)

        if isinstance(self.user_repr_graph, ReLUEmbedding):  # :76-79
            self.user_aux_dim = 5 * self.n_components
        if isinstance(self.item_repr_graph, ReLUEmbedding):
            self.item_aux_dim = 5 * self.n_components
        self.user_relu_bias = None
        self.user_relu_weight = None
        self.item_relu_bias = None
        self.item_relu_weight = None
        self.user_linear_bias = None
        self.item_linear_bias = None
        self.user_trainable = None
        self.item_trainable = None

        self.loss_history_ = []   # extension: mean loss of every epoch of the last fit
        self.fit_seconds_ = 0.0   # extension: time spent in the epoch loop of the last fit
        self.plan_seconds_ = 0.0  # extension: time spent building the index structures of the last fit
        self.verbose = True
        self.factor_dtype = torch.float32  # extension: torch.bfloat16 = bf16 factor storage, fp32 arithmetic
        self.predict_arithmetic = None     # extension: 'auto' (default: 'split' or 'fp32', both on whole fp32 factors) | 'fp32' | 'split' | 'half2' (opt-in, 22 bits) - _ops.predict_topk
        self.data_parallel = False         # extension: split the users over torch.distributed ranks (teamoflow_amd/dist.py)
        # extension: q >= 1 = item-row-sharded V in q windows per rank (dist.fit_item_sharded): the item table is owned in
        # row blocks and streamed window by window instead of being replicated - for catalogs beyond one GPU's memory
        self.shard_items = 0
        # extension, OFF by default (the reference is full-batch): B > 0 = one optimiser step per batch of B users
        # (teamoflow_amd/_minibatch.py)
        self.batch_users = 0
        # extension, OFF by default: 'adam' keeps Adam's moments across epochs.  The reference (and the default here,
        # 'fresh_adam') builds a new optimizer every epoch (:176), i.e. every step is Adam's first step.
        self.optimizer = 'fresh_adam'

    # ------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------
    def _on_fast_path(self, user_features, item_features):
        return (type(self.user_repr_graph) is LinearEmbedding and type(self.item_repr_graph) is LinearEmbedding
                and type(self.loss_graph) in (MSELoss, WMRBLoss)
                and is_indicator(user_features) and is_indicator(item_features))

    def fit(self, epochs, user_features, item_features, tf_interactions, lr=1e-2):
        """matrix_factorization.py:96-187.  Re-initialises the weights on every call, runs ``epochs``
        full-batch steps (loss -> gradient of the SUM of the per-interaction losses -> a fresh Adam
        step), then stores user_embedding / item_embedding / *_trainable.  Returns None."""
        n_users, n_user_features = user_features.shape
        n_items, n_item_features = item_features.shape
        if not isinstance(self.user_repr_graph, ReLUEmbedding):
            U = self.user_weight_graph.initialize_weights(n_user_features, self.n_components)
        else:
            U = self.user_weight_graph.initialize_weights(self.user_aux_dim, self.n_components)
        if not isinstance(self.item_repr_graph, ReLUEmbedding):
            V = self.item_weight_graph.initialize_weights(n_item_features, self.n_components)
        else:
            V = self.item_weight_graph.initialize_weights(self.item_aux_dim, self.n_components)
        interactions = _as_interactions(tf_interactions)
        if self._on_fast_path(user_features, item_features):
            self._fit_sparse(epochs, n_users, n_items, interactions, lr, U, V)
        else:
            self._fit_generic(epochs, user_features, item_features, interactions, lr, U, V)

    def _report(self, epoch, loss, seconds):
        if self.verbose and (epoch + 1) % 25 == 0:
            print(f'Epoch {epoch + 1} Complete | Loss {loss} | Runtime {seconds:.5} s')

    def _fit_sparse(self, epochs, n_users, n_items, interactions, lr, U0, V0):
        _lib.get()  # fail loudly here when the HIP engine cannot run
        self._sharded_epoch = None
        dev = default_device()
        t_plan = timeit.default_timer()
        if interactions.device != dev:
            interactions = interactions.to(dev)
        wmrb = isinstance(self.loss_graph, WMRBLoss)
        if wmrb and self.random_ind is None:
            raise SampleTableMissing('WMRBLoss needs generate_sample=True (random_ind is None)')
        if getattr(self, 'batch_users', 0):
            if getattr(self, 'shard_items', 0) or self.data_parallel:
                raise ValueError('batch_users cannot be combined with shard_items / data_parallel')
            from .. import _minibatch
            _minibatch.fit_minibatch(self, epochs, n_users, n_items, interactions, lr, U0, V0, self.batch_users)
            return
        if getattr(self, 'shard_items', 0):
            from .. import dist as tdist
            tdist.fit_item_sharded(self, epochs, n_users, n_items, interactions, lr, U0, V0, windows_per_rank=int(self.shard_items))
            return
        if self.data_parallel and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1 or (self.data_parallel == 'force'):
            from .. import dist as tdist
            tdist.fit_data_parallel(self, epochs, n_users, n_items, interactions, lr, U0, V0)
            return
        ld = _lib.padded_ld(self.n_components, self.factor_dtype)
        plan = _engine.InteractionPlan(interactions.indices, interactions.values, n_users, n_items,
                                       user_chunks=1 if wmrb else _engine.mse_user_chunks(), csc=not wmrb)
        wplan, c = None, 0.0
        if wmrb:
            if self.random_ind is None:
                raise SampleTableMissing('WMRBLoss needs generate_sample=True (random_ind is None)')
            R = torch.as_tensor(self.random_ind).to(device=dev, dtype=torch.int32).contiguous()
            if R.dim() != 2 or R.shape[0] != n_users:
                raise ValueError(f'random_ind has shape {tuple(R.shape)}, expected [{n_users}, n_samples]')
            if R.numel() and (int(R.min()) < 0 or int(R.max()) >= n_items):
                raise IndexError('random_ind holds item ids outside [0, n_items)')
            c = self.n_items / self.n_samples  # constructor ints, true division (:167)
            wplan = _engine.wmrb_plan_for(plan, R, self.n_components, self.factor_dtype)
        st = _engine.TrainState(U0, V0, plan, self.n_components, wplan, dtype=self.factor_dtype)
        adam = _engine.adam_constants(lr)
        loss_sums = torch.zeros(max(epochs, 1), dtype=torch.float64, device=dev)
        denom = plan.n_pos if wmrb else plan.nnz
        self.loss_history_ = []
        if self.optimizer not in ('fresh_adam', 'adam'):
            raise ValueError(f"optimizer={self.optimizer!r}: 'fresh_adam' (the reference's behaviour) or 'adam'")
        persistent = self.optimizer == 'adam'
        if persistent:
            if self.factor_dtype is not torch.float32:
                raise ValueError("optimizer='adam' (persistent moments) needs float32 factor tables")
            lib = _lib.get()
            gU, gV = torch.empty_like(st.U), torch.empty_like(st.V)
            mom = [torch.zeros_like(st.U), torch.zeros_like(st.U), torch.zeros_like(st.V), torch.zeros_like(st.V)]

        def run_epoch(epoch, out):
            if persistent:  # raw gradients of both sides from the pre-update tables, then one Adam step with state, in place
                a = lib.tmf_adam_step(float(lr), epoch + 1)
                if wmrb:
                    _engine.epoch_wmrb(st, a, c, out, _lib.EPI_GRAD, gV, None, _lib.EPI_GRAD, gU)
                else:
                    _engine.epoch_mse(st, a, out, _lib.EPI_GRAD, gV, None, _lib.EPI_GRAD, gU)
                for W, G, M, V2 in ((st.U, gU, mom[0], mom[1]), (st.V, gV, mom[2], mom[3])):
                    _lib.check(lib.tmf_adam_state_rows_f32(_lib.ptr(W), _lib.ptr(G), _lib.ptr(M), _lib.ptr(V2), W.shape[0],
                                                           self.n_components, a, _lib.stream_ptr()), lib)
                return
            if wmrb:
                _engine.epoch_wmrb(st, adam, c, out)
            else:
                _engine.epoch_mse(st, adam, out)
            st.swap()

        # Launch-bound problems (a few hundred microseconds of kernels per epoch): capture an even number of
        # epochs into one hipGraph and replay it - the per-launch host cost disappears from the loop.
        work = plan.nnz + (plan.n_users * wplan.S if wmrb else 0)
        G = min(epochs - epochs % 2, GRAPH_EPOCHS)
        use_graph = G >= 4 and work <= GRAPH_MAX_WORK and os.environ.get('TMF_NO_GRAPH') is None and not persistent
        torch.cuda.synchronize(dev)
        t0 = timeit.default_timer()
        self.plan_seconds_ = t0 - t_plan  # extension: index structures + table set-up of this fit (once, not per epoch)
        done = 0
        if use_graph:
            block = torch.zeros(G, dtype=torch.float64, device=dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for e in range(G):
                    run_epoch(e, block[e:e + 1])
            # capture only records; the tables are still the initial ones (G is even: buffers line up again)
            t_prev = timeit.default_timer() - t0
            while done + G <= epochs:
                graph.replay()
                loss_sums[done:done + G].copy_(block)
                done += G
                if self.verbose and denom:
                    t_now = None
                    for e in range(done - G, done):
                        if (e + 1) % 25 == 0:
                            if t_now is None:
                                float(loss_sums[e])  # syncs: the replay has finished
                                t_now = timeit.default_timer() - t0
                            # cumulative runtime at epoch e, interpolated inside this replay
                            self._report(e, float(loss_sums[e]) / denom, t_prev + (t_now - t_prev) * (e + 1 - (done - G)) / G)
                    if t_now is not None:
                        t_prev = t_now
        for epoch in range(done, epochs):
            run_epoch(epoch, loss_sums[epoch:epoch + 1])
            if self.verbose and (epoch + 1) % 25 == 0:
                loss = float(loss_sums[epoch]) / denom if denom else float('nan')  # syncs
                self._report(epoch, loss, timeit.default_timer() - t0)
        torch.cuda.synchronize(dev)
        self.fit_seconds_ = timeit.default_timer() - t0
        sums = loss_sums[:epochs].cpu().numpy()
        self.loss_history_ = (sums / denom if denom else np.full(epochs, np.nan)).tolist()
        self._state = st
        r = self.n_components
        self.user_embedding = st.U[:, :r]
        self.item_embedding = st.V[:, :r]
        self.user_trainable = [self.user_embedding]
        self.item_trainable = [self.item_embedding]

    def _fit_generic(self, epochs, user_features, item_features, interactions, lr, U, V):
        """The reference's dense loop (:128-187) over arbitrary plug-ins, differentiated by autograd."""
        self._sharded_epoch = None
        dev = U.device
        interactions = interactions.to(dev)
        idx = interactions.indices
        alpha, omb1, omb2, eps = (float(x) for x in _adam_scalars(lr))
        self.loss_history_ = []
        cumulative_time = 0.0
        random_ind = None if self.random_ind is None else torch.as_tensor(self.random_ind).to(dev)
        for epoch in range(epochs):
            start = timeit.default_timer()
            user_embedding, self.user_trainable = self.user_repr_graph.get_repr(
                features=user_features, weights=U, relu_weight=self.user_relu_weight, relu_bias=self.user_relu_bias,
                linear_bias=self.user_linear_bias)
            item_embedding, self.item_trainable = self.item_repr_graph.get_repr(
                features=item_features, weights=V, relu_weight=self.item_relu_weight, relu_bias=self.item_relu_bias,
                linear_bias=self.item_linear_bias)
            self._keep_aux_variables()
            predictions = user_embedding @ item_embedding.T
            tf_sample_predictions = tf_prediction_serial = None
            if isinstance(self.loss_graph, WMRBLoss):
                if random_ind is None:
                    raise SampleTableMissing('WMRBLoss needs generate_sample=True (random_ind is None)')
                tf_sample_predictions = torch.gather(predictions, 1, random_ind.to(torch.int64))
                tf_prediction_serial = predictions[idx[:, 0], idx[:, 1]]
                predictions = None
            elif isinstance(self.loss_graph, KLDivergenceLoss):
                tf_prediction_serial = predictions[idx[:, 0], idx[:, 1]]
                predictions = None
            loss_fn = self.loss_graph.get_loss(tf_interactions=interactions, tf_sample_predictions=tf_sample_predictions,
                                               tf_prediction_serial=tf_prediction_serial, predictions=predictions,
                                               n_items=self.n_items, n_samples=self.n_samples)
            variables = self.user_trainable + self.item_trainable
            grads = torch.autograd.grad(loss_fn.sum(), variables, allow_unused=True)
            with torch.no_grad():
                for w, g in zip(variables, grads):
                    if g is not None:  # fresh Adam, t = 1 (:176)
                        w -= ((g * omb1) * alpha) / (torch.sqrt((g * g) * omb2) + eps)
            cumulative_time += timeit.default_timer() - start
            loss_one_epoch = float(loss_fn.detach().mean())
            self.loss_history_.append(loss_one_epoch)
            self._report(epoch, loss_one_epoch, cumulative_time)
        self.fit_seconds_ = cumulative_time
        with torch.no_grad():
            self.user_embedding, self.user_trainable = self.user_repr_graph.get_repr(
                features=user_features, weights=U, relu_weight=self.user_relu_weight, relu_bias=self.user_relu_bias,
                linear_bias=self.user_linear_bias)
            self.item_embedding, self.item_trainable = self.item_repr_graph.get_repr(
                features=item_features, weights=V, relu_weight=self.item_relu_weight, relu_bias=self.item_relu_bias,
                linear_bias=self.item_linear_bias)

    def _keep_aux_variables(self):
        if isinstance(self.user_repr_graph, BiasedLinearEmbedding):
            _, self.user_linear_bias = self.user_trainable
        if isinstance(self.user_repr_graph, ReLUEmbedding):
            _, self.user_relu_weight, self.user_relu_bias = self.user_trainable
        if isinstance(self.item_repr_graph, BiasedLinearEmbedding):
            _, self.item_linear_bias = self.item_trainable
        if isinstance(self.item_repr_graph, ReLUEmbedding):
            _, self.item_relu_weight, self.item_relu_bias = self.item_trainable

    # ------------------------------------------------------------------------------------------
    # prediction and ranking
    # ------------------------------------------------------------------------------------------
    def predict(self, A=None):
        """:189-201.  All scores [n_users, n_items]; with A also the scores where A == 0 (row-major)."""
        ep = getattr(self, '_sharded_epoch', None)
        if ep is not None and ep.world > 1:
            raise NotImplementedError('item-row-sharded fit: this rank holds only its item rows, the dense [n_users, n_items] score '
                                      'matrix is not available; recall_at_k / precision_at_k / retrieve_user_recs rank over the '
                                      'windows (dist.sharded_top_items), dist.gather_item_embedding assembles the table where it fits')
        all_predictions = _ops.predict_gemm(self.user_embedding, self.item_embedding)
        if A is not None:
            A = torch.as_tensor(A).to(all_predictions.device)
            return all_predictions, all_predictions[A == 0]
        return all_predictions

    def predict_ranks(self, A):
        """:203-216.  Global descending ranking of the flattened unobserved predictions."""
        _, unobserved = self.predict(A)
        return torch.sort(unobserved, descending=True, stable=True)[1]

    def _user_blocks(self):
        m, n = self.user_embedding.shape[0], self.item_embedding.shape[0]
        rows = max(1, min(m, PREDICT_CHUNK_BYTES // (4 * max(n, 1))))
        return [(b, min(b + rows, m)) for b in range(0, m, rows)]

    def _top_items(self, k, clamp, users=None):
        """Top-k item ids (int32) for every user, scored block by block: the [m, n] matrix is only
        ever materialised one block of users at a time."""
        ep = getattr(self, '_sharded_epoch', None)
        if ep is not None and ep.world > 1:
            # item-row-sharded fit: item_embedding holds only this rank's rows - rank over the windows (a collective)
            from .. import dist as tdist
            top = tdist.sharded_top_items(self, k, clamp, users=users)
            return top[0] if users is not None else top
        if users is not None:
            scores = _ops.predict_gemm(self.user_embedding[users:users + 1], self.item_embedding)
            return _ops.topk_stable(scores, k, clamp_negatives=clamp)[0]
        if _ops.fused_topk_supported(self.user_embedding, self.item_embedding, k):
            return _ops.predict_topk(self.user_embedding, self.item_embedding, k, clamp_negatives=clamp,
                                     arithmetic=getattr(self, 'predict_arithmetic', None))
        out = []
        for b, e in self._user_blocks():
            scores = _ops.predict_gemm(self.user_embedding[b:e], self.item_embedding)
            out.append(_ops.topk_stable(scores, k, clamp_negatives=clamp))
        return torch.cat(out) if len(out) > 1 else out[0]

    def _hits_and_relevant(self, A, k):
        """hits[u] = #top-k items with a non-zero entry in A, relevant[u] = #entries of A > 0
        (:245-254).  A: dense [m, n] tensor, or SparseInteractions (extension for shapes whose dense
        table does not fit)."""
        top = self._top_items(k, clamp=True)
        if isinstance(A, SparseInteractions):
            A = A.to(top.device)
            m, n = A.dense_shape
            nz = A.values != 0
            # column first, mask second: masked ROW selection of a [nnz, 2] tensor is unreliable beyond ~6e7 rows on this
            # PyTorch-ROCm build (tools/torch_row_index_probe.py)
            keys = A.indices[:, 0][nz] * n + A.indices[:, 1][nz]
            if keys.numel() > 1 and not bool((keys[1:] >= keys[:-1]).all()):   # row-major input (the reference's format) is sorted already
                keys = torch.sort(keys)[0]
            q = torch.arange(m, device=top.device)[:, None] * n + top.to(torch.int64)
            pos = torch.clamp(torch.searchsorted(keys, q.reshape(-1)), max=max(keys.numel() - 1, 0))
            found = (keys[pos] == q.reshape(-1)).reshape(q.shape) if keys.numel() else torch.zeros_like(q, dtype=torch.bool)
            hits = found.sum(dim=1).to(torch.float32)
            relevant = torch.bincount(A.indices[:, 0][A.values > 0], minlength=m).to(torch.float32)
            return hits, relevant
        A = torch.as_tensor(A).to(device=top.device, dtype=torch.float32)
        res_top_k = gather_matrix_indices(A, top.to(torch.int64))
        hits = torch.count_nonzero(res_top_k, dim=1).to(torch.float32)
        relevant = torch.count_nonzero(A > 0.0, dim=1).to(torch.float32)
        return hits, relevant

    def recall_at_k(self, A, k=10, preserve_rows=False):
        """:218-269.  Per-user hits@k / #positives; the caller takes the mean."""
        hits, relevant = self._hits_and_relevant(A, k)
        if not preserve_rows:
            mask = relevant != 0.0
            return hits[mask] / relevant[mask]
        recall = hits / relevant
        return torch.where(torch.isnan(recall), torch.zeros_like(recall), recall)

    def precision_at_k(self, A, k=10, preserve_rows=False):
        """:271-304."""
        hits, relevant = self._hits_and_relevant(A, k)
        if not preserve_rows:
            return hits[relevant != 0.0] / k
        return hits / k

    def f1_at_k(self, A, k=10, beta=1.0):
        """:306-318 (the reference's formula, denominator beta^2 (p + r))."""
        prec = self.precision_at_k(A, k=k).mean()
        rec = self.recall_at_k(A, k=k).mean()
        return ((1 + beta ** 2) * prec * rec) / (beta ** 2 * (prec + rec))

    def _dcg_terms(self, dense_interactions):
        predictions = self.predict()
        m, n = predictions.shape
        ranks = _ops.topk_stable(predictions, n).to(torch.int64)
        A = torch.as_tensor(dense_interactions).to(device=predictions.device, dtype=torch.float32)
        numerator = torch.pow(2.0, gather_matrix_indices(A, ranks)) - 1.0
        order = torch.arange(1, n + 1, dtype=torch.float32, device=predictions.device)
        denominator = torch.log1p(order) / float(np.log(np.float32(2.0)))
        return numerator, denominator

    def dcg_at_k(self, dense_interactions, k=10):
        """:320-351."""
        numerator, denominator = self._dcg_terms(dense_interactions)
        return (numerator / denominator[None, :])[:, :k].sum(dim=1)

    def idcg_at_k(self, dense_interactions, k=10):
        """:353-384."""
        numerator, denominator = self._dcg_terms(dense_interactions)
        ideal = torch.sort(numerator, dim=1, descending=True, stable=True)[0]
        return (ideal / denominator[None, :])[:, :k].sum(dim=1)

    def ndcg_at_k(self, A, k=10, preserve_rows=False):
        """:386-413."""
        ndcg = self.dcg_at_k(A, k) / self.idcg_at_k(A, k)
        if not preserve_rows:
            A = torch.as_tensor(A).to(ndcg.device)
            return ndcg[torch.count_nonzero(A, dim=1) > 0]
        return torch.where(~torch.isnan(ndcg), ndcg, torch.zeros_like(ndcg))

    def retrieve_user_recs(self, user=None, k=None):
        """:416-438.  Item ids ranked by score (numpy int32, like tf.math.top_k(...).indices.numpy())."""
        num_items = self.item_embedding.shape[0]
        kk = num_items if k is None else k
        return self._top_items(kk, clamp=False, users=user).cpu().numpy()

    # ------------------------------------------------------------------------------------------
    # persistence
    # ------------------------------------------------------------------------------------------
    def save_model(self):
        """:440-462.  (config dict with the reference's display keys, results dict)."""
        dict_config = {'Latent Dimension': self.n_components, 'User Embedding': self.user_repr_graph,
                       'Item Embedding': self.item_repr_graph, 'Loss': self.loss_graph,
                       'User Initialization': self.user_weight_graph, 'Item Initialization': self.item_weight_graph,
                       'Number of Users': self.n_users, 'Number of Items': self.n_items,
                       'Number of Samples': self.n_samples, 'Generate Sample': self.generate_sample}
        dict_results = {'User Embedding': self.user_embedding, 'Item Embedding': self.item_embedding,
                        'User Variables': self.user_trainable, 'Item Variables': self.item_trainable}
        return dict_config, dict_results

    _DISPLAY_KEYS = {'Latent Dimension': 'n_components', 'User Embedding': 'user_repr_graph',
                     'Item Embedding': 'item_repr_graph', 'Loss': 'loss_graph',
                     'User Initialization': 'user_weight_graph', 'Item Initialization': 'item_weight_graph',
                     'Number of Users': 'n_users', 'Number of Items': 'n_items', 'Number of Samples': 'n_samples',
                     'Generate Sample': 'generate_sample'}

    def save(self, path, include_samples=True, allow_pickle=False):
        """Extension: save_model()'s two dicts plus the tables in ONE file (torch.save).  The built-in
        plug-ins are stored as plain data (class name + constructor state), so the file loads with
        ``torch.load(weights_only=True)``; a user-defined plug-in object can only be stored pickled
        (``allow_pickle=True`` here AND in ``load``).  ``include_samples=False`` leaves the [m, S] negative table out."""
        ep = getattr(self, '_sharded_epoch', None)
        if ep is not None and ep.world > 1:
            raise ValueError('item-row-sharded model: this rank holds users %s and %d of the item rows only - assemble the tables '
                             'first (dist.gather_user_embedding / gather_item_embedding) or save one file per rank from them'
                             % (self.user_block, int(self.item_rows.numel())))
        _save_to_disk(self, path, include_samples, allow_pickle)

    @classmethod
    def load(cls, path, device=None, allow_pickle=False):
        """Inverse of ``save``: a model ready for predict / recall_at_k / a further fit.  ``allow_pickle=True`` unpickles
        arbitrary objects from the file - only for files you wrote yourself."""
        return _load_from_disk(cls, path, device, allow_pickle)

    @classmethod
    def from_saved(cls, config):
        """:465-475: ``cls(**config)``.  Also accepts save_model()'s display-key dict (in the
        reference that raises TypeError - SURVEY.md §5)."""
        return cls(**{cls._DISPLAY_KEYS.get(k, k): v for k, v in config.items()})


_PLUGIN_KEYS = ('User Embedding', 'Item Embedding', 'Loss', 'User Initialization', 'Item Initialization')


def _builtin_plugins():
    from . import embedding_graphs, initializer_graphs, loss_graphs
    out = {}
    for mod in (embedding_graphs, initializer_graphs, loss_graphs):
        for name in dir(mod):
            obj = getattr(mod, name)
            if isinstance(obj, type) and obj.__module__ == mod.__name__ and not name.startswith('_'):
                out[name] = obj
    return out


def _encode_plugin(obj, allow_pickle):
    """Built-in plug-in -> {'plugin': class name, 'state': plain data}; anything else only as a pickled object."""
    cls = _builtin_plugins().get(type(obj).__name__)
    if cls is not None and type(obj) is cls:
        state = {}
        for k, v in vars(obj).items():
            if torch.is_tensor(v) or isinstance(v, np.ndarray):
                v = torch.as_tensor(v).detach().cpu()
            elif not isinstance(v, (int, float, bool, str, type(None))):
                raise TypeError(f'{type(obj).__name__}.{k} of type {type(v).__name__} cannot be stored as plain data')
            state[k] = v
        return {'plugin': type(obj).__name__, 'state': state}
    if not allow_pickle:
        raise TypeError(f'{type(obj).__name__} is not a built-in plug-in: pass allow_pickle=True to store it pickled')
    return {'pickled': obj}


def _decode_plugin(entry):
    if 'pickled' in entry:
        return entry['pickled']
    cls = _builtin_plugins()[entry['plugin']]
    obj = cls.__new__(cls)
    for k, v in entry['state'].items():
        setattr(obj, k, v)
    return obj


def _save_to_disk(model, path, include_samples=True, allow_pickle=False):
    config, results = model.save_model()
    config = {k: (_encode_plugin(v, allow_pickle) if k in _PLUGIN_KEYS else v) for k, v in config.items()}
    blob = {'format': 'teamoflow_amd.mf/2', 'config': config,
            'user_embedding': None if model.user_embedding is None else model.user_embedding.detach().float().cpu(),
            'item_embedding': None if model.item_embedding is None else model.item_embedding.detach().float().cpu(),
            'factor_dtype': str(model.factor_dtype).replace('torch.', ''),
            'optimizer': getattr(model, 'optimizer', 'fresh_adam'),
            'loss_history': [float(x) for x in (getattr(model, 'loss_history_', []) or [])],
            'random_ind': (torch.as_tensor(model.random_ind).cpu() if include_samples and model.random_ind is not None else None)}
    torch.save(blob, path)


def _load_from_disk(cls, path, device=None, allow_pickle=False):
    blob = torch.load(path, map_location='cpu', weights_only=not allow_pickle)
    if blob.get('format') == 'teamoflow_amd.mf/1':
        raise ValueError(f'{path}: saved by an older version of this package (format /1, pickled plug-ins); load it with that '
                         'version and save it again - the current format /2 stores plain data only')
    if blob.get('format') != 'teamoflow_amd.mf/2':
        raise ValueError(f'{path}: not a teamoflow_amd model file')
    cfg = {k: (_decode_plugin(v) if k in _PLUGIN_KEYS else v) for k, v in blob['config'].items()}
    generate = cfg['Generate Sample']
    cfg['Generate Sample'] = False          # the table comes from the file (or is absent), never redrawn
    model = cls.from_saved(cfg)
    model.generate_sample = generate
    dev = default_device() if device is None else torch.device(device)
    dt = getattr(torch, blob['factor_dtype'])
    model.factor_dtype = dt
    model.optimizer = blob.get('optimizer', 'fresh_adam')
    for name in ('user_embedding', 'item_embedding'):
        t = blob[name]
        setattr(model, name, None if t is None else t.to(device=dev, dtype=dt))
    model.user_trainable = [model.user_embedding] if model.user_embedding is not None else None
    model.item_trainable = [model.item_embedding] if model.item_embedding is not None else None
    model.loss_history_ = blob['loss_history']
    if blob['random_ind'] is not None:
        model.random_ind = blob['random_ind'].to(dev)
    return model


def _adam_scalars(lr):
    f = np.float32
    one, b1, b2 = f(1.0), f(0.9), f(0.999)
    return f(f(lr) * np.sqrt(f(one - b2)) / f(one - b1)), f(one - b1), f(one - b2), f(1e-7)
