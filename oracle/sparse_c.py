"""ctypes front of oracle/sparse_ref.c - the C/OpenMP restatement of one training epoch.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) - parity unpinned except ``gather_matrix_indices``.
Same call shape as ``oracle.sparse_ref.mse_epoch`` / ``wmrb_epoch`` so the tests can swap one for the
other; ``Plan`` holds the per-fit index structures (built once, outside any timed epoch)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SPARSE_LIB: another build of the same file (the sanitized one of `make -C oracle asan`)
LIB_PATH = os.environ.get('ORACLE_SPARSE_LIB') or os.path.join(HERE, 'liboracle_sparse.so')
_lib = None


class Adam(ctypes.Structure):
    _fields_ = [('alpha', ctypes.c_float), ('one_m_b1', ctypes.c_float), ('one_m_b2', ctypes.c_float), ('eps', ctypes.c_float)]


def build():
    subprocess.run(['make', '-C', HERE, 'liboracle_sparse.so'], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def adam_constants(lr):
    f = np.float32
    one, b1, b2 = f(1.0), f(0.9), f(0.999)
    return Adam(f(f(lr) * np.sqrt(f(one - b2)) / f(one - b1)), f(one - b1), f(one - b2), f(1e-7))


def threads():
    return int(lib().oracle_threads())


def set_threads(n):
    lib().oracle_set_threads(int(n))


class Plan:
    """CSR by user + grouping by item of the interactions (+ of the negative table R for WMRB)."""

    def __init__(self, indices, values, m, n, R=None):
        L = lib()
        indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
        order = np.lexsort((indices[:, 1], indices[:, 0]))      # row-major, as the reference's SparseTensor is
        self.order = order
        self.m, self.n, self.nnz = int(m), int(n), len(order)
        self.ent_user = np.ascontiguousarray(indices[order, 0], dtype=np.int32)
        self.col = np.ascontiguousarray(indices[order, 1], dtype=np.int32)
        self.val = np.ascontiguousarray(np.asarray(values, dtype=np.float32)[order])
        self.rowptr = np.zeros(self.m + 1, dtype=np.int64)
        np.cumsum(np.bincount(self.ent_user, minlength=self.m), out=self.rowptr[1:])
        self.colptr = np.zeros(self.n + 1, dtype=np.int64)
        self.centry = np.zeros(max(self.nnz, 1), dtype=np.int64)
        assert L.oracle_group_by_key(ctypes.c_int64(self.nnz), ctypes.c_int64(self.n), _p(self.col), _p(self.colptr),
                                     _p(self.centry)) == 0
        self.R = None
        if R is not None:
            self.R = np.ascontiguousarray(R, dtype=np.int32)
            self.S = int(self.R.shape[1])
            self.negptr = np.zeros(self.n + 1, dtype=np.int64)
            self.negentry = np.zeros(max(self.R.size, 1), dtype=np.int64)
            assert L.oracle_group_by_key(ctypes.c_int64(self.R.size), ctypes.c_int64(self.n), _p(self.R), _p(self.negptr),
                                         _p(self.negentry)) == 0
            self.D = np.zeros((self.m, self.S), dtype=np.float32)
        self.delta = np.zeros(max(self.nnz, 1), dtype=np.float32)


def mse_epoch(U, V, plan, lr, want_grads=True):
    """-> (U_new, V_new, mean_loss, terms{delta (input order), gU, gV, loss_sum})."""
    L = lib()
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    r = U.shape[1]
    Un, Vn = np.empty_like(U), np.empty_like(V)
    gU = np.empty_like(U) if want_grads else None
    gV = np.empty_like(V) if want_grads else None
    loss = ctypes.c_double(0.0)
    adam = adam_constants(lr)
    rc = L.oracle_mse_epoch(ctypes.c_int64(plan.m), ctypes.c_int64(plan.n), ctypes.c_int(r), _p(plan.rowptr), _p(plan.col),
                            _p(plan.val), _p(plan.ent_user), _p(plan.colptr), _p(plan.centry), _p(U), _p(V),
                            ctypes.byref(adam), _p(Un), _p(Vn), ctypes.byref(loss), _p(plan.delta), _p(gU), _p(gV))
    assert rc == 0
    delta = np.empty(plan.nnz, dtype=np.float32)
    delta[plan.order] = plan.delta[:plan.nnz]
    mean = loss.value / plan.nnz if plan.nnz else float('nan')
    return Un, Vn, mean, dict(delta=delta, gU=gU, gV=gV, loss_sum=loss.value)


def wmrb_epoch(U, V, plan, n_items, n_samples, lr, want_grads=True):
    """-> (U_new, V_new, mean loss over positives, terms{delta (input order, 0 on non-positives), D, gU, gV})."""
    L = lib()
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    r = U.shape[1]
    Un, Vn = np.empty_like(U), np.empty_like(V)
    gU = np.empty_like(U) if want_grads else None
    gV = np.empty_like(V) if want_grads else None
    loss, npos = ctypes.c_double(0.0), ctypes.c_int64(0)
    adam = adam_constants(lr)
    rc = L.oracle_wmrb_epoch(ctypes.c_int64(plan.m), ctypes.c_int64(plan.n), ctypes.c_int(r), ctypes.c_int64(plan.S),
                             ctypes.c_float(np.float32(n_items / n_samples)), _p(plan.rowptr), _p(plan.col), _p(plan.val),
                             _p(plan.ent_user), _p(plan.colptr), _p(plan.centry), _p(plan.R), _p(plan.negptr),
                             _p(plan.negentry), _p(U), _p(V), ctypes.byref(adam), _p(Un), _p(Vn), ctypes.byref(loss),
                             ctypes.byref(npos), _p(plan.delta), _p(plan.D), _p(gU), _p(gV))
    assert rc == 0
    delta = np.empty(plan.nnz, dtype=np.float32)
    delta[plan.order] = plan.delta[:plan.nnz]
    mean = loss.value / npos.value if npos.value else float('nan')
    return Un, Vn, mean, dict(delta=delta, D=plan.D.copy(), gU=gU, gV=gV, loss_sum=loss.value, n_pos=npos.value)


def wmrb_boundary_slack(U, V, plan, n_items, n_samples, tol_rel=None):
    """How far D / delta / gU / gV may move when hinge terms whose argument lies within ``tol_rel`` (the tolerance the
    predictions are compared at) of the kink switch between active and inactive - see oracle_wmrb_boundary_slack in
    sparse_ref.c.  -> dict(D, delta (input order), gU, gV, pairs)."""
    L = lib()
    L.oracle_wmrb_boundary_slack.restype = ctypes.c_int64
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    r = U.shape[1]
    if tol_rel is None:
        tol_rel = max(1e-5, 2 * r * 2.0 ** -24)
    D = np.zeros((plan.m, plan.S), dtype=np.float32)
    dl = np.zeros(max(plan.nnz, 1), dtype=np.float32)
    gU, gV = np.zeros_like(U), np.zeros_like(V)
    pairs = L.oracle_wmrb_boundary_slack(ctypes.c_int64(plan.m), ctypes.c_int64(plan.n), ctypes.c_int(r), ctypes.c_int64(plan.S),
                                         ctypes.c_float(np.float32(n_items / n_samples)), ctypes.c_float(tol_rel), _p(plan.rowptr),
                                         _p(plan.col), _p(plan.val), _p(plan.R), _p(U), _p(V), _p(D), _p(dl), _p(gU), _p(gV))
    assert pairs >= 0
    delta = np.empty(plan.nnz, dtype=np.float32)
    delta[plan.order] = dl[:plan.nnz]
    return dict(D=D, delta=delta, gU=gU, gV=gV, pairs=int(pairs))

