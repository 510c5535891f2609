"""ctypes binding of libtmf.so (C ABI in include/tmf.h).

The engine has no CPU implementation: every entry point needs the gfx950 code object and a
visible MI355X.  ``get()`` raises ``EngineUnavailable`` loudly instead of falling back to anything.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TMF_LIB', os.path.join(_HERE, 'libtmf.so'))
CSRC = os.path.join(_HERE, 'csrc')

EPI_ADAM, EPI_GRAD = 0, 1


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


class Adam(ctypes.Structure):
    _fields_ = [('alpha', ctypes.c_float), ('one_minus_b1', ctypes.c_float), ('one_minus_b2', ctypes.c_float),
                ('eps', ctypes.c_float)]


class SliceLists(ctypes.Structure):
    _fields_ = [('R_sorted', ctypes.c_void_p), ('slice_off', ctypes.c_void_p), ('rowptr', ctypes.c_void_p),
                ('col', ctypes.c_void_p), ('pos_off', ctypes.c_void_p), ('n_users', ctypes.c_int32),
                ('n_samples', ctypes.c_int32), ('n_slices', ctypes.c_int32), ('slice_begin', ctypes.c_int32),
                ('slice_count', ctypes.c_int32), ('item_base', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('n_items', ctypes.c_int32)]


class Segments(ctypes.Structure):
    _fields_ = [('rowptr', ctypes.c_void_p), ('seg_row', ctypes.c_void_p), ('seg_chunk', ctypes.c_void_p),
                ('seg_slab', ctypes.c_void_p), ('nseg', ctypes.c_int64), ('chunk', ctypes.c_int32),
                ('row_mod', ctypes.c_int32)]


_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_SEG = ctypes.POINTER(Segments)
_SL = ctypes.POINTER(SliceLists)
_I32, _SZ = ctypes.c_int32, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/tmf.h declares (tests check this)
SIGNATURES = {}
_BASE_SIGNATURES = {
    'tmf_version': (_I, []),
    'tmf_last_error': (ctypes.c_char_p, []),
    'tmf_padded_ld': (_I, [_I]),
    'tmf_padded_ld_bf16': (_I, [_I]),
    'tmf_adam_fresh': (Adam, [_F]),
    'tmf_csr_build_workspace_bytes': (_SZ, [_L]),
    'tmf_csr_build': (_I, [_P, _P, _L, _I32, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    'tmf_stable_order_workspace_bytes': (_SZ, [_L]),
    'tmf_csc_perm': (_I, [_P, _L, _I32, _P, _P, _P, _SZ, _P]),
    'tmf_stable_order_i32': (_I, [_P, _L, _L, _P, _P, _P, _P, _SZ, _P]),
    'tmf_sort_samples_workspace_bytes': (_SZ, [_I32, _I32]),
    'tmf_sort_samples': (_I, [_P, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    'tmf_slice_offsets': (_I, [_P, _P, _L, _I32, _I32, _I32, _P, _P]),
    'tmf_wmrb_entry_lists_workspace_bytes': (_SZ, [_L, _I32, _I32]),
    'tmf_wmrb_entry_lists': (_I, [_P, _P, _P, _L, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _SZ, _P]),
    'tmf_mse_pass_f32': (_I, [_SEG, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wsum_pass_f32': (_I, [_SEG, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wsum_rows4_rows_per_group': (_I, [_I, _I]),
    'tmf_wsum_rows4_workspace_bytes': (_SZ, [_I32, _I32, _I32]),
    'tmf_wsum_rows4_f32': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _I32, _P, _SZ, _P]),
    'tmf_wsum_rows5_workspace_bytes': (_SZ, [_I32, _I32, _I32]),
    'tmf_wsum_rows5_f32': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I, _I, Adam, _I32, _P, _SZ, _P]),
    'tmf_combine_rows_f32': (_I, [_P, _P, _L, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wmrb_user_pass_f32': (_I, [_P, _P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wmrb_user_pass_fits': (_I, [_I32, _I]),
    'tmf_wmrb_scores3_f32': (_I, [_SL, _P, _P, _P, _P, _I, _P]),
    'tmf_wmrb_scores6_users_per_group': (_I, []),
    'tmf_wmrb_scores6_supported': (_I, [_I, _I, _L]),
    'tmf_wmrb_scores6_f32': (_I, [_P, _P, _P, _L, _I32, _L, _L, _P, _P, _P, _P, _I, _P]),
    'tmf_wmrb_scores5_users_per_workgroup': (_I, []),
    'tmf_wmrb_scores5_supported': (_I, [_I, _I, _L]),
    'tmf_wmrb_scores5_workspace_bytes': (_SZ, [_L, _I32, _I]),
    'tmf_wmrb_scores5_f32': (_I, [_P, _P, _P, _L, _L, _L, _P, _P, _P, _P, _I, _I, _P, _I32, _I, _P, _SZ, _P]),
    'tmf_wmrb_hinge2': (_I, [_P, _P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P]),
    'tmf_wmrb_hinge2_ordered': (_I, [_P, _P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _P]),
    'tmf_wmrb_gradu3_f32': (_I, [_SL, _P, _P, _P, _P, _I, _I, _P]),
    'tmf_wmrb_gradu4_supported': (_I, [_I, _I]),
    'tmf_wmrb_gradu4_workspace_bytes': (_SZ, [_I32, _I32, _I32]),
    'tmf_wmrb_gradu4_f32': (_I, [_SL, _P, _P, _P, _P, _P, _I, _I, Adam, _I32, _P, _SZ, _P]),
    'tmf_wmrb_finish_f32': (_I, [_P, _I32, _I32, _P, _P, _I, _I, Adam, _P]),
    'tmf_adam_fresh_rows_f32': (_I, [_P, _P, _L, _I, Adam, _P]),
    'tmf_adam_step': (Adam, [_F, _I]),
    'tmf_adam_state_rows_f32': (_I, [_P, _P, _P, _P, _L, _I, Adam, _P]),
    'tmf_mse_pass_bf16': (_I, [_SEG, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wsum_rows4_bf16': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _I32, _P, _SZ, _P]),
    'tmf_wsum_rows5_bf16': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I, _I, Adam, _I32, _P, _SZ, _P]),
    'tmf_wsum_pass_bf16': (_I, [_SEG, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_combine_rows_bf16': (_I, [_P, _P, _L, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_wmrb_user_pass_bf16': (_I, [_P, _P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, Adam, _P]),
    'tmf_adam_fresh_rows_bf16': (_I, [_P, _P, _L, _I, Adam, _P]),
    'tmf_sum_f32': (_I, [_P, _L, _P, _P]),
    'tmf_gather_rows_cols_f32': (_I, [_P, _P, _P, _L, _L, _L, _P]),
    'tmf_predict_gemm_f32': (_I, [_P, _P, _P, _L, _L, _I, _L, _L, _L, _P]),
    'tmf_topk_workspace_bytes': (_SZ, [_L, _L, _I]),
    'tmf_topk_stable_f32': (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _P, _SZ, _P]),
    'tmf_predict_topk_f32': (_I, [_P, _P, _L, _L, _I, _L, _L, _I, _I, _P, _P, _P]),
    'tmf_predict_topk_bf16': (_I, [_P, _P, _L, _L, _I, _L, _L, _I, _I, _P, _P, _P]),
    'tmf_predict_topk_split_supported': (_I, [_I, _I]),
    'tmf_predict_topk_split_workspace_bytes': (_SZ, [_L, _I]),
    'tmf_predict_topk_split_f32': (_I, [_P, _P, _L, _L, _I, _L, _L, _I, _I, _P, _P, _P, _SZ, _P]),
    'tmf_predict_topk_half2_supported': (_I, [_I, _I]),
    'tmf_predict_topk_half2_workspace_bytes': (_SZ, [_L, _I]),
    'tmf_predict_topk_half2_f32': (_I, [_P, _P, _L, _L, _I, _L, _L, _I, _I, _P, _P, _P, _SZ, _P]),
}

SIGNATURES.update(_BASE_SIGNATURES)
for _name in ('tmf_wmrb_scores3', 'tmf_wmrb_scores5', 'tmf_wmrb_scores6', 'tmf_wmrb_gradu3', 'tmf_wmrb_gradu4', 'tmf_wmrb_finish'):
    SIGNATURES[_name + '_bf16'] = SIGNATURES[_name + '_f32']

_lib = None
MIN_LIB_VERSION = 203   # include/tmf.h TMF_VERSION: 203 = tmf_wsum_rows5, tmf_slice_lists.flags (72-byte struct; n_items counts only with SLICE_N_ITEMS_STATED)
SLICE_XCD_MAJOR, SLICE_N_ITEMS_STATED = 1, 2   # tmf_slice_lists.flags


def build(force=False, verbose=False):
    """Compile libtmf.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.run(['make', '-C', CSRC, 'clean'], check=True, capture_output=not verbose)
    subprocess.run(['make', '-C', CSRC, '-j4'], check=True, capture_output=not verbose)
    return LIB_PATH


def load_library():
    """dlopen libtmf.so and attach the signatures.  Needs no GPU (used by the CPU symbol test).
    torch is imported first on purpose: its bundled libamdhip64 (SONAME libamdhip64.so.7) is then
    already in the process and libtmf's DT_NEEDED entry resolves to that same runtime, so torch's
    streams and allocations are valid inside the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(f'{LIB_PATH} not found - run `python -c "import __graft_entry__ as g; g.build()"` '
                                f'or `make -C {CSRC}`')
    lib = ctypes.CDLL(LIB_PATH)
    older = os.environ.get('TMF_LIB_OLDER') == '1'   # A/B runs against a library built from an older commit (tools/c4_ab.sh)
    # the version first: a stale library then says so instead of failing on the first symbol it lacks
    try:
        lib.tmf_version.restype, lib.tmf_version.argtypes = SIGNATURES['tmf_version']
        version = lib.tmf_version()
    except AttributeError:
        version = 0
    if version < MIN_LIB_VERSION and not older:
        raise EngineUnavailable(f'{LIB_PATH} is version {version}, this package needs {MIN_LIB_VERSION} or newer (struct layouts and '
                                f'entry points changed) - rebuild it: `make -C {CSRC}`')
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if older:
                continue
            raise EngineUnavailable(f'{LIB_PATH} (version {version}) lacks {name} - rebuild it: `make -C {CSRC}`') from None
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def get():
    """The library, ready to launch on the current torch device.  Raises if there is no GPU."""
    if not torch.cuda.is_available():
        raise EngineUnavailable('teamoflow_amd needs an MI355X: torch.cuda.is_available() is False and the '
                                'engine has no CPU fallback')
    return load_library()


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        raise EngineError(f'libtmf error {rc}: {lib.tmf_last_error().decode()}')


def padded_ld(n_components, dtype=None):
    """Mirror of tmf_padded_ld / tmf_padded_ld_bf16, usable without loading the library."""
    r = int(n_components)
    if r < 1 or r > 1024:
        raise ValueError(f'n_components={r} outside the supported range [1, 1024]')
    if dtype is torch.bfloat16:
        if r > 512:
            return 1024
        lanes, g = (r + 7) // 8, 1
        while g < lanes:
            g *= 2
        return 8 * g
    if r <= 256:
        lanes, g = (r + 3) // 4, 1
        while g < lanes:
            g *= 2
        return 4 * g
    nv = (r + 255) // 256
    if nv == 3:
        nv = 4
    return 256 * nv
