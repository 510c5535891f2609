"""The yardstick under sanitizers (CPU box only - never on the GPU box).  Every full-size parity claim (C3 at full size, the
1.1M-user grid test, bench.py's cpu_baseline) rests on oracle/sparse_ref.c - OpenMP C with 64-bit index arithmetic over up to
1e9 list entries - and the compiled ABI caller's checks on tests/cabi/host_ref.h.  Both are rebuilt here with
-fsanitize=address,undefined and exercised in child processes: a heap overflow, a use-after-free, a signed overflow or a
misaligned access in the oracle would fail this test instead of quietly bending a parity verdict."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def asan_runtime():
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    path = subprocess.run([gcc, '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip('no libasan for this gcc')
    return os.path.realpath(path)


def test_c_oracle_cases_under_asan_and_ubsan():
    """tests/test_oracle.py's C-oracle cases (golden fixtures, random shapes x thread counts, boundary slack) against
    liboracle_sparse_asan.so, in a child Python with libasan preloaded (leak checking off: CPython itself 'leaks')."""
    rt = asan_runtime()
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan'], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rt, ORACLE_SPARSE_LIB=os.path.join(ROOT, 'oracle', 'liboracle_sparse_asan.so'),
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
               OMP_NUM_THREADS='4', TMF_ORACLE_SANITIZED='1')
    p = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_oracle.py'), '-x', '-q', '-p', 'no:cacheprovider',
                        '-k', 'c_oracle or boundary_slack'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert 'passed' in p.stdout and 'AddressSanitizer' not in tail and 'runtime error' not in tail, tail


def test_the_sanitized_library_is_the_one_that_ran():
    """The child really loads the instrumented build (a wrong path would silently test the -O3 one)."""
    rt = asan_runtime()
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan'], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, 'oracle', 'liboracle_sparse_asan.so')
    env = dict(os.environ, LD_PRELOAD=rt, ORACLE_SPARSE_LIB=lib, ASAN_OPTIONS='detect_leaks=0')
    code = ('from oracle import sparse_c as C\nC.lib()\nmaps = open("/proc/self/maps").read()\n'
            'assert "liboracle_sparse_asan.so" in maps and "liboracle_sparse.so" not in maps and C.LIB_PATH.endswith("_asan.so")\n'
            'import numpy as np\n'
            'idx = np.array([[0, 1], [1, 0], [2, 2]]); val = np.ones(3, np.float32)\n'
            'p = C.Plan(idx, val, 3, 3)\n'
            'U = np.ones((3, 2), np.float32); V = np.ones((3, 2), np.float32)\n'
            'print(C.mse_epoch(U, V, p, 0.01, want_grads=False)[2])\n')
    p = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]


def test_cabi_host_restatement_under_asan_and_ubsan():
    """tests/cabi/host_ref.h - the fp64 restatement the compiled ABI caller (cabi_fit.cpp) judges the GPU by - built stand-alone
    with g++ under the sanitizers; its self-test checks the closed-form gradients against central differences of the loss and
    the top-k checker against planted violations."""
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    asan_runtime()
    d = os.path.join(ROOT, 'tests', 'cabi')
    subprocess.run(['make', '-C', d, 'host_ref_selftest'], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    p = subprocess.run([os.path.join(d, 'host_ref_selftest')], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, (p.stdout + p.stderr)[-3000:]
