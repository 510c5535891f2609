"""The C ABI used by a compiled caller (tests/cabi/cabi_fit.cpp): no Python and no torch in that process - hipMalloc'd buffers,
tmf_csr_build / tmf_csc_perm, MSE epochs through tmf_mse_pass_f32, the loss through tmf_sum_f32, fused predict + top-k, WMRB
epochs through the sliced kernels and the item-side gather-sum, and the error path, each checked inside the program against its own fp64 restatement."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
EXE = os.path.join(ROOT, 'tests', 'cabi', 'cabi_fit')


@pytest.mark.gpu
def test_c_abi_from_a_compiled_caller():
    if not os.path.exists(EXE):   # built by __graft_entry__.build(); hipcc is on the GPU box as well
        subprocess.run(['make', '-C', os.path.dirname(EXE)], check=True)
    res = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode()
    assert res.returncode == 0 and 'PASS' in out, out


def test_compiled_caller_builds_against_the_header():
    """CPU: the caller compiles and links against include/tmf.h + libtmf.so (hipcc cross-compiles without a GPU)."""
    subprocess.run(['make', '-C', os.path.dirname(EXE)], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)
