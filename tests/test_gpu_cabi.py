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


HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
LIB = os.path.join(ROOT, 'teamoflow_amd', 'libtmf.so')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='build test: needs hipcc')
@pytest.mark.skipif(not os.path.exists(LIB), reason='build test: needs libtmf.so (built by __graft_entry__.build())')
def test_compiled_caller_builds_against_the_header(tmp_path):
    """CPU: the caller compiles and links against include/tmf.h + the EXISTING libtmf.so (hipcc cross-compiles without a GPU).
    Built into a temporary directory: the test neither rebuilds the library nor touches the source tree."""
    out = tmp_path / 'cabi_fit'
    subprocess.run([HIPCC, '-O2', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'cabi', 'cabi_fit.cpp'),
                    '-o', str(out), '-L' + os.path.dirname(LIB), '-ltmf', '-Wl,-rpath,' + os.path.dirname(LIB)],
                   check=True, stdout=subprocess.DEVNULL)
    assert out.exists()
