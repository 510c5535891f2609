"""Guard against a code-generation regression that cost 25 % of the C4 item pass and 20 % of every bf16 epoch before it
was noticed: the compiler placing ``s_waitcnt vmcnt(0)`` directly behind each unrolled row load of a gather loop (one
row in flight per lane group instead of four).  The kernels are cross-compiled to gfx950 assembly (no GPU needed) and
the row loads (``global_load_dwordx4``) of the gather kernels must not be immediately waited for.

The single-lane geometries (G = 1: n_components <= 4 in fp32, <= 8 in bf16) are exempt - their rows are one load."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CSRC = os.path.join(ROOT, 'teamoflow_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
GATHER_KERNELS = ('k_mse_pass', 'k_wsum_pass', 'k_wmrb_user', 'k_wmrb_scores2', 'k_wmrb_gradu2')


def _immediately_waited(asm_text):
    """{kernel symbol: (row loads, row loads whose next instructions include s_waitcnt vmcnt(0))}"""
    lines = asm_text.split('\n')
    out, name = {}, None
    for i, line in enumerate(lines):
        m = re.match(r'^(_ZN3tmf\w+):', line)
        if m:
            name = m.group(1)
            out[name] = [0, 0]
        if name and 'global_load_dwordx4' in line:
            out[name][0] += 1
            nxt = [x for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(';')][:2]
            if any('s_waitcnt vmcnt(0)' in x for x in nxt):
                out[name][1] += 1
        if 's_endpgm' in line:
            name = None
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src', ['tmf_train.hip', 'tmf_wmrb.hip'])
def test_gather_loops_keep_their_row_loads_in_flight(src, tmp_path):
    asm = tmp_path / (src + '.s')
    subprocess.run([HIPCC, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=on', '-S', '--cuda-device-only',
                    os.path.join(CSRC, src), '-o', str(asm)], check=True, stderr=subprocess.DEVNULL)
    stats = _immediately_waited(asm.read_text())
    checked = 0
    for sym, (loads, waited) in stats.items():
        if not any(k in sym for k in GATHER_KERNELS) or 'ILi1E' in sym:
            continue
        checked += 1
        assert waited <= 1, f'{sym}: {waited} of {loads} row loads are waited for immediately (serialised gathers)'
    assert checked >= 10
