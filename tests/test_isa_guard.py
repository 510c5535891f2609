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
# every kernel with a gather loop, by source file (the list named k_wmrb_scores2 / k_wmrb_gradu2, which no longer
# exist - the dominant slice kernels went unguarded while `checked >= 10` still passed on the others).  k_wmrb_scores3 with rows of
# 32 lanes contains the lean walk (slice_scores_lean) next to the general one.
GATHER_KERNELS = {'tmf_train.hip': ('k_mse_pass', 'k_wsum_pass', 'k_wsum_pass_pg'),
                  'tmf_wmrb.hip': ('k_wmrb_user', 'k_wmrb_scores3', 'k_wmrb_gradu3')}


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
        if line.startswith('.Lfunc_end'):   # not s_endpgm: a kernel may hold several (early exits)
            name = None
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src', ['tmf_train.hip', 'tmf_wmrb.hip'])
def test_gather_loops_keep_their_row_loads_in_flight(src, tmp_path):
    asm = tmp_path / (src + '.s')
    subprocess.run([HIPCC, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=on', '-S', '--cuda-device-only',
                    os.path.join(CSRC, src), '-o', str(asm)], check=True, stderr=subprocess.DEVNULL)
    stats = _immediately_waited(asm.read_text())
    found = {k: 0 for k in GATHER_KERNELS[src]}
    for sym, (loads, waited) in stats.items():
        # the kernel's own name as mangled (length-prefixed), so that k_wsum_pass does not also match k_wsum_pass_pg
        name = next((k for k in GATHER_KERNELS[src] if f'_ZN3tmf{len(k)}{k}I' in sym), None)
        if name is None or 'ILi1E' in sym:
            continue
        found[name] += 1
        if any(f'{name}ILi{g}E' in sym for g in (16, 32, 64)):   # the wide-row geometries the benchmarks run: an unrolled gather loop is there
            assert loads >= 4, f'{sym}: only {loads} row loads found - is this still the gather kernel?'
        assert waited <= 1, f'{sym}: {waited} of {loads} row loads are waited for immediately (serialised gathers)'
    # every named kernel was actually found, in the fp32 and the bf16 instantiations of several row widths
    assert all(v >= 6 for v in found.values()), found
    if src == 'tmf_wmrb.hip':   # the lean scores walk: rows of 32 lanes, four waves and eight, both storage types
        lean = [s for s in stats if '_ZN3tmf14k_wmrb_scores3ILi32E' in s]
        assert len(lean) >= 4, lean
