"""Condenses a gpurun_out/prof_<tag>/ directory (see tools_profile.sh) into the small files kept under
profiles/: <tag>_kernel_stats.csv (top kernels) and <tag>_pmc.json (FETCH/WRITE per tmf kernel)."""
import collections
import csv
import glob
import json
import os
import sys

import shutil
import subprocess


def demangle(name, _cache={}):
    if name not in _cache:
        tool = shutil.which('llvm-cxxfilt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'
        try:
            _cache[name] = subprocess.run([tool, name], stdout=subprocess.PIPE, check=True).stdout.decode().strip()
        except Exception:
            import re
            m_ = re.match(r'_ZN3tmf(\d+)', name)   # no demangler: at least cut the kernel's own name out of the mangled one
            _cache[name] = ('tmf::' + name[len(m_.group(0)):len(m_.group(0)) + int(m_.group(1))]) if m_ else name
    return _cache[name]


src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(src, 'summary')
os.makedirs(dst, exist_ok=True)
rows = list(csv.DictReader(open(glob.glob(src + '/trace/*/*_kernel_stats.csv')[0])))
with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    cols = ['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev']
    w.writerow(cols)
    for r in rows[:25]:
        w.writerow([r[c][:160] if c == 'Name' else r[c] for c in cols])
def kernel_key(r, seen):
    """Kernel name without arguments; the two launches per epoch of tmf::k_mse_pass are told apart by dispatch order
    (first = user pass, CSR; second = item pass, CSC) and keyed '... [user pass]' / '... [item pass]'."""
    name = r['Kernel_Name']
    if name.startswith('_ZN3tmf'):   # bf16 instantiations come out mangled (the _BFloat16 template argument)
        name = demangle(name)
    if 'tmf::' not in name:
        return None
    k = name.split('(')[0].replace('void ', '')
    if 'k_mse_pass' in k:
        ids = seen.setdefault(k, [])
        if r['Dispatch_Id'] not in ids:
            ids.append(r['Dispatch_Id'])
        k += ' [user pass]' if ids.index(r['Dispatch_Id']) % 2 == 0 else ' [item pass]'
    return k


out = {}
for kind, cn in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    agg, meta, seen = collections.defaultdict(list), {}, {}
    rows_ = sorted(csv.DictReader(open(glob.glob(f'{src}/{kind}/*/*_counter_collection.csv')[0])), key=lambda r: int(r['Dispatch_Id']))
    for r in rows_:
        k = kernel_key(r, seen)
        if k is not None:
            agg[k].append(float(r['Counter_Value']))
            meta[k] = dict(vgpr=r['VGPR_Count'], agpr=r['Accum_VGPR_Count'], sgpr=r['SGPR_Count'], lds=r['LDS_Block_Size'],
                           workgroup=r['Workgroup_Size'], grid=r['Grid_Size'])
    for k, v in agg.items():
        d = out.setdefault(k, {})
        d[cn + '_KB_mean_per_launch'] = sum(v) / len(v)
        d['launches_' + cn] = len(v)
        d.update(meta[k])
tcc = glob.glob(f'{src}/pmc_tcc/*/*_counter_collection.csv')
if tcc:   # L2 hit rate per kernel: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), summed over the launches
    hm, seen = collections.defaultdict(lambda: [0.0, 0.0]), {}
    for r in sorted(csv.DictReader(open(tcc[0])), key=lambda r: int(r['Dispatch_Id'])):
        k = kernel_key(r, seen)
        if k is not None and r['Counter_Name'] in ('TCC_HIT_sum', 'TCC_MISS_sum'):
            hm[k][r['Counter_Name'] == 'TCC_MISS_sum'] += float(r['Counter_Value'])
    for k, (h, mi) in hm.items():
        if h + mi > 0:
            out.setdefault(k, {}).update(TCC_HIT_sum=h, TCC_MISS_sum=mi, l2_hit_rate=h / (h + mi))
for k, d in list(out.items()):
    f, w = d.get('FETCH_SIZE_KB_mean_per_launch', 0), d.get('WRITE_SIZE_KB_mean_per_launch', 0)
    d['hbm_traffic_bytes_per_launch_corrected'] = 2 * f * 1024 + w * 1024
sys.path.insert(0, root)
import bench  # noqa: E402
out['_epochs_in_pmc_runs'] = 3   # tools/profile.sh: --steps 2 --warmup 1; launches / epochs = launches of a kernel per epoch
out['_csrc_sha'] = bench.csrc_sha()   # bench.py quotes these counters only for this version of the kernels
out['_note'] = ('traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: on gfx950 FETCH_SIZE counts half of the bytes of 16 B/lane '
                'coalesced reads (MI355X_MICROARCH.md, HBM section); FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes')
json.dump(out, open(os.path.join(dst, f'{tag}_pmc.json'), 'w'), indent=1)

# The roofline from THIS directory alone: per kernel the rocprofv3 average duration of the traced run, the bytes
# of bench.py's byte model for the same run (gather = every gathered row counted; compulsory = what must cross HBM at least
# once) and both fractions priced on the trace's own durations - next to the HIP-event milliseconds the bench line quotes.
roof = {}
try:
    ex = json.load(open(os.path.join(src, 'bench_trace_extras.json')))
    stats = {}
    for r in rows:
        nm = demangle(r['Name']) if r['Name'].startswith('_ZN3tmf') else r['Name']
        stats[nm.split('(')[0].replace('void ', '')] = r
    entries = list(ex.get('roofline', {}).get('kernels', []))
    for e in entries:
        pat = bench.PMC_KERNELS.get(e['kernel'])
        prefix = pat[0] if isinstance(pat, tuple) else pat
        if isinstance(pat, tuple) and pat[1]:   # the two launches per epoch of k_mse_pass share one row of the kernel statistics
            continue
        hit = [(k, v) for k, v in stats.items() if prefix and k.startswith(prefix)]
        if not hit:
            continue
        calls = sum(int(v['Calls']) for _, v in hit)
        total_ns = sum(float(v['TotalDurationNs']) for _, v in hit)
        per_epoch = max(1, round(calls / max(ex.get('steps', 1) + ex.get('warmup', 0), 1)))      # launches of the kernel per epoch
        ns = total_ns / calls * per_epoch                                                          # per epoch, like the event brackets
        d = dict(rocprof_symbol=hit[0][0][:80], rocprof_calls=calls, launches_per_epoch=per_epoch, rocprof_avg_ns_per_epoch=ns,
                 hip_event_ms=e['ms'], rocprof_over_event=ns * 1e-6 / e['ms'], compulsory_hbm_bytes=e['hbm_bytes'],
                 useful_hbm_frac_rocprof=e['hbm_bytes'] / (ns * 1e-9) / bench.HBM_PEAK)
        if e.get('gather_bytes'):
            d.update(gather_bytes=e['gather_bytes'], l2_frac_rocprof=e['gather_bytes'] / (ns * 1e-9) / bench.L2_PEAK,
                     l2_frac_events=e.get('l2_frac'))
        roof[e['kernel']] = d
    mse = [v for k, v in stats.items() if k.startswith('tmf::k_mse_pass')]
    if mse:
        ev = sum(e['ms'] for e in entries if e['kernel'].startswith('mse_'))
        ns = sum(float(v['TotalDurationNs']) for v in mse) / max(ex.get('steps', 1) + ex.get('warmup', 0), 1)
        roof['mse_user_pass + mse_item_pass'] = dict(note='both launches of k_mse_pass per epoch (one row of the kernel statistics)',
                                                    rocprof_avg_ns_per_epoch=ns, hip_event_ms=ev, rocprof_over_event=ns * 1e-6 / ev)
    roof['_epoch'] = dict(ms_per_step=ex.get('ms_per_step'), algorithmic_bytes=ex.get('roofline', {}).get('epoch_algorithmic_bytes'),
                          algorithmic_over_hbm_peak=ex.get('roofline', {}).get('algorithmic_over_hbm_peak'),
                          epoch_hbm_frac=ex.get('roofline', {}).get('epoch_hbm_frac'), csrc_sha=out['_csrc_sha'],
                          note='rocprof durations are of the TRACED run (kernel-trace adds a few per cent on this pool); the bench line quotes '
                               'HIP-event brackets of an untraced run; rocprof_over_event says how far apart the two are')
    json.dump(roof, open(os.path.join(dst, f'{tag}_roofline.json'), 'w'), indent=1)
except (OSError, ValueError, KeyError) as err:
    print(f'no roofline file: {err!r}')
print(open(os.path.join(dst, f'{tag}_kernel_stats.csv')).read()[:1500])
if roof:
    print(json.dumps({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in
                          ('rocprof_avg_ns_per_epoch', 'hip_event_ms', 'rocprof_over_event', 'l2_frac_rocprof', 'useful_hbm_frac_rocprof')}
                      for k, v in roof.items() if not k.startswith('_')}, indent=1))
print(json.dumps({k: dict(traffic_GB=round(v.get('hbm_traffic_bytes_per_launch_corrected', 0) / 1e9, 2), l2_hit_rate=v.get('l2_hit_rate'))
                  for k, v in out.items() if not k.startswith('_')}, indent=1))
