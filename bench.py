#!/usr/bin/env python3
"""Benchmark of the matrix-factorization training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` starts its own ranks -
`python -m torch.distributed.run --standalone --nproc-per-node N bench.py ...` as a CHILD process, before this process has made
any GPU call (`spawn_ranks`; a process that has initialised the GPU must never exec another program on this pool) -
relays rank 0's JSON line and exits with the child's code.

A "step" is one full-batch training epoch (user pass + item pass + fresh-Adam updates + loss), the
unit the reference times at matrix_factorization.py:129-177.  Workload = BASELINE.json's metric
configuration: 1M users x 100K items, r = 128, WMRB with S = 1024 static negatives, ~1e8 interactions
("C4"), fp32, synthetic data generated on the device.  With N > 1 the users are split over
the ranks, V is replicated and its gradient is exchanged by RCCL reduce-scatter / all-gather
(teamoflow_amd/dist.py).  Default for N > 1 is STRONG scaling: the ONE 1M-user problem (BASELINE config 4 as written) cut
into contiguous cost-balanced user blocks; `--scaling weak` gives every rank its own 1M users and says so in the line.
The default N = 1 run also times one rank's shard of the 8-way split on this one GPU (`strong_scaling_projection`: a
PROJECTION of the 1 -> 8 curve from measured shard time + modelled wire time, not a measured N > 1 run).
Rank 0 prints ONE JSON line; DESIGN.md §4 defines every field of its `roofline` object.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from teamoflow_amd import _engine, _lib, _ops  # noqa: E402
from teamoflow_amd import dist as tdist  # noqa: E402
from teamoflow_amd.mf.utils import random_sampler_device  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; per-XCD L2 aggregate ~34.5 TB/s; Infinity Cache 256 MiB
HBM_PEAK, L2_PEAK, MALL_BYTES = 8.0e12, 34.5e12, 256 << 20
L2_GATHER_MEASURED = (16.8e12, 18.8e12)  # the guide's measured chip-wide rate of L2-resident row gathers
HBM_COPY_MEASURED = 6.29e12                # the guide's measured float4 copy from HBM (79 % of the 8 TB/s spec)
FABRIC_GATHER_MEASURED = 8.6e12            # the guide's measured rate of row gathers served by the Infinity Cache (38 MB table)
FABRIC_BOUND_FROM = 0.75                   # measured fabric traffic from this share of the HBM peak on = the kernel is bound there
FABRIC_LABEL = 'fabric (L2-miss traffic incl. Infinity-Cache hits)'
FABRIC_LABEL_FROM = 6.5e12                 # measured fabric rate from which a kernel is priced on the fabric roof: clearly above the ~6.3 TB/s an
                                           # HBM copy reaches (a streaming kernel AT that ceiling - the combine: 6.3 TB/s, L2 hit rate 0.01 - is HBM-bound)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


GEN_CHUNK_DRAWS = 1 << 25     # item draws generated at a time (bounds the transient memory of the generator: ~2 GB)
CALIBRATION_USERS = 65536     # users 0 .. 65535 of the problem calibrate the duplicate rate of the item draws (every rank draws them)


def drawn_degrees(m, n, nnz_target, seed, dev, inflate=1.0):
    """Item draws per user of the WHOLE problem (int64 [m]): lognormal(4.2, 0.8) weights - Box-Muller on two counter_hash words
    of (seed, user) - rescaled so that the draws sum to nnz_target * inflate.  O(m) work, the same vector on every rank."""
    from teamoflow_amd.mf.utils import counter_hash, hash_unit
    u = torch.arange(m, device=dev, dtype=torch.int64)
    z = torch.sqrt(-2.0 * torch.log(hash_unit(counter_hash(seed, u, 1)))) * torch.cos(2.0 * np.pi * hash_unit(counter_hash(seed, u, 2)))
    w = torch.exp(4.2 + 0.8 * z)
    return torch.clamp(torch.round(w * (nnz_target * inflate / float(w.sum()))), 1, max(1, n // 4)).to(torch.int64)


def item_permutation(n, dev):
    """The fixed permutation the power-law ranks are mapped through (the popular items are spread over the id range)."""
    from teamoflow_amd.mf.utils import _srl, counter_hash
    return torch.argsort(_srl(counter_hash(4242, torch.arange(n, device=dev, dtype=torch.int64)), 1), stable=True)


def draw_block(deg, b, e, n, items, seed, dev, perm):
    """Unique sorted keys user * n + item of users b .. e: draw t of user u is a function of (seed, u, t) alone."""
    from teamoflow_amd.mf.utils import counter_hash, hash_below, hash_unit
    keys, cum = [], torch.cumsum(deg[b:e], 0)
    u0 = b
    while u0 < e:
        # as many whole users as fit GEN_CHUNK_DRAWS draws (at least one)
        base = int(cum[u0 - b - 1]) if u0 > b else 0
        u1 = b + max(int(torch.searchsorted(cum, base + GEN_CHUNK_DRAWS, right=True)), u0 - b + 1)
        u1 = min(u1, e)
        d = deg[u0:u1]
        total = int(d.sum())
        u = torch.repeat_interleave(torch.arange(u0, u1, device=dev, dtype=torch.int64), d, output_size=total)
        t = torch.arange(total, device=dev, dtype=torch.int64) - torch.repeat_interleave(torch.cumsum(d, 0) - d, d, output_size=total)
        h = counter_hash(seed, u, t, 3)
        del t
        if items == 'zipf':
            ranks = torch.clamp(torch.pow(float(n + 1), hash_unit(h)).to(torch.int64) - 1, 0, n - 1)
            j = perm[ranks]
            del ranks
        else:
            j = hash_below(h, n)
        del h
        keys.append(torch.unique(u * n + j))
        del u, j
        u0 = u1
    return torch.cat(keys) if len(keys) != 1 else keys[0]


def calibrated_degrees(m, n, nnz_target, items, seed, dev):
    """Drawn degrees of the whole problem such that the UNIQUE pairs come out within ~1 % of nnz_target.  Duplicate
    (user, item) draws collapse (19 % of them at C4 with the power law), so the draw is inflated - calibrated on users
    0 .. CALIBRATION_USERS (users are independent and identically distributed, so a prefix estimates the global rate),
    never on the whole problem.  Deterministic: every rank arrives at the same vector."""
    perm = item_permutation(n, dev) if items == 'zipf' else None
    inflate, cal = 1.0, min(m, CALIBRATION_USERS)
    for attempt in range(4):
        deg = drawn_degrees(m, n, nnz_target, seed, dev, inflate)
        unique_est = float(draw_block(deg, 0, cal, n, items, seed, dev, perm).numel()) / float(deg[:cal].sum()) * float(deg.sum())
        if abs(unique_est - nnz_target) <= 0.005 * nnz_target:
            break
        inflate *= nnz_target / unique_est
    return deg, perm


def gen_interactions(m, n, nnz_target, items, seed, dev, users=None, plan=None):
    """Synthetic interactions of SURVEY.md §8d: user degrees lognormal(4.2, 0.8) rescaled to the
    target; item ids from a power law (alpha = 1) over a fixed permutation ('zipf') or uniform;
    values in {1..5}; unique row-major pairs.

    PER-USER SEEDED: the degree, item draws and values of user u are functions of (seed, u) only (counter_hash), so
    `users=(b, e)` generates exactly that block of the m-user problem - the union over any partition of the users IS the
    `users=None` problem, bit for bit (tests/test_bench_cpu.py) - and a rank of a strong-scaling job never builds more than
    its own share.  The only global quantities are the O(m) degree vector and its duplicate-rate calibration
    (`plan = calibrated_degrees(...)`), computed identically by every rank.
    Returns (indices [nnz, 2] with GLOBAL user ids, values [nnz])."""
    from teamoflow_amd.mf.utils import counter_hash, hash_below
    deg, perm = plan if plan is not None else calibrated_degrees(m, n, nnz_target, items, seed, dev)
    b, e = users if users is not None else (0, m)
    key = draw_block(deg, b, e, n, items, seed, dev, perm)
    vals = (1 + hash_below(counter_hash(seed, key, 7), 5)).to(torch.float32)
    return torch.stack([key // n, key % n], dim=1), vals


def init_table(rows, r, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(rows, r, device=dev, generator=g)
    return x * torch.rsqrt(torch.clamp((x * x).sum(), min=1e-12))


# ---------------------------------------------------------------------------------------------
# Byte models.  Per kernel:
#   gather  = bytes of factor rows the kernel pulls into registers, every gathered row counted (what the
#             L2 / Infinity Cache / HBM hierarchy has to deliver to the CUs) - SURVEY §8d's "algorithmic" figure;
#   hbm     = bytes that have to cross the HBM interface at least once: every streamed array once, a factor table
#             (or the cache-sized window the kernel is blocked into) once per sweep when it is cache-resident
#             (<= 256 MiB Infinity Cache), every gathered row when it is not.
# ---------------------------------------------------------------------------------------------
def wmrb_kernel_models(m, n, S, nnz, P, ld, s, ns, sliced, E_lists, n_slab, C, part_layers=None, u_sweeps=1, flat_streams=False,
                       v_sweeps=0):
    row = ld * s          # bytes of one factor row as stored
    row32 = ld * 4        # fp32 partial / gradient rows
    U_tab, V_tab = m * row, n * row
    v_resident = V_tab <= MALL_BYTES or ns > 1   # sliced: the slice being walked is L2-resident
    k = {}
    if sliced:
        off = 2 * m * (ns + 1) * 4
        layers = ns if part_layers is None else part_layers
        # gradU partial rows: one fp32 layer per slice written once, or (memory-light) ONE layer read-modified-written per slice
        part_wr = ns * m * row32 if layers == ns else (2 * ns - 1) * m * row32
        # flat_streams (tmf_wmrb_scores6): 8 bytes of stream per entry (packed id + place) instead of 4 bytes of id + the offsets
        k['wmrb_scores'] = dict(rows=m * S + nnz, gather=(m * S + nnz) * row,
                                hbm=((m * S + nnz) * 8 if flat_streams else m * S * 4 + off + nnz * 4) + U_tab
                                + (V_tab if v_resident else (m * S + nnz) * row) + m * S * 4 + nnz * 4,
                                roof='l2' if v_resident else 'hbm',
                                what=('sampled + interaction scores: one workgroup per (slice, 64 users) chunk of a flat entry stream, the users\' rows '
                                      'in LDS, V rows gathered from the L2-resident slice' if flat_streams else
                                      'sampled + interaction scores: V rows gathered from the L2-resident slice, ids staged in LDS'))
        k['wmrb_hinge'] = dict(rows=0, gather=0, hbm=m * S * 4 + 2 * nnz * 4 + m * 8 + m * S * 4 + nnz * 4, roof='issue',
                               what='sort + binary search + scans per user (one wave per user); streams sp / p in, D / delta out')
        if v_sweeps:   # row-stationary gradU (tmf_wmrb_gradu4): no partial rows, no finish kernel; every launch of resident lane groups walks the whole catalog
            k['wmrb_gradu'] = dict(rows=m * S + nnz, gather=(m * S + nnz) * row,
                                   hbm=2 * m * S * 4 + off + 2 * nnz * 4 + v_sweeps * V_tab + 2 * U_tab, roof='l2' if v_resident else 'hbm',
                                   what='row-stationary gradU + fresh Adam: lane groups own users, keep their gradient rows in registers and walk all slices')
        else:
            k['wmrb_gradu'] = dict(rows=m * S + nnz, gather=(m * S + nnz) * row,
                                   hbm=2 * m * S * 4 + off + 2 * nnz * 4 + (V_tab if v_resident else (m * S + nnz) * row) + part_wr,
                                   roof='l2' if v_resident else 'hbm',
                                   what='D- and delta-weighted V rows per (user, slice); rows with weight 0 are skipped (but counted here)')
            k['wmrb_finish'] = dict(rows=0, gather=0, hbm=layers * m * row32 + 2 * U_tab, roof='hbm', what='ordered sum of the slice partials + fresh Adam')
    else:
        k['wmrb_user_pass'] = dict(rows=2 * (m * S + nnz), gather=2 * (m * S + nnz) * row,
                                   hbm=m * S * 4 + 2 * nnz * 4 + 2 * U_tab + (V_tab if V_tab <= MALL_BYTES else 2 * (m * S + nnz) * row) + m * S * 4 + nnz * 4,
                                   roof='l2' if V_tab <= MALL_BYTES else 'hbm', what='fused user pass (one workgroup per user)')
    u_resident = U_tab <= MALL_BYTES or C > 1   # user-blocked lists: the block being gathered from is L2-resident
    # u_sweeps > 1: the row-stationary form (tmf_wsum_rows5) - every launch of resident lane groups walks ALL user blocks, so U
    # crosses HBM once per launch; its slab holds only the parts of the rows it cut (popular items), the rest is finished in place
    k['wmrb_item_pass'] = dict(rows=E_lists, gather=E_lists * row,
                               hbm=E_lists * 12 + (u_sweeps * U_tab if u_resident else E_lists * row)
                               + (n_slab * row32 + (2 * V_tab if u_sweeps > 1 else 0) if n_slab else 2 * V_tab),
                               roof='l2' if u_resident else 'hbm',
                               what=('row-stationary weighted U-row gather-sum: lane groups own (virtual) item rows and walk the user blocks; '
                                     if u_sweeps > 1 else 'weighted U-row gather-sum over the (user block, item) lists; ') + '4-byte weight gathers')
    if n_slab:
        k['wmrb_combine'] = dict(rows=0, gather=0, hbm=n_slab * row32 + 2 * V_tab, roof='hbm', what='ordered sum of the per-block partial rows + fresh Adam')
    return k


def mse_kernel_models(m, n, nnz, ld, s):
    row = ld * s
    U_tab, V_tab = m * row, n * row
    k = {}
    # the gathered table is either cache-resident (Infinity Cache: priced against the L2 roof, the upper bound of anything
    # served on chip) or streams from HBM row by row
    k['mse_user_pass'] = dict(rows=nnz, gather=nnz * row, hbm=nnz * 8 + 2 * U_tab + (V_tab if V_tab <= MALL_BYTES else nnz * row),
                              roof='l2' if V_tab <= MALL_BYTES else 'hbm', what='gather V rows -> dot -> loss -> gradient -> Adam, CSR by user')
    k['mse_item_pass'] = dict(rows=nnz, gather=nnz * row, hbm=nnz * 8 + 2 * V_tab + (U_tab if U_tab <= MALL_BYTES else nnz * row),
                              roof='l2' if U_tab <= MALL_BYTES else 'hbm', what='the same with U rows, CSC by item')
    return k


def survey_algorithmic_bytes(loss, m, n, S, nnz, P, r, s):
    """SURVEY.md §8d's ALGORITHMIC bytes of one epoch (every gathered row counted, wherever it is served from):
    MSE  nnz (2 r s + 24) + (m + n) 2 r s;   WMRB  m S 3 r s + m S 16 + P (2 r s + 24) + (m + n) 2 r s."""
    if loss == 'mse':
        return float(nnz) * (2 * r * s + 24) + float(m + n) * 2 * r * s
    return float(m) * S * 3 * r * s + float(m) * S * 16 + float(P) * (2 * r * s + 24) + float(m + n) * 2 * r * s


def epoch_hbm_frac(models, seconds):
    """Compulsory HBM bytes of the epoch over its time against the HBM peak; None when the no-reuse model of a gather from a
    table beyond the Infinity Cache over-counts (skewed row popularity) and the figure would exceed 1."""
    f = sum(k['hbm'] for k in models.values()) / seconds / HBM_PEAK
    return f if f <= 1.0 else None


def roofline_report(models, prof, pmc=None):
    """One entry per kernel: its measured duration (HIP events on the launch stream inside the timed region), the rate
    it moves its gather bytes at against the L2 roof, its compulsory HBM bytes against the HBM roof (`useful_hbm_frac`), and -
    when a PMC profile of this code version is committed - the measured fabric traffic and how many times the compulsory
    bytes it is (`traffic_over_compulsory`).  Fractions are RAW ratios, never clamped: a measured fabric rate above the
    6.29 TB/s the guide measures for an HBM copy cannot all be HBM bytes (FETCH_SIZE counts Infinity-Cache hits), so such a
    kernel is labelled `bound: fabric` and priced against the guide's 8.6 TB/s Infinity-Cache gather rate."""
    entries = []
    for name, k in models.items():
        ms = prof.mean_ms(name)
        if not ms == ms:   # NaN: kernel not launched
            continue
        t = ms * 1e-3
        e = dict(kernel=name, ms=ms, what=k['what'], hbm_bytes=k['hbm'], hbm_rate_GBps=k['hbm'] / t / 1e9,
                 hbm_frac=k['hbm'] / t / HBM_PEAK, useful_hbm_frac=k['hbm'] / t / HBM_PEAK)
        if k['gather']:
            e.update(rows_gathered=k['rows'], gather_bytes=k['gather'], gather_rate_GBps=k['gather'] / t / 1e9,
                     l2_frac=k['gather'] / t / L2_PEAK)
        meas = pmc.get(name) if pmc else None
        rate = meas['bytes'] / t if meas is not None else None

        def fabric(extra=''):
            e.update(bound=FABRIC_LABEL, achieved=rate / 1e9, peak=FABRIC_GATHER_MEASURED / 1e9, frac=rate / FABRIC_GATHER_MEASURED,
                     frac_source='pmc', hbm_peak_frac=rate / HBM_PEAK,
                     note=f"measured fabric rate {rate / 1e12:.2f} TB/s (L2-miss + write traffic, L2 hit rate {meas.get('l2_hit_rate')}) is above the "
                          f"{HBM_COPY_MEASURED / 1e12:.2f} TB/s the guide measures for a float4 copy from HBM: FETCH_SIZE counts Infinity-Cache "
                          f"hits too, so these are NOT all HBM bytes; priced against the guide's {FABRIC_GATHER_MEASURED / 1e12:.1f} TB/s "
                          f"Infinity-Cache row-gather rate; traffic is {meas['bytes'] / k['hbm']:.1f}x the compulsory bytes" + extra
                          + ('; ABOVE that rate: the x2 correction of FETCH_SIZE (16-byte-per-lane reads) also doubles whatever the kernel '
                             'reads 4 bytes per lane (the item pass: its gathered weights), so the true figure is lower'
                             if rate > FABRIC_GATHER_MEASURED else ''))
        if k['roof'] == 'l2' and rate is not None and rate / HBM_PEAK >= FABRIC_BOUND_FROM and rate / HBM_PEAK > e['l2_frac']:
            # blocked for the L2s, but the counters say the memory-side fabric is the roof that binds (config-5 shard: L2 hit rate
            # 0.41, 7 - 9 TB/s of fabric traffic - at or above the measured copy ceiling): price the kernel where it is bound.
            # (The C4 item pass moves 4.5 TB/s over the fabric - 0.56 of the peak, more than its L2 fraction, but nowhere near a
            # roof: it stays on the L2 roof it is built against.)
            if rate > FABRIC_LABEL_FROM:
                fabric(f"; blocked for the L2s (gather rate {e['gather_rate_GBps'] / 1e3:.1f} TB/s = {e['l2_frac']:.2f} of the L2 roof)")
            else:
                e.update(bound='hbm', achieved=rate / 1e9, peak=HBM_PEAK / 1e9, frac=rate / HBM_PEAK, frac_source='pmc',
                         note=f"blocked for the L2s (gather rate {e['gather_rate_GBps'] / 1e3:.1f} TB/s = {e['l2_frac']:.2f} of the L2 roof) but bound by "
                              f"the memory side: measured {rate / 1e12:.2f} TB/s of L2-miss + write traffic, L2 hit rate {meas.get('l2_hit_rate')}")
        elif k['roof'] == 'l2':
            e.update(bound='l2', achieved=e['gather_rate_GBps'], peak=L2_PEAK / 1e9, frac=e['l2_frac'])
        elif k['roof'] == 'hbm' and meas is not None:
            # HBM-bound kernel with counters of this code version: the fraction is MEASURED fabric bytes (L2 misses + writes,
            # FETCH_SIZE / WRITE_SIZE) over the kernel's time - not the byte model
            if rate > FABRIC_LABEL_FROM:
                fabric()
            else:
                e.update(bound='hbm', achieved=rate / 1e9, peak=HBM_PEAK / 1e9, frac=rate / HBM_PEAK, frac_source='pmc')
            e['hbm_model_frac'] = e['hbm_frac']
        elif k['roof'] == 'hbm' and e['hbm_frac'] > 1.0 and k['gather']:
            # no counters for this code version and the no-reuse model (every gathered row of a table beyond the Infinity Cache comes
            # from HBM) exceeds the peak: rows are re-served on chip.  Price the kernel on the L2 roof and say so.
            e.update(bound='l2', achieved=e['gather_rate_GBps'], peak=L2_PEAK / 1e9, frac=e['l2_frac'],
                     note=f"UNVALIDATED byte model: without reuse it would give {e['hbm_frac']:.2f} of the HBM peak, so rows are served on chip; "
                          "no PMC profile of this code version to say how many")
            e['hbm_frac'] = e['useful_hbm_frac'] = None
        elif k['roof'] == 'hbm':
            e.update(bound='hbm', achieved=e['hbm_rate_GBps'], peak=HBM_PEAK / 1e9, frac=e['hbm_frac'], frac_source='byte model (no counters)')
        else:
            e.update(bound='issue', achieved=e['hbm_rate_GBps'], peak=HBM_PEAK / 1e9, frac=e['hbm_frac'],
                     note='instruction-issue / LDS-latency bound; the HBM figure only shows it is far from that roof')
        if meas is not None:
            e.update(traffic=meas['bytes'], hbm_traffic_frac=rate / HBM_PEAK, traffic_over_compulsory=meas['bytes'] / k['hbm'],
                     traffic_uncorrected=meas['bytes_uncorrected'], traffic_rate_GBps=rate / 1e9,
                     l2_hit_rate=meas.get('l2_hit_rate'))
        entries.append(e)
    entries.sort(key=lambda e: -e['ms'])
    return entries


# kernel symbol prefixes of the per-kernel timer names, for matching the committed PMC profile
PMC_KERNELS = {'wmrb_scores': ('tmf::k_wmrb_scores', ''), 'wmrb_hinge': 'tmf::k_wmrb_hinge2', 'wmrb_gradu': ('tmf::k_wmrb_gradu', ''),
               'wmrb_finish': 'tmf::k_wmrb_finish', 'wmrb_item_pass': ('tmf::k_wsum_', ''), 'wmrb_combine': 'tmf::k_combine_rows',
               'wmrb_user_pass': 'tmf::k_wmrb_user',
               # the two launches per epoch of one kernel, told apart by dispatch order in tools/profile_summary.py
               'mse_user_pass': ('tmf::k_mse_pass', '[user pass]'), 'mse_item_pass': ('tmf::k_mse_pass', '[item pass]')}
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_c4_latest.json')
PMC_LEG_FILES = {'c4_mse': os.path.join(ROOT, 'profiles', 'pmc_c4_mse_latest.json'),
                 'c5_shard_bf16': os.path.join(ROOT, 'profiles', 'pmc_c5_latest.json')}


def csrc_sha():
    """Fingerprint of the kernel sources a PMC profile belongs to (tools/profile_summary.py stamps the same value): every
    source of the TRAINING path - kernels, index preparation, shared header, the ABI.  tmf_predict.hip and tmf_predict_split.hip
    (ranking only: none of their kernels runs in a profiled epoch) are left out, so that a change there does not orphan the
    training profiles."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'teamoflow_amd', 'csrc')
    files = sorted(f for f in os.listdir(d) if f.endswith(('.hip', '.h')) and f not in ('tmf_predict.hip', 'tmf_predict_split.hip'))
    for path in [os.path.join(d, f) for f in files] + [os.path.join(ROOT, 'include', 'tmf.h')]:
        h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(path=None):
    """Fabric bytes per launch from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and WRITE_SIZE
    in separate runs; TCC_HIT / TCC_MISS in a third -> l2_hit_rate).  bench.py cannot collect counters itself; the workload is seeded, so the profile applies to this
    run - but only when it was taken on THIS version of the kernels: the file carries the sha of csrc/ and is ignored on
    a mismatch.  FETCH_SIZE counts half the bytes of 16-byte-per-lane reads on gfx950 (MI355X_MICROARCH.md, HBM):
    `bytes` doubles it for the row-gather kernels (their traffic is 16-byte row loads), `bytes_uncorrected` does not."""
    path = path or PMC_FILE
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, 'no profile committed'
    if d.get('_csrc_sha') != csrc_sha():
        return None, f"profile is for csrc {d.get('_csrc_sha')}, this is {csrc_sha()}"
    out = {}
    for name, pat in PMC_KERNELS.items():
        prefix, suffix = pat if isinstance(pat, tuple) else (pat, '')
        for k, v in d.items():
            if k.startswith(prefix) and k.endswith(suffix) and 'FETCH_SIZE_KB_mean_per_launch' in v:
                # per EPOCH (the timers bracket all launches of a kernel in an epoch: the config-5 gradU goes out once per slice)
                per_epoch = max(1.0, v.get('launches_FETCH_SIZE', 1) / float(d.get('_epochs_in_pmc_runs', v.get('launches_FETCH_SIZE', 1))))
                f, w = v['FETCH_SIZE_KB_mean_per_launch'] * 1024 * per_epoch, v.get('WRITE_SIZE_KB_mean_per_launch', 0) * 1024 * per_epoch
                wide = name != 'wmrb_hinge'   # the hinge kernel reads 4 bytes per lane: correction not calibrated, left out
                out[name] = dict(bytes=(2 * f if wide else f) + w, bytes_uncorrected=f + w, l2_hit_rate=v.get('l2_hit_rate'),
                                 launches_per_epoch=per_epoch)
    return out, os.path.relpath(path, ROOT)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box
    shows all 256 host CPUs but grants a 16-core share; spinning up 256 OpenMP threads there stalls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, 16) if n > 64 else n


def baseline_sample(idx, val, users, n):
    """The interactions of users 0 .. users-1 as host arrays.  Selected COLUMN BY COLUMN (_engine.take_interactions):
    row-indexing a [1e8, 2] int64 device tensor with a mask returns wrong rows on this PyTorch-ROCm build - the count
    would still look right.  The sample is checked: every user below `users`, every item below n, unique row-major order kept."""
    sidx, sval = (t.cpu().numpy() for t in _engine.take_interactions(idx, val, idx[:, 0] < users))
    if len(sval):
        key = sidx[:, 0].astype(np.int64) * n + sidx[:, 1]
        if sidx.shape[0] != len(sval) or sidx.min() < 0 or int(sidx[:, 0].max()) >= users or int(sidx[:, 1].max()) >= n \
                or not bool(np.all(np.diff(key) > 0)):
            raise AssertionError('cpu_baseline: the selected interactions are not the row-major prefix of the workload')
    return sidx, sval


def cpu_baseline(loss, idx, val, R, U0, V0, n, S, lr, users=None, epochs=3):
    """The C/OpenMP closed-form epoch (oracle/sparse_ref.c - "sparse CPU restatement, not the reference
    formulation") on the first `users` users of the same workload with every host core this
    process may use.  A reported baseline, never the measured path."""
    from oracle import sparse_c
    cores = host_cores()
    sparse_c.set_threads(cores)
    users = min(users or (131072 if loss == 'wmrb' else 262144), int(U0.shape[0]))   # ~10-20 s of CPU work at C4 on 16 cores
    sidx, sval = baseline_sample(idx, val, users, n)
    Us, Vs = U0[:users].cpu().numpy(), V0.cpu().numpy()
    plan = sparse_c.Plan(sidx, sval, users, n, R[:users].cpu().numpy() if loss == 'wmrb' else None)

    def epoch(Uc, Vc):
        if loss == 'wmrb':
            return sparse_c.wmrb_epoch(Uc, Vc, plan, n, S, lr, want_grads=False)[:2]
        return sparse_c.mse_epoch(Uc, Vc, plan, lr, want_grads=False)[:2]

    Uc, Vc = epoch(Us, Vs)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    for _ in range(epochs):
        Uc, Vc = epoch(Uc, Vc)
    dt = (time.perf_counter() - t0) / epochs
    return dict(value=len(sval) / dt, unit='interactions/s', cores=cores, kind='port',
                sample=f'{epochs} {loss.upper()} epochs of oracle/sparse_ref.c (C, OpenMP, {cores} threads) on the first {users} '
                       f'users of the same workload against all {n} items ({len(sval)} interactions'
                       + (f', S={S}' if loss == 'wmrb' else '') + f', r={Us.shape[1]}), {dt:.2f} s per epoch; '
                       'sparse closed-form restatement, not the reference\'s dense formulation (that one: reference_formulation_cpu)')


def recall_parity(dev):
    """recall@10 of the engine vs the oracle on the C1 golden case (BASELINE 'recall@10 parity')."""
    from oracle import dense_ref
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'c1_mse.npz')))
    model = MatrixFactorization(5)
    model.user_embedding = torch.tensor(g['U_450']).to(dev)
    model.item_embedding = torch.tensor(g['V_450']).to(dev)
    got = float(model.recall_at_k(torch.tensor(g['A'])).mean())
    want = float(dense_ref.recall_at_k_dense(g['U_450'], g['V_450'], g['A'], 10).mean())
    return got, want


def small_configs(dev, quick=False):
    """BASELINE configs 1-3 (CPU-runnable shapes): fit() wall time on the GPU (hipGraph-replayed epochs)
    next to the dense-faithful CPU restatement of the reference formulation (oracle/dense_ref.py: identity
    feature matmuls, [m, n] scores, autograd, fresh Adam) timed over the same region the reference times."""
    from oracle import datagen as G
    from oracle import dense_ref
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import WMRBLoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    out = {}
    cases = [('C1', 100, 50, 5, 0.05, 'mse', 1e-2, 450, 450, None),
             ('C2', 943, 1682, 32, 100000 / (0.9 * 943 * 1682), 'mse', 1e-3, 100, 100, None),
             ('C3', 6040, 3706, 64, 1000209 / (0.9 * 6040 * 3706), 'wmrb', 0.1, 100, 2, 3706 // 2)]
    if quick:  # default bench run: C2 in full, one dense CPU epoch of C3 (~10 s of CPU work in all)
        cases = [cases[1], cases[2][:8] + (1,) + cases[2][9:]]
    for name, m, n, r, density, loss, lr, epochs, cpu_epochs, S in cases:
        np.random.seed(0)
        idx, val, shape, A = G.generate_random_interaction(m, n, density=density)
        U0 = G.normal_init(m, r, 1) if loss == 'mse' else G.uniform_init(m, r, 1)
        V0 = G.normal_init(n, r, 2) if loss == 'mse' else G.uniform_init(n, r, 2)
        kw = dict(user_weight_graph=FixedInitializer(U0), item_weight_graph=FixedInitializer(V0))
        R = None
        if loss == 'wmrb':
            R = np.stack([np.random.choice(n, S, replace=False) for _ in range(m)])
            kw.update(loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S)
        model = MatrixFactorization(r, **kw)
        model.verbose = False
        if R is not None:
            model.random_ind = torch.as_tensor(R)
        inter = SparseInteractions(idx, val, shape)
        model.fit(epochs, eye(m), eye(n), inter, lr=lr)      # includes graph capture
        first = model.fit_seconds_
        model.fit(epochs, eye(m), eye(n), inter, lr=lr)
        gpu_s = model.fit_seconds_
        torch.set_num_threads(host_cores())
        ref = dense_ref.fit_dense(U0, V0, idx, val, loss, cpu_epochs, lr, random_ind=R, n_items=n, n_samples=S)
        cpu_per_epoch = ref['seconds'] / cpu_epochs
        k = min(len(ref['loss']), len(model.loss_history_))
        out[name] = dict(shape=[m, n], r=r, loss=loss, nnz=int(len(val)), epochs=epochs,
                         gpu_fit_seconds=gpu_s, gpu_fit_seconds_first_call=first,
                         gpu_interactions_per_sec=len(val) * epochs / gpu_s,
                         cpu_dense_seconds_per_epoch=cpu_per_epoch, cpu_epochs_timed=cpu_epochs,
                         cpu_interactions_per_sec=len(val) / cpu_per_epoch, cpu_threads=torch.get_num_threads(),
                         loss_rel_diff_first_epochs=float(np.abs(np.array(model.loss_history_[:k]) - ref['loss'][:k]).max()
                                                          / np.abs(ref['loss'][:k]).max()),
                         recall_at_10=float(model.recall_at_k(torch.tensor(A)).mean()))
        if cpu_epochs == epochs:  # the CPU restatement ran the whole fit: end-to-end recall@10 parity (BASELINE: within 1e-3)
            want = float(dense_ref.recall_at_k_dense(ref['U'], ref['V'], A, 10).mean())
            out[name].update(recall_at_10_cpu_restatement=want, recall_at_10_abs_diff=abs(out[name]['recall_at_10'] - want))
        if loss == 'wmrb' and not quick:
            # end-to-end WMRB parity: 20 epochs by the engine and by the C/OpenMP closed-form restatement from the same start
            from oracle import sparse_c
            sparse_c.set_threads(host_cores())
            e2e = 20
            plan_c = sparse_c.Plan(idx, val, m, n, R)
            Uc, Vc, closs = U0, V0, []
            for _ in range(e2e):
                Uc, Vc, mean, _t = sparse_c.wmrb_epoch(Uc, Vc, plan_c, n, S, lr, want_grads=False)
                closs.append(mean)
            model.fit(e2e, eye(m), eye(n), inter, lr=lr)
            got = float(model.recall_at_k(torch.tensor(A)).mean())
            want = float(dense_ref.recall_at_k_dense(Uc, Vc, A, 10).mean())
            out[name]['end_to_end_20_epochs'] = dict(
                recall_at_10_engine=got, recall_at_10_c_restatement=want, abs_diff=abs(got - want),
                loss_rel_diff=float(np.abs(np.array(model.loss_history_) - np.array(closs)).max() / np.abs(closs).max()))
        log(f'[bench] {name}: {out[name]}')
    return out


LINE_HARD_CAP = 8192   # bytes: the driver keeps only a tail of stdout, so the ONE line must stay small (round 3's 20.5 KB line was cut)
LINE_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
             'data', 'config', 'roofline', 'cpu_baseline', 'hbm_legs', 'predict_rows_per_sec', 'predict_tflops', 'predict_arithmetic',
             'predict_top10_rows_identical_to_fp64', 'recall_at_10',
             'hinge_terms_per_sec', 'collectives', 'extras')
ROOFLINE_KEYS = ('bound', 'kernel', 'kernel_ms', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'hbm_traffic_frac', 'traffic_over_compulsory',
                 'useful_hbm_frac', 'l2_hit_rate', 'traffic_source', 'csrc_sha', 'epoch_hbm_frac', 'algorithmic_over_hbm_peak', 'kernels_ms')


def _short(x, digits=6):
    """Floats to `digits` significant digits (the full-precision numbers are in the extras file)."""
    if isinstance(x, float):
        return float(f'{x:.{digits}g}') if x == x and abs(x) != float('inf') else None
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def compact_line(out):
    """The ONE JSON line of the bench contract, cut down to what the driver and the judge read (target <= 4 KB, hard cap
    LINE_HARD_CAP): headline metric, config, the dominant kernel's roofline entry, the CPU baseline, the ranking numbers.
    Everything else of `out` (per-kernel entries, HBM legs, projection, per-arithmetic predict table, API fit, small configs,
    notes) goes to the extras file (write_extras)."""
    line = {k: out[k] for k in LINE_KEYS if k in out}
    cfg = out.get('config', {})
    line['config'] = {k: cfg[k] for k in ('workload', 'interactions_total', 'interactions_per_gpu', 'parallelism', 'lr') if k in cfg}
    r = out.get('roofline', {})
    roof = {k: r.get(k) for k in ROOFLINE_KEYS if k in r}
    if 'kernels' in r:
        roof['kernels_ms'] = {e['kernel']: [round(e['ms'], 3), e['bound'].split(' ')[0], round(e['frac'], 3) if e.get('frac') is not None else None]
                              for e in r['kernels']}
    line['roofline'] = roof
    if 'cpu_baseline' in out:
        c = out['cpu_baseline']
        line['cpu_baseline'] = {k: c[k] for k in ('value', 'unit', 'cores', 'kind') if k in c}
        line['cpu_baseline']['sample'] = str(c.get('sample', ''))[:240]
    if 'hbm_legs' in out:
        # the two side legs whose factor rows really come from HBM, in short: epoch time, interactions per second, and
        # per kernel [ms, roof, fraction of that roof] - driver-parsed numbers instead of extras-file ones
        line['hbm_legs'] = {name: dict(ms_per_step=leg.get('ms_per_step'), value=leg.get('value'), unit=leg.get('unit'), dtype=leg.get('dtype'),
                                       epoch_hbm_frac=leg.get('epoch_hbm_frac'),
                                       kernels_ms={e['kernel']: [round(e['ms'], 3), e['bound'].split(' ')[0],
                                                                 round(e['frac'], 3) if e.get('frac') is not None else None]
                                                   for e in leg.get('kernels', [])})
                            for name, leg in out['hbm_legs'].items()}
    if 'recall_at_10' in out:
        q = out['recall_at_10']
        line['recall_at_10'] = {k: q[k] for k in ('engine', 'oracle', 'abs_diff') if k in q}
        if 'end_to_end_C2' in q:
            line['recall_at_10']['end_to_end_C2_abs_diff'] = q['end_to_end_C2'].get('abs_diff')
    if 'collectives' in out:
        c = out['collectives']
        line['collectives'] = {k: v for k, v in c.items() if not isinstance(v, (dict, list, str)) or k in ('backend',)}
    line = _short(line)
    text = json.dumps(line, separators=(',', ':'))
    if len(text) > LINE_HARD_CAP:   # never print a line the driver cannot take: drop the optional parts, largest first
        for k in ('collectives', 'hinge_terms_per_sec', 'extras', 'hbm_legs'):
            line.pop(k, None)
        line['roofline'].pop('kernels_ms', None)
        line['cpu_baseline'] = {k: v for k, v in line.get('cpu_baseline', {}).items() if k != 'sample'}
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) <= LINE_HARD_CAP, len(text)
    return text


def write_extras(out, path=None):
    """Everything the bench measured, in full, beside the compact line: gpurun_out/bench_extras.json (TMF_BENCH_EXTRAS overrides)."""
    path = path or os.environ.get('TMF_BENCH_EXTRAS') or os.path.join(ROOT, 'gpurun_out', 'bench_extras.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        log(f'[bench] extras not written: {e!r}')
        return None


class Workload:
    """One rank's problem, resident in HBM: interactions, negative table, plans, factor tables."""

    def __init__(self, args, m, n, nnz_target, r, S, loss, dtype, rank, world, dev, strong=False):
        self.n, self.r, self.S, self.loss, self.dtype = n, r, S, loss, dtype
        t0 = time.perf_counter()

        def trace(what):   # TMF_BENCH_TRACE=1: where every rank is in its preparation (stderr)
            if os.environ.get('TMF_BENCH_TRACE') == '1':
                torch.cuda.synchronize()
                log(f'[bench rank {rank}] {what} at {time.perf_counter() - t0:.1f} s, {torch.cuda.memory_allocated() / 2**30:.1f} GiB allocated, '
                    f'{torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved')
        trace('start')
        n_pad = tdist.padded_rows(n, world)
        if strong and world > 1:
            # the ONE global problem, cut into contiguous user blocks of balanced cost (drawn interactions + negatives).  Every
            # rank computes the O(m) degree vector and generates ONLY its own block (gen_interactions is per-user seeded)
            gplan = calibrated_degrees(m, n, nnz_target, args.item_dist, 0, dev)
            bounds = tdist.partition_users(_engine._excl_cumsum(gplan[0]), world, per_user_cost=S if loss == 'wmrb' else 0)
            if torch.distributed.is_initialized():
                # every rank derived the partition by itself: they must agree (they do when the ranks compute alike - same device
                # type, same library; a disagreement would silently drop or duplicate users)
                red = 'cpu' if torch.distributed.get_backend() == 'gloo' else dev
                lo = torch.tensor(bounds, dtype=torch.int64, device=red)
                hi = lo.clone()
                torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
                torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
                if not torch.equal(lo, hi):
                    raise RuntimeError(f'ranks disagree on the user partition: {lo.tolist()} .. {hi.tolist()}')
            b, e = bounds[rank], bounds[rank + 1]
            idx, val = gen_interactions(m, n, nnz_target, args.item_dist, 0, dev, users=(b, e), plan=gplan)
            idx[:, 0] -= b
            U0 = init_table(m, r, 11, dev)[b:e].clone()
            self.user_block, m = (b, e), e - b
            del gplan
            trace('own block of the global problem generated')
        else:
            idx, val = gen_interactions(m, n, nnz_target, args.item_dist, rank, dev)
            U0 = init_table(m, r, 11 + rank, dev)
            self.user_block = (rank * m, (rank + 1) * m)
        self.m = m
        self.idx, self.val, self.nnz = idx, val, int(val.numel())
        V0 = torch.zeros(n_pad, r, device=dev)
        V0[:n] = init_table(n, r, 7, dev)  # identical on every rank
        self.U0, self.V0, self.n_pad = U0, V0, n_pad
        ld = _lib.padded_ld(r)
        self.plan = _engine.InteractionPlan(idx, val, m, n_pad, user_chunks=_engine.mse_user_chunks() if loss == 'mse' else 1,
                                            csc=loss == 'mse')
        self.wplan, self.R = None, None
        trace('interaction plan built')
        if loss == 'wmrb':
            if strong and world > 1:
                # the ONE problem: its negative table is the table of the N = 1 run (seed 100), of which this rank draws its users' rows
                self.R = random_sampler_device(n, m, S, seed=100, device=dev, user_offset=self.user_block[0])
            else:
                self.R = random_sampler_device(n, m, S, seed=100 + rank, device=dev)
            trace('negative table drawn')
            self.wplan = _engine.wmrb_plan_for(self.plan, self.R, r, torch.bfloat16 if dtype == 'bf16' else torch.float32)
        trace('WMRB plan built')
        self.st = _engine.TrainState(U0, V0, self.plan, r, self.wplan, dtype=torch.bfloat16 if dtype == 'bf16' else torch.float32)
        trace('training state allocated')
        self.adam = _engine.adam_constants(args.lr)
        self.c = n / S
        torch.cuda.synchronize()
        self.prep_seconds = time.perf_counter() - t0

    def models(self):
        st, p, w = self.st, self.plan, self.wplan
        s = 2 if self.dtype == 'bf16' else 4
        if self.loss == 'wmrb':
            vs = -(-self.m // st.users_per_launch) if getattr(st, 'row_stationary', False) else 0
            if w.rows4:   # row-stationary item pass: slab = the parts of the cut rows, one sweep of U per launch of resident lane groups
                work = w.vrows.n_vrows if w.vrows is not None else self.n
                return wmrb_kernel_models(self.m, self.n, self.S, self.nnz, p.n_pos, st.ld, s, w.n_slices, w.sliced, int(w.rowptr_e[-1]),
                                          w.vrows.n_slab if w.vrows is not None else 0, w.user_chunks, getattr(st, 'part_layers', None),
                                          u_sweeps=max(2, -(-work // st.rows4_per_launch)), flat_streams=getattr(w, 's6', None) is not None, v_sweeps=vs)
            return wmrb_kernel_models(self.m, self.n, self.S, self.nnz, p.n_pos, st.ld, s, w.n_slices, w.sliced,
                                      int(w.rowptr_e[-1]), w.seg_e.n_slab if w.seg_e is not None else 0, w.user_chunks,
                                      getattr(st, 'part_layers', None), flat_streams=getattr(w, 's6', None) is not None, v_sweeps=vs)
        return mse_kernel_models(self.m, self.n, self.nnz, st.ld, s)

    def describe(self, tag):
        return (f'{tag}: {self.m} users x {self.n} items per GPU, r={self.r}, {self.loss.upper()}'
                + (f' S={self.S}' if self.loss == 'wmrb' else '') + ', item ids zipf-like, lognormal user degrees')


def run_steps(wl, steps, warmup, dp=None, backend=None):
    """W untimed + K timed epochs -> (seconds of the K epochs, seconds of the W epochs, KernelTimer, loss buffer)."""
    st, prof = wl.st, _engine.KernelTimer()
    loss_buf = torch.zeros(steps + warmup + 1, dtype=torch.float64, device=st.U.device)

    def step(i, p):
        if dp is not None:
            backend.prof = p
            loss_buf[i] = dp.step()
        else:
            if wl.loss == 'wmrb':
                _engine.epoch_wmrb(st, wl.adam, wl.c, loss_buf[i:i + 1], prof=p)
            else:
                _engine.epoch_mse(st, wl.adam, loss_buf[i:i + 1], prof=p)
            st.swap()

    def fence():
        torch.cuda.synchronize()
        if dp is not None:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    fence()
    tw = time.perf_counter()
    for i in range(warmup):
        step(i, None)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i, prof)
    fence()
    t1 = time.perf_counter()
    return t1 - t0, t0 - tw, prof, loss_buf


def hbm_leg(args, dev, name, m, n, nnz, r, S, loss, dtype, steps, warmup, pmc_key=None):
    """A side leg on a workload whose factor tables really stream from HBM: same timed-region rules.
    `traffic` per kernel = fabric bytes from the committed PMC passes of the same workload and code version (or absent)."""
    wl = Workload(args, m, n, nnz, r, S, loss, dtype, 0, 1, dev)
    elapsed, _, prof, loss_buf = run_steps(wl, steps, warmup)
    models = wl.models()
    pmc, pmc_src = pmc_traffic(PMC_LEG_FILES[pmc_key]) if pmc_key else (None, None)
    out = dict(workload=wl.describe(name), interactions=wl.nnz, steps=steps, warmup=warmup, ms_per_step=elapsed / steps * 1e3,
               value=wl.nnz / (elapsed / steps), unit='interactions/s', dtype='f32' if dtype == 'f32' else 'bf16 storage / f32 arithmetic',
               kernels=roofline_report(models, prof, pmc), traffic_source=pmc_src, prep_seconds=wl.prep_seconds,
               epoch_hbm_bytes=sum(k['hbm'] for k in models.values()),
               epoch_hbm_frac=epoch_hbm_frac(models, elapsed / steps),
               hbm_gib_peak=torch.cuda.max_memory_allocated() / 2 ** 30)
    del wl
    torch.cuda.empty_cache()
    return out


XGMI_LINKS, XGMI_LINK_BPS = 7, 153e9   # per GPU: 7 point-to-point links x ~153 GB/s (the task's hardware notes)


def strong_scaling_projection(args, dev, full_ms, full_kernel_ms, G=8, steps=10, warmup=10):
    """What ONE rank of the G-way strong split of this workload does, timed on this one GPU, and the 1 -> G scaling it
    projects to.  NOT a measured N > 1 run: the shard's compute (local passes with the raw-gradient epilogue for V,
    tmf_adam_fresh_rows on the rank's 1/G of V) is measured; the wire time of the two collectives is MODELLED from their byte
    counts over 7 xGMI links; the 1-rank RCCL calls are measured only as a call floor (no peer: a device-local copy).
    The reference loop being split: /root/reference/src/teamoflow/mf/matrix_factorization.py:128-176.
    Ten warm-up epochs (120 ms) before ten timed ones: the card idles for a second or two while the shard is prepared on the host.
    (That was not why round 4 saw the hinge kernel at 1.02 ms on the shard against 0.57 ms for the same users as a problem of their
    own - it is 1.00 ms after ten warm-up epochs too; the kernel's time depends on how the scores cluster, profiles/r05_c5_shard.txt.)"""
    out = dict(label=f'PROJECTION from one GPU - no N > 1 run has been measured; G = {G}', G=G, full_problem_ms=full_ms, shards={})
    worst, worst_kernels = 0.0, {}
    for rank in (0, G - 1):
        wl = Workload(args, args.users, args.items, args.nnz, args.r, args.samples, args.loss, args.dtype, rank, G, dev, strong=True)
        backend = tdist.HipBackend(wl.st, args.loss, wl.c, wl.adam, prof=None)
        st, rows = wl.st, wl.n_pad // G
        prof = _engine.KernelTimer()

        def step(p):
            backend.prof = p
            gV, _ = backend.local_passes()
            if p:
                p.start('v_table_copy_and_adam_shard')
            st.V_nxt.copy_(st.V)                                   # stands in for the local part of the all-gather
            backend.adam_rows(st.V_nxt[rank * rows:(rank + 1) * rows], gV[rank * rows:(rank + 1) * rows])
            if p:
                p.stop('v_table_copy_and_adam_shard')
            backend.finish()
        for _ in range(warmup):
            step(None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(prof)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        kern = {k: prof.mean_ms(k) for k in prof.spans}
        out['shards'][f'rank{rank}'] = dict(users=wl.m, interactions=wl.nnz, ms_per_epoch=ms, kernels_ms=kern)
        if ms > worst:
            worst, worst_kernels = ms, kern
        del wl, backend, st
        torch.cuda.empty_cache()
    # collectives: bytes every rank puts on the wire, spread over its 7 links (reduce-scatter and all-gather are all-to-all
    # shaped on a fully connected xGMI node: (G-1)/G of the table leaves through G-1 <= 7 links at once)
    ld = _lib.padded_ld(args.r, torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    n_pad = tdist.padded_rows(args.items, G)
    rs_bytes = (G - 1) / G * n_pad * ld * 4
    ag_bytes = (G - 1) / G * n_pad * ld * (2 if args.dtype == 'bf16' else 4)
    links = min(G - 1, XGMI_LINKS)
    wire_ms = (rs_bytes + ag_bytes) / (links * XGMI_LINK_BPS) * 1e3
    floor = None
    try:
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29578')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        a = torch.zeros(n_pad, ld, device=dev)
        b = torch.empty_like(a)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(3):   # the first call builds the communicator
            ev[0].record(); dist.reduce_scatter_tensor(b, a); ev[1].record()
            ev[2].record(); dist.all_gather_into_tensor(a, b); ev[3].record()
        torch.cuda.synchronize()
        floor = dict(reduce_scatter_ms=ev[0].elapsed_time(ev[1]), all_gather_ms=ev[2].elapsed_time(ev[3]),
                     note='RCCL with ONE rank: no peer, a device-local copy of the whole table - a floor for the call, not a wire time')
        if own:
            dist.destroy_process_group()
    except Exception as e:   # the projection does not depend on it
        floor = dict(error=repr(e))
    call_ms = (floor.get('reduce_scatter_ms', 0.0) + floor.get('all_gather_ms', 0.0)) if floor else 0.0
    t_G = worst + wire_ms + call_ms
    # what does not shrink with G: per kernel, the shard's time beyond 1/G of the full problem's
    fixed = {k: dict(full_ms=full_kernel_ms.get(k), shard_ms=v, beyond_ideal_ms=(v - full_kernel_ms[k] / G) if k in full_kernel_ms else v)
             for k, v in worst_kernels.items()}
    out.update(slowest_shard_ms=worst, wire_ms_modelled=wire_ms, wire_model=f'(reduce-scatter {rs_bytes / 1e6:.1f} MB + all-gather {ag_bytes / 1e6:.1f} MB per rank) '
               f'/ ({links} links x {XGMI_LINK_BPS / 1e9:.0f} GB/s)', rccl_one_rank_floor=floor, projected_ms_per_epoch=t_G,
               projected_scaling=full_ms / t_G, ideal_scaling=G, per_kernel_beyond_ideal=fixed,
               fixed_costs_note='beyond_ideal_ms > 0: the part of a kernel that does not shrink with G - the item pass still walks all '
                                'items (one partial row per (user block, item) list, V-sized slab + combine over the whole catalog), '
                                'the Adam step / table copy touch V-sized data, short per-user ranges lose their slice efficiency')
    return out


def api_fit(dev, wl, args, epochs, shard_items=0):
    """The same workload through the PUBLIC class surface: MatrixFactorization(...).fit(epochs) - plan build and epoch loop
    timed by the model itself (plan_seconds_, fit_seconds_).  shard_items = q > 0: the item-row-sharded form (one rank
    streaming its own V in q windows, two staged at a time - what every rank of a sharded job runs, minus the wire)."""
    from teamoflow_amd.mf.initializer_graphs import FixedInitializer
    from teamoflow_amd.mf.loss_graphs import MSELoss, WMRBLoss
    from teamoflow_amd.mf.matrix_factorization import MatrixFactorization
    from teamoflow_amd.mf.sparse import SparseInteractions, eye
    kw = dict(user_weight_graph=FixedInitializer(wl.U0), item_weight_graph=FixedInitializer(wl.V0[:wl.n]))
    if wl.loss == 'wmrb':
        kw.update(loss_graph=WMRBLoss(), n_users=wl.m, n_items=wl.n, n_samples=wl.S)
    else:
        kw.update(loss_graph=MSELoss())
    model = MatrixFactorization(wl.r, **kw)
    model.verbose, model.shard_items = False, shard_items
    if wl.loss == 'wmrb':
        model.random_ind = wl.R
    model.fit(epochs, eye(wl.m), eye(wl.n), SparseInteractions(wl.idx, wl.val, (wl.m, wl.n)), lr=args.lr)
    out = dict(epochs=epochs, plan_seconds=model.plan_seconds_, epoch_loop_seconds=model.fit_seconds_,
               ms_per_epoch=model.fit_seconds_ / epochs * 1e3, loss_first_last=[model.loss_history_[0], model.loss_history_[-1]])
    del model
    torch.cuda.empty_cache()
    return out


def sharded_run(args, rank, world, dev, rehearse, red_dev, json_out):
    """--shard-items Q: the same workload with the item table row-sharded over the ranks (every rank owns 1/world of every
    catalog window; dist.ItemShardedEpoch).  Weak scaling: every rank its own users.  Same timed-region rules as the main path."""
    from teamoflow_amd import _windowed
    m, n, r, S = args.users, args.items, args.r, args.samples
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    esz = 2 if args.dtype == 'bf16' else 4
    T = world * args.shard_items
    ld = _lib.padded_ld(r, dtype)
    idx, val = gen_interactions(m, n, args.nnz, args.item_dist, rank, dev)
    U0 = init_table(m, r, 11 + rank, dev)
    rows, k, n_pad = _windowed.window_geometry(n, T, ld, esz, world)
    V_own = torch.zeros(T * rows // world, ld, dtype=dtype, device=dev)
    V_all = init_table(n, r, 7, dev)                                    # same seed on every rank; only the owned rows are kept
    for l0, g0, cnt in tdist.owned_blocks(rows, T, world, rank, n):
        V_own[l0:l0 + cnt, :r] = V_all[g0:g0 + cnt]
    del V_all
    R = random_sampler_device(n, m, S, seed=100 + rank, device=dev) if args.loss == 'wmrb' else None
    backend = _windowed.WindowedHipBackend(U0, V_own, idx, val, R, m, n, T, r, args.loss, n / S if args.loss == 'wmrb' else 0.0, args.lr,
                                           dtype=dtype, world=world)
    ep = tdist.ItemShardedEpoch(backend, backend.n_loss, T)
    multi = world > 1
    losses = torch.zeros(args.steps + args.warmup + 1, dtype=torch.float64, device=dev)

    def fence():
        torch.cuda.synchronize()
        if multi:
            torch.distributed.barrier()
            torch.cuda.synchronize()
    fence()
    tw = time.perf_counter()
    for i in range(args.warmup):
        losses[i] = ep.step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses[args.warmup + i] = ep.step()
    fence()
    elapsed, warm = time.perf_counter() - t0, t0 - tw
    nnz_total = float(val.numel())
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
        tot = torch.tensor([nnz_total], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(tot)
        nnz_total = float(tot)
    if rank != 0:
        return
    sec = elapsed / args.steps
    rows_gathered = (3 if args.loss == 'wmrb' else 2) * (m * S * (args.loss == 'wmrb') + float(val.numel()))   # per rank and epoch
    gather = rows_gathered * ld * esz
    walks = 2 if args.loss == 'wmrb' else 1
    lh = losses[:args.steps + args.warmup].cpu().tolist()
    out = dict(metric='train_interactions_per_sec', value=nnz_total / sec, unit='interactions/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=sec * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32' if args.dtype == 'f32' else 'bf16 storage / f32 arithmetic',
               data='synthetic' + (' (REHEARSAL: all ranks on one card, host-staged gloo collectives - timings invalid)' if rehearse else ''),
               config=dict(workload=f'{m} users x {n} items per GPU, r={r}, {args.loss.upper()}' + (f' S={S}' if args.loss == 'wmrb' else '')
                           + f', ITEM ROWS SHARDED over {world} rank(s): {T} windows of {rows} rows ({k} slices each), every rank owns '
                             f'{rows // world} rows of every window',
                           interactions_per_gpu=int(val.numel()), interactions_total=nnz_total, parallelism=f'user-partition dp{world} + item-row-sharded V',
                           lr=args.lr, warmup_ms_per_step=warm / max(args.warmup, 1) * 1e3),
               roofline=dict(bound='l2', kernel='whole windowed epoch (per-kernel timers are not wired on this path)', achieved=gather / sec / 1e9,
                             peak=L2_PEAK / 1e9, unit='GB/s', frac=gather / sec / L2_PEAK, traffic=None,
                             note='gathered factor-row bytes of the epoch (every row counted once per gather) over the epoch time against the aggregate L2 rate'),
               collectives=dict(ranks=world, backend=torch.distributed.get_backend() if multi else None, walks_per_epoch=walks,
                                all_gather_bytes_received_per_rank=walks * (world - 1) * (T * rows // world) * ld * esz,
                                reduce_scatter_bytes_sent_per_rank=(world - 1) * (T * rows // world) * ld * 4,
                                note='per window: all-gather of the rows (async, prefetched one window ahead), reduce-scatter of the fp32 '
                                     'gradient (async); overlapped with the window\'s kernels'),
               loss_first_last=[lh[0], lh[-1]])
    out['extras'] = write_extras(out)
    print(compact_line(out), file=json_out, flush=True)


def launcher_command(n, argv, port=None):
    """The command line of the N ranks of `bench.py --gpus N` (one process per GPU over RCCL; the reference has no launcher:
    /root/reference/src/teamoflow/mf/matrix_factorization.py:96 is one process).  127.0.0.1 rendezvous: a container's hostname may not resolve."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the ranks as a CHILD process,
    pass rank 0's JSON line through and return the child's exit code.  Nothing in this process has touched the GPU at this
    point and nothing will (no torch.cuda call, no _lib.get()): the parent only waits.  Returns None when this process IS a
    rank (or N = 1) and should run the bench itself."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    import subprocess
    rehearse = os.environ.get('TMF_BENCH_REHEARSE') == '1'
    have = torch.cuda.device_count()   # counts devices without initialising the GPU on this image
    if have < args.gpus and not rehearse:
        raise SystemExit(f'--gpus {args.gpus} but this node shows {have} GPU(s) (TMF_BENCH_REHEARSE=1 puts all ranks on card 0 for a functional rehearsal)')
    assert not torch.cuda.is_initialized(), 'the launcher process must not have initialised the GPU'
    cmd = launcher_command(args.gpus, argv)
    log('[bench] no launcher around --gpus %d: starting %s' % (args.gpus, ' '.join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in child.stdout if ln.strip()]
    rc = child.wait()
    found = [ln for ln in lines if ln.lstrip().startswith('{"metric"')]
    for ln in lines:   # anything else a rank wrote to stdout goes to stderr: the contract is ONE line on stdout
        if ln not in found[-1:]:
            log(ln.rstrip())
    if found:
        sys.stdout.write(found[-1] if found[-1].endswith('\n') else found[-1] + '\n')
        sys.stdout.flush()
    elif rc == 0:
        log('[bench] the ranks exited 0 without a JSON line')
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                    help='N > 1: strong (default) = the ONE --users x --items problem split over the ranks - BASELINE config 4 as '
                         'written; weak = every rank its own --users users (labelled as such in the JSON line)')
    ap.add_argument('--no-projection', action='store_true', help='skip the one-GPU timing of the 8-way strong-scaling shard')
    ap.add_argument('--users', type=int, default=1_000_000)
    ap.add_argument('--items', type=int, default=100_000)
    ap.add_argument('--rank', type=int, default=128, dest='r')
    ap.add_argument('--nnz', type=int, default=100_000_000)
    ap.add_argument('--samples', type=int, default=1024)
    ap.add_argument('--loss', choices=['wmrb', 'mse'], default='wmrb')
    ap.add_argument('--item-dist', choices=['zipf', 'uniform'], default='zipf')
    ap.add_argument('--lr', type=float, default=0.1)
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32', help='factor storage (arithmetic is fp32 either way)')
    ap.add_argument('--small-configs', action='store_true', help='time BASELINE configs 1-3 in full (GPU fit vs dense CPU restatement)')
    ap.add_argument('--no-extras', action='store_true', help='skip cpu baseline / predict / API / HBM legs / small configs')
    ap.add_argument('--shard-items', type=int, default=0, metavar='Q',
                    help='item-row-sharded V (BASELINE config 4 as written): Q windows per rank, per-window all-gather / reduce-scatter '
                         '(dist.ItemShardedEpoch) instead of the replicated table; one JSON line with an epoch-level roofline entry')
    ap.add_argument('--no-legs', action='store_true', help='skip the two HBM-streaming side legs (C4 MSE, config-5 shard)')
    args = ap.parse_args()
    rc = spawn_ranks(args, sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)

    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner at communicator
    # creation) write to file descriptor 1 as well, so keep a private copy of stdout for the JSON line and point
    # fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: start it with --nproc-per-node {args.gpus}, or '
                         'without a launcher (bench.py then starts its own ranks)')
    # TMF_BENCH_REHEARSE=1: every rank on card 0 with a gloo group (dist.py stages the collectives through the host) -
    # a functional rehearsal of the N>1 path on a one-GPU box; its timings mean nothing
    rehearse = os.environ.get('TMF_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dp_mode = world > 1 or os.environ.get('TMF_BENCH_FORCE_DP') == '1'  # the env knob rehearses the N>1 path on one GPU
    if dp_mode:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29577')
            torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        elif rehearse:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=dev)
    red_dev = 'cpu' if rehearse else dev   # where the scalar reductions of the report live
    _lib.get()

    if args.shard_items > 0:
        sharded_run(args, rank, world, dev, rehearse, red_dev, json_out)
        if dp_mode:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    strong = args.scaling == 'strong'
    wl = Workload(args, args.users, args.items, args.nnz, args.r, args.samples, args.loss, args.dtype, rank, world, dev, strong)
    if rank == 0:
        log(f'[bench] prepared {wl.nnz} interactions, m={wl.m} n={wl.n} r={wl.r} S={wl.S} in {wl.prep_seconds:.1f} s; '
            f'{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak')
    dp = backend = None
    if dp_mode:
        backend = tdist.HipBackend(wl.st, args.loss, wl.c, wl.adam, prof=None)
        dp = tdist.DataParallelEpoch(backend, wl.plan.n_pos if args.loss == 'wmrb' else wl.nnz)
    elapsed, warm_elapsed, prof, loss_buf = run_steps(wl, args.steps, args.warmup, dp, backend)
    comm = None
    if dp_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
        tot = torch.tensor([float(wl.nnz)], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(tot)
        nnz_total = float(tot)
        comm = dp.comm_report()
    else:
        nnz_total = float(wl.nnz)
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline: one entry per kernel, the dominant one (by time) on top ----
    default_workload = (args.users, args.items, args.r, args.samples, args.nnz, args.loss, args.item_dist, args.dtype, world) == \
        (1_000_000, 100_000, 128, 1024, 100_000_000, 'wmrb', 'zipf', 'f32', 1)
    pmc, pmc_src = pmc_traffic() if default_workload else (None, 'only quoted on the default workload')
    models = wl.models()
    kernels = roofline_report(models, prof, pmc)
    top = kernels[0]
    roofline = dict(bound=top['bound'], kernel=top['kernel'], achieved=top['achieved'], peak=top['peak'], unit='GB/s', frac=top['frac'],
                    traffic=top.get('traffic'), hbm_traffic_frac=top.get('hbm_traffic_frac'), traffic_source=pmc_src,
                    traffic_over_compulsory=top.get('traffic_over_compulsory'), useful_hbm_frac=top.get('useful_hbm_frac'),
                    l2_hit_rate=top.get('l2_hit_rate'),
                    kernel_ms=top['ms'], kernels=kernels, csrc_sha=csrc_sha(),
                    epoch_hbm_bytes=sum(k['hbm'] for k in models.values()),
                    epoch_hbm_frac=epoch_hbm_frac(models, ms_per_step * 1e-3),
                    # SURVEY §8d's figure next to the L2 one, so that nobody reads `frac` as an HBM fraction: the epoch's algorithmic bytes
                    # (every gathered row counted) over its time against the 8 TB/s HBM peak - above 1 at C4 because the path is blocked
                    # so that the gathered rows are served by the L2s, not by HBM
                    epoch_algorithmic_bytes=survey_algorithmic_bytes(args.loss, wl.m, wl.n, wl.S, wl.nnz, wl.plan.n_pos, wl.r, 2 if args.dtype == 'bf16' else 4),
                    algorithmic_over_hbm_peak=survey_algorithmic_bytes(args.loss, wl.m, wl.n, wl.S, wl.nnz, wl.plan.n_pos, wl.r,
                                                                       2 if args.dtype == 'bf16' else 4) / (ms_per_step * 1e-3) / HBM_PEAK,
                    epoch_gather_bytes=sum(k['gather'] for k in models.values()),
                    epoch_gather_rate_GBps=sum(k['gather'] for k in models.values()) / (ms_per_step * 1e-3) / 1e9,
                    l2_gather_rate_measured_by_guide_GBps=[x / 1e9 for x in L2_GATHER_MEASURED],
                    note='bound=l2: the kernel is blocked so that the rows it gathers come from the XCD L2s; achieved = gathered row bytes / '
                         'kernel time against the 34.5 TB/s aggregate L2 rate (the guide measures 16.8-18.8 TB/s for L2-resident row '
                         'gathers). hbm_* = bytes that must cross HBM at least once against 8 TB/s; traffic = PMC fabric bytes of this '
                         'code version (csrc_sha) or null.')
    wplan = wl.wplan
    if wplan is not None:
        roofline.update(wmrb_user_pass_ms=prof.mean_ms('wmrb_user_pass'), wmrb_item_pass_ms=prof.mean_ms('wmrb_item_pass'),
                        wmrb_user_pass_form=(f'sliced: scores + hinge + gradU + finish kernels over {wplan.n_slices} item slice(s)'
                                             if wplan.sliced else 'fused single kernel'),
                        wmrb_item_lists_user_blocks=wplan.user_chunks)

    out = dict(metric='train_interactions_per_sec', value=nnz_total / (elapsed / args.steps), unit='interactions/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
               scaling=args.scaling, vs_baseline=None,
               dtype='f32' if args.dtype == 'f32' else 'bf16 storage / f32 arithmetic',
               data='synthetic' + (' (REHEARSAL: all ranks on one card, host-staged gloo collectives - timings invalid)' if rehearse else ''),
               config=dict(workload=wl.describe('C4' if (args.users, args.items, args.r) == (1_000_000, 100_000, 128) else 'custom')
                           + (f' (strong scaling: ONE problem over {world} ranks, users {wl.user_block[0]}..{wl.user_block[1]} of {args.users} on rank 0)'
                              if strong and world > 1 else '')
                           + (f' (WEAK scaling: every one of the {world} ranks has its own {args.users} users - not BASELINE config 4 as written)'
                              if not strong and world > 1 else ''),
                           interactions_per_gpu=wl.nnz, interactions_total=nnz_total, positives_per_gpu=wl.plan.n_pos,
                           parallelism=f'user-partition dp{world}', lr=args.lr, warmup_ms_per_step=warm_elapsed / max(args.warmup, 1) * 1e3),
               roofline=roofline)
    if args.loss == 'wmrb':   # hinge terms per second = positives x negatives per user / epoch time
        out['hinge_terms_per_sec'] = float(wl.plan.n_pos) * wl.S * world / (elapsed / args.steps)
    if comm is not None:
        out['collectives'] = comm

    if rank == 0 and not args.no_extras and world == 1:
        denom = wl.plan.n_pos if args.loss == 'wmrb' else wl.nnz
        losses = loss_buf[:args.steps + args.warmup].cpu().numpy() / denom
        out['loss_first_last'] = [float(losses[0]), float(losses[-1])]
        out['cpu_baseline'] = cpu_baseline(args.loss, wl.idx, wl.val, wl.R, wl.U0, wl.V0[:wl.n], wl.n, wl.S, args.lr)
        # context: what the reference's dense formulation of this epoch would cost - forward + two backward
        # matmuls of [m, n, r] - it cannot run at this size (the [m, n] score matrix alone is 4 m n bytes)
        out['cpu_baseline']['reference_dense_formulation'] = dict(flops_per_epoch=6.0 * wl.m * wl.n * wl.r,
                                                                  score_matrix_bytes=4.0 * wl.m * wl.n)
        # predict rows/s: stable top-10 over the full catalog, fused GEMM + top-k (no [m, n] matrix)
        Ue, Ve = wl.st.U[:, :wl.r], wl.st.V[:wl.n, :wl.r]
        rows = min(wl.m, 262144)

        def time_topk(arith):
            _ops.predict_topk(Ue[:rows], Ve, 10, clamp_negatives=True, arithmetic=arith)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _ops.predict_topk(Ue[:rows], Ve, 10, clamp_negatives=True, arithmetic=arith)
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        # what the class surface runs at this size (arithmetic='auto'): fp32 tables keep ALL 24 bits of every factor - on the
        # bf16 matrix cores as three exact bf16 planes and six exact plane products (tmf_predict_topk_split_f32) where that
        # kernel applies (width <= 128, k <= 32), else on the fp32 MFMA; bf16 tables on the bf16 MFMA.  The two-plane fp16
        # form (22 bits) is an opt-in approximation: timed below for the extras file, never the headline
        dt = time_topk(None)
        flops = 2.0 * rows * wl.n * wl.r
        planes = args.dtype != 'bf16' and _ops.split_topk_supported(wl.r, 10) and rows * wl.n >= _ops.SPLIT_MIN_SCORES
        chosen = 'split' if planes else ('bf16' if args.dtype == 'bf16' else 'fp32')
        out['predict_rows_per_sec'] = rows / dt
        out['predict_tflops'] = flops / dt / 1e12
        out['predict_arithmetic'] = chosen
        kernels = {'bf16': 'tmf_predict_topk_bf16 (bf16 tables: bf16 MFMA, fp32 accumulate; dense bf16 peak ~2500 TF)',
                   'fp32': 'tmf_predict_topk_f32 (fp32 MFMA, peak 157.3 TF)',
                   'split': 'tmf_predict_topk_split_f32 (fp32 tables split exactly into 3 bf16 planes, 6 plane products on v_mfma_f32_32x32x16_bf16, '
                            'fp32 accumulate: fp32-accurate; predict_tflops is fp32-EQUIVALENT, the matrix-core work is 6x that; dense bf16 peak '
                            '~2500 TF = 417 TF fp32-equivalent)',
                   'half2': 'tmf_predict_topk_half2_f32 (fp32 tables as 2 fp16 planes under per-user-row / per-table power-of-two scales, 3 plane '
                            'products on v_mfma_f32_32x32x16_f16, fp32 accumulate: 22 bits of every factor, values at the fp32 kernel\'s error; '
                            'predict_tflops is fp32-EQUIVALENT, the matrix-core work is 3x that; dense fp16 peak ~2500 TF = 833 TF fp32-equivalent)'}
        out['predict_note'] = (f'stable top-10 of U.V^T over all {wl.n} items for {rows} users, fused MFMA GEMM + top-k, as '
                               f'MatrixFactorization.retrieve_user_recs / recall_at_k run it: ' + kernels[chosen])
        if planes:
            sample = min(rows, 2048)
            ref = (Ue[:sample].double() @ Ve.double().T).clamp_min(0)
            norm = float(ref.abs().max())
            by = {}
            for arith, work in (('fp32', 1), ('split', 6), ('half2', 3)):
                t = dt if arith == chosen else time_topk(arith)
                v, i = _ops.predict_topk(Ue[:sample], Ve, 10, clamp_negatives=True, return_values=True, arithmetic=arith)
                by[arith] = dict(rows_per_sec=rows / t, tflops_fp32_equivalent=flops / t / 1e12, matrix_core_tflops=work * flops / t / 1e12,
                                 max_abs_err_over_max_score_vs_fp64=float((v.double() - torch.gather(ref, 1, i.long())).abs().max()) / max(norm, 1e-30),
                                 top10_rows_identical_to_fp64=float((i.long() == torch.topk(ref, 10, dim=1)[1]).all(1).float().mean()))
            del ref
            out['predict_by_arithmetic'] = dict(by, err_sample_users=sample,
                                                note="model.predict_arithmetic = 'fp32' | 'split' | 'half2' selects ('auto' never takes the 22-bit "
                                                     "'half2'); errors of the returned top-10 values against an fp64 product of the same tables")
            out['predict_top10_rows_identical_to_fp64'] = by[chosen]['top10_rows_identical_to_fp64']
            out['predict_fp32_mfma'] = dict(rows_per_sec=by['fp32']['rows_per_sec'], tflops=by['fp32']['tflops_fp32_equivalent'])
        got, want = recall_parity(dev)
        out['recall_at_10'] = dict(engine=got, oracle=want, abs_diff=abs(got - want),
                                   case='C1 golden fixture: ranking of the oracle-trained tables (450 epochs)')
        # the same epochs through the public API (fit re-initialises, so it runs warmup + steps epochs from the same start)
        engine_ms = (elapsed + warm_elapsed) / (args.steps + args.warmup) * 1e3
        del Ue, Ve
        wl.st = None   # its memory stays in torch's caching allocator: the model's tables and plans reuse it (no hipMalloc in plan_seconds)
        out['api_fit'] = api_fit(dev, wl, args, args.steps + args.warmup)
        out['api_fit'].update(engine_ms_per_epoch_same_epochs=engine_ms,
                              api_over_engine=out['api_fit']['ms_per_epoch'] / engine_ms)
        if default_workload:
            q = 4
            out['item_sharded_fit'] = api_fit(dev, wl, args, args.steps + args.warmup, shard_items=q)
            out['item_sharded_fit'].update(windows=q, over_resident=out['item_sharded_fit']['ms_per_epoch'] / out['api_fit']['ms_per_epoch'],
                                           note='model.shard_items = 4 on ONE rank: the catalog walked in 4 windows (twice for WMRB), per-window '
                                                'launches of the same kernels, user-gradient partials summed in window order; no collective runs')
    if rank == 0 and world == 1 and (args.small_configs or not args.no_extras):
        # the reference's own (dense, full-batch) formulation on the host cores next to the engine, BASELINE configs 1-3
        out['reference_formulation_cpu'] = small_configs(dev, quick=not args.small_configs)
        c2 = out['reference_formulation_cpu'].get('C2', {})
        if 'recall_at_10_abs_diff' in c2 and 'recall_at_10' in out:
            out['recall_at_10']['end_to_end_C2'] = dict(engine=c2['recall_at_10'], oracle=c2['recall_at_10_cpu_restatement'],
                                                        abs_diff=c2['recall_at_10_abs_diff'],
                                                        case='C2 (943 x 1682, r=32, MSE): 100 epochs trained by each side from the same start')
    if rank == 0 and world == 1 and not args.no_extras and not args.no_projection and default_workload:
        try:
            wl.st = None
        except NameError:
            pass
        torch.cuda.empty_cache()
        kms = {k: prof.mean_ms(k) for k in prof.spans}
        out['strong_scaling_projection'] = strong_scaling_projection(args, dev, ms_per_step, kms)
        log(f"[bench] projected 1 -> 8 scaling {out['strong_scaling_projection']['projected_scaling']:.2f}x "
            f"(slowest shard {out['strong_scaling_projection']['slowest_shard_ms']:.2f} ms, full problem {ms_per_step:.2f} ms)")
    if rank == 0 and world == 1 and not args.no_extras and not args.no_legs and default_workload:
        # workloads whose factor rows really come from HBM (the C4 tables sit in L2 / Infinity Cache): DESIGN.md §4
        del wl
        torch.cuda.empty_cache()
        out['hbm_legs'] = dict(
            c4_mse=hbm_leg(args, dev, 'C4 shape, MSE', 1_000_000, 100_000, 100_000_000, 128, 1024, 'mse', 'f32', 20, 5, 'c4_mse'),
            c5_shard_bf16=hbm_leg(args, dev, 'config-5 shard (1/8 of 10M x 1M)', 1_250_000, 1_000_000, 125_000_000, 256, 1024, 'wmrb',
                                  'bf16', 5, 2, 'c5_shard_bf16'))
    if rank == 0:
        out['extras'] = write_extras(out)
        log('[bench] roofline kernels: ' + json.dumps(_short({e['kernel']: dict(ms=e['ms'], bound=e['bound'], frac=e['frac'],
                                                                               traffic_over_compulsory=e.get('traffic_over_compulsory'),
                                                                               useful_hbm_frac=e.get('useful_hbm_frac'))
                                                             for e in out['roofline']['kernels']}, 4)))
        print(compact_line(out), file=json_out, flush=True)
    if dp_mode:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
