"""bench.py on the CPU: the ONE JSON line stays small enough for the driver (round 3's 20.5 KB line was cut: BENCH_r03 parsed =
null), the synthetic problem is per-user seeded (a rank of a strong-scaling job generates exactly its own block and the union
over the ranks IS the N = 1 problem), and roofline fractions are raw ratios with the binding roof named."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def bench():
    import bench as b
    return b


class StubTimer:
    def __init__(self, ms):
        self.ms, self.spans = ms, list(ms)

    def mean_ms(self, name):
        return self.ms.get(name, float('nan'))


def stub_out(bench):
    """A result dict of the size a full default run produces (per-kernel notes, legs, projection, per-arithmetic table)."""
    models = bench.wmrb_kernel_models(1_000_000, 100_000, 1024, 99_800_000, 99_800_000, 128, 4, 13, True, 1_123_800_000, 4_000_000, 27)
    prof = StubTimer(dict(wmrb_scores=26.6, wmrb_hinge=4.7, wmrb_gradu=25.0, wmrb_finish=1.25, wmrb_item_pass=30.6, wmrb_combine=1.3))
    pmc = {k: dict(bytes=float(v), bytes_uncorrected=float(v) * 0.6, l2_hit_rate=0.9, launches_per_epoch=1.0)
           for k, v in dict(wmrb_scores=32e9, wmrb_hinge=7.6e9, wmrb_gradu=34e9, wmrb_finish=7.7e9, wmrb_item_pass=114e9, wmrb_combine=8.5e9).items()}
    kernels = bench.roofline_report(models, prof, pmc)
    top = kernels[0]
    roofline = dict(bound=top['bound'], kernel=top['kernel'], achieved=top['achieved'], peak=top['peak'], unit='GB/s', frac=top['frac'],
                    traffic=top.get('traffic'), hbm_traffic_frac=top.get('hbm_traffic_frac'), traffic_source='profiles/pmc_c4_latest.json',
                    traffic_over_compulsory=top.get('traffic_over_compulsory'), useful_hbm_frac=top.get('useful_hbm_frac'),
                    l2_hit_rate=top.get('l2_hit_rate'), kernel_ms=top['ms'], kernels=kernels, csrc_sha='0123456789abcdef',
                    epoch_hbm_bytes=7.3e10, epoch_hbm_frac=0.1, algorithmic_over_hbm_peak=2.42, epoch_gather_bytes=1.7e12, note='x' * 700)
    filler = {f'key{i}': dict(a=1.23456789012345, b='y' * 300, c=list(range(40))) for i in range(40)}
    return dict(metric='train_interactions_per_sec', value=1.1123456789e9, unit='interactions/s', n_gpus=1, steps=20, warmup=5,
                ms_per_step=89.123456, higher_is_better=True, scaling='strong', vs_baseline=None, dtype='f32', data='synthetic',
                config=dict(workload='C4: 1000000 users x 100000 items per GPU, r=128, WMRB S=1024, item ids zipf-like, lognormal user degrees',
                            interactions_per_gpu=99_800_000, interactions_total=99_800_000.0, positives_per_gpu=99_800_000,
                            parallelism='user-partition dp1', lr=0.1, warmup_ms_per_step=90.0),
                roofline=roofline, hinge_terms_per_sec=1.1e12,
                cpu_baseline=dict(value=5.1e6, unit='interactions/s', cores=16, kind='port', sample='s' * 600,
                                  reference_dense_formulation=dict(flops_per_epoch=7.7e13, score_matrix_bytes=4e11)),
                predict_rows_per_sec=9.2e6, predict_tflops=236.0, predict_arithmetic='split', predict_note='n' * 900,
                predict_top10_rows_identical_to_fp64=1.0, predict_by_arithmetic=filler,
                recall_at_10=dict(engine=0.5, oracle=0.5, abs_diff=6e-8, case='C1', end_to_end_C2=dict(engine=0.1, oracle=0.1, abs_diff=5e-10, case='C2')),
                api_fit=filler, item_sharded_fit=filler, reference_formulation_cpu=filler, strong_scaling_projection=filler,
                hbm_legs=dict(c4_mse=dict(kernels=kernels, ms_per_step=10.79, value=9.25e9, unit='interactions/s', dtype='f32', epoch_hbm_frac=0.6, **filler),
                              c5_shard_bf16=dict(kernels=kernels, ms_per_step=237.4, value=5.26e8, unit='interactions/s',
                                                 dtype='bf16 storage / f32 arithmetic', epoch_hbm_frac=None, **filler)))


def test_the_bench_line_is_compact_and_complete(bench, tmp_path):
    out = stub_out(bench)
    assert len(json.dumps(out)) > 30_000            # the full result is far beyond what the driver keeps
    out['extras'] = bench.write_extras(out, str(tmp_path / 'extras.json'))
    text = bench.compact_line(out)
    assert '\n' not in text and len(text) < 4096, len(text)     # target 4 KB; the hard cap is asserted inside compact_line
    assert bench.LINE_HARD_CAP <= 8192
    line = json.loads(text)
    contract = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}
    assert contract <= set(line), contract - set(line)
    assert {'predict_rows_per_sec', 'predict_tflops', 'predict_arithmetic', 'recall_at_10', 'extras', 'hbm_legs'} <= set(line)
    # the HBM-streaming side legs are driver-parsed numbers now: epoch time, rate, per kernel [ms, roof, fraction]
    assert line['hbm_legs']['c4_mse']['value'] == 9.25e9 and len(line['hbm_legs']['c5_shard_bf16']['kernels_ms']) == 6
    assert set(line['hbm_legs']['c4_mse']) == {'ms_per_step', 'value', 'unit', 'dtype', 'epoch_hbm_frac', 'kernels_ms'}
    assert set(line) <= set(bench.LINE_KEYS)
    assert {'workload', 'interactions_total', 'parallelism'} <= set(line['config'])
    assert {'bound', 'kernel', 'kernel_ms', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'hbm_traffic_frac', 'traffic_source', 'csrc_sha',
            'epoch_hbm_frac', 'traffic_over_compulsory', 'useful_hbm_frac', 'algorithmic_over_hbm_peak'} <= set(line['roofline'])
    assert {'value', 'unit', 'cores', 'kind', 'sample'} == set(line['cpu_baseline'])
    assert line['roofline']['kernel'] == 'wmrb_item_pass' and len(line['roofline']['kernels_ms']) == 6
    assert abs(line['value'] - out['value']) < 1e-5 * out['value'] and line['recall_at_10']['abs_diff'] == 6e-8
    # nothing is lost: the extras file holds the whole result
    full = json.load(open(tmp_path / 'extras.json'))
    assert set(full) >= set(out) - {'extras'} and len(full['roofline']['kernels']) == 6


def test_an_oversized_line_sheds_its_optional_parts_rather_than_being_printed(bench):
    out = stub_out(bench)
    out['config']['workload'] = 'w' * 3000
    out['collectives'] = {f'c{i}': 1.5 for i in range(600)}
    text = bench.compact_line(out)
    assert len(text) <= bench.LINE_HARD_CAP
    line = json.loads(text)
    assert 'collectives' not in line and 'roofline' in line and 'cpu_baseline' in line


def test_roofline_fractions_are_raw_and_the_binding_roof_is_named(bench):
    """no clamp to 1.0; a measured fabric rate above the 6.29 TB/s copy ceiling is labelled 'fabric' and
    priced against the guide's 8.6 TB/s Infinity-Cache gather rate; every PMC-covered entry says how many times the
    compulsory bytes its traffic is."""
    # the config-5 shard of round 3: scores 446 GB in 62.9 ms, item pass 706 GB in 81.2 ms, gradU 616 GB in 78.3 ms
    m, n, S, nnz = 1_250_000, 1_000_000, 1024, 124_000_000
    models = bench.wmrb_kernel_models(m, n, S, nnz, nnz, 256, 2, 64, True, nnz + m * S, 70_000_000, 33, part_layers=1)
    prof = StubTimer(dict(wmrb_scores=62.9, wmrb_hinge=5.8, wmrb_gradu=78.3, wmrb_item_pass=81.2, wmrb_combine=11.8, wmrb_finish=1.6))
    pmc = {k: dict(bytes=v, bytes_uncorrected=v / 2, l2_hit_rate=0.44) for k, v in
           dict(wmrb_scores=446e9, wmrb_gradu=616e9, wmrb_item_pass=706e9, wmrb_combine=70e9).items()}
    e = {x['kernel']: x for x in bench.roofline_report(models, prof, pmc)}
    ip = e['wmrb_item_pass']
    assert ip['bound'].startswith('fabric') and ip['peak'] == 8600.0
    assert ip['achieved'] == pytest.approx(706e9 / 81.2e-3 / 1e9) and ip['frac'] == pytest.approx(ip['achieved'] / 8600.0)
    assert ip['hbm_traffic_frac'] == pytest.approx(706e9 / 81.2e-3 / 8e12) and ip['hbm_traffic_frac'] > 1.0    # raw, not clamped
    assert 6 < ip['traffic_over_compulsory'] < 10 and 30 < e['wmrb_scores']['traffic_over_compulsory'] < 40
    assert 3 < e['wmrb_gradu']['traffic_over_compulsory'] < 4
    for x in e.values():
        if 'traffic' in x:
            assert x['useful_hbm_frac'] == pytest.approx(x['hbm_bytes'] / (x['ms'] * 1e-3) / 8e12)
            assert x['kernel'] == 'wmrb_combine' or x['useful_hbm_frac'] < x['hbm_traffic_frac']
    # below the copy ceiling an HBM-streaming kernel stays on the HBM roof with its measured rate
    assert e['wmrb_combine']['bound'] == 'hbm' and e['wmrb_combine']['frac'] == pytest.approx(70e9 / 11.8e-3 / 8e12)
    assert e['wmrb_scores']['bound'].startswith('fabric')


@pytest.mark.parametrize('items', ['zipf', 'uniform'])
def test_generator_is_per_user_seeded(bench, items, monkeypatch):
    """The union over any user partition equals the whole problem bit for bit - interactions, values and negative table - so a
    rank never has to build more than its block (the 4-rank rehearsal that built 4 whole C4
    problems on one card and was killed at 300 s)."""
    from teamoflow_amd.mf.utils import random_sampler_device
    monkeypatch.setattr(bench, 'GEN_CHUNK_DRAWS', 5000)      # several generator chunks per block, cut at different users
    monkeypatch.setattr(bench, 'CALIBRATION_USERS', 256)
    m, n, target, S, dev = 1500, 400, 60_000, 24, torch.device('cpu')
    plan = bench.calibrated_degrees(m, n, target, items, 0, dev)
    idx, val = bench.gen_interactions(m, n, target, items, 0, dev, plan=plan)
    assert abs(len(val) - target) < 0.05 * target                                   # duplicates compensated (calibrated on a prefix)
    key = idx[:, 0] * n + idx[:, 1]
    assert bool((key[1:] > key[:-1]).all()) and int(idx[:, 0].max()) < m and int(idx[:, 1].max()) < n    # unique, row-major sorted
    assert set(val.unique().tolist()) <= {1.0, 2.0, 3.0, 4.0, 5.0}
    R = random_sampler_device(n, m, S, seed=100, device=dev)
    for bounds in ([0, 700, 1500], [0, 1, 399, 400, 1100, 1500], list(range(0, 1501, 125))):
        parts = [bench.gen_interactions(m, n, target, items, 0, dev, users=(b, e)) for b, e in zip(bounds[:-1], bounds[1:])]
        assert torch.equal(torch.cat([p[0] for p in parts]), idx) and torch.equal(torch.cat([p[1] for p in parts]), val)
        for (b, e), p in zip(zip(bounds[:-1], bounds[1:]), parts):
            assert p[0].numel() == 0 or (int(p[0][:, 0].min()) >= b and int(p[0][:, 0].max()) < e)
        Rp = [random_sampler_device(n, e - b, S, seed=100, device=dev, user_offset=b, rows_per_block=97) for b, e in zip(bounds[:-1], bounds[1:])]
        assert torch.equal(torch.cat(Rp), R)
    # another seed is another problem; the plan (degrees + calibration) is the same vector whoever computes it
    assert not torch.equal(bench.gen_interactions(m, n, target, items, 1, dev)[0][:1000], idx[:1000])
    assert torch.equal(bench.calibrated_degrees(m, n, target, items, 0, dev)[0], plan[0])
    if items == 'zipf':   # the power law is there: the most popular item has far more than the mean
        cnt = torch.bincount(idx[:, 1], minlength=n)
        assert int(cnt.max()) > 8 * len(val) / n


def test_strong_partition_is_balanced_on_drawn_degrees(bench):
    from teamoflow_amd import _engine
    from teamoflow_amd import dist as tdist
    m, n, target, dev = 20000, 5000, 600_000, torch.device('cpu')
    deg, _ = bench.calibrated_degrees(m, n, target, 'zipf', 0, dev)
    bounds = tdist.partition_users(_engine._excl_cumsum(deg), 8, per_user_cost=64)
    idx, _ = bench.gen_interactions(m, n, target, 'zipf', 0, dev)
    real = torch.bincount(idx[:, 0], minlength=m)
    cost = [int(real[b:e].sum()) + 64 * (e - b) for b, e in zip(bounds[:-1], bounds[1:])]
    assert max(cost) < 1.05 * (sum(cost) / 8), cost     # cut on DRAWN degrees, balanced on the realised (unique) ones to a few %


def test_gpus_n_without_a_launcher_starts_its_own_ranks_before_any_gpu_call(bench, monkeypatch, capsys):
    """the driver runs plain `python3 bench.py --gpus N`.  The parent must build the torch.distributed.run
    command, start it as a CHILD, relay rank 0's line and exit with the child's code - without a single GPU call of its own
    (a process that has initialised the GPU must never start another program by exec on this pool)."""
    import argparse
    import subprocess
    import torch

    def boom(*a, **k):
        raise AssertionError('GPU call in the launcher process')
    for name in ('set_device', 'synchronize', 'init', 'current_device', 'get_device_properties'):
        monkeypatch.setattr(torch.cuda, name, boom)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(bench._lib, 'get', boom)
    monkeypatch.setattr(bench._lib, 'load_library', boom)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('TMF_BENCH_REHEARSE', raising=False)
    seen = {}

    class Child:
        def __init__(self, cmd, stdout=None, env=None, text=None):
            seen.update(cmd=cmd, env=env, stdout=stdout)
            self.stdout = iter(['NCCL version 2.x banner\n', '{"metric":"train_interactions_per_sec","n_gpus":4}\n'])

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, 'Popen', Child)
    argv = ['--gpus', '4', '--steps', '3', '--warmup', '1']
    rc = bench.spawn_ranks(argparse.Namespace(gpus=4), argv)
    assert rc == 0
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-len(argv):] == argv
    assert os.path.samefile(cmd[-len(argv) - 1], os.path.join(ROOT, 'bench.py'))
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and seen['stdout'] == subprocess.PIPE
    out = capsys.readouterr()
    assert out.out == '{"metric":"train_interactions_per_sec","n_gpus":4}\n'      # ONE line on stdout, the banner went to stderr
    assert 'banner' in out.err
    # a failing child: its code is the parent's
    Child.wait = lambda self: 3
    assert bench.spawn_ranks(argparse.Namespace(gpus=4), argv) == 3
    # this process IS a rank (launcher around it) or N = 1: run the bench here
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert bench.spawn_ranks(argparse.Namespace(gpus=4), argv) is None
    monkeypatch.delenv('WORLD_SIZE')
    assert bench.spawn_ranks(argparse.Namespace(gpus=1), ['--gpus', '1']) is None
    # fewer cards than ranks: a clear refusal, not eight ranks fighting over one card
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    with pytest.raises(SystemExit, match='shows 1 GPU'):
        bench.spawn_ranks(argparse.Namespace(gpus=4), argv)


def test_importing_bench_and_parsing_arguments_initialises_no_gpu():
    """The launcher path of `python bench.py --gpus N` runs in a fresh interpreter: importing bench.py must not touch the GPU."""
    import subprocess
    code = ("import sys, torch; sys.argv=['bench.py']; import bench; "
            "assert not torch.cuda.is_initialized() and bench._lib._lib is None; print('ok')")
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


def test_cpu_baseline_sample_is_selected_column_by_column(bench, monkeypatch):
    """the baseline's sample went through `idx[mask]` on the [1e8, 2] int64 list - the row-indexing trap of
    this PyTorch-ROCm build.  It goes through _engine.take_interactions now and the sample is checked."""
    calls = []
    real = bench._engine.take_interactions
    monkeypatch.setattr(bench._engine, 'take_interactions', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    idx, val = bench.gen_interactions(300, 50, 3000, 'zipf', 0, 'cpu')
    sidx, sval = bench.baseline_sample(idx, val, 100, 50)
    assert calls and sidx[:, 0].max() < 100 and len(sval) == int((idx[:, 0] < 100).sum())
    assert np.array_equal(sidx, idx[: len(sval)].numpy())
    # rows that are not the prefix (what the trap returns: right count, wrong rows) are refused
    monkeypatch.setattr(bench._engine, 'take_interactions', lambda i, v, keep, **k: (i[-int(keep.sum()):], v[-int(keep.sum()):]))
    with pytest.raises(AssertionError, match='row-major prefix'):
        bench.baseline_sample(idx, val, 100, 50)
