"""utils.random_sampler_device: the negative table of a model too large for the reference's host loop
(/root/reference/src/teamoflow/mf/utils.py:8-22 - one O(n_items) np.random.choice(replace=False) per user; the constructor
switches above HOST_SAMPLER_MAX_WORK, matrix_factorization.py:65-73).  Same distribution, another stream: every row holds
distinct items of the catalog, uniformly, in random order - checked here by the properties np.random.choice guarantees."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def chi2_z(counts, expected, var_scale=1.0):
    """(chi2 - dof) / sqrt(2 dof) of observed counts against a flat expectation; ~N(0, 1) under the hypothesis."""
    counts = np.asarray(counts, np.float64)
    chi2 = ((counts - expected) ** 2 / (expected * var_scale)).sum()
    dof = counts.size - 1
    return float((chi2 - dof) / np.sqrt(2.0 * dof))


def check_table(R, n_items, n_samples):
    assert R.dtype in (torch.int32, torch.int64) and tuple(R.shape[1:]) == (n_samples,)
    assert int(R.min()) >= 0 and int(R.max()) < n_items
    srt = torch.sort(R, dim=1)[0]
    assert not bool((srt[:, 1:] == srt[:, :-1]).any()), 'a row repeats an item (replace=False, utils.py:20)'


def test_device_sampler_rows_are_distinct_uniform_and_shuffled():
    from teamoflow_amd.mf.utils import random_sampler_device
    n, m, S = 1000, 40_000, 200
    R = random_sampler_device(n, m, S, seed=3)
    check_table(R, n, S)
    assert R.shape[0] == m
    # item frequencies over the whole table: m S / n each; sampling without replacement inside a row shrinks the variance
    # of a count by (1 - S / n)
    z = chi2_z(torch.bincount(R.flatten().to(torch.int64), minlength=n).cpu().numpy(), m * S / n, var_scale=1.0 - S / n)
    assert abs(z) < 5.0, z
    # every COLUMN is uniform too (the order inside a row is random, not sorted): first, middle and last column
    for c in (0, S // 2, S - 1):
        z = chi2_z(torch.bincount(R[:, c].to(torch.int64), minlength=n).cpu().numpy(), m / n)
        assert abs(z) < 5.0, (c, z)
    # pairs of columns are not ordered (a sorted row would give P = 1)
    assert 0.48 < float((R[:, 0] < R[:, 1]).float().mean()) < 0.52
    # rows differ from each other and from another seed; the same seed reproduces the table
    assert not torch.equal(R[0], R[1]) and torch.equal(R, random_sampler_device(n, m, S, seed=3))
    assert not torch.equal(R, random_sampler_device(n, m, S, seed=4))
    # n_samples == n_items: every row a permutation of the catalog; beyond it: the reference's ValueError (utils.py docstring :48)
    P = random_sampler_device(64, 100, 64, seed=1)
    assert torch.equal(torch.sort(P, dim=1)[0], torch.arange(64, device=P.device, dtype=P.dtype).expand(100, 64))
    with pytest.raises(ValueError):
        random_sampler_device(10, 5, 11)


def test_constructor_draws_on_the_device_above_the_host_limit():
    """generate_sample=True with n_users x n_items beyond HOST_SAMPLER_MAX_WORK: the ctor warns and draws the table with
    random_sampler_device; below it the global NumPy stream is used exactly like the reference (seed-reproducible)."""
    from teamoflow_amd.mf import matrix_factorization as mfm
    from teamoflow_amd.mf.loss_graphs import WMRBLoss
    m, n, S = 60_000, 50_000, 48
    assert m * n > mfm.HOST_SAMPLER_MAX_WORK
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        model = mfm.MatrixFactorization(8, loss_graph=WMRBLoss(), n_users=m, n_items=n, n_samples=S, generate_sample=True)
    assert any('random_sampler_device' in str(x.message) for x in w)
    R = model.random_ind
    assert R.is_cuda and tuple(R.shape) == (m, S)
    check_table(R, n, S)
    z = chi2_z(torch.bincount(R.flatten().to(torch.int64), minlength=n).cpu().numpy(), m * S / n, var_scale=1.0 - S / n)
    assert abs(z) < 5.0, z
    # below the limit: the reference's stream
    np.random.seed(7)
    small = mfm.MatrixFactorization(4, loss_graph=WMRBLoss(), n_users=30, n_items=40, n_samples=10, generate_sample=True)
    np.random.seed(7)
    want = np.array([np.random.choice(a=40, size=10, replace=False) for _ in range(30)])
    assert np.array_equal(small.random_ind.cpu().numpy(), want)


def test_bench_generator_blocks_equal_the_whole_problem_on_the_device():
    """bench.py's synthetic problem is per-user seeded (tests/test_bench_cpu.py on the CPU): on the device, with the transcendental
    functions of the GPU, the blocks four ranks would generate are the whole problem bit for bit - interactions, values, negatives."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from teamoflow_amd import _engine
    from teamoflow_amd import dist as tdist
    from teamoflow_amd.mf.utils import random_sampler_device
    dev = torch.device('cuda', 0)
    m, n, target, S = 40_000, 5_000, 2_000_000, 64
    plan = bench.calibrated_degrees(m, n, target, 'zipf', 0, dev)
    idx, val = bench.gen_interactions(m, n, target, 'zipf', 0, dev, plan=plan)
    assert abs(len(val) - target) < 0.02 * target
    bounds = tdist.partition_users(_engine._excl_cumsum(plan[0]), 4, per_user_cost=S)
    parts = [bench.gen_interactions(m, n, target, 'zipf', 0, dev, users=(b, e), plan=plan) for b, e in zip(bounds[:-1], bounds[1:])]
    assert torch.equal(torch.cat([p[0] for p in parts]), idx) and torch.equal(torch.cat([p[1] for p in parts]), val)
    R = random_sampler_device(n, m, S, seed=100, device=dev)
    Rp = [random_sampler_device(n, e - b, S, seed=100, device=dev, user_offset=b) for b, e in zip(bounds[:-1], bounds[1:])]
    assert torch.equal(torch.cat(Rp), R)
    # the CPU draws the same table (integer arithmetic only)
    assert torch.equal(random_sampler_device(n, 300, S, seed=100, device='cpu', user_offset=777), R[777:1077].cpu())
