import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load


def rel_err(x, ref):
    """max |x - ref| / max |ref| (norm-wise relative error used by every parity test)."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-30))


def report_slack(**fields):
    """One line per check into gpurun_out/slack_report.jsonl (merged back from the GPU box) and onto stdout (pytest -s): how
    many hinge terms sat on the kink, how many elements needed the slack they grant, and how much of it was consumed - a
    regression hiding inside the slack shows up as these numbers growing."""
    import json
    line = json.dumps(fields)
    print('[slack]', line)
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'slack_report.jsonl'), 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def assert_close_with_slack(x, ref, slack=None, rtol=1e-5, what=''):
    """max_i (|x_i - ref_i| - slack_i) <= rtol * max|ref|: norm-wise agreement except for what the boundary hinge terms
    may legitimately move (oracle.sparse_ref.wmrb_slack / oracle_wmrb_boundary_slack).
    -> dict(n_over, max_consumed): the number of elements that only pass BECAUSE of their slack, and the largest share of
    its slack any of them used."""
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    if not x.size:
        return dict(n_over=0, max_consumed=0.0)
    sl = np.zeros_like(ref) if slack is None else 1.0001 * np.asarray(slack, np.float64)
    d0 = np.abs(x - ref)
    d = d0 - sl
    lim = rtol * max(float(np.abs(ref).max()), 1e-30)
    if float(d.max()) > lim:
        i = np.unravel_index(int(d.argmax()), d.shape)
        raise AssertionError(f'{what}: {int((d > lim).sum())} of {d.size} elements differ by more than {lim:.3g} + slack; worst at {i}: '
                             f'got {x[i]!r}, reference {ref[i]!r}, slack {sl[i]!r}')
    over = d0 > lim
    consumed = float(((d0[over] - lim) / sl[over]).max()) if over.any() else 0.0
    return dict(n_over=int(over.sum()), max_consumed=consumed)


def step_bounds(W0, g_ref, lr, rtol=1e-5, slack=None):
    """Interval every element of a factor table must fall in after ONE fresh-Adam step if the gradient the
    kernel summed is within ``rtol`` (norm-wise relative) of ``g_ref`` (fp64 closed form from the oracle).

    The reference's step is w -= alpha*(g*(1-b1)) / (sqrt(g*g*(1-b2)) + 1e-7), i.e. lr*g/(|g| + 3.16e-6)
     - monotone in g and nearly a sign function, so the elements whose gradient is within
    ~1e-5 of zero move by a visible fraction of lr under an fp32 reordering of the gradient sum while all
    others are insensitive to it.  Comparing against the interval [step(g + tol), step(g - tol)] is the
    exact statement of "the gradient agrees to rtol"; 1e-6 relative slack covers the fp32 rounding of the
    update itself.
    """
    W0 = np.asarray(W0, dtype=np.float64)
    g = np.asarray(g_ref, dtype=np.float64)
    f = np.float32
    omb1, omb2, eps = float(f(1) - f(0.9)), float(f(1) - f(0.999)), float(f(1e-7))
    alpha = float(f(f(lr) * np.sqrt(f(f(1) - f(0.999))) / f(f(1) - f(0.9))))

    def step(gg):
        return W0 - (gg * omb1 * alpha) / (np.sqrt(gg * gg * omb2) + eps)
    tol = rtol * max(float(np.abs(g).max()) if g.size else 0.0, 1e-30)
    if slack is not None:  # what switching the boundary hinge terms may add to / take from every element
        tol = tol + 1.0001 * np.asarray(slack, dtype=np.float64)
    ulp = 1e-6 * np.maximum(np.abs(W0), lr) + 1e-12
    return step(g + tol) - ulp, step(g - tol) + ulp


def assert_step(W_new, W0, g_ref, lr, rtol=1e-5, what='', slack=None):
    lo, hi = step_bounds(W0, g_ref, lr, rtol, slack)
    W = np.asarray(W_new, dtype=np.float64)
    bad = (W < lo) | (W > hi)
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f'{what}: {int(bad.sum())} of {bad.size} elements outside the step interval; first at '
                             f'{tuple(i)}: got {W[tuple(i)]!r}, interval [{lo[tuple(i)]!r}, {hi[tuple(i)]!r}], '
                             f'g_ref {np.asarray(g_ref)[tuple(i)]!r}')
