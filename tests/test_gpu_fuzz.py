"""GPU: a fixed-seed slice of the randomised sweeps in profiles/tools/ (random heterogeneous graphs with hubs, empty
relations and multi-edges; random widths, heads and step counts; source slicing forced on small graphs) -- forward vs the
CPU oracle, single convs vs the float64 restatement, scoring entry points vs float64 torch, gradients vs float64 autograd.
The sharded sweep found a real defect once (profiles/README.md); it runs here at world 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))

pytestmark = pytest.mark.gpu


def _sweep(module, count, seed):
    mod = __import__(module)
    rng = np.random.default_rng(seed)
    failures = []
    for i in range(count):
        ok, desc = mod.one(rng, i)
        if not ok:
            failures.append(desc)
    for knob in ('PEA_SLICE_MIN_EDGES', 'PEA_SLICE_BYTES'):
        os.environ.pop(knob, None)
    assert not failures, '\n'.join(failures)


def test_forward_sweep():
    _sweep('fuzz_parity', 40, 101)


def test_hip_is_not_the_looser_side(capsys):
    """The per-row fallback of helpers.assert_fp32_close carries 16 fp32 ulps of the row's magnitude because the ratio of
    two independent fp32 realisations' row maxima exceeds 2 by chance, on either side.  That reading only holds while the
    kernels are no looser than the oracle; this pins it: over a fixed-seed slice of the fuzz configurations, every output
    vector of the stack scored against float64 on both sides (profiles/tools/error_symmetry.py), the HIP path's mean error
    is at most 5 % above the oracle's per conv kind, and it trips the bare `2x + atol` rule no more often than the oracle
    does (plus a small-sample allowance)."""
    import error_symmetry
    argv, sys.argv = sys.argv, ['error_symmetry.py', '60', '106']
    try:
        stats = error_symmetry.main()
    finally:
        sys.argv = argv
        for knob in ('PEA_SLICE_MIN_EDGES', 'PEA_SLICE_BYTES'):
            os.environ.pop(knob, None)
    assert stats
    for kind, st in stats.items():
        eh, eo = np.concatenate(st['eh']), np.concatenate(st['eo'])
        assert eh.mean() <= 1.05 * eo.mean(), (kind, eh.mean(), eo.mean())
        assert np.percentile(eh, 99.9) <= 1.25 * np.percentile(eo, 99.9) + 1.0, kind
        assert st['hip_loose'] <= st['orc_loose'] + max(20, st['n'] // 2000), (kind, st['hip_loose'], st['orc_loose'], st['n'])


def test_single_conv_sweep():
    _sweep('fuzz_convs', 40, 102)


def test_scoring_sweep():
    _sweep('fuzz_scoring', 30, 103)


def test_gradient_sweep():
    _sweep('fuzz_backward', 15, 104)


def test_sharded_sweep_world2():
    import fuzz_sharded
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded, args=(2, port, 12, 105), nprocs=2, join=True)


def _sharded(rank, world, port, count, seed):
    sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
    import fuzz_sharded
    fuzz_sharded.worker(rank, world, port, count, seed, strict=True)
