"""A short slice of the randomised parity campaign (tools/fuzz_parity.py) inside the GPU suite: every family once, fixed seeds. The full
campaign (~20 seeds x 240-360 cases, profiles/r02_fuzz_parity.txt) is run with the tool itself."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_slice(oracle, seed):
    import fuzz_parity as fz
    rng = np.random.default_rng(seed)
    lines = []
    assert fz.fuzz_extract(rng, 24, lines.append), lines[-8:]
    assert fz.fuzz_match(rng, 10, lines.append), lines[-4:]
    assert fz.fuzz_stereo(rng, 2, lines.append), lines[-2:]
    assert fz.fuzz_window(rng, 3, lines.append), lines[-4:]
    assert fz.fuzz_batch(rng, 4, lines.append), lines[-2:]
    assert fz.fuzz_optimize(rng, 4, lines.append), lines[-4:]
    assert fz.fuzz_reprojection(rng, 3, lines.append), lines[-4:]
    assert fz.fuzz_sim3(rng, 2, lines.append), lines[-4:]
    assert len(lines) > 50


@pytest.mark.parametrize("seed", [2024, 7])
def test_tie_heavy_small_cases(oracle, seed):
    """tools/tie_fuzz_gpu.py inside the suite: 150 small cases per seed whose outcome only the tie rules decide (three base descriptors, a 5-px
    lattice, six angles) through brute_force_match, area, both bow_tree matchers and projection::match_frame_and_landmarks, against the oracle:
    where a parallel resolver and a sequential loop would diverge if the replay order were wrong."""
    import tie_fuzz_gpu
    bad = tie_fuzz_gpu.run(150, seed)
    assert not any(bad.values()), bad
