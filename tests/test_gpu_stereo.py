"""GPU parity: match::stereo::compute through the C ABI == CPU oracle. Float outputs (stereo_x_right, depths) are compared by
bit pattern: every float operation of the path is individually rounded on both sides (tolerance 0 ulp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from openvslam_amd import feature, match, synth
    return feature, match, synth


@pytest.mark.parametrize("rows,cols,nfeat,seed", [(376, 1241, 2000, 1), (376, 1241, 2000, 2), (240, 400, 500, 3)])
def test_stereo_compute_kitti_geometry(mods, oracle, rows, cols, nfeat, seed):
    """BASELINE config 3: 1241x376 rectified pair, 2000 features per image, focal_x_baseline 386.1448 (KITTI 00-02)."""
    feature, match, synth = mods
    left, right, _ = synth.synth_stereo_pair(rows, cols, seed=seed)
    el = feature.orb_extractor(feature.orb_params(nfeat), max_rows=rows, max_cols=cols)
    er = feature.orb_extractor(feature.orb_params(nfeat), max_rows=rows, max_cols=cols)
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    oxl, oxr = oracle.OrbExtractor(oracle.make_params(nfeat)), oracle.OrbExtractor(oracle.make_params(nfeat))
    wkl, wdl = oxl.extract(left)
    wkr, wdr = oxr.extract(right)
    assert np.array_equal(kl.view(np.uint8), wkl.view(np.uint8)) and np.array_equal(dr, wdr)
    for fxb, b in ((386.1448, 0.5372), (60.0, 1.0)):   # second: max_disp = 60 px cuts the disparity window
        st = match.stereo(el, er, kl, dl, kr, dr, fxb, b)
        xr, dp = st.compute()
        wxr, wdp, wn = oracle.stereo_compute(oxl, oxr, wkl, wdl, wkr, wdr, fxb, b)
        assert st.num_valid_ == wn
        assert np.array_equal(xr.view(np.uint32), wxr.view(np.uint32)) and np.array_equal(dp.view(np.uint32), wdp.view(np.uint32))
        # ORACLE_SPEC rule 20's two L-tagged choices as run-time variants of BOTH sides: bit-equal in every setting, and not no-ops
        changed = 0
        for f21, pdbl in ((True, False), (False, True), (True, True)):
            st.set_variant("outlier_factor", int(f21))
            st.set_variant("parabola", int(pdbl))
            vxr, vdp = st.compute()
            oxr_, odp_, on = oracle.stereo_compute(oxl, oxr, wkl, wdl, wkr, wdr, fxb, b, outlier_factor_21=f21, parabola_double=pdbl)
            assert st.num_valid_ == on and np.array_equal(vxr.view(np.uint32), oxr_.view(np.uint32)) and np.array_equal(vdp.view(np.uint32), odp_.view(np.uint32))
            changed += not np.array_equal(vxr.view(np.uint32), xr.view(np.uint32))
            if f21 and not pdbl:
                assert on >= wn   # a larger factor keeps at least as many matches
        st.set_variant("outlier_factor", 0)
        st.set_variant("parabola", 0)
        assert changed >= 1
    assert wn > len(kl) // 10


def test_stereo_synthetic_keypoints_edge_cases(mods, oracle):
    """Keypoints placed by hand near the image borders / far octaves: exercises the window range checks and the octave filter
    (the extractor itself never emits such keypoints)."""
    feature, match, synth = mods
    rows, cols = 240, 400
    left, right, _ = synth.synth_stereo_pair(rows, cols, seed=9)
    el = feature.orb_extractor(feature.orb_params(300), max_rows=rows, max_cols=cols)
    er = feature.orb_extractor(feature.orb_params(300), max_rows=rows, max_cols=cols)
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    oxl, oxr = oracle.OrbExtractor(oracle.make_params(300)), oracle.OrbExtractor(oracle.make_params(300))
    oxl.extract(left)
    oxr.extract(right)
    rng = np.random.default_rng(0)
    kl, kr = kl.copy(), kr.copy()
    kl["x"][:20] = rng.uniform(0, 12, 20)           # left windows that leave the image
    kr["x"][:20] = rng.uniform(0, 14, 20)           # right windows that leave the image
    kr["x"][20:40] = cols - rng.uniform(0, 14, 20)
    kl["octave"][40:60] = 7                          # octave gaps > 1
    dr[:60] = dl[:60]                                # make them attractive matches
    kr["y"][:60] = kl["y"][:60]
    st = match.stereo(el, er, kl, dl, kr, dr, 386.1448, 0.5372)
    xr, dp = st.compute()
    wxr, wdp, wn = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372)
    assert st.num_valid_ == wn
    assert np.array_equal(xr.view(np.uint32), wxr.view(np.uint32)) and np.array_equal(dp.view(np.uint32), wdp.view(np.uint32))
