"""The C++ class shims (openvslam_amd/cpp: feature::orb_extractor, match::robust with upstream's signatures) produce the
oracle's results when driven the way tracking code drives them."""
import os
import subprocess

import numpy as np
import pytest

from openvslam_amd.synth import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "openvslam_amd", "cpp", "test_shim")


def test_shim_builds():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openvslam_amd", "cpp")])
    assert os.path.exists(SHIM)


@pytest.mark.gpu
def test_shim_matches_oracle(oracle, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openvslam_amd", "cpp")])
    rows, cols, nfeat = 480, 752, 1000
    a = synth_frame(rows, cols, seed=21)
    b = synth_frame(rows, cols, seed=21, shift=(4, 3), noise_seed=5)
    a.tofile(tmp_path / "a.raw")
    b.tofile(tmp_path / "b.raw")
    out = tmp_path / "out.bin"
    subprocess.check_call([SHIM, str(rows), str(cols), str(nfeat), str(tmp_path / "a.raw"), str(tmp_path / "b.raw"), str(out)])
    raw = out.read_bytes()
    na, nb, nm, r7, c7 = (int(v) for v in np.frombuffer(raw[:20], np.int32))
    off = 20
    ka = np.frombuffer(raw[off:off + 28 * na], np.uint8); off += 28 * na
    da = np.frombuffer(raw[off:off + 32 * na], np.uint8).reshape(na, 32); off += 32 * na
    kb = np.frombuffer(raw[off:off + 28 * nb], np.uint8); off += 28 * nb
    db = np.frombuffer(raw[off:off + 32 * nb], np.uint8).reshape(nb, 32); off += 32 * nb
    pairs = np.frombuffer(raw[off:off + 8 * nm], np.int32).reshape(nm, 2)
    ox = oracle.OrbExtractor(oracle.make_params(nfeat))
    wa, wda = ox.extract(a)
    wb, wdb = ox.extract(b)
    assert (r7, c7) == ox.level_image(7).shape
    assert np.array_equal(ka, wa.view(np.uint8)) and np.array_equal(da, wda)
    assert np.array_equal(kb, wb.view(np.uint8)) and np.array_equal(db, wdb)
    valid = np.array([(i % 10 != 3) and (i % 10 != 7) for i in range(nb)], np.uint8)
    assert np.array_equal(pairs, oracle.robust_brute_force_match(wda, wdb, valid, 0.9)) and nm > 100
