"""GPU parity: HIP orb_extractor (through the C ABI) == CPU oracle, bit for bit, stage by stage and end to end.
Oracle = from-spec restatement (PARITY UNPINNED vs upstream, see oracle/ovo_oracle.h)."""
import os

import numpy as np
import pytest

from openvslam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu

SIZES = [(1080, 1920, 2000), (480, 752, 1000), (376, 1241, 2000), (1920, 3840, 4000), (500, 644, 700)]   # BASELINE configs 2, 1, 3, 4 + an odd size


@pytest.fixture(scope="module")
def hip():
    from openvslam_amd import feature
    return feature


def _pair(hip, oracle, rows, cols, nfeat, seed=0, **kw):
    img = synth_frame(rows, cols, seed=seed, **kw)
    gp = hip.orb_params(max_num_keypts=nfeat)
    ex = hip.orb_extractor(gp, max_rows=rows, max_cols=cols)
    ox = oracle.OrbExtractor(oracle.make_params(nfeat))
    return img, ex, ox


@pytest.mark.parametrize("rows,cols,nfeat", SIZES)
def test_tables(hip, oracle, rows, cols, nfeat):
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=nfeat), max_rows=rows, max_cols=cols)
    t = oracle.orb_tables(oracle.make_params(nfeat))
    assert np.array_equal(ex.get_scale_factors(), t["scale_factors"])
    assert np.array_equal(ex.get_inv_scale_factors(), t["inv_scale_factors"])
    assert np.array_equal(ex.get_level_sigma_sq(), t["level_sigma_sq"])
    assert np.array_equal(ex.get_inv_level_sigma_sq(), t["inv_level_sigma_sq"])
    assert np.array_equal(ex.num_keypts_per_level_, t["num_keypts_per_level"])


@pytest.mark.parametrize("rows,cols,nfeat", SIZES)
def test_pyramid_and_candidates(hip, oracle, rows, cols, nfeat):
    img, ex, ox = _pair(hip, oracle, rows, cols, nfeat)
    ex.extract(img)
    ox.extract(img)
    for l in range(8):
        assert np.array_equal(ex.image_pyramid(l), ox.level_image(l)), "pyramid level %d" % l
    for l in range(8):
        gx, gy, gs = ex.debug_candidates(l)
        wx, wy, ws = ox.level_candidates(l)
        assert len(gx) == len(wx), "level %d candidate count %d vs %d" % (l, len(gx), len(wx))
        assert np.array_equal(gx, wx) and np.array_equal(gy, wy) and np.array_equal(gs, ws), "level %d candidates" % l


@pytest.mark.parametrize("rows,cols,nfeat", SIZES)
@pytest.mark.parametrize("seed", [0, 1])
def test_extract_bit_exact(hip, oracle, rows, cols, nfeat, seed):
    img, ex, ox = _pair(hip, oracle, rows, cols, nfeat, seed=seed)
    gk, gd = ex.extract(img)
    wk, wd = ox.extract(img)
    assert len(gk) == len(wk)
    assert np.array_equal(ex.debug_level_counts(), [ox.level_num_keypts(l) for l in range(8)])
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(gk[f], wk[f]), f
    # float tolerance stated: IC angle must be bit-identical (same op sequence, no FMA); tolerance 0
    assert np.array_equal(gk["angle"].view(np.uint32), wk["angle"].view(np.uint32))
    assert np.array_equal(gd, wd)


@pytest.mark.parametrize("factor,tie,taps,trig", [(1, 0, 0, 0), (3, 1, 0, 0), (3, 0, 1, 0), (1, 1, 1, 0), (3, 0, 0, 1), (1, 1, 1, 1)])
@pytest.mark.parametrize("rows,cols,nfeat", [(1080, 1920, 2000), (480, 752, 1000), (500, 644, 700)])
def test_extract_bit_exact_under_every_variant(hip, oracle, rows, cols, nfeat, factor, tie, taps, trig):
    """ORACLE_SPEC rules 6 (quad-tree switch factor), 7 (equal-count tie order), 10 (blur taps) and 11 (steering trigonometry) as run-time variants of BOTH sides
    (ovs_orb_set_variant / ovo_orb_set_variant): byte-equal keypoints and descriptors in every setting, and back to the defaults."""
    img, ex, ox = _pair(hip, oracle, rows, cols, nfeat, seed=2)
    gk0, gd0 = ex.extract(img)
    for e in (ex, ox):
        e.set_variant("tree_switch_factor", factor)
        e.set_variant("tree_tie_order", tie)
        e.set_variant("blur_taps", taps)
        e.set_variant("trig", trig)
    gk, gd = ex.extract(img)
    wk, wd = ox.extract(img)
    assert len(gk) == len(wk) and np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
    assert not (np.array_equal(gd, gd0) and len(gk) == len(gk0))   # the variants are not no-ops on this frame
    for e in (ex, ox):
        e.set_variant("tree_switch_factor", 3)
        e.set_variant("tree_tie_order", 0)
        e.set_variant("blur_taps", 0)
        e.set_variant("trig", 0)
    gk1, gd1 = ex.extract(img)
    assert np.array_equal(gk1.view(np.uint8), gk0.view(np.uint8)) and np.array_equal(gd1, gd0)
    with pytest.raises(RuntimeError):
        ex.set_variant("tree_switch_factor", 2)


def test_switch_factor_one_overshoot(hip, oracle):
    """tree_switch_factor = 1 lets the last all-at-once pass end far beyond N (dense corners: 64 -> 256 nodes when a level wants ~130): the
    outputs are sized 2 N + 3 per level in that variant (ovs_orb_max_keypoints changes) and stay byte-equal with the oracle."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    overshoot = 0
    for nfeat in (300, 420, 600, 900):
        ex = hip.orb_extractor(hip.orb_params(max_num_keypts=nfeat), max_rows=480, max_cols=640)
        ox = oracle.OrbExtractor(oracle.make_params(nfeat))
        cap3 = ex.max_keypoints
        for e in (ex, ox):
            e.set_variant("tree_switch_factor", 1)
        assert ex.max_keypoints >= cap3 and (nfeat < 900 or ex.max_keypoints > cap3)
        gk, gd = ex.extract(img)
        wk, wd = ox.extract(img)
        assert len(gk) == len(wk) and np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
        overshoot += len(gk) > nfeat + 3 * 8   # more than sum over levels of N + 3: some level's list passed its factor-3 capacity
        ex.set_variant("tree_switch_factor", 3)
        ox.set_variant("tree_switch_factor", 3)
        assert ex.max_keypoints == cap3
        gk, gd = ex.extract(img)
        wk, wd = ox.extract(img)
        assert len(gk) <= nfeat + 3 * 8 and np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
    assert overshoot >= 1   # at least one size really needed the larger capacity


@pytest.mark.parametrize("ini,mn", [(7, 5), (12, 3), (20, 7)])
def test_dense_cells(hip, oracle, ini, mn):
    """White noise at a low threshold: more than half of a cell's 4096 pixels pass the diameter pre-test (the pooled candidate list of a cell
    is at its fullest, the exact scoring runs many rounds per cell) next to a low-contrast patch whose cells fall back to min_fast_thr.
    Candidates, keypoints and descriptors stay byte-equal with the oracle."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 420), dtype=np.uint8)
    img[100:200, 150:300] = (120 + rng.integers(0, 24, (100, 150))).astype(np.uint8)   # a low-contrast patch: cells of mixed density
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=1500, num_levels=4, ini_fast_thr=ini, min_fast_thr=mn), max_rows=300, max_cols=420)
    ox = oracle.OrbExtractor(oracle.make_params(1500, 1.2, 4, ini, mn))
    gk, gd = ex.extract(img)
    wk, wd = ox.extract(img)
    assert len(gk) == len(wk) and np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
    dense = 0
    for level in range(4):
        gx, gy, gs = ex.debug_candidates(level)   # (sorted by the debug entry point as the oracle emits them)
        wx, wy, ws = ox.level_candidates(level)
        assert len(gx) == len(wx) and np.array_equal(gx, wx) and np.array_equal(gy, wy) and np.array_equal(gs, ws), "level %d candidates" % level
        dense += len(wx)
    assert dense > 3000   # (NMS survivors; the lists in front of them were several times longer)


def test_flat_and_tiny_images(hip, oracle):
    # flat image: no corners at either threshold -> zero keypoints; tiny image: levels without any cell
    for img in (np.full((480, 752), 77, np.uint8), synth_frame(120, 160, seed=3), synth_frame(64, 64, seed=4)):
        ex = hip.orb_extractor(hip.orb_params(max_num_keypts=500), max_rows=img.shape[0], max_cols=img.shape[1])
        ox = oracle.OrbExtractor(oracle.make_params(500))
        gk, gd = ex.extract(img)
        wk, wd = ox.extract(img)
        assert len(gk) == len(wk)
        assert np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
    ex = hip.orb_extractor(hip.orb_params(), max_rows=64, max_cols=64)
    k, d = ex.extract(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)


def test_low_texture_uses_min_threshold(hip, oracle):
    # weak texture: most cells have no FAST-20 corner and fall back to threshold 7
    rng = np.random.default_rng(5)
    img = np.clip(128 + rng.normal(0, 6, size=(480, 752)), 0, 255).astype(np.uint8)
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=1000), max_rows=480, max_cols=752)
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    gk, gd = ex.extract(img)
    wk, wd = ox.extract(img)
    assert len(wk) > 100
    assert np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)


def test_mask(hip, oracle):
    img = synth_frame(480, 752, seed=7)
    mask = np.full(img.shape, 255, np.uint8)
    mask[100:300, 200:500] = 0
    mask[:, 700:] = 0
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=1000), max_rows=480, max_cols=752)
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    gk, gd = ex.extract(img, mask)
    wk, wd = ox.extract(img, mask)
    assert len(gk) == len(wk) and len(wk) > 100
    assert np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)
    # rectangle masks from orb_params (Feature.mask_rectangles)
    ex2 = hip.orb_extractor(hip.orb_params(max_num_keypts=1000, mask_rects=[(0.2, 0.5, 0.1, 0.6)]), max_rows=480, max_cols=752)
    gk2, gd2 = ex2.extract(img)
    wk2, wd2 = ox.extract(img, ex2.create_rectangle_mask(752, 480))
    assert np.array_equal(gk2.view(np.uint8), wk2.view(np.uint8)) and np.array_equal(gd2, wd2)


@pytest.mark.parametrize("pipeline", [1, 2, 3])
def test_batch_dev_matches_host_api(hip, oracle, pipeline):
    """Device-resident batch (optionally issued as overlapping sub-batches on internal streams) == oracle, frame by frame."""
    import torch
    rows, cols, B = 480, 752, 5
    imgs = np.stack([synth_frame(rows, cols, seed=10 + b) for b in range(B)])
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=1000), max_rows=rows, max_cols=cols, max_batch=B)
    ex.set_pipeline(pipeline)
    cap = ex.max_keypoints
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    desc = d_desc.cpu().numpy()
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    for b in range(B):
        wk, wd = ox.extract(imgs[b])
        assert cnt[b] == len(wk)
        assert np.array_equal(kps[b, :cnt[b]].reshape(-1), wk.view(np.uint8).reshape(-1))
        assert np.array_equal(desc[b, :cnt[b]], wd)


def test_large_batch_takes_the_many_problem_tree_kernel(hip, oracle):
    """launch_tree runs 1024-thread workgroups when a launch holds <= 64 (level, frame) problems and 512-thread ones above: 70 small
    frames put both the level-0 launch and the level-1..7 launch on the 512-thread kernel (every other test here stays below 64)."""
    import torch
    rows, cols, B = 240, 320, 70
    imgs = np.stack([synth_frame(rows, cols, seed=200 + b) for b in range(B)])
    ex = hip.orb_extractor(hip.orb_params(max_num_keypts=400), max_rows=rows, max_cols=cols, max_batch=B)
    cap = ex.max_keypoints
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    for split in (True, False):
        ex.set_fast_split(split)
        ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy()
        kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
        desc = d_desc.cpu().numpy()
        ox = oracle.OrbExtractor(oracle.make_params(400))
        for b in range(0, B, 3):
            wk, wd = ox.extract(imgs[b])
            assert cnt[b] == len(wk)
            assert np.array_equal(kps[b, :cnt[b]].reshape(-1), wk.view(np.uint8).reshape(-1))
            assert np.array_equal(desc[b, :cnt[b]], wd)


def test_alternative_kernel_forms_give_the_same_bytes(tmp_path):
    """Round 6 left five process-wide switches between kernel forms (the fifth: OVS_PYR_PAIR=0, the pyramid level by level instead of two levels per launch;
    the other four: (read once per process, so each runs in a child): the one-wavefront-per-cell
    FAST (OVS_FAST_IMPL=2), the frames-fastest work order of rounds 3-5 (OVS_FAST_MAP=0), the quad-tree's sweep form only (OVS_TREE_GRID=0) and
    its grid form at a forced depth (OVS_TREE_GRID=3: most levels overflow and fall back inside the launch; 7: the deepest grid). Every one must
    reproduce the default's counts, keypoint records and descriptors byte for byte on a 70-frame batch (which the test above checks against
    the oracle)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import torch
from openvslam_amd import feature
from openvslam_amd.synth import synth_frame
rows, cols, B = 240, 320, 70
imgs = np.stack([synth_frame(rows, cols, seed=200 + b) for b in range(B)])
ex = feature.orb_extractor(feature.orb_params(max_num_keypts=400), max_rows=rows, max_cols=cols, max_batch=B)
cap = ex.max_keypoints
d_img = torch.from_numpy(imgs).cuda()
d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
h = hashlib.sha256()
for split in (True, False):
    ex.set_fast_split(split)
    ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    desc = d_desc.cpu().numpy()
    h.update(cnt.tobytes())
    for b in range(B):
        h.update(kps[b, :cnt[b]].tobytes())
        h.update(desc[b, :cnt[b]].tobytes())
print("SHA", h.hexdigest(), int(cnt.sum()))
"""
    out = {}
    for tag, env in (("default", {}), ("wave", {"OVS_FAST_IMPL": "2"}), ("frames_fastest", {"OVS_FAST_MAP": "0"}), ("sweeps", {"OVS_TREE_GRID": "0"}),
                     ("grid3", {"OVS_TREE_GRID": "3"}), ("grid7", {"OVS_TREE_GRID": "7"}), ("pyramid_per_level", {"OVS_PYR_PAIR": "0"})):
        r = subprocess.run([sys.executable, "-c", code % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("SHA")][-1].split()
        out[tag] = line[1]
        assert int(line[2]) > 1000
    for tag, sha in out.items():
        assert sha == out["default"], tag


@pytest.mark.parametrize("split", [True, False])
def test_timed_regime_1080p_batch_of_64_against_the_oracle(hip, oracle, split):
    """The regime bench.py times (SURVEY 8(d), configs[1]): a device-resident 1920x1080 batch of 64 frames, 2000 features -- six-cell FAST
    workgroups, XCD-sliced launches, k_tree<512> with ~10 k candidates per level-0 problem, level-0 split on and off. 16 of the 64 frames
    (every fourth; both 8-frame scene positions) are compared with the CPU oracle: 28-byte keypoint records and descriptors, byte for byte."""
    import torch
    from openvslam_amd.synth import synth_video
    rows, cols, B = 1080, 1920, 64
    imgs = synth_video(rows, cols, B, seed=321)
    ex = hip.orb_extractor(hip.orb_params(2000, 1.2, 8, 20, 7), max_rows=rows, max_cols=cols, max_batch=B)
    ex.set_fast_split(split)
    cap = ex.max_keypoints
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    for _ in range(2):   # twice: the second call runs on warm pools and recycled scratch, as every timed step does
        ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=s.cuda_stream)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    desc = d_desc.cpu().numpy()
    ox = oracle.OrbExtractor(oracle.make_params(2000), threads=min(8, os.cpu_count() or 1))
    for b in list(range(0, B, 4)) + [B - 1]:
        wk, wd = ox.extract(imgs[b])
        assert cnt[b] == len(wk) and len(wk) > 1500, b
        assert np.array_equal(kps[b, :cnt[b]].reshape(-1), wk.view(np.uint8).reshape(-1)), b
        assert np.array_equal(desc[b, :cnt[b]], wd), b


@pytest.mark.parametrize("rows,cols,nfeat,levels,scale", [(1080, 1920, 2000, 8, 1.2), (480, 752, 1000, 8, 1.2), (376, 1241, 2000, 8, 1.2), (500, 643, 700, 5, 1.5),
                                                          (1920, 3840, 4000, 8, 1.2), (301, 403, 300, 12, 1.1), (480, 640, 500, 4, 2.0)])
def test_one_launch_pyramid_equals_the_level_by_level_one(hip, oracle, rows, cols, nfeat, levels, scale):
    """k_pyramid_chain (all levels of a single frame in one launch: every workgroup chains its tile through the levels in LDS) against the
    level-by-level kernels and the oracle: every plane byte-equal, and the keypoints / descriptors that follow from them. Sizes: the four
    BASELINE geometries, odd widths, other scale factors / level counts (halo growth, tiny last levels), through the host entry and the
    device-batch entry with two frames in the launch (a stereo pair's shape) at a row pitch that is not the width."""
    import torch
    img = synth_frame(rows, cols, seed=77)
    gp = hip.orb_params(nfeat, scale, levels, 20, 7)
    ex = hip.orb_extractor(gp, max_rows=rows, max_cols=cols, max_batch=2)
    ox = oracle.OrbExtractor(oracle.make_params(nfeat, scale, levels, 20, 7))
    planes = {}
    for chain in (True, False):
        ex.set_pyramid_chain(chain)
        k, d = ex.extract(img)
        planes[chain] = ([ex.image_pyramid(l) for l in range(levels)], k, d)
    wk, wd = ox.extract(img)
    for l in range(levels):
        want = ox.level_image(l)
        assert np.array_equal(planes[True][0][l], want), ("chain", l)
        assert np.array_equal(planes[False][0][l], want), ("levels", l)
    for chain in (True, False):
        assert np.array_equal(planes[chain][1].view(np.uint8), wk.view(np.uint8)) and np.array_equal(planes[chain][2], wd)
    # device-batch entry, two frames, rows at the tensor's own (possibly unaligned) stride
    ex.set_pyramid_chain(2)
    img2 = synth_frame(rows, cols, seed=78)
    pitch = (cols + 3) // 4 * 4 + 4            # rows at a 4-byte aligned pitch that is not the width (the ABI's alignment rule)
    d_full = torch.zeros((2, rows, pitch), dtype=torch.uint8, device="cuda")
    d_full[:, :, :cols] = torch.from_numpy(np.stack([img, img2])).cuda()
    d_img = d_full[:, :, :cols]
    cap = ex.max_keypoints
    d_kps = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((2,), dtype=torch.int32, device="cuda")
    ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wk2, wd2 = ox.extract(img2)
    for l in range(levels):
        assert np.array_equal(ex.image_pyramid(l, frame=1), ox.level_image(l)), ("batch frame 1", l)
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(2, cap, 28)
    for b, (k_, d_) in enumerate(((wk, wd), (wk2, wd2))):
        assert cnt[b] == len(k_) and np.array_equal(kps[b, :cnt[b]].reshape(-1), k_.view(np.uint8).reshape(-1))
        assert np.array_equal(d_desc[b, :cnt[b]].cpu().numpy(), d_)


def test_two_extractors_run_concurrently_from_two_threads(hip, oracle):
    """Upstream extracts the left and right image of a stereo frame on two std::threads with two extractor instances: handles are
    independent (own stream, own buffers), so two host threads may call extract() on two handles at the same time."""
    import threading
    rows, cols = 376, 1241
    imgs = [synth_frame(rows, cols, seed=40 + i) for i in range(2)]
    exs = [hip.orb_extractor(hip.orb_params(max_num_keypts=2000), max_rows=rows, max_cols=cols) for _ in range(2)]
    results = [[None] * 6, [None] * 6]

    def work(k):
        for rep in range(6):
            results[k][rep] = exs[k].extract(imgs[k])

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ox = oracle.OrbExtractor(oracle.make_params(2000))
    for k in range(2):
        wk, wd = ox.extract(imgs[k])
        for rep in range(6):
            gk, gd = results[k][rep]
            assert np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd)


def test_error_statuses(hip):
    """The ABI never throws and never falls back: bad arguments and exceeded capacities come back as status codes."""
    import ctypes as C
    from openvslam_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    p = _lib.OrbParams(2000, 1.2, 8, 20, 7)
    assert L.ovs_orb_create(C.byref(p), 0, 640, 1, 0, C.byref(h)) == -1                      # OVS_ERR_INVALID
    assert L.ovs_orb_create(C.byref(p), 480, 640, 1, 99, C.byref(h)) == -2                   # OVS_ERR_NO_DEVICE
    bad = _lib.OrbParams(2000, 1.0, 8, 20, 7)
    assert L.ovs_orb_create(C.byref(bad), 480, 640, 1, 0, C.byref(h)) == -1                  # scale factor must exceed 1
    assert L.ovs_orb_create(C.byref(p), 480, 640, 1, 0, C.byref(h)) == 0
    img = np.zeros((600, 640), np.uint8)
    kps = np.zeros(4096 * 7, np.float32)
    desc = np.zeros((4096, 32), np.uint8)
    n = C.c_int32(-5)
    st = L.ovs_orb_extract(h, img.ctypes.data_as(C.c_void_p), 600, 640, 640, None, 0, kps.ctypes.data_as(C.c_void_p),
                           desc.ctypes.data_as(C.c_void_p), 4096, C.byref(n))
    assert st == -4 and n.value == 0                                                         # OVS_ERR_CAPACITY: taller than max_rows
    st = L.ovs_orb_extract(h, None, 0, 0, 0, None, 0, None, None, 0, C.byref(n))
    assert st == 0 and n.value == 0                                                          # empty image: upstream's early return
    assert L.ovs_orb_destroy(h) == 0
    m = C.c_void_p()
    assert L.ovs_matcher_create(70000, 100, 1, 0, C.byref(m)) == -1                          # indices are 16 bit
    assert L.ovs_wmatcher_create(60000, 60000, 1 << 20, 0, C.byref(m)) == -4                 # resolver state would not fit LDS


def test_more_than_1024_nodes_per_level(hip, oracle):
    """8000 features at 3840x1920: level 0 asks for ~1740 nodes, above the quad-tree's direct-ranking limit (kRankDirect = 1024 in
    csrc/orb_tree.hip), so the sorted phase takes its bitonic-sort path; order-exact against the oracle like every other size."""
    img, ex, ox = _pair(hip, oracle, 1920, 3840, 8000, seed=2)
    gk, gd = ex.extract(img)
    wk, wd = ox.extract(img)
    assert len(gk) == len(wk) and ox.level_num_keypts(0) > 1024
    for f in ("x", "y", "response", "octave"):
        assert np.array_equal(gk[f], wk[f]), f
    assert np.array_equal(gd, wd)


@pytest.mark.gpu
def test_extract_pair_equals_two_extracts():
    """ovs_orb_extract_pair (a stereo rig's two images as one batch of two): bit-identical to two ovs_orb_extract calls, with and without
    masks, on a size that is not a multiple of the tile sizes; refused with OVS_ERR_CAPACITY on a handle created for one frame and with
    OVS_ERR_INVALID for one mask out of two."""
    from openvslam_amd import feature
    from openvslam_amd.synth import synth_frame
    rows, cols = 486, 754
    a = synth_frame(rows, cols, seed=11)
    b = synth_frame(rows, cols, seed=11, shift=(7, 0), noise_seed=2)
    one = feature.orb_extractor(feature.orb_params(1000), max_rows=rows, max_cols=cols)
    two = feature.orb_extractor(feature.orb_params(1000), max_rows=rows, max_cols=cols, max_batch=2)
    m = np.ones((rows, cols), np.uint8)
    m[100:220, 300:500] = 0
    for masks in (None, (m, m)):
        want = [one.extract(img, None if masks is None else m) for img in (a, b)]
        got = two.extract_pair(a, b, *(masks or ()))
        for (wk, wd), (gk, gd) in zip(want, got):
            assert len(wk) > 300 and np.array_equal(wk, gk) and np.array_equal(wd, gd)
    again = two.extract_pair(a, b)
    assert np.array_equal(again[0][0], two.extract(a)[0])   # the pair path and the single-frame path share the handle
    with pytest.raises(RuntimeError):
        one.extract_pair(a, b)
    with pytest.raises(ValueError):
        two.extract_pair(a, b, m, None)


@pytest.mark.parametrize("rows,cols,levels,scale", [(1080, 1920, 8, 1.2), (480, 752, 8, 1.2), (376, 1241, 8, 1.2), (500, 643, 7, 1.2), (1920, 3840, 8, 1.2),
                                                    (301, 403, 12, 1.1), (333, 1027, 6, 1.25), (480, 640, 4, 2.0), (97, 4000, 5, 1.2), (2000, 131, 8, 1.2)])
def test_two_levels_per_launch_pyramid_planes(hip, oracle, rows, cols, levels, scale):
    """k_resize_pair_u8 (round 6: batches compute levels l + 1 and l + 2 from a staged rectangle of level l; the middle level is stored by the tile
    that owns each 4-pixel group, the halos are recomputed) against the oracle's cv::resize restatement: every plane of every frame of a
    three-frame batch byte-equal. Geometries: the four BASELINE sizes, odd widths and heights (ragged last tiles in both levels), very wide and very
    tall strips (one tile row / one tile column), 7 levels (three pairs, no single level left) and 8 (one left), scale factors whose pairs have no
    plan (2.0: the launcher falls back to one level per launch) and others that do (1.1, 1.25)."""
    import torch
    B = 3
    imgs = np.stack([synth_frame(rows, cols, seed=900 + b) for b in range(B)])
    ex = hip.orb_extractor(hip.orb_params(300, scale, levels, 20, 7), max_rows=rows, max_cols=cols, max_batch=B)
    ex.set_pyramid_chain(0)   # never the one-launch chain: this test is about the batch form
    ox = oracle.OrbExtractor(oracle.make_params(300, scale, levels, 20, 7))
    cap = ex.max_keypoints
    pitch = (cols + 3) // 4 * 4   # (the ABI's alignment rule; 16-byte aligned rows take the two-level kernel from level 0, others from level 1)
    d_full = torch.zeros((B, rows, pitch), dtype=torch.uint8, device="cuda")
    d_full[:, :, :cols] = torch.from_numpy(imgs).cuda()
    d_img = d_full[:, :, :cols]
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for b in range(B):
        ox.extract(imgs[b])
        for l in range(levels):
            got, want = ex.image_pyramid(l, frame=b), ox.level_image(l)
            assert got.shape == want.shape and np.array_equal(got, want), (b, l, np.argwhere(got != want)[:5].tolist() if got.shape == want.shape else got.shape)
