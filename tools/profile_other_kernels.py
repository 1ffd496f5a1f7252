#!/usr/bin/env python3
"""Workload for rocprofv3: every kernel family that is NOT on bench.py's config-2 timed path, at its BASELINE config size, a few calls each --
config 3 (KITTI stereo: extract x2 + k_stereo_*), config 4 (3840x1920 / 4000 kp / 10 000 landmarks: k_grid_assign, k_window_lists,
k_list_resolve<Projection>), area / bow / fuse / Sim3 matchers (k_reproject_queries, k_fuse_best, k_cross_check, k_bow_lists), config 5
(k_ba_linearize, the local_ba_optimize loop), k_pose_optimize and k_bow_transform. Run under tools/gpu_profile_others.sh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from openvslam_amd import ba, bow, feature, match, synth  # noqa: E402
from openvslam_amd import _lib  # noqa: E402

N = int(os.environ.get("OVS_PROFILE_ITERS", "6"))
SF = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
LSF = float(np.log(np.float32(1.2)))

# ---- config 3: stereo
left, right, _ = synth.synth_stereo_pair(376, 1241, seed=1)
el = feature.orb_extractor(feature.orb_params(2000), max_rows=376, max_cols=1241)
er = feature.orb_extractor(feature.orb_params(2000), max_rows=376, max_cols=1241)
kl, dl = el.extract(left)
kr, dr = er.extract(right)
st = match.stereo(el, er, kl, dl, kr, dr, 386.1448, 0.5372)
for _ in range(N):
    st.compute()

# ---- config 4: projection::match_frame_and_landmarks at 3840x1920
k, d = synth.synth_keypoints(4000, 1920, 3840, seed=1)
lm = synth.synth_landmarks(k, d, 10000, 1920, 3840, seed=2, n_from_frame=5200)
gp = match.grid_params(3840, 1920)
pm = match.projection(0.8, True, max_targets=4096, max_queries=10240)
for _ in range(N):
    pm.match_frame_and_landmarks(gp, k, d, SF, lm["xy"], lm["level"], lm["desc"], 5.0, lm_valid=lm["valid"])
# a 3840x1920 extraction as well (config 4's frame size)
big = synth.synth_frame(1920, 3840, seed=4)
eb = feature.orb_extractor(feature.orb_params(4000), max_rows=1920, max_cols=3840)
for _ in range(3):
    eb.extract(big)

# ---- area / bow / projection-with-pose / fuse / Sim3 on 752x480 and 1280x720 scenes (the shapes of tests/test_gpu_window.py)
from test_gpu_window import _last_and_current  # noqa: E402

a = synth.synth_frame(480, 752, seed=0)
b = synth.synth_frame(480, 752, seed=0, shift=(5, 0), noise_seed=4242)
ex = feature.orb_extractor(feature.orb_params(1000), max_rows=480, max_cols=752)
ka, da = ex.extract(a)
kb, db = ex.extract(b)
gp0 = match.grid_params(752, 480)
am = match.area(0.9, True, max_targets=2048, max_queries=2048)
prev0 = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
fa, fb = synth.synth_bow(da, seed=4, n_nodes=90), synth.synth_bow(db, seed=4, n_nodes=90)
bt = match.bow_tree(0.75, True, max_targets=2048, max_queries=2048)
for _ in range(N):
    am.match_in_consistent_area(gp0, ka, da, kb, db, prev0.copy(), 100)
    bt.match_keyframes(ka, da, fa, kb, db, fb, None, None)
rows, cols, n = 720, 1280, 2000
ck, cd, Tc, lk, lpw, ld, Tl, valid, (fx, fy, cx, cy) = _last_and_current(synth, 0, rows, cols, n, 5, 0.0)
cam = _lib.Camera(0, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
gp1 = match.grid_params(cols, rows)
pw = match.projection(0.9, True, max_targets=4096, max_queries=4096)
R, t = Tc[:, :3], Tc[:, 3]
dist = np.linalg.norm(lpw - (-R.T @ t), axis=1)
dmax = (dist * SF[np.clip(lk["octave"], 0, 7)] * 0.93).astype(np.float32)
dmm = np.ascontiguousarray(np.stack([(dmax / SF[7] * 0.5).astype(np.float32), dmax], 1))
nrm = (lpw - (-R.T @ t)) / dist[:, None]
ils = (1.0 / (SF * SF)).astype(np.float32)
fz = match.fuse(0.6, max_targets=4096, max_queries=4096)
for _ in range(N):
    pw.match_current_and_last_frames(cam, gp1, ck, cd, Tc, lk, lpw, ld, Tl, SF, 15.0, last_valid=valid)
    pw.match_frame_and_keyframe(cam, gp1, ck, cd, Tc, lk, lpw, dmm, ld, SF, LSF, 10.0, 100, kf_valid=valid)
    fz.replace_duplication(cam, gp1, ck, cd, Tc, lpw, dmm, nrm, ld, SF, ils, LSF, 3.0, lm_valid=valid)

# ---- config 5: local BA linearisation (device resident) + the full optimize loop; pose optimisation; BoW transform
d5 = synth.synth_local_ba(seed=0, pose_noise=0.03, point_noise=0.03)
for _ in range(N):
    ba.linearize(d5["poses"], d5["pose_fixed"], d5["points"], d5["edges"], d5["cam"], d5["huber_delta"])
ba.local_ba_optimize(d5["poses"], d5["pose_fixed"], d5["points"], d5["edges"], d5["cam"])
from oracle import binding as ob  # noqa: E402  (only for the record dtype of the synthetic pose frame)

T0, pobs, pcam, pbf, _ = synth.synth_pose_frame(ob.POSE_OBS_DTYPE, 2000, 7)
for _ in range(N):
    ba.pose_optimize(T0, pobs, pcam, pbf)
v = bow.vocabulary(synth.synth_vocabulary(k=10, depth=5, seed=3))
for _ in range(N):
    v.transform(dl, 4)
print("profile workload done")
