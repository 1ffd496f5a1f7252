"""CPU known-answer tests of the oracle's equirectangular optimiser edges (equirectangular_pose_opt_edge in pose_optimizer::optimize,
equirectangular_reproj_edge in local_bundle_adjuster::optimize; BASELINE configs[3] camera model). The GPU parity tests compare the HIP
path with these functions; here the functions themselves are pinned: exact recovery on noise-free data, the pose edge's Jacobian
against central differences of its own residual, outlier flags on planted outliers, behaviour at the seam and the poles."""
import numpy as np

from openvslam_amd.synth import equirect_project, synth_pose_frame_equirect


def test_pose_edge_recovers_the_pose_on_exact_observations(oracle):
    T0, obs, cols, rows, (Rt, tt, bad) = synth_pose_frame_equirect(oracle.POSE_OBS_DTYPE, 400, 3, outlier_frac=0.0, pose_err=1.0)
    pc = obs["pos_w"] @ Rt.T + tt
    obs["obs_x"], obs["obs_y"] = equirect_project(pc, cols, rows)
    T, out, nv = oracle.pose_optimize_equirect(T0, obs, cols, rows)
    assert nv == 400 and not out.any()
    assert np.abs(T[:, :3] - Rt).max() < 1e-9 and np.abs(T[:, 3] - tt).max() < 1e-9


def test_pose_edge_flags_planted_outliers_and_ignores_stereo_fields(oracle):
    T0, obs, cols, rows, (Rt, tt, bad) = synth_pose_frame_equirect(oracle.POSE_OBS_DTYPE, 1200, 4, outlier_frac=0.15, pose_err=1.5)
    T, out, nv = oracle.pose_optimize_equirect(T0, obs, cols, rows)
    assert (out == bad).mean() > 0.95 and nv == int((~out).sum())
    assert np.linalg.norm(T[:, 3] - tt) < 0.01
    o2 = obs.copy()
    o2["is_stereo"] = 1            # an equirectangular rig has no stereo keypoints: the fields must not matter
    o2["obs_x_right"] = 123.0
    T2, out2, nv2 = oracle.pose_optimize_equirect(T0, o2, cols, rows)
    assert np.array_equal(T, T2) and np.array_equal(out, out2) and nv == nv2


def test_pose_edge_seam_and_poles(oracle):
    """Landmarks within pixels of the +-180 degree seam and of the poles: the estimate still converges (no wrap-around is applied, so an
    observation whose noise carried it across the seam is simply an outlier: ORACLE_SPEC rule 26)."""
    T0, obs, cols, rows, (Rt, tt, bad) = synth_pose_frame_equirect(oracle.POSE_OBS_DTYPE, 900, 5, outlier_frac=0.0, pose_err=1.0, seam_frac=0.3,
                                                                   pole_frac=0.2)
    T, out, nv = oracle.pose_optimize_equirect(T0, obs, cols, rows)
    assert np.linalg.norm(T[:, 3] - tt) < 0.02 and np.abs(T[:, :3] - Rt).max() < 2e-3
    u, _ = equirect_project(obs["pos_w"] @ T[:, :3].T + T[:, 3], cols, rows)
    crossed = np.abs(obs["obs_x"] - u) > cols / 2
    assert out[crossed].all()      # every observation that ended up on the other side of the seam is flagged
    assert nv >= 900 - crossed.sum() - 60


def test_fewer_than_five_observations_return_zero(oracle):
    T0, obs, cols, rows, _ = synth_pose_frame_equirect(oracle.POSE_OBS_DTYPE, 4, 6)
    T, out, nv = oracle.pose_optimize_equirect(T0, obs, cols, rows)
    assert nv == 0 and np.array_equal(T, T0) and not out.any()


def test_local_ba_equirect_oracle_converges(oracle):
    from oracle import lba
    rng = np.random.default_rng(2)
    n_pose, n_pt, per, cols, rows = 6, 500, 220, 3840, 1920
    pts = rng.normal(size=(n_pt, 3))
    pts *= (rng.uniform(3.0, 9.0, n_pt) / np.linalg.norm(pts, axis=1))[:, None]
    poses = np.zeros((n_pose, 7))
    poses[:, 6] = 1.0
    poses[:, :3] = rng.normal(0, 0.5, (n_pose, 3))
    edges = np.zeros(n_pose * per, oracle.BA_EDGE_DTYPE)
    for i in range(n_pose):
        sel = rng.choice(n_pt, per, replace=False)
        u, v = equirect_project(pts[sel] + poses[i, :3], cols, rows)
        e = edges[i * per:(i + 1) * per]
        e["pose_idx"], e["point_idx"], e["obs_x"], e["obs_y"], e["inv_sigma_sq"] = i, sel, u, v, 1.0
    bad = rng.random(len(edges)) < 0.03
    edges["obs_y"][bad] += 60.0
    fixed = np.zeros(n_pose, np.uint8)
    fixed[:2] = 1
    p0, x0 = poses.copy(), pts + rng.normal(0, 0.01, pts.shape)
    p0[2:, :3] += rng.normal(0, 0.01, (n_pose - 2, 3))
    r = lba.local_ba_optimize_equirect(p0, fixed, x0, edges, cols, rows)
    # a landmark seen once more than it is constrained can absorb a planted outlier, and an inlier can sit above the gate after round 1
    # (level-1 edges keep their round-1 chi2): agreement, not identity
    assert (r["mono_outlier"] == bad).mean() > 0.98 and r["mono_outlier"][bad].mean() > 0.8
    assert r["info"][3] < 1e-3 * r["info"][0]        # exact observations: the inlier chi2 collapses
    assert np.abs(r["poses"][2:, :3] - poses[2:, :3]).max() < 1e-3
    assert np.array_equal(r["poses"][:2], p0[:2])
