"""Seeded synthetic inputs for the parity tests and bench.py (there is no dataset in the container).

Frame generator (config 2 of BASELINE.json): sum of 3 octaves of value noise (uniform[0,255] lattices at 8/32/128 px pitch,
bilinear-upsampled, weights 0.5/0.3/0.2), 300 axis-aligned rectangles (side 8..120 px, uniform grey) painted on top, then
i.i.d. N(0,3) pixel noise, clamped to u8. RNG: numpy PCG64 seeded with `seed`.
"""
import numpy as np


def _value_noise(rng, rows, cols, pitch):
    gr, gc = rows // pitch + 2, cols // pitch + 2
    lat = rng.uniform(0.0, 255.0, size=(gr, gc)).astype(np.float32)
    y = np.arange(rows, dtype=np.float32) / pitch
    x = np.arange(cols, dtype=np.float32) / pitch
    y0 = y.astype(np.int32)
    x0 = x.astype(np.int32)
    fy = (y - y0)[:, None]
    fx = (x - x0)[None, :]
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


PAD = 64   # scene margin: |shift| <= PAD


def synth_scene(rows=1080, cols=1920, seed=0, n_rect=300):
    """The noise-free float32 scene (rows+2*PAD, cols+2*PAD) frames are cropped from."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R, Cc = rows + 2 * PAD, cols + 2 * PAD
    img = 0.5 * _value_noise(rng, R, Cc, 8) + 0.3 * _value_noise(rng, R, Cc, 32) + 0.2 * _value_noise(rng, R, Cc, 128)
    for _ in range(n_rect):
        w = int(rng.integers(8, 121))
        h = int(rng.integers(8, 121))
        x = int(rng.integers(0, Cc - w))
        y = int(rng.integers(0, R - h))
        img[y:y + h, x:x + w] = float(rng.uniform(0.0, 255.0))
    return img


def frame_from_scene(scene, rows, cols, shift=(0, 0), noise_seed=0, noise_sigma=3.0):
    dx, dy = shift
    assert abs(dx) <= PAD and abs(dy) <= PAD
    img = scene[PAD + dy:PAD + dy + rows, PAD + dx:PAD + dx + cols]
    nrng = np.random.Generator(np.random.PCG64(noise_seed))
    img = img + nrng.normal(0.0, noise_sigma, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_frame(rows=1080, cols=1920, seed=0, n_rect=300, noise_sigma=3.0, shift=(0, 0), noise_seed=None):
    """Returns a (rows, cols) uint8 frame. `shift`=(dx,dy) translates the underlying scene (frame k+1 = frame k shifted),
    `noise_seed` draws fresh pixel noise on the same scene."""
    scene = synth_scene(rows, cols, seed, n_rect)
    return frame_from_scene(scene, rows, cols, shift, seed * 7919 + 17 if noise_seed is None else noise_seed, noise_sigma)


def synth_video(rows, cols, n_frames, seed=0, frames_per_scene=8, step=(3, 2)):
    """n_frames frames; every `frames_per_scene` consecutive frames pan over one scene by `step` px per frame with fresh
    pixel noise (BASELINE config 2: frame k+1 = frame k shifted by (3, 2) px + fresh noise)."""
    out = np.empty((n_frames, rows, cols), np.uint8)
    scene = None
    for i in range(n_frames):
        k = i % frames_per_scene
        if k == 0:
            scene = synth_scene(rows, cols, seed * 1000 + i // frames_per_scene)
        out[i] = frame_from_scene(scene, rows, cols, (step[0] * k, step[1] * k), noise_seed=seed * 100003 + i)
    return out


def synth_local_ba(n_pose=50, n_pt=20000, obs_per_pose=2000, seed=0, pose_noise=0.0, point_noise=0.0, n_fixed=2):
    """BASELINE config 5: n_pose keyframes on a 10 m circle looking inward, n_pt landmarks uniform in a 4 m
    cube, each keyframe observes its obs_per_pose nearest landmarks; perspective fx=fy=700, cx=960, cy=540; observation =
    exact projection + N(0,1) px; octave U{0..7} (information = 1/1.2^(2*octave)); the first n_fixed poses are fixed.
    pose_noise / point_noise perturb the INITIAL estimate (for Gauss-Newton tests). Returns a dict of numpy arrays."""
    from .ba import EDGE_DTYPE, rot_to_quat
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = (700.0, 700.0, 960.0, 540.0)
    pts = rng.uniform(-2.0, 2.0, size=(n_pt, 3))
    poses = np.zeros((n_pose, 7))
    Rs, ts = [], []
    for k in range(n_pose):
        a = 2 * np.pi * k / n_pose
        Cw = np.array([10 * np.cos(a), 0.0, 10 * np.sin(a)])
        z = -Cw / np.linalg.norm(Cw)
        up = np.array([0.0, -1.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        t = -R @ Cw
        Rs.append(R); ts.append(t)
        poses[k, :3] = t
        poses[k, 3:] = rot_to_quat(R)
    edges = np.zeros(n_pose * obs_per_pose, EDGE_DTYPE)
    sig = np.float32(1.0)
    inv_sig = []
    for _ in range(8):
        inv_sig.append(1.0 / float(np.float32(sig * sig)))
        sig = np.float32(1.2) * sig
    inv_sig = np.array(inv_sig)
    for k in range(n_pose):
        pc = pts @ Rs[k].T + ts[k]
        near = np.argsort(pc[:, 2], kind="stable")[:obs_per_pose]
        near.sort()
        u = cam[0] * pc[near, 0] / pc[near, 2] + cam[2] + rng.normal(0, 1, len(near))
        v = cam[1] * pc[near, 1] / pc[near, 2] + cam[3] + rng.normal(0, 1, len(near))
        e = edges[k * obs_per_pose:(k + 1) * obs_per_pose]
        e["pose_idx"] = k
        e["point_idx"] = near
        e["obs_x"] = u
        e["obs_y"] = v
        e["inv_sigma_sq"] = inv_sig[rng.integers(0, 8, len(near))]
    fixed = np.zeros(n_pose, np.uint8)
    fixed[:n_fixed] = 1
    poses0, pts0 = poses.copy(), pts.copy()
    if pose_noise:
        poses0[n_fixed:, :3] += rng.normal(0, pose_noise, size=(n_pose - n_fixed, 3))
    if point_noise:
        pts0 += rng.normal(0, point_noise, size=pts.shape)
    return dict(cam=cam, poses=poses0, points=pts0, poses_true=poses, points_true=pts, edges=edges, pose_fixed=fixed,
                huber_delta=float(np.sqrt(5.991)))


def flip_bits(rng, desc, max_flip=40):
    """A copy of a 32-byte descriptor with U{0..max_flip} random bit flips."""
    bits = np.unpackbits(desc)
    k = int(rng.integers(0, max_flip + 1))
    bits[rng.permutation(256)[:k]] ^= 1
    return np.packbits(bits)


def synth_keypoints(n, rows, cols, seed=0, num_levels=8, scale=1.2):
    """n cv::KeyPoint records + random descriptors as an extractor would emit them (level-major, level-0 coordinates), for
    matcher tests that do not need images."""
    from .match import KP_DTYPE
    rng = np.random.Generator(np.random.PCG64(seed))
    k = np.zeros(n, KP_DTYPE)
    octs = np.sort(rng.integers(0, num_levels, n))
    sf = np.float32(scale) ** octs.astype(np.float32)
    k["octave"] = octs
    k["x"] = (np.floor(rng.uniform(22, cols / sf - 22)) * sf).astype(np.float32)
    k["y"] = (np.floor(rng.uniform(22, rows / sf - 22)) * sf).astype(np.float32)
    k["size"] = 31 * sf
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["response"] = rng.integers(7, 120, n)
    k["class_id"] = -1
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    return k, d


def synth_landmarks(kps, desc, n_lm, rows, cols, seed=0, n_from_frame=None, jitter=2.0, with_stereo=False):
    """BASELINE config 4's landmark set: n_from_frame landmarks are frame keypoints (wrapping around, so some
    keypoints are wanted by several landmarks) with U{0..40} flipped descriptor bits, reprojected within `jitter` px of their
    keypoint and predicted at the keypoint's level or one above; the rest are distractors: uniform-random descriptors,
    reprojections uniform over the image, levels U{0..7}. Returns dict(xy, level, desc, valid[, x_right])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(kps)
    if n_from_frame is None:
        n_from_frame = min(n_lm, n)
    xy = np.zeros((n_lm, 2), np.float32)
    level = np.zeros(n_lm, np.int32)
    d = rng.integers(0, 256, size=(n_lm, 32), dtype=np.uint8)
    src = np.full(n_lm, -1, np.int64)
    order = rng.permutation(n_lm)
    for j, l in enumerate(order[:n_from_frame]):
        i = int(rng.integers(0, n)) if j >= n else j
        src[l] = i
        xy[l] = (kps["x"][i] + rng.normal(0, jitter), kps["y"][i] + rng.normal(0, jitter))
        level[l] = min(7, int(kps["octave"][i]) + int(rng.integers(0, 2)))
        d[l] = flip_bits(rng, desc[i])
    for l in order[n_from_frame:]:
        xy[l] = (rng.uniform(0, cols), rng.uniform(0, rows))
        level[l] = int(rng.integers(0, 8))
    out = dict(xy=xy, level=level, desc=d, valid=(rng.random(n_lm) < 0.95).astype(np.uint8), src=src)
    if with_stereo:
        out["x_right"] = (xy[:, 0] - rng.uniform(2, 60, n_lm)).astype(np.float32)
    return out


def synth_bow(desc, seed=0, n_nodes=200):
    """A stand-in for DBoW2's feature vector (node id -> keypoint indices): keypoints are bucketed by a locality-sensitive
    hash of their descriptor (majority bit of 8 fixed byte groups), so near-duplicate descriptors tend to share a node as
    they do in a vocabulary tree. Only the bucket structure matters to the matcher."""
    rng = np.random.Generator(np.random.PCG64(seed))
    groups = rng.permutation(256)[:8 * 16].reshape(8, 16)
    bits = np.unpackbits(np.ascontiguousarray(desc, np.uint8).reshape(-1, 32), axis=1)
    code = np.zeros(len(bits), np.int64)
    for b in range(8):
        code |= (bits[:, groups[b]].sum(1) >= 8).astype(np.int64) << b
    node = (code * 7919) % n_nodes
    fv = {}
    for i, nd in enumerate(node):
        fv.setdefault(int(nd), []).append(i)
    return fv


def synth_stereo_pair(rows=376, cols=1241, seed=0, d_min=4.0, d_max=60.0, noise_sigma=2.0):
    """BASELINE config 3 input: a rectified pair. The right image sees, at column x, the scene point the left
    image shows at x + d(y): the disparity varies smoothly with the row inside [d_min, d_max] px (fractional, so the sub-pixel
    refinement has something to find); each image gets its own pixel noise. Returns (left, right, disparity_per_row)."""
    pad = int(np.ceil(d_max)) + 4
    scene = synth_scene(rows, cols + pad, seed)
    rng = np.random.Generator(np.random.PCG64(seed * 7919 + 13))
    y = np.arange(rows)
    d = d_min + (d_max - d_min) * (0.5 + 0.5 * np.sin(2 * np.pi * 1.5 * y / rows))
    x = np.arange(cols)[None, :] + d[:, None]
    x0 = np.floor(x).astype(np.int64)
    f = (x - x0).astype(np.float32)
    rowsel = y[:, None]
    right = scene[rowsel, x0] * (1 - f) + scene[rowsel, x0 + 1] * f
    left = scene[:rows, :cols]
    left = np.clip(np.rint(left + rng.normal(0, noise_sigma, left.shape)), 0, 255).astype(np.uint8)
    right = np.clip(np.rint(right + rng.normal(0, noise_sigma, right.shape)), 0, 255).astype(np.uint8)
    return left, right, d.astype(np.float32)


def _rot_axis(axis, deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    x, y, z = axis
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


def synth_pose_frame(dtype, n, seed, stereo_frac=0.4, outlier_frac=0.1, pose_err=1.0):
    """A frame for optimize::pose_optimizer: n landmarks in front of a camera at a known pose, observations with level-dependent
    pixel noise, a fraction of stereo keypoints and of gross outliers, and a perturbed initial pose. dtype = the POSE_OBS record
    dtype. Returns (T0 3x4, obs, cam, focal_x_baseline, (R_true, t_true, outlier_mask))."""
    rng = np.random.default_rng(seed)
    Rt = _rot_axis((0, 1, 0), 5) @ _rot_axis((1, 0, 0), -3)
    tt = np.array([0.3, -0.1, 0.2])
    X = np.stack([rng.uniform(-5, 5, n), rng.uniform(-3, 3, n), rng.uniform(4, 20, n)], 1)
    cam, bf = (700.0, 700.0, 960.0, 540.0), 70.0
    pc = X @ Rt.T + tt
    u = cam[0] * pc[:, 0] / pc[:, 2] + cam[2]
    v = cam[1] * pc[:, 1] / pc[:, 2] + cam[3]
    obs = np.zeros(n, dtype)
    obs["pos_w"] = X
    sig = 1.2 ** rng.integers(0, 8, n)
    obs["obs_x"] = u + rng.normal(0, 1, n) * sig
    obs["obs_y"] = v + rng.normal(0, 1, n) * sig
    obs["inv_sigma_sq"] = 1 / sig ** 2
    st = rng.random(n) < stereo_frac
    obs["is_stereo"] = st
    obs["obs_x_right"] = np.where(st, u - bf / pc[:, 2] + rng.normal(0, 1, n) * sig, 0)
    bad = rng.random(n) < outlier_frac
    obs["obs_x"][bad] += rng.uniform(20, 100, int(bad.sum()))
    T0 = np.concatenate([_rot_axis((0, 1, 0), 5 + 0.8 * pose_err) @ _rot_axis((1, 0, 0), -3 + 0.4 * pose_err),
                         (tt + pose_err * np.array([0.05, 0.03, -0.04]))[:, None]], 1)
    return T0, obs, cam, bf, (Rt, tt, bad)


def equirect_project(pc, cols, rows):
    """camera::equirectangular::reproject_to_image on camera-frame points (numpy; for generating test inputs)."""
    theta = np.arctan2(pc[:, 0], pc[:, 2])
    phi = -np.arcsin(pc[:, 1] / np.linalg.norm(pc, axis=1))
    return cols * (0.5 + theta / (2 * np.pi)), rows * (0.5 - phi / np.pi)


def synth_pose_frame_equirect(dtype, n, seed, cols=3840, rows=1920, outlier_frac=0.1, pose_err=1.0, seam_frac=0.0, pole_frac=0.0):
    """An equirectangular frame for optimize::pose_optimizer (BASELINE configs[3] geometry): n landmarks ALL AROUND the camera (bearings over
    the whole sphere), observations with level-dependent pixel noise, gross outliers, a perturbed initial pose. seam_frac / pole_frac of the
    landmarks are placed within a few pixels of the +-180 degree seam (behind the camera) / of the poles, where the projection is most
    sensitive. Returns (T0 3x4, obs, cols, rows, (R_true, t_true, outlier_mask))."""
    rng = np.random.default_rng(seed)
    Rt = _rot_axis((0, 1, 0), 25) @ _rot_axis((1, 0, 0), -7)
    tt = np.array([0.3, -0.1, 0.2])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    n_seam, n_pole = int(n * seam_frac), int(n * pole_frac)
    if n_seam:   # camera-frame bearings just either side of theta = +-pi
        th = np.pi - rng.uniform(0.0005, 0.01, n_seam) * rng.choice([-1, 1], n_seam)
        ph = rng.uniform(-1.0, 1.0, n_seam)
        d[:n_seam] = np.stack([np.sin(th) * np.cos(ph), -np.sin(ph), np.cos(th) * np.cos(ph)], 1)
    if n_pole:
        ph = (np.pi / 2 - rng.uniform(0.002, 0.03, n_pole)) * rng.choice([-1, 1], n_pole)
        th = rng.uniform(-np.pi, np.pi, n_pole)
        d[n_seam:n_seam + n_pole] = np.stack([np.sin(th) * np.cos(ph), -np.sin(ph), np.cos(th) * np.cos(ph)], 1)
    pc = d * rng.uniform(3, 25, n)[:, None]
    X = (pc - tt) @ Rt                     # world points whose camera-frame positions are pc
    u, v = equirect_project(pc, cols, rows)
    obs = np.zeros(n, dtype)
    obs["pos_w"] = X
    sig = 1.2 ** rng.integers(0, 8, n)
    obs["obs_x"] = u + rng.normal(0, 1, n) * sig
    obs["obs_y"] = v + rng.normal(0, 1, n) * sig
    obs["inv_sigma_sq"] = 1 / sig ** 2
    bad = rng.random(n) < outlier_frac
    obs["obs_x"][bad] += rng.uniform(20, 100, int(bad.sum()))
    T0 = np.concatenate([_rot_axis((0, 1, 0), 25 + 0.8 * pose_err) @ _rot_axis((1, 0, 0), -7 + 0.4 * pose_err),
                         (tt + pose_err * np.array([0.05, 0.03, -0.04]))[:, None]], 1)
    return T0, obs, cols, rows, (Rt, tt, bad)


def synth_vocabulary(k=10, depth=4, seed=0, flip=28):
    """A stand-in ORB vocabulary tree (the real orb_vocab file is not in the container): node 0 is the root, every inner node has k
    children (the last inner level keeps between k/2 and k so ragged nodes are exercised), a child's descriptor is its parent's with
    `flip` random bits flipped (k-majority clusters look like that), leaves are the words with TF-IDF-like weights; some leaves carry
    weight 0 (DBoW2 drops those features). Node ids are in creation order, children of a node are consecutive, as DBoW2 creates them.
    Returns dict(child_start, children, desc, weight, word_id, depth)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    desc = [rng.integers(0, 256, 32, dtype=np.uint8)]
    kids = [[]]
    level_nodes = [0]
    for lvl in range(depth):
        nxt = []
        for n in level_nodes:
            nk = k if lvl < depth - 1 else int(rng.integers(max(2, k // 2), k + 1))
            for _ in range(nk):
                d = desc[n].copy()
                pos = rng.choice(256, size=flip, replace=False)
                bits = np.unpackbits(d)
                bits[pos] ^= 1
                desc.append(np.packbits(bits))
                kids.append([])
                kids[n].append(len(desc) - 1)
                nxt.append(len(desc) - 1)
        level_nodes = nxt
    n_nodes = len(desc)
    child_start = np.zeros(n_nodes + 1, np.int32)
    children = []
    for n in range(n_nodes):
        children.extend(kids[n])
        child_start[n + 1] = len(children)
    word_id = -np.ones(n_nodes, np.int32)
    leaves = [n for n in range(n_nodes) if not kids[n]]
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    weight = np.zeros(n_nodes)
    weight[leaves] = np.where(rng.random(len(leaves)) < 0.03, 0.0, rng.uniform(0.5, 9.0, len(leaves)))
    return dict(child_start=child_start, children=np.asarray(children, np.int32), desc=np.stack(desc), weight=weight, word_id=word_id,
                depth=depth)


def synth_map(n_pose=8, n_pt=600, obs_per_pose=250, seed=0, stereo_frac=0.0, pose_noise=0.02, point_noise=0.02):
    """A map database (openvslam_amd.io.map_database) holding the synth_local_ba scene: one keyframe per pose whose keypoints are its
    observations (undistorted = observed position, octave from the information, random descriptors), one landmark per observed point.
    Returns (db, scene) where scene is the synth_local_ba dict the map was built from."""
    from . import io
    from .match import KP_DTYPE
    d = synth_local_ba(n_pose=n_pose, n_pt=n_pt, obs_per_pose=obs_per_pose, seed=seed, pose_noise=pose_noise, point_noise=point_noise, n_fixed=1)
    rng = np.random.default_rng(seed + 7)
    inv = io.inv_level_sigma_sq(1.2, 8)
    db = io.map_database()
    db.cameras = {"cam": {"model_type": "Perspective", "setup_type": "Stereo" if stereo_frac > 0 else "Monocular", "color_order": "Gray",
                          "cols": 1920, "rows": 1080, "fps": 30.0, "fx": d["cam"][0], "fy": d["cam"][1], "cx": d["cam"][2], "cy": d["cam"][3],
                          "k1": 0.0, "k2": 0.0, "p1": 0.0, "p2": 0.0, "k3": 0.0, "focal_x_baseline": 0.12 * d["cam"][0]}}
    e = d["edges"]
    bf = 0.12 * d["cam"][0]
    from .ba import quat_to_rot
    for k in range(n_pose):
        ek = e[e["pose_idx"] == k]
        n = len(ek)
        kp = np.zeros(n, KP_DTYPE)
        kp["x"], kp["y"] = ek["obs_x"], ek["obs_y"]
        kp["octave"] = [int(np.argmin(np.abs(inv - w))) for w in ek["inv_sigma_sq"]]
        kp["angle"] = rng.uniform(0, 360, n)
        kp["class_id"] = -1
        xr = np.full(n, -1.0, np.float32)
        dep = np.full(n, -1.0, np.float32)
        if stereo_frac > 0:
            z = (d["points_true"][ek["point_idx"]] @ quat_to_rot(d["poses_true"][k, 3:]).T + d["poses_true"][k, :3])[:, 2]
            st = rng.random(n) < stereo_frac
            xr[st] = (ek["obs_x"][st] - bf / z[st] + rng.normal(0, 1, int(st.sum()))).astype(np.float32)
            dep[st] = z[st]
        db.keyframes[k] = io.keyframe(id=k, src_frm_id=10 * k, ts=0.1 * k, cam="cam", depth_thr=40.0, rot_cw=d["poses"][k, 3:].copy(),
                                      trans_cw=d["poses"][k, :3].copy(), keypts=kp,
                                      undists=np.stack([ek["obs_x"], ek["obs_y"]], 1).astype(np.float32), x_rights=xr, depths=dep,
                                      descs=rng.integers(0, 256, (n, 32), dtype=np.uint8), lm_ids=ek["point_idx"].astype(np.int64),
                                      span_parent=k - 1, span_children=[k + 1] if k + 1 < n_pose else [], loop_edges=[])
    seen = np.unique(e["point_idx"])
    for j in seen:
        first = int(e["pose_idx"][e["point_idx"] == j].min())
        db.landmarks[int(j)] = io.landmark(id=int(j), first_keyfrm=first, pos_w=d["points"][j].copy(), ref_keyfrm=first,
                                           n_vis=int((e["point_idx"] == j).sum()), n_fnd=int((e["point_idx"] == j).sum()))
    db.frame_next_id, db.keyframe_next_id, db.landmark_next_id = 10 * n_pose, n_pose, int(seen.max()) + 1
    return db, d
