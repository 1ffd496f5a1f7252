"""A SECOND, independent restatement of the arithmetic stages of the extraction -- cv::resize(INTER_LINEAR, 8UC1), cv::FAST(TYPE_9_16) with its
corner score and non-maximum suppression, the 7x7 sigma-2 fixed-point GaussianBlur, ic_angle + cv::fastAtan2, and the steered rBRIEF-256 --
written in vectorised numpy from the algorithm text of oracle/ORACLE_SPEC.md (rules 3, 5, 9, 10, 11) and the public description of those
OpenCV / OpenVSLAM functions, NOT from oracle/ovo_orb.cc (whole-array formulation, no per-pixel loops, different decomposition).
tests/test_nversion.py compares it with the C oracle bit for bit, and reproduces the committed golden angles and descriptors of all 1008
keypoints with the oracle's C code out of the loop (only the keypoint positions -- cell loop + quad-tree, which have their own executable
model in tools/tree_model.py -- are taken as given).

Why: the reference tree is absent, so the oracle cannot be pinned to it ("What's missing" #1). Two implementations
written separately from the same specification that agree on every pixel / bit of the golden inputs is the strongest evidence available here
that the oracle implements its specification -- it says nothing about whether the SPECIFICATION is upstream's (that is what the run-time
variants of ORACLE_SPEC.md hedge). Test infrastructure only: nothing under openvslam_amd/ imports this."""
import numpy as np


# ---- rule 3: cv::resize, INTER_LINEAR, CV_8UC1, OpenCV's 11-bit fixed-point path ---------------------------------------------------------
def _linear_taps(src, dst):
    """Per destination index: (left source index, coefficient pair as int16 Q11), with OpenCV's border handling on the horizontal axis:
    a tap left of the image or at/after the last column collapses onto one pixel with weight 1."""
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)          # evaluated in double, stored as float
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _q11(c):
    """saturate_cast<short>(c * 2048): round half to even (cvRound)."""
    return np.rint(c.astype(np.float32) * np.float32(2048.0)).astype(np.int64)


def resize_linear_u8(img, drows, dcols):
    img = np.asarray(img, np.uint8)
    srows, scols = img.shape
    # horizontal: indices outside collapse (fx = 0)
    sx, fx = _linear_taps(scols, dcols)
    lo = sx < 0
    hi = sx >= scols - 1
    sx = np.where(lo, 0, np.where(hi, scols - 1, sx))
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    a0, a1 = _q11(np.float32(1.0) - fx), _q11(fx)
    sx1 = np.minimum(sx + 1, scols - 1)
    S = img.astype(np.int64)
    h = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]      # one value per (source row, destination column), Q11
    # vertical: the two source ROWS are clipped, the coefficients are not touched
    sy, fy = _linear_taps(srows, drows)
    b0, b1 = _q11(np.float32(1.0) - fy), _q11(fy)
    r0 = np.clip(sy, 0, srows - 1)
    r1 = np.clip(sy + 1, 0, srows - 1)
    t0 = (b0[:, None] * (h[r0, :] >> 4)) >> 16
    t1 = (b1[:, None] * (h[r1, :] >> 4)) >> 16
    return ((t0 + t1 + 2) >> 2).astype(np.uint8)


# ---- rule 5: FAST-9/16, corner score, strict 3x3 non-maximum suppression ----------------------------------------------------------------------
_RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_strength(img):
    """S(p) for every pixel at least 3 px from the border (0 elsewhere): the largest t' such that 9 contiguous ring pixels are all
    >= p + t' or all <= p - t', i.e. max over the 16 arcs of the arc's minimum signed difference, both polarities. A pixel is a
    FAST-9 corner at threshold t iff S > t, and OpenCV's cornerScore (the largest threshold at which it still is one) is S - 1."""
    I = np.asarray(img, np.uint8).astype(np.int32)
    H, W = I.shape
    out = np.zeros((H, W), np.int32)
    if H < 7 or W < 7:
        return out
    c = I[3:H - 3, 3:W - 3]
    d = np.stack([I[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] - c for dx, dy in _RING])          # (16, h, w): ring - centre
    ext = np.concatenate([d, d[:8]])                                                          # circular: 24 entries
    # sliding minimum / maximum of width 9 over the ring by doubling: windows of 2, 4, 8, then 8 + 1
    mn2, mx2 = np.minimum(ext[:-1], ext[1:]), np.maximum(ext[:-1], ext[1:])
    mn4, mx4 = np.minimum(mn2[:-2], mn2[2:]), np.maximum(mx2[:-2], mx2[2:])
    mn8, mx8 = np.minimum(mn4[:-4], mn4[4:]), np.maximum(mx4[:-4], mx4[4:])
    mn9 = np.minimum(mn8[:16], ext[8:24])
    mx9 = np.maximum(mx8[:16], ext[8:24])
    bright = mn9.max(0)          # ring brighter than the centre by at least this much along the best arc
    dark = (-mx9).max(0)         # ... darker
    out[3:H - 3, 3:W - 3] = np.maximum(np.maximum(bright, dark), 0)
    return out


def fast9_16(img, threshold, nonmax=True):
    """cv::FAST(img, kps, threshold, nonmax, TYPE_9_16): (x, y, response) in row-major order."""
    S = fast_strength(img)
    score = np.where(S > threshold, S - 1, 0)
    keep = S > threshold
    if nonmax:
        p = np.pad(score, 1)
        nb = np.stack([p[1 + dy:p.shape[0] - 1 + dy, 1 + dx:p.shape[1] - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
        keep &= score > nb.max(0)
    ys, xs = np.nonzero(keep)
    return xs.astype(np.int32), ys.astype(np.int32), score[ys, xs].astype(np.int32)


# ---- rule 10: GaussianBlur(7x7, sigma 2), 8-bit fixed point, BORDER_REFLECT_101 ---------------------------------------------------------------
_TAPS = {0: np.array([18, 34, 48, 56, 48, 34, 18], np.int64), 1: np.array([18, 34, 49, 55, 49, 34, 18], np.int64)}


def gaussian_blur_7x7(img, taps_variant=0):
    k = _TAPS[taps_variant]
    P = np.pad(np.asarray(img, np.uint8).astype(np.int64), 3, mode="reflect")     # numpy 'reflect' = REFLECT_101 (edge pixel not repeated)
    H, W = np.asarray(img).shape
    row = sum(k[i] * P[:, i:i + W] for i in range(7))                              # 8.8 fixed point (unsigned 16-bit, saturating)
    row = np.minimum(row, 0xFFFF)
    col = sum(k[i] * row[i:i + H, :] for i in range(7))                            # 16.16
    col = np.minimum(col, 0xFFFFFFFF)
    return np.minimum((col + 32768) >> 16, 255).astype(np.uint8)


# ---- rule 9: intensity-centroid orientation, cv::fastAtan2 (scalar form), every operation rounded to float32 ----------------------------
_UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
_F = np.float32


def fast_atan2_deg(y, x):
    """cv::fastAtan2(y, x) for float32 arrays, degrees in [0, 360]."""
    y, x = np.asarray(y, _F), np.asarray(x, _F)
    s = _F(180.0 / np.pi)
    p1, p3, p5, p7 = _F(0.9997878412794807) * s, _F(-0.3258083974640975) * s, _F(0.1555786518463281) * s, _F(-0.04432655554792128) * s
    eps = _F(2.2204460492503131e-16)
    ax, ay = np.abs(x), np.abs(y)
    swap = ax < ay
    num, den = np.where(swap, ax, ay), np.where(swap, ay, ax) + eps
    with np.errstate(invalid="ignore", divide="ignore"):
        c = (num / den).astype(_F)
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(swap, _F(90.0) - a, a).astype(_F)
    a = np.where(x < 0, _F(180.0) - a, a).astype(_F)
    a = np.where(y < 0, _F(360.0) - a, a).astype(_F)
    return a


def ic_angle(img, xs, ys):
    """orb_extractor::ic_angle at integer keypoint positions of one (unblurred) level: integer moments over the radius-15 disc."""
    I = np.asarray(img, np.uint8).astype(np.int64)
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    m10 = np.zeros(len(xs), np.int64)
    m01 = np.zeros(len(xs), np.int64)
    for v in range(-15, 16):
        um = _UMAX[abs(v)]
        u = np.arange(-um, um + 1)
        row = I[(ys + v)[:, None], xs[:, None] + u[None, :]]
        m10 += (row * u[None, :]).sum(1)
        m01 += v * row.sum(1)
    return fast_atan2_deg(m01.astype(_F), m10.astype(_F))


# ---- rule 11: steered rBRIEF-256 on the blurred level ---------------------------------------------------------------------------------------
def _util_cos(v):
    """openvslam::util::cos: reduction by floor(v / 2 pi), even polynomial on [0, pi / 2], float32 throughout."""
    v = np.asarray(v, _F)
    two_pi, half_pi, pi, three_half_pi = _F(6.28318530717958647692), _F(1.57079632679489661923), _F(3.14159265358979323846), _F(4.71238898038468985769)
    v = v - np.floor(v * _F(0.15915494309189533577)).astype(_F) * two_pi
    v = np.abs(v).astype(_F)

    def poly(t):
        t2 = t * t
        return _F(0.99940307) + t2 * (_F(-0.49558072) + _F(0.03679168) * t2)

    return np.where(v < half_pi, poly(v), np.where(v < pi, -poly(pi - v), np.where(v < three_half_pi, -poly(v - pi), poly(two_pi - v)))).astype(_F)


def orb_descriptors(blurred, xs, ys, angles_deg, pattern):
    """compute_orb_descriptor for keypoints (xs, ys) of one blurred level; pattern = the 256 x 4 int8 rBRIEF table."""
    B = np.asarray(blurred, np.uint8)
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    rad = (np.asarray(angles_deg, _F).astype(np.float64) * np.pi / 180.0).astype(_F)   # double product and quotient, ONE rounding to float
    c, s = _util_cos(rad), _util_cos(_F(1.57079632679489661923) - rad)
    pat = np.asarray(pattern, np.int64).reshape(256, 4).astype(_F)

    def sample(px, py):   # (256,) pattern coordinates against (n,) keypoints -> (n, 256) intensities
        fx, fy = px[None, :], py[None, :]
        dy = np.rint(fx * s[:, None] + fy * c[:, None]).astype(np.int64)
        dx = np.rint(fx * c[:, None] - fy * s[:, None]).astype(np.int64)
        return B[ys[:, None] + dy, xs[:, None] + dx]

    bits = sample(pat[:, 0], pat[:, 1]) < sample(pat[:, 2], pat[:, 3])
    return np.packbits(bits, axis=1, bitorder="little")


# ---- rule 14: the brute-force Hamming matcher (match::robust) -----------------------------------------------------------------------------
def hamming_matrix(a, b):
    """All-pairs Hamming distances of two sets of 256-bit descriptors, (len(a), len(b)): |x ^ y| = |x| + |y| - 2 x.y over the bit vectors
    (an integer matrix product -- no popcount, no xor)."""
    A = np.unpackbits(np.asarray(a, np.uint8).reshape(-1, 32), axis=1).astype(np.int32)
    B = np.unpackbits(np.asarray(b, np.uint8).reshape(-1, 32), axis=1).astype(np.int32)
    return A.sum(1)[:, None] + B.sum(1)[None, :] - 2 * (A @ B.T)


def robust_brute_force_match(desc_frm, desc_kf, kf_valid=None, lowe_ratio=0.8, frm_valid=None):
    """Rule 14 from its text: keyframe keypoints in index order (those with a live landmark), each against all frame keypoints nobody has
    claimed yet; the nearest (lowest index on ties) is accepted iff its distance is at most 50 and not `ratio * second < best` in float;
    an accepted frame keypoint is claimed. Distances of claimed / masked frame keypoints count as the 256 both minima start from.
    Returns (frame index, keyframe index) pairs in keyframe order."""
    D = hamming_matrix(desc_kf, desc_frm)
    free = np.ones(len(desc_frm), bool) if frm_valid is None else np.asarray(frm_valid).astype(bool).copy()
    pairs = []
    for j in range(len(desc_kf)):
        if kf_valid is not None and not kf_valid[j]:
            continue
        d = np.where(free, D[j], 256)
        if len(d) == 0:
            continue
        i = int(np.argmin(d))
        best = int(d[i])
        second = int(np.partition(d, 1)[1]) if len(d) > 1 else 256
        if best > 50 or np.float32(lowe_ratio) * np.float32(second) < np.float32(best):
            continue
        pairs.append((i, j))
        free[i] = False
    return np.asarray(pairs, np.int32).reshape(-1, 2)


def hamming_best2(q, t, t_valid=None):
    """Nearest and second-nearest target of every query (lowest index on ties; masked targets do not take part; 256 where there is none)."""
    D = hamming_matrix(q, t)
    if t_valid is not None:
        D = np.where(np.asarray(t_valid).astype(bool)[None, :], D, 256)
    if D.shape[1] == 0:
        return np.full(len(D), -1, np.int32), np.full(len(D), 256, np.uint16), np.full(len(D), 256, np.uint16)
    bi = np.argmin(D, 1).astype(np.int32)
    b = D[np.arange(len(D)), bi]
    s = np.partition(D, 1, axis=1)[:, 1] if D.shape[1] > 1 else np.full(len(D), 256)
    bi = np.where(b < 256, bi, -1).astype(np.int32)
    return bi, b.astype(np.uint16), np.asarray(s).astype(np.uint16)


# ---- rule 16: the keypoint grid (data::assign_keypoints_to_grid / get_keypoints_in_cell) -----------------------------------------------------
def _grid_inv(min_v, max_v, n_cells):
    return np.float32(np.float64(n_cells) / np.float64(np.float32(max_v) - np.float32(min_v)))


def grid_cells(xs, ys, min_x, min_y, max_x, max_y, n_cols, n_rows):
    """Cell of every keypoint, (cx, cy, inside): cvRound((pt - min) * inv) in float, half to even; outside the grid = in no cell."""
    F = np.float32
    cx = np.rint((np.asarray(xs, F) - F(min_x)) * _grid_inv(min_x, max_x, n_cols)).astype(np.int64)
    cy = np.rint((np.asarray(ys, F) - F(min_y)) * _grid_inv(min_y, max_y, n_rows)).astype(np.int64)
    return cx, cy, (cx >= 0) & (cx < n_cols) & (cy >= 0) & (cy < n_rows)


def keypoints_in_cell(xs, ys, octaves, ref_x, ref_y, margin, min_x, min_y, max_x, max_y, n_cols, n_rows, min_level=-1, max_level=-1):
    """Indices of the keypoints in the cells the square (ref +- margin) touches, cell columns first, then cell rows, ascending keypoint index
    inside a cell; level filter only when (0 < min_level) or (0 <= max_level); strictly inside the square."""
    F = np.float32
    xs, ys = np.asarray(xs, F), np.asarray(ys, F)
    cx, cy, inside = grid_cells(xs, ys, min_x, min_y, max_x, max_y, n_cols, n_rows)
    iw, ih = _grid_inv(min_x, max_x, n_cols), _grid_inv(min_y, max_y, n_rows)
    rx, ry, m = F(ref_x), F(ref_y), F(margin)
    x_lo = max(0, int(np.floor((rx - F(min_x) - m) * iw)))
    x_hi = min(n_cols - 1, int(np.ceil((rx - F(min_x) + m) * iw)))
    y_lo = max(0, int(np.floor((ry - F(min_y) - m) * ih)))
    y_hi = min(n_rows - 1, int(np.ceil((ry - F(min_y) + m) * ih)))
    if x_lo >= n_cols or x_hi < 0 or y_lo >= n_rows or y_hi < 0:
        return np.zeros(0, np.int32)
    sel = inside & (cx >= x_lo) & (cx <= x_hi) & (cy >= y_lo) & (cy <= y_hi)
    if (0 < min_level) or (0 <= max_level):
        oc = np.asarray(octaves)
        sel &= ~((oc < min_level) | ((0 <= max_level) & (oc > max_level)))
    sel &= (np.abs(xs - rx) < m) & (np.abs(ys - ry) < m)
    idx = np.nonzero(sel)[0]
    order = np.lexsort((idx, cy[idx], cx[idx]))   # primary: cell column, then cell row, then keypoint index
    return idx[order].astype(np.int32)


# ---- rule 17: match::angle_checker ------------------------------------------------------------------------------------------------------------
def angle_checker_invalid(delta_angles, keep_rule=0):
    """True for the entries outside the three fullest 30-degree bins: delta wrapped once into [0, 360), bin = cvRound(delta * (1 / 30)) in
    float (half to even), equal counts -> the lower bin first. keep_rule = 1 (the variant of rule 17, ORB-SLAM2's ComputeThreeMaxima): a second
    bin holding less than 0.1 x the fullest is dropped together with the third, a third bin below that alone."""
    F = np.float32
    d = np.asarray(delta_angles, F).copy()
    d = np.where(d < 0, d + F(360.0), d).astype(F)
    d = np.where(d >= F(360.0), d - F(360.0), d).astype(F)
    b = np.rint(d * F(1.0 / 30.0)).astype(np.int64)
    counts = np.bincount(b, minlength=30)[:30]
    keep = np.argsort(-counts, kind="stable")[:3]
    if keep_rule == 1:
        floor = F(0.1) * F(counts[keep[0]])
        if F(counts[keep[1]]) < floor:
            keep = keep[:1]
        elif F(counts[keep[2]]) < floor:
            keep = keep[:2]
    return ~np.isin(b, keep)


# ---- rule 18: projection::match_frame_and_landmarks -------------------------------------------------------------------------------------------
def projection_match_frame_and_landmarks(xs, ys, octaves, desc, scale_factors, lm_xy, lm_level, lm_desc, cols, rows, margin=5.0, lowe_ratio=0.6,
                                         x_right=None, occupied=None, lm_x_right=None, lm_valid=None):
    """assigned[l] = the frame keypoint given to local landmark l (or -1): landmarks in order; candidates = the grid's answer around the
    reprojection with radius margin * scale_factors[pred] on levels [pred - 1, pred], in the grid's order; keypoints that hold a landmark (also
    one given earlier in this call) are skipped; a stereo keypoint must agree on x_right within the radius; best / second with strict < and
    their levels; accepted iff best <= 100 and not (both on one level and best > ratio * second)."""
    F = np.float32
    occ = np.zeros(len(xs), bool) if occupied is None else np.asarray(occupied).astype(bool).copy()
    D = hamming_matrix(lm_desc, desc)
    assigned = np.full(len(lm_xy), -1, np.int32)
    for l in range(len(lm_xy)):
        if lm_valid is not None and not lm_valid[l]:
            continue
        pred = int(lm_level[l])
        r = F(margin) * F(scale_factors[pred])
        cand = keypoints_in_cell(xs, ys, octaves, lm_xy[l][0], lm_xy[l][1], r, 0.0, 0.0, cols, rows, 64, 48, pred - 1, pred)
        best = second = 256
        best_level = second_level = best_idx = -1
        for i in cand:
            if occ[i]:
                continue
            if x_right is not None and 0 < x_right[i] and r < abs(F(lm_x_right[l]) - F(x_right[i])):
                continue
            d = int(D[l, i])
            if d < best:
                second, second_level = best, best_level
                best, best_level, best_idx = d, int(octaves[i]), int(i)
            elif d < second:
                second, second_level = d, int(octaves[i])
        if best > 100 or (best_level == second_level and F(best) > F(lowe_ratio) * F(second)):
            continue
        assigned[l] = best_idx
        occ[best_idx] = True
    return assigned


# ---- rule 21: camera::reproject_to_image and projection::match_current_and_last_frames ---------------------------------------------------------
def reproject_to_image(model, cam, pose_cw, pos_w, min_x, min_y, max_x, max_y):
    """(in image, u, v, x_right) of world points (n, 3) under pose_cw (3 x 4): perspective (cam = fx, fy, cx, cy, focal_x_baseline): z <= 0
    is rejected, x_right = u - fx_b / z as float; equirectangular (cam = cols, rows): bearing -> (longitude, latitude); bounds inclusive."""
    T = np.asarray(pose_cw, float)
    p = np.asarray(pos_w, float) @ T[:, :3].T + T[:, 3]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    if model == 0:
        fx, fy, cx, cy, fxb = cam
        with np.errstate(divide="ignore", invalid="ignore"):
            iz = 1.0 / z
            u, v = fx * x * iz + cx, fy * y * iz + cy
            xr = (u - fxb * iz).astype(np.float32)
        ok = z > 0
    else:
        cols, rows = cam[0], cam[1]
        L = np.sqrt((p * p).sum(1))
        b = p / L[:, None]
        lat, lon = -np.arcsin(b[:, 1]), np.arctan2(b[:, 0], b[:, 2])
        u, v = cols * (0.5 + lon / (2 * np.pi)), rows * (0.5 - lat / np.pi)
        xr = np.full(len(p), -1.0, np.float32)
        ok = np.ones(len(p), bool)
    ok = ok & (u >= min_x) & (u <= max_x) & (v >= min_y) & (v <= max_y)
    return ok, u, v, xr


def projection_match_current_and_last_frames(model, setup, cam, true_baseline, cols, rows, xs, ys, octaves, angles, desc, pose_cw_curr,
                                             last_octaves, last_angles, last_pos_w, last_lm_desc, pose_cw_last, scale_factors, margin,
                                             check_orientation=True, x_right=None, occupied=None, last_valid=None):
    """assigned[i] = the current keypoint that takes over last frame's landmark i (or -1). The last frame's landmarks in order, reprojected with
    the current pose; radius margin * scale_factors[level in the last frame]; level window [l - 1, l + 1], or [l, top] / [0, l] when a
    non-monocular rig moved forward / backward by more than the baseline (z of the current camera centre in the last camera's frame); occupied
    keypoints (also those taken earlier in this call) skipped; stereo keypoints must agree on x_right within the radius; the nearest
    descriptor wins if its distance is <= 100; with the orientation check, the matches outside the three fullest bins of
    (last angle - current angle) are undone."""
    F = np.float32
    Tc, Tl = np.asarray(pose_cw_curr, float), np.asarray(pose_cw_last, float)
    centre_curr = -Tc[:, :3].T @ Tc[:, 3]
    z_lc = (Tl[:, :3] @ centre_curr + Tl[:, 3])[2]
    forward = setup != 0 and z_lc > true_baseline
    backward = setup != 0 and z_lc < -true_baseline
    ok, u, v, xr_lm = reproject_to_image(model, cam, Tc, last_pos_w, 0.0, 0.0, float(cols), float(rows))
    occ = np.zeros(len(xs), bool) if occupied is None else np.asarray(occupied).astype(bool).copy()
    D = hamming_matrix(last_lm_desc, desc)
    n_levels = len(scale_factors)
    assigned = np.full(len(last_pos_w), -1, np.int32)
    deltas, owners = [], []
    for i in range(len(last_pos_w)):
        if (last_valid is not None and not last_valid[i]) or not ok[i]:
            continue
        lvl = int(last_octaves[i])
        r = F(margin) * F(scale_factors[lvl])
        lo, hi = (lvl, n_levels - 1) if forward else ((0, lvl) if backward else (lvl - 1, lvl + 1))
        cand = keypoints_in_cell(xs, ys, octaves, F(u[i]), F(v[i]), r, 0.0, 0.0, cols, rows, 64, 48, lo, hi)
        best, best_idx = 256, -1
        for k in cand:
            if occ[k]:
                continue
            if x_right is not None and 0 < x_right[k] and r < abs(F(xr_lm[i]) - F(x_right[k])):
                continue
            d = int(D[i, k])
            if d < best:
                best, best_idx = d, int(k)
        if best > 100:
            continue
        assigned[i] = best_idx
        occ[best_idx] = True
        deltas.append(F(last_angles[i]) - F(angles[best_idx]))
        owners.append(i)
    if check_orientation and deltas:
        bad = angle_checker_invalid(np.asarray(deltas, F))
        assigned[np.asarray(owners)[bad]] = -1
    return assigned


# ---- rule 19: match::area::match_in_consistent_area (the initialiser's matcher) ----------------------------------------------------------------
def area_match_in_consistent_area(oct_1, ang_1, desc_1, xs_2, ys_2, oct_2, ang_2, desc_2, prev_matched_pts, cols, rows, margin=10, lowe_ratio=0.9,
                                  check_orientation=True):
    """(number of matches, matched[i1] = keypoint of frame 2 or -1, updated prev_matched_pts): level-0 keypoints of frame 1 in order, each
    against the level-0 keypoints of frame 2 inside the window around its previously matched point; a target currently matched at a
    distance <= d does not take part; accepted iff best <= 50 and not (ratio * second < best); an accepted target is taken from its earlier
    owner; every acceptance enters the orientation histogram (also those stolen later); matches outside the three fullest bins are undone;
    the surviving matches move prev_matched_pts to their target."""
    F = np.float32
    n1, n2 = len(oct_1), len(xs_2)
    D = hamming_matrix(desc_1, desc_2)
    m12 = np.full(n1, -1, np.int32)
    m21 = np.full(n2, -1, np.int32)
    mdist = np.full(n2, 256, np.int64)
    hist_delta, hist_owner = [], []
    for i1 in range(n1):
        if oct_1[i1] > 0:
            continue
        cand = keypoints_in_cell(xs_2, ys_2, oct_2, prev_matched_pts[i1][0], prev_matched_pts[i1][1], F(margin), 0.0, 0.0, cols, rows, 64, 48, 0, 0)
        best = second = 256
        best_idx = -1
        for i2 in cand:
            d = int(D[i1, i2])
            if mdist[i2] <= d:
                continue
            if d < best:
                second, best, best_idx = best, d, int(i2)
            elif d < second:
                second = d
        if best > 50 or F(second) * F(lowe_ratio) < F(best):
            continue
        if m21[best_idx] >= 0:
            m12[m21[best_idx]] = -1
        m12[i1], m21[best_idx], mdist[best_idx] = best_idx, i1, best
        hist_delta.append(F(ang_1[i1]) - F(ang_2[best_idx]))
        hist_owner.append(i1)
    if check_orientation and hist_delta:
        bad = angle_checker_invalid(np.asarray(hist_delta, F))
        m12[np.asarray(hist_owner)[bad]] = -1
    prev = np.array(prev_matched_pts, F, copy=True)
    ok = m12 >= 0
    prev[ok, 0], prev[ok, 1] = np.asarray(xs_2, F)[m12[ok]], np.asarray(ys_2, F)[m12[ok]]
    return int(ok.sum()), m12, prev


# ---- rule 19: match::bow_tree -----------------------------------------------------------------------------------------------------------
def _bow_pairs(ang_a, desc_a, fv_a, live_a, ang_b, desc_b, fv_b, usable_b, lowe_ratio, check_orientation):
    """The walk both bow_tree matchers share: vocabulary nodes present on both sides in ascending id; side a's keypoints of the node in
    their stored order (those with a live landmark), each against side b's keypoints of the same node that are still free; strict-less
    best / second; accept iff best <= 50 and not (ratio * second < best). Returns match_of_b[idx_b] = idx_a (or -1) after the orientation
    filter (delta = angle_a - angle_b)."""
    F = np.float32
    D = hamming_matrix(desc_a, desc_b)
    match_of_b = np.full(len(ang_b), -1, np.int32)
    free_b = np.array(usable_b, bool, copy=True)
    deltas, keys = [], []
    for node in sorted(set(fv_a) & set(fv_b)):
        members_b = np.asarray(fv_b[node], np.int64)
        for ia in fv_a[node]:
            if not live_a[ia]:
                continue
            cand = members_b[free_b[members_b]]
            if len(cand) == 0:
                continue
            d = D[ia, cand]
            first = int(np.argmin(d))                       # np.argmin: the first of equals, as the strict '<' scan
            best = int(d[first])
            second = int(np.delete(d, first).min()) if len(d) > 1 else 256
            if best > 50 or F(lowe_ratio) * F(second) < F(best):
                continue
            ib = int(cand[first])
            match_of_b[ib], free_b[ib] = ia, False
            deltas.append(F(ang_a[ia]) - F(ang_b[ib]))
            keys.append(ib)
    if check_orientation and deltas:
        match_of_b[np.asarray(keys)[angle_checker_invalid(np.asarray(deltas, F))]] = -1
    return match_of_b


def bow_match_frame_and_keyframe(kf_angles, kf_desc, kf_fv, kf_has_landmark, frm_angles, frm_desc, frm_fv, lowe_ratio=0.6, check_orientation=True):
    """(number of matches, landmark_source[frame keypoint] = keyframe keypoint whose landmark it receives, or -1)."""
    live = np.ones(len(kf_angles), bool) if kf_has_landmark is None else np.asarray(kf_has_landmark, bool)
    m = _bow_pairs(kf_angles, kf_desc, kf_fv, live, frm_angles, frm_desc, frm_fv, np.ones(len(frm_angles), bool), lowe_ratio, check_orientation)
    return int((m >= 0).sum()), m


def bow_match_keyframes(ang_1, desc_1, fv_1, has_lm_1, ang_2, desc_2, fv_2, has_lm_2, lowe_ratio=0.6, check_orientation=True):
    """(number of matches, matched_2_in_1[keyframe-1 keypoint] = keyframe-2 keypoint or -1): as above with keyframe 1 as the walking side,
    keyframe 2's keypoints usable when they carry a live landmark and are not matched yet."""
    live_1 = np.ones(len(ang_1), bool) if has_lm_1 is None else np.asarray(has_lm_1, bool)
    live_2 = np.ones(len(ang_2), bool) if has_lm_2 is None else np.asarray(has_lm_2, bool)
    m21 = _bow_pairs(ang_1, desc_1, fv_1, live_1, ang_2, desc_2, fv_2, live_2, lowe_ratio, check_orientation)
    m12 = np.full(len(ang_1), -1, np.int32)
    hit = np.nonzero(m21 >= 0)[0]
    m12[m21[hit]] = hit
    return len(hit), m12


# ---- rule 23: match::robust::match_for_triangulation ----------------------------------------------------------------------------------------
def robust_match_for_triangulation(ang_1, oct_1, desc_1, fv_1, bearings_1, has_lm_1, x_right_1, ang_2, desc_2, fv_2, bearings_2, has_lm_2, x_right_2,
                                   E_12, epipole_in_2, scale_factors, check_orientation=True):
    """(number of matches, matched_2_in_1): the bow_tree walk over keypoints WITHOUT a landmark on either side. A keyframe-2 keypoint is
    eligible for keyframe-1 keypoint i iff it is free, its distance is <= 50, it is not within 3 degrees of the epipole (unless one of
    the two is a stereo keypoint) and the pair passes check_epipolar_constraint; the scan keeps 'd <= best', so among the eligible ones
    the smallest distance wins and the LAST of equals."""
    F = np.float32
    n1, n2 = len(ang_1), len(ang_2)
    D = hamming_matrix(desc_1, desc_2)
    b1, b2 = np.asarray(bearings_1, np.float64), np.asarray(bearings_2, np.float64)
    E, ep = np.asarray(E_12, np.float64).reshape(3, 3), np.asarray(epipole_in_2, np.float64)
    stereo_1 = np.zeros(n1, bool) if x_right_1 is None else np.asarray(x_right_1, F) >= 0
    stereo_2 = np.zeros(n2, bool) if x_right_2 is None else np.asarray(x_right_2, F) >= 0
    lm_1 = np.zeros(n1, bool) if has_lm_1 is None else np.asarray(has_lm_1, bool)
    lm_2 = np.zeros(n2, bool) if has_lm_2 is None else np.asarray(has_lm_2, bool)
    near_epipole = 0.99862953475 < (ep[0] * b2[:, 0] + ep[1] * b2[:, 1]) + ep[2] * b2[:, 2]
    # epipolar planes of all keyframe-2 bearings, seen from keyframe 1: e = E_12 b_2 (rows left to right)
    e = np.stack([(E[r, 0] * b2[:, 0] + E[r, 1] * b2[:, 1]) + E[r, 2] * b2[:, 2] for r in range(3)], 1)
    e_norm = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2])
    thr = 0.2 * np.pi / 180.0
    free_2 = ~lm_2
    m12 = np.full(n1, -1, np.int32)
    deltas, owners = [], []
    for node in sorted(set(fv_1) & set(fv_2)):
        members_2 = np.asarray(fv_2[node], np.int64)
        for i1 in fv_1[node]:
            if lm_1[i1]:
                continue
            cand = members_2[free_2[members_2]]
            cand = cand[D[i1, cand] <= 50]
            if not stereo_1[i1]:
                cand = cand[stereo_2[cand] | ~near_epipole[cand]]
            if len(cand) == 0:
                continue
            with np.errstate(invalid="ignore"):
                cos_res = ((e[cand, 0] * b1[i1, 0] + e[cand, 1] * b1[i1, 1]) + e[cand, 2] * b1[i1, 2]) / e_norm[cand]
                residual = np.pi / 2.0 - np.abs(np.arccos(cos_res))        # (the sign is upstream's: a negative cosine always passes)
            cand = cand[residual < thr * float(scale_factors[oct_1[i1]])]
            if len(cand) == 0:
                continue
            d = D[i1, cand]
            i2 = int(cand[np.nonzero(d == d.min())[0][-1]])
            m12[i1], free_2[i2] = i2, False
            deltas.append(F(ang_1[i1]) - F(ang_2[i2]))
            owners.append(i1)
    if check_orientation and deltas:
        m12[np.asarray(owners)[angle_checker_invalid(np.asarray(deltas, F))]] = -1
    return int((m12 >= 0).sum()), m12


# ---- rules 21 (match_frame_and_keyframe) and 22 (fuse::replace_duplication): landmarks with a valid distance range ----------------------------
def predict_scale_level(max_valid_dist, cam_to_lm_dist, log_scale_factor, n_levels):
    """landmark::predict_scale_level: ceil(logf(max_valid_dist_ / dist) / log_scale_factor) in float, clamped to [0, n_levels - 1]."""
    F = np.float32
    with np.errstate(divide="ignore", invalid="ignore"):
        lvl = np.ceil(np.log(F(max_valid_dist) / F(cam_to_lm_dist), dtype=F) / F(log_scale_factor))
    return int(min(max(int(lvl), 0), n_levels - 1))


def _in_valid_range(dist_min_max, dist):
    """The getters' gate: (float)(0.7 * min_valid_dist_) <= dist <= (float)(1.3 * max_valid_dist_)."""
    F = np.float32
    return not (dist < float(F(0.7 * float(dist_min_max[0]))) or float(F(1.3 * float(dist_min_max[1]))) < dist)


def projection_match_frame_and_keyframe(model, cam, cols, rows, xs, ys, octaves, angles, desc, pose_cw_curr, kf_angles, kf_pos_w, kf_dist_min_max,
                                        kf_lm_desc, scale_factors, log_scale_factor, margin, hamm_dist_thr, check_orientation=True, occupied=None,
                                        kf_valid=None):
    """assigned[i] = current keypoint that receives the keyframe's landmark i (or -1): reprojection with the current pose, the landmark's
    valid distance range, the predicted level's radius and the level window [pred - 1, pred + 1], free keypoints only (claims are
    sequential), nearest descriptor if <= hamm_dist_thr, orientation histogram of (keyframe angle - current angle)."""
    F = np.float32
    T = np.asarray(pose_cw_curr, float)
    centre = -T[:, :3].T @ T[:, 3]
    ok, u, v, _ = reproject_to_image(model, cam, T, kf_pos_w, 0.0, 0.0, float(cols), float(rows))
    occ = np.zeros(len(xs), bool) if occupied is None else np.asarray(occupied).astype(bool).copy()
    D = hamming_matrix(kf_lm_desc, desc)
    assigned = np.full(len(kf_pos_w), -1, np.int32)
    deltas, owners = [], []
    for i in range(len(kf_pos_w)):
        if (kf_valid is not None and not kf_valid[i]) or not ok[i]:
            continue
        dist = float(np.linalg.norm(np.asarray(kf_pos_w[i], float) - centre))
        if not _in_valid_range(kf_dist_min_max[i], dist):
            continue
        pred = predict_scale_level(kf_dist_min_max[i][1], dist, log_scale_factor, len(scale_factors))
        cand = keypoints_in_cell(xs, ys, octaves, F(u[i]), F(v[i]), F(margin) * F(scale_factors[pred]), 0.0, 0.0, cols, rows, 64, 48, pred - 1, pred + 1)
        cand = [k for k in cand if not occ[k]]
        if not cand:
            continue
        d = D[i, cand]
        best = int(np.argmin(d))
        if d[best] > hamm_dist_thr:
            continue
        assigned[i] = cand[best]
        occ[cand[best]] = True
        deltas.append(F(kf_angles[i]) - F(angles[cand[best]]))
        owners.append(i)
    if check_orientation and deltas:
        assigned[np.asarray(owners)[angle_checker_invalid(np.asarray(deltas, F))]] = -1
    return assigned


def fuse_replace_duplication(model, cam, cols, rows, xs, ys, octaves, desc, pose_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                             inv_level_sigma_sq, log_scale_factor, margin=3.0, x_right=None, lm_valid=None):
    """best_idx[l] = the keyframe keypoint landmark l would be fused with (or -1). No claims: every landmark sees all keypoints."""
    F = np.float32
    T = np.asarray(pose_cw, float)
    centre = -T[:, :3].T @ T[:, 3]
    ok, u, v, xr_lm = reproject_to_image(model, cam, T, lm_pos_w, 0.0, 0.0, float(cols), float(rows))
    D = hamming_matrix(lm_desc, desc)
    xs64, ys64 = np.asarray(xs, F).astype(float), np.asarray(ys, F).astype(float)
    best_idx = np.full(len(lm_pos_w), -1, np.int32)
    for l in range(len(lm_pos_w)):
        if (lm_valid is not None and not lm_valid[l]) or not ok[l]:
            continue
        ray = np.asarray(lm_pos_w[l], float) - centre
        dist = float(np.linalg.norm(ray))
        if not _in_valid_range(lm_dist_min_max[l], dist):
            continue
        if float(ray @ np.asarray(lm_normal[l], float)) < 0.5 * dist:
            continue
        pred = predict_scale_level(lm_dist_min_max[l][1], dist, log_scale_factor, len(scale_factors))
        cand = np.asarray(keypoints_in_cell(xs, ys, octaves, F(u[l]), F(v[l]), F(margin) * F(scale_factors[pred]), 0.0, 0.0, cols, rows, 64, 48), np.int64)
        if len(cand) == 0:
            continue
        lv = np.asarray(octaves)[cand]
        cand = cand[(lv >= pred - 1) & (lv <= pred)]
        ex, ey = u[l] - xs64[cand], v[l] - ys64[cand]
        e2 = ex * ex + ey * ey
        gate = np.full(len(cand), float(F(5.99146)))
        if x_right is not None:
            st = np.asarray(x_right, F)[cand] >= 0
            er = (F(xr_lm[l]) - np.asarray(x_right, F)[cand]).astype(float)        # float - float, widened for the sum
            e2 = np.where(st, e2 + er * er, e2)
            gate = np.where(st, float(F(7.81473)), gate)
        cand = cand[~(gate < e2 * np.asarray(inv_level_sigma_sq, F)[np.asarray(octaves)[cand]].astype(float))]
        if len(cand) == 0:
            continue
        d = D[l, cand]
        k = int(np.argmin(d))
        if d[k] <= 50:
            best_idx[l] = cand[k]
    return best_idx


# ---- rule 20: match::stereo::compute ----------------------------------------------------------------------------------------------------------
def stereo_compute(pyr_left, pyr_right, kps_left, desc_left, kps_right, desc_right, scale_factors, inv_scale_factors, focal_x_baseline,
                   true_baseline, outlier_factor=2.0, parabola_double=False):
    """(stereo_x_right, depths) per left keypoint (-1 where there is none). pyr_* = the extractors' level images (lists of uint8 arrays);
    keypoints as structured arrays with x, y, octave."""
    F = np.float32
    nl = len(kps_left)
    rows0 = pyr_left[0].shape[0]
    xr_out, depth_out = np.full(nl, -1.0, F), np.full(nl, -1.0, F)
    if nl == 0 or len(kps_right) == 0:
        return xr_out, depth_out
    D = hamming_matrix(desc_left, desc_right)
    sf, isf = np.asarray(scale_factors, F), np.asarray(inv_scale_factors, F)
    xr_all, yr, oct_r = np.asarray(kps_right["x"], F), np.asarray(kps_right["y"], F), np.asarray(kps_right["octave"])
    # the right keypoints of every image row: those whose band y +- 2 * scale covers it, in keypoint order
    band = F(2.0) * sf[oct_r]
    row_lo = np.maximum(np.floor(yr - band).astype(np.int64), 0)
    row_hi = np.minimum(np.ceil(yr + band).astype(np.int64), rows0 - 1)
    max_disp = F(focal_x_baseline) / F(true_baseline)
    accepted = []                                             # (L1 distance, left keypoint)
    for i in range(nl):
        x_l, y_l, lvl = F(kps_left["x"][i]), F(kps_left["y"][i]), int(kps_left["octave"][i])
        row = int(y_l)
        cand = np.nonzero((row_lo <= row) & (row <= row_hi))[0]
        cand = cand[np.abs(oct_r[cand] - lvl) <= 1]
        cand = cand[(xr_all[cand] >= x_l - max_disp) & (xr_all[cand] <= x_l)]
        if len(cand) == 0 or x_l < 0:
            continue
        d = D[i, cand]
        b = int(np.argmin(d))
        if d[b] >= 75:
            continue
        # sub-pixel: slide an 11 x 11 window (centre value removed) over the right level image, shifts -5 .. +5
        s = isf[lvl]
        cx_l, cy_l, cx_r = int(np.rint(x_l * s)), int(np.rint(y_l * s)), int(np.rint(xr_all[cand[b]] * s))
        img_l, img_r = pyr_left[lvl], pyr_right[lvl]
        if not (0 <= cx_r - 10 and cx_r + 11 < img_r.shape[1]):
            continue
        win_l = img_l[cy_l - 5:cy_l + 6, cx_l - 5:cx_l + 6].astype(F)
        win_l = win_l - win_l[5, 5]
        cost = np.empty(11, F)
        for k, shift in enumerate(range(-5, 6)):
            win_r = img_r[cy_l - 5:cy_l + 6, cx_r + shift - 5:cx_r + shift + 6].astype(F)
            cost[k] = np.abs(win_l - (win_r - win_r[5, 5])).sum(dtype=np.float64)      # integers < 2^24: exact in any order
        k = int(np.argmin(cost))
        if k == 0 or k == 10:
            continue
        c1, c2, c3 = cost[k - 1], cost[k], cost[k + 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            if parabola_double:      # rule 20's variant: the quotient in double, rounded to float once
                delta = F((float(c1) - float(c3)) / (2.0 * (float(c1) + float(c3) - 2.0 * float(c2))))
            else:
                delta = (c1 - c3) / (F(2.0) * (c1 + c3 - F(2.0) * c2))
        if delta < -1 or 1 < delta:
            continue
        x_r = sf[lvl] * (F(cx_r) + F(k - 5) + delta)
        disp = x_l - x_r
        if not (0 <= disp < max_disp):
            continue
        if disp <= 0:
            disp, x_r = F(0.01), x_l - F(0.01)
        xr_out[i], depth_out[i] = x_r, F(focal_x_baseline) / disp
        accepted.append((float(c2), i))
    if accepted:
        dists = sorted(a[0] for a in accepted)
        median = dists[len(dists) // 2]
        for c, i in accepted:
            if F(outlier_factor) * F(median) < F(c):
                xr_out[i] = depth_out[i] = -1.0
    return xr_out, depth_out


# ---- rule 27: the loop closer's matchers -------------------------------------------------------------------------------------------------------
def _sim3_to_pose(sim3_cw):
    """[sR | t'] -> (R | t'/s) with s = the length of sR's first row."""
    S = np.asarray(sim3_cw, float)
    s = np.sqrt(S[0, :3] @ S[0, :3])
    return np.concatenate([S[:, :3] / s, (S[:, 3] / s)[:, None]], 1)


def _sim3_candidates(model, cam, cols, rows, xs, ys, octaves, T, pos_w, dist_min_max, normal, scale_factors, log_scale_factor, margin, valid):
    """Per landmark that passes rule 22's gates under pose T: (landmark, candidate keypoints of levels [pred - 1, pred] inside the radius)."""
    F = np.float32
    centre = -T[:, :3].T @ T[:, 3]
    ok, u, v, _ = reproject_to_image(model, cam, T, pos_w, 0.0, 0.0, float(cols), float(rows))
    for l in range(len(pos_w)):
        if (valid is not None and not valid[l]) or not ok[l]:
            continue
        ray = np.asarray(pos_w[l], float) - centre
        dist = float(np.linalg.norm(ray))
        if not _in_valid_range(dist_min_max[l], dist) or float(ray @ np.asarray(normal[l], float)) < 0.5 * dist:
            continue
        pred = predict_scale_level(dist_min_max[l][1], dist, log_scale_factor, len(scale_factors))
        cand = np.asarray(keypoints_in_cell(xs, ys, octaves, F(u[l]), F(v[l]), F(margin) * F(scale_factors[pred]), 0.0, 0.0, cols, rows, 64, 48), np.int64)
        if len(cand):
            lv = np.asarray(octaves)[cand]
            cand = cand[(lv >= pred - 1) & (lv <= pred)]
        yield l, cand


def fuse_detect_duplication(model, cam, cols, rows, xs, ys, octaves, desc, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                            log_scale_factor, margin, lm_valid=None):
    """best_idx[l]: rule 22 without the chi-square gate, under the pose recovered from the Sim3; no claims."""
    D = hamming_matrix(lm_desc, desc)
    best_idx = np.full(len(lm_pos_w), -1, np.int32)
    for l, cand in _sim3_candidates(model, cam, cols, rows, xs, ys, octaves, _sim3_to_pose(sim3_cw), lm_pos_w, lm_dist_min_max, lm_normal, scale_factors,
                                    log_scale_factor, margin, lm_valid):
        if len(cand):
            k = int(np.argmin(D[l, cand]))
            if D[l, cand[k]] <= 50:
                best_idx[l] = cand[k]
    return best_idx


def projection_match_by_sim3_transform(model, cam, cols, rows, xs, ys, octaves, desc, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc,
                                       scale_factors, log_scale_factor, margin, occupied=None, lm_valid=None):
    """assigned[l]: the same gates; a keypoint holding a match (given, or made earlier in this call) is not a candidate; best <= 50."""
    D = hamming_matrix(lm_desc, desc)
    occ = np.zeros(len(xs), bool) if occupied is None else np.asarray(occupied).astype(bool).copy()
    assigned = np.full(len(lm_pos_w), -1, np.int32)
    for l, cand in _sim3_candidates(model, cam, cols, rows, xs, ys, octaves, _sim3_to_pose(sim3_cw), lm_pos_w, lm_dist_min_max, lm_normal, scale_factors,
                                    log_scale_factor, margin, lm_valid):
        cand = cand[~occ[cand]] if len(cand) else cand
        if len(cand):
            k = int(np.argmin(D[l, cand]))
            if D[l, cand[k]] <= 50:
                assigned[l] = cand[k]
                occ[cand[k]] = True
    return assigned


def projection_match_keyframes_mutually(cam, cols, rows, kps_1, desc_1, pose_cw_1, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, kps_2, desc_2, pose_cw_2,
                                        lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, s_12, rot_12, trans_12, scale_factors, log_scale_factor, margin):
    """(number of pairs, matched_2_in_1): each keyframe's landmarks moved into the other keyframe's camera by Sim3_12 resp. its inverse
    (perspective cameras), nearest descriptor (<= 100) among the keypoints of levels [pred - 1, pred] within the radius, no claims; a pair
    stays iff both directions choose each other."""
    F = np.float32
    fx, fy, cx, cy = cam
    R12, t12 = np.asarray(rot_12, float).reshape(3, 3), np.asarray(trans_12, float)

    def one_way(T_from, pos_w, dist_mm, lm_desc, valid, A, b, kps_to, desc_to):
        D = hamming_matrix(lm_desc, desc_to)
        p = (np.asarray(pos_w, float) @ T_from[:, :3].T + T_from[:, 3]) @ A.T + b
        pick = np.full(len(pos_w), -1, np.int64)
        for l in range(len(pos_w)):
            x, y, z = p[l]
            if (valid is not None and not valid[l]) or z <= 0:
                continue
            u, v = fx * x / z + cx, fy * y / z + cy
            if not (0.0 <= u <= cols and 0.0 <= v <= rows):
                continue
            dist = float(np.sqrt(x * x + y * y + z * z))
            if not _in_valid_range(dist_mm[l], dist):
                continue
            pred = predict_scale_level(dist_mm[l][1], dist, log_scale_factor, len(scale_factors))
            cand = np.asarray(keypoints_in_cell(kps_to["x"], kps_to["y"], kps_to["octave"], F(u), F(v), F(margin) * F(scale_factors[pred]), 0.0, 0.0, cols,
                                                rows, 64, 48), np.int64)
            if len(cand):
                lv = kps_to["octave"][cand]
                cand = cand[(lv >= pred - 1) & (lv <= pred)]
            if len(cand):
                k = int(np.argmin(D[l, cand]))
                if D[l, cand[k]] <= 100:
                    pick[l] = cand[k]
        return pick

    A21 = R12.T / s_12
    in_2 = one_way(np.asarray(pose_cw_1, float), lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, A21, -A21 @ t12, kps_2, desc_2)
    in_1 = one_way(np.asarray(pose_cw_2, float), lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, s_12 * R12, t12, kps_1, desc_1)
    out = np.full(len(kps_1), -1, np.int32)
    for i1, i2 in enumerate(in_2):
        if i2 >= 0 and in_1[i2] == i1:
            out[i1] = i2
    return int((out >= 0).sum()), out


# ---- rule 29: DBoW2 TemplatedVocabulary::transform ---------------------------------------------------------------------------------------------
def bow_transform(vocab, desc, levelsup=4):
    """(word id, weight, node id at `levelsup` levels above the words) of every descriptor: all features descend together, level by level, each
    to its nearest child (first of equals in child order); a feature stops at a leaf. vocab = dict(child_start, children, desc, weight,
    word_id, depth)."""
    cs, ch = np.asarray(vocab["child_start"], np.int64), np.asarray(vocab["children"], np.int64)
    bits_nodes = np.unpackbits(np.asarray(vocab["desc"], np.uint8), axis=1).astype(np.int32)
    bits = np.unpackbits(np.asarray(desc, np.uint8).reshape(-1, 32), axis=1).astype(np.int32)
    n, depth = len(bits), int(vocab["depth"])
    cur = np.zeros(n, np.int64)
    node_at = np.zeros(n, np.int64)                     # the root, unless the descent passes level depth - levelsup
    level = 0
    while True:
        if level == depth - levelsup:
            node_at = cur.copy()
        n_kids = cs[cur + 1] - cs[cur]
        moving = np.nonzero(n_kids > 0)[0]
        if len(moving) == 0:
            break
        width = int(n_kids[moving].max())
        slot = np.arange(width)[None, :]
        kid = ch[np.minimum(cs[cur[moving]][:, None] + slot, len(ch) - 1)]                                   # (m, width), ragged tails masked below
        dist = (bits[moving][:, None, :] != bits_nodes[kid]).sum(2)
        dist = np.where(slot < n_kids[moving][:, None], dist, 1 << 20)
        cur[moving] = kid[np.arange(len(moving)), np.argmin(dist, axis=1)]
        level += 1
    if levelsup <= 0:
        node_at = cur.copy()
    return np.asarray(vocab["word_id"])[cur].astype(np.int32), np.asarray(vocab["weight"], np.float64)[cur], node_at.astype(np.int32)
