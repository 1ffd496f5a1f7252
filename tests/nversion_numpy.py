"""A SECOND, independent restatement of three OpenCV stages of the extraction -- cv::resize(INTER_LINEAR, 8UC1), cv::FAST(TYPE_9_16) with
its corner score and non-maximum suppression, and the 7x7 sigma-2 fixed-point GaussianBlur -- written in vectorised numpy from the
algorithm text of oracle/ORACLE_SPEC.md (rules 3, 5, 10) and the public description of those OpenCV functions, NOT from oracle/ovo_orb.cc
(whole-array formulation, no per-pixel loops, different decomposition). tests/test_nversion.py compares it with the C oracle bit for bit.

Why: the reference tree is absent, so the oracle cannot be pinned to it ("What's missing" #1). Two implementations
written separately from the same specification that agree on every pixel of the golden inputs is the strongest evidence available here
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
