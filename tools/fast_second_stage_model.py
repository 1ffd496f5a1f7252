"""Numpy model for round 5's first FAST experiment: how many of the candidates the 6-bit four-diameter pre-test of k_fast_cells lets
through would survive a SECOND, exact stage evaluated on the candidates only -- (a) the four odd diameters on 8-bit values (with the pre-test's
four even ones that is OpenCV's own eight-diameter test), (b) the same plus the exact 8-bit form of the even ones, (c) a cheap arc test on the
16 brighter / darker bits (9 contiguous set bits, the corner condition itself minus the score) -- on the bench's synthetic video at the real
pyramid scales, weighted by the level areas. Every stage must be NECESSARY for S > t (asserted). Usage: python tools/fast_second_stage_model.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import nversion_numpy as nv   # noqa: E402
from openvslam_amd.synth import synth_frame   # noqa: E402

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def stages(img, t):
    H, W = img.shape
    c = img[3:H - 3, 3:W - 3].astype(np.int32)
    r = np.stack([img[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx].astype(np.int32) for dx, dy in RING])
    S = nv.fast_strength(img)[3:H - 3, 3:W - 3]
    corner = S > t

    def diam(vals, centre, thr, pos, ge):
        br = np.ones(c.shape, bool)
        dk = np.ones(c.shape, bool)
        for i in pos:
            hi, lo = np.maximum(vals[i], vals[i + 8]), np.minimum(vals[i], vals[i + 8])
            br &= (hi >= centre + thr) if ge else (hi > centre + thr)
            dk &= (lo <= centre - thr) if ge else (lo < centre - thr)
        return br, dk

    b6, d6 = diam(r >> 2, c >> 2, (t + 1) >> 2, (0, 2, 4, 6), True)
    pre = b6 | d6                                     # what the kernel scores today
    bo, do = diam(r, c, t, (1, 3, 5, 7), False)
    stage_a = (b6 & bo) | (d6 & do)                   # + the odd diameters, exact, polarity-wise
    be, de = diam(r, c, t, (0, 2, 4, 6), False)
    stage_b = (b6 & bo & be) | (d6 & do & de)         # + the even ones in exact form as well
    bright = r > c + t
    dark = r < c - t
    ext_b, ext_d = np.concatenate([bright, bright[:8]]), np.concatenate([dark, dark[:8]])
    arc_b = np.zeros(c.shape, bool)
    arc_d = np.zeros(c.shape, bool)
    for j in range(16):
        arc_b |= ext_b[j:j + 9].all(0)
        arc_d |= ext_d[j:j + 9].all(0)
    stage_c = arc_b | arc_d                           # the corner condition itself (== S > t)
    assert not (corner & ~pre).any() and not (corner & ~stage_a).any() and not (corner & ~stage_b).any() and np.array_equal(stage_c, corner)
    return np.array([c.size, corner.sum(), pre.sum(), stage_a.sum(), stage_b.sum()], np.float64)


def main():
    tot = {20: np.zeros(5), 7: np.zeros(5)}
    sf = np.float32(1.0)
    img0 = synth_frame(1080, 1920, seed=3)
    cur = img0
    for level in range(8):
        if level:
            sf = np.float32(1.2) * sf
            cur = nv.resize_linear_u8(cur, int(np.floor(1080 / float(sf) + 0.5)), int(np.floor(1920 / float(sf) + 0.5)))
        for t in (20, 7):
            s = stages(cur, t)
            tot[t] += s
            print("level %d %4dx%-4d t=%2d: corners %5.2f %%  6-bit pre-test %5.2f %%  + odd diameters %5.2f %%  + exact even %5.2f %%" %
                  (level, cur.shape[1], cur.shape[0], t, *(100 * s[1:] / s[0])))
    for t in (20, 7):
        s = tot[t]
        print("whole pyramid t=%2d: corners %5.2f %%  pre-test %5.2f %%  + odd %5.2f %% (x %.2f of today's candidates)  + exact even %5.2f %% (x %.2f)" %
              (t, 100 * s[1] / s[0], 100 * s[2] / s[0], 100 * s[3] / s[0], s[3] / s[2], 100 * s[4] / s[0], s[4] / s[2]))


if __name__ == "__main__":
    main()
