"""Numpy model of k_fast_cells pre-tests (round 3): checks that the quantised four- / eight-diameter conditions are NECESSARY for the exact
FAST-9/16 corner condition S > t on synthetic frames at three pyramid scales, and prints the pass rates (exact 8-bit, 7-bit, 6-bit).
Usage: python tools/fast_pretest_model.py"""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from openvslam_amd.synth import synth_frame
RING = [(0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),(-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)]
def ring_vals(img):
    H,W = img.shape
    c = img[3:H-3,3:W-3].astype(np.int32)
    r = np.stack([img[3+dy:H-3+dy, 3+dx:W-3+dx].astype(np.int32) for dx,dy in RING])
    return c, r
def exact_S(c, r):
    d = r - c[None]
    best = np.zeros_like(c)
    for j in range(16):
        idx = [(j+k) % 16 for k in range(9)]
        a = d[idx].min(0); b = (-d[idx]).min(0)
        best = np.maximum(best, np.maximum(a, b))
    return best
def diam_exact(c, r, t, pos=(0,2,4,6)):
    br = np.ones(c.shape, bool); dk = np.ones(c.shape, bool)
    for i in pos:
        br &= np.maximum(r[i], r[i+8]) > c + t
        dk &= np.minimum(r[i], r[i+8]) < c - t
    return br | dk
def diam_q(c, r, t, sh, th, pos=(0,2,4,6)):
    cq = c >> sh; rq = r >> sh
    br = np.ones(c.shape, bool); dk = np.ones(c.shape, bool)
    for i in pos:
        br &= np.maximum(rq[i], rq[i+8]) >= cq + th
        dk &= np.minimum(rq[i], rq[i+8]) <= cq - th
    return br | dk
for lvl, scale in enumerate([1.0, 1.2**3, 1.2**7]):
    img = synth_frame(1080, 1920, seed=3)
    if scale > 1:
        # crude resize by area sampling for statistics only
        H, W = int(round(1080/scale)), int(round(1920/scale))
        ys = (np.arange(H)*scale).astype(int); xs = (np.arange(W)*scale).astype(int)
        img = img[ys][:, xs]
    c, r = ring_vals(img)
    for t in (20, 7):
        S = exact_S(c, r)
        corner = S > t
        e4 = diam_exact(c, r, t)
        e8 = diam_exact(c, r, t, range(8))
        q7 = diam_q(c, r, t, 1, (t+1)>>1)
        q7s = diam_q(c, r, t, 1, ((t+1)>>1)-1)
        q6 = diam_q(c, r, t, 2, (t+1)>>2)
        q78 = diam_q(c, r, t, 1, (t+1)>>1, range(8))
        q68 = diam_q(c, r, t, 2, (t+1)>>2, range(8))
        assert not (corner & ~e4).any() and not (corner & ~q7).any() and not (corner & ~q6).any() and not (corner & ~q78).any() and not (corner & ~q68).any()
        print(f"scale {scale:.2f} t={t}: corners {corner.mean()*100:.2f}%  exact4 {e4.mean()*100:.2f}  exact8 {e8.mean()*100:.2f}  q7 {q7.mean()*100:.2f}  q7slack {q7s.mean()*100:.2f}  q6 {q6.mean()*100:.2f}  q7x8 {q78.mean()*100:.2f} q6x8 {q68.mean()*100:.2f}")
