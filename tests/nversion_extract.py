"""TEST INFRASTRUCTURE. orb_extractor::extract as ONE numpy pipeline with none of the oracle's C code in it: the tables of rule 1, the pyramid
of rules 2 / 3, the cell loop of rule 4 over the numpy FAST of rule 5, the quad-tree of rules 6 - 8 in its closed-form version
(tools/tree_model.py), orientation (rule 9), blur (rule 10), steered rBRIEF (rule 11) and the scaling of rule 13 -- all written from
oracle/ORACLE_SPEC.md. tests/test_nversion.py requires that it reproduces the committed golden keypoints and descriptors and equals the C
oracle on other frames. Never imported by the product."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import nversion_numpy as nv        # noqa: E402
from tree_model import tree_model  # noqa: E402

F = np.float32
KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
BORDER, CELL, OVERLAP, PATCH = 19, 64, 6, 31


def tables(max_num_keypts, scale_factor, num_levels):
    """Rule 1: scale factors by repeated float multiplication; per-level budgets = the rounded geometric shares (double), the last level takes
    what is left."""
    sf = np.ones(num_levels, F)
    for l in range(1, num_levels):
        sf[l] = F(scale_factor) * sf[l - 1]
    q = 1.0 / float(F(scale_factor))
    share = max_num_keypts * (1.0 - q) / (1.0 - q ** num_levels)
    budget = []
    for _ in range(num_levels - 1):
        budget.append(int(np.floor(share + 0.5)))
        share *= q
    budget.append(max(max_num_keypts - sum(budget), 0))
    return sf, budget


def pyramid(img, sf):
    """Rules 2 / 3: level l has round(size / (double)sf[l]) of the ORIGINAL size and is resized from level l - 1."""
    rows, cols = img.shape
    out = [np.ascontiguousarray(img)]
    for l in range(1, len(sf)):
        out.append(nv.resize_linear_u8(out[-1], int(np.floor(rows / float(sf[l]) + 0.5)), int(np.floor(cols / float(sf[l]) + 0.5))))
    return out


def fast_candidates(level_img, ini_thr, min_thr):
    """Rule 4: 64-px cells with 6 px of overlap between the 19-px borders; per cell FAST at ini_thr, at min_thr if that finds nothing; the
    cell's keypoints are shifted by the cell's origin (coordinates relative to the border). Emission order = cell rows, cell columns,
    row-major inside a cell."""
    rows, cols = level_img.shape
    max_x, max_y = cols - BORDER, rows - BORDER
    if max_x <= BORDER or max_y <= BORDER:
        return np.zeros(0, F), np.zeros(0, F), np.zeros(0, F)
    n_cols, n_rows = (max_x - BORDER) // CELL + 1, (max_y - BORDER) // CELL + 1
    xs, ys, sc = [], [], []
    for i in range(n_rows):
        y0 = BORDER + i * CELL
        if max_y - OVERLAP <= y0:
            continue
        y1 = min(y0 + CELL + OVERLAP, max_y)
        for j in range(n_cols):
            x0 = BORDER + j * CELL
            if max_x - OVERLAP <= x0:
                continue
            x1 = min(x0 + CELL + OVERLAP, max_x)
            cell = level_img[y0:y1, x0:x1]
            cx, cy, cs = nv.fast9_16(cell, ini_thr, True)
            if len(cx) == 0:
                cx, cy, cs = nv.fast9_16(cell, min_thr, True)
            xs.append(cx + j * CELL)
            ys.append(cy + i * CELL)
            sc.append(cs)
    if not xs:
        return np.zeros(0, F), np.zeros(0, F), np.zeros(0, F)
    return np.concatenate(xs).astype(F), np.concatenate(ys).astype(F), np.concatenate(sc).astype(F)


def extract(img, pattern, max_num_keypts=1000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7, tree_switch_factor=3, tree_tie_order=0,
            blur_taps=0):
    """(keypoints, descriptors, candidates per level) of one grey image; the last three arguments are ORACLE_SPEC's run-time variants."""
    sf, budget = tables(max_num_keypts, scale_factor, num_levels)
    kps, descs, n_cand = [], [], []
    for l, lvl_img in enumerate(pyramid(np.asarray(img, np.uint8), sf)):
        cx, cy, cs = fast_candidates(lvl_img, ini_fast_thr, min_fast_thr)
        n_cand.append(len(cx))
        if len(cx) == 0:
            continue
        rows, cols = lvl_img.shape
        keep = np.asarray(tree_model(cx, cy, cs, BORDER, cols - BORDER, BORDER, rows - BORDER, budget[l], tree_switch_factor, tree_tie_order), np.int64)
        px, py = cx[keep].astype(np.int64) + BORDER, cy[keep].astype(np.int64) + BORDER
        ang = nv.ic_angle(lvl_img, px, py)
        k = np.zeros(len(keep), KP)
        k["x"], k["y"] = px.astype(F) * sf[l], py.astype(F) * sf[l]                     # rule 13
        k["size"] = F(np.uint32(F(PATCH) * sf[l]))
        k["angle"], k["response"], k["octave"], k["class_id"] = ang, cs[keep], l, -1
        kps.append(k)
        descs.append(nv.orb_descriptors(nv.gaussian_blur_7x7(lvl_img, blur_taps), px, py, ang, pattern))
    if not kps:
        return np.zeros(0, KP), np.zeros((0, 32), np.uint8), n_cand
    return np.concatenate(kps), np.concatenate(descs), n_cand
