#!/usr/bin/env python3
"""Executable model of the data-parallel quad-tree distribution used by the HIP kernel (csrc/orb_tree.hip).

The oracle (oracle/ovo_orb.cc distribute_via_tree) mutates a std::list exactly like upstream. The kernel cannot; it uses
closed-form positions instead:
  * a pass splits a set S of nodes in a processing order (phase 1: list order of all non-leaf nodes; phase 2: sorted by
    (count desc, list position asc), cut at the first split that makes #nodes >= N);
  * children are push_front'ed in creation order => new list = reverse(creation sequence) ++ (old list minus S);
  * per-node max response with the first-in-emission-order tie rule = max over key (score, -order).
This file is test infrastructure: tests/test_tree_model.py checks it against the oracle on adversarial inputs.
"""
import math
import numpy as np


def tree_model(xs, ys, scores, min_x, max_x, min_y, max_y, N, switch_factor=3, tie_order=0):
    # switch_factor / tie_order: ORACLE_SPEC's run-time variants of rules 6 and 7 (3 | 1; equal counts: later-created node first | earlier first)
    n = len(xs)
    if n == 0:
        return []
    xs = np.asarray(xs, np.int64)
    ys = np.asarray(ys, np.int64)
    order = np.arange(n)
    W, H = max_x - min_x, max_y - min_y
    ratio = W / H
    if ratio > 1:
        gx, gy = int(np.round(ratio)), 1          # np.round: half-to-even vs std::round half-away; ratio .5 never for ints? handled below
        gx = int(math.floor(ratio + 0.5))
        dx, dy = W / gx, float(H)
    else:
        gx = 1
        gy = int(math.floor(1 / ratio + 0.5))
        dx, dy = float(W), H / gy
    nroot = gx * gy
    rb = []
    for i in range(nroot):
        ix, iy = i % gx, i // gx
        rb.append((int(dx * ix), int(dx * (ix + 1)), int(dy * iy), int(dy * (iy + 1))))
    ridx = np.minimum((xs / dx).astype(np.int64), gx - 1) + np.minimum((ys / dy).astype(np.int64), gy - 1) * gx
    rcount = np.bincount(ridx, minlength=nroot)
    # list = non-empty roots in index order
    pos_of_root = -np.ones(nroot, np.int64)
    nodes = []   # (bx, ex, by, ey, count)
    for i in range(nroot):
        if rcount[i] > 0:
            pos_of_root[i] = len(nodes)
            nodes.append((rb[i][0], rb[i][1], rb[i][2], rb[i][3], int(rcount[i])))
    nd = pos_of_root[ridx]
    phase = 1
    while True:
        prev = len(nodes)
        L = len(nodes)
        bx = np.array([t[0] for t in nodes]); ex = np.array([t[1] for t in nodes])
        by = np.array([t[2] for t in nodes]); ey = np.array([t[3] for t in nodes])
        cnt = np.array([t[4] for t in nodes])
        nonleaf = cnt > 1
        cx = bx + np.ceil((ex - bx) / 2.0).astype(np.int64)
        cy = by + np.ceil((ey - by) / 2.0).astype(np.int64)
        child = (xs >= cx[nd]).astype(np.int64) + 2 * (ys >= cy[nd]).astype(np.int64)
        cc = np.zeros((L, 4), np.int64)
        np.add.at(cc, (nd, child), 1)
        cc[~nonleaf] = 0
        nch = (cc > 0).sum(1)
        # processing order
        if phase == 1:
            proc = [i for i in range(L) if nonleaf[i]]
        else:
            pool = [i for i in range(L) if nonleaf[i]]
            pool.sort(key=lambda i: (-cnt[i], i if tie_order == 0 else -i))
            size = L
            proc = []
            for i in pool:
                proc.append(i)
                size += nch[i] - 1
                if N <= size:
                    break
        in_s = np.zeros(L, bool)
        in_s[proc] = True
        total_new = int(sum(nch[i] for i in proc))
        new_nodes = [None] * (total_new + int((~in_s).sum()))
        cmap = -np.ones((L, 4), np.int64)
        c = 0
        for i in proc:
            for k in range(4):
                if cc[i, k] > 0:
                    p = total_new - 1 - c
                    cmap[i, k] = p
                    nbx, nex = (bx[i], cx[i]) if (k & 1) == 0 else (cx[i], ex[i])
                    nby, ney = (by[i], cy[i]) if (k & 2) == 0 else (cy[i], ey[i])
                    new_nodes[p] = (int(nbx), int(nex), int(nby), int(ney), int(cc[i, k]))
                    c += 1
        keep_pos = -np.ones(L, np.int64)
        r = 0
        for i in range(L):
            if not in_s[i]:
                keep_pos[i] = total_new + r
                new_nodes[total_new + r] = nodes[i]
                r += 1
        nd = np.where(in_s[nd], cmap[nd, child], keep_pos[nd])
        nodes = new_nodes
        size = len(nodes)
        npool = sum(1 for t in nodes[:total_new] if t[4] > 1)
        if phase == 1:
            if N <= size or size == prev:
                break
            if N < size + switch_factor * npool:
                phase = 2
        else:
            if N <= size or size == prev:
                break
    # per-node max response, first in emission order
    best = {}
    for i in range(n):
        key = (int(scores[i]), -int(order[i]))
        if nd[i] not in best or key > best[nd[i]][0]:
            best[nd[i]] = (key, i)
    return [best[p][1] for p in range(len(nodes))]


# ---- candidate formulation for the next version of csrc/orb_tree.hip (modelled here, not in the kernel yet) -----------------------------
# A node's split point is a function of its bounds alone and the bounds are a function of the root patch and the path from it, so the
# quadrant a candidate falls into at every depth can be computed ONCE, before the first pass: a PATH CODE of two bits per depth. With the
# candidates sorted by (root, path code) every node of every pass is a contiguous range of the sorted array: a node's count is the length
# of its range, its four children are found by four binary searches, and no pass touches the candidates at all; what remains per pass is
# work on <= 4 N nodes. Only the last step looks at the candidates again (maximum response per final node = a segmented maximum over the
# sorted array). Whether that pays is open: the bench frame's level 0 (7 899 candidates, N = 434) needs only FOUR passes (2 -> 8 -> 32 -> 128
# -> 435 nodes, `stats`), so the kernel's 50 us per frame are not mostly sweeps, and the sort is not free either.
kTreeDepth = 12   # two bits per depth; bounds of a 4096-px patch collapse to one pixel within 12 halvings


def path_codes(xs, ys, bx, ex, by, ey, depth=kTreeDepth):
    """Quadrant index at every depth for candidates inside the root patch [bx, ex) x [by, ey): code = sum child_d << 2 (depth-1-d)."""
    xs = np.asarray(xs, np.int64)
    ys = np.asarray(ys, np.int64)
    bx = np.full(len(xs), bx, np.int64)
    ex = np.full(len(xs), ex, np.int64)
    by = np.full(len(xs), by, np.int64)
    ey = np.full(len(xs), ey, np.int64)
    code = np.zeros(len(xs), np.int64)
    for _ in range(depth):
        cx = bx + (ex - bx + 1) // 2          # ceil((ex - bx) / 2) for non-negative widths
        cy = by + (ey - by + 1) // 2
        right, low = xs >= cx, ys >= cy
        code = code * 4 + right.astype(np.int64) + 2 * low.astype(np.int64)
        bx = np.where(right, cx, bx)
        ex = np.where(right, ex, cx)
        by = np.where(low, cy, by)
        ey = np.where(low, ey, cy)
    return code


def tree_model_sorted(xs, ys, scores, min_x, max_x, min_y, max_y, N, stats=None):
    """Same result as tree_model() (and the oracle), computed without per-pass candidate sweeps. stats (optional dict): passes, deepest node,
    nodes searched per pass."""
    n = len(xs)
    if n == 0:
        return []
    xs = np.asarray(xs, np.int64)
    ys = np.asarray(ys, np.int64)
    W, H = max_x - min_x, max_y - min_y
    ratio = W / H
    if ratio > 1:
        gx, gy = int(math.floor(ratio + 0.5)), 1
        dx, dy = W / gx, float(H)
    else:
        gx, gy = 1, int(math.floor(1 / ratio + 0.5))
        dx, dy = float(W), H / gy
    nroot = gx * gy
    ridx = np.minimum((xs / dx).astype(np.int64), gx - 1) + np.minimum((ys / dy).astype(np.int64), gy - 1) * gx
    key = np.zeros(n, np.int64)
    for r in range(nroot):
        ix, iy = r % gx, r // gx
        m = ridx == r
        key[m] = (r << (2 * kTreeDepth)) | path_codes(xs[m], ys[m], int(dx * ix), int(dx * (ix + 1)), int(dy * iy), int(dy * (iy + 1)))
    perm = np.argsort(key, kind="stable")
    skey = key[perm]

    def rng_of(prefix, depth):   # candidates of the node (root bits included in prefix) = a range of the sorted keys
        sh = 2 * (kTreeDepth - depth)
        return int(np.searchsorted(skey, prefix << sh, "left")), int(np.searchsorted(skey, (prefix + 1) << sh, "left"))

    nodes = []   # (prefix, depth, lo, hi) in list order
    for r in range(nroot):
        lo, hi = rng_of(r, 0)
        if hi > lo:
            nodes.append((r, 0, lo, hi))
    phase = 1
    if stats is not None:
        stats.update(passes=0, max_depth=0, searched=[])
    while True:
        prev = L = len(nodes)
        cnt = np.array([t[3] - t[2] for t in nodes])
        nonleaf = cnt > 1
        kids = {}
        nch = np.zeros(L, np.int64)
        for i in range(L):
            if nonleaf[i]:
                p, d = nodes[i][0], nodes[i][1]
                assert d < kTreeDepth
                kids[i] = [(4 * p + k, d + 1) + rng_of(4 * p + k, d + 1) for k in range(4)]
                nch[i] = sum(1 for c in kids[i] if c[3] > c[2])
        if phase == 1:
            proc = [i for i in range(L) if nonleaf[i]]
        else:
            pool = [i for i in range(L) if nonleaf[i]]
            pool.sort(key=lambda i: (-cnt[i], i))
            size, proc = L, []
            for i in pool:
                proc.append(i)
                size += nch[i] - 1
                if N <= size:
                    break
        in_s = np.zeros(L, bool)
        in_s[proc] = True
        total_new = int(sum(nch[i] for i in proc))
        new_nodes = [None] * (total_new + int((~in_s).sum()))
        c = 0
        for i in proc:
            for child in kids[i]:
                if child[3] > child[2]:
                    new_nodes[total_new - 1 - c] = child
                    c += 1
        r = 0
        for i in range(L):
            if not in_s[i]:
                new_nodes[total_new + r] = nodes[i]
                r += 1
        nodes = new_nodes
        size = len(nodes)
        if stats is not None:
            stats["passes"] += 1
            stats["max_depth"] = max([stats["max_depth"]] + [t[1] for t in nodes])
            stats["searched"].append(int(nonleaf.sum()))
        npool = sum(1 for t in nodes[:total_new] if t[3] - t[2] > 1)
        if phase == 1:
            if N <= size or size == prev:
                break
            if N < size + 3 * npool:
                phase = 2
        elif N <= size or size == prev:
            break
    out = []
    for (_, _, lo, hi) in nodes:   # maximum response, first in emission order
        members = perm[lo:hi]
        out.append(int(max(members, key=lambda i: (int(scores[i]), -int(i)))))
    return out
