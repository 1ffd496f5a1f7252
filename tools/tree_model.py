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


def tree_model(xs, ys, scores, min_x, max_x, min_y, max_y, N):
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
            pool.sort(key=lambda i: (-cnt[i], i))
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
            if N < size + 3 * npool:
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
