"""CPU ORACLE (test infrastructure, see ovo_oracle.h) for B4: optimize::local_bundle_adjuster::optimize behind the graph build
(expected: src/openvslam/optimize/local_bundle_adjuster.cc; g2o OptimizationAlgorithmLevenberg, BlockSolver_6_3, RobustKernelHuber).

Restated from the published g2o algorithm (upstream and g2o sources are absent: parity unpinned), ORACLE_SPEC rules 25 and 28:
  round 1: all edges, Huber (one delta per rig: Monocular sqrtf(5.99146f), otherwise sqrtf(7.81473f)), num_first_iter LM iterations;
  outliers (chi2 > 5.99146f mono edge / 7.81473f stereo edge, or depth <= 0) go to level 1, kernels are dropped;
  round 2: num_second_iter iterations on the inliers; final outlier test (level-1 edges keep their round-1 chi2, depth is re-tested).
LM: lambda0 = 1e-5 max|diag H| over the active vertices; trial (H + lambda I) dx = b by landmark elimination + dense Cholesky;
rho = (chi - chi_new) / (dx.(lambda dx + b) + 1e-3); accept (rho > 0): lambda *= clamp(1 - (2 rho - 1)^3, 1/3, 2/3), ni = 2; reject:
lambda *= ni, ni *= 2, at most 10 trials; an iteration that ends with rho == 0 or 10 rejections ends the round.
Update: T <- exp([omega, upsilon]) T on rotation matrices (SE3Quat::exp, V = R = I + O + O^2 below 1e-5 rad), X += dx.
The blocks come from the C oracle (ovo_ba_linearize / ovo_ba_linearize_stereo); the solve is numpy.
"""
import numpy as np

from . import binding as ob

# upstream: constexpr float chi_sq_2D = 5.99146, chi_sq_3D = 7.81473 and their FLOAT square roots (ORACLE_SPEC rules 25 / 28)
CHI2_MONO, CHI2_STEREO = float(np.float32(5.99146)), float(np.float32(7.81473))
SQRT_CHI2_MONO, SQRT_CHI2_STEREO = float(np.sqrt(np.float32(5.99146))), float(np.sqrt(np.float32(7.81473)))


def _quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rot_to_quat(R):
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr > 0:
        s = np.sqrt(tr + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0], q[1], q[2] = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.sqrt(q @ q)


def _oplus(R, t, u):
    w, v = u[:3], u[3:]
    th = np.sqrt(w @ w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    O2 = O @ O
    if th < 0.00001:
        E = np.eye(3) + O + O2
        V = E
    else:
        s, c = np.sin(th), np.cos(th)
        E = np.eye(3) + s / th * O + (1 - c) / (th * th) * O2
        V = np.eye(3) + (1 - c) / (th * th) * O + (th - s) / (th * th * th) * O2
    return E @ R, E @ t + V @ v


class _Graph:
    def __init__(self, n_pose, n_pt, fixed, mono, stereo, cam, bf, setup_type=0, equirect=None):
        self.n_pose, self.n_pt, self.cam, self.bf = n_pose, n_pt, cam, bf
        self.equirect = equirect   # None: perspective edges; (cols, rows): equirectangular_reproj_edge (mono edges only)
        self.setup_type = setup_type
        self.fixed = np.zeros(n_pose, np.uint8) if fixed is None else np.ascontiguousarray(fixed, np.uint8)
        self.free = np.flatnonzero(self.fixed == 0)
        self.slot = -np.ones(n_pose, np.int64)
        self.slot[self.free] = np.arange(len(self.free))
        self.set_edges(mono, stereo)

    def set_edges(self, mono, stereo):
        self.mono, self.stereo = mono, stereo
        self.pi = np.concatenate([mono["pose_idx"], stereo["pose_idx"]]).astype(np.int64)
        self.li = np.concatenate([mono["point_idx"], stereo["point_idx"]]).astype(np.int64)
        # all ordered pairs of edges that share a landmark and whose poses are free, upper block triangle
        order = np.argsort(self.li, kind="stable")
        li_s = self.li[order]
        starts = np.flatnonzero(np.r_[True, li_s[1:] != li_s[:-1]]) if len(li_s) else np.zeros(0, np.int64)
        counts = np.diff(np.r_[starts, len(li_s)]) if len(li_s) else np.zeros(0, np.int64)
        a, b = [], []
        for s, c in zip(starts, counts):
            idx = order[s:s + c]
            ii, jj = np.meshgrid(idx, idx, indexing="ij")
            a.append(ii.ravel())
            b.append(jj.ravel())
        a = np.concatenate(a) if a else np.zeros(0, np.int64)
        b = np.concatenate(b) if b else np.zeros(0, np.int64)
        sa, sb = self.slot[self.pi[a]], self.slot[self.pi[b]]
        keep = (sa >= 0) & (sb >= 0) & (sb >= sa)
        self.pa, self.pb, self.sa, self.sb = a[keep], b[keep], sa[keep], sb[keep]

    def linearize(self, T, X, robust):
        poses = np.zeros((self.n_pose, 7))
        for k, (R, t) in enumerate(T):
            poses[k, :3] = t
            poses[k, 3:] = _rot_to_quat(R)
        if self.equirect is not None:
            return ob.ba_linearize_equirect(poses, self.fixed, X, self.mono, self.equirect[0], self.equirect[1], SQRT_CHI2_MONO if robust else 0.0)
        out = ob.ba_linearize(poses, self.fixed, X, self.mono, self.cam,
                              (SQRT_CHI2_MONO if self.setup_type == 0 else SQRT_CHI2_STEREO) if robust else 0.0)
        if len(self.stereo):
            s = ob.ba_linearize_stereo(poses, self.fixed, X, self.stereo, self.cam, self.bf, SQRT_CHI2_STEREO if robust else 0.0)
            for k in ("Hpp", "bp", "Hll", "bl", "chi2"):
                out[k] = out[k] + s[k]
            out["Hpl"] = np.concatenate([out["Hpl"], s["Hpl"]])
        return out

    def edge_chi2(self, T, X):
        if self.equirect is not None:   # depth_is_positive() is always true for the equirectangular model
            poses = np.zeros((self.n_pose, 7))
            for k, (R, t) in enumerate(T):
                poses[k, :3] = t
                poses[k, 3:] = _rot_to_quat(R)
            chi = ob.ba_edge_chi2_equirect(poses, X, self.mono, self.equirect[0], self.equirect[1])
            return chi, np.ones(len(chi), bool)
        R = np.stack([r for r, _ in T])
        t = np.stack([tt for _, tt in T])
        fx, fy, cx, cy = self.cam
        chi, depth = [], []
        for e, stereo in ((self.mono, False), (self.stereo, True)):
            if not len(e):
                continue
            p = np.einsum("eab,eb->ea", R[e["pose_idx"]], X[e["point_idx"]]) + t[e["pose_idx"]]
            invz = 1.0 / p[:, 2]
            u = fx * p[:, 0] * invz + cx
            ss = (e["obs_x"] - u) ** 2 + (e["obs_y"] - (fy * p[:, 1] * invz + cy)) ** 2
            if stereo:
                ss = ss + (e["obs_x_right"] - (u - self.bf * invz)) ** 2
            chi.append(e["inv_sigma_sq"] * ss)
            depth.append(p[:, 2] > 0)
        if not chi:
            return np.zeros(0), np.zeros(0, bool)
        return np.concatenate(chi), np.concatenate(depth)

    def solve(self, B, lam):
        nf = len(self.free)
        Hll = B["Hll"] + lam * np.eye(3)[None]
        det = np.linalg.det(Hll)
        if not np.all(np.isfinite(det)) or np.any(det == 0):
            return None
        Hinv = np.linalg.inv(Hll)
        S = np.zeros((nf, nf, 6, 6))
        g = np.zeros((nf, 6))
        S[np.arange(nf), np.arange(nf)] = B["Hpp"][self.free] + lam * np.eye(6)[None]
        g[:] = B["bp"][self.free]
        W = B["Hpl"]
        Y = np.einsum("eab,ebc->eac", W, Hinv[self.li])
        fe = self.slot[self.pi] >= 0
        yb = np.einsum("eab,eb->ea", Y[fe], B["bl"][self.li[fe]])
        sl = self.slot[self.pi[fe]]
        g -= np.stack([np.bincount(sl, weights=yb[:, c], minlength=nf) for c in range(6)], 1)
        blk = np.einsum("pab,pcb->pac", Y[self.pa], W[self.pb]).reshape(-1, 36)
        lin = self.sa * nf + self.sb
        acc = np.stack([np.bincount(lin, weights=blk[:, c], minlength=nf * nf) for c in range(36)], 1)
        S -= acc.reshape(nf, nf, 6, 6)
        iu = np.triu_indices(nf, 1)
        S[iu[1], iu[0]] = np.transpose(S[iu[0], iu[1]], (0, 2, 1))
        A = S.transpose(0, 2, 1, 3).reshape(6 * nf, 6 * nf)
        try:
            Lc = np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            return None
        y = np.linalg.solve(Lc, g.reshape(-1))
        dxf = np.linalg.solve(Lc.T, y)
        dxp = np.zeros((self.n_pose, 6))
        dxp[self.free] = dxf.reshape(-1, 6)
        rhs = B["bl"].copy()
        wd = np.einsum("eab,ea->eb", W[fe], dxp[self.pi[fe]])
        rhs -= np.stack([np.bincount(self.li[fe], weights=wd[:, c], minlength=self.n_pt) for c in range(3)], 1)
        dxl = np.einsum("nab,nb->na", Hinv, rhs)
        return dxp, dxl

    def run_round(self, T, X, iters, robust, stop=None):
        cur = self.linearize(T, X, robust)
        chi = cur["chi2"][1]
        chi_start = chi
        n_iter = 0
        # (Terr, Xerr): the state the ACTIVE edges' errors were last computed at. g2o's edge->chi2() reads the stored error: after a rejected
        # trial the estimate is popped back but the errors stay those of the trial state until the next computeActiveErrors()
        Terr, Xerr = T, X
        if iters <= 0 or len(self.pi) == 0:
            return T, X, chi_start, chi, 0, Terr, Xerr
        has_edge = np.bincount(self.li, minlength=self.n_pt) > 0
        md = 0.0
        if len(self.free):
            md = max(md, np.abs(np.einsum("kii->ki", cur["Hpp"][self.free])).max())
        if has_edge.any():
            md = max(md, np.abs(np.einsum("kii->ki", cur["Hll"][has_edge])).max())
        lam, ni = 1e-5 * md, 2.0
        for _ in range(iters):
            if stop is not None and stop[0]:
                break
            n_iter += 1
            rho, qmax = 0.0, 0
            Terr, Xerr = T, X   # solve() starts with computeActiveErrors() at the current estimate
            while True:
                sol = self.solve(cur, lam)
                temp, scale = np.finfo(np.float64).max, 1e-3
                if sol is not None:
                    dxp, dxl = sol
                    Tn = [(_oplus(R, t, dxp[k]) if self.slot[k] >= 0 else (R, t)) for k, (R, t) in enumerate(T)]
                    Xn = X + dxl
                    trial = self.linearize(Tn, Xn, robust)
                    Terr, Xerr = Tn, Xn   # computeActiveErrors() ran on the trial state, accepted or not
                    temp = trial["chi2"][1]
                    scale = (dxp[self.free] * (lam * dxp[self.free] + cur["bp"][self.free])).sum() + (dxl * (lam * dxl + cur["bl"])).sum() + 1e-3
                rho = (chi - temp) / scale
                if sol is not None and rho > 0 and np.isfinite(temp):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha)
                    ni = 2.0
                    chi = temp
                    T, X, cur = Tn, Xn, trial
                else:
                    lam *= ni
                    ni *= 2
                    if not np.isfinite(lam):
                        break
                qmax += 1
                if not (rho < 0 and qmax < 10 and not (stop is not None and stop[0])):
                    break
            if qmax == 10 or rho == 0 or not np.isfinite(lam):
                break
        return T, X, chi_start, chi, n_iter, Terr, Xerr


def local_ba_optimize_equirect(poses, pose_fixed, points, mono, cols, rows, num_first_iter=5, num_second_iter=10):
    """The same two rounds over equirectangular_reproj_edge (every edge monocular, Monocular rig)."""
    return local_ba_optimize(poses, pose_fixed, points, mono, (float(cols), float(rows), 0.0, 0.0), None, 0.0, num_first_iter, num_second_iter,
                             setup_type=0, equirect=(int(cols), int(rows)))


def local_ba_optimize(poses, pose_fixed, points, mono, cam, stereo=None, focal_x_baseline=0.0, num_first_iter=5, num_second_iter=10,
                      setup_type=None, equirect=None):
    if setup_type is None:
        setup_type = 1 if focal_x_baseline != 0.0 else 0
    poses = np.array(poses, np.float64).reshape(-1, 7)
    X = np.array(points, np.float64).reshape(-1, 3).copy()
    mono = np.ascontiguousarray(mono if mono is not None else np.zeros(0, ob.BA_EDGE_DTYPE), ob.BA_EDGE_DTYPE)
    stereo = np.ascontiguousarray(stereo if stereo is not None else np.zeros(0, ob.BA_EDGE_STEREO_DTYPE), ob.BA_EDGE_STEREO_DTYPE)
    n_pose, n_pt, nm = len(poses), len(X), len(mono)
    T = [(_quat_to_rot(p[3:]), p[:3].copy()) for p in poses]
    G = _Graph(n_pose, n_pt, pose_fixed, mono, stereo, tuple(cam), focal_x_baseline, setup_type, equirect)
    info = np.zeros(6)
    T, X, info[0], info[1], info[4], Terr, Xerr = G.run_round(T, X, num_first_iter, True)
    # edge->chi2() is the error stored by the last computeActiveErrors() (the last TRIAL state when the round ended on a rejected step);
    # edge->depth_is_positive() is evaluated from the vertices' current (accepted) estimates
    chi_r1 = G.edge_chi2(Terr, Xerr)[0]
    depth = G.edge_chi2(T, X)[1]
    gate = np.concatenate([np.full(nm, CHI2_MONO), np.full(len(stereo), CHI2_STEREO)])
    out_r1 = (gate < chi_r1) | ~depth
    G.set_edges(mono[~out_r1[:nm]], stereo[~out_r1[nm:]])
    T, X, info[2], info[3], info[5], Terr, Xerr = G.run_round(T, X, num_second_iter, False)
    G.set_edges(mono, stereo)
    chi = G.edge_chi2(Terr, Xerr)[0]
    depth = G.edge_chi2(T, X)[1]
    c = np.where(out_r1, chi_r1, chi)
    outlier = (gate < c) | ~depth
    P = poses.copy()
    for k in G.free:
        P[k, :3] = T[k][1]
        P[k, 3:] = _rot_to_quat(T[k][0])
    return dict(poses=P, points=X, mono_outlier=outlier[:nm], stereo_outlier=outlier[nm:], info=info)
