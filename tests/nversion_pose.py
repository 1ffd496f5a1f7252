"""A SECOND, independent restatement of optimize::pose_optimizer::optimize (perspective mono / stereo edges) in plain numpy, written from
oracle/ORACLE_SPEC.md rules 15 and 25 and the published g2o algorithm (OptimizationAlgorithmLevenberg, RobustKernelHuber, SE3Quat::exp) --
not from oracle/ovo_pose.cc: whole-array edge arithmetic, numpy's LAPACK solve instead of a hand-written Cholesky. tests/test_nversion.py
compares it with the C oracle. The normal equations are summed in another order (pairwise numpy sums), so agreement is to the stated
tolerance of the optimiser (2e-8; rule 25, "How far the result is defined"), with identical inlier flags away from the chi2 gates.
Test infrastructure only."""
import numpy as np

CHI2_2D = float(np.float32(5.99146))
CHI2_3D = float(np.float32(7.81473))
SQRT_CHI2_2D = float(np.sqrt(np.float32(5.99146)))
SQRT_CHI2_3D = float(np.sqrt(np.float32(7.81473)))


def _se3_exp(u):
    w, v = u[:3], u[3:]
    th = np.sqrt(w @ w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    O2 = O @ O
    if th < 0.00001:
        R = np.eye(3) + O + O2
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / (th * th) * O2
        V = np.eye(3) + (1 - np.cos(th)) / (th * th) * O + (th - np.sin(th)) / (th ** 3) * O2
    return R, V @ v


def _edges(R, t, obs, cam, bf):
    """Residuals (n, 3), Jacobians (n, 3, 6) w.r.t. (omega, upsilon), stereo mask; the third row is zero for monocular observations."""
    fx, fy, cx, cy = cam
    p = obs["pos_w"] @ R.T + t
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    iz = 1.0 / z
    iz2 = iz * iz
    st = obs["is_stereo"] != 0
    u = fx * x * iz + cx
    e = np.stack([obs["obs_x"] - u, obs["obs_y"] - (fy * y * iz + cy), np.where(st, obs["obs_x_right"] - (u - bf * iz), 0.0)], 1)
    J = np.zeros((len(obs), 3, 6))
    J[:, 0] = np.stack([x * y * iz2 * fx, -(1 + x * x * iz2) * fx, y * iz * fx, -iz * fx, np.zeros_like(x), x * iz2 * fx], 1)
    J[:, 1] = np.stack([(1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, np.zeros_like(x), -iz * fy, y * iz2 * fy], 1)
    J2 = J[:, 0].copy()
    J2[:, 0] -= bf * y * iz2
    J2[:, 1] += bf * x * iz2
    J2[:, 5] -= bf * iz2
    J[:, 2] = np.where(st[:, None], J2, 0.0)
    return e, J, st


def _edges_equirect(R, t, obs, cam, bf):
    """Rule 26: u = cols (1/2 + atan2(x, z) / 2 pi), v = rows (1/2 + asin(y / |p|) / pi), e = z - (u, v), no wrap-around at the seam;
    J = -d(u, v) / d(omega, upsilon) with dp = (-[p]x | I). cam = (cols, rows, -, -)."""
    cols, rows = cam[0], cam[1]
    p = obs["pos_w"] @ R.T + t
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    L = np.sqrt((p * p).sum(1))
    rxz2 = x * x + z * z
    e = np.stack([obs["obs_x"] - cols * (0.5 + np.arctan2(x, z) / (2 * np.pi)), obs["obs_y"] - rows * (0.5 + np.arcsin(y / L) / np.pi),
                  np.zeros_like(x)], 1)
    dp = np.zeros((len(obs), 3, 6))   # d p / d (omega, upsilon): columns 0..2 = -[p]x, 3..5 = I
    dp[:, 0, 1], dp[:, 0, 2] = z, -y
    dp[:, 1, 0], dp[:, 1, 2] = -z, x
    dp[:, 2, 0], dp[:, 2, 1] = y, -x
    dp[:, 0, 3] = dp[:, 1, 4] = dp[:, 2, 5] = 1.0
    dL = (p[:, :, None] * dp).sum(1) / L[:, None]
    du = (cols / (2 * np.pi)) * (z[:, None] * dp[:, 0] - x[:, None] * dp[:, 2]) / rxz2[:, None]
    dv = (rows / np.pi) * (L[:, None] * dp[:, 1] - y[:, None] * dL) / (L * np.sqrt(rxz2))[:, None]
    J = np.zeros((len(obs), 3, 6))
    J[:, 0], J[:, 1] = -du, -dv
    return e, J, np.zeros(len(obs), bool)


def _chi2(R, t, obs, cam, bf, edges=None):
    e, _, st = (edges or _edges)(R, t, obs, cam, bf)
    return obs["inv_sigma_sq"] * (e * e).sum(1), st


def _robust_sum(c2, delta):
    if delta <= 0:
        return c2.sum()
    return np.where(c2 > delta * delta, 2 * np.sqrt(c2) * delta - delta * delta, c2).sum()


def pose_optimize_equirect(T0, obs, cols, rows):
    """Equirectangular frames (monocular rig: Huber sqrtf(5.99146f), gate 5.99146f)."""
    return pose_optimize(T0, obs, (float(cols), float(rows), 0.0, 0.0), 0.0, 0, _edges_equirect)


def pose_optimize(T0, obs, cam, bf=0.0, setup_type=None, edges=None, reset_each_round=False):
    """Returns (pose 3x4, outlier flags, num_valid) as the C oracle's ovo_pose_optimize."""
    edges = edges or _edges
    if setup_type is None:
        setup_type = 1 if bf != 0.0 else 0
    huber = SQRT_CHI2_2D if setup_type == 0 else SQRT_CHI2_3D
    R, t = np.array(T0[:, :3], float), np.array(T0[:, 3], float)
    n = len(obs)
    out = np.zeros(n, bool)
    if n < 5:
        return np.concatenate([R, t[:, None]], 1), out, 0
    active = np.ones(n, bool)
    num_bad = 0
    for rnd in range(4):
        delta = huber if rnd < 3 else 0.0   # Huber in rounds 0 .. 2
        if reset_each_round:                # rule 25 (iv)'s variant: every round starts from the input pose again
            R, t = np.array(T0[:, :3], float), np.array(T0[:, 3], float)
        o = obs[active]
        lam, ni = 0.0, 2.0
        Rn, tn = R, t
        err_at_trial = False
        for it in range(10):
            err_at_trial = False
            e, J, _ = edges(R, t, o, cam, bf)
            c2 = o["inv_sigma_sq"] * (e * e).sum(1)
            rho1 = np.ones(len(o))
            if delta > 0:
                big = c2 > delta * delta
                rho1 = np.where(big, delta / np.sqrt(np.where(big, c2, 1.0)), 1.0)
            W = rho1 * o["inv_sigma_sq"]
            H = np.einsum("n,nra,nrb->ab", W, J, J)
            b = -np.einsum("n,nra,nr->a", W, J, e)
            chi = _robust_sum(c2, delta)
            if it == 0:
                lam, ni = 1e-5 * np.abs(np.diag(H)).max(), 2.0
            rho, qmax = 0.0, 0
            while True:
                A = H + lam * np.eye(6)
                ok = True
                try:
                    np.linalg.cholesky(A)   # g2o: the step exists iff the damped system is positive definite
                    dx = np.linalg.solve(A, b)
                except np.linalg.LinAlgError:
                    ok = False
                temp, scale = np.finfo(float).max, 1e-3
                if ok:
                    E, et = _se3_exp(dx)
                    Rn, tn = E @ R, E @ t + et
                    temp = _robust_sum(_chi2(Rn, tn, o, cam, bf, edges)[0], delta)
                    scale = dx @ (lam * dx + b) + 1e-3
                    err_at_trial = True
                rho = (chi - temp) / scale
                if rho > 0 and np.isfinite(temp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    chi = temp
                    R, t = Rn, tn
                else:
                    lam *= ni
                    ni *= 2
                qmax += 1
                if not (rho < 0 and qmax < 10):
                    break
            if qmax == 10 or rho == 0:
                break
        # re-classification: previous outliers at the estimate, inliers where their errors were last computed (the last trial state)
        c_est, st = _chi2(R, t, obs, cam, bf, edges)
        c_err = _chi2(Rn, tn, obs, cam, bf, edges)[0] if err_at_trial else c_est
        c2 = np.where(active, c_err, c_est)
        out = np.where(st, CHI2_3D, CHI2_2D) < c2
        active = ~out
        num_bad = int(out.sum())
        if n - num_bad < 5:
            break
    return np.concatenate([R, t[:, None]], 1), out, n - num_bad


# ---- rule 15: one Levenberg-Marquardt linearisation of local BA (the blocks ovo_ba_linearize* return) --------------------------------------
def _quat_rot(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)


def ba_linearize(poses, pose_fixed, points, edges, cam, huber_delta, bf=None, equirect=False, rot=None):
    """Hpp (n_pose, 6, 6), bp, Hll (n_pt, 3, 3), bl, Hpl (n_edge, 6, 3), chi2 (plain, robust) of one set of edges: e = z - pi(R X + t),
    J_point = -(d pi / d p) R, J_pose = -(d pi / d p) (-[p]x | I), Huber on chi2 = w e.e with the second-derivative term dropped,
    b = -J^T W e; fixed keyframes get no pose blocks (their edges still feed the landmark blocks). bf: stereo edges (third residual
    u_right = u - bf / z); equirect: cam = (cols, rows, -, -), the projection of rule 26."""
    poses = np.asarray(poses, float).reshape(-1, 7)
    points = np.asarray(points, float).reshape(-1, 3)
    R_all = _quat_rot(poses[:, 3:]) if rot is None else np.asarray(rot, float)      # rot: rotation matrices given directly (local BA's state)
    k, j = edges["pose_idx"], edges["point_idx"]
    R = R_all[k]
    p = np.einsum("nab,nb->na", R, points[j]) + poses[k, :3]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    n = len(edges)
    rows_e = 3 if bf is not None else 2
    dpi = np.zeros((n, rows_e, 3))   # d pi / d p
    if equirect:
        cols, rws = cam[0], cam[1]
        L = np.sqrt((p * p).sum(1))
        rxz2 = x * x + z * z
        e = np.stack([edges["obs_x"] - cols * (0.5 + np.arctan2(x, z) / (2 * np.pi)), edges["obs_y"] - rws * (0.5 + np.arcsin(y / L) / np.pi)], 1)
        dpi[:, 0, 0], dpi[:, 0, 2] = (cols / (2 * np.pi)) * z / rxz2, -(cols / (2 * np.pi)) * x / rxz2
        s = (rws / np.pi) / (L * np.sqrt(rxz2))
        # d asin(y / L) = (L dy - y dL) / (L sqrt(x^2 + z^2)), dL = p.dp / L
        dpi[:, 1] = s[:, None] * (np.stack([np.zeros(n), L, np.zeros(n)], 1) - y[:, None] * p / L[:, None])
    else:
        fx, fy, cx, cy = cam
        iz = 1.0 / z
        u = fx * x * iz + cx
        cols_e = [edges["obs_x"] - u, edges["obs_y"] - (fy * y * iz + cy)]
        dpi[:, 0, 0], dpi[:, 0, 2] = fx * iz, -fx * x * iz * iz
        dpi[:, 1, 1], dpi[:, 1, 2] = fy * iz, -fy * y * iz * iz
        if bf is not None:
            cols_e.append(edges["obs_x_right"] - (u - bf * iz))
            dpi[:, 2, 0], dpi[:, 2, 2] = fx * iz, -fx * x * iz * iz + bf * iz * iz
        e = np.stack(cols_e, 1)
    dp = np.zeros((n, 3, 6))
    dp[:, 0, 1], dp[:, 0, 2] = z, -y
    dp[:, 1, 0], dp[:, 1, 2] = -z, x
    dp[:, 2, 0], dp[:, 2, 1] = y, -x
    dp[:, 0, 3] = dp[:, 1, 4] = dp[:, 2, 5] = 1.0
    Jl = -np.einsum("nrc,ncd->nrd", dpi, R)
    Jp = -np.einsum("nrc,ncd->nrd", dpi, dp)
    w = edges["inv_sigma_sq"]
    c2 = w * (e * e).sum(1)
    rho0, rho1 = c2.copy(), np.ones(n)
    if huber_delta > 0:
        big = c2 > huber_delta * huber_delta
        sq = np.sqrt(np.where(big, c2, 1.0))
        rho0 = np.where(big, 2 * sq * huber_delta - huber_delta * huber_delta, c2)
        rho1 = np.where(big, huber_delta / sq, 1.0)
    W = rho1 * w
    free = np.ones(len(poses), bool) if pose_fixed is None else ~np.asarray(pose_fixed).astype(bool)
    fe = free[k]
    out = dict(Hpp=np.zeros((len(poses), 6, 6)), bp=np.zeros((len(poses), 6)), Hll=np.zeros((len(points), 3, 3)), bl=np.zeros((len(points), 3)))
    np.add.at(out["Hll"], j, W[:, None, None] * np.einsum("nra,nrb->nab", Jl, Jl))
    np.add.at(out["bl"], j, -W[:, None] * np.einsum("nra,nr->na", Jl, e))
    np.add.at(out["Hpp"], k[fe], (W[:, None, None] * np.einsum("nra,nrb->nab", Jp, Jp))[fe])
    np.add.at(out["bp"], k[fe], (-W[:, None] * np.einsum("nra,nr->na", Jp, e))[fe])
    out["Hpl"] = np.where(fe[:, None, None], W[:, None, None] * np.einsum("nra,nrb->nab", Jp, Jl), 0.0)
    out["chi2"] = np.array([c2.sum(), rho0.sum()])
    return out


# ---- rule 28: optimize::local_bundle_adjuster::optimize behind the graph build ---------------------------------------------------------------
def _edge_chi2_depth(R, t, X, edges, cam, bf=None, equirect=False):
    """(chi2, depth_is_positive) of every edge at one state."""
    p = np.einsum("nab,nb->na", R[edges["pose_idx"]], X[edges["point_idx"]]) + t[edges["pose_idx"]]
    if equirect:
        cols, rows = cam[0], cam[1]
        L = np.sqrt((p * p).sum(1))
        ss = (edges["obs_x"] - cols * (0.5 + np.arctan2(p[:, 0], p[:, 2]) / (2 * np.pi))) ** 2 + \
             (edges["obs_y"] - rows * (0.5 + np.arcsin(p[:, 1] / L) / np.pi)) ** 2
        return edges["inv_sigma_sq"] * ss, np.ones(len(p), bool)
    fx, fy, cx, cy = cam
    u = fx * p[:, 0] / p[:, 2] + cx
    ss = (edges["obs_x"] - u) ** 2 + (edges["obs_y"] - (fy * p[:, 1] / p[:, 2] + cy)) ** 2
    if bf is not None:
        ss = ss + (edges["obs_x_right"] - (u - bf / p[:, 2])) ** 2
    return edges["inv_sigma_sq"] * ss, p[:, 2] > 0


def local_ba_optimize(poses, pose_fixed, points, mono, cam, stereo=None, bf=0.0, num_first_iter=5, num_second_iter=10, setup_type=None, equirect=False):
    """The two rounds of rule 28 WITHOUT the elimination of the landmarks: every trial solves the whole damped system over (free keyframes,
    landmarks with an active edge) with LAPACK -- mathematically the same step as the Schur form the oracle and the library take.
    Returns dict(R, t, points, mono_outlier, stereo_outlier, info = (chi2 at the start / end of round 1, of round 2, iterations of each))."""
    poses = np.asarray(poses, float).reshape(-1, 7)
    R, t, X = _quat_rot(poses[:, 3:]), poses[:, :3].copy(), np.array(points, float).reshape(-1, 3)
    n_pose, n_pt = len(poses), len(X)
    free = np.ones(n_pose, bool) if pose_fixed is None else ~np.asarray(pose_fixed).astype(bool)
    slot = np.cumsum(free) - 1
    nf = int(free.sum())
    if setup_type is None:
        setup_type = 1 if bf != 0.0 else 0
    huber = SQRT_CHI2_2D if setup_type == 0 else SQRT_CHI2_3D
    stereo = np.zeros(0, mono.dtype) if stereo is None else stereo
    sets = [(mono, None)] + ([(stereo, bf)] if len(stereo) else [])

    def linearize(R, t, X, active, delta):
        B = None
        for (edges, b), act in zip(sets, active):
            o = ba_linearize(np.concatenate([t, np.zeros((n_pose, 4))], 1), ~free, X, edges[act], cam, delta, bf=b, equirect=equirect, rot=R)
            o["edges"] = [edges[act]]
            o["Hpl"] = [o["Hpl"]]
            if B is None:
                B = o
            else:
                for key in ("Hpp", "bp", "Hll", "bl", "chi2"):
                    B[key] = B[key] + o[key]
                B["Hpl"] += o["Hpl"]
                B["edges"] += o["edges"]
        return B

    def solve(B, lam):
        k = np.concatenate([e["pose_idx"] for e in B["edges"]])
        j = np.concatenate([e["point_idx"] for e in B["edges"]])
        W = np.concatenate(B["Hpl"])
        used = np.zeros(n_pt, bool)
        used[j] = True
        col = np.cumsum(used) - 1
        na = int(used.sum())
        n = 6 * nf + 3 * na
        A, g = np.zeros((n, n)), np.zeros(n)
        for q in np.nonzero(free)[0]:
            s = 6 * slot[q]
            A[s:s + 6, s:s + 6] = B["Hpp"][q]
            g[s:s + 6] = B["bp"][q]
        for a in np.nonzero(used)[0]:
            s = 6 * nf + 3 * col[a]
            A[s:s + 3, s:s + 3] = B["Hll"][a]
            g[s:s + 3] = B["bl"][a]
        for e in np.nonzero(free[k])[0]:
            r, c = 6 * slot[k[e]], 6 * nf + 3 * col[j[e]]
            A[r:r + 6, c:c + 3] += W[e]
            A[c:c + 3, r:r + 6] += W[e].T
        A[np.arange(n), np.arange(n)] += lam
        try:
            np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            return None
        dx = np.linalg.solve(A, g)
        dxp, dxl = np.zeros((n_pose, 6)), np.zeros((n_pt, 3))
        dxp[free] = dx[:6 * nf].reshape(-1, 6)
        dxl[used] = dx[6 * nf:].reshape(-1, 3)
        return dxp, dxl, float(dx @ (lam * dx + g))

    def run_round(R, t, X, active, iters, delta):
        cur = linearize(R, t, X, active, delta)
        chi = chi_start = cur["chi2"][1]
        err_state = (R, t, X)
        n_edges = sum(int(a.sum()) for a in active)
        if iters <= 0 or n_edges == 0:
            return R, t, X, chi_start, chi, 0, err_state
        seen = np.zeros(n_pt, bool)
        for (edges, _), act in zip(sets, active):
            seen[edges["point_idx"][act]] = True
        diag = [np.abs(np.einsum("kii->ki", cur["Hpp"][free])).max()] if nf else []
        if seen.any():
            diag.append(np.abs(np.einsum("kii->ki", cur["Hll"][seen])).max())
        lam, ni, n_iter = 1e-5 * max(diag), 2.0, 0
        for _ in range(iters):
            n_iter += 1
            rho, qmax = 0.0, 0
            err_state = (R, t, X)
            while True:
                sol = solve(cur, lam)
                temp, scale = np.finfo(float).max, 1e-3
                if sol is not None:
                    dxp, dxl, gain = sol
                    Rn, tn = R.copy(), t.copy()
                    for q in np.nonzero(free)[0]:
                        E, et = _se3_exp(dxp[q])
                        Rn[q], tn[q] = E @ R[q], E @ t[q] + et
                    Xn = X + dxl
                    trial = linearize(Rn, tn, Xn, active, delta)
                    err_state = (Rn, tn, Xn)
                    temp, scale = trial["chi2"][1], gain + 1e-3
                rho = (chi - temp) / scale
                if sol is not None and rho > 0 and np.isfinite(temp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni, chi = 2.0, temp
                    R, t, X, cur = Rn, tn, Xn, trial
                else:
                    lam *= ni
                    ni *= 2
                    if not np.isfinite(lam):
                        break
                qmax += 1
                if not (rho < 0 and qmax < 10):
                    break
            if qmax == 10 or rho == 0 or not np.isfinite(lam):
                break
        return R, t, X, chi_start, chi, n_iter, err_state

    def judge(err_state, R, t, X):
        chi = [_edge_chi2_depth(*err_state, edges, cam, b, equirect)[0] for edges, b in sets]
        depth = [_edge_chi2_depth(R, t, X, edges, cam, b, equirect)[1] for edges, b in sets]
        return chi, depth

    gates = [CHI2_2D, CHI2_3D]
    info = np.zeros(6)
    everything = [np.ones(len(edges), bool) for edges, _ in sets]
    R, t, X, info[0], info[1], info[4], err = run_round(R, t, X, everything, num_first_iter, huber)
    chi_r1, depth = judge(err, R, t, X)
    out_r1 = [(gates[i] < chi_r1[i]) | ~depth[i] for i in range(len(sets))]
    R, t, X, info[2], info[3], info[5], err = run_round(R, t, X, [~o for o in out_r1], num_second_iter, 0.0)
    chi_r2, depth = judge(err, R, t, X)
    flags = [(gates[i] < np.where(out_r1[i], chi_r1[i], chi_r2[i])) | ~depth[i] for i in range(len(sets))]
    return dict(R=R, t=t, points=X, mono_outlier=flags[0], stereo_outlier=flags[1] if len(sets) > 1 else np.zeros(0, bool), info=info)
