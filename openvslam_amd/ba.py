"""Host side of the local-BA linearisation (config 5 of BASELINE.json): mirrors what g2o does between
optimize::local_bundle_adjuster::optimize's graph construction and its linear solve (expected:
src/openvslam/optimize/local_bundle_adjuster.cc), with the residual/Jacobian/block accumulation on the GPU.

Multi-GPU shape: edges are partitioned BY KEYFRAME across ranks (one process per GPU). Hpp/bp are then
complete on the owning rank (summed across ranks only to replicate them: 50 x 42 doubles), Hpl is per edge (stays local),
and the landmark blocks Hll|bl -- landmarks are seen from several keyframes -- need ONE sum across ranks: a single
all-reduce of a dense n_pt x 12 fp64 buffer (1.92 MB at 20 000 landmarks; RCCL over xGMI on GPUs, gloo in the CPU tests).
The sparse Schur solve stays on the host (`schur_solve`)."""
import ctypes as C

import numpy as np

from . import _lib

EDGE_DTYPE = np.dtype([("pose_idx", "<i4"), ("point_idx", "<i4"), ("obs_x", "<f8"), ("obs_y", "<f8"), ("inv_sigma_sq", "<f8")])
assert EDGE_DTYPE.itemsize == 32


class BaCam(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def linearize(poses, pose_fixed, points, edges, cam, huber_delta, device=0):
    """Single-GPU, host buffers in / host buffers out (ovs_ba_linearize). cam = (fx, fy, cx, cy)."""
    L = _lib.lib()
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
               Hpl=np.zeros((max(n_edge, 1), 6, 3)), chi2=np.zeros(2))
    c = BaCam(*cam)
    _lib.check(L.ovs_ba_linearize(device, _p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, C.byref(c), float(huber_delta),
                                  _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]), _p(out["bl"]), _p(out["Hpl"]), _p(out["chi2"])),
               "ovs_ba_linearize")
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


def linearize_equirect(poses, pose_fixed, points, edges, cols, rows, huber_delta, device=0):
    """Equirectangular reprojection edges, host buffers in / out (ovs_ba_linearize_equirect)."""
    L = _lib.lib()
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
               Hpl=np.zeros((max(n_edge, 1), 6, 3)), chi2=np.zeros(2))
    _lib.check(L.ovs_ba_linearize_equirect(device, _p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, int(cols), int(rows),
                                           float(huber_delta), _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]), _p(out["bl"]), _p(out["Hpl"]),
                                           _p(out["chi2"])), "ovs_ba_linearize_equirect")
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


EDGE_STEREO_DTYPE = np.dtype([("pose_idx", "<i4"), ("point_idx", "<i4"), ("obs_x", "<f8"), ("obs_y", "<f8"), ("obs_x_right", "<f8"),
                              ("inv_sigma_sq", "<f8")])
assert EDGE_STEREO_DTYPE.itemsize == 40


def linearize_stereo(poses, pose_fixed, points, edges, cam, focal_x_baseline, huber_delta, device=0):
    """Stereo reprojection edges (3 residuals), host buffers in / out (ovs_ba_linearize_stereo). cam = (fx, fy, cx, cy)."""
    L = _lib.lib()
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, EDGE_STEREO_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
               Hpl=np.zeros((max(n_edge, 1), 6, 3)), chi2=np.zeros(2))
    c = BaCam(*cam)
    _lib.check(L.ovs_ba_linearize_stereo(device, _p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, C.byref(c),
                                         float(focal_x_baseline), float(huber_delta), _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]),
                                         _p(out["bl"]), _p(out["Hpl"]), _p(out["chi2"])), "ovs_ba_linearize_stereo")
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


POSE_OBS_DTYPE = np.dtype([("pos_w", "<f8", (3,)), ("obs_x", "<f8"), ("obs_y", "<f8"), ("obs_x_right", "<f8"), ("inv_sigma_sq", "<f8"),
                           ("is_stereo", "<i4"), ("pad", "<i4")])
assert POSE_OBS_DTYPE.itemsize == 64


def local_ba_set_solver(where):
    """ovs_local_ba_set_solver: "device" (default: csrc/ba_solve.hip) | "host" (the Cholesky of rounds 1-3), process-wide."""
    _lib.check(_lib.lib().ovs_local_ba_set_solver({"device": 0, "host": 1}[where]), "ovs_local_ba_set_solver")


def local_ba_get_solver():
    return ("device", "host")[_lib.lib().ovs_local_ba_get_solver()]


def dense_solve(S, rhs, device=0):
    """ovs_ba_dense_solve: the reduced camera system's device solver alone (S symmetric positive definite, n <= 1024)."""
    S = np.ascontiguousarray(S, np.float64)
    rhs = np.ascontiguousarray(rhs, np.float64)
    n = S.shape[0]
    assert S.shape == (n, n) and rhs.shape == (n,)
    x = np.empty(n, np.float64)
    _lib.check(_lib.lib().ovs_ba_dense_solve(device, S.ctypes.data, rhs.ctypes.data, n, x.ctypes.data), "ovs_ba_dense_solve")
    return x


def pose_set_variant(which, value):
    """ovs_pose_set_variant: "reset_each_round" (0 | 1), process-wide (oracle/ORACLE_SPEC.md rule 25 (iv))."""
    _lib.check(_lib.lib().ovs_pose_set_variant({"reset_each_round": 0}[which], int(value)), "ovs_pose_set_variant")


def pose_optimize(pose_cw, obs, cam, focal_x_baseline=0.0, device=0, setup_type=None):
    """optimize::pose_optimizer::optimize(frm) on the device (ovs_pose_optimize). pose_cw: 3x4 [R|t]; obs: POSE_OBS_DTYPE records.
    setup_type: camera::setup_type_t of the rig (0 Monocular, 1 Stereo, 2 RGBD; default: Stereo iff focal_x_baseline != 0) -- it selects
    the frame's one Huber delta.
    Returns (pose_cw 3x4, outlier_flags bool[n], num_valid)."""
    L = _lib.lib()
    o = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
    T = np.asarray(pose_cw, np.float64)
    pin = np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))
    pout = np.zeros(12)
    out = np.zeros(max(len(o), 1), np.uint8)
    nv = C.c_int32()
    c = BaCam(*cam)
    if setup_type is None:
        setup_type = 1 if focal_x_baseline != 0.0 else 0
    _lib.check(L.ovs_pose_optimize(device, _p(pin), _p(o), len(o), C.byref(c), float(focal_x_baseline), int(setup_type), _p(pout), _p(out),
                                   C.byref(nv)),
               "ovs_pose_optimize")
    return np.concatenate([pout[:9].reshape(3, 3), pout[9:, None]], 1), out[:len(o)].astype(bool), nv.value


def pose_optimize_equirect(pose_cw, obs, cols, rows, device=0):
    """optimize::pose_optimizer::optimize(frm) for an equirectangular frame (ovs_pose_optimize_equirect: equirectangular_pose_opt_edge, every
    edge monocular, Monocular rig). obs: POSE_OBS_DTYPE records (is_stereo / obs_x_right ignored); cols / rows = camera->cols_ / rows_.
    Returns (pose_cw 3x4, outlier_flags bool[n], num_valid)."""
    L = _lib.lib()
    o = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
    T = np.asarray(pose_cw, np.float64)
    pin = np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))
    pout = np.zeros(12)
    out = np.zeros(max(len(o), 1), np.uint8)
    nv = C.c_int32()
    _lib.check(L.ovs_pose_optimize_equirect(device, _p(pin), _p(o), len(o), int(cols), int(rows), _p(pout), _p(out), C.byref(nv)),
               "ovs_pose_optimize_equirect")
    return np.concatenate([pout[:9].reshape(3, 3), pout[9:, None]], 1), out[:len(o)].astype(bool), nv.value


def local_ba_optimize(poses, pose_fixed, points, edges, cam, stereo_edges=None, focal_x_baseline=0.0, num_first_iter=5, num_second_iter=10,
                      force_stop_flag=None, device=0, setup_type=None):
    """optimize::local_bundle_adjuster::optimize behind the graph build (ovs_local_ba_optimize): returns dict(poses, points,
    mono_outlier, stereo_outlier, info). force_stop_flag: a 1-element uint8 array another thread may set."""
    L = _lib.lib()
    P = np.array(poses, np.float64).reshape(-1, 7).copy()
    X = np.array(points, np.float64).reshape(-1, 3).copy()
    em = np.ascontiguousarray(edges if edges is not None else np.zeros(0, EDGE_DTYPE), EDGE_DTYPE)
    es = np.ascontiguousarray(stereo_edges if stereo_edges is not None else np.zeros(0, EDGE_STEREO_DTYPE), EDGE_STEREO_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    om, os_ = np.zeros(max(len(em), 1), np.uint8), np.zeros(max(len(es), 1), np.uint8)
    info = np.zeros(6)
    c = BaCam(*cam)
    if setup_type is None:   # camera::setup_type_t of the rig (selects the Huber delta): Stereo iff there is a baseline
        setup_type = 1 if focal_x_baseline != 0.0 else 0
    _lib.check(L.ovs_local_ba_optimize(device, _p(P), _p(fixed), len(P), _p(X), len(X), _p(em) if len(em) else None, len(em),
                                       _p(es) if len(es) else None, len(es), C.byref(c), float(focal_x_baseline), int(setup_type), int(num_first_iter),
                                       int(num_second_iter), _p(force_stop_flag), _p(om), _p(os_), _p(info)), "ovs_local_ba_optimize")
    return dict(poses=P, points=X, mono_outlier=om[:len(em)].astype(bool), stereo_outlier=os_[:len(es)].astype(bool), info=info)


def local_ba_optimize_equirect(poses, pose_fixed, points, edges, cols, rows, num_first_iter=5, num_second_iter=10, force_stop_flag=None, device=0):
    """local_bundle_adjuster::optimize for an equirectangular local map (ovs_local_ba_optimize_equirect: equirectangular_reproj_edge, every
    edge monocular). Returns dict(poses, points, mono_outlier, info)."""
    L = _lib.lib()
    P = np.array(poses, np.float64).reshape(-1, 7).copy()
    X = np.array(points, np.float64).reshape(-1, 3).copy()
    em = np.ascontiguousarray(edges if edges is not None else np.zeros(0, EDGE_DTYPE), EDGE_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    om = np.zeros(max(len(em), 1), np.uint8)
    info = np.zeros(6)
    _lib.check(L.ovs_local_ba_optimize_equirect(device, _p(P), _p(fixed), len(P), _p(X), len(X), _p(em) if len(em) else None, len(em), int(cols),
                                                int(rows), int(num_first_iter), int(num_second_iter), _p(force_stop_flag), _p(om), _p(info)),
               "ovs_local_ba_optimize_equirect")
    return dict(poses=P, points=X, mono_outlier=om[:len(em)].astype(bool), info=info)


def shard_edges_by_keyframe(edges, n_pose, rank, world):
    """Edges of keyframes [rank*ceil(n_pose/world), ...): contiguous keyframe blocks, 2000 edges each in config 5."""
    per = -(-n_pose // world)
    sel = (edges["pose_idx"] // per) == rank
    return np.ascontiguousarray(edges[sel])


class graph:
    """ovs_ba_graph: a local-BA edge set indexed once (by landmark and by keyframe) and kept in HBM; linearize() is atomics-free and
    bit-reproducible. edges / stereo_edges: EDGE_DTYPE / EDGE_STEREO_DTYPE records (host). For a multi-rank shard pass the shard's edges."""

    def __init__(self, n_pose, pose_fixed, n_pt, edges, cam, stereo_edges=None, focal_x_baseline=0.0, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        em = np.ascontiguousarray(edges if edges is not None else np.zeros(0, EDGE_DTYPE), EDGE_DTYPE)
        es = np.ascontiguousarray(stereo_edges if stereo_edges is not None else np.zeros(0, EDGE_STEREO_DTYPE), EDGE_STEREO_DTYPE)
        fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
        self.n_pose, self.n_pt, self.n_edge = int(n_pose), int(n_pt), len(em) + len(es)
        h = C.c_void_p()
        c = BaCam(*cam)
        _lib.check(self._L.ovs_ba_graph_create(device, self.n_pose, _p(fixed), self.n_pt, _p(em) if len(em) else None, len(em),
                                               _p(es) if len(es) else None, len(es), C.byref(c), float(focal_x_baseline), C.byref(h)),
                   "ovs_ba_graph_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ovs_ba_graph_destroy(h)

    def linearize_dev(self, poses_t, points_t, huber_mono, huber_stereo=0.0, out=None, stream=None):
        """torch CUDA tensors in; returns dict of views into TWO device buffers: `pose` = Hpp | bp (complete on the shard that owns the
        keyframe) and `packed` = Hll | bl | chi2[3] (what a multi-rank run all-reduces, in ONE collective), plus Hpl."""
        import torch
        dev = poses_t.device
        n_pose, n_pt, n_edge = self.n_pose, self.n_pt, self.n_edge
        if out is None:
            out = dict(pose=torch.empty((42 * n_pose,), dtype=torch.float64, device=dev),
                       packed=torch.empty((12 * n_pt + 4,), dtype=torch.float64, device=dev),
                       hpl=torch.empty((max(n_edge, 1), 18), dtype=torch.float64, device=dev))
        bp_, bl_ = out["pose"].data_ptr(), out["packed"].data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self._L.ovs_ba_graph_linearize_dev(self._h, poses_t.data_ptr(), points_t.data_ptr(), float(huber_mono), float(huber_stereo),
                                                      bp_, bp_ + 8 * 36 * n_pose, bl_, bl_ + 8 * 9 * n_pt, out["hpl"].data_ptr(),
                                                      bl_ + 8 * 12 * n_pt, stream), "ovs_ba_graph_linearize_dev")
        return out

    @staticmethod
    def views(out, n_pose, n_pt, n_edge):
        return dict(Hpp=out["pose"][:36 * n_pose].view(n_pose, 6, 6), bp=out["pose"][36 * n_pose:].view(n_pose, 6),
                    Hll=out["packed"][:9 * n_pt].view(n_pt, 3, 3), bl=out["packed"][9 * n_pt:12 * n_pt].view(n_pt, 3),
                    chi2=out["packed"][12 * n_pt:12 * n_pt + 2], max_diag=out["packed"][12 * n_pt + 2:12 * n_pt + 3],
                    Hpl=out["hpl"][:n_edge].view(-1, 6, 3))


def hip_backend(poses_t, fixed_t, points_t, edges_t, cam, huber_delta):
    """Local shard on the current CUDA device through ovs_ba_linearize_dev (the round-1 atomics kernel; kept for comparison). Tensors are
    torch CUDA tensors; edges_t is a uint8 tensor viewing EDGE_DTYPE records. Returns (HppBp [n_pose*42], packed [12*n_pt + 4] =
    Hll | bl | chi2, Hpl [n_edge,18])."""
    import torch
    L = _lib.lib()
    n_pose, n_pt, n_edge = poses_t.shape[0], points_t.shape[0], edges_t.numel() // 32
    dev = poses_t.device
    hppbp = torch.empty((n_pose * 42,), dtype=torch.float64, device=dev)
    packed = torch.zeros((n_pt * 12 + 4,), dtype=torch.float64, device=dev)
    hpl = torch.empty((max(n_edge, 1), 18), dtype=torch.float64, device=dev)
    c = BaCam(*cam)
    base_pp, base_ll = hppbp.data_ptr(), packed.data_ptr()
    _lib.check(L.ovs_ba_linearize_dev(poses_t.data_ptr(), fixed_t.data_ptr() if fixed_t is not None else None, n_pose, points_t.data_ptr(),
                                      n_pt, edges_t.data_ptr() if n_edge else None, n_edge, C.byref(c), float(huber_delta),
                                      base_pp, base_pp + 8 * 36 * n_pose, base_ll, base_ll + 8 * 9 * n_pt, hpl.data_ptr(),
                                      base_ll + 8 * 12 * n_pt, torch.cuda.current_stream().cuda_stream), "ovs_ba_linearize_dev")
    return hppbp, packed, hpl[:n_edge]


class graph_backend:
    """Default shard backend of local_ba_linearizer: the atomics-free graph (built once per edge set, cached by the edge tensor)."""

    def __init__(self):
        self._key, self._g, self._out = None, None, None

    def __call__(self, poses_t, fixed_t, points_t, edges_t, cam, huber_delta):
        key = (edges_t.data_ptr(), edges_t.numel(), poses_t.shape[0], points_t.shape[0])
        if key != self._key:
            edges = np.ascontiguousarray(edges_t.cpu().numpy()).view(EDGE_DTYPE)
            fixed = fixed_t.cpu().numpy() if fixed_t is not None else None
            self._g = graph(poses_t.shape[0], fixed, points_t.shape[0], edges, cam, device=poses_t.device.index or 0)
            self._key, self._out = key, None
        self._out = self._g.linearize_dev(poses_t, points_t, huber_delta, 0.0, out=self._out)
        return self._out["pose"], self._out["packed"], self._out["hpl"][:self._g.n_edge]


class local_ba_linearizer:
    """One Levenberg-Marquardt linearisation of the local map, sharded over the ranks of a torch.distributed group.

    Every rank holds all poses and points (0.48 MB at config 5: replicated by the caller) and ITS shard of the edges (whole keyframes).
    `backend(poses, fixed, points, edges, cam, huber)` computes the shard's blocks and returns (HppBp, packed = Hll | bl | chi2, Hpl).
    THE exchange step of this path is ONE all-reduce of `packed` (landmarks are shared between keyframes on different ranks; 1.92 MB at
    20 k landmarks -- latency-dominated, so never three collectives where one does). Hpp | bp need no collective: a keyframe's block is
    complete on the rank that owns its edges and zero elsewhere; `gather_pose_blocks=True` sums them onto every rank (only the rank that
    solves the reduced camera system needs them -- pass dst to reduce instead of all-reduce)."""

    def __init__(self, cam, huber_delta, group=None, backend=None, gather_pose_blocks=True, dst=None):
        self.cam = tuple(float(v) for v in cam)
        self.huber_delta = float(huber_delta)
        self.group = group
        self.backend = backend if backend is not None else graph_backend()
        self.gather_pose_blocks = gather_pose_blocks
        self.dst = dst

    def linearize(self, poses_t, fixed_t, points_t, edges_t):
        import torch.distributed as dist
        hppbp, packed, hpl = self.backend(poses_t, fixed_t, points_t, edges_t, self.cam, self.huber_delta)
        n_pose, n_pt = poses_t.shape[0], points_t.shape[0]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(packed[:12 * n_pt + 2], op=dist.ReduceOp.SUM, group=self.group)
            if self.gather_pose_blocks:
                if self.dst is None:
                    dist.all_reduce(hppbp, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.reduce(hppbp, self.dst, op=dist.ReduceOp.SUM, group=self.group)
        return dict(Hpp=hppbp[:36 * n_pose].view(n_pose, 6, 6), bp=hppbp[36 * n_pose:].view(n_pose, 6), Hll=packed[:9 * n_pt].view(n_pt, 3, 3),
                    bl=packed[9 * n_pt:12 * n_pt].view(n_pt, 3), Hpl=hpl.view(-1, 6, 3), chi2=packed[12 * n_pt:12 * n_pt + 2])


# ---------------------------------------------------------------------------------------------------------------------
# Host-side reduced-camera solve (what g2o's BlockSolver_6_3 + linear solver do; dense here: 50 keyframes -> 300 x 300)
# ---------------------------------------------------------------------------------------------------------------------
def schur_solve(Hpp, bp, Hll, bl, Hpl, edges, pose_fixed, lam=0.0):
    """Solve [Hpp Hpl; Hpl^T Hll] [dp; dl] = [bp; bl] by eliminating the landmarks. Returns (dp [n_pose,6], dl [n_pt,3])."""
    Hpp, bp, Hll, bl, Hpl = (np.asarray(a, np.float64) for a in (Hpp, bp, Hll, bl, Hpl))
    n_pose, n_pt = len(Hpp), len(Hll)
    free = np.nonzero(~np.asarray(pose_fixed, bool))[0] if pose_fixed is not None else np.arange(n_pose)
    slot = -np.ones(n_pose, np.int64)
    slot[free] = np.arange(len(free))
    Hll_d = Hll + lam * np.eye(3)[None] * np.maximum(np.einsum("nii->n", Hll)[:, None, None] / 3, 1e-12) if lam else Hll
    # a landmark seen from a single keyframe has a rank-2 block (depth unobservable): leave it untouched, as local BA does
    # by only inserting landmarks with >= 2 observations
    seen = np.bincount(edges["point_idx"], minlength=n_pt) >= 2
    Hinv = np.zeros_like(Hll)
    Hinv[seen] = np.linalg.inv(Hll_d[seen])
    S = np.zeros((6 * len(free), 6 * len(free)))
    g = np.zeros(6 * len(free))
    for k in free:
        s = slot[k]
        S[6 * s:6 * s + 6, 6 * s:6 * s + 6] = Hpp[k] * (1 + lam) if lam else Hpp[k]
        g[6 * s:6 * s + 6] = bp[k]
    pi, li = edges["pose_idx"], edges["point_idx"]
    keep = slot[pi] >= 0
    pi, li, W = pi[keep], li[keep], Hpl[keep]
    WH = np.einsum("eab,ebc->eac", W, Hinv[li])                 # W_kj Hll_j^-1   (6x3)
    np.subtract.at(g.reshape(-1, 6), slot[pi], np.einsum("eab,eb->ea", WH, bl[li]))
    order = np.argsort(li, kind="stable")
    pi_s, li_s, W_s, WH_s = pi[order], li[order], W[order], WH[order]
    starts = np.flatnonzero(np.r_[True, li_s[1:] != li_s[:-1], True])
    for a, b in zip(starts[:-1], starts[1:]):                   # all keyframe pairs seeing landmark j
        for i in range(a, b):
            si = slot[pi_s[i]]
            for j in range(a, b):
                sj = slot[pi_s[j]]
                S[6 * si:6 * si + 6, 6 * sj:6 * sj + 6] -= WH_s[i] @ W_s[j].T
    dp_free = np.linalg.solve(S, g)
    dp = np.zeros((n_pose, 6))
    dp[free] = dp_free.reshape(-1, 6)
    rhs = bl.copy()
    np.subtract.at(rhs, li, np.einsum("eab,ea->eb", W, dp[pi]))
    dl = np.einsum("nab,nb->na", Hinv, rhs)
    return dp, dl


def se3_oplus(poses, dp):
    """g2o VertexSE3Expmap::oplusImpl: T <- exp(update) * T with update = (omega, upsilon)."""
    out = np.array(poses, np.float64).reshape(-1, 7).copy()
    for i, u in enumerate(np.asarray(dp, np.float64)):
        om, up = u[:3], u[3:]
        th = np.linalg.norm(om)
        Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
        if th < 1e-5:
            R = np.eye(3) + Om + 0.5 * Om @ Om
            V = R
        else:
            R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * (Om @ Om)
            V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * (Om @ Om)
        Rold = quat_to_rot(out[i, 3:7])
        Rn = R @ Rold
        tn = R @ out[i, :3] + V @ up
        out[i, :3] = tn
        out[i, 3:7] = rot_to_quat(Rn)
    return out


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)
