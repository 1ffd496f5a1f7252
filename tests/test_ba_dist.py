"""The N>1 path of the local-BA linearisation on CPU: world_size 2, 4 and 8 (the driver's scaling run), gloo. Edges are sharded by keyframe, every rank computes
its shard's partial blocks (here with the ORACLE as the shard backend -- this test is about the sharding and the exchange
step, the HIP backend runs under a 1-rank nccl group in tests/test_gpu_ba.py::test_graph_backend_under_nccl_group), then Hll|bl|chi2 are all-reduced in ONE packed collective. Result must equal the one-process
linearisation within the stated multi-rank tolerance 1e-10 (rel.; summation order differs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openvslam_amd import ba
from openvslam_amd.synth import synth_local_ba


def _oracle_backend(poses_t, fixed_t, points_t, edges_t, cam, huber_delta):
    from oracle import binding as ob
    edges = edges_t.numpy().view(ob.BA_EDGE_DTYPE)
    o = ob.ba_linearize(poses_t.numpy(), fixed_t.numpy() if fixed_t is not None else None, points_t.numpy(), edges, cam, huber_delta)
    hppbp = torch.from_numpy(np.concatenate([o["Hpp"].ravel(), o["bp"].ravel()]))
    packed = torch.from_numpy(np.concatenate([o["Hll"].ravel(), o["bl"].ravel(), o["chi2"].ravel(), np.zeros(2)]))   # ONE buffer, ONE collective
    return hppbp, packed, torch.from_numpy(o["Hpl"].reshape(-1, 18).copy())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth_local_ba(n_pose=8, n_pt=500, obs_per_pose=200, seed=4)
        shard = ba.shard_edges_by_keyframe(d["edges"], 8, rank, world)
        lin = ba.local_ba_linearizer(d["cam"], d["huber_delta"], backend=_oracle_backend)
        out = lin.linearize(torch.from_numpy(d["poses"]), torch.from_numpy(d["pose_fixed"]), torch.from_numpy(d["points"]),
                            torch.from_numpy(shard.view(np.uint8)))
        q.put((rank, {k: v.numpy().copy() for k, v in out.items() if k != "Hpl"}, out["Hpl"].numpy().copy(), shard["pose_idx"].copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_world_size_n_gloo(oracle, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    d = synth_local_ba(n_pose=8, n_pt=500, obs_per_pose=200, seed=4)
    want = oracle.ba_linearize(d["poses"], d["pose_fixed"], d["points"], d["edges"], d["cam"], d["huber_delta"])
    for rank, out, hpl, pose_idx in res:
        for k in ("Hpp", "bp", "Hll", "bl", "chi2"):
            scale = np.abs(want[k]).max()
            assert np.allclose(out[k], want[k], rtol=1e-10, atol=1e-10 * scale), (rank, k)
        # Hpl stays local: this rank's edges only, bit-identical to the one-process value
        sel = (d["edges"]["pose_idx"] // (8 // world)) == rank
        assert np.array_equal(hpl, want["Hpl"][sel])
    # the two ranks hold the SAME reduced landmark blocks
    assert np.array_equal(res[0][1]["Hll"], res[1][1]["Hll"]) and np.array_equal(res[0][1]["bl"], res[1][1]["bl"])
