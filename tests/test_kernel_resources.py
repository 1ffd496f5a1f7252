"""The occupancy figures DESIGN.md argues with, read from the gfx950 code objects hipcc emits here (no device): the hot kernels stay free of
scratch and inside the register / LDS budgets their workgroups-per-CU depend on. A compiler or source change that pushes one of them over
its tier shows up here and not as an unexplained slowdown on the device."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources as kr
    rows = kr.collect(files={"orb_fast.hip", "orb_describe.hip", "orb_tree.hip", "orb_pyramid.hip"})
    return {name: dict(vgpr=v, agpr=a, sgpr=s, scratch=p, lds=g, max_wg=w) for (_, name, v, a, s, p, g, w) in rows}


def test_hot_kernels_have_no_scratch_and_keep_their_occupancy_tier(kernels):
    fast = kernels["ovs::k_fast_cells<false>"]
    # seven 256-thread workgroups per CU: <= 72 registers (512 / 7 waves per SIMD) and <= 160 KiB / 7 of LDS
    assert fast["scratch"] == 0 and fast["vgpr"] + fast["agpr"] <= 72 and fast["lds"] <= 160 * 1024 // 7
    desc = kernels["ovs::k_describe"]
    assert desc["scratch"] == 0 and desc["vgpr"] <= 32 and desc["lds"] <= 6400          # 25 one-wave workgroups per CU by LDS
    tree = kernels["ovs::k_tree<512>"]
    assert tree["scratch"] == 0 and tree["vgpr"] <= 80 and tree["lds"] == 0             # three 512-thread workgroups per CU (dynamic LDS aside)
    assert kernels["ovs::k_tree<1024>"]["scratch"] == 0 and kernels["ovs::k_tree<1024>"]["vgpr"] <= 128
    pyr = kernels["ovs::k_resize_linear_u8"]
    assert pyr["scratch"] == 0 and pyr["vgpr"] <= 64 and pyr["lds"] <= 20 * 1024        # eight workgroups per CU


@pytest.fixture(scope="module")
def ba_kernels():
    import kernel_resources as kr
    rows = kr.collect(files={"ba_solve.hip", "ba_graph.hip", "pose_opt.hip"})
    return {name: dict(vgpr=v, agpr=a, sgpr=s, scratch=p, lds=g, max_wg=w) for (_, name, v, a, s, p, g, w) in rows}


def test_optimiser_kernels_keep_their_register_budgets(ba_kernels):
    """Round 4's kernels of the optimisers: the one-workgroup Cholesky holds its in-flight trailing tiles in registers (512 threads: two waves
    per SIMD, 256 registers each) and must not spill; the pair kernel of the reduced camera system keeps three waves per SIMD; the
    256-thread pose optimiser (the form a frame spread over several workgroups uses) stays free of scratch."""
    solve = [v for k, v in ba_kernels.items() if k.startswith("ovs::k_chol_solve")]
    assert len(solve) == 1 and solve[0]["scratch"] == 0 and solve[0]["vgpr"] + solve[0]["agpr"] <= 256
    pairs = [v for k, v in ba_kernels.items() if k == "ovs::k_schur"]   # pair blocks + right-hand side rows in one launch (round 5)
    assert len(pairs) == 1 and pairs[0]["scratch"] == 0 and pairs[0]["vgpr"] + pairs[0]["agpr"] <= 168     # 512 / 3 waves per SIMD
    pose = {k: v for k, v in ba_kernels.items() if k.startswith("ovs::k_pose_optimize<")}
    # (third template argument, round 6: 2 = a thread's at most two observations held in registers, 0 = re-read from memory per pass)
    for name in ("ovs::k_pose_optimize<0, 256, 0>", "ovs::k_pose_optimize<1, 256, 0>", "ovs::k_pose_optimize<0, 256, 2>", "ovs::k_pose_optimize<1, 256, 2>"):
        assert pose[name]["scratch"] == 0 and pose[name]["vgpr"] + pose[name]["agpr"] <= 256, name   # two waves per SIMD or more


def test_no_kernel_touches_scratch_memory():
    """Every kernel of the library, read from the code objects: none uses scratch (private memory spilled to HBM). Until round 5 the 512-thread
    builds of the pose optimiser did (120 / 208 bytes: literal constants of sincos / the Newton steps hoisted out of the iteration loops by
    machine LICM and reloaded one by one on the lane every trial waits for); pose_opt.o is built with -mllvm -disable-machine-licm since."""
    import kernel_resources as kr
    spilled = {name: scratch for (_, name, _v, _a, _s, scratch, _l, _w) in kr.collect() if scratch}
    assert not spilled, spilled
