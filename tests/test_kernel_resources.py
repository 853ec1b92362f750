"""Register / scratch budget of the HIP kernels, from the compiler's own report written by the build
(__graft_entry__.build_hip -> csrc/libpqp_hip.resources.txt).  The solve kernels keep 86 doubles of lane state in registers;
LLVM gives that up silently for innocent-looking source changes, and the kernel then runs several times
slower while every parity test still passes.  This test is the tripwire."""
import os
import re

import pytest

import __graft_entry__ as g


def _report():
    g.build_hip()
    if not os.path.exists(g.RESOURCES):
        g.build_hip(force=True)
    kernels, cur = {}, None
    for line in open(g.RESOURCES):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.fixture(scope="module")
def kernels():
    return _report()


def _find(kernels, *parts):
    hits = [(k, v) for k, v in kernels.items() if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, [k for k, _ in hits])
    return hits[0][1]


@pytest.mark.parametrize("nw", [1, 2, 4])
@pytest.mark.parametrize("cert", [0, 1])
def test_path_solve_kernel_keeps_its_lane_state_in_registers(kernels, nw, cert):
    r = _find(kernels, "path_solve_kernel", f"ILi{nw}ELb{cert}E")
    assert r["ScratchSize"] <= 256, r         # 32-128 B today: a few spill slots of the cold code (none in the ADMM loop); the lane
                                              # struct in scratch would be 1 KB
    assert r["Occupancy"] >= 1


@pytest.mark.parametrize("cert", [0, 1])
def test_path_solve_kernel_of_512_lanes_spills_no_more_than_today(kernels, cert):
    """NW = 8 (257-512 waypoints): a 512-lane workgroup is 8 wavefronts on one compute unit's 4 SIMDs, i.e. two per SIMD and 256 registers per
    lane whatever the launch bounds say - half of what the lane state + temporaries take (496 at NW <= 4).  The spill slots that follow are
    the price of one lane per waypoint at that size (1 024-1 184 B today); beyond them the lane struct itself would go."""
    r = _find(kernels, "path_solve_kernel", f"ILi8ELb{cert}E")
    assert r["ScratchSize"] <= 1280, r
    assert r["VGPRs"] <= 256 and r["Occupancy"] >= 2


@pytest.mark.parametrize("staged", [0, 1])
def test_path_stream_kernel_register_budget(kernels, staged):
    """the lane-per-QP kernel in both of its forms (register prefetch one waypoint ahead / records staged in LDS two ahead): one wavefront per SIMD, all 512
    registers, at most a few spill slots of the once-per-pass code (0-360 B from build to build: the allocator's noise at this size; two waypoints ahead IN
    REGISTERS spill 500-2900 B and are slower, profiles/r03a_stream_first.txt); the staged form's 30 KB of LDS leave room for four workgroups per CU"""
    r = _find(kernels, "path_stream_kernel", f"ILb{staged}E")
    assert r["ScratchSize"] <= 512, r
    assert r["Occupancy"] >= 1
    assert r["LDS Size"] == (30720 if staged else 0), r


@pytest.mark.parametrize("b,maxt,stage", [(3, 256, 1), (4, 256, 1), (9, 256, 1), (3, 512, 1), (4, 512, 1), (3, 512, 0), (4, 512, 0)])
def test_banded_solve_kernel_registers(kernels, b, maxt, stage):
    r = _find(kernels, "banded_solve_kernel", f"ILi{b}ELi{maxt}ELb{stage}E")
    assert r["ScratchSize"] <= 512, r         # spill slots of the setup / polish code (180-408 B today); a BqLane<9> in scratch would be 700 B


@pytest.mark.parametrize("name,tag", [("post_exact_kernel", "ILi1E"), ("post_exact_kernel", "ILi2E"), ("post_exact_kernel", "ILi4E"), ("post_exact_kernel", "ILi6E"),
                                      ("tension_exact_kernel", "ILi1E"), ("tension_exact_kernel", "ILi2E"), ("tension_exact_kernel", "ILi4E"), ("tension_exact_kernel", "ILi6E"),
                                      ("tension2_exact_kernel", ""), ("tension2_stage_kernel", "")])
def test_exact_smoother_kernels_live_in_registers(kernels, name, tag):
    """one wavefront (or one lane) per scenario, up to six layers / points per lane in register arrays: a dynamic index into one of them
    would put it in scratch"""
    r = _find(kernels, name, tag)
    assert r["ScratchSize"] == 0, r


def test_kernels_around_the_qps_do_not_use_scratch(kernels):
    for name in ("corridor_bounds_kernel", "reference_states_kernel", "spline_fit_kernel", "dp_corridor_kernel"):
        assert _find(kernels, name)["ScratchSize"] == 0, name
