"""CPU-side check of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/pqp.h declares.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

from path_optimizer_2_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pqp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pqp_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported(hip_lib):
    names = _declared()
    assert set(names) == set(capi.EXPORTS), (names, capi.EXPORTS)
    for nm in names:
        assert hasattr(hip_lib, nm), nm


def test_default_params_match_reference_flags(hip_lib):
    p = capi.default_params(hip_lib)
    assert (p.front_length, p.rear_length, p.wheel_base) == (3.9, -1.0, 2.5)
    assert p.expected_safety_margin == 0.6 and p.precise_planning_length == 30.0
    assert p.constraint_end_heading == 1 and p.rough_constraints_far_away == 0
    assert (p.weight_l, p.weight_kappa, p.weight_dkappa, p.weight_slack) == (0.0, 20.0, 100.0, 10.0)
    assert p.eps_abs == 2e-3 and p.eps_rel == 2e-3                       # base_solver.cpp:61-62
    assert (p.rho, p.sigma, p.alpha, p.max_iter, p.scaling) == (0.1, 1e-6, 1.6, 4000, 10)


def test_sizes_is_pure_host_logic(hip_lib):
    s = capi.PqpSizes()
    p = capi.default_params(hip_lib)
    assert hip_lib.pqp_path_sizes(C.byref(p), 80, None, C.byref(s)) == 0
    assert (s.vars, s.cons, s.nnz_a, s.nnz_p) == (479, 482, 1355, 319)
    assert hip_lib.pqp_path_sizes(C.byref(p), 200, None, C.byref(s)) == 0
    assert (s.vars, s.cons, s.nnz_a, s.nnz_p) == (1199, 1202, 3395, 799)


def test_no_device_fails_loudly(hip_lib):
    """There is no CPU fallback: without a HIP device pqp_create must fail, not degrade."""
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    p = capi.default_params(hip_lib)
    rc = hip_lib.pqp_create(C.byref(h), C.byref(p), 0, 8, 80)
    assert rc != 0 and not h.value
    assert b"no HIP device" in hip_lib.pqp_last_error() or rc == -3


def test_stream_batch_default_is_the_measured_crossover(hip_lib):
    """PQP_OPT_STREAM_BATCH's default (pure host logic): from how many QPs of n waypoints on a cold call runs on the lane-per-QP kernel - the crossover of the two
    path kernels measured on one MI355X (profiles/r06ay_crossover_hybrid.txt, r06az_crossover_other_n.txt: 15 k at 80 waypoints, 20 k at 100, 29 k at 120,
    49 k at 256, 11.5 k at 300, 24.5 k at 512); an 8192-QP shard of configs[3] stays on the lane-per-waypoint kernel at every length up to 128 waypoints."""
    f = lambda n: capi.stream_batch_default(n, hip_lib)
    assert f(80) == 15360 and f(60) == 15360 and f(1) == 0
    assert 21000 < f(100) < 22000 and 28000 < f(120) < 28500 and f(128) > 30000
    assert f(200) == 30000 and f(256) == 49152
    assert f(300) == 14400 and f(512) == 24576
    assert all(f(n) > 8192 for n in range(2, 129))
