"""The C++ side of the drop-in boundary: PathOptimizationNS::BaseSolver (include/pqp_base_solver.hpp) over the C ABI.
CPU: it must compile and link against libpqp_hip.so and fail cleanly (solve() == false) without a GPU.
GPU: the optimizePath call sequence must reproduce the oracle's two-pass result."""
import os
import subprocess

import numpy as np
import pytest

import pqp_oracle as O
from path_optimizer_2_amd.synth import make_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "path_optimizer_2_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cpp", "shim_demo")


@pytest.fixture(scope="module")
def shim_exe(hip_lib):
    src = [os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"), os.path.join(CSRC, "base_solver_shim.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", EXE] + src + ["-L" + CSRC, "-lpqp_hip", "-Wl,-rpath," + CSRC], check=True)
    return EXE


@pytest.fixture(scope="module")
def latency_exe(hip_lib):
    exe = os.path.join(ROOT, "tests", "cpp", "shim_latency")
    src = [os.path.join(ROOT, "tests", "cpp", "shim_latency.cpp"), os.path.join(CSRC, "base_solver_shim.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe] + src + ["-L" + CSRC, "-lpqp_hip", "-Wl,-rpath," + CSRC], check=True)
    return exe


REF_INCLUDE = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference tree is not on this box")
@pytest.mark.parametrize("unit", ["path_optimizer_2_amd/csrc/base_solver_shim.cpp", "include/pqp_batched_solver.hpp"])
def test_drop_in_mode_compiles_against_the_reference_headers(unit):
    """with PQP_USE_REFERENCE_TYPES the shim and the batched solver use the reference's OWN ReferencePath /
    VehicleState / SlState (include/data_struct/*.hpp, std-only) instead of include/pqp_types.hpp.  Pinned here as a compile check of
    exactly that configuration (syntax + types; the link step needs the reference's objects)."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-DPQP_USE_REFERENCE_TYPES", "-I" + REF_INCLUDE, "-I" + os.path.join(ROOT, "include"),
           "-include", "data_struct/data_struct.hpp", "-include", "data_struct/reference_path.hpp", "-include", "data_struct/vehicle_state_frenet.hpp",
           os.path.join(ROOT, unit)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "error" not in r.stderr


def _scenario_text(b, q):
    n = b["ref"].shape[1]
    lines = [str(n)]
    for i in range(n):
        lines.append(" ".join(repr(float(v)) for v in list(b["ref"][q, i]) + list(b["bounds"][q, i])))
    lines.append(" ".join(repr(float(v)) for v in b["scal"][q]))
    return "\n".join(lines) + "\n"


def test_shim_builds_and_fails_cleanly_without_gpu(shim_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    b = make_batch(1, 20)
    r = subprocess.run([shim_exe], input=_scenario_text(b, 0), capture_output=True, text=True)
    assert r.returncode == 1 and "Pre solving failed" in r.stderr     # bool false, like a failed initSolver(); no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reference", "polish"])
def test_shim_reproduces_optimize_path(shim_exe, mode):
    b = make_batch(2, 80)
    for q in range(2):
        args = [shim_exe] + (["polish"] if mode == "polish" else [])
        r = subprocess.run(args, input=_scenario_text(b, q), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.array([[float(v) for v in ln.split()] for ln in r.stdout.strip().splitlines()])
        assert got.shape == (80, 7)
        if mode == "reference":      # the reference's own setting: eps 2e-3, no polish -> same ADMM, same stopping check
            ref = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=O.OsqpSettings())
            assert np.abs(got - ref[-1]["out"]).max() < 1e-7
        else:
            ref = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
            assert np.abs(got[:, 3:5] - ref[-1]["out"][:, 3:5]).max() < 1e-6


@pytest.mark.gpu
def test_rough_constraints_mode_through_the_drop_in(shim_exe):
    """FLAGS_rough_constraints_far_away (base_solver.cpp:22-37,201-205,241-247) set through setParams(): vars() / cons() / the precise planning size
    follow the parameters (the reference reads the flag at construction), and the path is the oracle's in that mode."""
    import re
    b = make_batch(1, 80)
    length = 12.0
    r = subprocess.run([shim_exe, "polish", "rough=%r" % length], input=_scenario_text(b, 0), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.search(r"vars (\d+) cons (\d+) precise (\d+)", r.stderr)
    n = 80
    precise = int(np.searchsorted(b["ref"][0, :, 0], length, side="left"))          # std::lower_bound on s (base_solver.cpp:25-34)
    assert 0 < precise < n
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (3 * n + (n - 1) + precise + n, 4 * n + precise + n + 2, precise)
    prm = O.PathQpParams(rough_constraints_far_away=True, precise_planning_length=length)
    want = O.solve_path(b["ref"][0], b["bounds"][0], b["scal"][0], prm=prm, st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
    got = np.array([[float(v) for v in ln.split()] for ln in r.stdout.strip().splitlines()])
    assert np.abs(got[:, 3:5] - want[-1]["out"][:, 3:5]).max() < 1e-6
    # ... and the default mode reports the default sizes
    r0 = subprocess.run([shim_exe, "polish"], input=_scenario_text(b, 0), capture_output=True, text=True)
    m0 = re.search(r"vars (\d+) cons (\d+) precise (\d+)", r0.stderr)
    assert (int(m0.group(1)), int(m0.group(2)), int(m0.group(3))) == (6 * n - 1, 6 * n + 2, n)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["nan_s", "equal_s", "inf_bound", "nan_start"])
def test_a_scenario_that_is_not_a_number_makes_solve_return_false(shim_exe, what):
    """NaN / Inf in the scenario, or an arclength that does not increase (the reference divides by ds, base_solver.cpp:174,180): solve() == false,
    PQP_STATUS_NUMERICAL, as when OSQP hands back non-finite iterates (base_solver.cpp:80-88)."""
    b = make_batch(1, 60)
    if what == "nan_s": b["ref"][0, 17, 0] = np.nan
    if what == "equal_s": b["ref"][0, 30, 0] = b["ref"][0, 29, 0]
    if what == "inf_bound": b["bounds"][0, 5, 1] = np.inf
    if what == "nan_start": b["scal"][0, 1] = np.nan
    for args in ([shim_exe], [shim_exe, "polish"]):
        r = subprocess.run(args, input=_scenario_text(b, 0), capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "Pre solving failed" in r.stderr and "status 3" in r.stderr, r.stderr


@pytest.mark.gpu
def test_device_ordinal_comes_from_the_environment(shim_exe):
    """PQP_DEVICE selects the GPU of a BaseSolver (the reference has no notion of a device); an ordinal this box does not have makes
    solve() return false, like any failed set-up."""
    import torch
    b = make_batch(1, 40)
    ok = subprocess.run([shim_exe], input=_scenario_text(b, 0), capture_output=True, text=True, env=dict(os.environ, PQP_DEVICE="0"))
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run([shim_exe], input=_scenario_text(b, 0), capture_output=True, text=True, env=dict(os.environ, PQP_DEVICE=str(torch.cuda.device_count())))
    assert bad.returncode == 1 and "bad device ordinal" in bad.stderr


@pytest.mark.gpu
def test_one_solver_per_planning_cycle_reuses_its_handle(latency_exe):
    """The reference's pattern - a BaseSolver per cycle (path_optimizer.cpp:138) - on the pooled handle: later cycles do not pay pqp_create."""
    import json
    b = make_batch(1, 60)
    runs = {}
    for cache in (1, 0):
        r = subprocess.run([latency_exe, "12", str(cache), "0"], input=_scenario_text(b, 0), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        runs[cache] = json.loads(r.stdout)
    assert runs[1]["n"] == 60 and runs[1]["admm_iters"][0] > 0
    assert runs[1]["cycle_us_median"] < runs[1]["first_cycle_us"]
    assert runs[1]["cycle_us_median"] < runs[0]["cycle_us_median"]          # (uncached: every cycle creates stream, events and workspaces)
