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
