"""Build + drive tests/emu/liblq_emu.so: the lane-per-QP solver source (csrc/pqp_path_lq.hpp) compiled for the host (test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

from path_optimizer_2_amd.capi import PqpParams

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "emu", "lq_emu.cpp")
LIB = os.path.join(HERE, "emu", "liblq_emu.so")
# PQP_SANITIZED_LIBS=<dir>: load a prebuilt (-fsanitize=address,undefined) library from there instead (tools/sanitize_cpu.sh)
if os.environ.get("PQP_SANITIZED_LIBS"):
    LIB = os.path.join(os.environ["PQP_SANITIZED_LIBS"], "liblq_emu.so")
_DEPS = [SRC] + [os.path.join(ROOT, "path_optimizer_2_amd", "csrc", f) for f in ("pqp_path_lq.hpp", "pqp_path_lane.hpp", "pqp_defaults.hpp")] + \
        [os.path.join(ROOT, "include", "pqp.h")]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.environ.get("PQP_SANITIZED_LIBS") and (not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in _DEPS)):
            subprocess.run(["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
        _lib = C.CDLL(LIB)
    return _lib


def production(**over):
    p = PqpParams()
    load().pqp_emu_lq_production_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def solve(ref, bounds, scal, passes=1, n_of=None, lin=None, prm=None, sorted_launch=False, chunk_layout=False):
    lib = load()
    B, n = ref.shape[:2]
    prm = prm or production()
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    out = np.zeros((B, n, 7)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); info = np.zeros((B, 8))
    ref = np.ascontiguousarray(ref, dtype=np.float64); bounds = np.ascontiguousarray(bounds, dtype=np.float64)
    scal = np.ascontiguousarray(scal, dtype=np.float64)
    lin = None if lin is None else np.ascontiguousarray(lin, dtype=np.float64)
    n_of = None if n_of is None else np.ascontiguousarray(n_of, dtype=np.int32)
    # sorted_launch: as in a launch with Args::order (the re-linearised pass starts with active-set rounds on the previous pass's set)
    if chunk_layout:          # the solver over lq::ChunkWs (the workspace layout of the device's staged form)
        lib.pqp_emu_lq_solve_chunk_layout(C.byref(prm), B, n, vp(n_of), vp(ref), vp(lin), vp(bounds), vp(scal), passes, vp(out), vp(st), vp(it), vp(info), 1 if sorted_launch else 0)
        return dict(out=out, status=st, iters=it, info=info)
    (lib.pqp_emu_lq_solve_sorted if sorted_launch else lib.pqp_emu_lq_solve)(C.byref(prm), B, n, vp(n_of), vp(ref), vp(lin), vp(bounds), vp(scal), passes, vp(out), vp(st), vp(it), vp(info))
    return dict(out=out, status=st, iters=it, info=info)


class Carried:
    """A batch whose per-QP workspaces persist between solves (the device keeps them on the handle): PQP_OPT_CARRY_CYCLES in emulation."""

    def __init__(self, batch, n):
        self.ws = np.zeros((batch, n * load().pqp_emu_lq_fields()))
        self.first = True

    def solve(self, ref, bounds, scal, passes=1, prm=None, carry=True):
        lib = load()
        B, n = ref.shape[:2]
        prm = prm or production()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        out = np.zeros((B, n, 7)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); info = np.zeros((B, 8))
        ref = np.ascontiguousarray(ref, dtype=np.float64); bounds = np.ascontiguousarray(bounds, dtype=np.float64); scal = np.ascontiguousarray(scal, dtype=np.float64)
        lib.pqp_emu_lq_solve_carry(C.byref(prm), B, n, None, vp(ref), None, vp(bounds), vp(scal), passes, vp(out), vp(st), vp(it), vp(info), vp(self.ws),
                                   1 if (carry and not self.first) else 0)
        self.first = False
        return dict(out=out, status=st, iters=it, info=info)


def timed_rate(make_sample, n, budget_s=6.0):
    """bench.py's cpu_baseline.same_algorithm_on_host: the product's lane-per-QP algorithm source compiled for the host (test infrastructure,
    tests/emu/lq_emu.cpp), one QP per OpenMP task over all host threads, on a bounded sample of the bench workload."""
    import time
    lib = load()
    cores = lib.pqp_emu_lq_threads()
    probe = make_sample(64 * cores)
    t0 = time.perf_counter()
    solve(probe["ref"], probe["bounds"], probe["scal"])
    per = (time.perf_counter() - t0) / (64 * cores)
    k = int(min(1 << 20, max(64 * cores, budget_s / max(per, 1e-8))))
    b = make_sample(k)
    t0 = time.perf_counter()
    r = solve(b["ref"], b["bounds"], b["scal"])
    dt = time.perf_counter() - t0
    return {"value": k / dt, "unit": "paths/s", "cores": cores, "per_core": k / dt / cores, "solved": int((r["status"] == 1).sum()),
            "sample": f"{k} paths of the bench distribution (N={n}) in {dt:.1f} s: csrc/pqp_path_lq.hpp (interior-point + active-set rounds, every path the "
                      f"exact optimum) compiled for the host with g++ -O3 -march=native -fopenmp, one QP per task over {cores} threads"}
