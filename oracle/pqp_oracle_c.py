"""ctypes wrapper of oracle/libpqp_oracle.so (the plain-C restatement, oracle/pqp_oracle.c).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pqp_oracle.c")
LIB = os.path.join(HERE, "libpqp_oracle.so")
# PQP_SANITIZED_LIBS=<dir>: load a prebuilt (-fsanitize=address,undefined) library from there instead (tools/sanitize_cpu.sh)
_SANITIZED = os.environ.get("PQP_SANITIZED_LIBS")
if _SANITIZED:
    LIB = os.path.join(_SANITIZED, "libpqp_oracle.so")


class PqoParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("front_length", "rear_length", "wheel_base", "expected_safety_margin", "weight_l",
                                          "weight_kappa", "weight_dkappa", "weight_slack", "end_l_bound", "end_psi_tol",
                                          "end_psi_max", "min_clearance")] + \
               [("constraint_end_heading", C.c_int)] + \
               [(k, C.c_double) for k in ("eps_abs", "eps_rel", "rho", "sigma", "alpha")] + \
               [(k, C.c_int) for k in ("max_iter", "scaling", "adaptive_rho", "adaptive_rho_interval", "check_termination")] + \
               [("adaptive_rho_tolerance", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not _SANITIZED and (not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB)):
        subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"], check=True)
    _lib = C.CDLL(LIB)
    return _lib


def params(**over):
    p = PqoParams()
    load().pqo_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_path(prm, ref, bounds, scal, passes=1, lin0=None):
    lib = load()
    n = ref.shape[0]
    ref = np.ascontiguousarray(ref); bounds = np.ascontiguousarray(bounds); scal = np.ascontiguousarray(scal)
    lin0 = None if lin0 is None else np.ascontiguousarray(lin0)
    out = np.zeros((n, 7)); x = np.zeros(6 * n - 1); y = np.zeros(6 * n + 2)
    its = np.zeros(passes + 1, dtype=np.int32); st = np.zeros(passes + 1, dtype=np.int32); rho = C.c_double()
    ok = lib.pqo_solve_path(C.byref(prm), n, _vp(ref), _vp(lin0), _vp(bounds), _vp(scal), passes, _vp(out), _vp(x), _vp(y),
                            _vp(its), _vp(st), C.byref(rho))
    return dict(ok=bool(ok), out=out, x=x, y=y, iters=its, status=st, rho=rho.value)


def solve_batch(prm, ref, bounds, scal, passes=1, threads=0):
    lib = load()
    B, n = ref.shape[:2]
    out = np.zeros((B, n, 7)); its = np.zeros(B, dtype=np.int32)
    solved = lib.pqo_solve_batch(C.byref(prm), B, n, _vp(np.ascontiguousarray(ref)), _vp(np.ascontiguousarray(bounds)),
                                 _vp(np.ascontiguousarray(scal)), passes, threads, _vp(out), _vp(its))
    return dict(out=out, iters=its, solved=solved)


def timed_baseline(make_sample, n, eps, budget_s=15.0, rho_interval=100):
    """bench.py's cpu_baseline: the C restatement of the OSQP-paper algorithm (no polish: that is what the reference
    runs) with one path per task over all host cores, on a bounded sample of the SAME workload:
    make_sample(k) returns k scenarios of the bench's distribution; k is sized to ~budget_s seconds of wall time.
    Both assembly modes of SURVEY.md 8(d): direct O(N) structural fill (`value`) and the reference's dense cons x vars fill +
    sparseView scan, O(N^2) (`reference_faithful_assembly`), each on half the budget."""
    lib = load()
    cores = lib.pqo_num_threads()
    prm = params(eps_abs=eps, eps_rel=eps, adaptive_rho_interval=rho_interval)

    def run(dense, budget):
        lib.pqo_set_dense_assembly(1 if dense else 0)
        try:
            probe = make_sample(4 * cores)
            t0 = time.perf_counter()
            solve_batch(prm, probe["ref"], probe["bounds"], probe["scal"])
            per = (time.perf_counter() - t0) / (4 * cores)
            k = int(min(262144, max(4 * cores, budget / max(per, 1e-7))))
            b = make_sample(k)
            t0 = time.perf_counter()
            r = solve_batch(prm, b["ref"], b["bounds"], b["scal"])
            return k, time.perf_counter() - t0, r
        finally:
            lib.pqo_set_dense_assembly(0)

    k, dt, r = run(False, 0.5 * budget_s)
    kd, dtd, rd = run(True, 0.5 * budget_s)
    # one thread alone (128 SMT threads hide what one core does)
    one = make_sample(24)
    t0 = time.perf_counter()
    solve_batch(prm, one["ref"], one["bounds"], one["scal"], threads=1)
    dt1 = time.perf_counter() - t0
    solve_batch(prm, one["ref"][:1], one["bounds"][:1], one["scal"][:1], threads=cores)      # (omp_set_num_threads is process-wide: back to all)
    return {"single_thread": {"value": 24 / dt1, "unit": "paths/s", "sample": f"24 paths on one thread in {dt1:.2f} s"},
            "value": k / dt, "unit": "paths/s", "cores": cores, "kind": "port",
            "sample": f"{k} paths of the bench distribution (N={n}) in {dt:.1f} s, OSQP-paper restatement in C (oracle/pqp_oracle.c), "
                      f"eps {eps:g}, no polish, direct O(N) assembly, one path per OpenMP task over {cores} threads; mean ADMM iterations "
                      f"{float(r['iters'].mean()):.0f}; solved {r['solved']}/{k}",
            "per_core": k / dt / cores,
            "reference_faithful_assembly": {"value": kd / dtd, "unit": "paths/s", "per_core": kd / dtd / cores,
                                            "sample": f"{kd} paths in {dtd:.1f} s with the reference's dense (6N+2) x (6N-1) fill + sparseView scan "
                                                      f"twice per path (base_solver.cpp:122,145,159,210); solved {rd['solved']}/{kd}"}}
