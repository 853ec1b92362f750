"""A third-party pin of the oracle's QPs (round 6): HiGHS's QP solver - bundled with scipy >= 1.15 as scipy.optimize._highspy, nothing installed for it - solves
the QPs the oracle assembles line by line from the reference (the path QP of base_solver.cpp:119-261 in both linearisations and in rough-constraints mode,
TensionSmoother2's, TensionSmoother's, postSmooth's) and must return the optimum the oracle's OSQP restatement converges to, multipliers included.  The
committed golden vectors (tests/golden/path_n8.npz, path_n80.npz, smoothers.npz - produced by the oracle) are pinned the same way.

What this does and does not settle: OSQP's ITERATES at the reference's eps 2e-3 stay unpinned (upstream OSQP is not in this image: test_upstream_osqp.py);
that the oracle assembles a QP whose unique optimum a solver nobody here wrote agrees with - to 2e-7 in every variable and multiplier - is settled here.

HiGHS's QP solver regularises the Hessian by 1e-7 I (a constant of its active-set solver; this scipy's HiGHS has no option for it).  The path QP has
weight_l = 0 (planning_flags.cpp): its lateral offsets are held by the constraints and the curvature cost alone, and 1e-7 l^2 moves them by up to 5e-5
along that flat direction.  So the tight comparison is with the oracle on P + 1e-7 I (2e-7: it is also the proof that the regularisation is all that
differs); on the original P the two agree in the objective to 1e-9 and in the variables to the flat direction's slack."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import pqp_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import highs_qp as H
from path_optimizer_2_amd.synth import make_batch

pytestmark = pytest.mark.skipif(not H.available(), reason="this scipy does not bundle the HiGHS QP interface (scipy.optimize._highspy._core)")

HIGHS_REG = 1e-7
TIGHT = O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=400000)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pin(P, q, A, lo, up, x_star=None, y_star=None, flat=None, tol_x=2e-7, tol_y=2e-6):
    """HiGHS on (P, q, A, lo, up) against the oracle's ADMM on the same QP with HiGHS's regularisation (tight), and on the original QP (objective;
    variables outside `flat` - a mask of the weakly determined ones - to 1e-5).  x_star / y_star: a stored optimum of the original QP."""
    Pm = sp.diags(np.asarray(P, dtype=np.float64)) if np.ndim(P) == 1 else sp.csc_matrix(P)
    xh, yh, _ = H.solve_qp(Pm, q, A, lo, up)
    Ax = A @ xh
    assert np.maximum(lo - Ax, Ax - up).max() < 1e-8                                     # feasible
    reg = O.osqp_admm(sp.csc_matrix(Pm + HIGHS_REG * sp.identity(Pm.shape[0])), q, A, lo, up, TIGHT)
    assert reg["status"] == "solved"
    assert np.abs(xh - reg["x"]).max() < tol_x, np.abs(xh - reg["x"]).max()              # every variable: states, controls, slacks
    assert np.abs(yh - reg["y"]).max() < tol_y * max(1.0, np.abs(reg["y"]).max()), np.abs(yh - reg["y"]).max()       # every multiplier
    if x_star is None:
        r = O.osqp_admm(sp.csc_matrix(Pm), q, A, lo, up, TIGHT)
        assert r["status"] == "solved"
        x_star, y_star = r["x"], r["y"]
    f = lambda z: 0.5 * z @ (Pm @ z) + q @ z
    assert abs(f(xh) - f(x_star)) < 1e-8 * max(1.0, abs(f(x_star))), (f(xh), f(x_star))   # the same optimal value on the ORIGINAL QP
    firm = np.ones(len(xh), bool) if flat is None else ~flat
    assert not firm.any() or np.abs(xh - x_star)[firm].max() < 1e-5, np.abs(xh - x_star)[firm].max()
    assert np.abs(xh - x_star).max() < 2e-4                                                # the flat direction: what 1e-7 l^2 can move
    if y_star is not None:
        assert np.abs(yh - y_star).max() < 1e-4 * max(1.0, np.abs(y_star).max())
    return xh, x_star


def lateral(n, sz):
    """mask of the path QP's weakly determined variables: the lateral offsets l_i (weight_l = 0) and the collision rows' slacks, which follow them"""
    m = np.zeros(sz["vars"], bool)
    m[0:3 * n:3] = True
    m[4 * n - 1:] = True
    return m


@pytest.mark.parametrize("name", ["path_n8", "path_n80"])
def test_golden_path_qps_against_highs(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    n = z["ref"].shape[1]
    for q in range(z["ref"].shape[0]):
        Pd, A, lo, up, sz = O.assemble_path_qp(z["ref"][q], z["lin"][q], z["bounds"][q], z["scal"][q])
        assert np.array_equal(np.asarray(A[z["rows"], z["cols"]]).ravel(), z["a_val"][q]) and np.array_equal(lo, z["lower"][q]) and np.array_equal(up, z["upper"][q])
        xh, _ = pin(Pd, np.zeros(sz["vars"]), sp.csc_matrix(A), lo, up, x_star=z["x_star"][q], y_star=z["y_star"][q], flat=lateral(n, sz))
        out = O.unpack_path(xh, z["ref"][q])
        assert np.abs(out - z["out_star"][q])[:, 4:7].max() < 1e-5 and np.abs(out - z["out_star"][q])[:, 0:4].max() < 2e-4


@pytest.mark.parametrize("n,profile,seed,rough", [(80, "uniform", None, False), (60, "varied", 7, False), (120, "varied", 2, False), (80, "varied", 5, True), (200, "uniform", 4, False)])
def test_both_passes_of_optimize_path_against_highs(n, profile, seed, rough):
    """PathOptimizer::optimizePath's two QPs (path_optimizer.cpp:124-161): around (0, 0, k_ref), then around the first optimum - each against HiGHS."""
    b = make_batch(2, n, profile) if seed is None else make_batch(2, n, profile, seed=seed)
    prm = O.PathQpParams()
    if rough:
        prm.rough_constraints_far_away = True
        prm.precise_planning_length = 12.0
    for q in range(2 if n <= 80 else 1):
        ref, bounds, scal = b["ref"][q], b["bounds"][q], b["scal"][q]
        lin = O.first_linearization(ref)
        for _ in range(2):
            Pd, A, lo, up, sz = O.assemble_path_qp(ref, lin, bounds, scal, prm)
            flat = np.zeros(sz["vars"], bool); flat[0:3 * n:3] = True; flat[4 * n - 1:] = True
            _, x_star = pin(Pd, np.zeros(sz["vars"]), sp.csc_matrix(A), lo, up, flat=flat)
            lin = O.unpack_path(x_star, ref)[:, 3:6].copy()


def test_smoother_qps_against_highs():
    """S1-S3 on the golden inputs (tests/golden/smoothers.npz): TensionSmoother2 (equality rows only: the regularised optimum from the dense KKT system),
    TensionSmoother and postSmooth (the oracle's ADMM)."""
    g = np.load(os.path.join(GOLD, "smoothers.npz"))
    for tag in ("a", "b"):
        n = len(g[tag + "_x"])
        P, q, A, lo, up = O.assemble_tension2(g[tag + "_x"], g[tag + "_y"], g[tag + "_angle"], g[tag + "_k"], g[tag + "_s"])
        xh, yh, _ = H.solve_qp(P, q, A, lo, up)
        m = A.shape[0]
        sol = np.linalg.solve(np.block([[P + HIGHS_REG * np.eye(len(q)), A.T], [A, np.zeros((m, m))]]), np.r_[-q, lo])
        assert np.abs(xh - sol[:len(q)]).max() < 1e-6 and np.abs(yh - sol[len(q):]).max() < 1e-6 * max(1.0, np.abs(sol[len(q):]).max())
        f = lambda z_: 0.5 * z_ @ (P @ z_) + q @ z_
        star = np.linalg.solve(np.block([[P, A.T], [A, np.zeros((m, m))]]), np.r_[-q, lo])[:len(q)]
        assert np.abs(star[:n] - g[tag + "_t2_x"]).max() < 1e-9                            # (the golden IS that optimum)
        assert abs(f(xh) - f(star)) < 1e-8 * max(1.0, abs(f(star)))
        assert np.abs(xh[:2 * n] - star[:2 * n]).max() < 1e-3                              # x, y of order 10 ... 50 m: 1e-5 relative, the regularisation's reach
    for tag in ("a",):
        P, q, A, lo, up = O.assemble_tension(g[tag + "_x"], g[tag + "_y"], g[tag + "_angle"], g[tag + "_clearance"])
        n = len(g[tag + "_x"])
        xh, _ = pin(P, q, sp.csc_matrix(A), lo, up, flat=np.ones(3 * n, bool), tol_x=1e-6, tol_y=1e-5)
        assert np.abs(xh[:n] - g[tag + "_t_x"]).max() < 2e-4 and np.abs(xh[n:2 * n] - g[tag + "_t_y"]).max() < 2e-4
    for tag in ("c", "d"):
        s, lb, ub, l0 = g[tag + "_s"], g[tag + "_lb"], g[tag + "_ub"], float(g[tag + "_l0"])
        P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
        xh, _ = pin(P, q, sp.csc_matrix(A), lo, up, tol_x=1e-6, tol_y=1e-5)
        assert np.abs(xh[:len(s)] - g[tag + "_l"]).max() < 1e-6


@pytest.mark.parametrize("n,profile,batch,rough", [(80, "uniform", 1024, False), (120, "varied", 256, False), (80, "varied", 64, True)])
def test_the_device_algorithms_on_the_host_against_highs(n, profile, batch, rough):
    """Both path solvers' device sources compiled for the host (tests/emu: the lane-per-waypoint ADMM + KKT-verified polish, the lane-per-QP interior-point +
    active-set rounds), both passes of optimizePath, against HiGHS on the reference's QP around the same linearisation point - what
    tests/test_gpu_highs_pin.py then repeats with the HIP kernels through the C ABI."""
    import emu_util as EL
    import lq_emu_util as E
    from highs_util import against_highs
    b = make_batch(batch, n, profile)
    idx = np.linspace(0, batch - 1, 3).astype(int)
    sub = {k: np.ascontiguousarray(v[idx]) for k, v in b.items()}
    prm, over = None, {}
    if rough:
        prm = O.PathQpParams(); prm.rough_constraints_far_away = True; prm.precise_planning_length = 12.0
        over = dict(rough_constraints_far_away=1, precise_planning_length=12.0)
    runs = {"lane_per_qp": lambda passes: E.solve(sub["ref"], sub["bounds"], sub["scal"], passes=passes, prm=E.production(**over)),
            "lane_per_waypoint": lambda passes: EL.solve(EL.production(**over), sub["ref"], sub["bounds"], sub["scal"], passes=passes)}
    for name, run in runs.items():
        r0, r1 = run(0), run(1)
        assert (r0["status"] == 1).all() and (r1["status"] == 1).all(), name
        for q in range(len(idx)):
            ref, bounds, scal = sub["ref"][q], sub["bounds"][q], sub["scal"][q]
            against_highs(ref, O.first_linearization(ref), bounds, scal, r0["out"][q], prm)
            against_highs(ref, r0["out"][q][:, 3:6], bounds, scal, r1["out"][q], prm)
