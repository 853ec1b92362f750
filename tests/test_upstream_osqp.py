"""The one-command upstream pin (task 6): the oracle's optima against REAL OSQP.

The reference solves its QPs with OsqpEigen / OSQP (base_solver.cpp:59-64,80-89), which this image does not hold (no osqp, no
network): every solver-parity claim of this repo is "against the build's own OSQP-paper restatement" - DESIGN.md says "parity
unpinned".  Anyone with `pip install osqp` turns that into "pinned" by running

    python -m pytest tests/test_upstream_osqp.py -q

These tests are SKIPPED when `osqp` cannot be imported (as here and on the GPU box).  They need no GPU: they compare what the
oracle (oracle/pqp_oracle.py: the matrices exactly as BaseSolver assembles them, x* of the committed goldens) says with what OSQP
returns for the same matrices - settings as base_solver.cpp:59-62 sets them (warm start on, verbosity off, OSQP defaults otherwise),
eps tightened so that OSQP's answer is the optimum to 1e-7 rather than to its 2e-3."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

osqp = pytest.importorskip("osqp", reason="upstream OSQP is not installed in this image: parity stays 'unpinned' (DESIGN.md section 6)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqp_oracle as O  # noqa: E402
from path_optimizer_2_amd.synth import make_batch  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
TIGHT = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=400000)


def osqp_solve_general(P, q, A, lo, up, **over):
    """min 1/2 x' P x + q' x, lo <= A x <= up, for the smoother QPs (dense P with off-diagonals, q != 0): the call pattern of
    tension_smoother_2.cpp:32-55 / tension_smoother.cpp:61-83 / reference_path_smoother.cpp:533-558 (OSQP defaults, verbosity off, warm start on)."""
    P = sp.triu(sp.csc_matrix(np.asarray(P, dtype=np.float64)), format="csc")
    A = sp.csc_matrix(np.asarray(A, dtype=np.float64))
    settings = dict(TIGHT)
    settings.update(over)
    m = osqp.OSQP()
    lo = np.maximum(np.asarray(lo, dtype=np.float64), -1e30)
    up = np.minimum(np.asarray(up, dtype=np.float64), 1e30)
    try:
        m.setup(P=P, q=np.asarray(q, dtype=np.float64), A=A, l=lo, u=up, verbose=False, warm_start=True, polish=False, **settings)
    except TypeError:
        m.setup(P=P, q=np.asarray(q, dtype=np.float64), A=A, l=lo, u=up, verbose=False, warm_starting=True, polishing=False, **settings)
    r = m.solve()
    return np.asarray(r.x), np.asarray(r.y), r.info.status


def osqp_solve(Pd, A, lo, up, warm=None, **over):
    """One OSQP solve of min 1/2 x' diag(Pd) x s.t. lo <= A x <= up with the reference's call pattern (base_solver.cpp:59-64, 80-89:
    settings -> data -> initSolver -> solve; :106-110: updateBounds / updateLinearConstraintsMatrix -> solve on the warm solver, which is
    what `warm` = (x, y) of the previous solve stands for).  Written for the Python interface of OSQP 0.6.x and 1.x."""
    n = len(Pd)
    P = sp.triu(sp.diags(np.asarray(Pd, dtype=np.float64)), format="csc")
    A = sp.csc_matrix(A)
    q = np.zeros(n)
    settings = dict(TIGHT)
    settings.update(over)
    m = osqp.OSQP()
    lo = np.maximum(np.asarray(lo, dtype=np.float64), -1e30)       # OSQP_INFTY
    up = np.minimum(np.asarray(up, dtype=np.float64), 1e30)
    try:                      # 0.6.x names
        m.setup(P=P, q=q, A=A, l=lo, u=up, verbose=False, warm_start=True, polish=False, **settings)
    except TypeError:         # 1.x names
        m.setup(P=P, q=q, A=A, l=lo, u=up, verbose=False, warm_starting=True, polishing=False, **settings)
    if warm is not None:
        m.warm_start(x=warm[0], y=warm[1])
    r = m.solve()
    status = r.info.status
    return np.asarray(r.x), np.asarray(r.y), status


@pytest.mark.parametrize("name", ["path_n8", "path_n80"])
def test_golden_optima_are_what_osqp_returns(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    batch = g["ref"].shape[0]
    worst = 0.0
    for q in range(batch):
        Pd, A, lo, up, sz = O.assemble_path_qp(g["ref"][q], g["lin"][q], g["bounds"][q], g["scal"][q])
        # the committed matrices are these matrices (values at the structural pattern, bounds)
        np.testing.assert_allclose(np.asarray(sp.csc_matrix(A)[g["rows"], g["cols"]]).ravel(), g["a_val"][q], rtol=0, atol=1e-13)
        x, y, status = osqp_solve(Pd, A, lo, up)
        assert str(status).startswith("solved"), status
        n = g["ref"].shape[1]
        worst = max(worst, float(np.abs(x[:3 * n] - g["x_star"][q][:3 * n]).max()))
        c = O.kkt_certificate(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, x, y)
        assert max(c["pri"], c["stat"], c["comp"]) < 1e-6, c
    assert worst <= 1e-6, worst


def test_two_pass_pipeline_on_bench_scenarios_matches_osqp():
    """PathOptimizer::optimizePath on 64 of bench.py's configs[1] scenarios: cold solve around (0, 0, k_ref), re-linearise, warm
    re-solve (path_optimizer.cpp:124-161) - OSQP's final (l, d_heading) against the oracle's converged ones.  (1e-5, not the single QP's 1e-6:
    the second QP is built around the first one's solution, and two solvers' 1e-9-accurate first solutions already move its optimum by ~1e-6 on
    paths with a nearly degenerate contact - measured between two tight runs of the oracle itself.  The parity bar is 1e-4.)"""
    b = make_batch(64, 80)
    st = O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=200000)
    worst = 0.0
    for q in range(64):
        want = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=st)[-1]["out"]
        lin = O.first_linearization(b["ref"][q])
        warm = None
        for _ in range(2):
            Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin, b["bounds"][q], b["scal"][q])
            x, y, status = osqp_solve(Pd, A, lo, up, warm=warm)
            assert str(status).startswith("solved"), status
            out = O.unpack_path(x, b["ref"][q])
            lin = out[:, 3:6].copy()
            warm = (x, y)
        worst = max(worst, float(np.abs(out[:, 3:5] - want[:, 3:5]).max()))
    assert worst <= 1e-5, worst


def test_reference_setting_iteration_counts_are_in_the_oracles_range():
    """At the reference's own eps (2e-3, base_solver.cpp:61-62) OSQP and the restatement stop within a few checks of each other (OSQP's first
    rho update is wall-clock dependent with profiling on, so the counts need not be equal) and at points 1e-2-close to the optimum."""
    b = make_batch(16, 80)
    st = O.OsqpSettings()
    for q in range(16):
        lin = O.first_linearization(b["ref"][q])
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin, b["bounds"][q], b["scal"][q])
        mine = O.osqp_admm(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, st)
        x, y, status = osqp_solve(Pd, A, lo, up, eps_abs=2e-3, eps_rel=2e-3, max_iter=4000)
        assert str(status).startswith("solved"), status
        assert float(np.abs(x[:240] - mine["x"][:240]).max()) < 5e-2



@pytest.mark.parametrize("tag,n", [("a", 24), ("b", 60)])
def test_smoother_goldens_s1_s2_are_what_osqp_returns(tag, n):
    """tests/golden/smoothers.npz, rows S1 (TensionSmoother2, tension_smoother_2.cpp:74-158) and S2 (TensionSmoother, tension_smoother.cpp:102-177): the
    committed optima (the fixtures the GPU kernels are tested against) are what OSQP returns for the oracle's assembly of the reference's QP."""
    g = np.load(os.path.join(GOLDEN, "smoothers.npz"))
    x, y, ang, k, s, cl = (g[f"{tag}_{key}"] for key in ("x", "y", "angle", "k", "s", "clearance"))
    P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
    sol, _, status = osqp_solve_general(P, q, A, lo, up)
    assert str(status).startswith("solved"), status
    assert np.abs(sol[:n] - g[f"{tag}_t2_x"]).max() <= 1e-6 and np.abs(sol[n:2 * n] - g[f"{tag}_t2_y"]).max() <= 1e-6
    P2, q2, A2, lo2, up2 = O.assemble_tension(x, y, ang, cl)
    # (S2 has no deviation weight - cartesian_deviation_weight = 0, planning_flags.cpp:55: the optimum is flat along the line's tail, so a residual of
    #  1e-9 is 3e-5 in the points; the fixture was run to 1e-11.  The same setting here; the bar is the parity bar's decade below 1e-4)
    sol2, _, status2 = osqp_solve_general(P2, q2, A2, lo2, up2, eps_abs=1e-11, eps_rel=1e-11, max_iter=800000)
    assert str(status2).startswith("solved"), status2
    assert np.abs(sol2[:n] - g[f"{tag}_t_x"]).max() <= 1e-5 and np.abs(sol2[n:2 * n] - g[f"{tag}_t_y"]).max() <= 1e-5


@pytest.mark.parametrize("tag,m", [("c", 18), ("d", 60)])
def test_smoother_goldens_s3_are_what_osqp_returns(tag, m):
    """row S3 (postSmooth, reference_path_smoother.cpp:582-636): the committed lateral offsets of the DP corridor's QP."""
    g = np.load(os.path.join(GOLDEN, "smoothers.npz"))
    s, lb, ub, l0 = g[f"{tag}_s"], g[f"{tag}_lb"], g[f"{tag}_ub"], float(g[f"{tag}_l0"])
    P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
    sol, _, status = osqp_solve_general(P, q, A, lo, up)
    assert str(status).startswith("solved"), status
    assert np.abs(sol[:m] - g[f"{tag}_l"]).max() <= 1e-6


def test_smoothers_at_the_references_own_eps():
    """the reference runs the smoother QPs at OSQP's default eps 1e-3 (tension_smoother_2.cpp:32-36 sets only verbosity and warm start): OSQP stops there
    within 1e-2 of the committed optimum"""
    g = np.load(os.path.join(GOLDEN, "smoothers.npz"))
    x, y, ang, k, s = (g[f"b_{key}"] for key in ("x", "y", "angle", "k", "s"))
    P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
    sol, _, status = osqp_solve_general(P, q, A, lo, up, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000)
    assert str(status).startswith("solved"), status
    assert np.abs(sol[:60] - g["b_t2_x"]).max() < 1e-2 and np.abs(sol[60:120] - g["b_t2_y"]).max() < 1e-2
