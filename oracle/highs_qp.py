"""A THIRD-PARTY solver for the oracle's QPs: HiGHS (its active-set QP solver), as bundled with scipy >= 1.15 (scipy.optimize._highspy._core - the
pybind11 module scipy's linprog drives; nothing is installed for it).  TEST INFRASTRUCTURE like the rest of oracle/: it pins the optimum of the QPs the
oracle assembles line by line from the reference (base_solver.cpp:119-261, tension_smoother_2.cpp:74-158, tension_smoother.cpp:102-177,
reference_path_smoother.cpp:582-636) against a solver nobody here wrote.  It says nothing about OSQP's ITERATES at eps 2e-3 (upstream OSQP is not in this image:
tests/test_upstream_osqp.py); it says that the unique optimum those iterates converge to is the point the oracle - and the HIP kernels - return.

    min 1/2 x' P x + q' x   s.t.  l <= A x <= u        (OSQP's form, osqp.h)
"""
import numpy as np
import scipy.sparse as sp

INF = 1e30          # OSQP_INFTY: a bound beyond this is no bound


def available():
    try:
        from scipy.optimize._highspy import _core  # noqa: F401
        return hasattr(_core, "_Highs") and hasattr(_core, "HighsHessian")
    except Exception:
        return False


def version():
    from scipy.optimize._highspy import _core as c
    import scipy
    return f"HiGHS {c._Highs().githash()} (scipy {scipy.__version__})"


def _solve_once(P, q, A, lo, up, tol, time_limit=None):
    from scipy.optimize._highspy import _core as c
    m, n = A.shape
    Pl = sp.csc_matrix(sp.tril(P))
    Pl.sort_indices(); A.sort_indices()
    inf = c.kHighsInf
    lp = c.HighsLp()
    lp.num_col_, lp.num_row_ = n, m
    lp.col_cost_ = np.asarray(q, dtype=np.float64)
    lp.col_lower_ = np.full(n, -inf); lp.col_upper_ = np.full(n, inf)
    lp.row_lower_ = np.where(lo <= -INF, -inf, lo); lp.row_upper_ = np.where(up >= INF, inf, up)
    lp.a_matrix_.format_ = c.MatrixFormat.kColwise
    lp.a_matrix_.num_col_, lp.a_matrix_.num_row_ = n, m
    lp.a_matrix_.start_ = A.indptr.astype(np.int32); lp.a_matrix_.index_ = A.indices.astype(np.int32); lp.a_matrix_.value_ = A.data.astype(np.float64)
    hes = c.HighsHessian()
    hes.dim_ = n; hes.format_ = c.HessianFormat.kTriangular
    hes.start_ = Pl.indptr.astype(np.int32); hes.index_ = Pl.indices.astype(np.int32); hes.value_ = Pl.data.astype(np.float64)
    model = c.HighsModel(); model.lp_ = lp; model.hessian_ = hes
    h = c._Highs()
    h.setOptionValue("output_flag", False)
    h.setOptionValue("primal_feasibility_tolerance", tol); h.setOptionValue("dual_feasibility_tolerance", tol)
    if time_limit is not None:
        h.setOptionValue("time_limit", float(time_limit))          # (an ordering its active-set solver cycles on: status "Time limit reached", the next ordering)
    if h.passModel(model) == c.HighsStatus.kError:          # (kWarning: e.g. coefficients below its small-matrix-value threshold were dropped)
        raise RuntimeError("HiGHS refused the model")
    h.run()
    status = h.modelStatusToString(h.getModelStatus())
    if status != "Optimal":
        return None, None, status
    sol = h.getSolution()
    # HiGHS: row dual = d objective / d row activity bound (>= 0 at a lower bound for a minimisation); OSQP's y is its negative
    return np.array(sol.col_value), -np.array(sol.row_dual), h.getObjectiveValue()


def solve_qp(P, q, A, lo, up, tol=1e-9, tries=6, time_limit=None):
    """Returns (x, row duals y in OSQP's sign convention, objective).  P: dense / sparse symmetric PSD or a 1-D diagonal; A: dense / sparse.
    HiGHS's active-set QP solver now and then calls a point "Optimal" that misses two equality rows by ~5e-5 (one in ten path QPs in the reference's
    row / column order; its row activities drift over the iterations).  The point it returns is therefore checked against the rows here, and a QP it
    fails on is handed over again with its rows and columns in another (seeded) order - the same QP; every one of them then came back feasible to 1e-14."""
    A = sp.csc_matrix(A)
    m, n = A.shape
    P = sp.csc_matrix(sp.diags(np.asarray(P, dtype=np.float64)) if np.ndim(P) == 1 else sp.csc_matrix(P))
    q = np.asarray(q, dtype=np.float64); lo = np.asarray(lo, dtype=np.float64); up = np.asarray(up, dtype=np.float64)
    rng = np.random.default_rng(20260926)
    worst = float("nan")
    for t in range(tries):
        rp = np.arange(m) if t == 0 else rng.permutation(m)
        cp = np.arange(n) if t == 0 else rng.permutation(n)
        x2, y2, obj = _solve_once(sp.csc_matrix(P[cp][:, cp]), q[cp], sp.csc_matrix(A[rp][:, cp]), lo[rp], up[rp], tol, time_limit)
        if x2 is None:          # (its active-set solver gave up in this ordering - seen once in ~200 path QPs: "Unknown" - the next ordering)
            worst = obj
            continue
        x = np.empty(n); x[cp] = x2
        y = np.empty(m); y[rp] = y2
        Ax = A @ x
        worst = float(np.maximum(lo - Ax, Ax - up).max())
        if worst <= 100.0 * tol:
            return x, y, obj
    raise RuntimeError(f"HiGHS: no feasible 'Optimal' point in {tries} orderings (last: {worst if isinstance(worst, str) else 'rows missed by %.1e' % worst})")
