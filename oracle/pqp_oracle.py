"""CPU restatement (numpy/scipy) of the reference path-QP hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (LiJiangnanBit/path_optimizer_2) holds no tests, golden vectors or
fixtures for this path, and its ADMM arithmetic lives in the un-vendored, un-pinned third-party
solver OSQP (oxfordcontrol/osqp, cloned at HEAD by script/install_deps.sh:102; v0.6.x era) reached
through robotology/osqp-eigen (install_deps.sh:116).  Neither is present in this image.  What pins
results instead: uniqueness of the QP optimum + a solver-independent KKT certificate
(`kkt_certificate` below) + agreement of two independent formulations (this file: full quasi-definite
KKT via sparse LU; oracle/pqp_oracle.c: reduced banded Cholesky).
Round 6: the OPTIMUM (not OSQP's iterates) is pinned against a third-party solver - HiGHS's QP solver as bundled
with scipy, oracle/highs_qp.py, tests/test_highs_pin.py: every variable of every QP assembled here to 2e-7.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing in the product path (path_optimizer_2_amd/, include/) may.

What each function follows (file:line relative to /root/reference):
  path_qp_sizes        src/solver/base_solver.cpp:15-39
  assemble_path_qp     src/solver/base_solver.cpp:119-148 (setCost), :150-261 (setConstraints),
                       :290-295 (getSoftBounds)
  unpack_path          src/solver/base_solver.cpp:263-288 (getOptimizedPath)
  constrain_angle      include/tools/tools.hpp:24-35
  first_linearization  src/path_optimizer.cpp:128-137
  solve_path           src/path_optimizer.cpp:138-153 + base_solver.cpp:56-117
  osqp_admm            the OSQP paper (Stellato et al., Math. Prog. Comp. 2020) with the documented
                       v0.6 defaults; written from the paper, not from source.
  assemble_tension2 / assemble_tension / assemble_post
                       src/reference_path_smoother/tension_smoother_2.cpp:74-158,
                       tension_smoother.cpp:102-177, reference_path_smoother.cpp:582-636
"""
import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

OSQP_INFTY = 1e30          # OsqpEigen::INFTY == OSQP_INFTY (osqp 0.6.x) [upstream, unverified here]
MIN_SCALING = 1e-4
MAX_SCALING = 1e4
RHO_MIN = 1e-6
RHO_MAX = 1e6
RHO_TOL = 1e-4
RHO_EQ_OVER_RHO_INEQ = 1e3


@dataclass
class PathQpParams:
    """The ~20 scalars the hot path reads."""
    front_length: float = 3.9                       # planning_flags.cpp:20
    rear_length: float = -1.0                       # planning_flags.cpp:18
    wheel_base: float = 2.5                         # planning_flags.cpp:16
    max_steering_angle: float = 35.0 * math.pi / 180.0   # planning_flags.cpp:22
    expected_safety_margin: float = 0.6             # planning_flags.cpp:95
    constraint_end_heading: bool = True             # planning_flags.cpp:98
    rough_constraints_far_away: bool = False        # planning_flags.cpp:112
    precise_planning_length: float = 30.0           # planning_flags.cpp:114
    weight_l: float = 0.0                           # base_solver.cpp:123
    weight_kappa: float = 20.0                      # base_solver.cpp:124
    weight_dkappa: float = 100.0                    # base_solver.cpp:125
    weight_slack: float = 10.0                      # base_solver.cpp:126
    end_l_bound: float = 1.0                        # base_solver.cpp:250-251
    end_psi_tol: float = 0.087                      # base_solver.cpp:257-258
    end_psi_max: float = 70.0 * math.pi / 180.0     # base_solver.cpp:256
    min_clearance: float = 0.1                      # base_solver.cpp:292


@dataclass
class OsqpSettings:
    eps_abs: float = 2e-3          # base_solver.cpp:61
    eps_rel: float = 2e-3          # base_solver.cpp:62
    rho: float = 0.1
    sigma: float = 1e-6
    alpha: float = 1.6
    max_iter: int = 4000
    scaling: int = 10
    adaptive_rho: bool = True
    adaptive_rho_interval: int = 100   # OSQP "auto" without profiling = 4*check_termination
    adaptive_rho_tolerance: float = 5.0
    check_termination: int = 25
    eps_prim_inf: float = 1e-4     # OSQP default
    eps_dual_inf: float = 1e-4     # OSQP default


def constrain_angle(a):
    """include/tools/tools.hpp:24-35 (recursive wrap to [-pi, pi], boundaries inclusive)."""
    while True:
        if a > math.pi:
            a -= 2 * math.pi
        elif a < -math.pi:
            a += 2 * math.pi
        else:
            return a


def path_qp_sizes(n, s_list, prm):
    """base_solver.cpp:15-39.  Returns dict(state, control, precise, slack, vars, cons)."""
    state = 3 * n
    control = n - 1
    precise = n
    if prm.rough_constraints_far_away:
        # std::lower_bound on input_path[i].s < precise_planning_length (base_solver.cpp:26-33)
        precise = int(np.searchsorted(np.asarray(s_list), prm.precise_planning_length, side="left"))
    slack = precise + n
    nvars = state + control + slack
    ncons = 4 * n + precise + n + 2
    return dict(n=n, state=state, control=control, precise=precise, slack=slack, vars=nvars, cons=ncons)


def soft_bounds(lb, ub, margin, min_clearance=0.1):
    """base_solver.cpp:290-295."""
    clearance = ub - lb
    remain = max(min_clearance, clearance - 2 * margin)
    shrink = max(0.0, (clearance - remain) / 2.0)
    return lb + shrink, ub - shrink


def assemble_path_qp(ref, lin, bounds, scal, prm=None):
    """Dense restatement of setCost + setConstraints in the REFERENCE variable/row order.

    ref    [n][5]  s, k, heading, x, y           (ReferencePath::getReferenceStates)
    lin    [n][3]  l, d_heading, k               (input_path_: the linearisation point)
    bounds [n][6]  front lb,ub, rear lb,ub, center lb,ub  (ReferencePath::getBounds)
    scal   [6]     init_err[0], init_err[1], start.k, target.heading, blocked(0/1), max_steering_angle
    Returns (Pdiag[vars], A[cons][vars] dense, l[cons], u[cons], sizes)
    """
    prm = prm or PathQpParams()
    ref = np.asarray(ref, dtype=np.float64)
    lin = np.asarray(lin, dtype=np.float64)
    bounds = np.asarray(bounds, dtype=np.float64)
    n = ref.shape[0]
    sz = path_qp_sizes(n, ref[:, 0], prm)
    state, control, P_, nvars, ncons = sz["state"], sz["control"], sz["precise"], sz["vars"], sz["cons"]
    # ---- setCost (:119-148)
    Pd = np.zeros(nvars)
    for i in range(n):
        Pd[3 * i] += prm.weight_l
        Pd[3 * i + 2] += prm.weight_kappa
        if i < P_:
            Pd[state + control + 2 * i] += prm.weight_slack
            Pd[state + control + 2 * i + 1] += prm.weight_slack
        else:
            Pd[state + control + 2 * P_ + (i - P_)] += prm.weight_slack
        if i != n - 1:
            Pd[state + i] += prm.weight_dkappa
    # ---- setConstraints (:150-210)
    trans_idx = 0
    kappa_idx = trans_idx + 3 * n
    precise_idx = kappa_idx + n
    rough_idx = precise_idx + 2 * P_
    end_idx = rough_idx + n - P_
    A = np.zeros((ncons, nvars))
    for i in range(state):
        A[i, i] = -1.0
    c_list = []
    for i in range(n - 1):
        xl, xpsi, xk = lin[i]
        xk_next = lin[i + 1, 2]
        df_x = np.array([
            [-xk * math.tan(xpsi), (1 - xk * xl) / math.cos(xpsi) ** 2, 0.0],
            [-xk * xk / math.cos(xpsi), (1 - xk * xl) * xk * math.tan(xpsi) / math.cos(xpsi),
             (1 - xk * xl) / math.cos(xpsi)],
            [0.0, 0.0, 0.0]])
        df_u = np.array([0.0, 0.0, 1.0])
        ds = ref[i + 1, 0] - ref[i, 0]
        Ai = ds * df_x + np.eye(3)
        Bi = ds * df_u
        A[3 * (i + 1):3 * (i + 1) + 3, 3 * i:3 * i + 3] = Ai
        A[3 * (i + 1):3 * (i + 1) + 3, state + i] = Bi
        u_in = (xk_next - xk) / ds
        f = np.array([(1 - xk * xl) * math.tan(xpsi),
                      (1 - xk * xl) * xk / math.cos(xpsi) - ref[i, 1],
                      u_in])
        xv = np.array([xl, xpsi, xk])
        c_list.append(ds * (f - df_x @ xv - df_u * u_in))
    for i in range(n):
        A[kappa_idx + i, 3 * i + 2] = 1.0
    for i in range(n):
        if i < P_:
            A[precise_idx + 2 * i, 3 * i] = 1.0
            A[precise_idx + 2 * i, 3 * i + 1] = prm.front_length
            A[precise_idx + 2 * i + 1, 3 * i] = 1.0
            A[precise_idx + 2 * i + 1, 3 * i + 1] = prm.rear_length
            A[precise_idx + 2 * i, state + control + 2 * i] = 1.0
            A[precise_idx + 2 * i + 1, state + control + 2 * i + 1] = 1.0
        else:
            li = i - P_
            A[rough_idx + li, 3 * i] = 1.0
            A[rough_idx + li, state + control + 2 * P_ + li] = 1.0
    A[end_idx, state - 3] = 1.0
    A[end_idx + 1, state - 2] = 1.0
    # ---- bounds (:212-260)
    lo = np.zeros(ncons)
    up = np.zeros(ncons)
    x0 = np.array([scal[0], scal[1], scal[2]])
    lo[0:3] = -x0
    up[0:3] = -x0
    for i in range(n - 1):
        lo[3 * (i + 1):3 * (i + 1) + 3] = -c_list[i]
        up[3 * (i + 1):3 * (i + 1) + 3] = -c_list[i]
    kappa_limit = math.tan(scal[5]) / prm.wheel_base
    lo[kappa_idx:kappa_idx + n] = -kappa_limit
    up[kappa_idx:kappa_idx + n] = kappa_limit
    m = prm.expected_safety_margin
    for i in range(n):
        if i < P_:
            lo[precise_idx + 2 * i], up[precise_idx + 2 * i] = soft_bounds(bounds[i, 0], bounds[i, 1], m, prm.min_clearance)
            lo[precise_idx + 2 * i + 1], up[precise_idx + 2 * i + 1] = soft_bounds(bounds[i, 2], bounds[i, 3], m, prm.min_clearance)
        else:
            li = i - P_
            lo[rough_idx + li], up[rough_idx + li] = soft_bounds(bounds[i, 4], bounds[i, 5], m, prm.min_clearance)
    lo[end_idx] = -prm.end_l_bound
    up[end_idx] = prm.end_l_bound
    lo[end_idx + 1] = -OSQP_INFTY
    up[end_idx + 1] = OSQP_INFTY
    if prm.constraint_end_heading and not (scal[4] != 0.0):
        end_psi = constrain_angle(scal[3] - ref[-1, 2])
        if end_psi < prm.end_psi_max:      # signed compare, no fabs (base_solver.cpp:256)
            lo[end_idx + 1] = end_psi - prm.end_psi_tol
            up[end_idx + 1] = end_psi + prm.end_psi_tol
    return Pd, A, lo, up, sz


def structural_pattern(n, precise):
    """Value-independent (row, col) list of A in CSC order (col-major, rows ascending) —
    the 17N-5 slots of SURVEY.md Appendix A generalised to P<=N — and P's diagonal columns."""
    state, control = 3 * n, n - 1
    kappa_idx = 3 * n
    precise_idx = kappa_idx + n
    rough_idx = precise_idx + 2 * precise
    end_idx = rough_idx + n - precise
    ent = []
    for r in range(state):
        ent.append((r, r))
    for i in range(n - 1):
        r0 = 3 * (i + 1)
        ent += [(r0, 3 * i), (r0, 3 * i + 1),
                (r0 + 1, 3 * i), (r0 + 1, 3 * i + 1), (r0 + 1, 3 * i + 2),
                (r0 + 2, 3 * i + 2), (r0 + 2, state + i)]
    for i in range(n):
        ent.append((kappa_idx + i, 3 * i + 2))
    for i in range(n):
        if i < precise:
            ent += [(precise_idx + 2 * i, 3 * i), (precise_idx + 2 * i, 3 * i + 1),
                    (precise_idx + 2 * i + 1, 3 * i), (precise_idx + 2 * i + 1, 3 * i + 1),
                    (precise_idx + 2 * i, state + control + 2 * i),
                    (precise_idx + 2 * i + 1, state + control + 2 * i + 1)]
        else:
            li = i - precise
            ent += [(rough_idx + li, 3 * i), (rough_idx + li, state + control + 2 * precise + li)]
    ent += [(end_idx, state - 3), (end_idx + 1, state - 2)]
    ent.sort(key=lambda rc: (rc[1], rc[0]))
    rows = np.array([e[0] for e in ent], dtype=np.int32)
    cols = np.array([e[1] for e in ent], dtype=np.int32)
    nvars = state + control + precise + n
    colptr = np.zeros(nvars + 1, dtype=np.int32)
    for c in cols:
        colptr[c + 1] += 1
    colptr = np.cumsum(colptr).astype(np.int32)
    pcols = []
    for i in range(n):
        pcols.append(3 * i + 2)
    for i in range(n - 1):
        pcols.append(state + i)
    for i in range(precise + n):
        pcols.append(state + control + i)
    return rows, cols, colptr, np.array(sorted(pcols), dtype=np.int32)


def first_linearization(ref):
    """path_optimizer.cpp:128-137: l = d_heading = 0, k = k_ref."""
    ref = np.asarray(ref)
    lin = np.zeros((ref.shape[0], 3))
    lin[:, 2] = ref[:, 1]
    return lin


def unpack_path(x, ref):
    """base_solver.cpp:263-288.  out [n][7] = x, y, heading, l, d_heading, k, d_k."""
    ref = np.asarray(ref)
    n = ref.shape[0]
    out = np.zeros((n, 7))
    for i in range(n):
        angle = ref[i, 2]
        out[i, 2] = constrain_angle(angle + x[3 * i + 1])
        out[i, 4] = x[3 * i + 1]
        out[i, 3] = x[3 * i]
        new_angle = constrain_angle(angle + math.pi / 2)
        out[i, 0] = ref[i, 3] + x[3 * i] * math.cos(new_angle)
        out[i, 1] = ref[i, 4] + x[3 * i] * math.sin(new_angle)
        out[i, 5] = x[3 * i + 2]
        if i < n - 1:
            out[i, 6] = x[3 * n + i]
    return out


# ------------------------------------------------------------------------------------------------
# OSQP-paper ADMM (full quasi-definite KKT formulation, sparse LU).
# ------------------------------------------------------------------------------------------------
def _limit_scaling(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.minimum(v, MAX_SCALING)


def ruiz_scale(P, q, A, l, u, passes):
    """Modified Ruiz equilibration (paper Alg. 2). P is full symmetric."""
    n, m = P.shape[0], A.shape[0]
    D = np.ones(n)
    E = np.ones(m)
    c = 1.0
    P = sp.csc_matrix(P, dtype=np.float64, copy=True)
    A = sp.csc_matrix(A, dtype=np.float64, copy=True)
    q = np.array(q, dtype=np.float64)
    for _ in range(passes):
        pn = np.asarray(abs(P).max(axis=0).todense()).ravel() if P.nnz else np.zeros(n)
        an = np.asarray(abs(A).max(axis=0).todense()).ravel() if A.nnz else np.zeros(n)
        en = np.asarray(abs(A).max(axis=1).todense()).ravel() if A.nnz else np.zeros(m)
        Dt = 1.0 / np.sqrt(_limit_scaling(np.maximum(pn, an)))
        Et = 1.0 / np.sqrt(_limit_scaling(en))
        P = sp.diags(Dt) @ P @ sp.diags(Dt)
        A = sp.diags(Et) @ A @ sp.diags(Dt)
        q = Dt * q
        D *= Dt
        E *= Et
        pn = np.asarray(abs(sp.csc_matrix(P)).max(axis=0).todense()).ravel() if P.nnz else np.zeros(n)
        c_tmp = float(np.mean(pn))
        qn = float(np.max(np.abs(q))) if n else 0.0
        qn = 1.0 if qn < MIN_SCALING else min(qn, MAX_SCALING)
        c_tmp = max(c_tmp, qn)
        c_tmp = 1.0 if c_tmp < MIN_SCALING else min(c_tmp, MAX_SCALING)
        ct = 1.0 / c_tmp
        P = P * ct
        q = q * ct
        c *= ct
    return sp.csc_matrix(P), q, sp.csc_matrix(A), E * l, E * u, D, E, c


def _rho_vec(l, u, rho):
    rv = np.full(l.shape, rho)
    free = (l < -OSQP_INFTY * MIN_SCALING) & (u > OSQP_INFTY * MIN_SCALING)
    eq = (~free) & ((u - l) < RHO_TOL)
    rv[free] = RHO_MIN
    rv[eq] = RHO_EQ_OVER_RHO_INEQ * rho
    return rv


def osqp_admm(P, q, A, l, u, st=None, warm_x=None, warm_y=None, rho_init=None):
    """OSQP-paper ADMM.  P: full symmetric (sparse or dense), A: (m,n).  Returns dict."""
    st = st or OsqpSettings()
    P = sp.csc_matrix(P)
    A = sp.csc_matrix(A)
    n, m = P.shape[0], A.shape[0]
    l = np.maximum(np.asarray(l, dtype=np.float64), -OSQP_INFTY)
    u = np.minimum(np.asarray(u, dtype=np.float64), OSQP_INFTY)
    if st.scaling:
        Ps, qs, As, ls, us, D, E, c = ruiz_scale(P, q, A, l, u, st.scaling)
    else:
        Ps, qs, As, ls, us = P, np.array(q, dtype=np.float64), A, l.copy(), u.copy()
        D, E, c = np.ones(n), np.ones(m), 1.0
    Dinv, Einv, cinv = 1.0 / D, 1.0 / E, 1.0 / c
    rho = st.rho if rho_init is None else rho_init

    def factor(rho):
        rv = _rho_vec(ls, us, rho)
        K = sp.bmat([[Ps + st.sigma * sp.eye(n), As.T], [As, -sp.diags(1.0 / rv)]], format="csc")
        return rv, spla.splu(K)

    rv, lu = factor(rho)
    x = np.zeros(n)
    z = np.zeros(m)
    y = np.zeros(m)
    if warm_x is not None:
        x = Dinv * np.asarray(warm_x, dtype=np.float64)
        z = As @ x
    if warm_y is not None:
        y = c * Einv * np.asarray(warm_y, dtype=np.float64)
    status = "max_iter"
    n_refactor = 0
    it = 0
    pri = dua = float("nan")
    inf_u = us > OSQP_INFTY * MIN_SCALING
    inf_l = ls < -OSQP_INFTY * MIN_SCALING
    for it in range(1, st.max_iter + 1):
        x_prev, y_prev = x, y
        rhs = np.concatenate([st.sigma * x - qs, z - y / rv])
        sol = lu.solve(rhs)
        xt = sol[:n]
        zt = z + (sol[n:] - y) / rv
        x = st.alpha * xt + (1 - st.alpha) * x
        zh = st.alpha * zt + (1 - st.alpha) * z
        z_new = np.clip(zh + y / rv, ls, us)
        y = y + rv * (zh - z_new)
        z = z_new
        check = st.check_termination and it % st.check_termination == 0
        adapt = st.adaptive_rho and st.adaptive_rho_interval and it % st.adaptive_rho_interval == 0
        if check or adapt:
            Ax = As @ x
            Px = Ps @ x
            Aty = As.T @ y
            pri = np.max(np.abs(Einv * (Ax - z))) if m else 0.0
            dua = cinv * np.max(np.abs(Dinv * (Px + qs + Aty)))
            n_ax = np.max(np.abs(Einv * Ax)) if m else 0.0
            n_z = np.max(np.abs(Einv * z)) if m else 0.0
            n_px = cinv * np.max(np.abs(Dinv * Px))
            n_aty = cinv * np.max(np.abs(Dinv * Aty))
            n_q = cinv * np.max(np.abs(Dinv * qs))
            if check:
                eps_p = st.eps_abs + st.eps_rel * max(n_ax, n_z)
                eps_d = st.eps_abs + st.eps_rel * max(n_px, n_aty, n_q)
                if pri <= eps_p and dua <= eps_d:
                    status = "solved"
                    break
                # infeasibility certificates (OSQP paper section 3.4; osqp/src/auxil.c is_primal_infeasible / is_dual_infeasible)
                # on the difference of the last two iterates, norms unscaled
                dy = y - y_prev
                dy = np.where(inf_u & inf_l, 0.0, np.where(inf_u, np.minimum(dy, 0.0), np.where(inf_l, np.maximum(dy, 0.0), dy)))
                n_dy = np.max(np.abs(E * dy)) if m else 0.0
                if n_dy > st.eps_prim_inf:
                    lhs = float(np.sum(np.where(dy > 0, us * dy, 0.0)) + np.sum(np.where(dy < 0, ls * dy, 0.0)))
                    if lhs < -st.eps_prim_inf * n_dy and np.max(np.abs(Dinv * (As.T @ dy))) < st.eps_prim_inf * n_dy:
                        status = "primal_infeasible"
                        break
                dx = x - x_prev
                n_dx = np.max(np.abs(D * dx))
                if n_dx > st.eps_dual_inf and float(qs @ dx) < -c * st.eps_dual_inf * n_dx:
                    if np.max(np.abs(Dinv * (Ps @ dx))) < c * st.eps_dual_inf * n_dx:
                        Adx = Einv * (As @ dx)
                        tol = st.eps_dual_inf * n_dx
                        ok = np.all(np.where(inf_u, True, Adx < tol) & np.where(inf_l, True, Adx > -tol))
                        if ok:
                            status = "dual_infeasible"
                            break
            if adapt:
                pn = pri / (max(n_ax, n_z) + 1e-10)
                dn = dua / (max(n_px, n_aty, n_q) + 1e-10)
                rho_new = float(np.clip(rho * math.sqrt(pn / (dn + 1e-10)), RHO_MIN, RHO_MAX))
                if rho_new > rho * st.adaptive_rho_tolerance or rho_new < rho / st.adaptive_rho_tolerance:
                    rho = rho_new
                    rv, lu = factor(rho)
                    n_refactor += 1
    return dict(x=D * x, y=cinv * E * y, z=Einv * z, iters=it, status=status, rho=rho,
                pri_res=float(pri), dua_res=float(dua), refactors=n_refactor)


def kkt_certificate(P, q, A, l, u, x, y):
    """Solver-independent optimality certificate:
    primal infeasibility, stationarity, and complementarity of y against the bounds."""
    P = sp.csc_matrix(P)
    A = sp.csc_matrix(A)
    Ax = A @ x
    pri = float(np.max(np.maximum(0.0, np.maximum(l - Ax, Ax - u))))
    stat = float(np.max(np.abs(P @ x + q + A.T @ y)))
    # y_i > 0 only if Ax_i at upper bound; y_i < 0 only if at lower bound
    fin_u = u < OSQP_INFTY * MIN_SCALING
    fin_l = l > -OSQP_INFTY * MIN_SCALING
    gap_u = np.where(fin_u, np.abs(u - Ax), np.inf)
    gap_l = np.where(fin_l, np.abs(Ax - l), np.inf)
    comp_u = np.where(y > 0, np.minimum(y, gap_u), 0.0)
    comp_l = np.where(y < 0, np.minimum(-y, gap_l), 0.0)
    comp = float(max(np.max(comp_u), np.max(comp_l)))
    return dict(pri=pri, stat=stat, comp=comp)


def solve_path(ref, bounds, scal, prm=None, st=None, passes=1, lin0=None):
    """PathOptimizer::optimizePath (path_optimizer.cpp:124-161): cold solve around (0,0,k_ref),
    then `passes` re-linearised warm solves (reference: 1).  Returns dict with per-pass results."""
    prm = prm or PathQpParams()
    st = st or OsqpSettings()
    ref = np.asarray(ref, dtype=np.float64)
    n = ref.shape[0]
    lin = first_linearization(ref) if lin0 is None else np.asarray(lin0, dtype=np.float64)
    res = []
    warm_x = warm_y = None
    rho = None
    for p in range(passes + 1):
        Pd, A, lo, up, sz = assemble_path_qp(ref, lin, bounds, scal, prm)
        r = osqp_admm(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, st, warm_x, warm_y, rho)
        out = unpack_path(r["x"], ref)
        r["out"] = out
        r["qp"] = (Pd, A, lo, up)
        res.append(r)
        if r["status"] != "solved":       # solve() == false: optimizePath returns (path_optimizer.cpp:143-146,154-157)
            break
        lin = out[:, 3:6].copy()          # input_path_ = first solution (base_solver.cpp:100)
        warm_x, warm_y, rho = r["x"], r["y"], r["rho"]
    return res


# ------------------------------------------------------------------------------------------------
# Smoother QPs, reference variable order.
# ------------------------------------------------------------------------------------------------
def assemble_tension2(x_list, y_list, angle_list, k_list, s_list, w_dev=0.005, w_k=1.0, w_dk=10.0):
    """tension_smoother_2.cpp:74-158.  Returns (P dense, q, A dense, l, u)."""
    n = len(x_list)
    nv, nc = 4 * n - 1, 3 * (n - 1) + 2
    xs, ys, ts, ks = 0, n, 2 * n, 3 * n
    P = np.zeros((nv, nv))
    for i in range(n):
        P[xs + i, xs + i] = P[ys + i, ys + i] = w_dev * 2
        if i != n - 1:
            P[ks + i, ks + i] = w_k * 2
    coeff = np.array([[1.0, -1.0], [-1.0, 1.0]])
    for i in range(n - 2):
        P[ks + i:ks + i + 2, ks + i:ks + i + 2] += 2 * w_dk * coeff
    q = np.zeros(nv)
    for i in range(n):
        q[xs + i] = -2 * w_dev * x_list[i]
        q[ys + i] = -2 * w_dev * y_list[i]
    A = np.zeros((nc, nv))
    lo = np.zeros(nc)
    cx, cy, ct = 0, n - 1, 2 * (n - 1)
    cx0, cy0 = 3 * (n - 1), 3 * (n - 1) + 1
    for i in range(n - 1):
        ds = s_list[i + 1] - s_list[i]
        A[cx + i, xs + i + 1] = A[cy + i, ys + i + 1] = A[ct + i, ts + i + 1] = 1
        A[cx + i, xs + i] = A[cy + i, ys + i] = A[ct + i, ts + i] = -1
        A[cx + i, ts + i] = ds * math.sin(angle_list[i])
        A[cy + i, ts + i] = -ds * math.cos(angle_list[i])
        A[ct + i, ks + i] = -ds
        lo[cx + i] = ds * math.cos(angle_list[i])
        lo[cy + i] = ds * math.sin(angle_list[i])
        lo[ct + i] = -ds * k_list[i]
    A[cx0, xs] = A[cy0, ys] = 1
    lo[cx0] = x_list[0]
    lo[cy0] = y_list[0]
    return P, q, A, lo, lo.copy()


def assemble_tension(x_list, y_list, angle_list, clearance, w_k=1.0, w_dk=50.0, w_dev=0.0):
    """tension_smoother.cpp:102-177.  `clearance[i]` = Map::getObstacleDistance at point i
    (the map lookup itself is out of scope)."""
    n = len(x_list)
    xs, ys, dsx = 0, n, 2 * n
    P = np.zeros((3 * n, 3 * n))
    dds = np.outer([1, -2, 1], [1, -2, 1]) * w_k
    ddds = np.outer([-1, 3, -3, 1], [-1, 3, -3, 1]) * w_dk
    for i in range(n - 2):
        P[xs + i:xs + i + 3, xs + i:xs + i + 3] += dds
        P[ys + i:ys + i + 3, ys + i:ys + i + 3] += dds
        if i != n - 3:
            P[xs + i:xs + i + 4, xs + i:xs + i + 4] += ddds
            P[ys + i:ys + i + 4, ys + i:ys + i + 4] += ddds
    for i in range(n):
        P[dsx + i, dsx + i] = w_dev
    A = np.zeros((3 * n, 3 * n))
    lo = np.zeros(3 * n)
    up = np.zeros(3 * n)
    for i in range(n):
        A[xs + i, xs + i] = A[ys + i, ys + i] = 1
        th = angle_list[i] + math.pi / 2
        A[xs + i, dsx + i] = -math.cos(th)
        A[ys + i, dsx + i] = -math.sin(th)
        A[dsx + i, dsx + i] = 1
        lo[xs + i] = up[xs + i] = x_list[i]
        lo[ys + i] = up[ys + i] = y_list[i]
    lo[dsx] = up[dsx] = 0
    lo[dsx + n - 1], up[dsx + n - 1] = -0.5, 0.5
    for i in range(1, n - 1):
        c = min(clearance[i], 2.0)
        lo[dsx + i], up[dsx + i] = -c, c
    return P, np.zeros(3 * n), A, lo, up


def assemble_post(layers_s, layers_bounds, vehicle_l):
    """reference_path_smoother.cpp:582-636."""
    n = len(layers_s)
    P = np.zeros((3 * n, 3 * n))
    for i in range(n):
        P[i, i] = 1.0
        P[n + i, n + i] = 100.0
        P[2 * n + i, 2 * n + i] = 1000.0
    A = np.zeros((3 * n - 2, 3 * n))
    for i in range(n):
        A[i, i] = 1
    for i in range(n - 1):
        d = layers_s[i + 1] - layers_s[i]
        A[n + i, i + 1] = 1
        A[n + i, i] = -1
        A[n + i, n + i] = -d
        A[2 * n - 1 + i, n + i + 1] = 1
        A[2 * n - 1 + i, n + i] = -1
        A[2 * n - 1 + i, 2 * n + i] = -d
    lo = np.zeros(3 * n - 2)
    up = np.zeros(3 * n - 2)
    lo[0] = up[0] = vehicle_l
    for i in range(1, n):
        lo[i], up[i] = layers_bounds[i]
    return P, np.zeros(3 * n), A, lo, up
