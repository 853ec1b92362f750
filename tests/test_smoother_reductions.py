"""The algebra behind the exact smoother kernels (csrc/pqp_smoother_kernels.inc: tension2_exact_kernel, tension_exact_kernel,
post_exact_kernel), checked on the CPU against the oracle's formulation of the reference's QPs: the eliminations leave the optimum where it
was, and the Riccati sweep solves TensionSmoother2's QP.  (The kernels themselves are checked on the GPU, tests/test_gpu_smoothers.py.)"""
import numpy as np
import scipy.sparse as sp

import pqp_oracle as O
from smoother_cases import post_inputs, post_reduced_kkt, tension_inputs, tension_kkt_certificate

TIGHT = O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=400000)


def test_post_smooth_box_qp_in_the_offsets_has_the_oracle_s_optimum():
    """reference_path_smoother.cpp:582-636 with l' and l'' eliminated through the two difference rows (the last layer's l' in closed form):
    the oracle's optimum satisfies the KKT conditions of the reduced box QP, a perturbed point does not."""
    for m, seed in ((4, 1), (5, 2), (18, 3), (40, 4)):
        s, lb, ub, l0 = post_inputs(m, seed=seed)
        if seed == 3:
            s = np.concatenate([[0.0], np.cumsum(np.random.default_rng(0).uniform(0.8, 2.2, size=m - 1))])
        P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
        assert post_reduced_kkt(s, lb, ub, l0, ref["x"][:m]) < 1e-6, m
        assert post_reduced_kkt(s, lb, ub, l0, ref["x"][:m] + 1e-3 * np.cos(np.arange(m))) > 1e-4


def test_tension_box_qp_in_the_lateral_shifts_has_the_oracle_s_optimum():
    """tension_smoother.cpp:102-177 with x = X + c d, y = Y + s d: the same for TensionSmoother's QP."""
    for n, seed in ((12, 1), (30, 2)):
        x, y, ang, k, s, cl = tension_inputs(n, seed=seed)
        P, q, A, lo, up = O.assemble_tension(x, y, ang, cl)
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=1e-11, eps_rel=1e-11, max_iter=800000))
        assert tension_kkt_certificate(x, y, ang, cl, ref["x"][:n], ref["x"][n:2 * n]) < 2e-5, n      # (the ADMM oracle's own accuracy on this QP)


def test_interior_start_of_the_exact_tension_kernel_bounds_the_factorisations():
    """tension_exact_kernel's first active set (round 3): interior-point iterations on the box QP in the lateral shifts, restated in numpy by
    tools/active_set_sweep.py.  From the split 'multiplier > slack' the exact rounds need 1-3 factorisations where OSQP's cold-start rule needs 10-35,
    and they end at the same KKT point (the rounds' own acceptance test IS the box QP's KKT test)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import active_set_sweep as A
    cold, warm = [], []
    for seed in range(1000, 1012):
        qp = A.tension_box_qp(48, seed)
        act, its = A.interior_start(*qp)
        rounds = A.active_set(*qp, cautious=0.5, act=act)
        assert 0 < rounds <= 3 and its <= 14, (seed, its, rounds)
        # the set the interior start predicts is the optimal one up to a box or two: the cold rule's first set is not
        cold.append(A.active_set(*qp, cautious=0.5)); warm.append(its + rounds)
    assert max(warm) <= 16 and np.mean(cold) > 1.3 * np.mean(warm) and max(cold) > max(warm)


def riccati_tension2(X, Y, phi, kl, s, w_dev=0.005, w_k=1.0, w_dk=10.0):
    """The sweep of tension2_exact_kernel in numpy: state (x, y, theta, previous k), control k; returns x, y."""
    n = len(X)
    X = X - X[0]; Yr = Y - Y[0]
    wd2 = 2 * w_dev
    S = np.diag([wd2, wd2, 0.0, 0.0]); v = np.array([-wd2 * X[-1], -wd2 * Yr[-1], 0.0, 0.0])
    gains = [None] * (n - 1)
    for i in range(n - 2, -1, -1):
        h = s[i + 1] - s[i]; a, b = h * np.sin(phi[i]), h * np.cos(phi[i])
        F = np.array([[1, 0, -a, 0], [0, 1, b, 0], [0, 0, 1, 0], [0, 0, 0, 0.0]]); G = np.array([0, 0, h, 1.0]); c = np.array([b, a, -h * kl[i], 0.0])
        dk = 2 * w_dk if i >= 1 else 0.0
        t = S @ c + v
        quu = 2 * w_k + dk + G @ S @ G
        qux = G @ S @ F + np.array([0, 0, 0, -dk])
        qu = G @ t
        Qxx = np.diag([wd2, wd2, 0.0, dk]) + F.T @ S @ F
        qx = np.array([-wd2 * X[i], -wd2 * Yr[i], 0.0, 0.0]) + F.T @ t
        K, kf = qux / quu, qu / quu
        gains[i] = (K, kf, F, G, c)
        S = Qxx - np.outer(qux, qux) / quu; v = qx - qux * kf
    xi = np.array([0.0, 0.0, -v[2] / S[2, 2], 0.0])
    xs, ys = [0.0], [0.0]
    for i in range(n - 1):
        K, kf, F, G, c = gains[i]
        u = -(K @ xi + kf)
        xi = F @ xi + G * u + c
        xs.append(xi[0]); ys.append(xi[1])
    return np.array(xs), np.array(ys)                        # relative to the first raw point


def test_riccati_sweep_solves_tension_smoother_2():
    """tension_smoother_2.cpp:74-158 has equality rows only, so its optimum is one linear system (the dense KKT matrix of the oracle's P, A);
    as a linear-quadratic control problem the same optimum comes out of one backward and one forward sweep."""
    for n, seed, ds in ((3, 1, 1.0), (4, 2, 1.0), (25, 3, 0.8), (90, 4, 1.0)):
        x, y, ang, k, s, _ = tension_inputs(n, seed=seed, ds=ds)
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        assert (lo == up).all()
        sol = np.linalg.solve(np.block([[P, A.T], [A, np.zeros((A.shape[0], A.shape[0]))]]), np.r_[-q, lo])
        gx, gy = riccati_tension2(x, y, ang, k, s)
        assert np.abs(gx + x[0] - sol[:n]).max() < 1e-8 and np.abs(gy + y[0] - sol[n:2 * n]).max() < 1e-8, n


def test_golden_smoother_fixture():
    """tests/golden/smoothers.npz (make_golden.py): the committed optima are what the oracle gives today, satisfy the reduced problems' KKT
    conditions, and TensionSmoother2's is what the Riccati sweep gives."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smoothers.npz"))
    for tag in ("a", "b"):
        x, y, ang, k, s, cl = (g[f"{tag}_{key}"] for key in ("x", "y", "angle", "k", "s", "clearance"))
        n = len(x)
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        sol = np.linalg.solve(np.block([[P, A.T], [A, np.zeros((A.shape[0], A.shape[0]))]]), np.r_[-q, lo])
        assert np.abs(sol[:n] - g[f"{tag}_t2_x"]).max() < 1e-9 and np.abs(sol[n:2 * n] - g[f"{tag}_t2_y"]).max() < 1e-9
        gx, gy = riccati_tension2(x, y, ang, k, s)
        assert np.abs(gx + x[0] - g[f"{tag}_t2_x"]).max() < 1e-8 and np.abs(gy + y[0] - g[f"{tag}_t2_y"]).max() < 1e-8
        assert tension_kkt_certificate(x, y, ang, cl, g[f"{tag}_t_x"], g[f"{tag}_t_y"]) < 2e-5
    for tag in ("c", "d"):
        s, lb, ub, l0, l = (g[f"{tag}_{key}"] for key in ("s", "lb", "ub", "l0", "l"))
        assert post_reduced_kkt(s, lb, ub, float(l0), l) < 1e-6
