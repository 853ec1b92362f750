"""Synthetic inputs for the three smoother QPs (shared by CPU and GPU tests)."""
import numpy as np


def tension_inputs(n, seed=0, ds=1.0):
    """A noisy curved polyline re-sampled at ~1 m (what segmentRawReference hands to osqpSmooth,
    reference_path_smoother.cpp:47-105): x, y, heading, curvature, arclength lists."""
    rng = np.random.default_rng(seed)
    s = np.arange(n) * ds
    k = 0.05 * np.sin(s / 9.0 + rng.uniform(0, 6.28)) + rng.uniform(-0.01, 0.01)
    ang = np.concatenate([[0.3], 0.3 + np.cumsum(0.5 * (k[1:] + k[:-1]) * ds)])
    x = np.concatenate([[1.0], 1.0 + np.cumsum(np.cos(0.5 * (ang[1:] + ang[:-1])) * ds)]) + rng.normal(scale=0.05, size=n)
    y = np.concatenate([[-2.0], -2.0 + np.cumsum(np.sin(0.5 * (ang[1:] + ang[:-1])) * ds)]) + rng.normal(scale=0.05, size=n)
    clearance = rng.uniform(0.3, 3.0, size=n)
    return x, y, ang, k, s, clearance


def post_inputs(m, seed=0):
    rng = np.random.default_rng(seed)
    s = np.arange(m) * 1.5
    centre = 0.8 * np.sin(s / 11.0 + rng.uniform(0, 6.28))
    half = rng.uniform(0.4, 1.5, size=m)
    lb, ub = centre - half, centre + half
    return s, lb, ub, float(centre[0] + rng.uniform(-0.2, 0.2))


def tension_kkt_certificate(x_list, y_list, ang, cl, gx, gy):
    """Solver-free optimality check of a TensionSmoother result: eliminate the oracle's equality rows (x = X + c d, y = Y + s d), and test
    the box QP's KKT conditions in d - feasibility, stationarity of the free shifts, the gradient's sign at the active ones - with the
    oracle's P.  Returns the largest violation relative to 1 + |d|max."""
    import pqp_oracle as O           # (imported here: tools/ share the input generators above and must not depend on oracle/)
    n = len(x_list)
    P, q, A, lo, up = O.assemble_tension(x_list, y_list, ang, cl)
    c, s = -np.diag(A[:n, 2 * n:]), -np.diag(A[n:2 * n, 2 * n:])
    d = (gx - x_list) * c + (gy - y_list) * s                       # (c, s) is a unit vector
    assert np.abs(gx - x_list - c * d).max() < 1e-9 and np.abs(gy - y_list - s * d).max() < 1e-9       # the equality rows hold
    H = P[:n, :n]
    g = c * (H @ gx) + s * (H @ gy) + np.diag(P[2 * n:, 2 * n:]) * d
    dl, du = lo[2 * n:], up[2 * n:]
    viol = 0.0
    for i in range(n):
        if dl[i] == du[i]:
            viol = max(viol, abs(d[i] - dl[i])); continue
        viol = max(viol, dl[i] - d[i], d[i] - du[i])
        if d[i] <= dl[i] + 1e-9: viol = max(viol, -g[i])
        elif d[i] >= du[i] - 1e-9: viol = max(viol, g[i])
        else: viol = max(viol, abs(g[i]))
    return viol / (1.0 + np.abs(d).max())


def post_reduced_kkt(s, lb, ub, l0, l):
    """Solver-free optimality check of a postSmooth result: V maps the offsets to the oracle's variables (l' and l'' from the two difference
    rows, the last layer's l' minimised out, its l'' = 0); the box QP's KKT conditions with g = V^T P V l, P the oracle's.  Returns the
    largest violation relative to 1 + |l|max."""
    import pqp_oracle as O
    m = len(s)
    P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
    h = np.diff(s)
    V = np.zeros((3 * m, m))
    V[:m] = np.eye(m)
    for i in range(m - 1):
        V[m + i, i + 1] += 1 / h[i]; V[m + i, i] -= 1 / h[i]
    if m >= 2:
        V[2 * m - 1] = V[2 * m - 2] * (10.0 / (h[m - 2] ** 2 + 10.0))        # argmin over the last l' of 50 b^2 + 500 ((b - l'_{m-2}) / h)^2
    for i in range(m - 1):
        V[2 * m + i] = (V[m + i + 1] - V[m + i]) / h[i]
    v = V @ l
    assert np.abs(A @ v - np.clip(A @ v, lo, up))[m:].max(initial=0.0) < 1e-9           # the difference rows hold by construction
    g = V.T @ (P @ v)
    lo_l, up_l = lo[:m], up[:m]
    viol = 0.0
    for i in range(m):
        if lo_l[i] == up_l[i]:
            viol = max(viol, abs(l[i] - lo_l[i])); continue
        viol = max(viol, lo_l[i] - l[i], l[i] - up_l[i])
        if l[i] <= lo_l[i] + 1e-9: viol = max(viol, -g[i])
        elif l[i] >= up_l[i] - 1e-9: viol = max(viol, g[i])
        else: viol = max(viol, abs(g[i]))
    return viol / (1.0 + np.abs(l).max())
