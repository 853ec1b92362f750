"""The iterations of the exact TensionSmoother / postSmooth kernels (csrc/pqp_smoother_kernels.inc: tension_exact_kernel, post_exact_kernel)
restated in numpy - same first active set (TensionSmoother, round 3: from interior-point iterations; postSmooth: OSQP's cold-start rule), same
acceptance test, same cautious rule - and swept over many random lines on the CPU: factorisations needed, failures.  The kernel's counts
(tools/smoother_rounds.py, info[5]) reproduce these.
Usage: python tools/active_set_sweep.py [cases=200] [cold]       (CPU only; inputs from tests/smoother_cases.py, no oracle involved;
       `cold`: the TensionSmoother rounds from the cold-start rule, as round 2 ran them)"""
import os, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from smoother_cases import post_inputs, tension_inputs


def chain_matrix(n, w_k, w_dk):
    H = np.zeros((n, n))
    for i in range(n - 2):
        H[i:i + 3, i:i + 3] += w_k * np.outer([1, -2, 1], [1, -2, 1])
    for i in range(n - 3):
        H[i:i + 4, i:i + 4] += w_dk * np.outer([-1, 3, -3, 1], [-1, 3, -3, 1])
    return H


def tension_box_qp(n, seed, w_k=1.0, w_dk=50.0, w_dev=0.0):
    x, y, ang, _, _, cl = tension_inputs(n, seed=seed)
    c, s = np.cos(ang + np.pi / 2), np.sin(ang + np.pi / 2)
    H = chain_matrix(n, w_k, w_dk)
    Hd = c[:, None] * H * c[None, :] + s[:, None] * H * s[None, :] + w_dev * np.eye(n)
    lin = c * (H @ x) + s * (H @ y)
    lo, up = -np.minimum(cl, 2.0), np.minimum(cl, 2.0)
    lo[0] = up[0] = 0.0; lo[-1], up[-1] = -0.5, 0.5
    return Hd, lin, lo, up


def post_box_qp(m, seed):
    s, lb, ub, l0 = post_inputs(m, seed=seed)
    h = np.diff(s)
    V1 = np.zeros((m - 1, m))
    for i in range(m - 1):
        V1[i, i + 1], V1[i, i] = 1 / h[i], -1 / h[i]
    W = np.full(m - 1, 100.0); W[-1] += 1000.0 / (h[-1] ** 2 + 10.0)
    V2 = np.array([(V1[i + 1] - V1[i]) / h[i] for i in range(m - 2)])
    Hd = np.eye(m) + V1.T @ (W[:, None] * V1) + 1000.0 * V2.T @ V2
    lo, up = lb.copy(), ub.copy(); lo[0] = up[0] = l0
    return Hd, np.zeros(m), lo, up


def interior_start(Hd, lin, lo, up, gam=1.0):
    """Primal-dual interior-point iterations on the box QP (centring 0.2 twice, then 0.05; 0.995 of the way to the boundary; complementarity from
    max(0.01 gam, 3e-4 |g| t) to 1e-8 gam; boxes narrower than 1 mm take no part); returns the active set 'multiplier > gam * slack' and the iterations.  gam: the cost's scale relative to the reference's weights."""
    n = len(lin); pinned = lo == up
    w = up - lo
    narrow = ~pinned & (w < 1e-3)                   # stay at their centre, set by the gradient's sign afterwards
    fr = ~pinned & ~narrow; nf = int(fr.sum())
    d = np.where(fr, np.minimum(np.maximum(0.0, lo + 0.1 * w), up - 0.1 * w), 0.5 * (lo + up))
    tl, tu = np.where(fr, d - lo, 1.0), np.where(fr, up - d, 1.0)
    mu0 = 0.01 * gam
    if nf:
        mu0 = max(mu0, 3e-4 * np.abs((Hd @ d + lin)[fr]).max() * np.minimum(tl, tu)[fr].mean())
    zl, zu = np.where(fr, mu0 / tl, 0.0), np.where(fr, mu0 / tu, 0.0)
    its = 0
    for it in range(40 if nf else 0):
        g = Hd @ d + lin
        mu = (tl[fr] @ zl[fr] + tu[fr] @ zu[fr]) / (2 * nf)
        if mu < 1e-8 * gam and np.abs((g - zl + zu)[fr]).max() < 1e-6 * (gam + np.abs(g).max()):
            break
        smu = (0.2 if it < 2 else 0.05) * mu
        A = Hd + np.diag(zl / tl + zu / tu)
        dd = np.zeros(n)
        dd[fr] = np.linalg.solve(A[np.ix_(fr, fr)], (-g + smu / tl - smu / tu)[fr])
        its += 1
        dzl = np.where(fr, (smu - tl * zl - zl * dd) / tl, 0.0); dzu = np.where(fr, (smu - tu * zu + zu * dd) / tu, 0.0)
        a = np.inf
        for v, dv in ((tl, dd), (tu, -dd), (zl, dzl), (zu, dzu)):
            m = (dv < 0) & fr
            if m.any(): a = min(a, (-v[m] / dv[m]).min())
        a = min(1.0, 0.995 * a)
        d = d + a * dd; tl = np.where(fr, tl + a * dd, 1.0); tu = np.where(fr, tu - a * dd, 1.0); zl = zl + a * dzl; zu = zu + a * dzu
    g = Hd @ d + lin
    return np.where(pinned, -1, np.where(narrow, np.where(g > 0, -1, 1), np.where(zl > gam * tl, -1, np.where(zu > gam * tu, 1, 0)))), its


def active_set(Hd, lin, lo, up, cautious, tol=1e-7, act=None):
    """act: a first active set (the interior start's: the rounds are then cautious from the first one on, threshold 0.9); None: OSQP's cold-start rule."""
    n = len(lin); pinned = lo == up
    interior = act is not None
    if act is None:
        act = np.where(pinned, -1, np.where(lo > 0, -1, np.where(up < 0, 1, 0)))
    best, stall, cons = 1e300, 0, interior
    if interior:
        cautious = 0.9
    for rnd in range(6 * n + 40):
        fix = np.where(act < 0, lo, np.where(act > 0, up, 0.0)); fr = act == 0
        d = fix.copy()
        if fr.any():
            d[fr] = np.linalg.solve(Hd[np.ix_(fr, fr)], -lin[fr] - Hd[np.ix_(fr, ~fr)] @ fix[~fr])
        g = Hd @ d + lin
        viol = np.where(pinned, 0.0, np.where(fr, np.maximum(np.maximum(lo - d, d - up), 0), np.where(act < 0, np.maximum(-g, 0), np.maximum(g, 0))))
        vmax, scale = viol.max(), 1 + np.abs(d).max()
        if vmax <= tol * scale:
            return rnd + 1
        if vmax < 0.7 * best: best, stall = vmax, 0
        else: stall += 1
        if stall >= 3: cons = True
        c = 0.9 if (cautious < 0.9 and rnd >= 40 + n // 4) else cautious          # (the TensionSmoother kernel: single moves once a line is far beyond the usual rounds)
        thr = max(tol * scale, c * vmax) if cons else tol * scale
        mv = (viol >= thr) & (viol > 0) & ~pinned
        na = act.copy(); na[mv & fr] = np.where((lo - d > d - up)[mv & fr], -1, 1); na[mv & ~fr] = 0; act = na
    return -1


def job(a):
    kind, n, seed, cold = a
    if kind == "tension":
        qp = tension_box_qp(n, seed)
        if cold:
            return active_set(*qp, cautious=0.5)
        act, its = interior_start(*qp)
        r = active_set(*qp, cautious=0.5, act=act)
        return r + its if r > 0 else r
    return active_set(*post_box_qp(n, seed), cautious=0.9)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cold = "cold" in sys.argv[2:]
    with ProcessPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        for kind, sizes in (("tension", (24, 48, 80, 130, 200, 300)), ("post", (8, 18, 28, 60, 150, 341))):
            for n in sizes:
                r = np.array(list(ex.map(job, [(kind, n, 1000 + s, cold) for s in range(cases)])))
                ok = r[r > 0]
                print(f"{kind:8s} n {n:4d}: {cases} lines, failed {int((r < 0).sum())}, factorisations mean {ok.mean():5.1f} p90 {np.percentile(ok, 90):4.0f} max {ok.max():4d}", flush=True)
