"""numpy prototype of the per-waypoint ("lane") formulation the HIP kernel implements.
Design aid only (not product, not oracle): checks the structured Ruiz metrics, the metric-form
ADMM, the analytic slack/control elimination and the block cyclic reduction against the generic
oracle in oracle/pqp_oracle.py.   Run: python tests/prototype/lane_prototype.py
"""
import math, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import scipy.sparse as sp
import pqp_oracle as O
from path_optimizer_2_amd.synth import make_batch

INF = 1e30

def lane_data(ref, lin, bounds, scal, prm):
    """Per-waypoint compact data (what assemble produces for the solve kernel)."""
    n = ref.shape[0]
    a = np.zeros((n, 6))        # a00 a01 a10 a11 a12 ds  (transition i -> i+1; zero for i = n-1)
    bT = np.zeros((n, 3))       # equality rhs of transition rows of waypoint i
    bT[0] = -np.array(scal[:3])
    for i in range(n - 1):
        l, psi, k = lin[i]
        ds = ref[i + 1, 0] - ref[i, 0]
        t, c = math.tan(psi), math.cos(psi)
        dfx = np.array([[-k * t, (1 - k * l) / c ** 2, 0], [-k * k / c, (1 - k * l) * k * t / c, (1 - k * l) / c], [0, 0, 0]])
        A = ds * dfx + np.eye(3)
        a[i] = [A[0, 0], A[0, 1], A[1, 0], A[1, 1], A[1, 2], ds]
        u_in = (lin[i + 1, 2] - k) / ds
        f = np.array([(1 - k * l) * t, (1 - k * l) * k / c - ref[i, 1], u_in])
        cvec = ds * (f - dfx @ np.array([l, psi, k]) - np.array([0, 0, 1.0]) * u_in)
        bT[i + 1] = -cvec
    lo = np.zeros((n, 3)); up = np.zeros((n, 3))     # K, F, R rows
    kap = math.tan(scal[5]) / prm.wheel_base
    lo[:, 0] = -kap; up[:, 0] = kap
    for i in range(n):
        lo[i, 1], up[i, 1] = O.soft_bounds(bounds[i, 0], bounds[i, 1], prm.expected_safety_margin)
        lo[i, 2], up[i, 2] = O.soft_bounds(bounds[i, 2], bounds[i, 3], prm.expected_safety_margin)
    elo = np.array([-1.0, -INF]); eup = np.array([1.0, INF])
    if prm.constraint_end_heading and scal[4] == 0:
        ep = O.constrain_angle(scal[3] - ref[-1, 2])
        if ep < prm.end_psi_max:
            elo[1], eup[1] = ep - prm.end_psi_tol, ep + prm.end_psi_tol
    return dict(n=n, a=a, bT=bT, lo=lo, up=up, elo=elo, eup=eup, cf=prm.front_length, cr=prm.rear_length,
                Pd=np.array([prm.weight_l, 0.0, prm.weight_kappa, prm.weight_dkappa, prm.weight_slack, prm.weight_slack]))

def lim(v):
    v = np.where(v < 1e-4, 1.0, v); return np.minimum(v, 1e4)

def shift_next(v):   # value of waypoint i+1 seen at i (zero past the end)
    r = np.zeros_like(v); r[:-1] = v[1:]; return r
def shift_prev(v):   # value of waypoint i-1 seen at i
    r = np.zeros_like(v); r[1:] = v[:-1]; return r

def ruiz(d, passes=10):
    n = d['n']; a = np.abs(d['a']); cf, cr = abs(d['cf']), abs(d['cr'])
    has_next = np.arange(n) < n - 1
    last = np.arange(n) == n - 1
    Pd = np.tile(d['Pd'], (n, 1)); Pd[n - 1, 3] = 0.0      # no control at last waypoint
    D = np.ones((n, 6)); E = np.ones((n, 6)); Ee = np.ones(2); c = 1.0
    for _ in range(passes):
        En = shift_next(E[:, :3])          # E of T rows of waypoint i+1
        # column norms
        cn = np.zeros((n, 6))
        cn[:, 0] = np.maximum.reduce([E[:, 0], En[:, 0] * a[:, 0], En[:, 1] * a[:, 2], E[:, 4], E[:, 5], np.where(last, Ee[0], 0)])
        cn[:, 1] = np.maximum.reduce([E[:, 1], En[:, 0] * a[:, 1], En[:, 1] * a[:, 3], E[:, 4] * cf, E[:, 5] * cr, np.where(last, Ee[1], 0)])
        cn[:, 2] = np.maximum.reduce([E[:, 2], En[:, 1] * a[:, 4], np.where(has_next, En[:, 2], 0), E[:, 3]])
        cn[:, 3] = En[:, 2] * a[:, 5]
        cn[:, 4] = E[:, 4]; cn[:, 5] = E[:, 5]
        cn = np.maximum(cn * D, c * D * D * Pd)
        cn[n - 1, 3] = 1.0    # dummy control at the last waypoint
        # row norms
        Dp = shift_prev(D); ap = shift_prev(a)
        rn = np.zeros((n, 6))
        rn[:, 0] = np.maximum.reduce([D[:, 0], Dp[:, 0] * ap[:, 0], Dp[:, 1] * ap[:, 1]])
        rn[:, 1] = np.maximum.reduce([D[:, 1], Dp[:, 0] * ap[:, 2], Dp[:, 1] * ap[:, 3], Dp[:, 2] * ap[:, 4]])
        rn[:, 2] = np.maximum.reduce([D[:, 2], np.where(np.arange(n) > 0, Dp[:, 2], 0), Dp[:, 3] * ap[:, 5]])
        rn[:, 3] = D[:, 2]
        rn[:, 4] = np.maximum.reduce([D[:, 0], D[:, 1] * cf, D[:, 4]])
        rn[:, 5] = np.maximum.reduce([D[:, 0], D[:, 1] * cr, D[:, 5]])
        rn = rn * E
        ren = np.array([D[n - 1, 0], D[n - 1, 1]]) * Ee
        D = D / np.sqrt(lim(cn)); E = E / np.sqrt(lim(rn)); Ee = Ee / np.sqrt(lim(ren))
        D[n - 1, 3] = 1.0
        pcn = c * D * D * Pd
        nvars = 6 * n - 1
        ct = pcn.sum() / nvars if False else (np.abs(pcn).sum() / nvars)
        ct = max(ct, 1.0)          # q == 0 -> ||q|| limited to 1
        ct = 1.0 if ct < 1e-4 else min(ct, 1e4)
        c = c / ct
    return D, E, Ee, c

def to_ref_order(n, v6, fill_u_last=False):
    """per-waypoint [n][6] (l,psi,k,u,sf,sr) -> reference variable order (6n-1)."""
    out = np.zeros(6 * n - 1)
    for i in range(n):
        out[3 * i:3 * i + 3] = v6[i, :3]
        if i < n - 1: out[3 * n + i] = v6[i, 3]
        out[4 * n - 1 + 2 * i] = v6[i, 4]; out[4 * n - 1 + 2 * i + 1] = v6[i, 5]
    return out

def rows_to_ref_order(n, r6, re):
    out = np.zeros(6 * n + 2)
    for i in range(n):
        out[3 * i:3 * i + 3] = r6[i, :3]; out[3 * n + i] = r6[i, 3]
        out[4 * n + 2 * i] = r6[i, 4]; out[4 * n + 2 * i + 1] = r6[i, 5]
    out[6 * n:6 * n + 2] = re
    return out

# ---------------- structured operators --------------------------------------------------------
def A_mul(d, x):
    """x [n][6] -> rows [n][6], end rows [2]"""
    n = d['n']; a = d['a']
    out = np.zeros((n, 3))
    out[:, 0] = a[:, 0] * x[:, 0] + a[:, 1] * x[:, 1]
    out[:, 1] = a[:, 2] * x[:, 0] + a[:, 3] * x[:, 1] + a[:, 4] * x[:, 2]
    out[:, 2] = x[:, 2] + a[:, 5] * x[:, 3]
    out[n - 1] = 0
    r = np.zeros((n, 6))
    r[:, :3] = shift_prev(out) - x[:, :3]
    r[:, 3] = x[:, 2]
    r[:, 4] = x[:, 0] + d['cf'] * x[:, 1] + x[:, 4]
    r[:, 5] = x[:, 0] + d['cr'] * x[:, 1] + x[:, 5]
    return r, np.array([x[n - 1, 0], x[n - 1, 1]])

def At_mul(d, w, we):
    n = d['n']; a = d['a']
    wn = shift_next(w[:, :3])
    g = np.zeros((n, 6))
    g[:, 0] = -w[:, 0] + a[:, 0] * wn[:, 0] + a[:, 2] * wn[:, 1] + w[:, 4] + w[:, 5]
    g[:, 1] = -w[:, 1] + a[:, 1] * wn[:, 0] + a[:, 3] * wn[:, 1] + d['cf'] * w[:, 4] + d['cr'] * w[:, 5]
    g[:, 2] = -w[:, 2] + a[:, 4] * wn[:, 1] + np.where(np.arange(n) < n - 1, wn[:, 2], 0) + w[:, 3]
    g[:, 3] = a[:, 5] * wn[:, 2]
    g[:, 4] = w[:, 4]; g[:, 5] = w[:, 5]
    g[n - 1, 0] += we[0]; g[n - 1, 1] += we[1]
    return g

def build_blocks(d, rho, rhoe, sig):
    """Eliminate slacks + control analytically -> block tridiagonal (Dg[n][3][3], Eg[n][3][3] coupling i,i+1)
    plus the elimination coefficients."""
    n = d['n']; a = d['a']; cf, cr = d['cf'], d['cr']
    Pd = np.tile(d['Pd'], (n, 1))
    rn = shift_next(rho[:, :3])
    Dg = np.zeros((n, 3, 3)); Eg = np.zeros((n, 3, 3))
    dsf = Pd[:, 4] + sig[:, 4] + rho[:, 4]; dsr = Pd[:, 5] + sig[:, 5] + rho[:, 5]
    gf = rho[:, 4] - rho[:, 4] ** 2 / dsf; gr = rho[:, 5] - rho[:, 5] ** 2 / dsr     # effective row weights
    du = Pd[:, 3] + sig[:, 3] + rn[:, 2] * a[:, 5] ** 2
    tu = rn[:, 2] * a[:, 5]                 # coupling u_i <-> k_i (and -tu to k_{i+1})
    gu = rn[:, 2] - tu ** 2 / du            # effective weight of row T_{i+1,2} restricted to (k_i, k_{i+1})
    gu[n - 1] = 0; tu[n - 1] = 0
    for i in range(n):
        Ai = np.array([[a[i, 0], a[i, 1], 0], [a[i, 2], a[i, 3], a[i, 4]], [0, 0, 0.0]])  # rows 0,1 of A_i (row 2 handled via gu)
        R2 = np.diag([rn[i, 0], rn[i, 1], 0.0])
        M = np.diag(Pd[i, :3] + sig[i, :3] + rho[i, :3]) + Ai.T @ R2 @ Ai
        M[2, 2] += gu[i] + rho[i, 3]
        v = np.array([1, cf, 0.0]); M += gf[i] * np.outer(v, v)
        v = np.array([1, cr, 0.0]); M += gr[i] * np.outer(v, v)
        if i == n - 1:
            M[0, 0] += rhoe[0]; M[1, 1] += rhoe[1]
        if i > 0:
            M[2, 2] += gu[i - 1] - rho[i, 2]      # row T_i,2 acts on k_i with weight: rho -> gu (after eliminating u_{i-1})
        Dg[i] = M
        if i < n - 1:
            C = -(Ai.T @ R2)                      # coupling X_i (rows) x X_{i+1} (cols)
            C[2, 2] += -gu[i]
            Eg[i] = C
    return Dg, Eg, dict(dsf=dsf, dsr=dsr, du=du, tu=tu)

def cr_factor(Dg, Eg):
    """Block cyclic reduction factorisation of blocktridiag(Dg, Eg). Nodes padded to power of two."""
    n = Dg.shape[0]; Np = 1 << max(1, (n - 1).bit_length())
    D = np.tile(np.eye(3), (Np, 1, 1)); D[:n] = Dg
    L = np.zeros((Np, 3, 3)); R = np.zeros((Np, 3, 3))      # L[j]: coupling (j-h, j) as block [j-h rows][j cols]; R[j]: (j, j+h)
    R[:n] = Eg; L[1:n] = Eg[:n - 1]
    fac = dict(Np=Np, Dinv=np.zeros((Np, 3, 3)), GL=np.zeros((Np, 3, 3)), GR=np.zeros((Np, 3, 3)), levels=[])
    h = 1
    while h < Np:
        elim = [j for j in range(Np) if j % (2 * h) == h]
        fac['levels'].append((h, elim))
        newD = D.copy(); newL = L.copy(); newR = R.copy()
        for j in elim:
            Di = np.linalg.inv(D[j]); fac['Dinv'][j] = Di
            GL = L[j] @ Di            # multiplies r_j, subtracts from r_{j-h}
            fac['GL'][j] = GL
            newD[j - h] -= GL @ L[j].T
            if j + h < Np:
                GR = R[j].T @ Di; fac['GR'][j] = GR
                newD[j + h] -= GR @ R[j]
                # new coupling (j-h, j+h)
                newR[j - h] = -GL @ R[j]
                newL[j + h] = -GL @ R[j]
            else:
                newR[j - h] = 0
            fac['L_' + str(j)] = L[j].copy(); fac['R_' + str(j)] = R[j].copy()
        D, L, R = newD, newL, newR
        h *= 2
    fac['Dinv'][0] = np.linalg.inv(D[0])
    return fac

def cr_solve(fac, rhs):
    Np = fac['Np']; n = rhs.shape[0]
    r = np.zeros((Np, 3)); r[:n] = rhs
    for h, elim in fac['levels']:
        upd = np.zeros_like(r)
        for j in elim:
            upd[j - h] -= fac['GL'][j] @ r[j]
            if j + h < Np: upd[j + h] -= fac['GR'][j] @ r[j]
        r += upd
    x = np.zeros((Np, 3)); x[0] = fac['Dinv'][0] @ r[0]
    for h, elim in reversed(fac['levels']):
        for j in elim:
            t = fac['Dinv'][j] @ r[j] - fac['GL'][j].T @ x[j - h]
            if j + h < Np: t -= fac['GR'][j].T @ x[j + h]
            x[j] = t
    return x[:n]

def lane_solve_system(d, fac, el, rho, r):
    """Solve S xt = r (r [n][6]) using the eliminations + CR."""
    n = d['n']; a = d['a']; cf, cr = d['cf'], d['cr']
    rX = r[:, :3].copy()
    rX[:, 0] -= rho[:, 4] / el['dsf'] * r[:, 4] + rho[:, 5] / el['dsr'] * r[:, 5]
    rX[:, 1] -= cf * rho[:, 4] / el['dsf'] * r[:, 4] + cr * rho[:, 5] / el['dsr'] * r[:, 5]
    tud = el['tu'] / el['du'] * r[:, 3]
    rX[:, 2] -= tud
    rX[:, 2] += shift_prev(tud)
    X = cr_solve(fac, rX)
    xt = np.zeros((n, 6)); xt[:, :3] = X
    kn = shift_next(X[:, 2])
    xt[:, 3] = (r[:, 3] - el['tu'] * (X[:, 2] - kn)) / el['du']; xt[n - 1, 3] = 0
    xt[:, 4] = (r[:, 4] - rho[:, 4] * (X[:, 0] + cf * X[:, 1])) / el['dsf']
    xt[:, 5] = (r[:, 5] - rho[:, 5] * (X[:, 0] + cr * X[:, 1])) / el['dsr']
    return xt

def lane_admm(d, eps, rho0=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, warm=None, rho_init=None):
    n = d['n']
    D, E, Ee, c = ruiz(d)
    lo6 = np.zeros((n, 6)); up6 = np.zeros((n, 6))
    lo6[:, :3] = d['bT']; up6[:, :3] = d['bT']; lo6[:, 3:] = d['lo']; up6[:, 3:] = d['up']
    elo, eup = d['elo'], d['eup']
    def rho_vectors(rho):
        cls = np.full((n, 6), rho); 
        free = (E * lo6 < -INF * 1e-4) & (E * up6 > INF * 1e-4); eq = (~free) & (E * (up6 - lo6) < 1e-4)
        cls[free] = 1e-6; cls[eq] = 1e3 * rho
        clse = np.full(2, rho); fe = (Ee * elo < -INF * 1e-4) & (Ee * eup > INF * 1e-4); ee = (~fe) & (Ee * (eup - elo) < 1e-4)
        clse[fe] = 1e-6; clse[ee] = 1e3 * rho
        return cls * E * E / c, clse * Ee * Ee / c
    sig = sigma / (c * D * D)
    rho = rho0 if rho_init is None else rho_init
    def factor(rho):
        rv, rve = rho_vectors(rho)
        Dg, Eg, el = build_blocks(d, rv, rve, sig)
        return rv, rve, cr_factor(Dg, Eg), el
    rv, rve, fac, el = factor(rho)
    x = np.zeros((n, 6)); y = np.zeros((n, 6)); ye = np.zeros(2); z = np.zeros((n, 6)); ze = np.zeros(2)
    if warm is not None:
        x, y, ye = warm[0].copy(), warm[1].copy(), warm[2].copy()
        z, ze = A_mul(d, x)
    Pd = np.tile(d['Pd'], (n, 1)); Pd[n - 1, 3] = 0
    status = 'max_iter'
    for it in range(1, max_iter + 1):
        w = rv * z - y; we = rve * ze - ye
        r = sig * x + At_mul(d, w, we)
        r[n - 1, 3] = 0
        xt = lane_solve_system(d, fac, el, rv, r)
        zt, zte = A_mul(d, xt)
        x = alpha * xt + (1 - alpha) * x
        zh = alpha * zt + (1 - alpha) * z; zhe = alpha * zte + (1 - alpha) * ze
        zn = np.clip(zh + y / rv, lo6, up6); zne = np.clip(zhe + ye / rve, elo, eup)
        y = y + rv * (zh - zn); ye = ye + rve * (zhe - zne)
        z, ze = zn, zne
        if it % 25 == 0:
            Ax, Axe = A_mul(d, x)
            Aty = At_mul(d, y, ye); Aty[n - 1, 3] = 0
            Px = Pd * x
            pri = max(np.abs(Ax - z).max(), np.abs(Axe - ze).max())
            dua = np.abs(Px + Aty).max()
            nax = max(np.abs(Ax).max(), np.abs(Axe).max()); nz = max(np.abs(z).max(), np.abs(ze).max())
            npx = np.abs(Px).max(); naty = np.abs(Aty).max()
            if pri <= eps + eps * max(nax, nz) and dua <= eps + eps * max(npx, naty):
                status = 'solved'; break
            if it % 100 == 0:
                pn = pri / (max(nax, nz) + 1e-10); dn = dua / (max(npx, naty) + 1e-10)
                rn_ = float(np.clip(rho * math.sqrt(pn / (dn + 1e-10)), 1e-6, 1e6))
                if rn_ > 5 * rho or rn_ < rho / 5:
                    rho = rn_; rv, rve, fac, el = factor(rho)
    return dict(x=x, y=y, ye=ye, iters=it, status=status, rho=rho)

if __name__ == '__main__':
    prm = O.PathQpParams()
    b = make_batch(4, 80)
    for q in range(4):
        ref, bounds, scal = b['ref'][q], b['bounds'][q], b['scal'][q]
        lin = O.first_linearization(ref)
        d = lane_data(ref, lin, bounds, scal, prm)
        Pd, A, lo, up, sz = O.assemble_path_qp(ref, lin, bounds, scal, prm)
        n = d['n']
        # 1. Ruiz
        Ps, qs, As, ls, us, Dg, Eg, cg = O.ruiz_scale(sp.diags(Pd), np.zeros(sz['vars']), A, lo, up, 10)
        D, E, Ee, c = ruiz(d)
        print('ruiz D err', np.abs(to_ref_order(n, D) - Dg).max(), 'E err', np.abs(rows_to_ref_order(n, E, Ee) - Eg).max(), 'c', c, cg)
        # 2. operators
        rng = np.random.default_rng(0); xr = rng.standard_normal((n, 6)); xr[n - 1, 3] = 0
        r6, re = A_mul(d, xr)
        print('A_mul err', np.abs(rows_to_ref_order(n, r6, re) - A @ to_ref_order(n, xr)).max())
        wr = rng.standard_normal((n, 6)); wre = rng.standard_normal(2)
        g = At_mul(d, wr, wre); g[n - 1, 3] = 0
        print('At_mul err', np.abs(to_ref_order(n, g) - A.T @ rows_to_ref_order(n, wr, wre)).max())
        # 3. solve
        rho6 = np.abs(rng.standard_normal((n, 6))) + 0.1; rhoe = np.abs(rng.standard_normal(2)) + 0.1; sig = np.abs(rng.standard_normal((n, 6))) * 1e-3 + 1e-6
        Dgb, Egb, el = build_blocks(d, rho6, rhoe, sig)
        fac = cr_factor(Dgb, Egb)
        rr = rng.standard_normal((n, 6)); rr[n - 1, 3] = 0
        xt = lane_solve_system(d, fac, el, rho6, rr)
        sigr = to_ref_order(n, sig); sigr_full = sigr.copy()
        S = np.diag(Pd + sigr) + A.T @ np.diag(rows_to_ref_order(n, rho6, rhoe)) @ A
        xs = np.linalg.solve(S, to_ref_order(n, rr))
        print('solve err', np.abs(to_ref_order(n, xt) - xs).max(), 'cond', np.linalg.cond(S))
        # 4. admm vs oracle
        for eps in (1e-4, 1e-7):
            st = O.OsqpSettings(eps_abs=eps, eps_rel=eps)
            ro = O.osqp_admm(sp.diags(Pd), np.zeros(sz['vars']), A, lo, up, st)
            rl = lane_admm(d, eps)
            print('eps', eps, 'iters oracle', ro['iters'], 'lane', rl['iters'], 'rho', ro['rho'], rl['rho'], 'x diff', np.abs(to_ref_order(n, rl['x']) - ro['x']).max())
