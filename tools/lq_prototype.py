#!/usr/bin/env python
"""numpy prototype of the lane-per-QP path-QP solver (csrc/pqp_path_lq.hpp): every array op below is over the batch axis, the way the
64 lanes of a wavefront run 64 QPs in lock-step.

The path QP of base_solver.cpp:119-261 is a linear-quadratic control problem: state (l, psi, kappa)_i, control u_i = kappa', dynamics
= the transition rows, slacks eliminated in closed form (a soft row is the penalty w_s/2 dist(a^T x, [lo, up])^2), hard boxes on kappa_i
and on the end state.  For a fixed active set it is solved EXACTLY by one backward Riccati sweep + one forward roll-out; hard active rows
carry the penalty 1/delta with a multiplier shift (method of multipliers across rounds).  Rounds = primal-dual active-set updates.

Usage: python tools/lq_prototype.py [batch] [n] [profile] [seed]   -> rounds statistics and the distance to the converged C oracle.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


class Prm:
    front_length = 3.9; rear_length = -1.0; wheel_base = 2.5; margin = 0.6; min_clearance = 0.1
    w_l = 0.0; w_k = 20.0; w_u = 100.0; w_s = 10.0
    end_l = 1.0; end_psi_tol = 0.087; end_psi_max = 70.0 * np.pi / 180.0
    delta = 1e-9
    tol = 1e-7


def constrain_angle(a):
    a = a.copy()
    for _ in range(64):
        hi = a > np.pi; lo = a < -np.pi
        if not (hi.any() or lo.any()):
            break
        a = np.where(hi, a - 2 * np.pi, np.where(lo, a + 2 * np.pi, a))
    return a


def soft_bounds(lb, ub, p):
    cl = ub - lb
    rem = np.maximum(p.min_clearance, cl - 2 * p.margin)
    sh = np.maximum(0.0, (cl - rem) / 2.0)
    return lb + sh, ub - sh


def stage_data(ref, lin, bounds, scal, p):
    """Per pass: dynamics M (5 entries), c (2), ds; soft boxes; hard boxes."""
    s, kref = ref[:, :, 0], ref[:, :, 1]
    l, psi, k = lin[:, :-1, 0], lin[:, :-1, 1], lin[:, :-1, 2]
    knext = lin[:, 1:, 2]
    ds = s[:, 1:] - s[:, :-1]
    t, cs = np.tan(psi), np.cos(psi)
    df00 = -k * t; df01 = (1 - k * l) / cs ** 2
    df10 = -k * k / cs; df11 = (1 - k * l) * k * t / cs; df12 = (1 - k * l) / cs
    u_in = (knext - k) / ds
    f0 = (1 - k * l) * t; f1 = (1 - k * l) * k / cs - kref[:, :-1]
    c0 = ds * (f0 - (df00 * l + df01 * psi))
    c1 = ds * (f1 - (df10 * l + df11 * psi + df12 * k))
    d = dict(m00=1 + ds * df00, m01=ds * df01, m10=ds * df10, m11=1 + ds * df11, m12=ds * df12, c0=c0, c1=c1, ds=ds)
    d["lo_f"], d["up_f"] = soft_bounds(bounds[:, :, 0], bounds[:, :, 1], p)
    d["lo_r"], d["up_r"] = soft_bounds(bounds[:, :, 2], bounds[:, :, 3], p)
    d["kl"] = np.tan(scal[:, 5]) / p.wheel_base
    end_psi = constrain_angle(scal[:, 3] - ref[:, -1, 2])
    on = (scal[:, 4] == 0.0) & (end_psi < p.end_psi_max)
    d["psi_lo"] = np.where(on, end_psi - p.end_psi_tol, -1e30)
    d["psi_hi"] = np.where(on, end_psi + p.end_psi_tol, 1e30)
    d["x0"] = scal[:, 0:3].copy()
    return d


def riccati_solve(d, W, p):
    """Backward Riccati sweep + forward roll-out for per-row quadratic terms 1/2 w (a^T x - target)^2.
    W: dict wf, tf, wr, tr, wk, tk [B,N]; wel, tel, wep, tep [B].  Returns x [B,N,3], u [B,N-1]."""
    B, N = W["wf"].shape
    Lf, Lr = p.front_length, p.rear_length
    K = np.zeros((B, N - 1, 3)); k0 = np.zeros((B, N - 1))

    def stage_cost(i):
        Q = np.zeros((B, 6)); q = np.zeros((B, 3))
        Q[:, 0] += p.w_l; Q[:, 5] += p.w_k
        for wk_, tk_, L in ((W["wf"][:, i], W["tf"][:, i], Lf), (W["wr"][:, i], W["tr"][:, i], Lr)):
            Q[:, 0] += wk_; Q[:, 1] += wk_ * L; Q[:, 3] += wk_ * L * L
            q[:, 0] -= wk_ * tk_; q[:, 1] -= wk_ * tk_ * L
        Q[:, 5] += W["wk"][:, i]; q[:, 2] -= W["wk"][:, i] * W["tk"][:, i]
        if i == N - 1:
            Q[:, 0] += W["wel"]; q[:, 0] -= W["wel"] * W["tel"]
            Q[:, 3] += W["wep"]; q[:, 1] -= W["wep"] * W["tep"]
        return Q, q

    P, pv = stage_cost(N - 1)
    for i in range(N - 2, -1, -1):
        ds = d["ds"][:, i]
        g0, g1, g2 = P[:, 2], P[:, 4], P[:, 5]
        S = p.w_u + ds * ds * g2
        r = ds * ds / S
        wS = p.w_u / S
        Pb00 = P[:, 0] - r * g0 * g0; Pb01 = P[:, 1] - r * g0 * g1; Pb11 = P[:, 3] - r * g1 * g1
        Pb02 = g0 * wS; Pb12 = g1 * wS; Pb22 = g2 * wS
        pb0 = pv[:, 0] - r * pv[:, 2] * g0; pb1 = pv[:, 1] - r * pv[:, 2] * g1; pb2 = pv[:, 2] * wS
        m00, m01, m10, m11, m12 = d["m00"][:, i], d["m01"][:, i], d["m10"][:, i], d["m11"][:, i], d["m12"][:, i]
        c0, c1 = d["c0"][:, i], d["c1"][:, i]
        f = ds / S
        K[:, i, 0] = f * (m00 * g0 + m10 * g1)
        K[:, i, 1] = f * (m01 * g0 + m11 * g1)
        K[:, i, 2] = f * (m12 * g1 + g2)
        k0[:, i] = f * (g0 * c0 + g1 * c1 + pv[:, 2])
        h0 = Pb00 * c0 + Pb01 * c1 + pb0
        h1 = Pb01 * c0 + Pb11 * c1 + pb1
        h2 = Pb02 * c0 + Pb12 * c1 + pb2
        T00 = Pb00 * m00 + Pb01 * m10; T01 = Pb00 * m01 + Pb01 * m11; T02 = Pb01 * m12 + Pb02
        T10 = Pb01 * m00 + Pb11 * m10; T11 = Pb01 * m01 + Pb11 * m11; T12 = Pb11 * m12 + Pb12
        T22 = Pb12 * m12 + Pb22
        Q, q = stage_cost(i)
        Pn = np.empty((B, 6))
        Pn[:, 0] = Q[:, 0] + m00 * T00 + m10 * T10
        Pn[:, 1] = Q[:, 1] + m00 * T01 + m10 * T11
        Pn[:, 2] = Q[:, 2] + m00 * T02 + m10 * T12
        Pn[:, 3] = Q[:, 3] + m01 * T01 + m11 * T11
        Pn[:, 4] = Q[:, 4] + m01 * T02 + m11 * T12
        Pn[:, 5] = Q[:, 5] + m12 * T12 + T22
        pn = np.empty((B, 3))
        pn[:, 0] = q[:, 0] + m00 * h0 + m10 * h1
        pn[:, 1] = q[:, 1] + m01 * h0 + m11 * h1
        pn[:, 2] = q[:, 2] + m12 * h1 + h2
        P, pv = Pn, pn
    x = np.zeros((B, N, 3)); u = np.zeros((B, N - 1))
    x[:, 0] = d["x0"]
    for i in range(N - 1):
        xi = x[:, i]
        u[:, i] = -(K[:, i, 0] * xi[:, 0] + K[:, i, 1] * xi[:, 1] + K[:, i, 2] * xi[:, 2]) - k0[:, i]
        x[:, i + 1, 0] = d["m00"][:, i] * xi[:, 0] + d["m01"][:, i] * xi[:, 1] + d["c0"][:, i]
        x[:, i + 1, 1] = d["m10"][:, i] * xi[:, 0] + d["m11"][:, i] * xi[:, 1] + d["m12"][:, i] * xi[:, 2] + d["c1"][:, i]
        x[:, i + 1, 2] = xi[:, 2] + d["ds"][:, i] * u[:, i]
    return x, u


def riccati_round(d, act, lam, p):
    """One active-set round: the rows of `act` (int8 in {-1, 0, +1}) as exact quadratic terms - soft rows with weight w_s at their bound,
    hard rows with 1/delta at their bound shifted by delta * lam (method of multipliers across rounds)."""
    inv_d = 1.0 / p.delta
    W = {}
    for key, lo, up in (("f", d["lo_f"], d["up_f"]), ("r", d["lo_r"], d["up_r"])):
        a = act[key]
        W["w" + key] = np.where(a != 0, p.w_s, 0.0)
        W["t" + key] = np.where(a > 0, up, np.where(a < 0, lo, 0.0))
    a = act["k"]
    W["wk"] = np.where(a != 0, inv_d, 0.0); W["tk"] = np.where(a != 0, a * d["kl"][:, None] - p.delta * lam["k"], 0.0)
    a = act["el"]
    W["wel"] = np.where(a != 0, inv_d, 0.0); W["tel"] = np.where(a != 0, a * p.end_l - p.delta * lam["el"], 0.0)
    a = act["ep"]
    W["wep"] = np.where(a != 0, inv_d, 0.0); W["tep"] = np.where(a != 0, np.where(a > 0, d["psi_hi"], d["psi_lo"]) - p.delta * lam["ep"], 0.0)
    return riccati_solve(d, W, p)


def new_active_set(d, act, lam, x, p, rule):
    """The active set the point x asks for + multiplier estimates of the hard rows; returns (act', lam', changed [B], viol [B])."""
    B, N = act["f"].shape
    tol = p.tol
    new = {k: v.copy() for k, v in act.items()}
    nlam = {k: v.copy() for k, v in lam.items()}
    worst = np.zeros(B)
    for key, L, lo, up in (("f", p.front_length, d["lo_f"], d["up_f"]), ("r", p.rear_length, d["lo_r"], d["up_r"])):
        v = x[:, :, 0] + L * x[:, :, 1]
        a = act[key]
        want = np.where(v > up, 1, np.where(v < lo, -1, 0)).astype(np.int8)
        # consistent within tol: keep the old status
        near = ((a == 1) & (v >= up - tol)) | ((a == -1) & (v <= lo + tol)) | ((a == 0) & (v <= up + tol) & (v >= lo - tol))
        if rule in ("peaks", "softpeaks"):
            # adds (inactive -> active): only the local maxima of the violation along the path
            viol_s = np.where(a == 0, np.maximum(v - up, lo - v), 0.0)
            viol_s = np.where(viol_s > tol, viol_s, 0.0)
            pad = np.pad(viol_s, ((0, 0), (1, 1)))
            peak = (viol_s > 0) & (viol_s >= pad[:, :-2]) & (viol_s >= pad[:, 2:])
            want = np.where((a == 0) & ~peak, 0, want).astype(np.int8)
        new[key] = np.where(near, a, want)
        worst = np.maximum(worst, np.where(near, 0.0, np.abs(np.where(a == 1, v - up, np.where(a == -1, v - lo, np.maximum(v - up, lo - v))))).max(axis=1))
    # hard rows: kappa_i, i >= 1
    kap = x[:, :, 2]; kl = d["kl"][:, None]
    a = act["k"]
    y = lam["k"] + (kap - a * kl) / p.delta                     # multiplier estimate of an active row
    y = np.where(a != 0, y, 0.0)
    release = (a != 0) & (a * y < -tol)
    add_hi = (a == 0) & (kap > kl + tol); add_lo = (a == 0) & (kap < -kl - tol)
    add_hi[:, 0] = False; add_lo[:, 0] = False
    viol = np.where(add_hi, kap - kl, np.where(add_lo, -kl - kap, 0.0))
    if rule == "peaks":
        pad = np.pad(viol, ((0, 0), (1, 1)))
        peak = (viol >= pad[:, :-2]) & (viol >= pad[:, 2:])
        add_hi &= peak; add_lo &= peak
    nk = np.where(release, 0, np.where(add_hi, 1, np.where(add_lo, -1, a))).astype(np.int8)
    new["k"] = nk
    nlam["k"] = np.where(nk != 0, np.where(a != 0, y, 0.0), 0.0)
    worst = np.maximum(worst, np.maximum(viol.max(axis=1), np.where(release, np.abs(y), 0.0).max(axis=1)))
    pin_err = np.where((a != 0) & ~release, np.abs(kap - a * kl), 0.0).max(axis=1)
    # end rows
    for key, val, lo, up in (("el", x[:, -1, 0], -p.end_l * np.ones(B), p.end_l * np.ones(B)), ("ep", x[:, -1, 1], d["psi_lo"], d["psi_hi"])):
        a = act[key]
        bnd = np.where(a > 0, up, lo)
        y = np.where(a != 0, lam[key] + (val - bnd) / p.delta, 0.0)
        release = (a != 0) & (a * y < -tol)
        add_hi = (a == 0) & (val > up + tol); add_lo = (a == 0) & (val < lo - tol)
        na = np.where(release, 0, np.where(add_hi, 1, np.where(add_lo, -1, a))).astype(np.int8)
        new[key] = na
        nlam[key] = np.where(na != 0, y, 0.0)
        worst = np.maximum(worst, np.where(add_hi, val - up, np.where(add_lo, lo - val, np.where(release, np.abs(y), 0.0))))
        pin_err = np.maximum(pin_err, np.where((a != 0) & ~release, np.abs(val - bnd), 0.0))
    changed = np.zeros(B, dtype=bool)
    for k in act:
        diff = new[k] != act[k]
        changed |= diff.reshape(B, -1).any(axis=1)
    return new, nlam, changed, worst, pin_err


def solve_pass(d, p, act0=None, max_rounds=40, rule="all", verbose=False, lam0=None):
    B, N = d["lo_f"].shape
    if act0 is None:
        act = dict(f=np.zeros((B, N), np.int8), r=np.zeros((B, N), np.int8), k=np.zeros((B, N), np.int8), el=np.zeros(B, np.int8), ep=np.zeros(B, np.int8))
    else:
        act = {k: v.copy() for k, v in act0.items()}
    lam = dict(k=np.zeros((B, N)), el=np.zeros(B), ep=np.zeros(B)) if lam0 is None else {k: v.copy() for k, v in lam0.items()}
    done = np.zeros(B, dtype=bool)
    rounds = np.zeros(B, dtype=np.int32)
    xs = np.zeros((B, N, 3)); us = np.zeros((B, N - 1))
    for rnd in range(max_rounds):
        x, u = riccati_round(d, act, lam, p)
        new, nlam, changed, worst, pin_err = new_active_set(d, act, lam, x, p, rule)
        fin = ~changed & (pin_err <= 1e-9) & ~done
        live = ~done
        xs[live] = x[live]; us[live] = u[live]
        rounds[live] += 1
        done |= fin
        # lanes that are done keep their set
        for k in act:
            m = done if act[k].ndim == 1 else done[:, None]
            act[k] = np.where(m, act[k], new[k])
        for k in lam:
            m = done if lam[k].ndim == 1 else done[:, None]
            lam[k] = np.where(m, lam[k], nlam[k])
        if verbose:
            print(f"  round {rnd + 1}: live {int(live.sum())}, finished now {int(fin.sum())}, worst {worst[live].max():.3e}, pin {pin_err[live].max():.2e}")
        if done.all():
            break
    return xs, us, act, rounds, done, lam


def ipm_pass(d, p, max_iter=30, mu_stop=1e-6, sigma_lo=0.05, sigma_hi=0.4, verbose=False, mu0=1.0, w0=1.0, warm=None, mu_w=1e-3):
    """Primal-dual interior point iterations on the rows (dynamics exact through the Riccati solve).  Row state per two-sided row:
    g (row value incl. slack), t_l = g - lo + r_l .. (kept > 0), z_l, z_u.  Returns the last point, an active-set guess and the
    iteration count per QP."""
    B, N = d["lo_f"].shape
    Lf, Lr = p.front_length, p.rear_length
    kl = d["kl"][:, None] * np.ones((1, N))
    has_ep = d["psi_hi"] < 1e29
    rows = {   # name: (lo, up, soft?)
        "f": (d["lo_f"], d["up_f"], True), "r": (d["lo_r"], d["up_r"], True), "k": (-kl, kl, False),
        "el": (-p.end_l * np.ones(B), p.end_l * np.ones(B), False), "ep": (np.where(has_ep, d["psi_lo"], -1.0), np.where(has_ep, d["psi_hi"], 1.0), False)}
    # --- initial point: every row pulled to the middle of its box with a unit weight
    W = {}
    for key in ("f", "r", "k"):
        lo, up, soft = rows[key]
        W["w" + key] = np.full((B, N), p.w_s * w0 / (p.w_s + w0) if soft else w0); W["t" + key] = 0.5 * (lo + up)
    W["wk"][:, 0] = 0.0
    W["wel"] = np.full(B, w0); W["tel"] = np.zeros(B)
    W["wep"] = np.where(has_ep, w0, 0.0); W["tep"] = 0.5 * (rows["ep"][0] + rows["ep"][1])
    theta = 0.05
    st = {}

    def row_value(key, xx):
        if key == "f": return xx[:, :, 0] + Lf * xx[:, :, 1]
        if key == "r": return xx[:, :, 0] + Lr * xx[:, :, 1]
        if key == "k": return xx[:, :, 2]
        if key == "el": return xx[:, -1, 0]
        return xx[:, -1, 1]

    if warm is None:
        x, u = riccati_solve(d, W, p)
        for key, (lo, up, soft) in rows.items():
            v = row_value(key, x)
            wd = up - lo
            if soft:
                g = np.clip(v, lo + theta * wd, up - theta * wd)
            else:
                g = v
            tl = np.maximum(g - lo, theta * wd); tu = np.maximum(up - g, theta * wd)
            st[key] = dict(g=g, tl=tl, tu=tu, zl=mu0 / tl, zu=mu0 / tu)
    else:
        x, u, act_w, lam_w = warm          # the previous pass's exact solution, its active set and hard-row multipliers
        sq = np.sqrt(mu_w)
        for key, (lo, up, soft) in rows.items():
            v = row_value(key, x)
            if soft:
                y = np.where(act_w[key] > 0, p.w_s * (v - up), np.where(act_w[key] < 0, p.w_s * (v - lo), 0.0))
                g = np.clip(v, lo, up)
            else:
                y = lam_w[key]; g = v
            zu = np.maximum(y, 0.0); zl = np.maximum(-y, 0.0)
            tl_min = mu_w / np.maximum(zl, sq); tu_min = mu_w / np.maximum(zu, sq)
            if soft:
                g = np.minimum(np.maximum(g, lo + tl_min), up - tu_min)
                tl = g - lo; tu = up - g
            else:
                tl = np.maximum(g - lo, tl_min); tu = np.maximum(up - g, tu_min)
            zl = np.maximum(zl, mu_w / np.maximum(tl, sq)); zu = np.maximum(zu, mu_w / np.maximum(tu, sq))
            st[key] = dict(g=g, tl=tl, tu=tu, zl=zl, zu=zu)
    live_row = {"f": np.ones((B, N), bool), "r": np.ones((B, N), bool), "k": np.ones((B, N), bool), "el": np.ones(B, bool), "ep": has_ep.copy()}
    live_row["k"][:, 0] = False
    iters = np.zeros(B, np.int32)
    done = np.zeros(B, bool)
    alpha_prev = np.ones(B)
    xs = x.copy(); us = u.copy()
    for it in range(max_iter):
        # mu per QP
        num = np.zeros(B); cnt = np.zeros(B)
        for key in rows:
            r_ = st[key]; m = live_row[key]
            comp = np.where(m, r_["tl"] * r_["zl"] + r_["tu"] * r_["zu"], 0.0)
            num += comp.reshape(B, -1).sum(axis=1); cnt += 2 * m.reshape(B, -1).sum(axis=1)
        mu = num / cnt
        res = np.zeros(B)
        for key, (lo, up, soft) in rows.items():
            r_ = st[key]; m = live_row[key]
            rr = np.where(m, np.maximum(np.abs(r_["g"] - lo - r_["tl"]), np.abs(up - r_["g"] - r_["tu"])), 0.0)
            res = np.maximum(res, rr.reshape(B, -1).max(axis=1))
        newly = ~done & (mu < mu_stop) & (res < 1e-6)
        done |= newly
        if verbose:
            print(f"  ipm {it}: live {int((~done).sum())}, mu max {mu[~done].max() if (~done).any() else 0:.2e}, res {res.max():.2e}, alpha min {alpha_prev.min():.3f}")
        if done.all():
            break
        sigma = np.where(alpha_prev > 0.9, sigma_lo, sigma_hi)
        if it == 0:
            sigma = np.full(B, sigma_hi)
        W = {}
        aux = {}
        for key, (lo, up, soft) in rows.items():
            r_ = st[key]; m = live_row[key]
            sm = sigma[:, None] * mu[:, None] if r_["g"].ndim == 2 else sigma * mu
            rl = r_["g"] - lo - r_["tl"]; ru = up - r_["g"] - r_["tu"]
            dd = r_["zu"] / r_["tu"] + r_["zl"] / r_["tl"]
            e = sm * (1.0 / r_["tu"] - 1.0 / r_["tl"]) - (r_["zu"] / r_["tu"]) * ru + (r_["zl"] / r_["tl"]) * rl
            tgt = r_["g"] - e / dd
            w = p.w_s * dd / (p.w_s + dd) if soft else dd
            W["w" + key] = np.where(m, w, 0.0); W["t" + key] = np.where(m, tgt, 0.0)
            aux[key] = (dd, tgt, rl, ru, sm)
        xp, up_ = riccati_solve(d, W, p)
        alpha = np.ones(B)
        steps = {}
        for key, (lo, up, soft) in rows.items():
            r_ = st[key]; m = live_row[key]
            dd, tgt, rl, ru, sm = aux[key]
            v = row_value(key, xp)
            if soft:
                gp = v - (dd / (p.w_s + dd)) * (v - tgt)          # + s^+
            else:
                gp = v
            dg = gp - r_["g"]
            dtl = dg + rl; dtu = -dg + ru
            dzl = (sm - r_["tl"] * r_["zl"] - r_["zl"] * dtl) / r_["tl"]
            dzu = (sm - r_["tu"] * r_["zu"] - r_["zu"] * dtu) / r_["tu"]
            steps[key] = (dg, dtl, dtu, dzl, dzu)
            for val, dv in ((r_["tl"], dtl), (r_["tu"], dtu), (r_["zl"], dzl), (r_["zu"], dzu)):
                ratio = np.where(m & (dv < 0), -val / np.where(dv < 0, dv, -1.0), np.inf)
                alpha = np.minimum(alpha, 0.995 * ratio.reshape(B, -1).min(axis=1))
        alpha = np.where(done, 0.0, alpha)
        for key in rows:
            r_ = st[key]
            dg, dtl, dtu, dzl, dzu = steps[key]
            a = alpha[:, None] if dg.ndim == 2 else alpha
            r_["g"] = r_["g"] + a * dg; r_["tl"] = r_["tl"] + a * dtl; r_["tu"] = r_["tu"] + a * dtu
            r_["zl"] = r_["zl"] + a * dzl; r_["zu"] = r_["zu"] + a * dzu
        live = ~done
        xs[live] = xs[live] + alpha[live, None, None] * (xp[live] - xs[live])
        us[live] = us[live] + alpha[live, None] * (up_[live] - us[live])
        iters[live] += 1
        alpha_prev = alpha
    # active-set guess: a side is active when its multiplier outweighs its slack
    act = {}
    lam = {}
    for key, (lo, up, soft) in rows.items():
        r_ = st[key]; m = live_row[key]
        a = np.where(m & (r_["zu"] > r_["tu"]), 1, np.where(m & (r_["zl"] > r_["tl"]), -1, 0)).astype(np.int8)
        act[key] = a
        if not soft:
            lam[key] = np.where(a > 0, r_["zu"], np.where(a < 0, -r_["zl"], 0.0))
    return xs, us, act, lam, iters, done


def unpack_lin(x):
    return x.copy()


def solve_paths(ref, bounds, scal, p, rule="all", max_rounds=40, verbose=False, ipm=False, mu_stop=1e-6):
    lin = np.stack([np.zeros_like(ref[:, :, 1]), np.zeros_like(ref[:, :, 1]), ref[:, :, 1]], axis=2)
    d = stage_data(ref, lin, bounds, scal, p)
    infeasible = np.abs(d["x0"][:, 2]) > d["kl"] + 1e-12
    if ipm:
        xi, ui, acti, lami, it_ipm, okipm = ipm_pass(d, p, verbose=verbose, mu_stop=mu_stop)
        x1, u1, act, r1, ok1, lam1 = solve_pass(d, p, acti, max_rounds, rule, verbose, lam0=lami)
        print(f"  ipm iterations: mean {it_ipm.mean():.2f} median {np.median(it_ipm):.0f} p99 {np.percentile(it_ipm, 99):.0f} max {it_ipm.max()}; ipm converged {int(okipm.sum())}")
    else:
        x1, u1, act, r1, ok1, lam1 = solve_pass(d, p, None, max_rounds, rule, verbose)
    d2 = stage_data(ref, x1, bounds, scal, p)
    if ipm and os.environ.get("WARM_IPM"):
        xi, ui, acti, lami, it2, okipm = ipm_pass(d2, p, verbose=verbose, mu_stop=mu_stop, warm=(x1, u1, act, lam1), mu_w=float(os.environ["WARM_IPM"]))
        print(f"  warm ipm iterations: mean {it2.mean():.2f} median {np.median(it2):.0f} p99 {np.percentile(it2, 99):.0f} max {it2.max()}; converged {int(okipm.sum())}; "
              f"max over groups of 64: {np.mean([it2[i:i + 64].max() for i in range(0, len(it2), 64)]):.1f}")
        x2, u2, act2, r2, ok2, lam2 = solve_pass(d2, p, acti, max_rounds, rule, verbose, lam0=lami)
    else:
        x2, u2, act2, r2, ok2, lam2 = solve_pass(d2, p, act, max_rounds, rule, verbose, lam0=lam1 if os.environ.get('LAM2') else None)
    return dict(x=x2, u=u2, x1=x1, rounds1=r1, rounds2=r2, ok=ok1 & ok2 & ~infeasible, act=act2)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    profile = sys.argv[3] if len(sys.argv) > 3 else "uniform"
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else None
    rule = sys.argv[5] if len(sys.argv) > 5 else "all"
    from path_optimizer_2_amd.synth import BASE_SEED, make_batch
    b = make_batch(batch, n, profile, seed=seed if seed is not None else BASE_SEED)
    p = Prm()
    t0 = time.time()
    r = solve_paths(b["ref"], b["bounds"], b["scal"], p, rule=rule if rule != "ipm" else "all", verbose=batch <= 64, ipm=rule == "ipm", mu_stop=float(os.environ.get("MU_STOP", "1e-6")))
    print(f"prototype: {time.time() - t0:.1f} s; ok {int(r['ok'].sum())}/{batch}")
    for nm in ("rounds1", "rounds2"):
        v = r[nm]
        print(f"  {nm}: mean {v.mean():.2f} median {np.median(v):.0f} p90 {np.percentile(v, 90):.0f} p99 {np.percentile(v, 99):.0f} max {v.max()}   "
              f"max over groups of 64: mean {np.mean([v[i:i + 64].max() for i in range(0, batch, 64)]):.1f}")
    if os.environ.get("NO_ORACLE"):
        return
    import pqp_oracle_c as OC
    prm = OC.params(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)
    k = min(batch, int(os.environ.get("ORACLE_N", "128")))
    t0 = time.time()
    o = OC.solve_batch(prm, b["ref"][:k], b["bounds"][:k], b["scal"][:k], passes=1)
    err = np.abs(o["out"][:, :, 3:5] - r["x"][:k, :, 0:2]).max(axis=(1, 2))
    errk = np.abs(o["out"][:, :, 5] - r["x"][:k, :, 2]).max(axis=1)
    print(f"oracle ({time.time() - t0:.1f} s, solved {o['solved']}/{k}): |l, psi| error median {np.median(err):.2e} p99 {np.percentile(err, 99):.2e} max {err.max():.2e}; "
          f"kappa max {errk.max():.2e}; worst QPs {np.argsort(err)[-4:]}")
    bad = np.where(r["ok"][:k] & (err > 1e-5))[0]
    print("converged but > 1e-5 from the oracle:", bad[:16], err[bad][:16])


if __name__ == "__main__":
    main()
