"""Runtime parameters of the production setting on the host emulation of the lane-per-waypoint kernel (tools/emu_policy_sweep.py is the compile-time counterpart): modelled cost
4 x reduced solves + 13 x factorisations over 4 seeds x 8192 QPs.  Usage: python tools/emu_param_sweep.py      (CPU only)"""
import ctypes as C, os, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import emu_policy_sweep as E
from path_optimizer_2_amd.capi import PqpParams
from path_optimizer_2_amd.synth import make_batch
def task(a):
    lib_path, n, profile, seed, first, count, over = a
    lib = C.CDLL(lib_path)
    prm = PqpParams(); lib.pqp_emu_production_params(C.byref(prm))
    for k, v in over.items(): setattr(prm, k, v)
    h = make_batch(count, n, profile, seed=seed, first_qp=first)
    vp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.c_void_p)
    B = count
    out = np.zeros((B, n, 7)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32); info = np.zeros((B, 8))
    wx = np.zeros((B, n, 6)); wy = np.zeros((B, n, 6)); wye = np.zeros((B, 2)); wrho = np.zeros(B)
    ref, bounds, scal = (np.ascontiguousarray(h[k]) for k in ("ref", "bounds", "scal"))
    lib.pqp_emu_set_counts(None)
    lib.pqp_emu_path_solve(C.byref(prm), B, n, vp(ref), None, vp(bounds), vp(scal), 1, 0, vp(out), vp(st), vp(it), vp(info), vp(wx), vp(wy), vp(wye), vp(wrho))
    return info[:, 5].copy(), info[:, 6].copy(), st.copy(), info[:, 4].copy(), it.copy()
if __name__ == "__main__":
    name, lib = E.build("base")
    combos = [dict(), dict(polish_every=7, check_termination=7, adaptive_rho_interval=7), dict(polish_every=9, check_termination=9, adaptive_rho_interval=9),
              dict(polish_every=6, check_termination=6, adaptive_rho_interval=6), dict(polish_every=8, check_termination=4, adaptive_rho_interval=8),
              dict(polish_every=8, check_termination=8, adaptive_rho_interval=4), dict(polish_lazy=4), dict(polish_lazy=6), dict(polish_lazy=7), dict(polish_refine_iter=1), dict(rho=0.2), dict(rho=0.05), dict(scaling=3), dict(scaling=5)]
    with ProcessPoolExecutor(96) as ex:
        for n, profile in ((80, "uniform"), (120, "varied")):
            for over in combos:
                tasks = [(lib, n, profile, 1000 + s, f, 512, over) for s in range(4) for f in range(0, 8192, 512)]
                r = list(ex.map(task, tasks))
                k = np.concatenate([x[0] for x in r]); f = np.concatenate([x[1] for x in r]); st = np.concatenate([x[2] for x in r]); pol = np.concatenate([x[3] for x in r]); it = np.concatenate([x[4] for x in r])
                c = 4 * k + 13 * f
                print(f"n {n} {profile} {over}: bad {int(((st != 1) | (pol != 2)).sum())} cost mean {c.mean():.1f} p99 {np.percentile(c, 99):.0f} p99.99 {np.percentile(c, 99.99):.0f} | solves {k.mean():.2f} fac {f.mean():.2f} admm iters {it.mean():.1f}", flush=True)
