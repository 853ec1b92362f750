"""GPU micro-costs of the solve kernel's building blocks (run on the GPU box):
   t_iter   one ADMM iteration           (fixed 200 iterations, no adaptation)
   t_factor one re-factorisation         (adaptive-rho forced every 25 iterations)
   t_setup  assemble + Ruiz + factor     (max_iter = 25)
Usage: python tools/kernel_costs.py [batch] [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
host = make_batch(batch, n)
dev = torch.device("cuda", 0)
ref, bounds, scal = (torch.from_numpy(host[k]).to(dev) for k in ("ref", "bounds", "scal"))
out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
iters = torch.zeros(batch, dtype=torch.int32, device=dev)

def timed(**kw):
    prm = capi.default_params(**kw)
    h = capi.Handle(prm, device=0, max_batch=batch, max_n=n)
    ms = []
    for _ in range(4):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=0, iters=iters)
        ms.append(h.last_kernel_ms())
    h.sync()
    it = iters.cpu().numpy()
    h.close()
    return float(np.median(ms[1:])), it

base = dict(eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=0, check_termination=25)
t25, it = timed(max_iter=25, **base)
t225, it2 = timed(max_iter=225, **base)
t_iter = (t225 - t25) / 200
tf, it3 = timed(max_iter=225, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=1, adaptive_rho_interval=25, adaptive_rho_tolerance=1.0, check_termination=25)
t_factor = (tf - t225) / 9
t_nosc, _ = timed(max_iter=25, scaling=0, **base)
print(f"batch {batch} n {n}: kernel(25 it) {t25*1e3:.1f} us  kernel(225 it) {t225*1e3:.1f} us -> t_iter {t_iter*1e3:.2f} us;"
      f" t_factor {t_factor*1e3:.1f} us; setup(assemble+ruiz+factor) ~ {(t25-25*t_iter)*1e3:.1f} us; ruiz ~ {(t25-t_nosc)*1e3:.1f} us")
