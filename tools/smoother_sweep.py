"""Many distinct smoother QPs through the exact setting (polish = 1: active-set solve from the cold start, ADMM only as the fallback):
how many end solved, how many needed the fallback (iters > 0), the worst deviation from the reference setting's ADMM answer.
Usage: python tools/smoother_sweep.py [cases=2048] [n=80]   (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, tension_inputs

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
tc = [tension_inputs(n, seed=1000 + b) for b in range(cases)]
x, y, ang, kk, s, cl = (np.stack([c[k] for c in tc]) for k in range(6))
pc = [post_inputs(n, seed=5000 + b) for b in range(cases)]
ps = np.stack([c[0] for c in pc]); plb = np.stack([c[1] for c in pc]); pub = np.stack([c[2] for c in pc]); pl0 = np.array([c[3] for c in pc])
exact = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25, polish_refine_iter=2), max_batch=cases, max_n=n)
ref = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=25), max_batch=cases, max_n=n)
for name, call in (("tension2", lambda h: h.smooth_tension2(x, y, ang, kk, s)), ("tension", lambda h: h.smooth_tension(x, y, ang, cl)),
                   ("post", lambda h: h.post_smooth(ps, plb, pub, pl0))):
    a, b = call(exact), call(ref)
    key = "l" if name == "post" else "x"
    dev = np.abs(a[key] - b[key]).max(axis=1)
    print(f"{name:9s} n {n}: {cases} QPs, exact setting solved {int((a['status'] == 1).sum())}, fallback to ADMM (iters > 0) {int((a['iters'] > 0).sum())} "
          f"(max iters {int(a['iters'].max())}); reference setting solved {int((b['status'] == 1).sum())}, mean iters {b['iters'].mean():.0f}; "
          f"|exact - reference setting| median {np.median(dev):.2e} max {dev.max():.2e}")
