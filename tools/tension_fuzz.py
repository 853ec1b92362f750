"""The exact TensionSmoother kernel (interior start + exact rounds) on many DISTINCT random lines per size: every line must end SOLVED, the factorisation
count stays bounded, and a sample of the results is checked by the KKT certificate of the oracle's matrices (tests/smoother_cases.py; no solver involved).
Usage: python tools/tension_fuzz.py [lines=1024]      (run on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from path_optimizer_2_amd import capi
from smoother_cases import tension_inputs, tension_kkt_certificate

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prm = capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25, polish_refine_iter=2)
for n in (24, 48, 80, 130, 200, 300, 384, 500, 700, 1000):
    cases = [tension_inputs(n, seed=5000 + b) for b in range(lines)]
    x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
    h = capi.Handle(prm, max_batch=lines, max_n=n)
    r = h.smooth_tension(x, y, ang, cl, info=True)
    h.close()
    fac, ipm = r["info"][:, 5], r["info"][:, 3]
    worst = max(tension_kkt_certificate(x[b], y[b], ang[b], cl[b], r["x"][b], r["y"][b]) for b in list(range(0, lines, max(1, lines // 12))) + [int(fac.argmax())])
    print(f"n {n:3d}: {lines} lines, solved {(r['status'] == 1).sum()}, factorisations mean {fac.mean():.1f} max {fac.max():.0f} (interior iterations max {ipm.max():.0f}), KKT certificate of 13 lines incl. the slowest <= {worst:.1e}", flush=True)
