"""The exact smoother kernels (S2 TensionSmoother, S3 postSmooth) on long lines: HIP-event kernel time and factorisation counts per size, through the
register kernels (up to 1024 points / layers) and the HBM-workspace kernels beyond (tension_exact_kernel<0>, post_exact_kernel<0>).
Usage: python tools/bench_smoothers_long.py [batch] [n ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, tension_inputs

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sizes = [int(v) for v in sys.argv[2:]] or [500, 1000, 1024, 1025, 1500, 2000, 4000]
prm = capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25)
for n in sizes:
    tc = [tension_inputs(n, seed=b) for b in range(16)]
    x, y, ang, cl = (np.stack([tc[b % 16][k] for b in range(batch)]) for k in (0, 1, 2, 5))
    pc = [post_inputs(n, seed=b) for b in range(16)]
    s, lb, ub = (np.stack([pc[b % 16][k] for b in range(batch)]) for k in (0, 1, 2))
    l0 = np.array([pc[b % 16][3] for b in range(batch)])
    h = capi.Handle(prm, device=0, max_batch=batch, max_n=n)
    for name, run in (("exact tension (S2)", lambda: h.smooth_tension(x, y, ang, cl, info=True)),
                      ("exact post (S3)   ", lambda: h.post_smooth_var(s, lb, ub, l0, np.full(batch, n, dtype=np.int32), info=True))):
        ms = []
        for _ in range(3):
            r = run()
            ms.append(h.last_kernel_ms())
        f = r["info"][:, 5]
        where = "registers" if n <= 1024 else "HBM workspace"
        print(f"{name} n = {n:5d} batch {batch}: {min(ms):8.3f} ms = {batch / min(ms):8.1f} k scenarios/s; solved {(r['status'] == 1).sum()}/{batch}; "
              f"factorisations mean {f.mean():.1f} max {f.max():.0f}; lane state in {where}")
    h.close()
