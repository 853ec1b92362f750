"""Throughput of the three smoother QPs on one GPU (device-resident inputs, HIP-event kernel time of the banded core)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, tension_inputs

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda", 0)
cases = [tension_inputs(n, seed=b) for b in range(64)]
rep = lambda k: torch.from_numpy(np.stack([cases[b % 64][k] for b in range(batch)])).to(dev)
x, y, ang, kk, s, cl = (rep(k) for k in range(6))
pc = [post_inputs(n, seed=b) for b in range(64)]
ps = torch.from_numpy(np.stack([pc[b % 64][0] for b in range(batch)])).to(dev)
plb = torch.from_numpy(np.stack([pc[b % 64][1] for b in range(batch)])).to(dev)
pub = torch.from_numpy(np.stack([pc[b % 64][2] for b in range(batch)])).to(dev)
pl0 = torch.from_numpy(np.array([pc[b % 64][3] for b in range(batch)])).to(dev)
ox, oy, os_ = (torch.zeros((batch, n), dtype=torch.float64, device=dev) for _ in range(3))
st = torch.zeros(batch, dtype=torch.int32, device=dev); it = torch.zeros(batch, dtype=torch.int32, device=dev)
p = lambda t: capi.C.c_void_p(t.data_ptr())
for label, prm in (("reference setting (eps 1e-3, no polish)", capi.default_params(eps_abs=1e-3, eps_rel=1e-3)),
                   ("... rho adapted every 50 iterations", capi.default_params(eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=50)),
                   ("... rho adapted every 25 iterations", capi.default_params(eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=25)),
                   ("polish = 2: TensionSmoother2 exact (Riccati sweep), the others plain ADMM", capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=2, adaptive_rho_interval=25, polish_refine_iter=2)),
                   ("polish = 1: exact optima (S1 Riccati sweep, S2 and S3 box QPs in a wavefront)", capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25, polish_refine_iter=2))):
    h = capi.Handle(prm, device=0, max_batch=batch, max_n=n)
    lib = h.lib
    runs = {
        "tension2": lambda: lib.pqp_smooth_tension2_device(h._h, batch, n, p(x), p(y), p(ang), p(kk), p(s), p(ox), p(oy), p(os_), p(st), p(it), None),
        "tension": lambda: lib.pqp_smooth_tension_device(h._h, batch, n, p(x), p(y), p(ang), p(cl), p(ox), p(oy), p(os_), p(st), p(it), None),
        "post": lambda: lib.pqp_post_smooth_device(h._h, batch, n, p(ps), p(plb), p(pub), p(pl0), p(ox), p(st), p(it), None),
    }
    for name, fn in runs.items():
        for _ in range(2):
            assert fn() == 0
        h.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        h.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"{label:45s} {name:9s} batch {batch} n {n}: {batch / dt:10.0f} QP/s  ({dt * 1e3:.2f} ms/batch, solved {(st == 1).sum().item()}/{batch}, mean iters {it.double().mean().item():.0f})")
    h.close()
