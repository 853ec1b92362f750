"""The exact TensionSmoother / postSmooth kernels (polish = 1) with and without PQP_OPT_CARRY_CYCLES on lines that move a little from one planning
cycle to the next (points shifted by up to 5 cm along their normals, clearances scaled by 1 +- 5 %).  Usage: python tools/bench_smoothers_carry.py [batch] [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, tension_inputs

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
p = lambda t: capi.C.c_void_p(t.data_ptr())
for n in ([int(sys.argv[2])] if len(sys.argv) > 2 else [48, 80, 200]):
    cases = [tension_inputs(n, seed=b) for b in range(64)]
    base = [np.stack([cases[b % 64][k] for b in range(batch)]) for k in range(6)]
    pc = [post_inputs(n, seed=b) for b in range(64)]
    pbase = [np.stack([pc[b % 64][k] for b in range(batch)]) for k in range(3)] + [np.array([pc[b % 64][3] for b in range(batch)])]
    rng = np.random.default_rng(1)
    cycles = []
    for v in range(6):
        sh = rng.uniform(-0.05, 0.05, (batch, n)) if v else np.zeros((batch, n))
        x = base[0] + sh * np.cos(base[2] + np.pi / 2); y = base[1] + sh * np.sin(base[2] + np.pi / 2)
        cl = base[5] * (1 + (rng.uniform(-0.05, 0.05, (batch, 1)) if v else 0.0))
        lb = pbase[1] * (1 + (rng.uniform(-0.05, 0.05, (batch, 1)) if v else 0.0)); ub = pbase[2] * (1 + (rng.uniform(-0.05, 0.05, (batch, 1)) if v else 0.0))
        cycles.append([torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, y, base[2], cl, pbase[0], lb, ub, pbase[3])])
    ox, oy, os_ = (torch.zeros((batch, n), dtype=torch.float64, device=dev) for _ in range(3))
    st = torch.zeros(batch, dtype=torch.int32, device=dev); it = torch.zeros(batch, dtype=torch.int32, device=dev); info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
    for carry in (0, 1):
        h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25, polish_refine_iter=2), device=0, max_batch=batch, max_n=n)
        h.set_option(capi.OPT_CARRY_CYCLES, carry)
        lib = h.lib
        for name in ("tension", "post"):
            def run(c):
                if name == "tension":
                    return lib.pqp_smooth_tension_device(h._h, batch, n, p(c[0]), p(c[1]), p(c[2]), p(c[3]), p(ox), p(oy), p(os_), p(st), p(it), p(info))
                return lib.pqp_post_smooth_device(h._h, batch, n, p(c[4]), p(c[5]), p(c[6]), p(c[7]), p(ox), p(st), p(it), p(info))
            torch.cuda.synchronize()
            for v in range(2):
                assert run(cycles[v]) == 0
            h.sync()
            t0 = time.perf_counter()
            reps = 10
            for k in range(reps):
                run(cycles[2 + k % 4])
            h.sync()
            dt = (time.perf_counter() - t0) / reps
            print(f"n {n:3d} {name:8s} carry {carry}: {batch / dt / 1e3:9.0f} k QP/s ({dt * 1e3:.3f} ms per {batch}); solved {(st == 1).sum().item()}/{batch}; factorisations mean {info[:, 5].mean().item():.1f} max {info[:, 5].max().item():.0f}")
        h.close()
