"""Does the lane-per-waypoint kernel gain from starting a cycle's first solve from the previous cycle's final iterate (warm == 1 with lin = NULL)?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch, jitter_batch
dev = torch.device("cuda", 0)
batch, n = 1024, 80
host = make_batch(batch, n)
prm = capi.production_params()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ref = t(host["ref"])
var = [(t(jitter_batch(host, v)["bounds"]), t(jitter_batch(host, v)["scal"])) for v in range(8)]
for warm in (False, True):
    h = capi.Handle(prm, device=0, max_batch=batch, max_n=n)
    h.set_option(capi.OPT_STORE_WARM, 1 if warm else 0)
    h.set_option(capi.OPT_ORDER_BY_COST, 1)
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev); st = torch.zeros(batch, dtype=torch.int32, device=dev)
    it = torch.zeros(batch, dtype=torch.int32, device=dev); info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    h.solve_device(batch, n, ref, var[0][0], var[0][1], out, passes=1, status=st, iters=it, info=info); h.sync()
    outs = []
    t0 = time.perf_counter()
    for k in range(1, 41):
        b, s = var[k % 8]
        h.solve_device(batch, n, ref, b, s, out, passes=1, warm=warm, status=st, iters=it, info=info)
    h.sync()
    dt = (time.perf_counter() - t0) / 40
    inf = info.cpu().numpy()
    print(f"warm={warm}: {dt*1e3:.3f} ms/step = {batch/dt/1e6:.2f} M paths/s; solved {(st.cpu().numpy()==1).sum()}; admm iters mean {it.cpu().numpy().mean():.1f}; kkt solves {inf[:,5].mean():.1f} factorisations {inf[:,6].mean():.1f}")
    outs.append(out.cpu().numpy())
    if warm:
        hc = capi.Handle(prm, device=0, max_batch=batch, max_n=n); hc.set_option(capi.OPT_STORE_WARM, 0)
        o2 = torch.zeros_like(out); b, s = var[40 % 8]
        hc.solve_device(batch, n, ref, b, s, o2, passes=1); hc.sync()
        print("   warm-cycle result vs cold result: max |diff| = %.2e" % np.abs(o2.cpu().numpy() - outs[0]).max())
