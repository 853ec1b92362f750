"""Active-set rounds of the exact TensionSmoother kernel (info[5]) and its time per round.  Usage: python tools/smoother_rounds.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d_ in ("", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d_))
from path_optimizer_2_amd import capi
from smoother_cases import tension_inputs
C = capi.C
for n in (48, 80, 200):
    B = 64
    cases = [tension_inputs(n, seed=b) for b in range(B)]
    x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
    h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1), max_batch=B, max_n=n)
    dev = torch.device("cuda", 0)
    t = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d = [t(a) for a in (x, y, ang, cl)]
    o = [torch.zeros((B, n), dtype=torch.float64, device=dev) for _ in range(3)]
    st, it = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
    info = torch.zeros((B, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    p = lambda a: C.c_void_p(a.data_ptr())
    for _ in range(3):
        assert h.lib.pqp_smooth_tension_device(h._h, B, n, p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(o[0]), p(o[1]), p(o[2]), p(st), p(it), p(info)) == 0
    h.sync()
    f = info.cpu().numpy()[:, 5]
    print(f"n {n}: rounds mean {f.mean():.1f} p90 {np.percentile(f, 90):.0f} max {f.max():.0f}; kernel {h.last_kernel_ms()*1e3:.0f} us for {B} QPs -> {h.last_kernel_ms()*1e3/f.max():.1f} us per round of the slowest")
    h.close()
