"""Throughput of the layered DP corridor search (+ the postSmooth QP it feeds) on one GPU.
Usage: python tools/bench_dp.py [batch] [n_maps] [length]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import scenes
from path_optimizer_2_amd import capi

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_maps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
length = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
Lmax = 64
h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), device=0, max_batch=batch, max_n=Lmax)
sc = scenes.build(h, range(n_maps), 10)
geom = sc["geom"]
dev = torch.device("cuda", 0)
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
rep = lambda key: np.stack([sc[key][b % n_maps] for b in range(batch)])
tab, ext = t(rep("tab")), t(rep("ext"))
m = tab.shape[2]
dist = t(np.transpose(sc["dist"], (0, 2, 1)), torch.float32)
map_of = torch.arange(batch, dtype=torch.int32, device=dev) % n_maps
lens = torch.full((batch,), length, dtype=torch.float64, device=dev)
start = t(np.stack([np.array([sc["ref"][b % n_maps][0, 3] + 0.2, sc["ref"][b % n_maps][0, 4] + 0.5, sc["ref"][b % n_maps][0, 2]]) for b in range(batch)]))
z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
ls, lb, ub, vl, cnt = z(batch, Lmax), z(batch, Lmax), z(batch, Lmax), z(batch), z(batch, dt=torch.int32)
lib, hh = h.lib, h._h
p = lambda x: capi.C.c_void_p(x.data_ptr())
prm = capi.PqpDpParams(); lib.pqp_dp_default_params(capi.C.byref(prm))
ms = []
for _ in range(8):
    assert lib.pqp_dp_corridor_device(hh, batch, m, Lmax, p(tab), p(ext), p(lens), p(start), p(dist), p(map_of), capi.C.byref(geom), capi.C.byref(prm),
                                      p(ls), p(lb), p(ub), p(cnt), p(vl)) == 0
    ms.append(h.last_kernel_ms())
h.sync()
k_ms = float(np.median(ms[2:]))
c = cnt.cpu().numpy()
print(f"dp corridor search: batch {batch}, {n_maps} maps, {length} m: {k_ms * 1e3:.1f} us per launch = {batch / k_ms * 1e3:.0f} scenarios/s; layers {c.min()}..{c.max()}")
# the postSmooth QP on its output (all scenarios of this bench reach the same number of layers)
k = int(c.min())
if k >= 4 and c.max() == k:
    out_l = z(batch, k); st = z(batch, dt=torch.int32); it = z(batch, dt=torch.int32)
    a = lambda x: x[:, :k].contiguous()
    lsk, lbk, ubk = a(ls), a(lb), a(ub)
    qs = []
    for _ in range(5):
        assert lib.pqp_post_smooth_device(hh, batch, k, p(lsk), p(lbk), p(ubk), p(vl), p(out_l), p(st), p(it), None) == 0
        qs.append(h.last_kernel_ms())
    h.sync()
    print(f"postSmooth QP on it (m = {k} layers, eps 1e-3): {float(np.median(qs[1:])) * 1e3:.1f} us per launch, solved {int((st == 1).sum())}/{batch}, mean iterations {it.double().mean().item():.0f}")
