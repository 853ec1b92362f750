"""Throughput of the corridor-bounds step on one GPU: `batch` scenarios of n waypoints over `n_maps` distinct distance maps
(device-resident inputs, HIP-event kernel time).  Usage: python tools/bench_corridor.py [batch] [n] [n_maps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import scenes
from path_optimizer_2_amd import capi

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
n_maps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
h = capi.Handle(capi.default_params(), device=0, max_batch=batch, max_n=n)
sc = scenes.build(h, range(n_maps), n)
geom = sc["geom"]
dev = torch.device("cuda", 0)
rep = lambda key: torch.from_numpy(np.stack([sc[key][b % n_maps] for b in range(batch)])).to(dev)
ref, tab, ext = rep("ref"), rep("tab"), rep("ext")
dist = torch.from_numpy(np.ascontiguousarray(np.transpose(sc["dist"], (0, 2, 1)))).to(dev)     # column major
map_of = torch.arange(batch, dtype=torch.int32, device=dev) % n_maps
bounds = torch.zeros((batch, n, 6), dtype=torch.float64, device=dev)
nv = torch.zeros(batch, dtype=torch.int32, device=dev)
prm = h.corridor_params()
p = lambda t: capi.C.c_void_p(t.data_ptr())
m = tab.shape[2]
ms = []
for _ in range(10):
    assert h.lib.pqp_corridor_bounds_device(h._h, batch, n, m, p(ref), None, p(tab), p(ext), p(dist), p(map_of), capi.C.byref(geom), capi.C.byref(prm), p(bounds), p(nv)) == 0
    ms.append(h.last_kernel_ms())
h.sync()
k_ms = float(np.median(ms[2:]))
samples = batch * n * 3 * 2 * 25     # upper bound: 3 circles x 2 sides x (20 coarse + 5 fine) bilinear samples
print(f"corridor bounds: batch {batch} n {n} maps {n_maps} ({geom.rows}x{geom.cols} cells): {k_ms * 1e3:.1f} us per launch = {batch / k_ms * 1e3:.0f} scenarios/s, "
      f"{batch * n / k_ms * 1e3 / 1e6:.1f} M waypoints/s; <= {samples * 16 / k_ms / 1e6:.1f} GB/s of 4-float gathers; blocked {int((nv < n).sum())}/{batch}")
