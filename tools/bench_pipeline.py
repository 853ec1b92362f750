"""Device pipeline spline points -> spline fit -> reference states -> corridor bounds -> path QP (waypoint count per scenario),
per-stage HIP-event kernel times.  Usage: python tools/bench_pipeline.py [batch] [n_maps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_scene

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_maps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_max, length = 128, 24.0            # 24 m of reference line: 80 - 128 states with dynamic segmentation
cs = [make_scene(seed=s, n=10, n_obstacles=25) for s in range(n_maps)]
dev = torch.device("cuda", 0)
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
rep = lambda f: np.stack([f(cs[b % n_maps]) for b in range(batch)])
ks = t(rep(lambda c: c["knots_s"])); kx = t(rep(lambda c: c["knots_x"])); ky = t(rep(lambda c: c["knots_y"]))
m = ks.shape[1]
dist = t(np.transpose(np.stack([c["dist"] for c in cs]), (0, 2, 1)), torch.float32)
map_of = torch.arange(batch, dtype=torch.int32, device=dev) % n_maps
max_s = torch.full((batch,), length, dtype=torch.float64, device=dev)
hd0 = lambda c: np.arctan2(c["knots_y"][1] - c["knots_y"][0], c["knots_x"][1] - c["knots_x"][0])
start = t(rep(lambda c: np.array([c["knots_x"][0] + 0.1, c["knots_y"][0] + 0.1, hd0(c)])))
z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
tab, ext, ref, err = z(batch, 9, m), z(batch, 4), z(batch, n_max, 5), z(batch, 2)
count, nv, status = z(batch, dt=torch.int32), z(batch, dt=torch.int32), z(batch, dt=torch.int32)
bounds, out, scal = z(batch, n_max, 6), z(batch, n_max, 7), z(batch, 6)
h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n_max)
lib, hh = h.lib, h._h
p = lambda x: capi.C.c_void_p(x.data_ptr())
c0 = cs[0]
geom = capi.PqpGridGeometry(c0["rows"], c0["cols"], c0["resolution"], c0["length"][0], c0["length"][1], c0["pos"][0], c0["pos"][1])
prm = h.corridor_params()
times = {k: [] for k in ("spline_fit", "reference_states", "corridor_bounds", "path_solve")}
for rep_i in range(6):
    assert lib.pqp_spline_fit_device(hh, batch, m, p(ks), p(kx), p(ky), p(tab), p(ext)) == 0
    times["spline_fit"].append(h.last_kernel_ms())
    assert lib.pqp_reference_states_device(hh, batch, n_max, m, p(tab), p(ext), p(max_s), p(start), 0.15, 0.3, 1, p(ref), p(count), p(err)) == 0
    times["reference_states"].append(h.last_kernel_ms())
    assert lib.pqp_corridor_bounds_device(hh, batch, n_max, m, p(ref), p(count), p(tab), p(ext), p(dist), p(map_of), capi.C.byref(geom), capi.C.byref(prm), p(bounds), p(nv)) == 0
    times["corridor_bounds"].append(h.last_kernel_ms())
    h.sync()
    # scal = (init offset, init heading error, start k, target heading, blocked, max steering): a few device-side torch ops
    idx = (nv.long() - 1).clamp(min=0)
    scal[:, 0:2] = err; scal[:, 2] = ref[:, 0, 1]; scal[:, 3] = ref[torch.arange(batch, device=dev), idx, 2]
    scal[:, 4] = (nv < count).double(); scal[:, 5] = 35.0 * np.pi / 180.0
    torch.cuda.synchronize()
    h.solve_var_device(batch, n_max, nv, ref, bounds, scal, out, passes=1, status=status)
    times["path_solve"].append(h.last_kernel_ms())
h.sync()
cnt, nvh, st = count.cpu().numpy(), nv.cpu().numpy(), status.cpu().numpy()
print(f"pipeline batch {batch}, {n_maps} maps ({geom.rows}x{geom.cols}), {length} m of reference line: states {cnt.min()}..{cnt.max()}, "
      f"usable {nvh.min()}..{nvh.max()} (blocked {int((nvh < cnt).sum())}), solved {int((st == 1).sum())}/{int((nvh >= 2).sum())}")
tot = 0.0
for k, v in times.items():
    ms = float(np.median(v[1:])); tot += ms
    print(f"  {k:18s} {ms * 1e3:9.1f} us")
print(f"  {'sum of kernels':18s} {tot * 1e3:9.1f} us  = {batch / tot * 1e3:.0f} scenarios/s end to end")
