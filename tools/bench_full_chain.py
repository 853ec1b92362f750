"""Every device step of PathOptimizer::solve in the reference's order, input points -> optimised path, with the kernel time of each
(HIP events around the kernels; the host-pointer entry points are used, so copies happen between the steps but are not timed):

  bSpline -> spline fit -> segmentRawReference -> TensionSmoother2 QP -> spline fit -> graphSearchDp -> postSmooth QP -> offsets to
  points -> spline fit -> reference states + initial error -> corridor bounds -> path QP (cold + re-linearised warm solve)

The smoother QPs share one sparsity pattern per launch, so a batch must agree on the point counts of the intermediate lines; the
scenarios are drawn with equal polygon lengths and the few whose counts differ from the majority are dropped (a caller groups them).
Usage: python tools/bench_full_chain.py [batch=1024] [n_maps=8]   (run on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_scene

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_maps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cs = [make_scene(seed=s, n=40, n_obstacles=25, knots_every=3.05) for s in range(n_maps)]
rng = np.random.default_rng(1)
P = len(cs[0]["knots_x"])
pts = np.zeros((batch, P, 2)); map_of = np.arange(batch, dtype=np.int32) % n_maps
for b in range(batch):
    c = cs[b % n_maps]
    pts[b, :, 0] = c["knots_x"]; pts[b, :, 1] = c["knots_y"] + rng.normal(scale=0.15, size=P)      # a noisy polygon per scenario
n_pts = np.full(batch, P, dtype=np.int32)
dist = np.stack([c["dist"] for c in cs])
c0 = cs[0]
geom = capi.PqpGridGeometry(c0["rows"], c0["cols"], c0["resolution"], c0["length"][0], c0["length"][1], c0["pos"][0], c0["pos"][1])

h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=128)                    # path QP: production setting
hs = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), device=0, max_batch=batch, max_n=64)   # smoother QPs: the reference's
times = []


def stage(name, handle=h):
    times.append((name, handle.last_kernel_ms()))


def keep_majority(count, *arrays):
    """the scenarios whose count is the batch's most common one"""
    vals, freq = np.unique(count, return_counts=True)
    k = int(vals[np.argmax(freq)])
    sel = count == k
    return k, sel


def chain():
    del times[:]
    r = h.bspline_resample(pts, n_pts, 96); stage("bSpline (B-spline resampling)")
    n0, sel = keep_majority(r["count"])
    idx = np.nonzero(sel)[0]
    x0, y0, s0 = (r[k][idx, :n0] for k in ("x", "y", "s"))
    tab, ext = h.spline_fit(s0, x0, y0); stage("spline fit (raw line)")
    seg = h.segment_raw_reference(tab, ext, s0[:, -1].copy(), 96); stage("segmentRawReference")
    n1, sel = keep_majority(seg["count"])
    idx = idx[sel]
    seg = {k: seg[k][sel][:, :n1] for k in ("x", "y", "angle", "k", "s")}
    sm = hs.smooth_tension2(seg["x"], seg["y"], seg["angle"], seg["k"], seg["s"]); stage("TensionSmoother2 QP", hs)
    ok = sm["status"] == 1
    tab, ext = h.spline_fit(sm["s"], sm["x"], sm["y"]); stage("spline fit (smoothed line)")
    length = sm["s"][:, -1].copy()
    hd = np.arctan2(sm["y"][:, 1] - sm["y"][:, 0], sm["x"][:, 1] - sm["x"][:, 0])
    start = np.column_stack([sm["x"][:, 0] + 0.1, sm["y"][:, 0] + 0.1, hd])
    ls, lb, ub, cnt, vl = h.dp_corridor(tab, ext, length, start, dist, geom, max_layers=64, map_of=map_of[idx]); stage("graphSearchDp")
    k, sel = keep_majority(cnt)
    sel &= ok
    idx = idx[sel]
    tab, ext, start = tab[sel], ext[sel], start[sel]
    ls, lb, ub, vl = ls[sel][:, :k].copy(), lb[sel][:, :k].copy(), ub[sel][:, :k].copy(), vl[sel].copy()
    ps = hs.post_smooth(ls, lb, ub, vl); stage("postSmooth QP", hs)
    ok = ps["status"] == 1
    x2, y2, s2 = h.offsets_to_points(tab, ext, ls, ps["l"]); stage("offsets -> points")
    tab, ext = h.spline_fit(s2, x2, y2); stage("spline fit (final line)")
    n_max = 128
    ref, count, err = h.reference_states(tab, ext, s2[:, -1].copy(), n_max, start=start); stage("reference states + initial error")
    bounds, nv = h.corridor_bounds(ref, tab, ext, dist, geom, map_of=map_of[idx], n_of=count); stage("corridor bounds")
    B = len(idx)
    scal = np.zeros((B, 6))
    last = np.clip(nv - 1, 0, None)
    scal[:, 0:2] = err; scal[:, 2] = ref[:, 0, 1]; scal[:, 3] = ref[np.arange(B), last, 2]
    scal[:, 4] = (nv < count).astype(float); scal[:, 5] = 35.0 * np.pi / 180.0
    res = h.solve_var(nv, ref, bounds, scal, passes=1); stage("path QP (2 passes)")
    solved = (res["status"] == 1) & ok & (nv >= 2)

    return B, n0, n1, k, count, nv, solved, res


chain()                                  # first pass: module load, buffer growth
B, n0, n1, k, count, nv, solved, res = chain()
print(f"full chain: {batch} scenarios over {n_maps} maps, {P} input points each; {B} with the majority shapes (raw line {n0} points, "
      f"{n1} samples, {k} layers, {int(count.min())}..{int(count.max())} states); paths solved {int(solved.sum())}/{B}, blocked {int((nv < count).sum())}")
kk, it = res["info"][:, 5], res["iters"]
print(f"  path QP: ADMM iterations median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}; reduced solves mean {kk.mean():.1f} "
      f"p99 {np.percentile(kk, 99):.0f} max {kk.max():.0f}; polished passes {np.bincount(res['info'][:, 4].astype(int)).tolist()}")
tot = 0.0
for name, ms in times:
    tot += ms
    print(f"  {name:34s} {ms * 1e3:9.1f} us")
print(f"  {'sum of kernels':34s} {tot * 1e3:9.1f} us  (the launches before the shape filters ran on all {batch} scenarios)")
print(f"  = {B / tot * 1e3:.0f} scenarios/s through the whole of PathOptimizer::solve, kernels only")
