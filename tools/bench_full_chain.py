"""The whole of PathOptimizer::solve, input points -> optimised path, as ONE device-resident call (pqp_optimize_path_device): ragged
scenarios (polygons of 7..13 input points over 8 obstacle maps: every intermediate count differs per scenario), nothing copied to the
host between the twelve steps.  Prints scenarios/s (host clock around enqueue + sync over several repetitions) and the stage census.
Usage: python tools/bench_full_chain.py [batch=1024] [n_maps=8] [reps=10] [--exact-smoothers] [--tension] [--inflight-2] [--moving] [--carry] [--graph | --graph-unfenced]   (run on the GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_scene

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_maps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 10
cs = [make_scene(seed=s, n=40, n_obstacles=25, knots_every=3.05) for s in range(n_maps)]
rng = np.random.default_rng(1)
p_max = len(cs[0]["knots_x"])
pts = np.zeros((batch, p_max, 2)); n_pts = np.zeros(batch, dtype=np.int32); map_of = (np.arange(batch) % n_maps).astype(np.int32)
start = np.zeros((batch, 3)); target = np.zeros((batch, 3))
for b in range(batch):
    c = cs[b % n_maps]
    P = int(rng.integers(7, p_max + 1))
    n_pts[b] = P
    pts[b, :P, 0] = c["knots_x"][:P]; pts[b, :P, 1] = c["knots_y"][:P] + rng.normal(scale=0.15, size=P)
    h0 = np.arctan2(pts[b, 1, 1] - pts[b, 0, 1], pts[b, 1, 0] - pts[b, 0, 0])
    start[b] = (pts[b, 0, 0] + 0.1, pts[b, 0, 1] + 0.1, h0)
    h1 = np.arctan2(pts[b, P - 1, 1] - pts[b, P - 2, 1], pts[b, P - 1, 0] - pts[b, P - 2, 0])
    target[b] = (pts[b, P - 1, 0], pts[b, P - 1, 1], h1)
c0 = cs[0]
geom = capi.PqpGridGeometry(c0["rows"], c0["cols"], c0["resolution"], c0["length"][0], c0["length"][1], c0["pos"][0], c0["pos"][1])
dist = np.stack([c["dist"] for c in cs])

inflight = 2 if "--inflight-2" in sys.argv else 1      # independent batches in flight (each on its own pair of handles / streams)
dev = torch.device("cuda", 0)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
d_pts, d_np, d_st, d_tg, d_map = t(pts, np.float64), t(n_pts, np.int32), t(start, np.float64), t(target, np.float64), t(map_of, np.int32)
d_dist = t(np.transpose(dist, (0, 2, 1)), np.float32)
p = lambda x: capi.C.c_void_p(x.data_ptr())
lanes = []
for _ in range(inflight):
    h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=256)                    # path QP: production setting
    # smoother QPs: the reference's setting (OSQP defaults, eps 1e-3).  OSQP's default adaptive_rho_interval is time based - the iteration at which
    # 40 % of the setup time has passed, rounded to a multiple of check_termination = 25 and at least 25; here: 25 (--rho-interval-100: the
    # fixed 100 of pqp_default_params, with which the postSmooth QP needs 129 instead of 57 iterations)
    # --exact-smoothers (polish = 1, attempts every 25 iterations): TensionSmoother2's QP has no inequality rows - solved as one KKT system
    # instead of 25 ADMM iterations -, postSmooth's ends in a KKT-verified polish after 25 iterations: both return the exact optimum.
    # --exact-s1 (polish = 2): only the former; postSmooth runs the reference's plain ADMM
    pol = 1 if "--exact-smoothers" in sys.argv else (2 if "--exact-s1" in sys.argv else 0)
    hs = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=100 if "--rho-interval-100" in sys.argv else 25,
                                         polish=pol, polish_every=25 if pol == 1 else 0, polish_refine_iter=2 if pol else 4), device=0, max_batch=batch, max_n=128)
    h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_ORDER_BY_COST, 1)
    if "--carry" in sys.argv:            # PQP_OPT_CARRY_CYCLES on both handles: every QP of the chain starts from its slot's previous planning cycle
        h.set_option(capi.OPT_CARRY_CYCLES, 1); hs.set_option(capi.OPT_CARRY_CYCLES, 1)
    h.set_option(capi.OPT_CHAIN_GRAPH, 2 if "--graph-unfenced" in sys.argv else (1 if "--graph" in sys.argv else 0))           # PQP_OPT_CHAIN_GRAPH: the chain replayed as a captured hipGraph
    h.set_option(capi.OPT_RESERVE_CUS, int(os.environ.get("PQP_RESERVE_CUS", "0")))     # (experiments: CUs the path QP leaves to the other kernels in flight)
    # capacities sized to the workload (lines of 18..36 m): every smoother QP of the batch runs at the padded maximum size
    cfg = h.chain_config(raw_max=64, sample_max=48, layer_max=32, n_max=128) if "--default-capacities" not in sys.argv else h.chain_config()
    if "--tension" in sys.argv:            # FLAGS_smoothing_method = TENSION: clearance lookup + TensionSmoother's QP instead of TensionSmoother2's
        cfg.smoothing_method = capi.SMOOTHING_TENSION
    out = torch.zeros((batch, cfg.n_max, 7), dtype=torch.float64, device=dev)
    n_out, status, stage, iters = (torch.zeros(batch, dtype=torch.int32, device=dev) for _ in range(4))
    lanes.append((h, hs, cfg, out, n_out, status, stage, iters))
torch.cuda.synchronize()


# --moving (implied by --carry): the scenarios change from call to call as they do between planning cycles - input points shifted by ~2 cm, the start pose by
# ~5 cm / 0.01 rad - six variants in turn, so that a carried start is the PREVIOUS cycle's solution, not this one's
moving = "--moving" in sys.argv or "--carry" in sys.argv
variants = [(d_pts, d_st)]
if moving:
    rv = np.random.default_rng(9)
    for v in range(5):
        pv = pts.copy(); pv[:, :, 1] += rv.normal(scale=0.02, size=pts.shape[:2]) * (pts[:, :, 0] != 0)
        sv = start.copy(); sv[:, :2] += rv.normal(scale=0.05, size=(batch, 2)); sv[:, 2] += rv.normal(scale=0.01, size=batch)
        variants.append((t(pv, np.float64), t(sv, np.float64)))


def run(k):
    h, hs, cfg, out, n_out, status, stage, iters = lanes[k % inflight]
    d_pts, d_st = variants[(k // inflight) % len(variants)]
    h._check(h.lib.pqp_optimize_path_device(h._h, hs._h, capi.C.byref(cfg), batch, p_max, p(d_pts), p(d_np), p(d_st), p(d_tg), p(d_dist), p(d_map),
                                            capi.C.byref(geom), None, p(out), p(n_out), p(status), p(stage), p(iters)))


def sync():
    for h, hs, *_ in lanes:
        h.sync(); hs.sync()


for k in range((3 if not any(a.startswith("--graph") for a in sys.argv) else 2 * max(len(variants), 3)) * inflight):          # (--graph: plain call + capture per argument set, before the clock)
    run(k)
sync()
t0 = time.perf_counter()
for k in range(reps * inflight):
    run(k)
sync()
dt = (time.perf_counter() - t0) / (reps * inflight)
h, hs, cfg, out, n_out, status, stage, iters = lanes[0]
sg, no = stage.cpu().numpy(), n_out.cpu().numpy()
names = ["ok", "few points", "smoother failed", "search failed", "short reference", "post smooth failed", "heading", "blocked", "path QP failed", "capacity"]
print(f"pqp_optimize_path_device: {batch} ragged scenarios over {n_maps} maps ({int(n_pts.min())}..{int(n_pts.max())} input points): "
      f"{dt * 1e3:.3f} ms per batch = {batch / dt:.0f} scenarios/s, input points -> optimised path, device resident"
      + (f", {inflight} batches in flight" if inflight > 1 else "") + (", scenarios moving from call to call" if moving else "")
      + (", PQP_OPT_CARRY_CYCLES" if "--carry" in sys.argv else "") + (", PQP_OPT_CHAIN_GRAPH" if "--graph" in sys.argv else (", PQP_OPT_CHAIN_GRAPH = 2 (unfenced)" if "--graph-unfenced" in sys.argv else "")))
print("  stages: " + ", ".join(f"{names[k]} {int((sg == k).sum())}" for k in range(10) if (sg == k).any()))
ok = sg == 0
print(f"  paths: {int(ok.sum())} solved, waypoints {int(no[ok].min())}..{int(no[ok].max())} (mean {no[ok].mean():.0f}); "
      f"path-QP ADMM iterations median {np.median(iters.cpu().numpy()[ok]):.0f}")
