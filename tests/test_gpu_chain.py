"""pqp_optimize_path_device: the whole of PathOptimizer::solve (reference src/path_optimizer.cpp:34-71) for a ragged batch as one
device-resident call, against the same twelve steps run one scenario at a time through the per-step entry points (each of which is
checked against its oracle in test_gpu_corridor.py / test_gpu_smoothers.py / test_gpu_parity.py), and the reference's `return false`
sites as stages.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

import corridor_oracle as K
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _smoother_params():
    # the reference's smoother setting (OSQP default eps 1e-3) + the KKT-verified polish, so that both sides return the QP's optimum
    return capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25)


def _scenarios(B, n_maps=4, seed=5):
    cs = [make_scene(seed=s, n=40, n_obstacles=25, knots_every=3.05) for s in range(n_maps)]
    rng = np.random.default_rng(seed)
    p_max = len(cs[0]["knots_x"])
    pts = np.zeros((B, p_max, 2)); n_pts = np.zeros(B, dtype=np.int32); map_of = (np.arange(B) % n_maps).astype(np.int32)
    start = np.zeros((B, 3)); target = np.zeros((B, 3))
    for b in range(B):
        c = cs[b % n_maps]
        P = int(rng.integers(7, p_max + 1))                       # polygons of different length: every count downstream differs
        n_pts[b] = P
        pts[b, :P, 0] = c["knots_x"][:P]; pts[b, :P, 1] = c["knots_y"][:P] + rng.normal(scale=0.15, size=P)
        h0 = np.arctan2(pts[b, 1, 1] - pts[b, 0, 1], pts[b, 1, 0] - pts[b, 0, 0])
        start[b] = (pts[b, 0, 0] + 0.1, pts[b, 0, 1] + 0.1, h0 + 0.03)
        h1 = np.arctan2(pts[b, P - 1, 1] - pts[b, P - 2, 1], pts[b, P - 1, 0] - pts[b, P - 2, 0])
        target[b] = (pts[b, P - 1, 0], pts[b, P - 1, 1], h1)
    c0 = cs[0]
    geom = capi.PqpGridGeometry(c0["rows"], c0["cols"], c0["resolution"], c0["length"][0], c0["length"][1], c0["pos"][0], c0["pos"][1])
    return dict(pts=pts, n_pts=n_pts, map_of=map_of, start=start, target=target, dist=np.stack([c["dist"] for c in cs]), geom=geom)


def _one_by_one(h, hs, sc, b, cfg):
    """The chain for scenario b alone, exact sizes, through the per-step host-pointer entry points."""
    P = int(sc["n_pts"][b])
    mo = sc["map_of"][b:b + 1]
    st, tg = sc["start"][b:b + 1], sc["target"][b:b + 1]
    r = h.bspline_resample(sc["pts"][b:b + 1, :P], np.array([P], dtype=np.int32), cfg.raw_max)
    n0 = int(r["count"][0])
    x0, y0, s0 = (r[k][:, :n0] for k in ("x", "y", "s"))
    tab, ext = h.spline_fit(s0, x0, y0)
    seg = h.segment_raw_reference(tab, ext, s0[:, -1].copy(), cfg.sample_max)
    n1 = int(seg["count"][0])
    if cfg.smoothing_method == capi.SMOOTHING_TENSION:
        gx, gy = seg["x"][:, :n1], seg["y"][:, :n1]
        clr = np.array([[K.obstacle_distance(sc["dist"][mo[0]], sc["geom"], gx[0, i], gy[0, i]) for i in range(n1)]])      # Map::getObstacleDistance, the oracle's
        sm = hs.smooth_tension(gx, gy, seg["angle"][:, :n1], clr)
    else:
        sm = hs.smooth_tension2(*(seg[k][:, :n1] for k in ("x", "y", "angle", "k", "s")))
    assert sm["status"][0] == 1
    tab, ext = h.spline_fit(sm["s"], sm["x"], sm["y"])
    ls, lb, ub, cnt, vl = h.dp_corridor(tab, ext, sm["s"][:, -1] + cfg.smoothed_length_margin, st, sc["dist"], sc["geom"], max_layers=cfg.layer_max, map_of=mo)
    k = int(cnt[0])
    assert k >= 4
    ps = hs.post_smooth(ls[:, :k].copy(), lb[:, :k].copy(), ub[:, :k].copy(), vl)
    assert ps["status"][0] == 1
    x2, y2, s2 = h.offsets_to_points(tab, ext, ls[:, :k].copy(), ps["l"])
    tab, ext = h.spline_fit(s2, x2, y2)
    max_s = h.reference_length(tab, ext, s2[:, -1].copy(), tg)
    ref, count, err = h.reference_states(tab, ext, max_s, cfg.n_max, start=st, ds_small=cfg.output_spacing / 2, ds_large=cfg.output_spacing, dynamic=True)
    bounds, nv = h.corridor_bounds(ref, tab, ext, sc["dist"], sc["geom"], map_of=mo, n_of=count)
    scal = np.array([[err[0, 0], err[0, 1], 0.0, tg[0, 2], 1.0 if nv[0] < count[0] else 0.0, cfg.max_steering_angle]])
    res = h.solve_var(nv, ref, bounds, scal, passes=1)
    return dict(n0=n0, n1=n1, layers=k, count=int(count[0]), nv=int(nv[0]), out=res["out"][0], status=int(res["status"][0]), ref=ref, bounds=bounds, scal=scal)


def test_ragged_batch_equals_the_steps_run_one_scenario_at_a_time(hip_lib):
    B = 24
    sc = _scenarios(B)
    h = capi.Handle(capi.production_params(), max_batch=B, max_n=256)
    hs = capi.Handle(_smoother_params(), max_batch=B, max_n=128)
    cfg = h.chain_config()
    got = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=cfg)
    assert (got["stage"] == 0).sum() >= B - 2, got["stage"]
    seen = set()
    for b in range(B):
        if got["stage"][b] != 0:
            continue
        want = _one_by_one(h, hs, sc, b, cfg)
        seen.add((want["n0"], want["n1"], want["layers"], want["nv"]))
        assert got["n_out"][b] == want["nv"] and got["status"][b] == want["status"] == 1
        nv = want["nv"]
        # both sides solve their QPs to the optimum (polish); the padded smoother QPs and the exact-size ones agree to the polish
        # tolerance, and so does everything downstream
        assert np.abs(got["out"][b, :nv] - want["out"][:nv]).max() < 2e-5, (b, np.abs(got["out"][b, :nv] - want["out"][:nv]).max())
        assert np.all(got["out"][b, nv:] == 0.0)
    assert len(seen) >= 6                          # the batch really was ragged
    h.close(); hs.close()


def test_configs0_the_demo_s_single_path(hip_lib):
    """BASELINE configs[0]: one scenario, a line of about 60 waypoints, fixed start and goal on an obstacle map - the reference's demo
    (CPU OSQP).  Here: batch 1 through the device-resident chain, the reference's smoother setting (OSQP defaults) and the production path
    QP; the path against the C restatement of OSQP run to 1e-9 on the reference states and corridor the chain produced - the north_star
    bar: 1e-4 in lateral offset and heading."""
    import pqp_oracle_c as OC
    sc = _scenarios(1, n_maps=1, seed=21)
    sc["n_pts"][0] = 9                                             # ~24 m of input polygon -> ~18 m of line after the search: ~60 waypoints
    h = capi.Handle(capi.production_params(), max_batch=1, max_n=256)
    hs = capi.Handle(_smoother_params(), max_batch=1, max_n=128)
    cfg = h.chain_config()
    got = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=cfg)
    assert got["stage"][0] == 0 and got["status"][0] == 1
    nv = int(got["n_out"][0])
    assert 40 <= nv <= 90, nv
    want = _one_by_one(h, hs, sc, 0, cfg)
    assert want["nv"] == nv
    ref = OC.solve_batch(OC.params(eps_abs=1e-9, eps_rel=1e-9, max_iter=200000), want["ref"][:, :nv].copy(), want["bounds"][:, :nv].copy(), want["scal"], passes=1)
    assert ref["solved"] == 1
    assert np.abs(got["out"][0, :nv, 3:5] - ref["out"][0, :, 3:5]).max() < 1e-4
    assert np.abs(got["out"][0, :nv, 0:2] - ref["out"][0, :, 0:2]).max() < 1e-4
    h.close(); hs.close()


def test_capacities_beyond_the_register_kernels(hip_lib):
    """pqp_chain_config has no upper bounds (the reference has none): 300 samples are more than TensionSmoother2's generic core holds (256),
    400 layers more than postSmooth's (341), 600 waypoints more than the lane-per-waypoint kernel's (512 - the handle is created for them;
    a line this short still runs there).  Same paths as with the default capacities; a smoother handle in the reference's ADMM setting gets
    the exact kernels where the generic core does not fit instead of PQP_ERR_CAPACITY."""
    sc = _scenarios(3, n_maps=2, seed=33)
    h = capi.Handle(capi.production_params(), max_batch=3, max_n=600)
    hs = capi.Handle(_smoother_params(), max_batch=3, max_n=400)
    big_cfg = h.chain_config(raw_max=400, sample_max=300, layer_max=400, n_max=600)
    for method in (capi.SMOOTHING_TENSION2, capi.SMOOTHING_TENSION):
        cfg0 = h.chain_config(smoothing_method=method)
        cfg1 = h.chain_config(raw_max=400, sample_max=300, layer_max=400, n_max=600, smoothing_method=method)
        a = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=cfg0)
        b = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=cfg1)
        assert (a["stage"] == 0).all() and (b["stage"] == 0).all() and (b["status"] == 1).all()
        assert np.array_equal(a["n_out"], b["n_out"])
        for q in range(3):
            nv = int(a["n_out"][q])
            assert np.abs(a["out"][q, :nv] - b["out"][q, :nv]).max() < 1e-7, (method, q)
    plain = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=3, max_n=400)       # the reference's smoother setting
    c = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=plain, cfg=big_cfg)
    assert (c["stage"] == 0).all() and (c["status"] == 1).all()
    h.close(); hs.close(); plain.close()


def test_the_tension_smoothing_method(hip_lib):
    """FLAGS_smoothing_method = TENSION (planning_flags.cpp:27, ReferencePathSmoother::create reference_path_smoother.cpp:18-29): the chain
    looks the clearance of the raw line's samples up on the device and runs TensionSmoother's QP (tension_smoother.cpp:49-177) in place of
    TensionSmoother2's; against the steps run one scenario at a time with the clearance from the oracle's grid_map restatement."""
    B = 10
    sc = _scenarios(B, seed=11)
    h = capi.Handle(capi.production_params(), max_batch=B, max_n=256)
    hs = capi.Handle(_smoother_params(), max_batch=B, max_n=128)
    cfg = h.chain_config(smoothing_method=capi.SMOOTHING_TENSION)
    got = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=cfg)
    assert (got["stage"] == 0).sum() >= B - 2, got["stage"]
    base = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs)
    differs = 0
    for b in range(B):
        if got["stage"][b] != 0:
            continue
        want = _one_by_one(h, hs, sc, b, cfg)
        nv = want["nv"]
        assert got["n_out"][b] == nv and got["status"][b] == want["status"] == 1
        # (TensionSmoother's QP is ill-conditioned - test_tension: both sides sit on a ~1e-5 round-off floor of the smoothed line)
        assert np.abs(got["out"][b, :nv] - want["out"][:nv]).max() < 2e-4, (b, np.abs(got["out"][b, :nv] - want["out"][:nv]).max())
        if base["stage"][b] == 0 and (base["n_out"][b] != nv or np.abs(base["out"][b, :nv] - got["out"][b, :nv]).max() > 1e-3):
            differs += 1
    assert differs >= 1                           # it is another smoother: the paths are not the TENSION2 ones
    with pytest.raises(capi.PqpError):             # "No such smoother!" (reference_path_smoother.cpp:25-28)
        h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs, cfg=h.chain_config(smoothing_method=7))
    h.close(); hs.close()


def test_chain_carries_the_previous_planning_cycle(hip_lib):
    """PQP_OPT_CARRY_CYCLES through the whole device-resident chain: the vehicle advances a little every cycle (start pose and input points move), so
    the raw line, the layer and waypoint counts of a scenario change from call to call - the path QP's warm state is kept per waypoint, the exact
    smoother kernels start from their slot's previous active set.  Every cycle's paths equal those of handles that start cold (what the carried
    start saves is time: tools/bench_full_chain.py --carry, 892 k -> 989 k scenarios/s)."""
    B = 24
    sc = _scenarios(B, seed=21)
    pair = lambda: (capi.Handle(capi.production_params(), max_batch=B, max_n=256), capi.Handle(_smoother_params(), max_batch=B, max_n=128))
    h, hs = pair(); hc, hsc = pair()
    for x in (hc, hsc):
        x.set_option(capi.OPT_CARRY_CYCLES, 1)
    rng = np.random.default_rng(3)
    counts = []
    for cycle in range(4):
        pts = sc["pts"].copy(); start = sc["start"].copy()
        if cycle:
            pts[:, :, 1] += rng.normal(scale=0.02, size=pts.shape[:2]) * (pts[:, :, 0] != 0)
            start[:, :2] += rng.normal(scale=0.05, size=(B, 2)); start[:, 2] += rng.normal(scale=0.01, size=B)
        a = (pts, sc["n_pts"], start, sc["target"], sc["dist"], sc["geom"])
        cold = h.optimize_path(*a, map_of=sc["map_of"], smoother=hs)
        got = hc.optimize_path(*a, map_of=sc["map_of"], smoother=hsc)
        np.testing.assert_array_equal(got["stage"], cold["stage"]); np.testing.assert_array_equal(got["n_out"], cold["n_out"])
        np.testing.assert_array_equal(got["status"], cold["status"])
        ok = np.flatnonzero(cold["stage"] == 0)
        assert len(ok) >= B - 4
        for b in ok:
            nv = cold["n_out"][b]
            assert np.abs(got["out"][b, :nv] - cold["out"][b, :nv]).max() < 2e-5, (cycle, b, np.abs(got["out"][b, :nv] - cold["out"][b, :nv]).max())
        counts.append(cold["n_out"].copy())
    assert any((counts[k] != counts[0]).any() for k in range(1, 4))          # the waypoint counts did move between the cycles
    for x in (h, hs, hc, hsc):
        x.close()


def test_stages_are_the_reference_s_return_false_sites(hip_lib):
    sc = _scenarios(6, seed=9)
    sc["n_pts"][1] = 3                                               # "Few reference points" (reference_path_smoother.cpp:33-36)
    sc["start"][2, :2] += np.array([-np.sin(sc["start"][2, 2]), np.cos(sc["start"][2, 2])]) * 14.0      # 14 m beside the line: graphSearchDp quits
    sc["start"][3, 2] += 1.6                                         # 92 degrees off the line's heading: path_optimizer.cpp:113-116
    h = capi.Handle(capi.production_params(), max_batch=6, max_n=256)
    hs = capi.Handle(_smoother_params(), max_batch=6, max_n=128)
    got = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"], smoother=hs)
    assert got["stage"][1] == 1 and got["stage"][2] == 3 and got["stage"][3] == 6, got["stage"]
    for b in (1, 2, 3):
        assert got["status"][b] == 0 and got["n_out"][b] == 0
    for b in (0, 4, 5):
        assert got["stage"][b] == 0 and got["status"][b] == 1 and got["n_out"][b] >= 2 and np.isfinite(got["out"][b]).all()
    # one handle for everything (smoother QPs at the path QP's setting) works too
    alone = h.optimize_path(sc["pts"], sc["n_pts"], sc["start"], sc["target"], sc["dist"], sc["geom"], map_of=sc["map_of"])
    assert list(alone["stage"]) == list(got["stage"])
    h.close(); hs.close()


def test_clearance_lookup_on_the_device(hip_lib):
    """pqp_clearance_device = Map::getObstacleDistance at the raw line's points (tension_smoother.cpp:168), against the restatement of
    grid_map's bilinear lookup in the corridor oracle; its output feeds pqp_smooth_tension."""
    import torch
    sc = _scenarios(3, n_maps=3, seed=2)
    h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=3, max_n=64)
    rng = np.random.default_rng(0)
    n = 40
    x = rng.uniform(-30, 30, size=(3, n)); y = rng.uniform(-18, 18, size=(3, n))
    x[0, 0], y[0, 0] = 1e3, 0.0                                   # outside the map: 0 (Map.cpp:17-21)
    dev = torch.device("cuda", 0)
    t = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d_dist = t(np.transpose(sc["dist"], (0, 2, 1)), np.float32)
    d_x, d_y, d_map = t(x), t(y), t(np.arange(3), np.int32)
    out = torch.zeros((3, n), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    p = lambda a: capi.C.c_void_p(a.data_ptr())
    assert h.lib.pqp_clearance_device(h._h, 3, n, p(d_x), p(d_y), p(d_dist), p(d_map), capi.C.byref(sc["geom"]), p(out)) == 0
    h.sync()
    got = out.cpu().numpy()
    for b in range(3):
        for i in range(n):
            assert got[b, i] == K.obstacle_distance(sc["dist"][b], sc["geom"], x[b, i], y[b, i])
    assert got[0, 0] == 0.0
    r = h.smooth_tension(np.cumsum(np.ones((3, n)), axis=1), np.zeros((3, n)), np.zeros((3, n)), got)
    assert (r["status"] == 1).all()
    h.close()


def test_chain_replayed_as_a_hip_graph_is_bit_identical(hip_lib):
    """PQP_OPT_CHAIN_GRAPH: the chain's ~25 launches on two streams captured once per argument set and replayed (csrc/pqp_chain.inc).  Same
    device buffers call after call, their CONTENTS moving like planning cycles: every call's result equals the plainly launched chain's bit for bit -
    through the plain first calls, the two captures (one per parity of the cost-order double buffer) and the replays - and other arguments fall
    back to plain launches."""
    import torch
    B = 40
    sc = _scenarios(B, seed=11)
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    p = lambda x: capi.C.c_void_p(x.data_ptr())
    d_np, d_tg, d_map = t(sc["n_pts"], np.int32), t(sc["target"], np.float64), t(sc["map_of"], np.int32)
    d_dist = t(np.transpose(sc["dist"], (0, 2, 1)), np.float32)
    rng = np.random.default_rng(3)
    cycles = []
    for _ in range(9):
        pv = sc["pts"].copy(); pv[:, :, 1] += rng.normal(scale=0.02, size=pv.shape[:2]) * (pv[:, :, 0] != 0)
        sv = sc["start"].copy(); sv[:, :2] += rng.normal(scale=0.05, size=(B, 2))
        cycles.append((pv, sv))
    results = {}
    for graph in (0, 1, 2):          # 2: replays without the fences against the smoother handle's stream (this test syncs both streams after every call)
        h = capi.Handle(capi.production_params(), device=0, max_batch=B, max_n=256)
        hs = capi.Handle(_smoother_params(), device=0, max_batch=B, max_n=128)
        h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_ORDER_BY_COST, 1); h.set_option(capi.OPT_CHAIN_GRAPH, graph)
        cfg = h.chain_config(raw_max=64, sample_max=48, layer_max=32, n_max=128)
        d_pts, d_st = t(cycles[0][0], np.float64), t(cycles[0][1], np.float64)
        out = torch.zeros((B, cfg.n_max, 7), dtype=torch.float64, device=dev)
        n_out, status, stage, iters = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(4))
        got = []
        for pv, sv in cycles:
            d_pts.copy_(torch.from_numpy(pv)); d_st.copy_(torch.from_numpy(sv))
            torch.cuda.synchronize()
            h._check(h.lib.pqp_optimize_path_device(h._h, hs._h, capi.C.byref(cfg), B, sc["pts"].shape[1], p(d_pts), p(d_np), p(d_st), p(d_tg), p(d_dist), p(d_map),
                                                    capi.C.byref(sc["geom"]), None, p(out), p(n_out), p(status), p(stage), p(iters)))
            h.sync(); hs.sync()
            got.append((out.cpu().numpy().copy(), n_out.cpu().numpy().copy(), status.cpu().numpy().copy(), stage.cpu().numpy().copy()))
        if graph:
            # other arguments (another output buffer): not the captured graph's - plain launches, same result as the last cycle's
            out2 = torch.zeros_like(out)
            h._check(h.lib.pqp_optimize_path_device(h._h, hs._h, capi.C.byref(cfg), B, sc["pts"].shape[1], p(d_pts), p(d_np), p(d_st), p(d_tg), p(d_dist), p(d_map),
                                                    capi.C.byref(sc["geom"]), None, p(out2), p(n_out), p(status), p(stage), p(iters)))
            h.sync(); hs.sync()
            o2, no = out2.cpu().numpy(), got[-1][1]
            for b in range(B):          # (rows beyond a scenario's waypoint count are not written: the long-lived buffer keeps older cycles' rows there)
                np.testing.assert_array_equal(o2[b, :no[b]], got[-1][0][b, :no[b]])
            # and the handle still solves plain batches afterwards (the ticket counter of the path kernel was reset inside the graph)
            from path_optimizer_2_amd.synth import make_batch
            b = make_batch(16, 80)
            r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
            assert (r["status"] == 1).all()
        results[graph] = got
        h.close(); hs.close()
    assert (results[0][0][3] == 0).sum() >= B // 2          # most scenarios give a path
    for k in range(len(cycles)):
        for g in (1, 2):
            for a, b_ in zip(results[0][k], results[g][k]):
                np.testing.assert_array_equal(a, b_)


@pytest.mark.gpu
def test_chain_graph_on_the_lane_per_qp_kernel_leaves_the_ticket_counter_alone(hip_lib):
    """A captured chain whose path solve runs on path_stream_kernel (PQP_OPT_STREAM_BATCH reached) never touches the lane-per-waypoint kernel's
    ticket counter: plain pqp_path_solve calls on the same handle between two replays keep working (round 4's replay rewound the host's
    ticket base to the value at capture time - the next plain launch then drew only tickets beyond its batch and solved nothing), and the
    replays stay bit-identical to plain launches."""
    import torch
    from path_optimizer_2_amd.synth import make_batch
    B = 24
    sc = _scenarios(B, seed=5)
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    p = lambda x: capi.C.c_void_p(x.data_ptr())
    d_np, d_tg, d_map = t(sc["n_pts"], np.int32), t(sc["target"], np.float64), t(sc["map_of"], np.int32)
    d_dist = t(np.transpose(sc["dist"], (0, 2, 1)), np.float32)
    plain = make_batch(16, 80)
    ref_h = capi.Handle(capi.production_params(), device=0, max_batch=16, max_n=80)
    want = ref_h.solve(plain["ref"], plain["bounds"], plain["scal"], passes=1)
    ref_h.close()
    results = {}
    for graph in (0, 1):
        h = capi.Handle(capi.production_params(), device=0, max_batch=B, max_n=256)
        hs = capi.Handle(_smoother_params(), device=0, max_batch=B, max_n=128)
        h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1); h.set_option(capi.OPT_CHAIN_GRAPH, graph)
        cfg = h.chain_config(raw_max=64, sample_max=48, layer_max=32, n_max=128)
        d_pts, d_st = t(sc["pts"], np.float64), t(sc["start"], np.float64)
        out = torch.zeros((B, cfg.n_max, 7), dtype=torch.float64, device=dev)
        n_out, status, stage, iters = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(4))
        got = []
        for cycle in range(6):
            h._check(h.lib.pqp_optimize_path_device(h._h, hs._h, capi.C.byref(cfg), B, sc["pts"].shape[1], p(d_pts), p(d_np), p(d_st), p(d_tg), p(d_dist), p(d_map),
                                                    capi.C.byref(sc["geom"]), None, p(out), p(n_out), p(status), p(stage), p(iters)))
            h.sync(); hs.sync()
            assert h.last_path_kernel() == capi.KERNEL_LANE_PER_QP
            got.append((out.cpu().numpy().copy(), n_out.cpu().numpy().copy(), status.cpu().numpy().copy()))
            # plain solves on the lane-per-waypoint kernel in between (the option is put back before the next chain call: same key)
            h.set_option(capi.OPT_STREAM_BATCH, 0)
            for _ in range(1 + cycle % 2):
                r = h.solve(plain["ref"], plain["bounds"], plain["scal"], passes=1)
                assert h.last_path_kernel() == capi.KERNEL_LANE_PER_WAYPOINT
                assert (r["status"] == 1).all(), (graph, cycle, r["status"])
                np.testing.assert_array_equal(r["out"], want["out"])
            h.set_option(capi.OPT_STREAM_BATCH, 1)
        results[graph] = got
        h.close(); hs.close()
    assert (results[0][0][2] == 1).sum() >= B // 2          # most scenarios give a path
    for k in range(6):
        for a, b_ in zip(results[0][k], results[1][k]):
            np.testing.assert_array_equal(a, b_)
