"""Corridor bounds from the distance map on the GPU (pqp_corridor_bounds) against the CPU restatement
(oracle/corridor_oracle.py).  Sample positions are computed in the reference's operation order with FMA contraction off, so
the step counts of the ray search - and with them the bounds - are the oracle's; only sin/cos differ (ocml vs libm, <= 1 ulp),
which can move a bound by one search step when a sample sits within round-off of the 0.5 m threshold."""
import os

import numpy as np
import pytest

import corridor_oracle as K
import corridor_util as U
from path_optimizer_2_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle(hip_lib):
    h = capi.Handle(capi.default_params(), device=0, max_batch=64, max_n=128)
    yield h
    h.close()


def _geom(g):
    return capi.PqpGridGeometry(g.rows, g.cols, g.resolution, g.length_x, g.length_y, g.pos_x, g.pos_y)


def _compare(got, n_valid_got, c, prm=K.CorridorParams()):
    want, n_valid, blocked = K.update_bounds_improved(c["ref"], c["sx"], c["sy"], c["dist"], c["geom"], prm)
    assert n_valid_got == n_valid
    rows = np.vstack([want, np.array(blocked)[None]]) if blocked is not None else want
    diff = np.abs(got[:len(rows)] - rows)
    exact = diff < 1e-9
    # a disagreement must be a whole search step (0.05 or 0.3 m), and rare
    off = diff[~exact]
    assert exact.mean() > 0.99, (exact.mean(), off)
    for d in off:
        assert min(abs(d - 0.05 * k) for k in range(1, 8)) < 1e-9 or min(abs(d - 0.3 * k) for k in range(1, 4)) < 1e-9, d
    return exact.mean()


@pytest.mark.parametrize("seed,n", [(0, 40), (1, 80), (2, 80), (5, 120)])
def test_bounds_match_the_oracle(handle, seed, n):
    c = U.build(seed=seed, n=n)
    got, nv = handle.corridor_bounds(c["ref"][None], c["tab"][None], c["ext"][None], c["dist"], _geom(c["geom"]))
    _compare(got[0], int(nv[0]), c)


def test_batch_with_one_map_per_scenario_and_a_blocked_road(handle):
    cs = [U.build(seed=s, n=60) for s in (7, 8, 9)]
    g = cs[0]["geom"]
    # scenario 2: a wall across the road at waypoint 35
    d2 = cs[2]["dist"].copy()
    x_wall = cs[2]["ref"][35, 3]
    for i in range(g.rows):
        x, _ = K.grid_cell_position(g, i, 0)
        d2[i, :] = np.minimum(d2[i, :], np.float32(abs(x - x_wall)))
    cs[2]["dist"] = d2
    got, nv = handle.corridor_bounds(np.stack([c["ref"] for c in cs]), np.stack([c["tab"] for c in cs]), np.stack([c["ext"] for c in cs]),
                                     np.stack([c["dist"] for c in cs]), _geom(g), map_of=[0, 1, 2])
    for q, c in enumerate(cs):
        _compare(got[q], int(nv[q]), c)
    assert nv[2] < 35 and nv[0] == 60


def test_paths_longer_than_the_lds_go_through_in_tiles(handle):
    """corridor_bounds_kernel keeps the line's spline table and the probes of its waypoints in one CU's LDS (9 m + 33 n doubles); where a
    scenario does not fit the waypoints go through in tiles.  Forced here with a table of 2150 knots (the same line, refitted through a
    dense resampling), which leaves room for 33 waypoints at a time: 80 waypoints in three tiles, a wall in the third; against the oracle's
    walk on the same dense splines."""
    c = U.build(seed=8, n=80)
    s_end = c["scene"]["knots_s"][-1]
    ks = np.linspace(0.0, s_end, 2150)
    dx = K.spline_fit(ks, np.array([K.spline_eval(c["sx"], v) for v in ks]))
    dy = K.spline_fit(ks, np.array([K.spline_eval(c["sy"], v) for v in ks]))
    tab, ext = K.pack_spline(dx, dy)
    assert tab.shape[-1] == 2150
    g = c["geom"]
    d2 = c["dist"].copy()
    x_wall = c["ref"][72, 3]
    for i in range(g.rows):
        x, _ = K.grid_cell_position(g, i, 0)
        d2[i, :] = np.minimum(d2[i, :], np.float32(abs(x - x_wall)))
    for dist in (c["dist"], d2):
        got, nv = handle.corridor_bounds(c["ref"][None], tab[None], ext[None], dist, _geom(g))
        _compare(got[0], int(nv[0]), dict(ref=c["ref"], sx=dx, sy=dy, dist=dist, geom=g))
    assert 36 <= int(nv[0]) < 72


def test_non_default_vehicle_parameters(handle):
    c = U.build(seed=11, n=50)
    prm_o = K.CorridorParams(front_length=3.2, rear_length=-0.8, car_width=1.8, safety_margin=0.2)
    prm = handle.corridor_params(front_length=3.2, rear_length=-0.8, car_width=1.8, safety_margin=0.2)
    got, nv = handle.corridor_bounds(c["ref"][None], c["tab"][None], c["ext"][None], c["dist"], _geom(c["geom"]), prm=prm)
    _compare(got[0], int(nv[0]), c, prm_o)


def test_feeds_the_path_qp(handle):
    """bounds computed on the device go straight into pqp_path_solve (same layout) and give the oracle's path"""
    import pqp_oracle as O
    c = U.build(seed=4, n=60)
    got, nv = handle.corridor_bounds(c["ref"][None], c["tab"][None], c["ext"][None], c["dist"], _geom(c["geom"]))
    n = int(nv[0])
    assert n >= 20
    ref = c["ref"][None, :n].copy()
    scal = np.array([[0.05, 0.01, c["ref"][0, 1], c["ref"][n - 1, 2], 0.0, 35.0 * np.pi / 180.0]])
    hp = capi.Handle(capi.production_params(), device=0, max_batch=4, max_n=128)
    res = hp.solve(ref, got[:, :n].copy(), scal, passes=1)
    assert res["status"][0] == 1
    want = O.solve_path(ref[0], got[0, :n], scal[0], st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
    assert np.abs(res["out"][0][:, 3:5] - want[-1]["out"][:, 3:5]).max() < 1e-6
    hp.close()


def test_reference_states_and_initial_error(handle):
    cs = [U.build(seed=s, n=10) for s in (20, 21, 22)]
    max_s = np.array([30.0, 41.7, 18.3])
    start = np.array([[c["ref"][0, 3] + 0.3, c["ref"][0, 4] - 0.2, c["ref"][0, 2] + 0.1] for c in cs])
    ref, count, err = handle.reference_states(np.stack([c["tab"] for c in cs]), np.stack([c["ext"] for c in cs]), max_s, 400, start=start)
    for q, c in enumerate(cs):
        want = K.build_reference_from_spline(c["sx"], c["sy"], float(max_s[q]))
        assert count[q] == len(want)
        np.testing.assert_allclose(ref[q, :count[q]], want, rtol=0, atol=1e-11)       # pow / atan2: ocml vs libm
        assert np.all(ref[q, count[q]:] == 0.0)
        off, dpsi = K.process_init_state(c["sx"], c["sy"], *start[q])
        assert err[q, 0] == pytest.approx(off, abs=1e-12) and err[q, 1] == pytest.approx(dpsi, abs=1e-12)
    # fixed spacing, and a capacity smaller than the line needs: count still reports what the loop produces
    ref2, count2, _ = handle.reference_states(cs[0]["tab"][None], cs[0]["ext"][None], max_s[:1], 50, dynamic=False)
    want2 = K.build_reference_from_spline(cs[0]["sx"], cs[0]["sy"], 30.0, dynamic=False)
    assert count2[0] == len(want2) == 100          # 100 additions of 0.3 give 30.000000000000004 > 30
    np.testing.assert_allclose(ref2[0], want2[:50], rtol=0, atol=1e-11)


def test_reference_length_up_to_the_target(handle):
    """pqp_reference_length against the restatement of setReferencePathLength: targets beyond the end, beside the line, at its
    start, and the cut length fed to pqp_reference_states."""
    cs = [U.build(seed=s, n=10) for s in (40, 41, 42, 43)]
    tab, ext = np.stack([c["tab"] for c in cs]), np.stack([c["ext"] for c in cs])
    length = np.array([30.0, 30.0, 25.5, 28.0])
    def at(c, s, off):
        x, y = K.spline_eval(c["sx"], s), K.spline_eval(c["sy"], s)
        h = np.arctan2(K.spline_deriv(c["sy"], 1, s), K.spline_deriv(c["sx"], 1, s))
        return [x - off * np.sin(h), y + off * np.cos(h), h]
    target = np.array([at(cs[0], 30.0, 0.0)[:2] + [0.0], at(cs[1], 17.4, 1.2), at(cs[2], 0.3, -0.8), at(cs[3], 27.9, 0.5)])
    target[0, 0] += 3.0 * np.cos(at(cs[0], 30.0, 0.0)[2]); target[0, 1] += 3.0 * np.sin(at(cs[0], 30.0, 0.0)[2])      # 3 m beyond the end
    got = handle.reference_length(tab, ext, length, target)
    for q, c in enumerate(cs):
        want = K.reference_length(c["sx"], c["sy"], float(length[q]), target[q, 0], target[q, 1])
        assert got[q] == pytest.approx(want, abs=1e-9)
    assert got[0] == 30.0 and abs(got[1] - 17.4) < 1e-3 and abs(got[2] - 0.3) < 1e-3
    ref, count, _ = handle.reference_states(tab[1:2], ext[1:2], got[1:2], 200)
    assert count[0] == len(K.build_reference_from_spline(cs[1]["sx"], cs[1]["sy"], float(got[1])))


def test_bspline_resampling_of_the_input_points(handle):
    """pqp_bspline_resample against the restatement of ReferencePathSmoother::bSpline (all three degrees, ragged point counts, a
    scenario with too few points), then on into pqp_spline_fit -> pqp_segment_raw_reference as ReferencePathSmoother::solve chains them."""
    rng = np.random.default_rng(11)
    sets = []
    for n, spacing in ((7, 4.0), (12, 3.0), (9, 7.0), (6, 12.0), (3, 5.0)):
        sets.append(np.cumsum(np.column_stack([np.full(n, spacing), rng.uniform(-1.5, 1.5, n)]), axis=0))
    p_max = 12
    pts = np.zeros((len(sets), p_max, 2)); n_pts = np.array([len(p) for p in sets], dtype=np.int32)
    for q, p in enumerate(sets):
        pts[q, :len(p)] = p
    r = handle.bspline_resample(pts, n_pts, 96)
    assert r["count"][4] == 0                                                  # "Few reference points" (reference_path_smoother.cpp:33)
    for q, p in enumerate(sets[:4]):
        x, y, s = K.bspline_resample(p)
        n = len(x)
        assert r["count"][q] == n
        np.testing.assert_allclose(r["x"][q, :n], x, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["y"][q, :n], y, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["s"][q, :n], s, rtol=0, atol=1e-11)
        assert np.all(r["x"][q, n:] == 0.0) and np.all(r["s"][q, n:] == 0.0)
    assert handle.bspline_resample(pts[:1], n_pts[:1], 10)["count"][0] == r["count"][0]         # capacity smaller than the line needs
    q, n = 1, int(r["count"][1])                                               # on into the next two steps of the chain
    tab, ext = handle.spline_fit(r["s"][q:q + 1, :n], r["x"][q:q + 1, :n], r["y"][q:q + 1, :n])
    seg = handle.segment_raw_reference(tab, ext, r["s"][q:q + 1, n - 1], 64)
    sx, sy = K.spline_fit(r["s"][q, :n], r["x"][q, :n]), K.spline_fit(r["s"][q, :n], r["y"][q, :n])
    want = K.segment_raw_reference(sx, sy, float(r["s"][q, n - 1]))
    m = len(want[2])
    assert seg["count"][0] == m
    np.testing.assert_allclose(seg["x"][0, :m], want[0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(seg["k"][0, :m], want[4], rtol=0, atol=1e-11)


def test_raw_reference_segmentation_feeds_the_tension_smoother(handle):
    """pqp_segment_raw_reference against the restatement of segmentRawReference, then the chain TensionSmoother::smooth runs
    (tension_smoother.cpp:21-39): raw points -> spline -> 1 m samples -> TensionSmoother2 QP -> spline of the smoothed points."""
    import scipy.sparse as sp
    import pqp_oracle as O
    cs = [U.build(seed=s, n=10) for s in (30, 31, 32)]
    max_s = np.array([24.3, 24.0, 24.9])                    # 26, 25, 26 samples (the reference's loop overshoots the line)
    tab, ext = np.stack([c["tab"] for c in cs]), np.stack([c["ext"] for c in cs])
    seg = handle.segment_raw_reference(tab, ext, max_s, 40)
    for q, c in enumerate(cs):
        x, y, s, ang, k = K.segment_raw_reference(c["sx"], c["sy"], float(max_s[q]))
        n = len(s)
        assert seg["count"][q] == n
        np.testing.assert_array_equal(seg["s"][q, :n], s)
        for name, want in (("x", x), ("y", y), ("angle", ang), ("k", k)):
            np.testing.assert_allclose(seg[name][q, :n], want, rtol=0, atol=1e-11)      # pow / atan2: ocml vs libm
            assert np.all(seg[name][q, n:] == 0.0)
    # a capacity smaller than the line needs: count still reports what the loop produces
    assert handle.segment_raw_reference(tab[:1], ext[:1], max_s[:1], 10)["count"][0] == 26
    # the two 26-sample lines as one smoother batch (shared sparsity needs equal n), checked against the converged oracle QP
    idx, n = [0, 2], 26
    hs = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25), max_batch=2, max_n=n)
    r = hs.smooth_tension2(*(seg[k][idx, :n] for k in ("x", "y", "angle", "k", "s")))
    assert (r["status"] == 1).all()
    for j, q in enumerate(idx):
        P, qv, A, lo, up = O.assemble_tension2(*(seg[k][q, :n] for k in ("x", "y", "angle", "k", "s")))
        ref = O.osqp_admm(sp.csc_matrix(P), qv, A, lo, up, O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=100000))
        assert np.abs(r["x"][j] - ref["x"][:n]).max() < 1e-5 and np.abs(r["y"][j] - ref["x"][n:2 * n]).max() < 1e-5
    fit_tab, fit_ext = handle.spline_fit(r["s"], r["x"], r["y"])                         # tension_smoother.cpp:36-38
    sx = K.spline_fit(r["s"][0], r["x"][0])
    _tab_close(fit_tab[0], None, K.pack_spline(sx, K.spline_fit(r["s"][0], r["y"][0]))[0])
    hs.close()


def test_pipeline_spline_to_path_on_the_device(handle):
    """spline coefficients -> reference states -> corridor bounds -> path QP with a waypoint count per scenario, all on the
    device, against the same chain of oracles (reference call order: path_optimizer.cpp:110-161)."""
    import torch
    import pqp_oracle as O
    dev = torch.device("cuda", 0)
    cs = [U.build(seed=s, n=10) for s in (30, 31, 33)]
    g = cs[0]["geom"]
    B, n_max = len(cs), 128
    max_s = np.array([24.0, 30.0, 36.0])
    tab = torch.from_numpy(np.stack([c["tab"] for c in cs])).to(dev); ext = torch.from_numpy(np.stack([c["ext"] for c in cs])).to(dev)
    dist = torch.from_numpy(np.ascontiguousarray(np.transpose(np.stack([c["dist"] for c in cs]), (0, 2, 1)))).to(dev)
    start = np.array([[c["ref"][0, 3] + 0.1, c["ref"][0, 4] + 0.15, c["ref"][0, 2] - 0.05] for c in cs])
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    ref = torch.zeros((B, n_max, 5), dtype=torch.float64, device=dev); count = torch.zeros(B, dtype=torch.int32, device=dev)
    err = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    bounds = torch.zeros((B, n_max, 6), dtype=torch.float64, device=dev); nv = torch.zeros(B, dtype=torch.int32, device=dev)
    out = torch.zeros((B, n_max, 7), dtype=torch.float64, device=dev); status = torch.zeros(B, dtype=torch.int32, device=dev)
    p = lambda x: capi.C.c_void_p(x.data_ptr())
    hp = capi.Handle(capi.production_params(), device=0, max_batch=B, max_n=n_max)
    lib, hh = hp.lib, hp._h
    m = tab.shape[2]
    ms, st_d, mo = t(max_s), t(start), torch.arange(B, dtype=torch.int32, device=dev)
    assert lib.pqp_reference_states_device(hh, B, n_max, m, p(tab), p(ext), p(ms), p(st_d), 0.15, 0.3, 1, p(ref), p(count), p(err)) == 0
    geom = _geom(g); prm = hp.corridor_params()
    assert lib.pqp_corridor_bounds_device(hh, B, n_max, m, p(ref), p(count), p(tab), p(ext), p(dist), p(mo), capi.C.byref(geom), capi.C.byref(prm), p(bounds), p(nv)) == 0
    hp.sync()
    ref_h, nv_h, err_h = ref.cpu().numpy(), nv.cpu().numpy(), err.cpu().numpy()
    scal = np.zeros((B, 6))
    for q in range(B):
        nq = int(nv_h[q])
        scal[q] = (err_h[q, 0], err_h[q, 1], ref_h[q, 0, 1], ref_h[q, max(nq - 1, 0), 2], 1.0 if nq < int(count[q]) else 0.0, 35.0 * np.pi / 180.0)
    hp.solve_var_device(B, n_max, nv, ref, bounds, t(scal), out, passes=1, status=status)
    hp.sync()
    out_h, bounds_h = out.cpu().numpy(), bounds.cpu().numpy()
    solved = 0
    for q, c in enumerate(cs):
        want_ref = K.build_reference_from_spline(c["sx"], c["sy"], float(max_s[q]))
        assert int(count[q]) == len(want_ref) <= n_max
        want_b, want_nv, _ = K.update_bounds_improved(want_ref, c["sx"], c["sy"], c["dist"], g)
        assert nv_h[q] == want_nv
        nq = int(want_nv)
        if nq < 2:
            continue
        assert status[q] == 1
        want = O.solve_path(ref_h[q, :nq], bounds_h[q, :nq], scal[q], st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
        assert np.abs(out_h[q, :nq, 3:5] - want[-1]["out"][:, 3:5]).max() < 1e-6
        solved += 1
    assert solved >= 2
    hp.close()


SPLINE_TOL = 1e-13      # of the largest coefficient of a table row: the device solves the moment equations by a Thomas recurrence of its own (FMA
                        # contraction allowed), the reference by a row-normalised band LU - same spline, different round-off (measured: < 1e-15)


def _ref_spline_lib():
    """the reference's own tk::spline, compiled from where it lies (oracle/_ref, built by oracle/Makefile; travels to the GPU box prebuilt)"""
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_spline.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_spline_new.restype = C.c_void_p; lib.ref_spline_new.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.ref_spline_deriv.restype = C.c_double; lib.ref_spline_deriv.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.ref_spline_eval.restype = C.c_double; lib.ref_spline_eval.argtypes = [C.c_void_p, C.c_double]
    lib.ref_spline_free.argtypes = [C.c_void_p]
    return lib


def _rows_close(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    scale = max(float(np.abs(want).max()), 1e-300)
    err = float(np.abs(got - want).max()) / scale
    assert err <= SPLINE_TOL, (what, err)
    return err


def _tab_close(tab, ext, want_tab, want_ext=None):
    """a device spline table against a restatement / golden one: knots and values exact, coefficient rows to SPLINE_TOL of the row's largest"""
    assert np.array_equal(np.asarray(tab)[[0, 1, 5]], np.asarray(want_tab)[[0, 1, 5]])
    for r in (2, 3, 4, 6, 7, 8):
        _rows_close(tab[r], want_tab[r], ("row", r))
    if want_ext is not None:
        assert np.abs(np.asarray(ext) - np.asarray(want_ext)).max() <= SPLINE_TOL * max(1.0, float(np.abs(want_ext).max()))


def test_spline_fit_matches_the_reference_build(handle):
    """The natural cubic spline of tk::spline::set_points on the device (its own Thomas recurrence on the moment equations): the coefficient
    table against (i) the reference's compiled tk::spline itself - cubic, quadratic and linear coefficient of every segment read off its
    third, second and first derivative just right of the knot - and (ii) the restatement (which equals the reference build bit for bit,
    tests/test_corridor_oracle.py), to 1e-13 of a row's largest coefficient; knots and values are copied: exact."""
    import ctypes as C
    rng = np.random.default_rng(5)
    B, m = 6, 37
    s = np.cumsum(rng.uniform(0.2, 3.0, (B, m)), axis=1)
    x = np.cumsum(rng.normal(size=(B, m)), axis=1); y = np.cumsum(rng.normal(size=(B, m)), axis=1)
    tab, ext = handle.spline_fit(s, x, y)
    lib = _ref_spline_lib()
    worst = 0.0
    for q in range(B):
        want_tab, want_ext = K.pack_spline(K.spline_fit(s[q], x[q]), K.spline_fit(s[q], y[q]))
        assert np.array_equal(tab[q][[0, 1, 5]], want_tab[[0, 1, 5]])
        for r in (2, 3, 4, 6, 7, 8):
            worst = max(worst, _rows_close(tab[q][r], want_tab[r], ("restatement", q, r)))
        assert np.abs(ext[q] - want_ext).max() <= SPLINE_TOL * max(1.0, np.abs(want_ext).max())
        if lib is not None:
            for vals, r0 in ((x[q], 1), (y[q], 5)):
                sq, vq = np.ascontiguousarray(s[q]), np.ascontiguousarray(vals)
                hd = lib.ref_spline_new(m, sq.ctypes.data, vq.ctypes.data)
                just_right = np.nextafter(sq[:-1], np.inf)
                a_ref = np.array([lib.ref_spline_deriv(hd, 3, t) for t in just_right]) / 6.0
                b_ref = np.array([lib.ref_spline_deriv(hd, 2, t) for t in just_right]) / 2.0
                c_ref = np.array([lib.ref_spline_deriv(hd, 1, t) for t in just_right])
                _rows_close(tab[q][r0 + 1][:-1], a_ref, ("reference a", q, r0))
                _rows_close(tab[q][r0 + 2][:-1], b_ref, ("reference b", q, r0))
                _rows_close(tab[q][r0 + 3][:-1], c_ref, ("reference c", q, r0))
                # the right-hand extrapolation (slope at the last knot) and the left one
                beyond = float(sq[-1] + 2.5)
                assert abs(lib.ref_spline_deriv(hd, 1, beyond) - (2.0 * tab[q][r0 + 2][-1] * 2.5 + tab[q][r0 + 3][-1])) <= SPLINE_TOL * max(1.0, np.abs(c_ref).max())
                before = float(sq[0] - 1.5)
                e0 = 0 if r0 == 1 else 2
                assert abs(lib.ref_spline_deriv(hd, 1, before) - (2.0 * ext[q][e0] * (-1.5) + ext[q][e0 + 1])) <= SPLINE_TOL * max(1.0, np.abs(c_ref).max())
                lib.ref_spline_free(hd)
    # smallest legal size
    tab3, ext3 = handle.spline_fit(s[:1, :3], x[:1, :3], y[:1, :3])
    w3, e3 = K.pack_spline(K.spline_fit(s[0, :3], x[0, :3]), K.spline_fit(s[0, :3], y[0, :3]))
    for r in range(9):
        _rows_close(tab3[0][r], w3[r], ("n = 3", r))
    assert np.abs(ext3[0] - e3).max() <= SPLINE_TOL * max(1.0, np.abs(e3).max())


@pytest.mark.parametrize("seed,length", [(0, 40.0), (1, 33.0), (2, 40.0), (6, 5.0), (12, 25.7)])
def test_dp_corridor_search(handle, seed, length):
    """graphSearchDp on the device against the restatement: same number of layers, same layer abscissae, same bounds
    (they are sums of 0.6 m and 0.2 m steps: equal to round-off unless a sample sits on the 1.2 m threshold)."""
    c = U.build(seed=seed, n=10)
    start = np.array([[c["ref"][0, 3] + 0.2, c["ref"][0, 4] + 0.5, c["ref"][0, 2] + 0.05]])
    ls, lb, ub, count, vl = handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([length]), start, c["dist"], _geom(c["geom"]))
    want = K.graph_search_dp(c["sx"], c["sy"], length, tuple(start[0]), c["dist"], c["geom"])
    assert want is not None and count[0] == len(want["layers_s"])
    k = int(count[0])
    assert vl[0] == pytest.approx(want["vehicle_l"], abs=1e-12)
    np.testing.assert_allclose(ls[0, :k], want["layers_s"], rtol=0, atol=1e-11)
    same = (np.abs(lb[0, :k] - want["lb"]) < 1e-9) & (np.abs(ub[0, :k] - want["ub"]) < 1e-9)
    assert same.mean() > 0.95, (lb[0, :k], want["lb"], ub[0, :k], want["ub"])


@pytest.mark.parametrize("rng,spacing,lon,length", [(10.0, 0.6, 1.5, 40.0),      # the reference's grid: 34 samples, window of 7 predecessors
                                                    (3.0, 0.2, 1.5, 30.0),       # 31 samples, window of 17
                                                    (1.5, 0.6, 1.5, 30.0),       # 6 samples: the window covers the whole layer
                                                    (6.0, 1.0, 0.5, 40.0)])      # 80 layers: more than one chunk of 32 prepared layers
def test_dp_corridor_search_on_other_grids(handle, rng, spacing, lon, length):
    """The admissible-predecessor window, the chunked preparation of the layers' nodes and the parallel bound refinement must give what
    the reference's serial loops give, whatever the lateral grid and the number of layers."""
    c = U.build(seed=2, n=10)
    start = np.array([[c["ref"][0, 3] + 0.1, c["ref"][0, 4] + 0.3, c["ref"][0, 2] + 0.02]])
    prm = capi.PqpDpParams(rng, lon, spacing, 2.0)
    ls, lb, ub, count, vl = handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([length]), start, c["dist"], _geom(c["geom"]), prm=prm, max_layers=96)
    want = K.graph_search_dp(c["sx"], c["sy"], length, tuple(start[0]), c["dist"], c["geom"],
                             prm=K.DpParams(lateral_range=rng, longitudinal_spacing=lon, lateral_spacing=spacing))
    if want is None:
        assert count[0] == 0
        return
    assert count[0] == len(want["layers_s"])
    k = int(count[0])
    np.testing.assert_allclose(ls[0, :k], want["layers_s"], rtol=0, atol=1e-11)
    same = (np.abs(lb[0, :k] - want["lb"]) < 1e-9) & (np.abs(ub[0, :k] - want["ub"]) < 1e-9)
    assert same.mean() > 0.95, (lb[0, :k], want["lb"], ub[0, :k], want["ub"])


def test_dp_corridor_edge_cases(handle):
    c = U.build(seed=3, n=10)
    g = _geom(c["geom"])
    far = np.array([[c["ref"][0, 3], c["ref"][0, 4] + 15.0, c["ref"][0, 2]]])            # vehicle 15 m off the line: graphSearchDp returns false
    _, _, _, count, vl = handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([30.0]), far, c["dist"], g)
    assert count[0] == 0 and abs(vl[0]) > 10.0
    assert K.graph_search_dp(c["sx"], c["sy"], 30.0, tuple(far[0]), c["dist"], c["geom"]) is None
    near = np.array([[c["ref"][0, 3], c["ref"][0, 4], c["ref"][0, 2]]])
    _, _, _, count, _ = handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([30.0]), near, c["dist"], g, max_layers=8)
    assert count[0] == -1                                                               # 21 layers do not fit 8


def test_dp_corridor_feeds_post_smooth(handle):
    """layers_s / lb / ub / vehicle_l of the search are the inputs of the postSmooth QP (reference_path_smoother.cpp:44)"""
    c = U.build(seed=1, n=10)
    start = np.array([[c["ref"][0, 3] + 0.1, c["ref"][0, 4] + 0.3, c["ref"][0, 2]]])
    ls, lb, ub, count, vl = handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([33.0]), start, c["dist"], _geom(c["geom"]))
    k = int(count[0])
    assert k >= 4
    res = handle.post_smooth(ls[:, :k].copy(), lb[:, :k].copy(), ub[:, :k].copy(), vl.copy())
    assert res["status"][0] == 1
    l = res["l"][0]
    assert abs(l[0] - vl[0]) < 1e-3 and np.all(l[1:] >= lb[0, 1:k] - 1e-3) and np.all(l[1:] <= ub[0, 1:k] + 1e-3)
    # the tail of postSmooth (:559-576): offsets -> points with chord-length abscissae -> the final reference line's splines
    full_l = np.zeros_like(ls); full_l[0, :k] = l
    x, y, s = handle.offsets_to_points(c["tab"][None], c["ext"][None], ls, full_l, m_of=count)
    wx, wy, ws = K.offsets_to_points(c["sx"], c["sy"], ls[0, :k], l)
    np.testing.assert_allclose(x[0, :k], wx, rtol=0, atol=1e-11)           # atan2 / sin / cos: ocml vs libm
    np.testing.assert_allclose(y[0, :k], wy, rtol=0, atol=1e-11)
    np.testing.assert_allclose(s[0, :k], ws, rtol=0, atol=1e-10)
    assert np.all(x[0, k:] == 0.0) and np.all(s[0, k:] == 0.0)
    tab, ext = handle.spline_fit(s[:, :k].copy(), x[:, :k].copy(), y[:, :k].copy())
    _tab_close(tab[0], None, K.pack_spline(K.spline_fit(s[0, :k], x[0, :k]), K.spline_fit(s[0, :k], y[0, :k]))[0])
    ref, cnt, _ = handle.reference_states(tab, ext, s[:, k - 1].copy(), 200)      # and on to the path QP's reference states
    assert cnt[0] == len(K.build_reference_from_spline(K.spline_fit(s[0, :k], x[0, :k]), K.spline_fit(s[0, :k], y[0, :k]), float(s[0, k - 1])))


@pytest.mark.parametrize("name", ["scene_a", "scene_b"])
def test_golden_scene_fixtures_through_the_hip_chain(handle, name):
    """tests/golden/scene_*.npz: knots -> spline -> states -> bounds, and the DP search, all through the C ABI (no oracle here)."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    gv = f["geom"]
    geom = capi.PqpGridGeometry(int(gv[0]), int(gv[1]), *[float(v) for v in gv[2:]])
    tab, ext = handle.spline_fit(f["knots_s"][None], f["knots_x"][None], f["knots_y"][None])
    _tab_close(tab[0], ext[0], f["spline"], f["spline_ext"])
    ref, count, err = handle.reference_states(tab, ext, np.array([float(f["length"])]), 128, start=f["start"][None])
    n = int(count[0])
    assert n == len(f["ref"])
    np.testing.assert_allclose(ref[0, :n], f["ref"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(err[0], f["init_err"], rtol=0, atol=1e-12)
    bounds, nv = handle.corridor_bounds(ref, tab, ext, f["dist"], geom, n_of=count)
    assert int(nv[0]) == int(f["n_valid"])
    k = int(nv[0])
    assert (np.abs(bounds[0, :k] - f["bounds"]) < 1e-9).mean() > 0.99
    ls, lb, ub, cnt, vl = handle.dp_corridor(tab, ext, np.array([float(f["length"])]), f["start"][None], f["dist"], geom)
    c = int(cnt[0])
    assert c == len(f["dp_layers_s"])
    np.testing.assert_allclose(ls[0, :c], f["dp_layers_s"], rtol=0, atol=1e-11)
    assert ((np.abs(lb[0, :c] - f["dp_lb"]) < 1e-9) & (np.abs(ub[0, :c] - f["dp_ub"]) < 1e-9)).mean() > 0.95
    assert vl[0] == pytest.approx(float(f["dp_vehicle_l"]), abs=1e-12)


def test_golden_line_fixture_through_the_hip_chain(handle):
    """tests/golden/line_a.npz: input points -> pqp_bspline_resample -> pqp_spline_fit -> pqp_segment_raw_reference, plus
    pqp_offsets_to_points and pqp_reference_length on that spline, against the committed vectors."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "line_a.npz"))
    pts = f["points"]
    r = handle.bspline_resample(pts[None], np.array([len(pts)]), 64)
    n = len(f["raw_x"])
    assert r["count"][0] == n
    np.testing.assert_allclose(r["x"][0, :n], f["raw_x"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["s"][0, :n], f["raw_s"], rtol=0, atol=1e-11)
    tab, ext = handle.spline_fit(f["raw_s"][None], f["raw_x"][None], f["raw_y"][None])
    _tab_close(tab[0], ext[0], f["spline"], f["spline_ext"])
    seg = handle.segment_raw_reference(tab, ext, f["raw_s"][-1:].copy(), 64)
    m = len(f["seg_s"])
    assert seg["count"][0] == m
    np.testing.assert_array_equal(seg["s"][0, :m], f["seg_s"])
    for key, name in (("x", "seg_x"), ("y", "seg_y"), ("angle", "seg_angle"), ("k", "seg_k")):
        np.testing.assert_allclose(seg[key][0, :m], f[name], rtol=0, atol=1e-11)
    x, y, s = handle.offsets_to_points(tab, ext, f["at_s"][None], f["offsets"][None])
    np.testing.assert_allclose(x[0], f["off_x"], rtol=0, atol=1e-11); np.testing.assert_allclose(s[0], f["off_s"], rtol=0, atol=1e-10)
    cut = handle.reference_length(tab, ext, f["raw_s"][-1:].copy(), f["target"][None])
    assert cut[0] == pytest.approx(float(f["cut_length"]), abs=1e-9)


def test_argument_errors(handle):
    """bad arguments come back as PQP_ERR_INVALID / PQP_ERR_CAPACITY (no exception crosses the ABI, nothing is launched)"""
    c = U.build(seed=0, n=8)
    g = _geom(c["geom"])
    with pytest.raises(capi.PqpError):
        handle.spline_fit(np.zeros((1, 2)), np.zeros((1, 2)), np.zeros((1, 2)))                      # fewer than 3 knots (spline.cpp:164)
    with pytest.raises(capi.PqpError):
        handle.reference_states(c["tab"][None], c["ext"][None], np.array([10.0]), 64, ds_small=0.3, ds_large=0.1)   # CHECK_LE (:315)
    bad = capi.PqpGridGeometry(1, 1, 0.2, 0.2, 0.2, 0.0, 0.0)
    with pytest.raises(capi.PqpError):
        handle.corridor_bounds(c["ref"][None], c["tab"][None], c["ext"][None], np.zeros((1, 1), dtype=np.float32), bad)
    with pytest.raises(capi.PqpError):
        prm = capi.PqpDpParams(10.0, 1.5, 0.1, 2.0)                                                # 201 lateral samples: more than one wavefront
        handle.dp_corridor(c["tab"][None], c["ext"][None], np.array([10.0]), np.zeros((1, 3)), c["dist"], g, prm=prm)
    # a batch of one and a line shorter than one step are fine
    ref, count, _ = handle.reference_states(c["tab"][None], c["ext"][None], np.array([0.1]), 8)
    assert count[0] == 1 and ref[0, 0, 0] == 0.0
    # the entry points added around the smoother QPs
    with pytest.raises(capi.PqpError):
        handle.bspline_resample(np.zeros((1, 3, 2)), np.array([3]), 32)                             # p_max < 4 (reference_path_smoother.cpp:33)
    with pytest.raises(capi.PqpError):
        handle.segment_raw_reference(c["tab"][None], c["ext"][None], np.array([10.0]), 32, delta_s=0.0)
    with pytest.raises(capi.PqpError):
        handle.offsets_to_points(c["tab"][None, :, :2], c["ext"][None], np.zeros((1, 4)), np.zeros((1, 4)))   # a 2-knot spline
    with pytest.raises(capi.PqpError):
        handle.bspline_resample(np.zeros((1, 8, 2)), np.array([8]), 8192)                           # 3 x 8192 samples do not fit one CU's LDS
    # a degenerate input polygon (all points equal): length 0, one sample at t = 0 and the one at t = 1
    r = handle.bspline_resample(np.ones((1, 6, 2)), np.array([6]), 16)
    assert r["count"][0] == 2 and np.all(r["x"][0, :2] == 1.0) and np.all(r["s"][0, :2] == 0.0)
    # a target exactly at the line's end leaves the length alone or cuts it at the end: the same number either way
    L = float(c["tab"][0, -1])
    end = [K.spline_eval(c["sx"], L), K.spline_eval(c["sy"], L), 0.0]
    assert handle.reference_length(c["tab"][None], c["ext"][None], np.array([L]), np.array([end]))[0] == pytest.approx(L, abs=1e-9)
