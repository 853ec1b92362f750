"""The CPU restatement of the corridor-bounds step (oracle/corridor_oracle.py) against what can be pinned here:
the reference's own tk::spline compiled from /root/reference (oracle/_ref, bit for bit), and analytic known answers
for the restated grid_map lookups and the clearance search."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import corridor_oracle as K
import corridor_util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SPLINE = os.path.join(ROOT, "oracle", "_ref", "libref_spline.so")


@pytest.mark.skipif(not os.path.exists(REF_SPLINE), reason="oracle/_ref not built (needs /root/reference: `make -C oracle ref`)")
def test_spline_restatement_is_the_reference_bit_for_bit():
    lib = C.CDLL(REF_SPLINE)
    lib.ref_spline_new.restype = C.c_void_p; lib.ref_spline_new.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.ref_spline_eval.restype = C.c_double; lib.ref_spline_eval.argtypes = [C.c_void_p, C.c_double]
    lib.ref_spline_deriv.restype = C.c_double; lib.ref_spline_deriv.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.ref_spline_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(0)
    for _ in range(12):
        n = int(rng.integers(3, 60))
        x = np.cumsum(rng.uniform(0.2, 3.0, n)); y = np.cumsum(rng.normal(size=n))
        sp = K.spline_fit(x, y)
        h = lib.ref_spline_new(n, x.ctypes.data, y.ctypes.data)
        for t in np.concatenate([x, rng.uniform(x[0] - 3, x[-1] + 3, 120)]):      # knots, interior, both extrapolations
            t = float(t)
            assert K.spline_eval(sp, t) == lib.ref_spline_eval(h, t)
            for order in (1, 2, 3):
                assert K.spline_deriv(sp, order, t) == lib.ref_spline_deriv(h, order, t)
        lib.ref_spline_free(h)


def test_spline_interpolates_and_is_c2():
    x = np.array([0.0, 1.0, 2.5, 4.0, 7.0]); y = np.array([1.0, -1.0, 0.5, 0.0, 2.0])
    sp = K.spline_fit(x, y)
    for xi, yi in zip(x, y):
        assert K.spline_eval(sp, float(xi)) == pytest.approx(yi, abs=1e-14)
    for xi in x[1:-1]:
        for order in (1, 2):
            assert K.spline_deriv(sp, order, xi - 1e-9) == pytest.approx(K.spline_deriv(sp, order, xi + 1e-9), abs=1e-6)
    assert K.spline_deriv(sp, 2, 0.0) == pytest.approx(0.0, abs=1e-14) and K.spline_deriv(sp, 2, 7.0) == pytest.approx(0.0, abs=1e-12)


def _plane_map(g, a, b, c):
    """distance layer = a + b x + c y sampled at the cell centres (bilinear interpolation reproduces a plane exactly)"""
    d = np.zeros((g.rows, g.cols), dtype=np.float32)
    for i in range(g.rows):
        for j in range(g.cols):
            x, y = K.grid_cell_position(g, i, j)
            d[i, j] = a + b * x + c * y
    return d


def test_grid_conventions():
    g = K.GridGeom.make(10.0, 6.0, 0.5, pos=(1.0, -2.0))
    assert (g.rows, g.cols) == (20, 12)
    # cell (0, 0) is the +x / +y corner, indices grow towards -x / -y
    assert K.grid_cell_position(g, 0, 0) == pytest.approx((1.0 + 5.0 - 0.25, -2.0 + 3.0 - 0.25))
    assert K.grid_index(g, 5.9, 0.9) == (0, 0) and K.grid_index(g, -3.9, -4.9) == (19, 11)
    for i, j in ((0, 0), (3, 7), (19, 11)):
        assert K.grid_index(g, *K.grid_cell_position(g, i, j)) == (i, j)
    assert K.grid_is_inside(g, 5.99, 0.99) and not K.grid_is_inside(g, 6.01, 0.0) and not K.grid_is_inside(g, 0.0, -5.01)
    assert K.obstacle_distance(np.ones((20, 12), dtype=np.float32), g, 100.0, 0.0) == 0.0          # outside -> 0 (Map.cpp:20)


def test_bilinear_lookup_reproduces_a_plane():
    g = K.GridGeom.make(12.0, 8.0, 0.25)
    d = _plane_map(g, 3.0, 0.2, -0.1)
    rng = np.random.default_rng(1)
    for _ in range(300):
        x, y = rng.uniform(-5.5, 5.5), rng.uniform(-3.5, 3.5)
        assert K.obstacle_distance(d, g, x, y) == pytest.approx(3.0 + 0.2 * x - 0.1 * y, abs=2e-6)      # float32 layer


def test_clearance_between_two_walls():
    """Straight corridor along x between walls at y = +3 and y = -2: distance field = min(3 - y, y + 2).  The search walks
    0.3 m steps until the field drops below 0.5, then 0.05 m steps, then subtracts car_width/2 - 0.5 and the safety margin."""
    g = K.GridGeom.make(40.0, 16.0, 0.1)
    d = np.zeros((g.rows, g.cols), dtype=np.float32)
    for j in range(g.cols):
        _, y = K.grid_cell_position(g, 0, j)
        d[:, j] = max(min(3.0 - y, y + 2.0), 0.0)
    left, right = K.clearance_strict(0.0, 0.0, 0.0, d, g)
    # left: the last 0.3-step with field >= 0.5 is 2.4 (at 2.7 the field is 0.3); fine steps 2.45, 2.5 pass, 2.55 fails -> 2.5;
    # minus (car_width/2 - 0.5) = 0.5, minus the safety margin 0.3
    assert left == pytest.approx(2.5 - 0.5 - 0.3, abs=1e-9)
    # right: the coarse search stops at 1.8 -> -1.5.  Two reference quirks are reproduced as written (:289-299):
    #   int(0.3 / 0.05) = 5 (5.999...), so the fine loop makes 4 steps, not 5;
    #   the fine probe is state + right_bound * (cos, sin)(right_angle) with right_bound NEGATIVE, i.e. it samples the LEFT
    #   side of the road - free here, so all 4 steps pass and the right bound moves out to -1.7 although the wall is at -1.5
    assert right == pytest.approx(-(1.5 + 4 * 0.05) + 0.5 + 0.3, abs=1e-9)
    assert K.clearance_strict(0.0, 2.8, 0.0, d, g) == (0.0, 0.0)                  # starts inside the inflated obstacle


def test_update_bounds_shapes_offsets_and_blocking():
    c = U.build(seed=3, n=30)
    bounds, n_valid, blocked = K.update_bounds_improved(c["ref"], c["sx"], c["sy"], c["dist"], c["geom"])
    assert bounds.shape == (n_valid, 6)
    assert (bounds[:, 1] >= bounds[:, 0]).all() and (bounds[:, 3] >= bounds[:, 2]).all()        # ub >= lb
    # a wall across the road: everything beyond it is cut off and the blocked row has an empty front or rear interval
    d2 = c["dist"].copy()
    x_wall = c["ref"][20, 3]
    g = c["geom"]
    for i in range(g.rows):
        x, _ = K.grid_cell_position(g, i, 0)
        d2[i, :] = np.minimum(d2[i, :], np.float32(abs(x - x_wall)))
    b2, nv2, blocked2 = K.update_bounds_improved(c["ref"], c["sx"], c["sy"], d2, g)
    assert nv2 < 20 and blocked2 is not None
    assert abs(blocked2[1] - blocked2[0]) < 1e-6 or abs(blocked2[3] - blocked2[2]) < 1e-6


def test_reference_state_sampling():
    """buildReferenceFromSpline: on a circle of radius R the curvature is 1/R everywhere (up to the spline's error), the
    heading turns with s, and the spacing follows the curvature rule (0.3 below k = 0.08, 0.15 above 0.2, linear between)."""
    for R, step in ((50.0, 0.3), (4.0, 0.15), (1.0 / 0.14, 0.3 - 0.5 * 0.15)):
        s = np.linspace(0.0, 12.0, 25)
        sx = K.spline_fit(s, R * np.sin(s / R)); sy = K.spline_fit(s, R * (1.0 - np.cos(s / R)))
        ref = K.build_reference_from_spline(sx, sy, 9.0)
        inner = ref[(ref[:, 0] > 3.0) & (ref[:, 0] < 7.0)]          # away from the natural spline's end effects
        assert np.abs(inner[:, 1] - 1.0 / R).max() < 2e-3
        assert np.abs(inner[:, 2] - inner[:, 0] / R).max() < 2e-3
        assert np.abs(np.diff(inner[:, 0]) - step).max() < 5e-3
        assert ref[0, 0] == 0.0 and ref[-1, 0] <= 9.0 < ref[-1, 0] + 0.3
    fixed = K.build_reference_from_spline(sx, sy, 9.0, dynamic=False)
    assert np.allclose(np.diff(fixed[:, 0]), 0.3)


def test_offsets_to_points():
    """postSmooth's tail: on a straight line along x the points are (s_i, l_i) and the new abscissa is their chord length."""
    s = np.linspace(0.0, 40.0, 41)
    sx = K.spline_fit(s, s.copy()); sy = K.spline_fit(s, np.zeros_like(s))
    at = np.array([0.0, 3.0, 6.0, 10.0]); l = np.array([0.5, -1.0, 0.0, 2.0])
    x, y, ss = K.offsets_to_points(sx, sy, at, l)
    assert np.abs(x - at).max() < 1e-12 and np.abs(y - l).max() < 1e-12
    assert np.abs(ss - np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(at), np.diff(l)))])).max() < 1e-12


def test_reference_length_up_to_the_target():
    """setReferencePathLength: a target beyond the end of the line leaves the length alone, a target beside the line cuts it at
    the target's projection (on a straight line: the target's own abscissa)."""
    s = np.linspace(0.0, 40.0, 41)
    sx = K.spline_fit(s, s.copy()); sy = K.spline_fit(s, np.zeros_like(s))
    assert K.reference_length(sx, sy, 40.0, 45.0, 1.0) == 40.0
    assert K.reference_length(sx, sy, 40.0, 40.0, -2.0) == 40.0                # x == 0 in the end frame: cut ... at the end itself
    assert abs(K.reference_length(sx, sy, 40.0, 27.3, 1.5) - 27.3) < 1e-9
    R = 30.0
    cx = K.spline_fit(s, R * np.sin(s / R)); cy = K.spline_fit(s, R * (1 - np.cos(s / R)))
    phi = 22.0 / R
    got = K.reference_length(cx, cy, 40.0, (R - 2.0) * np.sin(phi), R - (R - 2.0) * np.cos(phi))    # 2 m inside the arc at s = 22
    assert abs(got - 22.0) < 1e-3


def test_bspline_resampling():
    """bSpline(): the restated clamped knot vector + de Boor against scipy's independent B-spline evaluation (tinyspline itself is
    not in this image: parity unpinned, see the oracle), the degree rule, the sampling loop and the chord-length abscissae."""
    from scipy.interpolate import BSpline
    rng = np.random.default_rng(5)
    for n, spacing, degree in ((7, 4.0, 5), (9, 7.0, 4), (6, 12.0, 3)):
        pts = np.cumsum(np.column_stack([np.full(n, spacing), rng.uniform(-1.0, 1.0, n)]), axis=0)
        kn = K.clamped_knots(n, degree)
        assert len(kn) == n + degree + 1 and list(kn[:degree + 1]) == [0.0] * (degree + 1) and list(kn[-degree - 1:]) == [1.0] * (degree + 1)
        assert np.allclose(np.diff(kn[degree:n + 1]), 1.0 / (n - degree))               # uniform interior knots
        sp = BSpline(kn, pts, degree)
        for t in np.linspace(0.0, 1.0, 41):
            assert np.abs(K.bspline_eval(pts, kn, degree, t) - sp(t)).max() < 1e-12
        x, y, s = K.bspline_resample(pts)
        length = np.hypot(*np.diff(pts, axis=0).T).sum()
        assert len(x) in (int(np.ceil(length)) + 1, int(np.ceil(length)) + 2)           # t = k / length while < 1 (accumulated), then t = 1
        assert (x[0], y[0]) == tuple(pts[0]) and (x[-1], y[-1]) == tuple(pts[-1])        # a clamped spline starts and ends on its polygon
        assert np.abs(K.bspline_eval(pts, kn, degree, 3.0 / length) - (x[3], y[3])).max() < 1e-9
        assert s[0] == 0.0 and np.all(np.diff(s) > 0) and s[-1] <= length + 1e-9 and s[-1] > 0.9 * length
    line = np.column_stack([np.linspace(0.0, 30.0, 8), np.linspace(0.0, 15.0, 8)])
    x, y, s = K.bspline_resample(line)                                                  # collinear control points: the spline is the segment
    assert np.abs(y - 0.5 * x).max() < 1e-12 and np.all(np.diff(x) > 0)


def test_raw_reference_segmentation():
    """segmentRawReference: samples every metre from 0 up to and including the first abscissa >= max_s (the reference's loop
    overshoots the line unless its length is a whole number of metres); on a circle: angle = s / R, k = 1 / R."""
    R = 20.0
    s = np.linspace(0.0, 30.0, 61)
    sx = K.spline_fit(s, R * np.sin(s / R)); sy = K.spline_fit(s, R * (1.0 - np.cos(s / R)))
    x, y, sl, ang, k = K.segment_raw_reference(sx, sy, 24.3)
    assert list(sl) == [float(i) for i in range(26)]                 # 0 .. 25: the last one lies 0.7 m beyond max_s
    inner = (sl > 3) & (sl < 22)
    assert np.abs(ang[inner] - sl[inner] / R).max() < 1e-4 and np.abs(k[inner] - 1 / R).max() < 1e-4
    assert np.abs(x - R * np.sin(sl / R)).max() < 1e-4 and np.abs(y - R * (1 - np.cos(sl / R))).max() < 1e-4
    assert len(K.segment_raw_reference(sx, sy, 24.0)[2]) == 25       # a whole number of metres: 0 .. 24, no overshoot
    assert list(K.segment_raw_reference(sx, sy, 2.1, delta_s=0.5)[2]) == [0.0, 0.5, 1.0, 1.5, 2.0, 2.5]


def test_initial_error():
    s = np.linspace(0.0, 10.0, 11)
    sx = K.spline_fit(s, s); sy = K.spline_fit(s, 0.0 * s)          # the x axis
    off, dpsi = K.process_init_state(sx, sy, 0.0, 0.7, 0.2)         # vehicle 0.7 m to the left of the line, heading +0.2
    assert off == pytest.approx(0.7) and dpsi == pytest.approx(0.2)
    off, dpsi = K.process_init_state(sx, sy, 0.0, -0.4, -0.1)
    assert off == pytest.approx(-0.4) and dpsi == pytest.approx(-0.1)


def test_dp_search_on_an_empty_map_and_past_a_single_obstacle():
    g = K.GridGeom.make(80.0, 40.0, 0.2)
    s = np.linspace(0.0, 40.0, 21)
    sx = K.spline_fit(s, s - 30.0); sy = K.spline_fit(s, 0.0 * s)                     # the x axis from x = -30
    free = np.full((g.rows, g.cols), 20.0, dtype=np.float32)
    r = K.graph_search_dp(sx, sy, 30.0, (-30.0, 0.0, 0.0), free, g)
    # layers every 1.5 m from the projection of the vehicle (s = 0) plus the end; the vehicle starts on sample 16 (l = -0.4) and
    # stays there: a lateral move of 0.6 m over 1.5 m costs 16 * atan(0.4) / (pi/2) = 3.9, the offset only 0.04
    assert len(r["layers_s"]) == 21 and r["layers_s"][1] == 1.5 and r["layers_s"][-1] == 30.0
    assert r["vehicle_l"] == pytest.approx(0.0, abs=1e-12)
    assert (r["lb"][0], r["ub"][0]) == (-10.0, 10.0)
    # everything is free: the rough bounds are the whole +-10 m range, already beyond the 6 m refinement limit
    assert np.allclose(r["ub"][1:], 0.2 + (-10.0 + 33 * 0.6)) and np.allclose(r["lb"][1:], -0.2 - 10.0)
    # a disc of radius 1 at (-15, 0.5): samples closer than 1.2 m to it are infeasible, the corridor passes on the lower side
    d = free.copy()
    for i in range(g.rows):
        for j in range(g.cols):
            x, y = K.grid_cell_position(g, i, j)
            d[i, j] = max(math.hypot(x + 15.0, y - 0.5) - 1.0, 0.0)
    r2 = K.graph_search_dp(sx, sy, 30.0, (-30.0, 0.0, 0.0), d, g)
    k = int(np.argmin(np.abs(r2["layers_s"] - 15.0)))
    assert r2["ub"][k] < -1.0 and r2["lb"][k] < r2["ub"][k]                            # squeezed below the obstacle
    assert len(r2["layers_s"]) == 21
    assert K.graph_search_dp(sx, sy, 30.0, (-30.0, 12.0, 0.0), free, g) is None        # vehicle more than 10 m off the line


@pytest.mark.parametrize("name", ["scene_a", "scene_b"])
def test_scene_fixtures_are_reproduced(name):
    """tests/golden/scene_*.npz (made by tests/golden/make_golden.py) pin the restatement against silent changes."""
    f = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sx = K.spline_fit(f["knots_s"], f["knots_x"]); sy = K.spline_fit(f["knots_s"], f["knots_y"])
    tab, ext = K.pack_spline(sx, sy)
    assert np.array_equal(tab, f["spline"]) and np.array_equal(ext, f["spline_ext"])
    gv = f["geom"]
    g = K.GridGeom(int(gv[0]), int(gv[1]), *[float(v) for v in gv[2:]])
    ref = K.build_reference_from_spline(sx, sy, float(f["length"]))
    assert np.array_equal(ref, f["ref"])
    bounds, n_valid, _ = K.update_bounds_improved(ref, sx, sy, f["dist"], g)
    assert n_valid == int(f["n_valid"]) and np.array_equal(bounds, f["bounds"])
    dp = K.graph_search_dp(sx, sy, float(f["length"]), tuple(f["start"]), f["dist"], g)
    assert np.array_equal(dp["layers_s"], f["dp_layers_s"]) and np.array_equal(dp["lb"], f["dp_lb"]) and np.array_equal(dp["ub"], f["dp_ub"])


def test_line_fixture_is_reproduced():
    """tests/golden/line_a.npz pins the head of ReferencePathSmoother::solve (bSpline -> spline -> segmentRawReference), the tail
    of postSmooth and setReferencePathLength against silent changes of the restatement."""
    f = np.load(os.path.join(ROOT, "tests", "golden", "line_a.npz"))
    x, y, s = K.bspline_resample(f["points"])
    assert np.array_equal(x, f["raw_x"]) and np.array_equal(y, f["raw_y"]) and np.array_equal(s, f["raw_s"])
    sx, sy = K.spline_fit(s, x), K.spline_fit(s, y)
    assert np.array_equal(K.pack_spline(sx, sy)[0], f["spline"])
    seg = K.segment_raw_reference(sx, sy, float(s[-1]))
    for got, key in zip(seg, ("seg_x", "seg_y", "seg_s", "seg_angle", "seg_k")):
        assert np.array_equal(got, f[key])
    op = K.offsets_to_points(sx, sy, f["at_s"], f["offsets"])
    assert np.array_equal(op[0], f["off_x"]) and np.array_equal(op[2], f["off_s"])
    assert K.reference_length(sx, sy, float(s[-1]), f["target"][0], f["target"][1]) == float(f["cut_length"])
