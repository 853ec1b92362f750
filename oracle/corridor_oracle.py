"""corridor_oracle.py — CPU restatement of the step that produces the `bounds` input of every path QP.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench tools' cpu_baseline legs may import this module.

PARITY UNPINNED.  The reference has no tests or vectors for this path, and two of its ingredients are third-party:
  * grid_map_core (ANYbotics/grid_map, un-pinned `find_package(grid_map_core)`; not in /root/reference, not in this image).
    `GridMap::isInside`, `getIndex`, `getPosition` and `atPosition(..., INTER_LINEAR)` are restated below from the published
    1.6.x sources (GridMapMath.cpp, GridMap.cpp::atPositionLinearInterpolated), including their border behaviour.
  * libm's sin/cos/atan2 (the GPU uses ocml's): bounds agree except where a clearance sample lies within round-off of
    the 0.5 m threshold.
tk::spline IS in the reference tree (src/tools/spline.cpp, std-only): oracle/Makefile builds it from where it lies into
oracle/_ref/libref_spline.so and tests/test_corridor_oracle.py checks the restatement below against it bit for bit.

What is restated (reference file:line):
  spline_fit / spline_eval / spline_deriv      tk::spline::set_points / operator() / deriv       src/tools/spline.cpp:69-318
  directional_projection_newton                getDirectionalProjectionByNewton                  src/tools/tools.cpp:156-189
  global2local_y                               global2Local(...).y                               src/tools/tools.cpp:57-64
  obstacle_distance                            Map::getObstacleDistance                          src/tools/Map.cpp:16-22
  build_reference_from_spline                  ReferencePathImpl::buildReferenceFromSpline       src/data_struct/reference_path_impl.cpp:314-338
  process_init_state                           PathOptimizer::processInitState                   src/path_optimizer.cpp:73-85
  projection / projection_newton               getProjection / getProjectionByNewton             src/tools/tools.cpp:66-126
  graph_search_dp                              ReferencePathSmoother::graphSearchDp (+ calculateCostAt)  src/reference_path_smoother/reference_path_smoother.cpp:107-295
  clearance_strict                             ReferencePathImpl::getClearanceWithDirectionStrict src/data_struct/reference_path_impl.cpp:232-312
  update_bounds_improved                       ReferencePathImpl::updateBoundsImproved           src/data_struct/reference_path_impl.cpp:177-230
Python floats are IEEE doubles and every operation below is written in the reference's order, so the arithmetic is the
reference's (gcc does not contract to FMA on baseline x86-64).
"""
import bisect
import math
from dataclasses import dataclass

import numpy as np


# ----------------------------------------------------------------------------------------------------------------------
# parameters (src/config/planning_flags.cpp)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class CorridorParams:
    front_length: float = 3.9        # :20
    rear_length: float = -1.0        # :18
    car_width: float = 2.0           # :10
    safety_margin: float = 0.3       # :14
    epsilon: float = 1e-6            # :108  isEqual tolerance
    search_radius: float = 0.5       # reference_path_impl.cpp:241 (static local)
    delta_s: float = 0.3             # :238
    smaller_ds: float = 0.05         # :277
    search_range: float = 6.0        # :243
    min_space: float = 0.2           # :304
    projection_window: float = 5.0   # :194  max_s = s + 5.0


def constrain_angle(a):              # include/tools/tools.hpp:24-35
    while a > math.pi:
        a -= 2 * math.pi
    while a < -math.pi:
        a += 2 * math.pi
    return a


# ----------------------------------------------------------------------------------------------------------------------
# tk::spline (natural cubic spline; the reference never calls set_boundary)
# ----------------------------------------------------------------------------------------------------------------------
def spline_fit(x, y):
    """set_points (spline.cpp:161-249) with band_matrix::lu_solve (:69-148): tridiagonal, rows pre-scaled to unit diagonal."""
    x = [float(v) for v in x]
    y = [float(v) for v in y]
    n = len(x)
    assert n > 2 and all(x[i] < x[i + 1] for i in range(n - 1))
    lo = [0.0] * n     # A(i, i-1)
    di = [0.0] * n     # A(i, i)
    up = [0.0] * n     # A(i, i+1)
    rhs = [0.0] * n
    for i in range(1, n - 1):
        lo[i] = 1.0 / 3.0 * (x[i] - x[i - 1])
        di[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1])
        up[i] = 1.0 / 3.0 * (x[i + 1] - x[i])
        rhs[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1])
    di[0] = 2.0; up[0] = 0.0; rhs[0] = 0.0                  # second_deriv = 0 on the left
    di[n - 1] = 2.0; lo[n - 1] = 0.0; rhs[n - 1] = 0.0      # ... and on the right
    # lu_decompose: normalise every row by its diagonal (saved), then Gauss elimination
    saved = [0.0] * n
    for i in range(n):
        saved[i] = 1.0 / di[i]
        if i > 0:
            lo[i] *= saved[i]
        if i < n - 1:
            up[i] *= saved[i]
        di[i] = 1.0
    for k in range(n - 1):
        i = k + 1
        xm = -lo[i] / di[k]
        lo[i] = -xm
        di[i] = di[i] + xm * up[k]
    # l_solve
    yv = [0.0] * n
    for i in range(n):
        s = 0.0
        if i > 0:
            s += lo[i] * yv[i - 1]
        yv[i] = (rhs[i] * saved[i]) - s
    # r_solve
    b = [0.0] * n
    for i in range(n - 1, -1, -1):
        s = 0.0
        if i < n - 1:
            s += up[i] * b[i + 1]
        b[i] = (yv[i] - s) / di[i]
    a = [0.0] * n
    c = [0.0] * n
    for i in range(n - 1):
        a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (x[i + 1] - x[i])
        c[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (x[i + 1] - x[i])
    b0, c0 = b[0], c[0]
    h = x[n - 1] - x[n - 2]
    a[n - 1] = 0.0
    c[n - 1] = 3.0 * a[n - 2] * h * h + 2.0 * b[n - 2] * h + c[n - 2]
    return dict(x=x, y=y, a=a, b=b, c=c, b0=b0, c0=c0)


def _segment(sp, x):
    # std::lower_bound: first knot >= x; idx = max(that - 1, 0)
    return max(bisect.bisect_left(sp["x"], x) - 1, 0)


def spline_eval(sp, x):              # operator()  spline.cpp:251-272
    n = len(sp["x"])
    idx = _segment(sp, x)
    h = x - sp["x"][idx]
    if x < sp["x"][0]:
        return (sp["b0"] * h + sp["c0"]) * h + sp["y"][0]
    if x > sp["x"][n - 1]:
        return (sp["b"][n - 1] * h + sp["c"][n - 1]) * h + sp["y"][n - 1]
    return ((sp["a"][idx] * h + sp["b"][idx]) * h + sp["c"][idx]) * h + sp["y"][idx]


def spline_deriv(sp, order, x):      # deriv  spline.cpp:274-318 (the left-extrapolated 2nd derivative is the reference's)
    n = len(sp["x"])
    idx = _segment(sp, x)
    h = x - sp["x"][idx]
    if x < sp["x"][0]:
        return {1: 2.0 * sp["b0"] * h + sp["c0"], 2: 2.0 * sp["b0"] * h}.get(order, 0.0)
    if x > sp["x"][n - 1]:
        return {1: 2.0 * sp["b"][n - 1] * h + sp["c"][n - 1], 2: 2.0 * sp["b"][n - 1]}.get(order, 0.0)
    if order == 1:
        return (3.0 * sp["a"][idx] * h + 2.0 * sp["b"][idx]) * h + sp["c"][idx]
    if order == 2:
        return 6.0 * sp["a"][idx] * h + 2.0 * sp["b"][idx]
    if order == 3:
        return 6.0 * sp["a"][idx]
    return 0.0


def pack_spline(sx, sy):
    """The flat layout the C ABI takes: [7][m] doubles = knots, then (y, a, b, c) of x(s), then of y(s) ... see include/pqp.h.
    Row 0: knots s; rows 1-4: y, a, b, c of x(s); rows 5-8: y, a, b, c of y(s); extra[4] = b0x, c0x, b0y, c0y."""
    m = len(sx["x"])
    tab = np.zeros((9, m))
    tab[0] = sx["x"]
    tab[1], tab[2], tab[3], tab[4] = sx["y"], sx["a"], sx["b"], sx["c"]
    tab[5], tab[6], tab[7], tab[8] = sy["y"], sy["a"], sy["b"], sy["c"]
    return tab, np.array([sx["b0"], sx["c0"], sy["b0"], sy["c0"]])


# ----------------------------------------------------------------------------------------------------------------------
# tools.cpp
# ----------------------------------------------------------------------------------------------------------------------
def directional_projection_newton(sx, sy, tx, ty, angle, max_s, hint_s):      # tools.cpp:156-189
    hint_s = min(hint_s, max_s)
    cur_s = hint_s
    prev_s = hint_s
    v1 = math.sin(angle)
    v2 = -math.cos(angle)
    for _ in range(20):
        x = spline_eval(sx, cur_s)
        y = spline_eval(sy, cur_s)
        dx = spline_deriv(sx, 1, cur_s)
        dy = spline_deriv(sy, 1, cur_s)
        ddx = spline_deriv(sx, 2, cur_s)
        ddy = spline_deriv(sy, 2, cur_s)
        p1 = v1 * (x - tx) + v2 * (y - ty)
        p2 = v1 * dx + v2 * dy
        j = p1 * p2
        h = p1 * (v1 * ddx + v2 * ddy) + p2 * p2
        cur_s -= j / h
        if abs(cur_s - prev_s) < 1e-5:
            break
        prev_s = cur_s
    cur_s = min(cur_s, max_s)
    return spline_eval(sx, cur_s), spline_eval(sy, cur_s), cur_s      # heading is overwritten by the caller (:207)


def global2local_y(rx, ry, rheading, tx, ty):                                 # tools.cpp:57-64, .y only
    dx = tx - rx
    dy = ty - ry
    return -dx * math.sin(rheading) + dy * math.cos(rheading)


def build_reference_from_spline(sx, sy, max_s, ds_small=0.15, ds_large=0.3, dynamic=True):
    """ReferencePathImpl::buildReferenceFromSpline (reference_path_impl.cpp:314-338), called with (output_spacing / 2,
    output_spacing) = (0.15, 0.3) at path_optimizer.cpp:119.  Rows (s, k, heading, x, y): the ABI layout of the path QP."""
    large_k, small_k = 0.2, 0.08
    out = []
    tmp_s = 0.0
    while tmp_s <= max_s:
        x = spline_eval(sx, tmp_s)
        y = spline_eval(sy, tmp_s)
        dx, dy = spline_deriv(sx, 1, tmp_s), spline_deriv(sy, 1, tmp_s)
        ddx, ddy = spline_deriv(sx, 2, tmp_s), spline_deriv(sy, 2, tmp_s)
        h = math.atan2(dy, dx)                                                            # getHeading   tools.cpp:32-36
        k = (dx * ddy - dy * ddx) / math.pow(math.pow(dx, 2) + math.pow(dy, 2), 1.5)      # getCurvature tools.cpp:38-44
        out.append((tmp_s, k, h, x, y))
        if dynamic:
            ak = abs(k)
            k_share = 1 if ak > large_k else (0 if ak < small_k else (ak - small_k) / (large_k - small_k))
            tmp_s += ds_large - k_share * (ds_large - ds_small)
        else:
            tmp_s += ds_large
    return np.array(out).reshape(-1, 5)


# ----------------------------------------------------------------------------------------------------------------------
# ReferencePathSmoother::bSpline (reference_path_smoother.cpp:490-521) over tinyspline
# PARITY UNPINNED: tinyspline (the un-vendored, un-pinned ROS package qutas/tinyspline_ros, CMakeLists.txt:12, package.xml:56) is
# not in /root/reference and not in this image.  What is restated here is its published behaviour for the one call pattern the
# reference uses - tinyspline::BSpline(n_control_points, 2, degree) [type TS_CLAMPED], setControlPoints, eval(u).result():
#   knots of a clamped spline on [0, 1]: `order` zeros, then (i - degree) / (n_knots - 2 degree - 1) for the interior ones, then
#   `order` ones (ts_bspline_new / fill_knots); eval = de Boor's algorithm on the knot span containing u; u = 0 and u = 1 return the
#   first and the last control point.  tinyspline treats u within 1e-4 of a knot as lying on it, which changes the result by
#   O(1e-4 ^ degree) only (the pieces join with degree - 1 continuous derivatives) and is not reproduced.
# ----------------------------------------------------------------------------------------------------------------------
def clamped_knots(n_ctrl, degree):
    order = degree + 1
    n_knots = n_ctrl + order
    fac = 1.0 / (n_knots - 2 * degree - 1)
    return np.array([0.0] * order + [(i - degree) * fac for i in range(order, n_knots - order)] + [1.0] * order)


def bspline_eval(ctrl, knots, degree, u):
    """de Boor: the point of the B-spline with control points ctrl [n][2] at parameter u in [0, 1]."""
    n = len(ctrl)
    if u <= knots[0]:
        return np.array(ctrl[0], dtype=float)
    if u >= knots[-1]:
        return np.array(ctrl[n - 1], dtype=float)
    k = degree
    while knots[k + 1] <= u:               # knots[k] <= u < knots[k + 1], degree <= k <= n - 1
        k += 1
    d = [np.array(ctrl[j + k - degree], dtype=float) for j in range(degree + 1)]
    for r in range(1, degree + 1):
        for j in range(degree, r - 1, -1):
            i = j + k - degree
            a = (u - knots[i]) / (knots[i + degree - r + 1] - knots[i])
            d[j] = (1.0 - a) * d[j - 1] + a * d[j]
    return d[degree]


def bspline_resample(points):
    """ReferencePathSmoother::bSpline (reference_path_smoother.cpp:490-521): the input points are the control points of a clamped
    B-spline whose degree depends on their average spacing (> 10 m: 3, > 5 m: 4, else 5); it is sampled at t = 0, 1/length,
    2/length, ... while t < 1 (t accumulated as written) and at t = 1; s is the accumulated chord length.  Returns x, y, s."""
    pts = np.asarray(points, dtype=float)
    length = 0.0
    for i in range(len(pts) - 1):
        length += math.sqrt(math.pow(pts[i, 0] - pts[i + 1, 0], 2) + math.pow(pts[i, 1] - pts[i + 1, 1], 2))     # tools.hpp distance()
    average_length = length / (len(pts) - 1)
    degree = 3 if average_length > 10 else (4 if average_length > 5 else 5)
    knots = clamped_knots(len(pts), degree)
    delta_t = 1.0 / length
    xs, ys = [], []
    tmp_t = 0.0
    while tmp_t < 1:
        p = bspline_eval(pts, knots, degree, tmp_t)
        xs.append(p[0]); ys.append(p[1])
        tmp_t += delta_t
    p = bspline_eval(pts, knots, degree, 1.0)
    xs.append(p[0]); ys.append(p[1])
    s = [0.0]
    for i in range(1, len(xs)):
        s.append(s[-1] + math.sqrt(math.pow(xs[i] - xs[i - 1], 2) + math.pow(ys[i] - ys[i - 1], 2)))
    return np.array(xs), np.array(ys), np.array(s)


def segment_raw_reference(sx, sy, max_s, delta_s=1.0):
    """ReferencePathSmoother::segmentRawReference (reference_path_smoother.cpp:48-85): the raw point list's splines sampled
    every delta_s = 1.0 for the smoother QPs.  The loop (:64-67) pushes back + delta_s while back < max_s, so the last abscissa is
    the first one >= max_s - beyond the line unless max_s is a multiple of delta_s - and the test of :68-70 can never hold
    after it; both kept as written.  Returns x, y, s, angle, k lists (the argument order of osqpSmooth)."""
    s_list = [0.0]
    while s_list[-1] < max_s:
        s_list.append(s_list[-1] + delta_s)
    if max_s - s_list[-1] > 1:
        s_list.append(max_s)
    x, y, angle, k = [], [], [], []
    for s in s_list:
        dx, dy = spline_deriv(sx, 1, s), spline_deriv(sy, 1, s)
        ddx, ddy = spline_deriv(sx, 2, s), spline_deriv(sy, 2, s)
        angle.append(math.atan2(dy, dx))
        k.append((dx * ddy - dy * ddx) / math.pow(dx * dx + dy * dy, 1.5))
        x.append(spline_eval(sx, s))
        y.append(spline_eval(sy, s))
    return np.array(x), np.array(y), np.array(s_list), np.array(angle), np.array(k)


def process_init_state(sx, sy, start_x, start_y, start_heading):
    """PathOptimizer::processInitState (path_optimizer.cpp:73-85) -> (initial_offset, initial_heading_error)."""
    ix, iy = spline_eval(sx, 0.0), spline_eval(sy, 0.0)
    ih = math.atan2(spline_deriv(sy, 1, 0.0), spline_deriv(sx, 1, 0.0))
    dx, dy = ix - start_x, iy - start_y
    local_y = -dx * math.sin(start_heading) + dy * math.cos(start_heading)                # global2Local(start, init_point).y
    min_distance = math.sqrt(math.pow(start_x - ix, 2) + math.pow(start_y - iy, 2))       # distance()  tools.cpp:46-48
    offset = min_distance if local_y < 0.0 else -min_distance
    return offset, constrain_angle(start_heading - ih)


# ----------------------------------------------------------------------------------------------------------------------
# grid_map_core 1.6.x (restated; see the header)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class GridGeom:
    rows: int              # cells along x  (getSize()(0))
    cols: int              # cells along y  (getSize()(1))
    resolution: float
    length_x: float
    length_y: float
    pos_x: float           # map centre
    pos_y: float

    @staticmethod
    def make(length_x, length_y, resolution, pos=(0.0, 0.0)):
        # GridMap::setGeometry: size = round(length / resolution), length = size * resolution
        rows, cols = int(round(length_x / resolution)), int(round(length_y / resolution))
        return GridGeom(rows, cols, resolution, rows * resolution, cols * resolution, float(pos[0]), float(pos[1]))


def grid_is_inside(g, px, py):       # checkIfPositionWithinMap: -(p - c - L/2) in [0, L)
    tx = -(px - g.pos_x - 0.5 * g.length_x)
    ty = -(py - g.pos_y - 0.5 * g.length_y)
    return tx >= 0.0 and ty >= 0.0 and tx < g.length_x and ty < g.length_y


def grid_index(g, px, py):           # getIndexFromPosition: trunc toward zero of (p - L/2 - c) / res, negated
    vx = (px - 0.5 * g.length_x - g.pos_x) / g.resolution
    vy = (py - 0.5 * g.length_y - g.pos_y) / g.resolution
    return -int(vx), -int(vy)        # Python int() truncates toward zero like the C++ double -> int conversion


def grid_cell_position(g, ix, iy):   # getPositionFromIndex: c + (L/2 - res/2) + res * (-idx)
    ox = 0.5 * g.length_x - 0.5 * g.resolution
    oy = 0.5 * g.length_y - 0.5 * g.resolution
    return (g.pos_x + ox) + g.resolution * float(-ix), (g.pos_y + oy) + g.resolution * float(-iy)


def _lin(g, ix, iy):                 # getLinearIndexFromIndex (column major), as size_t: negative values wrap to huge ones
    v = iy * g.rows + ix
    return v if v >= 0 else (1 << 64) + v


def grid_at_linear(dist, g, px, py):
    """GridMap::atPositionLinearInterpolated; `dist` is the layer as a [rows][cols] float32 array. Returns None on failure."""
    i0 = grid_index(g, px, py)
    cx, cy = grid_cell_position(g, *i0)
    idx = [i0, None, None, None]
    if px >= cx:
        idx[1] = (i0[0] - 1, i0[1]); tmp_dir = True
    else:
        idx[1] = (i0[0] + 1, i0[1]); tmp_dir = False
    if py >= cy:
        idx[2] = (i0[0], i0[1] - 1)
        shift = (0, 1, 2, 3) if tmp_dir else (1, 0, 3, 2)
    else:
        idx[2] = (i0[0], i0[1] + 1)
        shift = (2, 3, 0, 1) if tmp_dir else (3, 2, 1, 0)
    idx[3] = (idx[1][0], idx[2][1])
    end_lin = g.rows * g.cols          # startIndexLin = 0 for a map that was never moved
    f = []
    for k in range(4):
        ix, iy = idx[shift[k]]
        lin = _lin(g, ix, iy)
        if lin > end_lin:              # (the upstream test is `>`; lin == end_lin would read one past the buffer)
            return None
        if lin == end_lin:
            return None                # guard the restatement against the out-of-bounds read; unreachable for inside points
        # the quirk: a neighbour with ix = -1 and iy >= 1 passes the linear test and reads the element the wrapped linear
        # index points at
        f.append(float(dist.reshape(-1, order="F")[lin]))
    qx, qy = grid_cell_position(g, *idx[shift[0]])
    rx = (px - qx) / g.resolution
    ry = (py - qy) / g.resolution
    fx = 1.0 - rx
    fy = 1.0 - ry
    v = f[0] * fx * fy + f[1] * rx * fy + f[2] * fx * ry + f[3] * rx * ry
    return float(np.float32(v))


def obstacle_distance(dist, g, px, py):          # Map::getObstacleDistance (Map.cpp:16-22)
    if not grid_is_inside(g, px, py):
        return 0.0
    v = grid_at_linear(dist, g, px, py)
    if v is not None:
        return v
    ix, iy = grid_index(g, px, py)               # INTER_NEAREST fallback of GridMap::atPosition
    return float(dist[ix, iy])


# ----------------------------------------------------------------------------------------------------------------------
# reference_path_impl.cpp
# ----------------------------------------------------------------------------------------------------------------------
def clearance_strict(x, y, heading, dist, g, prm=CorridorParams()):          # :232-312 -> (left_bound, right_bound)
    delta_s = prm.delta_s
    left_angle = constrain_angle(heading + math.pi / 2)
    right_angle = constrain_angle(heading - math.pi / 2)
    n = int(prm.search_range / delta_s)
    if not (obstacle_distance(dist, g, x, y) > prm.search_radius):
        return 0.0, 0.0
    right_s = 0.0
    for _ in range(n):
        right_s += delta_s
        if obstacle_distance(dist, g, x + right_s * math.cos(right_angle), y + right_s * math.sin(right_angle)) < prm.search_radius:
            break
    left_s = 0.0
    for _ in range(n):
        left_s += delta_s
        if obstacle_distance(dist, g, x + left_s * math.cos(left_angle), y + left_s * math.sin(left_angle)) < prm.search_radius:
            break
    right_bound = -(right_s - delta_s)
    left_bound = left_s - delta_s
    smaller_ds = prm.smaller_ds
    steps = int(delta_s / smaller_ds)          # static_cast<int>(0.3 / 0.05) = 5: four fine steps
    for _ in range(1, steps):
        left_bound += smaller_ds
        if obstacle_distance(dist, g, x + left_bound * math.cos(left_angle), y + left_bound * math.sin(left_angle)) < prm.search_radius:
            left_bound -= smaller_ds
            break
    for _ in range(1, steps):
        right_bound -= smaller_ds
        # as written in the reference (:291-294): right_bound is negative, so this probes the LEFT side of the state
        if obstacle_distance(dist, g, x + right_bound * math.cos(right_angle), y + right_bound * math.sin(right_angle)) < prm.search_radius:
            right_bound += smaller_ds
            break
    diff_radius = prm.car_width * 0.5 - prm.search_radius
    left_bound -= diff_radius
    right_bound += diff_radius
    if left_bound < right_bound:
        return 0.0, 0.0
    space = left_bound - right_bound
    max_safety_margin = max(0.0, (space - prm.min_space) / 2.0)
    safety_margin = min(prm.safety_margin, max_safety_margin)
    left_bound -= safety_margin
    right_bound += safety_margin
    return left_bound, right_bound


def update_bounds_improved(ref, sx, sy, dist, g, prm=CorridorParams()):
    """ref [n][5] = (s, k, heading, x, y) per waypoint (the ABI layout of the path QP).  Returns (bounds [n_valid][6] in the
    ABI order f_lb f_ub r_lb r_ub c_lb c_ub, n_valid, blocked row or None) — :177-230: the loop stops at the first
    waypoint whose front or rear interval is empty, and the reference path is cut there."""
    out = []
    blocked = None
    for i in range(len(ref)):
        s, _, heading, x, y = (float(v) for v in ref[i])
        fcx = x + prm.front_length * math.cos(heading)
        fcy = y + prm.front_length * math.sin(heading)
        rcx = x + prm.rear_length * math.cos(heading)
        rcy = y + prm.rear_length * math.sin(heading)
        fpx, fpy, _ = directional_projection_newton(sx, sy, fcx, fcy, heading + math.pi / 2, s + prm.projection_window, s + prm.front_length)
        rpx, rpy, _ = directional_projection_newton(sx, sy, rcx, rcy, heading + math.pi / 2, s + prm.projection_window, s + prm.rear_length)
        f_ub, f_lb = clearance_strict(fpx, fpy, heading, dist, g, prm)
        off = global2local_y(fcx, fcy, heading, fpx, fpy)
        f_ub += off; f_lb += off
        r_ub, r_lb = clearance_strict(rpx, rpy, heading, dist, g, prm)
        off = global2local_y(rcx, rcy, heading, rpx, rpy)
        r_ub += off; r_lb += off
        c_ub, c_lb = clearance_strict(x, y, heading, dist, g, prm)
        row = [f_lb, f_ub, r_lb, r_ub, c_lb, c_ub]
        if abs(f_ub - f_lb) < prm.epsilon or abs(r_ub - r_lb) < prm.epsilon:
            blocked = row
            break
        out.append(row)
    return np.array(out).reshape(-1, 6), len(out), blocked


# ----------------------------------------------------------------------------------------------------------------------
# layered DP corridor search between the smoother QP and the postSmooth QP
# ----------------------------------------------------------------------------------------------------------------------
DBL_MAX = 1.7976931348623157e308


@dataclass
class DpParams:
    lateral_range: float = 10.0          # FLAGS_search_lateral_range         planning_flags.cpp:38
    longitudinal_spacing: float = 1.5    # FLAGS_search_longitudial_spacing   :40
    lateral_spacing: float = 0.6         # FLAGS_search_lateral_spacing       :42
    car_width: float = 2.0               # :10  -> search_threshold = car_width / 2 + 0.2   reference_path_smoother.cpp:171


def projection_newton(sx, sy, tx, ty, max_s, hint_s):                        # tools.cpp:98-126
    hint_s = min(hint_s, max_s)
    cur_s = hint_s
    prev_s = hint_s
    for _ in range(20):
        x, y = spline_eval(sx, cur_s), spline_eval(sy, cur_s)
        dx, dy = spline_deriv(sx, 1, cur_s), spline_deriv(sy, 1, cur_s)
        ddx, ddy = spline_deriv(sx, 2, cur_s), spline_deriv(sy, 2, cur_s)
        j = (x - tx) * dx + (y - ty) * dy
        h = dx * dx + (x - tx) * ddx + dy * dy + (y - ty) * ddy
        cur_s -= j / h
        if abs(cur_s - prev_s) < 1e-5:
            break
        prev_s = cur_s
    return min(cur_s, max_s)


def projection(sx, sy, tx, ty, max_s, start_s=0.0):                           # getProjection tools.cpp:66-96 -> s
    if max_s <= start_s:
        return 0.0                                # State{xs(start_s), ys(start_s)}: s stays 0
    tmp_s, min_dis_s, min_dis = start_s, start_s, DBL_MAX
    while tmp_s <= max_s:
        d = math.sqrt(math.pow(spline_eval(sx, tmp_s) - tx, 2) + math.pow(spline_eval(sy, tmp_s) - ty, 2))
        if d < min_dis:
            min_dis, min_dis_s = d, tmp_s
        tmp_s += 1.0
    d_end = math.sqrt(math.pow(spline_eval(sx, max_s) - tx, 2) + math.pow(spline_eval(sy, max_s) - ty, 2))
    if d_end < min_dis:
        return max_s
    return projection_newton(sx, sy, tx, ty, max_s, min_dis_s)


def offsets_to_points(sx, sy, at_s, l):
    """The tail of ReferencePathSmoother::postSmooth (reference_path_smoother.cpp:559-573): offsets l_i at abscissae s_i of the
    smoothed line -> x_list, y_list and the accumulated chord length s_list."""
    xs, ys, ss = [], [], []
    s = 0.0
    for i in range(len(at_s)):
        ref_s = at_s[i]
        ref_dir = math.atan2(spline_deriv(sy, 1, ref_s), spline_deriv(sx, 1, ref_s))           # getHeading tools.cpp:32-36
        xs.append(spline_eval(sx, ref_s) + l[i] * math.cos(ref_dir + math.pi / 2))
        ys.append(spline_eval(sy, ref_s) + l[i] * math.sin(ref_dir + math.pi / 2))
        if i > 0:
            s += math.sqrt(math.pow(xs[i] - xs[i - 1], 2) + math.pow(ys[i] - ys[i - 1], 2))
        ss.append(s)
    return np.array(xs), np.array(ys), np.array(ss)


def reference_length(sx, sy, length, tx, ty):
    """PathOptimizer::setReferencePathLength (path_optimizer.cpp:87-104): the line's length, or the abscissa of the target state's
    projection when the target lies behind the line's end (x <= 0 in the end state's frame)."""
    ex, ey = spline_eval(sx, length), spline_eval(sy, length)
    eh = math.atan2(spline_deriv(sy, 1, length), spline_deriv(sx, 1, length))
    local_x = (tx - ex) * math.cos(eh) + (ty - ey) * math.sin(eh)              # global2Local(...).x  tools.cpp:57-64
    if local_x > 0.0:
        return length
    return projection(sx, sy, tx, ty, length, 0.0)


def graph_search_dp(sx, sy, length, start, dist, g, prm=DpParams()):
    """graphSearchDp (reference_path_smoother.cpp:142-295) with calculateCostAt (:107-140).
    start = (x, y, heading) of the vehicle.  Returns None when the reference returns false, else a dict with
    layers_s, lb, ub (layers_bounds_), vehicle_l (vehicle_l_wrt_smoothed_ref_): the inputs of the postSmooth QP."""
    thr = prm.car_width / 2.0 + 0.2
    tmp_s = projection(sx, sy, start[0], start[1], length)
    layers = []
    search_ds = prm.longitudinal_spacing if length > 6 else 0.5
    while tmp_s < length:
        layers.append(tmp_s)
        tmp_s += search_ds
    layers.append(length)
    vs = layers[0]
    px, py = spline_eval(sx, vs), spline_eval(sy, vs)
    ph = math.atan2(spline_deriv(sy, 1, vs), spline_deriv(sx, 1, vs))
    vehicle_l = global2local_y(px, py, ph, start[0], start[1])
    if abs(vehicle_l) > prm.lateral_range:
        return None
    start_idx = int((prm.lateral_range + vehicle_l) / prm.lateral_spacing)
    samples = []
    for i, cur_s in enumerate(layers):
        rx, ry = spline_eval(sx, cur_s), spline_eval(sy, cur_s)
        dx, dy = spline_deriv(sx, 1, cur_s), spline_deriv(sy, 1, cur_s)
        ddx, ddy = spline_deriv(sx, 2, cur_s), spline_deriv(sy, 2, cur_s)
        rh = math.atan2(dy, dx)
        rk = (dx * ddy - dy * ddx) / math.pow(math.pow(dx, 2) + math.pow(dy, 2), 1.5)
        rr = math.inf if rk == 0.0 else 1 / rk
        pts = []
        cur_l = -prm.lateral_range
        j = 0
        while cur_l <= prm.lateral_range:
            x = rx + cur_l * math.cos(rh + math.pi / 2)
            y = ry + cur_l * math.sin(rh + math.pi / 2)
            d = obstacle_distance(dist, g, x, y) if grid_is_inside(g, x, y) else -1.0
            feas = not ((rk < 0 and cur_l < rr) or (rk > 0 and cur_l > rr) or d < thr)
            pt = dict(x=x, y=y, heading=rh, s=cur_s, l=cur_l, cost=DBL_MAX, dir=0.0, dis=d, parent=None, feas=feas, j=j, i=i)
            if i == 0:
                pt["feas"] = j == start_idx
                if j == start_idx:
                    pt["dir"], pt["cost"] = start[2], 0.0
            pts.append(pt)
            cur_l += prm.lateral_spacing
            j += 1
        for j in range(len(pts)):
            pts[j]["rlo"] = pts[j]["l"] if (j == 0 or not pts[j - 1]["feas"] or not pts[j]["feas"]) else pts[j - 1]["rlo"]
        for j in range(len(pts) - 1, -1, -1):
            pts[j]["rup"] = pts[j]["l"] if (j == len(pts) - 1 or not pts[j + 1]["feas"] or not pts[j]["feas"]) else pts[j + 1]["rup"]
        samples.append(pts)
    # costs (calculateCostAt)
    max_layer = 0
    for i, layer in enumerate(samples):
        layer_feasible = False
        for pt in layer:
            if i > 0 and pt["feas"]:
                self_cost = 0.0
                if pt["dis"] < 3.0:
                    self_cost += (3.0 - pt["dis"]) / 3.0 * 0.5
                self_cost += abs(pt["l"]) / prm.lateral_range * 1.0
                min_cost = DBL_MAX
                for pre in samples[i - 1]:
                    if not pre["feas"]:
                        continue
                    if abs(pre["l"] - pt["l"]) > (pt["s"] - pre["s"]):
                        continue
                    direction = math.atan2(pt["y"] - pre["y"], pt["x"] - pre["x"])
                    edge = abs(constrain_angle(direction - pre["dir"])) / (math.pi / 2) * 16.0 \
                        + abs(constrain_angle(direction - pt["heading"])) / (math.pi / 2) * 0.5
                    total = self_cost + edge + pre["cost"]
                    if total < min_cost:
                        min_cost = total
                        pt["parent"] = pre
                        pt["dir"] = direction
                if pt["parent"] is not None:
                    pt["cost"] = min_cost
            if pt["parent"] is not None:
                layer_feasible = True
        if i != 0 and not layer_feasible:
            break
        max_layer = i
    # retrieve
    ptr, min_cost = None, DBL_MAX
    for pt in samples[max_layer]:
        if pt["cost"] < min_cost:
            ptr, min_cost = pt, pt["cost"]
    bounds = []
    while ptr is not None:
        if ptr["i"] == 0:
            bounds.append((-10.0, 10.0))
        else:
            check_s, limit = 0.2, 6.0
            ub = check_s + ptr["rup"]
            lb = -check_s + ptr["rlo"]
            rx, ry = spline_eval(sx, ptr["s"]), spline_eval(sy, ptr["s"])
            ca, sa = math.cos(ptr["heading"] + math.pi / 2), math.sin(ptr["heading"] + math.pi / 2)
            while ub < limit:
                x, y = rx + ub * ca, ry + ub * sa
                if grid_is_inside(g, x, y) and obstacle_distance(dist, g, x, y) > thr:
                    ub += check_s
                else:
                    ub -= check_s
                    break
            while lb > -limit:
                x, y = rx + lb * ca, ry + lb * sa
                if grid_is_inside(g, x, y) and obstacle_distance(dist, g, x, y) > thr:
                    lb -= check_s
                else:
                    lb += check_s
                    break
            bounds.append((lb, ub))
        ptr = ptr["parent"]
    bounds.reverse()
    n = len(bounds)
    return dict(layers_s=np.array(layers[:n]), lb=np.array([b[0] for b in bounds]), ub=np.array([b[1] for b in bounds]), vehicle_l=vehicle_l,
                path=[None] * n, n_layers_sampled=len(layers))
