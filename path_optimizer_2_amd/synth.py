"""Synthetic scenario generator G(seed, N, profile) for the batched path QP.

Plain data plumbing (numpy): produces the arrays the C-ABI consumes.  Counter-based RNG
(splitmix64 finaliser keyed by (base_seed, qp_index, field, k)) so any QP of any batch can be
regenerated independently and identically on any host.

Layouts (all float64, C-contiguous) — the ones `include/pqp.h` documents:
  ref    [batch][n][5]  s, k, heading, x, y      (ReferencePath::getReferenceStates, reference
                                                   include/data_struct/data_struct.hpp:14-26)
  bounds [batch][n][6]  front lb,ub, rear lb,ub, center lb,ub   (VehicleStateBound, :74-93)
  scal   [batch][6]     init_err_l, init_err_psi, start_k, target_heading, blocked(0/1),
                        max_steering_angle
"""
import numpy as np

BASE_SEED = 20260926
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

F_LAMBDA, F_PHI, F_C, F_A, F_PHI1, F_PHI2, F_OBS_J, F_OBS_W, F_OBS_SIDE, F_OBS_VAL, F_X0, F_ENDPSI, F_STEER = range(13)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def u01(seed, qp, field, k=0):
    """Uniform [0,1) from the counter (seed, qp, field, k); qp may be an array."""
    with np.errstate(over="ignore"):
        qp = np.asarray(qp, dtype=np.uint64)
        a = _mix(np.uint64(seed) + _GOLDEN * (qp + np.uint64(1)))
        b = _mix(a + _GOLDEN * np.uint64(field * 4096 + k + 1))
    return (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _uni(seed, qp, field, lo, hi, k=0):
    return lo + (hi - lo) * u01(seed, qp, field, k)


def make_batch(batch, n, profile="uniform", seed=BASE_SEED, first_qp=0):
    """Return dict(ref, bounds, scal) for QPs first_qp .. first_qp+batch-1.

    profile "uniform": ds = 0.3 m, a = 0.08, end_psi = 0.05, max_steer = 35 deg  (configs 2, 4)
    profile "varied" : a~U[0.02,0.15], ds by the dynamic-segmentation rule of the reference
                       (reference_path_impl.cpp:331-335, 0.15..0.3 m), end_psi~U[-0.2,0.2],
                       max_steer~U[25,40] deg                                       (config 3)
    """
    qp = np.arange(first_qp, first_qp + batch, dtype=np.uint64)
    varied = profile == "varied"
    lam = _uni(seed, qp, F_LAMBDA, 15.0, 40.0)
    phi = _uni(seed, qp, F_PHI, 0.0, 2 * np.pi)
    c = _uni(seed, qp, F_C, -0.03, 0.03)
    a = _uni(seed, qp, F_A, 0.02, 0.15) if varied else np.full(batch, 0.08)

    def kref(sv):
        if sv.ndim == 2:
            return a[:, None] * np.sin(2 * np.pi * sv / lam[:, None] + phi[:, None]) + c[:, None]
        return a * np.sin(2 * np.pi * sv / lam + phi) + c

    s = np.zeros((batch, n))
    for i in range(1, n):
        if varied:
            ak = np.abs(kref(s[:, i - 1]))
            share = np.where(ak > 0.2, 1.0, np.where(ak < 0.08, 0.0, (ak - 0.08) / (0.2 - 0.08)))
            s[:, i] = s[:, i - 1] + (0.3 - share * 0.15)
        else:
            s[:, i] = 0.3 * i
    k = kref(s)
    ds = np.diff(s, axis=1)
    heading = np.zeros((batch, n))
    heading[:, 1:] = np.cumsum(0.5 * (k[:, 1:] + k[:, :-1]) * ds, axis=1)
    x = np.zeros((batch, n))
    y = np.zeros((batch, n))
    ch, sh = np.cos(heading), np.sin(heading)
    x[:, 1:] = np.cumsum(0.5 * (ch[:, 1:] + ch[:, :-1]) * ds, axis=1)
    y[:, 1:] = np.cumsum(0.5 * (sh[:, 1:] + sh[:, :-1]) * ds, axis=1)
    ref = np.stack([s, k, heading, x, y], axis=2)

    phi1 = _uni(seed, qp, F_PHI1, 0.0, 2 * np.pi)[:, None]
    phi2 = _uni(seed, qp, F_PHI2, 0.0, 2 * np.pi)[:, None]
    ub = 2.5 + 0.8 * np.sin(s / 5.0 + phi1)
    lb = -(2.5 + 0.8 * np.sin(s / 7.0 + phi2))
    j0 = n // 4 + np.floor(u01(seed, qp, F_OBS_J) * (3 * n // 4 - n // 4 + 1)).astype(np.int64)
    w = 4 + np.floor(u01(seed, qp, F_OBS_W) * 8).astype(np.int64)
    left = u01(seed, qp, F_OBS_SIDE) < 0.5
    val = _uni(seed, qp, F_OBS_VAL, -0.3, 0.6)
    idx = np.arange(n)[None, :]
    win = (idx >= j0[:, None]) & (idx < (j0 + w)[:, None])
    ub = np.where(win & left[:, None], val[:, None], ub)
    lb = np.where(win & ~left[:, None], -val[:, None], lb)
    bounds = np.stack([lb, ub, lb, ub, lb, ub], axis=2)

    scal = np.zeros((batch, 6))
    scal[:, 0] = _uni(seed, qp, F_X0, -0.5, 0.5, 0)
    scal[:, 1] = _uni(seed, qp, F_X0, -0.15, 0.15, 1)
    scal[:, 2] = k[:, 0] + _uni(seed, qp, F_X0, -0.02, 0.02, 2)
    end_psi = _uni(seed, qp, F_ENDPSI, -0.2, 0.2) if varied else np.full(batch, 0.05)
    scal[:, 3] = heading[:, -1] + end_psi
    scal[:, 4] = 0.0
    scal[:, 5] = np.deg2rad(_uni(seed, qp, F_STEER, 25.0, 40.0)) if varied else np.full(batch, 35.0 * np.pi / 180.0)
    return dict(ref=np.ascontiguousarray(ref), bounds=np.ascontiguousarray(bounds), scal=np.ascontiguousarray(scal))


F_JITTER = 13


def jitter_batch(host, variant, rel=0.05, seed=BASE_SEED, first_qp=0):
    """The same scenarios one planning cycle later: every QP's corridor sides and its start state (init_err_l, init_err_psi, start_k) are
    scaled by independent factors 1 + rel * U[-1, 1] drawn from the counter (seed, qp, F_JITTER, 8 * variant + j).  variant 0 returns the
    batch itself.  What bench.py cycles through so that PQP_OPT_ORDER_BY_COST sees SIMILAR, not identical, batches from step to step."""
    if variant == 0:
        return host
    batch = host["ref"].shape[0]
    qp = np.arange(first_qp, first_qp + batch, dtype=np.uint64)
    f = [1.0 + rel * (2.0 * u01(seed, qp, F_JITTER, 8 * variant + j) - 1.0) for j in range(5)]
    bounds = host["bounds"].copy()
    bounds[:, :, 0::2] *= f[0][:, None, None]       # the three lower bounds (right side)
    bounds[:, :, 1::2] *= f[1][:, None, None]       # the three upper bounds (left side)
    scal = host["scal"].copy()
    for j in range(3):
        scal[:, j] *= f[2 + j]
    return dict(ref=host["ref"], bounds=np.ascontiguousarray(bounds), scal=np.ascontiguousarray(scal))


# ----------------------------------------------------------------------------------------------------------------------
# synthetic scenes for the corridor-bounds step (obstacle distance map + reference line), deterministic in (seed)
# ----------------------------------------------------------------------------------------------------------------------
def make_scene(seed=0, n=80, spacing=0.6, resolution=0.2, length=(70.0, 40.0), n_obstacles=60, knots_every=3.0):
    """One planning scene: a float32 obstacle-distance layer dist[rows][cols] on a grid_map-style grid (cell (0,0) at the
    +x/+y corner, indices grow towards -x/-y), a smooth reference line through it as two knot lists (s, x, y) for
    tk::spline::set_points, and n reference states (s, k, heading, x, y) sampled every `spacing` metres.
    Obstacles are discs scattered off the line, some close enough to squeeze the corridor."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    rows, cols = int(round(length[0] / resolution)), int(round(length[1] / resolution))
    lx, ly = rows * resolution, cols * resolution
    # reference line: heading wanders slowly; starts near the -x edge
    total = (n - 1) * spacing + 8.0
    ks = np.arange(0.0, total + knots_every, knots_every)
    curv = 0.04 * np.sin(ks / 9.0 + rng.uniform(0, 6.28)) + rng.normal(scale=0.01, size=ks.size)
    head = np.cumsum(curv * knots_every) + rng.uniform(-0.3, 0.3)
    kx = -0.5 * lx + 6.0 + np.concatenate([[0.0], np.cumsum(np.cos(head[:-1]) * knots_every)])
    ky = rng.uniform(-4.0, 4.0) + np.concatenate([[0.0], np.cumsum(np.sin(head[:-1]) * knots_every)])
    # obstacle mask in cell coordinates; cell (i, j) centre = (lx/2 - res/2 - res i, ly/2 - res/2 - res j)
    cx = 0.5 * lx - 0.5 * resolution - resolution * np.arange(rows)
    cy = 0.5 * ly - 0.5 * resolution - resolution * np.arange(cols)
    free = np.ones((rows, cols), dtype=bool)
    for _ in range(n_obstacles):
        k = rng.integers(0, ks.size)
        side = rng.choice([-1.0, 1.0])
        off = rng.uniform(2.2, 9.0)
        ox = kx[k] - side * off * np.sin(head[k]) + rng.normal(scale=0.5)
        oy = ky[k] + side * off * np.cos(head[k]) + rng.normal(scale=0.5)
        rad = rng.uniform(0.3, 1.2)
        free &= ((cx[:, None] - ox) ** 2 + (cy[None, :] - oy) ** 2) > rad * rad
    dist = (ndimage.distance_transform_edt(free) * resolution).astype(np.float32)
    return dict(dist=dist, rows=rows, cols=cols, resolution=resolution, length=(lx, ly), pos=(0.0, 0.0), knots_s=ks, knots_x=kx, knots_y=ky,
                n=n, spacing=spacing)
