"""Corridor bounds from the distance map on the GPU (pqp_corridor_bounds) against the CPU restatement
(oracle/corridor_oracle.py).  Sample positions are computed in the reference's operation order with FMA contraction off, so
the step counts of the ray search - and with them the bounds - are the oracle's; only sin/cos differ (ocml vs libm, <= 1 ulp),
which can move a bound by one search step when a sample sits within round-off of the 0.5 m threshold."""
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
