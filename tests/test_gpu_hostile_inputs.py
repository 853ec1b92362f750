"""Scenarios that are not numbers - NaN / Inf in the reference states, bounds or start state, an arclength that does not increase (the reference
divides by ds: base_solver.cpp:174,180) - through BOTH path-QP kernels and both kinds of entry point.  The reference's OSQP would hand back
non-finite iterates and BaseSolver::solve() false (base_solver.cpp:80-88); here such a QP must end PQP_STATUS_NUMERICAL with a zero output
record, in bounded time, and - the workgroups are persistent and draw one QP after the other, the lane-per-QP kernel runs 64 QPs in lock-step -
leave every other QP of the batch bit-identical to a batch without it."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

NUMERICAL = 3
BATCH, N, BAD = 32, 80, 7


def _poison(b, what):
    if what == "nan_s": b["ref"][BAD, 10, 0] = np.nan
    elif what == "inf_k": b["ref"][BAD, 40, 1] = np.inf
    elif what == "nan_heading": b["ref"][BAD, 3, 2] = np.nan
    elif what == "nan_bound": b["bounds"][BAD, 5, 0] = np.nan
    elif what == "inf_bound": b["bounds"][BAD, 77, 3] = np.inf
    elif what == "nan_start": b["scal"][BAD, 0] = np.nan
    elif what == "inf_steer": b["scal"][BAD, 5] = np.inf
    elif what == "equal_s": b["ref"][BAD, 30, 0] = b["ref"][BAD, 29, 0]
    elif what == "decreasing_s": b["ref"][BAD, 50, 0] = b["ref"][BAD, 48, 0]
    elif what == "nan_last": b["ref"][BAD, N - 1, 3] = np.nan
    else: raise ValueError(what)


KINDS = ["nan_s", "inf_k", "nan_heading", "nan_bound", "inf_bound", "nan_start", "inf_steer", "equal_s", "decreasing_s", "nan_last"]


@pytest.fixture(scope="module")
def clean(hip_lib):
    """the same batch without the hostile QP, through both kernels"""
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    b = make_batch(BATCH, N, seed=5)
    res = {}
    for stream in (0, 1):
        h = capi.Handle(capi.production_params(), device=0, max_batch=BATCH, max_n=N)
        h.set_option(capi.OPT_STORE_WARM, 0)
        h.set_option(capi.OPT_STREAM_BATCH, 1 if stream else 0)
        res[stream] = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        assert (res[stream]["status"] == 1).all()
        h.close()
    return b, res


@pytest.mark.parametrize("stream", [0, 1], ids=["lane_per_waypoint", "lane_per_qp"])
@pytest.mark.parametrize("what", KINDS)
def test_one_hostile_qp_among_ordinary_ones(clean, what, stream):
    import torch
    from path_optimizer_2_amd import capi
    b0, ref = clean
    b = {k: v.copy() for k, v in b0.items()}
    _poison(b, what)
    others = np.arange(BATCH) != BAD
    h = capi.Handle(capi.production_params(), device=0, max_batch=BATCH, max_n=N)
    h.set_option(capi.OPT_STORE_WARM, 0)
    h.set_option(capi.OPT_STREAM_BATCH, 1 if stream else 0)
    # host-pointer entry point
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert r["status"][BAD] == NUMERICAL, (what, r["status"][BAD])
    assert np.all(r["out"][BAD] == 0.0)
    assert np.array_equal(r["status"][others], ref[stream]["status"][others])
    assert np.array_equal(r["out"][others], ref[stream]["out"][others])          # bit for bit
    # device-pointer entry point, twice on the same handle (the slot that ran the hostile QP runs ordinary ones afterwards)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_ref, d_bounds, d_scal = t(b["ref"]), t(b["bounds"]), t(b["scal"])
    for _ in range(2):
        out = torch.full((BATCH, N, 7), 123.0, dtype=torch.float64, device=dev)
        status = torch.zeros(BATCH, dtype=torch.int32, device=dev)
        h.solve_device(BATCH, N, d_ref, d_bounds, d_scal, out, passes=1, status=status)
        h.sync()
        st, o = status.cpu().numpy(), out.cpu().numpy()
        assert st[BAD] == NUMERICAL and np.all(o[BAD] == 0.0)
        assert np.array_equal(st[others], ref[stream]["status"][others]) and np.array_equal(o[others], ref[stream]["out"][others])
    assert h.last_path_kernel() == (capi.KERNEL_LANE_PER_QP if stream else capi.KERNEL_LANE_PER_WAYPOINT)
    h.close()


def test_every_qp_hostile_and_the_reference_solver_setting(hip_lib):
    """a whole batch of hostile QPs (one kind each) in the reference's ADMM setting (no polish, the certificate inside the loop): all NUMERICAL, all zero"""
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    b = make_batch(len(KINDS), N, seed=9)
    global BAD
    keep = BAD
    try:
        for q, what in enumerate(KINDS):
            BAD = q
            _poison(b, what)
    finally:
        BAD = keep
    for prm in (capi.default_params(), capi.production_params()):
        h = capi.Handle(prm, device=0, max_batch=len(KINDS), max_n=N)
        r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        h.close()
        assert (r["status"] == NUMERICAL).all(), r["status"]
        assert np.all(r["out"] == 0.0)
