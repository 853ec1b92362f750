"""BASELINE.json configs[4] under test: TensionSmoother2 QP + path QP (N = 200) of one GPU's shard (512 scenarios) on two handles /
HIP streams, ordered only by events (path_optimizer_2_amd/pipeline.py).  And configs[2] / configs[3] at their full batch through
size-independent properties.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

import pqp_oracle as O
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.pipeline import SmootherPathPipeline
from path_optimizer_2_amd.synth import make_batch

pytestmark = pytest.mark.gpu


def test_two_stream_pipeline_equals_the_serial_run(hip_lib):
    """512 scenarios x N = 200: smoother chain of batch k + 1 overlapping the path QP of batch k gives bit for bit what the
    host-synchronised one-after-the-other run gives, for every one of six batches with three distinct input sets cycling through two
    buffer slots (a path QP that read a stale or half-written buffer would differ)."""
    B, n, steps = 512, 200, 6
    want = []
    ser = SmootherPathPipeline(B, n, variants=3)
    for k in range(steps):
        ser.step_serial(k)
        want.append(ser.result(k))
    ser.close()
    assert all((w["st"] == 1).all() and (w["sm_st"] == 1).all() and (w["count"] == n).all() for w in want)
    assert not np.array_equal(want[0]["out"], want[1]["out"])                       # the variants really differ
    pipe = SmootherPathPipeline(B, n, variants=3)
    got = {}
    for k in range(steps):
        pipe.step_pipelined(k)           # no host wait between the batches
    pipe.sync()
    for k in (steps - 2, steps - 1):     # the two batches whose buffers are still in the slots
        got[k] = pipe.result(k)
    pipe.close()
    for k, g in got.items():
        for key in ("sx", "ref", "scal", "out", "st", "it"):
            same_as = [j for j in range(steps) if np.array_equal(g[key], want[j][key])]
            rows = np.where((g[key] != want[k][key]).reshape(B, -1).any(axis=1))[0]
            np.testing.assert_array_equal(g[key], want[k][key], err_msg=f"batch {k} {key} (equal to the serial run's batches {same_as}; {len(rows)} scenarios "
                                          f"differ, first {rows[:8]}; smoother solved {int((g['sm_st'] == 1).sum())}/{B}; nan {int(np.isnan(g[key]).sum())})")
    # the path QP's answer is the optimum of the QP its reference states define (a sample against the converged oracle)
    g = got[steps - 1]
    bounds = make_batch(B, n, seed=20260926)["bounds"]
    for q in (0, 17, 511):
        ref = O.solve_path(g["ref"][q], bounds[q], g["scal"][q], st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
        assert np.abs(g["out"][q][:, 3:5] - ref[-1]["out"][:, 3:5]).max() < 1e-4


def test_the_event_is_what_orders_the_two_streams(hip_lib):
    """The same schedule WITHOUT the smoother -> path event: the path QP starts while the smoother chain (>= 1 ms) is still running and
    reads the reference-state buffer as it was (poisoned with NaN here), so its result is wrong; with the event it is right."""
    import torch
    B, n = 512, 200
    ser = SmootherPathPipeline(B, n)
    ser.step_serial(0)
    want = ser.result(0)
    ser.close()
    def run(gate):
        pipe = SmootherPathPipeline(B, n)
        for b in pipe.buf:
            b["ref"].fill_(float("nan")); b["scal"][:, 0:4].fill_(float("nan")); b["count"].fill_(n)
        torch.cuda.synchronize()
        pipe.step_pipelined(0, gate=gate)
        pipe.sync()
        got = pipe.result(0)
        pipe.close()
        return got
    got = run(True)
    np.testing.assert_array_equal(got["out"], want["out"])
    assert (got["st"] == 1).all()
    # without the event the outcome is a race the path QP normally loses by a millisecond - but it is a race: a host that is slow to enqueue the
    # path kernel lets the smoother finish first.  Observed once in a few attempts, or the negative half of the test is skipped (never failed)
    for _ in range(5):
        got = run(False)
        if not (got["st"] == 1).all() and not np.array_equal(got["out"], want["out"]):
            return
    pytest.skip("the un-gated path QP never overtook the smoother chain in 5 attempts on this box")


def _properties(b, r, kap_wheel_base=2.5):
    out = r["out"]
    assert (r["status"] == 1).all()
    assert np.isfinite(out).all()
    kap = np.tan(b["scal"][:, 5]) / kap_wheel_base
    assert (np.abs(out[:, :, 5]) <= kap[:, None] + 1e-4).all()                 # curvature box (per-QP steering limit)
    assert (np.abs(out[:, -1, 3]) <= 1.0 + 1e-4).all()                          # end-l box
    ds = np.diff(b["ref"][:, :, 0], axis=1)
    assert np.abs(out[:, 1:, 5] - out[:, :-1, 5] - ds * out[:, :-1, 6]).max() < 1e-4      # k_{i+1} = k_i + ds dk_i (exact row)
    assert np.abs(out[:, 0, 3] - b["scal"][:, 0]).max() < 1e-4                  # initial state pinned
    assert np.abs(out[:, 0, 4] - b["scal"][:, 1]).max() < 1e-4
    assert np.abs(out[:, 0, 5] - b["scal"][:, 2]).max() < 1e-4
    # unpack: (x, y) = ref + l * normal, heading = ref heading + d_heading (base_solver.cpp:263-288)
    nx, ny = np.cos(b["ref"][:, :, 2] + np.pi / 2), np.sin(b["ref"][:, :, 2] + np.pi / 2)
    assert np.abs(out[:, :, 0] - (b["ref"][:, :, 3] + out[:, :, 3] * nx)).max() < 1e-9
    assert np.abs(out[:, :, 1] - (b["ref"][:, :, 4] + out[:, :, 3] * ny)).max() < 1e-9


@pytest.mark.parametrize("cfg,batch,n,profile", [(2, 8192, 120, "varied"), (3, 8192, 80, "uniform")])
def test_full_batch_of_configs_2_and_3(hip_lib, cfg, batch, n, profile):
    """configs[2] (8192 x N = 120, varied start / goal / curvature limits) and one GPU's shard of configs[3] (8192 x N = 80) at full
    size: every path solved and polished, the size-independent properties hold, a sample equals the converged oracle, the result
    does not depend on the order the QPs are started in (cost order on / off) and is bit-reproducible."""
    b = make_batch(batch, n, profile)
    h = capi.Handle(capi.production_params(), max_batch=batch, max_n=n)
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    _properties(b, r)
    assert (r["info"][:, 4] >= 1).all()
    for q in (0, batch // 2, batch - 1):
        ref = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000))
        assert np.abs(r["out"][q][:, 3:5] - ref[-1]["out"][:, 3:5]).max() < 1e-4
    h.set_option(capi.OPT_ORDER_BY_COST, 1)
    h.set_option(capi.OPT_STORE_WARM, 0)
    for _ in range(2):                       # the second call runs most-expensive-first by the first one's costs
        r2 = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        np.testing.assert_array_equal(r2["out"], r["out"])
        np.testing.assert_array_equal(r2["iters"], r["iters"])
        np.testing.assert_array_equal(r2["info"][:, 3:7], r["info"][:, 3:7])
    h.close()
