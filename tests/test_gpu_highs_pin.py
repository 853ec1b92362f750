"""The HIP kernels against a third-party QP solver (round 6): both passes of optimizePath as each path kernel returns them through the C ABI, against HiGHS
(oracle/highs_qp.py: the QP solver bundled with scipy) on the QP the oracle assembles line by line from the reference around the same linearisation point.
See tests/test_highs_pin.py for what HiGHS's own 1e-7 Hessian regularisation does to the lateral offsets (weight_l = 0: a flat direction)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import pqp_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import highs_qp as H
from highs_util import against_highs
from path_optimizer_2_amd.synth import make_batch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not H.available(), reason="this scipy does not bundle the HiGHS QP interface")]


@pytest.mark.parametrize("n,profile,batch,rough", [(80, "uniform", 1024, False), (120, "varied", 256, False), (60, "varied", 64, False), (200, "uniform", 64, False), (80, "varied", 64, True)])
@pytest.mark.parametrize("kernel", ["lane_per_waypoint", "lane_per_qp"])
def test_both_passes_of_both_kernels_against_highs(hip_lib, kernel, n, profile, batch, rough):
    from path_optimizer_2_amd import capi
    b = make_batch(batch, n, profile)
    over = dict(rough_constraints_far_away=1, precise_planning_length=12.0) if rough else {}
    h = capi.Handle(capi.production_params(**over), device=0, max_batch=batch, max_n=n)
    if kernel == "lane_per_qp":
        h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1)
    else:
        h.set_option(capi.OPT_STREAM_BATCH, 0)
    r0 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    r1 = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert h.last_path_kernel() == (capi.KERNEL_LANE_PER_QP if kernel == "lane_per_qp" else capi.KERNEL_LANE_PER_WAYPOINT)
    h.close()
    assert (r0["status"] == 1).all() and (r1["status"] == 1).all()
    prm = None
    if rough:
        prm = O.PathQpParams(); prm.rough_constraints_far_away = True; prm.precise_planning_length = 12.0
    for q in np.linspace(0, batch - 1, 4 if n <= 120 else 2).astype(int):
        ref, bounds, scal = b["ref"][q], b["bounds"][q], b["scal"][q]
        against_highs(ref, O.first_linearization(ref), bounds, scal, r0["out"][q], prm)
        against_highs(ref, r0["out"][q][:, 3:6], bounds, scal, r1["out"][q], prm)
