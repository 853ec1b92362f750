#!/usr/bin/env python
"""The third-party pin of the optimum over WHOLE bench batches (tests/test_gpu_highs_pin.py checks 2-4 QPs per shape): every QP of a batch, both passes, both HIP
kernels' outputs through the C ABI against HiGHS's QP solver (oracle/highs_qp.py, bundled with scipy) on the oracle's line-by-line assembly of the reference's QP.
    python tools/highs_sweep.py gpu <out.npz>            # on the GPU box: solve the batches on both kernels, keep the output records
    python tools/highs_sweep.py check <out.npz> [procs]  # anywhere (CPU): HiGHS on every QP, one summary line per shape / kernel / pass
The check imports oracle/ (test infrastructure): this tool is a checker, not part of the product path."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [(80, "uniform", 1024), (120, "varied", 256), (200, "uniform", 96), (60, "varied", 128)]      # configs[1]'s batch, configs[2]'s / configs[4]'s / the demo's lengths
KERNELS = ("lane_per_waypoint", "lane_per_qp")


def gpu(path):
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    out = {}
    for n, profile, batch in SHAPES:
        b = make_batch(batch, n, profile)
        for kernel in KERNELS:
            h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
            if kernel == "lane_per_qp":
                h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1)
            else:
                h.set_option(capi.OPT_STREAM_BATCH, 0)
            r0 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0)
            r1 = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
            want = capi.KERNEL_LANE_PER_QP if kernel == "lane_per_qp" else capi.KERNEL_LANE_PER_WAYPOINT
            assert h.last_path_kernel() == want
            h.close()
            k = f"{n}_{profile}_{batch}_{kernel}"
            out[k + "_out0"], out[k + "_out1"], out[k + "_st0"], out[k + "_st1"] = r0["out"], r1["out"], r0["status"], r1["status"]
            print(k, "solved", int((r0["status"] == 1).sum()), int((r1["status"] == 1).sum()), "of", batch, flush=True)
    np.savez_compressed(path, **out)


def one(args):
    """(feasibility violation of the kernel's point, (f - f_highs) / max(1, f_highs), max |d| of (heading offset, curvature, curvature rate), max |d| of the lateral offset)"""
    import scipy.sparse as sp
    import pqp_oracle as O
    import highs_qp as H
    from highs_util import qp_point
    ref, lin, bounds, scal, o = args
    Pd, A, lo, up, sz = O.assemble_path_qp(ref, lin, bounds, scal, None)
    A = sp.csr_matrix(A)
    try:
        xh, _, _ = H.solve_qp(Pd, np.zeros(sz["vars"]), A, lo, up, tries=24, time_limit=3.0)
    except RuntimeError:          # (HiGHS's active-set solver returned no feasible "Optimal" point in 24 orderings of this QP: oracle/highs_qp.py)
        return float("nan"), float("nan"), float("nan"), float("nan")
    x = qp_point(o, A, lo, up, sz)
    Ax = A @ x
    f = lambda z: 0.5 * np.sum(Pd * z * z)
    d = np.abs(O.unpack_path(xh, ref) - o)
    return float(np.maximum(lo - Ax, Ax - up).max()), float((f(x) - f(xh)) / max(1.0, f(xh))), float(d[:, 4:7].max()), float(d[:, 3].max())


def check(path, procs):
    import pqp_oracle as O
    from path_optimizer_2_amd.synth import make_batch
    z = np.load(path)
    import multiprocessing as mp
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:          # (forked workers hang inside HiGHS)
        for n, profile, batch in SHAPES:
            b = make_batch(batch, n, profile)
            for kernel in KERNELS:
                k = f"{n}_{profile}_{batch}_{kernel}"
                o0, o1 = z[k + "_out0"], z[k + "_out1"]
                assert (z[k + "_st0"] == 1).all() and (z[k + "_st1"] == 1).all()
                for p, (outs, lins) in enumerate(((o0, None), (o1, o0))):
                    jobs = [(b["ref"][q], O.first_linearization(b["ref"][q]) if lins is None else lins[q][:, 3:6], b["bounds"][q], b["scal"][q], outs[q]) for q in range(batch)]
                    r = np.array(list(ex.map(one, jobs, chunksize=8)))
                    none = int(np.isnan(r[:, 0]).sum())
                    r = r[~np.isnan(r[:, 0])]
                    print(f"n {n:3d} {profile:8s} {batch:5d} QPs  {kernel:18s} pass {p + 1}: HiGHS answered {len(r)} (no feasible point from it in 24 orderings: {none}); rows violated by at most {r[:, 0].max():.1e}; objective (relative to HiGHS's) at most {max(r[:, 1].max(), 0.0):.1e} above, at most {max(-r[:, 1].min(), 0.0):.1e} below, lower than HiGHS's in {int((r[:, 1] < 0).sum())} QPs; "
                          f"|d(heading offset, curvature, curvature rate)| max {r[:, 2].max():.1e} p99 {np.percentile(r[:, 2], 99):.1e}; |d lateral offset| max {r[:, 3].max():.1e} p99 {np.percentile(r[:, 3], 99):.1e} median {np.median(r[:, 3]):.1e}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "gpu":
        gpu(sys.argv[2])
    else:
        check(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
