"""path_stream_kernel's sorted launches (PQP_OPT_ORDER_BY_COST with a map: the re-linearised pass starts from the first pass's active set) against its unsorted
ones over seeded batches: QPs that do not end SOLVED, statuses that differ, the largest difference of the paths, the share of second passes without an
interior-point iteration, sweeps per path.  Usage: python tools/sorted_launch_sweep.py [seeds=4]      (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
for n, profile, batch in ((80, "uniform", 65536), (80, "varied", 65536), (120, "varied", 49152), (200, "uniform", 49152), (37, "varied", 65536), (300, "varied", 49152)):
    h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
    h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1); h.set_option(capi.OPT_ORDER_BY_COST, 1)
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev); st = torch.zeros(batch, dtype=torch.int32, device=dev)
    info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
    tot = bad0 = bad1 = differ = 0; worst = 0.0; hits = []; sw0 = []; sw1 = []; r2max = 0
    for s in range(seeds):
        b = make_batch(batch, n, profile, seed=5000 + s)
        ref, bounds, scal = (torch.from_numpy(b[k]).to(dev) for k in ("ref", "bounds", "scal"))
        h.set_option(capi.OPT_ORDER_BY_COST, 1)          # (forgets the map: the next launch is unsorted)
        res = []
        for k in range(3):
            h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, info=info); h.sync()
            res.append((out.cpu().numpy().copy(), st.cpu().numpy().copy(), info.cpu().numpy().copy()))
        (o0, s0, i0), (o1, s1, i1), (o2, s2, i2) = res
        assert h.last_path_kernel() == capi.KERNEL_LANE_PER_QP
        assert (i0[s0 == 1, 3] - i0[s0 == 1, 2] > 0).all(), "the first launch of a shape is unsorted"
        assert (o1 == o2).all() and (s1 == s2).all(), "sorted launches reproduce each other"
        tot += batch; bad0 += int((s0 != 1).sum()); bad1 += int((s1 != 1).sum()); differ += int((s0 != s1).sum())
        ok = (s0 == 1) & (s1 == 1)
        worst = max(worst, float(np.abs(o0[ok] - o1[ok])[:, :, 3:6].max()))
        hits.append((i1[ok, 3] - i1[ok, 2] == 0).mean()); sw0.append(i0[:, 6].mean()); sw1.append(i1[:, 6].mean()); r2max = max(r2max, int((i1[:, 7] - i1[:, 5]).max()))
    print(f"n {n:3d} {profile:8s}: {tot} QPs; not SOLVED unsorted {bad0} sorted {bad1}, statuses that differ {differ}; largest |l, d_heading, kappa| difference {worst:.1e}; "
          f"second passes without an interior-point iteration {np.mean(hits):.3f}; Riccati sweeps per path {np.mean(sw0):.1f} -> {np.mean(sw1):.1f}; second-pass rounds max {r2max}", flush=True)
    h.close()
    del out, st, info
    torch.cuda.empty_cache()
