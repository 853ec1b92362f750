"""Which QPs of path_stream_kernel's sorted launches differ from the unsorted launch's by more than 1e-6, and which of the two is the optimum?  (GPU box)
Saves the worst QPs' inputs and both outputs to gpurun_out/sorted_diff_qps.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
n, profile, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 300, "varied", 49152
dev = torch.device("cuda", 0)
h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1)
out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev); st = torch.zeros(batch, dtype=torch.int32, device=dev); info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
keep = []
for s in range(4):
    b = make_batch(batch, n, profile, seed=5000 + s)
    ref, bounds, scal = (torch.from_numpy(b[k]).to(dev) for k in ("ref", "bounds", "scal"))
    h.set_option(capi.OPT_ORDER_BY_COST, 1)
    res = []
    for k in range(2):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, info=info); h.sync()
        res.append((out.cpu().numpy().copy(), st.cpu().numpy().copy(), info.cpu().numpy().copy()))
    (o0, s0, i0), (o1, s1, i1) = res
    d = np.abs(o0 - o1)[:, :, 3:6].max(axis=(1, 2))
    idx = np.argsort(-d)[:6]
    print(f"seed {5000 + s}: QPs beyond 1e-6: {(d > 1e-6).sum()}, beyond 1e-5: {(d > 1e-5).sum()}")
    for q in idx[:4]:
        print(f"   qp {q}: diff {d[q]:.2e} at waypoint {np.abs(o0[q] - o1[q])[:, 3:6].max(axis=1).argmax()}; unsorted info {i0[q, 2:8]} sorted info {i1[q, 2:8]}")
    # the lane-per-waypoint kernel on the worst ones
    hl = capi.Handle(capi.production_params(), device=0, max_batch=8, max_n=n)
    hl.set_option(capi.OPT_STREAM_BATCH, 0)
    rl = hl.solve(b["ref"][idx], b["bounds"][idx], b["scal"][idx], passes=1)
    hl.close()
    for j, q in enumerate(idx[:4]):
        print(f"   qp {q}: lane-per-waypoint kernel status {rl['status'][j]}: |unsorted - it| {np.abs(o0[q] - rl['out'][j])[:, 3:6].max():.2e}  |sorted - it| {np.abs(o1[q] - rl['out'][j])[:, 3:6].max():.2e}")
    keep.append(dict(ref=b["ref"][idx], bounds=b["bounds"][idx], scal=b["scal"][idx], unsorted=o0[idx], sorted=o1[idx], lane=rl["out"], qp=idx, seed=5000 + s))
np.savez("gpurun_out/sorted_diff_qps.npz", **{f"{k}_{i}": v for i, kk in enumerate(keep) for k, v in kk.items()})
