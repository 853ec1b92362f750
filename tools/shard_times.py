"""Kernel time of each rank's shard of the N-GPU bench workload, measured one after the other on ONE GPU: the multi-GPU figure is
set by the slowest shard (bench.py takes the max over ranks).  Usage: python tools/shard_times.py [ranks=8] [batch=1024] [n=80]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dev = torch.device("cuda", 0)
h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
ts = []
for r in range(ranks):
    host = make_batch(batch, n, first_qp=r * batch)
    ref, bounds, scal = (torch.from_numpy(host[k]).to(dev) for k in ("ref", "bounds", "scal"))
    for _ in range(8):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, info=info)
    h.sync()
    ms = float(np.mean(h.kernel_ms_history(5)))
    k = info.cpu().numpy()[:, 5]
    ts.append(ms)
    print(f"rank {r}: kernel {ms:.3f} ms, reduced solves mean {k.mean():.1f} max {k.max():.0f}")
print(f"slowest / rank 0 = {max(ts) / ts[0]:.2f}; weak-scaling efficiency bound from the shards alone = {ts[0] / max(ts):.2f}")
