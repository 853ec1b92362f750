#!/usr/bin/env python
"""Probe (round 5): what would wavefronts of similar work buy path_stream_kernel at 65 536 QPs?  No kernel change: the batch itself is permuted on the
host by keys taken from a first solve of the same batch (perfect foresight - the upper bound of any predictor), uploaded again and timed.
Keys: total Riccati sweeps (round 3's experiment), and the per-phase counts (interior-point iterations and active-set rounds of each pass) a
wavefront actually runs in lock-step.  Usage: python tools/stream_sorted_probe.py [batch=65536] [n=80]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch, jitter_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda", 0)
host = make_batch(batch, n)
h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_STREAM_BATCH, 1)
out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
st = torch.zeros(batch, dtype=torch.int32, device=dev); it = torch.zeros(batch, dtype=torch.int32, device=dev)
info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)


def run(perm, label, steps=6, src=None):
    src = src or host
    ref, bounds, scal = (torch.from_numpy(np.ascontiguousarray(src[k][perm])).to(dev) for k in ("ref", "bounds", "scal"))
    torch.cuda.synchronize()
    for _ in range(2):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, iters=it, info=info)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, iters=it, info=info)
    h.sync()
    dt = (time.perf_counter() - t0) / steps
    inf = info.cpu().numpy()
    sweeps = inf[:, 6]
    # what a wavefront runs: the maxima over its 64 lanes of each phase's count
    ph = np.stack([inf[:, 2], inf[:, 5], inf[:, 3] - inf[:, 2], inf[:, 7] - inf[:, 5]], axis=1).reshape(-1, 64, 4)
    wave = ph.max(axis=1).sum(axis=1)            # interior-point iterations + active-set rounds per wavefront (each is one backward + one forward sweep or one sweep pair)
    lane = ph.sum(axis=2).mean()
    print(f"{label:46s} {dt * 1e3:8.3f} ms = {batch / dt / 1e6:5.2f} M paths/s; solved {(st == 1).sum().item()}/{batch}; sweeps mean {sweeps.mean():.1f} max {sweeps.max():.0f}; "
          f"lock-step phases per wavefront {wave.mean():.1f} (max {wave.max():.0f}) for {lane:.1f} per lane")
    return inf


inf0 = run(np.arange(batch), "index order")
rng = np.random.default_rng(0)
run(rng.permutation(batch), "random permutation")
run(np.argsort(-inf0[:, 6], kind="stable"), "sorted by total sweeps (descending)")
key = np.stack([inf0[:, 2], inf0[:, 5], inf0[:, 3] - inf0[:, 2], inf0[:, 7] - inf0[:, 5]], axis=1).astype(np.int64)
order = np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0]))[::-1]
run(order, "sorted by the four phase counts (descending)")
tot = key.sum(axis=1)
order2 = np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0], tot))[::-1]
run(order2, "sorted by total, then the phase counts")
run(order2[::-1].copy(), "... ascending (cheapest wavefronts first)")
# imperfect foresight: the keys of THIS batch applied to the same scenarios one planning cycle later (synth.jitter_batch, +-5 %), as bench.py cycles them
for v in (1, 2):
    hv = jitter_batch(host, v)
    infv = run(np.arange(batch), f"jittered variant {v}: index order", src=hv)
    run(order, f"jittered variant {v}: sorted by variant 0's phase counts", src=hv)
    kv = np.stack([infv[:, 2], infv[:, 5], infv[:, 3] - infv[:, 2], infv[:, 7] - infv[:, 5]], axis=1).astype(np.int64)
    run(np.lexsort((kv[:, 3], kv[:, 2], kv[:, 1], kv[:, 0]))[::-1], f"jittered variant {v}: sorted by its own phase counts", src=hv)
    print("   share of QPs whose four counts equal variant 0's: %.2f; first-pass iterations equal: %.2f" % ((kv == key).all(axis=1).mean(), (kv[:, 0] == key[:, 0]).mean()))
# which key order survives imperfect foresight best: variant 1 in the order of variant 0's counts
hv = jitter_batch(host, 1)
for name, cols in (("it1 s1 it2 s2", (0, 1, 2, 3)), ("it1 it2 s1 s2", (0, 2, 1, 3)), ("it2 it1 s1 s2", (2, 0, 1, 3)), ("s1 it1 s2 it2", (1, 0, 3, 2)), ("it1 it2", (0, 2)),
                   ("it1+it2, s1+s2", None)):
    if cols is None:
        kk = [key[:, 1] + key[:, 3], key[:, 0] + key[:, 2]]
    else:
        kk = [key[:, c] for c in reversed(cols)]
    run(np.lexsort(tuple(kk))[::-1], f"variant 1 by variant 0's keys, order {name}", src=hv)
# coarser keys: what a cheap predictor might still know
for name, kk in (("first-pass interior-point iterations only", key[:, 0]), ("first pass: iterations, then set rounds", key[:, 0] * 16 + key[:, 1])):
    run(np.argsort(-kk, kind="stable"), "sorted by " + name)
h.close()
