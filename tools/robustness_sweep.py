"""Solve many seeded batches with the production setting and report what did not end in an accepted polish, and the slowest QPs.
Usage: python tools/robustness_sweep.py [seeds=16] [batch=8192] [first_seed=1000] [scaling=<production's>]   (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
base = int(sys.argv[3]) if len(sys.argv) > 3 else 1000          # first seed
over = {"scaling": int(sys.argv[4])} if len(sys.argv) > 4 else {}   # Ruiz passes (0: the lean kernel variant)
print("production setting" + (f" with {over}" if over else ""))
for n, profile in ((80, "uniform"), (80, "varied"), (120, "varied"), (200, "uniform"), (37, "varied"), (300, "varied")):
    b = batch if n <= 120 else batch // 4
    h = capi.Handle(capi.production_params(**over), device=0, max_batch=b, max_n=n)
    hs = capi.Handle(capi.production_params(**over), device=0, max_batch=b, max_n=n)        # the same QPs on the lane-per-QP kernel
    hs.set_option(capi.OPT_STORE_WARM, 0); hs.set_option(capi.OPT_STREAM_BATCH, 1)
    bad_s = 0; worst_pair = 0.0
    tot = bad = 0
    worst = []
    kk = []
    for s in range(seeds):
        host = make_batch(b, n, profile, seed=base + s)
        r = h.solve(host["ref"], host["bounds"], host["scal"], passes=1)
        ok = (r["status"] == 1) & (r["info"][:, 4] == 2)
        tot += b; bad += int((~ok).sum())
        rs = hs.solve(host["ref"], host["bounds"], host["scal"], passes=1)
        oks = rs["status"] == 1
        bad_s += int((~oks).sum())
        both = ok & oks
        if both.any():
            worst_pair = max(worst_pair, float(np.abs(r["out"][both][:, :, 3:5] - rs["out"][both][:, :, 3:5]).max()))
        for q in np.nonzero(~ok)[0][:3]:
            print(f"    n {n} {profile} seed {base + s} qp {q}: status {r['status'][q]} iters {r['iters'][q]} polished passes {r['info'][q, 4]:.0f}")
        k = r["info"][:, 5]; kk.append(k)
        q = int(np.argmax(k)); worst.append((float(k[q]), base + s, q))
    kk = np.concatenate(kk)
    worst.sort(reverse=True)
    print(f"n {n:3d} {profile:8s}: {tot} QPs, {bad} not solved+polished; reduced solves mean {kk.mean():.1f} p99 {np.percentile(kk, 99):.0f} "
          f"p99.99 {np.percentile(kk, 99.99):.0f} max {kk.max():.0f}; slowest (solves, seed, qp): {worst[:3]}")
    print(f"              lane-per-QP kernel on the same QPs: {bad_s} not SOLVED; largest |l, d_heading| difference between the two kernels {worst_pair:.1e}")
    h.close(); hs.close()
