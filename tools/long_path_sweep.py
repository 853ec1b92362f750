#!/usr/bin/env python
"""Long paths through the lane-per-QP kernel on the GPU (the only kernel beyond 512 waypoints): QPs that do not end SOLVED, active-set rounds, Riccati sweeps
over seeds and both scenario profiles.  The host emulation of the same source found 1 QP in 16 384 at 512 waypoints and 5 in 131 072 at 1000 whose plain
active-set rounds cycle (tests/test_lq_emulation.py::CYCLING) before the guarded rounds of round 5.
Usage: python tools/long_path_sweep.py [seeds=8] [batch=8192] [sizes=512,700,1000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    sizes = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "512,700,1000").split(",")]
    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    dev = torch.device("cuda", 0)
    for n in sizes:
        h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
        h.set_option(capi.OPT_STORE_WARM, 0)
        h.set_option(capi.OPT_STREAM_BATCH, 1)
        out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
        st = torch.zeros(batch, dtype=torch.int32, device=dev); it = torch.zeros(batch, dtype=torch.int32, device=dev)
        info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
        tot = bad = 0; rounds = []; sweeps = []; guarded = 0
        for s in range(seeds):
            for prof in ("varied", "uniform"):
                b = make_batch(batch, n, prof, seed=3000 + s)
                ref, bounds, scal = (torch.from_numpy(b[k]).to(dev) for k in ("ref", "bounds", "scal"))
                h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, iters=it, info=info)
                h.sync()
                assert h.last_path_kernel() == capi.KERNEL_LANE_PER_QP
                stat = st.cpu().numpy(); inf = info.cpu().numpy()
                tot += batch; bad += int((stat != 1).sum())
                for q in np.nonzero(stat != 1)[0][:4]:
                    print(f"   n {n} {prof} seed {3000 + s} qp {q}: status {stat[q]} info {inf[q]}")
                rounds.append(inf[:, 7]); sweeps.append(inf[:, 6])
                guarded += int((inf[:, 7] > 36).sum())
        h.close()
        rounds = np.concatenate(rounds); sweeps = np.concatenate(sweeps)
        print(f"n {n}: {tot} QPs, {bad} not SOLVED; active-set rounds mean {rounds.mean():.2f} p99.9 {np.percentile(rounds, 99.9):.0f} max {rounds.max():.0f} "
              f"({guarded} QPs with more than 36 rounds over the two passes); Riccati sweeps mean {sweeps.mean():.1f} max {sweeps.max():.0f}", flush=True)


if __name__ == "__main__":
    main()
