"""Would an interior-point predictor shorten the tail of a 1024-QP launch (task 5)?  A trace-driven model, run on the GPU box.

The lane-per-waypoint kernel's slowest QP of bench.py's batch runs 45 reduced solves + 24 factorisations against a mean of 14.3 / 7.8; one launch at a
time lasts as long as that QP.  The lane-per-QP kernel predicts the active set by interior-point iterations instead of ADMM and has the tighter tail
RATIO (configs[3]: 13.8 iterations + 2.4 rounds on average, p99 23, max 30 Riccati sweeps).  In the lane-per-waypoint kernel an interior-point
iteration would be: barrier weights per row -> factor() -> one reduced solve -> step length (a reduction) -> update, i.e. one factorisation + one
solve + one residual-like phase each.  This script takes THE SAME 1024 scenarios through both kernels, reads every QP's own counts (reduced solves
/ factorisations in the lane kernel; interior-point iterations / active-set rounds per pass in the lane-per-QP kernel) and prices both with the
per-operation times measured this round on the device clock (profiles/r06i_timeline_*: factorisation 5.7 us, reduced solve 2.5 us, residuals 1.4 us,
polish set operations 1.6 us per round, fixed part of a path 38 us).
Usage: python tools/ipm_tail_model.py [batch] [n] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch

T_FACTOR, T_SOLVE, T_RES, T_SETOPS, T_FIXED = 5.7, 2.5, 1.4, 1.6, 38.0       # microseconds (profiles/r06i_timeline_after_lds_round_trips.txt)

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
seed = int(sys.argv[3]) if len(sys.argv) > 3 else None
host = make_batch(batch, n) if seed is None else make_batch(batch, n, seed=seed)


def run(stream):
    h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
    h.set_option(capi.OPT_STORE_WARM, 0)
    h.set_option(capi.OPT_STREAM_BATCH, 1 if stream else 0)
    r = h.solve(host["ref"], host["bounds"], host["scal"], passes=1)
    h.close()
    assert (r["status"] == 1).all()
    return r


lane, lq = run(0), run(1)
kkt, fac = lane["info"][:, 5], lane["info"][:, 6]
ipm, rounds = lq["info"][:, 3], lq["info"][:, 7]                                  # both passes: interior-point iterations, active-set rounds
# the lane kernel as it is: every factorisation is followed by a look at the residuals and a set operation; the first one of each pass is not
t_lane = T_FIXED + T_SOLVE * kkt + T_FACTOR * fac + (T_RES + T_SETOPS) * np.maximum(fac - 2, 0) + T_RES * 2
# the same path predicted by interior-point iterations in the lane kernel: an iteration = factorisation + solve + step-length reduction (a residual-like
# phase) + weight update (a set-operation-like phase); an active-set round = factorisation + 2 solves + residuals + set operation
t_ipm = T_FIXED + ipm * (T_FACTOR + T_SOLVE + T_RES + T_SETOPS) + rounds * (T_FACTOR + 2 * T_SOLVE + T_RES + T_SETOPS)
print(f"batch {batch} n {n}: lane-per-waypoint kernel reduced solves mean {kkt.mean():.1f} max {kkt.max():.0f}, factorisations mean {fac.mean():.1f} max {fac.max():.0f}; "
      f"lane-per-QP kernel interior-point iterations mean {ipm.mean():.1f} p99 {np.percentile(ipm, 99):.0f} max {ipm.max():.0f}, rounds mean {rounds.mean():.1f} max {rounds.max():.0f}")
print(f"modelled time per path, us: as it is mean {t_lane.mean():.0f} p99 {np.percentile(t_lane, 99):.0f} max {t_lane.max():.0f}   |   interior-point predictor for every QP: "
      f"mean {t_ipm.mean():.0f} p99 {np.percentile(t_ipm, 99):.0f} max {t_ipm.max():.0f}")
order = np.argsort(-t_lane)[:16]
print("the 16 slowest paths of the launch (qp: solves / factorisations -> us as it is | interior-point iterations / rounds -> us with the predictor):")
for q in order:
    print(f"  qp {q:5d}: {kkt[q]:3.0f} / {fac[q]:2.0f} -> {t_lane[q]:4.0f}   |   {ipm[q]:3.0f} / {rounds[q]:2.0f} -> {t_ipm[q]:4.0f}")
# the predictor for the tail only: a QP switches once it has spent `k` active-set rounds without acceptance in pass 1 (no previous-cycle state)
for k in (4, 6, 8, 10):
    tail = fac >= k + 2 + 1                                                     # (first pass: 2 set-up factorisations, then one per round)
    spent = T_FIXED / 2 + 5 * T_SOLVE + 2 * T_FACTOR + k * (T_FACTOR + 1.5 * T_SOLVE + T_RES + T_SETOPS)
    t_mix = np.where(tail, spent + t_ipm - T_FIXED / 2, t_lane)
    slots = 512
    def launch(t):           # most expensive first on `slots` persistent workgroups (what PQP_OPT_ORDER_BY_COST does with the previous cycle's costs)
        end = np.zeros(slots)
        for v in np.sort(t)[::-1]:
            i = int(np.argmin(end)); end[i] += v
        return end.max()
    print(f"switch after {k:2d} rounds: {int(tail.sum()):4d} QPs switch; slowest path {t_lane.max():.0f} -> {t_mix.max():.0f} us, mean {t_lane.mean():.1f} -> {t_mix.mean():.1f} us; "
          f"one launch at a time {launch(t_lane):.0f} -> {launch(t_mix):.0f} us (modelled)")
