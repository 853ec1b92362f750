"""Indices and counters of the slowest QPs of a batch (run on the GPU box); the emulator can then replay them one by one.
Usage: python tools/stragglers.py [batch] [n] [profile] [key=value ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
profile = sys.argv[3] if len(sys.argv) > 3 else "uniform"
over = {}
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    over[k] = float(v) if ("." in v or "e" in v) else int(v)
seed = int(over.pop("seed", -1))
host = make_batch(batch, n, profile) if seed < 0 else make_batch(batch, n, profile, seed=seed)
res = capi.Handle(capi.production_params(**over), device=0, max_batch=batch, max_n=n).solve(host["ref"], host["bounds"], host["scal"], passes=1)
kkt, fac = res["info"][:, 5], res["info"][:, 6]
order = np.argsort(-kkt)[:12]
print(f"batch {batch} n {n} {profile} {over}: kkt mean {kkt.mean():.1f}")
for q in order:
    print(f"  qp {q}: status {res['status'][q]} iters {res['iters'][q]} kkt {kkt[q]:.0f} fac {fac[q]:.0f} polished {res['info'][q, 4]:.0f}")
