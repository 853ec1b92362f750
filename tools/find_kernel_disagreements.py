import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
import pqp_oracle_c as OC
for n, profile, b in ((300, "varied", 2048), (200, "uniform", 2048)):
    h = capi.Handle(capi.production_params(), device=0, max_batch=b, max_n=n)
    hs = capi.Handle(capi.production_params(), device=0, max_batch=b, max_n=n)
    hs.set_option(capi.OPT_STORE_WARM, 0); hs.set_option(capi.OPT_STREAM_BATCH, 1)
    found = []
    for s in range(16):
        host = make_batch(b, n, profile, seed=1000 + s)
        r = h.solve(host["ref"], host["bounds"], host["scal"], passes=1)
        rs = hs.solve(host["ref"], host["bounds"], host["scal"], passes=1)
        d = np.abs(r["out"][:, :, 3:5] - rs["out"][:, :, 3:5]).max(axis=(1, 2))
        for q in np.nonzero(d > 5e-5)[0]:
            found.append((float(d[q]), 1000 + s, int(q)))
            o = OC.solve_batch(OC.params(eps_abs=1e-10, eps_rel=1e-10, max_iter=400000), host["ref"][q:q+1], host["bounds"][q:q+1], host["scal"][q:q+1], passes=1)
            e1 = np.abs(o["out"][0][:, 3:5] - r["out"][q][:, 3:5]).max(); e2 = np.abs(o["out"][0][:, 3:5] - rs["out"][q][:, 3:5]).max()
            print(f"n {n} seed {1000+s} qp {q}: kernels differ by {d[q]:.2e}; lane-per-waypoint vs oracle {e1:.2e} (status {r['status'][q]}, polished passes {r['info'][q,4]:.0f}, solves {r['info'][q,5]:.0f}); lane-per-QP vs oracle {e2:.2e} (status {rs['status'][q]}, info {rs['info'][q].tolist()}); oracle status {o['status'][0] if 'status' in o else None} iters {o.get('iters', [None])[0] if isinstance(o.get('iters'), np.ndarray) else None}")
    print(n, profile, 'pairs above 5e-5:', len(found))
    h.close(); hs.close()
