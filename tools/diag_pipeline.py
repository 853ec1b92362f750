"""Stress of the two-stream pipeline (debug tool): many fresh pipelines, every batch still in a buffer slot compared with the
host-synchronised serial run.  Usage: python tools/diag_pipeline.py [trials=30] [slots=2]"""
import sys, numpy as np
sys.path.insert(0, '.')
from path_optimizer_2_amd.pipeline import SmootherPathPipeline
B, n, steps = 512, 200, 6
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
want = []
ser = SmootherPathPipeline(B, n, variants=3, slots=steps)
for k in range(steps):
    ser.step_serial(k); want.append(ser.result(k))
ser.close()
bad_total = 0
for trial in range(trials):
    pipe = SmootherPathPipeline(B, n, variants=3, slots=SL)
    for k in range(steps):
        pipe.step_pipelined(k)
    pipe.sync()
    for k in range(steps - SL, steps):
        g = pipe.result(k)
        for key in ("sx", "ref", "scal", "out"):
            if not np.array_equal(g[key], want[k][key]):
                rows = np.where((g[key] != want[k][key]).reshape(B, -1).any(axis=1))[0]
                print(f"trial {trial} batch {k} {key}: {len(rows)} scenarios differ (first {rows[:6]}), smoother status ok {int((g['sm_st'] == 1).sum())}/{B}, nan {int(np.isnan(g[key]).sum())}")
                bad_total += 1
                break
    pipe.close()
print("mismatching batches:", bad_total)
