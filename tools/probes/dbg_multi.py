import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
b1 = make_batch(1024, 80)
two = capi.MultiHandle(capi.production_params(), devices=(0, 0), max_batch_per_shard=1024, max_n=80)
bad = {k: v.copy() for k, v in b1.items()}
bad["bounds"][7, 11, 0] = 9.0
r = two.solve(bad["ref"][:64], bad["bounds"][:64], bad["scal"][:64], passes=1)
print(r["status"], np.abs(r["out"][7]).max(), r["iters"][:10])
