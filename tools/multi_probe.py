import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
prm = capi.production_params()
b1 = make_batch(1024, 80); b2 = make_batch(2048, 80)
one = capi.MultiHandle(prm, devices=(0,), max_batch_per_shard=1024, max_n=80)
two = capi.MultiHandle(prm, devices=(0, 0), max_batch_per_shard=1024, max_n=80)
h = capi.Handle(prm, max_batch=2048, max_n=80)
def rate(f, n, reps=20):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return n / ((time.perf_counter() - t0) / reps)
print("single handle host-pointer 1024: %.3g" % rate(lambda: h.solve(b1["ref"], b1["bounds"], b1["scal"], passes=1), 1024))
print("single handle host-pointer 2048: %.3g" % rate(lambda: h.solve(b2["ref"], b2["bounds"], b2["scal"], passes=1), 2048))
print("multi 1 shard 1024: %.3g" % rate(lambda: one.solve(b1["ref"], b1["bounds"], b1["scal"], passes=1), 1024))
print("multi 2 shards 2048: %.3g" % rate(lambda: two.solve(b2["ref"], b2["bounds"], b2["scal"], passes=1), 2048))
os.environ["PQP_MULTI_TRACE"] = "1"
