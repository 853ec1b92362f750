"""BASELINE.json configs[4] on one GPU's shard: TensionSmoother2 QP (n points) + path QP (N = n waypoints), pipelined on two HIP
streams -- the smoother of batch k+1 runs on its handle's stream while the path QP of batch k runs on the other handle's stream;
an event recorded after smoother k gates path QP k ("smoother-QP stream feeds path-QP stream via an event").
The two QPs' inputs are synthetic and independent (tests/smoother_cases.py, synth.make_batch); the event is the only coupling.
Usage: python tools/bench_two_stream.py [batch=512] [n=200] [steps=12]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
from smoother_cases import tension_inputs

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda", 0)
cases = [tension_inputs(n, seed=b) for b in range(64)]
rep = lambda k: torch.from_numpy(np.stack([cases[b % 64][k] for b in range(batch)])).to(dev)
x, y, ang, kk, s = (rep(k) for k in range(5))
host = make_batch(batch, n)
ref, bounds, scal = (torch.from_numpy(host[k]).to(dev) for k in ("ref", "bounds", "scal"))
z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
sm_out = [(z(batch, n), z(batch, n), z(batch, n)) for _ in range(2)]            # double-buffered smoother outputs
sm_st, sm_it = z(batch, dt=torch.int32), z(batch, dt=torch.int32)
out, st = z(batch, n, 7), z(batch, dt=torch.int32)
p = lambda t: capi.C.c_void_p(t.data_ptr())
hs = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), device=0, max_batch=batch, max_n=n)   # the reference's smoother setting
hp = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)
s_sm, s_qp = torch.cuda.ExternalStream(hs.stream()), torch.cuda.ExternalStream(hp.stream())


def smooth(k):
    ox, oy, os_ = sm_out[k % 2]
    assert hs.lib.pqp_smooth_tension2_device(hs._h, batch, n, p(x), p(y), p(ang), p(kk), p(s), p(ox), p(oy), p(os_), p(sm_st), p(sm_it), None) == 0


def path(k):
    hp.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st)


def both_sync():
    hs.sync(); hp.sync()


def timed(body):
    for k in range(2):
        body(k)
    both_sync()
    t0 = time.perf_counter()
    for k in range(steps):
        body(k)
    both_sync()
    return (time.perf_counter() - t0) / steps


def serial(k):                          # one after the other, host waits in between (what a single-stream host loop does)
    smooth(k); hs.sync(); path(k); hp.sync()


def pipelined(k):                       # no host wait: smoother k+1 overlaps path QP k; the event orders smoother k before path QP k
    smooth(k)
    ev = torch.cuda.Event()
    ev.record(s_sm)
    s_qp.wait_event(ev)
    path(k)


t_sm = timed(lambda k: smooth(k))
t_qp = timed(lambda k: path(k))
t_ser = timed(serial)
t_pipe = timed(pipelined)
ok_sm, ok_qp = int((sm_st == 1).sum().item()), int((st == 1).sum().item())
print(f"configs[4] shard: batch {batch}, TensionSmoother2 n {n} + path QP N {n}; smoother solved {ok_sm}/{batch} (mean iters {sm_it.double().mean().item():.0f}), paths solved {ok_qp}/{batch}")
print(f"  smoother alone         {t_sm * 1e3:8.2f} ms/batch")
print(f"  path QP alone          {t_qp * 1e3:8.2f} ms/batch")
print(f"  one after the other    {t_ser * 1e3:8.2f} ms/batch = {batch / t_ser:9.0f} scenarios/s")
print(f"  two streams + event    {t_pipe * 1e3:8.2f} ms/batch = {batch / t_pipe:9.0f} scenarios/s  ({t_ser / t_pipe:.2f}x)")
