"""BASELINE.json configs[4] on one GPU's shard: the TensionSmoother2 QP of scenario batch k and the path QP of batch k on two
handles (= two HIP streams), coupled only by events ("smoother-QP stream feeds path-QP stream via an event"), so that
the smoother of batch k + 1 overlaps the path QP of batch k.

Per scenario, on the smoother handle's stream (the reference's order, reference_path_smoother.cpp:31-45 -> path_optimizer.cpp:106-161):
    TensionSmoother2 QP over n points  ->  tk::spline through the smoothed points  ->  N reference states + initial error
and on the path handle's stream, behind the smoother handle's mark:
    path QP (cold solve + re-linearised warm re-solve) on those reference states.
The path QP READS what the smoother chain wrote (reference states, initial error), so the event is a real dependency: without it the
path QP runs on whatever the buffers held (tests/test_gpu_pipeline.py shows exactly that).  Corridor bounds are synthetic (Frenet
frame, synth.make_batch) - the map-based bounds step is pqp_corridor_bounds and needs an obstacle map per scenario.

Python here is plumbing (buffers, call order); every step is a C-ABI call on device pointers.  The three scalars a scenario's `scal`
row takes from the reference states (start curvature, target heading) are copied by two strided device copies enqueued on the smoother
stream through torch.
"""
import os

import numpy as np

from . import capi
from .synth import make_batch


def raw_lines(batch, n, spacing=0.31, seed=0):
    """`batch` noisy curved polylines of n points (what segmentRawReference hands to the smoother QP): x, y, heading, curvature,
    arclength lists [batch][n], drawn from 64 distinct lines."""
    out = []
    for b in range(min(batch, 64)):
        rng = np.random.default_rng(seed * 1000 + b)
        s = np.arange(n) * spacing
        k = 0.05 * np.sin(s / 9.0 + rng.uniform(0, 6.28)) + rng.uniform(-0.01, 0.01)
        ang = np.concatenate([[0.3], 0.3 + np.cumsum(0.5 * (k[1:] + k[:-1]) * spacing)])
        x = np.concatenate([[1.0], 1.0 + np.cumsum(np.cos(0.5 * (ang[1:] + ang[:-1])) * spacing)]) + rng.normal(scale=0.02, size=n)
        y = np.concatenate([[-2.0], -2.0 + np.cumsum(np.sin(0.5 * (ang[1:] + ang[:-1])) * spacing)]) + rng.normal(scale=0.02, size=n)
        out.append((x, y, ang, k, s))
    return [np.ascontiguousarray(np.stack([out[b % len(out)][j] for b in range(batch)])) for j in range(5)]


class SmootherPathPipeline:
    """slots = buffers in flight (2: smoother k + 1 may overwrite what path QP k - 1 read, not what path QP k reads)."""

    def __init__(self, batch, n, device=0, slots=2, seed=0, ds=0.3, smoother_params=None, path_params=None, variants=1):
        import torch
        self.torch, self.batch, self.n, self.slots, self.ds = torch, batch, n, slots, ds
        dev = torch.device("cuda", device)
        self.dev = dev
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
        # x, y, angle, k, s  [batch][n]; step k smooths variant k % variants (distinct inputs make a stale buffer visible in tests)
        self.raws = [[t(a) for a in raw_lines(batch, n, seed=seed + v)] for v in range(variants)]
        self.raw = self.raws[0]
        self.max_s = torch.full((batch,), ds * (n - 1) + 0.1, dtype=torch.float64, device=dev)   # N = n states 0, ds, ..., ds (n - 1)
        start = np.stack([self.raw[0][:, 0].cpu().numpy() + 0.05, self.raw[1][:, 0].cpu().numpy() + 0.1, np.full(batch, 0.28)], axis=1)
        self.start = t(start)
        host = make_batch(batch, n, seed=20260926 + seed)
        self.bounds = t(host["bounds"])
        self.scal0 = t(host["scal"])                                                      # columns 4, 5 (blocked, steering limit) are kept
        self.buf = [dict(sx=z(batch, n), sy=z(batch, n), ss=z(batch, n), tab=z(batch, 9, n), ext=z(batch, 4), ref=z(batch, n, 5),
                         count=z(batch, dt=torch.int32), err=z(batch, 2), scal=self.scal0.clone(), out=z(batch, n, 7),
                         st=z(batch, dt=torch.int32), it=z(batch, dt=torch.int32), sm_st=z(batch, dt=torch.int32), sm_it=z(batch, dt=torch.int32))
                    for _ in range(slots)]
        # the reference runs its smoother QPs at OSQP's default eps 1e-3 (tension_smoother_2.cpp:32-36)
        self.hs = capi.Handle(smoother_params or capi.default_params(eps_abs=1e-3, eps_rel=1e-3), device=device, max_batch=batch, max_n=n)
        self.hp = capi.Handle(path_params or capi.production_params(), device=device, max_batch=batch, max_n=n)
        self.hp.set_option(capi.OPT_STORE_WARM, 0)
        # the path QP's wavefronts own their SIMDs' whole register files: 32 of the 256 compute units are left to the smoother stream's kernels of
        # the next batch, which otherwise only get onto the chip in the path kernel's tail (0.87 -> 0.73 ms per step; 8: 0.81, 64: 0.77, 96: 0.99)
        self.hp.set_option(capi.OPT_RESERVE_CUS, int(os.environ.get("PQP_RESERVE_CUS", "32")))
        self.s_sm = torch.cuda.ExternalStream(self.hs.stream(), device=dev)
        torch.cuda.synchronize(dev)

    def close(self):
        self.sync()
        self.hs.close(); self.hp.close()

    def sync(self):
        self.hs.sync(); self.hp.sync()

    # -- the two halves of a step ------------------------------------------------------------------------------------------------
    def smoother_chain(self, k):
        """smoother QP -> spline -> reference states + initial error -> scal row, all on the smoother handle's stream."""
        b, B, n, p = self.buf[k % self.slots], self.batch, self.n, (lambda x: capi.C.c_void_p(x.data_ptr()))
        lib, h = self.hs.lib, self.hs._h
        x, y, ang, kk, s = self.raws[k % len(self.raws)]
        chk = self.hs._check
        chk(lib.pqp_smooth_tension2_device(h, B, n, p(x), p(y), p(ang), p(kk), p(s), p(b["sx"]), p(b["sy"]), p(b["ss"]), p(b["sm_st"]), p(b["sm_it"]), None))
        chk(lib.pqp_spline_fit_device(h, B, n, p(b["ss"]), p(b["sx"]), p(b["sy"]), p(b["tab"]), p(b["ext"])))
        chk(lib.pqp_reference_states_device(h, B, n, n, p(b["tab"]), p(b["ext"]), p(self.max_s), p(self.start), self.ds, self.ds, 0,
                                            p(b["ref"]), p(b["count"]), p(b["err"])))
        with self.torch.cuda.stream(self.s_sm):                    # scal = (init_err[0], init_err[1], start k, target heading, blocked, steer)
            b["scal"][:, 0:2].copy_(b["err"])
            b["scal"][:, 2].copy_(b["ref"][:, 0, 1])
            b["scal"][:, 3].copy_(b["ref"][:, n - 1, 2])

    def path_qp(self, k):
        b = self.buf[k % self.slots]
        self.hp.solve_var_device(self.batch, self.n, b["count"], b["ref"], self.bounds, b["scal"], b["out"], passes=1, status=b["st"], iters=b["it"])

    # -- schedules -----------------------------------------------------------------------------------------------------------------
    def step_pipelined(self, k, gate=True):
        """No host wait.  gate=False leaves out the event that orders smoother k before path QP k (for the test that shows the
        event is what makes the result right)."""
        slot = k % self.slots
        if k >= self.slots:
            self.hs.wait_mark(self.hp, slot)          # the buffers of batch k - slots must have been read by their path QP
        self.smoother_chain(k)
        self.hs.mark(slot)
        if gate:
            self.hp.wait_mark(self.hs, slot)
        self.path_qp(k)
        self.hp.mark(slot)

    def step_serial(self, k):
        """One after the other with the host waiting in between (a single-stream host loop)."""
        self.smoother_chain(k)
        self.hs.sync()
        self.path_qp(k)
        self.hp.sync()

    def result(self, k):
        b = self.buf[k % self.slots]
        return {key: b[key].cpu().numpy() for key in ("out", "st", "it", "count", "ref", "scal", "sm_st", "sx", "sy")}
