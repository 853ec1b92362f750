#!/bin/bash
# Round 5, GPU call 3: full GPU suite (exact smoother kernels at 700 / 1000 points, chain graph on the lane-per-QP kernel); carry tails with sticky keys.
o=gpurun_out/r05c; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "$F" | tail -25) > ${o}_pytest.log 2>&1
tail -14 ${o}_pytest.log
(time timeout 600 python bench.py --no-cpu-baseline --pmc off > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
python - <<PY
import json
for f in ("${o}_bench_n1.json",):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"])
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.4g" % v["value"], v.get("kkt_solves_mean"), v.get("kkt_solves_max"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python - <<PY 2>&1 | grep -v "$F" | tee ${o}_smoothers_long.txt
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, tension_inputs
dev = torch.device("cuda", 0)
batch = 256
p = lambda t: capi.C.c_void_p(t.data_ptr())
for n in (384, 500, 700, 1000):
    cases = [tension_inputs(n, seed=b) for b in range(16)]
    rep = lambda k: torch.from_numpy(np.stack([cases[b % 16][k] for b in range(batch)])).to(dev)
    x, y, ang, kk, s, cl = (rep(k) for k in range(6))
    pc = [post_inputs(n, seed=b) for b in range(16)]
    ps, plb, pub = (torch.from_numpy(np.stack([pc[b % 16][j] for b in range(batch)])).to(dev) for j in range(3))
    pl0 = torch.from_numpy(np.array([pc[b % 16][3] for b in range(batch)])).to(dev)
    ox, oy, os_ = (torch.zeros((batch, n), dtype=torch.float64, device=dev) for _ in range(3))
    st = torch.zeros(batch, dtype=torch.int32, device=dev); it = torch.zeros(batch, dtype=torch.int32, device=dev)
    info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
    h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25, polish_refine_iter=2), device=0, max_batch=batch, max_n=n)
    lib = h.lib
    runs = {"tension (S2)": lambda: lib.pqp_smooth_tension_device(h._h, batch, n, p(x), p(y), p(ang), p(cl), p(ox), p(oy), p(os_), p(st), p(it), p(info)),
            "post (S3)": lambda: lib.pqp_post_smooth_device(h._h, batch, n, p(ps), p(plb), p(pub), p(pl0), p(ox), p(st), p(it), p(info))}
    for name, fn in runs.items():
        for _ in range(2): assert fn() == 0
        h.sync(); t0 = time.perf_counter()
        for _ in range(5): assert fn() == 0
        h.sync(); dt = (time.perf_counter() - t0) / 5
        print(f"exact {name:13s} n = {n:5d} batch {batch}: {dt * 1e3:8.3f} ms = {batch / dt / 1e3:8.1f} k scenarios/s; solved {(st == 1).sum().item()}/{batch}; factorisations mean {info[:, 5].mean().item():.1f} max {info[:, 5].max().item():.0f}")
    h.close()
PY
