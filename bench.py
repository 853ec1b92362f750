#!/usr/bin/env python
"""bench.py — paths/sec of the batched path-QP hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic scenarios: for every QP of the batch
    assemble -> cold ADMM solve -> unpack -> re-linearise -> warm re-solve -> unpack
(PathOptimizer::optimizePath, reference src/path_optimizer.cpp:124-161), inputs already resident in HBM.

Default workload = BASELINE.json configs[1]: batch 1024 QPs, N = 80, one GPU, solved to the engine's production
setting: ADMM to eps_abs = eps_rel = 1e-4 with the KKT-verified polish (every returned path is the exact QP optimum,
i.e. inside the 1e-4 parity bar; tests/test_gpu_parity.py).  `--no-polish` runs the plain OSQP termination instead.

With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank generates and solves its own
contiguous shard of `--batch` QPs (weak scaling: per-GPU work fixed).  The QPs are independent, so the timed region has
no data-path collective: barrier + synchronize on both sides, MAX over ranks of the elapsed time.  After the timing one
RCCL all_gather of the result slabs (path_optimizer_2_amd/shard.py) checks the gather a caller would use.  Rank 0 prints
ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)


def algorithmic_bytes(n, kkt_solves, factors, setups):
    """Streaming-model bytes of SURVEY.md §8(d), fp64, default flags:
         B_io   = 152 N + 40   per solve   (read 12 doubles/waypoint + 5 scalars, write 7 doubles/waypoint)
         B_asm  = 656 N        per solve   (assembled P, A, l, u, q written once and read once)
         B_iter = 1040 N       per reduced-KKT solve (band factor 48, A values 2x17, x/z/y 18 + 18, l/u 12 doubles/waypoint)
       plus the two terms §8(d) leaves out because OSQP refactors at most ~3 times but the polish refactors every
       active-set round:
         B_fac  = 752 N        per factorisation (read A 17, rho 6, sigma 6, P 6; write the band 48, rho 11 doubles/waypoint)
         B_ruiz = 3440 N       per setup   (10 equilibration passes x (read A 17, P 6, D 6, E 6; write D 6, E 2) doubles/waypoint)
    """
    b_io, b_asm, b_iter, b_fac, b_ruiz = 152 * n + 40, 656 * n, 1040 * n, 752 * n, 3440 * n
    base = float(np.sum(setups * (b_io + b_asm) + np.asarray(kkt_solves, dtype=np.float64) * b_iter))
    ext = base + float(np.sum(np.asarray(factors, dtype=np.float64) * b_fac + setups * b_ruiz))
    return base, ext


def cpu_baseline(make_sample, n, eps, budget_s):
    """The oracle (C restatement of the OSQP-paper algorithm, oracle/pqp_oracle.c) timed on this box's host cores over
    a bounded sample of the same workload.  kind = "port": OSQP itself is not in this image."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pqp_oracle_c as OC
    return OC.timed_baseline(make_sample, n, eps, budget_s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="QPs per GPU")
    ap.add_argument("--n", type=int, default=80, help="waypoints per path")
    ap.add_argument("--eps", type=float, default=1e-4, help="eps_abs = eps_rel of the ADMM termination test")
    ap.add_argument("--no-polish", action="store_true", help="plain OSQP termination (the reference setting), no polish")
    ap.add_argument("--rho-interval", type=int, default=15, help="adaptive_rho_interval (iterations)")
    ap.add_argument("--polish-every", type=int, default=15, help="also try the KKT-verified polish every k ADMM iterations")
    ap.add_argument("--polish-refine", type=int, default=2, help="refinement solves per active-set round of the polish")
    ap.add_argument("--polish-max-rounds", type=int, default=0, help="active-set rounds before a polish attempt gives up (0: max(24, n/5 - 8))")
    ap.add_argument("--seed", type=int, default=None, help="seed of the synthetic scenarios (default: synth.BASE_SEED)")
    ap.add_argument("--rho-tolerance", type=float, default=2.0, help="adaptive_rho_tolerance")
    ap.add_argument("--polish-warm-set", type=int, default=2, help="1: pass 2 starts with a polish on pass 1's active set; 2: and keeps its equilibration")
    ap.add_argument("--check-termination", type=int, default=15, help="residual check interval (iterations)")
    ap.add_argument("--inflight", type=int, default=1, help="batches in flight: k > 1 runs consecutive steps on k handles / HIP streams "
                    "(what a server does with independent batches; the next batch fills the slots the slow tail of this one leaves idle)")
    ap.add_argument("--profile", default="uniform", choices=["uniform", "varied"])
    ap.add_argument("--reference-setting", action="store_true", help="the reference's solver setting instead of the production one: "
                    "pqp_default_params (OSQP defaults, no polish, infeasibility certificate on) at --eps (the reference runs 2e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of host CPU work for the cpu_baseline sample")
    args = ap.parse_args()

    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.shard import gather_paths
    from path_optimizer_2_amd.synth import BASE_SEED, make_batch
    if args.seed is None:
        args.seed = BASE_SEED

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    batch, n = args.batch, args.n
    total = batch * world
    host = make_batch(batch, n, args.profile, seed=args.seed, first_qp=rank * batch)          # this rank's shard of the global batch
    ref = torch.from_numpy(host["ref"]).to(dev)
    bounds = torch.from_numpy(host["bounds"]).to(dev)
    scal = torch.from_numpy(host["scal"]).to(dev)
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
    status = torch.zeros(batch, dtype=torch.int32, device=dev)
    iters = torch.zeros(batch, dtype=torch.int32, device=dev)
    info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)

    polish = not args.no_polish and not args.reference_setting
    prm = capi.default_params(eps_abs=args.eps, eps_rel=args.eps) if args.reference_setting else capi.production_params(eps_abs=args.eps, eps_rel=args.eps, polish=1 if polish else 0, polish_every=args.polish_every,
                              adaptive_rho_interval=args.rho_interval, polish_warm_set=args.polish_warm_set if polish else 0, check_termination=args.check_termination, polish_refine_iter=args.polish_refine,
                              polish_max_rounds=args.polish_max_rounds, adaptive_rho_tolerance=args.rho_tolerance)
    h = capi.Handle(prm, device=local_rank, max_batch=batch, max_n=n)
    # --inflight k: k - 1 more handles (own stream, own warm state, own outputs) used round-robin
    extra = [(capi.Handle(prm, device=local_rank, max_batch=batch, max_n=n), torch.zeros_like(out), torch.zeros_like(status),
              torch.zeros_like(iters), torch.zeros_like(info)) for _ in range(max(args.inflight, 1) - 1)]
    lanes = [(h, out, status, iters, info)] + extra
    counter = [0]

    def step():
        hh, o, st, it, inf = lanes[counter[0] % len(lanes)]
        counter[0] += 1
        hh.solve_device(batch, n, ref, bounds, scal, o, passes=1, status=st, iters=it, info=inf)

    def sync_all():
        for hh, *_ in lanes:
            hh.sync()

    for _ in range(args.warmup):
        step()
    sync_all()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gathered_ok = None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        full = gather_paths(out, total)              # untimed: what a caller that wants every path on every rank would do
        gathered_ok = bool(full.shape[0] == total and torch.equal(full[rank * batch:(rank + 1) * batch], out))

    # per-launch duration of the dominant kernel over the timed region: HIP events the handle recorded around every launch on the
    # stream it launched on, read back now (nothing was synchronised between the launches)
    ev_ms = []
    for li, (hh, *_) in enumerate(lanes):
        k = min(sum(1 for i in range(args.steps) if (args.warmup + i) % len(lanes) == li), 256)      # timed launches of this handle
        if k > 0:
            ev_ms.extend(hh.kernel_ms_history(k).tolist())

    it_np = iters.cpu().numpy()
    st_np = status.cpu().numpy()
    info_np = info.cpu().numpy()
    kkt_np, fac_np = info_np[:, 5], info_np[:, 6]
    setups = 2.0
    avg_kernel_s = float(np.mean(ev_ms)) * 1e-3
    abytes, abytes_ext = algorithmic_bytes(n, kkt_np, fac_np, setups)
    achieved = abytes / avg_kernel_s / 1e9
    # measured HBM bytes per launch come from separate rocprofv3 --pmc passes of THIS command (FETCH_SIZE, WRITE_SIZE; the
    # gfx950 x2 correction of FETCH_SIZE for wide reads applied; MI355X_MICROARCH.md HBM section), committed under profiles/
    traffic, issue = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r01_bench_n1_pmc_per_launch.json")
    if world == 1 and batch == 1024 and n == 80 and polish and os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            # what does bound the kernel (SURVEY.md 8d "secondary ceilings"): share of the wave cycles spent issuing fp64 VALU /
            # LDS instructions and waiting, from the same PMC passes
            wc = pmc["SQ_WAVE_CYCLES"]
            issue = {"valu_active_frac": pmc["SQ_ACTIVE_INST_VALU"] / wc, "lds_active_frac": pmc["SQ_ACTIVE_INST_LDS"] / wc,
                     "any_active_frac": pmc["SQ_ACTIVE_INST_ANY"] / wc, "wait_frac": pmc["SQ_WAIT_ANY"] / wc, "waves_per_simd": 1}
        except Exception:
            traffic = None
    if rank == 0:
        line = {
            "metric": "paths/sec (QP solves/sec) at N=80 waypoints; ADMM iters to 1e-4",
            "value": total * args.steps / dt, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={batch} QPs per GPU, N={n}, shared sparsity, synthetic obstacle bounds ({args.profile} profile)",
                       "batch_per_gpu": batch, "n_waypoints": n, "eps_abs": args.eps, "eps_rel": args.eps, "polish": polish,
                       "setting": "reference (pqp_default_params)" if args.reference_setting else "production (pqp_production_params)",
                       "polish_every": args.polish_every if polish else 0, "adaptive_rho_interval": 100 if args.reference_setting else args.rho_interval,
                       "polish_refine_iter": args.polish_refine, "polish_max_rounds": args.polish_max_rounds, "polish_warm_set": args.polish_warm_set,
                       "passes": "cold solve + 1 re-linearised warm re-solve (PathOptimizer::optimizePath)",
                       "parallelism": f"{world} independent shard(s), no collective in the timed region", "batches_in_flight": max(args.inflight, 1)},
            "admm_iters": {"min": int(it_np.min()), "median": float(np.median(it_np)), "p99": float(np.percentile(it_np, 99)),
                           "max": int(it_np.max()), "mean": float(it_np.mean())},
            "kkt_solves": {"mean": float(kkt_np.mean()), "p99": float(np.percentile(kkt_np, 99)), "max": float(kkt_np.max())},
            "factorisations": {"mean": float(fac_np.mean()), "max": float(fac_np.max())},
            "gather_check": gathered_ok, "solved": int((st_np == 1).sum()), "polished": int((info_np[:, 4] >= 2).sum()), "batch": batch,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "path_solve_kernel", "kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": abytes, "issue": issue,
                         "achieved_incl_factor_and_scaling": abytes_ext / avg_kernel_s / 1e9,
                         "note": "SURVEY.md 8(d) streaming-model bytes; the iterates are register/LDS resident, so measured HBM "
                                 "traffic is far below this (profiles/, DESIGN.md 5)"},
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(lambda k: make_batch(k, n, args.profile, seed=args.seed), n, args.eps, args.cpu_budget)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
