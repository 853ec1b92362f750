#!/usr/bin/env python
"""bench.py — paths/sec of the batched path-QP hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic scenarios: for every QP of the batch
assemble -> cold ADMM solve -> unpack -> re-linearise -> warm ADMM re-solve -> unpack
(PathOptimizer::optimizePath, reference src/path_optimizer.cpp:124-161), inputs already resident in HBM.
Default workload = BASELINE.json configs[1]: batch 1024 QPs, N = 80, one GPU.  With --gpus N (launched by
torch.distributed.run) every rank solves its own 1024-QP shard (weak scaling, no data-path collective; the
only collective is the result gather over RCCL, outside the timed region's critical path is NOT assumed: it
is inside it when --gather is given).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(n, iters_per_qp):
    """B_path = 2*B_io + 2*B_asm + iters*B_iter (fp64, streaming model),
    B_io = 152 N + 40, B_asm = 656 N, B_iter = 1040 N bytes."""
    b_io = 152 * n + 40
    b_asm = 656 * n
    b_iter = 1040 * n
    return float(np.sum(2 * b_io + 2 * b_asm + np.asarray(iters_per_qp, dtype=np.float64) * b_iter))


def cpu_baseline(batch_np, n, eps, budget_s=20.0):
    """The oracle (C restatement of the OSQP-paper algorithm, oracle/pqp_oracle.c) timed on this box's host
    cores over a bounded sample of the same workload.  kind = "port": OSQP itself is not in this image."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import pqp_oracle_c as OC
    except Exception as e:      # C oracle not built: report the (much slower) numpy restatement on 4 paths
        import pqp_oracle as O
        t0 = time.perf_counter()
        k = 4
        for q in range(k):
            O.solve_path(batch_np["ref"][q], batch_np["bounds"][q], batch_np["scal"][q], st=O.OsqpSettings(eps_abs=eps, eps_rel=eps))
        dt = time.perf_counter() - t0
        return {"value": k / dt, "unit": "paths/s", "cores": 1, "kind": "port",
                "sample": f"{k} paths N={n} numpy restatement ({e.__class__.__name__}: C oracle unavailable)"}
    return OC.timed_baseline(batch_np, n, eps, budget_s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="QPs per GPU")
    ap.add_argument("--n", type=int, default=80, help="waypoints per path")
    ap.add_argument("--eps", type=float, default=1e-4, help="eps_abs = eps_rel of the ADMM termination test")
    ap.add_argument("--no-polish", action="store_true", help="plain OSQP termination (the reference setting), no polish")
    ap.add_argument("--rho-interval", type=int, default=25, help="adaptive_rho_interval (iterations)")
    ap.add_argument("--polish-every", type=int, default=25, help="also try the KKT-verified polish every k ADMM iterations")
    ap.add_argument("--profile", default="uniform", choices=["uniform", "varied"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="all_gather the result slabs over RCCL inside the timed region")
    args = ap.parse_args()

    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    batch, n = args.batch, args.n
    host = make_batch(batch, n, args.profile, first_qp=rank * batch)          # this rank's shard
    ref = torch.from_numpy(host["ref"]).to(dev)
    bounds = torch.from_numpy(host["bounds"]).to(dev)
    scal = torch.from_numpy(host["scal"]).to(dev)
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
    status = torch.zeros(batch, dtype=torch.int32, device=dev)
    iters = torch.zeros(batch, dtype=torch.int32, device=dev)
    info = torch.zeros((batch, 6), dtype=torch.float64, device=dev)
    gathered = [torch.empty_like(out) for _ in range(world)] if (args.gather and world > 1) else None

    prm = capi.default_params(eps_abs=args.eps, eps_rel=args.eps, polish=0 if args.no_polish else 1,
                              polish_every=args.polish_every, adaptive_rho_interval=args.rho_interval,
                              polish_warm_set=0 if args.no_polish else 1, polish_refine_iter=3)
    h = capi.Handle(prm, device=local_rank, max_batch=batch, max_n=n)

    def step():
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=status, iters=iters, info=info)
        if gathered is not None:
            h.sync()
            dist.all_gather(gathered, out)

    for _ in range(args.warmup):
        step()
    h.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)
    h.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    # per-launch kernel duration from HIP events recorded on the handle's stream (last launch) + a second,
    # event-timed sweep for the average (events on the launch stream, not torch's current stream)
    ev_ms = []
    for _ in range(min(args.steps, 10)):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=status, iters=iters, info=info)
        ev_ms.append(h.last_kernel_ms())
    h.sync()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    it_np = iters.cpu().numpy()
    info_np = info.cpu().numpy()
    kkt_np = info_np[:, 5]
    st_np = status.cpu().numpy()
    solved = int((st_np == 1).sum())
    total_paths = batch * world * args.steps
    value = total_paths / dt
    avg_kernel_s = float(np.mean(ev_ms)) * 1e-3
    abytes = algorithmic_bytes(n, kkt_np)      # every reduced-KKT solve (ADMM iterations + polish refinement) moves B_iter
    achieved = abytes / avg_kernel_s / 1e9
    if rank == 0:
        line = {
            "metric": "paths/sec (QP solves/sec) at N=80 waypoints; ADMM iters to 1e-4",
            "value": value, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={batch} QPs/GPU, N={n}, shared sparsity, synthetic obstacle bounds ({args.profile})",
                       "batch_per_gpu": batch, "n_waypoints": n, "eps_abs": args.eps, "eps_rel": args.eps, "polish": not args.no_polish, "polish_every": args.polish_every, "polish_warm_set": not args.no_polish, "polish_refine_iter": 3, "adaptive_rho_interval": args.rho_interval,
                       "passes": "cold solve + 1 re-linearised warm re-solve (optimizePath)",
                       "parallelism": f"{world} x independent shards" + (", RCCL all_gather of results" if gathered is not None else "")},
            "admm_iters": {"min": int(it_np.min()), "median": float(np.median(it_np)), "p99": float(np.percentile(it_np, 99)),
                           "max": int(it_np.max()), "mean": float(it_np.mean())},
            "kkt_solves": {"mean": float(kkt_np.mean()), "max": float(kkt_np.max())},
            "polished_frac": float((info_np[:, 4] >= 2).mean()),
            "solved": solved, "batch": batch,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "path_solve_kernel", "kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": abytes,
                         "note": "streaming-model bytes (SURVEY.md 8d); iterates are register/LDS resident, see DESIGN.md"},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host, n, args.eps)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
