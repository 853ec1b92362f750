#!/usr/bin/env python
"""bench.py — paths/sec of the batched path-QP hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic scenarios: for every QP of the batch
    assemble -> cold ADMM solve -> unpack -> re-linearise -> warm re-solve -> unpack
(PathOptimizer::optimizePath, reference src/path_optimizer.cpp:124-161), inputs already resident in HBM.

Workloads = BASELINE.json configs (`--config`; explicit --batch / --n / --profile override a preset's shape):
    1  batch 1024, N = 80, shared sparsity, synthetic obstacle bounds, one GPU                  (default at --gpus 1)
    2  batch 8192, N = 120, varied start / goal / curvature limits, one GPU
    3  batch 65 536, N = 80 over 8 GPUs = 8192 QPs per GPU                                        (default at --gpus > 1)
    4  batch 4096, N = 200 over 8 GPUs = 512 per GPU: TensionSmoother2 QP + path QP on two HIP streams (pipeline.py)
Two batches are kept in flight by default (--inflight 2: consecutive steps alternate between two handles / HIP streams; every step is
still one pass over one whole batch, and K steps are timed); --inflight 1 and its figure in "secondary" = one launch strictly after the other.
Consecutive steps solve SIMILAR, not identical, batches (--variants 8: the scenarios one planning cycle later, synth.jitter_batch: corridor
sides and start state scaled by 1 +- 5 %), so that the most-expensive-first start order (previous step's cost) has no perfect foresight;
`secondary.identical_batch_every_step` is round 2's headline (the same batch re-solved every step).
Batches of at least pqp_stream_batch_default(N) QPs (PQP_OPT_STREAM_BATCH: the measured crossover, 15 360 at N = 80) run on the lane-per-QP kernel (path_stream_kernel: HBM-streaming, its roofline is
measured traffic): `--config 3 --batch 65536` is configs[3]'s whole batch on one GPU, timed in every default run under
`secondary.configs3_whole_batch_one_gpu_stream_kernel`.
Solver setting: the engine's production setting (pqp_production_params: ADMM to eps 1e-4 + KKT-verified polish — every returned path
is the exact optimum of its QP, inside the 1e-4 parity bar).  The literal metric ("ADMM iters to 1e-4", plain OSQP termination, no
polish) and the reference's own setting (eps 2e-3) are timed in the same run and reported under "secondary".

With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank generates and solves its own contiguous shard (weak
scaling: per-GPU work fixed).  The QPs are independent, so the timed region has no data-path collective: barrier + synchronize on both
sides, MAX over ranks of the elapsed time.  After the timing one RCCL all_gather of the result slabs (shard.py) checks the gather a
caller would use.  Rank 0 prints ONE JSON line.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
SIMDS = 256 * 4            # 256 CUs x 4 SIMDs
FP64_CYCLES_PER_VALU = 4   # a wave64 fp64 VALU instruction occupies its SIMD for 4 cycles (78.6 TFLOP/s fp64 vector peak / 2 flop /
                           # 1024 SIMDs / 2.4 GHz = 16 lanes per cycle); SQ_ACTIVE_INST_* count quad-cycles (MICROARCH guide)

CONFIGS = {
    1: dict(batch=1024, n=80, profile="uniform", what="configs[1]: batch=1024 QPs, N=80, shared sparsity, synthetic obstacle bounds, 1 GPU"),
    2: dict(batch=8192, n=120, profile="varied", what="configs[2]: batch=8192 QPs, N=120, varied start/goal + curvature limits, 1 GPU"),
    3: dict(batch=8192, n=80, profile="uniform", what="configs[3]: batch=65536 QPs, N=80, sharded over 8 GPUs = 8192 QPs per GPU"),
    4: dict(batch=512, n=200, profile="uniform", what="configs[4]: batch=4096, N=200 over 8 GPUs = 512 scenarios per GPU, TensionSmoother2 QP "
                                                      "+ path QP pipelined on two HIP streams"),
}


def algorithmic_bytes(n, kkt_solves, admm_iters, factors, setups):
    # kkt_solves / admm_iters / factors: one entry per QP of the launch (their length is the batch); setups: assemblies per QP (2)
    """Streaming-model bytes of SURVEY.md §8(d), fp64, default flags:
         B_io   = 152 N + 40   per solve   (read 12 doubles/waypoint + 5 scalars, write 7 doubles/waypoint)
         B_asm  = 656 N        per solve   (assembled P, A, l, u, q written once and read once)
         B_iter = 1040 N       per reduced-KKT solve (band factor 48, A values 2x17, x/z/y 18 + 18, l/u 12 doubles/waypoint)
       Returned: (bytes charging every reduced-KKT solve = ADMM iterations + polish refinement solves, bytes charging ADMM iterations
       only - §8(d)'s literal wording -, bytes with factorisations and Ruiz passes added:
         B_fac  = 752 N per factorisation, B_ruiz = 3440 N per setup)."""
    b_io, b_asm, b_iter, b_fac, b_ruiz = 152 * n + 40, 656 * n, 1040 * n, 752 * n, 3440 * n
    batch = len(np.atleast_1d(kkt_solves))
    fixed = float(batch * setups * (b_io + b_asm))              # (round 2 forgot `batch` here: 132 MB per 1024 x 80 launch)
    with_kkt = fixed + float(np.sum(np.asarray(kkt_solves, dtype=np.float64) * b_iter))
    admm_only = fixed + float(np.sum(np.asarray(admm_iters, dtype=np.float64) * b_iter))
    ext = with_kkt + float(np.sum(np.asarray(factors, dtype=np.float64) * b_fac)) + float(batch * setups * b_ruiz)
    return with_kkt, admm_only, ext


# bytes per waypoint of one sweep of path_stream_kernel (fields read + written, fp64 problem data / gains / point,
# fp32 interior-point state); "1": a pass linearised around (0, 0, k_ref) streams 3 of the 8 transition doubles, "2": a re-linearised pass all 8
STREAM_BYTES = {"prep1": 184, "init": 232, "ipm1": 368, "ipm2": 448, "guess1": 172, "guess2": 212, "fset1": 144, "fset2": 184,
                "bset1": 104, "bset2": 144, "prep2": 104, "warm": 128, "unpack": 136, "stash": 80}


STREAM_DIRECT_ROUNDS = 3        # lq::kDirectRounds (csrc/pqp_path_lq.hpp)


def stream_direct_rounds(info):
    """Was this a sorted launch of the lane-per-QP kernel (its re-linearised passes began with active-set rounds on the first pass's set)?  Only there does a QP
    end its second pass without an interior-point iteration: a pass started from the previous optimum runs at least one."""
    return STREAM_DIRECT_ROUNDS if ((info[:, 4] >= 2) & (info[:, 3] - info[:, 2] == 0)).any() else 0


def stream_algorithmic_bytes(n, info, direct=0):
    """Workspace + I/O bytes the lane-per-QP kernel's algorithm moves for the sweeps each QP actually ran (info rows of pqp_path_solve:
    [2] interior-point iterations of the first pass, [3] of both, [5] active-set rounds of the first pass, [7] of both).  Every byte is HBM
    (or Infinity-Cache) traffic by construction: nothing is kept on chip between two sweeps.  A wavefront also moves the lines of lanes that
    have finished while a neighbour in the same 128-byte line has not, so measured traffic is 1.0-1.35x this.
    direct = k > 0: a sorted launch (PQP_OPT_ORDER_BY_COST with a map) - the re-linearised pass keeps the first pass's optimum aside ("stash": 40 bytes
    read + 40 written per waypoint) and runs up to k active-set rounds on its set (one backward + one forward sweep each); a QP whose set was
    confirmed ran no interior-point iteration in that pass, the others ran k rounds and then the pass of an unsorted launch."""
    B = STREAM_BYTES
    it1, it = info[:, 2], info[:, 3]
    s1, st = info[:, 5], info[:, 7]
    it2, s2 = it - it1, st - s1
    two = (info[:, 4] >= 2) | (it2 > 0) | (s2 > 0)
    first = B["prep1"] + B["init"] + it1 * B["ipm1"] + B["guess1"] + s1 * B["fset1"] + np.maximum(s1 - 1, 0) * B["bset1"] + B["unpack"]
    from_ipm = lambda rounds: B["warm"] + B["guess2"] + it2 * B["ipm2"] + rounds * B["fset2"] + np.maximum(rounds - 1, 0) * B["bset2"]
    if direct > 0:
        hit = two & (it2 == 0)
        second = B["prep2"] + B["stash"] + np.where(hit, s2 * (B["bset2"] + B["fset2"]), direct * (B["bset2"] + B["fset2"]) + from_ipm(np.maximum(s2 - direct, 0)))
    else:
        second = B["prep2"] + from_ipm(s2)
    return float(np.sum(first + two * second) * n)


def fetch_calibration():
    """Factors that turn rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB) into bytes for THIS library's access patterns, measured by
    tools/fetch_calib.py on a known byte count (tools/probes/fetch_calib.hip; committed record profiles/fetch_calib.json).  Without a
    record: the MI355X guide's x2 for reads (its figure for 16 B/lane streaming reads) and x1 for writes, labelled as such."""
    path = os.path.join(ROOT, "profiles", "fetch_calib.json")
    try:
        f = json.load(open(path))["factors"]
        stream_rd, stream_wr = f.get("sweep_like_read") or f["read_8B_per_lane"], f.get("sweep_like_write") or f["write_8B_per_lane"]
        return {"source": "profiles/fetch_calib.json (tools/fetch_calib.py: known bytes / reported bytes, 4 GiB footprint)",
                "stream_read": float(stream_rd), "stream_write": float(stream_wr),          # path_stream_kernel: mixed 8 B / 4 B per lane lines
                "lane_read": float(f["read_8B_per_lane"]), "lane_write": float(f["write_8B_per_lane"]), "all": f}      # path_solve_kernel: 8 B per lane
    except Exception:
        return {"source": "uncalibrated: MI355X_MICROARCH.md's x2 for wide streaming reads, x1 for writes (no profiles/fetch_calib.json)",
                "stream_read": 2.0, "stream_write": 1.0, "lane_read": 2.0, "lane_write": 1.0, "all": None}


def shim_batch1(n_list=(60, 80), cycles=40):
    """The reference's own call pattern timed through the C++ drop-in (tests/cpp/shim_latency.cpp over csrc/base_solver_shim.cpp): one
    BaseSolver per planning cycle at batch 1 - construct -> solve() -> updateProblemFormulationAndSolve() -> destruct
    (src/path_optimizer.cpp:138-153).  Microseconds per cycle with the shim's handle pool on and off, reference setting and production
    setting.  Never raises."""
    try:
        from path_optimizer_2_amd.synth import make_batch
        csrc = os.path.join(ROOT, "path_optimizer_2_amd", "csrc")
        exe = os.path.join(tempfile.gettempdir(), "pqp_shim_latency")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "shim_latency.cpp"), os.path.join(csrc, "base_solver_shim.cpp"),
                        "-L" + csrc, "-lpqp_hip", "-Wl,-rpath," + csrc], check=True, capture_output=True, timeout=300)
        res = {"what": "one BaseSolver per planning cycle at batch 1 (path_optimizer.cpp:138-153): construct -> solve -> updateProblemFormulationAndSolve -> destruct, "
                       "host-pointer entry points (PCIe copies included), microseconds per cycle; first_cycle_us pays pqp_create"}
        for n in n_list:
            b = make_batch(1, n)
            lines = [str(n)] + [" ".join(repr(float(v)) for v in list(b["ref"][0, i]) + list(b["bounds"][0, i])) for i in range(n)] + [" ".join(repr(float(v)) for v in b["scal"][0])]
            text = "\n".join(lines) + "\n"
            for key, cache, polish in ((f"n{n}_reference_setting", 1, 0), (f"n{n}_reference_setting_no_handle_pool", 0, 0), (f"n{n}_production_setting", 1, 1)):
                r = subprocess.run([exe, str(cycles), str(cache), str(polish)], input=text, capture_output=True, text=True, timeout=300)
                res[key] = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-300:]}
        return res
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline(make_sample, n, eps, budget_s):
    """The oracle (C restatement of the OSQP-paper algorithm, oracle/pqp_oracle.c) timed on this box's host cores over
    a bounded sample of the same workload.  kind = "port": OSQP itself is not in this image.  Beside it, as a second CPU line, the
    engine's OWN lane-per-QP algorithm (csrc/pqp_path_lq.hpp) compiled for the host by the test infrastructure (tests/emu): what the
    same arithmetic does on the box's cores - a baseline measurement, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pqp_oracle_c as OC
    base = OC.timed_baseline(make_sample, n, eps, 0.7 * budget_s)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lq_emu_util as LE
        base["same_algorithm_on_host"] = LE.timed_rate(make_sample, n, 0.3 * budget_s)
    except Exception as e:
        base["same_algorithm_on_host"] = {"error": f"{type(e).__name__}: {e}"}
    # a third line, for scale: a third-party production QP solver on the reference's QPs (HiGHS's active-set QP solver as bundled with scipy,
    # oracle/highs_qp.py - the solver tests/test_highs_pin.py pins the optimum against), one thread, both passes of optimizePath, exact optimum
    try:
        import highs_qp as HQ
        import pqp_oracle as ON
        if HQ.available():
            smp = make_sample(6)
            t0 = time.perf_counter()
            for q in range(6):
                lin = ON.first_linearization(smp["ref"][q])
                for _ in range(2):
                    Pd, A, lo, up, sz = ON.assemble_path_qp(smp["ref"][q], lin, smp["bounds"][q], smp["scal"][q])
                    xq, _, _ = HQ.solve_qp(Pd, np.zeros(sz["vars"]), A, lo, up)
                    lin = ON.unpack_path(xq, smp["ref"][q])[:, 3:6].copy()
            dtq = time.perf_counter() - t0
            base["third_party_qp_solver_single_thread"] = {"value": 6 / dtq, "unit": "paths/s", "cores": 1, "solver": HQ.version() + ", active-set QP solver",
                                                           "sample": f"6 paths (N={n}) in {dtq:.1f} s: the oracle's line-by-line assembly of the reference's QP (numpy, included in the time) + HiGHS, "
                                                                     "two passes per path, exact optimum"}
    except Exception as e:
        base["third_party_qp_solver_single_thread"] = {"error": f"{type(e).__name__}: {e}"}
    return base


def pmc_child(argv_core, kernel_substr, timeout_s, steps=6, passes=None):
    """Hardware counters of THIS command's dominant kernel, measured now: three rocprofv3 --pmc passes (SQ counters; the TCC byte counters
    FETCH_SIZE / WRITE_SIZE in a pass each: they do not fit one) of a short child run of bench.py, --kernel-trace only
    beside --pmc (MI355X_MICROARCH.md, rocprofv3 PMC section).  Returns per-launch averages or None (never raises)."""
    passes = passes or ["SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU",
                        "FETCH_SIZE GRBM_GUI_ACTIVE", "WRITE_SIZE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"]
    acc, cnt = {}, {}
    try:
        for pmc in passes:
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                cmd = ["rocprofv3", "--kernel-trace", "--pmc", *pmc.split(), "-f", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                       *argv_core, "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-secondary", "--pmc", "off", "--sustain", "0",
                       "--inflight", "1", "--prewarm", "0"]          # (one kernel at a time: the counters of a launch are its own)
                env = dict(os.environ, TMPDIR="/tmp")
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if kernel_substr in row["Kernel_Name"]:
                            c = row["Counter_Name"]
                            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
                            cnt[c] = cnt.get(c, 0) + 1
        if not acc:
            return None
        res = {c: acc[c] / cnt[c] for c in acc}
        res["_launches"] = min(cnt.values())
        return res
    except Exception as e:      # rocprofv3 missing, counters unavailable, time-out: the bench line simply carries nulls
        sys.stderr.write(f"[bench] pmc pass skipped: {type(e).__name__}: {e}\n")
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (default: 0.65 s of configs[1] steps, so that the timed region is visible to a 10 Hz sampler)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm", type=float, default=0.25, help="seconds of untimed steps before the W warm-up steps (clocks, start order); 0: none")
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4], help="BASELINE.json configs[k] (default: 1 at --gpus 1, else 3)")
    ap.add_argument("--batch", type=int, default=None, help="QPs per GPU (default: the config's)")
    ap.add_argument("--total", type=int, default=None, help="QPs of the WHOLE job, split contiguously over the ranks (shard.shard_range: the first total %% N ranks get "
                    "one more - ragged shards); overrides --batch; the line then says \"scaling\": \"strong\" (the total stays as N grows)")
    ap.add_argument("--n", type=int, default=None, help="waypoints per path (default: the config's)")
    ap.add_argument("--profile", default=None, choices=["uniform", "varied"])
    ap.add_argument("--eps", type=float, default=1e-4, help="eps_abs = eps_rel of the ADMM termination test")
    ap.add_argument("--no-polish", action="store_true", help="plain OSQP termination, no polish")
    ap.add_argument("--rho-interval", type=int, default=None, help="adaptive_rho_interval (iterations; default: the production setting's: 5 up to 90 waypoints, 8 beyond)")
    ap.add_argument("--polish-every", type=int, default=None, help="also try the KKT-verified polish every k ADMM iterations (default: 5 up to 90 waypoints, 8 beyond)")
    ap.add_argument("--polish-lazy", type=int, default=None, help="rounds of a polish attempt that move rows after one solve (default: the production setting's 5)")
    ap.add_argument("--polish-refine", type=int, default=2, help="refinement solves per active-set round of the polish")
    ap.add_argument("--polish-max-rounds", type=int, default=0, help="active-set rounds before a polish attempt gives up (0: max(24, n/5 - 8))")
    ap.add_argument("--scaling", type=int, default=None, help="Ruiz equilibration passes (default: the production setting's)")
    ap.add_argument("--seed", type=int, default=None, help="seed of the synthetic scenarios (default: synth.BASE_SEED)")
    ap.add_argument("--rho-tolerance", type=float, default=2.0, help="adaptive_rho_tolerance")
    ap.add_argument("--rho", type=float, default=None, help="initial ADMM step size rho (default: the production setting's 0.1)")
    ap.add_argument("--polish-warm-set", type=int, default=2, help="1: pass 2 starts with a polish on pass 1's active set; 2: and keeps its equilibration")
    ap.add_argument("--check-termination", type=int, default=None, help="residual check interval (iterations; default: 5 up to 90 waypoints, 8 beyond)")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight: consecutive steps (independent batches) go round-robin to k handles / HIP "
                    "streams, as a planning server keeps independent batches in flight: the next batch's QPs fill the slots the slow tail of "
                    "this one leaves idle.  1 = strictly one launch after the other (reported under secondary.one_batch_at_a_time)")
    ap.add_argument("--variants", type=int, default=8, help="input variants cycled through by consecutive steps: variant v = the batch one planning cycle "
                    "later (synth.jitter_batch: corridor sides and start state scaled by 1 +- 5 %%); 1: the identical batch every step")
    ap.add_argument("--no-cost-order", action="store_true", help="start the QPs of a batch in index order instead of most-expensive-first by "
                    "their cost in the previous step (PQP_OPT_ORDER_BY_COST)")
    ap.add_argument("--reference-setting", action="store_true", help="the reference's solver setting instead of the production one: "
                    "pqp_default_params (OSQP defaults, no polish, infeasibility certificate on) at --eps (the reference runs 2e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the plain-ADMM (literal metric) and index-order measurements")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of host CPU work for the cpu_baseline sample")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"], help="auto (1 GPU only): hardware counters of the dominant kernel from "
                    "rocprofv3 --pmc child runs of this command, after the timed region")
    ap.add_argument("--pmc-timeout", type=float, default=150.0)
    ap.add_argument("--sustain", type=float, default=0.6, help="seconds of back-to-back steps timed after the K steps (reported as `sustained`; 0: skip)")
    args = ap.parse_args()

    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.shard import gather_paths
    from path_optimizer_2_amd.synth import BASE_SEED, jitter_batch, make_batch
    if args.seed is None:
        args.seed = BASE_SEED

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
        # (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...), and hand its exit code on
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stderr.write(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}\n")
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: launch one rank per GPU (or give --gpus alone and let bench.py start them)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # PQP_BENCH_SHARED_GPU=1 (tests only): every rank on GPU 0, collectives over gloo on host copies - the multi-rank code path of this file on a
    # one-GPU box (RCCL refuses two ranks on one device).  Its numbers mean nothing; the line says so.
    shared_gpu = world > 1 and os.environ.get("PQP_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node shows {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    coll = (lambda t: t.cpu()) if shared_gpu else (lambda t: t)          # where a collective's tensors live

    cfg_id = args.config if args.config is not None else (1 if max(world, args.gpus) == 1 else 3)
    cfg = CONFIGS[cfg_id]
    batch = args.batch if args.batch is not None else cfg["batch"]
    n = args.n if args.n is not None else cfg["n"]
    profile = args.profile or cfg["profile"]
    preset_shape = (batch, n, profile) == (cfg["batch"], cfg["n"], cfg["profile"])
    total = batch * world
    first_qp = rank * batch                     # this rank's contiguous shard [first_qp, first_qp + batch) of the job's QPs
    if args.total is not None:
        from path_optimizer_2_amd.shard import shard_range
        if args.total < world:
            raise SystemExit(f"bench.py: --total {args.total} is less than one QP per rank")
        total = args.total
        first_qp, batch = shard_range(total, world, rank)
        preset_shape = False
    polish = not args.no_polish and not args.reference_setting
    cost_order = not args.no_cost_order
    # the handle's default PQP_OPT_STREAM_BATCH (include/pqp.h: the measured crossover of the two kernels, pqp_stream_batch_default):
    # from here on the lane-per-QP kernel runs
    STREAM_BATCH = capi.stream_batch_default(n)
    stream = polish and cfg_id != 4 and batch >= STREAM_BATCH

    def production(**over):
        kw = dict(eps_abs=args.eps, eps_rel=args.eps, polish=1 if polish else 0, polish_warm_set=args.polish_warm_set if polish else 0,
                  polish_refine_iter=args.polish_refine, polish_max_rounds=args.polish_max_rounds, adaptive_rho_tolerance=args.rho_tolerance)
        for key, val in (("scaling", args.scaling), ("adaptive_rho_interval", args.rho_interval), ("polish_every", args.polish_every),
                         ("check_termination", args.check_termination), ("polish_lazy", args.polish_lazy), ("rho", args.rho)):
            if val is not None:
                kw[key] = val
        if not polish:
            kw["polish_every"] = 0
        kw.update(over)
        return capi.production_params(**kw)

    prm = capi.default_params(eps_abs=args.eps, eps_rel=args.eps) if args.reference_setting else production()

    # ---------------------------------------------------------------------------------------------------------------------------
    # the workload: `step()` enqueues one pass of the hot path over one batch; `sync_all()` waits for everything enqueued
    # ---------------------------------------------------------------------------------------------------------------------------
    pipe = None
    if cfg_id == 4:
        from path_optimizer_2_amd.pipeline import SmootherPathPipeline
        # the smoother of the production setting: TensionSmoother2's QP has no inequality rows, so it is solved as ONE KKT system (polish = 2:
        # the exact optimum, no ADMM iterations); --reference-setting: the reference's 25 ADMM iterations to eps 1e-3
        sm_prm = capi.default_params(eps_abs=1e-3, eps_rel=1e-3) if args.reference_setting else capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=2, scaling=0, polish_refine_iter=2)
        pipe = SmootherPathPipeline(batch, n, device=local_rank, seed=rank, path_params=prm, smoother_params=sm_prm)
        if cost_order:
            pipe.hp.set_option(capi.OPT_ORDER_BY_COST, 1)
        counter = [0]

        def step():
            pipe.step_pipelined(counter[0])
            counter[0] += 1

        sync_all = pipe.sync
        main_handles = [pipe.hp]
        out = pipe.buf[0]["out"]
    else:
        host = make_batch(batch, n, profile, seed=args.seed, first_qp=first_qp)          # this rank's shard of the global batch
        ref = torch.from_numpy(host["ref"]).to(dev)
        n_var = max(1, args.variants)
        # variant v: the same scenarios one planning cycle later (corridor sides and start state moved by up to 5 %); the reference line stays
        var_in = []
        for v in range(n_var):
            hv = jitter_batch(host, v, seed=args.seed, first_qp=first_qp)
            var_in.append((torch.from_numpy(hv["bounds"]).to(dev), torch.from_numpy(hv["scal"]).to(dev)))
        bounds, scal = var_in[0]

        def make_lane(p, order):
            h = capi.Handle(p, device=local_rank, max_batch=batch, max_n=n)
            h.set_option(capi.OPT_STORE_WARM, 0)          # every step is a complete optimizePath: nobody reads the warm state
            h.set_option(capi.OPT_ORDER_BY_COST, 1 if order else 0)
            return (h, torch.zeros((batch, n, 7), dtype=torch.float64, device=dev), torch.zeros(batch, dtype=torch.int32, device=dev),
                    torch.zeros(batch, dtype=torch.int32, device=dev), torch.zeros((batch, 8), dtype=torch.float64, device=dev))

        lanes = [make_lane(prm, cost_order) for _ in range(max(args.inflight, 1))]       # --inflight k: k handles used round-robin
        main_handles = [ln[0] for ln in lanes]
        out, status, iters, info = lanes[0][1:]
        counter = [0]

        def step():
            hh, o, st, it, inf = lanes[counter[0] % len(lanes)]
            bv, sv = var_in[counter[0] % n_var]
            counter[0] += 1
            hh.solve_device(batch, n, ref, bv, sv, o, passes=1, status=st, iters=it, info=inf)

        def sync_all():
            for hh, *_ in lanes:
                hh.sync()

    torch.cuda.synchronize()          # the inputs were produced on torch's stream; the handles launch on their own (non-blocking) streams
    # Untimed, before the W warm-up steps: the same step for --prewarm seconds.  A fresh process finds the GPU at its idle clocks and the handles
    # without a start order; W = 3 ... 5 steps are ~1.5 ms of work - the K steps behind them would measure the ramp, not the kernel.
    if args.prewarm > 0:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm:
            for _ in range(16):
                step()
            sync_all()
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
    gather_info = None
    if dist is not None:
        tmax = coll(torch.tensor([dt], dtype=torch.float64, device=dev))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        full = gather_paths(coll(out), total)        # untimed: what a caller that wants every path on every rank would do
        gathered_ok = bool(full.shape[0] == total and torch.equal(full[first_qp:first_qp + batch], coll(out)))
        # ... and what that RCCL all-gather over xGMI costs (outside the timed region; second call: communicators are warm)
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        full = gather_paths(coll(out), total)
        torch.cuda.synchronize(); dist.barrier()
        gather_s = time.perf_counter() - tg
        gather_info = {"ms": gather_s * 1e3, "bytes_per_rank_received": int(full.numel() * 8), "what": "torch.distributed all_gather (RCCL) of every rank's [batch][n][7] result slab "
                       "to every rank, outside the timed region: the only collective of the path (SURVEY.md 8e)"}
        del full
        # the communicator's size as the collective library itself reports it: an all-reduce of ones over RCCL (gloo in the shared-GPU test mode)
        ones = coll(torch.ones(1, dtype=torch.float64, device=dev))
        dist.all_reduce(ones)
        gather_info["rccl_ranks"] = int(round(float(ones.item())))
        gather_info["backend"] = dist.get_backend()

    # The scaling curve's own references, measured in THIS run on rank 0's GPU while the other ranks wait at a barrier: (i) the per-GPU shard
    # alone (weak scaling: N GPUs should do N times that), (ii) the WHOLE job's batch on one GPU (strong scaling: what sharding buys over
    # keeping the batch on one device - a large batch runs on the lane-per-QP kernel there, PQP_OPT_STREAM_BATCH).
    scaling_ref = None
    if dist is not None and pipe is None:
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:
            try:
                for _ in range(args.warmup):
                    step()
                sync_all(); torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(args.steps):
                    step()
                sync_all(); torch.cuda.synchronize()
                shard_alone = batch * args.steps / (time.perf_counter() - ta)
                hw = make_batch(total, n, profile, seed=args.seed)
                w_ref = torch.from_numpy(hw["ref"]).to(dev)
                # (as the shards: consecutive launches solve the scenarios one planning cycle later, so that a start / wavefront order by the previous solve's
                #  costs has no perfect foresight)
                w_var = []
                for v in range(min(n_var, 4)):
                    hv = jitter_batch(hw, v, seed=args.seed)
                    w_var.append((torch.from_numpy(hv["bounds"]).to(dev), torch.from_numpy(hv["scal"]).to(dev)))
                h_w = capi.Handle(prm, device=local_rank, max_batch=total, max_n=n)
                h_w.set_option(capi.OPT_STORE_WARM, 0); h_w.set_option(capi.OPT_ORDER_BY_COST, 1 if cost_order else 0)
                o_w = torch.zeros((total, n, 7), dtype=torch.float64, device=dev); st_w = torch.zeros(total, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                for i in range(2):
                    h_w.solve_device(total, n, w_ref, w_var[i % len(w_var)][0], w_var[i % len(w_var)][1], o_w, passes=1, status=st_w)
                h_w.sync()
                kw = max(2, min(args.steps, 8))
                ta = time.perf_counter()
                for i in range(kw):
                    h_w.solve_device(total, n, w_ref, w_var[i % len(w_var)][0], w_var[i % len(w_var)][1], o_w, passes=1, status=st_w)
                h_w.sync()
                whole = total * kw / (time.perf_counter() - ta)
                whole_kernel = {1: "path_solve_kernel (lane per waypoint)", 2: "path_stream_kernel (lane per QP)"}.get(h_w.last_path_kernel(), "?")
                scaling_ref = {"shard_alone_on_one_gpu": {"value": shard_alone, "unit": "paths/s", "batch": batch, "steps": args.steps, "batches_in_flight": len(main_handles)},
                               "whole_batch_on_one_gpu": {"value": whole, "unit": "paths/s", "batch": total, "launches": kw, "kernel": whole_kernel,
                                                          "solved": int((st_w == 1).sum().item()), "setting": "one launch after the other"},
                               "note": "measured on rank 0's GPU after the timed region, the other ranks idle at a barrier"}
                h_w.close()
                del w_ref, w_var, o_w, st_w
            except Exception as e:          # (the headline does not depend on it)
                scaling_ref = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize(); dist.barrier()

    # per-launch duration of the dominant kernel over the timed region: HIP events the handle recorded around every launch on the
    # stream it launched on, read back now (nothing was synchronised between the launches)
    ev_ms = []
    for li, hh in enumerate(main_handles):
        k = min(sum(1 for i in range(args.steps) if (args.warmup + i) % len(main_handles) == li), 256)      # timed launches of this handle
        if k > 0:
            ev_ms.extend(hh.kernel_ms_history(k).tolist())
    avg_kernel_s = float(np.mean(ev_ms)) * 1e-3

    # longer back-to-back run of the same step (the K steps above are what `value` is; a 20-step region lasts ~15 ms)
    sustained = None
    if args.sustain > 0:
        k_long = int(max(args.steps, min(20000, args.sustain / max(dt / args.steps, 1e-6))))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(k_long):
            step()
        sync_all()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt_long = time.perf_counter() - t1
        if dist is not None:
            tl = coll(torch.tensor([dt_long], dtype=torch.float64, device=dev))
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            dt_long = float(tl.item())
        sustained = {"value": total * k_long / dt_long, "unit": "paths/s", "steps": k_long, "seconds": dt_long}

    if pipe is None:
        # the checksum's step: variant 0 on the first handle, whatever K and the number of variants were
        hh, o, st, it, inf = lanes[0]
        hh.solve_device(batch, n, ref, var_in[0][0], var_in[0][1], o, passes=1, status=st, iters=it, info=inf)
        sync_all()
    if pipe is not None:
        res = pipe.result((counter[0] - 1))
        it_np, st_np = res["it"], res["st"]
        info_np = None
        kkt_np = fac_np = None
        sm_solved = int((res["sm_st"] == 1).sum())
    else:
        it_np, st_np, info_np = iters.cpu().numpy(), status.cpu().numpy(), info.cpu().numpy()
        kkt_np, fac_np = info_np[:, 5], info_np[:, 6]
    out_sha = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    solved_all = None
    if dist is not None:          # QPs solved over the whole job (every rank's shard): the one scalar reduction of the path (shard.reduce_stats)
        sa = coll(torch.tensor([int((st_np == 1).sum())], dtype=torch.int64, device=dev))
        dist.all_reduce(sa)
        solved_all = int(sa.item())

    # ---------------------------------------------------------------------------------------------------------------------------
    # secondary measurements on the same inputs (N = 1, path-QP configs): index order, the literal metric, the reference's setting
    # ---------------------------------------------------------------------------------------------------------------------------
    secondary = None
    if not args.no_secondary and world == 1 and pipe is None and not args.reference_setting and polish:
        def timed(p, order, steps, warm=3, inflight=1, variants=None, carry=False):
            """`steps` steps round-robin over `inflight` handles; variants: the (bounds, scal) sets consecutive steps cycle through.  Two batches in
            flight run on the headline's own two handles (parameters / start order set for the measurement): streams created later may share a
            hardware queue with each other and then do not overlap."""
            variants = variants or var_in
            own = inflight == 1
            if own:
                lns = [make_lane(p, order)]
            else:
                lns = lanes[:inflight]
                for ln in lns:
                    ln[0].set_params(p); ln[0].set_option(capi.OPT_ORDER_BY_COST, 1 if order else 0)
            for ln in lns:
                ln[0].set_option(capi.OPT_CARRY_CYCLES, int(carry))          # (True: every QP, k >= 2: the previous launch's most expensive 1 / k only)
            k = [0]

            def one():
                hh, o, st, it, inf = lns[k[0] % len(lns)]
                bv, sv = variants[k[0] % len(variants)]
                k[0] += 1
                hh.solve_device(batch, n, ref, bv, sv, o, passes=1, status=st, iters=it, info=inf)

            def wait():
                for ln in lns:
                    ln[0].sync()
            for _ in range(warm * inflight):
                one()
            wait()
            ta = time.perf_counter()
            for _ in range(steps):
                one()
            wait()
            tb = time.perf_counter() - ta
            hh, o, st, it, inf = lns[0]
            info_rows = stream and p.polish != 0
            sweeps_mean = float(inf.cpu().numpy()[:, 6].mean()) if info_rows else None          # (of the last timed step on this handle)
            if carry and not stream:
                kkt_carry = (float(inf.cpu().numpy()[:, 5].mean()), float(inf.cpu().numpy()[:, 6].mean()), float(inf.cpu().numpy()[:, 5].max()))
            for ln in lns:
                ln[0].set_option(capi.OPT_CARRY_CYCLES, 0)                  # (the checksum's step below is a cold one)
            hh.solve_device(batch, n, ref, variants[0][0], variants[0][1], o, passes=1, status=st, iters=it, info=inf)      # the checksum's step
            hh.sync()
            itn, stn = it.cpu().numpy(), st.cpu().numpy()
            r = {"value": batch * steps / tb, "unit": "paths/s", "steps": steps, "ms_per_step": tb / steps * 1e3, "batches_in_flight": inflight,
                 "kernel_ms": float(np.mean(np.concatenate([ln[0].kernel_ms_history(min(max(steps // inflight, 1), 256)) for ln in lns]))),
                 "solved": int((stn == 1).sum()),
                 "admm_iters": {"min": int(itn.min()), "median": float(np.median(itn)), "p99": float(np.percentile(itn, 99)), "max": int(itn.max()),
                                "mean": float(itn.mean())}, "out_sha1": hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:16]}
            for ln in lns:
                if own:
                    ln[0].close()
                else:
                    ln[0].set_params(prm); ln[0].set_option(capi.OPT_ORDER_BY_COST, 1 if cost_order else 0); ln[0].set_option(capi.OPT_CARRY_CYCLES, 0)
            if info_rows:
                r["riccati_sweeps_mean"] = sweeps_mean
            if carry and not stream:
                r["kkt_solves_mean"], r["factorisations_mean"], r["kkt_solves_max"] = kkt_carry
            return r
        nfl = len(lanes)
        # (the driver times K = 20 steps = 7 ms of configs[1]: the side measurements take at least 400 steps each, so that they mean something there too)
        sec_steps = max(args.steps, min(400, max(20, int(0.25 / max(dt / args.steps, 1e-6)))))
        secondary = {
            "one_batch_at_a_time": dict(timed(prm, cost_order, sec_steps), setting="the headline setting, one launch strictly after the other (--inflight 1)")
            if nfl > 1 else None,
            "index_order": dict(timed(prm, not cost_order, sec_steps), setting="one launch after the other with the QPs started in index order"
                                if cost_order else "one launch after the other with the QPs started most-expensive-first (previous step's cost)"),
            "inflight2_index_order": dict(timed(prm, False, sec_steps, inflight=2), setting="two batches in flight, QPs started in index order") if nfl >= 2 else None,
            "identical_batch_every_step": dict(timed(prm, cost_order, sec_steps, inflight=nfl, variants=var_in[:1]),
                                               setting="the headline setting with the SAME batch re-solved every step (round 2's headline: the start "
                                                       "order then has perfect foresight of every QP's cost)") if n_var > 1 else None,
            "plain_admm_eps_1e-4": dict(timed(capi.default_params(eps_abs=1e-4, eps_rel=1e-4), False, max(3, sec_steps // 8)),
                                        setting="the literal metric: OSQP termination at eps_abs = eps_rel = 1e-4, OSQP defaults, no polish "
                                                "(pqp_default_params); paths 1e-5..2e-3 from the optimum"),
            "plain_admm_eps_1e-4_as_the_headline_runs": dict(timed(capi.default_params(eps_abs=1e-4, eps_rel=1e-4), cost_order, max(6, sec_steps // 8), inflight=nfl),
                                        setting="the literal metric under the headline's launch conditions: two batches in flight, QPs started most-expensive-first by the "
                                                "previous step's cost (a launch of 1024 QPs on 512 slots otherwise lasts as long as its slowest QP: 1100 iterations against 247 on average)") if nfl > 1 else None,
            "reference_setting_eps_2e-3": dict(timed(capi.default_params(), False, max(3, sec_steps // 4)),
                                               setting="what base_solver.cpp:61-62 runs: eps 2e-3, OSQP defaults, no polish; paths 2e-4..2e-2 from the optimum"),
        }
        if cfg_id == 1 and preset_shape:
            secondary["base_solver_shim_batch1"] = shim_batch1()
            # the same batch at the reference's own path length: its demo plans N ~ 60 waypoints (BASELINE configs[0]), which fit ONE wavefront per QP -
            # four QPs per CU instead of two (64 -> 65 waypoints costs 1.8x)
            try:
                n60 = 60
                h60 = make_batch(batch, n60, profile, seed=args.seed)
                r60 = torch.from_numpy(h60["ref"]).to(dev)
                v60 = []
                for v in range(4):
                    hv = jitter_batch(h60, v, seed=args.seed)
                    v60.append((torch.from_numpy(hv["bounds"]).to(dev), torch.from_numpy(hv["scal"]).to(dev)))
                l60 = []
                for _ in range(nfl):
                    hh = capi.Handle(prm, device=local_rank, max_batch=batch, max_n=n60)
                    hh.set_option(capi.OPT_STORE_WARM, 0); hh.set_option(capi.OPT_ORDER_BY_COST, 1 if cost_order else 0)
                    l60.append((hh, torch.zeros((batch, n60, 7), dtype=torch.float64, device=dev), torch.zeros(batch, dtype=torch.int32, device=dev)))
                torch.cuda.synchronize()

                def run60(k):
                    for i in range(k):
                        hh, o, st = l60[i % len(l60)]
                        hh.solve_device(batch, n60, r60, v60[i % 4][0], v60[i % 4][1], o, passes=1, status=st)
                    for hh, *_ in l60:
                        hh.sync()
                run60(8)
                ta = time.perf_counter()
                run60(sec_steps)
                t60 = time.perf_counter() - ta
                secondary["reference_demo_length_n60"] = {"value": batch * sec_steps / t60, "unit": "paths/s", "steps": sec_steps, "ms_per_step": t60 / sec_steps * 1e3,
                                                          "batch": batch, "n_waypoints": n60, "batches_in_flight": len(l60), "solved": int((l60[0][2] == 1).sum().item()),
                                                          "setting": "the headline setting at N = 60 waypoints: one wavefront per QP (64 lanes), four QPs per CU = 1024 slots - a batch of 1024 is then ONE QP per slot and a "
                                                                     "launch lasts as long as its slowest QP (handles created after the headline's: their streams may share a hardware queue, so this is a lower bound - on its own, `bench.py --batch 1024 --n 60`: 5.3 M paths/s, profiles/r05ac_interval_rule_ends.txt); 8192 QPs of N = 60: 7.7 M paths/s (profiles/r05ab_n_sweep_batch8192.txt)"}
                for hh, *_ in l60:
                    hh.close()
                del l60, v60, r60
            except Exception as e:          # (the headline does not depend on it)
                secondary["reference_demo_length_n60"] = {"error": f"{type(e).__name__}: {e}"}
        if not stream and n_var > 1:
            txt = ("PQP_OPT_CARRY_CYCLES: the first solve of a cycle starts from the final iterate and active set the handle kept from the "
                   "previous cycle (the jittered variant {} step(s) earlier) instead of cold - what a planner that re-solves its scenarios every cycle would switch "
                   "on (same paths: the optimum is unique); the reference constructs a fresh solver per cycle, so `value` is measured without it")
            secondary["carry_cycles"] = dict(timed(prm, cost_order, sec_steps, inflight=nfl, carry=True), setting="the headline setting with " + txt.format(nfl))
            secondary["carry_cycles_one_batch_at_a_time"] = dict(timed(prm, cost_order, sec_steps, carry=True), setting="one launch after the other with " + txt.format(1))
            if cost_order:
                tails = ("PQP_OPT_CARRY_CYCLES = {0}: only the QPs that were among the most expensive 1/{0} of the handle's previous solve (by the cost keys the start order "
                         "uses) start from their previous cycle's optimum, all others start cold; same paths; not `value` for the same reason")
                secondary["carry_tails"] = dict(timed(prm, cost_order, sec_steps, inflight=nfl, carry=8), setting="the headline setting with " + tails.format(8))
                secondary["carry_tails_one_batch_at_a_time"] = dict(timed(prm, cost_order, sec_steps, carry=8), setting="one launch after the other with " + tails.format(8))
                secondary["carry_tails_quarter_one_batch_at_a_time"] = dict(timed(prm, cost_order, sec_steps, carry=4), setting="one launch after the other with " + tails.format(4))
        if stream and n_var > 1:
            secondary["carry_cycles"] = dict(timed(prm, cost_order, sec_steps, inflight=nfl, carry=True),
                                             setting="the headline setting with PQP_OPT_CARRY_CYCLES: the first pass of every QP starts from the optimum its slot had in the handle's "
                                                     "previous solve (the jittered variant two steps earlier) instead of cold - what a planner that re-solves its scenarios every "
                                                     "cycle would switch on; the reference constructs a fresh solver per cycle, so `value` is measured without it")
        if not stream and cfg_id == 1 and preset_shape:
            # BASELINE configs[3]'s WHOLE batch (65 536 QPs, N = 80) on this one GPU: the size at which the lane-per-QP kernel takes over
            # (PQP_OPT_STREAM_BATCH) - measured HBM roofline: `python bench.py --config 3 --batch 65536`, profiles/r03*_bench_stream*
            try:
                bb = 65536
                hb = make_batch(bb, 80, "uniform", seed=args.seed)
                t_ref, t_b, t_s = (torch.from_numpy(hb[k]).to(dev) for k in ("ref", "bounds", "scal"))
                res_w = {}
                for name, thr in (("lane_per_qp_stream_kernel", 1), ("lane_per_waypoint_kernel", 0)):
                    hh = capi.Handle(prm, device=local_rank, max_batch=bb, max_n=80)
                    hh.set_option(capi.OPT_STORE_WARM, 0); hh.set_option(capi.OPT_ORDER_BY_COST, 1); hh.set_option(capi.OPT_STREAM_BATCH, thr)
                    o = torch.zeros((bb, 80, 7), dtype=torch.float64, device=dev); stt = torch.zeros(bb, dtype=torch.int32, device=dev)
                    inf = torch.zeros((bb, 8), dtype=torch.float64, device=dev)
                    torch.cuda.synchronize()
                    for _ in range(2):
                        hh.solve_device(bb, 80, t_ref, t_b, t_s, o, passes=1, status=stt, info=inf)
                    hh.sync()
                    ta = time.perf_counter()
                    ks = 4
                    for _ in range(ks):
                        hh.solve_device(bb, 80, t_ref, t_b, t_s, o, passes=1, status=stt, info=inf)
                    hh.sync()
                    tb = (time.perf_counter() - ta) / ks
                    kms = float(np.mean(hh.kernel_ms_history(ks)))
                    res_w[name] = {"value": bb / tb, "unit": "paths/s", "ms_per_step": tb * 1e3, "kernel_ms": kms, "solved": int((stt == 1).sum().item()),
                                   "out_sha1": hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:16]}
                    if thr:
                        inf_timed = inf.cpu().numpy()          # the counts of the last TIMED launch (a sorted one): the roofline's bytes below belong to kms
                        res_w[name]["riccati_sweeps_mean"] = float(inf_timed[:, 6].mean())
                        # the same with PQP_OPT_CARRY_CYCLES over 4 jittered variants of the batch (the scenarios one planning cycle later)
                        var_b = []
                        for v in range(4):
                            hv = jitter_batch(hb, v + 1, seed=args.seed)
                            var_b.append((torch.from_numpy(hv["bounds"]).to(dev), torch.from_numpy(hv["scal"]).to(dev)))
                        # (the launches above re-solve the IDENTICAL batch: the wavefront order - PQP_OPT_ORDER_BY_COST, wavefronts of QPs that ran the same
                        #  phases in the previous solve - then has perfect foresight.  The scenarios one planning cycle later:)
                        torch.cuda.synchronize()
                        for v in range(4):
                            hh.solve_device(bb, 80, t_ref, var_b[v][0], var_b[v][1], o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tj = time.perf_counter()
                        for v in range(4):
                            hh.solve_device(bb, 80, t_ref, var_b[v][0], var_b[v][1], o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tjd = (time.perf_counter() - tj) / 4
                        res_w[name]["on_jittered_planning_cycles"] = {"value": bb / tjd, "unit": "paths/s", "ms_per_step": tjd * 1e3, "solved": int((stt == 1).sum().item()),
                                                                      "setting": "4 jittered variants of the batch cycled: the wavefront order comes from the previous cycle's phase counts"}
                        hh.set_option(capi.OPT_ORDER_BY_COST, 0)
                        torch.cuda.synchronize()
                        for _ in range(2):
                            hh.solve_device(bb, 80, t_ref, t_b, t_s, o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tu = time.perf_counter()
                        for _ in range(ks):
                            hh.solve_device(bb, 80, t_ref, t_b, t_s, o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tud = (time.perf_counter() - tu) / ks
                        res_w[name]["index_order"] = {"value": bb / tud, "unit": "paths/s", "ms_per_step": tud * 1e3, "setting": "PQP_OPT_ORDER_BY_COST off: QP k in slot k (round 4's launch)"}
                        hh.set_option(capi.OPT_ORDER_BY_COST, 1)
                        hh.set_option(capi.OPT_CARRY_CYCLES, 1)
                        torch.cuda.synchronize()
                        for v in range(4):
                            hh.solve_device(bb, 80, t_ref, var_b[v][0], var_b[v][1], o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tc = time.perf_counter()
                        for v in range(4):
                            hh.solve_device(bb, 80, t_ref, var_b[v][0], var_b[v][1], o, passes=1, status=stt, info=inf)
                        hh.sync()
                        tcd = (time.perf_counter() - tc) / 4
                        res_w[name]["with_carry_cycles"] = {"value": bb / tcd, "unit": "paths/s", "ms_per_step": tcd * 1e3, "solved": int((stt == 1).sum().item()),
                                                            "riccati_sweeps_mean": float(inf.cpu().numpy()[:, 6].mean()),
                                                            "setting": "PQP_OPT_CARRY_CYCLES on 4 jittered variants of the batch: every QP's first pass starts from its slot's previous optimum"}
                        hh.set_option(capi.OPT_CARRY_CYCLES, 0)
                        del var_b
                        hh.solve_device(bb, 80, t_ref, t_b, t_s, o, passes=1, status=stt, info=inf)
                        hh.sync()
                        ab = stream_algorithmic_bytes(80, inf_timed, direct=stream_direct_rounds(inf_timed))
                        res_w[name]["roofline"] = {"bound": "hbm", "kernel": "path_stream_kernel", "algorithmic_bytes_per_launch": ab, "achieved": ab / (kms * 1e-3) / 1e9,
                                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                                   "note": "bytes the algorithm streams through its HBM workspace for the sweeps each QP ran (DESIGN.md 3b) / kernel time; "
                                                           "measured traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE): the roofline of `python bench.py --config 3 --batch 65536`, profiles/"}
                    else:
                        oa = o.cpu().numpy()
                    hh.close()
                    del o, inf
                secondary["configs3_whole_batch_one_gpu"] = dict(res_w, workload="configs[3]: batch = 65 536 QPs, N = 80, the whole batch on ONE GPU, one launch after the other",
                                                                 speedup_stream_over_lane_per_waypoint=res_w["lane_per_qp_stream_kernel"]["value"] / res_w["lane_per_waypoint_kernel"]["value"])
                del t_ref, t_b, t_s
                torch.cuda.empty_cache()
                if args.pmc == "auto":
                    # measured HBM traffic of THIS kernel in this run too (the two TCC byte counters, a child pass each; the SQ counters are
                    # `python bench.py --config 3 --batch 65536`'s)
                    core3 = ["--config", "3", "--batch", str(bb), "--n", "80", "--profile", "uniform", "--eps", str(args.eps), "--seed", str(args.seed), "--variants", "1"]
                    pmc3 = pmc_child(core3, "path_stream_kernel", args.pmc_timeout, steps=3, passes=["FETCH_SIZE GRBM_GUI_ACTIVE", "WRITE_SIZE SQ_WAVES"])
                    rf = res_w["lane_per_qp_stream_kernel"]["roofline"]
                    if pmc3 and "FETCH_SIZE" in pmc3 and "WRITE_SIZE" in pmc3:
                        cal = fetch_calibration()
                        tr = (cal["stream_read"] * pmc3["FETCH_SIZE"] + cal["stream_write"] * pmc3["WRITE_SIZE"]) * 1024.0
                        kms3 = res_w["lane_per_qp_stream_kernel"]["kernel_ms"]
                        rf.update(traffic=tr, fetch_size_kib=pmc3["FETCH_SIZE"], write_size_kib=pmc3["WRITE_SIZE"], fetch_size_factor=cal["stream_read"],
                                  write_size_factor=cal["stream_write"], traffic_over_algorithmic=tr / rf["algorithmic_bytes_per_launch"],
                                  hbm_measured_frac=tr / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  traffic_source=f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of `bench.py --config 3 --batch {bb}` (one launch at a time), "
                                                 f"{pmc3['_launches']} launches averaged; factors: {cal['source']}")
            except Exception as e:          # (out of memory on a shared box, ...: the headline does not depend on it)
                secondary["configs3_whole_batch_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}

    # ---------------------------------------------------------------------------------------------------------------------------
    # roofline of the dominant kernel
    # ---------------------------------------------------------------------------------------------------------------------------
    roofline, roofline_issue = None, None
    if pipe is None:
        setups = 2.0
        kernel_name = "path_stream_kernel" if stream else "path_solve_kernel"
        if stream:
            abytes = stream_algorithmic_bytes(n, info_np, direct=stream_direct_rounds(info_np))
            abytes_admm = abytes_ext = abytes
        else:
            abytes, abytes_admm, abytes_ext = algorithmic_bytes(n, kkt_np, it_np, fac_np, setups)
        pmc = None
        if args.pmc == "auto" and world == 1 and rank == 0:
            core = ["--config", str(cfg_id), "--batch", str(batch), "--n", str(n), "--profile", profile, "--eps", str(args.eps), "--seed", str(args.seed),
                    "--variants", "1"]            # (the counters of the batch whose sweep / solve counts the byte model above uses)
            core += ["--no-cost-order"] if not cost_order else []
            core += ["--no-polish"] if args.no_polish else []
            core += ["--reference-setting"] if args.reference_setting else []
            pmc = pmc_child(core, kernel_name, args.pmc_timeout, steps=3 if stream else 6)
        traffic = None
        calib = fetch_calibration()
        f_rd, f_wr = (calib["stream_read"], calib["stream_write"]) if stream else (calib["lane_read"], calib["lane_write"])
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # FETCH_SIZE / WRITE_SIZE are in KiB; what a KiB of them is worth in bytes for this kernel's access pattern was measured
            # (fetch_calibration(): gfx950 tallies a wide read at half its bytes - MICROARCH guide, HBM - but 8 B / 4 B per lane lines differ)
            traffic = (f_rd * pmc["FETCH_SIZE"] + f_wr * pmc["WRITE_SIZE"]) * 1024.0
        # launches of different handles overlap: mean number of solve kernels running at a time over the timed region
        concurrency = max(1.0, args.steps * avg_kernel_s / dt) if len(main_handles) > 1 else 1.0
        per_launch = lambda b: b / avg_kernel_s / 1e9 / HBM_PEAK_GBS
        common = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": kernel_name, "kernel_ms": avg_kernel_s * 1e3,
                  "launches_in_flight": len(main_handles), "kernels_running_at_a_time": concurrency, "traffic": traffic,
                  "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (one launch at a time, the identical batch), after the "
                                     f"timed region, {pmc['_launches']} launches averaged; bytes = {f_rd:.3f} x FETCH_SIZE + {f_wr:.3f} x WRITE_SIZE, factors: {calib['source']}") if traffic else None,
                  "fetch_size_factor": f_rd, "write_size_factor": f_wr,
                  "fetch_size_kib": pmc.get("FETCH_SIZE") if pmc else None, "write_size_kib": pmc.get("WRITE_SIZE") if pmc else None,
                  # what HBM really moved per second of ONE launch's own duration, as a fraction of the 8 TB/s peak
                  "hbm_measured_frac": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                  "hbm_measured_frac_chip_wide": (traffic * concurrency / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None}
        if stream:
            roofline = dict(common, achieved=abytes / avg_kernel_s / 1e9, frac=per_launch(abytes), frac_chip_wide=per_launch(abytes) * concurrency,
                            algorithmic_bytes_per_launch=abytes, traffic_over_algorithmic=(traffic / abytes) if traffic else None,
                            note="path_stream_kernel keeps nothing on chip between two sweeps: achieved = the bytes its algorithm reads and writes in its HBM workspace "
                                 "for the sweeps every QP ran (DESIGN.md 3b: 368 / 448 B per waypoint and interior-point iteration, 248 / 328 B per active-set round, "
                                 "+ prep / init / unpack) / the launch's event-timed duration.  `traffic` (rocprofv3) is larger: a wavefront moves a 128-byte line "
                                 "as long as ONE of its 16 lanes is still iterating.  hbm_measured_frac = traffic / kernel time / 8 TB/s")
        else:
            true_io = 2.0 * batch * (152 * n + 40)
            roofline = dict(common, achieved=abytes_admm / avg_kernel_s / 1e9, frac=per_launch(abytes_admm), frac_chip_wide=per_launch(abytes_admm) * concurrency,
                            frac_all_solves=per_launch(abytes), frac_all_solves_chip_wide=per_launch(abytes) * concurrency,
                            algorithmic_bytes_per_launch=abytes_admm, algorithmic_bytes_all_solves_per_launch=abytes,
                            achieved_all_solves=abytes / avg_kernel_s / 1e9, achieved_incl_factor_and_scaling=abytes_ext / avg_kernel_s / 1e9,
                            true_io_bytes_per_launch=true_io, traffic_over_true_io=(traffic / true_io) if traffic else None,
                            note="achieved / frac: SURVEY.md 8(d)'s STREAMING MODEL taken literally - B_path = 2 B_io + 2 B_asm + (ADMM iterations) x 1040 N bytes - per LAUNCH "
                                 "(one launch's bytes / its own event-timed duration); frac_all_solves also charges 1040 N for every polish refinement solve; *_chip_wide "
                                 "= x kernels running at a time (two launches in flight share the chip).  A MODEL, not traffic: the iterates of path_solve_kernel are "
                                 "register / LDS resident, it is bound by fp64 VALU issue + LDS latency (roofline_issue); what HBM really moved is `traffic`, "
                                 "hbm_measured_frac = traffic / kernel time / 8 TB/s.  The kernel whose roofline IS measured traffic is path_stream_kernel "
                                 "(batches >= 15 360 at N = 80: secondary.configs3_whole_batch_one_gpu, `bench.py --config 3 --batch 65536`)")
        if secondary and secondary.get("plain_admm_eps_1e-4") and not stream:
            # what the model says about the solver the metric names: OSQP's plain ADMM streaming its data from HBM every iteration
            its = secondary["plain_admm_eps_1e-4"]["admm_iters"]["mean"]
            b_path = 2.0 * (152 * n + 40 + 656 * n) + its * 1040 * n
            bound = HBM_PEAK_GBS * 1e9 / b_path
            roofline["streaming_plain_admm"] = {"admm_iterations_per_path": its, "model_bytes_per_path": b_path, "hbm_bound_paths_per_s": bound,
                                                "this_run_over_that_bound": batch * args.steps / dt / bound,
                                                "note": "SURVEY.md 8(d)'s B_path for plain ADMM to eps 1e-4 (iterations measured in this run, `secondary`): the "
                                                        "paths/s at which an HBM-streaming implementation of the metric's literal algorithm saturates 8 TB/s"}
        if pmc and "SQ_ACTIVE_INST_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
            wc = pmc["SQ_WAVE_CYCLES"]
            gui = pmc.get("GRBM_GUI_ACTIVE")                                      # summed over the 8 XCDs
            kernel_cycles = gui / 8.0 if gui else avg_kernel_s * 2.4e9
            valu_cycles = pmc["SQ_ACTIVE_INST_VALU"] * 4.0                       # quad-cycles -> cycles
            roofline_issue = {"bound": "fp64_valu_issue", "kernel": kernel_name, "achieved": valu_cycles / (SIMDS * kernel_cycles), "peak": 1.0, "unit": "fraction of the "
                              "chip's VALU issue cycles (VALU-active cycles / (1024 SIMDs x kernel cycles))", "frac": valu_cycles / (SIMDS * kernel_cycles),
                              "valu_active_frac_of_wave_cycles": pmc["SQ_ACTIVE_INST_VALU"] / wc, "lds_active_frac": pmc.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
                              "scalar_active_frac": pmc.get("SQ_ACTIVE_INST_SCA", 0.0) / wc, "any_active_frac": pmc.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                              "wait_frac": pmc.get("SQ_WAIT_ANY", 0.0) / wc, "valu_instructions_per_launch": pmc.get("SQ_INSTS_VALU"),
                              "waves_per_launch": pmc.get("SQ_WAVES"), "kernel_cycles": kernel_cycles,
                              # the dynamic instruction mix (round 6): what share of the VALU instructions that RAN is fp64 arithmetic - the rest moves
                              # registers (AGPR <-> VGPR copies, lane moves of spilled SGPRs, selects, DPP exchanges); tools/isa_mix.py gives the static view
                              "fp64_share_of_valu_instructions": (sum(pmc.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")) / pmc["SQ_INSTS_VALU"])
                                                                 if pmc.get("SQ_INSTS_VALU") and "SQ_INSTS_VALU_FMA_F64" in pmc else None,
                              "wait_inst_frac": pmc.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                              "source": f"rocprofv3 --pmc child pass of this command, {pmc['_launches']} launches averaged"}

    if rank == 0:
        setting = "reference (pqp_default_params)" if args.reference_setting else "production (pqp_production_params)"
        workload = cfg["what"] if preset_shape else f"custom shape on {cfg['what'].split(':')[0]}: batch={batch} per GPU, N={n}, {profile} profile"
        line = {
            "metric": "paths/sec (QP solves/sec) at N=80 waypoints; ADMM iters to 1e-4",
            "value": total * args.steps / dt, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total is not None else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "config_id": cfg_id, "batch_per_gpu": batch, **({"total": total, "shards": "contiguous, ragged: the first total % N ranks hold one QP more (batch_per_gpu is rank 0's)"} if args.total is not None else {}),
                       "n_waypoints": n, "profile": profile,
                       "eps_abs": args.eps, "eps_rel": args.eps, "polish": polish, "setting": setting,
                       "solver": ("lane-per-QP kernel: interior-point rounds + KKT-verifying active-set rounds on the QP as a linear-quadratic control problem "
                                  "(every path is the exact QP optimum; iterations below are interior-point iterations, not ADMM's)") if stream else
                                 ("lane-per-waypoint kernel: ADMM iterations + KKT-verified active-set polish (every path is the exact QP optimum, inside the metric's "
                                  "1e-4; the metric's literal 'ADMM iters to 1e-4' is secondary.plain_admm_eps_1e-4)") if polish else "plain OSQP termination (ADMM to eps)",
                       "kernel": "path_stream_kernel" if stream else "path_solve_kernel",
                       "scenarios_per_step": f"{n_var} variants cycled: the batch and its +-5 % jittered planning cycles (synth.jitter_batch)" if pipe is None and n_var > 1 else "the identical batch every step",
                       "polish_every": capi.path_interval(prm.polish_every, n), "adaptive_rho_interval": capi.path_interval(prm.adaptive_rho_interval, n), "check_termination": capi.path_interval(prm.check_termination, n),
                       "ruiz_passes": abs(prm.scaling), "ruiz_evaluated_on": "one interior waypoint's blocks, taken by every waypoint (pqp_params.scaling < 0: a valid "
                       "diagonal scaling, equal to the full passes' D, E, c on these scenario families)" if prm.scaling < 0 else "every waypoint (OSQP's passes)", "polish_lazy": prm.polish_lazy,
                       "polish_refine_iter": args.polish_refine, "polish_max_rounds": args.polish_max_rounds, "polish_warm_set": args.polish_warm_set,
                       "passes": "cold solve + 1 re-linearised warm re-solve (PathOptimizer::optimizePath)",
                       "qp_start_order": "most expensive first by the previous step's cost (PQP_OPT_ORDER_BY_COST)" if cost_order else "index order",
                       "parallelism": f"{world} independent shard(s), no collective in the timed region" + (" - TEST MODE: all ranks share GPU 0 (PQP_BENCH_SHARED_GPU), numbers meaningless" if shared_gpu else ""),
                       "batches_in_flight": max(args.inflight, 1), "prewarm_s": args.prewarm,
                       **({"smoother": "TensionSmoother2 QP (equality rows only) " + ("as the reference runs it: ADMM to eps 1e-3" if args.reference_setting else
                                       "solved exactly by one Riccati sweep per scenario (pqp_params.polish = 2: tension2_exact_kernel, no ADMM iterations)")} if pipe is not None else {})},
            "admm_iters": {"min": int(it_np.min()), "median": float(np.median(it_np)), "p99": float(np.percentile(it_np, 99)),
                           "max": int(it_np.max()), "mean": float(it_np.mean())},
            "out_sha1": out_sha, "gather_check": gathered_ok, "gather": gather_info, "rccl_ranks": gather_info["rccl_ranks"] if gather_info else None,
            "solved": int((st_np == 1).sum()), "batch": batch, **({"solved_all_ranks": solved_all, "total": total} if solved_all is not None else {}),
            "sustained": sustained, "secondary": secondary, "roofline": roofline, "roofline_issue": roofline_issue,
        }
        if pipe is None:
            if stream:
                line["lanes_active_frac"] = {"value": batch / (64.0 * ((batch + 63) // 64)), "what": "QPs per 64-lane wavefront of path_stream_kernel (one lane per QP)"}
            else:
                t_lanes = 64
                while t_lanes < n:
                    t_lanes *= 2
                line["lanes_active_frac"] = {"value": n / float(t_lanes), "lanes_per_qp": t_lanes,
                                             "what": "waypoints per lane of a QP's workgroup (one lane per waypoint, 64 x 2^k lanes per QP): at N = 80 a QP holds 128 lanes - the kernel's time "
                                                     "per QP follows the lanes (the depth of the reduction tree), not N; filling the idle 37.5 % with a third QP per two workgroups "
                                                     "runs the three in lock-step: x0.95 ... x1.07 by the trace-driven model of profiles/r05_lockstep_three_qps_per_workgroup.txt"}
        if scaling_ref is not None and "error" not in scaling_ref:
            line["weak_scaling_efficiency"] = line["value"] / (world * scaling_ref["shard_alone_on_one_gpu"]["value"])
            line["strong_scaling_vs_one_gpu_whole_batch"] = line["value"] / scaling_ref["whole_batch_on_one_gpu"]["value"]
            # north_star's "scaling 1 -> N GPUs" read as STRONG scaling: the whole job (N shards on N GPUs) against the same total batch kept
            # on ONE GPU with the kernel this library picks for that size there - next to "scaling": "weak" (per-GPU work fixed), which
            # is how the ranks are loaded
            line["scaling_strong"] = line["strong_scaling_vs_one_gpu_whole_batch"]
        if scaling_ref is not None:
            line["scaling_reference"] = scaling_ref
        if info_np is not None and stream:
            line["riccati_sweeps"] = {"mean": float(fac_np.mean()), "p99": float(np.percentile(fac_np, 99)), "max": float(fac_np.max())}
            line["active_set_rounds"] = {"mean": float(info_np[:, 7].mean()), "max": float(info_np[:, 7].max())}
            line["verified"] = int((info_np[:, 4] >= 2).sum())
        elif info_np is not None:
            line["kkt_solves"] = {"mean": float(kkt_np.mean()), "p99": float(np.percentile(kkt_np, 99)), "max": float(kkt_np.max())}
            line["factorisations"] = {"mean": float(fac_np.mean()), "max": float(fac_np.max())}
            line["polished"] = int((info_np[:, 4] >= 2).sum())
        if pipe is not None:
            line["unit"] = "paths/s"
            line["smoother_solved"] = sm_solved
            line["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                "kernel": "path_solve_kernel (behind banded_solve_kernel on the other stream)", "kernel_ms": avg_kernel_s * 1e3,
                                "note": "two kernels overlap on two streams; the streaming-model figure is reported for the path-QP configs"}
        if not args.no_cpu_baseline and world == 1 and pipe is None:          # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(lambda k: make_batch(k, n, profile, seed=args.seed), n, args.eps, args.cpu_budget)
        print(json.dumps(line))
    if pipe is not None:
        pipe.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
