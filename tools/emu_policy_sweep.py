"""Statistics of the active-set rounds' policy on the HOST emulation of the lane-per-waypoint kernel (tests/emu/lane_emu.cpp: the kernel's own source,
csrc/pqp_path_lane.hpp, compiled for the CPU - test infrastructure): hundreds of thousands of QPs per shape and rule variant without a GPU.
Reports the modelled cost 4 x reduced solves + 13 x factorisations (the weights of the kernel's cost order) and the reduced solves: mean, p99, p99.99, max,
QPs without an accepted polish, the slowest QPs.  Which QP of a hundred thousand wanders depends on every constant and on the last bit of the arithmetic
(host libm vs device), so single QPs differ from the GPU's; the distributions are what carries over.
Usage: python tools/emu_policy_sweep.py <seeds> <batch> name[=DEFINE[,DEFINE...]] ...      e.g.  16 8192 base old=PQP_FULL_MOVE_SHARE=0.0,PQP_CAUTIOUS_FROM_ROUND=1073741824"""
import ctypes as C
import os, subprocess, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_optimizer_2_amd.capi import PqpParams
from path_optimizer_2_amd.synth import make_batch
SRC = os.path.join(ROOT, "tests", "emu", "lane_emu.cpp")
OUT = "/tmp/emu_variants"
SHAPES = ((80, "uniform"), (80, "varied"), (120, "varied"), (200, "uniform"), (37, "varied"), (300, "varied"))
CHUNK = 512


def build(spec):
    name, _, defs = spec.partition("=")
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f"lib_{name}.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DPQP_EMU_DIET=0", *[f"-D{d}" for d in defs.split(",") if d], "-o", lib, SRC], check=True)
    return name, lib


def task(a):
    lib_path, n, profile, seed, first, count = a
    lib = C.CDLL(lib_path)
    prm = PqpParams(); lib.pqp_emu_production_params(C.byref(prm))
    h = make_batch(count, n, profile, seed=seed, first_qp=first)
    vp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.c_void_p)
    B = count
    out = np.zeros((B, n, 7)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32); info = np.zeros((B, 8))
    wx = np.zeros((B, n, 6)); wy = np.zeros((B, n, 6)); wye = np.zeros((B, 2)); wrho = np.zeros(B)
    ref, bounds, scal = (np.ascontiguousarray(h[k]) for k in ("ref", "bounds", "scal"))
    lib.pqp_emu_set_counts(None)
    lib.pqp_emu_path_solve(C.byref(prm), B, n, vp(ref), None, vp(bounds), vp(scal), 1, 0, vp(out), vp(st), vp(it), vp(info), vp(wx), vp(wy), vp(wye), vp(wrho))
    return seed, first, info[:, 5].copy(), info[:, 6].copy(), st.copy(), info[:, 4].copy()


if __name__ == "__main__":
    seeds, batch = int(sys.argv[1]), int(sys.argv[2])
    specs = sys.argv[3:] or ["base"]
    shapes = SHAPES
    if os.environ.get("SHAPES"):
        shapes = tuple((int(s.split(":")[0]), s.split(":")[1]) for s in os.environ["SHAPES"].split(","))
    libs = [build(s) for s in specs]
    with ProcessPoolExecutor(min(96, os.cpu_count() or 1)) as ex:
        for n, profile in shapes:
            b = batch if n <= 120 else batch // 4
            for name, lib in libs:
                tasks = [(lib, n, profile, 1000 + s, f, min(CHUNK, b - f)) for s in range(seeds) for f in range(0, b, CHUNK)]
                kk, ff, bad, worst = [], [], 0, []
                for seed, first, k, f, st, pol in ex.map(task, tasks):
                    kk.append(k); ff.append(f); bad += int(((st != 1) | (pol != 2)).sum())
                    q = int(np.argmax(k)); worst.append((float(k[q]), seed, first + q))
                k = np.concatenate(kk); f = np.concatenate(ff); c = 4 * k + 13 * f
                worst.sort(reverse=True)
                print(f"n {n:3d} {profile:8s} {name:10s}: {len(k)} QPs, {bad} not solved+polished; cost mean {c.mean():6.1f} p99 {np.percentile(c, 99):5.0f} p99.99 {np.percentile(c, 99.99):5.0f} max {c.max():6.0f}; "
                      f"reduced solves mean {k.mean():.2f} p99.99 {np.percentile(k, 99.99):.0f} max {k.max():.0f}; slowest {worst[:3]}", flush=True)
