#!/usr/bin/env python
"""Probe (round 6): the re-linearised pass of the lane-per-QP solver started with active-set rounds on the previous pass's set (PQP_LQ_DIRECT_ROUNDS = k
rounds before the interior-point rounds get their turn), in the host emulation: per-QP counts, the stream kernel's algorithmic bytes, and the
lock-step phases of wavefronts sorted by their phase keys.  Usage: python tools/lq_direct_probe.py [batch=4096] [n=80] [profile=uniform]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_optimizer_2_amd.synth import make_batch
from path_optimizer_2_amd.capi import PqpParams
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
profile = sys.argv[3] if len(sys.argv) > 3 else "uniform"
b = make_batch(batch, n, profile)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
base = None
for k in (0, 1, 2, 3, 4):
    lib_path = os.path.join(ROOT, "ab", "emu", f"liblq_emu_d{k}.so")
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    subprocess.run(["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", "-shared", "-fPIC", f"-DPQP_LQ_DIRECT_ROUNDS={max(k, 1)}", "-o", lib_path,
                    os.path.join(ROOT, "tests", "emu", "lq_emu.cpp")], check=True)
    lib = C.CDLL(lib_path)
    prm = PqpParams(); lib.pqp_emu_lq_production_params(C.byref(prm))
    out = np.zeros((batch, n, 7)); st = np.zeros(batch, np.int32); it = np.zeros(batch, np.int32); info = np.zeros((batch, 8))
    (lib.pqp_emu_lq_solve_sorted if k else lib.pqp_emu_lq_solve)(C.byref(prm), batch, n, None, vp(b["ref"]), None, vp(b["bounds"]), vp(b["scal"]), 1, vp(out), vp(st), vp(it), vp(info))
    it1, itt, s1, stt = info[:, 2], info[:, 3], info[:, 5], info[:, 7]
    hits = (itt - it1 == 0).astype(float)
    inf = info
    by = bench.stream_algorithmic_bytes(n, inf) / batch / n
    key = np.stack([it1, s1, itt - it1, stt - s1], axis=1).astype(np.int64)
    order = np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0]))[::-1]
    ph = key[order][: batch // 64 * 64].reshape(-1, 64, 4).max(axis=1)
    if base is None: base = out.copy()
    print(f"direct rounds {k}: solved {(st == 1).sum()}/{batch}; ipm iterations pass 1 {it1.mean():.2f} pass 2 {(itt - it1).mean():.2f} (max {(itt - it1).max():.0f}); "
          f"set rounds pass 1 {s1.mean():.2f} pass 2 {(stt - s1).mean():.2f} (max {(stt - s1).max():.0f}); direct hits {hits.mean():.3f}; "
          f"bytes/waypoint {by:.0f}; sorted wavefronts: phases {ph.sum(axis=1).mean():.1f}, pass-2 ipm {ph[:, 2].mean():.2f} rounds {ph[:, 3].mean():.2f}; "
          f"max |dout| vs k=0 {np.abs(out - base).max():.2e}")
