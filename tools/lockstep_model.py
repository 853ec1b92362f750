#!/usr/bin/env python
"""Round 5, VERDICT task 1: what would THREE 80-waypoint QPs per 4-wave workgroup (one 240-node chain, 240 of 256 lanes) buy?

A trace-driven model, no GPU: the host emulation of the solver (tests/emu, the device source compiled for the host) is built from a
temporary copy of pqp_path_lane.hpp with trace hooks (ctx.trace) in iterate / residuals / factor / do_cold, and run on bench.py's
configs[1] batch and its jittered planning cycles.  Every QP's trace is then the exact sequence of operations the kernel executes for it
(the iteration and factorisation counts are the GPU's: 16.77 reduced solves, 7.38 factorisations per QP).

Lock-step execution of G QPs as one chain: every unfinished QP advances by one "round" = [cold operation(s)] + iterate + [residuals] per
global round; a global round costs the union of what its QPs need (a factorisation if ANY of them factorises, each distinct cold
operation once - the lanes of the other QPs are predicated off).  Costs per operation: the device-clock breakdown of
profiles/r02o_timeline_batch1024_n80.txt (T = 128); at T = 256 every operation is x1.13 - measured: N = 200 on T = 256 runs 20.0 solves +
10.0 factorisations per QP in 238 us against 16.84 + 7.42 in 160.5 us at N = 80 on T = 128 (profiles/r04m_bench_n200_batch512.json,
r04m_bench_config3_shard.json: 238 / (4 x 20.0 + 13 x 10.0) against 160.5 / (4 x 16.84 + 13 x 7.42)).
Throughput ratio against today's two 2-wave workgroups per CU: (G / T_group) / (2 / T_single).

Usage: python tools/lockstep_model.py [batch=1024] [n=80]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_optimizer_2_amd.capi import PqpParams  # noqa: E402
from path_optimizer_2_amd import synth  # noqa: E402

COST = {1: 3.05, 2: 1.6, 3: 5.8, 10: 9.0, 11: 9.3, 20: 1.0, 21: 1.5, 22: 2.0, 23: 2.0, 30: 2.5, 40: 3.9, 50: 3.0}      # us at T = 128
T256 = 1.13


def build_tracing_emulation(tmp):
    for sub in ("path_optimizer_2_amd/csrc", "tests/emu", "include"):
        os.makedirs(os.path.join(tmp, sub))
    shutil.copy(os.path.join(ROOT, "include", "pqp.h"), os.path.join(tmp, "include"))
    for f in os.listdir(os.path.join(ROOT, "path_optimizer_2_amd", "csrc")):
        if f.endswith(".hpp"):
            shutil.copy(os.path.join(ROOT, "path_optimizer_2_amd", "csrc", f), os.path.join(tmp, "path_optimizer_2_amd", "csrc"))
    p = os.path.join(tmp, "path_optimizer_2_amd", "csrc", "pqp_path_lane.hpp")
    s = open(p).read()
    for head, code in (("    PQP_HD void iterate() {\n", "ctx.trace(1);"), ("    PQP_HD void residuals(double (&res)[6]) {\n", "ctx.trace(2);"),
                       ("    PQP_HD void factor() {\n", "ctx.trace(3);"),
                       ("    PQP_HD void do_cold(int op, int i0, int i1, double d0) {\n",
                        "ctx.trace(10 + op * 10 + (op == COLD_REFACTOR ? i0 : (op == COLD_BEGIN_PASS ? (i1 & 2 ? 1 : 0) : 0)));")):
        assert head in s
        s = s.replace(head, head + "        " + code + "\n")
    open(p, "w").write(s)
    e = open(os.path.join(ROOT, "tests", "emu", "lane_emu.cpp")).read()
    e = e.replace("static int g_wave_order = 1;", "static int g_wave_order = 1;\nstatic std::vector<int> g_trace;\nextern \"C\" int pqp_emu_trace(int* out, int cap) "
                  "{ int n = (int)g_trace.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = g_trace[i]; g_trace.clear(); return n; }")
    e = e.replace("    static double uni(double x) { return x; }\n", "    static double uni(double x) { return x; }\n    void trace(int c) { g_trace.push_back(c); }\n")
    e = e.replace("        HostCtx ctx(T);\n        if (prm->eps_prim_inf", "        HostCtx ctx(T);\n        g_trace.push_back(-1 - q);\n        if (prm->eps_prim_inf")
    src = os.path.join(tmp, "tests", "emu", "lane_emu.cpp")
    open(src, "w").write(e)
    lib = os.path.join(tmp, "liblane_emu_trace.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, src], check=True)
    return C.CDLL(lib)


def traces_of(lib, prm, b):
    ref, bounds, scal = [np.ascontiguousarray(b[k]) for k in ("ref", "bounds", "scal")]
    B, n = ref.shape[:2]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.zeros((B, n, 7)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32); info = np.zeros((B, 8))
    wx = np.zeros((B, n, 6)); wy = np.zeros((B, n, 6)); wye = np.zeros((B, 2)); wrho = np.zeros(B)
    lib.pqp_emu_set_counts(None)
    lib.pqp_emu_path_solve(C.byref(prm), B, n, vp(ref), None, vp(bounds), vp(scal), 1, 0, vp(out), vp(st), vp(it), vp(info), vp(wx), vp(wy), vp(wye), vp(wrho))
    buf = (C.c_int * (4000 * B))()
    m = lib.pqp_emu_trace(buf, 4000 * B)
    traces, cur = [], None
    for c in np.frombuffer(buf, dtype=np.int32)[:m]:
        if c < 0:
            cur = []; traces.append(cur)
        else:
            cur.append(int(c))
    assert (st == 1).all()
    return traces, info


def rounds(t):
    out, cur = [], dict(cold=[], F=False, I=False, R=False)
    for c in t:
        if c >= 10:
            if cur["I"]:
                out.append(cur); cur = dict(cold=[], F=False, I=False, R=False)
            cur["cold"].append(c)
        elif c == 3:
            cur["F"] = True
        elif c == 1:
            if cur["I"]:
                out.append(cur); cur = dict(cold=[], F=False, I=False, R=False)
            cur["I"] = True
        else:
            cur["R"] = True
    out.append(cur)
    return out


def t_single(t):
    return sum(COST[c] for c in t)


def t_group(trs, f):
    rs = [rounds(t) for t in trs]
    tot = 0.0
    for k in range(max(len(r) for r in rs)):
        act = [r[k] for r in rs if k < len(r)]
        kinds = set()
        for a in act:
            kinds.update(a["cold"])
        tot += f * (sum(COST[c] for c in kinds) + (COST[3] if any(a["F"] for a in act) else 0.0) + (COST[1] if any(a["I"] for a in act) else 0.0)
                    + (COST[2] if any(a["R"] for a in act) else 0.0))
    return tot


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    tmp = tempfile.mkdtemp(prefix="pqp_lockstep_")
    try:
        lib = build_tracing_emulation(tmp)
        prm = PqpParams(); lib.pqp_emu_production_params(C.byref(prm))
        base = synth.make_batch(B, n)
        tr0, info0 = traces_of(lib, prm, base)
        tr1, info1 = traces_of(lib, prm, synth.jitter_batch(base, 1))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"batch {B} x {n} waypoints, production setting: reduced solves {info1[:, 5].mean():.2f}, factorisations {info1[:, 6].mean():.2f} per QP (the GPU's counts)")
    ts = np.array([t_single(t) for t in tr1]); prev = np.array([t_single(t) for t in tr0])
    print(f"one QP alone (model, T = 128): mean {ts.mean():.1f} us, p99 {np.percentile(ts, 99):.1f}, max {ts.max():.1f}   (measured: 164 us)")
    rng = np.random.default_rng(0)
    print("G QPs in lock-step on one 4-wave workgroup against two 2-wave workgroups of one QP each (same four SIMDs):")
    for name, order in (("grouped at random", rng.permutation(B)), ("grouped by their cost in the previous planning cycle (what a handle knows)", np.argsort(-prev, kind="stable")),
                        ("grouped by their own cost (perfect foresight)", np.argsort(-ts))):
        for G in (2, 3):
            tg = np.array([t_group([tr1[q] for q in order[i:i + G]], T256) for i in range(0, B - G + 1, G)])
            t1 = np.array([t_group([tr1[q] for q in order[i:i + G]], 1.0) for i in range(0, B - G + 1, G)])
            print(f"  G = {G}  {name:78s} group {tg.mean():6.1f} us   paths/s x {(G / tg.mean()) / (2 / ts.mean()):.3f}   (lock-step loss alone: x {(G / t1.mean()) / (G / ts.mean()):.3f})")


if __name__ == "__main__":
    main()
