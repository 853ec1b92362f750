#!/usr/bin/env python
"""Probe (round 6): what the re-linearised pass of a SORTED launch buys the lane-per-QP solver (lq::kDirectRounds active-set rounds on the first pass's set
before the interior-point rounds), in the host emulation of the device source: per-QP counts, the stream kernel's algorithmic bytes, and the lock-step
phases / bytes of wavefronts sorted by their phase keys.  (profiles/r06ae_lq_direct_probe_emulation.txt has the sweep over 0 ... 4 rounds made while the
number was a compile-time switch.)  Usage: python tools/lq_direct_probe.py [batch=8192] [n=80] [profile=uniform]      (CPU only)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_optimizer_2_amd.synth import make_batch
import lq_emu_util as E
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
profile = sys.argv[3] if len(sys.argv) > 3 else "uniform"
b = make_batch(batch, n, profile)
base = None
for sorted_launch in (False, True):
    r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, sorted_launch=sorted_launch)
    info, st, out = r["info"], r["status"], r["out"]
    d = bench.STREAM_DIRECT_ROUNDS if sorted_launch else 0
    it1, itt, s1, stt = info[:, 2], info[:, 3], info[:, 5], info[:, 7]
    by = bench.stream_algorithmic_bytes(n, info, direct=d) / batch / n
    key = np.stack([it1, s1, itt - it1, stt - s1], axis=1).astype(np.int64)
    order = np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0]))[::-1]
    ph = key[order][: batch // 64 * 64].reshape(-1, 64, 4).max(axis=1)
    w = np.zeros((len(ph), 8)); w[:, 2] = ph[:, 0]; w[:, 3] = ph[:, 0] + ph[:, 2]; w[:, 5] = ph[:, 1]; w[:, 7] = ph[:, 1] + ph[:, 3]; w[:, 4] = 2
    wby = bench.stream_algorithmic_bytes(n, w, direct=d) / len(ph) / n
    if base is None: base = out.copy()
    print(f"{'sorted launch' if sorted_launch else 'unsorted launch':16s}: solved {(st == 1).sum()}/{batch}; ipm iterations pass 1 {it1.mean():.2f} pass 2 {(itt - it1).mean():.2f} (max {(itt - it1).max():.0f}); "
          f"set rounds pass 1 {s1.mean():.2f} pass 2 {(stt - s1).mean():.2f} (max {(stt - s1).max():.0f}); second passes without an iteration {(itt - it1 == 0).mean():.3f}; "
          f"bytes/waypoint per lane {by:.0f}, per sorted wavefront {wby:.0f} ({ph.sum(axis=1).mean():.1f} phases); max |dout| vs unsorted {np.abs(out - base).max():.2e}")
