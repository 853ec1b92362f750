"""Where the solve kernel's time goes, per QP: builds a -DPQP_TIMING variant of the library (device wall clock, 100 MHz,
accumulated per category by thread 0 and written over the info record) and prints the mean / max microseconds per category.
Debug tool; the timing build is never the shipped library.
Usage: python tools/kernel_timeline.py [batch] [n] [key=value ...]      (run on the GPU box)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "path_optimizer_2_amd", "csrc")
MASK = os.environ.get("PQP_TIMING_MASK") or None       # e.g. 0x10: only iterate() is timed (and the total): one category per build perturbs least
LIB = os.path.join(CSRC, "libpqp_hip_timing.so" if MASK is None else f"libpqp_hip_timing_{MASK}.so")


def build():
    """The library with the device-clock instrumentation of pqp_path_lane.hpp compiled into the lane-per-waypoint kernels (a debug build of its own:
    the four pqp_path_solve.hip objects with -DPQP_TIMING, linked with the shipped objects of the other translation units)."""
    import __graft_entry__ as G
    from concurrent.futures import ThreadPoolExecutor
    G.build_hip()
    defs = ["-DPQP_TIMING"] + ([] if MASK is None else [f"-DPQP_TIMING_MASK={MASK}"]) + (["-DPQP_TIMING_ITER"] if MASK == "0x0" else [])
    odir = os.path.join(CSRC, "build", "timing" if MASK is None else f"timing_{MASK}")
    units = [(s, d + defs, o) for s, d, o in G.HIP_UNITS if s == "pqp_path_solve.hip"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = [o for o, _ in ex.map(lambda u: G.compile_unit(*u, odir=odir), units)]
    objs += [os.path.join(CSRC, "build", "libpqp_hip", o) for s, _, o in G.HIP_UNITS if s != "pqp_path_solve.hip"]
    G.link_units(objs, LIB)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(); sys.exit(0)
    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    if not os.path.exists(LIB):         # (built where the objects of the other translation units are: `python tools/kernel_timeline.py build` before the GPU call)
        build()
    capi.LIB_PATH = LIB
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    over = {}                       # the production setting (bench.py's) unless overridden on the command line: key=value ...
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        over[k] = float(v) if ("." in v or "e" in v) else int(v)
    host = make_batch(batch, n)
    dev = torch.device("cuda", 0)
    ref, bounds, scal = (torch.from_numpy(host[k]).to(dev) for k in ("ref", "bounds", "scal"))
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
    info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
    h = capi.Handle(capi.production_params(**over), device=0, max_batch=batch, max_n=n)
    h.set_option(capi.OPT_STORE_WARM, 0); h.set_option(capi.OPT_ORDER_BY_COST, int(os.environ.get("PQP_ORDER", "1")))     # bench.py's handle options
    for _ in range(3):
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, info=info)
    h.sync()
    t = info.cpu().numpy() / 100.0          # ticks of 10 ns -> microseconds
    names = ["begin_pass(assemble+ruiz+factor)", "hot loop (iterate+residuals+policy)", "refactor(polish set)", "end_pass(unpack)", "iterate", "residuals",
             "finish phase (record, stores)", "TOTAL"]
    print(f"batch {batch} n {n} {over}: kernel {h.last_kernel_ms():.3f} ms")
    for k, nm in enumerate(names):
        print(f"  {nm:36s} mean {t[:, k].mean():8.1f} us   p99 {np.percentile(t[:, k], 99):8.1f}   max {t[:, k].max():8.1f}")
    print(f"  sum of means (0..6) {t[:, :7].sum(1).mean():.1f} us")
    tk = out.cpu().numpy().reshape(batch, -1)[:, 8] / 100.0
    print(f"    kernel loop: ticket -> QP known     mean {tk.mean():8.1f} us   max {tk.max():8.1f}")
    sch = out.cpu().numpy().reshape(batch, -1)[:, 9:12]
    t0 = sch[:, 0].min(); st, en, slot = (sch[:, 0] - t0) / 100.0, (sch[:, 1] - t0) / 100.0, sch[:, 2].astype(int)
    nslot = slot.max() + 1
    slot_end = np.zeros(nslot); slot_busy = np.zeros(nslot); np.maximum.at(slot_end, slot, en); np.add.at(slot_busy, slot, en - st)
    last = int(np.argmax(en))
    print(f"    schedule: {nslot} slots, span {en.max():.1f} us, busy {slot_busy.sum() / (nslot * en.max()):.2f} of slot time; the QP that ends last: qp {last} "
          f"start {st[last]:.1f} end {en[last]:.1f} (slot's QPs: {int((slot == slot[last]).sum())}); longest QP {np.max(en - st):.1f} us starts at {st[np.argmax(en - st)]:.1f}")
    print(f"    slot end times: p10 {np.percentile(slot_end, 10):.0f} p50 {np.percentile(slot_end, 50):.0f} p90 {np.percentile(slot_end, 90):.0f} p99 {np.percentile(slot_end, 99):.0f} max {slot_end.max():.0f} us;"
          f" first-round starts: max {np.sort(st)[nslot - 1]:.1f} us")
    dur = en - st
    print("    the 12 longest QPs (us, start): " + "  ".join(f"{dur[q]:.0f}@{st[q]:.0f}" for q in np.argsort(-dur)[:12]))
    first = np.argsort(st)[:nslot]
    print(f"    first-round QPs: duration min {dur[first].min():.0f} p10 {np.percentile(dur[first], 10):.0f} median {np.median(dur[first]):.0f}; later QPs: median {np.median(np.delete(dur, first)):.0f} p90 {np.percentile(np.delete(dur, first), 90):.0f} max {np.delete(dur, first).max():.0f}")
    if MASK == "0x0":        # the build with shader-clock ticks inside iterate() (and nothing else timed)
        fa = out.cpu().numpy().reshape(batch, -1)[:, 20:28]
        it = out.cpu().numpy().reshape(batch, -1)[:, 12:20]
        kk = info.cpu().numpy()[:, 5] if False else None
        tot = it.sum(1).mean()
        for k, nm in enumerate(["I1 right-hand side", "forward levels 1..8 (DPP)", "forward levels 16, 32 (in-wave)", "barrier + cross-wave levels / root", "backward levels >= 16",
                                "backward levels 8..1 (DPP)", "I3 update"]):
            print(f"    iterate: {nm:36s} mean {it[:, k].mean() / 100.0:8.2f} us per QP  {it[:, k].mean() / tot:5.1%}")
        print(f"    iterate: total {tot / 100.0:.1f} us per QP")
        ftot = fa.sum(1).mean()
        for k, nm in enumerate(["F1 own block + message", "F2 receive", "level 1", "levels 2 .. 8", "levels 16, 32", "levels >= 64"]):
            print(f"    factor:  {nm:36s} mean {fa[:, k].mean() / 100.0:8.2f} us per QP  {fa[:, k].mean() / ftot:5.1%}")
        print(f"    factor:  total {ftot / 100.0:.1f} us per QP")
    sub = out.cpu().numpy().reshape(batch, -1)[:, :8] / 100.0      # the timing build writes the cold operations' sub-times over out[qp][0][0..7]
    for k, nm in enumerate(["load", "assemble", "ruiz", "start_transition_rows", "polish begin / apply set", "factor", "polish update set", "polish end (reject)"]):
        print(f"    cold: {nm:28s} mean {sub[:, k].mean():8.1f} us   max {sub[:, k].max():8.1f}")
