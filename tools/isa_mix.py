"""Instruction mix of the gfx950 code objects in the built library, per kernel, per loop region and (with line tables) per source function.

    python tools/isa_mix.py                                   # path_solve_kernel<2,false> of the built libpqp_hip.so
    python tools/isa_mix.py --kernel 'path_solve_kernelILi1ELb0' --regions 12
    python tools/isa_mix.py --lib some.o --lines              # an object built with -gline-tables-only: mix per source function

Static counts: what the compiler emitted, not what ran (the PMC pass of tools/pmc_valu_mix.sh gives the dynamic shares).  Classes:
  f64      fp64 arithmetic / compare / conversion (v_*_f64, v_cmp*_f64, v_rcp/rsq_f64, v_ldexp_f64, v_cvt_*f64*)
  agpr     v_accvgpr_read / v_accvgpr_write  (AGPR <-> VGPR copies: values parked in the accumulator half of the register file)
  lane     v_readlane / v_writelane / v_readfirstlane (spilled SGPRs, wave-uniform values, cross-lane reads)
  mov      v_mov / v_cndmask without DPP (copies and selects)
  dpp      any VALU instruction with a DPP modifier (row_shr, row_shl, row_bcast, quad_perm ...)
  valu     every other VALU instruction (integer, fp32, bit operations)
  ds       LDS, mem: global / flat / scratch / buffer, salu: scalar ALU + s_load, wait: s_waitcnt / s_nop / s_barrier, br: branches
A loop region is [target, branch] of a backward branch; nested regions are listed on their own (innermost first by address).

    python tools/isa_mix.py --lds-trips [--lines]             # LDS round trips per barrier interval
An LDS round trip = an s_waitcnt that completes a ds_read issued after the previous such wait: a load -> wait -> use chain.  At one wavefront per
SIMD nothing hides them; a phase that needs k values from LDS should show ONE (all loads, one wait), not k (round 6: the root of the
cyclic-reduction tree showed 7, the end rows 7, the workgroup reduction 6).  With --lines every interval names the source lines it spans."""
import argparse
import bisect
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CLASSES = ["f64", "agpr", "lane", "mov", "dpp", "valu", "ds", "mem", "salu", "wait", "br"]
VALU = ["f64", "agpr", "lane", "mov", "dpp", "valu"]


def classify(mn, ops):
    if mn.startswith("v_accvgpr"):
        return "agpr"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if mn.startswith("v_"):
        if re.search(r"\b(row_shr|row_shl|row_ror|row_bcast|row_mirror|row_half_mirror|quad_perm|wave_shr|wave_shl|row_newbcast|row_share)", ops):
            return "dpp"
        if "f64" in mn:
            return "f64"
        if mn.startswith(("v_mov_b", "v_cndmask", "v_pk_mov", "v_swap")):
            return "mov"
        return "valu"
    if mn.startswith("ds_"):
        return "ds"
    if mn.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "mem"
    if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call", "s_getpc")):
        return "br"
    if mn.startswith("s_"):
        return "salu"
    return "valu"


def code_objects(lib, tmp):
    """gfx950 code objects inside a host library / object (offload bundles), or the file itself when it already is one."""
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-h", lib], capture_output=True, text=True).stdout
    if "AMDGPU" in out or "EM_AMDGPU" in out:
        return [lib]
    cp = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, cp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], capture_output=True, text=True, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
SYM = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
LINE = re.compile(r"^; (\S+):(\d+)")


def disassemble(co, want, lines):
    """-> OrderedDict kernel -> list of (addr, class, mnemonic, operands, (file, line) or None)"""
    cmd = [f"{LLVM}/llvm-objdump", "-d"] + (["-l"] if lines else []) + [co]
    txt = subprocess.run(cmd, capture_output=True, text=True).stdout
    kernels, cur, loc = OrderedDict(), None, None
    for ln in txt.splitlines():
        m = SYM.match(ln)
        if m:
            name = m.group(2)
            cur = kernels.setdefault(name, []) if re.search(want, name) else None
            loc = None
            continue
        if cur is None:
            continue
        m = LINE.match(ln)
        if m:
            loc = (os.path.basename(m.group(1)), int(m.group(2)))
            continue
        m = INSN.match(ln)
        if m:
            mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
            cur.append((addr, classify(mn, ops), mn, ops, loc))
    return kernels


def fmt(counts, total=None):
    n = sum(counts.values())
    v = sum(counts[c] for c in VALU)
    moves = counts["agpr"] + counts["lane"]
    s = f"{n:6d} insns | VALU {v:5d}: f64 {counts['f64']:5d} ({counts['f64'] / max(v, 1):4.0%})  agpr {counts['agpr']:4d}  lane {counts['lane']:4d}  " \
        f"(agpr+lane {moves / max(v, 1):4.0%})  mov {counts['mov']:4d}  dpp {counts['dpp']:4d}  other {counts['valu']:4d} | " \
        f"ds {counts['ds']:4d}  mem {counts['mem']:4d}  salu {counts['salu']:5d}  wait {counts['wait']:4d}  br {counts['br']:4d}"
    return s


def regions(insns):
    """loop regions from backward branches: [(start_addr, end_addr)] sorted by start; duplicates by start keep the longest"""
    addrs = [a for a, *_ in insns]
    regs = {}
    for i, (a, cls, mn, ops, _) in enumerate(insns):
        if cls != "br" or not mn.startswith(("s_cbranch", "s_branch")):
            continue
        m = re.match(r"(-?\d+)", ops)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt <= a:
            regs[tgt] = max(regs.get(tgt, 0), a)
    return sorted(regs.items()), addrs


def source_functions(paths):
    """line -> enclosing function of the solver sources: (file, [(first_line, name)]) by a scan for function headers"""
    table = {}
    head = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:PQP_HD|__device__|__global__|inline|static)[^;=]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\([^;]*$")
    for p in paths:
        rows = []
        for i, ln in enumerate(open(p), 1):
            m = head.match(ln)
            if m and not ln.lstrip().startswith(("//", "return", "if", "for")):
                rows.append((i, m.group(1)))
        table[os.path.basename(p)] = rows
    return table


def lds_trips(insns, min_insns):
    """[(first index, last index, instructions, VALU, trips)] per interval between s_barrier instructions"""
    from collections import deque
    out, q, trips, start, n, valu, at = [], deque(), 0, 0, 0, 0, []
    for i, (addr, cls, mn, ops, loc) in enumerate(insns):
        n += 1
        valu += cls in VALU
        if cls == "ds":
            q.append((trips, mn.startswith("ds_read") or "_rtn" in mn, loc))
        elif mn.startswith("s_waitcnt") and "lgkmcnt" in ops:
            c = int(re.search(r"lgkmcnt\((\d+)\)", ops).group(1))
            hit = None
            while len(q) > c:
                g, is_read, where = q.popleft()
                if is_read and g == trips:
                    hit = where or ("?", 0)
            if hit:
                trips += 1
                at.append(hit[1])
        elif mn.startswith("s_barrier") or i == len(insns) - 1:
            if n >= min_insns or trips > 1:
                out.append((start, i, n, valu, trips, at))
            q, trips, start, n, valu, at = deque(), 0, i + 1, 0, 0, []
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--lib", default=os.path.join(ROOT, "path_optimizer_2_amd", "csrc", "libpqp_hip.so"))
    ap.add_argument("--kernel", default="path_solve_kernelILi2ELb0", help="regex on the (mangled) kernel name")
    ap.add_argument("--regions", type=int, default=16, help="loop regions to list per kernel (largest first)")
    ap.add_argument("--lines", action="store_true", help="the object carries line tables: mix per source function")
    ap.add_argument("--min-insns", type=int, default=60, help="loop regions smaller than this are not listed")
    ap.add_argument("--lds-trips", action="store_true", help="LDS round trips (load -> wait -> use chains) per barrier interval instead of the mix")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    try:
        cos = code_objects(a.lib, tmp)
        if not cos:
            sys.exit(f"no gfx950 code object in {a.lib}")
        for co in cos:
            for name, insns in disassemble(co, a.kernel, a.lines).items():
                if not insns:
                    continue
                total = Counter(c for _, c, *_ in insns)
                base = insns[0][0]
                print(f"== {name}")
                if a.lds_trips:
                    rows = lds_trips(insns, a.min_insns)
                    print(f"   {sum(r[4] for r in rows)} LDS round trips in {len(rows)} barrier intervals (static: every interval once)")
                    for s0, s1, n, v, tr, at in rows:
                        where = f"  loads waited for at lines {at}" if a.lines and at else ""
                        print(f"   +{insns[s0][0] - base:#07x}..+{insns[s1][0] - base:#07x} {n:5d} insns  VALU {v:4d}  LDS round trips {tr:2d}{where}")
                    continue
                print(f"   whole kernel      {fmt(total)}")
                regs, addrs = regions(insns)
                rows = []
                for s, e in regs:
                    i0, i1 = bisect.bisect_left(addrs, s), bisect.bisect_right(addrs, e)
                    if i1 - i0 >= a.min_insns:
                        rows.append((i1 - i0, s, e, Counter(c for _, c, *_ in insns[i0:i1])))
                rows.sort(key=lambda r: -r[0])
                for n, s, e, c in rows[:a.regions]:
                    print(f"   +{s - base:#07x}..+{e - base:#07x} {fmt(c)}")
                if a.lines:
                    srcs = [os.path.join(ROOT, "path_optimizer_2_amd", "csrc", f) for f in ("pqp_path_lane.hpp", "pqp_kernels.hip")]
                    tab = source_functions([s for s in srcs if os.path.exists(s)])
                    per = {}
                    for _, cls, _, _, loc in insns:
                        fn = "?"
                        if loc and loc[0] in tab:
                            rows_ = tab[loc[0]]
                            k = bisect.bisect_right([r[0] for r in rows_], loc[1]) - 1
                            fn = f"{loc[0].split('.')[0][4:]}:{rows_[k][1]}" if k >= 0 else loc[0]
                        elif loc:
                            fn = loc[0]
                        per.setdefault(fn, Counter())[cls] += 1
                    print("   -- by source function (innermost inlined line)")
                    for fn, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
                        if sum(c.values()) >= 20:
                            print(f"   {fn[:28]:28s} {fmt(c)}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
