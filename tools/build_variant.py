"""An experimental build of the library for A/B runs on the GPU box: the lane-per-waypoint kernels (pqp_path_solve.hip, four widths) compiled with extra
flags / definitions, linked with the shipped objects of the other translation units -> ab/<name>/libpqp_hip.so (+ resources.txt, isa_mix.txt).
    python tools/build_variant.py <name> [-DPQP_X ...] [-mllvm -some-flag ...]
Select it with PQP_LIB=ab/<name>/libpqp_hip.so (tools/gpu_job.sh ab: LIBS="name=path ...").  ab/ is not tracked; it travels to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    G.build_hip()
    odir = os.path.join(ROOT, "ab", name)
    os.makedirs(odir, exist_ok=True)
    units = [(s, d + extra, o) for s, d, o in G.HIP_UNITS if s == "pqp_path_solve.hip"]
    for _, _, o in units:       # (always recompiled: the flags are the variable)
        for f in (os.path.join(odir, o), os.path.join(odir, o + ".remarks")):
            if os.path.exists(f):
                os.remove(f)
    with ThreadPoolExecutor(max_workers=4) as ex:
        done = list(ex.map(lambda u: G.compile_unit(*u, odir=odir), units))
    objs = [o for o, _ in done] + [os.path.join(G.CSRC, "build", "libpqp_hip", o) for s, _, o in G.HIP_UNITS if s != "pqp_path_solve.hip"]
    lib = os.path.join(odir, "libpqp_hip.so")
    G.link_units(objs, lib)
    with open(os.path.join(odir, "resources.txt"), "w") as f:
        f.write("".join(t for _, t in done))
    mix = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), "--lib", lib, "--kernel", "path_solve_kernelILi[12]ELb0", "--regions", "0"],
                         capture_output=True, text=True).stdout
    with open(os.path.join(odir, "isa_mix.txt"), "w") as f:
        f.write(mix)
    import re
    for t in (t for _, t in done):
        for b in re.split(r"(?=remark: [^\n]*Function Name:)", t):
            m = re.search(r"Function Name: \S*path_solve_kernelILi(\d)ELb0", b)
            if m:
                g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
                print(f"  NW={m.group(1)}: VGPR {g(' VGPRs')} AGPR {g('AGPRs')} scratch {g('ScratchSize .bytes/lane.')} spilled SGPR {g('SGPRs Spill')} VGPR {g('VGPRs Spill')}")
    print("\n".join(l[:215] for l in mix.splitlines() if "whole kernel" in l or l.startswith("==")))
    print(lib)


if __name__ == "__main__":
    main()
