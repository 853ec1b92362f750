"""An experimental build of the library for A/B runs: pqp_path_stream.hip compiled with extra definitions, linked with the shipped objects of the other
translation units -> ab/<name>/libpqp_hip.so.     python tools/build_stream_variant.py <name> [-DPQP_X ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
G.build_hip()
odir = os.path.join(ROOT, "ab", name)
os.makedirs(odir, exist_ok=True)
(src, defs, o), = [u for u in G.HIP_UNITS if u[0] == "pqp_path_stream.hip"]
for f in (os.path.join(odir, o), os.path.join(odir, o + ".remarks")):
    if os.path.exists(f):
        os.remove(f)
obj, text = G.compile_unit(src, defs + extra, o, odir=odir)
objs = [obj] + [os.path.join(G.CSRC, "build", "libpqp_hip", oo) for s, _, oo in G.HIP_UNITS if s != "pqp_path_stream.hip"]
G.link_units(objs, os.path.join(odir, "libpqp_hip.so"))
print("".join(l + "\n" for l in text.splitlines() if "ILb1" in l or "Spill" in l or "Scratch" in l)[:1500])
