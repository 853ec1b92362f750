"""Alternative builds of libpqp_hip.so for A/B experiments on the GPU box: build_variants/libpqp_<name>.so, loaded through PQP_LIB.
Usage: python tools/build_variants.py name=DEFINE[,DEFINE...] ...      (e.g. occ2=PQP_SOLVE_OCC=2)"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def one(spec):
    name, _, defs = spec.partition("=")
    out = os.path.join(ROOT, "build_variants", f"libpqp_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    g.build_hip(defines=[d for d in defs.split(",") if d], out=out)
    return out

with ThreadPoolExecutor(max_workers=6) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print("built", o)
