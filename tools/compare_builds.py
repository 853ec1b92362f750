"""Two builds of the library on the same batches: how far apart are their paths?  (Run on the GPU box; each build in a process of its own.)
Usage: python tools/compare_builds.py <libA> <libB> [batch n profile]...   - prints max / p99 / median |A - B| over (l, d_heading) and the statuses"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
out = {}
for spec in sys.argv[2:]:
    batch, n, profile = spec.split(":")
    b = make_batch(int(batch), int(n), profile)
    h = capi.Handle(capi.production_params(), device=0, max_batch=int(batch), max_n=int(n))
    h.set_option(capi.OPT_STORE_WARM, 0)
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h.close()
    out[spec + "_out"] = r["out"]; out[spec + "_status"] = r["status"]; out[spec + "_info"] = r["info"]
np.savez(sys.argv[1], **out)
''' % ROOT
libs = sys.argv[1:3]
specs = sys.argv[3:] or ["1024:80:uniform", "8192:80:uniform", "8192:120:varied", "512:200:uniform", "8192:64:uniform"]
res = []
for lib in libs:
    f = tempfile.mktemp(suffix=".npz")
    subprocess.run([sys.executable, "-c", CHILD, f, *specs], env=dict(os.environ, PQP_LIB=os.path.abspath(lib)), check=True, stderr=subprocess.DEVNULL)
    res.append(np.load(f))
for spec in specs:
    a, b = res[0][spec + "_out"], res[1][spec + "_out"]
    d = np.abs(a[:, :, 3:5] - b[:, :, 3:5]).max(axis=(1, 2))
    same = int((d == 0).sum())
    print(f"{spec:18s} statuses equal {bool((res[0][spec + '_status'] == res[1][spec + '_status']).all())}  solves / factorisations equal "
          f"{bool((res[0][spec + '_info'][:, 5:7] == res[1][spec + '_info'][:, 5:7]).all())}  paths bit-identical {same} of {len(d)}  |A - B| median {np.median(d):.1e} p99 {np.percentile(d, 99):.1e} max {d.max():.1e}")
