"""One launch pattern for rocprofv3: batch QPs, fixed number of ADMM iterations (no early exit), 3 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_batch
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
iters_fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 200
host = make_batch(batch, n)
dev = torch.device("cuda", 0)
ref, bounds, scal = (torch.from_numpy(host[k]).to(dev) for k in ("ref", "bounds", "scal"))
out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
prm = capi.default_params(eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=0, max_iter=iters_fixed)
h = capi.Handle(prm, device=0, max_batch=batch, max_n=n)
for _ in range(3):
    h.solve_device(batch, n, ref, bounds, scal, out, passes=0)
h.sync()
print("kernel ms", h.last_kernel_ms())
