#!/bin/bash
# round 4: GPU test suite + default bench line on the current build
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['out_sha1'], d['solved'], d['roofline']['traffic_source'][:120])"
timeout 300 python __graft_entry__.py 2>&1 | tail -1; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1
