#!/bin/bash
# round 4, first GPU call: FETCH_SIZE / WRITE_SIZE calibration, the GPU test suite, the default bench line
set -x
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fetch_calib.py $O/fetch_calib.json > $O/fetch_calib.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.err
cat $O/fetch_calib.log | tail -3
