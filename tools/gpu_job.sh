#!/bin/bash
# One GPU call, parametrised (replaces the per-call scripts of earlier rounds):  tools/gpu_job.sh <tag> <section> [<section> ...]
# Outputs go to gpurun_out/<tag>_*; copy what is to be judged into profiles/.
#   tests              python -m pytest tests -m gpu
#   bench              the driver's default line + configs 2, 3 (shard), 4 + one launch at a time + the stream kernel's 65 536
#   ab                 A/B of library builds on the path-QP lines: LIBS="name=path name=path ..." (default: the shipped build), REPS repetitions
#                      alternating between the builds (boxes differ by +-1.5 %: only same-call comparisons count)
#   valumix            rocprofv3 --pmc passes over the default bench command: SQ_INSTS_VALU against its fp64 classes -> <tag>_valumix.json
#   timeline           tools/kernel_timeline.py (the -DPQP_TIMING build must have been built before the call: python tools/kernel_timeline.py build)
#   nsweep             8192 QPs of N = 48 ... 128
#   counters           rocprofv3 -L (the counter list of the box)
tag=$1; shift
o=gpurun_out/${tag}; mkdir -p gpurun_out
export TMPDIR=/tmp
root=$PWD
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
line() { python -c "
import json,sys
ls=[l for l in sys.stdin.readlines() if l.startswith('{\"metric\"')]
if not ls: print('$1: no bench line'); sys.exit(0)
d=json.loads(ls[-1])
k, f = d.get('kkt_solves') or d.get('riccati_sweeps') or {}, d.get('factorisations') or d.get('active_set_rounds') or {}
print('%-44s %9.0f paths/s  step %.4f ms  solved %d  kkt %.1f (max %.0f)  fac %.1f (max %.0f)  sha %s' % ('$1', d['value'], d['ms_per_step'], d['solved'], k.get('mean', 0), k.get('max', 0), f.get('mean', 0), f.get('max', 0), d['out_sha1'][:10]))"; }
for sec in "$@"; do case $sec in
tests)
  (timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -15) > ${o}_pytest.log 2>&1; tail -5 ${o}_pytest.log ;;
bench)
  (time timeout 900 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
  timeout 300 python bench.py --steps 20 --warmup 3 > ${o}_bench_driver_style.json 2> /dev/null
  timeout 600 python bench.py --config 3 --batch 65536 --steps 40 --no-cpu-baseline > ${o}_bench_stream_65536.json 2> /dev/null
  timeout 300 python bench.py --config 2 --steps 40 --no-cpu-baseline > ${o}_bench_config2.json 2> /dev/null
  timeout 300 python bench.py --config 3 --steps 100 --no-cpu-baseline > ${o}_bench_config3_shard.json 2> /dev/null
  timeout 300 python bench.py --config 4 --steps 200 --no-cpu-baseline > ${o}_bench_config4.json 2> /dev/null
  for f in n1 driver_style stream_65536 config2 config3_shard config4; do cat ${o}_bench_$f.json | line $f; done | tee ${o}_bench_summary.txt
  (cd /tmp && rm -rf /tmp/rp1 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp1 -- python $root/bench.py $Q > ${root}/${o}_bench_n1_under_rocprof.json 2> /dev/null)
  cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) ${o}_bench_n1_kernel_stats.csv; head -3 ${o}_bench_n1_kernel_stats.csv ;;
ab)
  LIBS=${LIBS:-"shipped=$root/path_optimizer_2_amd/csrc/libpqp_hip.so"}
  for rep in $(seq 1 ${REPS:-3}); do for nl in $LIBS; do name=${nl%%=*}; lib=${nl#*=}
    for a in "--steps 2000" "--steps 1000 --inflight 1" "--config 3 --steps 100" "--batch 8192 --n 64 --steps 60" ${AB_EXTRA:+"$AB_EXTRA"}; do
      PQP_LIB=$lib timeout 300 python bench.py $a $Q 2>/dev/null | line "$name | $a"
    done; done; done | tee ${o}_ab.txt ;;
valumix)
  cd /tmp
  i=0
  for pmc in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVES"; do
    i=$((i+1)); rm -rf /tmp/vm_$i
    rocprofv3 --kernel-trace --pmc $pmc -f csv -d /tmp/vm_$i -- python $root/bench.py $Q --prewarm 0 --steps 10 --warmup 3 ${VALUMIX_ARGS} > /tmp/vm_$i.log 2>&1 || tail -3 /tmp/vm_$i.log
  done
  cd $root
  mkdir -p /tmp/vm && rm -rf /tmp/vm/* && for k in 1 2 3; do [ -d /tmp/vm_$k ] && cp -r /tmp/vm_$k /tmp/vm/pmc_$k; done
  python tools/pmc_aggregate.py /tmp/vm path_solve_kernel ${o}_valumix.json | python -c "
import json,sys
d=json.load(sys.stdin); v=d.get('SQ_INSTS_VALU',0) or 1
f=sum(d.get('SQ_INSTS_VALU_%s_F64'%k,0) for k in ('ADD','MUL','FMA','TRANS'))
print('VALU instructions per launch %.4g; fp64 add/mul/fma/trans %.4g = %.3f of VALU' % (v, f, f/v))
for k in sorted(d):
    if k.startswith('SQ_'): print('  %-28s %.5g' % (k, d[k]))" | tee ${o}_valumix.txt ;;
timeline)
  for m in "" 0x10 0x20 0x2000; do
    if [ -z "$m" ]; then lib=libpqp_hip_timing.so; else lib=libpqp_hip_timing_$m.so; fi
    [ -f path_optimizer_2_amd/csrc/$lib ] || continue
    echo "== PQP_TIMING_MASK=${m:-all}"; PQP_TIMING_MASK=$m timeout 300 python tools/kernel_timeline.py ${TIMELINE_ARGS:-1024 80} 2>&1 | grep -v "$F"
  done | tee ${o}_timeline.txt ;;
nsweep)
  for nn in 48 60 64 80 96 128; do timeout 200 python bench.py --batch 8192 --n $nn --steps 40 $Q 2>/dev/null | line "batch 8192 N = $nn"; done | tee ${o}_n_sweep_batch8192.txt ;;
counters)
  (cd /tmp && rocprofv3 -L 2>&1 | grep -o "SQ_INSTS_[A-Z0-9_]*\|SQ_ACTIVE_[A-Z0-9_]*\|SQ_WAIT[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*" | sort -u | tr '\n' ' ') > ${o}_counters.txt; cat ${o}_counters.txt ;;
*) echo "unknown section $sec" ;;
esac; done
