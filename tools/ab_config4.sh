for v in ${VARIANTS:-base}; do lib=$PWD/build_variants/libpqp_$v.so; [ $v = base ] && lib=$PWD/path_optimizer_2_amd/csrc/libpqp_hip.so; echo "== $v"
  PQP_LIB=$lib timeout 100 python bench.py --config 4 --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[4] %.0f scen/s %.3f ms' % (d['value'], d['ms_per_step']))"
  PQP_LIB=$lib timeout 100 python tools/stragglers.py 2048 300 varied seed=1004 2>&1 | grep -v amdgpu | head -2 | tail -1
  PQP_LIB=$lib timeout 100 python tools/stragglers.py 2048 200 uniform seed=1002 2>&1 | grep -v amdgpu | head -2 | tail -1
done
