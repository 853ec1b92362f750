for v in ${VARIANTS:-base c8}; do
  if [ $v = base ]; then lib=$PWD/path_optimizer_2_amd/csrc/libpqp_hip.so; else lib=$PWD/build_variants/libpqp_$v.so; fi
  for seed in ${SEEDS:-default 6 13}; do
    s=""; [ $seed != default ] && s="--seed $seed"
    PQP_LIB=$lib timeout 120 python bench.py --no-cpu-baseline --steps 200 --warmup 8 --pmc off --sustain 0 $s 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print('$v seed %-7s value %.3f M | one at a time %.3f M | solved %d sha %s | kkt solves mean %.1f max %d factorisations mean %.1f max %d' % ('$seed', d['value']/1e6, s['one_batch_at_a_time']['value']/1e6, d['solved'], d['out_sha1'], d['kkt_solves']['mean'], d['kkt_solves']['max'], d['factorisations']['mean'], d['factorisations']['max']))"
  done
done
