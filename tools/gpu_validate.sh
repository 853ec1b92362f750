mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
rm -f gpurun_out/b_var.log
for a in "--batch 8192" "--batch 8192 --n 120" "--batch 4096 --n 200" "--batch 4096 --n 200 --polish-max-rounds 8" "--batch 1024 --n 120" "--profile varied"; do
  echo "== $a" >> gpurun_out/b_var.log
  timeout 300 python bench.py --no-cpu-baseline --steps 20 $a 2>&1 | tail -1 >> gpurun_out/b_var.log
done
python tools/kernel_timeline.py 1024 80 > gpurun_out/timeline.log 2>&1
cat gpurun_out/t_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/bench_n1.json
python - <<'PY'
import json
for l in open('gpurun_out/b_var.log'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); print(round(d['value']), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],3), d['kkt_solves'], d['factorisations'], d['solved'], d['polished'])
    except Exception as e: print(l[:300])
PY
cat gpurun_out/timeline.log
