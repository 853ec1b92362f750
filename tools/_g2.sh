timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for a in "" "--batch 8192" "--batch 8192 --n 120" "--batch 4096 --n 200" "--profile varied" "--batch 8192 --n 120 --profile varied" "--inflight 2"; do
  echo "== $a"; python bench.py --no-cpu-baseline --steps 20 $a 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],3), d['admm_iters'], d['kkt_solves'], d['factorisations'], d['solved'])"
done
bash tools/seed_sweep.sh
