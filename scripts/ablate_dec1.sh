for v in 0 16 32 48 64 112; do
  python bench.py --workload c4 --users 1600 --steps 1 --warmup 1 --no-cpu-baseline --variant $v 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('variant $v', d['kernels_us']['dec1_fwd'], d['kernels_us']['dh2'])"
done
