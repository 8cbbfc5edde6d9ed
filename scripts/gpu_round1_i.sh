for b in 8 12 16 24 32; do
  echo "blocks_per_cu $b"
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --blocks-per-cu $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"
done
