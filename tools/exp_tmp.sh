run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['solves_per_s']), d['kernel_ms'])"; }
run base
