run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['solves_per_s']), d['kernel_ms'], d['config']['chunk'])"; }
run nstr4 "--nstr 4"
run nstr8 "--nstr 8 --nwl 16384"
run nstr20 "--nstr 20 --nwl 16384"
run nstr32_50 "--nstr 32 --nlyr 50 --nwl 4096"
run nstr16_nwl751 "--nwl 751"
python tools/bench_radiance.py 2>&1 | tail -2
