#!/bin/bash
# Other BASELINE shapes on the GPU box, with rocprof kernel-trace stats: config D/E (nstr 32, 50 layers,
# flux) through bench.py, config C (radiance nstr 32, 20 x 16 angles) through tools/bench_radiance.py.
# usage: tools/profile_shapes.sh TAG   (writes gpurun_out/TAG/...; copy what is to be judged into profiles/)
tag=${1:-r02_shapes}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/traceD -- python bench.py --nstr 32 --nlyr 50 --nwl 12288 --steps 5 --warmup 1 --no-cpu-baseline > $o/traceD.log 2>&1
grep '^{"metric' $o/traceD.log | tail -1 > $o/cfgD_bench_under_rocprof.json
find $o/traceD -name "*kernel_stats.csv" -exec cp {} $o/cfgD_kernel_stats.csv \;
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/traceC -- python tools/bench_radiance.py > $o/traceC.log 2>&1
grep 'radiance nstr' $o/traceC.log | tail -1 > $o/cfgC_bench_under_rocprof.txt
find $o/traceC -name "*kernel_stats.csv" -exec cp {} $o/cfgC_kernel_stats.csv \;
timeout 900 python bench.py --nstr 32 --nlyr 50 --nwl 12288 --steps 10 --warmup 2 2>&1 | tail -1 > $o/cfgD_bench.json
timeout 600 python tools/bench_radiance.py 2>&1 | tail -1 > $o/cfgC_bench.txt
head -8 $o/cfgD_kernel_stats.csv; head -8 $o/cfgC_kernel_stats.csv; cat $o/cfgC_bench.txt
python -c "import json;d=json.load(open('$o/cfgD_bench.json'));print(d['value'],d['ms_per_step'],d['kernel_ms'])"
