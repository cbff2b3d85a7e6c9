#!/bin/bash
# Other BASELINE shapes on the GPU box, profiled like the headline: config D/E (NSTR 32, 50 layers,
# flux; bench.py --nstr 32 --nlyr 50) and config C (radiance NSTR 32, 20 x 16 angles; tools/bench_radiance.py 384):
# rocprofv3 kernel-trace stats, the SQ counter sets of tools/pmc_run.sh, and FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in
# passes of their own -> <shape>_traffic.json, <shape>_valu.json (what bench.py's other_shapes reads, by kernel-source hash).
# usage: tools/profile_shapes.sh TAG   (writes gpurun_out/TAG/...; copy what is to be judged into profiles/)
tag=${1:-r04_shapes}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
cmdD="python bench.py --nstr 32 --nlyr 50 --nwl 6144 --steps 1 --warmup 0 --no-cpu-baseline --no-side-lines"
cmdC="python tools/bench_radiance.py 384"
# kernel-trace stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/traceD -- python bench.py --nstr 32 --nlyr 50 --nwl 6144 --steps 5 --warmup 1 --no-cpu-baseline --no-side-lines > $o/traceD.log 2>&1
grep '^{"metric' $o/traceD.log | tail -1 > $o/cfgD_bench_under_rocprof.json
find $o/traceD -name "*kernel_stats.csv" -exec cp {} $o/cfgD_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/traceC -- $cmdC > $o/traceC.log 2>&1
grep 'radiance nstr' $o/traceC.log | tail -1 > $o/cfgC_bench_under_rocprof.txt
find $o/traceC -name "*kernel_stats.csv" -exec cp {} $o/cfgC_kernel_stats.csv \;
# counters, one pass each
WD=$(python -c "import json;print(json.loads(open('$o/cfgD_bench_under_rocprof.json').read())['roofline']['solves_per_launch'])")   # (one PASS of the batch)
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmcD/$c -- $cmdD > $o/pmcD_$c.log 2>&1 || echo "D $c pass failed"
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmcC/$c -- $cmdC > $o/pmcC_$c.log 2>&1 || echo "C $c pass failed"
done
python tools/make_traffic_profile.py $o/pmcD $o/cfgD_traffic.json $WD 32 50 > /dev/null
python tools/make_valu_profile.py $o/pmcD/SQ_INSTS_VALU $o/cfgD_valu.json $WD 32 50 > /dev/null
python tools/make_traffic_profile.py $o/pmcC $o/cfgC_traffic.json 512 32 33 > /dev/null
python tools/make_valu_profile.py $o/pmcC/SQ_INSTS_VALU $o/cfgC_valu.json 512 32 33 > /dev/null
# SQ counter sets (instruction mix, waits, LDS)
bash tools/pmc_run.sh $o/sqD --nstr 32 --nlyr 50 --nwl 6144 --steps 1 --warmup 0 --no-cpu-baseline --no-side-lines > $o/cfgD_pmc.txt 2>&1
PMC_CMD="$cmdC" bash tools/pmc_run.sh $o/sqC > $o/cfgC_pmc.txt 2>&1
head -8 $o/cfgD_kernel_stats.csv; head -8 $o/cfgC_kernel_stats.csv; cat $o/cfgC_bench_under_rocprof.txt
python -c "import json;d=json.load(open('$o/cfgD_bench_under_rocprof.json'));print(d['value'],d['ms_per_step'],d['kernel_ms'])"
