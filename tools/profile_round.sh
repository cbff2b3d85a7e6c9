#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats of the bench command, PMC passes (instruction
# mix at a reduced size, HBM traffic at the bench's own launch size), bench line.
# usage: tools/profile_round.sh TAG    (writes gpurun_out/TAG/...; copy what is to be judged into profiles/)
tag=${1:-r05}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
# 1. per-kernel time of the bench command itself (no counters in this run).  --headline-only: the timed steps and the
#    per-kernel timing pass, every launch at the bench's own launch size -- the csv's averages ARE bench-size launches
#    (with the host-entry-point legs in the same trace the averages mixed launch sizes)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-lines --headline-only > $o/trace.log 2>&1
grep '^{"metric' $o/trace.log | tail -1 > $o/bench_under_rocprof.json
find $o/trace -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
# 2. instruction mix / stalls (SQ counters), one run per counter set, reduced batch
bash tools/pmc_run.sh $o/pmc > $o/pmc_summary.txt 2>&1
# 3. HBM bytes per launch at the bench's launch size: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/traffic/$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-lines --headline-only > $o/traffic_$c.log 2>&1 || echo "$c pass failed"
done
pass=$(python -c "import json;d=json.loads(open('$o/bench_under_rocprof.json').read());print(d['roofline']['solves_per_launch'])")
python tools/make_traffic_profile.py $o/traffic $o/traffic.json $pass 16 33 > /dev/null
python tools/make_valu_profile.py $o/traffic/SQ_INSTS_VALU $o/valu.json $pass 16 33 > /dev/null
# 4. the bench line proper (with the CPU baselines)
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $o/bench.json
cat $o/bench.json
head -12 $o/kernel_stats.csv
