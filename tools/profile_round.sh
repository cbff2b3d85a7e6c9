#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats of the bench command, PMC passes, bench line.
# usage: tools/profile_round.sh TAG    (writes gpurun_out/TAG/...)
tag=${1:-r01}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $o/trace.log 2>&1
tail -1 $o/trace.log > $o/bench_under_rocprof.json
find $o/trace -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
bash tools/pmc_run.sh $o/pmc > $o/pmc_summary.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 > $o/bench.json
cat $o/bench.json
head -8 $o/kernel_stats.csv
