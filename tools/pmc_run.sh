#!/bin/bash
# rocprofv3 PMC passes (one run per counter set, no tracing) over a short bench run.
# usage: tools/pmc_run.sh OUTDIR [bench args...]; summary: python tools/pmc_summary.py OUTDIR
# (PMC_CMD="python tools/bench_radiance.py 384": another command instead of bench.py)
out=${1:-gpurun_out/pmc}; shift
args=${@:---steps 1 --warmup 0 --nwl 6144 --no-cpu-baseline --no-side-lines}
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $out
sets=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT"
)
i=0
for s in "${sets[@]}"; do
  timeout 240 rocprofv3 --pmc $s --output-format csv -d $out/set$i -- ${PMC_CMD:-python bench.py $args} > $out/set$i.log 2>&1 || { echo "set $i ($s) failed or timed out"; tail -3 $out/set$i.log; }
  i=$((i+1))
done
python tools/pmc_summary.py $out
