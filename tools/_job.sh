cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p gpurun_out/r04a
rocprofv3 -L > gpurun_out/r04a/avail.txt 2>&1
bash tools/pmc_run.sh gpurun_out/r04a/pmcD --nstr 32 --nlyr 50 --nwl 6144 --steps 1 --warmup 0 --no-cpu-baseline --no-side-lines > gpurun_out/r04a/pmcD.txt 2>&1
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU --output-format csv -d gpurun_out/r04a/pmcD/set4 -- python bench.py --nstr 32 --nlyr 50 --nwl 6144 --steps 1 --warmup 0 --no-cpu-baseline --no-side-lines > gpurun_out/r04a/pmcD/set4.log 2>&1
python tools/pmc_summary.py gpurun_out/r04a/pmcD > gpurun_out/r04a/pmcD.txt 2>&1
tail -40 gpurun_out/r04a/pmcD.txt
