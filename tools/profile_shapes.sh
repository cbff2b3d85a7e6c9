#!/bin/bash
# Other BASELINE shapes on the GPU box, profiled like the headline: config D/E (NSTR 32, 50 layers, flux) and config C
# (radiance NSTR 32, 20 x 16 angles) -- each through `bench.py --shape X`, the very line bench.py's other_shapes prints
# (the same sweep, the same launch sizes: VERDICT r04 #2 -- cfgC's counters were recorded at 512 solves per launch while the
# bench launches 511, and were refused): rocprofv3 kernel-trace stats, the SQ counter sets of tools/pmc_run.sh, and
# FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in passes of their own -> <shape>_traffic.json, <shape>_valu.json (what
# bench.py's other_shapes reads, by kernel-source hash).
# usage: tools/profile_shapes.sh TAG   (writes gpurun_out/TAG/...; copy what is to be judged into profiles/)
tag=${1:-r05_shapes}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for S in D C 40; do      # (40: params.f's largest stream count on the headline's layers -- band_rows_kernel)
  pre=$([ $S = 40 ] && echo nstr40 || echo cfg$S)
  cmd="python bench.py --shape $pre"
  key=$([ $S = D ] && echo cfgD_nstr32_50layers_flux || ([ $S = C ] && echo cfgC_nstr32_radiance_20x16 || echo nstr40_33layers_flux))
  nstr=$([ $S = 40 ] && echo 40 || echo 32); nlyr=$([ $S = D ] && echo 50 || echo 33)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace$S -- $cmd --headline-only > $o/trace$S.log 2>&1   # (no serialized pass: every launch in the timed configuration)
  grep '^{"shape' $o/trace$S.log | tail -1 > $o/${pre}_bench_under_rocprof.json
  find $o/trace$S -name "*kernel_stats.csv" -exec cp {} $o/${pre}_kernel_stats.csv \;
  W=$(python -c "import json;d=json.loads(open('$o/${pre}_bench_under_rocprof.json').read());print(d['$key']['roofline']['solves_per_launch'])")
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmc$S/$c -- $cmd > $o/pmc${S}_$c.log 2>&1 || echo "$S $c pass failed"
  done
  python tools/make_traffic_profile.py $o/pmc$S $o/${pre}_traffic.json $W $nstr $nlyr > /dev/null
  python tools/make_valu_profile.py $o/pmc$S/SQ_INSTS_VALU $o/${pre}_valu.json $W $nstr $nlyr > /dev/null
  PMC_CMD="$cmd" bash tools/pmc_run.sh $o/sq$S > $o/${pre}_pmc.txt 2>&1
  head -8 $o/${pre}_kernel_stats.csv
  python -c "import json;d=json.loads(open('$o/${pre}_bench_under_rocprof.json').read())['$key'];print('$key', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
