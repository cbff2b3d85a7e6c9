#!/bin/bash
# profiling helper: kernel times with parts of the work disabled (results are WRONG with flags != 0)
# bit0 one Jacobi sweep, bit1 no UPBEAM LU, bit2 no U/y stores in the band kernel, bit3 no row prefetch loads
for f in ${@:-0 1 2 3}; do
  SBD_DBG_FLAGS=$f python bench.py --steps 2 --warmup 1 --nwl 16384 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags $f layer_ms', d['kernel_ms']['layer_kernel'], 'band_ms', d['kernel_ms']['band_kernel'])"
done
