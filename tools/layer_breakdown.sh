#!/bin/bash
# profiling helper: layer-kernel time with parts of the work disabled (results are WRONG with flags != 0)
for f in 0 1 2 3; do
  SBD_DBG_FLAGS=$f python bench.py --steps 2 --warmup 1 --nwl 16384 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags $f layer_ms', d['kernel_ms']['layer_kernel'], 'band_ms', d['kernel_ms']['band_kernel'])"
done
