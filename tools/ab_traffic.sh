#!/bin/bash
# Developer A/B: does the layer -> band HBM round trip bind?  Two builds with WRONG results but the
# same instruction streams: (a) the band kernel reads system 0's layer outputs for every system (its 53 KB of HBM reads
# per solve become L2 hits), (b) the layer kernel does not write GC's quarters (33 of its 57 KB per solve).  Built in the
# build container into sbdart_amd/lib/libsbdart_amd_ab{a,b}.so (not kept), timed on the GPU box:
#    tools/ab_traffic.sh build     (here)        tools/ab_traffic.sh run   (gpurun)
cd "$(dirname "$0")/.."
c=sbdart_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC"
if [ "$1" = build ]; then
  mkdir -p /tmp/ab
  /opt/rocm/bin/hipcc $FL -DSBD_AB_SHARED_INPUTS -c -o /tmp/ab/k_band4.o $c/sbd_k_band4.hip
  /opt/rocm/bin/hipcc $FL -DSBD_AB_NO_GC_STORES -c -o /tmp/ab/k_layer2f.o $c/sbd_k_layer2f.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sbdart_amd/lib/libsbdart_amd_aba.so $c/build/sbd_engine.o $(ls $c/build/sbd_k_*.o | grep -v k_band4.o) /tmp/ab/k_band4.o -ldl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sbdart_amd/lib/libsbdart_amd_abb.so $c/build/sbd_engine.o $(ls $c/build/sbd_k_*.o | grep -v k_layer2f.o) /tmp/ab/k_layer2f.o -ldl
else
  for i in 1 2 3; do
    for lib in "" aba abb; do
      echo -n "lib=${lib:-product} "
      if [ -z "$lib" ]; then python tools/bench_switch.py 16 33 49152 | tail -1
      else SBDART_AMD_LIB=sbdart_amd/lib/libsbdart_amd_$lib.so python tools/bench_switch.py 16 33 49152 | tail -1; fi
    done
  done
fi
