#!/usr/bin/env bash
# INPUT -> stdout wall time of sbdart_amd (band model + engine + writers) on the full short-wave sweep at
# three spectral resolutions, next to the reference executable on the same box.  Run on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; mkdir -p e2e; cd e2e
now() { python3 -c 'import time; print(time.time())'; }
since() { python3 -c "import time,sys; print('total %.3f s' % (time.time() - float(sys.argv[1])))" "$1"; }
for inc in .005 .0002 .00005; do
  printf "\n &INPUT\n idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=$inc nstr=16 iout=10\n /\n" > INPUT
  for th in ${THREADS:-default}; do
    echo "== sbdart_amd, wlinc=$inc, OMP_NUM_THREADS=$th"
    for rep in 1 2 3; do
      t0=$(now)
      if [ "$th" = default ]; then SBD_TIMING=1 SBD_OPTICS=/nonexistent "$ROOT/sbdart_amd/bin/sbdart_amd"
      else OMP_NUM_THREADS=$th SBD_TIMING=1 SBD_OPTICS=/nonexistent "$ROOT/sbdart_amd/bin/sbdart_amd"; fi
      since "$t0"
    done
  done
done
printf "\n &INPUT\n idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 nstr=16 iout=10\n /\n" > INPUT
echo "== reference (one core), wlinc=.005"
t0=$(now); "$ROOT/oracle/_ref/sbdart_ref"; since "$t0"
echo "host cores: $(nproc)"
