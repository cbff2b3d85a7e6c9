#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Builds the *reference* (paulricchiazzi/SBDART) from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored,
# but shipped to the GPU box).  Nothing from the reference is copied into the
# repository; only binaries land in oracle/_ref/.
#
#   oracle/_ref/sbdart_ref       the unmodified reference executable (makefile:17-33 order)
#   oracle/_ref/sbdart_capture   same objects, with DISORT/depthscl/filter call
#                                sites routed through oracle/ref/sbd_ref_capture.f90
#                                (symbol renames by llvm-objcopy -- no source edits); also writes
#                                the level altitudes/pressures (absint call site) beside the records
#   oracle/_ref/disort_ref_cli   reference DISORT behind a record-file CLI
#                                (oracle/ref/sbd_ref_cli.f90)
#   oracle/_ref/libsbdart_ref.so the reference's routines as a shared library (ctypes: table extraction,
#                                band-model unit parity)
#   oracle/_ref/ref_units_cli    single reference routines (QGAUSN, PLKAVG, ASYMTX,
#                                LEPOLY, SGBCO/SGBSL) behind a CLI (oracle/ref/sbd_ref_units.f90)
#
# Compiler: amdflang (flang, ROCm 7.2); gfortran is not in the image.  The
# flang runtime is linked statically, so the binaries run on the GPU box.
set -euo pipefail
REF=${SBD_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
OBJ="$OUT/obj"
FC=${FC:-/opt/rocm/bin/amdflang}
OBJCOPY=${OBJCOPY:-/opt/rocm/lib/llvm/bin/llvm-objcopy}
FFLAGS=${FFLAGS:--O2}

if [ ! -d "$REF" ]; then
  echo "build_ref.sh: $REF not present -- keeping prebuilt oracle/_ref (if any)" >&2
  exit 0
fi
mkdir -p "$OBJ"
cd "$OBJ"
for f in params tauaero taugas spectra taucloud atms disutil disort drt; do
  if [ ! -f $f.o ] || [ "$REF/$f.f" -nt $f.o ]; then
    "$FC" $FFLAGS -ffixed-form -w -c "$REF/$f.f" -o $f.o
  fi
done
"$FC" $FFLAGS -o "$OUT/sbdart_ref" params.o tauaero.o taugas.o spectra.o taucloud.o atms.o disutil.o disort.o drt.o

# --- capture build: rename the three reference definitions, link wrappers ---
"$OBJCOPY" --redefine-sym disort_=disort_ref_     disort.o  disort_cap.o
"$OBJCOPY" --redefine-sym depthscl_=depthscl_ref_ --redefine-sym absint_=absint_ref_ --redefine-sym readk_=readk_ref_ taugas.o  taugas_cap.o
"$OBJCOPY" --redefine-sym filter_=filter_ref_     spectra.o spectra_cap.o
"$FC" $FFLAGS -c "$HERE/ref/sbd_ref_capture.f90" -o sbd_ref_capture.o
"$FC" $FFLAGS -o "$OUT/sbdart_capture" params.o tauaero.o taugas_cap.o spectra_cap.o \
     taucloud.o atms.o disutil.o disort_cap.o drt.o sbd_ref_capture.o

# --- reference DISORT behind a CLI (BDREF stubbed: Lambertian only) ---
"$FC" $FFLAGS -c "$HERE/ref/sbd_ref_cli.f90" -o sbd_ref_cli.o
"$FC" $FFLAGS -o "$OUT/disort_ref_cli" sbd_ref_cli.o disort.o disutil.o params.o
# --- individual reference routines behind a CLI (unit pinning) ---
"$FC" $FFLAGS -c "$HERE/ref/sbd_ref_units.f90" -o sbd_ref_units.o
"$FC" $FFLAGS -o "$OUT/ref_units_cli" sbd_ref_units.o disort.o disutil.o params.o
# --- the reference as a shared library (position-independent objects of the same sources): single
#     routines (atms, absint, taugas, gasset, rayleigh, solirr ...) are called through ctypes by
#     tools/extract_tables.py and by the band-model parity tests ---
mkdir -p "$OUT/pic"
( cd "$OUT/pic"
  for f in params tauaero taugas spectra taucloud atms disutil disort drt; do
    if [ ! -f $f.o ] || [ "$REF/$f.f" -nt $f.o ]; then
      "$FC" $FFLAGS -fPIC -ffixed-form -w -c "$REF/$f.f" -o $f.o
    fi
  done
  "$FC" -shared -o "$OUT/libsbdart_ref.so" params.o tauaero.o taugas.o spectra.o taucloud.o atms.o disutil.o disort.o drt.o )
echo "built: $(ls "$OUT" | grep -v obj | tr '\n' ' ')"
