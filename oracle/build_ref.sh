#!/usr/bin/env bash
# build_ref.sh -- compile the REFERENCE's own hot-path sources, where they lie
# under /root/reference/src, into oracle/_ref/ (git-ignored, travels with gpurun).
#
# TEST INFRASTRUCTURE ONLY.  Nothing from the reference is copied into the repo:
# the compiler reads the sources in place and only shared objects / one binary
# are written to oracle/_ref/.  The reference's own build system is not used
# (its serial build fails: src/matrixBinfile.c includes mpi.h unconditionally,
# and gcc 11 rejects -std=c23), so the files are compiled directly with ROCm
# clang, which accepts -std=c23.  main.c and matrixBinfile.c are not on the hot
# path and are left out of the serial libraries; oracle/ref_shim.c (ours) is the
# caller instead.
#
# Outputs
#   libsbref_crs.so       CRS, strict IEEE (-O2 -fno-fast-math -ffp-contract=off)
#   libsbref_scs.so       SCS as-is (self-consistent for C=1 only: the reference
#                         clobbers C and sigma, src/matrix-SCS.c:42-43)
#   libsbref_scs_fix.so   SCS with exactly those two assignments deleted by sed
#                         in a temp copy -- a PATCHED
#                         reference, used only as a second opinion next to the
#                         reference's own layout fixtures
#   libsbref_crs_omp.so   CRS, upstream flags -O3 -ffast-math + OpenMP: the
#                         cpu_baseline "reference" leg of bench.py (timing only)
#   sb_ref_mpi            full reference (main.c, MPI) for multi-rank golden
#                         histories, only if an MPI compiler wrapper is present
#   sb_ref_mpi_omp        full reference, upstream flags -O3 -ffast-math + OpenMP + MPI (timing only)
#   refmain_{CRS,SCS}_hip the reference's main.c (unchanged) linked with OUR drop-in
#                         libraries instead of the reference's objects
set -euo pipefail
REF=${SB_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
S=$REF/src
if [ ! -d "$S" ]; then
  echo "build_ref: $S not present (expected on the GPU box) -- using prebuilt oracle/_ref" >&2
  exit 0
fi
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
mkdir -p "$OUT"
DEFS="-DPRECISION=2 -DUINT_TYPE=1 -D_GNU_SOURCE -DARRAY_ALIGNMENT=64 -DOMP_SCHEDULE=static"
STRICT="-O2 -fno-fast-math -ffp-contract=off -std=c23 -w -fPIC"
COMMON="$S/CGSolver.c $S/solver.c $S/matrix.c $S/mmio.c $S/allocate.c $S/comm.c $S/bstree.c $S/timing.c $S/profiler.c $S/util.c"

$CLANG -DCRS $DEFS $STRICT -I"$S" -shared -o "$OUT/libsbref_crs.so" \
  "$HERE/ref_shim.c" $COMMON "$S/matrix-CRS.c" -Wl,--wrap=ddot -Wl,-Bsymbolic -lm

$CLANG -DSCS $DEFS $STRICT -I"$S" -shared -o "$OUT/libsbref_scs.so" \
  "$HERE/ref_shim.c" $COMMON "$S/matrix-SCS.c" -Wl,--wrap=ddot -Wl,-Bsymbolic -lm

TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
sed -e '/m->C        = (CG_UINT)1;/d' -e '/m->sigma    = (CG_UINT)1;/d' "$S/matrix-SCS.c" > "$TMP/matrix-SCS-fix.c"
$CLANG -DSCS $DEFS $STRICT -I"$S" -shared -o "$OUT/libsbref_scs_fix.so" \
  "$HERE/ref_shim.c" $COMMON "$TMP/matrix-SCS-fix.c" -Wl,--wrap=ddot -Wl,-Bsymbolic -lm

# upstream optimisation flags (mk/include_CLANG.mk:15) + OpenMP, timing only
if $CLANG -fopenmp -x c -o /dev/null -c - <<<'int main(void){return 0;}' 2>/dev/null; then
  $CLANG -DCRS $DEFS -O3 -ffast-math -std=c23 -w -fPIC -fopenmp -I"$S" -shared \
    -o "$OUT/libsbref_crs_omp.so" "$HERE/ref_shim.c" $COMMON "$S/matrix-CRS.c" \
    -Wl,--wrap=ddot -Wl,-Bsymbolic -lm -Wl,-rpath,/opt/rocm/lib/llvm/lib || echo "build_ref: OpenMP variant failed (non-fatal)" >&2
fi

# full MPI reference, strict, with the ddot log -> multi-rank golden histories
MPICC=${MPICC:-/opt/conda/bin/mpicc}
if [ -x "$MPICC" ]; then
  MPICH_CC=$CLANG "$MPICC" -DCRS -D_MPI $DEFS -O2 -fno-fast-math -ffp-contract=off -std=c23 -w \
    -I"$S" -o "$OUT/sb_ref_mpi" "$HERE/ref_mpi_log.c" $S/main.c $COMMON "$S/matrix-CRS.c" \
    "$S/matrixBinfile.c" "$S/parameter.c" "$S/affinity.c" -Wl,--wrap=ddot -Wl,-Bsymbolic -lm \
    || echo "build_ref: MPI variant failed (non-fatal)" >&2
fi
# the same full reference with upstream optimisation flags + OpenMP (MPI ranks x OpenMP threads: the hybrid mode
# north_star names for the CPU baseline; bench.py's cpu_baseline.mpi_openmp leg, timing only)
if [ -x "$MPICC" ] && $CLANG -fopenmp -x c -o /dev/null -c - <<<'int main(void){return 0;}' 2>/dev/null; then
  MPICH_CC=$CLANG "$MPICC" -DCRS -D_MPI -D_OPENMP $DEFS -O3 -ffast-math -fopenmp -std=c23 -w \
    -I"$S" -o "$OUT/sb_ref_mpi_omp" $S/main.c $COMMON "$S/matrix-CRS.c" \
    "$S/matrixBinfile.c" "$S/parameter.c" "$S/affinity.c" -lm -Wl,-rpath,/opt/rocm/lib/llvm/lib \
    || echo "build_ref: MPI+OpenMP variant failed (non-fatal)" >&2
fi

# The reference's OWN driver, unchanged, on top of the MI355X drop-in: src/main.c compiled against the
# forwarding headers (include/sparsebench/compat) and linked with libsparsebench_<fmt>.so -- no reference
# object besides main.o.  Lets the GPU box RUN "main.c drives it" (tests/test_gpu_dropin.py).
# (main.c is compiled from a scratch copy in $TMP: `#include "comm.h"` looks in the including file's own directory
#  first, so compiled in place it would pick the reference's headers, not the forwarding ones)
REPO=$(cd "$HERE/.." && pwd)
if [ -f "$REPO/sparsebench_amd/lib/libsparsebench_crs.so" ]; then
  cp "$S/main.c" "$TMP/main.c"
  for F in CRS SCS; do
    f=$(echo $F | tr A-Z a-z)
    gcc -std=gnu11 -O1 -w -D$F -DPRECISION=2 -DUINT_TYPE=1 -DARRAY_ALIGNMENT=64 \
      -I"$REPO/include/sparsebench/compat" -I"$REPO/include" "$TMP/main.c" -o "$OUT/refmain_${F}_hip" \
      -L"$REPO/sparsebench_amd/lib" -lsparsebench_$f -lsparsebench_host -lsbhip \
      -Wl,-rpath,'$ORIGIN/../../sparsebench_amd/lib' -lm || echo "build_ref: refmain_$F failed (non-fatal)" >&2
  done
fi
ls -la "$OUT"
