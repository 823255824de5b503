#!/bin/bash
# tools/refprobe/build.sh [outdir]: compiles the reference's core from /root/reference/src (read-only) with this directory's
# stand-in headers into $outdir/refprobe (default /tmp/refprobe).  Build container only; see README.md.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="${REFERENCE_SRC:-/root/reference/src}"
OUT="${1:-${REFPROBE_OUT:-/tmp/refprobe}}"
[ -d "$SRC" ] || { echo "no reference sources at $SRC" >&2; exit 3; }
mkdir -p "$OUT"
if [ "$OUT/refprobe" -nt "$HERE/driver.cpp" ] && [ "$OUT/refprobe" -nt "$HERE/build.sh" ] && [ -z "$REFPROBE_FORCE" ]; then echo "$OUT/refprobe"; exit 0; fi
g++ -O2 -fopenmp -std=c++11 -w -D__GAPS_OPENMP__ -DBOOST_MATH_PROMOTE_DOUBLE_POLICY=0 -DGAPS_DISABLE_CHECKPOINTS \
    -I"$HERE/shim" -I"$SRC" "$HERE/driver.cpp" \
    "$SRC"/GapsParameters.cpp "$SRC"/GapsResult.cpp "$SRC"/GapsRunner.cpp "$SRC"/GapsStatistics.cpp \
    "$SRC"/atomic/*.cpp "$SRC"/data_structures/*.cpp "$SRC"/file_parser/*.cpp "$SRC"/math/*.cpp \
    "$SRC"/gibbs_sampler/AlphaParameters.cpp "$SRC"/gibbs_sampler/DenseNormalModel.cpp "$SRC"/gibbs_sampler/SparseNormalModel.cpp \
    -o "$OUT/refprobe.tmp.$$"
mv "$OUT/refprobe.tmp.$$" "$OUT/refprobe"
echo "$OUT/refprobe"
