#!/bin/bash
# Build the REFERENCE's own cost function into oracle/_ref/libvg_ref.so -- only with REAL Eigen3 and Ceres headers and
# libraries (found through pkg-config or the usual system paths); no stand-in headers, no copies of reference sources.
#   exit 0   built          (then: python tools/gen_ref_fixtures.py  ->  tests/golden/ref_eval_block.json)
#   exit 77  Eigen3 or Ceres (or the reference tree) is absent: nothing is produced, parity stays "unpinned"
# The reference's own build asks for them at CMakeLists.txt:8 (Ceres) and :13 (EIGEN3_INCLUDE_DIR); only
# src/calibration/calib_cost_functions.cpp and the headers it includes are compiled, from where they lie.
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${VG_REFERENCE_ROOT:-/root/reference}
OUT="$HERE/_ref"
[ -f "$REF/src/calibration/calib_cost_functions.cpp" ] || { echo "build_ref: no reference tree at $REF"; exit 77; }

EIGEN_INC=""
if pkg-config --exists eigen3 2>/dev/null; then EIGEN_INC=$(pkg-config --cflags eigen3)
else
  for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/eigen3/include/eigen3; do
    [ -f "$d/Eigen/Eigen" ] && EIGEN_INC="-I$d" && break
  done
fi
[ -n "$EIGEN_INC" ] || { echo "build_ref: Eigen3 headers not found (pkg-config eigen3, /usr/include/eigen3)"; exit 77; }

CERES_INC=""; CERES_LIB=""
if pkg-config --exists ceres 2>/dev/null; then CERES_INC=$(pkg-config --cflags ceres); CERES_LIB=$(pkg-config --libs ceres)
else
  for d in /usr/include /usr/local/include /opt/ceres/include; do
    [ -f "$d/ceres/ceres.h" ] && CERES_INC="-I$d" && break
  done
  [ -n "$CERES_INC" ] && CERES_LIB="-lceres -lglog"
fi
[ -n "$CERES_INC" ] || { echo "build_ref: Ceres headers not found (pkg-config ceres, /usr/include/ceres/ceres.h)"; exit 77; }

mkdir -p "$OUT"
set -x
g++ -std=c++14 -O2 -fPIC -shared -Wno-deprecated -I"$REF/include" $EIGEN_INC $CERES_INC \
    "$HERE/ref_harness.cpp" "$REF/src/calibration/calib_cost_functions.cpp" \
    -o "$OUT/libvg_ref.so" $CERES_LIB || exit 1
echo "build_ref: $OUT/libvg_ref.so"
