#!/bin/bash
# Build libwct_hip.so for gfx950 (cross-compiles without a GPU).  In-tree so that it travels with the repo snapshot.
set -e
cd "$(dirname "$0")"
SRC="csrc/conv3x3.hip csrc/conv3x3_f16.hip csrc/conv3x3_sp.hip csrc/level1.hip csrc/moments.hip csrc/solve.hip csrc/misc.hip csrc/resize.hip csrc/wct_api.hip"
OUT=libwct_hip.so
if [ -f "$OUT" ] && [ "$1" != "-f" ]; then
  newer=0
  for f in $SRC csrc/wct_common.h csrc/conv_f16_dev.h ../include/wct_hip.h build.sh; do [ "$f" -nt "$OUT" ] && newer=1; done
  [ $newer -eq 0 ] && { echo "libwct_hip.so up to date"; exit 0; }
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$OUT" $SRC
echo "built $(pwd)/$OUT"
