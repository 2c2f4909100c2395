#!/bin/bash
# Build libwct_hip.so for gfx950 (cross-compiles without a GPU).  In-tree so that it travels with the repo snapshot.
# One object per source, compiled in parallel and only when the source or a header changed (build/ is git-ignored).
#   build.sh [-f]          -f: rebuild everything
#   WCT_DEFS="-DWCT_SP_TIMING" build.sh -f     instrumented builds (tools/experiments/*.sh)
set -e
cd "$(dirname "$0")"
SRC="conv3x3 conv3x3_f16 conv3x3_sp level1 moments solve misc resize wct_api"
HDR="csrc/wct_common.h csrc/conv_f16_dev.h csrc/wct_sharded_impl.h ../include/wct_hip.h build.sh"
OUT=${WCT_OUT:-libwct_hip.so}
OBJ=build/obj${WCT_DEFS:+_$(echo "$WCT_DEFS" | md5sum | cut -c1-8)}
mkdir -p "$OBJ"
[ "$1" = "-f" ] && rm -f "$OBJ"/*.o
pids=""
for s in $SRC; do
  o="$OBJ/$s.o"
  stale=0
  [ -f "$o" ] || stale=1
  for f in csrc/$s.hip $HDR; do [ "$f" -nt "$o" ] && stale=1; done
  if [ $stale -eq 1 ]; then
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $WCT_DEFS -c csrc/$s.hip -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids="$pids $!"
  fi
done
fail=0
for p in $pids; do wait $p || fail=1; done
[ $fail -eq 0 ] || { echo "compile failed"; exit 1; }
if [ -n "$pids" ] || [ ! -f "$OUT" ]; then
  objs=""; for s in $SRC; do objs="$objs $OBJ/$s.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT" $objs -ldl
  echo "built $(pwd)/$OUT"
else
  echo "libwct_hip.so up to date"
fi
