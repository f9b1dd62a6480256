#!/bin/bash
# dev helper: build a second librigl_hip with extra -D flags for ONE source file (A/B runs via RIGL_HIP_LIB)
# usage: tools/build_alt.sh NAME FILE -DFOO [-DBAR ...]   ->  build/alt/librigl_NAME.so   (FILE: conv, prune_regrow, bn, ...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
src="$1"; shift
mkdir -p "$ROOT/build/alt"
make -C "$ROOT/rigl_amd/csrc" >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function "$@" \
  -c "$ROOT/rigl_amd/csrc/$src.hip" -o "$ROOT/build/alt/${src}_$name.o"
objs=""
for o in runtime prune_regrow optimizer conv conv_ref conv_f32 bn depthwise pool random head; do
  [ "$o" = "$src" ] || objs="$objs $ROOT/build/$o.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/alt/librigl_$name.so" "$ROOT/build/alt/${src}_$name.o" $objs
echo "$ROOT/build/alt/librigl_$name.so"
