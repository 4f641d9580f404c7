#!/bin/bash
# Build a variant of the library with extra compiler flags for same-box A/B runs (select with VDN_LIB).
# Usage: [ONLY="gemm_x8 attn"] tools/build_variant.sh NAME -DFLAG [...]   ->  video-depth-normal-v2_amd/lib/abl/libvdn_NAME.so
# ONLY: recompile just these translation units with the flags and take every other object from the regular build (lib/*.o).
set -e
PKG="$(cd "$(dirname "$0")/.." && pwd)/video-depth-normal-v2_amd"
name=$1; shift
out="$PKG/lib/abl/$name"; mkdir -p "$out"
ALL="gemm_big_f16 gemm_big_bf16 gemm_small_f16 gemm_small_bf16 gemm_x8 gemm attn norm spatial tail pack stitch refine"
for f in $ALL; do
  if [ -n "$ONLY" ] && ! [[ " $ONLY " == *" $f "* ]]; then cp "$PKG/lib/$f.o" "$out/$f.o"; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-comment "$@" -c "$PKG/csrc/$f.hip" -o "$out/$f.o" &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/abl/libvdn_$name.so" "$out"/*.o
rm -rf "$out"
ls -la "$PKG/lib/abl/libvdn_$name.so"
