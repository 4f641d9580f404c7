#!/bin/bash
# Build a variant of the whole library with extra compiler flags for same-box A/B runs (select with VDN_LIB).
# Usage: tools/build_variant.sh NAME -DFLAG [...]   ->  video-depth-normal-v2_amd/lib/abl/libvdn_NAME.so
set -e
PKG="$(cd "$(dirname "$0")/.." && pwd)/video-depth-normal-v2_amd"
name=$1; shift
out="$PKG/lib/abl/$name"; mkdir -p "$out"
for f in gemm_big_f16 gemm_big_bf16 gemm_small_f16 gemm_small_bf16 gemm_x8 gemm attn norm spatial tail pack stitch refine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c "$PKG/csrc/$f.hip" -o "$out/$f.o" &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/abl/libvdn_$name.so" "$out"/*.o
rm -rf "$out"
ls -la "$PKG/lib/abl/libvdn_$name.so"
