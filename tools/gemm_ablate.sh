#!/bin/bash
# Build ablated variants of the big-tile GEMM (main-loop anatomy; results are WRONG by construction,
# used only by tools/gemm_bench.py through VDN_LIB). Usage: tools/gemm_ablate.sh 1 2 3 4 5
set -e
PKG="$(cd "$(dirname "$0")/.." && pwd)/video-depth-normal-v2_amd"
mkdir -p "$PKG/lib/abl"
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DVDN_ABLATE=$v \
     -c "$PKG/csrc/gemm_big_f16.hip" -o "$PKG/lib/abl/gemm_big_f16_$v.o" &
done
wait
for v in "$@"; do
  objs=""
  for o in gemm_big_bf16 gemm_small_f16 gemm_small_bf16 gemm attn norm spatial; do objs="$objs $PKG/lib/$o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/abl/libvdn_abl$v.so" "$PKG/lib/abl/gemm_big_f16_$v.o" $objs
done
ls -la "$PKG/lib/abl/"
