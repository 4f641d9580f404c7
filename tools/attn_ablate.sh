#!/bin/bash
# Variant libraries of the attention kernel with parts of its loop removed (VDN_ATTN_ABL bits, csrc/attn.hip) for
# same-box timing: where do the cycles of a 64-key tile go? Usage: [FLAG=VDN_ATTN_OPT VARS="0 1 ..."] tools/attn_ablate.sh build | run
# (VDN_ATTN_ABL bits: 1 no DMA, 2 no barrier, 4 no softmax VALU, 8 no MFMA; VDN_ATTN_OPT: schedule switches, csrc/attn.hip)
PKG="$(cd "$(dirname "$0")/.." && pwd)/video-depth-normal-v2_amd"
VARS="${VARS:-0 1 2 3 4 7 8 11}"; FLAG="${FLAG:-VDN_ATTN_ABL}"
if [ "$1" = build ]; then
  mkdir -p "$PKG/lib/abl"
  for v in $VARS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -D$FLAG=$v $EXTRA -c "$PKG/csrc/attn.hip" -o /tmp/attn_abl_$TAG$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/abl/libvdn_attn$TAG$v.so" /tmp/attn_abl_$TAG$v.o $(ls "$PKG"/lib/*.o | grep -v /attn.o) ) &
    if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
  done
  wait; ls -la "$PKG"/lib/abl/
else
  for v in $VARS; do
    echo "== $FLAG=$v $EXTRA"
    VDN_LIB="$PKG/lib/abl/libvdn_attn$TAG$v.so" timeout -k 10 120 python3 "$(dirname "$0")/attn_bench.py" || exit 1
  done
fi
