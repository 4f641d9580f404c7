// One-time weight packing on the device (include/vdn.h: vdn_pack_weight): the reference's fp32 parameter tensors, in
// the layouts torch.nn keeps them, -> the K-contiguous 16-bit (hi [, lo]) planes vdn_gemm / vdn_depth_tail read.
// A host in any language can therefore feed libvdn_hip.so from a raw checkpoint without Python. One thread per output
// element computes its source index from the layout kind; K is zero padded to the plane stride `ldb`.
#include "common.hpp"

namespace {

struct PackArgs {
  int kind, d0, d1, d2, rows, K, ldb;
};

__device__ __forceinline__ long src_index(const PackArgs& a, int n, int k) {
  switch (a.kind) {
    case VDN_PACK_LINEAR:  // [N, K] (also conv1x1 [Co, Ci, 1, 1] and the patch embedding [C, 3*14*14])
      return (long)n * a.d1 + k;
    case VDN_PACK_CONV3X3: {  // [Co, Ci, 3, 3] -> K = (ci / 64, tap, ci % 64) when Ci % 64 == 0, else (tap, ci)
      const int Ci = a.d1;
      int tap, ci;
      if (Ci % 64 == 0) {
        const int c64 = k / 576, r = k - c64 * 576;
        tap = r >> 6;
        ci = c64 * 64 + (r & 63);
      } else {
        tap = k / Ci;
        ci = k - tap * Ci;
      }
      return ((long)n * Ci + ci) * 9 + tap;
    }
    case VDN_PACK_CONV3X3_TAPS: {  // [Co, Ci, 3, 3] -> K = (tap, ci) always (vdn_depth_tail)
      const int Ci = a.d1, tap = k / Ci, ci = k - tap * Ci;
      return ((long)n * Ci + ci) * 9 + tap;
    }
    case VDN_PACK_CONVT: {  // ConvTranspose2d [Ci, Co, k, k], kernel == stride -> rows (ky, kx, co), K = ci
      const int Co = a.d1, ks = a.d2;
      const int kk = n / Co, co = n - kk * Co;
      return ((long)k * Co + co) * ks * ks + kk;
    }
    case VDN_PACK_GEGLU: {  // [2 Nh, K] = [h ; gate] -> 16-row blocks alternating h / gate
      const int Nh = a.d0 / 2, t = n >> 5, r = n & 31;
      const int row = r < 16 ? t * 16 + r : Nh + t * 16 + (r - 16);
      return (long)row * a.d1 + k;
    }
    case VDN_PACK_ROPE: {  // per 64-row head: (2i, 2i+1) pairs -> [re 0-15 | im 0-15 | re 16-31 | im 16-31]
      const int head = n >> 6, p = n & 63, blk = p >> 4, r = p & 15;
      const int row = head * 64 + 2 * ((blk >> 1) * 16 + r) + (blk & 1);
      return (long)row * a.d1 + k;
    }
  }
  return -1;
}

template <int DT>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, typename Half<DT>::T* __restrict__ hi,
                                                   typename Half<DT>::T* __restrict__ lo, PackArgs a) {
  using T = typename Half<DT>::T;
  const size_t total = (size_t)a.rows * a.ldb;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / a.ldb), k = (int)(i - (size_t)n * a.ldb);
    const float v = k < a.K ? w[src_index(a, n, k)] : 0.f;
    const T h = (T)v;  // weights: hi rounded to nearest, lo = nearest(v - hi)
    hi[i] = h;
    if (lo) lo[i] = (T)(v - (float)h);
  }
}

// bias vector of a packed projection: the row order of the packed weight (GEGLU / RoPE permutations, ConvTranspose repeat)
__global__ void pack_bias_kernel(const float* __restrict__ b, float* __restrict__ out, PackArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.rows) return;
  PackArgs row_only = a;
  row_only.d1 = 1;  // src_index(..., k = 0) with a one-column source = the source ROW
  long src;
  if (a.kind == VDN_PACK_CONVT) src = n % a.d1;
  else if (a.kind == VDN_PACK_GEGLU || a.kind == VDN_PACK_ROPE) src = src_index(row_only, n, 0);
  else src = n;
  out[n] = b[src];
}

bool pack_geometry(int kind, int d0, int d1, int d2, int* rows, int* K) {
  if (d0 <= 0 || d1 <= 0) return false;
  switch (kind) {
    case VDN_PACK_LINEAR: *rows = d0; *K = d1; return true;
    case VDN_PACK_CONV3X3:
    case VDN_PACK_CONV3X3_TAPS: *rows = d0; *K = 9 * d1; return (d1 & 7) == 0;
    case VDN_PACK_CONVT: *rows = d2 * d2 * d1; *K = d0; return d2 > 0;
    case VDN_PACK_GEGLU: *rows = d0; *K = d1; return (d0 & 31) == 0;
    case VDN_PACK_ROPE: *rows = d0; *K = d1; return (d0 & 63) == 0;
  }
  return false;
}

}  // namespace

extern "C" int vdn_pack_rows(int kind, int d0, int d1, int d2) {
  int rows, K;
  return pack_geometry(kind, d0, d1, d2, &rows, &K) ? rows : VDN_EINVAL;
}

extern "C" int vdn_pack_ldb(int kind, int d0, int d1, int d2) {
  int rows, K;
  return pack_geometry(kind, d0, d1, d2, &rows, &K) ? (K + 63) / 64 * 64 : VDN_EINVAL;
}

extern "C" int vdn_pack_weight(int dt, int kind, const float* w, int d0, int d1, int d2, void* hi, void* lo, int ldb,
                               vdn_stream stream) {
  PackArgs a;
  a.kind = kind; a.d0 = d0; a.d1 = d1; a.d2 = d2;
  if (!w || !hi || !pack_geometry(kind, d0, d1, d2, &a.rows, &a.K)) return VDN_EINVAL;
  if (ldb < a.K || (ldb & 63)) return VDN_EALIGN;
  a.ldb = ldb;
  const size_t total = (size_t)a.rows * ldb;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(pack_kernel<VDN_F16>, dim3(blocks), dim3(256), 0, s, w, (_Float16*)hi, (_Float16*)lo, a);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(pack_kernel<VDN_BF16>, dim3(blocks), dim3(256), 0, s, w, (__bf16*)hi, (__bf16*)lo, a);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_pack_bias(int kind, const float* b, int d0, int d1, int d2, float* out, vdn_stream stream) {
  PackArgs a;
  a.kind = kind; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.ldb = 0;
  if (!b || !out || !pack_geometry(kind, d0, d1, d2, &a.rows, &a.K)) return VDN_EINVAL;
  hipLaunchKernelGGL(pack_bias_kernel, dim3((a.rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, b, out, a);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

namespace {
// Operand planes of the 8-bit cross-term GEMM from fp16 split planes (include/vdn.h: vdn_pack_x8): the hi plane again
// K-tile-major and the two planes of 6-bit rows (hi, remainder; common.hpp 'x6 rows'), K-tile-major or row-major. One thread
// per row, 64-wide K slab and half: 32 values gathered in the stream order of the activation's producer.
template <int ORDER>   // compile-time stream order: the gather below is register renaming, the 64 values come in as 16-byte loads
__global__ __launch_bounds__(256) void pack_x8_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, int rows,
                                                      int ld, _Float16* __restrict__ hi_kt, uint8_t* __restrict__ p8, int kt) {
  const size_t total = (size_t)rows * (ld >> 5);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / (ld >> 5)), hb = (int)(i - (size_t)r * (ld >> 5)), slab = hb >> 1, h = hb & 1;
    const _Float16* hr = hi + (size_t)r * ld + slab * 64;
    const _Float16* lr = lo + (size_t)r * ld + slab * 64;
    // a half's 32 values are runs of 32, 8 or 4 consecutive columns (common.hpp x6_col): vector loads at addresses that depend
    // on h, register positions fixed at compile time
    f16x32 hv, lv;
    if constexpr (ORDER == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {   // position 4 q .. 4 q + 3 = columns 32 (q >> 2) + 8 (q & 3) + 4 h + {0..3}
        const int c = 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
        const f16x4 a = *(const f16x4*)(hr + c), b = *(const f16x4*)(lr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[4 * q + e] = a[e]; lv[4 * q + e] = b[e]; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // position 8 q .. 8 q + 7: order 0 columns 32 h + 8 q, order 1 columns 32 (q >> 1) + 16 (q & 1) + 8 h
        const int c = ORDER == 0 ? 32 * h + 8 * q : 32 * (q >> 1) + 16 * (q & 1) + 8 * h;
        const f16x8 a = *(const f16x8*)(hr + c), b = *(const f16x8*)(lr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { hv[8 * q + e] = a[e]; lv[8 * q + e] = b[e]; }
      }
    }
    if (hi_kt) {  // natural column order: K tile 2 slab + h of this row
      _Float16* d = hi_kt + ((size_t)(2 * slab + h) * rows + r) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(f16x8*)(d + 8 * q) = *(const f16x8*)(hr + 32 * h + 8 * q);
    }
    const int sb = x6_scale_byte(hv);
    uint8_t* d8 = p8 + (kt ? ((size_t)slab * rows + r) * 64 + 32 * h : (size_t)r * ld + slab * 64 + 32 * h);
    x6_store_half(d8, hv, sb);
    x6_store_half(d8 + (size_t)rows * ld, lv, sb - 10);
  }
}

// the same from an fp32 activation [rows, ld] (natural order, K-tile-major): split toward zero, hi plane + the two planes of 6-bit rows
__global__ __launch_bounds__(256) void pack_x8_f32_kernel(const float* __restrict__ x, int rows, int ld, _Float16* __restrict__ hi_kt,
                                                          uint8_t* __restrict__ p8) {
  const size_t total = (size_t)rows * (ld >> 5);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / (ld >> 5)), hb = (int)(i - (size_t)r * (ld >> 5));
    const float* xr = x + (size_t)r * ld + hb * 32;
    f16x32 hv, lv;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 a = *(const f32x4*)(xr + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        _Float16 h0, h1, l0, l1;
        split2_rtz(a[e], a[e + 1], h0, h1, l0, l1);
        hv[4 * q + e] = h0; hv[4 * q + e + 1] = h1; lv[4 * q + e] = l0; lv[4 * q + e + 1] = l1;
      }
    }
    _Float16* d = hi_kt + ((size_t)hb * rows + r) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = hv[8 * q + e];
      *(f16x8*)(d + 8 * q) = o;
    }
    const int sb = x6_scale_byte(hv);
    uint8_t* d8 = p8 + ((size_t)(hb >> 1) * rows + r) * 64 + 32 * (hb & 1);
    x6_store_half(d8, hv, sb);
    x6_store_half(d8 + (size_t)rows * ld, lv, sb - 10);
  }
}

}  // namespace

extern "C" int vdn_pack_x8_f32(const float* x, int rows, int ld, void* hi_kt, void* planes8, vdn_stream stream) {
  if (!x || !hi_kt || !planes8 || rows <= 0 || ld <= 0 || (ld & 63)) return VDN_EINVAL;
  if (((uintptr_t)x | (uintptr_t)hi_kt | (uintptr_t)planes8) & 15) return VDN_EALIGN;
  const size_t work = (size_t)rows * (ld >> 5);
  const dim3 g((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192));
  hipLaunchKernelGGL(pack_x8_f32_kernel, g, dim3(256), 0, (hipStream_t)stream, x, rows, ld, (_Float16*)hi_kt, (uint8_t*)planes8);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_pack_x8(const void* hi, const void* lo, int rows, int ld, void* hi_kt, void* planes8, int kt, int order,
                           vdn_stream stream) {
  if (!hi || !lo || !planes8 || rows <= 0 || ld <= 0 || (ld & 63) || order < 0 || order > 2) return VDN_EINVAL;
  if (((uintptr_t)hi | (uintptr_t)lo | (uintptr_t)hi_kt | (uintptr_t)planes8) & 15) return VDN_EALIGN;
  if (hi_kt && !kt) return VDN_EINVAL;
  const size_t work = (size_t)rows * (ld >> 5);
  const dim3 g((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192));
  const auto kern = order == 0 ? pack_x8_kernel<0> : order == 1 ? pack_x8_kernel<1> : pack_x8_kernel<2>;
  hipLaunchKernelGGL(kern, g, dim3(256), 0, (hipStream_t)stream, (const _Float16*)hi, (const _Float16*)lo, rows, ld,
                     (_Float16*)hi_kt, (uint8_t*)planes8, kt);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

// Scratch a caller should hand to vdn_gemm (splitk_ws) so that this launch may split K: 0 when the shape never does.
extern "C" size_t vdn_gemm_workspace_bytes(const vdn_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || !d->A_lo || !d->W_lo) return 0;
  const int depth = d->a_mode == VDN_A_CONV3X3 ? d->ldb : d->K;
  if (depth < 1024) return 0;
  return (size_t)8 * d->M * d->N * sizeof(float);  // up to 8 K slices of raw f32 partial sums
}

extern "C" size_t vdn_groupnorm_workspace_bytes(int frames, int groups, int nsplit) {
  if (frames <= 0 || groups <= 0 || nsplit <= 0) return 0;
  return (size_t)frames * nsplit * groups * 2 * sizeof(float);
}
