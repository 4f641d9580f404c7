// Normalisation / elementwise kernels (HBM-bound; one pass over the data, 8-16 B per lane).
#include "common.hpp"

namespace {

template <typename XT> struct Vec4;
template <> struct Vec4<float> { using V = f32x4; };
template <> struct Vec4<_Float16> { using V = f16x4; };
template <> struct Vec4<__bf16> { using V = bf16x4; };

// ---------------------------------------------------------------- LayerNorm: one wave per row, persistent
// A wave walks rows wave_id, wave_id + nwaves, ... and issues the loads of its NEXT row before it reduces the
// current one, so two rows (8 KiB at C = 1024) per wave are in flight and the grid (<= 8 blocks per CU) has no
// partial last round: round 1 ran one row per wave with 2740 blocks on 2048 slots (1.34 rounds of pure latency)
// at 2.7 TB/s. gamma / beta (+ alpha * addvec) are loaded once per wave. NV = 256-channel steps (4 per lane each).
// X6: the launch also writes the planes of 6-bit rows (out8) — a separate instantiation, so that the plain one keeps its register
// count and occupancy (the staging + conversion code cost a block per CU when it was compiled into every launch)
template <typename XT, int DT, int NV, bool X6>
__global__ __launch_bounds__(256) void layernorm_kernel(const XT* __restrict__ x, int rows, int C,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, const float* __restrict__ addvec, float alpha,
                                                        const float* __restrict__ addtab, int tab_div, int tab_mod,
                                                        int out_group, typename Half<DT>::T* __restrict__ out_h,
                                                        typename Half<DT>::T* __restrict__ out_l,
                                                        float* __restrict__ out_f, uint8_t* __restrict__ out8, int kt) {
  using T = typename Half<DT>::T;
  using XV = typename Vec4<XT>::V;
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  extern __shared__ __attribute__((aligned(16))) _Float16 stage_all[];   // X6: per wave, hi and remainder of a row (4 x 2 x NV x 256 halves)
  _Float16* const stage = stage_all + (X6 ? (threadIdx.x >> 6) * 2 * NV * 256 : 0);
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  f32x4 wv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < C) {
      wv[i] = *(const f32x4*)(w + c);
      bv[i] = *(const f32x4*)(b + c);
      if (addvec) bv[i] += alpha * *(const f32x4*)(addvec + c);
    }
  }
  auto load = [&](int r, XV (&dst)[NV]) {
    const XT* xr = x + (size_t)r * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;
      if (c < C) dst[i] = *(const XV*)(xr + c);
    }
  };
  XV cur[NV], nxt[NV];
  if (row < rows) load(row, cur);
  for (; row < rows; row += nwaves) {
    const int nrow = row + nwaves;
    if (nrow < rows) load(nrow, nxt);
    // out_group > 0: drop the first row of every group (the cls token) and compact the rest
    if (!(out_group > 0 && row % out_group == 0)) {
      const size_t orow = out_group > 0 ? (size_t)(row - row / out_group - 1) : (size_t)row;
      f32x4 v[NV];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i * 256 + lane * 4 < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = (float)cur[i][e];
          sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
      }
      const float mean = wave_sum(sum) / (float)C;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i * 256 + lane * 4 < C) {
          v[i] -= mean;
          sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
      }
      const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
      const float* tab = addtab ? addtab + (size_t)((row / tab_div) % tab_mod) * C : nullptr;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < C) {
          f32x4 y = v[i] * rstd * wv[i] + bv[i];
          if (tab) y += *(const f32x4*)(tab + c);
          if (out_f) *(f32x4*)(out_f + orow * C + c) = y;
          if (out_h) {
            typename Half<DT>::V4 hv, lv;
            // kt: K-tile-major planes for the 8-bit cross-term GEMM (include/vdn.h a_kt): [C/32][rows][32] halves, [C/64][rows][64] bytes
            const size_t oh = kt ? ((size_t)(c >> 5) * rows + orow) * 32 + (c & 31) : orow * C + c;
            if (out_l || out8) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                T h0, h1, l0, l1;
                split2_rtz(y[e], y[e + 1], h0, h1, l0, l1);
                hv[e] = h0; hv[e + 1] = h1; lv[e] = l0; lv[e + 1] = l1;
              }
              if (out_l) *(typename Half<DT>::V4*)(out_l + oh) = lv;
              if constexpr (X6 && DT == VDN_F16) {   // staged for the 6-bit rows below: this wave's row as fp16 hi and remainder
                *(f16x4*)(stage + c) = hv;
                *(f16x4*)(stage + NV * 256 + c) = lv;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = (T)y[e];
            }
            *(typename Half<DT>::V4*)(out_h + oh) = hv;
          }
        }
      }
      if constexpr (X6 && DT == VDN_F16) {
        {
          // 6-bit rows of the consuming GEMM's A operand (common.hpp x6 rows, natural order): the row goes through LDS so that a
          // lane holds the 32 consecutive values of one half; lanes take (plane, half) pairs: hi rows first, then remainder rows
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own staging writes (LDS is in order per wave)
          const int nb = C >> 5;
          for (int t = lane; t < 2 * nb; t += 64) {
            const int plane = t >= nb, hb = t - plane * nb;
            f16x32 hv, xv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f16x8 a = *(const f16x8*)(stage + hb * 32 + 8 * q);
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[8 * q + e] = a[e];
            }
            const int sb = x6_scale_byte(hv);
            if (plane) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f16x8 a = *(const f16x8*)(stage + NV * 256 + hb * 32 + 8 * q);
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[8 * q + e] = a[e];
              }
            } else xv = hv;
            uint8_t* d8 = out8 + (size_t)plane * rows * C +
                          (kt ? ((size_t)(hb >> 1) * rows + orow) * 64 + 32 * (hb & 1) : orow * C + (size_t)hb * 32);
            x6_store_half(d8, xv, plane ? sb - 10 : sb);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the next row overwrites it
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
  }
}

// ---------------------------------------------------------------- GroupNorm (NHWC), 2 kernels
// stats: grid (F, nsplit); deterministic: per-channel column sums in fixed order, then per group.
template <int DT>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const typename Half<DT>::T* __restrict__ x,
                                                              const typename Half<DT>::T* __restrict__ xl, int HW,
                                                              int C, int groups, float* __restrict__ partial,
                                                              int nsplit) {
  using V8 = typename Half<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = blockIdx.x, sp = blockIdx.y;
  const int cv = C >> 3;              // 8-channel vectors per pixel
  const int pl = 256 / cv;            // pixel lanes (>= 1 because C <= 2048)
  const int tid = threadIdx.x;
  const int my_pl = tid / cv, my_cv = tid - my_pl * cv;
  const int p0 = (int)(((long long)HW * sp) / nsplit), p1 = (int)(((long long)HW * (sp + 1)) / nsplit);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  if (my_pl < pl) {
    const size_t xo = (size_t)f * HW * C + my_cv * 8;
    for (int p = p0 + my_pl; p < p1; p += pl) {
      const V8 v = *(const V8*)(x + xo + (size_t)p * C);
      V8 vl;
      if (xl) vl = *(const V8*)(xl + xo + (size_t)p * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float fv = xl ? (float)v[e] + (float)vl[e] : (float)v[e];
        s[e] += fv;
        ss[e] += fv * fv;
      }
    }
  }
  float* col = (float*)smem;  // [pl][C][2]
  if (my_pl < pl) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      col[((size_t)my_pl * C + my_cv * 8 + e) * 2] = s[e];
      col[((size_t)my_pl * C + my_cv * 8 + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  const int cg = C / groups;
  for (int g = tid; g < groups; g += 256) {
    float a = 0.f, bq = 0.f;
    for (int k = 0; k < pl; ++k)
      for (int c = g * cg; c < (g + 1) * cg; ++c) {
        a += col[((size_t)k * C + c) * 2];
        bq += col[((size_t)k * C + c) * 2 + 1];
      }
    float* o = partial + (((size_t)f * nsplit + sp) * groups + g) * 2;
    o[0] = a;
    o[1] = bq;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const typename Half<DT>::T* __restrict__ x,
                                                              const typename Half<DT>::T* __restrict__ xl,
                                                              typename Half<DT>::T* __restrict__ y,
                                                              typename Half<DT>::T* __restrict__ yl, int HW, int C,
                                                              int groups, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps,
                                                              const float* __restrict__ partial, int nsplit) {
  using T = typename Half<DT>::T;
  using V8 = typename Half<DT>::V8;
  __shared__ float mr[2 * 64];
  const int f = blockIdx.x;
  const int cg = C / groups;
  for (int g = threadIdx.x; g < groups; g += 256) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < nsplit; ++k) {
      const float* pp = partial + (((size_t)f * nsplit + k) * groups + g) * 2;
      a += pp[0];
      q += pp[1];
    }
    const float n = (float)HW * (float)cg;
    const float mean = a / n;
    const float var = fmaxf(q / n - mean * mean, 0.f);
    mr[2 * g] = mean;
    mr[2 * g + 1] = rsqrtf(var + eps);
  }
  __syncthreads();
  const size_t nvec = (size_t)HW * (C >> 3);
  const size_t per = (nvec + gridDim.y - 1) / gridDim.y;
  const size_t v0 = per * blockIdx.y, v1 = (v0 + per < nvec) ? v0 + per : nvec;
  const size_t fo = (size_t)f * HW * C;
  const int cv = C >> 3;
  for (size_t i = v0 + threadIdx.x; i < v1; i += 256) {
    const int c0 = (int)(i % cv) * 8;
    const V8 v = *(const V8*)(x + fo + i * 8);
    V8 vl, o, ol;
    if (xl) vl = *(const V8*)(xl + fo + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e, g = c / cg;
      const float xv = xl ? (float)v[e] + (float)vl[e] : (float)v[e];
      const float r = (xv - mr[2 * g]) * mr[2 * g + 1] * w[c] + b[c];
      if (yl) { T a, b2; split_rtz(r, a, b2); o[e] = a; ol[e] = b2; }
      else o[e] = (T)r;
    }
    *(V8*)(y + fo + i * 8) = o;
    if (yl) *(V8*)(yl + fo + i * 8) = ol;
  }
}

__global__ void add_vec_kernel(const float* __restrict__ x, const float* __restrict__ vec, float alpha,
                               float* __restrict__ y, size_t n4, int C) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    const f32x4 a = *(const f32x4*)(x + i * 4);
    const f32x4 v = *(const f32x4*)(vec + c);
    *(f32x4*)(y + i * 4) = f32x4{a[0] + alpha * v[0], a[1] + alpha * v[1], a[2] + alpha * v[2], a[3] + alpha * v[3]};
  }
}

template <int DT>
__global__ void addtab_cast_kernel(const float* __restrict__ x, const float* __restrict__ tab, int tab_div, int tab_mod,
                                   typename Half<DT>::T* __restrict__ y, typename Half<DT>::T* __restrict__ yl, size_t rows,
                                   int C) {
  using T = typename Half<DT>::T;
  const int cv = C >> 2;
  const size_t n4 = rows * cv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / cv;
    const int c = (int)(i - row * cv) * 4;
    f32x4 a = *(const f32x4*)(x + row * C + c);
    if (tab) a += *(const f32x4*)(tab + (size_t)((row / tab_div) % tab_mod) * C + c);
    typename Half<DT>::V4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (yl) { T x0, x1; split_rtz(a[e], x0, x1); h[e] = x0; l[e] = x1; }
      else h[e] = (T)a[e];
    }
    *(typename Half<DT>::V4*)(y + row * C + c) = h;
    if (yl) *(typename Half<DT>::V4*)(yl + row * C + c) = l;
  }
}

__global__ void cast_kernel(const void* __restrict__ x, int xdt, void* __restrict__ y, int ydt, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_from_float(y, ydt, i, load_as_float(x, xdt, i));
}

template <typename XT, int NV>
int ln_launch_nv(const void* x, int rows, int C, const float* w, const float* b, float eps, const float* addvec,
                 float alpha, const float* addtab, int tab_div, int tab_mod, int out_group, void* out_h, void* out_l,
                 int h_dt, float* out_f, void* out8, int kt, hipStream_t s) {
  // grid = what is resident at once (CUs x blocks the register budget admits), persistent beyond that
  static const int resident = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, layernorm_kernel<XT, VDN_F16, NV, false>, 256, 0) != hipSuccess || per_cu <= 0)
      per_cu = 4;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256 * per_cu;
    return prop.multiProcessorCount * per_cu;
  }();
  constexpr size_t lds6 = (size_t)4 * 2 * NV * 256 * sizeof(_Float16);
  static const int resident6 = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, layernorm_kernel<XT, VDN_F16, NV, true>, 256, lds6) != hipSuccess || per_cu <= 0)
      per_cu = 3;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256 * per_cu;
    return prop.multiProcessorCount * per_cu;
  }();
  const int blocks = (rows + 3) / 4;
  const int res = out8 ? resident6 : resident;
  const dim3 grid(blocks < res ? blocks : res);
  if (h_dt == VDN_BF16)
    hipLaunchKernelGGL((layernorm_kernel<XT, VDN_BF16, NV, false>), grid, dim3(256), 0, s, (const XT*)x, rows, C, w, b, eps,
                       addvec, alpha, addtab, tab_div, tab_mod, out_group, (__bf16*)out_h, (__bf16*)out_l, out_f, (uint8_t*)nullptr, 0);
  else if (out8)
    hipLaunchKernelGGL((layernorm_kernel<XT, VDN_F16, NV, true>), grid, dim3(256), lds6, s, (const XT*)x, rows, C, w, b, eps,
                       addvec, alpha, addtab, tab_div, tab_mod, out_group, (_Float16*)out_h, (_Float16*)out_l, out_f, (uint8_t*)out8, kt);
  else
    hipLaunchKernelGGL((layernorm_kernel<XT, VDN_F16, NV, false>), grid, dim3(256), 0, s, (const XT*)x, rows, C, w, b, eps,
                       addvec, alpha, addtab, tab_div, tab_mod, out_group, (_Float16*)out_h, (_Float16*)out_l, out_f, (uint8_t*)nullptr, kt);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <typename XT>
int ln_launch(const void* x, int rows, int C, const float* w, const float* b, float eps, const float* addvec,
              float alpha, const float* addtab, int tab_div, int tab_mod, int out_group, void* out_h, void* out_l,
              int h_dt, float* out_f, void* out8, int kt, hipStream_t s) {
  if (C <= 512)
    return ln_launch_nv<XT, 2>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_l, h_dt, out_f, out8, kt, s);
  if (C <= 1024)
    return ln_launch_nv<XT, 4>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_l, h_dt, out_f, out8, kt, s);
  return ln_launch_nv<XT, 8>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_l, h_dt, out_f, out8, kt, s);
}

}  // namespace

extern "C" int vdn_layernorm(const void* x, int x_dt, int rows, int C, const float* w, const float* b, float eps,
                             const float* addvec, float alpha, const float* addtab, int tab_div, int tab_mod,
                             int out_group, void* out_h, void* out_h_lo, int h_dt, float* out_f, void* out8, int kt,
                             vdn_stream stream) {
  if (!x || !w || !b || rows <= 0 || C <= 0 || (!out_h && !out_f)) return VDN_EINVAL;
  if ((C & 3) || C > 2048) return VDN_EALIGN;
  if (addtab && (tab_div <= 0 || tab_mod <= 0)) return VDN_EINVAL;
  if (out_h && h_dt != VDN_F16 && h_dt != VDN_BF16) return VDN_EUNSUPPORTED;
  if ((out8 || kt) && (!out_h || h_dt != VDN_F16 || out_group > 0 || (C & 63) || ((uintptr_t)out8 & 15))) return VDN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!addtab) { tab_div = 1; tab_mod = 1; }
  switch (x_dt) {
    case VDN_F32: return ln_launch<float>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_h_lo, h_dt, out_f, out8, kt, s);
    case VDN_F16: return ln_launch<_Float16>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_h_lo, h_dt, out_f, out8, kt, s);
    case VDN_BF16: return ln_launch<__bf16>(x, rows, C, w, b, eps, addvec, alpha, addtab, tab_div, tab_mod, out_group, out_h, out_h_lo, h_dt, out_f, out8, kt, s);
    default: return VDN_EUNSUPPORTED;
  }
}

extern "C" int vdn_groupnorm(int dt, const void* x, const void* x_lo, void* y, void* y_lo, int F, int HW, int C,
                             int groups, const float* w, const float* b, float eps, float* partial, int nsplit,
                             vdn_stream stream) {
  if (!x || !y || !w || !b || !partial || F <= 0 || HW <= 0 || groups <= 0 || groups > 64 || C % groups || nsplit <= 0)
    return VDN_EINVAL;
  if ((C & 7) || C > 2048) return VDN_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int pl = 256 / (C >> 3);
  const size_t lds = (size_t)pl * C * 2 * sizeof(float);
  const int chunks = (int)(((size_t)HW * (C >> 3) + 256 * 8 - 1) / (256 * 8));
  if (dt == VDN_F16) {
    hipLaunchKernelGGL(groupnorm_stats_kernel<VDN_F16>, dim3(F, nsplit), dim3(256), lds, s, (const _Float16*)x,
                       (const _Float16*)x_lo, HW, C, groups, partial, nsplit);
    hipLaunchKernelGGL(groupnorm_apply_kernel<VDN_F16>, dim3(F, chunks), dim3(256), 0, s, (const _Float16*)x,
                       (const _Float16*)x_lo, (_Float16*)y, (_Float16*)y_lo, HW, C, groups, w, b, eps, partial, nsplit);
  } else if (dt == VDN_BF16) {
    hipLaunchKernelGGL(groupnorm_stats_kernel<VDN_BF16>, dim3(F, nsplit), dim3(256), lds, s, (const __bf16*)x,
                       (const __bf16*)x_lo, HW, C, groups, partial, nsplit);
    hipLaunchKernelGGL(groupnorm_apply_kernel<VDN_BF16>, dim3(F, chunks), dim3(256), 0, s, (const __bf16*)x,
                       (const __bf16*)x_lo, (__bf16*)y, (__bf16*)y_lo, HW, C, groups, w, b, eps, partial, nsplit);
  } else {
    return VDN_EUNSUPPORTED;
  }
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_add_vec(const float* x, const float* vec, float alpha, float* y, int rows, int C,
                           vdn_stream stream) {
  if (!x || !vec || !y || rows <= 0 || C <= 0) return VDN_EINVAL;
  if (C & 3) return VDN_EALIGN;
  const size_t n4 = (size_t)rows * C / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_vec_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, vec, alpha, y, n4, C);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_addtab_cast(int dt, const float* x, const float* tab, int tab_div, int tab_mod, void* y, void* y_lo,
                               size_t rows, int C, vdn_stream stream) {
  if (!x || !y || rows == 0 || C <= 0 || (tab && (tab_div <= 0 || tab_mod <= 0))) return VDN_EINVAL;
  if (C & 3) return VDN_EALIGN;
  const size_t n4 = rows * (C >> 2);
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (!tab) { tab_div = 1; tab_mod = 1; }
  if (dt == VDN_F16)
    hipLaunchKernelGGL(addtab_cast_kernel<VDN_F16>, dim3(blocks), dim3(256), 0, s, x, tab, tab_div, tab_mod, (_Float16*)y,
                       (_Float16*)y_lo, rows, C);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(addtab_cast_kernel<VDN_BF16>, dim3(blocks), dim3(256), 0, s, x, tab, tab_div, tab_mod, (__bf16*)y,
                       (__bf16*)y_lo, rows, C);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_cast(const void* x, int x_dt, void* y, int y_dt, size_t n, vdn_stream stream) {
  if (!x || !y || n == 0) return VDN_EINVAL;
  if (x_dt < 0 || x_dt > VDN_F32 || y_dt < 0 || y_dt > VDN_F32) return VDN_EUNSUPPORTED;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_dt, y, y_dt, n);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
