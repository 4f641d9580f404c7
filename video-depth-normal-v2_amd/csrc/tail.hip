// Fused depth tail of the DPT head (depth_anything_v2/dpt.py:146-151, video_depth_anything/dpt_temporal.py:106-111):
//   out = F.interpolate(out1, (14 ph, 14 pw), bilinear, align_corners=True)      [B, C, OH, OW]   (C = features / 2)
//   out = ReLU(Conv3x3(C -> 32, pad 1)(out))
//   depth = [ReLU](Conv1x1(32 -> 1)(out))                                          [B, OH, OW] f32
// in ONE kernel: the x1.75 up-sampled map (1.1 GB of split planes at batch 8, ViT-L) and the 32-channel map never
// reach HBM. Round 1 ran this as upsample_kernel -> implicit-GEMM conv (N = 32: 9.9 GB moved L2 -> LDS per launch,
// feed-bound) -> head_out_kernel = 1.6 ms per batch-8 step.
//
// One workgroup (4 waves) owns a 16 x 16 output tile. For each block of 32 input channels:
//   fill:  the 18 x 18 halo patch of the UP-SAMPLED map is computed from out1 (4 corner loads of 16 B per plane,
//          fp32 lerp, split into hi/lo halves) straight into LDS: 64-byte rows per pixel and plane, 16-byte chunk
//          XOR-swizzled by the pixel index so that the 16 consecutive pixels of a fragment read hit 16 bank groups;
//          pixels outside the image are the convolution's zero padding;
//   mma:   per tap a wave multiplies its 4 rows x 16 pixels by the tap's 32 x 32 weight block on
//          v_mfma_f32_16x16x32 (weights as the A operand: a lane then holds 4 output channels of ONE pixel), three
//          products per term (hi*lo + lo*hi + hi*hi). Weight fragments come straight from L2 (147 KB in all), the
//          next tap's are fetched while this tap multiplies.
// Epilogue: bias + ReLU + the 32 -> 1 dot product in registers, two cross-lane adds, one 64-byte store per row.
// Roofline: MFMA (19.8 GF per 518 x 518 frame at C = 128, x3 executed); HBM traffic = out1 once (+ halo) + depth.
#include "common.hpp"

namespace {

constexpr int TW = 16, TH = 16, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;  // 18 x 18 = 324 halo pixels
constexpr int CB = 32;                                                      // input channels per pass
constexpr int PLANE = NPIX * 64;                                            // bytes per plane of the patch
#ifndef VDN_TAIL_FB
#define VDN_TAIL_FB 1  // fill items whose corner loads are issued together (tools/build_variant.sh A/B: 2 needs 226 VGPRs)
#endif

__device__ __forceinline__ void ac_coord(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float src = scale * (float)o;  // PyTorch's align_corners=True source index (same as upsample_kernel)
  i0 = (int)src;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}

template <int DT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VDN_TAIL_FB > 1 ? 2 : 3, VDN_TAIL_FB > 1 ? 2 : 3))) void depth_tail_kernel(const typename Half<DT>::T* __restrict__ x,
                                                         const typename Half<DT>::T* __restrict__ xl, int B, int IH, int IW,
                                                         int C, const typename Half<DT>::T* __restrict__ w,
                                                         const typename Half<DT>::T* __restrict__ wl, int ldb,
                                                         const float* __restrict__ b2, const float* __restrict__ w1, float b1,
                                                         float* __restrict__ depth, int OH, int OW, int relu) {
  using H = Half<DT>;
  using T = typename H::T;
  using V8 = typename H::V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [hi | lo] x NPIX x 64 B

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // an XCD walks consecutive tiles of one image: halo rows hit its L2
  const int b = tile / (tiles_x * tiles_y);
  const int trem = tile - b * tiles_x * tiles_y;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
  const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  const T* xb = x + (size_t)b * IH * IW * C;
  const T* xlb = xl + (size_t)b * IH * IW * C;

  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight fragment (A operand) of output-channel block j, tap `tap`, channel block cb: W[16 j + fr][tap C + cb + 8 fq ..]
  const T* wrow[2] = {w + (size_t)fr * ldb + fq * 8, w + (size_t)(16 + fr) * ldb + fq * 8};
  const ptrdiff_t wdelta = (const char*)wl - (const char*)w;
  auto load_w = [&](int k0, V8 (&h)[2], V8 (&l)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      h[j] = *(const V8*)(wrow[j] + k0);
      l[j] = *(const V8*)((const char*)(wrow[j] + k0) + wdelta);
    }
  };

  for (int cb = 0; cb < C; cb += CB) {
    __syncthreads();  // the previous pass's fragment reads are done
    // ---- fill: up-sampled halo patch of channels cb .. cb+31, 4 chunks of 8 channels per pixel; FB items per thread
    // at a time (their 8 FB corner loads are issued together, then interpolated)
    constexpr int FB = VDN_TAIL_FB, NIT = (NPIX * 4 + 255) / 256;
    for (int base = 0; base < NIT; base += FB) {
      V8 cv[FB][8];
      float lyv[FB], lxv[FB];
      bool ok[FB];
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int it = tid + (base + u) * 256;
        const int pix = it >> 2, ch = it & 3;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
        ok[u] = it < NPIX * 4 && gy >= 0 && gy < OH && gx >= 0 && gx < OW;
        if (ok[u]) {
          int y0, y1, x0, x1;
          ac_coord(gy, sy, IH, y0, y1, lyv[u]);
          ac_coord(gx, sx, IW, x0, x1, lxv[u]);
          const size_t co = (size_t)cb + ch * 8;
          const size_t o00 = ((size_t)y0 * IW + x0) * C + co, o01 = ((size_t)y0 * IW + x1) * C + co;
          const size_t o10 = ((size_t)y1 * IW + x0) * C + co, o11 = ((size_t)y1 * IW + x1) * C + co;
          cv[u][0] = *(const V8*)(xb + o00); cv[u][1] = *(const V8*)(xb + o01);
          cv[u][2] = *(const V8*)(xb + o10); cv[u][3] = *(const V8*)(xb + o11);
          cv[u][4] = *(const V8*)(xlb + o00); cv[u][5] = *(const V8*)(xlb + o01);
          cv[u][6] = *(const V8*)(xlb + o10); cv[u][7] = *(const V8*)(xlb + o11);
        }
      }
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int it = tid + (base + u) * 256;
        if (it >= NPIX * 4) continue;
        const int pix = it >> 2, ch = it & 3;
        V8 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) { oh[e] = (T)0.f; ol[e] = (T)0.f; }
        if (ok[u]) {
          const float ly = lyv[u], lx = lxv[u];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            float r[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const float a00 = (float)cv[u][0][e + q] + (float)cv[u][4][e + q], a01 = (float)cv[u][1][e + q] + (float)cv[u][5][e + q];
              const float a10 = (float)cv[u][2][e + q] + (float)cv[u][6][e + q], a11 = (float)cv[u][3][e + q] + (float)cv[u][7][e + q];
              const float top = (1.f - lx) * a00 + lx * a01;
              const float bot = (1.f - lx) * a10 + lx * a11;
              r[q] = (1.f - ly) * top + ly * bot;
            }
            T h0, h1, q0, q1;
            split2_rtz(r[0], r[1], h0, h1, q0, q1);
            oh[e] = h0; oh[e + 1] = h1; ol[e] = q0; ol[e + 1] = q1;
          }
        }
        const int off = pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4);
        *(V8*)(smem + off) = oh;
        *(V8*)(smem + PLANE + off) = ol;
      }
    }
    __syncthreads();
    // ---- mma: 9 taps x (32 channels = one K step)
    V8 wh[2][2], wlo[2][2];
    load_w(cb, wh[0], wlo[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) load_w((tap + 1) * C + cb, wh[cur ^ 1], wlo[cur ^ 1]);
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = (wave * 4 + i + ky) * PW + fr + kx;
        const int off = pix * 64 + ((fq ^ ((pix >> 2) & 3)) << 4);
        const V8 ah = *(const V8*)(smem + off), al = *(const V8*)(smem + PLANE + off);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 c = acc[i][j];
          c = H::mfma16(wh[cur][j], al, c);
          c = H::mfma16(wlo[cur][j], ah, c);
          c = H::mfma16(wh[cur][j], ah, c);
          acc[i][j] = c;
        }
      }
    }
  }

  // ---- epilogue: lane (fr, fq) holds output channels 16 j + 4 fq + e of pixel (row wave*4 + i, column fr)
  f32x4 bias4[2], w14[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bias4[j] = *(const f32x4*)(b2 + j * 16 + fq * 4);
    w14[j] = *(const f32x4*)(w1 + j * 16 + fq * 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] + bias4[j][e];
        v = (v < 0.f) ? 0.f : v;  // NaN passes through
        s = fmaf(v, w14[j][e], s);
      }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    s += b1;
    if (relu) s = (s < 0.f) ? 0.f : s;
    const int oy = ty0 + wave * 4 + i, ox = tx0 + fr;
    if (fq == 0 && oy < OH && ox < OW) depth[((size_t)b * OH + oy) * OW + ox] = s;
  }
}

}  // namespace

extern "C" int vdn_depth_tail(int dt, const void* x, const void* x_lo, int B, int IH, int IW, int C, const void* w,
                              const void* w_lo, int ldb, const float* bias2, const float* w1, float b1, float* depth, int OH,
                              int OW, int relu, vdn_stream stream) {
  if (!x || !x_lo || !w || !w_lo || !bias2 || !w1 || !depth || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return VDN_EINVAL;
  if (C <= 0 || (C % CB) || ldb < 9 * C || (ldb & 7)) return VDN_EALIGN;
  if (((uintptr_t)x | (uintptr_t)x_lo | (uintptr_t)w | (uintptr_t)w_lo | (uintptr_t)bias2 | (uintptr_t)w1) & 15) return VDN_EALIGN;
  const int tiles = B * ((OH + TH - 1) / TH) * ((OW + TW - 1) / TW);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(depth_tail_kernel<VDN_F16>, dim3(tiles), dim3(256), 2 * PLANE, s, (const _Float16*)x, (const _Float16*)x_lo, B,
                       IH, IW, C, (const _Float16*)w, (const _Float16*)w_lo, ldb, bias2, w1, b1, depth, OH, OW, relu);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(depth_tail_kernel<VDN_BF16>, dim3(tiles), dim3(256), 2 * PLANE, s, (const __bf16*)x, (const __bf16*)x_lo, B, IH,
                       IW, C, (const __bf16*)w, (const __bf16*)w_lo, ldb, bias2, w1, b1, depth, OH, OW, relu);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
