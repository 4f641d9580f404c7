// Spatial / gather kernels (HBM-bound): bilinear resize, patchify, pos-embed bicubic, depth tail,
// MaskDownSampler stages, depthwise 7x7.
#include "common.hpp"
#include <stddef.h>

namespace {

// PyTorch's align_corners=True source index: scale = (in-1)/(out-1) in float, src = scale * dst
__device__ __forceinline__ void ac_coord(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float src = scale * (float)o;
  i0 = (int)src;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}

template <int DT>
__global__ __launch_bounds__(256) void upsample_kernel(const typename Half<DT>::T* __restrict__ x,
                                                       const typename Half<DT>::T* __restrict__ xl,
                                                       typename Half<DT>::T* __restrict__ y,
                                                       typename Half<DT>::T* __restrict__ yl, int B, int IH, int IW,
                                                       int OH, int OW, int C) {
  using T = typename Half<DT>::T;
  using V8 = typename Half<DT>::V8;
  const int cv = C >> 3;
  const size_t total = (size_t)B * OH * OW * cv;
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
  const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    size_t pix = i / cv;
    const int ox = (int)(pix % OW);
    pix /= OW;
    const int oy = (int)(pix % OH);
    const int b = (int)(pix / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    ac_coord(oy, sy, IH, y0, y1, ly);
    ac_coord(ox, sx, IW, x0, x1, lx);
    const size_t xo = (size_t)b * IH * IW * C + c8 * 8;
    const size_t o00 = xo + ((size_t)y0 * IW + x0) * C, o01 = xo + ((size_t)y0 * IW + x1) * C;
    const size_t o10 = xo + ((size_t)y1 * IW + x0) * C, o11 = xo + ((size_t)y1 * IW + x1) * C;
    const V8 v00 = *(const V8*)(x + o00), v01 = *(const V8*)(x + o01);
    const V8 v10 = *(const V8*)(x + o10), v11 = *(const V8*)(x + o11);
    V8 l00, l01, l10, l11, o, ol;
    if (xl) {
      l00 = *(const V8*)(xl + o00); l01 = *(const V8*)(xl + o01);
      l10 = *(const V8*)(xl + o10); l11 = *(const V8*)(xl + o11);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a00 = (float)v00[e], a01 = (float)v01[e], a10 = (float)v10[e], a11 = (float)v11[e];
      if (xl) { a00 += (float)l00[e]; a01 += (float)l01[e]; a10 += (float)l10[e]; a11 += (float)l11[e]; }
      const float top = (1.f - lx) * a00 + lx * a01;
      const float bot = (1.f - lx) * a10 + lx * a11;
      const float r = (1.f - ly) * top + ly * bot;
      if (yl) { T a, b2; split_rtz(r, a, b2); o[e] = a; ol[e] = b2; }
      else o[e] = (T)r;
    }
    *(V8*)(y + i * 8) = o;
    if (yl) *(V8*)(yl + i * 8) = ol;
  }
}

__global__ __launch_bounds__(256) void upsample_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                           int IH, int IW, int OH, int OW, int relu) {
  const size_t total = (size_t)B * OH * OW;
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
  const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const size_t t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    ac_coord(oy, sy, IH, y0, y1, ly);
    ac_coord(ox, sx, IW, x0, x1, lx);
    const float* xb = x + (size_t)b * IH * IW;
    const float top = (1.f - lx) * xb[y0 * IW + x0] + lx * xb[y0 * IW + x1];
    const float bot = (1.f - lx) * xb[y1 * IW + x0] + lx * xb[y1 * IW + x1];
    float v = (1.f - ly) * top + ly * bot;
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       typename Half<DT>::T* __restrict__ rows,
                                                       typename Half<DT>::T* __restrict__ rows_lo, int B, int H, int W,
                                                       int ldk) {
  using T = typename Half<DT>::T;
  using V8 = typename Half<DT>::V8;
  const int ph = H / 14, pw = W / 14;
  const int kv = ldk >> 3;
  const size_t total = (size_t)B * ph * pw * kv;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k8 = (int)(i % kv);
    size_t row = i / kv;
    const int px = (int)(row % pw);
    const size_t t = row / pw;
    const int py = (int)(t % ph);
    const int b = (int)(t / ph);
    V8 o, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k8 * 8 + e;
      float v = 0.f;
      if (k < 588) {
        const int c = k / 196, rem = k - c * 196;
        const int ky = rem / 14, kx = rem - ky * 14;
        v = img[(((size_t)b * 3 + c) * H + py * 14 + ky) * W + px * 14 + kx];
      }
      if (rows_lo) { T a, b2; split_rtz(v, a, b2); o[e] = a; ol[e] = b2; }
      else o[e] = (T)v;
    }
    *(V8*)(rows + i * 8) = o;
    if (rows_lo) *(V8*)(rows_lo + i * 8) = ol;
  }
}

__global__ void fill_row_kernel(float* __restrict__ x, const float* __restrict__ vec, int B, int rows_per_b, int row,
                                int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  x[((size_t)b * rows_per_b + row) * C + c] = vec[c];
}

// torch upsample_bicubic2d (A = -0.75), align_corners=False with an explicit scale factor:
// src = (dst + 0.5) / scale - 0.5, border-clamped taps.
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ void bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int ih, int iw, int oh, int ow,
                               int C, float inv_sy, float inv_sx) {
  const size_t total = (size_t)oh * ow * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int ox = (int)(p % ow), oy = (int)(p / ow);
    const float fy = ((float)oy + 0.5f) * inv_sy - 0.5f;
    const float fx = ((float)ox + 0.5f) * inv_sx - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - (float)iy, wy);
    cubic_w(fx - (float)ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a;
      yy = yy < 0 ? 0 : (yy > ih - 1 ? ih - 1 : yy);
      float rowv = 0.f;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        int xx = ix - 1 + bq;
        xx = xx < 0 ? 0 : (xx > iw - 1 ? iw - 1 : xx);
        rowv += wx[bq] * src[((size_t)yy * iw + xx) * C + c];
      }
      acc += wy[a] * rowv;
    }
    dst[i] = acc;
  }
}

// vdn_preprocess: the whole pre-processing of a batch of frames in one launch — u8 [n,h,w,3] (RGB, or BGR with swap_rb) ->
// /255 -> cubic resize to (H, W) (same taps as bicubic_kernel: A = -0.75, half-pixel centres, border clamp; identity-sized
// inputs hit the weights {0,1,0,0} exactly) -> (v - mean[c]) / std[c] -> f32 NCHW. One thread per output pixel, 3 channels.
struct PrepNorm { float mean[3], inv_std[3]; };
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int n, int ih, int iw,
                                                         int oh, int ow, float inv_sy, float inv_sx, int swap_rb, PrepNorm nm) {
  const size_t per = (size_t)oh * ow, total = per * n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i / per);
    const size_t p = i - (size_t)f * per;
    const int ox = (int)(p % ow), oy = (int)(p / ow);
    const float fy = ((float)oy + 0.5f) * inv_sy - 0.5f;
    const float fx = ((float)ox + 0.5f) * inv_sx - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - (float)iy, wy);
    cubic_w(fx - (float)ix, wx);
    const uint8_t* img = src + (size_t)f * ih * iw * 3;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a;
      yy = yy < 0 ? 0 : (yy > ih - 1 ? ih - 1 : yy);
      float rowv[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        int xx = ix - 1 + bq;
        xx = xx < 0 ? 0 : (xx > iw - 1 ? iw - 1 : xx);
        const uint8_t* px = img + ((size_t)yy * iw + xx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) rowv[c] += wx[bq] * ((float)px[c] / 255.0f);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy[a] * rowv[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = swap_rb ? 2 - c : c;   // output channel c reads source channel sc
      dst[((size_t)f * 3 + c) * per + p] = (acc[sc] - nm.mean[c]) * nm.inv_std[c];
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void head_out_kernel(const typename Half<DT>::T* __restrict__ feat,
                                                       const typename Half<DT>::T* __restrict__ feat_lo,
                                                       const float* __restrict__ w, float bias,
                                                       float* __restrict__ depth, int M, int C, int relu) {
  using V8 = typename Half<DT>::V8;
  __shared__ float sw[64];
  if (threadIdx.x < C) sw[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  for (size_t m = blockIdx.x * (size_t)256 + threadIdx.x; m < (size_t)M; m += (size_t)gridDim.x * 256) {
    float acc = bias;
    for (int c = 0; c < C; c += 8) {
      const V8 v = *(const V8*)(feat + m * C + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += sw[c + e] * (float)v[e];
      if (feat_lo) {
        const V8 vl = *(const V8*)(feat_lo + m * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += sw[c + e] * (float)vl[e];
      }
    }
    depth[m] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// sigmoid -> conv3x3 s2 p1 (1->4) -> LayerNorm2d(4, eps 1e-6) -> GELU -> conv1x1 (4->1)
// w: [conv w 4x9 | conv b 4 | ln w 4 | ln b 4 | proj w 4 | proj b 1]
__global__ __launch_bounds__(256) void mask_down1_kernel(const float* __restrict__ depth, float* __restrict__ out,
                                                         int B, int H, int W, int OH, int OW,
                                                         const float* __restrict__ w) {
  __shared__ float sw[53];
  if (threadIdx.x < 53) sw[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const size_t total = (size_t)B * OH * OW;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const size_t t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    float in[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        float v = 0.f;  // zero padding applies to sigmoid(depth), i.e. the padded value is 0
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = 1.f / (1.f + expf(-depth[((size_t)b * H + iy) * W + ix]));
        in[ky * 3 + kx] = v;
      }
    float ch[4], mean = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = sw[36 + c];
#pragma unroll
      for (int k = 0; k < 9; ++k) a += sw[c * 9 + k] * in[k];
      ch[c] = a;
      mean += a;
    }
    mean *= 0.25f;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) var += (ch[c] - mean) * (ch[c] - mean);
    const float rstd = 1.f / sqrtf(var * 0.25f + 1e-6f);
    float o = sw[52];
#pragma unroll
    for (int c = 0; c < 4; ++c) o += sw[48 + c] * gelu_erf((ch[c] - mean) * rstd * sw[40 + c] + sw[44 + c]);
    out[i] = o;
  }
}

// conv7x7 s7 (1->49) -> LayerNorm2d(49) -> GELU -> conv1x1 (49->1)
// w: [conv w 49x49 | conv b 49 | ln w 49 | ln b 49 | proj w 49 | proj b 1]
__global__ __launch_bounds__(128) void mask_down2_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                         int H, int W, int OH, int OW, const float* __restrict__ w) {
  __shared__ float sw[2598];
  for (int i = threadIdx.x; i < 2598; i += 128) sw[i] = w[i];
  __syncthreads();
  const size_t i = blockIdx.x * (size_t)128 + threadIdx.x;
  if (i >= (size_t)B * OH * OW) return;
  const int ox = (int)(i % OW);
  const size_t t = i / OW;
  const int oy = (int)(t % OH);
  const int b = (int)(t / OH);
  float px[49];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) px[ky * 7 + kx] = in[((size_t)b * H + oy * 7 + ky) * W + ox * 7 + kx];
  float ch[49], mean = 0.f;
#pragma unroll
  for (int c = 0; c < 49; ++c) {
    float a = sw[2401 + c];
#pragma unroll
    for (int k = 0; k < 49; ++k) a += sw[c * 49 + k] * px[k];
    ch[c] = a;
    mean += a;
  }
  mean *= (1.f / 49.f);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < 49; ++c) var += (ch[c] - mean) * (ch[c] - mean);
  const float rstd = 1.f / sqrtf(var * (1.f / 49.f) + 1e-6f);
  float o = sw[2597];
#pragma unroll
  for (int c = 0; c < 49; ++c)
    o += sw[2548 + c] * gelu_erf((ch[c] - mean) * rstd * sw[2450 + c] + sw[2499 + c]);
  out[i] = o;
}

// Depthwise 7x7 (pad 3) on f32 NHWC (sam2/modeling/memory_encoder.py:99-117 CXBlock.dwconv). One workgroup owns
// 8 output rows x all columns x 32 channels: the 14 input rows it needs and the 49 x 32 weights are staged in LDS once
// (round 1 read every input 49 times through L2: 1.3 GB for a 45 MB tensor), then a thread produces runs of 4 adjacent
// outputs for 4 channels, re-using each staged input row segment (10 values) for 28 multiply-adds.
constexpr int DW_TH = 8, DW_CB = 32, DW_PS = 36;  // rows per tile, channels per block, padded pixel stride (floats)
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                      int W, int C, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int WT) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;                 // [49][32]
  float* sx = sw + 49 * DW_CB;              // [DW_TH + 6][WT + 6][DW_PS]  (halo of 3 columns on both sides, zero outside the image)
  // WT: columns per tile (the whole row when it fits the LDS, else 64-column tiles: maps wider than 73 pixels)
  const int cblocks = C / DW_CB, rtiles = (H + DW_TH - 1) / DW_TH, ctiles = (W + WT - 1) / WT;
  int bid = blockIdx.x;
  const int cb = (bid % cblocks) * DW_CB;
  bid /= cblocks;
  const int x0 = (bid % ctiles) * WT;
  bid /= ctiles;
  const int oy0 = (bid % rtiles) * DW_TH, b = bid / rtiles;
  const int Wt = (W - x0) < WT ? (W - x0) : WT;   // columns of this tile
  const int WP = WT + 6;
  for (int i = threadIdx.x; i < 49 * (DW_CB / 4); i += 256) {
    const int t = i / (DW_CB / 4), v = i - t * (DW_CB / 4);
    *(f32x4*)(sw + t * DW_CB + v * 4) = *(const f32x4*)(w + (size_t)t * C + cb + v * 4);
  }
  for (int i = threadIdx.x; i < (DW_TH + 6) * WP * (DW_CB / 4); i += 256) {
    const int v = i % (DW_CB / 4);
    const int p = i / (DW_CB / 4);
    const int r = p / WP, cx = p - r * WP;
    const int iy = oy0 - 3 + r, ix = x0 + cx - 3;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = *(const f32x4*)(x + (((size_t)b * H + iy) * W + ix) * C + cb + v * 4);
    *(f32x4*)(sx + (size_t)(r * WP + cx) * DW_PS + v * 4) = val;
  }
  __syncthreads();
  const int v = threadIdx.x & 7, pl = threadIdx.x >> 3;  // 8 channel quads x 32 run lanes
  const int runs_per_row = (Wt + 3) / 4;
  const f32x4 bv = *(const f32x4*)(bias + cb + v * 4);
  for (int run = pl; run < DW_TH * runs_per_row; run += 32) {
    const int ry = run / runs_per_row, ox0 = (run - ry * runs_per_row) * 4;   // tile-local column
    if (oy0 + ry >= H) break;
    f32x4 acc[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const float* row = sx + (size_t)((ry + ky) * WP + ox0) * DW_PS + v * 4;
      f32x4 in[10], wk[7];
#pragma unroll
      for (int i = 0; i < 10; ++i) in[i] = (ox0 + i < WP) ? *(const f32x4*)(row + i * DW_PS) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) wk[kx] = *(const f32x4*)(sw + (ky * 7 + kx) * DW_CB + v * 4);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) acc[o] += in[o + kx] * wk[kx];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (ox0 + o < Wt) *(f32x4*)(y + (((size_t)b * H + oy0 + ry) * W + x0 + ox0 + o) * C + cb + v * 4) = acc[o];
  }
}

inline int grid_for(size_t work, int cap = 4096) {
  const size_t b = (work + 255) / 256;
  return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

}  // namespace

extern "C" int vdn_upsample_bilinear(int dt, const void* x, const void* x_lo, void* y, void* y_lo, int B, int IH,
                                     int IW, int OH, int OW, int C, vdn_stream stream) {
  if (!x || !y || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0) return VDN_EINVAL;
  if ((C & 7) || (((uintptr_t)x | (uintptr_t)y) & 15)) return VDN_EALIGN;
  const int g = grid_for((size_t)B * OH * OW * (C >> 3), 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(upsample_kernel<VDN_F16>, dim3(g), dim3(256), 0, s, (const _Float16*)x, (const _Float16*)x_lo,
                       (_Float16*)y, (_Float16*)y_lo, B, IH, IW, OH, OW, C);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(upsample_kernel<VDN_BF16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)x_lo,
                       (__bf16*)y, (__bf16*)y_lo, B, IH, IW, OH, OW, C);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_upsample_bilinear_f32(const float* x, float* y, int B, int IH, int IW, int OH, int OW, int relu,
                                         vdn_stream stream) {
  if (!x || !y || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return VDN_EINVAL;
  hipLaunchKernelGGL(upsample_f32_kernel, dim3(grid_for((size_t)B * OH * OW)), dim3(256), 0, (hipStream_t)stream, x, y,
                     B, IH, IW, OH, OW, relu);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_patchify(int dt, const float* img, void* rows, void* rows_lo, int B, int H, int W, int ldk,
                            vdn_stream stream) {
  if (!img || !rows || B <= 0 || H <= 0 || W <= 0 || H % 14 || W % 14) return VDN_EINVAL;
  if (ldk < 588 || (ldk & 63) || ((uintptr_t)rows & 15)) return VDN_EALIGN;
  const int g = grid_for((size_t)B * (H / 14) * (W / 14) * (ldk >> 3), 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(patchify_kernel<VDN_F16>, dim3(g), dim3(256), 0, s, img, (_Float16*)rows, (_Float16*)rows_lo, B, H, W,
                       ldk);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(patchify_kernel<VDN_BF16>, dim3(g), dim3(256), 0, s, img, (__bf16*)rows, (__bf16*)rows_lo, B, H, W,
                       ldk);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_fill_row(float* x, const float* vec, int B, int rows_per_b, int row, int C, vdn_stream stream) {
  if (!x || !vec || B <= 0 || rows_per_b <= 0 || row < 0 || row >= rows_per_b || C <= 0) return VDN_EINVAL;
  hipLaunchKernelGGL(fill_row_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, vec, B,
                     rows_per_b, row, C);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_bicubic(const float* src, float* dst, int ih, int iw, int oh, int ow, int C, float scale_rows,
                           float scale_cols, vdn_stream stream) {
  if (!src || !dst || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0 || C <= 0 || scale_rows <= 0.f || scale_cols <= 0.f)
    return VDN_EINVAL;
  hipLaunchKernelGGL(bicubic_kernel, dim3(grid_for((size_t)oh * ow * C)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     ih, iw, oh, ow, C, 1.0f / scale_rows, 1.0f / scale_cols);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_preprocess(const uint8_t* frames, int n, int h, int w, int swap_rb, float* out, int H, int W, const float* mean3,
                              const float* std3, vdn_stream stream) {
  if (!frames || !out || !mean3 || !std3 || n <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return VDN_EINVAL;
  PrepNorm nm;
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] > 0.f)) return VDN_EINVAL;
    nm.mean[c] = mean3[c];
    nm.inv_std[c] = 1.0f / std3[c];
  }
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream, frames, out, n, h, w, H, W,
                     (float)h / (float)H, (float)w / (float)W, swap_rb, nm);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_head_out(int dt, const void* feat, const void* feat_lo, const float* w, float bias, float* depth,
                            int M, int C, int relu, vdn_stream stream) {
  if (!feat || !w || !depth || M <= 0 || C <= 0 || C > 64) return VDN_EINVAL;
  if ((C & 7) || ((uintptr_t)feat & 15)) return VDN_EALIGN;
  const int g = grid_for((size_t)M, 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(head_out_kernel<VDN_F16>, dim3(g), dim3(256), 0, s, (const _Float16*)feat,
                       (const _Float16*)feat_lo, w, bias, depth, M, C, relu);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(head_out_kernel<VDN_BF16>, dim3(g), dim3(256), 0, s, (const __bf16*)feat,
                       (const __bf16*)feat_lo, w, bias, depth, M, C, relu);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_mask_down1(const float* depth, float* out, int B, int H, int W, int OH, int OW, const float* w,
                              vdn_stream stream) {
  if (!depth || !out || !w || B <= 0 || OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return VDN_EINVAL;
  hipLaunchKernelGGL(mask_down1_kernel, dim3(grid_for((size_t)B * OH * OW)), dim3(256), 0, (hipStream_t)stream, depth,
                     out, B, H, W, OH, OW, w);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_mask_down2(const float* in, float* out, int B, int H, int W, int OH, int OW, const float* w,
                              vdn_stream stream) {
  if (!in || !out || !w || B <= 0 || OH != (H - 7) / 7 + 1 || OW != (W - 7) / 7 + 1 || OH <= 0 || OW <= 0)
    return VDN_EINVAL;
  const size_t total = (size_t)B * OH * OW;
  hipLaunchKernelGGL(mask_down2_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, in,
                     out, B, H, W, OH, OW, w);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_dwconv7(const float* x, float* y, int B, int H, int W, int C, const float* w, const float* bias,
                           vdn_stream stream) {
  if (!x || !y || !w || !bias || B <= 0 || H <= 0 || W <= 0 || C <= 0) return VDN_EINVAL;
  if (C % DW_CB) return VDN_EALIGN;
  // the whole row per tile while it fits the 160 KiB of LDS (W <= 73: the 37 x 37 grid of 518 x 518 inputs), else 64-column tiles
  const int WT = ((size_t)49 * DW_CB + (size_t)(DW_TH + 6) * (W + 6) * DW_PS) * sizeof(float) <= 160 * 1024 ? W : 64;
  const size_t lds = ((size_t)49 * DW_CB + (size_t)(DW_TH + 6) * (WT + 6) * DW_PS) * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute((const void*)dwconv7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return -(1000 + (int)attr);
  hipLaunchKernelGGL(dwconv7_kernel, dim3(B * ((H + DW_TH - 1) / DW_TH) * ((W + WT - 1) / WT) * (C / DW_CB)), dim3(256), lds,
                     (hipStream_t)stream, x, y, B, H, W, C, w, bias, WT);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" size_t vdn_sizeof_gemm_desc(void) { return sizeof(vdn_gemm_desc); }
extern "C" size_t vdn_offsetof_gemm_zeros(void) { return offsetof(vdn_gemm_desc, zeros); }
extern "C" size_t vdn_offsetof_gemm_res2_lo(void) { return offsetof(vdn_gemm_desc, res2_lo); }

extern "C" const char* vdn_version(void) { return "vdn-hip 0.1 (gfx950)"; }

extern "C" int vdn_arch_ok(void) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  const char* a = prop.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}
