// Pre/post-processing kernels of the depth-refiner wrappers (models/video_depth_model_v5.py:63-87,160-192,
// models/video_depth_model_v4.py:117-148, utils/normal_utils.py:4-51):
//   vdn_frame_median  — torch.quantile(x, 0.5) per frame (linear interpolation between the two middle order
//                       statistics) as an exact 3-pass radix select on order-preserving keys (integer atomics:
//                       deterministic), both ranks in the same passes;
//   vdn_refine_scale  — x / max_depth * exp(tanh(w * median / max_depth + b) * max_log_scale)  (GlobalScaleHead);
//   vdn_refine_pack   — network input [F,3,H,W] = (d, nx, ny) with n = (-Ix, -Iy, 1)/|.|, Sobel/8 on a reflect pad;
//   vdn_refine_finish — (scaled + (w * depth + b)) * max_depth  (scalar 1x1 'ZeroConv' shift + residual).
// All one pass over HBM.
#include "common.hpp"

namespace {

constexpr int BITS0 = 11, BITS1 = 11, BITS2 = 10, NBIN = 2048;

__device__ __forceinline__ uint32_t ordered_key(float v) {
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __builtin_bit_cast(float, u);
}

struct SelState {  // per (frame, rank)
  uint32_t prefix;  // key bits decided so far (right-aligned)
  uint32_t k;       // remaining rank inside the prefix
};

__global__ void median_init_kernel(SelState* st, uint32_t* hist, int F, size_t n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F * 2) {
    const size_t lo = (n - 1) / 2, hi = n / 2;  // floor / ceil of 0.5 (n - 1)
    st[i].prefix = 0;
    st[i].k = (uint32_t)((i & 1) ? hi : lo);
  }
  for (size_t j = i; j < (size_t)F * 2 * NBIN; j += (size_t)gridDim.x * blockDim.x) hist[j] = 0;
}

// PASS 0: top 11 bits; PASS 1: next 11 among keys whose top 11 equal the prefix; PASS 2: last 10
template <int PASS>
__global__ __launch_bounds__(256) void median_hist_kernel(const float* __restrict__ x, size_t n, const SelState* __restrict__ st,
                                                          uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[2][NBIN];
  const int f = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * NBIN; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint32_t p0 = st[f * 2].prefix, p1 = st[f * 2 + 1].prefix;
  const float* xf = x + (size_t)f * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t key = ordered_key(xf[i]);
    uint32_t bin, pre;
    if (PASS == 0) { bin = key >> (32 - BITS0); pre = 0; }
    else if (PASS == 1) { bin = (key >> BITS2) & ((1u << BITS1) - 1); pre = key >> (BITS1 + BITS2); }
    else { bin = key & ((1u << BITS2) - 1); pre = key >> BITS2; }
    if (PASS == 0 || pre == p0) atomicAdd(&h[0][bin], 1u);
    if (PASS == 0 || pre == p1) atomicAdd(&h[1][bin], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * NBIN; i += 256) {
    const uint32_t v = (&h[0][0])[i];
    if (v) atomicAdd(hist + (size_t)f * 2 * NBIN + i, v);
  }
}

// one block per (frame, rank): walk the histogram to the bin holding rank k, extend the prefix, clear the bins
template <int PASS>
__global__ void median_scan_kernel(SelState* st, uint32_t* hist) {
  uint32_t* h = hist + (size_t)blockIdx.x * NBIN;
  if (threadIdx.x == 0) {
    SelState s = st[blockIdx.x];
    const int nb = PASS == 2 ? (1 << BITS2) : NBIN;
    uint32_t cum = 0;
    int b = 0;
    for (; b < nb - 1; ++b) {
      if (cum + h[b] > s.k) break;
      cum += h[b];
    }
    s.k -= cum;
    s.prefix = (s.prefix << (PASS == 0 ? BITS0 : (PASS == 1 ? BITS1 : BITS2))) | (uint32_t)b;
    st[blockIdx.x] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBIN; i += blockDim.x) h[i] = 0;
}

__global__ void median_final_kernel(const SelState* st, int F, size_t n, float* median) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const float a = key_value(st[f * 2].prefix), b = key_value(st[f * 2 + 1].prefix);
  const float w = (n & 1) ? 0.f : 0.5f;  // fractional part of 0.5 (n - 1)
  median[f] = w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);  // at::lerp's two-sided form
}

__global__ __launch_bounds__(256) void refine_scale_kernel(const float* __restrict__ x, const float* __restrict__ median, float w,
                                                           float b, float max_log_scale, float max_depth, float* __restrict__ out,
                                                           float* __restrict__ scale_out, size_t n) {
  const int f = blockIdx.y;
  const float s = __expf(tanhf(__fadd_rn(__fmul_rn(__fdiv_rn(median[f], max_depth), w), b)) * max_log_scale);
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) scale_out[f] = s;
  const float* xf = x + (size_t)f * n;
  float* of = out + (size_t)f * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    of[i] = __fmul_rn(__fdiv_rn(xf[i], max_depth), s);
}

__global__ __launch_bounds__(256) void refine_pack_kernel(const float* __restrict__ d, float* __restrict__ out, int F, int H, int W,
                                                          int normals) {
  const size_t hw = (size_t)H * W, total = (size_t)F * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i / hw);
    const int p = (int)(i - (size_t)f * hw);
    const int y = p / W, x = p - y * W;
    const float* df = d + (size_t)f * hw;
    float* of = out + (size_t)f * 3 * hw;
    const float c = df[p];
    of[p] = c;
    if (!normals) {
      of[hw + p] = c;
      of[2 * hw + p] = c;
      continue;
    }
    const int ym = y == 0 ? 1 : y - 1, yp = y == H - 1 ? H - 2 : y + 1;  // reflect (no edge repeat)
    const int xm = x == 0 ? 1 : x - 1, xp = x == W - 1 ? W - 2 : x + 1;
    const float a00 = df[ym * W + xm], a01 = df[ym * W + x], a02 = df[ym * W + xp];
    const float a10 = df[y * W + xm], a12 = df[y * W + xp];
    const float a20 = df[yp * W + xm], a21 = df[yp * W + x], a22 = df[yp * W + xp];
    // cross-correlation with kx = [[1,0,-1],[2,0,-2],[1,0,-1]]/8, ky = [[1,2,1],[0,0,0],[-1,-2,-1]]/8
    const float ix = ((a00 - a02) + 2.f * (a10 - a12) + (a20 - a22)) * 0.125f;
    const float iy = ((a00 - a20) + 2.f * (a01 - a21) + (a02 - a22)) * 0.125f;
    const float inv = 1.0f / sqrtf(ix * ix + iy * iy + 1.0f + 1e-8f);
    of[hw + p] = -ix * inv;
    of[2 * hw + p] = -iy * inv;
  }
}

__global__ __launch_bounds__(256) void refine_finish_kernel(const float* __restrict__ scaled, const float* __restrict__ depth, float w,
                                                            float b, float max_depth, int residual, float* __restrict__ out,
                                                            size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = depth[i];
    out[i] = residual ? __fmul_rn(__fadd_rn(scaled[i], __fadd_rn(__fmul_rn(d, w), b)), max_depth) : __fmul_rn(d, max_depth);
  }
}

inline unsigned blocks_for(size_t n, unsigned cap) {
  const size_t b = (n + 255) / 256;
  return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

}  // namespace

extern "C" size_t vdn_frame_median_workspace_bytes(int frames) {
  return frames <= 0 ? 0 : (size_t)frames * 2 * (NBIN * sizeof(uint32_t) + sizeof(SelState));
}

extern "C" int vdn_frame_median(const float* x, int frames, size_t n, float* median, void* workspace, vdn_stream stream) {
  if (!x || !median || !workspace || frames <= 0 || n == 0) return VDN_EINVAL;
  if (n >= ((size_t)1 << 32)) return VDN_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* hist = (uint32_t*)workspace;
  SelState* st = (SelState*)(hist + (size_t)frames * 2 * NBIN);
  const unsigned per_frame = blocks_for(n, 64);
  hipLaunchKernelGGL(median_init_kernel, dim3(blocks_for((size_t)frames * 2 * NBIN, 1024)), dim3(256), 0, s, st, hist, frames, n);
  hipLaunchKernelGGL(median_hist_kernel<0>, dim3(per_frame, frames), dim3(256), 0, s, x, n, st, hist);
  hipLaunchKernelGGL(median_scan_kernel<0>, dim3(frames * 2), dim3(256), 0, s, st, hist);
  hipLaunchKernelGGL(median_hist_kernel<1>, dim3(per_frame, frames), dim3(256), 0, s, x, n, st, hist);
  hipLaunchKernelGGL(median_scan_kernel<1>, dim3(frames * 2), dim3(256), 0, s, st, hist);
  hipLaunchKernelGGL(median_hist_kernel<2>, dim3(per_frame, frames), dim3(256), 0, s, x, n, st, hist);
  hipLaunchKernelGGL(median_scan_kernel<2>, dim3(frames * 2), dim3(256), 0, s, st, hist);
  hipLaunchKernelGGL(median_final_kernel, dim3((frames + 63) / 64), dim3(64), 0, s, st, frames, n, median);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_refine_scale(const float* x, const float* median, int frames, size_t n, float w, float b, float max_log_scale,
                                float max_depth, float* out, float* scale_out, vdn_stream stream) {
  if (!x || !median || !out || frames <= 0 || n == 0 || !(max_depth > 0.f)) return VDN_EINVAL;
  hipLaunchKernelGGL(refine_scale_kernel, dim3(blocks_for(n, 256), frames), dim3(256), 0, (hipStream_t)stream, x, median, w, b,
                     max_log_scale, max_depth, out, scale_out, n);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_refine_pack(const float* d, float* out, int frames, int H, int W, int normals, vdn_stream stream) {
  if (!d || !out || frames <= 0 || H < 2 || W < 2) return VDN_EINVAL;
  hipLaunchKernelGGL(refine_pack_kernel, dim3(blocks_for((size_t)frames * H * W, 16384)), dim3(256), 0, (hipStream_t)stream, d, out,
                     frames, H, W, normals);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_refine_finish(const float* scaled, const float* depth, float w, float b, float max_depth, int residual,
                                 float* out, size_t n, vdn_stream stream) {
  if (!depth || !out || n == 0 || (residual && !scaled)) return VDN_EINVAL;
  hipLaunchKernelGGL(refine_finish_kernel, dim3(blocks_for(n, 16384)), dim3(256), 0, (hipStream_t)stream, scaled, depth, w, b,
                     max_depth, residual, out, n);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
