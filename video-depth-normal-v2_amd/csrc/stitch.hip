// Device-side window stitcher for VideoDepthAnything.infer_video_depth (video_depth.py:118-156,
// utils/util.py:40-73): the closed-form scale/shift fit of a new 32-frame window to the running result
// on the 2 alignment frames, the clamp, the 8-frame linear cross-fade and the reference-frame update,
// so that a clip needs ONE device-to-host copy instead of 32 per window.
// All HBM-bound, one pass each; the sums are deterministic (fixed partial order, fp64).
#include "common.hpp"

namespace {

constexpr int SUM_BLOCKS = 256;

// partial[b] = {sum p*p, sum p, count, sum p*t, sum t} over this block's grid-stride share
__global__ __launch_bounds__(256) void align_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            size_t n, double* __restrict__ partial) {
  double s[5] = {0, 0, 0, 0, 0};
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 p = *(const f32x4*)(pred + 4 * i), t = *(const f32x4*)(target + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[0] += (double)p[e] * p[e];
      s[1] += p[e];
      s[3] += (double)p[e] * t[e];
      s[4] += t[e];
    }
    s[2] += 4.0;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail elements
    const size_t i = (n4 << 2) + threadIdx.x;
    s[0] += (double)pred[i] * pred[i];
    s[1] += pred[i];
    s[2] += 1.0;
    s[3] += (double)pred[i] * target[i];
    s[4] += target[i];
  }
  __shared__ double red[4][5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) partial[blockIdx.x * 5 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// coef = {scale, shift}: x = A^-1 b of compute_scale_and_shift_full (utils/util.py:49-60), (1, 0) when det == 0
__global__ void align_solve_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nblocks; ++b)
    for (int k = 0; k < 5; ++k) s[k] += partial[b * 5 + k];
  const double det = s[0] * s[2] - s[1] * s[1];
  if (det == 0.0) {
    coef[0] = 1.f;
    coef[1] = 0.f;
  } else {
    coef[0] = (float)((s[2] * s[3] - s[1] * s[4]) / det);
    coef[1] = (float)((-s[1] * s[3] + s[0] * s[4]) / det);
  }
}

// One pass over frames `first_fit` .. T-1 of the window (frame-major [T, hw]):
//   frames align_len .. overlap-1      : out_tail[i] = out_tail[i] * (1 - w_i) + fit(frame) * w_i   (i = 0 .. interp-1)
//   frames overlap .. T-1              : out_new[j]  = fit(frame)
//   frame ref_frame (the 2nd keyframe) : ref1        = fit(frame)                                    (extra write)
// fit(d) = max(d * scale + shift, 0) with separate multiply and add roundings, as numpy evaluates it.
__global__ __launch_bounds__(256) void align_apply_kernel(const float* __restrict__ win, const float* __restrict__ coef,
                                                          float* __restrict__ out_tail, float* __restrict__ out_new,
                                                          float* __restrict__ ref1, size_t hw, int T, int align_len,
                                                          int overlap, int ref_frame) {
  const float scale = coef[0], shift = coef[1];
  const int interp = overlap - align_len;
  const float step = 1.0f / (float)(interp - 1);
  const size_t total = (size_t)(T - align_len) * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = align_len + (int)(i / hw);
    const size_t px = i - (size_t)(f - align_len) * hw;
    const float d = fmaxf(__fadd_rn(__fmul_rn(win[(size_t)f * hw + px], scale), shift), 0.f);
    if (f < overlap) {
      const int k = f - align_len;
      const float w = k == 0 ? 0.f : (k == interp - 1 ? 1.f : (float)k * step);
      float* o = out_tail + (size_t)k * hw + px;
      *o = __fadd_rn(__fmul_rn(*o, 1.f - w), __fmul_rn(d, w));
    } else {
      out_new[(size_t)(f - overlap) * hw + px] = d;
    }
    if (f == ref_frame) ref1[px] = d;
  }
}

}  // namespace

extern "C" size_t vdn_stitch_workspace_bytes(void) { return (size_t)SUM_BLOCKS * 5 * sizeof(double); }

extern "C" int vdn_stitch_fit(const float* pred, const float* target, size_t n, void* workspace, float* coef,
                              vdn_stream stream) {
  if (!pred || !target || !workspace || !coef || n == 0) return VDN_EINVAL;
  if (((uintptr_t)pred & 15) || ((uintptr_t)target & 15) || ((uintptr_t)workspace & 7)) return VDN_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(align_partial_kernel, dim3(SUM_BLOCKS), dim3(256), 0, s, pred, target, n, (double*)workspace);
  hipLaunchKernelGGL(align_solve_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, SUM_BLOCKS, coef);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_stitch_apply(const float* window, const float* coef, float* out_tail, float* out_new, float* ref1,
                                size_t hw, int T, int align_len, int overlap, int ref_frame, vdn_stream stream) {
  if (!window || !coef || !out_tail || !out_new || !ref1 || hw == 0) return VDN_EINVAL;
  if (T <= overlap || align_len < 0 || overlap - align_len < 2 || ref_frame < align_len || ref_frame >= T) return VDN_EINVAL;
  const size_t total = (size_t)(T - align_len) * hw;
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(align_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, window, coef, out_tail, out_new,
                     ref1, hw, T, align_len, overlap, ref_frame);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
