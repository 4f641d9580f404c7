// Fused depth tail of the DPT head (depth_anything_v2/dpt.py:146-151, video_depth_anything/dpt_temporal.py:106-111):
//   out = F.interpolate(out1, (14 ph, 14 pw), bilinear, align_corners=True)      [B, C, OH, OW]   (C = features / 2)
//   out = ReLU(Conv3x3(C -> 32, pad 1)(out))
//   depth = [ReLU](Conv1x1(32 -> 1)(out))                                          [B, OH, OW] f32
// in ONE kernel: the x1.75 up-sampled map (1.1 GB of split planes at batch 8, ViT-L) and the 32-channel map never
// reach HBM. Round 1 ran this as upsample_kernel -> implicit-GEMM conv (N = 32: 9.9 GB moved L2 -> LDS per launch,
// feed-bound) -> head_out_kernel = 1.6 ms per batch-8 step.
//
// One workgroup (4 waves) owns a 16 x 16 output tile. For each block of 32 input channels:
//   (out1 arrives as ONE fp32 plane: reading split halves cost 12 conversions + adds per value in the fill, which is
//   VALU-bound: 300 -> ~110 instructions per 8-channel item)
//   stage: the <= 13 x 13 out1 pixels under the tile's halo are copied ONCE into LDS (loads issued before the previous
//          block's MFMAs, written after them). Fetching the 4 bilinear corners of every up-sampled pixel from global
//          memory instead re-read each source pixel ~9 times through L2 (5.7 GB per launch: 1.0 ms, L2-bound);
//   fill:  the 18 x 18 halo patch of the UP-SAMPLED map is interpolated from that copy (fp32 lerp, split into hi/lo
//          halves) into LDS: 64-byte rows per pixel and plane, 16-byte chunk XOR-swizzled by (row + column / 4) so that
//          the 16 consecutive pixels of a fragment read hit 16 bank groups; pixels outside the image are the
//          convolution's zero padding;
//   mma:   per tap a wave multiplies its 8 rows x 16 pixels by ITS 16 output channels' 16 x 32 weight block on
//          v_mfma_f32_16x16x32 (weights as the A operand: a lane then holds 4 output channels of ONE pixel), three
//          products per term (hi*lo + lo*hi + hi*hi). The pass's 9 weight fragments per plane live in registers.
// Epilogue: bias + ReLU + each wave's half of the 32 -> 1 dot product in registers, two cross-lane adds, the two
//          halves summed through LDS, one coalesced store per tile row.
// Roofline: MFMA (19.8 GF per 518 x 518 frame at C = 128, x3 executed); HBM traffic = out1 once (+ halo) + depth.
#include "common.hpp"

namespace {

constexpr int TW = 16, TH = 16, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;  // 18 x 18 = 324 halo pixels
constexpr int CB = 32;                                                      // input channels per pass
constexpr int PWS = 20;                                                     // LDS row stride of the patch in pixels
constexpr int PLANE = PH * PWS * 64;                                        // bytes per plane of the patch
constexpr int SP = 13, SPLANE = SP * SP * 64;                               // source (out1) patch: 13 x 13 pixels x 32 f32 = 2 SPLANE bytes
constexpr int SLOADS = (SP * SP * 4 * 2 + 255) / 256;                       // 16-byte source loads per thread and pass

__device__ __forceinline__ void ac_coord(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float src = scale * (float)o;  // PyTorch's align_corners=True source index (same as upsample_kernel)
  i0 = (int)src;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}

template <int DT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void depth_tail_kernel(const float* __restrict__ x, int B, int IH, int IW,
                                                         int C, const typename Half<DT>::T* __restrict__ w,
                                                         const typename Half<DT>::T* __restrict__ wl, int ldb,
                                                         const float* __restrict__ b2, const float* __restrict__ w1, float b1,
                                                         float* __restrict__ depth, int OH, int OW, int relu) {
  using H = Half<DT>;
  using T = typename H::T;
  using V8 = typename H::V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // up patch [hi | lo] x NPIX x 64 B, then source [hi | lo] x SP^2 x 64 B
  char* ssrc = smem + 2 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // an XCD walks consecutive tiles of one image: halo rows hit its L2
  const int b = tile / (tiles_x * tiles_y);
  const int trem = tile - b * tiles_x * tiles_y;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
  const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  const float* xb = x + (size_t)b * IH * IW * C;

  // wave (wj, wr): output channels 16 wj .. 16 wj + 15 of tile rows 8 wr .. 8 wr + 7. One 16-channel block per wave
  // means the WHOLE pass's weights (9 taps x 16 x 32, both planes) fit its registers (72 VGPRs): they are loaded at the
  // top of the pass and their L2 latency hides under the stage / fill phases. (Fetching each tap's fragments inside the
  // tap loop, one tap ahead, left ~0.7 us of L2 latency exposed per tap: 25 us per workgroup, 0.9 ms per launch.)
  const int fr = lane & 15, fq = lane >> 4;
  const int wj = wave & 1, wr = wave >> 1;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* wrow = w + (size_t)(16 * wj + fr) * ldb + fq * 8;  // A operand: W[16 wj + fr][tap C + cb + 8 fq ..]
  const ptrdiff_t wdelta = (const char*)wl - (const char*)w;
  u32x4 wh[9], wlo[9];  // opaque 16-byte registers: bit-cast to fragments at the MFMA
  // The patch is stored with a 20-pixel row stride and the 16-byte chunk of pixel (row, col) XOR-ed with
  // (row + col / 4) & 3: a fragment read then depends on (kx, row & 3) only — 12 base addresses per lane, everything else
  // an instruction immediate (with an 18-pixel stride the 72 (tap, row) offsets were all different: the compiler
  // hoisted them out of the channel loop and spilled).
  const char* abase[3][4];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int rm = 0; rm < 4; ++rm)
      abase[kx][rm] = smem + (wr * 8 * PWS + fr + kx) * 64 + ((fq ^ ((rm + ((fr + kx) >> 2)) & 3)) << 4);

  // ---- source patch: origin = the source pixel under the halo's first row / column
  int dummy;
  float dl;
  int sy0, sx0;
  ac_coord(ty0 > 0 ? ty0 - 1 : 0, sy, IH, sy0, dummy, dl);
  ac_coord(tx0 > 0 ? tx0 - 1 : 0, sx, IW, sx0, dummy, dl);
  // LDS-DMA of the patch: item = (pixel, 4-float piece) in LDS order, 16 B per lane, 1 KiB per wave-instruction; no
  // registers are held while it is in flight (a register-staged prefetch across the MFMA phase spilled)
  auto dma_src = [&](int cb) {
    int tid_d = tid;
    asm volatile("" : "+v"(tid_d));  // as in the fill loop: recompute the addresses per pass instead of keeping them
#pragma unroll
    for (int i = 0; i < SLOADS; ++i) {
      const int it = tid_d + i * 256;
      if (it < SP * SP * 8) {
        const int pix = it >> 3, c4 = it & 7;  // 128 B per pixel = 8 pieces of 4 floats
        int yy = sy0 + pix / SP, xx = sx0 + pix % SP;
        yy = yy < IH ? yy : IH - 1;
        xx = xx < IW ? xx : IW - 1;
        const float* src = xb + ((size_t)yy * IW + xx) * C + cb + c4 * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ssrc + (i * 4 + wave) * 1024), 16, 0, 0);
      }
    }
  };
  dma_src(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int cb = 0; cb < C; cb += CB) {
    __syncthreads();  // src(cb) is complete and visible; the previous pass's fragment reads (up patch) are done
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {  // this pass's weights: 18 loads per lane, in flight during the fill phase
      wh[tap] = *(const u32x4*)(wrow + tap * C + cb);
      wlo[tap] = *(const u32x4*)((const char*)(wrow + tap * C + cb) + wdelta);
    }
    // ---- fill: up-sampled halo patch of channels cb .. cb+31, 4 chunks of 8 channels per pixel, from the LDS copy
    // (the coordinates do not depend on the channel block: an opaque copy of tid keeps the compiler from hoisting ~60
    //  registers of them out of the channel loop, where they would live across the MFMA phase and spill)
    int tid_c = tid;
    asm volatile("" : "+v"(tid_c));
    for (int it = tid_c; it < NPIX * 4; it += 256) {
      const int pix = it >> 2, ch = it & 3;
      const int py = pix / PW, px = pix - py * PW;
      const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      V8 oh, ol;
#pragma unroll
      for (int e = 0; e < 8; ++e) { oh[e] = (T)0.f; ol[e] = (T)0.f; }
      if (gy >= 0 && gy < OH && gx >= 0 && gx < OW) {
        int y0, y1, x0, x1;
        float ly, lx;
        ac_coord(gy, sy, IH, y0, y1, ly);
        ac_coord(gx, sx, IW, x0, x1, lx);
        const float* c00 = (const float*)ssrc + ((y0 - sy0) * SP + (x0 - sx0)) * 32 + ch * 8;
        const float* c01 = (const float*)ssrc + ((y0 - sy0) * SP + (x1 - sx0)) * 32 + ch * 8;
        const float* c10 = (const float*)ssrc + ((y1 - sy0) * SP + (x0 - sx0)) * 32 + ch * 8;
        const float* c11 = (const float*)ssrc + ((y1 - sy0) * SP + (x1 - sx0)) * 32 + ch * 8;
        f32x4 a00[2], a01[2], a10[2], a11[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          a00[u] = *(const f32x4*)(c00 + 4 * u); a01[u] = *(const f32x4*)(c01 + 4 * u);
          a10[u] = *(const f32x4*)(c10 + 4 * u); a11[u] = *(const f32x4*)(c11 + 4 * u);
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float r[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int u = (e + q) >> 2, k = (e + q) & 3;
            const float top = (1.f - lx) * a00[u][k] + lx * a01[u][k];
            const float bot = (1.f - lx) * a10[u][k] + lx * a11[u][k];
            r[q] = (1.f - ly) * top + ly * bot;
          }
          T h0, h1, q0, q1;
          split2_rtz(r[0], r[1], h0, h1, q0, q1);
          oh[e] = h0; oh[e + 1] = h1; ol[e] = q0; ol[e + 1] = q1;
        }
      }
      const int off = (py * PWS + px) * 64 + ((ch ^ ((py + (px >> 2)) & 3)) << 4);
      *(V8*)(smem + off) = oh;
      *(V8*)(smem + PLANE + off) = ol;
    }
    // The weights have landed long ago; retire them HERE, explicitly, so that the compiler does not put its own
    // vmcnt(0) in front of the first MFMA — which would also wait for the source DMA issued just below.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]), "+v"(wh[4]), "+v"(wh[5]), "+v"(wh[6]), "+v"(wh[7]), "+v"(wh[8]),
                   "+v"(wlo[0]), "+v"(wlo[1]), "+v"(wlo[2]), "+v"(wlo[3]), "+v"(wlo[4]), "+v"(wlo[5]), "+v"(wlo[6]), "+v"(wlo[7]),
                   "+v"(wlo[8])
                 :
                 : "memory");
    __syncthreads();  // the up patch is complete; the source buffer is free
    if (cb + CB < C) dma_src(cb + CB);  // lands while this pass multiplies
    // ---- mma: patch row rho = i + ky of column shift kx feeds up to three taps (ky = 0, 1, 2 -> output rows rho - ky):
    // one fragment read per (kx, rho) = 30 per plane instead of 72 (one per (tap, row)), each followed by up to 9 MFMAs,
    // and the read of the NEXT (kx, rho) is issued before them (a read that 3 MFMAs wait for left the loop LDS-latency
    // bound: 124 cycles per MFMA measured).
    {
      V8 fh[2], fl[2];
      auto rd = [&](int idx, V8& h, V8& l) {  // idx = kx * 10 + rho
        const int kx = idx / 10, rho = idx - kx * 10;
        const char* ap = abase[kx][rho & 3] + rho * (PWS * 64);
        h = *(const V8*)ap;
        l = *(const V8*)(ap + PLANE);
      };
      rd(0, fh[0], fl[0]);
#pragma unroll
      for (int idx = 0; idx < 30; ++idx) {
        const int kx = idx / 10, rho = idx - kx * 10, cur = idx & 1;
        if (idx + 1 < 30) rd(idx + 1, fh[cur ^ 1], fl[cur ^ 1]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int i = rho - ky, tap = ky * 3 + kx;
          if (i >= 0 && i < 8) {
            f32x4 c = acc[i];
            c = H::mfma16(__builtin_bit_cast(V8, wh[tap]), fl[cur], c);
            c = H::mfma16(__builtin_bit_cast(V8, wlo[tap]), fh[cur], c);
            c = H::mfma16(__builtin_bit_cast(V8, wh[tap]), fh[cur], c);
            acc[i] = c;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next pass's source patch has landed (before the loop-top barrier)
  }

  // ---- epilogue: lane (fr, fq) holds output channels 16 wj + 4 fq + e of pixel (row 8 wr + i, column fr): bias + ReLU +
  // this wave's half of the 32 -> 1 dot product; the two halves (wj = 0, 1) meet through LDS
  const f32x4 bias4 = *(const f32x4*)(b2 + wj * 16 + fq * 4), w14 = *(const f32x4*)(w1 + wj * 16 + fq * 4);
  __syncthreads();  // the last pass's fragment reads are done: the patch memory is free
  float* part = (float*)smem;  // [2][TH * TW]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[i][e] + bias4[e];
      v = (v < 0.f) ? 0.f : v;  // NaN passes through
      sum = fmaf(v, w14[e], sum);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (fq == 0) part[wj * (TH * TW) + (wr * 8 + i) * TW + fr] = sum;
  }
  __syncthreads();
  {
    const int py = tid >> 4, px = tid & 15;  // 256 threads = 16 x 16 pixels
    float v = part[py * TW + px] + part[TH * TW + py * TW + px] + b1;
    if (relu) v = (v < 0.f) ? 0.f : v;
    const int oy = ty0 + py, ox = tx0 + px;
    if (oy < OH && ox < OW) depth[((size_t)b * OH + oy) * OW + ox] = v;
  }
}

}  // namespace

extern "C" int vdn_depth_tail(int dt, const float* x, int B, int IH, int IW, int C, const void* w,
                              const void* w_lo, int ldb, const float* bias2, const float* w1, float b1, float* depth, int OH,
                              int OW, int relu, vdn_stream stream) {
  if (!x || !w || !w_lo || !bias2 || !w1 || !depth || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return VDN_EINVAL;
  if (C <= 0 || (C % CB) || ldb < 9 * C || (ldb & 7)) return VDN_EALIGN;
  // the 13 x 13 source patch must cover the 18-pixel halo of a tile: floor(17 s) + 3 <= 13 for both scales
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  if ((int)(17.f * sy) + 3 > SP || (int)(17.f * sx) + 3 > SP) return VDN_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)w_lo | (uintptr_t)bias2 | (uintptr_t)w1) & 15) return VDN_EALIGN;
  const int tiles = B * ((OH + TH - 1) / TH) * ((OW + TW - 1) / TW);
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(depth_tail_kernel<VDN_F16>, dim3(tiles), dim3(256), 2 * PLANE + 2 * SPLANE, s, x, B,
                       IH, IW, C, (const _Float16*)w, (const _Float16*)w_lo, ldb, bias2, w1, b1, depth, OH, OW, relu);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(depth_tail_kernel<VDN_BF16>, dim3(tiles), dim3(256), 2 * PLANE + 2 * SPLANE, s, x, B, IH,
                       IW, C, (const __bf16*)w, (const __bf16*)w_lo, ldb, bias2, w1, b1, depth, OH, OW, relu);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
