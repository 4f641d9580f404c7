// 4-wave GEMM kernels (single-product loop, fused-plane 128x128 x3 loop) for f16 operands.
#define VDN_GEMM_NO_BIG 1
#include "gemm_kernels.hpp"
namespace vdn_gemm_impl {
int launch_f16(const vdn_gemm_desc& d, hipStream_t s) { return launch_dt<VDN_F16>(d, s); }
}
