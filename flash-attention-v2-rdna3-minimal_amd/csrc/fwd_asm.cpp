// fwd_asm.cpp — launchers of the hand-scheduled forward kernels (generated inline-asm bodies, csrc/gen/).
// Reference counterpart: the launch at the end of forward_fp16 / forward_bf16 (kernel_fp16.cu:808-851).
#include "fa2_launch.h"

#include "fa2_fwd_d128.hip.h"
#include "fa2_gfx950.h"

namespace {

// Persistent workgroups of the d128 kernel (non-causal launches): at most one workgroup per CU, each working through a
// strided list of (head, q block) items and fetching the next item's first tiles while the current one finishes
// (fa2_fwd_d128.hip.h).  Option "persist" = 0 launches one workgroup per item instead (A/B measurements, bit-identity tests).
template <bool BF16, bool CAUSAL>
int launch_d128_t(const fa2::FwdParams& p, hipStream_t stream) {
    constexpr auto kern = fa2::fwd_d128_kernel<BF16, CAUSAL>;
    if (int rc = fa2::set_lds<kern>(fa2::kD128LdsBytes)) return rc;
    int64_t grid = (int64_t)p.nbh * p.nqblk;
    const int pg = fa2::options().persist.load(std::memory_order_relaxed) ? fa2::device_cus() & ~7 : 0;   // a multiple of 8: an item stays on its head's XCD
    if (!CAUSAL && pg > 0 && grid > pg) grid = pg;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), fa2::kD128LdsBytes, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

namespace fa2 {

int launch_fwd_d128(bool bf16, const FwdParams& p, bool causal, hipStream_t stream) {
    if (bf16) return causal ? launch_d128_t<true, true>(p, stream) : launch_d128_t<true, false>(p, stream);
    return causal ? launch_d128_t<false, true>(p, stream) : launch_d128_t<false, false>(p, stream);
}

}  // namespace fa2
