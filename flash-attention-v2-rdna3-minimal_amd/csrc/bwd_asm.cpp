// bwd_asm.cpp — launchers of the hand-scheduled backward kernels for head dim 128 (generated inline-asm bodies, csrc/gen/bwd_d128_gen.py).
// Reference counterpart: backward_fp16 / backward_bf16 (kernel_fp16.cu:878-1028).
#include "fa2_launch.h"

#include "fa2_bwd_d128.hip.h"
#include "fa2_gfx950.h"

namespace {

// parts: bit 0 = dQ pass, bit 1 = dK/dV pass.  neg_delta: the dQ pass leaves -delta in the workspace (the hand-scheduled dK/dV pass follows).
template <bool BF16, bool CAUSAL, bool KFOLD, bool M16 = false>
int launch_dkv(const fa2::BwdParams& p, hipStream_t stream) {
    constexpr auto kern = fa2::bwd_dkv_d128_kernel<BF16, CAUSAL, KFOLD, M16>;
    if (int rc = fa2::set_lds<kern>(fa2::kBwdKvLdsBytes)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(256), fa2::kBwdKvLdsBytes, stream, p);
    return (int)hipGetLastError();
}

template <bool BF16, bool CAUSAL>
int launch_t(fa2::BwdParams p, int parts, bool neg_delta, bool kfold, bool dq16, bool dkv16, hipStream_t stream) {
    if (parts & 1) {        // dQ (+ delta): one workgroup per 256 Q rows
        p.nblk = (p.Nq + 255) / 256;
        const dim3 grid((unsigned)((int64_t)p.B * p.H * p.nblk));
        if (dq16 && neg_delta) {        // the body built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py)
            constexpr auto kern = fa2::bwd_dq_d128_kernel<BF16, CAUSAL, true, true>;
            if (int rc = fa2::set_lds<kern>(fa2::kBwdDqLdsBytes)) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(256), fa2::kBwdDqLdsBytes, stream, p);
        } else if (dq16) {
            constexpr auto kern = fa2::bwd_dq_d128_kernel<BF16, CAUSAL, false, true>;
            if (int rc = fa2::set_lds<kern>(fa2::kBwdDqLdsBytes)) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(256), fa2::kBwdDqLdsBytes, stream, p);
        } else if (neg_delta) {
            constexpr auto kern = fa2::bwd_dq_d128_kernel<BF16, CAUSAL, true>;
            if (int rc = fa2::set_lds<kern>(fa2::kBwdDqLdsBytes)) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(256), fa2::kBwdDqLdsBytes, stream, p);
        } else {
            constexpr auto kern = fa2::bwd_dq_d128_kernel<BF16, CAUSAL, false>;
            if (int rc = fa2::set_lds<kern>(fa2::kBwdDqLdsBytes)) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(256), fa2::kBwdDqLdsBytes, stream, p);
        }
        if (int rc = (int)hipGetLastError()) return rc;
    }
    if (parts & 2) {        // dK / dV: one workgroup per 128 KV rows
        p.nblk = (p.Nkv + 127) / 128;
        if ((int64_t)p.B * p.H * p.nblk > 0x7fffffffLL) return FA2_ERR_GRID;
        // (the bodies built on v_mfma_f32_16x16x32, csrc/gen/bwd_dkv_m16_gen.py, scale the f32 scores: a launch with the folded K keeps the 32x32x16 body)
        if (int rc = kfold ? launch_dkv<BF16, CAUSAL, true>(p, stream) : dkv16 ? launch_dkv<BF16, CAUSAL, false, true>(p, stream)
                                                                                : launch_dkv<BF16, CAUSAL, false>(p, stream)) return rc;
    }
    return 0;
}

}  // namespace

namespace fa2 {

int launch_bwd_d128(bool bf16, const BwdParams& p, bool causal, int parts, bool neg_delta, bool kfold, hipStream_t stream, bool dq16, bool dkv16) {
    if (bf16) return causal ? launch_t<true, true>(p, parts, neg_delta, kfold, dq16, dkv16, stream) : launch_t<true, false>(p, parts, neg_delta, kfold, dq16, dkv16, stream);
    return causal ? launch_t<false, true>(p, parts, neg_delta, kfold, dq16, dkv16, stream) : launch_t<false, false>(p, parts, neg_delta, kfold, dq16, dkv16, stream);
}

}  // namespace fa2
