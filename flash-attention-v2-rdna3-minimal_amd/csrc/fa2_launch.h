// fa2_launch.h — what the translation units of libfa2_gfx950.so share on the host side.
//
// The library is compiled as several translation units in parallel (build.py): host.cpp holds the C-ABI, the argument
// validation and the launch heuristics; the others hold the kernel instantiations of one family each and export one
// launcher per dtype.  Nothing here is part of the public boundary (include/fa2_gfx950.h): every symbol is hidden.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "fa2_bwd_kernel.hip.h"      // FwdParams / BwdParams (templates are only instantiated where a launcher names them)

#define FA2_HIDDEN __attribute__((visibility("hidden")))

namespace fa2 {

// Process-wide tuning switches (fa2_set_option / fa2_get_option in the public header; initial values from the environment,
// read once when the library is loaded).  They select between kernels that all satisfy the same contract.
struct Options {
    std::atomic<int> rows{0};      // FA2_ROWS: 0 = heuristic, 128 | 256 = rows per forward workgroup
    std::atomic<int> asm_mask{3};      // FA2_ASM: bit 0 = hand-scheduled forward bodies, bit 1 = hand-scheduled backward bodies
    std::atomic<int> persist{1};       // FA2_PERSIST: persistent workgroups of the hand-scheduled forward kernels
    std::atomic<int> bwd_parts{3};     // profiling only: bit 0 = run the dQ pass, bit 1 = run the dK / dV pass of fa2_bwd
    std::atomic<int> split{1};         // FA2_SPLIT: KV-split of the last, partly filled round of forward workgroups (fa2_fwd_ws)
};
FA2_HIDDEN Options& options();
FA2_HIDDEN int device_cus();           // compute units of the current device (cached per device index)

// Kernels that need more than 64 KiB of dynamic LDS must be opted in once per (kernel, device).  The cache is keyed on
// the kernel itself (a non-type template parameter: one flag array per instantiation, not per function-pointer type).
template <auto Kernel>
int set_lds(int bytes) {
    if (bytes <= 64 * 1024) return 0;
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && done[dev].load(std::memory_order_acquire)) return 0;
    const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc == 0 && dev >= 0) done[dev].store(true, std::memory_order_release);
    return rc;
}

// ---- one launcher per (family, dtype); `HD` is the kernel head dim (fa2_padded_head_dim)
// generic HIP forward (fwd_hip.cpp): rows = 256 (8 waves) or 128 (4 waves) per workgroup; bias: the BIAS kernels (always 128 rows)
FA2_HIDDEN int launch_fwd_hip_f16(int HD, const FwdParams& p, bool causal, int rows, bool bias, hipStream_t stream);
FA2_HIDDEN int launch_fwd_hip_bf16(int HD, const FwdParams& p, bool causal, int rows, bool bias, hipStream_t stream);
// merge of the KV-split parts a forward launch left in p.ws (fwd_hip.cpp)
FA2_HIDDEN int launch_fwd_combine_f16(int HD, const FwdParams& p, hipStream_t stream);
FA2_HIDDEN int launch_fwd_combine_bf16(int HD, const FwdParams& p, hipStream_t stream);
// hand-scheduled forward, head dim exactly 128 or 64 (fwd_asm.cpp)
FA2_HIDDEN int launch_fwd_asm(int HD, bool bf16, const FwdParams& p, bool causal, hipStream_t stream);
// HIP backward (bwd_hip.cpp): parts bit 0 = dQ pass (+ delta workspace), bit 1 = dK / dV pass(es)
FA2_HIDDEN int launch_bwd_hip_f16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
FA2_HIDDEN int launch_bwd_hip_bf16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
// HIP backward through a biased / masked forward (bwd_bias_hip.cpp): dQ, dV, dK; head dims up to 256
FA2_HIDDEN int launch_bwd_bias_hip_f16(int HD, const BwdParams& p, bool causal, hipStream_t stream);
FA2_HIDDEN int launch_bwd_bias_hip_bf16(int HD, const BwdParams& p, bool causal, hipStream_t stream);
// hand-scheduled backward, head dim exactly 128 (bwd_asm.cpp); same `parts`
// neg_delta: the dQ pass writes -delta (the hand-scheduled dK/dV pass reads it as such; the HIP dK/dV passes read +delta)
FA2_HIDDEN int launch_bwd_d128(bool bf16, const BwdParams& p, bool causal, int parts, bool neg_delta, hipStream_t stream);

constexpr int kBwdAsmParts = 3;      // passes the hand-scheduled backward covers: bit 0 = dQ, bit 1 = dK / dV

}  // namespace fa2
