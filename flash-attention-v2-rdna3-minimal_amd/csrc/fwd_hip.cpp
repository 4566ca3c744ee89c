// fwd_hip.cpp — instantiations and launchers of the compiler-scheduled forward kernels (fa2_fwd_kernel.hip.h) for ONE dtype:
// build.py compiles this file twice, -DFA2_TU_BF16=0 and =1, in parallel with the other translation units.
// Reference counterpart: the template dispatch at the end of forward_fp16 / forward_bf16 (kernel_fp16.cu:841-851).
#include "fa2_launch.h"

#include "fa2_fwd_short.hip.h"
#include "fa2_gfx950.h"

#ifndef FA2_TU_BF16
#error "compile with -DFA2_TU_BF16=0 or 1"
#endif
// -DFA2_TU_TRIM=1: this unit holds the TRIMMED instantiations instead (head dims below the kernel's HD run only the MFMA k-steps and O
// column blocks that hold real columns: fa2_fwd_kernel.hip.h, KSQ / DTN / RTD) and exports launch_fwd_hip_trim_{f16,bf16}.
#ifndef FA2_TU_TRIM
#define FA2_TU_TRIM 0
#endif
#ifndef FA2_TRIM          // 0: never dispatch to the trimmed kernels (A/B builds, tools/kbench.py)
#define FA2_TRIM 1
#endif
#ifndef FA2_TRIM256_MODE
#define FA2_TRIM256_MODE 2
#endif

namespace {

constexpr bool kBF16 = FA2_TU_BF16 != 0;

template <int HD, bool CAUSAL, int NW, int BIAS = 0, int KSQ = HD / 16, int DTN = (HD > 128 ? 128 : HD) / 32, bool RTD = false, int HDV_ = 0>
int launch_shape(const fa2::FwdParams& p0, hipStream_t stream) {
    constexpr int HDV = HDV_ ? HDV_ : HD > 128 ? 128 : HD;       // (HDV_ = 256: one pass over all columns, trimmed head dims 129..192 only)
    constexpr int lds_kv = 2 * fa2::Geo<HD, NW>::TILEB + 2 * fa2::Geo<HDV, NW>::TILEB;
    constexpr int lds_epi = FA2_EPI_LDS ? NW * 32 * (HDV * 2 + 16) : 0;     // epilogue image (reuses the K/V space)
    // bias kernels: + NW wave-private 32-row images of the "tile" bias form where they fit (not at D = 512: 160 KiB of K / V buffers)
    // (BIAS = 2, the LDS-DMA form: NW images of 8 KiB)
    constexpr int lds_bias = BIAS == 2 ? NW * 8192 : BIAS && lds_kv + NW * 32 * 272 <= 160 * 1024 ? NW * 32 * 272 : 0;
    constexpr int lds = lds_kv + lds_bias > lds_epi ? lds_kv + lds_bias : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    fa2::FwdParams p = p0;
    p.nqblk = (p.Nq + NW * 32 - 1) / (NW * 32);
    if ((int64_t)p.nbh * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;
    int64_t nblk = (int64_t)p.nbh * p.nqblk;
    if constexpr (NW == 8 && !BIAS && !CAUSAL && HD == HDV) {
        // KV-split tail (host.cpp: plan_split): the whole items from blk0 on, then split_items * nsplit parts
        if (p.nsplit > 1) nblk = (int64_t)p.full_items - p.blk0 + (int64_t)p.split_items * p.nsplit;
    } else {
        p.nsplit = 0;
    }
    const dim3 grid((unsigned)nblk, (p.D + HDV - 1) / HDV);      // column slabs that hold real columns (HD / HDV of them at most)
    constexpr auto kern = fa2::fwd_kernel<HD, HDV, kBF16, CAUSAL, NW, 1, BIAS, KSQ, DTN, RTD>;
    if (int rc = fa2::set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
    return (int)hipGetLastError();
}

#if FA2_TU_TRIM

// Trimmed kernels (no bias).  Returns -1 when the head dim has no trimmed instantiation (the caller runs the full kernel).
//   HD  64: D <= 32 -> 2 k-steps, 1 column block;  D <= 48 -> 3, 2 (SD 1.5's head dim 40)
//   HD 128: D <= 96 -> 6, 3 (head dims 80, 96)
//   HD 256: D <= 160 / 192 / 224 -> 10 / 12 / 14 k-steps; the second column half runs ceil((D - 128) / 32) of its 4 blocks
// Measured (tools/trim_ab.py, profiles/r08_trim_ab.txt; B1 H24 N4096 fp16, the reference harness's D-scan): D 16 / 32 +27 %, 40 / 48 +4 %,
// 80 / 96 +12 %, 144 / 160 +19 %, 176 / 192 +10 %, 208 / 224 +3 %; bf16 causal B2 H16 D96 N4096 +9.5 %, D192 N2048 +27 %.
template <int HD, int KSQ, int DTN, bool RTD>
int launch_trim(const fa2::FwdParams& p, bool causal, int rows, hipStream_t stream) {
    if (causal) return rows == 128 ? launch_shape<HD, true, 4, 0, KSQ, DTN, RTD>(p, stream) : launch_shape<HD, true, 8, 0, KSQ, DTN, RTD>(p, stream);
    return rows == 128 ? launch_shape<HD, false, 4, 0, KSQ, DTN, RTD>(p, stream) : launch_shape<HD, false, 8, 0, KSQ, DTN, RTD>(p, stream);
}

}  // namespace

namespace fa2 {

#if FA2_TU_BF16
int launch_fwd_hip_trim_bf16(int HD, const FwdParams& p, bool causal, int rows, hipStream_t stream) {
#else
int launch_fwd_hip_trim_f16(int HD, const FwdParams& p, bool causal, int rows, hipStream_t stream) {
#endif
    switch (HD) {
        case 64:
            if (p.D <= 32) return launch_trim<64, 2, 1, false>(p, causal, rows, stream);
            if (p.D <= 48) return launch_trim<64, 3, 2, false>(p, causal, rows, stream);
            return -1;
        case 128:
            if (p.D <= 80) return launch_trim<128, 5, 3, false>(p, causal, rows, stream);      // (SD 1.5's head dim 80: 5 k-steps exactly)
            if (p.D <= 96) return launch_trim<128, 6, 3, false>(p, causal, rows, stream);
            return -1;
        case 256:
#if FA2_TRIM256_MODE >= 1
            // Grids wide enough for 256-row workgroups: ONE pass over all columns instead of two column halves that each recompute Q.K^T
            // (D <= 160: per 32 x 64 score tile 20 + 20 MFMAs instead of 2 x 20 + 16 + 4).  D <= 160: the accumulators of 5 column blocks and
            // 10 Q fragments fit the 8-wave shape's 256 registers; above: 4-wave workgroups of 128 rows, one wave per SIMD (512 registers).
            // Measured (tools/trim_ab.py --dmin 129, profiles/r08_trim256_ab.txt, B1 H24 N4096): D 144 / 160 386 / 390 -> 283 / 287 us (898 TF at
            // D = 160), 176 / 192 430 / 436 -> 389 / 390; on small grids (rows == 128) the column halves — twice the workgroups — stay ahead.
            if (rows != 128) {
                if (p.D <= 160)
                    return causal ? launch_shape<256, true, 8, 0, 10, 5, false, 256>(p, stream) : launch_shape<256, false, 8, 0, 10, 5, false, 256>(p, stream);
                if (p.D <= 192)
                    return causal ? launch_shape<256, true, 4, 0, 12, 6, false, 256>(p, stream) : launch_shape<256, false, 4, 0, 12, 6, false, 256>(p, stream);
#if FA2_TRIM256_MODE >= 2
                if (p.D <= 224)
                    return causal ? launch_shape<256, true, 4, 0, 14, 7, false, 256>(p, stream) : launch_shape<256, false, 4, 0, 14, 7, false, 256>(p, stream);
                // every column (D <= 256), causal only: the non-causal instantiation spills 27 registers and measured 547 -> 700 us at B1 H24 N4096,
                // the causal one fits: bf16 B2 H16 N4096 D256 414 -> 351 us (profiles/r08_trim256_ab.txt)
                if (causal) return launch_shape<256, true, 4, 0, 16, 8, false, 256>(p, stream);
#endif
            }
#endif
            if (p.D <= 160) return launch_trim<256, 10, 4, true>(p, causal, rows, stream);
            if (p.D <= 192) return launch_trim<256, 12, 4, true>(p, causal, rows, stream);
            if (p.D <= 224) return launch_trim<256, 14, 4, true>(p, causal, rows, stream);
            return -1;
        case 512:
            // 128-column slabs of O per 128-row workgroup, each recomputing Q.K^T over the head dim: ceil(D / 16) k-steps instead of 32 (and only the
            // slabs that hold real columns are launched at all: launch_shape); the last slab runs ceil((D mod 128) / 32) of its 4 blocks
            if (p.D <= 320) return causal ? launch_shape<512, true, 4, 0, 20, 4, true>(p, stream) : launch_shape<512, false, 4, 0, 20, 4, true>(p, stream);
            if (p.D <= 384) return causal ? launch_shape<512, true, 4, 0, 24, 4, true>(p, stream) : launch_shape<512, false, 4, 0, 24, 4, true>(p, stream);
            if (p.D <= 448) return causal ? launch_shape<512, true, 4, 0, 28, 4, true>(p, stream) : launch_shape<512, false, 4, 0, 28, 4, true>(p, stream);
            return -1;
        default: return -1;
    }
}

}  // namespace fa2

#else   // !FA2_TU_TRIM

// D = 256 runs as two, D = 512 as four 128-column slabs of O per Q block (grid.y), recomputing QK^T per slab.
// D = 512 (the reference's D > 384 path, FlashAttn.py:65-67; the SD VAE attention block): 4-wave workgroups of 128 Q
// rows, one wave per SIMD — the 128 registers of Q fragments per wave need the 512-register budget — and all 160 KiB
// of LDS (two 64 KiB K tiles + two 16 KiB V tiles).  A correct path for a rare shape, not a tuned one.
// Attention bias / boolean mask: every head dim runs as 4-wave, 128-row workgroups — one wave per SIMD, so the 32 bias
// registers per tile come out of the 512-register budget instead of spilling (the 8-wave shape at D = 64 spills 62-67 VGPRs).
template <int HD, bool CAUSAL>
int launch_t(const fa2::FwdParams& p, int rows, bool bias, hipStream_t stream) {
    if constexpr (HD <= 128) {
        // a dense per-row bias whose geometry allows whole 16-byte granules (p.bias_vec == 3, host.cpp) on a grid that fills the chip: the 8-wave,
        // 256-row shape with the bias tile staged by LDS-DMA (fa2_fwd_kernel.hip.h, BIAS = 2)
        if (bias && p.bias_vec == 3) return launch_shape<HD, CAUSAL, 8, 2>(p, stream);
    }
    if (bias) return launch_shape<HD, CAUSAL, 4, 1>(p, stream);
    if constexpr (HD > 256) {
        return launch_shape<HD, CAUSAL, 4>(p, stream);
    } else {
        return rows == 128 ? launch_shape<HD, CAUSAL, 4>(p, stream) : launch_shape<HD, CAUSAL, 8>(p, stream);
    }
}

template <int HD>
int launch_hd(const fa2::FwdParams& p, bool causal, int rows, bool bias, hipStream_t stream) {
    return causal ? launch_t<HD, true>(p, rows, bias, stream) : launch_t<HD, false>(p, rows, bias, stream);
}

// KV sweeps of at most two tiles (fa2_fwd_short.hip.h): 128-row workgroups; one instantiation per count of 32-key blocks that hold a key
template <int HD, int NB>
int launch_short_nb(const fa2::FwdParams& p, hipStream_t stream) {
    constexpr auto kern = fa2::fwd_short_kernel<HD, kBF16, NB>;
    constexpr int lds = fa2::short_lds_bytes<HD>((NB + 1) / 2);
    if (int rc = fa2::set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.nbh * p.nqblk)), dim3(256), lds, stream, p);
    return (int)hipGetLastError();
}

template <int HD>
int launch_short(const fa2::FwdParams& p0, hipStream_t stream) {
    fa2::FwdParams p = p0;
    p.nqblk = (p.Nq + fa2::kShortRows - 1) / fa2::kShortRows;
    if ((int64_t)p.nbh * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;
    p.nsplit = 0;
    switch ((p.Nkv + 31) / 32) {
        case 1: return launch_short_nb<HD, 1>(p, stream);
        case 2: return launch_short_nb<HD, 2>(p, stream);
        case 3: return launch_short_nb<HD, 3>(p, stream);
        case 4: return launch_short_nb<HD, 4>(p, stream);
        default: return FA2_ERR_BAD_SHAPE;
    }
}

template <int HD>
int launch_combine(const fa2::FwdParams& p, hipStream_t stream) {
    const int64_t threads = (int64_t)p.split_items * fa2::kSplitRows * (HD / 8);
    hipLaunchKernelGGL((fa2::fwd_combine_kernel<HD, kBF16>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

namespace fa2 {

// merge of the KV-split parts (fwd_combine_kernel); p as handed to the launch that produced them (nqblk in 256-row blocks)
#if FA2_TU_BF16
int launch_fwd_combine_bf16(int HD, const FwdParams& p, hipStream_t stream) {
#else
int launch_fwd_combine_f16(int HD, const FwdParams& p, hipStream_t stream) {
#endif
    return HD == 64 ? launch_combine<64>(p, stream) : HD == 128 ? launch_combine<128>(p, stream) : FA2_ERR_HEAD_DIM;
}

#if FA2_TU_BF16
int launch_fwd_short_bf16(int HD, const FwdParams& p, hipStream_t stream) {
#else
int launch_fwd_short_f16(int HD, const FwdParams& p, hipStream_t stream) {
#endif
    return HD == 64 ? launch_short<64>(p, stream) : HD == 128 ? launch_short<128>(p, stream) : FA2_ERR_HEAD_DIM;
}

#if FA2_TU_BF16
int launch_fwd_hip_bf16(int HD, const FwdParams& p, bool causal, int rows, bool bias, hipStream_t stream) {
#else
int launch_fwd_hip_f16(int HD, const FwdParams& p, bool causal, int rows, bool bias, hipStream_t stream) {
#endif
    if (FA2_TRIM && !bias && (p.D < HD || (HD == 256 && causal && FA2_TRIM256_MODE >= 2))) {     // a trimmed kernel, where one exists
#if FA2_TU_BF16
        const int rc = launch_fwd_hip_trim_bf16(HD, p, causal, rows, stream);
#else
        const int rc = launch_fwd_hip_trim_f16(HD, p, causal, rows, stream);
#endif
        if (rc >= 0) return rc;
    }
    switch (HD) {
        case 64: return launch_hd<64>(p, causal, rows, bias, stream);
        case 128: return launch_hd<128>(p, causal, rows, bias, stream);
        case 256: return launch_hd<256>(p, causal, rows, bias, stream);
        case 512: return launch_hd<512>(p, causal, rows, bias, stream);
        default: return FA2_ERR_HEAD_DIM;
    }
}

}  // namespace fa2

#endif  // FA2_TU_TRIM
