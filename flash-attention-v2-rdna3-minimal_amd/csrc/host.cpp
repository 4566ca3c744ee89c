// host.cpp — C-ABI shim of the gfx950 FlashAttention-2 forward path (compiled with hipcc -x hip).
//
// Counterpart of the reference's host layer:
//   rocwmma_fattn/host.cpp:30-45        dtype switch forward()        -> fa2_fwd
//   rocwmma_fattn/kernel_fp16.cu:744-876 forward_fp16 host launcher   -> fa2_fwd_f16
//   rocwmma_fattn/kernel_bf16.cu:802-941 forward_bf16 host launcher   -> fa2_fwd_bf16
//   rocwmma_fattn/host.cpp:47-58         dtype switch backward()       -> fa2_bwd
//   rocwmma_fattn/kernel_fp16.cu:878-1028 backward_fp16 host launcher  -> fa2_bwd_f16 (bf16 twin -> fa2_bwd_bf16)
// Unlike the reference this layer owns no tensors and allocates nothing: padding, output allocation
// and the 6-tensor return contract live in the Python operator (rocwmma_fattn/FlashAttn.py), the
// launch is asynchronous on the caller's stream, and failures are returned, not printf'ed
// (reference: kernel_fp16.cu:854-863).
#include "fa2_fwd_kernel.hip.h"
#include "fa2_fwd_kernel16.hip.h"
#include "fa2_fwd_d128.hip.h"
#include "fa2_bwd_kernel.hip.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "fa2_gfx950.h"

namespace {

constexpr int kHeadDims[] = {64, 128, 256, 512};
constexpr int kMaxBwdHeadDim = 256;   // the backward kernels stop here
constexpr int kNumHeadDims = sizeof(kHeadDims) / sizeof(kHeadDims[0]);

// Workgroup shape: NW waves x QB 32-row Q blocks per wave (NW * QB * 32 = 256 Q rows).
#ifndef FA2_NW
#define FA2_NW 8
#endif
#ifndef FA2_QB
#define FA2_QB (8 / FA2_NW)
#endif
constexpr int kNW = FA2_NW, kQB = FA2_QB;
constexpr int kFwdRows = kNW * kQB * 32;   // Q rows per forward workgroup

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

// Rows per forward workgroup: 256 (8 waves; D = 128: the hand-scheduled 4-wave kernel) or 128 (4 waves, HIP kernel).
// FA2_FWD_ROWS=128|256 in the environment overrides pick_rows() (A/B runs, tools/rows_probe.py).
int forced_rows() {
    static const int v = [] {
        const char* e = std::getenv("FA2_FWD_ROWS");
        return e ? std::atoi(e) : 0;
    }();
    return v;
}

// A grid of 256-row workgroups that covers well under half of the 256 CUs leaves the chip idle: 128-row workgroups double the
// number of busy CUs at the price of staging every K/V tile for half as many rows.  Measured (tools/rows_probe.py, MI355X):
// 64 workgroups -> 128 of 128 rows: B1 H16 N1024 D128 25.1 -> 20.5 us, B1 H8 N2048 D128 44.1 -> 35.3, B2 H8 N1024 D80 24.0 -> 19.9;
// at 160 workgroups (SDXL 32x32 self-attention) and above the big shape wins (19.9 vs 23.9 us), 64-row workgroups never do.
// Between one and one and a half rounds of 256-row workgroups at head dims <= 64 (non-causal), 128-row workgroups all the way — two per CU since
// they fit the 256-register budget (fwd_min_waves_per_simd) — beat both the plain grid and the tail split (tools/rows_probe.py, r05m: SDXL 64x64
// self-attention B2 H10 N4096 D64, 320 workgroups: 130.8 plain / 121.2 tail split / 117.4 us; B1 H24 N3072 D64, 288: 97.0 / 90.9 / 84.5 us); at exactly one
// round (B2 H8 N4096: 66 vs 73 us) and below (160 workgroups: 20.0 vs 22.6 us) the 8-wave shape stays ahead.
bool short_second_round(const fa2::FwdParams& p, bool causal) {
    if (causal || p.D > 64) return false;
    const int64_t w = (int64_t)p.nbh * ((p.Nq + 255) / 256);
    return w > 256 && w <= 384;
}

int pick_rows(const fa2::FwdParams& p, bool causal = true) {
    const int f = forced_rows();
    if (f == 128 || f == 256) return f;
    if (p.rows_hint == 128 || p.rows_hint == 256) return p.rows_hint;
    if (short_second_round(p, causal)) return 128;
    return (int64_t)p.nbh * ((p.Nq + 255) / 256) <= 96 ? 128 : 256;
}

// Tail split (non-causal): B*H*ceil(Nq/256) equal workgroups on 256 CUs take ceil(x/256) rounds however empty the last one
// is — SDXL's 64x64 self-attention (B2 H10 N4096) is 320 workgroups, two rounds for 1.25 rounds of work.  When the last
// round would be at most half full, the heads that make it up run in a second launch as 128-row workgroups (twice as
// many, ~0.85x as long each: they stream the same K/V for half the rows).  Measured at D = 64 (tools/rows_probe.py): B2 H10
// N4096 130.8 -> 123.9 us, B1 H24 N3072 97.0 -> 92.5 us; one SDXL UNet step's attention 3.11 -> 3.02 ms.  Returns the number of
// heads of the main launch (= nbh: no split).  FA2_TAIL_SPLIT=0 in the environment disables it.
int tail_split_heads(const fa2::FwdParams& p, bool causal) {
    static const bool on = [] { const char* e = std::getenv("FA2_TAIL_SPLIT"); return !(e && e[0] == '0'); }();
    if (!on || causal || forced_rows() != 0) return p.nbh;
    if (short_second_round(p, causal)) return p.nbh;            // 128-row workgroups for the whole grid instead (pick_rows)
    // a second launch costs ~5 us: only sweeps of at least 16 KV tiles (a workgroup then runs >= ~15 us) can win it back.  SDXL's
    // cross-attention (B2 H10 N4096 x Nkv 77, two tiles) measured 13.8 us split into two launches against 10.1 us for torch SDPA.
    if (p.Nkv < 16 * fa2::kKvTile) return p.nbh;
    const int64_t cus = 256, nq = (p.Nq + 255) / 256, w = (int64_t)p.nbh * nq;
    if (w <= cus || nq > cus) return p.nbh;
    const int64_t main_heads = (w / cus) * cus / nq;           // whole heads that fit the full rounds
    const int64_t tail_w = (p.nbh - main_heads) * nq;           // 256-row workgroups of the remaining heads
    if (main_heads <= 0 || tail_w <= 0 || tail_w > cus / 2) return p.nbh;
    return (int)main_heads;
}

template <int HD, bool BF16, bool CAUSAL, int NW, int QB, bool BIAS = false>
int launch_shape(const fa2::FwdParams& p0, hipStream_t stream) {
    constexpr int HDV = HD > 128 ? 128 : HD;
    constexpr int lds_kv = 2 * fa2::Geo<HD, NW>::TILEB + 2 * fa2::Geo<HDV, NW>::TILEB;
    constexpr int lds_epi = FA2_EPI_LDS && QB == 1 ? NW * 32 * (HDV * 2 + 16) : 0;     // epilogue image (reuses the K/V space)
    // bias kernels: + NW wave-private 32-row images of the "tile" bias form where they fit (not at D = 512: 160 KiB of K / V buffers)
    constexpr int lds_bias = BIAS && lds_kv + NW * 32 * 272 <= 160 * 1024 ? NW * 32 * 272 : 0;
    constexpr int lds = lds_kv + lds_bias > lds_epi ? lds_kv + lds_bias : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    fa2::FwdParams p = p0;
    p.nqblk = (p.Nq + NW * QB * 32 - 1) / (NW * QB * 32);
    if ((int64_t)p.nbh * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;
    const dim3 grid((unsigned)((int64_t)p.nbh * p.nqblk), HD / HDV);
    constexpr auto kern = fa2::fwd_kernel<HD, HDV, BF16, CAUSAL, NW, QB, BIAS>;
    if (int rc = set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
    return (int)hipGetLastError();
}

// Attention bias / boolean mask (fa2_fwd_bias): every head dim runs the generic HIP kernel as 4-wave, 128-row workgroups —
// one wave per SIMD, so the 32 bias registers per tile come out of the 512-register budget instead of spilling.
template <int HD, bool BF16>
int launch_bias(const fa2::FwdParams& p, bool causal, hipStream_t stream) {
    // (the 8-wave shape at D = 64 spills 62-67 VGPRs with the bias registers: not used)
    return causal ? launch_shape<HD, BF16, true, 4, 1, true>(p, stream) : launch_shape<HD, BF16, false, 4, 1, true>(p, stream);
}

// The 8-wave kernel on v_mfma_f32_16x16x32 (fa2_fwd_kernel16.hip.h), head dims 64 and 128, 256-row workgroups.
// FA2_MFMA16=1|0 in the environment (read once) overrides the build-time default.
#ifndef FA2_MFMA16
#define FA2_MFMA16 0
#endif
bool use_mfma16() {
    static const bool on = [] {
        const char* e = std::getenv("FA2_MFMA16");
        if (e && e[0] == '1') return true;
        if (e && e[0] == '0') return false;
        return FA2_MFMA16 != 0;
    }();
    return on;
}

template <int HD, bool BF16, bool CAUSAL>
int launch_shape16(const fa2::FwdParams& p0, hipStream_t stream) {
    constexpr int lds_kv = 4 * fa2::Geo<HD, 8>::TILEB;
    constexpr int lds_epi = 8 * 32 * (HD * 2 + 16);
    constexpr int lds = lds_kv > lds_epi ? lds_kv : lds_epi;
    fa2::FwdParams p = p0;
    p.nqblk = (p.Nq + 255) / 256;
    if ((int64_t)p.nbh * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;
    constexpr auto kern = fa2::fwd_kernel16<HD, BF16, CAUSAL>;
    if (int rc = set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.nbh * p.nqblk)), dim3(512), lds, stream, p);
    return (int)hipGetLastError();
}

template <int HD, bool BF16, bool CAUSAL>
int launch_t(const fa2::FwdParams& p, hipStream_t stream) {
    // D = 256 runs as two, D = 512 as four 128-column slabs of O per Q block (grid.y), recomputing QK^T per slab.
    // D = 512 (the reference's D > 384 path, FlashAttn.py:65-67; the SD VAE attention block): 4-wave workgroups of 128 Q
    // rows, one wave per SIMD — the 128 registers of Q fragments per wave need the 512-register budget — and all 160 KiB
    // of LDS (two 64 KiB K tiles + two 16 KiB V tiles).  A correct path for a rare shape, not a tuned one.
    if constexpr (HD > 256) {
        return launch_shape<HD, BF16, CAUSAL, 4, 1>(p, stream);
    } else if constexpr (kNW == 8 && kQB == 1) {
        // (512-row workgroups <8, 2> for short KV sweeps over long Q — SDXL cross-attention, 320 workgroups = 1.25 rounds — were
        //  measured: 105 spilled VGPRs at D = 64, 20.1 us against 13.7 us: not kept)
        if (pick_rows(p, CAUSAL) == 128) return launch_shape<HD, BF16, CAUSAL, 4, 1>(p, stream);
        if constexpr (HD <= 128) {
            if (use_mfma16()) return launch_shape16<HD, BF16, CAUSAL>(p, stream);
        }
        return launch_shape<HD, BF16, CAUSAL, kNW, kQB>(p, stream);
    } else {
        return launch_shape<HD, BF16, CAUSAL, kNW, kQB>(p, stream);
    }
}

// Head dim exactly 128 with a positive scale runs the hand-scheduled 4-wave kernel (fa2_fwd_d128.hip.h); FA2_FWD_D128=hip
// in the environment selects the compiler-scheduled 8-wave kernel instead (A/B measurements, tools/kbench.py).
#ifndef FA2_D128_ASM
#define FA2_D128_ASM 1      // build-time default of the switch below
#endif
bool use_d128_asm() {
    static const bool on = [] {
        const char* e = std::getenv("FA2_FWD_D128");
        if (e && e[0] == 'h') return false;
        if (e && e[0] == 'a') return true;
        return FA2_D128_ASM != 0;
    }();
    return on;
}

#ifndef FA2_D128_FOLD
#define FA2_D128_FOLD 0     // build-time default of the switch below
#endif
// Folded scale (opt-in: FA2_D128_FOLD=1 in the environment, read once): Q * scale*log2e is rounded once to the I/O dtype — the
// reference oracle's contract, pure_torch_ver.py:61 — and the running reference enters the first QK^T k-step as its C operand,
// so the 64 v_fma per tile disappear (no extra MFMAs).  Only when scale*log2e <= 1; fa2_fwd_prescales_q() reports the choice.
// Measured on MI355X: +2 % throughput at config 2 (the chip is power-limited: the saved issue cycles mostly come back as stalls),
// for 16-bit-rounded logits (LSE error 2e-4 fp16 / 6e-3 bf16 instead of 2e-6): off by default.
bool use_d128_fold() {
    static const bool on = [] {
        const char* e = std::getenv("FA2_D128_FOLD");
        if (e && e[0] == '1') return true;
        if (e && e[0] == '0') return false;
        return FA2_D128_FOLD != 0;
    }();
    return on;
}
bool d128_eligible(int D, float scale) { return D == 128 && scale > 0.f && use_d128_asm(); }
// (the asm block addresses a head's Q rows with 32-bit byte offsets)
bool d128_q_span_ok(const fa2::FwdParams& p) { return ((int64_t)(p.Nq - 1) * p.qs[2] + 128) * 2 < ((int64_t)1 << 32); }
bool d128_folds(float c) { return use_d128_fold() && c <= 1.0f; }

// Persistent workgroups of the d128 kernel (non-causal launches): at most one workgroup per CU, each working through a
// strided list of (head, q block) items and fetching the next item's first tiles while the current one finishes
// (fa2_fwd_d128.hip.h).  FA2_D128_PERSIST=0 in the environment launches one workgroup per item instead (A/B measurements).
#ifndef FA2_D128_PERSIST
#define FA2_D128_PERSIST 1     // build-time default of the switch
#endif
int d128_persistent_grid() {
    static const int grid = [] {
        const char* e = std::getenv("FA2_D128_PERSIST");
        if (e ? e[0] == '0' : FA2_D128_PERSIST == 0) return 0;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cus & ~7;          // a multiple of 8: an item stays on the XCD its head is mapped to
    }();
    return grid;
}

template <bool BF16, bool CAUSAL, bool FOLD>
int launch_d128_t(const fa2::FwdParams& p, hipStream_t stream) {
    static_assert(kFwdRows == 256, "the d128 kernel covers 256 Q rows per workgroup, like the default shape");
    constexpr auto kern = fa2::fwd_d128_kernel<BF16, CAUSAL, FOLD>;
    if (int rc = set_lds<kern>(fa2::kD128LdsBytes)) return rc;
    int64_t grid = (int64_t)p.nbh * p.nqblk;
    const int pg = d128_persistent_grid();
    if (!CAUSAL && pg > 0 && grid > pg) grid = pg;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), fa2::kD128LdsBytes, stream, p);
    return (int)hipGetLastError();
}

template <bool BF16, bool CAUSAL>
int launch_d128(const fa2::FwdParams& p, hipStream_t stream) {
    return d128_folds(p.c) ? launch_d128_t<BF16, CAUSAL, true>(p, stream) : launch_d128_t<BF16, CAUSAL, false>(p, stream);
}

template <int HD, bool BF16>
int launch_range(const fa2::FwdParams& p, bool causal, hipStream_t stream);

template <int HD, bool BF16>
int launch(const fa2::FwdParams& p0, bool causal, hipStream_t stream) {
    if constexpr (HD <= 64) {     // (measured at D = 128, B1 H24 N4096: 188 -> 194 us — the 128-row shape is too slow there; D = 64: see below)
        const int main_heads = tail_split_heads(p0, causal);
        if (main_heads < p0.nbh) {
            fa2::FwdParams p = p0;
            p.nbh = main_heads;
            if (int rc = launch_range<HD, BF16>(p, causal, stream)) return rc;
            p.bh0 = p0.bh0 + main_heads;
            p.nbh = p0.nbh - main_heads;
            p.rows_hint = 128;
            return launch_range<HD, BF16>(p, causal, stream);
        }
    }
    return launch_range<HD, BF16>(p0, causal, stream);
}

template <int HD, bool BF16>
int launch_range(const fa2::FwdParams& p, bool causal, hipStream_t stream) {
    if constexpr (HD == 128) {
        if (d128_eligible(p.D, p.negate_q ? -1.f : 1.f) && pick_rows(p) == 256 && d128_q_span_ok(p))
            return causal ? launch_d128<BF16, true>(p, stream) : launch_d128<BF16, false>(p, stream);
    }
    return causal ? launch_t<HD, BF16, true>(p, stream) : launch_t<HD, BF16, false>(p, stream);
}

#ifndef FA2_BWD_FUSE_MAX_HD          // head dims up to this run dK and dV as one fused pass
#define FA2_BWD_FUSE_MAX_HD 64
#endif

// Backward at head dims 65..128: dK and dV as ONE sweep of wave pairs (7 GEMM-equivalents for the whole backward) instead of two
// separate sweeps (8).  FA2_BWD_PAIR=0|1 in the environment (read once) overrides the build-time default (A/B measurements).
#ifndef FA2_BWD_PAIR
#define FA2_BWD_PAIR 1
#endif
bool use_bwd_pair() {
    static const bool on = [] {
        const char* e = std::getenv("FA2_BWD_PAIR");
        if (e && e[0] == '1') return true;
        if (e && e[0] == '0') return false;
        return FA2_BWD_PAIR != 0;
    }();
    return on;
}

template <int HD, bool BF16, bool CAUSAL>
int launch_bwd_pair(const fa2::BwdParams& p, hipStream_t stream) {
    constexpr int lds = 2 * (4 * fa2::Geo<HD, 8>::TILEB + 512) + 4 * 4096;
    constexpr auto kern = fa2::bwd_dkv_pair_kernel<HD, BF16, CAUSAL>;
    if (int rc = set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(512), lds, stream, p);
    return (int)hipGetLastError();
}

template <int HD, bool BF16, bool CAUSAL>
int launch_bwd_t(fa2::BwdParams p, hipStream_t stream) {
    constexpr int NW = HD > 128 ? 4 : 8;          // D = 256: one wave per SIMD (512 registers), single LDS stage
    constexpr int kRows = NW * 32, kStages = NW == 8 ? 2 : 1;
    constexpr int TILEB = fa2::Geo<HD, NW>::TILEB;
    int rc;
    // dQ: one workgroup per kRows Q rows; also writes D_i = rowsum(dO * O) to the delta workspace for the dK pass.
    // Grids that would cover at most half of the CUs with 256-row workgroups (B*H*ceil(Nq/256) <= 128: SD-size training shapes) run
    // as 128-row, 4-wave workgroups instead — twice as many, one wave per SIMD each.  FA2_BWD_DQ_ROWS=256|128 in the environment pins the shape.
    bool dq_small = false;
    if constexpr (NW == 8) {
        static const int forced = [] { const char* e = std::getenv("FA2_BWD_DQ_ROWS"); return e ? std::atoi(e) : 0; }();
        dq_small = forced == 128 || (forced != 256 && (int64_t)p.B * p.H * ((p.Nq + 255) / 256) <= 128);
    }
    if (dq_small) {
        if constexpr (NW == 8) {
            constexpr int lds = 2 * 3 * fa2::Geo<HD, 4>::TILEB;
            constexpr auto kern = fa2::bwd_dq_kernel<HD, BF16, CAUSAL, 4>;
            if ((rc = set_lds<kern>(lds))) return rc;
            p.nblk = (p.Nq + 127) / 128;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(256), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
    } else {
        constexpr int lds = kStages * 3 * TILEB;
        constexpr auto kern = fa2::bwd_dq_kernel<HD, BF16, CAUSAL, NW>;
        if ((rc = set_lds<kern>(lds))) return rc;
        p.nblk = (p.Nq + kRows - 1) / kRows;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
    }
    if constexpr (HD == 128 && NW == 8) {
        // D in 65..128: dK and dV in one sweep by wave pairs (bwd_dkv_pair_kernel): 128 KV rows per workgroup, S and P formed once
        if (use_bwd_pair()) {
            p.nblk = (p.Nkv + 127) / 128;
            if ((int64_t)p.B * p.H * p.nblk > 0x7fffffffLL) return FA2_ERR_GRID;
            return launch_bwd_pair<HD, BF16, CAUSAL>(p, stream);
        }
    }
    p.nblk = (p.Nkv + kRows - 1) / kRows;   // dV, dK: one workgroup per kRows KV rows
    if constexpr (HD <= FA2_BWD_FUSE_MAX_HD && NW == 8) {
        // D = 64: both accumulators fit, one sweep forms S and P once for dK and dV
        constexpr int lds = kStages * (4 * TILEB + 512);
        constexpr auto kern = fa2::bwd_dkv_kernel<HD, BF16, CAUSAL, true, NW, true>;
        if ((rc = set_lds<kern>(lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
        return (int)hipGetLastError();
    } else {
        {
            constexpr int lds = kStages * (2 * TILEB + 512);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, BF16, CAUSAL, false, NW>;
            if ((rc = set_lds<kern>(lds))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        {
            constexpr int lds = kStages * (3 * TILEB + 512);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, BF16, CAUSAL, true, NW>;
            if ((rc = set_lds<kern>(lds))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
    }
    return 0;
}

template <int HD, bool BF16>
int launch_bwd(const fa2::BwdParams& p, bool causal, hipStream_t stream) {
    return causal ? launch_bwd_t<HD, BF16, true>(p, stream) : launch_bwd_t<HD, BF16, false>(p, stream);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool strides_ok(const int64_t* s) { return s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0 && s[2] > 0; }

}  // namespace

extern "C" {

int fa2_supported_head_dims(int* dims, int cap) {
    for (int i = 0; i < kNumHeadDims && i < cap; ++i)
        if (dims) dims[i] = kHeadDims[i];
    return kNumHeadDims;
}

int fa2_padded_head_dim(int D) {
    if (D < 1) return -1;
    for (int i = 0; i < kNumHeadDims; ++i)
        if (D <= kHeadDims[i]) return kHeadDims[i];
    return -1;
}

int fa2_tile_rows(int D, int* q_rows_per_block, int* kv_rows_per_tile) {
    if (fa2_padded_head_dim(D) != D) return FA2_ERR_HEAD_DIM;
    if (q_rows_per_block) *q_rows_per_block = D > 256 ? 128 : kFwdRows;
    if (kv_rows_per_tile) *kv_rows_per_tile = fa2::kKvTile;
    return FA2_OK;
}

int fa2_fwd_prescales_q(int D, float scale) {
    if (fa2_padded_head_dim(D) < 0) return -1;
    const float c = std::fabs(scale) * 1.4426950408889634f;
    return d128_eligible(D, scale) && d128_folds(c < 1e-30f ? 1e-30f : c) ? 1 : 0;
}

const char* fa2_error_string(int code) {
    switch (code) {
        case FA2_OK: return "ok";
        case FA2_ERR_NULL_POINTER: return "fa2: null pointer argument";
        case FA2_ERR_BAD_SHAPE: return "fa2: B, H, Nq, Nkv, D must be >= 1 and one head's K/V must span < 4 GiB";
        case FA2_ERR_HEAD_DIM: return "fa2: head dim not supported (pad D to fa2_padded_head_dim(D))";
        case FA2_ERR_ALIGNMENT: return "fa2: pointers must be 16-byte aligned, strides multiples of 8 elements, last dim contiguous";
        case FA2_ERR_DTYPE: return "fa2: dtype must be FA2_DTYPE_F16 or FA2_DTYPE_BF16";
        case FA2_ERR_SCALE: return "fa2: scale must be finite";
        case FA2_ERR_GRID: return "fa2: B*H*ceil(Nq/256) exceeds the grid limit";
        case FA2_ERR_BIAS: return "fa2: bias_kind must be FA2_BIAS_{NONE,IO_DTYPE,F32,BOOL} and bias strides >= 0";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "fa2: unknown error code";
}

const char* fa2_version(void) { return "fa2_gfx950 0.6 (D=128: hand-scheduled 4-wave 256x64 asm body; other head dims up to 512: 8-wave HIP kernels; mfma32x32x16, lds-dma; fwd+bwd; attention bias / mask)"; }

static int fwd_impl(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                    int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                    const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                    float scale, int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream) {
    if (!q || !k || !v || !o || !lse || !q_strides || !k_strides || !v_strides || !o_strides || !lse_strides)
        return FA2_ERR_NULL_POINTER;
    if (bias_kind != FA2_BIAS_NONE) {
        if (bias_kind != FA2_BIAS_IO_DTYPE && bias_kind != FA2_BIAS_F32 && bias_kind != FA2_BIAS_BOOL) return FA2_ERR_BIAS;
        if (!bias || !bias_strides) return FA2_ERR_NULL_POINTER;
        if (bias_strides[0] < 0 || bias_strides[1] < 0 || bias_strides[2] < 0) return FA2_ERR_BIAS;
        const uintptr_t esize = bias_kind == FA2_BIAS_F32 ? 4 : bias_kind == FA2_BIAS_IO_DTYPE ? 2 : 1;
        if (reinterpret_cast<uintptr_t>(bias) % esize) return FA2_ERR_ALIGNMENT;
    }
    if (dtype != FA2_DTYPE_F16 && dtype != FA2_DTYPE_BF16) return FA2_ERR_DTYPE;
    if (B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1) return FA2_ERR_BAD_SHAPE;
    const int HD = fa2_padded_head_dim(D);       // kernel head dim; columns [D, HD) are masked in-kernel
    if (HD < 0 || (D & 7)) return FA2_ERR_HEAD_DIM;
    if (!std::isfinite(scale)) return FA2_ERR_SCALE;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !strides_ok(q_strides) ||
        !strides_ok(k_strides) || !strides_ok(v_strides) || !strides_ok(o_strides))
        return FA2_ERR_ALIGNMENT;
    const int64_t k_bytes = ((int64_t)(Nkv - 1) * k_strides[2] + D) * 2;
    const int64_t v_bytes = ((int64_t)(Nkv - 1) * v_strides[2] + D) * 2;
    // fa2::kOobOffset relies on every in-range offset, plus one tile of rows past the end, staying below 2 GiB
    if (k_bytes + 64 * k_strides[2] * 2 > 0x7fffffffLL || v_bytes + 64 * v_strides[2] * 2 > 0x7fffffffLL) return FA2_ERR_BAD_SHAPE;

    fa2::FwdParams p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.Nq = Nq; p.Nkv = Nkv; p.D = D;
    for (int i = 0; i < 3; ++i) {
        p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; p.os[i] = o_strides[i];
    }
    p.ls[0] = lse_strides[0]; p.ls[1] = lse_strides[1];
    p.c = std::fabs(scale) * 1.4426950408889634f;  // fold log2(e): reference kernel_fp16.cu:827
    // scale == 0 is the uniform softmax (O = mean of V, as the reference's arithmetic gives).  With c == 0 the first
    // tile's (max - (-inf)) * c would be NaN; a factor too small to move any f32 score off zero gives the same result.
    if (p.c < 1e-30f) p.c = 1e-30f;
    p.negate_q = scale < 0.f;
    p.nqblk = (Nq + kFwdRows - 1) / kFwdRows;
    p.bh0 = 0;
    p.nbh = B * H;
    p.rows_hint = 0;
    p.k_bytes = (uint32_t)k_bytes;
    p.v_bytes = (uint32_t)v_bytes;
    p.bias = bias;
    p.bias_kind = bias_kind;
    for (int i = 0; i < 3; ++i) p.bs[i] = bias_kind != FA2_BIAS_NONE ? bias_strides[i] : 0;
    p.bias_vec = 0;
    if (bias_kind != FA2_BIAS_NONE) {      // groups of four consecutive kv can be fetched with one aligned load
        const uintptr_t esize = bias_kind == FA2_BIAS_F32 ? 4 : bias_kind == FA2_BIAS_IO_DTYPE ? 2 : 1;
        p.bias_vec = Nkv % 4 == 0 && reinterpret_cast<uintptr_t>(bias) % (4 * esize) == 0 && p.bs[0] % 4 == 0 && p.bs[1] % 4 == 0 &&
                     p.bs[2] % 4 == 0;
        // 2: a per-row bias whose geometry allows whole 16-byte granules: coalesced tile loads through LDS (a row-broadcast bias —
        // bs[2] == 0, e.g. a key-padding mask — is one cache line for the whole wave already)
        const int64_t gran = 16 / (int64_t)esize;
        if (p.bias_vec && HD <= 256 && Nkv % gran == 0 && reinterpret_cast<uintptr_t>(bias) % 16 == 0 && p.bs[0] % gran == 0 &&
            p.bs[1] % gran == 0 && p.bs[2] % gran == 0 && p.bs[2] != 0)
            p.bias_vec = 2;
    }
    if ((int64_t)B * H * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;

    hipStream_t stream = (hipStream_t)hip_stream;
    const bool bf16 = dtype == FA2_DTYPE_BF16;
    if (bias_kind != FA2_BIAS_NONE) {
        if ((int64_t)B * H * ((Nq + 127) / 128) > 0x7fffffffLL) return FA2_ERR_GRID;
        switch (HD) {
            case 64: return bf16 ? launch_bias<64, true>(p, causal != 0, stream) : launch_bias<64, false>(p, causal != 0, stream);
            case 128: return bf16 ? launch_bias<128, true>(p, causal != 0, stream) : launch_bias<128, false>(p, causal != 0, stream);
            case 256: return bf16 ? launch_bias<256, true>(p, causal != 0, stream) : launch_bias<256, false>(p, causal != 0, stream);
            case 512: return bf16 ? launch_bias<512, true>(p, causal != 0, stream) : launch_bias<512, false>(p, causal != 0, stream);
            default: return FA2_ERR_HEAD_DIM;
        }
    }
    switch (HD) {
        case 64: return bf16 ? launch<64, true>(p, causal != 0, stream) : launch<64, false>(p, causal != 0, stream);
        case 128: return bf16 ? launch<128, true>(p, causal != 0, stream) : launch<128, false>(p, causal != 0, stream);
        case 256: return bf16 ? launch<256, true>(p, causal != 0, stream) : launch<256, false>(p, causal != 0, stream);
        case 512: return bf16 ? launch<512, true>(p, causal != 0, stream) : launch<512, false>(p, causal != 0, stream);
        default: return FA2_ERR_HEAD_DIM;
    }
}

int fa2_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
            int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
            const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
            float scale, int causal, void* hip_stream) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides, lse_strides,
                    scale, causal, nullptr, FA2_BIAS_NONE, nullptr, hip_stream);
}

int fa2_fwd_bias(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                 int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                 const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                 float scale, int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides, lse_strides,
                    scale, causal, bias, bias_kind, bias_strides, hip_stream);
}

int fa2_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
            void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
            const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
            const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
            const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
            int causal, void* hip_stream) {
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !delta_ws || !q_strides || !k_strides ||
        !v_strides || !o_strides || !do_strides || !dq_strides || !dk_strides || !dv_strides || !lse_strides)
        return FA2_ERR_NULL_POINTER;
    if (dtype != FA2_DTYPE_F16 && dtype != FA2_DTYPE_BF16) return FA2_ERR_DTYPE;
    if (B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1) return FA2_ERR_BAD_SHAPE;
    const int HD = fa2_padded_head_dim(D);              // columns [D, HD) are masked in-kernel
    if (HD < 0 || HD > kMaxBwdHeadDim || (D & 7)) return FA2_ERR_HEAD_DIM;
    if (!std::isfinite(scale)) return FA2_ERR_SCALE;
    const void* ptrs[] = {q, k, v, o, dout, dq, dk, dv};
    const int64_t* strides[] = {q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides};
    for (int i = 0; i < 8; ++i)
        if (!aligned16(ptrs[i]) || !strides_ok(strides[i])) return FA2_ERR_ALIGNMENT;
    const int64_t q_bytes = ((int64_t)(Nq - 1) * q_strides[2] + D) * 2, do_bytes = ((int64_t)(Nq - 1) * do_strides[2] + D) * 2;
    const int64_t k_bytes = ((int64_t)(Nkv - 1) * k_strides[2] + D) * 2, v_bytes = ((int64_t)(Nkv - 1) * v_strides[2] + D) * 2;
    const int64_t lim = 0x7fffffffLL;   // + one tile of rows past the end: masked lanes add fa2::kOobOffset to such an offset
    if (q_bytes + 64 * q_strides[2] * 2 > lim || do_bytes + 64 * do_strides[2] * 2 > lim || k_bytes + 64 * k_strides[2] * 2 > lim ||
        v_bytes + 64 * v_strides[2] * 2 > lim)
        return FA2_ERR_BAD_SHAPE;
    const int64_t blocks = (int64_t)B * H * (((Nq > Nkv ? Nq : Nkv) + fa2::kQBlock - 1) / fa2::kQBlock);
    if (blocks > 0x7fffffffLL || (int64_t)B * H * Nq / 16 > 0x7fffffffLL) return FA2_ERR_GRID;

    fa2::BwdParams p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.dout = dout; p.lse = lse; p.delta = delta_ws; p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.H = H; p.Nq = Nq; p.Nkv = Nkv; p.D = D;
    for (int i = 0; i < 3; ++i) {
        p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; p.os[i] = o_strides[i];
        p.dos[i] = do_strides[i]; p.dqs[i] = dq_strides[i]; p.dks[i] = dk_strides[i]; p.dvs[i] = dv_strides[i];
    }
    p.ls[0] = lse_strides[0]; p.ls[1] = lse_strides[1];
    p.scale = scale;
    p.c = scale * 1.4426950408889634f;
    p.nblk = 0;
    p.q_bytes = (uint32_t)q_bytes; p.k_bytes = (uint32_t)k_bytes; p.v_bytes = (uint32_t)v_bytes;
    p.do_bytes = (uint32_t)do_bytes; p.l_bytes = (uint32_t)Nq * 4u;
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool bf16 = dtype == FA2_DTYPE_BF16;
    switch (HD) {
        case 64: return bf16 ? launch_bwd<64, true>(p, causal != 0, stream) : launch_bwd<64, false>(p, causal != 0, stream);
        case 128: return bf16 ? launch_bwd<128, true>(p, causal != 0, stream) : launch_bwd<128, false>(p, causal != 0, stream);
        case 256: return bf16 ? launch_bwd<256, true>(p, causal != 0, stream) : launch_bwd<256, false>(p, causal != 0, stream);
        default: return FA2_ERR_HEAD_DIM;
    }
}

#define FA2_BWD_ARGS                                                                                                    \
    const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse, void *dq, void *dk, \
        void *dv, float *delta_ws, int B, int H, int Nq, int Nkv, int D, const int64_t q_strides[3],                    \
        const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],                             \
        const int64_t do_strides[3], const int64_t dq_strides[3], const int64_t dk_strides[3],                          \
        const int64_t dv_strides[3], const int64_t lse_strides[2], float scale, int causal, void *hip_stream
#define FA2_BWD_PASS                                                                                                  \
    q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides,        \
        do_strides, dq_strides, dk_strides, dv_strides, lse_strides, scale, causal, hip_stream

int fa2_bwd_f16(FA2_BWD_ARGS) { return fa2_bwd(FA2_DTYPE_F16, FA2_BWD_PASS); }
int fa2_bwd_bf16(FA2_BWD_ARGS) { return fa2_bwd(FA2_DTYPE_BF16, FA2_BWD_PASS); }

int fa2_fwd_f16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq,
                int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                float scale, int causal, void* hip_stream) {
    return fa2_fwd(FA2_DTYPE_F16, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides,
                   o_strides, lse_strides, scale, causal, hip_stream);
}

int fa2_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq,
                 int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                 const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                 float scale, int causal, void* hip_stream) {
    return fa2_fwd(FA2_DTYPE_BF16, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides,
                   o_strides, lse_strides, scale, causal, hip_stream);
}

}  // extern "C"
