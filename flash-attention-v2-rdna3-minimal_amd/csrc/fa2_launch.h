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
    std::atomic<int> asm_mask{1987};     // FA2_ASM: bit 0 = hand-scheduled forward bodies, bit 1 = hand-scheduled backward bodies, bits 6 / 7 / 8 = the head-dim-128 forward / dQ-pass / dK-dV-pass
                                       // bodies built on v_mfma_f32_16x16x32 (round 5; off: the 32x32x16 bodies everywhere)
    std::atomic<int> persist{1};       // FA2_PERSIST: persistent workgroups of the hand-scheduled forward kernels
    std::atomic<int> bwd_parts{3};     // profiling only: bit 0 = run the dQ pass, bit 1 = run the dK / dV pass of fa2_bwd
    std::atomic<int> split{1};         // FA2_SPLIT: KV-split of the last, partly filled round of forward workgroups (fa2_fwd_ws)
    std::atomic<int> epoch{0};         // bumped by every fa2_set_option: callers that cache a plan (the compiled front end) key it on this
    std::atomic<int> fold{1};          // FA2_FOLD: 0 = the hand-scheduled forward bodies scale the f32 product, 1 = fp16 launches fold the scale into Q, 2 = bf16 too
                                       // (never for calls flagged FA2_FLAG_EXACT_SCALE: the forward of a call that will be differentiated)
    std::atomic<int> short_kv{1};      // FA2_SHORT: 1 = non-causal sweeps of at most two KV tiles (cross-attention) run the single-pass kernel (fa2_fwd_short.hip.h)
    std::atomic<int> kfold{0};         // FA2_KFOLD: 1 = the hand-scheduled dK / dV pass folds the scale into its K fragments where `fold` would fold a forward of that
                                       // dtype (round 4's default; off since round 5: profiles/r16_fold_evidence.txt, tests/test_backward_gpu.py large-logit case)
};
FA2_HIDDEN Options& options();
FA2_HIDDEN int device_cus();           // compute units of the current device (cached per device index)

// KV-split (forward, dQ pass) / Q-split (dK-dV pass) of a partly filled last round of 256-row workgroups: `items` equal workgroups on
// `cus` CUs take ceil(items / cus) rounds however empty the last one is.  With scratch memory from the caller the r = items % cus items of
// that round are swept by S workgroups each ("parts") over disjoint tile ranges, which leave f32 partial tiles that a small kernel merges.
// S in 2..kMaxSplit minimises rounds(r * S) / S + the scheme's fixed costs, in units of one whole item:
//   an item sweeps `nt` tiles at `us_per_tile` each; the merge kernel, its launch and a part's own prologue / epilogue cost `fixed_us`;
//   the partial tiles (`tile_bytes` per part) cross memory twice at ~4 TB/s through L2 / Infinity Cache.
// Parts get at least 8 tiles, the workspace never exceeds 64 MiB, and a split must win at least 7 % of the last round.
struct SplitPlan { int full_items = 0, split_items = 0, nsplit = 0; int64_t bytes = 0; };
constexpr int64_t kMaxSplitWsBytes = 64ll << 20;

// underfilled = true additionally splits EVERY item of a grid that covers at most half of the CUs (a KV-owned pass over a short KV — the dK / dV
// pass of SD cross-attention, Nkv = 77: B*H workgroups in all — sweeps thousands of Q rows on a tenth of the chip).
inline SplitPlan plan_tail_split(int64_t items, int nt, double us_per_tile, double fixed_us, int64_t tile_bytes, int64_t cus, bool underfilled = false) {
    SplitPlan none;
    if (underfilled && items >= 1 && items <= cus / 2) {
        int S = (int)(cus / items < kMaxSplit ? cus / items : kMaxSplit);
        while (S > 1 && (nt / S < 8 || items * S * tile_bytes > kMaxSplitWsBytes)) --S;
        const double t_item = nt * us_per_tile;
        if (S < 2 || t_item * (1.0 - 1.0 / S) < 2.0 * fixed_us) return none;
        SplitPlan pl;
        pl.full_items = 0;
        pl.split_items = (int)items;
        pl.nsplit = S;
        pl.bytes = items * S * tile_bytes;
        return pl;
    }
    if (items <= cus || items > 0x7fffffffLL || cus <= 0) return none;
    const int64_t r = items % cus;
    if (r == 0) return none;
    const double t_item = nt * us_per_tile, fixed = fixed_us / t_item;
    double best = 0.93;
    SplitPlan pl;
    for (int S = 2; S <= kMaxSplit; ++S) {
        if (nt / S < 8) break;
        const int64_t bytes = r * S * tile_bytes;
        if (bytes > kMaxSplitWsBytes) break;
        const double cost = (double)((r * S + cus - 1) / cus) / S + fixed + 2.0 * bytes / 4.0e6 / t_item;
        if (cost < best) { best = cost; pl.nsplit = S; pl.bytes = bytes; }
    }
    if (!pl.nsplit) return none;
    pl.full_items = (int)(items - r);
    pl.split_items = (int)r;
    return pl;
}

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
// KV sweeps of at most two tiles, non-causal, no bias, head dims <= 128 (fa2_fwd_short.hip.h; round 6): 128-row workgroups, one memory round trip
FA2_HIDDEN int launch_fwd_short_f16(int HD, const FwdParams& p, hipStream_t stream);
FA2_HIDDEN int launch_fwd_short_bf16(int HD, const FwdParams& p, hipStream_t stream);
// trimmed instantiations of the same kernels for head dims well below HD (fwd_hip.cpp compiled with -DFA2_TU_TRIM=1); -1 = none for this p.D
FA2_HIDDEN int launch_fwd_hip_trim_f16(int HD, const FwdParams& p, bool causal, int rows, hipStream_t stream);
FA2_HIDDEN int launch_fwd_hip_trim_bf16(int HD, const FwdParams& p, bool causal, int rows, hipStream_t stream);
// merge of the KV-split parts a forward launch left in p.ws (fwd_hip.cpp)
FA2_HIDDEN int launch_fwd_combine_f16(int HD, const FwdParams& p, hipStream_t stream);
FA2_HIDDEN int launch_fwd_combine_bf16(int HD, const FwdParams& p, hipStream_t stream);
// hand-scheduled forward, head dim exactly 128 or 64 (fwd_asm.cpp)
// fold: the body that folds scale * log2(e) into Q (FA2_CONTRACT_PRESCALE_Q) instead of scaling the f32 product
// m16: the body built on v_mfma_f32_16x16x32 (head dim 128, f32 scale, whole items only; csrc/gen/fwd_m16_gen.py) where it applies
// m16: 0 = the 32x32x16 bodies, 1 = the 16x16x32 bodies with the sum check (option "asm" bit 6), 2 = ... with the row sums on the matrix pipe (bits 6 and 9)
FA2_HIDDEN int launch_fwd_asm(int HD, bool bf16, const FwdParams& p, bool causal, bool fold, hipStream_t stream, int m16 = 0);
inline int fwd_m16_mode(int asm_mask) { return (asm_mask & 64) ? ((asm_mask & 512) ? 2 : 1) : 0; }
// Does launch_fwd_asm(..., m16) run a body built on v_mfma_f32_16x16x32 (csrc/gen/fwd_m16_gen.py)?  ONE predicate: the launcher executes it, the plan
// reports its contract (the folded 16 x 16 bodies add the ROUNDED P into the row sums: FA2_CONTRACT_LSUM_P16).  fwd_asm.cpp has the measurements.
enum { kM16None = 0, kM16F32 = 1, kM16Fold = 2, kM16F32Lm = 3, kM16FoldNoLm = 4 };
// hand-scheduled forward for head dim exactly 256 (round 6; fa2_fwd_d256.hip.h, csrc/gen/fwd_m16_d256_gen.py): 128-row workgroups, f32 scale,
// row sums of the rounded P (FA2_CONTRACT_LSUM_P16); host.cpp: plan_range decides which calls it takes
FA2_HIDDEN int launch_fwd_asm_d256(bool bf16, const FwdParams& p, bool causal, hipStream_t stream);
inline int fwd_asm_m16_kind(int HD, bool bf16, const FwdParams& p, bool fold, int m16) {
    if (!m16) return kM16None;
    const bool lm = m16 == 2;
    // folded scale: row sums on the matrix pipe; without (m16 == 1) head dim 128 takes the folded sum-check body, head dim 64 the 32x32x16 bodies
    // (p.D < HD, a head dim below the body's: the general form of the LDS-DMA offsets, any row pitch — csrc/gen/fwd_m16_gen.py: trim_offsets)
    if (fold) return (HD == 64 ? lm : (p.vs[2] % 32 == 0 || p.D < HD)) ? (lm ? kM16Fold : kM16FoldNoLm) : kM16None;
    // f32 scale.  A call flagged FA2_FLAG_EXACT_SCALE (a forward that will be differentiated) keeps the f32 row sums: fp16 at head dim 128 on the
    // 16 x 16 body with the sum check, everything else on the 32x32x16 bodies; other calls take the 16 x 16 bodies with the row sums on the matrix pipe
    // (head dim 128; at head dim 64 the f32-scale 16 x 16 bodies do not beat the 32x32x16 body: profiles/r18_kbench_f32lm*.txt)
    // (head dim 64: the f32-scale body with the sum check loses to the 32x32x16 body, the one with the row sums on the matrix pipe wins a little —
    //  bf16 B2 H16 N4096 non-causal 135.8 (8-wave HIP kernel) / 136.5 (32x32x16 body) -> 130.8 us, causal 75.6 -> 74.4: profiles/r18_kbench_d64_f32lm.txt)
    if (HD == 64) return (lm && !p.exact_scale) ? kM16F32Lm : kM16None;
    if (HD != 128) return kM16None;
    if (p.exact_scale) return !bf16 ? kM16F32 : kM16None;
    // (round 6: the lm bodies keep the conflict-free V image of the folded ones — the same condition on V's row pitch, see `fold` above)
    return (lm && (p.vs[2] % 32 == 0 || p.D < HD)) ? kM16F32Lm : (!bf16 ? kM16F32 : kM16None);
}
inline bool fwd_asm_is_m16(int HD, bool bf16, const FwdParams& p, bool fold, int m16) { return fwd_asm_m16_kind(HD, bf16, p, fold, m16) != kM16None; }
inline bool fwd_asm_lsum16(int HD, bool bf16, const FwdParams& p, bool fold, int m16) {
    const int k = fwd_asm_m16_kind(HD, bf16, p, fold, m16);
    return k == kM16Fold || k == kM16F32Lm;
}
// Plans of the backward's split passes (compiler-scheduled kernels; bwd_hip.cpp): the dQ pass splits its KV sweep (head dims <= 128), the fused
// dK / dV pass of head dims <= 64 its Q sweep.  Returns the workspace bytes fa2_bwd_ws can use (the passes run one after the other and share it).
// Tile costs (us per 64-row tile of a 256-row workgroup, 8-wave HIP kernels): dQ pass 3 GEMMs, fused dK / dV pass 4 — 1.5x / 2x the forward's 0.9 * HD / 64.
#ifndef FA2_BWD_DQ_UNDERFILLED
#define FA2_BWD_DQ_UNDERFILLED 1
#endif
inline int64_t plan_bwd_split(int HD, const BwdParams& p, bool causal, SplitPlan* dq, SplitPlan* dkv) {
    *dq = SplitPlan();
    *dkv = SplitPlan();
    if (causal || HD > 128 || !options().split.load(std::memory_order_relaxed)) return 0;
    if (options().rows.load(std::memory_order_relaxed) == 128) return 0;
    // the hand-scheduled passes (unmasked calls only: the masked backward runs the BIAS forms of the compiler-scheduled passes at every head dim)
    if (p.bias_kind == 0 && HD == 128 && p.D == 128 && (options().asm_mask.load(std::memory_order_relaxed) & 2)) return 0;
    const int64_t cus = device_cus(), bh = (int64_t)p.B * p.H, tile_bytes = (int64_t)kSplitRows * HD * 4;
    *dq = plan_tail_split(bh * ((p.Nq + 255) / 256), (p.Nkv + kKvTile - 1) / kKvTile, 1.35 * HD / 64.0, 10.0, tile_bytes, cus, FA2_BWD_DQ_UNDERFILLED != 0);
    if (HD <= 64) *dkv = plan_tail_split(bh * ((p.Nkv + 255) / 256), (p.Nq + kKvTile - 1) / kKvTile, 1.8 * HD / 64.0, 10.0, 2 * tile_bytes, cus, true);
    return dq->bytes > dkv->bytes ? dq->bytes : dkv->bytes;
}
// HIP backward (bwd_hip.cpp): parts bit 0 = dQ pass (+ delta workspace), bit 1 = dK / dV pass(es)
FA2_HIDDEN int launch_bwd_hip_f16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
FA2_HIDDEN int launch_bwd_hip_bf16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
// dQ pass (+ delta, or -delta for the hand-scheduled dK / dV pass) of KV sweeps of at most two tiles, non-causal, no bias, head dims <= 128 (fa2_bwd_short.hip.h; round 6)
FA2_HIDDEN int launch_bwd_short_dq_f16(int HD, const BwdParams& p, bool neg_delta, hipStream_t stream);
FA2_HIDDEN int launch_bwd_short_dq_bf16(int HD, const BwdParams& p, bool neg_delta, hipStream_t stream);
// trimmed instantiations of the same passes for head dims well below HD (bwd_hip.cpp compiled with -DFA2_TU_TRIM=1); -1 = none for this p.D
FA2_HIDDEN int launch_bwd_hip_trim_f16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
FA2_HIDDEN int launch_bwd_hip_trim_bf16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream);
// sum of the parts a split pass left in p.ws (bwd_merge_kernel, bwd_hip.cpp): which = 1: dQ, 2: dK and dV; head dims <= 128
FA2_HIDDEN int launch_bwd_merge_f16(int HD, const BwdParams& p, int which, hipStream_t stream);
FA2_HIDDEN int launch_bwd_merge_bf16(int HD, const BwdParams& p, int which, hipStream_t stream);
// HIP backward through a biased / masked forward (bwd_bias_hip.cpp): dQ, dV, dK; head dims up to 256
FA2_HIDDEN int launch_bwd_bias_hip_f16(int HD, const BwdParams& p, bool causal, hipStream_t stream);
FA2_HIDDEN int launch_bwd_bias_hip_bf16(int HD, const BwdParams& p, bool causal, hipStream_t stream);
// hand-scheduled backward, head dim exactly 128 (bwd_asm.cpp); same `parts`
// neg_delta: the dQ pass writes -delta (the hand-scheduled dK/dV pass reads it as such; the HIP dK/dV passes read +delta)
// kfold: the dK / dV body whose P side folds scale * log2(e) into its K fragments (option "fold"; host.cpp: bwd_folds)
// dq16: the dQ pass built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py)
// dkv16: likewise the dK / dV pass (csrc/gen/bwd_dkv_m16_gen.py)
FA2_HIDDEN int launch_bwd_d128(bool bf16, const BwdParams& p, bool causal, int parts, bool neg_delta, bool kfold, hipStream_t stream, bool dq16 = false,
                               bool dkv16 = false);

constexpr int kBwdAsmParts = 3;      // passes the hand-scheduled backward covers: bit 0 = dQ, bit 1 = dK / dV

}  // namespace fa2
