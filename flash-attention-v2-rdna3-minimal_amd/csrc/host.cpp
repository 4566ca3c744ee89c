// host.cpp — C-ABI shim of the gfx950 FlashAttention-2 path (compiled with hipcc -x hip): argument validation, launch
// heuristics and dispatch.  The kernels live in the other translation units of the library (fa2_launch.h).
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
#include "fa2_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fa2_gfx950.h"

namespace fa2 {

// The tuning switches start from the environment (read once, when the library is loaded) and can be changed at run time
// through fa2_set_option — tests flip them in-process instead of spawning a child process per variant.
Options& options() {
    static Options o;
    static const bool init = [] {
        if (const char* e = std::getenv("FA2_ROWS")) o.rows = std::atoi(e);
        if (const char* e = std::getenv("FA2_ASM")) o.asm_mask = std::atoi(e);
        if (const char* e = std::getenv("FA2_PERSIST")) o.persist = std::atoi(e);
        if (const char* e = std::getenv("FA2_SPLIT")) o.split = std::atoi(e) != 0;
        if (const char* e = std::getenv("FA2_FOLD")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) o.fold = v; }
        if (const char* e = std::getenv("FA2_KFOLD")) o.kfold = std::atoi(e) != 0;
        if (const char* e = std::getenv("FA2_SHORT")) o.short_kv = std::atoi(e) != 0;
        return true;
    }();
    (void)init;
    return o;
}

int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

}  // namespace fa2

namespace {

constexpr int kHeadDims[] = {64, 128, 256, 512};
constexpr int kMaxBwdHeadDim = 512;   // like the forward (above 256: slab kernels, bwd_hip.cpp)
constexpr int kNumHeadDims = sizeof(kHeadDims) / sizeof(kHeadDims[0]);
constexpr int kFwdRows = 256;         // Q rows per forward workgroup of the default shapes

int forced_rows() { return fa2::options().rows.load(std::memory_order_relaxed); }
bool asm_fwd() { return fa2::options().asm_mask.load(std::memory_order_relaxed) & 1; }

// A grid of 256-row workgroups that covers well under half of the CUs leaves the chip idle: 128-row workgroups double the
// number of busy CUs at the price of staging every K/V tile for half as many rows.  Measured (tools/rows_probe.py, MI355X, 256 CUs):
// 64 workgroups -> 128 of 128 rows: B1 H16 N1024 D128 25.1 -> 20.5 us, B1 H8 N2048 D128 44.1 -> 35.3, B2 H8 N1024 D80 24.0 -> 19.9;
// at 160 workgroups (SDXL 32x32 self-attention) and above the big shape wins (19.9 vs 23.9 us), 64-row workgroups never do.
// Between one and one and a half rounds of 256-row workgroups at head dims <= 64 (non-causal), 128-row workgroups all the way — two per CU since
// they fit the 256-register budget — beat both the plain grid and the tail split (SDXL 64x64 self-attention B2 H10 N4096 D64, 320 workgroups:
// 130.8 plain / 121.2 tail split / 117.4 us; B1 H24 N3072 D64, 288: 97.0 / 90.9 / 84.5 us); at exactly one round (B2 H8 N4096: 66 vs 73 us)
// and below (160 workgroups: 20.0 vs 22.6 us) the 8-wave shape stays ahead.  The thresholds scale with the device's CU count.
bool short_second_round(const fa2::FwdParams& p, bool causal) {
    if (causal || p.D > 64) return false;
    const int64_t w = (int64_t)p.nbh * ((p.Nq + 255) / 256), cus = fa2::device_cus();
    return w > cus && w <= cus + cus / 2;
}

int pick_rows(const fa2::FwdParams& p, bool causal = true) {
    const int f = forced_rows();
    if (f == 128 || f == 256) return f;
    if (p.rows_hint == 128 || p.rows_hint == 256) return p.rows_hint;
    if (short_second_round(p, causal)) return 128;
    return (int64_t)p.nbh * ((p.Nq + 255) / 256) <= fa2::device_cus() * 3 / 8 ? 128 : 256;
}

// Tail split (non-causal): B*H*ceil(Nq/256) equal workgroups on the CUs take ceil(x/CUs) rounds however empty the last one
// is — SDXL's 64x64 self-attention (B2 H10 N4096) is 320 workgroups, two rounds for 1.25 rounds of work.  When the last
// round would be at most half full, the heads that make it up run in a second launch as 128-row workgroups (twice as
// many, ~0.85x as long each: they stream the same K/V for half the rows).  Measured at D = 64 (tools/rows_probe.py): B2 H10
// N4096 130.8 -> 123.9 us, B1 H24 N3072 97.0 -> 92.5 us; one SDXL UNet step's attention 3.11 -> 3.02 ms.  Returns the number of
// heads of the main launch (= nbh: no split).
int tail_split_heads(const fa2::FwdParams& p, bool causal) {
    if (causal || forced_rows() != 0) return p.nbh;
    if (short_second_round(p, causal)) return p.nbh;            // 128-row workgroups for the whole grid instead (pick_rows)
    // a second launch costs ~5 us: only sweeps of at least 16 KV tiles (a workgroup then runs >= ~15 us) can win it back.  SDXL's
    // cross-attention (B2 H10 N4096 x Nkv 77, two tiles) measured 13.8 us split into two launches against 10.1 us for torch SDPA.
    if (p.Nkv < 16 * fa2::kKvTile) return p.nbh;
    const int64_t cus = fa2::device_cus(), nq = (p.Nq + 255) / 256, w = (int64_t)p.nbh * nq;
    if (w <= cus || nq > cus) return p.nbh;
    const int64_t main_heads = (w / cus) * cus / nq;           // whole heads that fit the full rounds
    const int64_t tail_w = (p.nbh - main_heads) * nq;           // 256-row workgroups of the remaining heads
    if (main_heads <= 0 || tail_w <= 0 || tail_w > cus / 2) return p.nbh;
    return (int)main_heads;
}

// Head dim exactly 128 with a positive scale runs the hand-scheduled 4-wave kernel (fa2_fwd_d128.hip.h) unless option "asm"
// bit 0 is cleared (A/B measurements, tools/kbench.py).  The asm block addresses a head's Q rows with 32-bit byte offsets.
// ... and derives a wave's further LDS-DMA source offsets from its first by flipping granule bits of the byte offset (the K image's swizzle follows
// the row), which is only the same as re-swizzling when a row's byte offset has those bits clear: K's row pitch must be a multiple of one tile
// row (2 * HD bytes).  Contiguous BHND and BNHD tensors are; a column slice of a wider matrix is not and runs on the HIP kernels (found by the
// randomised sweep, tools/fuzz_parity.py, once it drew grids wide enough for these kernels: profiles/r06_fuzz_parity_seed5.json).
bool asm_pitch_ok(int64_t row_stride_elems, int HD) { return row_stride_elems % HD == 0; }
// (round 5) Q is staged by LDS-DMA like K: its row pitch has to be a multiple of one tile row too, and the byte offsets of the (up to 63) rows a
// workgroup's last wave reads past Nq — zero-filled by the descriptor — must not wrap
// (any_pitch: launches of a head dim below the body's take the general form of the LDS-DMA offsets — csrc/gen/fwd_m16_gen.py: trim_offsets)
// (any_pitch launches mark a granule the row does not have with byte offset 0x80000000 — "beyond every descriptor": true only while the Q span
//  itself stays below 2 GiB, as K's and V's do)
bool asm_q_span_ok(const fa2::FwdParams& p, bool any_pitch = false) {
    return ((int64_t)(p.Nq + 64) * p.qs[2] + p.D) * 2 < ((int64_t)1 << (any_pitch ? 31 : 32)) && (any_pitch || p.qs[2] % (p.D > 0 ? p.D : 1) == 0);
}
// ... and pays a fixed head and tail per item (the Q tile through LDS, the first K / V tiles before any MFMA, the drain of the software pipeline):
// over a short KV sweep (cross-attention, low-resolution self-attention) the compiler-scheduled kernels — two waves per SIMD hiding each
// other's prologue — are faster.  tools/asm_kv_ab.py, one box, fp16, B4 H16 N4096, HIP time / hand-scheduled time at Nkv = 77, 256, 512,
// 768, 1024, 2048:  D = 64  0.88 0.90 0.96 0.995 1.015 1.06;  D = 128  1.006 0.86 0.955 1.00 1.03 1.08 (bf16 1.12 0.925 0.99 1.02 1.04 1.07;
// bf16 keeps its two-tile sweeps on the hand-scheduled body, fp16 measured 0.94 .. 1.006 there).  The crossover is the same at B2 and at Nq = 1024.
// Causal self-attention sweeps half the sequence on average: B8 H16, N = 512 768 1024 1536 2048: D = 64 0.90 0.90 0.97 0.88 1.10, D = 128 0.92 0.90
// 0.96 0.89 1.07 (bf16 alike) — the hand-scheduled body from N = 1792 (profiles/r06_asm_kv_ab.txt).  Option "asm" bit 5 and option "rows" = 256 ignore this rule (A/B measurements, tests).
bool asm_folds(bool bf16, const fa2::FwdParams& p);
bool asm_kv_len_ok(int HD, bool bf16, const fa2::FwdParams& p, bool causal) {
    if ((fa2::options().asm_mask.load(std::memory_order_relaxed) & 32) || forced_rows() == 256) return true;     // (option rows = 256 pins the
    if (causal) return p.Nkv >= 1792;                                                                              //  hand-scheduled kernels: tests)
    // Round 5, re-measured with the 16x16x32 bodies (tools/asm_kv_ab.py, profiles/r18_asm_kv_ab.txt; HIP time / hand-scheduled time at Nkv = 256, 512,
    // 768, 1024, 1536): D = 128 fp16 0.90 1.01 1.07 1.11 1.16, bf16 0.91 1.01 1.05 1.07 1.09 -> from 512 on; D = 64 fp16 (folded) 0.82 0.92 0.99 1.03
    // 1.10 -> from 896 on as before; D = 64 bf16 (f32 scale) 0.81 0.88 0.92 0.97 1.02 -> from 1280 on.
    if (p.Nkv >= (HD == 128 ? 512 : (asm_folds(bf16, p) ? 896 : 1280))) return true;
    return HD == 128 && bf16 && p.Nkv <= 2 * fa2::kKvTile;
}

// Does a launch of the hand-scheduled body fold scale * log2(e) into Q (rounded once to the I/O dtype: FA2_CONTRACT_PRESCALE_Q), or does it scale
// the f32 product like the reference kernel (kernel_fp16.cu:164) and every compiler-scheduled kernel here?  Option "fold": 0 never, 1 (default)
// fp16 launches, 2 bf16 launches too.  Only while c = scale * log2(e) <= 1: the prescaled Q must stay inside the I/O dtype's range whatever the
// caller's Q holds (|q| * c <= |q|); a larger scale runs the f32-scale body of the same schedule (scale > 0.69 could overflow fp16).
bool asm_folds(bool bf16, const fa2::FwdParams& p) {
    const int f = fa2::options().fold.load(std::memory_order_relaxed);
    return (bf16 ? f >= 2 : f >= 1) && p.c <= 1.0f && !p.exact_scale;
}

// What one launch over the heads [p.bh0, p.bh0 + p.nbh) runs: the ONE place that decides it (launch_range executes the plan, fa2_fwd_plan reports it).
struct RangePlan { int kernel, contract; int rows; bool fold; bool short_kv = false; };

// Non-causal sweeps of at most two KV tiles without a bias (cross-attention on a text prompt: Nkv = 77) at head dims <= 128: the single-pass kernel
// (fa2_fwd_short.hip.h; option "short").  Option "rows" pins the streaming kernels (tests, A/B).
bool short_kv_ok(int HD, const fa2::FwdParams& p, bool causal) {
    return !causal && HD <= 128 && p.Nkv <= 2 * fa2::kKvTile && p.bias_kind == FA2_BIAS_NONE && forced_rows() == 0 &&
           fa2::options().short_kv.load(std::memory_order_relaxed) != 0;
}

// A head dim below the body's that the 16x16x32 bodies take (see plan_range)
bool asm_trimmed(int HD, bool bf16, const fa2::FwdParams& p, bool fold) {
    const int m16_mode = fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed));
#ifndef FA2_TRIM_MIN64       // (developer A/B: tools/kbench.py build NAME:-DFA2_TRIM_MIN64=32,-DFA2_TRIM_MIN128=80,only=host)
#define FA2_TRIM_MIN64 40
#endif
#ifndef FA2_TRIM_MIN128
#define FA2_TRIM_MIN128 88      // (B2 H16 N4096 fp16, same file: D = 72 +4 %, 80 +2 %, 88 +7 %, 96 +6 %; D = 24 / 32 on the 64 body: -1 %)
#endif
    return p.D < HD && p.D >= (HD == 64 ? FA2_TRIM_MIN64 : fold ? FA2_TRIM_MIN128 : 104) && fa2::fwd_asm_m16_kind(HD, bf16, p, fold, m16_mode) != fa2::kM16None;
}

// Head dim exactly 256 (round 6): the hand-scheduled 128-row kernel (fa2_fwd_d256.hip.h; option "asm" bit 10).  It scales the f32 product and adds the
// rounded P into the row sums on the matrix pipe (FA2_CONTRACT_LSUM_P16), so calls flagged FA2_FLAG_EXACT_SCALE keep the compiler-scheduled kernels;
// its LDS-DMA pieces are derived from piece 0 by flipping offset bits: K / V row pitches must be multiples of one 512-byte tile row; Q and O are
// addressed with 32-bit offsets from the head base; short KV sweeps stay on the 8-wave kernels (two waves per SIMD hide each other's prologue).
bool asm_d256_ok(bool bf16, const fa2::FwdParams& p, bool causal) {
    (void)bf16;
    const int mask = fa2::options().asm_mask.load(std::memory_order_relaxed);
#ifndef FA2_D256_TRIM_MIN      // (developer A/B: tools/kbench.py build NAME:-DFA2_D256_TRIM_MIN=144,only=host)
#define FA2_D256_TRIM_MIN 136
#endif
    if (!(mask & 1) || !(mask & 1024) || p.D > 256 || p.D < FA2_D256_TRIM_MIN || p.negate_q || p.exact_scale || p.bias_kind != 0) return false;
    // Head dims below 256 run ceil(D / 32) k-steps of the body (fwd_asm.cpp).  tools/d256_ab.py, one box (profiles/r21_d256_ab.txt), against the trimmed
    // compiler-scheduled kernels: D 176 .. 248 1.25 .. 1.51x, causal or not; D 136 .. 160 (the 8-wave kernel runs those in ONE pass over all columns)
    // non-causal 1.03 .. 1.05x, causal 0.86x at N = 4096: those keep the 8-wave kernel when causal.
    if (p.D < 176 && causal) return false;
    // D == 256: the pieces of a wave's LDS-DMA share are derived from piece 0 by flipping offset bits — row pitches that are multiples of one 512-byte
    // tile row; below, the general form of the offsets: any pitch
    if (p.D == 256 && (p.ks[2] % 256 || p.vs[2] % 256)) return false;
    if (((int64_t)(p.Nq + 32) * p.qs[2] + 256) * 2 >= ((int64_t)1 << 32) || ((int64_t)(p.Nq + 32) * p.os[2] + 256) * 2 >= ((int64_t)1 << 32)) return false;
    if (forced_rows() == 256) return false;              // (option rows = 256 pins the 256-row kernels: tests, A/B)
    return (mask & 32) || p.Nkv >= (causal ? 1024 : 512);
}

RangePlan plan_range(int HD, bool bf16, const fa2::FwdParams& p, bool causal) {
    if (HD == 256 && asm_d256_ok(bf16, p, causal)) return {FA2_KERNEL_ASM, FA2_CONTRACT_LSUM_P16, 128, false};
    if (short_kv_ok(HD, p, causal)) return {FA2_KERNEL_HIP_128, 0, 128, false, true};
    const int rows = pick_rows(p, causal);
    // Head dim exactly 128 with a positive scale: the hand-scheduled 4-wave kernel (256-row workgroups).  Head dim 64 has its generated
    // body too (same generator, half the MFMAs per tile for the same softmax work, row sums on the matrix pipe), but a lone wave per SIMD
    // issues that VALU-bound mix no faster than the two waves of the compiler-scheduled 8-wave kernel: same box (tools/fwd_ab.py) B2 H16
    // N4096 933 vs 947 TF, B1 H24 N8192 1032 vs 1022, causal bf16 917 vs 869 — so it takes the causal launches (+5.5 %) only; option
    // "asm" bit 4 sends every D = 64 launch to it (A/B measurements).
    // (round 3) the body that folds the scale into Q beats the 8-wave kernel non-causal too (+9 %), so every launch that folds takes it.
    const bool fold = asm_folds(bf16, p);
    // (round 5: and every launch the f32-scale 16x16x32 body with the row sums on the matrix pipe takes: calls that are not flagged FA2_FLAG_EXACT_SCALE)
    const bool d64_asm = HD == 64 && (causal || fold || (fa2::options().asm_mask.load(std::memory_order_relaxed) & 16) ||
                                      fa2::fwd_asm_m16_kind(HD, bf16, p, fold, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed))) == fa2::kM16F32Lm);
    // Head dims BELOW the body's (round 5; the 16x16x32 bodies only): the padded columns of the Q / K / V images are zero-filled by the LDS-DMA itself (a
    // granule the row does not have gets a source offset beyond the descriptor) and the piece offsets take their general form, so any row pitch goes.
    // The body runs at D / HD of its rate: it takes the head dims where that still beats the trimmed compiler-scheduled kernels
    // (tools/trim_asm_ab.py, profiles/r18_trim_asm_ab.txt, one box): fp16 D = 40 (SD 1.5's 64 x 64 self-attention, B2 H8 N4096) 72.0 -> 64.0 us, D = 48
    // causal +5.8 %, D = 96 +6.5 %, D = 104 .. 120 +19 .. 20 %, D = 112 causal +13 %; bf16 (f32 scale) D = 112 +14 %, D = 96 causal -8.7 %: the folded
    // bodies from 40 / 88 on, the f32-scale ones from 104 on.
    const bool trimmed = asm_trimmed(HD, bf16, p, fold);
    if ((HD == 128 || d64_asm) && (p.D == HD || trimmed) && !p.negate_q && asm_fwd() && rows == 256 && pick_rows(p) == 256 && asm_q_span_ok(p, trimmed) &&
        (trimmed || asm_pitch_ok(p.ks[2], HD)) && asm_kv_len_ok(HD, bf16, p, causal)) {
        // (the folded bodies built on v_mfma_f32_16x16x32 add the rounded P into the row sums — on the matrix pipe; csrc/gen/fwd_m16_gen.py, opt=lm)
        const bool lsum16 = fa2::fwd_asm_lsum16(HD, bf16, p, fold, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed)));
        return {FA2_KERNEL_ASM, (fold ? FA2_CONTRACT_PRESCALE_Q : 0) | (lsum16 ? FA2_CONTRACT_LSUM_P16 : 0), 256, fold};
    }
    return {rows == 256 ? FA2_KERNEL_HIP_256 : FA2_KERNEL_HIP_128, 0, rows, false};
}

int launch_range(int HD, bool bf16, const fa2::FwdParams& p, bool causal, hipStream_t stream) {
    const RangePlan r = plan_range(HD, bf16, p, causal);
    // option "asm" bit 6 (default): head dim 128 launches of whole items take the bodies built on v_mfma_f32_16x16x32 (round 5; fwd_asm.cpp)
    if (r.kernel == FA2_KERNEL_ASM && HD == 256) return fa2::launch_fwd_asm_d256(bf16, p, causal, stream);
    if (r.kernel == FA2_KERNEL_ASM)
        return fa2::launch_fwd_asm(HD, bf16, p, causal, r.fold, stream, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed)));
    if (r.short_kv) return bf16 ? fa2::launch_fwd_short_bf16(HD, p, stream) : fa2::launch_fwd_short_f16(HD, p, stream);
    return bf16 ? fa2::launch_fwd_hip_bf16(HD, p, causal, r.rows, false, stream) : fa2::launch_fwd_hip_f16(HD, p, causal, r.rows, false, stream);
}

// non-causal launches the hand-scheduled persistent kernels take: head dim 128, and head dim 64 when the launch folds the scale (plan_range)
bool asm_noncausal_ok(int HD, bool bf16, const fa2::FwdParams& p) {
    const bool fold = asm_folds(bf16, p);
    const bool d64 = HD == 64 && (fold || (fa2::options().asm_mask.load(std::memory_order_relaxed) & 16) ||
                                  fa2::fwd_asm_m16_kind(HD, bf16, p, fold, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed))) == fa2::kM16F32Lm);
    const bool trimmed = asm_trimmed(HD, bf16, p, fold);
    return (HD == 128 || d64) && (p.D == HD || trimmed) && !p.negate_q && asm_fwd() && asm_q_span_ok(p, trimmed) && (trimmed || asm_pitch_ok(p.ks[2], HD)) &&
           asm_kv_len_ok(HD, bf16, p, false);
}

// KV-split tail (fa2_fwd_ws).  B*H*ceil(Nq/256) equal workgroups on the CUs take ceil(x / CUs) rounds however empty the last one is: SDXL's
// 64x64 self-attention (B2 H10 N4096 D64) is 320 workgroups — two rounds for 1.25 rounds of work —, the reference harness's own sweep
// (B1 H24 D64, bench_with_sdpa.py:201-224) saw-tooths between 670 and 930 TF with N for the same reason.  With a workspace from the caller the r
// items of the last round are swept by S workgroups each over disjoint KV ranges (r * S parts fill the CUs again) and a small kernel merges the
// partial results (fa2_fwd_kernel.hip.h).  S minimises rounds(r * S) / S plus the fixed cost of the scheme, in units of one whole item:
//   an item sweeps nt KV tiles at ~0.9 us * HD / 64 each (measured: D = 64 N4096 60 us, D = 128 112 us per 256-row item);
//   the merge kernel, its launch and a part's own prologue / epilogue cost ~10 us (+ ~4 us for the extra launch of the D = 128 path);
//   the f32 partial tiles cross memory twice.  The workspace never exceeds 64 MiB (fa2::plan_tail_split, fa2_launch.h).
fa2::SplitPlan plan_split(const fa2::FwdParams& p, int HD, bool bf16, bool causal) {
    fa2::SplitPlan none;
    if (causal || p.bias_kind != FA2_BIAS_NONE || HD > 128 || !fa2::options().split.load(std::memory_order_relaxed)) return none;
    if (short_kv_ok(HD, p, causal)) return none;
    const int f = forced_rows();
    if (f == 128 || p.rows_hint == 128) return none;
    const int64_t items = (int64_t)p.nbh * ((p.Nq + kFwdRows - 1) / kFwdRows);
    const int nt = (p.Nkv + fa2::kKvTile - 1) / fa2::kKvTile;
    // whole rounds on the hand-scheduled persistent kernel (head dim 128; head dim 64 in fp16): the parts need a launch of their own
    const bool asm_rounds = asm_noncausal_ok(HD, bf16, p);
    // (round 6) ... and EVERY item of a grid that covers at most half of the CUs over a long sweep — a decode-sized call: B1 H32 Nq1 Nkv8192 is 32
    // workgroups streaming 134 MB of K / V, 147 us for 30 us of bytes (plan_tail_split, underfilled; the parts run on the 8-wave kernel: plan_fwd)
    return fa2::plan_tail_split(items, nt, 0.9 * HD / 64.0, asm_rounds ? 14.0 : 10.0, fa2::split_ws_bytes(1, 1, HD), fa2::device_cus(), true);
}

// The whole forward call: at most two launches (+ the merge of a split).  plan_fwd decides, launch_fwd executes, fa2_fwd_plan reports.
struct FwdPlan {
    fa2::SplitPlan split;        // nsplit > 1: the last round's items run as KV-split parts (needs the caller's workspace)
    bool split_asm = false;      // ... inside the hand-scheduled persistent kernel (else the 8-wave HIP kernel)
    int main_heads = 0;          // heads [0, main_heads) run `main`, the others `tail` (HD <= 64 tail split; main_heads == nbh: one launch)
    RangePlan main{0, 0, 0, false}, tail{0, 0, 0, false};
};

FwdPlan plan_fwd(int HD, bool bf16, const fa2::FwdParams& p0, bool causal, bool have_ws, size_t ws_bytes) {
    FwdPlan f;
    f.main_heads = p0.nbh;
    if (have_ws) {
        const fa2::SplitPlan pl = plan_split(p0, HD, bf16, causal);
        if (pl.nsplit > 1 && (int64_t)ws_bytes >= fa2::split_ws_bytes(pl.split_items, pl.nsplit, HD)) {
            fa2::FwdParams p = p0;
            p.rows_hint = 256;
            f.split = pl;
            f.split_asm = asm_noncausal_ok(HD, bf16, p) && pick_rows(p, causal) == 256 &&
                          // (a grid of parts only — plan_split, underfilled: the persistent kernel where a part sweeps at least 24 tiles, else the 8-wave
                          //  kernel, whose prologue is shorter.  Same box, hand-scheduled / 8-wave parts: B1 H8 N4096 D128, 2 parts of 32 tiles, 67.6 / 78.5 us,
                          //  D40 44.5 / 48.3, bf16 D64 48.7 / 53.3; B1 H4 N2048 D128, 4 x 8 tiles, 29.2 / 27.5; B1 H32 Nq1 Nkv8192, 8 x 16, 48.4 / 44.9)
                          (pl.full_items > 0 || ((p.Nkv + fa2::kKvTile - 1) / fa2::kKvTile) / pl.nsplit >= 24);
            const bool fold = f.split_asm && asm_folds(bf16, p);
            const bool lsum16 = f.split_asm && fa2::fwd_asm_lsum16(HD, bf16, p, fold, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed)));
            f.main = f.split_asm ? RangePlan{FA2_KERNEL_ASM, (fold ? FA2_CONTRACT_PRESCALE_Q : 0) | (lsum16 ? FA2_CONTRACT_LSUM_P16 : 0), 256, fold}
                                 : RangePlan{FA2_KERNEL_HIP_256, 0, 256, false};
            return f;
        }
    }
    if (HD <= 64) {     // (measured at D = 128, B1 H24 N4096: 188 -> 194 us — the 128-row shape is too slow there)
        const int main_heads = tail_split_heads(p0, causal);
        if (main_heads < p0.nbh) {
            fa2::FwdParams p = p0;
            p.nbh = main_heads;
            f.main_heads = main_heads;
            f.main = plan_range(HD, bf16, p, causal);
            p.bh0 = p0.bh0 + main_heads;
            p.nbh = p0.nbh - main_heads;
            p.rows_hint = 128;
            f.tail = plan_range(HD, bf16, p, causal);
            return f;
        }
    }
    f.main = plan_range(HD, bf16, p0, causal);
    return f;
}

int launch_fwd(int HD, bool bf16, const fa2::FwdParams& p0, bool causal, hipStream_t stream, void* ws, size_t ws_bytes) {
    const bool have_ws = ws && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0;
    const FwdPlan f = plan_fwd(HD, bf16, p0, causal, have_ws, ws_bytes);
    if (f.split.nsplit > 1) {
        fa2::FwdParams p = p0;
        p.rows_hint = 256;
        p.full_items = f.split.full_items; p.split_items = f.split.split_items; p.nsplit = f.split.nsplit;
        p.ws = (float*)ws;
        int rc;
        if (f.split_asm) {
            // the hand-scheduled persistent kernel: every workgroup works through its whole items, then its parts (the item seam hides a
            // part's load phase like any other item's; the block stores a part's f32 tile itself)
            p.item_cap = f.split.full_items + f.split.split_items * f.split.nsplit;
            rc = fa2::launch_fwd_asm(HD, bf16, p, false, f.main.fold, stream, fa2::fwd_m16_mode(fa2::options().asm_mask.load(std::memory_order_relaxed)));
        } else {
            rc = bf16 ? fa2::launch_fwd_hip_bf16(HD, p, false, 256, false, stream) : fa2::launch_fwd_hip_f16(HD, p, false, 256, false, stream);
        }
        if (rc) return rc;
        return bf16 ? fa2::launch_fwd_combine_bf16(HD, p, stream) : fa2::launch_fwd_combine_f16(HD, p, stream);
    }
    if (f.main_heads < p0.nbh) {
        fa2::FwdParams p = p0;
        p.nbh = f.main_heads;
        if (int rc = launch_range(HD, bf16, p, causal, stream)) return rc;
        p.bh0 = p0.bh0 + f.main_heads;
        p.nbh = p0.nbh - f.main_heads;
        p.rows_hint = 128;
        return launch_range(HD, bf16, p, causal, stream);
    }
    return launch_range(HD, bf16, p0, causal, stream);
}

// Head dim exactly 128 runs the hand-scheduled backward kernels (fa2_bwd_d128.hip.h) unless option "asm" bit 1 is cleared; bits 2 / 3 of
// the option take only the dQ pass / only the dK-dV pass off them (A/B measurements of one pass at a time).
int launch_bwd(int HD, bool bf16, const fa2::BwdParams& p, bool causal, hipStream_t stream) {
    const int m = fa2::options().asm_mask.load(std::memory_order_relaxed);
    const int want = fa2::options().bwd_parts.load(std::memory_order_relaxed);      // 3 unless a profiling run asked for one pass only
    int asm_parts = 0;
    if (HD == 128 && p.D == 128 && (m & 2)) asm_parts = 3 & ~((m >> 2) & 3) & fa2::kBwdAsmParts;
    // row pitches the generated bodies' LDS-DMA offset arithmetic holds for (asm_pitch_ok): the dQ pass stages K and V, the dK / dV pass Q and dO
    if (!asm_pitch_ok(p.ks[2], 128) || !asm_pitch_ok(p.vs[2], 128)) asm_parts &= ~1;
    if (!asm_pitch_ok(p.qs[2], 128) || !asm_pitch_ok(p.dos[2], 128)) asm_parts &= ~2;
    if (p.Nq % 32 != 0) asm_parts &= ~2;              // the hand-scheduled dK/dV pass sweeps whole 32-row Q tiles
    // the sign delta crosses the workspace with: the hand-scheduled dK/dV pass takes -delta (C operand of its dP product); a dQ pass
    // that is not the hand-scheduled one writes +delta, so the two only go together
    if ((asm_parts & 2) && !(asm_parts & 1)) asm_parts = 0;
    const bool neg_delta = (asm_parts & 2) != 0;
    // Option "fold" reaches the backward too: the hand-scheduled dK / dV pass then recomputes P from K * scale*log2(e) rounded once to the I/O dtype
    // (the forward's folded contract on the other operand of Q.K^T) with L as the C operand of the product — 32 v_fma fewer per body on the wave role
    // every body waits for.  Same guard as the forward: |scale * log2(e)| <= 1 keeps the prescaled K inside the dtype's range.
    const int fopt = fa2::options().fold.load(std::memory_order_relaxed);
    // Round 5: off unless option "kfold" asks for it.  The forward of a differentiated call scales the f32 product (FA2_FLAG_EXACT_SCALE) and so do both
    // backward passes: P is recomputed from the very scores L was formed from.  With the fold, gradients at logits of +-30 and more were 2-4x the
    // suite's tolerance (tests/test_backward_gpu.py::test_hand_scheduled_backward_on_large_logits_under_both_fold_settings).
    const bool kfold = fa2::options().kfold.load(std::memory_order_relaxed) && (bf16 ? fopt >= 2 : fopt >= 1) && std::fabs(p.c) <= 1.0f;
    // (head dim 128 exactly keeps the hand-scheduled dQ pass where that one is available: B8 H16 N4096 x 77, whole backward, 236 us against 246)
    const bool short_dq = !causal && (HD <= 64 || (HD == 128 && !(asm_parts & 1))) && p.Nkv <= 2 * fa2::kKvTile && p.bias_kind == FA2_BIAS_NONE &&
                          forced_rows() == 0 && fa2::options().short_kv.load(std::memory_order_relaxed) != 0;
    for (int part = 1; part <= 2; part <<= 1) {       // the dQ pass first: it fills the delta workspace the dK / dV pass reads
        if (!(want & part)) continue;
        int rc;
        // (round 6) the dQ pass of a non-causal sweep of at most two KV tiles — cross-attention — is the short-sweep kernel's (fa2_bwd_short.hip.h;
        //  option "short"; option "rows" pins the streaming passes); it leaves delta in the sign the dK / dV pass that follows takes
        if (part == 1 && short_dq) rc = bf16 ? fa2::launch_bwd_short_dq_bf16(HD, p, neg_delta, stream) : fa2::launch_bwd_short_dq_f16(HD, p, neg_delta, stream);
        else if (asm_parts & part) rc = fa2::launch_bwd_d128(bf16, p, causal, part, neg_delta, kfold, stream, (m & 128) != 0, (m & 256) != 0);     // option "asm" bits 7 / 8: the 16x16x32 dQ / dK-dV pass
        else rc = bf16 ? fa2::launch_bwd_hip_bf16(HD, p, causal, part, stream) : fa2::launch_bwd_hip_f16(HD, p, causal, part, stream);
        if (rc) return rc;
    }
    return 0;
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
    // the fp16 launches that take the hand-scheduled bodies MAY fold (plan_range; fa2_fwd_plan says what one call does)
    return (D == 64 || D == 128) && scale > 0.f && scale * 1.4426950408889634f <= 1.0f && fa2::options().fold.load(std::memory_order_relaxed) >= 1 ? 1 : 0;
}

int fa2_set_option(const char* name, int value) {
    if (!name) return FA2_ERR_NULL_POINTER;
    fa2::Options& o = fa2::options();
    if (!std::strcmp(name, "rows")) { if (value != 0 && value != 128 && value != 256) return FA2_ERR_BAD_SHAPE; o.rows = value; }
    else if (!std::strcmp(name, "asm")) o.asm_mask = value;
    else if (!std::strcmp(name, "persist")) o.persist = value != 0;
    else if (!std::strcmp(name, "split")) o.split = value != 0;
    else if (!std::strcmp(name, "fold")) { if (value < 0 || value > 2) return FA2_ERR_BAD_SHAPE; o.fold = value; }
    else if (!std::strcmp(name, "bwd_parts")) { if (value < 1 || value > 3) return FA2_ERR_BAD_SHAPE; o.bwd_parts = value; }
    else if (!std::strcmp(name, "kfold")) o.kfold = value != 0;
    else if (!std::strcmp(name, "short")) o.short_kv = value != 0;
    else return FA2_ERR_BAD_SHAPE;
    o.epoch.fetch_add(1, std::memory_order_relaxed);
    return FA2_OK;
}

int fa2_get_option(const char* name) {
    if (!name) return FA2_ERR_NULL_POINTER;
    fa2::Options& o = fa2::options();
    if (!std::strcmp(name, "rows")) return o.rows.load();
    if (!std::strcmp(name, "asm")) return o.asm_mask.load();
    if (!std::strcmp(name, "persist")) return o.persist.load();
    if (!std::strcmp(name, "split")) return o.split.load();
    if (!std::strcmp(name, "fold")) return o.fold.load();
    if (!std::strcmp(name, "epoch")) return o.epoch.load() & 0x3fffffff;
    if (!std::strcmp(name, "bwd_parts")) return o.bwd_parts.load();
    if (!std::strcmp(name, "kfold")) return o.kfold.load();
    if (!std::strcmp(name, "short")) return o.short_kv.load();
    return FA2_ERR_BAD_SHAPE;
}

const char* fa2_error_string(int code) {
    switch (code) {
        case FA2_OK: return "ok";
        case FA2_ERR_NULL_POINTER: return "fa2: null pointer argument";
        case FA2_ERR_BAD_SHAPE: return "fa2: B, H, Nq, Nkv, D must be >= 1 and one head's matrix must span < 2 GiB";
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

const char* fa2_version(void) { return "fa2_gfx950 0.9 (D=128 forward (sum-check fast bodies) + backward and D=64 forward: hand-scheduled 4-wave asm bodies; other head dims up to 512: HIP kernels; mfma32x32x16, lds-dma; KV-split tail rounds; attention bias / mask, forward + backward)"; }

static int fwd_impl(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                    int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                    const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                    float scale, int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream,
                    void* ws = nullptr, size_t ws_bytes = 0, size_t* ws_need = nullptr, fa2_fwd_plan_t* plan_out = nullptr) {
    // `causal` carries the call's flags: bit 0 = causal mask, bit 1 = FA2_FLAG_EXACT_SCALE (this call scales the f32 product whatever option "fold" says)
    // Until round 5 any non-zero value meant "causal"; a caller that still passes another truthy int would silently get a non-causal forward: refuse it
    if (causal & ~(FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE)) return FA2_ERR_BAD_SHAPE;
    const bool exact_scale = (causal & FA2_FLAG_EXACT_SCALE) != 0;
    causal &= 1;
    // ws_need / plan_out: validate and plan only (fa2_fwd_workspace_bytes, fa2_fwd_plan) — the data pointers are stand-ins then
    const bool plan_only = ws_need || plan_out;
    if (ws_need) *ws_need = 0;
    if (!q || !k || !v || !o || !lse || !q_strides || !k_strides || !v_strides || !o_strides || !lse_strides)
        return FA2_ERR_NULL_POINTER;
    if (bias_kind != FA2_BIAS_NONE) {
        if (bias_kind != FA2_BIAS_IO_DTYPE && bias_kind != FA2_BIAS_F32 && bias_kind != FA2_BIAS_BOOL) return FA2_ERR_BIAS;
        if (plan_only) { static const int64_t z3[3] = {0, 0, 0}; if (!bias) bias = q; if (!bias_strides) bias_strides = z3; }
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
    p.exact_scale = exact_scale ? 1 : 0;
    p.persist = 1;
    p.full_items = p.split_items = p.nsplit = p.blk0 = p.item_cap = 0;
    p.ws = nullptr;
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
        // 3: the same geometry on a grid of more than 3/8 of the CUs' worth of 256-row workgroups, head dims <= 128, one (b, h) slice of the bias within
        // 32-bit byte offsets: the 8-wave shape with the tile staged by LDS-DMA (no bias registers).  Measured (tools/mask_bench.py): dense fp16 bias
        // shared by the heads, B2 H10 N4096 D64, 445 us as 4-wave workgroups; torch SDPA 339.
        if (p.bias_vec == 2 && HD <= 128 && forced_rows() != 128 && (int64_t)B * H * ((Nq + 255) / 256) > fa2::device_cus() * 3 / 8 &&
            ((int64_t)(Nq - 1) * p.bs[2] + Nkv + 64 * p.bs[2]) * (int64_t)esize < 0x7fffffffLL)
            p.bias_vec = 3;
        // 4: a bias broadcast over the Q rows (row stride 0 — the [B, 1, 1, Nkv] key-padding mask of padded token batches and of SD cross-attention),
        // any alignment: a wave fetches the tile's 64 values once and spreads them through its LDS image (head dims <= 256: the image exists)
        if (p.bs[2] == 0 && HD <= 256) p.bias_vec = 4;
    }
    if ((int64_t)B * H * p.nqblk > 0x7fffffffLL) return FA2_ERR_GRID;

    hipStream_t stream = (hipStream_t)hip_stream;
    const bool bf16 = dtype == FA2_DTYPE_BF16;
    if (bias_kind != FA2_BIAS_NONE) {
        if ((int64_t)B * H * ((Nq + 127) / 128) > 0x7fffffffLL) return FA2_ERR_GRID;
        if (plan_out) {
            std::memset(plan_out, 0, sizeof(*plan_out));
            plan_out->kernel = FA2_KERNEL_HIP_BIAS;
            plan_out->rows = 0;       // unspecified: the load form (and with it 128- or 256-row workgroups) depends on the bias strides and alignment, which the query does not take
            plan_out->heads_main = B * H;
        }
        if (plan_only) return FA2_OK;
        return bf16 ? fa2::launch_fwd_hip_bf16(HD, p, causal != 0, 128, true, stream) : fa2::launch_fwd_hip_f16(HD, p, causal != 0, 128, true, stream);
    }
    if (ws_need) {      // fa2_fwd_workspace_bytes
        // The size query has no scale argument, and the split plan depends on the scale at head dim 64 (a launch that folds the scale — c <= 1 — runs
        // the parts inside the hand-scheduled kernel: other fixed costs, possibly another S): the answer is the larger of the two plans, so a
        // workspace of this size serves the call whatever its scale (the callers' caches are keyed without the scale).
        // Likewise FA2_FLAG_EXACT_SCALE (the forward of a call that will be differentiated never folds: another kernel, other fixed costs, possibly
        // another S): both settings are planned, whatever the query's flag says — the callers' caches are keyed without it too.
        size_t need = 0;
        for (const float c_try : {0.5f, 2.0f})
            for (const int exact_try : {0, 1}) {
                fa2::FwdParams pt = p;
                pt.c = c_try;
                pt.exact_scale = exact_try;
                const fa2::SplitPlan pl = plan_split(pt, HD, dtype == FA2_DTYPE_BF16, causal != 0);
                if (pl.nsplit > 1) need = std::max(need, (size_t)fa2::split_ws_bytes(pl.split_items, pl.nsplit, HD));
            }
        *ws_need = need;
    }
    if (plan_out) {     // fa2_fwd_plan: what launch_fwd would do with a (16-byte aligned) workspace of ws_bytes bytes
        const FwdPlan f = plan_fwd(HD, bf16, p, causal != 0, ws_bytes > 0, ws_bytes);
        std::memset(plan_out, 0, sizeof(*plan_out));
        plan_out->kernel = f.main.kernel; plan_out->contract = f.main.contract; plan_out->rows = f.main.rows;
        plan_out->heads_main = f.main_heads;
        plan_out->kernel_tail = f.tail.kernel; plan_out->contract_tail = f.tail.contract; plan_out->rows_tail = f.tail.rows;
        plan_out->nsplit = f.split.nsplit > 1 ? f.split.nsplit : 0;
        plan_out->split_items = f.split.nsplit > 1 ? f.split.split_items : 0;
    }
    if (plan_only) return FA2_OK;
    return launch_fwd(HD, bf16, p, causal != 0, stream, ws, ws_bytes);
}

int fa2_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
            int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
            const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
            float scale, int causal, void* hip_stream) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides, lse_strides,
                    scale, causal, nullptr, FA2_BIAS_NONE, nullptr, hip_stream);
}

int fa2_fwd_ws(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
               int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
               const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
               float scale, int causal, void* workspace, size_t workspace_bytes, void* hip_stream) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides, lse_strides,
                    scale, causal, nullptr, FA2_BIAS_NONE, nullptr, hip_stream, workspace, workspace_bytes);
}

// Stand-in arguments of the plan-only calls: contiguous BHND tensors (the plan looks at the row pitches of Q and K: a contiguous tensor's are
// what the size query must assume — ADVICE r3: stand-in pitches of 8 elements made the query plan for the compiler-scheduled kernels and
// the launch for the hand-scheduled ones).
struct ContigStrides {
    int64_t q[3], k[3], ls[2];
    ContigStrides(int H, int Nq, int Nkv, int D) {
        const int64_t d = D > 0 ? D : 8;
        q[2] = d; q[1] = (int64_t)Nq * d; q[0] = (int64_t)H * q[1];
        k[2] = d; k[1] = (int64_t)Nkv * d; k[0] = (int64_t)H * k[1];
        ls[1] = Nq; ls[0] = (int64_t)H * Nq;
    }
};
alignas(16) static char g_plan_dummy[16];

size_t fa2_fwd_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal) {
    // the plan depends on the shape, the options and the device's CU count only: run the validation + planning half of the call on stand-in arguments
    if (B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1) return 0;
    const ContigStrides cs(H, Nq, Nkv, D);
    size_t need = 0;
    char* d = g_plan_dummy;
    if (fwd_impl(dtype, d, d, d, d, (float*)d, B, H, Nq, Nkv, D, cs.q, cs.k, cs.k, cs.q, cs.ls, (float)(1.0 / std::sqrt((double)D)), causal, nullptr, FA2_BIAS_NONE,
                 nullptr, nullptr, nullptr, 0, &need) != FA2_OK)
        return 0;
    return need;
}

int fa2_fwd_plan(int dtype, int B, int H, int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                 float scale, int causal, int bias_kind, size_t workspace_bytes, fa2_fwd_plan_t* plan) {
    if (!plan) return FA2_ERR_NULL_POINTER;
    if (B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1) return FA2_ERR_BAD_SHAPE;
    const ContigStrides cs(H, Nq, Nkv, D);
    char* d = g_plan_dummy;
    return fwd_impl(dtype, d, d, d, d, (float*)d, B, H, Nq, Nkv, D, q_strides ? q_strides : cs.q, k_strides ? k_strides : cs.k, k_strides ? k_strides : cs.k,
                    q_strides ? q_strides : cs.q, cs.ls, scale, causal, nullptr, bias_kind, nullptr, nullptr, nullptr, workspace_bytes, nullptr, plan);
}

int fa2_fwd_bias(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                 int Nq, int Nkv, int D, const int64_t q_strides[3], const int64_t k_strides[3],
                 const int64_t v_strides[3], const int64_t o_strides[3], const int64_t lse_strides[2],
                 float scale, int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides, lse_strides,
                    scale, causal, bias, bias_kind, bias_strides, hip_stream);
}

static const int64_t* ls3_zero() { static const int64_t z[3] = {0, 0, 0}; return z; }

static int bwd_impl(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
            void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
            const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
            const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
            const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
            int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream,
            void* ws = nullptr, size_t ws_bytes = 0, size_t* ws_need = nullptr) {
    if (causal & ~(FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE)) return FA2_ERR_BAD_SHAPE;      // (as in fwd_impl: no value but the documented flag bits)
    causal &= 1;          // (bit 1, FA2_FLAG_EXACT_SCALE, is what the backward does anyway unless option "kfold" is set)
    if (ws_need) *ws_need = 0;
    if (bias_kind != FA2_BIAS_NONE) {
        if (bias_kind != FA2_BIAS_IO_DTYPE && bias_kind != FA2_BIAS_F32 && bias_kind != FA2_BIAS_BOOL) return FA2_ERR_BIAS;
        if (!bias || !bias_strides) return FA2_ERR_NULL_POINTER;
        if (bias_strides[0] < 0 || bias_strides[1] < 0 || bias_strides[2] < 0) return FA2_ERR_BIAS;
        const uintptr_t esize = bias_kind == FA2_BIAS_F32 ? 4 : bias_kind == FA2_BIAS_IO_DTYPE ? 2 : 1;
        if (reinterpret_cast<uintptr_t>(bias) % esize) return FA2_ERR_ALIGNMENT;
    }
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
    p.bias = bias;
    p.bias_kind = bias_kind;
    for (int i = 0; i < 3; ++i) p.bs[i] = bias_kind != FA2_BIAS_NONE ? bias_strides[i] : 0;
    // bias tiles by LDS-DMA (fa2_bwd_kernel.hip.h: bwd_bias_tile_*): a per-row bias whose pointer, strides and Nkv are multiples of 16 bytes, one
    // (b, h) slice within 32-bit byte offsets; otherwise one guarded load per score
    p.bias_tile = 0;
    p.bias_img = 4096;
    if (bias_kind != FA2_BIAS_NONE) {
        const int64_t esize = bias_kind == FA2_BIAS_F32 ? 4 : bias_kind == FA2_BIAS_IO_DTYPE ? 2 : 1, gran = 16 / esize;
        // the backward addresses one (b, h) slice of the bias with 32-bit byte offsets (bounds-checked buffer loads), like K and V
        if (((int64_t)(Nq - 1) * p.bs[2] + Nkv + 64 * p.bs[2]) * esize >= 0x7fffffffLL) return FA2_ERR_BAD_SHAPE;
        if (bias_kind == FA2_BIAS_F32) p.bias_img = 8192;
        if (Nkv % gran == 0 && reinterpret_cast<uintptr_t>(bias) % 16 == 0 && p.bs[0] % gran == 0 && p.bs[1] % gran == 0 && p.bs[2] % gran == 0 &&
            p.bs[2] != 0)
            p.bias_tile = 1;
    }
    p.full_items = p.split_items = p.nsplit = 0;
    p.ws = (float*)ws;
    p.ws_bytes = ws_bytes;
    if (ws_need) {      // fa2_bwd_workspace_bytes: validate and plan only
        fa2::SplitPlan a, b2;
        *ws_need = (size_t)fa2::plan_bwd_split(HD, p, causal != 0, &a, &b2);
        return FA2_OK;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool bf16 = dtype == FA2_DTYPE_BF16;
    if (bias_kind != FA2_BIAS_NONE)
        return bf16 ? fa2::launch_bwd_bias_hip_bf16(HD, p, causal != 0, stream) : fa2::launch_bwd_bias_hip_f16(HD, p, causal != 0, stream);
    return launch_bwd(HD, bf16, p, causal != 0, stream);
}

int fa2_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
            void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
            const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
            const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
            const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
            int causal, void* hip_stream) {
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides,
                    do_strides, dq_strides, dk_strides, dv_strides, lse_strides, scale, causal, nullptr, FA2_BIAS_NONE, nullptr, hip_stream);
}

int fa2_bwd_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
               void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
               const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
               const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
               const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
               int causal, void* workspace, size_t workspace_bytes, void* hip_stream) {
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides,
                    do_strides, dq_strides, dk_strides, dv_strides, lse_strides, scale, causal, nullptr, FA2_BIAS_NONE, nullptr, hip_stream,
                    workspace, workspace_bytes);
}

size_t fa2_bwd_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal) {
    static const int64_t one[3] = {8, 8, 8};
    alignas(16) static char dummy[16];
    const int64_t ls[2] = {0, 0};
    size_t need = 0;
    if (bwd_impl(dtype, dummy, dummy, dummy, dummy, dummy, (const float*)dummy, dummy, dummy, dummy, (float*)dummy, B, H, Nq, Nkv, D, one, one, one, one,
                 one, one, one, one, ls, 1.0f, causal, nullptr, FA2_BIAS_NONE, nullptr, nullptr, nullptr, 0, &need) != FA2_OK)
        return 0;
    return need;
}

int fa2_bwd_bias_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                    void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
                    const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                    const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                    const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
                    int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* workspace, size_t workspace_bytes,
                    void* hip_stream) {
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides,
                    do_strides, dq_strides, dk_strides, dv_strides, lse_strides, scale, causal, bias, bias_kind, bias_strides, hip_stream,
                    workspace, workspace_bytes);
}

size_t fa2_bwd_bias_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal) {
    static const int64_t one[3] = {8, 8, 8};
    alignas(16) static char dummy[16];
    const int64_t ls[2] = {0, 0};
    size_t need = 0;
    // (the plan does not depend on the bias' kind or strides: any valid description will do)
    if (bwd_impl(dtype, dummy, dummy, dummy, dummy, dummy, (const float*)dummy, dummy, dummy, dummy, (float*)dummy, B, H, Nq, Nkv, D, one, one, one, one,
                 one, one, one, one, ls, 1.0f, causal, dummy, FA2_BIAS_BOOL, ls3_zero(), nullptr, nullptr, 0, &need) != FA2_OK)
        return 0;
    return need;
}

int fa2_bwd_bias(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                 const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                 const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2], float scale,
                 int causal, const void* bias, int bias_kind, const int64_t bias_strides[3], void* hip_stream) {
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, H, Nq, Nkv, D, q_strides, k_strides, v_strides, o_strides,
                    do_strides, dq_strides, dk_strides, dv_strides, lse_strides, scale, causal, bias, bias_kind, bias_strides, hip_stream);
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
