// fwd_asm.cpp — launchers of the hand-scheduled forward kernels (generated inline-asm bodies, csrc/gen/).
// Reference counterpart: the launch at the end of forward_fp16 / forward_bf16 (kernel_fp16.cu:808-851).
#include "fa2_launch.h"

#include "fa2_fwd_d128.hip.h"
#include "fa2_fwd_d256.hip.h"
#include "fa2_gfx950.h"

namespace {

// Persistent workgroups of the d128 kernel (non-causal launches): at most one workgroup per CU, each working through a
// strided list of (head, q block) items and fetching the next item's first tiles while the current one finishes
// (fa2_fwd_d128.hip.h).  Option "persist" = 0 launches one workgroup per item instead (A/B measurements, bit-identity tests).
template <int HD, bool BF16, bool CAUSAL, bool FOLD, bool M16 = false, bool LM = false>
int launch_asm_t(const fa2::FwdParams& p0, hipStream_t stream) {
    constexpr auto kern = fa2::fwd_asm_kernel<HD, BF16, CAUSAL, FOLD, M16, LM>;
    constexpr int lds = fa2::AsmGeo<HD>::LDS_BYTES;
    if (int rc = fa2::set_lds<kern>(lds)) return rc;
    fa2::FwdParams p = p0;
    p.persist = fa2::options().persist.load(std::memory_order_relaxed) ? 1 : 0;
    const int pg = fa2::device_cus() & ~7;       // a multiple of 8: a unit stays on its head's XCD
    if (CAUSAL && p.persist) {
        // Causal pair units against one item per workgroup in longest-first order, same box (tools/fwd_ab.py, MI355X): B8 H16 N4096 +5.2 %,
        // B1 H8 N16384 +1.4 %, B1 H32 N8192 +0.7 %, but B2 H16 N4096 -2.2 % and B4 H16 N2048 -2.0 %: with a single short unit per workgroup
        // the hardware's dynamic dispatch of single items is the better balancer.  Pairs when a workgroup gets several units or a head has >= 32 blocks.
        // Round 5 (the next item's Q staged through LDS during the item, read at the seam): a grid of exactly one unit per CU is ahead with pairs too —
        // B2 H16 N4096 bf16 112.7 -> 112.1 us, B4 H16 N2048 70.0 -> 69.4 (profiles/r16_kbench_pairs_ab.txt); fewer units than CUs keep single items
        // (twice the workgroups: more of the chip busy).
        const int64_t units = (int64_t)p.nbh * ((p.nqblk + 1) / 2);
#if defined(FA2_PAIRS_ROUND4)   // (developer A/B, tools/kbench.py build nopairs:-DFA2_PAIRS_ROUND4=1,only=fwd_asm: round 4's rule)
        if (!(units > pg || p.nqblk >= 32)) p.persist = 0;
#elif !defined(FA2_PAIRS_ALWAYS)
        if (!(units >= pg || p.nqblk >= 32)) p.persist = 0;
#endif
    }
    // work units: non-causal one per (head, q block); causal one per PAIR of q blocks of a head (fa2_fwd_d128.hip.h)
    const int64_t per_head = (CAUSAL && p.persist) ? (p.nqblk + 1) / 2 : p.nqblk;
    int64_t grid = (int64_t)p.nbh * per_head;
    if (!CAUSAL && p.item_cap > 0) grid = p.item_cap;
    if (p.persist && pg > 0 && grid > pg) grid = pg;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return (int)hipGetLastError();
}

// head dim 256 (round 6): 128-row workgroups, one item each (fa2_fwd_d256.hip.h)
template <bool BF16, bool CAUSAL, bool TRIM, int KS = 8>
int launch_d256_t(const fa2::FwdParams& p0, hipStream_t stream) {
    constexpr auto kern = fa2::fwd_asm_d256_kernel<BF16, CAUSAL, TRIM, KS>;
    if (int rc = fa2::set_lds<kern>(fa2::kD256LdsBytes)) return rc;
    fa2::FwdParams p = p0;
    p.nqblk = (p.Nq + fa2::kD256Rows - 1) / fa2::kD256Rows;
    const int64_t grid = (int64_t)p.nbh * p.nqblk;
    if (grid > 0x7fffffffLL) return FA2_ERR_GRID;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), fa2::kD256LdsBytes, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

namespace fa2 {

int launch_fwd_asm_d256(bool bf16, const FwdParams& p, bool causal, hipStream_t stream) {
    if (p.D < 256) {       // head dims 136 .. 248: the general form of the offsets, any row pitch; the body runs ceil(D / 32) k-steps
        const int ks = (p.D + 31) / 32;
#define FA2_D256_TRIM_CASE(K)                                                                                                              \
        if (ks == K) {                                                                                                                 \
            if (bf16) return causal ? launch_d256_t<true, true, true, K>(p, stream) : launch_d256_t<true, false, true, K>(p, stream);   \
            return causal ? launch_d256_t<false, true, true, K>(p, stream) : launch_d256_t<false, false, true, K>(p, stream);           \
        }
        FA2_D256_TRIM_CASE(5) FA2_D256_TRIM_CASE(6) FA2_D256_TRIM_CASE(7) FA2_D256_TRIM_CASE(8)
#undef FA2_D256_TRIM_CASE
        return FA2_ERR_HEAD_DIM;
    }
    if (bf16) return causal ? launch_d256_t<true, true, false>(p, stream) : launch_d256_t<true, false, false>(p, stream);
    return causal ? launch_d256_t<false, true, false>(p, stream) : launch_d256_t<false, false, false>(p, stream);
}

template <int HD, bool BF16>
static int launch_asm_hd(const FwdParams& p, bool causal, bool fold, hipStream_t stream) {
    if (fold) return causal ? launch_asm_t<HD, BF16, true, true>(p, stream) : launch_asm_t<HD, BF16, false, true>(p, stream);
    return causal ? launch_asm_t<HD, BF16, true, false>(p, stream) : launch_asm_t<HD, BF16, false, false>(p, stream);
}

int launch_fwd_asm(int HD, bool bf16, const FwdParams& p, bool causal, bool fold, hipStream_t stream, int m16) {
    // The v_mfma_f32_16x16x32 bodies: head dim 128.  Measured against the 32x32x16 bodies on one box (tools/kbench.py,
    // profiles/r16_kbench_m16_*.txt): folded scale fp16 c2 +4.7 %, c4 +4.6 %, B8 +3.8 %; f32 scale fp16 +3.3 % / +3.3 % / +1.1 %; f32 scale bf16
    // -0.5 .. +0.5 % (c3, c2-shape, B8: the bf16 32x32x16 MFMA is the cheaper one to begin with, profiles/mfma_peak.json) -> those stay where they were.
    // The folded 16 x 16 bodies derive the LDS-DMA source offsets of the odd V pieces by flipping bit 5 of the byte offset (their V image keeps the
    // 32-byte halves of those rows flipped): the same as re-deriving the granule only when a row's byte offset has that bit clear — V's row pitch must
    // be a multiple of 64 bytes (contiguous BHND / BNHD tensors are; a row-padded V is not and keeps the 32 x 32 body: found by tools/fuzz_parity.py,
    // profiles/fuzz_runs.md row r17_fuzz_rows256_seed503).
    // Round 5, later: the folded 16 x 16 bodies keep their row sums on the matrix pipe (gen opt=lm: 8 MFMAs for 64 v_add_f32 per tile, and fast bodies
    // that are exp + pack only): another -4.0 .. 4.7 % (profiles/r18_kbench_lm_windows.txt); contract FA2_CONTRACT_LSUM_P16 (plan_range, host.cpp).
    // Head dim 64 (same generator, hd = 64: 2 k-steps, 4 d groups, a 72-gap body): the folded body with the row sums on the matrix pipe against the
    // 32x32x16 folded body, one box (profiles/r18_kbench_d64_m16.txt): fp16 B2 H16 N4096 136.0 -> 123.7 us, B1 H24 N5120 151.1 -> 137.1 (+10 %);
    // the f32-scale 16 x 16 body (bf16 causal) LOSES 7 % there (75.4 -> 80.9 us: 64 more v_fma_f32 and the adds of the sum check on a body that is
    // VALU-bound to begin with) — those launches stay on the 32x32x16 body.
    // The f32-scale 16 x 16 bodies with the row sums on the matrix pipe (gen opt=lm without ct: constants in a[224:255], K fragments in the 32-register
    // pool): every head-dim-128 launch that scales the f32 product and is not flagged FA2_FLAG_EXACT_SCALE — bf16 by default.  Against the routing
    // before (bf16 on the 32x32x16 body), one box (profiles/r18_kbench_f32lm.txt, _c3.txt): bf16 B2 H16 N4096 non-causal 199.1 -> 187.8 us, causal (c3)
    // 110.0 -> 106.7, B1 H32 N8192 causal 393.8 -> 373.2, B8 797 -> 764; head dim 64 bf16 causal 74.9 -> 74.7 (not taken).
    const int kind = fwd_asm_m16_kind(HD, bf16, p, fold, m16);
    if (kind == kM16F32Lm) {
        if (HD == 64) {
            if (bf16) return causal ? launch_asm_t<64, true, true, false, true, true>(p, stream) : launch_asm_t<64, true, false, false, true, true>(p, stream);
            return causal ? launch_asm_t<64, false, true, false, true, true>(p, stream) : launch_asm_t<64, false, false, false, true, true>(p, stream);
        }
        if (bf16) return causal ? launch_asm_t<128, true, true, false, true, true>(p, stream) : launch_asm_t<128, true, false, false, true, true>(p, stream);
        return causal ? launch_asm_t<128, false, true, false, true, true>(p, stream) : launch_asm_t<128, false, false, false, true, true>(p, stream);
    }
    if (HD == 64 && kind != kM16None) {          // (kM16Fold)
        if (bf16) return causal ? launch_asm_t<64, true, true, true, true, true>(p, stream) : launch_asm_t<64, true, false, true, true, true>(p, stream);
        return causal ? launch_asm_t<64, false, true, true, true, true>(p, stream) : launch_asm_t<64, false, false, true, true, true>(p, stream);
    }
    if (kind == kM16FoldNoLm) {                  // option "asm" bit 9 clear: the folded sum-check bodies (in-place repair instead of the item redo)
        if (bf16) return causal ? launch_asm_t<128, true, true, true, true>(p, stream) : launch_asm_t<128, true, false, true, true>(p, stream);
        return causal ? launch_asm_t<128, false, true, true, true>(p, stream) : launch_asm_t<128, false, false, true, true>(p, stream);
    }
    if (kind != kM16None) {
        if (fold) {
            if (bf16) return causal ? launch_asm_t<128, true, true, true, true, true>(p, stream) : launch_asm_t<128, true, false, true, true, true>(p, stream);
            return causal ? launch_asm_t<128, false, true, true, true, true>(p, stream) : launch_asm_t<128, false, false, true, true, true>(p, stream);
        }
        if (bf16) return causal ? launch_asm_t<128, true, true, false, true>(p, stream) : launch_asm_t<128, true, false, false, true>(p, stream);
        return causal ? launch_asm_t<128, false, true, false, true>(p, stream) : launch_asm_t<128, false, false, false, true>(p, stream);
    }
    if (HD == 128) return bf16 ? launch_asm_hd<128, true>(p, causal, fold, stream) : launch_asm_hd<128, false>(p, causal, fold, stream);
    return bf16 ? launch_asm_hd<64, true>(p, causal, fold, stream) : launch_asm_hd<64, false>(p, causal, fold, stream);
}

}  // namespace fa2
