// fa2_fwd_d256.hip.h — HIP shell of the hand-scheduled forward for head dim 256 (round 6; generated body: csrc/gen/fwd_m16_d256_gen.py).
// Reference counterpart: fwd_kernel, kernel_fp16.cu:306-544 (its D > 128 shapes run with Br = 64 / 32 and an fp16 accumulator in LDS).
//
// Workgroup = 4 waves = 128 Q rows, one wave per SIMD, wave = 32 rows; KV tiles of 64 rows of 512 bytes, two-deep K and V rings = 128 KiB of LDS and
// nothing else: Q fragments come straight from memory, O leaves the registers by bounds-checked buffer stores (rows >= Nq fall outside the
// descriptor), the LSE in one register.  One item (head, q block of 128 rows) per workgroup: no seams.  Max-first bodies only, f32 scale, row sums of
// the rounded P on the matrix pipe (FA2_CONTRACT_LSUM_P16).  The launcher (fwd_asm.cpp) hands over calls with D == 256, a positive scale, K / V row
// pitches that are multiples of 512 bytes (the LDS-DMA pieces of a wave are derived from piece 0 by flipping offset bits) and Q / O spans below 4 GiB.
#pragma once
#include "fa2_fwd_d128.hip.h"

namespace fa2 {

constexpr int kD256Rows = 128;                                   // Q rows per workgroup
constexpr int kD256TileB = 64 * 512;                             // one K or V tile image
constexpr int kD256LdsBytes = 4 * kD256TileB + 16;

// (the generated bodies: FA2_D128_INC of fa2_fwd_d128.hip.h — tools/kbench.py points it at schedule variants)
#define FA2_D256_INC(name) FA2_D128_INC(name)

typedef uint32_t d256_u32x4s __attribute__((ext_vector_type(4)));

// TRIM: head dims 136 .. 248 on the same body (generator opt=trim): rows of p.D columns at any pitch, padded columns zero-filled by the loads themselves
// KS (TRIM only): the 32-column k-steps of Q.K^T the body runs, 5 .. 8 for head dims <= 160 / 192 / 224 / 256 (d groups of O: 2 KS) — k-steps and d groups
// without a real column are not in the instruction stream at all: 84 / 100 / 116 / 132 MFMAs per tile
template <bool BF16, bool CAUSAL, bool TRIM, int KS = 8>
__global__ __launch_bounds__(256, 1) void fwd_asm_d256_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bh, qblk;
    block_to_head_qblock<CAUSAL>(p, (int)blockIdx.x, bh, qblk);
    bh = __builtin_amdgcn_readfirstlane(bh);
    qblk = __builtin_amdgcn_readfirstlane(qblk);
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * kD256Rows, qw0 = q0 + 32 * wave;
    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u, o_rowb = (uint32_t)p.os[2] * 2u;
    const uint32_t n16 = lane & 15, g4 = lane >> 4;

    // KV sweep bounds (workgroup: staging + barriers; wave: compute)
    int ntiles = (p.Nkv + kKvTile - 1) / kKvTile;
    if (CAUSAL) {
        const int qmax = (q0 + kD256Rows < p.Nq ? q0 + kD256Rows : p.Nq) - 1;
        const int nt_c = qmax / kKvTile + 1;
        ntiles = nt_c < ntiles ? nt_c : ntiles;
    }
    int ntw = ntiles;
    if (CAUSAL) {
        const int nt_w = (qw0 + 31) / kKvTile + 1;
        ntw = nt_w < ntiles ? nt_w : ntiles;
    }
    // LDS-DMA: piece i of this wave fills image bytes [wave * 8192 + i * 1024, +1024): lane l supplies the source of image slot (row 16 wave + 2 i + l / 32,
    // granule l % 32) — the inverse of the read swizzles: K granule ^ (row & 15); V 64-byte chunk ^ (row & 3), 32-byte half flipped for rows with (row >> 2) & 1
    const uint32_t drow = 16u * wave + (lane >> 5), dslot = lane & 31;
    const uint32_t kd0 = drow * k_rowb + ((dslot ^ (drow & 15u)) << 4);
    const uint32_t vd0 = drow * v_rowb + (((((dslot >> 2) ^ (drow & 3u)) << 2) | ((dslot & 3u) ^ (((drow >> 2) & 1u) << 1))) << 4);
    const uint32_t kr0 = n16 * 512u + ((g4 ^ n16) << 4);
    const uint32_t trow = 4u * g4 + (n16 >> 2);
    const uint32_t vr0 = trow * 512u + ((trow & 3u) << 6) + 32u * (g4 & 1u) + 8u * (n16 & 3u);
    // masks of the wave's last tile: row 16 qg + n keeps kv_local = 16 kg + 4 g4 + i iff 16 kg + i <= min(lim0 + 16 qg, lim1)
    const int lim0 = (CAUSAL ? qw0 + (int)n16 : 0x3fff0000) - kKvTile * (ntw - 1) - 4 * (int)g4;
    const int lim1 = p.Nkv - 1 - kKvTile * (ntw - 1) - 4 * (int)g4;
    const uint32_t q_off = n16 * q_rowb + 16u * g4, o_off = n16 * o_rowb + 8u * g4;

    const uint64_t qa = (uint64_t)((const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1]);
    const uint64_t ka = (uint64_t)((const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1]);
    const uint64_t va = (uint64_t)((const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1]);
    const uint64_t oa = (uint64_t)((uint16_t*)p.o + b * p.os[0] + h * p.os[1]);
    const d256_u32x4s qrs = {(uint32_t)qa, (uint32_t)(qa >> 32) & 0xffffu, (uint32_t)(p.Nq - 1) * q_rowb + 2u * (uint32_t)p.D, 0x00020000u};
    const d256_u32x4s krs = {(uint32_t)ka, (uint32_t)(ka >> 32) & 0xffffu, p.k_bytes, 0x00020000u};
    const d256_u32x4s vrs = {(uint32_t)va, (uint32_t)(va >> 32) & 0xffffu, p.v_bytes, 0x00020000u};
    const d256_u32x4s ors = {(uint32_t)oa, (uint32_t)(oa >> 32) & 0xffffu, (uint32_t)(p.Nq - 1) * o_rowb + 2u * (uint32_t)p.D, 0x00020000u};
    const uint32_t qw = __builtin_amdgcn_readfirstlane((uint32_t)qw0 * q_rowb), ow = __builtin_amdgcn_readfirstlane((uint32_t)qw0 * o_rowb);
    const uint32_t k_tile = kKvTile * k_rowb, v_tile = kKvTile * v_rowb, k_row2 = 2u * k_rowb - 1024u, v_row2 = 2u * v_rowb - 1024u;
    const uint32_t ldsw = wave * (kD256TileB / 4);
    const uint32_t q_t16 = 16u * q_rowb, o_t16 = 16u * o_rowb;
    const float c = p.c;
    const uint32_t ng = (uint32_t)p.D >> 3;
    float lse0, lse1;
#define FA2_D256_OPERANDS                                                                                                     \
    : "=v"(lse0), "=v"(lse1)                                                                                                  \
    : "v"(q_off), "s"(qw), "s"(qrs), "s"(krs), "s"(vrs), "v"(kd0), "v"(vd0), "v"(kr0), "v"(vr0), "v"(lim0), "v"(lim1), "s"(c),   \
      "s"(ntw), "s"(ntiles), "s"(k_tile), "s"(v_tile), "s"(k_row2), "s"(v_row2), "s"(ldsw), "v"(o_off), "s"(ow), "s"(q_t16),     \
      "s"(o_t16), "s"(ors), "s"(ng)                                                                                           \
    :
    if constexpr (BF16 && TRIM && KS == 5) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_bf16_trim5.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (BF16 && TRIM && KS == 6) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_bf16_trim6.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (BF16 && TRIM && KS == 7) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_bf16_trim7.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (BF16 && TRIM && KS == 8) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_bf16_trim8.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (!BF16 && TRIM && KS == 5) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_f16_trim5.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (!BF16 && TRIM && KS == 6) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_f16_trim6.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (!BF16 && TRIM && KS == 7) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_f16_trim7.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (!BF16 && TRIM && KS == 8) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_f16_trim8.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else if constexpr (BF16) {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_bf16.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    } else {
        asm volatile(
#include FA2_D256_INC(fa2_fwd_m16_d256_f16.inc)
            FA2_D256_OPERANDS
#include FA2_D256_INC(fa2_fwd_d128_clobbers.inc)
        );
    }
#undef FA2_D256_OPERANDS
    (void)lse1;
    // the LSE: lane l < 32 hands over row l of the wave
    const int lane2 = threadIdx.x & 63, wave2 = threadIdx.x >> 6;
    const int row = qblk * kD256Rows + 32 * wave2 + lane2;
    if (lane2 < 32 && row < p.Nq) p.lse[b * p.ls[0] + h * p.ls[1] + row] = lse0;
}

}  // namespace fa2
