// fa2_bwd_d128.hip.h — backward kernels for head dim 128 whose tile sweeps are hand-scheduled inline-asm blocks.
//
// Same contract and same math as the compiler-scheduled passes of fa2_bwd_kernel.hip.h (reference: bwd_kernel,
// kernel_fp16.cu:547-740), different machine mapping: 4-wave workgroups with ONE wave per SIMD and the whole 512-entry
// register file per wave, KV (dQ pass) / Q (dK/dV pass) swept in tiles of 32 rows, every LDS fragment read feeding two
// MFMAs, the instruction stream between MFMAs placed by csrc/gen/bwd_d128_gen.py (register maps, pipelines and the LDS
// image formats are documented there) and validated instruction by instruction on the CPU by tools/asm_emu.py
// (tests/test_asm_emu_bwd.py).  This file is the shell: workgroup -> (head, block) mapping, the per-lane addresses the
// blocks take as operands, and the global stores of what they leave in LDS.
#pragma once
#include "fa2_bwd_kernel.hip.h"

#ifndef FA2_D128_INC_DIR
#define FA2_BWD_INC(name) #name
#else
#define FA2_BWD_STR2(x) #x
#define FA2_BWD_STR(x) FA2_BWD_STR2(x)
#define FA2_BWD_INC(name) FA2_BWD_STR(FA2_D128_INC_DIR/name)
#endif

namespace fa2 {

constexpr int kBwdEpiRowB = 272;                       // bytes per staged output row (EPI_ROWB of the generator)
constexpr int kBwdDqEpiBase = 49152;                   // DQ.EPI_BASE: above the row ring (2 x 16 KiB) and the transposed-read ring (2 x 8 KiB)
constexpr int kBwdDqLdsBytes = kBwdDqEpiBase + 4 * 64 * kBwdEpiRowB;      // DQ.LDS_BYTES = 118784
constexpr int kBwdTile = 32;                           // rows per swept tile of the hand-scheduled backward kernels

typedef uint32_t bwd_u32x4s __attribute__((ext_vector_type(4)));

// granule swizzle of the unified LDS image format (f_swz of the generator): one image serves ds_read_b128 row reads and
// ds_read_b64_tr_b16 transposed reads
__device__ __forceinline__ uint32_t bwd_swz(uint32_t r) { return ((r & 3u) << 2) | ((r >> 2) & 3u); }
// ... under the 16x16x32 dK / dV bodies (f_swz16 of the generator, round 6): their lane groups hit every bank twice with the one above
__device__ __forceinline__ uint32_t bwd_swz16(uint32_t r) { return (r & 7u) << 1; }
template <bool M16> __device__ __forceinline__ uint32_t bwd_swz_of(uint32_t r) { return M16 ? bwd_swz16(r) : bwd_swz(r); }

// ---------------------------------------------------------------------------------------------------------
// dQ pass: workgroup = 256 Q rows (4 waves x 64), sweep over KV tiles of 32.  Also writes delta = rowsum(dO * O).
// Requires D == 128 (host.cpp dispatch); any Nq, Nkv (clamped rows, masked tail tiles), causal or not.
// NEG_DELTA: the delta workspace receives -delta (what the hand-scheduled dK/dV pass takes as the C operand of its dP product).
// M16 (round 5): the body built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py): same pipeline, images and operand list; a lane is (n = lane % 16,
// g = lane / 16) there and a Q row is spread over four lanes, so operands 2..9 carry the lane's row of q group 0, 16 g, Nq - 1 and the three row pitches
// instead of eight clamped row offsets, the fragment / epilogue addresses and the mask limits differ, and delta leaves in one register (lane l = row l).
template <bool BF16, bool CAUSAL, bool NEG_DELTA, bool M16 = false>
__global__ __launch_bounds__(256, 1) void bwd_dq_d128_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (head, q block): as in bwd_dq_kernel (causal: longest-first across the heads of an XCD)
    const int nbh = p.B * p.H, bid = blockIdx.x;
    int bh, qblk;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;
        if (CAUSAL) { bh = (bid & 7) + 8 * (slot % hpx); qblk = p.nblk - 1 - slot / hpx; }
        else { bh = (bid & 7) + 8 * (slot / p.nblk); qblk = slot % p.nblk; }
    } else if (CAUSAL) { bh = bid % nbh; qblk = p.nblk - 1 - bid / nbh; }
    else { bh = bid / p.nblk; qblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * 256, qw0 = q0 + 64 * wave;

    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, g_rowb = (uint32_t)p.dos[2] * 2u, o_rowb = (uint32_t)p.os[2] * 2u;
    const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;

    // KV sweep bounds in tiles of 32 (workgroup: staging + barriers; wave: compute)
    const int ntiles = (p.Nkv + kBwdTile - 1) / kBwdTile;
    int ntwg = ntiles, ntw = ntiles;
    if (CAUSAL) {
        const int qmax = (q0 + 256 < p.Nq ? q0 + 256 : p.Nq) - 1;
        const int nt_c = qmax / kBwdTile + 1;
        ntwg = nt_c < ntiles ? nt_c : ntiles;
        const int nt_w = (qw0 + 63) / kBwdTile + 1;
        ntw = nt_w < ntwg ? nt_w : ntwg;
    }

    uint32_t qo[2], go[2], oo[2], lo[2];
    int lim[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = qw0 + 32 * qb + l31;
        const uint32_t qr = (uint32_t)(qrow < p.Nq ? qrow : p.Nq - 1);
        qo[qb] = qr * q_rowb + 16u * hi;
        go[qb] = qr * g_rowb + 16u * hi;
        oo[qb] = qr * o_rowb + 16u * hi;
        lo[qb] = qr * 4u;
        // the wave's last two tiles are masked where kv > min(q row (causal), Nkv - 1); local to the LAST tile, minus this lane half's 4*hi
        const int lim_c = CAUSAL ? qrow : 0x3fffffff;
        lim[qb] = (lim_c < p.Nkv - 1 ? lim_c : p.Nkv - 1) - kBwdTile * (ntw - 1) - 4 * hi;
    }
    const int n16 = lane & 15, g4 = lane >> 4;
    if constexpr (M16) {
        qo[0] = (uint32_t)(qw0 + n16);  qo[1] = 16u * g4;              // ROW0, 16 g
        go[0] = (uint32_t)(p.Nq - 1);   go[1] = q_rowb;                // Nq - 1, Q pitch
        oo[0] = g_rowb;                 oo[1] = o_rowb;                // dO pitch, O pitch
        lo[0] = lo[1] = 0;
        // row 16 qg + n keeps kv_local = 16 kg + 4 g + i of the wave's LAST tile (the one before: + 32) iff 16 kg + i <= min(lim[0] + 16 qg, lim[1])
        lim[0] = (CAUSAL ? qw0 + n16 : 0x3fff0000) - kBwdTile * (ntw - 1) - 4 * g4;
        lim[1] = p.Nkv - 1 - kBwdTile * (ntw - 1) - 4 * g4;
    }
    // LDS-DMA: piece i of this wave fills image bytes [wave*2048 + i*1024, +1024): lane l supplies the source of image slot
    // (row = 8*wave + 4*i + l/16, slot = l%16); the asm block derives piece 1 from piece 0
    const uint32_t drow = 8u * wave + (lane >> 4), dslot = lane & 15;
#ifdef FA2_BWD_DQ_UNI      // developer build (generator opt "uni"): every image in the unified format of the dK/dV pass
    const uint32_t kd0 = drow * k_rowb + ((dslot ^ bwd_swz(drow)) << 4), vd0 = drow * v_rowb + ((dslot ^ bwd_swz(drow)) << 4), td0 = kd0;
    const uint32_t kr0 = (uint32_t)l31 * 256u + (((uint32_t)hi ^ bwd_swz(l31)) << 4);
    const uint32_t pp = lane & 15, g1 = (lane >> 4) & 1, ti = pp >> 2, tj = pp & 3, trow = 4u * hi + ti;
    const uint32_t vr0 = trow * 256u + (((2u * g1 + (tj >> 1)) ^ bwd_swz(trow)) << 4) + 8u * (tj & 1);
#else
    const uint32_t kd0 = drow * k_rowb + ((dslot ^ (drow & 15u)) << 4);                                   // row images: granule ^ (row & 15)
    const uint32_t vd0 = drow * v_rowb + ((dslot ^ (drow & 15u)) << 4);
    // "tr" image: 64-B chunk ^ (row & 3); under the 16x16x32 body (round 6) the 32-byte half of a chunk is also flipped for rows with (row >> 2) & 1:
    // a transposed read serves rows r and r + 4 in one cycle there, which the plain image keeps in the same banks (SQ_LDS_BANK_CONFLICT 26 % of the
    // pass's LDS cycles, profiles/r19_bwd_c2_pmc.txt; the forward's "ct" V image is the same fix)
    const uint32_t td0 = drow * k_rowb + (((((dslot >> 2) ^ (drow & 3u)) << 2) | ((dslot & 3u) ^ (M16 ? ((drow >> 2) & 1u) << 1 : 0u))) << 4);
    const uint32_t pp = lane & 15, g1 = (lane >> 4) & 1;
    const uint32_t trow16 = 4u * g4 + (n16 >> 2);
    const uint32_t kr0 = M16 ? (uint32_t)n16 * 256u + (((uint32_t)g4 ^ (uint32_t)n16) << 4) : (uint32_t)l31 * 256u + (((uint32_t)hi ^ ((uint32_t)l31 & 15u)) << 4);
    const uint32_t vr0 = M16 ? trow16 * 256u + ((trow16 & 3u) << 6) + 32u * ((trow16 >> 2) & 1u) + 8u * (n16 & 3) : (4u * hi + (pp >> 2)) * 256u + ((pp >> 2) << 6) + 32u * g1 + 8u * (pp & 3);
#endif
    const uint32_t epi = M16 ? kBwdDqEpiBase + wave * 64 * kBwdEpiRowB + n16 * kBwdEpiRowB + g4 * 8
                             : kBwdDqEpiBase + wave * 64 * kBwdEpiRowB + l31 * kBwdEpiRowB + hi * 16;

    const uint64_t qbase = (uint64_t)((const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1]);
    const uint64_t gbase = (uint64_t)((const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1]);
    const uint64_t obase = (uint64_t)((const uint16_t*)p.o + b * p.os[0] + h * p.os[1]);
    const uint64_t lbase = (uint64_t)(p.lse + b * p.ls[0] + h * p.ls[1]);
    const uint64_t ka = (uint64_t)((const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1]);
    const uint64_t va = (uint64_t)((const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1]);
    const bwd_u32x4s krs = {(uint32_t)ka, (uint32_t)(ka >> 32) & 0xffffu, p.k_bytes, 0x00020000u};
    const bwd_u32x4s vrs = {(uint32_t)va, (uint32_t)(va >> 32) & 0xffffu, p.v_bytes, 0x00020000u};
    const uint32_t k_tile = kBwdTile * k_rowb, v_tile = kBwdTile * v_rowb, k_row4 = 4 * k_rowb - 1024, v_row4 = 4 * v_rowb - 1024;
    const uint32_t ldsw = wave * 2048;
    const float c = p.c, scale = p.scale, dsign = NEG_DELTA ? -1.0f : 1.0f;
    float d0, d1;

#define FA2_BWD_DQ_OPERANDS                                                                                                     \
    : "=&v"(d0), "=&v"(d1)                                                                                                      \
    : "v"(qo[0]), "v"(qo[1]), "v"(go[0]), "v"(go[1]), "v"(oo[0]), "v"(oo[1]), "v"(lo[0]), "v"(lo[1]), "v"(kd0), "v"(vd0),       \
      "v"(td0), "v"(kr0), "v"(vr0), "v"(lim[0]), "v"(lim[1]), "v"(epi), "s"(qbase), "s"(gbase), "s"(obase), "s"(lbase), "s"(krs), "s"(vrs), \
      "s"(c), "s"(scale), "s"(ntw), "s"(ntwg), "s"(k_tile), "s"(v_tile), "s"(k_row4), "s"(v_row4), "s"(ldsw), "s"(dsign)         \
    :
    if constexpr (M16 && BF16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dq_m16_bf16.inc)
            FA2_BWD_DQ_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dq_d128_clobbers.inc)
        );
    } else if constexpr (M16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dq_m16_f16.inc)
            FA2_BWD_DQ_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dq_d128_clobbers.inc)
        );
    } else if constexpr (BF16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dq_d128_bf16.inc)
            FA2_BWD_DQ_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dq_d128_clobbers.inc)
        );
    } else {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dq_d128_f16.inc)
            FA2_BWD_DQ_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dq_d128_clobbers.inc)
        );
    }
#undef FA2_BWD_DQ_OPERANDS

    // ---- delta out; the wave's 64 x 128 dQ tile is in its LDS image: whole-row stores (4 rows of 256 B per instruction)
    if constexpr (M16) {
        float* dp = p.delta + b * p.ls[0] + h * p.ls[1];
        if (qw0 + lane < p.Nq) dp[qw0 + lane] = d0;                    // one register: lane l hands over row l of the wave
    } else if (hi == 0) {
        float* dp = p.delta + b * p.ls[0] + h * p.ls[1];
        if (qw0 + l31 < p.Nq) dp[qw0 + l31] = d0;
        if (qw0 + 32 + l31 < p.Nq) dp[qw0 + 32 + l31] = d1;
    }
    const char* img = smem + kBwdDqEpiBase + wave * 64 * kBwdEpiRowB;
    const int rl = lane >> 4, cl = lane & 15;
    uint16_t* out = (uint16_t*)p.dq + b * p.dqs[0] + h * p.dqs[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = i * 4 + rl;
        const u32x4 w = *(const u32x4*)(img + r * kBwdEpiRowB + cl * 16);
        if (qw0 + r < p.Nq) *(u32x4*)(out + (int64_t)(qw0 + r) * p.dqs[2] + cl * 8) = w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// dK / dV pass: workgroup = 128 KV rows = two wave pairs (P side + dS side, csrc/gen/bwd_d128_gen.py class KV), sweep over Q tiles of 32.
// Requires D == 128 and Nq % 32 == 0 (host dispatch); any Nkv; causal or not.  Reads -delta from the workspace.
constexpr int kBwdKvLdsBytes = 65536 + 16384;                // KV.LDS_BYTES: Q ring 32 KiB | dO ring 16 KiB | L, -delta 1 KiB (below 64 KiB: LDS-DMA targets) | P slots 16 KiB
constexpr int kBwdKvLdBase = 49152, kBwdKvPSlots = 65536;

// KFOLD: the body whose P side folds scale * log2(e) into its K fragments (rounded once to the I/O dtype) and takes L as the C operand of the S product
// (csrc/gen/bwd_d128_gen.py, option "kfold"; host: option "fold", bwd_folds in host.cpp).
// M16 (round 5): the bodies built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dkv_m16_gen.py): same pipeline, rings, roles and operand list; a lane is (n = lane %
// 16, g = lane / 16) there, the wave's 64 KV rows are four 16-row groups: operands 0, 1, 7, 10 carry the own-row offsets of the four groups, 6 the causal
// limit of group 0 (the asm derives the others and the lane's L / -delta staging offset), the fragment / epilogue addresses differ.
template <bool BF16, bool CAUSAL, bool KFOLD, bool M16 = false>
__global__ __launch_bounds__(256, 1) void bwd_dkv_d128_kernel(const BwdParams p) {
    static_assert(!(M16 && KFOLD), "the 16x16x32 dK / dV bodies scale the f32 scores");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, role = wave & 1;

    // block -> (head, kv block of 128): as in bwd_dkv_pair_kernel (causal: the FIRST kv block sweeps the most Q tiles -> ascending, across heads)
    const int nbh = p.B * p.H, bid = blockIdx.x;
    int bh, kblk;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;
        if (CAUSAL) { bh = (bid & 7) + 8 * (slot % hpx); kblk = slot / hpx; }
        else { bh = (bid & 7) + 8 * (slot / p.nblk); kblk = slot % p.nblk; }
    } else if (CAUSAL) { bh = bid % nbh; kblk = bid / nbh; }
    else { bh = bid / p.nblk; kblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int kv0 = kblk * 128, kvw0 = kv0 + 64 * pair;

    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, g_rowb = (uint32_t)p.dos[2] * 2u;
    const uint32_t f_rowb = (uint32_t)(role ? p.vs[2] : p.ks[2]) * 2u;
    const int tile0 = CAUSAL ? kv0 / kBwdTile : 0;
    int n = p.Nq / kBwdTile - tile0;
    n = n < 1 ? 1 : n;                 // (a causal block whose rows all lie past Nq: one fully masked tile, the results are zeros)

    uint32_t fo[2];
    int lim[2];
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb) {
        const int kvrow = kvw0 + 32 * kvb + l31;
        const uint32_t kr = (uint32_t)(kvrow < p.Nkv ? kvrow : p.Nkv - 1);
        fo[kvb] = kr * f_rowb + 16u * hi;
        lim[kvb] = CAUSAL ? kvrow - kBwdTile * tile0 - 4 * hi : -(1 << 30);
    }
    const uint32_t n16 = lane & 15, g4 = lane >> 4;
    uint32_t fo2 = 0, fo3 = 0;
    if constexpr (M16) {
        uint32_t f[4];
#pragma unroll
        for (int kvg = 0; kvg < 4; ++kvg) {
            const int kvrow = kvw0 + 16 * kvg + (int)n16;
            f[kvg] = (uint32_t)(kvrow < p.Nkv ? kvrow : p.Nkv - 1) * f_rowb + 16u * g4;
        }
        fo[0] = f[0]; fo[1] = f[1]; fo2 = f[2]; fo3 = f[3];
        lim[0] = CAUSAL ? kvw0 + (int)n16 - kBwdTile * tile0 - 4 * (int)g4 : -(1 << 30);
    }
    const uint32_t drow = 8u * wave + (lane >> 4), dslot = lane & 15;
    const uint32_t gd0 = drow * g_rowb + ((dslot ^ bwd_swz_of<M16>(drow)) << 4);
#ifdef FA2_BWD_QSPLIT     // (bodies generated with option "qsplit": of a pair's four Q pieces the P side stages row quad 0, the dS side quads 1..3)
    const uint32_t qrow0 = 16u * pair + (role ? 4u : 0u);
#else
    const uint32_t qrow0 = 8u * wave;
#endif
    const uint32_t qdrow = qrow0 + (lane >> 4);
    const uint32_t qd0 = qdrow * q_rowb + ((dslot ^ bwd_swz_of<M16>(qdrow)) << 4);
    const uint32_t pp = lane & 15, g1 = (lane >> 4) & 1, ti = pp >> 2, tj = pp & 3, trow = 4u * hi + ti;
    const uint32_t tq16 = 4u * g4 + (n16 >> 2);
    const uint32_t kr0 = M16 ? n16 * 256u + ((g4 ^ bwd_swz16(n16)) << 4) : (uint32_t)l31 * 256u + (((uint32_t)hi ^ bwd_swz(l31)) << 4);
    const uint32_t vr0 = M16 ? tq16 * 256u + ((((n16 & 3u) >> 1) ^ bwd_swz16(tq16)) << 4) + 8u * (n16 & 1u)
                             : trow * 256u + (((2u * g1 + (tj >> 1)) ^ bwd_swz(trow)) << 4) + 8u * (tj & 1);
    const uint32_t pxa = kBwdKvPSlots + pair * 8192 + lane * 16, lda = (M16 ? 16u * g4 : 16u * hi) + 512u * role, l4 = M16 ? fo3 : 4u * lane;
    if constexpr (M16) lim[1] = (int)fo2;
    const uint32_t ldm0 = pair == 0 ? kBwdKvLdBase + 512u * role : 0u;        // pair 0's waves stage L (P side) / -delta (dS side) for the workgroup
    const uint32_t epi = M16 ? wave * 64 * kBwdEpiRowB + n16 * kBwdEpiRowB + g4 * 8 : wave * 64 * kBwdEpiRowB + l31 * kBwdEpiRowB + hi * 16;

    const uint64_t fbase = role ? (uint64_t)((const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1]) : (uint64_t)((const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1]);
    const uint64_t qa = (uint64_t)((const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1]);
    const uint64_t ga = (uint64_t)((const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1]);
    const uint64_t la = (uint64_t)((role ? p.delta : p.lse) + b * p.ls[0] + h * p.ls[1]);
    const bwd_u32x4s qrs = {(uint32_t)qa, (uint32_t)(qa >> 32) & 0xffffu, p.q_bytes, 0x00020000u};
    const bwd_u32x4s grs = {(uint32_t)ga, (uint32_t)(ga >> 32) & 0xffffu, p.do_bytes, 0x00020000u};
    const bwd_u32x4s lrs = {(uint32_t)la, (uint32_t)(la >> 32) & 0xffffu, p.l_bytes, 0x00020000u};
    const float c = p.c, oscale = role ? p.scale : 1.0f;
    const uint32_t qoff0 = (uint32_t)tile0 * kBwdTile * q_rowb, goff0 = (uint32_t)tile0 * kBwdTile * g_rowb, loff0 = (uint32_t)tile0 * 128u;
    const uint32_t q_tile = kBwdTile * q_rowb, g_tile = kBwdTile * g_rowb, q_row4 = 4 * q_rowb - 1024, g_row4 = 4 * g_rowb - 1024;
    const uint32_t ldsw = wave * 2048, ldswq = __builtin_amdgcn_readfirstlane(qrow0 * 256u);

#define FA2_BWD_KV_OPERANDS                                                                                                       \
    :                                                                                                                             \
    : "v"(fo[0]), "v"(fo[1]), "v"(qd0), "v"(gd0), "v"(kr0), "v"(vr0), "v"(lim[0]), "v"(lim[1]), "v"(pxa), "v"(lda), "v"(l4), "v"(epi), \
      "s"(fbase), "s"(qrs), "s"(grs), "s"(lrs), "s"(c), "s"(oscale), "s"(n), "s"(qoff0), "s"(goff0), "s"(loff0), "s"(q_tile),     \
      "s"(g_tile), "s"(q_row4), "s"(g_row4), "s"(ldsw), "s"(role), "s"(ldm0), "s"(ldswq)                                          \
    :
    if constexpr (M16 && BF16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_m16_bf16.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    } else if constexpr (M16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_m16_f16.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    } else if constexpr (BF16 && KFOLD) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_d128_bf16_fold.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    } else if constexpr (BF16) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_d128_bf16.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    } else if constexpr (KFOLD) {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_d128_f16_fold.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    } else {
        asm volatile(
#include FA2_BWD_INC(fa2_bwd_dkv_d128_f16.inc)
            FA2_BWD_KV_OPERANDS
#include FA2_BWD_INC(fa2_bwd_dkv_d128_clobbers.inc)
        );
    }
#undef FA2_BWD_KV_OPERANDS

    // ---- the wave's 64 x 128 tile (dV on the P side, dK on the dS side) is in its LDS image: whole-row stores
    const char* img = smem + wave * 64 * kBwdEpiRowB;
    const int rl = lane >> 4, cl = lane & 15;
    uint16_t* out = role ? (uint16_t*)p.dk + b * p.dks[0] + h * p.dks[1] : (uint16_t*)p.dv + b * p.dvs[0] + h * p.dvs[1];
    const int64_t orow = role ? p.dks[2] : p.dvs[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = i * 4 + rl;
        const u32x4 w = *(const u32x4*)(img + r * kBwdEpiRowB + cl * 16);
        if (kvw0 + r < p.Nkv) *(u32x4*)(out + (int64_t)(kvw0 + r) * orow + cl * 8) = w;
    }
}

}  // namespace fa2
