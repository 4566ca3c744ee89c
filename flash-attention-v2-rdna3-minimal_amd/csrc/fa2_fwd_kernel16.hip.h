// fa2_fwd_kernel16.hip.h — the 8-wave forward kernel of fa2_fwd_kernel.hip.h on v_mfma_f32_16x16x32 instead of 32x32x16.
//
// Why: the chip is power-limited on this path and a loop of nothing but MFMAs sustains 13 % more FLOP/s with the
// 16x16x32 form on the benchmark's operands (profiles/mfma_peak.json: 1 955-1 969 vs 1 708-1 738 TF) — but only with two waves per SIMD,
// which is this kernel's shape (the hand-scheduled D = 128 body runs one wave per SIMD and cannot use it).  Same contract, same math, same
// LDS images, staging, block mapping and software pipeline as fwd_kernel<HD, HD, BF16, CAUSAL, 8, 1> (reference counterpart: fwd_kernel,
// rocwmma_fattn/kernel_fp16.cu:306-544); what changes is the register layout of S, P and O:
//
//   wave w owns Q rows [32w, 32w+32) as two 16-row blocks sb = 0, 1; lane = (c = lane & 15, g = lane >> 4)
//   S^T block [16 kv x 16 q]  = K frag (A: row 16kb + c, d = 32ks + 8g..+7, one ds_read_b128) x Q frag (B: q = 16sb + c, same d slice)
//        C/D layout of a 16x16 tile: column j = c, rows i = 4g + r (r = 0..3)  ->  lane (c, g) holds kv = 16kb + 4g + r of Q row c:
//        row max / row sum are in-lane chains over 16 values plus TWO exchanges (v_permlane16_swap, v_permlane32_swap) across g
//   O^T block [16 d x 16 q]   = V^T frag (A: d = 16db + c, kv slots e) x P frag (B: q = c, kv slots e); k-step kk covers kv 32kk..32kk+31 with
//        slot e of lane group g bound to kv = 32kk + 4g + e (e < 4) / 32kk + 16 + 4g + (e - 4): exactly how two S^T blocks leave the QK^T
//        MFMA, so P still needs no cross-lane movement; the V^T fragment with the same binding is two ds_read_b64_tr_b16
//   per wave and KV tile: 32 + 32 MFMAs of 16 matrix-pipe cycles (was 16 + 16 of 32), the same 16 + 32 LDS fragment reads, each feeding two MFMAs.
#pragma once
#include "fa2_fwd_kernel.hip.h"

#ifndef FA2_IGLP16           // scheduler hint of this kernel's steady-state step (see FA2_IGLP)
#define FA2_IGLP16 FA2_IGLP
#endif

namespace fa2 {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool BF16>
__device__ __forceinline__ f32x4v mfma32(u32x4 a, u32x4 b, f32x4v c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// reductions over the four lanes that share lane & 15 (the four 16-lane rows of the wave)
__device__ __forceinline__ float rows4_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int HD, bool BF16, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void fwd_kernel16(const FwdParams p) {
    static_assert(HD == 64 || HD == 128, "fwd_kernel16 is instantiated for the one-slab head dims");
    constexpr int NW = 8, SB = 2;                 // waves per workgroup; 16-row Q blocks per wave
    constexpr int kRowsPerBlock = NW * 32;
    using G_ = Geo<HD, NW>;
    constexpr int ROWB = G_::ROWB, TILEB = G_::TILEB, NPASS = G_::NPASS;
    constexpr int KS = HD / 32;                   // QK^T k-steps of 32
    constexpr int DB = HD / 16;                   // 16-wide d blocks of O
    constexpr int KB = kKvTile / 16;              // 16-row kv blocks of a tile
    constexpr int VBASE = 2 * TILEB;              // LDS: K buf0 | K buf1 | V buf0 | V buf1
    constexpr int kThreads = NW * 64;
    // V image of this kernel: the 32-byte segment sg of row r sits at segment sg ^ ((r / RPB) & (NSEG - 1)) of its row, so that the
    // 8 rows x 32 bytes a transpose read of 32 lanes touches (rows 4g + (c >> 2), g = 0, 1) fall into 8 different 32-byte bank groups
    // (fwd_kernel's 64-byte swizzle serves 4 rows x 64 bytes; with it these reads are 2-way conflicted)
    constexpr int NSEG = ROWB / 32, RPBV = 256 / ROWB > 0 ? 256 / ROWB : 1;
    auto v_off16 = [](int row, int colbyte) __attribute__((always_inline)) {
        return row * ROWB + ((((colbyte >> 5) ^ ((row / RPBV) & (NSEG - 1)))) << 5) + (colbyte & 31);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15;
    const int g = lane >> 4;

    int bh, qblk;
    block_to_head_qblock<CAUSAL>(p, blockIdx.x, bh, qblk);
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * kRowsPerBlock;
    const int qw0 = q0 + wave * 32;
    int qrow[SB];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) qrow[sb] = qw0 + 16 * sb + c16;

    // ---- Q fragments (B operand): lane reads 8 consecutive d of its row per k-step of 32
    u32x4 qf[SB][KS];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        const int qr = qrow[sb] < p.Nq ? qrow[sb] : p.Nq - 1;
        const uint16_t* qp = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qr * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[sb][ks] = (32 * ks + 8 * g < p.D) ? *(const u32x4*)(qp + 32 * ks + 8 * g) : (u32x4){0u, 0u, 0u, 0u};
        if (p.negate_q) {
            const uint32_t sgn = 0x80008000u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[sb][ks] ^= (u32x4){sgn, sgn, sgn, sgn};
        }
    }

    // ---- K/V staging: exactly as fwd_kernel (buffer descriptors, LDS-DMA at D >= 128, register staging below)
    const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t* vbase = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1];
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes, 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, p.v_bytes, 0x00020000);
    const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;
    constexpr bool kDma = FA2_LDS_DMA && HD >= FA2_LDS_DMA_MIN_HD;
    uint32_t kg_off[NPASS], vg_off[NPASS], kd_off[NPASS], vd_off[NPASS];
    int kw_off[NPASS], vw_off[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / G_::G, gi = idx % G_::G;
        kg_off[i] = gi * 8 < p.D ? row * k_rowb + gi * 16 : kOobOffset;
        vg_off[i] = gi * 8 < p.D ? row * v_rowb + gi * 16 : kOobOffset;
        kw_off[i] = G_::k_off(row, gi);
        vw_off[i] = v_off16(row, gi * 16);
        const int gk = gi ^ ((row / G_::RPB) & G_::KMASK);                                   // DMA: source granule of image slot gi
        const int gv = ((((gi >> 1) ^ ((row / RPBV) & (NSEG - 1)))) << 1) | (gi & 1);
        kd_off[i] = gk * 8 < p.D ? row * k_rowb + gk * 16 : kOobOffset;
        vd_off[i] = gv * 8 < p.D ? row * v_rowb + gv * 16 : kOobOffset;
    }
    u32x4 kreg[NPASS], vreg[NPASS];
    auto load_k = [&](int tile, int buf) __attribute__((always_inline)) {
        const uint32_t soff = (uint32_t)tile * kKvTile * k_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if constexpr (kDma) dma16_to_lds(krs, smem + buf * TILEB + (wave * 64 + kThreads * i) * 16, FA2_TILE_OFF(kd_off[i], soff));
            else kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, FA2_TILE_OFF(kg_off[i], soff), 0);
        }
    };
    auto load_v = [&](int tile, int buf) __attribute__((always_inline)) {
        const uint32_t soff = (uint32_t)tile * kKvTile * v_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if constexpr (kDma) dma16_to_lds(vrs, smem + VBASE + buf * TILEB + (wave * 64 + kThreads * i) * 16, FA2_TILE_OFF(vd_off[i], soff));
            else vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, FA2_TILE_OFF(vg_off[i], soff), 0);
        }
    };
    auto write_k = [&](int buf) __attribute__((always_inline)) {
        if constexpr (!kDma) {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) *(u32x4*)(smem + buf * TILEB + kw_off[i]) = kreg[i];
        }
    };
    auto write_v = [&](int buf) __attribute__((always_inline)) {
        if constexpr (!kDma) {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) *(u32x4*)(smem + VBASE + buf * TILEB + vw_off[i]) = vreg[i];
        }
    };

    // ---- per-lane LDS read offsets (the swizzle terms of k_off / v_off do not change when 16 or 32 rows are added)
    int kr_off[KS];   // K fragment: row c of a 16-row kv block, granule 4ks + g
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kr_off[ks] = G_::k_off(c16, 4 * ks + g);
    // V^T fragment by transpose read: this lane addresses row 4g + (c >> 2), columns 16db + 4(c & 3) .. +3 = 8 bytes of segment db
    const int vrow = 4 * g + (c16 >> 2);
    const int vr_base = vrow * ROWB + 8 * (c16 & 3), vr_x = (vrow / RPBV) & (NSEG - 1);
    auto vr_off = [&](int db) __attribute__((always_inline)) { return vr_base + ((db ^ vr_x) << 5); };

    // ---- KV sweep bounds
    int ntiles = (p.Nkv + kKvTile - 1) / kKvTile;
    if (CAUSAL) {
        const int qmax = (q0 + kRowsPerBlock < p.Nq ? q0 + kRowsPerBlock : p.Nq) - 1;
        const int nt_c = qmax / kKvTile + 1;
        ntiles = nt_c < ntiles ? nt_c : ntiles;
    }
    int ntiles_w = ntiles;
    if (CAUSAL) {
        const int nt_w = (qw0 + 31) / kKvTile + 1;
        ntiles_w = nt_w < ntiles ? nt_w : ntiles;
    }

    f32x4v acc[SB][DB];
    float m_run[SB], l_run[SB];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        m_run[sb] = -INFINITY;
        l_run[sb] = 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db) acc[sb][db] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    const float c = p.c;

    // S^T = K Q^T: KB x SB blocks of 16 x 16; each K fragment read from LDS feeds SB MFMAs
    auto qk = [&](int buf, f32x4v (&s)[SB][KB]) __attribute__((always_inline)) {
        const char* kt = smem + buf * TILEB;
#pragma unroll
        for (int sb = 0; sb < SB; ++sb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) s[sb][kb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const u32x4 a = *(const u32x4*)(kt + kr_off[ks] + 16 * kb * ROWB);
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) s[sb][kb] = mfma32<BF16>(a, qf[sb][ks], s[sb][kb]);
            }
    };

    auto finish_scores = [&](int tile, auto masked, f32x4v (&s)[SB][KB]) __attribute__((always_inline)) {
        float mx[SB];
        bool grow = FA2_DEFER_THR < 0.f;
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
            if constexpr (decltype(masked)::value) {
                const int kv0 = tile * kKvTile;
                const bool need_causal = CAUSAL && (kv0 + kKvTile - 1 > qw0 + 16 * sb);
                const bool need_tail = kv0 + kKvTile > p.Nkv;
                if (need_causal || need_tail) {
                    const int lim_c = CAUSAL ? qrow[sb] : 0x7fffffff;
                    int lim = lim_c < p.Nkv - 1 ? lim_c : p.Nkv - 1;
                    int kvb = kv0 + 4 * g;
                    asm volatile("" : "+v"(lim), "+v"(kvb));     // opaque: or LICM hoists the tile-independent tail masks out of the sweep
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kvb + 16 * kb + r > lim) s[sb][kb][r] = -INFINITY;
                }
            }
            float m = max3(s[sb][0][0], s[sb][0][1], s[sb][0][2]);
            m = max3(m, s[sb][0][3], s[sb][1][0]);
            m = max3(m, s[sb][1][1], s[sb][1][2]);
            m = max3(m, s[sb][1][3], s[sb][2][0]);
            m = max3(m, s[sb][2][1], s[sb][2][2]);
            m = max3(m, s[sb][2][3], s[sb][3][0]);
            m = max3(m, s[sb][3][1], s[sb][3][2]);
            m = __builtin_fmaxf(m, s[sb][3][3]);
            mx[sb] = rows4_max(m);
            if (!(FA2_DEFER_THR < 0.f))
                grow = grow || (__builtin_amdgcn_ballot_w64((mx[sb] - m_run[sb]) * c > FA2_DEFER_THR) != 0);
        }
        if (grow) {
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                const float m_new = __builtin_fmaxf(m_run[sb], mx[sb]);
                const float alpha = __builtin_amdgcn_exp2f((m_run[sb] - m_new) * c);
                m_run[sb] = m_new;
                l_run[sb] *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db) acc[sb][db] *= alpha;
            }
        }
    };

    // P = 2^(S*c - m*c), this lane's part of the row sum, P -> 16-bit B fragments: k-step kk = blocks 2kk (slots 0..3) and 2kk+1 (4..7)
    auto exp_scores = [&](f32x4v (&s)[SB][KB], u32x4 (&pf)[SB][KB / 2]) __attribute__((always_inline)) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
            const float mc = m_run[sb] * c;
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[sb][kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sb][kb][r], c, -mc));
                    if (r & 1) rs1 += s[sb][kb][r];
                    else rs0 += s[sb][kb][r];
                }
            l_run[sb] += rs0 + rs1;
#pragma unroll
            for (int kk = 0; kk < KB / 2; ++kk)
                pf[sb][kk] = (u32x4){pack2<BF16>(s[sb][2 * kk][0], s[sb][2 * kk][1]), pack2<BF16>(s[sb][2 * kk][2], s[sb][2 * kk][3]),
                                     pack2<BF16>(s[sb][2 * kk + 1][0], s[sb][2 * kk + 1][1]), pack2<BF16>(s[sb][2 * kk + 1][2], s[sb][2 * kk + 1][3])};
        }
    };

    // O^T += V^T P^T; each V^T fragment (two transpose reads) feeds SB MFMAs
    auto pv = [&](int buf, const u32x4 (&pf)[SB][KB / 2]) __attribute__((always_inline)) {
        const char* vt = smem + VBASE + buf * TILEB;
#pragma unroll
        for (int kk = 0; kk < KB / 2; ++kk)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* va = vt + vr_off(db) + 32 * kk * ROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 16 * ROWB));
                const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi4);
                const u32x4 a = (u32x4){lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) acc[sb][db] = mfma32<BF16>(a, pf[sb][kk], acc[sb][db]);
            }
    };

    // One pipeline step: as fwd_kernel (MODE 1 = branch-free steady state, MODE 0 = generic)
    auto step = [&](int tile, auto par, auto mode, f32x4v (&sc)[SB][KB], f32x4v (&sn)[SB][KB]) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr int MODE = decltype(mode)::value;
        constexpr bool FAST = MODE != 0;
        const bool more1 = FAST || tile + 1 < ntiles, more2 = FAST || tile + 2 < ntiles;
        const bool next_w = FAST || tile + 1 < ntiles_w, cur_w = FAST || tile < ntiles_w;
#if FA2_IGLP16 >= 0
        if constexpr (FAST) __builtin_amdgcn_iglp_opt(FA2_IGLP16);
#endif
        if (more2) load_k(tile + 2, PAR);
        if (more1) load_v(tile + 1, PAR ^ 1);
        if (next_w) qk(PAR ^ 1, sn);
        if (cur_w) {
            u32x4 pf[SB][KB / 2];
            exp_scores(sc, pf);
            pv(PAR, pf);
        }
        if (more2) write_k(PAR);
        if (more1) write_v(PAR ^ 1);
        __syncthreads();
        if (next_w) finish_scores(tile + 1, std::integral_constant<bool, MODE != 1>{}, sn);
    };

    // ---- prologue
    load_k(0, 0);
    load_v(0, 0);
    write_k(0);
    write_v(0);
    if (ntiles > 1) { load_k(1, 1); write_k(1); }
    __syncthreads();
    f32x4v sa[SB][KB], sb_[SB][KB];
    qk(0, sa);
    __syncthreads();
    finish_scores(0, std::true_type{}, sa);

    int n_fast = ntiles - 2 < ntiles_w - 1 ? ntiles - 2 : ntiles_w - 1;
    {
        const int unmasked_kv = p.Nkv / kKvTile;
        const int unmasked_c = CAUSAL ? (qw0 + 1) / kKvTile : 0x7fffffff;
        const int unmasked = unmasked_kv < unmasked_c ? unmasked_kv : unmasked_c;
        n_fast = n_fast < unmasked - 1 ? n_fast : unmasked - 1;
        n_fast = n_fast < 0 ? 0 : n_fast & ~1;
    }
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::integral_constant<int, 0> GENERIC{};
    constexpr std::integral_constant<int, 1> STEADY{};
    int tile = 0;
    for (; tile < n_fast; tile += 2) {
        step(tile, P0, STEADY, sa, sb_);
        step(tile + 1, P1, STEADY, sb_, sa);
    }
    for (; tile + 1 < ntiles; tile += 2) {
        step(tile, P0, GENERIC, sa, sb_);
        step(tile + 1, P1, GENERIC, sb_, sa);
    }
    if (tile < ntiles) step(tile, P0, GENERIC, sa, sb_);

    // ---- epilogue: O = O / l through a wave-private LDS image (whole-row stores), lse = m*c + log2(l)
    constexpr int EROW = HD * 2 + 16;
    constexpr int LPR = HD * 2 / 16;
    constexpr int RPI = 64 / LPR;
    __syncthreads();
    char* img = smem + wave * (32 * EROW);
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        const float l_tot = rows4_sum(l_run[sb]);
        const float inv_l = 1.0f / l_tot;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4v a = acc[sb][db] * inv_l;       // d = 16db + 4g + r of row 16sb + c
            *(u32x2*)(img + (16 * sb + c16) * EROW + (16 * db + 4 * g) * 2) = (u32x2){pack2<BF16>(a[0], a[1]), pack2<BF16>(a[2], a[3])};
        }
        if (qrow[sb] < p.Nq && g == 0) p.lse[b * p.ls[0] + h * p.ls[1] + qrow[sb]] = m_run[sb] * c + __builtin_amdgcn_logf(l_tot);
    }
    const int rl = lane / LPR, cl = lane % LPR;
    uint16_t* obase = (uint16_t*)p.o + b * p.os[0] + h * p.os[1];
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + rl;
        const u32x4 w = *(const u32x4*)(img + r * EROW + cl * 16);
        if (qw0 + r < p.Nq && cl * 8 < p.D) *(u32x4*)(obase + (int64_t)(qw0 + r) * p.os[2] + cl * 8) = w;
    }
}

}  // namespace fa2
