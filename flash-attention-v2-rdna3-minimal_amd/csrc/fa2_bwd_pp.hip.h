// fa2_bwd_pp.hip.h — "ping-pong" form of the fused dK / dV pass for head dims <= 64 (round 3, late).
//
// Same contract, arithmetic and output owner as bwd_dkv_kernel<HD, BF16, CAUSAL, true, 8, true> (fa2_bwd_kernel.hip.h; reference counterpart
// bwd_kernel, rocwmma_fattn/kernel_fp16.cu:547-740): workgroup = 256 KV rows (8 waves x 32, lane = KV row), sweep over Q tiles of 64, S and P
// formed once for dK and dV.  What changes is WHEN the two waves of a SIMD do what.  In the plain kernel every wave runs
//     S, dP (16 MFMAs)  ->  P, dS (150 VALU, 32 of them v_exp)  ->  dV, dK (16 MFMAs)  ->  __syncthreads
// and the hardware's fair arbitration keeps the two waves of a SIMD in the same phase: rocprofv3 showed 197 k cycles of matrix pipe and 242 k of
// VALU per SIMD in 421 k, 71 k of them overlapped (profiles/r06_experiments.txt, item 9).  Here the body is two PHASES with a barrier after each,
//     M(t): dV, dK of tile t-1, then S, dP of tile t   (32 MFMAs, every LDS fragment read)
//     V(t): P, dS of tile t, packed into the fragments M(t+1) consumes; the LDS-DMA of tile t+2
// and waves 4..7 pass ONE barrier more than waves 0..3 before the sweep (and one fewer after it), so that at any time one wave of a SIMD is in
// its matrix phase and the other in its VALU phase.  Tiles live in a ring of FOUR stages: tile t+2 is requested while t-1 may still be read by the
// group that is half a tile behind.  The staging loads are inline asm (buffer_load ... lds with M0 set beside them): the compiler does not see
// them, so it places no conservative vmcnt(0) in front of the transposed LDS reads, and the one wait per tile (end of the M phase, a whole
// tile after the request) is written here.
#pragma once
#include "fa2_bwd_kernel.hip.h"

namespace fa2 {

__device__ __forceinline__ void pp_dma16(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ void pp_dma4(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ u32x4 pp_rsrc(const void* base, uint32_t bytes) {      // the descriptor __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000) makes
    const uint64_t a = (uint64_t)(uintptr_t)base;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// barrier between two phases: nothing moves across it, and (unlike __syncthreads) it does not wait for the staging loads in flight
__device__ __forceinline__ void pp_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

constexpr int kPpStages = 4;

template <int HD, bool BF16, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void bwd_dkv_pp_kernel(const BwdParams p) {
    using L_ = BwdLane<HD, 8>;
    constexpr int kRows = 256;
    constexpr int NPASS = L_::NPASS, KS = L_::KS, DT = L_::DT, ROWB = L_::ROWB, TILEB = L_::TILEB;
    constexpr int LOFF = 4 * TILEB;                  // stage: Q row | dO row | Q tr | dO tr | L[64] | D[64]
    constexpr int STAGEB = LOFF + 512;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const lds_char_ptr smem = (lds_char_ptr)smem_generic;
    const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD

    // block -> (head, kv block), split parts: as bwd_dkv_kernel
    const int nbh = p.B * p.H;
    int bid = blockIdx.x;
    int part = -1, sidx = 0;
    if constexpr (!CAUSAL) {
        if (p.nsplit > 1 && bid >= p.full_items) {
            const int j = bid - p.full_items;
            part = j / p.split_items;
            sidx = j % p.split_items;
            bid = p.full_items + sidx;
        }
    }
    int bh, kblk;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;
        if (CAUSAL) { bh = (bid & 7) + 8 * (slot % hpx); kblk = slot / hpx; }
        else { bh = (bid & 7) + 8 * (slot / p.nblk); kblk = slot % p.nblk; }
    } else if (CAUSAL) { bh = bid % nbh; kblk = bid / nbh; }
    else { bh = bid / p.nblk; kblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int kv0 = kblk * kRows, kvw0 = kv0 + 32 * wave, kvrow = kvw0 + l31;
    const int kr = kvrow < p.Nkv ? kvrow : p.Nkv - 1;

    L_ ln;
    ln.init(tid, lane, p.D);
    u32x4 kf[KS], vf[KS];
    {
        const uint16_t* kp = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1] + (int64_t)kr * p.ks[2];
        const uint16_t* vp = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1] + (int64_t)kr * p.vs[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool in = 16 * ks + 8 * hi < p.D;
            kf[ks] = in ? *(const u32x4*)(kp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
            vf[ks] = in ? *(const u32x4*)(vp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    const u32x4 qrs = pp_rsrc((const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1], p.q_bytes);
    const u32x4 grs = pp_rsrc((const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1], p.do_bytes);
    const u32x4 lrs = pp_rsrc(p.lse + b * p.ls[0] + h * p.ls[1], p.l_bytes);
    const u32x4 drs = pp_rsrc(p.delta + b * p.ls[0] + h * p.ls[1], p.l_bytes);
    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, g_rowb = (uint32_t)p.dos[2] * 2u;

    int ntiles = (p.Nq + kKvTile - 1) / kKvTile;
    int tile0 = CAUSAL ? kv0 / kKvTile : 0;
    if (part >= 0) { tile0 = part * ntiles / p.nsplit; ntiles = (part + 1) * ntiles / p.nsplit; }
    const int tile0_w = CAUSAL ? kvw0 / kKvTile : 0;                                  // this wave's first useful tile
    const int first_plain = CAUSAL ? (kvw0 + 31 + kKvTile - 1) / kKvTile : 0;         // tiles from here on lie below this wave's diagonal

    uint32_t qr_src[NPASS], qt_src[NPASS], gr_src[NPASS], gt_src[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        qr_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.r_src[i];
        qt_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.t_src[i];
        gr_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.r_src[i];
        gt_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.t_src[i];
    }
    auto stage_load = [&](int tile, int stage) __attribute__((always_inline)) {
        const uint32_t base = smem_addr + (uint32_t)stage * STAGEB;
        const uint32_t qsoff = (uint32_t)tile * kKvTile * q_rowb, gsoff = (uint32_t)tile * kKvTile * g_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const uint32_t dst = base + (uint32_t)(wave * 64 + 512 * i) * 16u;
            pp_dma16(qrs, dst, qr_src[i], qsoff);                  // Q row-form
            pp_dma16(grs, dst + TILEB, gr_src[i], gsoff);          // dO row-form
            pp_dma16(qrs, dst + 2 * TILEB, qt_src[i], qsoff);      // Q tr-form
            pp_dma16(grs, dst + 3 * TILEB, gt_src[i], gsoff);      // dO tr-form
        }
        const uint32_t lsoff = (uint32_t)tile * kKvTile * 4u;
        if ((wave & 3) == 0) pp_dma4(grp == 0 ? lrs : drs, base + LOFF + 256u * grp, (uint32_t)lane * 4u, lsoff);   // wave 0: L, wave 4: delta
    };

    f32x16 acc[DT], accv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[dt][r] = 0.f; accv[dt][r] = 0.f; }
    const float c = p.c, scale = p.scale;

    f32x16 s0, s1, d0, d1;          // S / P and dP / dS of the tile between its M and V phase
    u32x4 xfp[4], xfd[4];           // P and dS as B fragments (contraction index = q), from V(t) to M(t + 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) { xfp[i] = (u32x4){0u, 0u, 0u, 0u}; xfd[i] = (u32x4){0u, 0u, 0u, 0u}; }

    // S[q, kv] = Q K^T, dP[q, kv] = dO V^T  (lane = kv)
    auto m1 = [&](int st) __attribute__((always_inline)) {
        const lds_char_ptr qR = smem + st * STAGEB, gR = qR + TILEB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            s0 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks]), kf[ks], s0);
            s1 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks] + 32 * ROWB), kf[ks], s1);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            d0 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks]), vf[ks], d0);
            d1 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks] + 32 * ROWB), vf[ks], d1);
        }
    };
    // dV^T += dO^T P, dK^T += Q^T dS with the fragments the last V phase left
    auto m2 = [&](int st) __attribute__((always_inline)) {
        const lds_char_ptr qR = smem + st * STAGEB;
        const lds_char_ptr gT = qR + 3 * TILEB, tT = qR + 2 * TILEB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const lds_char_ptr va = gT + ln.vr_off[dt] + 16 * ks * ROWB;
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB)));
                accv[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, xfp[ks], accv[dt]);
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const lds_char_ptr va = tT + ln.vr_off[dt] + 16 * ks * ROWB;
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB)));
                acc[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, xfd[ks], acc[dt]);
            }
    };
    // P = 2^(S c - L[q]), dS / scale = P (dP - D[q]); rows q are spread over the registers: q = q0t + (r&3) + 8(r>>2) + 4hi (+32)
    auto vphase = [&](int tile, int st, bool masked) __attribute__((always_inline)) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const lds_char_ptr lt = smem + st * STAGEB + LOFF;
        const int q0t = tile * kKvTile;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 L0 = *(const __attribute__((address_space(3))) f32x4*)(lt + (8 * g4 + 4 * hi) * 4);
            const f32x4 L1 = *(const __attribute__((address_space(3))) f32x4*)(lt + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -L0[e]));
                s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -L1[e]));
            }
        }
        if (CAUSAL && masked) {                  // causal: pairs with kv > q contribute nothing (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = q0t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (kvrow > qi) s0[r] = 0.f;
                if (kvrow > qi + 32) s1[r] = 0.f;
            }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 D0 = *(const __attribute__((address_space(3))) f32x4*)(lt + 256 + (8 * g4 + 4 * hi) * 4);
            const f32x4 D1 = *(const __attribute__((address_space(3))) f32x4*)(lt + 256 + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                d0[r] = s0[r] * (d0[r] - D0[e]);
                d1[r] = s1[r] * (d1[r] - D1[e]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xfp[0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
            xfp[1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
            xfp[2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
            xfp[3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
            xfd[0][i] = pack2<BF16>(d0[2 * i], d0[2 * i + 1]);
            xfd[1][i] = pack2<BF16>(d0[8 + 2 * i], d0[8 + 2 * i + 1]);
            xfd[2][i] = pack2<BF16>(d1[2 * i], d1[2 * i + 1]);
            xfd[3][i] = pack2<BF16>(d1[8 + 2 * i], d1[8 + 2 * i + 1]);
        }
    };

    // The request for tile t + 2 goes out at the start of phase 2t + 2 of the workgroup's clock (V(t) of waves 0..3, M(t) of waves 4..7: the
    // stage it overwrites held tile t - 2, last read in phase 2t) and a wave waits for its pieces of tile t + 1 at the end of that same phase —
    // everything older than the request it has just made (counted vmcnt) — three phases after asking and one barrier before the first reader.
    const bool extra = (wave & 3) == 0;           // waves 0 and 4 also stage L / delta
    auto wait_older = [&](bool issued) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (extra) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NPASS + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NPASS) : "memory");
#endif
    };
    if (tile0 < ntiles) stage_load(tile0, 0);
    if (tile0 + 1 < ntiles) stage_load(tile0 + 1, 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)               // the compiler's own vmcnt wait for the K / V fragments goes HERE, not into the sweep
        asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
#endif
    wait_older(false);
    pp_barrier();
    if (grp == 1) pp_barrier();                   // waves 4..7 run half a tile behind waves 0..3
    for (int tile = tile0; tile < ntiles; ++tile) {
        const int st = (tile - tile0) & (kPpStages - 1);
        const bool act = !CAUSAL || tile >= tile0_w;
        const bool more = tile + 2 < ntiles;
        // ---- M phase
        if (grp == 1 && more) stage_load(tile + 2, (st + 2) & (kPpStages - 1));
        __builtin_amdgcn_s_setprio(1);
        if (tile > tile0 && (!CAUSAL || tile - 1 >= tile0_w)) m2((st + kPpStages - 1) & (kPpStages - 1));
        if (act) m1(st);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 1) wait_older(more);
        pp_barrier();
        // ---- V phase
        if (grp == 0 && more) stage_load(tile + 2, (st + 2) & (kPpStages - 1));
        if (act) vphase(tile, st, CAUSAL && tile < first_plain);
        if (grp == 0) wait_older(more);
        pp_barrier();
    }
    if (tile0 < ntiles && (!CAUSAL || ntiles - 1 >= tile0_w)) m2((ntiles - 1 - tile0) & (kPpStages - 1));
    if (grp == 0) pp_barrier();

    if constexpr (!CAUSAL) {
        if (part >= 0) {                   // unscaled f32 partial dK and dV tiles
            const int64_t slot = sidx * p.nsplit + part, ntile = (int64_t)p.split_items * p.nsplit;
            store_partial_t<DT>(acc, p.ws + slot * kSplitRows * HD, 32 * wave + l31, hi);
            store_partial_t<DT>(accv, p.ws + (ntile + slot) * kSplitRows * HD, 32 * wave + l31, hi);
            return;
        }
    }
    if (kvrow < p.Nkv) {
        uint16_t* op = (uint16_t*)p.dk + b * p.dks[0] + h * p.dks[1] + (int64_t)kvrow * p.dks[2];
        store_acc_t<BF16, DT>(acc, op, hi, scale, p.D);
        uint16_t* ov = (uint16_t*)p.dv + b * p.dvs[0] + h * p.dvs[1] + (int64_t)kvrow * p.dvs[2];
        store_acc_t<BF16, DT>(accv, ov, hi, 1.0f, p.D);
    }
}

}  // namespace fa2
