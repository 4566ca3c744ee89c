// fa2_bwd_short.hip.h — the dQ pass of the backward for KV sweeps of at most two tiles (Nkv <= 128), non-causal, no bias: the forward's twin
// (fa2_fwd_short.hip.h), for LoRA-style training through the cross-attention of an SD UNet (reference README.md:151-154).  Round 6.
// Reference counterpart: the dQ half of bwd_kernel (kernel_fp16.cu:547-740) and its `Di` (:605-631).
//
// Like the forward such a call is issue-bound: B8 H16 N4096 x 77 D64 runs the streaming dQ pass (fa2_bwd_kernel.hip.h: a two-stage pipeline over two
// tiles, the second one three-quarters masked) in 94 us against 60 us for moving Q, O, dO in and dQ out.  Here a workgroup (4 waves x 32 Q rows) loads
// its Q and dO fragments and forms delta = rowsum(dO * O), stages every K and V row block that holds a key at once (K in row form and in the
// transposed-read form, V in row form), waits once, and runs block by block: S^T = K Q^T, P^T = 2^(S^T c - L) (L is known: no running state at all),
// dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T.  One block of S and dP is live at a time: ~120 registers.  NB = ceil(Nkv / 32) as in the
// forward: blocks, 16-key steps of the dQ product and LDS-DMA passes without a key are not in the instruction stream.
// The delta workspace gets delta — or -delta when the hand-scheduled dK / dV pass (head dim 128) follows (host.cpp: launch_bwd).
#pragma once
#include "fa2_bwd_kernel.hip.h"

namespace fa2 {

constexpr int kBwdShortRows = 128;
// three images (K row-form, V row-form, K tr-form) of the 32 NB rows that hold a key (an image is row-major: nothing is kept for the rest of the last tile)
template <int HD>
__host__ __device__ constexpr int bwd_short_lds_bytes(int nb) { return 3 * 32 * nb * Geo<HD, 4>::ROWB; }

template <int HD, bool BF16, int NB>
__global__ __launch_bounds__(256, (HD <= 64 ? 3 : 2)) void bwd_short_dq_kernel(const BwdParams p, const int neg_delta) {
    using L_ = BwdLane<HD, 4>;
    using G_ = Geo<HD, 4>;
    constexpr int NPASS = L_::NPASS, KS = L_::KS, DT = L_::DT, ROWB = L_::ROWB, TILEB = L_::TILEB;
    constexpr int NT = (NB + 1) / 2;                       // KV tiles staged
    constexpr int RPP = 256 / G_::G;                       // tile rows one staging pass of the workgroup covers
    constexpr int IMGB = 32 * NB * ROWB;                   // one image: the rows of the blocks that hold a key
    constexpr int VOFF = IMGB, TOFF = 2 * IMGB;            // K row-form | V row-form | K tr-form
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const lds_char_ptr smem = (lds_char_ptr)smem_generic;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (head, q block): the non-causal mapping of bwd_dq_kernel (all q blocks of a head on one XCD)
    const int nbh = p.B * p.H, bid = blockIdx.x;
    int bh, qblk;
    if ((nbh & 7) == 0) { const int slot = bid >> 3; bh = (bid & 7) + 8 * (slot / p.nblk); qblk = slot % p.nblk; }
    else { bh = bid / p.nblk; qblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int qw0 = qblk * kBwdShortRows + 32 * wave, qrow = qw0 + l31;
    const int qr = qrow < p.Nq ? qrow : p.Nq - 1;

    L_ ln;
    ln.init(tid, lane, p.D);
    u32x4 qf[KS], gf[KS];
    {
        const uint16_t* qp = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qr * p.qs[2];
        const uint16_t* gp = (const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1] + (int64_t)qr * p.dos[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool in = 16 * ks + 8 * hi < p.D;
            qf[ks] = in ? *(const u32x4*)(qp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
            gf[ks] = in ? *(const u32x4*)(gp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    // ---- every K and V row block that holds a key, now: K row-form, V row-form, K tr-form (rows >= Nkv are outside the descriptors: zeros)
    {
        const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1];
        const uint16_t* vbase = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1];
        const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes, 0x00020000);
        const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, p.v_bytes, 0x00020000);
        const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const lds_char_ptr dst = smem + (wave * 64 + 256 * i) * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (64 * t + RPP * i < 32 * NB) {        // (compile time) the pass holds rows of a block that has a key
                    const uint32_t krow = (uint32_t)(t * kKvTile + ln.rowi[i]) * k_rowb, vrow = (uint32_t)(t * kKvTile + ln.rowi[i]) * v_rowb;
                    dma16_to_lds3(krs, dst + t * TILEB, krow + ln.r_src[i], 0);
                    dma16_to_lds3(vrs, dst + VOFF + t * TILEB, vrow + ln.r_src[i], 0);
                    dma16_to_lds3(krs, dst + TOFF + t * TILEB, krow + ln.t_src[i], 0);
                }
            }
        }
    }
    const float Lq = p.lse[b * p.ls[0] + h * p.ls[1] + qr];
    // D_i = sum_d dO[i,d] * O[i,d] for the lane's own row, from the dO fragments in registers; to the workspace for the dK / dV pass
    float Dq;
    {
        const uint16_t* orow = (const uint16_t*)p.o + b * p.os[0] + h * p.os[1] + (int64_t)qr * p.os[2];
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (16 * ks + 8 * hi < p.D) dsum += dot8<BF16>(*(const u32x4*)(orow + 16 * ks + 8 * hi), gf[ks]);
        Dq = half_swap_sum(dsum);
        if (hi == 0 && qrow < p.Nq) p.delta[b * p.ls[0] + h * p.ls[1] + qrow] = neg_delta ? -Dq : Dq;
    }
    __syncthreads();

    const float c = p.c;
    f32x16 acc[DT];
    const bool last_too = p.Nkv > 32 * (NB - 1) + 16;      // the last block's second 16-key step holds a key (wave-uniform)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const lds_char_ptr kR = smem + (j >> 1) * TILEB + (j & 1) * 32 * ROWB;
        const lds_char_ptr vR = kR + VOFF;
        f32x16 s, d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {                         // S^T = K Q^T, dP^T = V dO^T
            const u32x4 ka = lds_load128(kR + ln.kr_off[ks]), va = lds_load128(vR + ln.kr_off[ks]);
            if (ks == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                s = mfma16<BF16>(ka, qf[ks], z);
                d = mfma16<BF16>(va, gf[ks], z);
            } else {
                s = mfma16<BF16>(ka, qf[ks], s);
                d = mfma16<BF16>(va, gf[ks], d);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -Lq));        // P^T
        if (j == NB - 1) {                                        // keys past Nkv contribute nothing
            const int lim = p.Nkv - 1 - 32 * (NB - 1) - 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) > lim) s[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s[r] * (d[r] - Dq);   // dS^T / scale; `scale` is applied once, to the finished dQ
        u32x4 df[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            df[0][i] = pack2<BF16>(s[2 * i], s[2 * i + 1]);
            df[1][i] = pack2<BF16>(s[8 + 2 * i], s[8 + 2 * i + 1]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {                             // dQ^T += K^T dS^T, 16 keys per step
            const int x = 2 * j + e;
            if (x < 2 * NB - 1 || last_too) {
                const lds_char_ptr tT = smem + TOFF + (x >> 2) * TILEB + 16 * (x & 3) * ROWB;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const lds_char_ptr ta = tT + ln.vr_off[dt];
                    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ta)));
                    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ta + 8 * ROWB)));
                    const u32x4 a = {lo[0], lo[1], h2[0], h2[1]};
                    if (x == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[dt] = mfma16<BF16>(a, df[e], z);
                    } else {
                        acc[dt] = mfma16<BF16>(a, df[e], acc[dt]);
                    }
                }
            }
        }
    }
    if (qrow < p.Nq) {
        uint16_t* op = (uint16_t*)p.dq + b * p.dqs[0] + h * p.dqs[1] + (int64_t)qrow * p.dqs[2];
        store_acc_t<BF16, DT>(acc, op, hi, p.scale, p.D);
    }
}

}  // namespace fa2
