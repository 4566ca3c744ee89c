// fa2_fwd_short.hip.h — forward for KV sweeps of at most two tiles (Nkv <= 128), non-causal, no bias: the cross-attention calls of the reference's
// own use case (SD 1.5 / SDXL text conditioning: Nkv = 77; README.md:35-37 of the reference) and low-resolution self-attention.  Round 6.
// Reference counterpart: the same fwd_kernel (kernel_fp16.cu:306-544) — its loop runs one KV block there.
//
// Such a call moves Q in and O out and does next to no arithmetic: B2 H10 N4096 x Nkv77 D64 is 21 MB and 1.6 GFLOP.  The streaming kernels
// (fa2_fwd_kernel.hip.h) run it as a two-step software pipeline with five barriers and two dependent memory waits per workgroup and keep ~200 registers
// for the steady state they never reach, so two workgroups share a CU and 640 of them take two rounds of latency chains (11 us; torch SDPA 10.9).
// Here: ONE memory round trip and two barriers.  A workgroup (4 waves x 32 rows) issues the Q fragment loads and the LDS-DMA of EVERY K and V tile up
// front, waits once, forms all scores (four blocks of 32 keys per wave at most), takes the exact row max, exponentiates, multiplies by V and stores — no
// running state, no rescale.  f32 scale, f32 row sums (contract 0, like every compiler-scheduled kernel).  LDS: the K tiles, the V tiles, and the
// wave-private O images of the epilogue over the K tiles (free once every wave has its scores); registers for three (head dims <= 64) or two
// workgroups per CU.
#pragma once
#include "fa2_fwd_kernel.hip.h"

namespace fa2 {

constexpr int kShortMaxKv = 2 * kKvTile;
constexpr int kShortRows = 128;

template <int HD>
constexpr int short_epi_bytes() { return 4 * 32 * (HD * 2 + 16); }
// [K tiles | (rest of the O images)] [V tiles]
template <int HD>
__host__ __device__ constexpr int short_vbase(int ntiles) {
    return ntiles * Geo<HD, 4>::TILEB > short_epi_bytes<HD>() ? ntiles * Geo<HD, 4>::TILEB : short_epi_bytes<HD>();
}
template <int HD>
__host__ __device__ constexpr int short_lds_bytes(int ntiles) { return short_vbase<HD>(ntiles) + ntiles * Geo<HD, 4>::TILEB; }

// NB = ceil(Nkv / 32): the 32-key blocks that hold a key (1 .. 4).  Blocks, k-steps of P.V (16 keys) and LDS-DMA pieces (32 tile rows) without one
// are not in the instruction stream: a wave of this kernel is issue-bound (2.5 waves per SIMD at SDXL's 64 x 64 cross-attention: ~1 250 instructions
// each at NB = 4), not latency-bound — Nkv = 77 runs 12 + 10 MFMAs and 48 exps per lane instead of 16 + 16 and 64.
template <int HD, bool BF16, int NB>
__global__ __launch_bounds__(256, (HD <= 64 ? 3 : 2)) void fwd_short_kernel(const FwdParams p) {
    using G_ = Geo<HD, 4>;
    constexpr int ROWB = G_::ROWB, TILEB = G_::TILEB, NPASS = G_::NPASS, KS_QK = HD / 16, DT = HD / 32;
    constexpr int NT = (NB + 1) / 2;                       // KV tiles staged
    constexpr int RPP = 256 / G_::G;                       // tile rows one staging pass of the workgroup covers (32 at head dim 64, 16 at 128)
    constexpr int VBASE = short_vbase<HD>(NT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int bh, qblk;
    block_to_head_qblock<false>(p, (int)blockIdx.x, bh, qblk);
    const int b = bh / p.H, h = bh % p.H;
    const int qw0 = qblk * kShortRows + wave * 32, qrow = qw0 + l31;

    // ---- Q fragments (B operand), straight from memory: lane reads 8 consecutive d of its row per k-step; columns >= D are zeros
    u32x4 qf[KS_QK];
    {
        const int qr = qrow < p.Nq ? qrow : p.Nq - 1;
        const uint16_t* qp = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qr * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) qf[ks] = (16 * ks + 8 * hi < p.D) ? *(const u32x4*)(qp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
    }

    // ---- every K and V tile by LDS-DMA, now (the images of fa2_fwd_kernel.hip.h: lane l supplies the source of image slot wave * 64 + 256 i + l);
    // rows >= Nkv are outside the descriptors and arrive as zeros
    {
        const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;
        const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1];
        const uint16_t* vbase_p = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1];
        const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes, 0x00020000);
        const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase_p, 0, p.v_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / G_::G, slot = idx % G_::G;
            const int gk = slot ^ ((row / G_::RPB) & G_::KMASK);
            const int gv = ((((slot >> 2) ^ ((row / G_::RPB) & G_::VMASK))) << 2) | (slot & 3);
            const uint32_t kd = gk * 8 < p.D ? row * k_rowb + gk * 16 : kOobOffset;
            const uint32_t vd = gv * 8 < p.D ? row * v_rowb + gv * 16 : kOobOffset;
            char* kdst = smem + (wave * 64 + 256 * i) * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (64 * t + RPP * i < 32 * NB) {        // (compile time) the pass holds rows of a block that has a key
                    dma16_to_lds(krs, kdst + t * TILEB, kd, (uint32_t)(t * kKvTile) * k_rowb);
                    dma16_to_lds(vrs, kdst + VBASE + t * TILEB, vd, (uint32_t)(t * kKvTile) * v_rowb);
                }
            }
        }
    }
    if (p.negate_q) {
        const uint32_t sgn = 0x80008000u;
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) qf[ks] ^= (u32x4){sgn, sgn, sgn, sgn};
    }
    __syncthreads();

    // ---- S^T = K Q^T: block j = keys [32 j, 32 j + 32) = half j & 1 of tile j >> 1
    f32x16 s[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const char* kt = smem + (j >> 1) * TILEB + (j & 1) * 32 * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) {
            const u32x4 a = *(const u32x4*)(kt + G_::k_off(l31, 2 * ks + hi));
            if (ks == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                s[j] = mfma16<BF16>(a, qf[ks], z);
            } else {
                s[j] = mfma16<BF16>(a, qf[ks], s[j]);
            }
        }
    }
    // ---- the last block's keys past Nkv (the image holds zeros there: a score of 0, not of "nothing"), exact row max, P, row sum
    const float c = p.c;
    {
        const int lim = p.Nkv - 1 - 32 * (NB - 1) - 4 * hi;            // element r of the last block is key 32 (NB - 1) + 4 hi + (r & 3) + 8 (r >> 2)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) > lim) s[NB - 1][r] = -INFINITY;
    }
    float m = s[0][0];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int r = j == 0 ? 1 : 0;
#pragma unroll
        for (; r + 1 < 16; r += 2) m = max3(m, s[j][r], s[j][r + 1]);
        if (r < 16) m = __builtin_fmaxf(m, s[j][r]);
    }
    m = half_swap_max(m);
    const float mc = m * c;
    float l = 0.f;
    u32x4 pf[NB][2];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[j][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], c, -mc));
            rs += s[j][r];
        }
        l += rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[j][0][i] = pack2<BF16>(s[j][2 * i], s[j][2 * i + 1]);
            pf[j][1][i] = pack2<BF16>(s[j][8 + 2 * i], s[j][8 + 2 * i + 1]);
        }
    }
    // ---- O^T = V^T P^T: k-step x = keys [16 x, 16 x + 16); the last block's second k-step only if it holds a key (wave-uniform)
    f32x16 acc[DT];
    {
        const int pp = lane & 15, g1 = (lane >> 4) & 1;
        const bool last_too = p.Nkv > 32 * (NB - 1) + 16;
#pragma unroll
        for (int x = 0; x < 2 * NB; ++x) {
            if (x < 2 * NB - 1 || last_too) {
                const char* vt = smem + VBASE + (x >> 2) * TILEB + 16 * (x & 3) * ROWB;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* va = vt + G_::v_off(4 * hi + (pp >> 2), (32 * dt + 16 * g1 + 4 * (pp & 3)) * 2);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB));
                    const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi4);
                    const u32x4 a = {lo2[0], lo2[1], hi2[0], hi2[1]};
                    if (x == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[dt] = mfma16<BF16>(a, pf[0][0], z);
                    } else {
                        acc[dt] = mfma16<BF16>(a, pf[x >> 1][x & 1], acc[dt]);
                    }
                }
            }
        }
    }
    // ---- epilogue (reference: kernel_fp16.cu:510-543): O / l through a wave-private LDS image (over the K tiles: every wave has its scores once all
    // are here), whole rows to memory; lse = m c + log2(l)
    constexpr int EROW = HD * 2 + 16, LPR = HD * 2 / 16, RPI = 64 / LPR;
    const float l_tot = half_swap_sum(l);
    const float inv_l = 1.0f / l_tot;
    __syncthreads();
    char* img = smem + wave * (32 * EROW);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4 += 2) {
            const f32x16& a = acc[dt];
            uint32_t a0 = pack2<BF16>(a[4 * r4 + 0] * inv_l, a[4 * r4 + 1] * inv_l);
            uint32_t a1 = pack2<BF16>(a[4 * r4 + 2] * inv_l, a[4 * r4 + 3] * inv_l);
            uint32_t b0 = pack2<BF16>(a[4 * r4 + 4] * inv_l, a[4 * r4 + 5] * inv_l);
            uint32_t b1 = pack2<BF16>(a[4 * r4 + 6] * inv_l, a[4 * r4 + 7] * inv_l);
            auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            *(u32x4*)(img + l31 * EROW + (32 * dt + 8 * (r4 + hi)) * 2) = (u32x4){x0[0], x1[0], x0[1], x1[1]};
        }
    }
    if (qrow < p.Nq && hi == 0) p.lse[b * p.ls[0] + h * p.ls[1] + qrow] = mc + __builtin_amdgcn_logf(l_tot);
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
