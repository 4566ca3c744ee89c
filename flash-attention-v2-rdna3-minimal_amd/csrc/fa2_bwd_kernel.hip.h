// fa2_bwd_kernel.hip.h — FlashAttention-2 backward for MI355X (gfx950 / CDNA4), device side.
//
// Replaces the reference's bwd_kernel + mul_add_AT_B (rocwmma_fattn/kernel_fp16.cu:65-112, :547-740; bf16
// twin kernel_bf16.cu:87-135, :580-798).  The reference accumulates dQ with an unsynchronised global
// read-modify-write across KV blocks (kernel_fp16.cu:736, racy by design); here every output is owned by
// exactly one workgroup, so the result is deterministic and needs no atomics, at the price of recomputing
// S in each of the three passes:
//
//   (D_i = rowsum(dO_i * O_i), kernel_fp16.cu:605-631, is computed by bwd_dq for its own rows and stored for bwd_dkv)
//   bwd_dq_kernel                   workgroup = 256 Q rows, sweeps KV tiles:   dQ  = sum_j dS_ij K_j
//   bwd_dkv_kernel<WANT_DK=false>   workgroup = 256 KV rows, sweeps Q tiles:   dV  = sum_i P_ij^T dO_i
//   bwd_dkv_kernel<WANT_DK=true>    same sweep:                               dK  = sum_i dS_ij^T Q_i
// with P = 2^(S*c - L_i) (L = the forward's log2-domain LSE, so no running max is needed),
// dP = dO V^T, dS = scale * P * (dP - D_i)                                    (kernel_fp16.cu:684-735).
//
// All products reuse the forward kernel's two MFMA forms (fa2_fwd_kernel.hip.h):
//   "row" form   C^T[r, n] = A[r, :] . B[n, :]   A = 32 rows of an LDS tile (ds_read_b128, K-style image),
//                                                B = register fragments of the wave's own 32 rows
//   "tr" form    C^T[d, n] = sum_k T[k, d] X[k, n]  A = transpose-read of an LDS tile (V-style image),
//                                                X = 16-bit fragments straight from a row-form result
// In bwd_dq the wave's own rows are Q rows (lane = q: L_i, D_i are per-lane scalars, as in the forward);
// in bwd_dkv they are KV rows (lane = kv) and L_i, D_i vary along the accumulator registers, so they are
// staged through LDS and fetched as float4 per group of four consecutive q.
#pragma once
#include "fa2_fwd_kernel.hip.h"

namespace fa2 {

struct BwdParams {
    const void* q;
    const void* k;
    const void* v;
    const void* o;
    const void* dout;
    const float* lse;
    float* delta;     // workspace [B,H,Nq] f32, same strides as lse
    void* dq;
    void* dk;
    void* dv;
    int B, H, Nq, Nkv;
    int D;            // actual head dim (multiple of 8, <= the kernel's HD): columns >= D read as 0, are not stored
    int64_t qs[3], ks[3], vs[3], os[3], dos[3], dqs[3], dks[3], dvs[3];  // element strides: batch, head, row
    int64_t ls[2];                                                       // lse / delta strides: batch, head
    float scale, c;                                                      // scale, scale * log2(e)
    int nblk;                                                            // row blocks (of 256) of the swept-over owner
    uint32_t q_bytes, k_bytes, v_bytes, do_bytes, l_bytes;               // addressable bytes of one head's matrices
    // attention bias / boolean mask of a biased forward (BIAS kernels only; fa2_bwd_bias): as FwdParams::bias / bs / bias_kind
    const void* bias;
    int64_t bs[3];
    int bias_kind;
    int bias_tile;        // host: the bias geometry allows whole 16-byte granules -> a wave stages its tile of the bias by LDS-DMA (bwd_bias_tile_*);
                          // 0: one guarded load per score (any alignment, any strides)
    int bias_img;         // bytes of one wave's bias image (4096, or 8192 for f32)
    // split of a partly filled last round of 256-row workgroups (fa2_bwd_ws; non-causal, unbiased, 8-wave kernels; FwdParams has the
    // forward's twin): the last `split_items` workgroups of the pass being launched are each replaced by `nsplit` parts that sweep disjoint
    // tile ranges and leave f32 partial accumulators in `ws`; bwd_merge_kernel sums them.  The launcher fills these per pass.
    int full_items, split_items, nsplit;
    float* ws;            // [split_items * nsplit] tiles of 256 x HD floats (the fused dK / dV pass: all dK tiles, then as many dV tiles)
    size_t ws_bytes;      // what the caller handed over
};

// f32 partial accumulator tile of a part -> workspace.  Layout of a tile: float (((dt*4 + g) * 256 + row) * 8 + 4*hi + e) for
// d = 32dt + 8g + 4hi + e (the forward's partial O tiles: a wave's store instruction writes 1 KiB of consecutive bytes).
template <int DT>
__device__ __forceinline__ void store_partial_t(const f32x16 (&acc)[DT], float* tile, int row, int hi) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x16& a = acc[dt];
            *(f32x4*)(tile + ((dt * 4 + g) * kSplitRows + row) * 8 + 4 * hi) = (f32x4){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        }
}

// The backward kernels address LDS through address-space-3 pointers only (no generic pointers into LDS): besides
// sparing the aperture checks, this avoids a hipcc 7.2 miscompile of the generic<->LDS casts in the 4-wave kernels
// ("Illegal instruction detected: Operand has incorrect register class", a V_CMP against src_shared_base).
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_cptr;
__device__ __forceinline__ u32x4 lds_load128(lds_char_ptr p) { return *(lds_u32x4_cptr)p; }
template <typename RSRC>
__device__ __forceinline__ void dma16_to_lds3(RSRC rsrc, lds_char_ptr lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// 4 bytes per lane, global -> LDS at wave-uniform `lds_dst` + lane*4 (zero for out-of-range)
template <typename RSRC>
__device__ __forceinline__ void dma4_to_lds(RSRC rsrc, lds_char_ptr lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 4, voff, soff, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// f32 dot product of two fragments of eight 16-bit values.
template <bool BF16>
__device__ __forceinline__ float dot8(u32x4 a, u32x4 g) {
    float acc = 0.f;
    if constexpr (BF16) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            acc += __uint_as_float(a[w] << 16) * __uint_as_float(g[w] << 16) +
                   __uint_as_float(a[w] & 0xffff0000u) * __uint_as_float(g[w] & 0xffff0000u);
    } else {
        // (whole-vector bit_cast: a per-element __builtin_bit_cast(f16x2, a[w]) in an unrolled loop was folded to
        // element 0 by hipcc 7.2)
        const f16x8 ah = __builtin_bit_cast(f16x8, a), gh = __builtin_bit_cast(f16x8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)ah[e] * (float)gh[e];
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------------------
// Bias tiles of the masked backward (round 3; fa2_fwd_kernel.hip.h has the forward's, BIAS = 2).  One guarded load per score made the masked
// backward 15x the unmasked one (with lane = Q row every load instruction touches 32 cache lines for a few useful bytes).  Where the bias
// geometry allows whole 16-byte granules (pointer, strides, Nkv multiples of 16 bytes; a per-row bias) a wave stages its tile by LDS-DMA into a
// wave-private image and reads its scores' values back from there.  Rows / granules out of range of the descriptor read zeros: such scores
// are masked or belong to rows that are never stored.
//   dQ pass   (lane = Q row): 32 rows x 64 kv, granule g of row r at slot g ^ (r & MASK)  — the forward's image
//   dK/dV pass (lane = KV row): 64 q rows x 32 kv of the wave, row-major, no swizzle (a register's 32 lanes read 32 consecutive elements)
template <int ES>
__device__ __forceinline__ auto bwd_bias_rsrc(const BwdParams& p, int b, int h) {
    const char* base = (const char*)p.bias + (b * p.bs[0] + h * p.bs[1]) * ES;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(((int64_t)(p.Nq - 1) * p.bs[2] + p.Nkv) * ES), 0x00020000);
}

// element form: one bounds-checked buffer load per score, 32-bit offsets inside the (b, h) slice (host: the slice spans < 2 GiB); out of range -> 0
template <int ES, bool BF16, typename RSRC>
__device__ __forceinline__ float bwd_bias_elem(RSRC brs, uint32_t voff) {
    constexpr float kLog2e = 1.4426950408889634f;
    if constexpr (ES == 2) {
        const uint16_t w = __builtin_amdgcn_raw_buffer_load_b16(brs, voff, 0, 0);
        return (BF16 ? __uint_as_float((uint32_t)w << 16) : (float)__builtin_bit_cast(_Float16, w)) * kLog2e;
    } else if constexpr (ES == 4) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, voff, 0, 0)) * kLog2e;
    } else {
        return __builtin_amdgcn_raw_buffer_load_b8(brs, voff, 0, 0) ? 0.f : -__builtin_inff();
    }
}

template <int ES>
__device__ __forceinline__ void bwd_bias_tile_dq_load(const BwdParams& p, int b, int h, int qw0, int kv0, int lane, lds_char_ptr img) {
    constexpr int GPR = 64 * ES / 16, RPI = 64 / GPR, NI = 32 / RPI, MASK = (GPR < RPI ? GPR : RPI) - 1;
    const uint32_t rowb = (uint32_t)p.bs[2] * ES;
    const auto brs = bwd_bias_rsrc<ES>(p, b, h);
    const int lr = lane / GPR, g = (lane % GPR) ^ (lr & MASK);
    const int kvg = kv0 + g * (16 / ES);
    const uint32_t voff = kvg < p.Nkv ? (uint32_t)(qw0 + lr) * rowb + (uint32_t)kvg * ES : kOobOffset;
#pragma unroll
    for (int i = 0; i < NI; ++i) dma16_to_lds3(brs, img + i * 1024, voff, (uint32_t)(i * RPI) * rowb);
}

// log2-domain bias terms of the four scores kv = kv0 + 32 half + 8 g + 4 hi + e of this lane's row (dQ pass image)
template <int ES, bool BF16>
__device__ __forceinline__ void bwd_bias_tile_dq_read(lds_char_ptr img, int l31, int hi, int half, int g, float (&bv)[4]) {
    constexpr int GPR = 64 * ES / 16, RPI = 64 / GPR, MASK = (GPR < RPI ? GPR : RPI) - 1;
    constexpr float kLog2e = 1.4426950408889634f;
    const int byte = (32 * half + 8 * g + 4 * hi) * ES;
    const lds_char_ptr src = img + l31 * (GPR * 16) + (((byte >> 4) ^ (l31 & MASK)) << 4) + (byte & 15);
    if constexpr (ES == 2) {
        const u32x2 w = *(const __attribute__((address_space(3))) u32x2*)src;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t h16 = (e & 1) ? w[e >> 1] >> 16 : w[e >> 1] & 0xffffu;
            bv[e] = (BF16 ? __uint_as_float(h16 << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)h16)) * kLog2e;
        }
    } else if constexpr (ES == 4) {
        const u32x4 w = *(const __attribute__((address_space(3))) u32x4*)src;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = __uint_as_float(w[e]) * kLog2e;
    } else {
        const uint32_t w = *(const __attribute__((address_space(3))) uint32_t*)src;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (w & (0xffu << (8 * e))) ? 0.f : -__builtin_inff();
    }
}

template <int ES>
__device__ __forceinline__ void bwd_bias_tile_kv_load(const BwdParams& p, int b, int h, int q0t, int kvw0, int lane, lds_char_ptr img) {
    constexpr int GPR = 32 * ES / 16, RPI = 64 / GPR, NI = 64 / RPI;          // a row of the image: the wave's 32 kv
    const uint32_t rowb = (uint32_t)p.bs[2] * ES;
    const auto brs = bwd_bias_rsrc<ES>(p, b, h);
    const int lr = lane / GPR, g = lane % GPR;
    const int kvg = kvw0 + g * (16 / ES);
    const uint32_t voff = kvg < p.Nkv ? (uint32_t)(q0t + lr) * rowb + (uint32_t)kvg * ES : kOobOffset;
#pragma unroll
    for (int i = 0; i < NI; ++i) dma16_to_lds3(brs, img + i * 1024, voff, (uint32_t)(i * RPI) * rowb);
}

// log2-domain bias term of score (tile row ql, this lane's kv row l31) (dK/dV pass image)
template <int ES, bool BF16>
__device__ __forceinline__ float bwd_bias_tile_kv_read(lds_char_ptr img, int ql, int l31) {
    constexpr float kLog2e = 1.4426950408889634f;
    const lds_char_ptr src = img + ql * (32 * ES) + l31 * ES;
    if constexpr (ES == 2) {
        const uint16_t w = *(const __attribute__((address_space(3))) uint16_t*)src;
        return (BF16 ? __uint_as_float((uint32_t)w << 16) : (float)__builtin_bit_cast(_Float16, w)) * kLog2e;
    } else if constexpr (ES == 4) {
        return *(const __attribute__((address_space(3))) float*)src * kLog2e;
    } else {
        return *(const __attribute__((address_space(3))) uint8_t*)src ? 0.f : -__builtin_inff();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Shared per-lane LDS offsets and DMA source permutations (see the forward kernel for the image layouts).
template <int HD, int NW = 8>
struct BwdLane {
    using G_ = Geo<HD, NW>;
    static constexpr int NPASS = G_::NPASS, KS = G_::KS_QK, DT = G_::DT, ROWB = G_::ROWB, TILEB = G_::TILEB;
    int kr_off[KS];              // row-form A fragment: row lane&31 of a 32-row half tile, k-step ks
    int vr_off[DT];              // tr-form A fragment base for d block dt
    uint32_t r_src[NPASS];       // DMA source byte offset (minus row pitch term) for a row-form image
    uint32_t t_src[NPASS];       // ... for a tr-form image
    int rowi[NPASS];             // tile row this lane's DMA piece belongs to
    __device__ __forceinline__ void init(int tid, int lane, int D, int col0 = 0) {      // col0: first column of the slab a "tr" image holds
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kr_off[ks] = G_::k_off(l31, 2 * ks + hi);
        const int pp = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            vr_off[dt] = G_::v_off(4 * hi + (pp >> 2), (32 * dt + 16 * g1 + 4 * (pp & 3)) * 2);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int idx = tid + NW * 64 * i;
            const int row = idx / G_::G, slot = idx % G_::G;
            rowi[i] = row;
            const int gr = slot ^ ((row / G_::RPB) & G_::KMASK);                                  // source granules
            const int gt = ((((slot >> 2) ^ ((row / G_::RPB) & G_::VMASK))) << 2) | (slot & 3);
            r_src[i] = gr * 8 < D ? gr * 16 : kOobOffset;       // columns >= D: out of range for the descriptor -> zeros
            t_src[i] = col0 + gt * 8 < D ? col0 * 2 + gt * 16 : kOobOffset;
        }
    }
};

// 16-bit store of an O^T-layout accumulator (lane = column n = lane & 31, rows d) to row-major [n][d] memory
template <bool BF16, int DT>
__device__ __forceinline__ void store_acc_t(const f32x16 (&acc)[DT], uint16_t* rowp, int hi, float mul, int D, int col0 = 0) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4 += 2) {
            const f32x16& a = acc[dt];
            uint32_t a0 = pack2<BF16>(a[4 * r4 + 0] * mul, a[4 * r4 + 1] * mul);
            uint32_t a1 = pack2<BF16>(a[4 * r4 + 2] * mul, a[4 * r4 + 3] * mul);
            uint32_t b0 = pack2<BF16>(a[4 * r4 + 4] * mul, a[4 * r4 + 5] * mul);
            uint32_t b1 = pack2<BF16>(a[4 * r4 + 6] * mul, a[4 * r4 + 7] * mul);
            auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            const u32x4 w = {x0[0], x1[0], x0[1], x1[1]};
            if (col0 + 32 * dt + 8 * (r4 + hi) < D) *(u32x4*)(rowp + col0 + 32 * dt + 8 * (r4 + hi)) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// dQ: workgroup = 256 Q rows (8 waves x 32), sweep over KV tiles of 64.
// LDS per stage: K row-form | V row-form | K tr-form; two stages.
// NW = 8: two waves per SIMD, 256 VGPRs, two LDS stages (D <= 128).  NW = 4 (D = 256): one wave per SIMD with the
// 512-register budget, 128 rows per workgroup, ONE LDS stage (three 32-KiB images do not fit twice) — a correct, simple
// path for the rare large head dim (SD1.5's D = 160), not a tuned one.
// HDV < HD (head dims above 256, HD = 512): the workgroup produces the HDV-column slab blockIdx.y of dQ — S and dP are contracted over
// the whole head dim (Q / dO fragments of all HD columns in registers: the 512-register budget of one wave per SIMD), only the K^T image
// and the accumulator are slab-sized; every slab recomputes S and dP (a correct path for the SD-VAE-sized head dim, not a tuned one).
// BIAS: the forward was fa2_fwd_bias — P = 2^(S c + bias log2e - L); a fully masked row (L = -inf) has P = 0.
// KSN / DTN ("trimmed" instantiations, bwd_hip.cpp; the forward's twin, fa2_fwd_kernel.hip.h): a head dim well below HD keeps HD's LDS
// images and staging (columns >= D are never fetched) but runs only KSN = ceil(D / 16) k-steps of the products contracted over the head dim
// and DTN = ceil(D / 32) column blocks of the accumulators.  Defaults = the full kernel.
template <int HD, bool BF16, bool CAUSAL, int NW = 8, int HDV = HD, int BIAS = 0, int KSN = HD / 16, int DTN = HDV / 32>
__global__ __launch_bounds__(NW * 64, (NW == 8 || HD <= 128) ? 2 : 1) void bwd_dq_kernel(const BwdParams p) {
    using L_ = BwdLane<HD, NW>;
    using LV_ = BwdLane<HDV, NW>;             // geometry of the transposed-read image (the slab)
    constexpr int kRows = NW * 32;            // rows per workgroup (p.nblk = ceil(N / kRows))
    constexpr bool DBUF = NW == 8 || HD <= 128;   // (NW = 4 at head dims <= 128: the small-grid shape, 128 Q rows per workgroup, 256 registers, two stages)
    constexpr int NPASS = L_::NPASS, KS = KSN, ROWB = L_::ROWB, TILEB = L_::TILEB;
    constexpr int NPASSV = LV_::NPASS, DT = DTN, ROWBV = LV_::ROWB, TILEBV = LV_::TILEB;
    static_assert(KSN <= L_::KS && DTN <= LV_::DT, "trimmed loop bounds");
    constexpr int STAGEB = 2 * TILEB + TILEBV;
    const int vcol0 = HDV == HD ? 0 : blockIdx.y * HDV;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const lds_char_ptr smem = (lds_char_ptr)smem_generic;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (head, q block): as in the forward kernel (causal: longest-first across heads)
    const int nbh = p.B * p.H;
    int bid = blockIdx.x;
    // split last round: blocks [full_items, ...) are parts, part-major (fa2_fwd_kernel.hip.h has the forward's twin)
    int part = -1, sidx = 0;
    if constexpr (!CAUSAL && NW == 8 && HDV == HD) {
        if (p.nsplit > 1 && bid >= p.full_items) {
            const int j = bid - p.full_items;
            part = j / p.split_items;
            sidx = j % p.split_items;
            bid = p.full_items + sidx;
        }
    }
    int bh, qblk;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;
        if (CAUSAL) { bh = (bid & 7) + 8 * (slot % hpx); qblk = p.nblk - 1 - slot / hpx; }
        else { bh = (bid & 7) + 8 * (slot / p.nblk); qblk = slot % p.nblk; }
    } else if (CAUSAL) { bh = bid % nbh; qblk = p.nblk - 1 - bid / nbh; }
    else { bh = bid / p.nblk; qblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * kRows, qw0 = q0 + 32 * wave, qrow = qw0 + l31;
    const int qr = qrow < p.Nq ? qrow : p.Nq - 1;

    L_ ln;
    ln.init(tid, lane, p.D);
    LV_ lnv;                                  // (the same object when the slab is the whole head dim)
    if constexpr (HDV != HD) lnv.init(tid, lane, p.D, vcol0);
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
    float Lq = p.lse[b * p.ls[0] + h * p.ls[1] + qr];
    if (BIAS && Lq == -__builtin_inff()) Lq = __builtin_inff();        // fully masked row: every P below becomes 2^(-inf) = 0
    // D_i = sum_d dO[i,d] * O[i,d] (the reference's `Di`, kernel_fp16.cu:605-631) for the lane's own row, from the dO
    // fragments already in registers; stored to the delta workspace for the dK pass that follows on the stream.
    float Dq;
    {
        const uint16_t* orow = (const uint16_t*)p.o + b * p.os[0] + h * p.os[1] + (int64_t)qr * p.os[2];
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (16 * ks + 8 * hi < p.D) dsum += dot8<BF16>(*(const u32x4*)(orow + 16 * ks + 8 * hi), gf[ks]);
        }
        Dq = half_swap_sum(dsum);
        if (hi == 0 && qrow < p.Nq && vcol0 == 0 && part <= 0) p.delta[b * p.ls[0] + h * p.ls[1] + qrow] = Dq;
    }

    const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t* vbase = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1];
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes, 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, p.v_bytes, 0x00020000);
    const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;

    int ntiles = (p.Nkv + kKvTile - 1) / kKvTile;
    if (CAUSAL) {
        const int qmax = (q0 + kRows < p.Nq ? q0 + kRows : p.Nq) - 1;
        const int nt_c = qmax / kKvTile + 1;
        ntiles = nt_c < ntiles ? nt_c : ntiles;
    }
    int ntiles_w = ntiles;
    if (CAUSAL) { const int nt_w = (qw0 + 31) / kKvTile + 1; ntiles_w = nt_w < ntiles ? nt_w : ntiles; }
    asm volatile("" : "+s"(ntiles_w));   // opaque: stops the non-causal build from peeling the tail tile into a register-hungry shape

    // per-lane source offsets of the staging loads within a tile (loop-invariant); the tile's own byte offset rides in soffset
    // (kept for the 8-wave kernels only: the 4-wave ones have 8 passes per tile and no registers to spare)
    constexpr int NKEEP = NW == 8 ? NPASS : 1;
    uint32_t kr_src[NKEEP], vr_src[NKEEP], kt_src[NKEEP];
    if constexpr (NW == 8) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            kr_src[i] = (uint32_t)ln.rowi[i] * k_rowb + ln.r_src[i];
            vr_src[i] = (uint32_t)ln.rowi[i] * v_rowb + ln.r_src[i];
            kt_src[i] = (uint32_t)ln.rowi[i] * k_rowb + ln.t_src[i];
        }
    }
    auto stage_load = [&](int tile, int stage) __attribute__((always_inline)) {
        const lds_char_ptr base = smem + stage * STAGEB;
        const uint32_t ksoff = (uint32_t)tile * kKvTile * k_rowb, vsoff = (uint32_t)tile * kKvTile * v_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const lds_char_ptr dst = base + (wave * 64 + NW * 64 * i) * 16;
            if constexpr (NW == 8) {
                dma16_to_lds3(krs, dst, FA2_TILE_OFF(kr_src[i], ksoff));
                dma16_to_lds3(vrs, dst + TILEB, FA2_TILE_OFF(vr_src[i], vsoff));
                dma16_to_lds3(krs, dst + 2 * TILEB, FA2_TILE_OFF(kt_src[i], ksoff));
            } else {
                const uint32_t krow = (uint32_t)(tile * kKvTile + ln.rowi[i]) * k_rowb;
                const uint32_t vrow = (uint32_t)(tile * kKvTile + ln.rowi[i]) * v_rowb;
                dma16_to_lds3(krs, dst, krow + ln.r_src[i], 0);
                dma16_to_lds3(vrs, dst + TILEB, vrow + ln.r_src[i], 0);
                if constexpr (HDV == HD) dma16_to_lds3(krs, dst + 2 * TILEB, krow + ln.t_src[i], 0);
            }
        }
        if constexpr (HDV != HD) {            // the slab's K^T image has its own (narrower) geometry
#pragma unroll
            for (int i = 0; i < NPASSV; ++i) {
                const lds_char_ptr dst = base + 2 * TILEB + (wave * 64 + NW * 64 * i) * 16;
                dma16_to_lds3(krs, dst, (uint32_t)(tile * kKvTile + lnv.rowi[i]) * k_rowb + lnv.t_src[i], 0);
            }
        }
    };

    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const float c = p.c, scale = p.scale;

    // tiles [0, n_plain) need no mask for this wave: fully inside Nkv and fully below the wave's diagonal
    int n_plain = p.Nkv / kKvTile;
    if (CAUSAL) { const int nc = (qw0 + 1) / kKvTile; n_plain = nc < n_plain ? nc : n_plain; }
    n_plain = n_plain < ntiles_w ? n_plain : ntiles_w;

    // ONE tile body: two bodies (masked / plain) joined by a branch made the register allocator copy the 64 accumulator registers
    // at the join in every iteration (64 v_mov per tile in the ISA); the mask is a wave-uniform block instead.
    auto tile_body = [&](int tile, int st, bool masked) __attribute__((always_inline)) {
        {
            if constexpr (BIAS == 2) {
                {                           // this wave's bias tile: in flight under the S products
                    const lds_char_ptr bimg = smem + (DBUF ? 2 : 1) * STAGEB + wave * p.bias_img;
                    if (p.bias_kind == 1) bwd_bias_tile_dq_load<2>(p, b, h, qw0, tile * kKvTile, lane, bimg);
                    else if (p.bias_kind == 2) bwd_bias_tile_dq_load<4>(p, b, h, qw0, tile * kKvTile, lane, bimg);
                    else bwd_bias_tile_dq_load<1>(p, b, h, qw0, tile * kKvTile, lane, bimg);
                }
            }
            const lds_char_ptr kR = smem + st * STAGEB;
            const lds_char_ptr vR = kR + TILEB;
            const lds_char_ptr kT3 = kR + 2 * TILEB;
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                     // S^T = K Q^T
                s0 = mfma16<BF16>(lds_load128(kR + ln.kr_off[ks]), qf[ks], s0);
                s1 = mfma16<BF16>(lds_load128(kR + ln.kr_off[ks] + 32 * ROWB), qf[ks], s1);
            }
            const int kv0 = tile * kKvTile;
            const int lim_c = CAUSAL ? qrow : 0x7fffffff;
            const int lim = lim_c < p.Nkv - 1 ? lim_c : p.Nkv - 1;   // kv index must be <= lim (MASKED tiles only)
            // P^T = 2^(S^T c - L); masked entries -> 0
            constexpr bool tiled = BIAS == 2;       // (BIAS = 1: one load per score; the two forms are separate instantiations — together
                                                    //  they spilled hundreds of bytes per lane)
            if constexpr (BIAS == 1) {
                // element form (any alignment, broadcast rows: a [B,1,1,Nkv] key-padding mask): bounds-checked buffer loads with 32-bit offsets, the
                // kind dispatched once outside the loop (guarded 64-bit loads with the kind inside it spilled 400-1000 bytes per lane)
                auto one = [&](auto es_t) __attribute__((always_inline)) {
                    constexpr int ES = decltype(es_t)::value;
                    const auto brs = bwd_bias_rsrc<ES>(p, b, h);
                    const uint32_t v0 = (uint32_t)qr * (uint32_t)p.bs[2] * ES + (uint32_t)(kv0 + 4 * hi) * ES;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t vo = v0 + ((r & 3) + 8 * (r >> 2)) * ES;
                        s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, bwd_bias_elem<ES, BF16>(brs, vo) - Lq));
                        s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, bwd_bias_elem<ES, BF16>(brs, vo + 32 * ES) - Lq));
                    }
                };
                if (p.bias_kind == 1) one(std::integral_constant<int, 2>{});
                else if (p.bias_kind == 2) one(std::integral_constant<int, 4>{});
                else one(std::integral_constant<int, 1>{});
            } else if constexpr (tiled) {
                // the wave's 32 x 64 tile of the bias, staged by LDS-DMA at the top of the body (bias_issue below), read back in groups of four kv
                const lds_char_ptr bimg = smem + (DBUF ? 2 : 1) * STAGEB + wave * p.bias_img;
                __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0): the pieces have landed (wave-private image)
#pragma unroll
                for (int G = 0; G < 8; ++G) {
                    float bv[4];
                    if (p.bias_kind == 1) bwd_bias_tile_dq_read<2, BF16>(bimg, l31, hi, G >> 2, G & 3, bv);
                    else if (p.bias_kind == 2) bwd_bias_tile_dq_read<4, BF16>(bimg, l31, hi, G >> 2, G & 3, bv);
                    else bwd_bias_tile_dq_read<1, BF16>(bimg, l31, hi, G >> 2, G & 3, bv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * (G & 3) + e;
                        if (G >> 2) s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, bv[e] - Lq));
                        else s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, bv[e] - Lq));
                    }
                }
                // (kv >= Nkv: the image holds zeros or a neighbouring row's values, P stays finite; the MASKED block below zeroes those entries)
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float b0 = -Lq, b1 = -Lq;
                    s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, b0));
                    s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, b1));
                }
            }
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kvi = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kvi > lim) s0[r] = 0.f;
                    if (kvi + 32 > lim) s1[r] = 0.f;
                }
            }
            f32x16 d0, d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                     // dP^T = V dO^T
                d0 = mfma16<BF16>(lds_load128(vR + ln.kr_off[ks]), gf[ks], d0);
                d1 = mfma16<BF16>(lds_load128(vR + ln.kr_off[ks] + 32 * ROWB), gf[ks], d1);
            }
            // dS^T / scale = P^T * (dP^T - D); the factor `scale` is applied once, to the finished dQ
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = s0[r] * (d0[r] - Dq);
                s1[r] = s1[r] * (d1[r] - Dq);
            }
            u32x4 df[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                df[0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
                df[1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
                df[2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
                df[3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)                         // dQ^T += K^T dS^T
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const lds_char_ptr va = kT3 + (HDV == HD ? ln.vr_off[dt % L_::DT] : lnv.vr_off[dt]) + 16 * ks * ROWBV;
                    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWBV)));
                    acc[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, df[ks], acc[dt]);
                }
        }
    };

    int t_begin = 0;                       // a part sweeps its share of the whole KV tiles
    if (part >= 0) { t_begin = part * ntiles / p.nsplit; ntiles = (part + 1) * ntiles / p.nsplit; }
    stage_load(t_begin, 0);
    __syncthreads();
    for (int tile = t_begin; tile < ntiles; ++tile) {
        const int st = DBUF ? (tile - t_begin) & 1 : 0;
        if (DBUF && tile + 1 < ntiles) stage_load(tile + 1, st ^ 1);   // destination stage was last read before the previous barrier
        if (!CAUSAL || tile < ntiles_w) tile_body(tile, st, tile >= n_plain);   // (causal: a wave past its diagonal only keeps the barriers)
        __syncthreads();
        if (!DBUF && tile + 1 < ntiles) { stage_load(tile + 1, 0); __syncthreads(); }
    }
    if constexpr (!CAUSAL && NW == 8 && HDV == HD) {
        if (part >= 0) {                   // unscaled f32 partial dQ tile; bwd_merge_kernel sums the parts, scales and rounds once
            store_partial_t<DT>(acc, p.ws + (int64_t)(sidx * p.nsplit + part) * kSplitRows * HD, 32 * wave + l31, hi);
            return;
        }
    }
    if (qrow < p.Nq) {
        uint16_t* op = (uint16_t*)p.dq + b * p.dqs[0] + h * p.dqs[1] + (int64_t)qrow * p.dqs[2];
        store_acc_t<BF16, DT>(acc, op, hi, scale, p.D, vcol0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// dV (WANT_DK = false) or dK (WANT_DK = true): workgroup = 256 KV rows (8 waves x 32), sweep over Q tiles of 64.
// LDS per stage: Q row-form | (dK: dO row-form | Q tr-form)  (dV: dO tr-form) | L[64] | D[64]; two stages.
// BOTH (WANT_DK and D = 64, where two accumulators fit): dK and dV in one sweep — S and P are formed once, the stage
// additionally carries dO in tr-form (Q row | dO row | Q tr | dO tr).
// HDV < HD (HD = 512): the workgroup produces the HDV-column slab blockIdx.y of dK / dV; S (and dP) are contracted over the whole head
// dim, only the transposed-read image and the accumulator are slab-sized (see bwd_dq_kernel).
template <int HD, bool BF16, bool CAUSAL, bool WANT_DK, int NW = 8, bool BOTH = false, int HDV = HD, int BIAS = 0, int KSN = HD / 16, int DTN = HDV / 32>
__global__ __launch_bounds__(NW * 64, NW / 4) void bwd_dkv_kernel(const BwdParams p) {
    static_assert(!BOTH || WANT_DK, "the fused pass is the dK pass plus a dV accumulator");
    static_assert(!BOTH || HDV == HD, "slabs exist for the separate passes only");
    using L_ = BwdLane<HD, NW>;
    using LV_ = BwdLane<HDV, NW>;
    constexpr int kRows = NW * 32;
    constexpr bool DBUF = NW == 8;
    constexpr int NPASS = L_::NPASS, KS = KSN, ROWB = L_::ROWB, TILEB = L_::TILEB;
    constexpr int NPASSV = LV_::NPASS, DT = DTN, ROWBV = LV_::ROWB, TILEBV = LV_::TILEB;
    static_assert(KSN <= L_::KS && DTN <= LV_::DT, "trimmed loop bounds");
    constexpr int NT = BOTH ? 4 : WANT_DK ? 3 : 2;                 // images per stage; the last one is the transposed-read image
    constexpr int TROFF = (NT - 1) * TILEB;                        // ... which starts here (BOTH: Q tr at 2, dO tr at 3 tiles)
    constexpr int LOFF = HDV == HD ? NT * TILEB : TROFF + TILEBV;   // L | D values of the tile
    constexpr int STAGEB = LOFF + 512;
    const int vcol0 = HDV == HD ? 0 : blockIdx.y * HDV;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const lds_char_ptr smem = (lds_char_ptr)smem_generic;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (head, kv block); causal: the FIRST kv block sweeps the most Q tiles -> ascending kv block, across heads
    const int nbh = p.B * p.H;
    int bid = blockIdx.x;
    int part = -1, sidx = 0;               // split last round (fused pass only): see bwd_dq_kernel
    if constexpr (!CAUSAL && NW == 8 && BOTH) {
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
    const int kv0 = kblk * kRows, kvw0 = kv0 + 32 * wave, kvrow = kvw0 + l31;   // this lane's KV row
    const int kr = kvrow < p.Nkv ? kvrow : p.Nkv - 1;

    L_ ln;
    ln.init(tid, lane, p.D);
    LV_ lnv;
    if constexpr (HDV != HD) lnv.init(tid, lane, p.D, vcol0);
    u32x4 kf[KS], vf[WANT_DK ? KS : 1];
    {
        const uint16_t* kp = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1] + (int64_t)kr * p.ks[2];
        const uint16_t* vp = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1] + (int64_t)kr * p.vs[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool in = 16 * ks + 8 * hi < p.D;
            kf[ks] = in ? *(const u32x4*)(kp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
            if constexpr (WANT_DK) vf[ks] = in ? *(const u32x4*)(vp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    const uint16_t* qbase = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1];
    const uint16_t* gbase = (const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1];
    const auto qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, p.q_bytes, 0x00020000);
    const auto grs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, p.do_bytes, 0x00020000);
    const auto lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + b * p.ls[0] + h * p.ls[1]), 0, p.l_bytes, 0x00020000);
    const auto drs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.delta + b * p.ls[0] + h * p.ls[1]), 0, p.l_bytes, 0x00020000);
    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, g_rowb = (uint32_t)p.dos[2] * 2u;

    // Q tiles that can touch this workgroup's KV rows: causal keeps only q >= kv (top-left aligned)
    int ntiles = (p.Nq + kKvTile - 1) / kKvTile;
    int tile0 = CAUSAL ? kv0 / kKvTile : 0;
    if (part >= 0) { tile0 = part * ntiles / p.nsplit; ntiles = (part + 1) * ntiles / p.nsplit; }   // a part sweeps its share of the Q tiles
    const int tile0_w = CAUSAL ? kvw0 / kKvTile : 0;     // this wave's first useful tile

    // per-lane source offsets of the staging loads within a tile (loop-invariant); the tile's own byte offset rides in soffset
    // (kept for the 8-wave kernels only: the 4-wave ones have 8 passes per tile and no registers to spare)
    constexpr int NKEEP = NW == 8 ? NPASS : 1;
    uint32_t qr_src[NKEEP], qt_src[NKEEP], gr_src[NKEEP], gt_src[NKEEP];
    if constexpr (NW == 8) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            qr_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.r_src[i];
            qt_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.t_src[i];
            gr_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.r_src[i];
            gt_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.t_src[i];
        }
    }
    auto stage_load = [&](int tile, int stage) __attribute__((always_inline)) {
        const lds_char_ptr base = smem + stage * STAGEB;
        const uint32_t qsoff = (uint32_t)tile * kKvTile * q_rowb, gsoff = (uint32_t)tile * kKvTile * g_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const lds_char_ptr dst = base + (wave * 64 + NW * 64 * i) * 16;
            if constexpr (NW == 8) {
                dma16_to_lds3(qrs, dst, FA2_TILE_OFF(qr_src[i], qsoff));                               // Q row-form
                if constexpr (WANT_DK) {
                    dma16_to_lds3(grs, dst + TILEB, FA2_TILE_OFF(gr_src[i], gsoff));                   // dO row-form
                    dma16_to_lds3(qrs, dst + 2 * TILEB, FA2_TILE_OFF(qt_src[i], qsoff));               // Q tr-form
                    if constexpr (BOTH) dma16_to_lds3(grs, dst + 3 * TILEB, FA2_TILE_OFF(gt_src[i], gsoff));   // dO tr-form
                } else {
                    dma16_to_lds3(grs, dst + TILEB, FA2_TILE_OFF(gt_src[i], gsoff));                   // dO tr-form
                }
            } else {
                const uint32_t qrow_b = (uint32_t)(tile * kKvTile + ln.rowi[i]) * q_rowb;
                const uint32_t grow_b = (uint32_t)(tile * kKvTile + ln.rowi[i]) * g_rowb;
                dma16_to_lds3(qrs, dst, qrow_b + ln.r_src[i], 0);
                if constexpr (WANT_DK) {
                    dma16_to_lds3(grs, dst + TILEB, grow_b + ln.r_src[i], 0);
                    if constexpr (HDV == HD) dma16_to_lds3(qrs, dst + 2 * TILEB, qrow_b + ln.t_src[i], 0);
                    if constexpr (BOTH) dma16_to_lds3(grs, dst + 3 * TILEB, grow_b + ln.t_src[i], 0);   // dO tr-form (fused pass of the trimmed head dims 129..224)
                } else {
                    if constexpr (HDV == HD) dma16_to_lds3(grs, dst + TILEB, grow_b + ln.t_src[i], 0);
                }
            }
        }
        if constexpr (HDV != HD) {            // the slab's transposed-read image (Q for dK, dO for dV) has its own geometry
#pragma unroll
            for (int i = 0; i < NPASSV; ++i) {
                const lds_char_ptr dst = base + TROFF + (wave * 64 + NW * 64 * i) * 16;
                if constexpr (WANT_DK) dma16_to_lds3(qrs, dst, (uint32_t)(tile * kKvTile + lnv.rowi[i]) * q_rowb + lnv.t_src[i], 0);
                else dma16_to_lds3(grs, dst, (uint32_t)(tile * kKvTile + lnv.rowi[i]) * g_rowb + lnv.t_src[i], 0);
            }
        }
        const uint32_t lsoff = (uint32_t)tile * kKvTile * 4u;
        if (wave == 0) dma4_to_lds(lrs, base + LOFF, FA2_TILE_OFF((uint32_t)lane * 4u, lsoff));
        if (WANT_DK && wave == 1) dma4_to_lds(drs, base + LOFF + 256, FA2_TILE_OFF((uint32_t)lane * 4u, lsoff));
    };

    f32x16 acc[DT], accv[BOTH ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[dt][r] = 0.f; if constexpr (BOTH) accv[dt][r] = 0.f; }
    const float c = p.c, scale = p.scale;

    // Q tiles from first_plain on lie entirely at or below this wave's KV rows' diagonal (q >= kv for every pair)
    const int first_plain = CAUSAL ? (kvw0 + 31 + kKvTile - 1) / kKvTile : 0;

    // element form of a bias broadcast over the Q rows (row stride 0: a [B, 1, 1, Nkv] key-padding mask, the mask of SD cross-attention): the
    // score's bias depends on the lane's KV row only — ONE load before the sweep instead of 32 per tile and lane
    // (BIAS = 3, an instantiation of its own: with the per-score loads of BIAS = 1 in the same kernel the fused pass spilled 300 bytes per lane)
    [[maybe_unused]] float bias_row = 0.f;
    if constexpr (BIAS == 3) {
        if (p.bias_kind == 1) bias_row = bwd_bias_elem<2, BF16>(bwd_bias_rsrc<2>(p, b, h), (uint32_t)kr * 2u);
        else if (p.bias_kind == 2) bias_row = bwd_bias_elem<4, BF16>(bwd_bias_rsrc<4>(p, b, h), (uint32_t)kr * 4u);
        else bias_row = bwd_bias_elem<1, BF16>(bwd_bias_rsrc<1>(p, b, h), (uint32_t)kr);
    }

    auto tile_body = [&](int tile, int st, bool masked) __attribute__((always_inline)) {       // one body: see bwd_dq_kernel
        {
            constexpr bool tiled = BIAS == 2;
            lds_char_ptr bimg = smem;
            if constexpr (tiled) {
                {                           // this wave's 64 x 32 tile of the bias (Q tile rows x its own KV rows): in flight under the S products
                    bimg = smem + (DBUF ? 2 : 1) * STAGEB + wave * p.bias_img;
                    if (p.bias_kind == 1) bwd_bias_tile_kv_load<2>(p, b, h, tile * kKvTile, kvw0, lane, bimg);
                    else if (p.bias_kind == 2) bwd_bias_tile_kv_load<4>(p, b, h, tile * kKvTile, kvw0, lane, bimg);
                    else bwd_bias_tile_kv_load<1>(p, b, h, tile * kKvTile, kvw0, lane, bimg);
                }
            }
            const lds_char_ptr qR = smem + st * STAGEB;
            const lds_char_ptr lt = qR + LOFF;
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                     // S[q, kv] = Q K^T  (lane = kv)
                s0 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks]), kf[ks], s0);
                s1 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks] + 32 * ROWB), kf[ks], s1);
            }
            f32x16 d0, d1;
            if constexpr (WANT_DK) {
                const lds_char_ptr gR = qR + TILEB;
#pragma unroll
                for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {                 // dP[q, kv] = dO V^T
                    d0 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks]), vf[ks], d0);
                    d1 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks] + 32 * ROWB), vf[ks], d1);
                }
            }
            const int q0t = tile * kKvTile;
            if constexpr (tiled) __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): the bias tile has landed (wave-private image)
            // P = 2^(S c - L[q]); rows q are spread over the registers: q = q0t + (r&3) + 8(r>>2) + 4hi (+32).  The loop is a generic lambda over the
            // bias element size: the kind of a masked call is dispatched ONCE, outside it (inside, the three kinds' code per element spilled).
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            auto p_loop = [&](auto es_t, auto bc_t) __attribute__((always_inline)) {
                constexpr int ES = decltype(es_t)::value;
                constexpr bool BC = decltype(bc_t)::value;      // a bias broadcast over the Q rows: this lane's one value, loaded before the sweep
                [[maybe_unused]] const auto brs = bwd_bias_rsrc<ES>(p, b, h);
                [[maybe_unused]] const uint32_t rowb = (uint32_t)p.bs[2];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 L0 = *(const __attribute__((address_space(3))) f32x4*)(lt + (8 * g4 + 4 * hi) * 4);
                    const f32x4 L1 = *(const __attribute__((address_space(3))) f32x4*)(lt + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        float b0 = -L0[e], b1 = -L1[e];
                        if constexpr (BIAS) {        // (a fully masked row, L = -inf: P = 0)
                            const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi;
                            float t0, t1;
                            if constexpr (tiled) {
                                t0 = bwd_bias_tile_kv_read<ES, BF16>(bimg, ql, l31);
                                t1 = bwd_bias_tile_kv_read<ES, BF16>(bimg, ql + 32, l31);
                            } else if constexpr (BC) {
                                t0 = t1 = bias_row;
                            } else {
                                // element form: bounds-checked buffer loads, 32-bit offsets (rows >= Nq are out of range and read 0; their P is never used)
                                const uint32_t vq = (uint32_t)(q0t + ql) * rowb + (uint32_t)kr;
                                t0 = bwd_bias_elem<ES, BF16>(brs, vq * ES);
                                t1 = bwd_bias_elem<ES, BF16>(brs, (vq + 32 * rowb) * ES);
                            }
                            b0 = L0[e] == -__builtin_inff() ? -__builtin_inff() : b0 + t0;
                            b1 = L1[e] == -__builtin_inff() ? -__builtin_inff() : b1 + t1;
                        }
                        s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, b0));
                        s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, b1));
                    }
                }
            };
            if constexpr (BIAS) {
                if constexpr (BIAS == 3) p_loop(std::integral_constant<int, 1>{}, std::true_type{});
                else if (p.bias_kind == 1) p_loop(std::integral_constant<int, 2>{}, std::false_type{});
                else if (p.bias_kind == 2) p_loop(std::integral_constant<int, 4>{}, std::false_type{});
                else p_loop(std::integral_constant<int, 1>{}, std::false_type{});
            } else {
                p_loop(std::integral_constant<int, 2>{}, std::false_type{});
            }
            if (CAUSAL && masked) {               // causal: pairs with kv > q contribute nothing (wave-uniform branch)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = q0t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kvrow > qi) s0[r] = 0.f;
                    if (kvrow > qi + 32) s1[r] = 0.f;
                }
            }
            if constexpr (WANT_DK) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 D0 = *(const __attribute__((address_space(3))) f32x4*)(lt + 256 + (8 * g4 + 4 * hi) * 4);
                    const f32x4 D1 = *(const __attribute__((address_space(3))) f32x4*)(lt + 256 + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        if constexpr (BOTH) { d0[r] = s0[r] * (d0[r] - D0[e]); d1[r] = s1[r] * (d1[r] - D1[e]); }   // dS / scale, P kept
                        else {                        // dS / scale; `scale` is applied once, to the finished dK
                            s0[r] = s0[r] * (d0[r] - D0[e]);
                            s1[r] = s1[r] * (d1[r] - D1[e]);
                        }
                    }
                }
            }
            u32x4 xf[4];   // P (dV) or dS (dK) as B fragments: contraction index = q
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xf[0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
                xf[1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
                xf[2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
                xf[3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
            }
            if constexpr (BOTH) {                                  // dV^T += dO^T P with the P fragments above ...
                const lds_char_ptr gT = qR + 3 * TILEB;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const lds_char_ptr va = gT + ln.vr_off[dt] + 16 * ks * ROWB;
                        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                        const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB)));
                        accv[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, xf[ks], accv[dt]);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {                      // ... then the fragments are re-packed from dS for dK
                    xf[0][i] = pack2<BF16>(d0[2 * i], d0[2 * i + 1]);
                    xf[1][i] = pack2<BF16>(d0[8 + 2 * i], d0[8 + 2 * i + 1]);
                    xf[2][i] = pack2<BF16>(d1[2 * i], d1[2 * i + 1]);
                    xf[3][i] = pack2<BF16>(d1[8 + 2 * i], d1[8 + 2 * i + 1]);
                }
            }
            const lds_char_ptr tT = qR + (WANT_DK ? 2 : 1) * TILEB;       // Q tr-form (dK) or dO tr-form (dV)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)                         // dK^T += Q^T dS   /   dV^T += dO^T P
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const lds_char_ptr va = tT + (HDV == HD ? ln.vr_off[dt % L_::DT] : lnv.vr_off[dt]) + 16 * ks * ROWBV;
                    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWBV)));
                    acc[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, xf[ks], acc[dt]);
                }
        }
    };

    // A wave whose 32 KV rows all lie past Nkv (cross-attention on a text prompt: 77 keys in a 256-row workgroup — five of eight waves) stages and
    // syncs, nothing else: its accumulators stay zero (a whole item never stores them, a part stores the zeros).  Round 6: that pass is issue-bound
    const bool wave_live = kvw0 < p.Nkv;
    if (tile0 < ntiles) stage_load(tile0, 0);
    __syncthreads();
    for (int tile = tile0; tile < ntiles; ++tile) {
        const int st = DBUF ? (tile - tile0) & 1 : 0;
        if (DBUF && tile + 1 < ntiles) stage_load(tile + 1, st ^ 1);
        if ((!CAUSAL || tile >= tile0_w) && wave_live) tile_body(tile, st, CAUSAL && tile < first_plain);
        __syncthreads();
        if (!DBUF && tile + 1 < ntiles) { stage_load(tile + 1, 0); __syncthreads(); }
    }
    if constexpr (!CAUSAL && NW == 8 && BOTH) {
        if (part >= 0) {                   // unscaled f32 partial dK and dV tiles
            const int64_t slot = sidx * p.nsplit + part, ntile = (int64_t)p.split_items * p.nsplit;
            store_partial_t<DT>(acc, p.ws + slot * kSplitRows * HD, 32 * wave + l31, hi);
            store_partial_t<DT>(accv, p.ws + (ntile + slot) * kSplitRows * HD, 32 * wave + l31, hi);
            return;
        }
    }
    if (kvrow < p.Nkv) {
        uint16_t* op = WANT_DK ? (uint16_t*)p.dk + b * p.dks[0] + h * p.dks[1] + (int64_t)kvrow * p.dks[2]
                               : (uint16_t*)p.dv + b * p.dvs[0] + h * p.dvs[1] + (int64_t)kvrow * p.dvs[2];
        store_acc_t<BF16, DT>(acc, op, hi, WANT_DK ? scale : 1.0f, p.D, vcol0);
        if constexpr (BOTH) {
            uint16_t* ov = (uint16_t*)p.dv + b * p.dvs[0] + h * p.dvs[1] + (int64_t)kvrow * p.dvs[2];
            store_acc_t<BF16, DT>(accv, ov, hi, 1.0f, p.D);
        }
    }
}

// Mid-tile barrier of the pair pass: only the LDS hand-over of P has to be complete — unlike __syncthreads() it does not wait for the
// next tile's staging loads in flight (vmcnt), which have the whole tile to land (the end-of-tile __syncthreads() drains them).
__device__ __forceinline__ void pair_mid_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ---------------------------------------------------------------------------------------------------------
// dK AND dV in one sweep at head dims 65..128, where one wave cannot hold both accumulators: wave PAIRS.
// A workgroup owns 128 KV rows; waves g and g + 4 (same SIMD) share the 32 KV rows of group g:
//   wave g     ("P side")   S = Q K^T (K rows as B fragments), P = 2^(S c - L), dV^T += dO^T P
//   wave g + 4 ("dS side")  dP = dO V^T (V rows as B fragments), dS = P (dP - D), dK^T += Q^T dS
// P crosses from the first to the second wave once per tile through a 4-KiB LDS slot per pair, as the very 16-bit B
// fragments the dV product consumes (same lane, same registers: S^T and dP^T have the same layout) — so S and P are formed once
// for both products: 4 GEMMs per (kv, q) pair instead of the 5 of the two separate passes, 7 instead of 8 for the whole backward.
// Each tile has two phases separated by a barrier: {S, exp | dP} and {dV | dS, dK}; the transcendental / VALU stretch of one wave
// of a pair runs beside the MFMAs of the other.  Stage: Q row | dO row | Q tr | dO tr | L | D, two stages (129 KiB) + 16 KiB of slots.
template <int HD, bool BF16, bool CAUSAL, int KSN = HD / 16, int DTN = HD / 32>
__global__ __launch_bounds__(512, 2) void bwd_dkv_pair_kernel(const BwdParams p) {
    constexpr int NW = 8;
    using L_ = BwdLane<HD, NW>;
    constexpr int kRows = 128;
    constexpr int NPASS = L_::NPASS, KS = KSN, DT = DTN, ROWB = L_::ROWB, TILEB = L_::TILEB;
    constexpr int NT = 4, STAGEB = NT * TILEB + 512, XCHB = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const lds_char_ptr smem = (lds_char_ptr)smem_generic;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave & 3;
    const bool ds_side = wave >= 4;
    const lds_char_ptr xch = smem + 2 * STAGEB + grp * XCHB + lane * 16;

    const int nbh = p.B * p.H, bid = blockIdx.x;
    int bh, kblk;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;
        if (CAUSAL) { bh = (bid & 7) + 8 * (slot % hpx); kblk = slot / hpx; }
        else { bh = (bid & 7) + 8 * (slot / p.nblk); kblk = slot % p.nblk; }
    } else if (CAUSAL) { bh = bid % nbh; kblk = bid / nbh; }
    else { bh = bid / p.nblk; kblk = bid % p.nblk; }
    const int b = bh / p.H, h = bh % p.H;
    const int kv0 = kblk * kRows, kvw0 = kv0 + 32 * grp, kvrow = kvw0 + l31;
    const int kr = kvrow < p.Nkv ? kvrow : p.Nkv - 1;

    L_ ln;
    ln.init(tid, lane, p.D);
    u32x4 bf[KS];            // the pair's own rows as B fragments: K on the P side, V on the dS side
    {
        const uint16_t* rp = ds_side ? (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1] + (int64_t)kr * p.vs[2]
                                     : (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1] + (int64_t)kr * p.ks[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bf[ks] = 16 * ks + 8 * hi < p.D ? *(const u32x4*)(rp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
    }
    const uint16_t* qbase = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1];
    const uint16_t* gbase = (const uint16_t*)p.dout + b * p.dos[0] + h * p.dos[1];
    const auto qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, p.q_bytes, 0x00020000);
    const auto grs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, p.do_bytes, 0x00020000);
    const auto lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + b * p.ls[0] + h * p.ls[1]), 0, p.l_bytes, 0x00020000);
    const auto drs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.delta + b * p.ls[0] + h * p.ls[1]), 0, p.l_bytes, 0x00020000);
    const uint32_t q_rowb = (uint32_t)p.qs[2] * 2u, g_rowb = (uint32_t)p.dos[2] * 2u;

    const int ntiles = (p.Nq + kKvTile - 1) / kKvTile;
    const int tile0 = CAUSAL ? kv0 / kKvTile : 0;
    const int first_plain = CAUSAL ? (kvw0 + 31 + kKvTile - 1) / kKvTile : 0;

    // per-lane source offsets of the staging loads within a tile (loop-invariant); the tile's own byte offset rides in soffset
    uint32_t qr_src[NPASS], qt_src[NPASS], gr_src[NPASS], gt_src[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        qr_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.r_src[i];
        qt_src[i] = (uint32_t)ln.rowi[i] * q_rowb + ln.t_src[i];
        gr_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.r_src[i];
        gt_src[i] = (uint32_t)ln.rowi[i] * g_rowb + ln.t_src[i];
    }
    auto stage_load = [&](int tile, int stage) __attribute__((always_inline)) {
        const lds_char_ptr base = smem + stage * STAGEB;
        const uint32_t qsoff = (uint32_t)tile * kKvTile * q_rowb, gsoff = (uint32_t)tile * kKvTile * g_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const lds_char_ptr dst = base + (wave * 64 + NW * 64 * i) * 16;
            dma16_to_lds3(qrs, dst, FA2_TILE_OFF(qr_src[i], qsoff));                        // Q row-form
            dma16_to_lds3(grs, dst + TILEB, FA2_TILE_OFF(gr_src[i], gsoff));                // dO row-form
            dma16_to_lds3(qrs, dst + 2 * TILEB, FA2_TILE_OFF(qt_src[i], qsoff));            // Q tr-form
            dma16_to_lds3(grs, dst + 3 * TILEB, FA2_TILE_OFF(gt_src[i], gsoff));            // dO tr-form
        }
        const uint32_t lsoff = (uint32_t)tile * kKvTile * 4u;
        if (wave == 0) dma4_to_lds(lrs, base + NT * TILEB, FA2_TILE_OFF((uint32_t)lane * 4u, lsoff));
        if (wave == 1) dma4_to_lds(drs, base + NT * TILEB + 256, FA2_TILE_OFF((uint32_t)lane * 4u, lsoff));
    };

    f32x16 acc[DT];          // dV^T (P side) or dK^T / scale (dS side)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const float c = p.c;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_cptr;
    typedef __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;

    auto accumulate = [&](lds_char_ptr tT, const u32x4 (&xf)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const lds_char_ptr va = tT + ln.vr_off[dt] + 16 * ks * ROWB;
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va)));
                const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB)));
                acc[dt] = mfma16<BF16>((u32x4){lo[0], lo[1], h2[0], h2[1]}, xf[ks], acc[dt]);
            }
    };

    // One tile loop PER ROLE (the role branch outside the loop): with both roles inside one loop the allocator copied the 64
    // accumulator registers at the join in every iteration.  Both loops hold the same two barriers per tile.
    // (A pair whose rows all lie above a tile's q range — the first tile of groups 2, 3 under a causal mask — runs the tile like
    //  any other: every P is masked to zero, so it contributes nothing.)
    if (tile0 < ntiles) stage_load(tile0, 0);
    __syncthreads();
    if (!ds_side) {
        for (int tile = tile0; tile < ntiles; ++tile) {
            const int st = (tile - tile0) & 1;
            if (tile + 1 < ntiles) stage_load(tile + 1, st ^ 1);
            const lds_char_ptr qR = smem + st * STAGEB;
            const lds_char_ptr lt = qR + NT * TILEB;
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
            __builtin_amdgcn_s_setprio(3);            // the S products ahead of the partner wave's dP products: the exp stretch then runs beside those
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                         // S[q, kv] = Q K^T  (lane = kv)
                s0 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks]), bf[ks], s0);
                s1 = mfma16<BF16>(lds_load128(qR + ln.kr_off[ks] + 32 * ROWB), bf[ks], s1);
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 L0 = *(lds_f32x4_cptr)(lt + (8 * g4 + 4 * hi) * 4);
                const f32x4 L1 = *(lds_f32x4_cptr)(lt + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -L0[e]));
                    s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -L1[e]));
                }
            }
            if (CAUSAL && tile < first_plain) {                       // pairs with kv > q contribute nothing (wave-uniform branch)
                const int q0t = tile * kKvTile;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = q0t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kvrow > qi) s0[r] = 0.f;
                    if (kvrow > qi + 32) s1[r] = 0.f;
                }
            }
            u32x4 xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xf[0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
                xf[1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
                xf[2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
                xf[3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(lds_u32x4_ptr)(xch + 1024 * i) = xf[i];
            pair_mid_barrier();                                       // P is in the pair's slot
            accumulate(qR + 3 * TILEB, xf);                           // dV^T += dO^T P
            __syncthreads();
        }
    } else {
        for (int tile = tile0; tile < ntiles; ++tile) {
            const int st = (tile - tile0) & 1;
            if (tile + 1 < ntiles) stage_load(tile + 1, st ^ 1);
            const lds_char_ptr qR = smem + st * STAGEB;
            const lds_char_ptr lt = qR + NT * TILEB;
            const lds_char_ptr gR = qR + TILEB;
            f32x16 d0, d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                         // dP[q, kv] = dO V^T
                d0 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks]), bf[ks], d0);
                d1 = mfma16<BF16>(lds_load128(gR + ln.kr_off[ks] + 32 * ROWB), bf[ks], d1);
            }
            pair_mid_barrier();
            u32x4 xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *(lds_u32x4_ptr)(xch + 1024 * i);
            float pv[32];                                              // P as the dV product sees it (16-bit), in register order
            if constexpr (BF16) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pv[8 * j + 2 * i] = __uint_as_float(xf[j][i] << 16);
                        pv[8 * j + 2 * i + 1] = __uint_as_float(xf[j][i] & 0xffff0000u);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x8 hv = __builtin_bit_cast(f16x8, xf[j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[8 * j + e] = (float)hv[e];
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 D0 = *(lds_f32x4_cptr)(lt + 256 + (8 * g4 + 4 * hi) * 4);
                const f32x4 D1 = *(lds_f32x4_cptr)(lt + 256 + (32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    d0[r] = pv[r] * (d0[r] - D0[e]);                   // dS / scale; `scale` is applied once, to the finished dK
                    d1[r] = pv[16 + r] * (d1[r] - D1[e]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xf[0][i] = pack2<BF16>(d0[2 * i], d0[2 * i + 1]);
                xf[1][i] = pack2<BF16>(d0[8 + 2 * i], d0[8 + 2 * i + 1]);
                xf[2][i] = pack2<BF16>(d1[2 * i], d1[2 * i + 1]);
                xf[3][i] = pack2<BF16>(d1[8 + 2 * i], d1[8 + 2 * i + 1]);
            }
            accumulate(qR + 2 * TILEB, xf);                            // dK^T += Q^T dS
            __syncthreads();
        }
    }
    if (kvrow < p.Nkv) {
        uint16_t* op = ds_side ? (uint16_t*)p.dk + b * p.dks[0] + h * p.dks[1] + (int64_t)kvrow * p.dks[2]
                               : (uint16_t*)p.dv + b * p.dvs[0] + h * p.dvs[1] + (int64_t)kvrow * p.dvs[2];
        store_acc_t<BF16, DT>(acc, op, hi, ds_side ? p.scale : 1.0f, p.D);
    }
}

// Sum of the parts of a split backward pass (fa2_bwd_ws): out = round(mul * sum_i partial_i) for every row of a split item.  blockIdx.y
// selects the tensor (the fused dK / dV pass leaves two sets of tiles).  One thread per (row, 8 output columns), as fwd_combine_kernel.
struct BwdMergeParams {
    const float* ws;
    void* out[2];
    int64_t os[2][3];        // element strides of the outputs: batch, head, row
    float mul[2];
    int H, nbh, nblk, nrows, D;
    int full_items, split_items, nsplit;
};

template <int HD, bool BF16>
__global__ __launch_bounds__(256) void bwd_merge_kernel(const BwdMergeParams p) {
    constexpr int CPR = HD / 8;
    const int t = blockIdx.x * 256 + threadIdx.x, which = blockIdx.y;
    const int sidx = t / (kSplitRows * CPR), rem = t % (kSplitRows * CPR);
    const int row = rem / CPR, c8 = rem % CPR;
    if (sidx >= p.split_items) return;
    const int bid = p.full_items + sidx;
    int bh, blk;                         // the non-causal block order of bwd_dq_kernel / bwd_dkv_kernel
    if ((p.nbh & 7) == 0) { const int slot = bid >> 3; bh = (bid & 7) + 8 * (slot / p.nblk); blk = slot % p.nblk; }
    else { bh = bid / p.nblk; blk = bid % p.nblk; }
    const int r = blk * kSplitRows + row;
    if (r >= p.nrows || 8 * c8 >= p.D) return;
    const float* w = p.ws + ((int64_t)which * p.split_items * p.nsplit + (int64_t)sidx * p.nsplit) * kSplitRows * HD + (c8 * kSplitRows + row) * 8;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < p.nsplit; ++i) {
        const f32x4 lo = *(const f32x4*)(w + (int64_t)i * kSplitRows * HD), hi4 = *(const f32x4*)(w + (int64_t)i * kSplitRows * HD + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] += lo[e]; o[4 + e] += hi4[e]; }
    }
    const float m = p.mul[which];
    const u32x4 w16 = {pack2<BF16>(o[0] * m, o[1] * m), pack2<BF16>(o[2] * m, o[3] * m), pack2<BF16>(o[4] * m, o[5] * m), pack2<BF16>(o[6] * m, o[7] * m)};
    const int b = bh / p.H, h = bh % p.H;
    *(u32x4*)((uint16_t*)p.out[which] + b * p.os[which][0] + h * p.os[which][1] + (int64_t)r * p.os[which][2] + 8 * c8) = w16;
}

}  // namespace fa2
