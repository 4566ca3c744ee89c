// fa2_fwd_kernel.hip.h — FlashAttention-2 forward for MI355X (gfx950 / CDNA4), device side.
//
// Replaces the reference's fwd_kernel + mul_A_BT + mul_add_A_B (rocwmma_fattn/kernel_fp16.cu:115-232,
// :306-544; bf16 twin kernel_bf16.cu:138-255, :329-577).  Written for wave64 + MFMA, nothing is
// shared with the RDNA3/WMMA code.
//
// Work decomposition
//   workgroup = 8 waves (512 threads) = 256 consecutive Q rows of one (batch, head);
//   wave w owns Q rows [32w, 32w+32) for the whole KV sweep; KV is streamed in tiles of 64 rows.
//   Q fragments live in registers; K and V tiles are staged global -> registers -> LDS (double
//   buffered, one barrier per tile); S, P, m, l and the O accumulator never leave registers.
//
// MFMA orientation ("swapped" products, so that a Q row is lane-local)
//   S^T[kv, q] = K[kv, :] . Q[q, :]      mfma_f32_32x32x16(A = K frag, B = Q frag)
//   O^T[d , q] = V^T[d, kv] . P^T[kv, q] mfma_f32_32x32x16(A = V^T frag, B = P frag)
//   C/D layout of a 32x32 tile: column n = lane & 31, row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
//   Hence lane (q = lane & 31, hi = lane >> 5) holds, for its Q row q, the scores of kv rows
//   {(r&3) + 8(r>>2) + 4hi}: row max / row sum are in-lane reductions plus ONE half-wave exchange
//   (v_permlane32_swap), and the softmax rescale factor is a per-lane scalar.
//   A/B layout: lane holds 8 consecutive k: row/col = lane & 31, k = 8 * (lane >> 5) + j.
//   The contraction index of P.V is kv; k-slot j of lane-half hi is bound to
//   kv = 16*ks + (j&3) + 8*(j>>2) + 4*hi, which is exactly how S^T leaves the QK^T MFMA, so P needs
//   no cross-lane movement; V^T fragments with the same binding come from ds_read_b64_tr_b16.
//
// LDS images (row = kv row inside the tile, ROWB = HD*2 bytes per row)
//   K: 16-byte granule gi of row r stored at r*ROWB + ((gi ^ fK(r)) << 4)   (conflict-free b128 reads
//      by 16 lanes holding 16 different rows)
//   V: 64-byte chunk ci of row r stored at r*ROWB + ((ci ^ fV(r)) << 6) + (byte & 63)  (conflict-free
//      transpose reads: 32 lanes read 4 rows x 64 B)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;         // waves per workgroup
constexpr int kQRowsPerWave = 32; // one 32-wide MFMA column block
constexpr int kQBlock = kWaves * kQRowsPerWave;  // 256 Q rows per workgroup
constexpr int kKvTile = 64;       // KV rows per tile
constexpr int kThreads = kWaves * 64;

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, H, Nq, Nkv;
    int64_t qs[3], ks[3], vs[3], os[3];  // element strides: batch, head, row
    int64_t ls[2];                       // lse strides: batch, head
    float c;                             // |scale| * log2(e)
    int negate_q;                        // scale < 0: fold the sign into Q
    int nqblk;                           // ceil(Nq / kQBlock)
    uint32_t k_bytes, v_bytes;           // addressable bytes of one head's K / V matrix
};

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two f32 -> one dword of two 16-bit floats, round-to-nearest-even (v_cvt_pk_{f16,bf16}_f32)
template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f32x2 x = {lo, hi};
    if constexpr (BF16)
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    else
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
}

__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// LDS image geometry for head dim HD
template <int HD>
struct Geo {
    static constexpr int ROWB = HD * 2;                       // bytes per tile row
    static constexpr int TILEB = kKvTile * ROWB;              // bytes per K (or V) tile
    static constexpr int G = ROWB / 16;                       // 16-B granules per row
    static constexpr int RPB = (ROWB >= 256) ? 1 : 256 / ROWB;  // rows per 256-B bank row
    static constexpr int KMASK = (G < 16 ? G : 16) - 1;
    static constexpr int C = ROWB / 64;                       // 64-B chunks per row
    static constexpr int VMASK = (C < 4 ? C : 4) - 1;
    static constexpr int NPASS = (kKvTile * G) / kThreads;    // 16-B staging loads per thread per tile
    static constexpr int KS_QK = HD / 16;                     // MFMA k-steps of Q.K^T
    static constexpr int DT = HD / 32;                        // 32-wide d blocks of O
    static_assert(NPASS >= 1, "head dim too small for the 512-thread staging pattern");
    __device__ static __forceinline__ int k_off(int row, int gi) {
        return row * ROWB + ((gi ^ ((row / RPB) & KMASK)) << 4);
    }
    __device__ static __forceinline__ int v_off(int row, int colbyte) {
        return row * ROWB + ((((colbyte >> 6) ^ ((row / RPB) & VMASK))) << 6) + (colbyte & 63);
    }
};

template <int HD, bool BF16, bool CAUSAL>
__global__ __launch_bounds__(kThreads, 2) void fwd_kernel(const FwdParams p) {
    using G_ = Geo<HD>;
    constexpr int ROWB = G_::ROWB, TILEB = G_::TILEB, NPASS = G_::NPASS;
    constexpr int KS_QK = G_::KS_QK, DT = G_::DT;
    // LDS: K buf0 | K buf1 | V buf0 | V buf1
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // ---- workgroup -> (batch*head, q block); blocks of one head share an XCD (bid % 8) so that
    //      the head's K/V stay in that XCD's L2; causal runs the long (late) q blocks first.
    const int nbh = p.B * p.H;
    const int bid = blockIdx.x;
    int bh, qb;
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3;
        bh = (bid & 7) + 8 * (slot / p.nqblk);
        qb = slot % p.nqblk;
    } else {
        bh = bid / p.nqblk;
        qb = bid % p.nqblk;
    }
    if (CAUSAL) qb = p.nqblk - 1 - qb;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qb * kQBlock;
    const int qw0 = q0 + wave * kQRowsPerWave;  // first Q row of this wave
    const int qrow = qw0 + l31;                 // this lane's Q row (may be >= Nq)

    // ---- Q fragments (B operand): lane reads 8 consecutive d of its row per k-step
    u32x4 qf[KS_QK];
    {
        const int qr = qrow < p.Nq ? qrow : p.Nq - 1;
        const uint16_t* qp = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qr * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) qf[ks] = *(const u32x4*)(qp + 16 * ks + 8 * hi);
        if (p.negate_q) {
            const uint32_t sgn = 0x80008000u;
#pragma unroll
            for (int ks = 0; ks < KS_QK; ++ks) qf[ks] ^= (u32x4){sgn, sgn, sgn, sgn};
        }
    }

    // ---- K/V staging: buffer descriptors of this head's matrices (out-of-range rows read 0)
    const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t* vbase = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1];
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes, 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, p.v_bytes, 0x00020000);
    const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;
    uint32_t kg_off[NPASS], vg_off[NPASS];  // per-lane byte offsets into the head matrix, tile 0
    int kw_off[NPASS], vw_off[NPASS];       // per-lane LDS byte offsets inside a tile image
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / G_::G, gi = idx % G_::G;
        kg_off[i] = row * k_rowb + gi * 16;
        vg_off[i] = row * v_rowb + gi * 16;
        kw_off[i] = G_::k_off(row, gi);
        vw_off[i] = G_::v_off(row, gi * 16);
    }

    // ---- per-lane LDS read offsets
    int kr_off[KS_QK];  // K fragment (row l31 of the 32-row half tile), k-step ks
#pragma unroll
    for (int ks = 0; ks < KS_QK; ++ks) kr_off[ks] = G_::k_off(l31, 2 * ks + hi);
    int vr_off[DT];     // V^T fragment via transpose read: rows 4hi + (p>>2), cols 32dt + 16(g&1) + 4(p&3)
    {
        const int pp = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            vr_off[dt] = G_::v_off(4 * hi + (pp >> 2), (32 * dt + 16 * g1 + 4 * (pp & 3)) * 2);
    }

    // ---- KV sweep bounds
    int ntiles = (p.Nkv + kKvTile - 1) / kKvTile;
    if (CAUSAL) {
        const int qmax = (q0 + kQBlock < p.Nq ? q0 + kQBlock : p.Nq) - 1;
        const int nt_c = qmax / kKvTile + 1;
        ntiles = nt_c < ntiles ? nt_c : ntiles;
    }

    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -INFINITY;  // running row max, raw (unscaled) score units
    float l_run = 0.f;        // running row sum, this lane's half of the kv columns only
    const float c = p.c;

    u32x4 kreg[NPASS], vreg[NPASS];
    auto stage_load = [&](int tile) {
        const uint32_t ksoff = (uint32_t)tile * kKvTile * k_rowb;
        const uint32_t vsoff = (uint32_t)tile * kKvTile * v_rowb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kg_off[i], ksoff, 0);
            vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vg_off[i], vsoff, 0);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *(u32x4*)(smem + buf * TILEB + kw_off[i]) = kreg[i];
            *(u32x4*)(smem + (2 + buf) * TILEB + vw_off[i]) = vreg[i];
        }
    };

    // one KV tile against this wave's 32 Q rows; BUF is the LDS buffer holding the tile
    auto compute_tile = [&](int tile, int buf) {
        const int kv0 = tile * kKvTile;
        const char* kt = smem + buf * TILEB;
        const char* vt = smem + (2 + buf) * TILEB;
        // S^T = K Q^T : two 32(kv) x 32(q) tiles
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) {
            const u32x4 a0 = *(const u32x4*)(kt + kr_off[ks]);
            const u32x4 a1 = *(const u32x4*)(kt + kr_off[ks] + 32 * ROWB);
            s0 = mfma16<BF16>(a0, qf[ks], s0);
            s1 = mfma16<BF16>(a1, qf[ks], s1);
        }
        // masks: causal diagonal and the ragged last KV tile (wave-uniform tests)
        const bool need_causal = CAUSAL && (kv0 + kKvTile - 1 > qw0);
        const bool need_tail = kv0 + kKvTile > p.Nkv;
        if (need_causal || need_tail) {
            const int lim_c = CAUSAL ? qrow : 0x7fffffff;  // kv index must be <= lim_c
            const int lim = lim_c < p.Nkv - 1 ? lim_c : p.Nkv - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kvi = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (kvi > lim) s0[r] = -INFINITY;
                if (kvi + 32 > lim) s1[r] = -INFINITY;
            }
        }
        // online softmax (reference: kernel_fp16.cu:434-490), all in f32 registers
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = half_swap_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float mc = m_new * c;
        const float alpha = __builtin_amdgcn_exp2f(m_run * c - mc);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -mc));
            s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -mc));
            rs += s0[r] + s1[r];
        }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
        // P -> 16-bit B fragments; k-step ks uses registers [8(ks&1), 8(ks&1)+8) of tile ks>>1
        u32x4 pf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
            pf[1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
            pf[2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
            pf[3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
        }
        // O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* va = vt + vr_off[dt] + 16 * ks * ROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * ROWB));
                const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi4);
                const u32x4 a = {lo2[0], lo2[1], hi2[0], hi2[1]};
                acc[dt] = mfma16<BF16>(a, pf[ks], acc[dt]);
            }
        }
    };

    // ---- prologue: tile 0 -> LDS buffer 0
    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const bool more = tile + 1 < ntiles;
        if (more) stage_load(tile + 1);  // global loads fly under the MFMA work below
        // causal: a wave whose rows all lie above this tile has nothing to do (still stages + syncs)
        const bool active = !CAUSAL || (tile * kKvTile <= qw0 + kQRowsPerWave - 1);
        if (active) compute_tile(tile, buf);
        if (more) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (reference: kernel_fp16.cu:510-543): O = O / l, lse = m + log2(l) (log2 domain)
    const float l_tot = half_swap_sum(l_run);
    const float inv_l = 1.0f / l_tot;
    if (qrow < p.Nq) {
        uint16_t* op = (uint16_t*)p.o + b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4 += 2) {
                // lane holds d = 32dt + 8*r4 + 4*hi + {0..3} (group r4) and the same for r4+1
                uint32_t a0 = pack2<BF16>(acc[dt][4 * r4 + 0] * inv_l, acc[dt][4 * r4 + 1] * inv_l);
                uint32_t a1 = pack2<BF16>(acc[dt][4 * r4 + 2] * inv_l, acc[dt][4 * r4 + 3] * inv_l);
                uint32_t b0 = pack2<BF16>(acc[dt][4 * r4 + 4] * inv_l, acc[dt][4 * r4 + 5] * inv_l);
                uint32_t b1 = pack2<BF16>(acc[dt][4 * r4 + 6] * inv_l, acc[dt][4 * r4 + 7] * inv_l);
                // half exchange: lower lanes end with 8 consecutive d of group r4, upper of r4+1
                auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const u32x4 w = {x0[0], x1[0], x0[1], x1[1]};
                *(u32x4*)(op + 32 * dt + 8 * (r4 + hi)) = w;
            }
        }
        if (hi == 0)
            p.lse[b * p.ls[0] + h * p.ls[1] + qrow] = m_run * c + __builtin_amdgcn_logf(l_tot);
    }
}

}  // namespace fa2
