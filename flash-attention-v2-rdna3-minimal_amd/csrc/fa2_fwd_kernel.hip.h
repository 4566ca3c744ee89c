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
// Software pipeline (per wave): iteration j issues the QK^T MFMAs of tile j+1 and, independent of
// them, the softmax VALU work of tile j, then the PV MFMAs of tile j — so matrix and vector pipes
// have independent work in the same instruction window.
//
// LDS images (row = kv row inside the tile, ROWB = HD*2 bytes per row)
//   K: 16-byte granule gi of row r stored at r*ROWB + ((gi ^ fK(r)) << 4)   (conflict-free b128 reads
//      by 16 lanes holding 16 different rows)
//   V: 64-byte chunk ci of row r stored at r*ROWB + ((ci ^ fV(r)) << 6) + (byte & 63)  (conflict-free
//      transpose reads: 32 lanes read 4 rows x 64 B)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// ---- tuning knobs (A/B-tested on MI355X with tools/kbench.py; numbers at B2 H16 N4096 D128 fp16) ----
#ifndef FA2_DEFER_THR        // skip the O rescale while the row max grew by <= this (log2 units); <0: always rescale.
#define FA2_DEFER_THR 8.0f   // 0 = exact FA2 (rescale whenever any row's max grows); 8 keeps P <= 2^8 (fp16/bf16 safe): +5 %
#endif
// (The loop is the cross-tile software pipeline: QK^T of tile+1 beside the softmax of tile.  The plain-order variant,
//  1090-1125 TF against 1130-1180, and the FA2_ABL ablation switches that priced the parts of the step are in the git
//  history, up to commit f31c723.)
#ifndef FA2_LDS_DMA          // stage K/V tiles with buffer_load ... lds (no staging VGPRs, no ds_write); the LDS swizzle
#define FA2_LDS_DMA 1        // is applied to the per-lane SOURCE address, the LDS image stays lane-linear.
#endif                       // D=128: +3 % non-causal, +6 % causal; D=64: -3 % -> register staging below FA2_LDS_DMA_MIN_HD
#ifndef FA2_LDS_DMA_MIN_HD
#define FA2_LDS_DMA_MIN_HD 128
#endif
#ifndef FA2_EPI_LDS          // 1: O goes to global memory as whole rows through a wave-private LDS image
#define FA2_EPI_LDS 1
#endif
#ifndef FA2_IGLP             // __builtin_amdgcn_iglp_opt(n) in the steady-state step; -1 = none.  0: +1-2 %; 1: -18 %;
#define FA2_IGLP 0           // explicit uniform sched_group_barrier pipelines (1 MFMA : 4-6 VALU : 1-2 DS): -10 %
#endif
#ifndef FA2_TILE_IN_SOFFSET   // 1: a tile's byte offset rides in the buffer instruction's soffset (no VALU, 7 fewer VGPRs); 0: it is
#define FA2_TILE_IN_SOFFSET 1  // added to the per-lane voffset.  LLVM documents soffset as excluded from the bounds check, but on gfx950
#endif                         // it is included: tests/test_parity_gpu.py::test_ragged_tail_ignores_memory_past_nkv passes with both
#if FA2_TILE_IN_SOFFSET
#define FA2_TILE_OFF(voff, soff) (voff), (soff)
#else
#define FA2_TILE_OFF(voff, soff) (voff) + (soff), 0u
#endif
// Tried and dropped (git history has the code; DESIGN.md §4 the measurements): s_setprio variants, 2- and 3-phase
// ping-pong of the two waves of a SIMD, issuing all LDS fragment reads of a phase up front, packed-f32 softmax math.

namespace fa2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));

constexpr int kQBlock = 256;      // rows per workgroup of the default kernel shapes (8 waves x 32 rows)
constexpr int kKvTile = 64;       // KV rows per tile
// Buffer-load byte offset that is out of range for every head matrix (host.cpp keeps them below 2 GiB, and the
// per-tile scalar offset added to it stays below 2 GiB too): lanes that stage columns >= D use it and receive 0.
constexpr uint32_t kOobOffset = 0x80000000u;

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, H, Nq, Nkv;
    int D;                               // actual head dim (multiple of 8, <= the kernel's HD): columns >= D read as 0, are not stored
    int64_t qs[3], ks[3], vs[3], os[3];  // element strides: batch, head, row
    int64_t ls[2];                       // lse strides: batch, head
    float c;                             // |scale| * log2(e)
    int negate_q;                        // scale < 0: fold the sign into Q
    int nqblk;                           // ceil(Nq / rows per workgroup)
    uint32_t k_bytes, v_bytes;           // addressable bytes of one head's K / V matrix
    int bh0, nbh;                        // this launch covers the flattened (batch*H + head) range [bh0, bh0 + nbh)
    int rows_hint;                       // host only: rows per workgroup the launcher must use (0 = its own heuristic)
    int exact_scale;                     // host only: FA2_FLAG_EXACT_SCALE — this call never folds the scale into Q
    int persist;                         // hand-scheduled kernels: 1 = persistent workgroups (grid = work units, capped at the CU count), 0 = one item per workgroup
    // additive attention bias / boolean mask (BIAS kernels only; fa2_fwd_bias in include/fa2_gfx950.h): element (b,h,i,j) at
    // bias + b*bs[0] + h*bs[1] + i*bs[2] + j in elements of the bias type, strides may be 0 (broadcast)
    const void* bias;
    int64_t bs[3];
    int bias_kind;                       // 1: the I/O 16-bit dtype, 2: f32, 3: uint8 (non-zero = attend)
    int bias_vec;                        // host: 1 = one aligned load per group of four kv, 2 = coalesced tiles through LDS (4-wave kernels), 3 = tiles by LDS-DMA (8-wave kernels, BIAS = 2), 4 = row-broadcast bias, one load per wave and tile
    // KV-split tail (fa2_fwd_ws; non-causal, no bias, head dims <= 128, 256-row workgroups): B*H*nqblk equal items on the CUs take
    // ceil(items / CUs) rounds however empty the last one is.  The items of that last round — the last `split_items` of the item
    // order, after `full_items` whole ones — are each swept by `nsplit` workgroups ("parts") over disjoint KV ranges, which leave
    // normalised f32 partial O tiles and partial log2 LSEs in `ws`; fwd_combine_kernel merges them into o / lse.
    int full_items, split_items, nsplit; // nsplit <= 1: no split
    int blk0;                            // added to blockIdx.x (a launch that covers only the parts: blk0 = full_items)
    int item_cap;                        // hand-scheduled kernels: the launch covers the list entries [0, item_cap): whole items, then parts (0 = all whole items)
    float* ws;                           // [split_items * nsplit] partial O tiles of kSplitRows x HD floats, then as many LSE rows of kSplitRows
};

constexpr int kSplitRows = 256;          // rows of a split item (the 8-wave workgroup shape)
constexpr int kMaxSplit = 8;

// bytes of the KV-split workspace: partial O tiles + partial LSE rows
__host__ __device__ inline int64_t split_ws_bytes(int split_items, int nsplit, int HD) {
    return (int64_t)split_items * nsplit * kSplitRows * (HD + 1) * 4;
}

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

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// 16 bytes per lane, global (buffer descriptor, zero for out-of-range) -> LDS at wave-uniform `lds_dst` + lane*16
template <typename RSRC>
__device__ __forceinline__ void dma16_to_lds(RSRC rsrc, char* lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// LDS image geometry for head dim HD staged by NW waves
template <int HD, int NW>
struct Geo {
    static constexpr int ROWB = HD * 2;                       // bytes per tile row
    static constexpr int TILEB = kKvTile * ROWB;              // bytes per K (or V) tile
    static constexpr int G = ROWB / 16;                       // 16-B granules per row
    static constexpr int RPB = (ROWB >= 256) ? 1 : 256 / ROWB;  // rows per 256-B bank row
    static constexpr int KMASK = (G < 16 ? G : 16) - 1;
    static constexpr int C = ROWB / 64;                       // 64-B chunks per row
    static constexpr int VMASK = (C < 4 ? C : 4) - 1;
    static constexpr int NPASS = (kKvTile * G) / (NW * 64);   // 16-B staging loads per thread per tile
    static constexpr int KS_QK = HD / 16;                     // MFMA k-steps of Q.K^T
    static constexpr int DT = HD / 32;                        // 32-wide d blocks of O
    static_assert(NPASS >= 1, "head dim too small for the staging pattern");
    __device__ static __forceinline__ int k_off(int row, int gi) {
        return row * ROWB + ((gi ^ ((row / RPB) & KMASK)) << 4);
    }
    __device__ static __forceinline__ int v_off(int row, int colbyte) {
        return row * ROWB + ((((colbyte >> 6) ^ ((row / RPB) & VMASK))) << 6) + (colbyte & 63);
    }
};

#ifndef FA2_CAUSAL_HPG        // causal launch order: heads per group inside an XCD (0 = all heads of the XCD in one group).
#define FA2_CAUSAL_HPG 0      // Pairs of heads (2) cut config 4's HBM-side traffic from 2.17x to 1.40x of the algorithmic bytes but are
#endif                        // no faster there (1237 vs 1250 TF: the path is MFMA/power-bound, the re-reads hit the Infinity Cache) and
                              // cost 20 % at N = 2048 (fewer long blocks in flight): measured, left off

// Workgroup -> (batch*head, q block).  blockIdx % 8 is the XCD a block lands on (observed placement; speed only).
//   non-causal: all q blocks of a head run on one XCD, back to back, so the head's K/V stay in that XCD's L2 (equal work
//               per block: order is irrelevant for balance);
//   causal:     work grows with the q block index and blocks are handed to CUs in launch order, so blocks go longest
//               first ACROSS heads (with per-head ordering the long blocks of the later heads arrive last: makespan 100
//               instead of 68 tile-steps at B2 H16 N4096).  FA2_CAUSAL_HPG > 0 restricts the interleave to that many
//               heads at a time (interleaving all heads of an XCD cycles their K/V through one 4 MiB L2).
template <bool CAUSAL>
__device__ __forceinline__ void block_to_head_qblock(const FwdParams& p, int bid, int& bh, int& qblk) {
    const int nbh = p.nbh;               // (a launch may cover a sub-range of the heads: host.cpp, tail split)
    if ((nbh & 7) == 0) {
        const int slot = bid >> 3, hpx = nbh >> 3;   // hpx = heads per XCD
        if (CAUSAL) {
            const int hpg = (FA2_CAUSAL_HPG > 0 && hpx % FA2_CAUSAL_HPG == 0) ? FA2_CAUSAL_HPG : hpx;
            const int per_group = hpg * p.nqblk, g = slot / per_group, r = slot % per_group;
            bh = (bid & 7) + 8 * (g * hpg + r % hpg);
            qblk = p.nqblk - 1 - r / hpg;
        } else {
            bh = (bid & 7) + 8 * (slot / p.nqblk);
            qblk = slot % p.nqblk;
        }
    } else if (CAUSAL) {
        bh = bid % nbh;
        qblk = p.nqblk - 1 - bid / nbh;
    } else {
        bh = bid / p.nqblk;
        qblk = bid % p.nqblk;
    }
    bh += p.bh0;
}

// NW waves per workgroup, each owning QB consecutive 32-row Q blocks (NW * QB * 32 == 256):
//   <8, 1>: two waves per SIMD, 256 VGPRs each;  <4, 2>: one wave per SIMD with the 512-register
//   budget — every K / V^T fragment read from LDS then feeds two MFMAs instead of one.
//
// HD = head dim of Q and K (the QK^T contraction); HDV = the V / O columns one workgroup produces.
// HDV == HD except for D = 256, which is run as two column halves (blockIdx.y selects [0,128) or
// [128,256)): a 256-wide f32 O accumulator plus the Q fragments would not fit 256 VGPRs, so QK^T is
// recomputed per half (1.5x the MFMA work of an unsplit kernel; D = 256 only occurs at tiny N in practice).
//
// BIAS: scores = scale * Q K^T + bias[b, h, i, j] (or a boolean keep-mask) — the attention bias / mask argument the
// reference only reserves (`mask` is accepted and ignored, FlashAttn.py:49, :74; README.md:45 lists it as to do).  The bias of
// tile+1 is fetched into registers at the top of a step (the loads fly under the MFMAs), folded into the scores as
// s*c + bias*log2e after the causal / tail masks, and the softmax then works in log2 units (c = 1).
// A row whose every score is -inf (fully masked) produces O = 0 and lse = -inf.
// (BIAS at D = 64 with two workgroups per CU — 256 registers per wave instead of the 434 the compiler spreads one over — spills
//  112 VGPRs: not used)
// Register budget: 4-wave workgroups get the 512-register budget of one wave per SIMD only where they need it (bias registers, head dims
// above 128, two q blocks per wave); the plain 128-row kernels of head dims <= 128 fit 256 registers, and with the smaller budget the
// compiler keeps the accumulators out of the AGPRs (under the 512 budget it parked them there and wrapped every rescale in
// v_accvgpr_read / write: 11-13 register moves per MFMA in the ISA) and two workgroups can share a CU.
// BIAS = 2 (round 3): the bias tile of a step goes global -> LDS by LDS-DMA, swizzled like a K tile, into a wave-private image and is decoded
// group by group straight into the score registers — no 32 raw registers, so the 8-wave, 256-row shape (two waves per SIMD, 256 registers)
// carries a dense per-row bias too.  Geometry: the "tile" form's (pointer, strides and Nkv multiples of 16 bytes), head dims <= 128.
// (developer A/B, round 6: waves per SIMD the plain 128-row kernels are compiled for.  3 at head dim 64 spills 16-22 registers, 4 spills 300+, 3 at
//  head dim 128 spills 119-257: the streaming kernel keeps ~200 registers for its steady state — which is why short sweeps got a kernel of their own,
//  fa2_fwd_short.hip.h)
#ifndef FA2_OCC64
#define FA2_OCC64 2
#endif
#ifndef FA2_OCC128
#define FA2_OCC128 2
#endif
template <int HD, int NW, int QB, int BIAS>
constexpr int fwd_min_waves_per_simd() { return (NW == 4 && !BIAS && QB == 1 && HD <= 128) ? (HD <= 64 ? FA2_OCC64 : FA2_OCC128) : (NW + 3) / 4; }

//
// KSQ / DTN / RTD (round 3, "trimmed" instantiations, fwd_hip.cpp): a head dim D below the kernel's HD keeps the LDS images and
// the staging pattern of HD (columns >= D are never fetched: out-of-range granules) but runs only the MFMA k-steps and O column
// blocks that hold real columns — KSQ = ceil(D / 16) k-steps of Q.K^T (rounded up to the instantiated value), DTN = 32-wide O
// blocks per column half.  RTD (HD = 256, two column halves): the SECOND half holds D - 128 columns only, its block count is a
// workgroup-uniform run-time bound.  Defaults = the full kernel.  KV-split parts keep the workspace layout of HD (the merge kernel never
// reads columns >= D).  Reference counterpart: the host-side zero padding of D to a
// multiple of 32 (kernel_fp16.cu:763-779) — it multiplies the zeros.
template <int HD, int HDV, bool BF16, bool CAUSAL, int NW, int QB, int BIAS = 0, int KSQ = HD / 16, int DTN = HDV / 32, bool RTD = false>
__global__ __launch_bounds__(NW * 64, (fwd_min_waves_per_simd<HD, NW, QB, BIAS>())) void fwd_kernel(const FwdParams p) {
    constexpr int kRowsPerBlock = NW * QB * 32;   // Q rows per workgroup (p.nqblk = ceil(Nq / kRowsPerBlock))
    using G_ = Geo<HD, NW>;    // K tile image
    using GV_ = Geo<HDV, NW>;  // V tile image
    constexpr int ROWB = G_::ROWB, TILEB = G_::TILEB, NPASS = G_::NPASS;
    constexpr int VROWB = GV_::ROWB, VTILEB = GV_::TILEB, VNPASS = GV_::NPASS;
    constexpr int KS_QK = KSQ, DT = DTN;
    static_assert(KSQ <= G_::KS_QK && DTN <= GV_::DT, "trimmed loop bounds");
    constexpr int VBASE = 2 * TILEB;   // LDS: K buf0 | K buf1 | V buf0 | V buf1
    const int vcol0 = blockIdx.y * HDV;   // first V / O column of this workgroup
    const int ndt = RTD ? (p.D - vcol0 + 31) / 32 : DT;   // O column blocks of this half that hold real columns (uniform)
    constexpr int kThreads = NW * 64, kRowsPerWave = 32 * QB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // KV-split tail: blocks [full_items, full_items + split_items * nsplit) are parts; part-major order, so that with split_items a
    // multiple of 8 a part lands on the XCD (block index % 8) its item's head is mapped to
    int bid = blockIdx.x + p.blk0;
    int part = -1, sidx = 0;
    if constexpr (!CAUSAL && !BIAS && HD == HDV && NW * QB * 32 == kSplitRows) {
        if (p.nsplit > 1 && bid >= p.full_items) {
            const int j = bid - p.full_items;
            part = j / p.split_items;
            sidx = j % p.split_items;
            bid = p.full_items + sidx;
        }
    }
    int bh, qblk;
    block_to_head_qblock<CAUSAL>(p, bid, bh, qblk);
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * kRowsPerBlock;
    // this workgroup's KV range [kv_first, kv_first + nkv): everything, or a part's whole tiles
    int kv_first = 0, nkv = p.Nkv;
    if (part >= 0) {
        const int nt = (p.Nkv + kKvTile - 1) / kKvTile;
        const int t0 = part * nt / p.nsplit, t1 = (part + 1) * nt / p.nsplit;
        kv_first = t0 * kKvTile;
        nkv = (t1 * kKvTile < p.Nkv ? t1 * kKvTile : p.Nkv) - kv_first;
    }
    const int qw0 = q0 + wave * kRowsPerWave;   // first Q row of this wave
    int qrow[QB];                               // this lane's Q row in each of its blocks (may be >= Nq)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) qrow[qb] = qw0 + 32 * qb + l31;

    // ---- Q fragments (B operand): lane reads 8 consecutive d of its row per k-step
    u32x4 qf[QB][KS_QK];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qr = qrow[qb] < p.Nq ? qrow[qb] : p.Nq - 1;
        const uint16_t* qp = (const uint16_t*)p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qr * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks)     // head dims below HD: the missing columns are zeros (reference: host-side pad, kernel_fp16.cu:763-779)
            qf[qb][ks] = (16 * ks + 8 * hi < p.D) ? *(const u32x4*)(qp + 16 * ks + 8 * hi) : (u32x4){0u, 0u, 0u, 0u};
        if (p.negate_q) {
            const uint32_t sgn = 0x80008000u;
#pragma unroll
            for (int ks = 0; ks < KS_QK; ++ks) qf[qb][ks] ^= (u32x4){sgn, sgn, sgn, sgn};
        }
    }

    // ---- K/V staging: buffer descriptors of this head's matrices (out-of-range rows read 0)
    const uint32_t k_rowb = (uint32_t)p.ks[2] * 2u, v_rowb = (uint32_t)p.vs[2] * 2u;
    const uint16_t* kbase = (const uint16_t*)p.k + b * p.ks[0] + h * p.ks[1] + (int64_t)kv_first * p.ks[2];
    const uint16_t* vbase = (const uint16_t*)p.v + b * p.vs[0] + h * p.vs[1] + (int64_t)kv_first * p.vs[2];
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k_bytes - (uint32_t)kv_first * k_rowb, 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, p.v_bytes - (uint32_t)kv_first * v_rowb, 0x00020000);
    uint32_t kg_off[NPASS], vg_off[VNPASS];  // per-lane byte offsets into the head matrix, tile 0
    int kw_off[NPASS], vw_off[VNPASS];       // per-lane LDS byte offsets inside a tile image
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / G_::G, gi = idx % G_::G;
        kg_off[i] = gi * 8 < p.D ? row * k_rowb + gi * 16 : kOobOffset;
        kw_off[i] = G_::k_off(row, gi);
    }
#pragma unroll
    for (int i = 0; i < VNPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / GV_::G, gi = idx % GV_::G;
        vg_off[i] = gi * 8 + vcol0 < p.D ? row * v_rowb + gi * 16 + vcol0 * 2 : kOobOffset;
        vw_off[i] = GV_::v_off(row, gi * 16);
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
            vr_off[dt] = GV_::v_off(4 * hi + (pp >> 2), (32 * dt + 16 * g1 + 4 * (pp & 3)) * 2);
    }

    // ---- KV sweep bounds
    int ntiles = (nkv + kKvTile - 1) / kKvTile;
    if (CAUSAL) {
        const int qmax = (q0 + kRowsPerBlock < p.Nq ? q0 + kRowsPerBlock : p.Nq) - 1;
        const int nt_c = qmax / kKvTile + 1;
        ntiles = nt_c < ntiles ? nt_c : ntiles;
    }
    // causal: tiles this wave actually computes (the rest only stage + sync)
    int ntiles_w = ntiles;
    if (CAUSAL) {
        const int nt_w = (qw0 + kRowsPerWave - 1) / kKvTile + 1;
        ntiles_w = nt_w < ntiles ? nt_w : ntiles;
    }

    f32x16 acc[QB][DT];
    float m_run[QB], l_run[QB];  // running reference max (raw score units) / row sum (this lane's kv half)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][dt][r] = 0.f;
    }
    const float cs = p.c;                // factor of the raw Q.K^T product
    const float c = BIAS ? 1.0f : p.c;   // factor still to be applied to a finished score (BIAS: already in log2 units)

    // ---- attention bias (BIAS kernels): the 32 values of this lane's row in one KV tile, as loaded (raw words), in the register
    // order of the two score accumulators: element i = 16*half + 4*g + e <-> kv = kv0 + 32*half + 8*g + 4*hi + e, i.e. eight groups
    // G = 4*half + g of four consecutive kv.  Two load forms (wave-uniform p.bias_vec, chosen by the host):
    //   scalar  one guarded load per element, raw[i] = the element (any alignment, any Nkv — e.g. 77-token cross-attention rows)
    //   vector  one load per group — 8 / 16 / 4 bytes for the 16-bit / f32 / byte kinds — when the base pointer and every stride
    //           are multiples of four elements and Nkv % 4 == 0: raw[2G..2G+1] / raw[4G..4G+3] / raw[G]
    //   tile    (p.bias_vec == 2; pointer, strides and Nkv multiples of 16 bytes, head dims up to 256) the wave fetches its 32 rows x
    //           64 kv tile with COALESCED 16-byte loads (4 / 8 / 2 per lane; an instruction covers whole rows — with lane = Q row
    //           the other forms touch 32 cache lines per instruction for a few useful bytes each), parks it in a wave-private LDS
    //           image (rows of 272 bytes) when the scores are ready and reads its own row's groups back from there
    // The words are decoded where they are used (add_bias), so that the loads issued at the top of a step fly under its MFMAs.
    // (the words travel as ONE 32-element vector value per wave tile: as an array, written on several control paths, they were
    //  kept in scratch memory — 112 bytes per lane — instead of registers)
    constexpr int kBiasRowB = 272;                           // up to 256 bytes of one row's 64 bias values (f32) + 16 bytes of padding
    constexpr int kBiasLdsBase = 2 * TILEB + 2 * VTILEB;     // "tile" form: NW wave-private images of 32 rows above the K / V buffers
    static_assert(!BIAS || QB == 1, "the bias kernels run one 32-row Q block per wave");
    // BIAS == 2: per kind, a tile row is GPR granules of 16 bytes (64 elements), an LDS-DMA instruction covers RPI = 64 / GPR rows, the image keeps
    // granule g of row r at slot g ^ (r & MASK) (the swizzle is applied to the lane's SOURCE address); rows >= Nq and granules past the last row's
    // end are out of range for the descriptor and read zeros (such scores are masked or never stored)
    constexpr int kBiasDmaBase = 2 * TILEB + 2 * VTILEB, kBiasDmaImg = 8192;       // NW wave-private images above the K / V buffers
    auto load_bias_dma = [&](int tile) __attribute__((always_inline)) {
        if constexpr (BIAS == 2) {
            auto one = [&](auto es_t) __attribute__((always_inline)) {
                constexpr int ES = decltype(es_t)::value, GPR = 64 * ES / 16, RPI = 64 / GPR, NI = 32 / RPI, MASK = (GPR < RPI ? GPR : RPI) - 1;
                const char* base = (const char*)p.bias + (b * p.bs[0] + h * p.bs[1]) * ES;
                const uint32_t rowb = (uint32_t)p.bs[2] * ES;
                const auto brs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(((int64_t)(p.Nq - 1) * p.bs[2] + p.Nkv) * ES), 0x00020000);
                const int lr = lane / GPR, g = (lane % GPR) ^ (lr & MASK);
                const int kvg = tile * kKvTile + g * (16 / ES);                         // first kv of this lane's granule
                const uint32_t voff = kvg < p.Nkv ? (uint32_t)(qw0 + lr) * rowb + (uint32_t)kvg * ES : kOobOffset;
                char* img = smem + kBiasDmaBase + wave * kBiasDmaImg;
#pragma unroll
                for (int i = 0; i < NI; ++i) dma16_to_lds(brs, img + i * 1024, voff, (uint32_t)(i * RPI) * rowb);
            };
            if (p.bias_kind == 1) one(std::integral_constant<int, 2>{});
            else if (p.bias_kind == 2) one(std::integral_constant<int, 4>{});
            else one(std::integral_constant<int, 1>{});
        }
    };
    auto load_bias = [&](int tile, u32x32& raw) __attribute__((always_inline)) {
        if constexpr (BIAS == 2) {
            load_bias_dma(tile);
        } else if constexpr (BIAS) {
            const int kv0 = tile * kKvTile + 4 * hi;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int qr = qrow[qb] < p.Nq ? qrow[qb] : p.Nq - 1;
                const int64_t row = b * p.bs[0] + h * p.bs[1] + (int64_t)qr * p.bs[2];
                if (p.bias_vec == 4) {
                    // a bias broadcast over the Q rows (row stride 0: a [B, 1, 1, Nkv] key-padding mask): the tile's 64 values are the same for
                    // every row — lane l fetches the one of kv0 + l (one coalesced load per wave and tile instead of 8 .. 32 per lane); add_bias
                    // spreads them through the wave's LDS image.  raw[0] = the element, 0 past Nkv (those columns are masked by the tail anyway).
                    const int kvl = tile * kKvTile + lane;
                    const int64_t at = b * p.bs[0] + h * p.bs[1] + kvl;
                    if (p.bias_kind == 1) raw[0] = kvl < p.Nkv ? (uint32_t)((const uint16_t*)p.bias)[at] : 0u;
                    else if (p.bias_kind == 2) raw[0] = kvl < p.Nkv ? ((const uint32_t*)p.bias)[at] : 0u;
                    else raw[0] = kvl < p.Nkv ? (uint32_t)((const uint8_t*)p.bias)[at] : 0u;
                } else if (p.bias_vec == 2) {
                    // a tile row is 64 elements = LPR lanes of 16 bytes; instruction i covers rows [i*RPI, +RPI) of the wave's 32
                    // (RPI = 64 / LPR); its words land in raw[4i .. 4i+3]
                    auto tile_rows = [&](auto es_t, auto lpr_log_t) __attribute__((always_inline)) {
                        constexpr int ES = decltype(es_t)::value, LPRL = decltype(lpr_log_t)::value, RPI = 64 >> LPRL;
                        const char* bh_base = (const char*)p.bias + (b * p.bs[0] + h * p.bs[1]) * ES;
                        const int kvg = tile * kKvTile + (16 / ES) * (lane & ((1 << LPRL) - 1));      // first kv of this lane's granule
#pragma unroll
                        for (int i = 0; i < 32 / RPI; ++i) {
                            const int rr = qw0 + 32 * qb + i * RPI + (lane >> LPRL);
                            const int rc = rr < p.Nq ? rr : p.Nq - 1;
                            const u32x4 w = kvg < p.Nkv ? *(const u32x4*)(bh_base + ((int64_t)rc * p.bs[2] + kvg) * ES) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
                            for (int e = 0; e < 4; ++e) raw[4 * i + e] = w[e];
                        }
                    };
                    if (p.bias_kind == 1) tile_rows(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
                    else if (p.bias_kind == 2) tile_rows(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
                    else tile_rows(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                } else if (p.bias_vec) {
                    if (p.bias_kind == 1) {
                        const uint16_t* bp = (const uint16_t*)p.bias + row;
#pragma unroll
                        for (int G = 0; G < 8; ++G) {
                            const int kvi = kv0 + 32 * (G >> 2) + 8 * (G & 3);
                            const u32x2 w = kvi < p.Nkv ? *(const u32x2*)(bp + kvi) : (u32x2){0u, 0u};
                            raw[2 * G] = w[0];
                            raw[2 * G + 1] = w[1];
                        }
                    } else if (p.bias_kind == 2) {
                        const uint32_t* bp = (const uint32_t*)p.bias + row;
#pragma unroll
                        for (int G = 0; G < 8; ++G) {
                            const int kvi = kv0 + 32 * (G >> 2) + 8 * (G & 3);
                            const u32x4 w = kvi < p.Nkv ? *(const u32x4*)(bp + kvi) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
                            for (int e = 0; e < 4; ++e) raw[4 * G + e] = w[e];
                        }
                    } else {
                        const uint8_t* bp = (const uint8_t*)p.bias + row;
#pragma unroll
                        for (int G = 0; G < 8; ++G) {
                            const int kvi = kv0 + 32 * (G >> 2) + 8 * (G & 3);
                            raw[G] = kvi < p.Nkv ? *(const uint32_t*)(bp + kvi) : 0u;
                        }
                    }
                } else if (p.bias_kind == 1) {
                    const uint16_t* bp = (const uint16_t*)p.bias + row;
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int kvi = kv0 + 32 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2);
                        raw[e] = kvi < p.Nkv ? (uint32_t)bp[kvi] : 0u;
                    }
                } else if (p.bias_kind == 2) {
                    const uint32_t* bp = (const uint32_t*)p.bias + row;
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int kvi = kv0 + 32 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2);
                        raw[e] = kvi < p.Nkv ? bp[kvi] : 0u;
                    }
                } else {
                    const uint8_t* bp = (const uint8_t*)p.bias + row;
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int kvi = kv0 + 32 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2);
                        raw[e] = kvi < p.Nkv ? (uint32_t)bp[kvi] : 0u;
                    }
                }
            }
        }
    };
    // s <- s * cs + bias * log2(e)   (boolean mask: s * cs where kept, -inf where not).  One fully separate code path per bias
    // kind (compile-time KIND inside, wave-uniform dispatch outside): shared temporaries across the kinds ended up in scratch.
    auto add_bias = [&](f32x16& s0, f32x16& s1, const u32x32& raw) __attribute__((always_inline)) {
        if constexpr (BIAS == 2) {
            __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's LDS-DMA pieces of the tile have landed (the image is wave-private)
            auto one = [&](auto es_t) __attribute__((always_inline)) {
                constexpr int ES = decltype(es_t)::value, GPR = 64 * ES / 16, RPI = 64 / GPR, MASK = (GPR < RPI ? GPR : RPI) - 1;
                constexpr float kLog2e = 1.4426950408889634f;
                const char* row = smem + kBiasDmaBase + wave * kBiasDmaImg + l31 * (GPR * 16);
                const int sw = l31 & MASK;
#pragma unroll
                for (int G = 0; G < 8; ++G) {          // group G: kv = 32 (G >> 2) + 8 (G & 3) + 4 hi + e, e = 0..3 -> registers 4 (G & 3) + e of s0 / s1
                    const int byte = (32 * (G >> 2) + 8 * (G & 3) + 4 * hi) * ES;
                    const char* src = row + (((byte >> 4) ^ sw) << 4) + (byte & 15);
                    float bv[4];
                    if constexpr (ES == 2) {
                        const u32x2 w = *(const u32x2*)src;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t h16 = (e & 1) ? w[e >> 1] >> 16 : w[e >> 1] & 0xffffu;
                            bv[e] = (BF16 ? __uint_as_float(h16 << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)h16)) * kLog2e;
                        }
                    } else if constexpr (ES == 4) {
                        const u32x4 w = *(const u32x4*)src;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = __uint_as_float(w[e]) * kLog2e;
                    } else {
                        const uint32_t w = *(const uint32_t*)src;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = (w & (0xffu << (8 * e))) ? 0.f : -INFINITY;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {              // (vector elements cannot be bound to references: both halves spelled out)
                        const int rr = 4 * (G & 3) + e;
                        const float sc = (G >> 2) ? s1[rr] : s0[rr];
                        const float nv = ES == 1 ? (bv[e] == 0.f ? sc * cs : -INFINITY) : __builtin_fmaf(sc, cs, bv[e]);
                        if (G >> 2) s1[rr] = nv;
                        else s0[rr] = nv;
                    }
                }
            };
            if (p.bias_kind == 1) one(std::integral_constant<int, 2>{});
            else if (p.bias_kind == 2) one(std::integral_constant<int, 4>{});
            else one(std::integral_constant<int, 1>{});
        } else if constexpr (BIAS) {
            auto one_kind = [&](auto kind_t) __attribute__((always_inline)) {
                constexpr int KIND = decltype(kind_t)::value;
                constexpr float kLog2e = 1.4426950408889634f;
                constexpr int ES = KIND == 1 ? 2 : KIND == 2 ? 4 : 1;          // bytes per element
                constexpr int LPRL = KIND == 1 ? 3 : KIND == 2 ? 4 : 2;        // log2(lanes of 16 bytes per tile row)
                constexpr int RPI = 64 >> LPRL;                                // rows per load instruction
                constexpr int WPG = ES;                                        // words per group of four elements
                auto io16 = [](uint32_t bits16) __attribute__((always_inline)) -> float {      // low 16 bits -> f32
                    if constexpr (BF16) return __uint_as_float(bits16 << 16);
                    else return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16);
                };
                // element r (0..15) of half hf from the grouped word layout (group G = 4*hf + (r >> 2), element e = r & 3)
                auto decode = [&](const u32x32& wd, int hf, int r) __attribute__((always_inline)) -> float {
                    const int G = 4 * hf + (r >> 2), e = r & 3;
                    if constexpr (KIND == 2) return __uint_as_float(wd[4 * G + e]) * kLog2e;
                    else if constexpr (KIND == 1) {
                        const uint32_t w = wd[2 * G + (e >> 1)];
                        return io16((e & 1) ? w >> 16 : w & 0xffffu) * kLog2e;
                    } else return (wd[G] & (0xffu << (8 * e))) ? 0.f : -INFINITY;
                };
                auto fold = [&](float s, float bv) __attribute__((always_inline)) -> float {
                    if constexpr (KIND == 3) return bv == 0.f ? s * cs : -INFINITY;
                    else return __builtin_fmaf(s, cs, bv);
                };
                if (p.bias_vec == 4) {
                    // row-broadcast form: this lane's element -> log2-domain term (bool: 0 / -inf), 64 floats through the wave's LDS image, the
                    // lane's eight groups of four read back (LDS operations of one wave execute in order: no barrier)
                    char* bimg = smem + kBiasLdsBase + wave * (32 * kBiasRowB);
                    float t;
                    if constexpr (KIND == 3) t = raw[0] ? 0.f : -INFINITY;
                    else if constexpr (KIND == 2) t = __uint_as_float(raw[0]) * kLog2e;
                    else t = io16(raw[0]) * kLog2e;
                    *(float*)(bimg + lane * 4) = t;
                    typedef float f32x4b __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int G = 0; G < 8; ++G) {
                        const f32x4b w = *(const f32x4b*)(bimg + (32 * (G >> 2) + 8 * (G & 3) + 4 * hi) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rr = 4 * (G & 3) + e;
                            if (G >> 2) s1[rr] = __builtin_fmaf(s1[rr], cs, w[e]);
                            else s0[rr] = __builtin_fmaf(s0[rr], cs, w[e]);
                        }
                    }
                } else if (p.bias_vec == 2) {
                    // wave-private image above the K / V buffers: park the coalesced tile, read this lane's row back (the LDS
                    // operations of one wave execute in order, so neither a barrier nor a second buffer is needed)
                    char* bimg = smem + kBiasLdsBase + wave * (32 * kBiasRowB);
#pragma unroll
                    for (int i = 0; i < 32 / RPI; ++i)
                        *(u32x4*)(bimg + (RPI * i + (lane >> LPRL)) * kBiasRowB + 16 * (lane & ((1 << LPRL) - 1))) =
                            (u32x4){raw[4 * i], raw[4 * i + 1], raw[4 * i + 2], raw[4 * i + 3]};
                    u32x32 wd;
#pragma unroll
                    for (int G = 0; G < 8; ++G) {
                        const char* src = bimg + l31 * kBiasRowB + (32 * (G >> 2) + 8 * (G & 3) + 4 * hi) * ES;
                        if constexpr (KIND == 1) {
                            const u32x2 w = *(const u32x2*)src;
                            wd[2 * G] = w[0];
                            wd[2 * G + 1] = w[1];
                        } else if constexpr (KIND == 2) {
                            const u32x4 w = *(const u32x4*)src;
#pragma unroll
                            for (int e = 0; e < 4; ++e) wd[4 * G + e] = w[e];
                        } else {
                            wd[G] = *(const uint32_t*)src;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s0[r] = fold(s0[r], decode(wd, 0, r));
                        s1[r] = fold(s1[r], decode(wd, 1, r));
                    }
                } else if (p.bias_vec) {
                    u32x32 wd;
                    wd = raw;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s0[r] = fold(s0[r], decode(wd, 0, r));
                        s1[r] = fold(s1[r], decode(wd, 1, r));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {        // scalar form: one word per element
                        if constexpr (KIND == 3) {
                            s0[r] = raw[r] ? s0[r] * cs : -INFINITY;
                            s1[r] = raw[16 + r] ? s1[r] * cs : -INFINITY;
                        } else if constexpr (KIND == 2) {
                            s0[r] = __builtin_fmaf(s0[r], cs, __uint_as_float(raw[r]) * kLog2e);
                            s1[r] = __builtin_fmaf(s1[r], cs, __uint_as_float(raw[16 + r]) * kLog2e);
                        } else {
                            s0[r] = __builtin_fmaf(s0[r], cs, io16(raw[r]) * kLog2e);
                            s1[r] = __builtin_fmaf(s1[r], cs, io16(raw[16 + r]) * kLog2e);
                        }
                    }
                }
            };
            if (p.bias_kind == 1) one_kind(std::integral_constant<int, 1>{});
            else if (p.bias_kind == 2) one_kind(std::integral_constant<int, 2>{});
            else one_kind(std::integral_constant<int, 3>{});
        }
    };

    // Staging.  DMA form (head dims >= FA2_LDS_DMA_MIN_HD): buffer_load ... lds writes the tile image
    // directly — no staging VGPRs, no ds_write; wave-instruction i of this wave fills the 1 KiB of the
    // image starting at linear granule (wave*64 + kThreads*i), and lane l supplies the global address of
    // the granule that belongs at image slot (wave*64 + kThreads*i + l), i.e. the inverse of the k_off /
    // v_off swizzle (the image itself stays lane-linear).  The destination buffer must be free when the
    // load is ISSUED.  Register form: global -> VGPRs at issue, ds_write_b128 at the end of the step.
    constexpr bool kDma = FA2_LDS_DMA && HD >= FA2_LDS_DMA_MIN_HD;
    uint32_t kd_off[NPASS], vd_off[VNPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / G_::G, slot = idx % G_::G;
        const int gk = slot ^ ((row / G_::RPB) & G_::KMASK);   // source granule of this image slot
        kd_off[i] = gk * 8 < p.D ? row * k_rowb + gk * 16 : kOobOffset;
    }
#pragma unroll
    for (int i = 0; i < VNPASS; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / GV_::G, slot = idx % GV_::G;
        const int gv = ((((slot >> 2) ^ ((row / GV_::RPB) & GV_::VMASK))) << 2) | (slot & 3);
        vd_off[i] = gv * 8 + vcol0 < p.D ? row * v_rowb + gv * 16 + vcol0 * 2 : kOobOffset;
    }
    u32x4 kreg[NPASS], vreg[VNPASS];
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
        for (int i = 0; i < VNPASS; ++i) {
            if constexpr (kDma) dma16_to_lds(vrs, smem + VBASE + buf * VTILEB + (wave * 64 + kThreads * i) * 16, FA2_TILE_OFF(vd_off[i], soff));
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
            for (int i = 0; i < VNPASS; ++i) *(u32x4*)(smem + VBASE + buf * VTILEB + vw_off[i]) = vreg[i];
        }
    };

    // S^T = K Q^T for one KV tile: per Q block two 32(kv) x 32(q) accumulators; each K fragment
    // read from LDS feeds QB MFMAs
    auto qk = [&](int buf, f32x16 (&s)[QB][2]) __attribute__((always_inline)) {
        const char* kt = smem + buf * TILEB;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[qb][0][r] = s[qb][1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS_QK; ++ks) {
            const u32x4 a0 = *(const u32x4*)(kt + kr_off[ks]);
            const u32x4 a1 = *(const u32x4*)(kt + kr_off[ks] + 32 * ROWB);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                s[qb][0] = mfma16<BF16>(a0, qf[qb][ks], s[qb][0]);
                s[qb][1] = mfma16<BF16>(a1, qf[qb][ks], s[qb][1]);
            }
        }
    };

    // Scores of `tile` just left the MFMA: apply the masks (MASKED: causal diagonal / ragged last tile),
    // reduce the row max and, when some row's max grew by more than the threshold, move the running
    // reference max and rescale O and l (wave-uniform, rare branch).  With THR = 0 this is exact: rows
    // that did not grow have alpha == 1.  Runs BEFORE the tile's P is formed and AFTER the previous
    // tile's P.V has been accumulated, so everything at the old reference is scaled exactly once.
    // (reference: kernel_fp16.cu:396-451)
    auto finish_scores = [&](int tile, auto masked, f32x16 (&s)[QB][2], const u32x32& braw) __attribute__((always_inline)) {
        float mx[QB];
        bool grow = FA2_DEFER_THR < 0.f;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x16& s0 = s[qb][0];
            f32x16& s1 = s[qb][1];
            if constexpr (decltype(masked)::value) {
                const int kv0 = tile * kKvTile;
                const bool need_causal = CAUSAL && (kv0 + kKvTile - 1 > qw0 + 32 * qb);
                const bool need_tail = kv0 + kKvTile > nkv;
                if (need_causal || need_tail) {
                    const int lim_c = CAUSAL ? qrow[qb] : 0x7fffffff;  // kv index must be <= lim_c
                    int lim = lim_c < nkv - 1 ? lim_c : nkv - 1;
                    int kvb = kv0 + 4 * hi;
                    // opaque (D = 64, 256; measured neutral-to-negative at D = 128): the tail comparison of the non-causal
                    // build does not depend on the tile, so LICM hoists all 32 lane masks (64 SGPRs) or the 32
                    // per-register kv indices (32 VGPRs) to kernel entry — D = 256: 24 -> 3 spilled VGPRs, +2..+9 %
                    if constexpr (HD != 128) asm volatile("" : "+v"(lim), "+v"(kvb));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kvi = HD != 128 ? kvb + (r & 3) + 8 * (r >> 2) : kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (kvi > lim) s0[r] = -INFINITY;
                        if (kvi + 32 > lim) s1[r] = -INFINITY;
                    }
                }
            }
            add_bias(s0, s1, braw);
            float m = max3(s0[0], s1[0], s0[1]);
            m = max3(m, s1[1], s0[2]);
#pragma unroll
            for (int r = 2; r < 15; ++r) m = max3(m, s1[r], s0[r + 1]);
            m = __builtin_fmaxf(m, s1[15]);
            mx[qb] = half_swap_max(m);
            if (!(FA2_DEFER_THR < 0.f))
                grow = grow || (__builtin_amdgcn_ballot_w64((mx[qb] - m_run[qb]) * c > FA2_DEFER_THR) != 0);
        }
        if (grow) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float m_new = __builtin_fmaxf(m_run[qb], mx[qb]);
                float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
                if constexpr (BIAS) alpha = m_new == -INFINITY ? 1.0f : alpha;   // row fully masked so far: nothing accumulated
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[qb][dt][r] *= alpha;
            }
        }
    };

    // P = 2^(S*c - m*c) against the current reference max, row sum, P -> 16-bit B fragments
    // (reference: kernel_fp16.cu:455-479); k-step ks uses registers [8(ks&1), 8(ks&1)+8) of tile ks>>1
    auto exp_scores = [&](f32x16 (&s)[QB][2], u32x4 (&pf)[QB][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x16& s0 = s[qb][0];
            f32x16& s1 = s[qb][1];
            const float mc = (BIAS && m_run[qb] == -INFINITY) ? 0.f : m_run[qb] * c;   // fully masked so far: P = 2^(-inf - 0) = 0
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -mc));
                s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -mc));
                rs0 += s0[r];
                rs1 += s1[r];
            }
            l_run[qb] += rs0 + rs1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf[qb][0][i] = pack2<BF16>(s0[2 * i], s0[2 * i + 1]);
                pf[qb][1][i] = pack2<BF16>(s0[8 + 2 * i], s0[8 + 2 * i + 1]);
                pf[qb][2][i] = pack2<BF16>(s1[2 * i], s1[2 * i + 1]);
                pf[qb][3][i] = pack2<BF16>(s1[8 + 2 * i], s1[8 + 2 * i + 1]);
            }
        }
    };

    // O^T += V^T P^T; each V^T fragment read from LDS feeds QB MFMAs
    auto pv = [&](int buf, const u32x4 (&pf)[QB][4]) __attribute__((always_inline)) {
        const char* vt = smem + VBASE + buf * VTILEB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (RTD && dt >= ndt) continue;
                const char* va = vt + vr_off[dt] + 16 * ks * VROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * VROWB));
                const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi4);
                const u32x4 a = (u32x4){lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) acc[qb][dt] = mfma16<BF16>(a, pf[qb][ks], acc[qb][dt]);
            }
        }
    };

    // One pipeline step.  PAR = tile & 1 selects the LDS buffers statically:
    //   reads  K(tile+1) from K buf PAR^1, V(tile) from V buf PAR
    //   writes K(tile+2) to   K buf PAR,   V(tile+1) to V buf PAR^1
    // sc = finished scores of `tile` (QK^T one step earlier), sn receives the scores of tile+1.
    // MODE 1 = steady state: every load/compute condition is known true and tile+1 needs no mask, so the
    // QK^T MFMAs of tile+1, the exp/pack VALU work of `tile` and the P.V MFMAs of `tile` form ONE basic
    // block the scheduler can interleave; the only branch is the rare rescale at the end.  MODE 0 = generic
    // (masked tiles, pipeline tail, waves above the causal diagonal that only stage and sync).  A masked
    // steady-state variant for the diagonal tiles measured +0 % (and cost 23 VGPRs), so it is not kept.
    auto step = [&](int tile, auto par, auto mode, f32x16 (&sc)[QB][2], f32x16 (&sn)[QB][2]) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr int MODE = decltype(mode)::value;
        constexpr bool FAST = MODE != 0;
        const bool more1 = FAST || tile + 1 < ntiles, more2 = FAST || tile + 2 < ntiles;
        const bool next_w = FAST || tile + 1 < ntiles_w, cur_w = FAST || tile < ntiles_w;
#if FA2_IGLP >= 0
        if constexpr (FAST) __builtin_amdgcn_iglp_opt(FA2_IGLP);   // scheduler hint for the steady-state block
#endif
        if (more2) load_k(tile + 2, PAR);  // global loads fly under the MFMA work below
        if (more1) load_v(tile + 1, PAR ^ 1);
        u32x32 braw;
        if (BIAS && next_w) load_bias(tile + 1, braw);
        if (next_w) qk(PAR ^ 1, sn);
        if (cur_w) {
            u32x4 pf[QB][4];
            exp_scores(sc, pf);
            pv(PAR, pf);
        }
        if (more2) write_k(PAR);
        if (more1) write_v(PAR ^ 1);
        __syncthreads();
        if (next_w) finish_scores(tile + 1, std::integral_constant<bool, MODE != 1>{}, sn, braw);
    };

    // ---- prologue: K0, V0 -> buffers 0, K1 -> K buffer 1; scores of tile 0
    load_k(0, 0);
    load_v(0, 0);
    write_k(0);
    write_v(0);
    if (ntiles > 1) { load_k(1, 1); write_k(1); }
    __syncthreads();
    f32x16 sa[QB][2], sb[QB][2];
    u32x32 braw0;
    load_bias(0, braw0);
    qk(0, sa);
    __syncthreads();   // step(0) stages K2 into K buffer 0: every wave's tile-0 fragment reads must be behind us
    finish_scores(0, std::true_type{}, sa, braw0);

    // steady-state tiles [0, n_fast): tile+2 < ntiles, tile+1 < ntiles_w, tile+1 unmasked
    int n_fast = ntiles - 2 < ntiles_w - 1 ? ntiles - 2 : ntiles_w - 1;
    {
        const int unmasked_kv = nkv / kKvTile;                             // tiles fully inside this workgroup's KV range
        const int unmasked_c = CAUSAL ? (qw0 + 1) / kKvTile : 0x7fffffff;  // tiles fully below the diagonal
        const int unmasked = unmasked_kv < unmasked_c ? unmasked_kv : unmasked_c;
        n_fast = n_fast < unmasked - 1 ? n_fast : unmasked - 1;            // tile+1 <= unmasked-1
        n_fast = n_fast < 0 ? 0 : n_fast & ~1;
    }
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::integral_constant<int, 0> GENERIC{};
    constexpr std::integral_constant<int, 1> STEADY{};
    int tile = 0;
    for (; tile < n_fast; tile += 2) {
        step(tile, P0, STEADY, sa, sb);
        step(tile + 1, P1, STEADY, sb, sa);
    }
    for (; tile + 1 < ntiles; tile += 2) {
        step(tile, P0, GENERIC, sa, sb);
        step(tile + 1, P1, GENERIC, sb, sa);
    }
    if (tile < ntiles) step(tile, P0, GENERIC, sa, sb);


    // ---- epilogue (reference: kernel_fp16.cu:510-543): O = O / l, lse = m + log2(l) (log2 domain)
#if FA2_EPI_LDS
    // The normalised 16-bit O tile of a wave (32 rows x HDV columns) is written to a wave-private LDS image and read back
    // row-major, so that every global store instruction writes whole contiguous rows (64 lanes x 16 B = 4 rows of
    // 256 B) instead of 32 B of each of 32 rows.  The K/V buffers are free by now; rows are padded by 16 B so both the
    // column-wise writes and the row-wise reads are bank-conflict free.
    if constexpr (!CAUSAL && !BIAS && HD == HDV && NW * QB * 32 == kSplitRows) {
        if (part >= 0) {
            // a part: normalised f32 partial tile + partial LSE -> workspace.  Layout of a tile: float (((dt*4 + g) * 256 + row) * 8 + 4*hi + e)
            // for d = 32dt + 8g + 4hi + e — one store instruction of the wave writes 1 KiB of consecutive bytes.
            const int slot = sidx * p.nsplit + part;
            float* wo = p.ws + (int64_t)slot * kSplitRows * HD;
            float* wl = p.ws + (int64_t)p.split_items * p.nsplit * kSplitRows * HD + (int64_t)slot * kSplitRows;
            const int row = wave * 32 + l31;
            const float l_tot = half_swap_sum(l_run[0]);
            const float inv_l = 1.0f / l_tot;
            if (q0 + row >= p.Nq) return;          // (the merge never reads rows past Nq: a decode-sized call, Nq = 1, writes one row per part, not 256)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = acc[0][dt];
                    const f32x4 w = {a[4 * g] * inv_l, a[4 * g + 1] * inv_l, a[4 * g + 2] * inv_l, a[4 * g + 3] * inv_l};
                    *(f32x4*)(wo + ((dt * 4 + g) * kSplitRows + row) * 8 + 4 * hi) = w;
                }
            if (hi == 0) wl[row] = m_run[0] * c + __builtin_amdgcn_logf(l_tot);
            return;
        }
    }
    if constexpr (QB == 1) {
        constexpr int EROW = HDV * 2 + 16;                       // bytes per staged row
        constexpr int LPR = HDV * 2 / 16;                        // lanes (16-B pieces) per row
        constexpr int RPI = 64 / LPR;                            // rows per store instruction
        __syncthreads();                                         // every wave is done reading the K / V buffers
        char* img = smem + wave * (32 * EROW);
        const float l_tot = half_swap_sum(l_run[0]);
        const float inv_l = (BIAS && !(l_tot > 0.f)) ? 0.f : 1.0f / l_tot;   // fully masked row: O = 0 (lse = -inf)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4 += 2) {
                const f32x16& a = acc[0][dt];
                uint32_t a0 = pack2<BF16>(a[4 * r4 + 0] * inv_l, a[4 * r4 + 1] * inv_l);
                uint32_t a1 = pack2<BF16>(a[4 * r4 + 2] * inv_l, a[4 * r4 + 3] * inv_l);
                uint32_t b0 = pack2<BF16>(a[4 * r4 + 4] * inv_l, a[4 * r4 + 5] * inv_l);
                uint32_t b1 = pack2<BF16>(a[4 * r4 + 6] * inv_l, a[4 * r4 + 7] * inv_l);
                auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                *(u32x4*)(img + l31 * EROW + (32 * dt + 8 * (r4 + hi)) * 2) = (u32x4){x0[0], x1[0], x0[1], x1[1]};
            }
        }
        if (qrow[0] < p.Nq && hi == 0 && vcol0 == 0)
            p.lse[b * p.ls[0] + h * p.ls[1] + qrow[0]] = m_run[0] * c + __builtin_amdgcn_logf(l_tot);
        // read back row-major (wave-private image: the compiler's lgkmcnt wait orders write -> read)
        const int rl = lane / LPR, cl = lane % LPR;              // row within the group of RPI rows, 16-B piece within the row
        uint16_t* obase = (uint16_t*)p.o + b * p.os[0] + h * p.os[1] + vcol0;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int r = i * RPI + rl;
            const u32x4 w = *(const u32x4*)(img + r * EROW + cl * 16);
            if (qw0 + r < p.Nq && vcol0 + cl * 8 < p.D) *(u32x4*)(obase + (int64_t)(qw0 + r) * p.os[2] + cl * 8) = w;
        }
        return;
    }
#endif
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = half_swap_sum(l_run[qb]);
        const float inv_l = (BIAS && !(l_tot > 0.f)) ? 0.f : 1.0f / l_tot;
        if (qrow[qb] < p.Nq) {
            uint16_t* op = (uint16_t*)p.o + b * p.os[0] + h * p.os[1] + (int64_t)qrow[qb] * p.os[2] + vcol0;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r4 = 0; r4 < 4; r4 += 2) {
                    const f32x16& a = acc[qb][dt];
                    // lane holds d = 32dt + 8*r4 + 4*hi + {0..3} (group r4) and the same for r4+1
                    uint32_t a0 = pack2<BF16>(a[4 * r4 + 0] * inv_l, a[4 * r4 + 1] * inv_l);
                    uint32_t a1 = pack2<BF16>(a[4 * r4 + 2] * inv_l, a[4 * r4 + 3] * inv_l);
                    uint32_t b0 = pack2<BF16>(a[4 * r4 + 4] * inv_l, a[4 * r4 + 5] * inv_l);
                    uint32_t b1 = pack2<BF16>(a[4 * r4 + 6] * inv_l, a[4 * r4 + 7] * inv_l);
                    // half exchange: lower lanes end with 8 consecutive d of group r4, upper of r4+1
                    auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const u32x4 w = {x0[0], x1[0], x0[1], x1[1]};
                    if (vcol0 + 32 * dt + 8 * (r4 + hi) < p.D) *(u32x4*)(op + 32 * dt + 8 * (r4 + hi)) = w;
                }
            }
            if (hi == 0 && vcol0 == 0)
                p.lse[b * p.ls[0] + h * p.ls[1] + qrow[qb]] = m_run[qb] * c + __builtin_amdgcn_logf(l_tot);
        }
    }
}

// Merge of the KV-split parts (fa2_fwd_ws): for every row of a split item, lse = log2 sum_i 2^lse_i and O = sum_i 2^(lse_i - lse) O_i
// over the item's nsplit partial results, rounded once to the I/O dtype.  One thread per (row, 8 output columns): the two float4 reads
// per part are 32 consecutive bytes of the layout the parts wrote, the 16-byte stores of 16 (or 8) consecutive threads one output row.
template <int HD, bool BF16>
__global__ __launch_bounds__(256) void fwd_combine_kernel(const FwdParams p) {
    constexpr int CPR = HD / 8;                                  // threads per row
    const int t = blockIdx.x * 256 + threadIdx.x;                // over split_items * kSplitRows * CPR
    const int sidx = t / (kSplitRows * CPR), rem = t % (kSplitRows * CPR);
    const int row = rem / CPR, c8 = rem % CPR;
    if (sidx >= p.split_items) return;
    int bh, qblk;
    block_to_head_qblock<false>(p, p.full_items + sidx, bh, qblk);
    const int b = bh / p.H, h = bh % p.H;
    const int qr = qblk * kSplitRows + row;
    if (qr >= p.Nq || 8 * c8 >= p.D) return;
    const float* wl = p.ws + (int64_t)p.split_items * p.nsplit * kSplitRows * HD + (int64_t)sidx * p.nsplit * kSplitRows + row;
    float li[kMaxSplit], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxSplit; ++i)
        if (i < p.nsplit) { li[i] = wl[i * kSplitRows]; m = __builtin_fmaxf(m, li[i]); }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxSplit; ++i)
        if (i < p.nsplit) { li[i] = __builtin_amdgcn_exp2f(li[i] - m); sum += li[i]; }
    const float inv = 1.0f / sum;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* wo = p.ws + (int64_t)sidx * p.nsplit * kSplitRows * HD + (c8 * kSplitRows + row) * 8;
#pragma unroll
    for (int i = 0; i < kMaxSplit; ++i)
        if (i < p.nsplit) {
            const f32x4 lo = *(const f32x4*)(wo + (int64_t)i * kSplitRows * HD), hi4 = *(const f32x4*)(wo + (int64_t)i * kSplitRows * HD + 4);
            const float w = li[i] * inv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += w * lo[e]; o[4 + e] += w * hi4[e]; }
        }
    const u32x4 w16 = {pack2<BF16>(o[0], o[1]), pack2<BF16>(o[2], o[3]), pack2<BF16>(o[4], o[5]), pack2<BF16>(o[6], o[7])};
    *(u32x4*)((uint16_t*)p.o + b * p.os[0] + h * p.os[1] + (int64_t)qr * p.os[2] + 8 * c8) = w16;
    if (c8 == 0) p.lse[b * p.ls[0] + h * p.ls[1] + qr] = m + __builtin_amdgcn_logf(sum);
}

}  // namespace fa2
