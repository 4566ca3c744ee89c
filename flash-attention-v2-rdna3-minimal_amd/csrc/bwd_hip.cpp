// bwd_hip.cpp — instantiations and launchers of the compiler-scheduled backward kernels (fa2_bwd_kernel.hip.h) for ONE dtype:
// build.py compiles this file twice, -DFA2_TU_BF16=0 and =1.  Reference counterpart: backward_fp16 / backward_bf16
// (kernel_fp16.cu:878-1028).
#include "fa2_launch.h"

#include "fa2_bwd_short.hip.h"
#include "fa2_gfx950.h"

#ifndef FA2_TU_BF16
#error "compile with -DFA2_TU_BF16=0 or 1"
#endif
// -DFA2_TU_TRIM=1: this unit holds the TRIMMED instantiations instead (head dims well below the kernel's HD run only the MFMA k-steps and
// accumulator column blocks that hold real columns: fa2_bwd_kernel.hip.h, KSN / DTN) and exports launch_bwd_hip_trim_{f16,bf16}.
#ifndef FA2_TU_TRIM
#define FA2_TU_TRIM 0
#endif
#ifndef FA2_TRIM          // 0: never dispatch to the trimmed kernels (A/B builds, tools/kbench.py)
#define FA2_TRIM 1
#endif
#ifndef FA2_BWD_FUSE256   // 0: trimmed head dims 129..224 keep the separate dV and dK sweeps (A/B builds)
#define FA2_BWD_FUSE256 1
#endif

namespace {

constexpr bool kBF16 = FA2_TU_BF16 != 0;

template <int HD, bool CAUSAL, int KSN, int DTN>
int launch_bwd_pair(const fa2::BwdParams& p, hipStream_t stream) {
    constexpr int lds = 2 * (4 * fa2::Geo<HD, 8>::TILEB + 512) + 4 * 4096;
    constexpr auto kern = fa2::bwd_dkv_pair_kernel<HD, kBF16, CAUSAL, KSN, DTN>;
    if (int rc = fa2::set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(512), lds, stream, p);
    return (int)hipGetLastError();
}

// sum of the parts a split pass left in p.ws (bwd_merge_kernel): which = 1: dQ, 2: dK and dV
template <int HD>
int launch_merge(const fa2::BwdParams& p, int which, hipStream_t stream) {
    if constexpr (HD <= 128) {
        fa2::BwdMergeParams m;
        m.ws = p.ws;
        m.H = p.H; m.nbh = p.B * p.H; m.nblk = p.nblk; m.D = p.D;
        m.full_items = p.full_items; m.split_items = p.split_items; m.nsplit = p.nsplit;
        if (which == 1) {
            m.out[0] = p.dq; m.out[1] = nullptr; m.mul[0] = p.scale; m.mul[1] = 0.f; m.nrows = p.Nq;
            for (int i = 0; i < 3; ++i) { m.os[0][i] = p.dqs[i]; m.os[1][i] = 0; }
        } else {
            m.out[0] = p.dk; m.out[1] = p.dv; m.mul[0] = p.scale; m.mul[1] = 1.0f; m.nrows = p.Nkv;
            for (int i = 0; i < 3; ++i) { m.os[0][i] = p.dks[i]; m.os[1][i] = p.dvs[i]; }
        }
        const int64_t threads = (int64_t)p.split_items * fa2::kSplitRows * (HD / 8);
        hipLaunchKernelGGL((fa2::bwd_merge_kernel<HD, kBF16>), dim3((unsigned)((threads + 255) / 256), which == 1 ? 1 : 2), dim3(256), 0, stream, m);
        return (int)hipGetLastError();
    } else {
        return FA2_ERR_HEAD_DIM;
    }
}

// parts: bit 0 = the dQ pass (which also fills the delta workspace), bit 1 = the dK / dV pass(es)
template <int HD, bool CAUSAL, int KSN = HD / 16, int DTN = HD / 32>
int launch_bwd_t(fa2::BwdParams p, int parts, hipStream_t stream) {
    constexpr int NW = HD > 128 ? 4 : 8;          // D = 256: one wave per SIMD (512 registers), single LDS stage
    constexpr int kRows = NW * 32, kStages = NW == 8 ? 2 : 1;
    constexpr int TILEB = fa2::Geo<HD, NW>::TILEB;
    int rc;
    // split of a partly filled last round (fa2_bwd_ws: the caller handed over scratch memory; fa2_launch.h has the plan)
    fa2::SplitPlan sp_dq, sp_dkv;
    if (p.ws && (reinterpret_cast<uintptr_t>(p.ws) & 15u) == 0) {
        fa2::plan_bwd_split(HD, p, CAUSAL, &sp_dq, &sp_dkv);
        if ((size_t)sp_dq.bytes > p.ws_bytes) sp_dq = fa2::SplitPlan();
        if ((size_t)sp_dkv.bytes > p.ws_bytes) sp_dkv = fa2::SplitPlan();
    }
    p.nsplit = 0;
    // dQ: one workgroup per kRows Q rows; also writes D_i = rowsum(dO * O) to the delta workspace for the dK pass.
    // Grids that would cover at most half of the CUs with 256-row workgroups (SD-size training shapes) run as 128-row,
    // 4-wave workgroups instead — twice as many, one wave per SIMD each.
    bool dq_small = false;
    if constexpr (NW == 8) {
        const int forced = fa2::options().rows.load(std::memory_order_relaxed);      // option "rows" pins this shape too
        const int64_t w = (int64_t)p.B * p.H * ((p.Nq + 255) / 256), cus = fa2::device_cus();
        // ... and, at head dims <= 64, grids of one to one and a half rounds of 256-row workgroups (the forward's short_second_round): measured
        // (tools/bwd_rows_ab.py, whole backward) SDXL 64x64 B2 H10 N4096 428 -> 396 us, B1 H24 N3072 328 -> 300, N4096 430 -> 410, SD1.5 B3 H8 395 -> 371;
        // at exactly one round (SD1.5 B2 H8: 218 vs 229) and at D = 80 (349 vs 365) the 8-wave shape stays ahead
        dq_small = forced == 128 || (forced != 256 && (w <= cus / 2 || (HD <= 64 && w > cus && w <= cus + cus / 2)));
        if (sp_dq.nsplit > 1) dq_small = false;      // the split last round balances better than smaller workgroups
    }
    if (!(parts & 1)) {
    } else if (dq_small) {
        if constexpr (NW == 8) {
            constexpr int lds = 2 * 3 * fa2::Geo<HD, 4>::TILEB;
            constexpr auto kern = fa2::bwd_dq_kernel<HD, kBF16, CAUSAL, 4, HD, 0, KSN, DTN>;
            if ((rc = fa2::set_lds<kern>(lds))) return rc;
            p.nblk = (p.Nq + 127) / 128;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(256), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
    } else {
        constexpr int lds = kStages * 3 * TILEB;
        constexpr auto kern = fa2::bwd_dq_kernel<HD, kBF16, CAUSAL, NW, HD, 0, KSN, DTN>;
        if ((rc = fa2::set_lds<kern>(lds))) return rc;
        p.nblk = (p.Nq + kRows - 1) / kRows;
        int64_t grid = (int64_t)p.B * p.H * p.nblk;
        if constexpr (NW == 8 && !CAUSAL) {
            if (sp_dq.nsplit > 1) {
                p.full_items = sp_dq.full_items; p.split_items = sp_dq.split_items; p.nsplit = sp_dq.nsplit;
                grid = (int64_t)p.full_items + (int64_t)p.split_items * p.nsplit;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
        if (p.nsplit > 1 && (rc = launch_merge<HD>(p, 1, stream))) return rc;
        p.nsplit = 0;
    }
    if (!(parts & 2)) return 0;
    if constexpr (HD == 128) {
        // D in 65..128: dK and dV in one sweep by wave pairs (bwd_dkv_pair_kernel): 128 KV rows per workgroup, S and P formed once
        p.nblk = (p.Nkv + 127) / 128;
        if ((int64_t)p.B * p.H * p.nblk > 0x7fffffffLL) return FA2_ERR_GRID;
        return launch_bwd_pair<HD, CAUSAL, KSN, DTN>(p, stream);
    } else if constexpr (HD <= 64) {
        // D <= 64: both accumulators fit one wave, one sweep forms S and P once for dK and dV
        p.nblk = (p.Nkv + kRows - 1) / kRows;
        constexpr int lds = kStages * (4 * TILEB + 512);
        constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, true, HD, 0, KSN, DTN>;
        if ((rc = fa2::set_lds<kern>(lds))) return rc;
        int64_t grid = (int64_t)p.B * p.H * p.nblk;
        if constexpr (!CAUSAL) {
            if (sp_dkv.nsplit > 1) {
                p.full_items = sp_dkv.full_items; p.split_items = sp_dkv.split_items; p.nsplit = sp_dkv.nsplit;
                grid = (int64_t)p.full_items + (int64_t)p.split_items * p.nsplit;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
        return p.nsplit > 1 ? launch_merge<HD>(p, 2, stream) : 0;
    } else {
        p.nblk = (p.Nkv + kRows - 1) / kRows;   // dV, dK: one workgroup per kRows KV rows, two sweeps
#if FA2_BWD_FUSE256
        // trimmed head dims 129..224: both KV-owned accumulators (2 x DTN blocks) and the K / V fragments of KSN k-steps fit the 512-register budget
        // of one wave per SIMD, so dK and dV come from ONE sweep that forms S and dP once (4 GEMMs instead of 2 + 3, one launch less).  Measured
        // (tools/trim_ab.py --bwd --dmin 129, profiles/r08_fuse256_ab.txt, whole backward B1 H24 N4096): D 144 / 160 1 277 / 1 287 -> 1 143 / 1 142 us,
        // 176 / 192 1 490 / 1 498 -> 1 304 / 1 319, 208 / 224 1 707 / 1 771 -> 1 627 / 1 716, B1 H8 N1024 D160 115 -> 97
        if constexpr (HD == 256 && !CAUSAL && DTN <= 7) {   // (causal: 11 spilled registers at 5 blocks, 100+ above; measured -5 .. -9 %)
            constexpr int lds = kStages * (4 * TILEB + 512);
            static_assert(lds <= 160 * 1024, "LDS budget");
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, true, HD, 0, KSN, DTN>;
            if ((rc = fa2::set_lds<kern>(lds))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
            return (int)hipGetLastError();
        }
#endif
        {
            constexpr int lds = kStages * (2 * TILEB + 512);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, false, NW, false, HD, 0, KSN, DTN>;
            if ((rc = fa2::set_lds<kern>(lds))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        {
            constexpr int lds = kStages * (3 * TILEB + 512);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, false, HD, 0, KSN, DTN>;
            if ((rc = fa2::set_lds<kern>(lds))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, p);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        return 0;
    }
}

// Head dims above 256 (kernel head dim 512, the SD VAE attention block): 4-wave workgroups of 128 rows, one wave per SIMD with the 512-register
// budget, single LDS stage; every workgroup produces a 128-column slab of its output (grid.y = 4) and recomputes S (and dP) over the whole
// head dim.  Three launches: dQ (+ delta), dV, dK.  A correct path for a rare shape, not a tuned one.
// KSN (trimmed instantiations): ceil(D / 16) k-steps of the products contracted over the head dim; only the slabs that hold real columns are launched.
template <bool CAUSAL, int KSN = 32>
int launch_bwd_512(fa2::BwdParams p, int parts, hipStream_t stream) {
    constexpr int HD = 512, HDV = 128, NW = 4, kRows = NW * 32;
    constexpr int TILEB = fa2::Geo<HD, NW>::TILEB, TILEBV = fa2::Geo<HDV, NW>::TILEB;
    int rc;
    if (parts & 1) {
        constexpr int lds = 2 * TILEB + TILEBV;
        constexpr auto kern = fa2::bwd_dq_kernel<HD, kBF16, CAUSAL, NW, HDV, 0, KSN, HDV / 32>;
        if ((rc = fa2::set_lds<kern>(lds))) return rc;
        p.nblk = (p.Nq + kRows - 1) / kRows;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk), (p.D + HDV - 1) / HDV), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
    }
    if (!(parts & 2)) return 0;
    p.nblk = (p.Nkv + kRows - 1) / kRows;
    {
        constexpr int lds = TILEB + TILEBV + 512;
        constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, false, NW, false, HDV, 0, KSN, HDV / 32>;
        if ((rc = fa2::set_lds<kern>(lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk), (p.D + HDV - 1) / HDV), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
    }
    {
        constexpr int lds = 2 * TILEB + TILEBV + 512;
        constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, false, HDV, 0, KSN, HDV / 32>;
        if ((rc = fa2::set_lds<kern>(lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk), (p.D + HDV - 1) / HDV), dim3(NW * 64), lds, stream, p);
        if ((rc = (int)hipGetLastError())) return rc;
    }
    return 0;
}

#if FA2_TU_TRIM

template <int HD, int KSN, int DTN>
int launch_trim(const fa2::BwdParams& p, bool causal, int parts, hipStream_t stream) {
    return causal ? launch_bwd_t<HD, true, KSN, DTN>(p, parts, stream) : launch_bwd_t<HD, false, KSN, DTN>(p, parts, stream);
}

}  // namespace

namespace fa2 {

// Trimmed backward kernels; -1 = none for this p.D (the caller runs the full kernels).  The head dims the forward trims (fwd_hip.cpp):
//   HD  64: D <= 32 -> 2 k-steps, 1 column block;   HD 128: D <= 96 -> 6, 3;   HD 256: D <= 160 / 192 / 224 -> 10, 5 / 12, 6 / 14, 7
// Measured (tools/trim_ab.py --bwd, profiles/r08_trim_ab_bwd.txt): B1 H24 N4096 D 16 / 32 +30 %, 80 / 96 +17 %, 144 / 160 +36 %, 176 / 192 +19 %, 208 / 224 +7 %.
#if FA2_TU_BF16
int launch_bwd_hip_trim_bf16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream) {
#else
int launch_bwd_hip_trim_f16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream) {
#endif
    switch (HD) {
        case 64:
            if (p.D <= 32) return launch_trim<64, 2, 1>(p, causal, parts, stream);
            return -1;      // (3 k-steps, 2 blocks for D <= 48 measured -1 .. +3 %: the fused D <= 64 pass is not bound by its MFMAs; not instantiated)
        case 128:
            if (p.D <= 80) return launch_trim<128, 5, 3>(p, causal, parts, stream);
            if (p.D <= 96) return launch_trim<128, 6, 3>(p, causal, parts, stream);
            return -1;
        case 256:
            if (p.D <= 160) return launch_trim<256, 10, 5>(p, causal, parts, stream);
            if (p.D <= 192) return launch_trim<256, 12, 6>(p, causal, parts, stream);
            if (p.D <= 224) return launch_trim<256, 14, 7>(p, causal, parts, stream);
            return -1;
        case 512:
            if (p.D <= 320) return causal ? launch_bwd_512<true, 20>(p, parts, stream) : launch_bwd_512<false, 20>(p, parts, stream);
            if (p.D <= 384) return causal ? launch_bwd_512<true, 24>(p, parts, stream) : launch_bwd_512<false, 24>(p, parts, stream);
            if (p.D <= 448) return causal ? launch_bwd_512<true, 28>(p, parts, stream) : launch_bwd_512<false, 28>(p, parts, stream);
            return -1;
        default: return -1;
    }
}

}  // namespace fa2

#else   // !FA2_TU_TRIM

// dQ pass of KV sweeps of at most two tiles (fa2_bwd_short.hip.h): 128-row workgroups; one instantiation per count of 32-key blocks that hold a key
template <int HD, int NB>
int launch_short_dq_nb(const fa2::BwdParams& p, bool neg_delta, hipStream_t stream) {
    constexpr auto kern = fa2::bwd_short_dq_kernel<HD, kBF16, NB>;
    constexpr int lds = fa2::bwd_short_lds_bytes<HD>(NB);
    if (int rc = fa2::set_lds<kern>(lds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(256), lds, stream, p, (int)neg_delta);
    return (int)hipGetLastError();
}

template <int HD>
int launch_short_dq(const fa2::BwdParams& p0, bool neg_delta, hipStream_t stream) {
    fa2::BwdParams p = p0;
    p.nblk = (p.Nq + fa2::kBwdShortRows - 1) / fa2::kBwdShortRows;
    if ((int64_t)p.B * p.H * p.nblk > 0x7fffffffLL) return FA2_ERR_GRID;
    p.nsplit = 0;
    switch ((p.Nkv + 31) / 32) {
        case 1: return launch_short_dq_nb<HD, 1>(p, neg_delta, stream);
        case 2: return launch_short_dq_nb<HD, 2>(p, neg_delta, stream);
        case 3: return launch_short_dq_nb<HD, 3>(p, neg_delta, stream);
        case 4: return launch_short_dq_nb<HD, 4>(p, neg_delta, stream);
        default: return FA2_ERR_BAD_SHAPE;
    }
}

template <int HD>
int launch_bwd(const fa2::BwdParams& p, bool causal, int parts, hipStream_t stream) {
    return causal ? launch_bwd_t<HD, true>(p, parts, stream) : launch_bwd_t<HD, false>(p, parts, stream);
}

}  // namespace

namespace fa2 {

#if FA2_TU_BF16
int launch_bwd_hip_bf16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream) {
#else
int launch_bwd_hip_f16(int HD, const BwdParams& p, bool causal, int parts, hipStream_t stream) {
#endif
    if (FA2_TRIM && p.D < HD) {      // a trimmed instantiation, where one exists
#if FA2_TU_BF16
        const int rc = launch_bwd_hip_trim_bf16(HD, p, causal, parts, stream);
#else
        const int rc = launch_bwd_hip_trim_f16(HD, p, causal, parts, stream);
#endif
        if (rc >= 0) return rc;
    }
    switch (HD) {
        case 64: return launch_bwd<64>(p, causal, parts, stream);
        case 128: return launch_bwd<128>(p, causal, parts, stream);
        case 256: return launch_bwd<256>(p, causal, parts, stream);
        case 512: return causal ? launch_bwd_512<true>(p, parts, stream) : launch_bwd_512<false>(p, parts, stream);
        default: return FA2_ERR_HEAD_DIM;
    }
}

#if FA2_TU_BF16
int launch_bwd_short_dq_bf16(int HD, const BwdParams& p, bool neg_delta, hipStream_t stream) {
#else
int launch_bwd_short_dq_f16(int HD, const BwdParams& p, bool neg_delta, hipStream_t stream) {
#endif
    return HD == 64 ? launch_short_dq<64>(p, neg_delta, stream) : HD == 128 ? launch_short_dq<128>(p, neg_delta, stream) : FA2_ERR_HEAD_DIM;
}

#if FA2_TU_BF16
int launch_bwd_merge_bf16(int HD, const BwdParams& p, int which, hipStream_t stream) {
#else
int launch_bwd_merge_f16(int HD, const BwdParams& p, int which, hipStream_t stream) {
#endif
    return HD == 64 ? launch_merge<64>(p, which, stream) : HD == 128 ? launch_merge<128>(p, which, stream) : FA2_ERR_HEAD_DIM;
}

}  // namespace fa2

#endif  // FA2_TU_TRIM
