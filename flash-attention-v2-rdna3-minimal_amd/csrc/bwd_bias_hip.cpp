// bwd_bias_hip.cpp — backward through a biased / masked forward (fa2_bwd_bias) for ONE dtype: the BIAS instantiations of the
// compiler-scheduled passes (fa2_bwd_kernel.hip.h): dQ, then dK and dV in one sweep at head dims <= 64, dV and dK above.  build.py compiles this file twice (-DFA2_TU_BF16=0 / 1).
// The reference has no counterpart: its `mask` argument is accepted and ignored (FlashAttn.py:49, :74; README.md:45 "to do").
#include "fa2_launch.h"

#include "fa2_gfx950.h"

#ifndef FA2_TU_BF16
#error "compile with -DFA2_TU_BF16=0 or 1"
#endif

namespace {

constexpr bool kBF16 = FA2_TU_BF16 != 0;

template <int HD, bool CAUSAL, int FORM>
int launch_form(fa2::BwdParams p, hipStream_t stream);

// FORM 2: bias tiles by LDS-DMA where the geometry allows and the images fit the LDS; FORM 1: one guarded load per score; FORM 3: one per KV row
template <int HD, bool CAUSAL>
int launch_t(fa2::BwdParams p, hipStream_t stream) {
    constexpr int NW = HD > 128 ? 4 : 8;
    constexpr int TILEB = fa2::Geo<HD, NW>::TILEB, kStages = NW == 8 ? 2 : 1;
    if (p.bias_tile && kStages * ((HD <= 64 ? 4 : 3) * TILEB + 512) + NW * p.bias_img <= 160 * 1024) return launch_form<HD, CAUSAL, 2>(p, stream);
    p.bias_tile = 0;
    // a bias broadcast over the Q rows (a [B, 1, 1, Nkv] key-padding mask): the dK / dV pass loads its lanes' values once (FORM 3)
    if constexpr (HD <= 128) {
        if (p.bs[2] == 0) return launch_form<HD, CAUSAL, 3>(p, stream);
    }
    return launch_form<HD, CAUSAL, 1>(p, stream);
}

template <int HD, bool CAUSAL, int FORM>
int launch_form(fa2::BwdParams p, hipStream_t stream) {
    constexpr int NW = HD > 128 ? 4 : 8;          // D = 256: one wave per SIMD (512 registers), single LDS stage
    constexpr int kRows = NW * 32, kStages = NW == 8 ? 2 : 1;
    constexpr int TILEB = fa2::Geo<HD, NW>::TILEB;
    int rc;
    // bias tiles staged by LDS-DMA (p.bias_tile, host.cpp): NW wave-private images above the stages, where they fit the 160 KiB
    const int img = p.bias_tile ? NW * p.bias_img : 0;
    auto fits = [&](int base) { return base + img <= 160 * 1024; };
    // split of a partly filled last round / of an underfilled KV-owner grid (fa2_bwd_bias_ws: scratch memory from the caller), as in bwd_hip.cpp
    fa2::SplitPlan sp_dq, sp_dkv;
    if constexpr (!CAUSAL && NW == 8) {
        if (p.ws && (reinterpret_cast<uintptr_t>(p.ws) & 15u) == 0) {
            fa2::plan_bwd_split(HD, p, CAUSAL, &sp_dq, &sp_dkv);
            if ((size_t)sp_dq.bytes > p.ws_bytes) sp_dq = fa2::SplitPlan();
            if ((size_t)sp_dkv.bytes > p.ws_bytes) sp_dkv = fa2::SplitPlan();
        }
    }
    auto merge = [&](const fa2::BwdParams& q, int which) { return kBF16 ? fa2::launch_bwd_merge_bf16(HD, q, which, stream) : fa2::launch_bwd_merge_f16(HD, q, which, stream); };
    p.nsplit = 0;
    {
        constexpr int lds0 = kStages * 3 * TILEB;
        fa2::BwdParams pq = p;
        if (!fits(lds0)) return FA2_ERR_BAD_SHAPE;      // unreachable: launch_t sends a call here only when the worst-case stage size fits (the FORM 2 kernels always stage the bias image)
        const int lds = lds0 + (pq.bias_tile ? img : 0);
        constexpr auto kern = fa2::bwd_dq_kernel<HD, kBF16, CAUSAL, NW, HD, FORM == 3 ? 1 : FORM>;
        if ((rc = fa2::set_lds<kern>(160 * 1024))) return rc;
        pq.nblk = (p.Nq + kRows - 1) / kRows;
        int64_t grid = (int64_t)p.B * p.H * pq.nblk;
        if (sp_dq.nsplit > 1) {
            pq.full_items = sp_dq.full_items; pq.split_items = sp_dq.split_items; pq.nsplit = sp_dq.nsplit;
            grid = (int64_t)pq.full_items + (int64_t)pq.split_items * pq.nsplit;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, pq);
        if ((rc = (int)hipGetLastError())) return rc;
        if (pq.nsplit > 1 && (rc = merge(pq, 1))) return rc;
    }
    p.nblk = (p.Nkv + kRows - 1) / kRows;
    if constexpr (HD <= 64 && FORM != 1) {     // (with one load per score, FORM 1, the fused pass spills: two sweeps as above 64)
        // both accumulators fit one wave: dK and dV in ONE sweep (S, the bias and P formed once), like the unmasked backward
        constexpr int lds0 = kStages * (4 * TILEB + 512);
        fa2::BwdParams pk = p;
        if (!fits(lds0)) return FA2_ERR_BAD_SHAPE;      // unreachable: launch_t sends a call here only when the worst-case stage size fits (the FORM 2 kernels always stage the bias image)
        const int lds = lds0 + (pk.bias_tile ? img : 0);
        constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, true, HD, FORM>;
        if ((rc = fa2::set_lds<kern>(160 * 1024))) return rc;
        int64_t grid = (int64_t)p.B * p.H * p.nblk;
        if (sp_dkv.nsplit > 1) {
            pk.full_items = sp_dkv.full_items; pk.split_items = sp_dkv.split_items; pk.nsplit = sp_dkv.nsplit;
            grid = (int64_t)pk.full_items + (int64_t)pk.split_items * pk.nsplit;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, pk);
        if ((rc = (int)hipGetLastError())) return rc;
        return pk.nsplit > 1 ? merge(pk, 2) : 0;
    } else {
        {
            constexpr int lds0 = kStages * (2 * TILEB + 512);
            fa2::BwdParams pv = p;
            if (!fits(lds0)) return FA2_ERR_BAD_SHAPE;      // unreachable: launch_t sends a call here only when the worst-case stage size fits (the FORM 2 kernels always stage the bias image)
            const int lds = lds0 + (pv.bias_tile ? img : 0);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, false, NW, false, HD, FORM>;
            if ((rc = fa2::set_lds<kern>(160 * 1024))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, pv);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        {
            constexpr int lds0 = kStages * (3 * TILEB + 512);
            fa2::BwdParams pk = p;
            if (!fits(lds0)) return FA2_ERR_BAD_SHAPE;      // unreachable: launch_t sends a call here only when the worst-case stage size fits (the FORM 2 kernels always stage the bias image)
            const int lds = lds0 + (pk.bias_tile ? img : 0);
            constexpr auto kern = fa2::bwd_dkv_kernel<HD, kBF16, CAUSAL, true, NW, false, HD, FORM>;
            if ((rc = fa2::set_lds<kern>(160 * 1024))) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.H * p.nblk)), dim3(NW * 64), lds, stream, pk);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        return 0;
    }
}

template <int HD>
int launch_hd(const fa2::BwdParams& p, bool causal, hipStream_t stream) {
    return causal ? launch_t<HD, true>(p, stream) : launch_t<HD, false>(p, stream);
}

}  // namespace

namespace fa2 {

#if FA2_TU_BF16
int launch_bwd_bias_hip_bf16(int HD, const BwdParams& p, bool causal, hipStream_t stream) {
#else
int launch_bwd_bias_hip_f16(int HD, const BwdParams& p, bool causal, hipStream_t stream) {
#endif
    switch (HD) {
        case 64: return launch_hd<64>(p, causal, stream);
        case 128: return launch_hd<128>(p, causal, stream);
        case 256: return launch_hd<256>(p, causal, stream);
        default: return FA2_ERR_HEAD_DIM;       // (the slab kernels of head dims above 256 have no bias form)
    }
}

}  // namespace fa2
