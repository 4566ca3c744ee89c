/*
 * fa2_oracle.c — CPU restatement of the reference's FlashAttention-2 FORWARD algorithm.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file.  The product path
 * (flash-attention-v2-rdna3-minimal_amd/) never links or imports it.
 *
 * Parity pinned: tests/test_oracle.py checks this restatement against golden vectors produced by
 * importing the reference's own oracle pure_torch_ver.py in the build container
 * (tests/golden/make_golden.py, fixtures tests/golden/<case>.npz) and against fp32/fp64 dense attention.
 *
 * What is restated (citations relative to the reference tree):
 *   - tiling and loop order: one Q row-block of Br rows against Tc KV blocks of Bc rows
 *     (rocwmma_fattn/kernel_fp16.cu:381-508 hot loop; pure_torch_ver.py:52-79)
 *   - scores in the log2 domain: S = (Q K^T) * scale * log2(e)   (kernel_fp16.cu:392-395, :827)
 *   - causal mask, top-left aligned: column > row is masked       (kernel_fp16.cu:403-410,
 *     pure_torch_ver.py:64-69); ragged tail: columns >= Nkv masked (kernel_fp16.cu:415-431)
 *   - online softmax: m_new = max(m_old, rowmax S); alpha = 2^(m_old - m_new);
 *     P = 2^(S - m_new); l = alpha*l + rowsum P; O = alpha*O + P V   (kernel_fp16.cu:434-505)
 *   - epilogue: O = O / l; L = m + log2(l)  (log2-domain LSE)      (kernel_fp16.cu:510-543)
 *
 * Precision model.  P is always rounded to the I/O dtype before P.V (both the reference kernels and
 * the gfx950 kernel feed 16-bit P to the matrix unit).  `flags` selects where ELSE values are rounded:
 *   0                      f32 running state (m, l, O accumulate in f32; one final rounding of O) — the
 *                          contract of the gfx950 kernel (include/fa2_gfx950.h)
 *   FA2_ORACLE_ROUND_S     S rounded to the I/O dtype after scaling — the reference stores S as
 *                          16-bit in LDS (kernel_fp16.cu:164-171)
 *   FA2_ORACLE_ROUND_O     O accumulator rounded to the I/O dtype after every KV block, as the
 *                          reference's 16-bit LDS accumulator is (kernel_fp16.cu:223-228, :483-488),
 *                          and m, l kept in the I/O dtype as pure_torch_ver.py:54-58 does
 *   FA2_ORACLE_BF16_TRUNC  bf16 conversions truncate instead of RNE (kernel_bf16.cu:62-72)
 *   FA2_ORACLE_LSUM_P16    the row sum l adds the ROUNDED P (the values the P.V product consumes) instead of the f32 P: the
 *                          contract of the gfx950 head-dim-64 body, whose row sums ride the matrix pipe (csrc/gen/fwd_d128_gen.py "lmfma")
 *   FA2_ORACLE_PRESCALE_FUSED  (with PRESCALE_Q, fp16 only) the product Q * scale*log2(e) is rounded to fp16 ONCE, from the exact
 *                          product, as the gfx950 fp16 kernels' v_fma_mixlo_f16 does.  pure_torch_ver.py:61 (and PRESCALE_Q alone)
 *                          rounds it to f32 first and to fp16 after: the two differ by one fp16 ulp on the rare products whose f32
 *                          rounding lands on an fp16 tie — how rare depends on the bit pattern of scale*log2(e): D = 112 and 56
 *                          are unlucky (found on the GPU in round 6: 17 rows of 1024 off by > 5e-4 of LSE on N(0, 6^2) logits)
 *   FA2_ORACLE_PRESCALE_Q  Q is multiplied by scale*log2(e) and rounded back to the I/O dtype BEFORE the
 *                          product, as pure_torch_ver.py:61 does (`scale * q_frags[Tr_i]`); S = Q' K^T is then
 *                          not scaled again.  The gfx950 kernels use this contract where
 *                          fa2_fwd_prescales_q(D, scale) says so (include/fa2_gfx950.h).
 *
 * Build: oracle/Makefile (gcc -O3 -mavx2 -fopenmp -ffp-contract=off -shared -fPIC).  Plain C99.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FA2_ORACLE_DTYPE_F16 0
#define FA2_ORACLE_DTYPE_BF16 1
#define FA2_ORACLE_ROUND_S 1
#define FA2_ORACLE_ROUND_O 2
#define FA2_ORACLE_BF16_TRUNC 4
#define FA2_ORACLE_PRESCALE_Q 8
#define FA2_ORACLE_LSUM_P16 16
#define FA2_ORACLE_PRESCALE_FUSED 32

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals and inf/nan handled */
static float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        int e = -1;
        do { man <<= 1; ++e; } while (!(man & 0x400u));
        man &= 0x3ffu;
        return u2f(sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13));
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 112) << 23) | (man << 13));
}

static uint16_t f32_to_f16(float f) {
    const uint32_t x = f2u(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to >= 65520 -> inf */
    if (ax < 0x38800000u) {                                    /* subnormal half or zero */
        if (ax < 0x33000000u) return sign;                     /* < 2^-25 -> 0 */
        const int e = (int)(ax >> 23);
        const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                             /* 14..24 */
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ax - 0x38000000u;                             /* rebias */
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}

static float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }

static uint16_t f32_to_bf16(float f, int trunc) {
    uint32_t x = f2u(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u); /* nan */
    if (!trunc) x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

typedef struct {
    int dtype, trunc;
} cvt_t;

/* exact double -> nearest fp16 value (ties to even), returned as float: the single rounding of a fused multiply-to-half */
static float f16_round_double(double p) {
    if (!(p == p) || p == 0.0 || isinf(p)) return (float)p;
    int e;
    (void)frexp(p, &e);                 /* |p| = f * 2^e, f in [0.5, 1) -> exponent of the leading bit: e - 1 */
    int lead = e - 1;
    if (lead < -14) lead = -14;         /* subnormal halves share the exponent -14 */
    const double r = nearbyint(ldexp(p, 10 - lead));        /* 11 significant bits (fewer below 2^-14), round-to-nearest-even */
    const double v = ldexp(r, lead - 10);
    if (fabs(v) >= 65520.0) return (float)(p < 0 ? -INFINITY : INFINITY);
    return (float)v;                    /* exactly representable in fp16, hence in f32 */
}

static inline float load16(cvt_t c, uint16_t h) { return c.dtype == FA2_ORACLE_DTYPE_F16 ? f16_to_f32(h) : bf16_to_f32(h); }
static inline uint16_t store16(cvt_t c, float f) { return c.dtype == FA2_ORACLE_DTYPE_F16 ? f32_to_f16(f) : f32_to_bf16(f, c.trunc); }
static inline float round16(cvt_t c, float f) { return load16(c, store16(c, f)); }

/* exported so the tests can pin the converters against numpy/torch bit patterns */
uint16_t fa2_oracle_f32_to_f16(float f) { return f32_to_f16(f); }
float fa2_oracle_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint16_t fa2_oracle_f32_to_bf16(float f, int trunc) { return f32_to_bf16(f, trunc); }
float fa2_oracle_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }

/*
 * Forward attention on the CPU.  Layout and strides exactly as fa2_fwd (include/fa2_gfx950.h):
 * element strides {batch, head, row}, last dim contiguous; lse strides {batch, head}.
 * Br/Bc: row-block / KV-block sizes of the tiling (reference: 64/128, FlashAttn.py:56-57;
 * pure_torch_ver.py:24 defaults 64/256; the gfx950 kernel: 32 rows per wave / 64).
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
static int fwd_impl(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse,
                    int B, int H, int Nq, int Nkv, int D,
                    const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* ls,
                    float scale, int causal, int Br, int Bc, int flags, int nthreads,
                    const float* bias, const int64_t* bs) {
    if (!q || !k || !v || !o || !lse || B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1 || Br < 1 || Bc < 1) return -1;
    if (dtype != FA2_ORACLE_DTYPE_F16 && dtype != FA2_ORACLE_DTYPE_BF16) return -1;
    const cvt_t cv = {dtype, (flags & FA2_ORACLE_BF16_TRUNC) != 0};
    const float c = scale * 1.4426950408889634f; /* kernel_fp16.cu:827 */
    const int Tr = (Nq + Br - 1) / Br, Tc = (Nkv + Bc - 1) / Bc;
    const int round_s = flags & FA2_ORACLE_ROUND_S, round_o = flags & FA2_ORACLE_ROUND_O;
    const int prescale = flags & FA2_ORACLE_PRESCALE_Q, lsum16 = flags & FA2_ORACLE_LSUM_P16, fused = flags & FA2_ORACLE_PRESCALE_FUSED;
    const float cs = prescale ? 1.f : c; /* factor applied to the f32 dot product */
    int failed = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    /* f32 copies of K and V (dense [B*H][Nkv][D]), converted once and shared by all row-blocks */
    const size_t head_elems = (size_t)Nkv * D;
    float* kf_all = (float*)malloc((size_t)B * H * head_elems * sizeof(float));
    float* vf_all = (float*)malloc((size_t)B * H * head_elems * sizeof(float));
    if (!kf_all || !vf_all) { free(kf_all); free(vf_all); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)B * H * Nkv; ++row) {
        const int bh = (int)(row / Nkv), j = (int)(row % Nkv);
        const int b = bh / H, h = bh % H;
        const uint16_t* kr = k + b * ks[0] + h * ks[1] + (int64_t)j * ks[2];
        const uint16_t* vr = v + b * vs[0] + h * vs[1] + (int64_t)j * vs[2];
        for (int d = 0; d < D; ++d) {
            kf_all[(size_t)row * D + d] = load16(cv, kr[d]);
            vf_all[(size_t)row * D + d] = load16(cv, vr[d]);
        }
    }

#pragma omp parallel
    {
        /* per-thread scratch: one row-block of Q, S, O and the running statistics */
        float* qf = (float*)malloc((size_t)Br * D * sizeof(float));
        float* S = (float*)malloc((size_t)Br * Bc * sizeof(float));
        float* O = (float*)malloc((size_t)Br * D * sizeof(float));
        float* m = (float*)malloc((size_t)Br * sizeof(float));
        float* l = (float*)malloc((size_t)Br * sizeof(float));
        if (!qf || !S || !O || !m || !l) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for collapse(2) schedule(dynamic, 1)
            for (int bh = 0; bh < B * H; ++bh) {
                for (int tr = 0; tr < Tr; ++tr) {
                    const int b = bh / H, h = bh % H;
                    const float* kf = kf_all + (size_t)bh * head_elems;
                    const float* vf = vf_all + (size_t)bh * head_elems;
                    const int r0 = tr * Br, rows = (r0 + Br <= Nq) ? Br : Nq - r0;
                    const uint16_t* qb = q + b * qs[0] + h * qs[1];
                    for (int i = 0; i < rows; ++i) {
                        for (int d = 0; d < D; ++d) {
                            float x = load16(cv, qb[(int64_t)(r0 + i) * qs[2] + d]);
                            if (prescale && fused && dtype == FA2_ORACLE_DTYPE_F16) x = f16_round_double((double)x * (double)c);
                            else if (prescale) x = round16(cv, x * c); /* pure_torch_ver.py:61 */
                            qf[(size_t)i * D + d] = x;
                        }
                        m[i] = -INFINITY;
                        l[i] = 0.f;
                        memset(O + (size_t)i * D, 0, D * sizeof(float));
                    }
                    for (int tc = 0; tc < Tc; ++tc) {
                        const int c0 = tc * Bc, cols = (c0 + Bc <= Nkv) ? Bc : Nkv - c0;
                        /* blocks entirely above the diagonal contribute nothing (P == 0) */
                        if (causal && c0 > r0 + rows - 1) break;
                        for (int i = 0; i < rows; ++i) {
                            float* Si = S + (size_t)i * Bc;
                            const float* qi = qf + (size_t)i * D;
                            /* S = Q K^T * scale*log2e, f32 accumulate (kernel_fp16.cu:115-175) */
                            float mx = -INFINITY;
                            for (int j = 0; j < cols; ++j) {
                                const float* kj = kf + (size_t)(c0 + j) * D;
                                float acc = 0.f;
#pragma omp simd reduction(+ : acc)
                                for (int d = 0; d < D; ++d) acc += qi[d] * kj[d];
                                float s = acc * cs;
                                /* attention bias (fa2_fwd_bias): natural-log units -> log2 domain, added to the scaled product */
                                if (bias) s += bias[b * bs[0] + h * bs[1] + (int64_t)(r0 + i) * bs[2] + c0 + j] * 1.4426950408889634f;
                                if (round_s) s = round16(cv, s);
                                if (causal && c0 + j > r0 + i) s = -INFINITY; /* kernel_fp16.cu:403-410 */
                                Si[j] = s;
                                mx = s > mx ? s : mx;
                            }
                            /* online softmax (kernel_fp16.cu:434-490) */
                            const float m_new = mx > m[i] ? mx : m[i];
                            float alpha = exp2f(m[i] - m_new);
                            if (m_new == -INFINITY) alpha = 1.f; /* row fully masked so far: nothing accumulated */
                            float rs = 0.f;
                            for (int j = 0; j < cols; ++j) {
                                const float p = (m_new == -INFINITY) ? 0.f : exp2f(Si[j] - m_new);
                                Si[j] = round16(cv, p);  /* P fed to the matrix unit in the I/O dtype */
                                rs += lsum16 ? Si[j] : p; /* row sum of the unrounded P (kernel_fp16.cu:455-479), or of what P.V consumes */
                            }
                            float* Oi = O + (size_t)i * D;
                            l[i] = l[i] * alpha + rs;
                            if (round_o) { l[i] = round16(cv, l[i]); alpha = round16(cv, alpha); }
                            for (int d = 0; d < D; ++d) Oi[d] *= alpha;
                            /* O += P V (kernel_fp16.cu:178-232) */
                            for (int j = 0; j < cols; ++j) {
                                const float p = Si[j];
                                if (p == 0.f) continue;
                                const float* vj = vf + (size_t)(c0 + j) * D;
                                for (int d = 0; d < D; ++d) Oi[d] += p * vj[d];
                            }
                            if (round_o)
                                for (int d = 0; d < D; ++d) Oi[d] = round16(cv, Oi[d]);
                            m[i] = m_new;
                        }
                    }
                    /* epilogue (kernel_fp16.cu:510-543) */
                    uint16_t* ob = o + b * os[0] + h * os[1];
                    float* lb = lse + b * ls[0] + h * ls[1];
                    for (int i = 0; i < rows; ++i) {
                        const float inv = l[i] > 0.f ? 1.0f / l[i] : 0.f; /* fully masked row (bias only): O = 0, L = -inf */
                        for (int d = 0; d < D; ++d) ob[(int64_t)(r0 + i) * os[2] + d] = store16(cv, O[(size_t)i * D + d] * inv);
                        lb[r0 + i] = m[i] + log2f(l[i]);
                    }
                }
            }
        }
        free(qf); free(S); free(O); free(m); free(l);
    }
    free(kf_all); free(vf_all);
    return failed ? -1 : 0;
}

int fa2_oracle_fwd(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse,
                   int B, int H, int Nq, int Nkv, int D,
                   const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* ls,
                   float scale, int causal, int Br, int Bc, int flags, int nthreads) {
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, qs, ks, vs, os, ls, scale, causal, Br, Bc, flags, nthreads, NULL, NULL);
}

/*
 * Forward with an additive attention bias (the checker of fa2_fwd_bias, include/fa2_gfx950.h): S = Q K^T * scale + bias, then the
 * same recurrence.  The reference has no counterpart to pin this against — its `mask` argument is accepted and ignored
 * (rocwmma_fattn/FlashAttn.py:49, :74; README.md:45 "to do") — so this entry is pinned on dense float64 attention only
 * (tests/test_oracle.py).  bias: f32, element (b,h,i,j) at bias + b*bs[0] + h*bs[1] + i*bs[2] + j (strides may be 0); -inf masks a
 * position; a fully masked row yields O = 0, L = -inf.
 */
int fa2_oracle_fwd_bias(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse,
                        int B, int H, int Nq, int Nkv, int D,
                        const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* ls,
                        float scale, int causal, int Br, int Bc, int flags, int nthreads,
                        const float* bias, const int64_t* bs) {
    if (!bias || !bs) return -1;
    return fwd_impl(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, qs, ks, vs, os, ls, scale, causal, Br, Bc, flags, nthreads, bias, bs);
}

/*
 * Backward attention on the CPU — restates the reference's backward recurrence
 * (rocwmma_fattn/kernel_fp16.cu:547-740 bwd_kernel, pure_torch_ver.py:92-153):
 *   D_i  = rowsum(dO_i * O_i)                                    (kernel_fp16.cu:605-631, pure_torch_ver.py:146)
 *   P    = 2^(S*c - L_i)   with L the forward's log2-domain LSE   (kernel_fp16.cu:684-708), masked entries 0
 *   dV  += P^T dO                                                 (:724, pure_torch_ver.py:144)
 *   dP   = dO V^T                                                 (:725)
 *   dS   = scale * P * (dP - D_i)                                 (:727-735)
 *   dQ  += dS K,  dK += dS^T Q                                    (:736-737)
 * f32 accumulation; P and dS are rounded to the I/O dtype before they enter a matrix product (both the
 * reference kernels and the gfx950 kernels feed 16-bit operands to the matrix unit); one final rounding of
 * dQ, dK, dV.  Layouts as fa2_oracle_fwd; lse = the forward's output (log2 domain), length >= Nq per head.
 */
static int bwd_impl(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* o,
                   const uint16_t* dout, const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv,
                   int B, int H, int Nq, int Nkv, int D,
                   const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* dos,
                   const int64_t* ls, const int64_t* dqs, const int64_t* dks, const int64_t* dvs,
                   float scale, int causal, int flags, int nthreads, const float* bias, const int64_t* bs) {
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || B < 1 || H < 1 || Nq < 1 || Nkv < 1 || D < 1) return -1;
    if (dtype != FA2_ORACLE_DTYPE_F16 && dtype != FA2_ORACLE_DTYPE_BF16) return -1;
    const cvt_t cv = {dtype, (flags & FA2_ORACLE_BF16_TRUNC) != 0};
    const float c = scale * 1.4426950408889634f;
    int failed = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < B * H; ++bh) {
        const int b = bh / H, h = bh % H;
        float* qf = (float*)malloc((size_t)Nq * D * sizeof(float));
        float* dof = (float*)malloc((size_t)Nq * D * sizeof(float));
        float* kf = (float*)malloc((size_t)Nkv * D * sizeof(float));
        float* vf = (float*)malloc((size_t)Nkv * D * sizeof(float));
        float* dqa = (float*)calloc((size_t)Nq * D, sizeof(float));
        float* dka = (float*)calloc((size_t)Nkv * D, sizeof(float));
        float* dva = (float*)calloc((size_t)Nkv * D, sizeof(float));
        float* delta = (float*)malloc((size_t)Nq * sizeof(float));
        if (!qf || !dof || !kf || !vf || !dqa || !dka || !dva || !delta) {
#pragma omp atomic write
            failed = 1;
        } else {
            const uint16_t* qb = q + b * qs[0] + h * qs[1];
            const uint16_t* ob = o + b * os[0] + h * os[1];
            const uint16_t* gb = dout + b * dos[0] + h * dos[1];
            const uint16_t* kb = k + b * ks[0] + h * ks[1];
            const uint16_t* vb = v + b * vs[0] + h * vs[1];
            const float* lb = lse + b * ls[0] + h * ls[1];
            for (int i = 0; i < Nq; ++i) {
                float dsum = 0.f;
                for (int d = 0; d < D; ++d) {
                    qf[(size_t)i * D + d] = load16(cv, qb[(int64_t)i * qs[2] + d]);
                    dof[(size_t)i * D + d] = load16(cv, gb[(int64_t)i * dos[2] + d]);
                    dsum += dof[(size_t)i * D + d] * load16(cv, ob[(int64_t)i * os[2] + d]);
                }
                delta[i] = dsum;
            }
            for (int j = 0; j < Nkv; ++j)
                for (int d = 0; d < D; ++d) {
                    kf[(size_t)j * D + d] = load16(cv, kb[(int64_t)j * ks[2] + d]);
                    vf[(size_t)j * D + d] = load16(cv, vb[(int64_t)j * vs[2] + d]);
                }
            for (int i = 0; i < Nq; ++i) {
                const float* qi = qf + (size_t)i * D;
                const float* gi = dof + (size_t)i * D;
                float* dqi = dqa + (size_t)i * D;
                const int jmax = causal ? (i < Nkv - 1 ? i : Nkv - 1) : Nkv - 1;   /* column > row masked */
                for (int j = 0; j <= jmax; ++j) {
                    const float* kj = kf + (size_t)j * D;
                    const float* vj = vf + (size_t)j * D;
                    float s = 0.f, dp = 0.f;
#pragma omp simd reduction(+ : s, dp)
                    for (int d = 0; d < D; ++d) { s += qi[d] * kj[d]; dp += gi[d] * vj[d]; }
                    /* attention bias (fa2_bwd_bias): P = 2^(s c + bias log2e - L) with the L of the BIASED forward; a fully masked row
                       (L = -inf) and a masked position (bias = -inf) have P = 0.  The bias itself receives no gradient. */
                    float x = s * c;
                    if (bias) x += bias[b * bs[0] + h * bs[1] + (int64_t)i * bs[2] + j] * 1.4426950408889634f;
                    const float p = (lb[i] == -INFINITY || x == -INFINITY) ? 0.f : exp2f(x - lb[i]);
                    const float p16 = round16(cv, p);
                    const float ds16 = round16(cv, scale * p * (dp - delta[i]));
                    float* dvj = dva + (size_t)j * D;
                    float* dkj = dka + (size_t)j * D;
                    for (int d = 0; d < D; ++d) {
                        dvj[d] += p16 * gi[d];
                        dqi[d] += ds16 * kj[d];
                        dkj[d] += ds16 * qi[d];
                    }
                }
            }
            uint16_t* dqb = dq + b * dqs[0] + h * dqs[1];
            uint16_t* dkb = dk + b * dks[0] + h * dks[1];
            uint16_t* dvb = dv + b * dvs[0] + h * dvs[1];
            for (int i = 0; i < Nq; ++i)
                for (int d = 0; d < D; ++d) dqb[(int64_t)i * dqs[2] + d] = store16(cv, dqa[(size_t)i * D + d]);
            for (int j = 0; j < Nkv; ++j)
                for (int d = 0; d < D; ++d) {
                    dkb[(int64_t)j * dks[2] + d] = store16(cv, dka[(size_t)j * D + d]);
                    dvb[(int64_t)j * dvs[2] + d] = store16(cv, dva[(size_t)j * D + d]);
                }
        }
        free(qf); free(dof); free(kf); free(vf); free(dqa); free(dka); free(dva); free(delta);
    }
    return failed ? -1 : 0;
}

int fa2_oracle_bwd(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* o,
                   const uint16_t* dout, const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv,
                   int B, int H, int Nq, int Nkv, int D,
                   const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* dos,
                   const int64_t* ls, const int64_t* dqs, const int64_t* dks, const int64_t* dvs,
                   float scale, int causal, int flags, int nthreads) {
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, B, H, Nq, Nkv, D, qs, ks, vs, os, dos, ls, dqs, dks, dvs, scale, causal, flags,
                    nthreads, NULL, NULL);
}

/* Backward through a biased / masked forward (the counterpart of fa2_oracle_fwd_bias; C-ABI fa2_bwd_bias).  The reference has neither
 * (its `mask` argument is ignored, FlashAttn.py:49/:74): pinned on float64 autograd of the dense formula (tests/test_oracle.py). */
int fa2_oracle_bwd_bias(int dtype, const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* o,
                        const uint16_t* dout, const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv,
                        int B, int H, int Nq, int Nkv, int D,
                        const int64_t* qs, const int64_t* ks, const int64_t* vs, const int64_t* os, const int64_t* dos,
                        const int64_t* ls, const int64_t* dqs, const int64_t* dks, const int64_t* dvs,
                        float scale, int causal, int flags, int nthreads, const float* bias, const int64_t* bs) {
    if (!bias || !bs) return -1;
    return bwd_impl(dtype, q, k, v, o, dout, lse, dq, dk, dv, B, H, Nq, Nkv, D, qs, ks, vs, os, dos, ls, dqs, dks, dvs, scale, causal, flags,
                    nthreads, bias, bs);
}

int fa2_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
