// Micro-benchmarks of gfx950 issue rates that price the softmax VALU work next to MFMA (dev tool).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && ./valu_rate
// Each test runs `waves_per_simd` waves on every SIMD of every CU and reports cycles per instruction
// per SIMD derived from s_memtime (shader clock) on wave 0 of block 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define ITERS 2000

enum Test { T_FMA, T_EXP, T_CVT, T_MAX3, T_PKMUL, T_PKADD, T_ADD, T_MFMA, T_MFMA_FMA4, T_MFMA_EXP4, T_MFMA_FMA6,
            T_MFMA_MIX, T_EXP_FMA, T_MFMA_EXP2FMA2, T_MFMA_FMA8, T_NUM };
static const char* kNames[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_max3_f32", "v_pk_mul_f32", "v_pk_add_f32",
                               "v_add_f32", "mfma32x32x16 alone", "mfma + 4 fma", "mfma + 4 exp", "mfma + 6 fma",
                               "mfma + 2exp 2fma 1cvt 1add", "exp+fma pairs", "mfma + 2exp 2fma", "mfma + 8 fma"};
static const int kInstPerBody[] = {8, 8, 8, 8, 8, 8, 8, 8, 8 * 5, 8 * 5, 8 * 7, 8 * 7, 16, 8 * 5, 8 * 9};

template <int T>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f, a4 = a0 + 1, a5 = a0 + 2, a6 = a0 + 3, a7 = a0 + 4;
    float b0 = a0 + 5, b1 = a0 + 6, b2 = a0 + 7, b3 = a0 + 8;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(seed + i); fb[i] = (_Float16)(seed - i); }
    const float c = seed * 1e-3f;
    long long t0 = __builtin_readcyclecounter();
    t0 = wall_clock64();
    long long s0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (T == T_FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if constexpr (T == T_EXP) {
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == T_CVT) {
            asm volatile(REP8("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %1, %2, %3\n v_cvt_pk_f16_f32 %2, %3, %4\n v_cvt_pk_f16_f32 %3, %4, %5\n"
                              "v_cvt_pk_f16_f32 %4, %5, %6\n v_cvt_pk_f16_f32 %5, %6, %7\n v_cvt_pk_f16_f32 %6, %7, %0\n v_cvt_pk_f16_f32 %7, %0, %1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == T_MAX3) {
            asm volatile(REP8("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                              "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == T_PKMUL || T == T_PKADD) {
            f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {a1, a2}, p7 = {a3, a4}, pc = {c, c};
            if constexpr (T == T_PKMUL)
                asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                  "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            else
                asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                  "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            a0 = p0[0] + p1[0]; a1 = p2[0] + p3[1]; a2 = p4[0] + p5[1]; a3 = p6[0] + p7[1];
        } else if constexpr (T == T_ADD) {
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if constexpr (T == T_EXP_FMA) {
            asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %8\n"
                              "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else {
            // 8 MFMAs on 4 independent accumulators, each followed by a fixed VALU filler group
#define MF(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %12, %13, %" #ACC "\n"
#define F4 "v_fma_f32 %0, %0, %14, %14\n v_fma_f32 %1, %1, %14, %14\n v_fma_f32 %2, %2, %14, %14\n v_fma_f32 %3, %3, %14, %14\n"
#define F2 "v_fma_f32 %4, %4, %14, %14\n v_fma_f32 %5, %5, %14, %14\n"
#define E4 "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
#define E2 "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n"
#define MIX "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %14, %14\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %14, %14\n v_cvt_pk_f16_f32 %4, %5, %6\n v_add_f32 %7, %7, %14\n"
#define BODY(FILL) MF(8) FILL MF(9) FILL MF(10) FILL MF(11) FILL MF(8) FILL MF(9) FILL MF(10) FILL MF(11) FILL
#define RUN(FILL)                                                                                                   \
    asm volatile(BODY(FILL) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),       \
                 "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(fa), "v"(fb), "v"(c))
            if constexpr (T == T_MFMA) RUN("");
            else if constexpr (T == T_MFMA_FMA4) RUN(F4);
            else if constexpr (T == T_MFMA_EXP4) RUN(E4);
            else if constexpr (T == T_MFMA_FMA6) RUN(F4 F2);
            else if constexpr (T == T_MFMA_MIX) RUN(MIX);
            else if constexpr (T == T_MFMA_EXP2FMA2) RUN(E2 F2);
            else if constexpr (T == T_MFMA_FMA8) RUN(F4 F4);
        }
    }
    long long s1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = s1 - s0; cyc[1] = wall_clock64() - t0; }
}

template <int T>
void run(int waves_per_simd, float* out, long long* cyc) {
    const int threads = 64 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double insts = (double)ITERS * kInstPerBody[T] * (T >= T_MFMA && T != T_EXP_FMA ? 1 : 8) * waves_per_simd;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock); convert with the event time instead
    const double cycles_at = ms * 1e-3;  // seconds
    printf("%-28s waves/SIMD=%d  %8.3f ms  %7.3f ns per inst per SIMD  (= %5.2f cyc @2.4GHz, %5.2f @2.0GHz)  memtime=%lld\n",
           kNames[T], waves_per_simd, ms, cycles_at / insts * 1e9, cycles_at / insts * 2.4e9, cycles_at / insts * 2.0e9, h[0]);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 16);
    for (int w = 1; w <= 2; ++w) {
        run<T_FMA>(w, out, cyc); run<T_ADD>(w, out, cyc); run<T_EXP>(w, out, cyc); run<T_EXP_FMA>(w, out, cyc);
        run<T_CVT>(w, out, cyc); run<T_MAX3>(w, out, cyc);
        run<T_PKMUL>(w, out, cyc); run<T_PKADD>(w, out, cyc); run<T_MFMA>(w, out, cyc); run<T_MFMA_FMA4>(w, out, cyc);
        run<T_MFMA_FMA6>(w, out, cyc); run<T_MFMA_FMA8>(w, out, cyc); run<T_MFMA_EXP4>(w, out, cyc);
        run<T_MFMA_EXP2FMA2>(w, out, cyc); run<T_MFMA_MIX>(w, out, cyc);
    }
    return 0;
}
