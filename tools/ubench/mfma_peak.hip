// MFMA peak micro-benchmark for MI355X (gfx950) — the analogue of the reference's GPU_peak_perf_test.cu:38-64 /
// GPU_peak_perf_test.py:41-61 (a WMMA-only loop that prices the RDNA3 matrix rate the attention kernel is compared with).
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak > profiles/mfma_peak.json
//
// Every SIMD of every CU runs v_mfma_f32_32x32x16_{f16,bf16} back to back on 4 independent accumulators per wave,
// with one or two waves per SIMD, on RANDOM operands (uniform [-1,1): the chip clocks to its power budget, zero
// operands run ~20 % faster and are reported separately) — nothing but the matrix pipe, the loop counter and the
// final store.  Launches are sized like the attention kernel (8192 MFMAs per SIMD, ~110 us at 2.4 GHz) and repeated
// for ~0.4 s before the timed launches so that the clock has settled: `sustained_tflops_*` is the rate the matrix pipe
// sustains under load on this chip, the second ("sustained") roofline denominator bench.py reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMfmaPerSimd = 8192;   // per launch

template <bool BF16>
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// THREADS = 256: one wave per SIMD; 512: two waves per SIMD.  One workgroup per CU.
template <bool BF16, int THREADS>
__global__ __launch_bounds__(THREADS) void peak_kernel(const u32x4* __restrict__ operands, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(i * 2 + 0) * 512 + tid];
        b[i] = operands[(i * 2 + 1) * 512 + tid];
    }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma<BF16>(a[(i + r) & 3], b[i], acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * THREADS + tid] = s;   // keep the chains alive, never true in practice
}

static unsigned short f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}
static unsigned short f32_to_bf16_bits(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <bool BF16, int THREADS>
static double run(const u32x4* d_ops, float* d_out, int n_cu, double settle_s, int timed_launches, double* out_ms) {
    const int waves_per_simd = THREADS / 256;
    const int iters = kMfmaPerSimd / waves_per_simd / 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // settle the clock: keep launching for settle_s seconds
    CHECK(hipEventRecord(e0));
    float ms = 0.f;
    do {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((peak_kernel<BF16, THREADS>), dim3(n_cu), dim3(THREADS), 0, 0, d_ops, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < settle_s * 1e3);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < timed_launches; ++i) hipLaunchKernelGGL((peak_kernel<BF16, THREADS>), dim3(n_cu), dim3(THREADS), 0, 0, d_ops, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    const double per_launch_s = ms * 1e-3 / timed_launches;
    const double flops = 2.0 * 32 * 32 * 16 * (double)kMfmaPerSimd * 4 * n_cu;
    *out_ms = per_launch_s * 1e3;
    return flops / per_launch_s / 1e12;
}

// Same loop with v_mfma_f32_16x16x32_f16 (half the FLOPs per instruction, 16 instead of 32 pipe cycles; per FLOP it reads twice the
// A/B operand bytes and moves half the accumulator bytes of the 32x32x16 form): is the other MFMA shape cheaper in energy?
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void peak_kernel_16(const u32x4* __restrict__ operands, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(i * 2 + 0) * 512 + tid];
        b[i] = operands[(i * 2 + 1) * 512 + tid];
    }
    f32x4 acc[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + r) & 3]), __builtin_bit_cast(f16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * THREADS + tid] = s;
}

template <int THREADS>
static double run16(const u32x4* d_ops, float* d_out, int n_cu, double settle_s, int timed_launches, double* out_ms) {
    const int waves_per_simd = THREADS / 256;
    const int iters = 2 * kMfmaPerSimd / waves_per_simd / 32;   // same FLOPs per launch as run<>
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    float ms = 0.f;
    do {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((peak_kernel_16<THREADS>), dim3(n_cu), dim3(THREADS), 0, 0, d_ops, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < settle_s * 1e3);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < timed_launches; ++i) hipLaunchKernelGGL((peak_kernel_16<THREADS>), dim3(n_cu), dim3(THREADS), 0, 0, d_ops, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    const double per_launch_s = ms * 1e-3 / timed_launches;
    const double flops = 2.0 * 32 * 32 * 16 * (double)kMfmaPerSimd * 4 * n_cu;
    *out_ms = per_launch_s * 1e3;
    return flops / per_launch_s / 1e12;
}

// Operand-sharing probe (32x32x16 f16, U[0,1) operands): does the order in which the MFMAs of a tile are issued matter for power?
//   ORDER 0: consecutive MFMAs differ in A and in B (the loop of peak_kernel)
//   ORDER 1: runs of four MFMAs share B (the attention kernel's P.V order: one P fragment against four V^T fragments)
//   ORDER 2: runs of four MFMAs share A
//   ORDER 3: all MFMAs use the same A and the same B
template <int ORDER, int THREADS>
__global__ __launch_bounds__(THREADS) void order_kernel(const u32x4* __restrict__ operands, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(i * 2 + 0) * 512 + tid];
        b[i] = operands[(i * 2 + 1) * 512 + tid];
    }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 aa = ORDER == 0 ? a[(i + r) & 3] : ORDER == 1 ? a[i] : ORDER == 2 ? a[r] : a[0];
                const u32x4 bb = ORDER == 0 ? b[i] : ORDER == 1 ? b[r] : ORDER == 2 ? b[i] : b[0];
                acc[i] = mfma<false>(aa, bb, acc[i]);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * THREADS + tid] = s;
}

template <int ORDER>
static double run_order(const u32x4* d_ops, float* d_out, int n_cu, double settle_s, int timed_launches) {
    const int iters = kMfmaPerSimd / 2 / 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    float ms = 0.f;
    do {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((order_kernel<ORDER, 512>), dim3(n_cu), dim3(512), 0, 0, d_ops, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < settle_s * 1e3);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < timed_launches; ++i) hipLaunchKernelGGL((order_kernel<ORDER, 512>), dim3(n_cu), dim3(512), 0, 0, d_ops, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    const double per_launch_s = ms * 1e-3 / timed_launches;
    return 2.0 * 32 * 32 * 16 * (double)kMfmaPerSimd * 4 * n_cu / per_launch_s / 1e12;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<unsigned short> h_f16(8 * 512 * 8), h_bf16(8 * 512 * 8), h_u01(8 * 512 * 8);
    srand(1234);
    for (size_t i = 0; i < h_f16.size(); ++i) {
        const float x = 2.f * (float)rand() / (float)RAND_MAX - 1.f;
        h_f16[i] = f32_to_f16_bits(x);
        h_bf16[i] = f32_to_bf16_bits(x);
        h_u01[i] = f32_to_f16_bits(0.5f * (x + 1.f));     // U[0,1): the distribution of bench.py's q/k/v (torch.rand)
    }
    u32x4 *d_f16, *d_bf16, *d_zero, *d_u01;
    float* d_out;
    const size_t bytes = h_f16.size() * 2;
    CHECK(hipMalloc(&d_f16, bytes));
    CHECK(hipMalloc(&d_bf16, bytes));
    CHECK(hipMalloc(&d_zero, bytes));
    CHECK(hipMalloc(&d_u01, bytes));
    CHECK(hipMemcpy(d_u01, h_u01.data(), bytes, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 512 * 4));
    CHECK(hipMemcpy(d_f16, h_f16.data(), bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bf16, h_bf16.data(), bytes, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_zero, 0, bytes));

    double ms;
    const double settle = 0.4;
    const int timed = 300;
    const double f16_w2 = run<false, 512>(d_f16, d_out, n_cu, settle, timed, &ms);   const double f16_w2_ms = ms;
    const double f16_w1 = run<false, 256>(d_f16, d_out, n_cu, settle, timed, &ms);   const double f16_w1_ms = ms;
    const double bf16_w2 = run<true, 512>(d_bf16, d_out, n_cu, settle, timed, &ms);  const double bf16_w2_ms = ms;
    const double bf16_w1 = run<true, 256>(d_bf16, d_out, n_cu, settle, timed, &ms);  const double bf16_w1_ms = ms;
    const double u01_w2 = run<false, 512>(d_u01, d_out, n_cu, settle, timed, &ms);   const double u01_w2_ms = ms;
    const double u01_w1 = run<false, 256>(d_u01, d_out, n_cu, settle, timed, &ms);   const double u01_w1_ms = ms;
    const double s16_w2 = run16<512>(d_u01, d_out, n_cu, settle, timed, &ms);        const double s16_w2_ms = ms;
    const double s16_w1 = run16<256>(d_u01, d_out, n_cu, settle, timed, &ms);        const double s16_w1_ms = ms;
    const double u01_again = run<false, 512>(d_u01, d_out, n_cu, settle, timed, &ms); const double u01_again_ms = ms;
    const double ord0 = run_order<0>(d_u01, d_out, n_cu, settle, timed), ord1 = run_order<1>(d_u01, d_out, n_cu, settle, timed);
    const double ord2 = run_order<2>(d_u01, d_out, n_cu, settle, timed), ord3 = run_order<3>(d_u01, d_out, n_cu, settle, timed);
    const double ord0b = run_order<0>(d_u01, d_out, n_cu, settle, timed);
    const double zero_w2 = run<false, 512>(d_zero, d_out, n_cu, settle, timed, &ms); const double zero_w2_ms = ms;
    // cold: 20 launches straight after >= 1 s of idle (what a short driver-run benchmark sees)
    CHECK(hipDeviceSynchronize());
    struct timespec ts = {1, 200000000};
    nanosleep(&ts, nullptr);
    const double cold = run<false, 512>(d_f16, d_out, n_cu, 0.0, 20, &ms);           const double cold_ms = ms;

    const double cyc = 32.0 * kMfmaPerSimd;   // matrix-pipe cycles per SIMD per launch at 100 % issue
    printf("{\n");
    printf(" \"_comment\": \"tools/ubench/mfma_peak.hip on %s (%d CUs): v_mfma_f32_32x32x16 only, 4 accumulators per wave, %d MFMAs per SIMD per launch, random uniform [-1,1) operands unless stated, %d timed launches after %.1f s of settling\",\n",
           prop.gcnArchName, n_cu, kMfmaPerSimd, timed, settle);
    printf(" \"spec_peak_tflops\": 2500.0,\n");
    printf(" \"f16_2waves_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", f16_w2, f16_w2_ms, cyc / (f16_w2_ms * 1e-3) / 1e9);
    printf(" \"f16_1wave_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", f16_w1, f16_w1_ms, cyc / (f16_w1_ms * 1e-3) / 1e9);
    printf(" \"bf16_2waves_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", bf16_w2, bf16_w2_ms, cyc / (bf16_w2_ms * 1e-3) / 1e9);
    printf(" \"bf16_1wave_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", bf16_w1, bf16_w1_ms, cyc / (bf16_w1_ms * 1e-3) / 1e9);
    printf(" \"f16_uniform01_2waves_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", u01_w2, u01_w2_ms, cyc / (u01_w2_ms * 1e-3) / 1e9);
    printf(" \"f16_uniform01_1wave_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", u01_w1, u01_w1_ms, cyc / (u01_w1_ms * 1e-3) / 1e9);
    printf(" \"f16_uniform01_16x16x32_2waves_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f},\n", s16_w2, s16_w2_ms);
    printf(" \"f16_uniform01_16x16x32_1wave_per_simd\": {\"tflops\": %.1f, \"launch_ms\": %.5f},\n", s16_w1, s16_w1_ms);
    printf(" \"f16_uniform01_2waves_per_simd_repeat\": {\"tflops\": %.1f, \"launch_ms\": %.5f},\n", u01_again, u01_again_ms);
    printf(" \"f16_uniform01_operand_order_tflops\": {\"a_and_b_change_every_mfma\": %.1f, \"b_shared_by_runs_of_4\": %.1f, \"a_shared_by_runs_of_4\": %.1f, \"same_a_and_b_always\": %.1f, \"a_and_b_change_repeat\": %.1f},\n", ord0, ord1, ord2, ord3, ord0b);
    printf(" \"f16_zero_operands_2waves\": {\"tflops\": %.1f, \"launch_ms\": %.5f, \"implied_clock_ghz\": %.3f},\n", zero_w2, zero_w2_ms, cyc / (zero_w2_ms * 1e-3) / 1e9);
    printf(" \"f16_cold_20_launches_after_idle\": {\"tflops\": %.1f, \"launch_ms\": %.5f},\n", cold, cold_ms);
    printf(" \"sustained_tflops_f16_signed\": %.1f,\n", f16_w2 > f16_w1 ? f16_w2 : f16_w1);
    printf(" \"sustained_tflops_f16\": %.1f,\n", u01_w2 > u01_w1 ? u01_w2 : u01_w1);   // operands distributed like bench.py's inputs
    printf(" \"sustained_tflops_bf16\": %.1f\n", bf16_w2 > bf16_w1 ? bf16_w2 : bf16_w1);
    printf("}\n");
    return 0;
}
