#!/usr/bin/env python3
"""Which MFMA shape should the forward body use on a POWER-LIMITED chip?  (developer probe, round 5)

profiles/mfma_peak.json: a loop of nothing but v_mfma_f32_16x16x32_f16 sustains 1 969 TF on U[0,1) operands with two waves per SIMD where the
32x32x16 form sustains 1 738 — the smaller tile moves half the accumulator bytes per FLOP — but only 1 531 with ONE wave per SIMD (a lone wave cannot
issue a 16-cycle MFMA every 16 cycles).  The hand-scheduled forward is one wave per SIMD and is not MFMA-issue-bound but energy-bound (DESIGN
section 3), so the question is what its BODY would do with the other shape: the same FLOPs as 128 MFMAs of 16 pipe cycles instead of 64 of 32, the
same single-issue fillers (64 v_exp_f32, 64 v_add_f32, 32 v_cvt_pk, 16 ds_read_b128, 32 ds_read_b64_tr_b16 per 64 x 64 tile) spread over twice the
gaps.  This script writes two synthetic kernels — a tile body of each shape in one inline-asm loop, accumulator chains like the real body (Q.K^T: short
chains that restart from C = 0; P.V: long-lived accumulators), static U[0,1) fp16 operands, fillers on scratch registers — compiles them and, on the
GPU, times them interleaved.  Results are not attention; cycle counts, clocks and energy are the point.

    python tools/ubench/mfma_shape_probe.py build        (CPU container: writes mfma_shape_probe.hip and compiles tools/ubench/mfma_shape_probe)
    tools/ubench/mfma_shape_probe                         (GPU box: prints one JSON line)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.realpath(__file__))

FILL = {"exp": 64, "add": 64, "cvt": 32, "kread": 16, "vread": 32}
# the dQ pass of the hand-scheduled backward (csrc/gen/bwd_d128_gen.py): per body 48 MFMAs of 32x32x16 (dQ 16, then S and dP 32) and, per wave, 144 plain
# VALU instructions (fma, sub, mul, pack), 32 exp, 16 row-fragment + 16 transposed reads: would the other MFMA tile pay there too?
FILL_BWD = {"exp": 32, "add": 144, "cvt": 0, "kread": 16, "vread": 16}


def fillers(nofill=(), fill=None):
    """The body's single-issue work as a list of instruction strings, interleaved the way the generator's water-filling leaves them: evenly."""
    out = []
    streams = []
    for kind, n in (fill or FILL).items():
        if kind in nofill:
            continue
        lst = []
        for j in range(n):
            if kind == "exp":
                lst.append("v_exp_f32 v%d, v%d" % (200 + j % 8, 208 + j % 8))
            elif kind == "add":
                lst.append("v_add_f32 v%d, v%d, v%d" % (216 + j % 2, 216 + j % 2, 200 + (j * 3) % 8))
            elif kind == "cvt":
                lst.append("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (220 + j % 4, 200 + (2 * j) % 8, 200 + (2 * j + 1) % 8))
            elif kind == "cvtrtz":      # the pack with round-toward-zero / by byte permute (bf16 truncation): cheaper than the RNE pack?
                lst.append("v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (220 + j % 4, 200 + (2 * j) % 8, 200 + (2 * j + 1) % 8))
            elif kind == "perm":
                lst.append("v_perm_b32 v%d, v%d, v%d, v183" % (220 + j % 4, 200 + (2 * j) % 8, 200 + (2 * j + 1) % 8))
            elif kind == "cvtbf":
                lst.append("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (220 + j % 4, 200 + (2 * j) % 8, 200 + (2 * j + 1) % 8))
            elif kind == "kread":
                lst.append("ds_read_b128 v[%d:%d], %%3 offset:%d" % (224 + 4 * (j % 4), 227 + 4 * (j % 4), 1024 * (j % 16)))
            else:
                lst.append("ds_read_b64_tr_b16 v[%d:%d], %%4 offset:%d" % (240 + 2 * (j % 4), 241 + 2 * (j % 4), 16384 + 512 * (j % 32)))
        streams.append(lst)
    total = sum(len(s) for s in streams)
    pos = []
    for s in streams:
        for j, ins in enumerate(s):
            pos.append(((j + 0.5) / len(s), ins))
    pos.sort(key=lambda x: x[0])
    return [ins for _, ins in pos], total


# row sums on the matrix pipe instead of 64 v_add_f32 per tile (round 5 probe): "sum16" = one more 16x16x32 MFMA per (q group, k-step) with a constant
# A operand (ones in the rows m % 4 == q group: one 4-register accumulator collects all four q groups); "sum4" = v_mfma_f32_4x4x4_16b_f16 with a ones
# row as A: every lane's four packed P values land, summed, in one accumulator register (per-lane partial sums, as the v_add chain leaves them)
def sum_mfmas(kind):
    if kind == "sum16h":      # (32 Q rows per wave: two q groups x two k-steps)
        return ["v_mfma_f32_16x16x32_f16 a[192:195], v[%d:%d], v[%d:%d], a[192:195]" % (184 + 4 * (j % 4), 187 + 4 * (j % 4), 48 + 4 * (j % 8), 51 + 4 * (j % 8))
                for j in range(4)]
    if kind == "sum16":
        return ["v_mfma_f32_16x16x32_f16 a[192:195], v[%d:%d], v[%d:%d], a[192:195]" % (184 + 4 * (j % 4), 187 + 4 * (j % 4), 48 + 4 * (j % 8), 51 + 4 * (j % 8))
                for j in range(8)]
    if kind == "sum4":
        return ["v_mfma_f32_4x4x4_16b_f16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (192 + 4 * (j % 2), 195 + 4 * (j % 2), 184 + 2 * (j % 4), 185 + 2 * (j % 4),
                                                                                    48 + 2 * (j % 16), 49 + 2 * (j % 16), 192 + 4 * (j % 2), 195 + 4 * (j % 2)) for j in range(16)]
    return []


def body(shape, nofill=(), fill=None, pv_steps=None, extra=None, hd=128):
    """One tile body: MFMAs with the fillers spread evenly over the gaps.  pv_steps: k-steps of the P.V-like product (default: the forward's)."""
    fl, _ = fillers(nofill, fill)
    if extra in ("dot2", "dot2c"):          # the row sums from the PACKED 16-bit P: one v_dot2 with a ones operand per pair
        if extra == "dot2":
            add = ["v_dot2_f32_f16 v%d, v%d, v184, v%d" % (216 + j % 2, 220 + j % 4, 216 + j % 2) for j in range(32)]
        else:
            add = ["v_dot2c_f32_f16 v%d, v%d, v184" % (216 + j % 2, 220 + j % 4) for j in range(32)]
        fl = fl + add
        pos = sorted(range(len(fl)), key=lambda i: ((i + 0.5) / (len(fl) - 32) if i < len(fl) - 32 else (i - (len(fl) - 32) + 0.5) / 32))
        fl = [fl[i] for i in pos]
    if extra == "pkadd":
        fl = fl + ["v_pk_add_f32 v[216:217], v[216:217], v[%d:%d]" % (200 + 2 * (j % 4), 201 + 2 * (j % 4)) for j in range(32)]
        pos = sorted(range(len(fl)), key=lambda i: ((i + 0.5) / (len(fl) - 32) if i < len(fl) - 32 else (i - (len(fl) - 32) + 0.5) / 32))
        fl = [fl[i] for i in pos]
    mf = []
    if shape == 32:
        # P.V-like: 8 accumulators x 4 k-steps (a[64:191]); Q.K^T-like: 4 accumulators x 8 k-steps, the first from C = 0 (a[0:63])
        for ks in range(4 if pv_steps is None else pv_steps):
            for acc in range(8 if hd == 128 else 4):
                a0 = 64 + 16 * acc
                mf.append("v_mfma_f32_32x32x16_f16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (a0, a0 + 15, 16 + 4 * ((acc + ks) % 8), 19 + 4 * ((acc + ks) % 8),
                                                                                        48 + 4 * ((2 * ks + acc // 4) % 8), 51 + 4 * ((2 * ks + acc // 4) % 8), a0, a0 + 15))
        for ks in range(8 if hd == 128 else 4):
            for acc in range(4):
                a0 = 16 * acc
                c = "0" if ks == 0 else "a[%d:%d]" % (a0, a0 + 15)
                mf.append("v_mfma_f32_32x32x16_f16 a[%d:%d], v[%d:%d], v[%d:%d], %s" % (a0, a0 + 15, 16 + 4 * ((ks + acc) % 8), 19 + 4 * ((ks + acc) % 8),
                                                                                  48 + 4 * ((ks + 2 * acc) % 8), 51 + 4 * ((ks + 2 * acc) % 8), c))
    else:
        # the same products as 16x16x32 tiles: P.V-like: 32 accumulators (4 registers) x 2 k-steps; Q.K^T-like: 16 accumulators x 4 k-steps
        for ks in range(2 if pv_steps is None else max(1, pv_steps // 2)):
            for acc in range(32 if hd == 128 else 16):
                a0 = 64 + 4 * acc
                mf.append("v_mfma_f32_16x16x32_f16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (a0, a0 + 3, 16 + 4 * ((acc // 4 + ks) % 8), 19 + 4 * ((acc // 4 + ks) % 8),
                                                                                        48 + 4 * ((acc + 4 * ks) % 8), 51 + 4 * ((acc + 4 * ks) % 8), a0, a0 + 3))
        for ks in range(4 if hd == 128 else 2):
            for acc in range(16):
                a0 = 4 * acc
                c = "0" if ks == 0 else "a[%d:%d]" % (a0, a0 + 3)
                mf.append("v_mfma_f32_16x16x32_f16 a[%d:%d], v[%d:%d], v[%d:%d], %s" % (a0, a0 + 3, 16 + 4 * ((acc // 4 + ks) % 8), 19 + 4 * ((acc // 4 + ks) % 8),
                                                                                  48 + 4 * ((acc + ks) % 8), 51 + 4 * ((acc + ks) % 8), c))
    ex = sum_mfmas(extra)
    if ex:          # spread over the P.V-like phase (the first len(mf) / 2 MFMAs)
        half = len(mf) // 2
        step = half // len(ex)
        for j, ins in enumerate(reversed(ex)):
            mf.insert(half - j * step, ins)
    n = len(mf)
    lines = []
    fi = 0
    for g in range(n):
        lines.append(mf[g])
        want = (g + 1) * len(fl) // n
        while fi < want:
            lines.append(fl[fi])
            fi += 1
    lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def kernel(name, shape, nofill=(), fill=None, pv_steps=None, extra=None, hd=128):
    # operands: %0 = result (out), %1 = iters (s), %2 = operand pointer (s, 64 bit), %3 / %4 = LDS read addresses (v), %5 = this thread's byte offset (v)
    lines = ["s_mov_b32 s60, %1", "v_mov_b32 v250, %5"]
    for i in range(16):          # 8 A + 8 B operand quads: U[0,1) fp16 data
        lines.append("global_load_dwordx4 v[%d:%d], v250, %%2" % (16 + 4 * i, 19 + 4 * i))
        lines.append("v_add_u32 v250, 0x1000, v250")
    for i in range(8):
        lines.append("v_mov_b32 v%d, 0xbf000000" % (208 + i))        # exp sources: -0.5
        lines.append("v_mov_b32 v%d, 0" % (200 + i))
    lines += ["v_mov_b32 v216, 0", "v_mov_b32 v217, 0", "v_mov_b32 v183, 0x07060302"]
    for i in range(16):
        lines.append("v_mov_b32 v%d, 0x3c003c00" % (184 + i))       # the ones operand of the row-sum MFMAs
    for i in range(200):
        lines.append("v_accvgpr_write_b32 a%d, 0" % i)
    lines.append("s_waitcnt vmcnt(0)")
    lines.append(".Lprobe_%s_%%=:" % name)
    lines += body(shape, nofill, fill, pv_steps, extra, hd)
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_gt_i32 s60, 0", "s_cbranch_scc1 .Lprobe_%s_%%=" % name]
    lines.append("s_nop 7")
    lines.append("v_accvgpr_read_b32 %0, a64")     # keep something observable
    text = "\n".join('        "%s\\n"' % l for l in lines)
    clob = ", ".join('"v%d"' % i for i in range(16, 256)) + ", " + ", ".join('"a%d"' % i for i in range(256)) + ', "s60", "vcc", "scc", "memory"'
    return """
__global__ __launch_bounds__(256, 1) void %s(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 32768 / 16; i += 256) ((u32x4*)smem)[i] = ops[i & 1023];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t kaddr = lane * 16, vaddr = (lane & 15) * 8 + (lane >> 4) * 128, goff = threadIdx.x * 16;
    float r;
    asm volatile(
%s
        : "=v"(r) : "s"(iters), "s"(ops), "v"(kaddr), "v"(vaddr), "v"(goff)
        : %s);
    if (r == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = r;
}
""" % (name, text, clob)


def kernel_w2(name, sums=True):
    """Round 6: the lm body with TWO waves per SIMD — 8-wave workgroups, 32 Q rows and a 256-register budget per wave.  Per wave and
    tile: 32 P.V-like + 32 Q.K^T-like MFMAs of 16x16x32 (+ 4 row-sum MFMAs), 32 exp, 16 pack — half the 4-wave body's — but ALL of the tile's fragment
    reads (16 ds_read_b128 + 32 ds_read_b64_tr_b16: every wave needs the whole K and V tile, so the CU reads each twice as often per FLOP).  128 bodies per
    wave and launch: the same FLOPs per launch as the 4-wave variants.  Registers: operands v16..79, fillers v80..127, accumulators a0..127."""
    fl = []
    streams = [["v_exp_f32 v%d, v%d" % (80 + j % 8, 88 + j % 8) for j in range(32)],
               ["v_cvt_pk_f16_f32 v%d, v%d, v%d" % (100 + j % 4, 80 + (2 * j) % 8, 80 + (2 * j + 1) % 8) for j in range(16)],
               ["ds_read_b128 v[%d:%d], %%3 offset:%d" % (104 + 4 * (j % 4), 107 + 4 * (j % 4), 1024 * (j % 16)) for j in range(16)],
               ["ds_read_b64_tr_b16 v[%d:%d], %%4 offset:%d" % (120 + 2 * (j % 4), 121 + 2 * (j % 4), 16384 + 512 * (j % 32)) for j in range(32)]]
    pos = []
    for st in streams:
        for j, ins in enumerate(st):
            pos.append(((j + 0.5) / len(st), ins))
    pos.sort(key=lambda x: x[0])
    fl = [ins for _, ins in pos]
    mf = []
    for ks in range(2):                      # P.V-like: 16 accumulators (a64..127) x 2 k-steps
        for acc in range(16):
            a0 = 64 + 4 * acc
            mf.append("v_mfma_f32_16x16x32_f16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (a0, a0 + 3, 16 + 4 * ((acc // 4 + ks) % 8), 19 + 4 * ((acc // 4 + ks) % 8),
                                                                                    48 + 4 * ((acc + 4 * ks) % 8), 51 + 4 * ((acc + 4 * ks) % 8), a0, a0 + 3))
    if sums:                                 # 4 row-sum MFMAs (2 q groups x 2 k-steps), spread over the P.V-like phase
        for j in range(4):
            mf.insert(8 * j + 7 + j, "v_mfma_f32_16x16x32_f16 a[32:35], v[16:19], v[%d:%d], a[32:35]" % (48 + 4 * (j % 8), 51 + 4 * (j % 8)))
    for ks in range(4):                      # Q.K^T-like: 8 accumulators (a0..31) x 4 k-steps, the first from C = 0
        for acc in range(8):
            a0 = 4 * acc
            c = "0" if ks == 0 else "a[%d:%d]" % (a0, a0 + 3)
            mf.append("v_mfma_f32_16x16x32_f16 a[%d:%d], v[%d:%d], v[%d:%d], %s" % (a0, a0 + 3, 16 + 4 * ((acc // 4 + ks) % 8), 19 + 4 * ((acc // 4 + ks) % 8),
                                                                              48 + 4 * ((acc + ks) % 8), 51 + 4 * ((acc + ks) % 8), c))
    n = len(mf)
    body_ = []
    fi = 0
    for g in range(n):
        body_.append(mf[g])
        want = (g + 1) * len(fl) // n
        while fi < want:
            body_.append(fl[fi])
            fi += 1
    body_.append("s_waitcnt lgkmcnt(0)")
    lines = ["s_mov_b32 s60, %1", "v_mov_b32 v126, %5"]
    for i in range(16):
        lines.append("global_load_dwordx4 v[%d:%d], v126, %%2" % (16 + 4 * i, 19 + 4 * i))
        lines.append("v_add_u32 v126, 0x1000, v126")
    for i in range(8):
        lines.append("v_mov_b32 v%d, 0xbf000000" % (88 + i))
        lines.append("v_mov_b32 v%d, 0" % (80 + i))
    for i in range(128):
        lines.append("v_accvgpr_write_b32 a%d, 0" % i)
    lines.append("s_waitcnt vmcnt(0)")
    lines.append(".Lprobe_%s_%%=:" % name)
    lines += body_
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_gt_i32 s60, 0", "s_cbranch_scc1 .Lprobe_%s_%%=" % name]
    lines.append("s_nop 7")
    lines.append("v_accvgpr_read_b32 %0, a64")
    text = "\n".join('        "%s\\n"' % l for l in lines)
    clob = ", ".join('"v%d"' % i for i in range(16, 128)) + ", " + ", ".join('"a%d"' % i for i in range(128)) + ', "s60", "vcc", "scc", "memory"'
    return """
__global__ __launch_bounds__(512, 1) void %s(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 32768 / 16; i += 512) ((u32x4*)smem)[i] = ops[i & 1023];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t kaddr = lane * 16, vaddr = (lane & 15) * 8 + (lane >> 4) * 128, goff = (threadIdx.x & 255) * 16;
    float r;
    asm volatile(
%s
        : "=v"(r) : "s"(iters), "s"(ops), "v"(kaddr), "v"(vaddr), "v"(goff)
        : %s);
    if (r == 12345.678f) out[blockIdx.x * 256 + (threadIdx.x & 255)] = r;
}
""" % (name, text, clob)


HOST = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
KERNELS
typedef void (*kern_t)(const u32x4*, float*, int);
int main() {
    int dev = 0, n_cu = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    std::vector<unsigned short> h(16 * 1024 * 8);
    srand(1);
    for (auto& x : h) { _Float16 v = (_Float16)((float)rand() / RAND_MAX); memcpy(&x, &v, 2); }
    u32x4* d_ops; float* d_out;
    CHECK(hipMalloc(&d_ops, h.size() * 2));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 256 * 4));
    CHECK(hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 128;      // tile bodies per workgroup and launch (config 2: 2 items x 64 tiles)
    struct K { const char* name; kern_t fn; int threads; } ks[] = {NAMES};
    const int nk = sizeof(ks) / sizeof(ks[0]);
    for (int i = 0; i < nk; ++i) CHECK(hipFuncSetAttribute((const void*)ks[i].fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    CHECK(hipEventRecord(e0));
    do {       // settle the clock
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(ks[0].fn, dim3(n_cu), dim3(ks[0].threads), 65536, 0, d_ops, d_out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < 400.f);
    std::vector<std::vector<double>> t(nk);
    for (int round = 0; round < 9; ++round)
        for (int i = 0; i < nk; ++i) {
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(ks[i].fn, dim3(n_cu), dim3(ks[i].threads), 65536, 0, d_ops, d_out, iters);
            CHECK(hipEventRecord(e0));
            for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(ks[i].fn, dim3(n_cu), dim3(ks[i].threads), 65536, 0, d_ops, d_out, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipGetLastError());
            t[i].push_back(ms / 100.0 * 1e3);
        }
    const double flops = 64.0 * 2 * 32 * 32 * 16 * 4 * n_cu * iters;
    printf("{\"_comment\": \"tools/ubench/mfma_shape_probe.py: synthetic forward tile bodies, one wave per SIMD, %d CUs, %d bodies per launch, U[0,1) fp16 operands, interleaved rounds; us per launch (median of 9 x 100 launches), TF at the body's FLOPs\"", n_cu, iters);
    for (int i = 0; i < nk; ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf(", \"%s\": {\"us\": %.1f, \"tflops\": %.0f, \"best_us\": %.1f}", ks[i].name, med, flops / (med * 1e-6) / 1e12, t[i][0]);
    }
    printf("}\n");
    return 0;
}
"""


def main():
    variants = [("body_32x32x16", 32, (), None, None), ("body_16x16x32", 16, (), None, None), ("mfma_only_32x32x16", 32, tuple(FILL), None, None),
                ("mfma_only_16x16x32", 16, tuple(FILL), None, None), ("no_lds_32x32x16", 32, ("kread", "vread"), None, None),
                ("no_lds_16x16x32", 16, ("kread", "vread"), None, None),
                # a dQ-pass-like body: 48 (96) MFMAs, the backward's filler mix (its FLOPs are 0.75 of the forward body's: the TF printed for it are 4/3 too high)
                ("bwd_dq_like_32x32x16", 32, (), FILL_BWD, 2), ("bwd_dq_like_16x16x32", 16, (), FILL_BWD, 2)]
    variants = [v + (None,) for v in variants]
    # row sums on the matrix pipe / packed adds instead of the 64 v_add_f32; "noadd": the adds simply gone (the bound of what removing them can return)
    variants += [("body16_noadd", 16, ("add",), None, None, None), ("body16_sum16", 16, ("add",), None, None, "sum16"), ("body16_sum4", 16, ("add",), None, None, "sum4"),
                 ("body16_pkadd", 16, ("add",), None, None, "pkadd"), ("body16_dot2", 16, ("add",), None, None, "dot2"), ("body16_dot2c", 16, ("add",), None, None, "dot2c"),
                 ("body32_dot2", 32, ("add",), None, None, "dot2"), ("body16_nocvt", 16, ("cvt",), None, None, None), ("body16_noexp", 16, ("exp",), None, None, None), ("body32_noadd", 32, ("add",), None, None, None)]
    variants = [v + (128,) for v in variants]
    # head dim 64 (half the MFMAs and half the LDS reads for the same softmax work; its TF are printed at the 128 body's FLOPs: halve them)
    FILL64 = {"exp": 64, "add": 64, "cvt": 32, "kread": 8, "vread": 16}
    variants += [("d64_body32", 32, (), FILL64, None, None, 64), ("d64_body16", 16, (), FILL64, None, None, 64),
                 ("d64_body16_sum16", 16, ("add",), FILL64, None, "sum16", 64), ("d64_body32_noadd", 32, ("add",), FILL64, None, None, 64)]
    # the lm body (no adds, 8 sum MFMAs) with other ways to pack P
    LM = {"exp": 64, "cvt": 32, "kread": 16, "vread": 32}
    variants += [("lm_cvt", 16, (), LM, None, "sum16", 128), ("lm_cvtrtz", 16, (), dict(LM, cvt=0, cvtrtz=32), None, "sum16", 128),
                 ("lm_perm", 16, (), dict(LM, cvt=0, perm=32), None, "sum16", 128), ("lm_cvtbf", 16, (), dict(LM, cvt=0, cvtbf=32), None, "sum16", 128),
                 ("lm_nocvt", 16, (), dict(LM, cvt=0), None, "sum16", 128)]
    # round 6: the lm body without its 8 row-sum MFMAs (what dropping them could return at most), and the same work as two waves per SIMD (kernel_w2)
    variants += [("lm_nosum", 16, (), LM, None, None, 128)]
    # head dim 256 on the same structure: one wave per SIMD can keep O for 32 Q rows only (128 accumulator registers + 64 of Q
    # fragments), so a wave-tile is 32 rows x 64 keys: the same FLOPs as the head-dim-128 body's 64 x 64, half the exp / pack work, but TWICE the
    # fragment reads — a K or V^T fragment feeds two MFMAs instead of four (32 ds_read_b128 + 64 ds_read_b64_tr_b16 per wave-tile)
    D256 = {"exp": 32, "cvt": 16, "kread": 32, "vread": 64}
    variants += [("d256_rows32", 16, (), D256, None, "sum16h", 128), ("d256_rows32_nolds", 16, ("kread", "vread"), D256, None, "sum16h", 128)]
    w2 = [("w2_lm", True), ("w2_nosum", False)]
    if len(sys.argv) > 2:
        variants = [v for v in variants if v[0] in sys.argv[2].split(",")]
        w2 = [v for v in w2 if v[0] in sys.argv[2].split(",")]
    kernels = [kernel(n, s, nf, fl, pv, ex, hd) for n, s, nf, fl, pv, ex, hd in variants] + [kernel_w2(n, sm) for n, sm in w2]
    names = ['{"%s", %s, 256}' % (v[0], v[0]) for v in variants] + ['{"%s", %s, 512}' % (n, n) for n, _ in w2]
    src = HOST.replace("KERNELS", "\n".join(kernels)).replace("NAMES", ", ".join(names))
    path = os.path.join(HERE, "mfma_shape_probe.hip")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/ubench/mfma_shape_probe.py — do not edit.\n" + src)
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", path, "-o", os.path.join(HERE, "mfma_shape_probe")]
        print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    main()
