// Can an LDS-DMA (buffer_load_dword ... lds) land above 64 KiB of a workgroup's LDS?  (dev probe: M0 carries the destination base)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_hi.hip -o tools/ubench/lds_dma_hi && tools/ubench/lds_dma_hi
// One wave copies 256 B of a global buffer into LDS at byte address `dst` through M0 and reports what the LDS holds at `dst` and at `dst & 0xffff`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k(const uint32_t* src, uint32_t* out, uint32_t dst, uint32_t nbytes_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (uint32_t i = threadIdx.x * 4; i < nbytes_lds; i += 256) *(volatile uint32_t*)(smem + i) = 0xdead0000u;
    __syncthreads();
    const uint64_t a = (uint64_t)src;
    const u32x4 rs = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, 4096u, 0x00020000u};
    const uint32_t off = threadIdx.x * 4;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"v"(off), "s"(rs), "s"(dst) : "memory", "m0");
    __syncthreads();
    out[threadIdx.x] = *(volatile uint32_t*)(smem + dst + threadIdx.x * 4);
    out[64 + threadIdx.x] = *(volatile uint32_t*)(smem + (dst & 0xffffu) + threadIdx.x * 4);
}

int main() {
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    uint32_t *src, *out;
    hipMalloc(&src, 4096);
    hipMalloc(&out, 512);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const uint32_t lds = 160 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (uint32_t dst : {4096u, 65536u - 256u, 65536u, 65536u + 4096u, 100000u & ~255u, 159u * 1024u}) {
        hipMemset(out, 0, 512);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, src, out, dst, lds);
        uint32_t r[128];
        hipMemcpy(r, out, 512, hipMemcpyDeviceToHost);
        printf("dst %6u: LDS[dst] = %08x %08x ... %s | LDS[dst & 0xffff] = %08x ... %s   (%s)\n", dst, r[0], r[1], r[0] == 0x1000u ? "DATA" : "untouched", r[64],
               r[64] == 0x1000u ? "DATA" : "untouched", hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
