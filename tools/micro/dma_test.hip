// Standalone check of the LDS-DMA primitive used by the streaming GEMM: global_load_lds_dwordx4 with M0 as the
// wave-uniform LDS base (also above 64 KB), per-lane source swizzle, hand-counted vmcnt, raw barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// each block copies `rows` 128-byte rows into LDS at byte offset `base` with the chunk swizzle c ^ ((row>>1)&7),
// then every thread reads the logical layout back and writes it out.
__global__ void __launch_bounds__(256) k(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int rows, int base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)(smem + base);
    // wave w copies rows [w*rows/4, (w+1)*rows/4), 8 rows per instruction
    const int rpw = rows / 4;
    for (int r0 = wave * rpw; r0 < (wave + 1) * rpw; r0 += 8) {
        const int row = r0 + (lane >> 3), phys = lane & 7;
        const int logical = phys ^ ((row >> 1) & 7);
        const u32x4_t* g = src + (size_t)row * 8 + logical;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)r0 * 128u);
        dma16(g, ldst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < rows * 8; i += 256) {
        const int row = i >> 3, c = i & 7;
        dst[i] = *(const u32x4_t*)(smem + base + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    }
}

int main() {
    const int rows = 128;
    std::vector<unsigned> h(rows * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u + 12345u);
    unsigned *d_src, *d_dst;
    hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_dst, h.size() * 4);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int base : {0, 16384, 65536, 98304, 131072}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(d_dst, 0, h.size() * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), base + rows * 128, 0, (const u32x4_t*)d_src, (u32x4_t*)d_dst, rows, base);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> o(h.size());
        hipMemcpy(o.data(), d_dst, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < h.size(); ++i) bad += o[i] != h[i];
        printf("base %6d: err=%d mismatches=%d\n", base, (int)e, bad);
        bad_total += bad;
    }
    printf(bad_total ? "DMA TEST FAILED\n" : "DMA TEST OK\n");
    return bad_total != 0;
}
