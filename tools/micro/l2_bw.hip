// What can one CU pull through its vector-memory path?  Every wave streams 1 KB (64 lanes x 16 B) loads, UNROLL of
// them in flight, over a window of `window` bytes that is (a) L1-resident, (b) L2-resident and shared by all blocks
// (the weight re-read pattern of the conv GEMMs), (c) far larger than every cache (HBM streaming).
// Prints GB/s chip-wide and B/clk/CU at the measured duration.   hipcc --offload-arch=gfx950 -O3 l2_bw.hip -o l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(512) stream(const u32x4_t* __restrict__ src, unsigned* __restrict__ sink, size_t window_chunks,
                                              size_t block_stride_chunks, int iters, int row_chunks, int row_pitch_chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * block_stride_chunks;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    size_t pos = (size_t)wave * 64 * UNROLL;
    for (int it = 0; it < iters; ++it) {
        u32x4_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            size_t c = (pos + (size_t)u * 64 + lane) % window_chunks;           // linear chunk index inside the window
            if (row_chunks) c = (c / row_chunks) * row_pitch_chunks + c % row_chunks;   // rows of row_chunks*16 B at a larger pitch
            v[u] = src[base + c];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        pos += (size_t)8 * 64 * UNROLL;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}

// same streams, but through LDS-DMA (global_load_lds_dwordx4: data lands in LDS, no VGPR): is the DMA path as wide?
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int UNROLL>
__global__ void __launch_bounds__(512) stream_dma(const u32x4_t* __restrict__ src, unsigned* __restrict__ sink, size_t window_chunks,
                                                  size_t block_stride_chunks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = (size_t)blockIdx.x * block_stride_chunks;
    size_t pos = (size_t)wave * 64 * UNROLL;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t c = (pos + (size_t)u * 64 + lane) % window_chunks;
            dma16(src + base + c, (unsigned)((wave * UNROLL + u) * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pos += (size_t)8 * 64 * UNROLL;
    }
    if (smem[threadIdx.x] == 0x7f && iters < 0) sink[threadIdx.x] = 1;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    u32x4_t* d; unsigned* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4096);
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    struct Case { const char* name; size_t window; size_t stride; int row_chunks, pitch_chunks; };
    const Case cases[] = {
        {"L1-resident 8 KB per block          ", 8 << 10, 8 << 10, 0, 0},
        {"L2-resident 192 KB shared by all     ", 192 << 10, 0, 0, 0},
        {"L2-resident 192 KB, 64 B of 128 B rows", 96 << 10, 0, 4, 8},
        {"HBM stream, 4 MB per block, contiguous", 4 << 20, 4 << 20, 0, 0},
        {"HBM stream, 128 B of each 256 B row   ", 2 << 20, 4 << 20, 8, 16},
        {"HBM stream, 64 B of each 256 B row    ", 1 << 20, 4 << 20, 4, 16},
    };
    for (const Case& c : cases) {
        for (int unroll : {4, 16}) {
            const int iters = 4096 / unroll;                 // 8 waves x 4096 KB = 4 MB per block... per wave 4096 loads of 1 KB
            auto launch = [&]() {
                if (unroll == 4) hipLaunchKernelGGL(stream<4>, dim3(ncu), dim3(512), 0, 0, d, sink, c.window / 16, c.stride / 16, iters, c.row_chunks, c.pitch_chunks);
                else hipLaunchKernelGGL(stream<16>, dim3(ncu), dim3(512), 0, 0, d, sink, c.window / 16, c.stride / 16, iters, c.row_chunks, c.pitch_chunks);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double tot = (double)ncu * 8 * iters * unroll * 1024.0;
            printf("%s unroll %2d: %7.3f ms  %8.1f GB/s  %6.1f B/clk/CU @2.1GHz\n", c.name, unroll, ms, tot / ms / 1e6, tot / ncu / (ms * 1e-3) / 2.1e9);
        }
    }
    for (const Case& c : cases) {
        if (c.row_chunks) continue;
        for (int unroll : {4, 8}) {
            const int iters = 4096 / unroll;
            auto launch = [&]() {
                if (unroll == 4) hipLaunchKernelGGL(stream_dma<4>, dim3(ncu), dim3(512), 8 * 4 * 1024, 0, d, sink, c.window / 16, c.stride / 16, iters);
                else hipLaunchKernelGGL(stream_dma<8>, dim3(ncu), dim3(512), 8 * 8 * 1024, 0, d, sink, c.window / 16, c.stride / 16, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double tot = (double)ncu * 8 * iters * unroll * 1024.0;
            printf("LDS-DMA %s unroll %2d: %7.3f ms  %8.1f GB/s  %6.1f B/clk/CU @2.1GHz\n", c.name, unroll, ms, tot / ms / 1e6, tot / ncu / (ms * 1e-3) / 2.1e9);
        }
    }
    return 0;
}
