// What does the LDS-DMA path of one CU deliver when it carries the two streams of the resblock conv kernels at once?
//   W: a 16 KB weight slab per sub-step, from a 196 KB buffer every CU re-reads (L2 resident);
//   A: 32 KB of activations per K block (3 sub-steps), 128-byte pieces of 512-byte rows, streamed once from HBM.
// 256 persistent blocks x 8 waves, the kernel's own issue / wait / barrier pattern, no compute.  Modes:
//   0 W only   1 A only   2 both as adf_gemm_rb.h issues them (W first, A behind it in sub-step 0; waits vmcnt(4), 0, 0)
//   3 both, W by waves 0-3 and A by waves 4-7 (separate in-order queues: A is only waited for at the end of the K block)
//   4 as 2, A from an L2-resident window (is it the HBM latency or the path?)   5 as 2, A spread over the three sub-steps
//   6 as 3, but the A waves issue their 8 pieces over the three sub-steps (3 + 3 + 2)
//   7 as 2, but the weight slab travels global_load_dwordx4 -> registers -> ds_write_b128 instead of LDS-DMA (does the register return path
//     overlap the DMA path?)
// hipcc --offload-arch=gfx950 -O3 dma_mix.hip -o dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void dma(const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(base), "v"(voff), "s"(lds) : "memory");
}
#define WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int MODE>
__global__ void __launch_bounds__(512) mix(const char* __restrict__ a, const char* __restrict__ w, unsigned long long* __restrict__ cyc,
                                           int tiles_per_block, size_t a_window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voffA = (unsigned)(lane >> 3) * 512u + (unsigned)(lane & 7) * 16u;     // row (lane >> 3) of a piece, 16-byte chunk
    const unsigned voffW = (unsigned)lane * 16u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int g = 0, s = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        size_t tile_off = ((size_t)blockIdx.x * tiles_per_block + t) * (256 * 512);
        if (MODE == 4) tile_off %= a_window;
        for (int kb = 0; kb < 4; ++kb, ++g) {
            const char* abase = a + tile_off + kb * 128;
            const unsigned stA = (unsigned)((g % 3) * 32768);
            auto issue_w = [&](int npieces, int first) {
                const char* wb = w + (size_t)((s % 12) * 16384);
                const unsigned stW = 98304u + (unsigned)((s & 1) * 16384);
                for (int p = 0; p < npieces; ++p) dma(wb + (first + p) * 1024, voffW, stW + (unsigned)(first + p) * 1024u);
            };
            auto issue_a = [&](int npieces, int first) {
                for (int p = 0; p < npieces; ++p) dma(abase + (size_t)(first + p) * 8 * 512, voffA, stA + (unsigned)(first + p) * 1024u);
            };
            for (int sub = 0; sub < 3; ++sub, ++s) {
                if (MODE == 0) { issue_w(2, wave * 2); WAIT(0); }
                else if (MODE == 1) { if (sub == 0) issue_a(4, wave * 4); if (sub == 1) WAIT(0); }
                else if (MODE == 2 || MODE == 4) {
                    issue_w(2, wave * 2);
                    if (sub == 0) { issue_a(4, wave * 4); WAIT(4); } else WAIT(0);
                } else if (MODE == 5) {
                    issue_w(2, wave * 2);
                    if (sub == 0) { issue_a(2, wave * 4); WAIT(2); } else if (sub == 1) { issue_a(2, wave * 4 + 2); WAIT(2); } else WAIT(0);
                } else if (MODE == 3) {
                    if (wave < 4) { issue_w(4, wave * 4); WAIT(0); }
                    else { if (sub == 0) issue_a(8, (wave - 4) * 8); if (sub == 2) WAIT(0); }
                } else if (MODE == 7) {
                    const char* wb = w + (size_t)((s % 12) * 16384) + (size_t)wave * 2048;
                    const unsigned stW = 98304u + (unsigned)((s & 1) * 16384) + (unsigned)wave * 2048u + (unsigned)lane * 16u;
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 r0, r1;
                    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=&v"(r0), "=&v"(r1) : "v"(voffW), "s"(wb) : "memory");
                    if (sub == 0) { issue_a(4, wave * 4); asm volatile("s_waitcnt vmcnt(4)" : "+v"(r0), "+v"(r1) :: "memory"); }
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1) :: "memory");
                    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"(stW), "v"(r0), "v"(r1) : "memory");
                } else if (MODE == 6) {
                    if (wave < 4) { issue_w(4, wave * 4); WAIT(0); }
                    else { const int f = (wave - 4) * 8; if (sub == 0) issue_a(3, f); else if (sub == 1) issue_a(3, f + 3); else { issue_a(2, f + 6); WAIT(0); } }
                }
                BARRIER();
            }
        }
    }
    WAIT(0);
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    if (smem[threadIdx.x] == 0x7f && tiles_per_block < 0) cyc[0] = 1;
}

int main() {
    const size_t abytes = (size_t)3 << 30;
    char *a, *w; unsigned long long* cyc;
    hipMalloc(&a, abytes); hipMalloc(&w, 256 << 10); hipMalloc(&cyc, 256 * 8);
    hipMemset(a, 1, abytes); hipMemset(w, 1, 256 << 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = 16;                                  // 256 blocks x 16 tiles x 128 KB = 512 MB of A per launch
    const char* names[] = {"W only (16 KB / sub-step, L2)", "A only (32 KB / K block, HBM)", "both, one queue per wave (rb order)",
                           "both, W waves 0-3 / A waves 4-7", "both, A from an L2 window", "both, A spread over sub-steps", "split waves, A spread",
                           "both, W through registers"};
    for (int mode = 0; mode < 8; ++mode) {
        auto launch = [&]() {
#define L(M) hipFuncSetAttribute((const void*)mix<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(mix<M>, dim3(256), dim3(512), 131072, 0, a, w, cyc, tiles, (size_t)16 << 20)
            switch (mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break; default: L(7); }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
        const double kblocks = tiles * 4.0;
        const double wb = mode == 1 ? 0 : 48.0 * 1024, ab = mode == 0 ? 0 : 32.0 * 1024;
        printf("mode %d %-36s %7.3f ms  %7.0f cycles / K block  %5.1f B/clk/CU  (A %.0f GB/s, W %.0f GB/s chip-wide)\n", mode, names[mode], ms, mean / kblocks,
               (wb + ab) / (mean / kblocks), ab * kblocks * 256 / ms / 1e6, wb * kblocks * 256 / ms / 1e6);
    }
    return 0;
}
