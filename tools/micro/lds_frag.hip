// How much matrix-pipe time do LDS fragment reads leave?  Every wave runs K steps of { NR x ds_read_b128 of the NEXT step's fragments ;
// NM x v_mfma_f32_32x32x16_bf16 on the current ones ; s_waitcnt lgkmcnt(0) } -- the inner loop of the implicit-GEMM kernels (144-byte LDS rows,
// lane (r, h) reads the 16-byte chunk 2 ks + h of row r: conflict-free) for the wave tiles
//   32 x 64  : 1 A + 2 B fragments, 2 MFMAs per K step   (conv2d_tile_kernel WR = 1, conv2d_gemm_kernel, conv_gemm_kernel)
//   64 x 64  : 2 A + 2 B fragments, 4 MFMAs              (conv_gemm_rb_kernel half tiles, conv2d_tile_kernel WR = 2)
//   64 x 128 : 2 A + 4 B fragments, 8 MFMAs
//   32 x 64 with the B fragments NOT from LDS (register-resident: the weight-fragment ring of wn_layer_bf16_wide_kernel): 1 read, 2 MFMAs
// plus the two baselines (MFMAs only, reads only), at 1, 2 and 4 waves per SIMD (256 / 512 / 1024 threads, one workgroup per CU, every CU busy).
// Prints cycles per K step of a wave (median over workgroups) and the matrix-pipe utilisation that implies: NM x 32 cycles x waves per SIMD / cycles.
//   hipcc --offload-arch=gfx950 -O3 lds_frag.hip -o lds_frag
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NA, int NB, int NBR, bool MFMA, bool READ>
__global__ void __launch_bounds__(1024) k(float* __restrict__ out, unsigned long long* __restrict__ cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 40 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3f803f80u + (unsigned)i * 0x00010001u % 0x00400040u;
    __syncthreads();
    // A rows: 32 rows per fragment at row ((wave & 3) * 64 + i * 32 + r); B rows behind them
    const unsigned a0 = (unsigned)(((wave & 3) * 64 + r) * 144 + h * 16);
    const unsigned b0 = (unsigned)(18432 + (((wave >> 2) & 1) * 64 + r) * 144 + h * 16);
    u32x4_t fa[2][NA > 0 ? NA : 1], fb[2][NB > 0 ? NB : 1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < (NA > 0 ? NA : 1); ++i) fa[s][i] = u32x4_t{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int j = 0; j < (NB > 0 ? NB : 1); ++j) fb[s][j] = u32x4_t{0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u};
    }
    constexpr int NAe = NA > 0 ? NA : 1, NBe = NB > 0 ? NB : 1;       // fragments multiplied (register-resident when not read)
    f32x16_t acc[NAe][NBe];
#pragma unroll
    for (int i = 0; i < NAe; ++i)
#pragma unroll
        for (int j = 0; j < NBe; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    auto reads = [&](int ks, u32x4_t (&xa)[NAe], u32x4_t (&xb)[NBe]) __attribute__((always_inline)) {
        if constexpr (READ) {
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xa[i]) : "v"(a0 + (unsigned)(i * 32 * 144 + (ks & 3) * 32)) : "memory");
#pragma unroll
            for (int j = 0; j < NBR; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(xb[j]) : "v"(b0 + (unsigned)(j * 32 * 144 + (ks & 3) * 32)) : "memory");
        }
    };
    auto mfmas = [&](u32x4_t (&xa)[NAe], u32x4_t (&xb)[NBe]) __attribute__((always_inline)) {
        if constexpr (MFMA) {
#pragma unroll
            for (int i = 0; i < NAe; ++i)
#pragma unroll
                for (int j = 0; j < NBe; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, xa[i]), __builtin_bit_cast(bf16x8_t, xb[j]), acc[i][j], 0, 0, 0);
        }
    };
    reads(0, fa[0], fb[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int ks = 0; ks < steps; ks += 2) {
        reads(ks + 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reads(ks + 2, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NAe; ++i)
#pragma unroll
        for (int j = 0; j < NBe; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    s += __builtin_bit_cast(float, fa[0][0][0]) + __builtin_bit_cast(float, fb[0][0][0]) + __builtin_bit_cast(float, fa[1][0][1]) + __builtin_bit_cast(float, fb[1][0][1]);
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NA, int NB, int NBR, bool MFMA, bool READ>
static void run(const char* name, int nm, int threads, float* out, unsigned long long* cyc, int ncu) {
    const int steps = 4096;
    hipFuncSetAttribute((const void*)k<NA, NB, NBR, MFMA, READ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NA, NB, NBR, MFMA, READ>), dim3(ncu), dim3(threads), 96 * 1024, 0, out, cyc, steps);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NA, NB, NBR, MFMA, READ>), dim3(ncu), dim3(threads), 96 * 1024, 0, out, cyc, steps);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), cyc, ncu * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // s_memtime counts at 100 MHz on gfx950: convert with the clock measured by the MFMA-only run? -> print raw ticks per step; the
    // ratio between configurations is what matters, and the MFMA-only line calibrates ticks per 32-cycle MFMA
    const double per = (double)h[ncu / 2] / steps;
    const double ns = ms * 1e6 / steps;
    printf("%-36s threads %4d  ticks/step %8.2f  ns/step %8.2f  MFMAs/step %d  ns per MFMA-slot %7.3f  TFLOP/s (chip) %7.1f\n", name, threads, per, ns, nm,
           nm ? ns / (nm * (threads / 256.0)) : 0.0, nm ? 32768.0 * nm * (threads / 64) * ncu / ns / 1e3 : 0.0);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)ncu * 1024 * 4);
    hipMalloc(&cyc, (size_t)ncu * 8);
    printf("# %s, %d CUs; 96 KB of LDS per workgroup (one workgroup per CU); 'ticks per MFMA-slot' = ticks per step / (MFMAs per step x waves per SIMD):\n"
           "# equal to the MFMA-only value when the matrix pipe is saturated, larger by the factor the fragment reads cost (wall time by events)\n", p.name, ncu);
    for (int threads : {256, 512, 1024}) {
        run<1, 2, 2, true, false>("MFMA only 32x64 (2 per step)", 2, threads, out, cyc, ncu);
        run<2, 4, 4, true, false>("MFMA only 64x128 (8 per step)", 8, threads, out, cyc, ncu);
        run<1, 2, 2, false, true>("reads only 3 x b128", 0, threads, out, cyc, ncu);
        run<2, 4, 4, false, true>("reads only 6 x b128", 0, threads, out, cyc, ncu);
        run<1, 2, 2, true, true>("32x64: 3 reads + 2 MFMAs", 2, threads, out, cyc, ncu);
        run<2, 2, 2, true, true>("64x64: 4 reads + 4 MFMAs", 4, threads, out, cyc, ncu);
        run<2, 4, 4, true, true>("64x128: 6 reads + 8 MFMAs", 8, threads, out, cyc, ncu);
        run<1, 2, 0, true, true>("32x64: A from LDS, B in registers", 2, threads, out, cyc, ncu);
    }
    return 0;
}
