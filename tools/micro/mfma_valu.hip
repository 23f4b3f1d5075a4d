// What does vector work cost beside MFMAs?  Every wave runs  { v_mfma_f32_32x32x16_bf16 ; FILL } x 16 per "sub-step"
// (random bf16 operands in registers, four accumulators in rotation), with FILL one of
//   0 nothing | 1 N x v_fma_f32 | 2 N x v_exp_f32 | 3 the per-element SiLU(GN) chain of adf_gemm_pp.h (and/lshl, fma, mul,
//   exp, add, rcp, mul, cvt_pk every other gap) | 4 the same chain software-pipelined as in the kernel (exp of element e
//   beside rcp of element e-1) | 5 chain + one ds_read_b64 + one ds_write_b64 per 4 gaps
// at one wave per SIMD (256 threads) and two (512).  Prints s_memtime cycles per MFMA (median over workgroups, wave 0).
//   hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int KIND, int N>
__device__ __forceinline__ void fill(float (&x)[8], float c, float d, unsigned& w, char* lds, int q) {
    if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(c), "v"(d));
    } else if constexpr (KIND == 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
    } else if constexpr (KIND == 3) {
        // one element, serial chain
        float v, t;
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v) : "v"(w));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(d));
        asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t) : "v"(v));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[q & 1]) : "v"(v), "v"(t));
        if (q & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x[0]), "v"(x[1]));
    } else if constexpr (KIND >= 4) {
        // stage b of the previous element (x[2] = v, x[3] = exp) beside stage a of this one
        float v, t, u = x[3];
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v) : "v"(w));
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(u));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(d));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(u));
        asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t) : "v"(v));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[q & 1]) : "v"(x[2]), "v"(u));
        x[2] = v; x[3] = t;
        if (q & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x[0]), "v"(x[1]));
        if constexpr (KIND == 5) {
            if ((q & 3) == 3) {
                unsigned long long rv;
                asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)lds), "v"((unsigned long long)w) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:8192\n\ts_waitcnt lgkmcnt(0)" : "=v"(rv) : "v"((unsigned)(size_t)lds) : "memory");
                w ^= (unsigned)rv;
            }
        }
    }
}

template <int KIND, int N>
__global__ void __launch_bounds__(512) k(const bf16x8_t* __restrict__ ab, float* __restrict__ out, unsigned long long* __restrict__ cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8_t a0 = ab[tid], a1 = ab[tid + 512], b0 = ab[tid + 1024], b1 = ab[tid + 1536];
    f32x16_t acc[4] = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (float)(tid + i);
    unsigned w = 0x3f803f80u + tid;
    char* lds = smem + tid * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((q & 2) ? a1 : a0, (q & 1) ? b1 : b0, acc[q & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fill<KIND, N>(x, 1.0001f, 0.0001f, w, lds, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) r += acc[i][e];
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * blockDim.x + tid] = r + __uint_as_float(w);
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    (void)lane;
}

template <int KIND, int N>
static void run(const char* name, bf16x8_t* ab, float* out, unsigned long long* cyc, int threads) {
    const int steps = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(threads), 160 * 1024, 0, ab, out, cyc, steps);   // 160 KB: one block per CU
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[blocks / 2] / (steps * 16.0);
    const double mf = (double)blocks * threads / 64 * steps * 16;
    printf("%-34s waves/SIMD %d  cycles per MFMA (per wave) %7.1f   %.0f TF/s   clock %.2f GHz\n", name, threads / 256, per,
           mf * 32768.0 / (ms * 1e-3) / 1e12, (double)h[blocks / 2] / (ms * 1e-3) / 1e9);
}

int main() {
    bf16x8_t* ab; float* out; unsigned long long* cyc;
    hipMalloc(&ab, 2048 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned short> h(2048 * 8);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3f00 + ((s >> 16) & 0xff)) ^ (unsigned short)((s >> 8) & 0x8000); }
    hipMemcpy(ab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<0, 0>("mfma only", ab, out, cyc, threads);
        run<1, 2>("+ 2 v_fma", ab, out, cyc, threads);
        run<1, 4>("+ 4 v_fma", ab, out, cyc, threads);
        run<1, 6>("+ 6 v_fma", ab, out, cyc, threads);
        run<1, 8>("+ 8 v_fma", ab, out, cyc, threads);
        run<1, 12>("+ 12 v_fma", ab, out, cyc, threads);
        run<2, 1>("+ 1 v_exp", ab, out, cyc, threads);
        run<2, 2>("+ 2 v_exp", ab, out, cyc, threads);
        run<2, 4>("+ 4 v_exp", ab, out, cyc, threads);
        run<3, 0>("+ SiLU chain (serial)", ab, out, cyc, threads);
        run<4, 0>("+ SiLU chain (2-stage)", ab, out, cyc, threads);
        run<5, 0>("+ SiLU 2-stage + LDS r/w per 4", ab, out, cyc, threads);
    }
    return 0;
}
