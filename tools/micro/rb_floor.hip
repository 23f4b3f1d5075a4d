// The floor of conv_gemm_rb_kernel<1, false, 2>'s steady-state sub-step: its EXACT per-wave instruction mix with no data dependency between the
// streams -- what the SIMD's issue port, the matrix pipe, the LDS and the CU's LDS-DMA path deliver when nothing ever waits for anything but them.
//
// One sub-step of one wave of the shipped kernel (ISA of the product build, tools/kernel_resources.py / hipcc -S; profiles/README.md round 4):
//   16 x v_mfma_f32_32x32x16_bf16 on fragments read from LDS (per K step: 2 A + 2 B ds_read_b128 = 16 reads, double-buffered one K step ahead),
//   the SiLU(GroupNorm) prologue of 12 elements per lane riding in the 16 MFMA gaps (per element: unpack, fma, fma, v_exp, add, v_rcp, mul, half a
//   v_cvt_pk = 87 single-rate + 24 quarter-rate vector instructions per wave and sub-step incl. the 3 ds_read_b64 / 3 ds_write_b64 of the groups and the
//   table reads), 2 LDS-DMA pieces of a weight slab from an L2-resident buffer, 4.5 / 3 pieces of activations streamed once from HBM (pieces 0-1 + halo in tap
//   0, pieces 2-3 in tap 1), one workgroup barrier.
// Here: the same instructions on the same LDS layout, 256 persistent workgroups x 8 waves, but the MFMA operands, the prologue's inputs and the DMA targets
// are unrelated (the prologue reads and writes its own 8 KB, the DMAs land in a ring nobody reads), and the only waits are "at most 12 of this wave's
// DMAs in flight" (the depth the kernel's ring allows) and the barrier.  Modes knock streams out:
//   0 all four (MFMA + fragment reads, prologue, DMA, barrier)   1 without the DMAs   2 without the prologue   3 MFMA + fragment reads + barrier only
//   4 all, no barrier   5 MFMA + prologue, no fragment reads, no DMA, no barrier (the issue port alone: tools/micro/role_split.hip mode 6)
//   6 all, with the younger four waves at s_setprio 1
//   7 all, PROGRESS-BASED priority: every wave lowers its own priority as it advances through the sub-step (s_setprio 3 - K step, back to 3 behind the barrier):
//     of the two waves of a SIMD the one that is behind always wins the arbitration, so both reach the barrier together instead of the older one idling there
//   8 all, one barrier per THREE sub-steps (what a ring deep enough to drop two of three barriers would buy)
//   9 as 0 and 10 as 3, with the same FLOPs as 2 x v_mfma_f32_16x16x32_bf16 per 32x32x16 (MI355X_MICROARCH.md "DVFS give-back" (7): the 16x16x32 shape holds a
//     higher clock under load -- compare WALL time, not cycles)
// Output: cycles per sub-step (s_memtime over the loop / sub-steps), median over workgroups, and the clock (s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize rb_floor.hip -o bin/rb_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma2(const char* base, unsigned va, unsigned vb, unsigned la, unsigned lb) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
                 :: "s"(base), "v"(va), "v"(vb), "s"(la), "s"(lb) : "memory");
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}

constexpr int kStageA = 33 * 1024, kOffW = 3 * kStageA, kStageW = 16 * 1024, kOffP = kOffW + 2 * kStageW, kOffTab = kOffP + 8 * 1024 * 2, kLds = kOffTab + 4096;

template <int MODE>
__global__ void __launch_bounds__(512) floor_kernel(const char* __restrict__ act, const char* __restrict__ wgt, unsigned long long* __restrict__ cyc,
                                                    float* __restrict__ sink, int substeps, size_t act_bytes) {
    constexpr bool kMfma = true, kFrag = MODE != 5, kPro = MODE != 2 && MODE != 3 && MODE != 10, kDma = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6,
                   kBar = MODE != 4 && MODE != 5;
    constexpr bool kDma2 = MODE == 7 || MODE == 8 || MODE == 9;
    constexpr bool k16 = MODE == 9 || MODE == 10;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    f32x4_t acc16[2][2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q4 = 0; q4 < 4; ++q4) acc16[i][j][q4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    if (MODE == 6 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (MODE == 7) __builtin_amdgcn_s_setprio(3);
    for (int i = tid; i < kLds / 4; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u + (unsigned)i * 0x00010001u % 0x00400040u;     // finite bf16 pairs
    __syncthreads();
    // fragment addresses: the kernel's swizzled 128-byte rows
    unsigned aadr[4], wadr[4];
    {
        const int row = wm * 64 + r, f = (row >> 1) & 7, fw = (r >> 1) & 7;
        const unsigned ab = (unsigned)(row * 128 + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5));
        const unsigned wb = (unsigned)(kOffW + (wn * 64 + r) * 128 + ((h ^ (fw & 1)) << 4) + ((fw >> 1) << 5));
        for (int ks = 0; ks < 4; ++ks) { aadr[ks] = ab ^ (unsigned)(ks << 5); wadr[ks] = wb ^ (unsigned)(ks << 5); }
    }
    f32x16_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8_t fa[2][2], fb[2][2];
    for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) { fa[c][i] = *(const bf16x8_t*)(smem + aadr[0] + i * 32 * 128); fb[c][i] = *(const bf16x8_t*)(smem + wadr[0] + i * 32 * 128); }
    // prologue state: 3 groups of 4 elements per sub-step as in part_gap (8 stages of 4 independent instructions per group)
    char* const pbuf = smem + kOffP + wave * 1024 * 2 + lane * 16;
    const float* const tab = (const float*)(smem + kOffTab) + (lane & 7) * 16;
    float x[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f}, ta[4], tb[4];
    u32x2_t raw[3], pk = {0u, 0u};
    // DMA sources: activations streamed once (each workgroup its own region, rows of 512 B, 128-byte pieces), weights from a 196 KB L2-resident buffer
    const unsigned voffA = (unsigned)(wave * 8 + (lane >> 3)) * 512u + (unsigned)(lane & 7) * 16u;
    const unsigned voffW = (unsigned)(wave * 8 + (lane >> 3)) * 128u + (unsigned)(lane & 7) * 16u;
    const size_t per_block = act_bytes / gridDim.x;
    const char* abase = act + (size_t)blockIdx.x * per_block;
    size_t aoff = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < substeps; ++s) {
        const int tap = s % 3;
        const unsigned stA = (unsigned)(((s / 3) % 3) * kStageA), stW = (unsigned)(kOffW + (s & 1) * kStageW);
        if (kPro) {
            // part_begin: (a, b) of 4 channels from the table, the raw 8-byte halves of the groups
            const f32x2_t t0v = *(const f32x2_t*)(tab), t1v = *(const f32x2_t*)(tab + 2), t2v = *(const f32x2_t*)(tab + 4), t3v = *(const f32x2_t*)(tab + 6);
            ta[0] = t0v.x; tb[0] = t0v.y; ta[1] = t1v.x; tb[1] = t1v.y; ta[2] = t2v.x; tb[2] = t2v.y; ta[3] = t3v.x; tb[3] = t3v.y;
            for (int k = 0; k < 3; ++k) raw[k] = *(const u32x2_t*)(pbuf + k * 8192 % 2048);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (kFrag && ks + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { fa[nxt][i] = *(const bf16x8_t*)(smem + stA + aadr[ks + 1] + i * 32 * 128); fb[nxt][i] = *(const bf16x8_t*)(smem + (stW - kOffW) + wadr[ks + 1] + i * 32 * 128); }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (k16) {
                        // the same FLOPs: a 32 x 32 x 16 product = two 16 x 16 x 32 products (operands reused as they are: timing only)
                        acc16[i][j][(ks & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][i], fb[cur][j], acc16[i][j][(ks & 1) * 2], 0, 0, 0);
                        acc16[i][j][(ks & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][i], fb[cur][j], acc16[i][j][(ks & 1) * 2 + 1], 0, 0, 0);
                    } else if (kMfma) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kPro) {
                        // gap q = ks * 4 + i * 2 + j carries slots 6 q .. 6 q + 5 of the 96 of this sub-step (3 groups x 8 stages x 4 elements)
                        const int q = ks * 4 + i * 2 + j;
#pragma unroll
                        for (int n = q * 6; n < q * 6 + 6; ++n) {
                            const int k = n >> 5, st = (n >> 2) & 7, e = n & 3;
                            switch (st) {
                                case 0: { const unsigned w = (e & 2) ? raw[k].y : raw[k].x; x[e] = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)); break; }
                                case 1: x[e] = fmaf(x[e], ta[e], tb[e]); break;
                                case 2: u[e] = x[e] * -1.4426950408889634f; break;
                                case 3: u[e] = __builtin_amdgcn_exp2f(u[e]); break;
                                case 4: u[e] = u[e] + 1.0f; break;
                                case 5: u[e] = __builtin_amdgcn_rcpf(u[e]); break;
                                case 6: x[e] = x[e] * u[e]; break;
                                default:
                                    if (e == 0) pk.x = pack2(x[0], x[1]);
                                    else if (e == 1) pk.y = pack2(x[2], x[3]);
                                    else if (e == 2) *(u32x2_t*)(pbuf + (k * 8192) % 2048 + 8) = pk;
                                    break;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(x[e])); asm volatile("" : "+v"(u[e])); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (MODE == 7) {
                if (ks == 0) __builtin_amdgcn_s_setprio(2);
                else if (ks == 1) __builtin_amdgcn_s_setprio(1);
                else if (ks == 2) __builtin_amdgcn_s_setprio(0);
            }
            if (kDma || kDma2) {
                if (ks == 0) dma2(wgt + (size_t)((s % 12) * 16384), voffW, voffW + 64u * 128u, stW ^ (unsigned)kStageW, (stW ^ (unsigned)kStageW) + 8192u);   // (the other stage)
                else if (ks == 1 && tap < 2) {
                    const unsigned st2 = (unsigned)((((s / 3) + 2) % 3) * kStageA) + (unsigned)wave * 1024u + (tap ? 16384u : 0u);
                    dma2(abase + aoff, voffA, voffA + 64u * 512u, st2, st2 + 8192u);
                    aoff += 65536; if (aoff + 131072 > per_block) aoff = 0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kDma || kDma2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (kBar && (MODE != 8 || s % 3 == 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 7) __builtin_amdgcn_s_setprio(3);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = x[0] + u[1];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q4 = 0; q4 < 4; ++q4) sum += acc16[i][j][q4].x + acc16[i][j][q4].w;
    if (sum == 12345.678f) sink[0] = sum;
    // the SLOWEST wave's time (without a barrier the older wave of a SIMD pair runs ahead: wave 0's own time would flatter modes 4 / 5)
    __shared__ unsigned long long wt[8][2];
    if (lane == 0) { wt[wave][0] = t1 - t0; wt[wave][1] = r1 - r0; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long m0 = 0, m1 = 0;
        for (int w = 0; w < 8; ++w) { m0 = wt[w][0] > m0 ? wt[w][0] : m0; m1 = wt[w][1] > m1 ? wt[w][1] : m1; }
        cyc[blockIdx.x * 2] = m0; cyc[blockIdx.x * 2 + 1] = m1;
    }
}

template <int MODE>
static void run(const char* name, const char* act, const char* wgt, unsigned long long* dcyc, float* sink, size_t act_bytes) {
    const int substeps = 48 * 8;
    hipFuncSetAttribute((const void*)floor_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(floor_kernel<MODE>, dim3(256), dim3(512), kLds, 0, act, wgt, dcyc, sink, substeps, act_bytes);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(512);
    hipMemcpy(c.data(), dcyc, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> per, ghz;
    for (int b = 0; b < 256; ++b) { per.push_back((double)c[2 * b] / substeps); ghz.push_back((double)c[2 * b] / ((double)c[2 * b + 1] * 10.0) ); }
    std::sort(per.begin(), per.end()); std::sort(ghz.begin(), ghz.end());
    printf("mode %d  %-62s %7.0f cycles per sub-step (min %6.0f p98 %6.0f)  %.2f GHz  => %5.1f us per 48 sub-steps\n", MODE, name, per[128], per[0], per[250], ghz[128],
           per[128] * 48 / (ghz[128] * 1e3));
}

int main() {
    const size_t act_bytes = (size_t)2 << 30;
    char *act, *wgt; unsigned long long* dcyc; float* sink;
    hipMalloc(&act, act_bytes); hipMalloc(&wgt, 256 * 1024); hipMalloc(&dcyc, 512 * 8); hipMalloc(&sink, 64);
    hipMemset(act, 0x3f, act_bytes); hipMemset(wgt, 0x3f, 256 * 1024);
    printf("rb_floor: one sub-step of conv_gemm_rb_kernel<1,false,2> per wave = 16 MFMA + 16 ds_read_b128 + 111 vector instructions (24 of them v_exp / v_rcp) + 3.5 LDS-DMA pieces + 1 barrier\n");
    run<3>("MFMA + fragment reads + barrier", act, wgt, dcyc, sink, act_bytes);
    run<5>("MFMA + prologue (issue port alone: no LDS fragments, DMA, barrier)", act, wgt, dcyc, sink, act_bytes);
    run<2>("MFMA + fragment reads + DMA + barrier (no prologue)", act, wgt, dcyc, sink, act_bytes);
    run<1>("MFMA + fragment reads + prologue + barrier (no DMA)", act, wgt, dcyc, sink, act_bytes);
    run<4>("all four streams, no barrier", act, wgt, dcyc, sink, act_bytes);
    run<0>("all four streams (the kernel's sub-step)", act, wgt, dcyc, sink, act_bytes);
    run<6>("all four streams, younger four waves at s_setprio 1", act, wgt, dcyc, sink, act_bytes);
    run<7>("all four streams, progress-based priority (3 - K step)", act, wgt, dcyc, sink, act_bytes);
    run<8>("all four streams, one barrier per three sub-steps", act, wgt, dcyc, sink, act_bytes);
    run<10>("MFMA 16x16x32 (same FLOPs) + fragment reads + barrier", act, wgt, dcyc, sink, act_bytes);
    run<9>("all four streams with MFMA 16x16x32 (same FLOPs)", act, wgt, dcyc, sink, act_bytes);
    run<3>("MFMA + fragment reads + barrier (again)", act, wgt, dcyc, sink, act_bytes);
    run<0>("all four streams (again)", act, wgt, dcyc, sink, act_bytes);
    return 0;
}
