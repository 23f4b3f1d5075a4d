// WaveNetNoise (DiffWave-style, unconditional) on gfx950 -- BASELINE config 5, SURVEY.md 8f row 4.
// Reference: src/models/backbones/wavenet.py (WeightNorm :15-55, Conv :68-82, diffusion_embedding :88-92, ResidualBlock
// :94-116, ResidualGroup :117-152, WaveNetNoise :153-180).
//
// One launch per residual layer.  The layer is two GEMMs per position (dilated k = 3 conv C -> 2C, K = 3C; 1x1 conv C -> 2C)
// with a gate between them and a residual / skip epilogue: 1.05 MFLOP against ~2.5 KB of HBM traffic per position at C = 256,
// i.e. far on the matrix side of the ridge, so the throughput kernel is an MFMA kernel and the weights (1 MB per layer in
// bf16, re-read by every tile from L2) are its second stream:
//   wn_layer_bf16_kernel: 64 positions per 512-thread workgroup.  The three dilated windows of y (rows t0 + (tap - 1) d, zero
//     outside the sample) are staged once into LDS; each wave owns 32 gate columns AND the 32 matching filter columns of GEMM 1
//     (so sigmoid * tanh is register-local) and reads its weight fragments straight from a fragment-major copy in global
//     memory, 8 K steps ahead (each weight byte once per workgroup); the gated tile goes to LDS as bf16 and is the A operand of
//     GEMM 2, whose residual columns update the centre window in place (LDS) and whose skip columns are accumulated into the
//     fp32 skip sum in HBM; the centre window then leaves as 16-byte stores.
//   wn_layer_f32_kernel: the parity path -- the same fusion on the vector ALUs in exact fp32 (one thread per channel, 16
//     positions per workgroup, operands transposed in LDS so that a ds_read_b128 is a 4-position broadcast).
#include "adf_wavenet.h"
#include <type_traits>

#ifdef ADF_WN_STAMP
// diagnostic build (tools/build_variant.sh wnstamp -DADF_WN_STAMP; tools/wn_stamps.py): s_memtime at the phase boundaries of the
// bf16 layer kernel, all 8 waves of one workgroup in the middle of the grid
namespace adf { __device__ unsigned long long adf_wn_stamps[8 * 16]; }
extern "C" int adf_debug_wn_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(adf::adf_wn_stamps), sizeof(unsigned long long) * 8 * 16);
}
#define WN_STAMP(i) do { if (stamped && lane == 0) adf_wn_stamps[wave * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WN_STAMP(i) do { } while (0)
#endif

// The skip sum (fp32 read-modify-write, 2 KB per position and layer) and the y_next stores are pure streams; the layer's 1 MB of weights is re-read from L2 by
// every tile.  Non-temporal accesses for the streams (ADF_WN_NT, default on) keep them from evicting the weights.
#ifndef ADF_WN_XCD
#define ADF_WN_XCD 1
#endif
#ifndef ADF_WN_NT
#define ADF_WN_NT 1
#endif
#if ADF_WN_NT
#define WN_NT_LOAD(p) __builtin_nontemporal_load(p)
#define WN_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define WN_NT_LOAD(p) (*(p))
#define WN_NT_STORE(v, p) (*(p) = (v))
#endif

namespace adf {

typedef float f32x4_vec __attribute__((ext_vector_type(4)));

#define WN_LAUNCH_CHECK(name) (hipGetLastError() == hipSuccess ? nullptr : "launch failed: " name)

// ------------------------------------------------------------------------------------------------ weight preparation
__global__ void __launch_bounds__(1024) wn_sumsq_kernel(const float* __restrict__ v, long long numel, double* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < numel; i += 1024) { const double x = (double)v[i]; acc += x * x; }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = t;
    }
}
const char* launch_wn_sumsq(const float* v, long long numel, double* out, hipStream_t s) {
    hipLaunchKernelGGL(wn_sumsq_kernel, dim3(1), dim3(1024), 0, s, v, numel, out);
    return WN_LAUNCH_CHECK("wn_sumsq");
}

__global__ void __launch_bounds__(256) wn_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, const double* __restrict__ sumsq,
                                                      void* __restrict__ dst, int layout, int cout, int cin, int K) {
    const long long total = (long long)cout * cin * K;
    const float scale = g[0] / (float)sqrt(sumsq[0]);                 // wavenet.py:50
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        if (layout == 2) {
            ((float*)dst)[idx] = v[idx] * scale;
        } else if (layout == 0) {                                     // [K][cin][cout]
            const int co = (int)(idx % cout);
            const long long r = idx / cout;
            const int ci = (int)(r % cin), k = (int)(r / cin);
            ((float*)dst)[idx] = v[((long long)co * cin + ci) * K + k] * scale;
        } else {                                                      // [(tap, 16-channel step)][half][cout][8]
            const int e = (int)(idx & 7);
            long long r = idx >> 3;
            const int co = (int)(r % cout); r /= cout;
            const int hh = (int)(r & 1); r >>= 1;
            const int steps = cin / 16;
            const int j = (int)(r % steps), k = (int)(r / steps);
            const int ci = j * 16 + hh * 8 + e;
            ((uint16_t*)dst)[idx] = f32_to_bf16(v[((long long)co * cin + ci) * K + k] * scale);
        }
    }
}
const char* launch_wn_pack(const float* v, const float* g, const double* sumsq, void* dst, int layout, int cout, int cin, int K,
                           hipStream_t s) {
    if (layout == 1 && cin % 16) return "wn_pack: the fragment-major layout needs a multiple of 16 input channels";
    const long long total = (long long)cout * cin * K;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wn_pack_kernel, dim3(blocks), dim3(256), 0, s, v, g, sumsq, dst, layout, cout, cin, K);
    return WN_LAUNCH_CHECK("wn_pack");
}

// ------------------------------------------------------------------------------------------------ diffusion-step embedding
__global__ void __launch_bounds__(256) wn_step_embed_kernel(const float* __restrict__ t, int t_stride, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, int dim_in, int dim_mid, int dim_out,
                                                            float* __restrict__ pre) {
    __shared__ float f[1024];
    __shared__ float hdn[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float tv = t[(size_t)b * t_stride];
    const int half = dim_in / 2;
    for (int i = tid; i < half; i += 256) {                           // wavenet.py:89-91, sines first
        const float ang = tv * expf((-(float)i * 4.0f) / (float)(half - 1));
        f[i] = sinf(ang);
        f[half + i] = cosf(ang);
    }
    __syncthreads();
    for (int j = tid; j < dim_mid; j += 256) {
        float acc = b1[j];
        for (int i = 0; i < dim_in; ++i) acc = fmaf(w1[(size_t)j * dim_in + i], f[i], acc);
        hdn[j] = acc / (1.0f + expf(-acc));                            // swish, :84-86
    }
    __syncthreads();
    for (int j = tid; j < dim_out; j += 256) {
        float acc = b2[j];
        for (int i = 0; i < dim_mid; ++i) acc = fmaf(w2[(size_t)j * dim_mid + i], hdn[i], acc);
        pre[(size_t)b * dim_out + j] = acc;                            // the second swish is applied by film_kernel
    }
}
const char* launch_wn_step_embed(const float* t, int t_stride, int nb, const float* w1, const float* b1, const float* w2,
                                 const float* b2, int dim_in, int dim_mid, int dim_out, float* pre, hipStream_t s) {
    if (dim_in > 1024 || dim_mid > 1024 || dim_in % 2 || dim_in < 4) return "wn_step_embed: unsupported embedding widths";
    hipLaunchKernelGGL(wn_step_embed_kernel, dim3(nb), dim3(256), 0, s, t, t_stride, w1, b1, w2, b2, dim_in, dim_mid, dim_out, pre);
    return WN_LAUNCH_CHECK("wn_step_embed");
}

// ------------------------------------------------------------------------------------------------ input projection
template <typename T>
__global__ void __launch_bounds__(256) wn_input_kernel(const float* __restrict__ x, const float* __restrict__ coef, int coef_bstride,
                                                       const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                       const float* __restrict__ e, int e_bstride, T* __restrict__ y0, int B, int Tn, int C) {
    constexpr int EPC = Elem<T>::kPerChunk;
    const int cpr = C / EPC;
    const long long total = (long long)B * Tn * cpr;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int cc = (int)(idx % cpr);
        const long long row = idx / cpr;
        const int b = (int)(row / Tn);
        const float cin = coef ? coef[(size_t)b * coef_bstride] : 1.0f;
        const float xv = x[row] * cin;
        float f[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) {
            const int c = cc * EPC + k;
            f[k] = fmaxf(fmaf(w_in[c], xv, b_in[c]), 0.0f) + e[(size_t)b * e_bstride + c];
        }
        *(u32x4_t*)(y0 + row * C + cc * EPC) = pack16<T>(f);
    }
}
const char* launch_wn_input(const WnIO& io, const float* x, const float* coef, int coef_bstride, const float* w_in, const float* b_in,
                            void* y0, hipStream_t s) {
    const long long total = (long long)io.B * io.T * (io.C / (io.bf16 ? 8 : 4));
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (io.bf16) hipLaunchKernelGGL(wn_input_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, x, coef, coef_bstride, w_in, b_in, io.e, io.e_bstride, (bf16_t*)y0, io.B, io.T, io.C);
    else hipLaunchKernelGGL(wn_input_kernel<float>, dim3(blocks), dim3(256), 0, s, x, coef, coef_bstride, w_in, b_in, io.e, io.e_bstride, (float*)y0, io.B, io.T, io.C);
    return WN_LAUNCH_CHECK("wn_input");
}

// ------------------------------------------------------------------------------------------------ residual layer, fp32 (parity)
constexpr int kWnTP = 16;           // positions per workgroup of the fp32 kernels
constexpr float kInvSqrt2Div = 1.41421356237309504880f;

__global__ void __launch_bounds__(512) wn_layer_f32_kernel(const float* __restrict__ y, float* __restrict__ y_next, float* __restrict__ skip,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ e, int e_bstride, int n, int dil, int first,
                                                            int Tn, int C) {
    constexpr int TP = kWnTP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const At = (float*)smem;                    // [3][C][TP]: the three dilated windows, transposed
    float* const Gt = At + 3 * C * TP;                 // [C][TP]: gated activation, transposed
    const int c = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * TP;
    const float* const yb = y + (size_t)b * Tn * C;
    for (int idx = c; idx < 3 * TP * C; idx += C) {
        const int tap = idx / (TP * C), rem = idx - tap * TP * C;
        const int i = rem / C, ch = rem - i * C;
        const int t = t0 + i + (tap - 1) * dil;
        At[(tap * C + ch) * TP + i] = (t >= 0 && t < Tn) ? yb[(size_t)t * C + ch] : 0.0f;     // zero padding of y (wavenet.py:71)
    }
    __syncthreads();
    float ag[TP], af[TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) { ag[i] = b1[c]; af[i] = b1[C + c]; }
    for (int tap = 0; tap < 3; ++tap) {
        const float* wt = w1 + (size_t)tap * C * 2 * C;
        const float* at = At + (size_t)tap * C * TP;
        for (int k = 0; k < C; ++k) {
            const float wg = wt[(size_t)k * 2 * C + c], wf = wt[(size_t)k * 2 * C + C + c];
            const f32x4_vec* a4 = (const f32x4_vec*)(at + k * TP);
#pragma unroll
            for (int q = 0; q < TP / 4; ++q) {
                const f32x4_vec a = a4[q];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ag[q * 4 + u] = fmaf(wg, a[u], ag[q * 4 + u]); af[q * 4 + u] = fmaf(wf, a[u], af[q * 4 + u]); }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TP; ++i) Gt[c * TP + i] = (1.0f / (1.0f + expf(-ag[i]))) * tanhf(af[i]);       // :112-113 (first chunk = gate)
    __syncthreads();
    float ar[TP], as[TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) { ar[i] = b2[c]; as[i] = b2[C + c]; }
    for (int k = 0; k < C; ++k) {
        const float wr = w2[(size_t)k * 2 * C + c], ws = w2[(size_t)k * 2 * C + C + c];
        const f32x4_vec* a4 = (const f32x4_vec*)(Gt + k * TP);
#pragma unroll
        for (int q = 0; q < TP / 4; ++q) {
            const f32x4_vec a = a4[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ar[q * 4 + u] = fmaf(wr, a[u], ar[q * 4 + u]); as[q * 4 + u] = fmaf(ws, a[u], as[q * 4 + u]); }
        }
    }
    const float en = e[(size_t)b * e_bstride + (size_t)n * C + c];
    const float en1 = y_next ? e[(size_t)b * e_bstride + (size_t)(n + 1) * C + c] : 0.0f;
#pragma unroll
    for (int i = 0; i < TP; ++i) {
        const int t = t0 + i;
        if (t < Tn) {
            const size_t o = ((size_t)b * Tn + t) * C + c;
            if (y_next) y_next[o] = ((At[(C + c) * TP + i] - en) + ar[i]) / kInvSqrt2Div + en1;         // :116, then the next layer's :110
            skip[o] = first ? as[i] : skip[o] + as[i];                                                   // :149
        }
    }
}

// ------------------------------------------------------------------------------------------------ residual layer, bf16 (MFMA)
typedef __attribute__((ext_vector_type(8))) __bf16 wn_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float wn_f32x16_t;

__device__ __forceinline__ float wn_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float wn_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * 2.8853900817779268f)); }

// The MFMA stage shared by the layer kernel and the final kernel: a wave accumulates NN column tiles of 32 (columns nbase[j] + r)
// over KS K steps of 16 channels; A fragments from LDS (tap k starts `tap_rows` rows after tap k - 1, pitch PA), W fragments from the
// fragment-major global copy [K step][half][NCOLS][8], DEPTH steps ahead in a register ring.
template <int C, int TM>
struct WnTile {
    static constexpr int PA = C * 2 + 16;              // LDS row pitch: conflict-free 16-byte fragment reads (see adf_gemm.h)
    static constexpr int MT = TM / 32;
    static constexpr int SPT = C / 16;                 // K steps per tap
};

template <int C, int TM, int NCOLS, int NN, int TAPS, int RING = 8>
__device__ __forceinline__ void wn_gemm(const char* __restrict__ A, const void* __restrict__ W, const int (&nbase)[NN],
                                        wn_f32x16_t (&acc)[NN][TM / 32], int r, int hh, int tap_rows = TM) {
    using Tl = WnTile<C, TM>;
    constexpr int PA = Tl::PA, MT = Tl::MT, SPT = Tl::SPT, DEPTH = RING - 1;
    constexpr int KS = TAPS * SPT;
    static_assert(KS % RING == 0 && KS > DEPTH, "K steps must fill whole ring trips");
    const char* wl[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) wl[j] = (const char*)W + ((size_t)hh * NCOLS + nbase[j] + r) * 16;
    auto wfrag = [&](int j, int ks) __attribute__((always_inline)) -> wn_bf16x8_t {
        return __builtin_bit_cast(wn_bf16x8_t, *(const u32x4_t*)(wl[j] + (size_t)ks * 2 * NCOLS * 16));
    };
    auto afrag = [&](int ks, wn_bf16x8_t (&af)[MT]) __attribute__((always_inline)) {
        const int tap = ks / SPT, q = ks - tap * SPT;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const wn_bf16x8_t*)(A + ((size_t)tap * tap_rows + i * 32 + r) * PA + q * 32 + hh * 16);
    };
    wn_bf16x8_t wf[NN][RING];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < NN; ++j) wf[j][d] = wfrag(j, d);
    wn_bf16x8_t af[2][MT];
    afrag(0, af[0]);
#pragma unroll 1
    for (int kb = 0; kb < KS; kb += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int ks = kb + u;
            if (ks + DEPTH < KS) {
#pragma unroll
                for (int j = 0; j < NN; ++j) wf[j][(u + DEPTH) % RING] = wfrag(j, ks + DEPTH);
            }
            if (ks + 1 < KS) afrag(ks + 1, af[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NN; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][i], wf[j][u], acc[j][i], 0, 0, 0);
        }
    }
}

template <int C, int TM>
__global__ void __launch_bounds__(512) wn_layer_bf16_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ y_next, float* __restrict__ skip,
                                                            const void* __restrict__ w1, const float* __restrict__ b1,
                                                            const void* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ e, int e_bstride, int n, int dil, int first, int Tn) {
    using Tl = WnTile<C, TM>;
    constexpr int PA = Tl::PA, MT = Tl::MT;
    static_assert(C == 256, "a wave owns 32 of the C gate columns: 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufA = smem;                               // [3][TM][PA]: the dilated windows of y
    char* const bufG = smem + 3 * TM * PA;                 // [TM][PA]: gated activation
    float* const prm = (float*)(bufG + TM * PA);           // b1 (2C) | b2 (2C) | e_n (C) | e_{n+1} (C)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TM;
    const bf16_t* const yb = y + (size_t)b * Tn * C;
#ifdef ADF_WN_STAMP
    const bool stamped = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2;
#endif
    WN_STAMP(0);

    // ---- stage the three windows (zero outside the sample: the conv's padding of y) and the parameters --------------------
    // All loads of a thread are issued before its first LDS store (a load -> store loop pays the memory latency per trip:
    // 6 us of a 38 us tile in the first version of this kernel).
    constexpr int CPR = C / 8;
    constexpr int NST = 3 * TM * CPR / 512;
    {
        u32x4_t sv[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + k * 512;
            const int tap = idx / (TM * CPR), rem = idx - tap * TM * CPR;
            const int i = rem / CPR, cc = rem - i * CPR;
            const int t = t0 + i + (tap - 1) * dil;
            const bool in = t >= 0 && t < Tn;
            sv[k] = *(const u32x4_t*)(yb + (size_t)(in ? t : 0) * C + cc * 8);
            if (!in) sv[k] = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + k * 512;
            const int tap = idx / (TM * CPR), rem = idx - tap * TM * CPR;
            const int i = rem / CPR, cc = rem - i * CPR;
            *(u32x4_t*)(bufA + ((size_t)tap * TM + i) * PA + cc * 16) = sv[k];
        }
    }
    {   // every parameter load of a thread before its first LDS store (as load -> store per trip: 6 C / 512 serialised memory round trips per tile)
        constexpr int PT = (6 * C + 511) / 512;
        float pv[PT];
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int i = tid + k * 512;
            float v = 0.0f;
            if (i < 2 * C) v = b1[i];
            else if (i < 4 * C) v = b2[i - 2 * C];
            else if (i < 5 * C) v = e[(size_t)b * e_bstride + (size_t)n * C + (i - 4 * C)];
            else if (i < 6 * C && y_next) v = e[(size_t)b * e_bstride + (size_t)(n + 1) * C + (i - 5 * C)];
            pv[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) if (tid + k * 512 < 6 * C) prm[tid + k * 512] = pv[k];
    }
    WN_STAMP(1);
    __syncthreads();
    WN_STAMP(2);

    auto row_of = [&](int i, int q) __attribute__((always_inline)) -> int { return i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh; };
    const int col = wave * 32 + r;
    const int nb2[2] = {wave * 32, C + wave * 32};

    // ---- GEMM 1: gate columns and the matching filter columns; sigmoid * tanh -> bufG -------------------------------------
    {
        wn_f32x16_t acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;
        wn_gemm<C, TM, 2 * C, 2, 3>(bufA, w1, nb2, acc, r, hh);
        WN_STAMP(3);
        const float bg = prm[col], bf = prm[C + col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = wn_sigmoid(acc[0][i][q] + bg) * wn_tanh(acc[1][i][q] + bf);
                *(unsigned short*)(bufG + (size_t)row_of(i, q) * PA + col * 2) = f32_to_bf16_hw(v);
            }
    }
    WN_STAMP(4);
    __syncthreads();
    WN_STAMP(5);

    // ---- GEMM 2: residual columns -> centre window in place; skip columns -> fp32 sum in HBM ------------------------------
    // The skip tile is a read-modify-write of 64 KB per workgroup.  Its loads are issued BEFORE GEMM 2 (16-byte pieces, the
    // latency hidden behind the MFMAs) and the accumulators reach those pieces through LDS -- the two outer windows are dead
    // once every wave has left GEMM 1 -- so the stores are 16-byte and coalesced too.  (Element-wise `skip[..] += acc` from the
    // accumulator layout was a chain of 32 dependent HBM round trips per lane: 17 us of a 38 us tile.)
    constexpr int PS = C * 4 + 16;                         // fp32 row pitch of the skip tile in LDS
    static_assert((TM / 2) * PS <= TM * PA, "half the skip tile fits one dead window");
    constexpr int SCH = C / 4;                             // 16-byte chunks per fp32 row
    constexpr int NSK = TM * SCH / 512;
    auto srow = [&](int row) __attribute__((always_inline)) -> char* {
        return bufA + (row < TM / 2 ? (size_t)row * PS : (size_t)2 * TM * PA + (size_t)(row - TM / 2) * PS);
    };
    float* const sg = skip + ((size_t)b * Tn + t0) * C;
    u32x4_t sk[NSK];
#pragma unroll
    for (int k = 0; k < NSK; ++k) {
        const int idx = tid + k * 512;
        const int row = idx / SCH, cc = idx - row * SCH;
        const bool in = !first && t0 + row < Tn;
        sk[k] = *(const u32x4_t*)(sg + (size_t)(in ? row : 0) * C + cc * 4);
        if (!in) sk[k] = u32x4_t{0u, 0u, 0u, 0u};
    }
    {
        wn_f32x16_t acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;
        wn_gemm<C, TM, 2 * C, 2, 1>(bufG, w2, nb2, acc, r, hh);
        WN_STAMP(6);
        const float br = prm[2 * C + col], bs = prm[3 * C + col], en = prm[4 * C + col], en1 = prm[5 * C + col];
        char* const ctr = bufA + (size_t)TM * PA;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row_of(i, q);
                if (y_next) {
                    unsigned short* const p = (unsigned short*)(ctr + (size_t)row * PA + col * 2);
                    const float yo = bf16_to_f32(*p);
                    *p = f32_to_bf16_hw(((yo - en) + (acc[0][i][q] + br)) * 0.70710678118654752440f + en1);
                }
                *(float*)(srow(row) + col * 4) = acc[1][i][q] + bs;
            }
    }
    WN_STAMP(7);
    __syncthreads();
    WN_STAMP(8);
#pragma unroll
    for (int k = 0; k < NSK; ++k) {
        const int idx = tid + k * 512;
        const int row = idx / SCH, cc = idx - row * SCH;
        if (t0 + row < Tn) {
            const f32x4_vec a = *(const f32x4_vec*)(srow(row) + cc * 16);
            const f32x4_vec o = __builtin_bit_cast(f32x4_vec, sk[k]);
            *(f32x4_vec*)(sg + (size_t)row * C + cc * 4) = a + o;
        }
    }
    if (y_next) {
        bf16_t* const ob = y_next + ((size_t)b * Tn + t0) * C;
#pragma unroll
        for (int k = 0; k < TM * CPR / 512; ++k) {
            const int idx = tid + k * 512;
            const int i = idx / CPR, cc = idx - i * CPR;
            if (t0 + i < Tn) *(u32x4_t*)(ob + (size_t)i * C + cc * 8) = *(const u32x4_t*)(bufA + ((size_t)TM + i) * PA + cc * 16);
        }
    }
    WN_STAMP(9);
}

// The wide variant: 128 positions per workgroup (still eight waves; a wave's 32 gate + 32 filter columns x 128 rows are 2 x 4
// accumulator tiles = 128 registers).  A weight fragment now feeds four row tiles instead of two, which halves the L2 -> CU
// weight stream per position: the 64-position kernel spends its GEMM phases AT that stream's limit (768 KB per tile at
// ~34 B/clk, tools/wn_stamps.py).  (A four-wave version with 4 x 4 tiles per wave in the accumulation registers was tried
// first: hipcc spilled accumulators inside the K loop.)
// LDS holds two 128-row buffers: taps 0 and 1 of the dilated conv are staged into X and G, tap 2 waits in registers and
// replaces tap 0 once every wave has left the first two taps; the gated tile then replaces tap 1; after GEMM 2 both buffers
// together are the fp32 exchange tile through which the residual and the skip accumulators reach 16-byte global pieces.
template <int C, int TM, int CW = 32>
__global__ void __launch_bounds__(C / CW * 64, CW == 64 ? 2 : 1) wn_layer_bf16_wide_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ y_next, float* __restrict__ skip,
                                                                 const void* __restrict__ w1, const float* __restrict__ b1,
                                                                 const void* __restrict__ w2, const float* __restrict__ b2,
                                                                 const float* __restrict__ e, int e_bstride, int n, int dil, int first, int Tn, int pf_stride) {
    using Tl = WnTile<C, TM>;
    constexpr int PA = Tl::PA, MT = Tl::MT, SPT = Tl::SPT;
    constexpr int NT = C / CW * 64;                        // threads: C / CW waves, each CW gate columns and the CW matching filter columns
    constexpr int NG = CW / 32, NN = 2 * NG;               // column tiles of a wave: NG gate (residual) tiles, then NG filter (skip) tiles
    static_assert((C == 256 || C == 128 || C == 64) && (TM == 128 || TM == 64) && (CW == 32 || CW == 64) && C % CW == 0 && NN * (TM / 32) == 8,
                  "a wave per CW gate columns; two TM-row buffers; eight accumulator tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufX = smem;                               // [TM][PA]: tap 0, then tap 2
    char* const bufG = smem + TM * PA;                     // [TM][PA]: tap 1, then the gated activation
    float* const prm = (float*)(smem + 2 * TM * PA);       // b1 (2C) | b2 (2C) | e_n (C) | e_{n+1} (C)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // XCD-contiguous tile order (adf_xcd_tile, as the conv2d kernels): an XCD's L2 sees a run of consecutive tiles of the same samples, whose dilated
    // windows are each other's centre rows.  Measured twice: before the non-temporal streams 1531 -> 1518 ms per configs[4] step with the PMC traffic of a
    // launch UP (11.6 -> 12.2 GB: not kept then); with them 1499.6 -> 1486.4 ms (three A/B pairs) and 10.8 GB per launch (11.1 without): kept.
#if ADF_WN_XCD
    const unsigned ltile = adf_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = (int)(ltile / gridDim.x), t0 = (int)(ltile % gridDim.x) * TM;
    const unsigned lin_next = blockIdx.y * gridDim.x + blockIdx.x + (unsigned)pf_stride;
    const unsigned ltile_next = adf_xcd_tile(lin_next, gridDim.x * gridDim.y);
#else
    const unsigned lin_next = blockIdx.y * gridDim.x + blockIdx.x + (unsigned)pf_stride;
    const unsigned ltile_next = lin_next;
    const int b = blockIdx.y, t0 = blockIdx.x * TM;
#endif
    // Global pieces are addressed as a wave-uniform base + a 32-bit per-lane byte offset derived from a freshly pinned thread id
    // in every phase: 64-bit per-lane pointers, which the compiler otherwise computes once for all phases and keeps (or
    // spills), would take ~200 of the 256 vector registers of this kernel.
    const char* const yb = (const char*)(y + (size_t)b * Tn * C);
    auto pin = [&]() __attribute__((always_inline)) -> int { int t = tid; asm volatile("" : "+v"(t)); return t; };

#ifdef ADF_WN_STAMP
    const bool stamped = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2;
#endif
    WN_STAMP(0);
    constexpr int CPR = C / 8;
    constexpr int NST = TM * CPR / NT;                     // 16-byte pieces of one window per thread
    auto row_of = [&](int i, int q) __attribute__((always_inline)) -> int { return i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh; };
    int nb2[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) nb2[j] = (j < NG ? 0 : C) + wave * CW + (j % NG) * 32;
    const int col0 = wave * CW + r;                        // column of this lane in column tile g: col0 + 32 g
    wn_f32x16_t acc[NN][MT];
    auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;
    };
    auto ldprm = [&]() __attribute__((always_inline)) {
        // every parameter load of a thread before its first LDS store (as load -> store per trip: 6 C / NT serialised memory round trips per tile)
        constexpr int PT = (6 * C + NT - 1) / NT;
        float pv[PT];
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int i = tid + k * NT;
            float v = 0.0f;
            if (i < 2 * C) v = b1[i];
            else if (i < 4 * C) v = b2[i - 2 * C];
            else if (i < 5 * C) v = e[(size_t)b * e_bstride + (size_t)n * C + (i - 4 * C)];
            else if (i < 6 * C && y_next) v = e[(size_t)b * e_bstride + (size_t)(n + 1) * C + (i - 5 * C)];
            pv[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) if (tid + k * NT < 6 * C) prm[tid + k * NT] = pv[k];
    };
    zero();
    if (2 * dil <= TM) {
        // ---- small dilation: the three windows overlap -- rows t0 - d .. t0 + TM + d are staged ONCE into X|G (<= 2 TM rows) and
        // tap k reads them from row k d on; one GEMM over all three taps -------------------------------------------------------
        u32x4_t wv[2 * NST];
        const int rows = TM + 2 * dil;
        {
            const int tq = pin();
#pragma unroll
            for (int k = 0; k < 2 * NST; ++k) {
                const int idx = tq + k * NT;
                const int i = idx / CPR, cc = idx - i * CPR;
                const int t = t0 - dil + i;
                const bool in = i < rows && t >= 0 && t < Tn;
                wv[k] = *(const u32x4_t*)(yb + (unsigned)(in ? t * (C * 2) + cc * 16 : 0));
                if (!in) wv[k] = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        ldprm();
#pragma unroll
        for (int k = 0; k < 2 * NST; ++k) {
            const int idx = tid + k * NT;
            const int i = idx / CPR, cc = idx - i * CPR;
            *(u32x4_t*)(bufX + (size_t)i * PA + cc * 16) = wv[k];
        }
        WN_STAMP(1);
        __syncthreads();
        WN_STAMP(2);
        wn_gemm<C, TM, 2 * C, NN, 3, 4>(bufX, w1, nb2, acc, r, hh, dil);
        __syncthreads();                                   // every wave has left the windows: G may be overwritten
    } else {
        // ---- large dilation: three disjoint windows.  Taps 0 and 1 go to X and G, tap 2 waits in registers and replaces tap 0
        // once every wave has left the first two taps ----------------------------------------------------------------------------
        u32x4_t w2v[NST];
        {
            u32x4_t w0v[NST], w1v[NST];
            auto ldwin = [&](int tap, u32x4_t (&v)[NST]) __attribute__((always_inline)) {
                const int tq = pin();
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                    const int idx = tq + k * NT;
                    const int i = idx / CPR, cc = idx - i * CPR;
                    const int t = t0 + i + (tap - 1) * dil;
                    const bool in = t >= 0 && t < Tn;
                    v[k] = *(const u32x4_t*)(yb + (unsigned)(in ? t * (C * 2) + cc * 16 : 0));
                    if (!in) v[k] = u32x4_t{0u, 0u, 0u, 0u};
                }
            };
            ldwin(0, w0v); ldwin(1, w1v); ldwin(2, w2v);
            ldprm();
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int idx = tid + k * NT;
                const int i = idx / CPR, cc = idx - i * CPR;
                *(u32x4_t*)(bufX + (size_t)i * PA + cc * 16) = w0v[k];
                *(u32x4_t*)(bufG + (size_t)i * PA + cc * 16) = w1v[k];
            }
        }
        WN_STAMP(1);
        __syncthreads();
        WN_STAMP(2);
        wn_gemm<C, TM, 2 * C, NN, 2, 4>(bufX, w1, nb2, acc, r, hh);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + k * NT;
            const int i = idx / CPR, cc = idx - i * CPR;
            *(u32x4_t*)(bufX + (size_t)i * PA + cc * 16) = w2v[k];
        }
        __syncthreads();
        wn_gemm<C, TM, 2 * C, NN, 1, 4>(bufX, (const char*)w1 + (size_t)2 * SPT * 2 * (2 * C) * 16, nb2, acc, r, hh);
        // (every wave has left tap 1 at the barrier above: G may be overwritten)
    }
    WN_STAMP(3);
    // gate: sigmoid(a) tanh(b) = (E - 1) / ((1 + exp(-a)) (1 + E)), E = exp(2 b) clamped so that (1 + E) stays finite: one
    // reciprocal per element instead of two
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int col = col0 + 32 * g;
        const float bg = prm[col], bf = prm[C + col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float ea = __builtin_amdgcn_exp2f((acc[g][i][q] + bg) * -1.4426950408889634f);
                const float eb = __builtin_amdgcn_exp2f(fminf((acc[NG + g][i][q] + bf) * 2.8853900817779268f, 60.0f));
                const float v = (eb - 1.0f) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
                *(unsigned short*)(bufG + (size_t)row_of(i, q) * PA + col * 2) = f32_to_bf16_hw(v);
            }
    }
    WN_STAMP(4);
    __syncthreads();
    WN_STAMP(5);

    // ---- GEMM 2 ----------------------------------------------------------------------------------------------------------
    zero();
    wn_gemm<C, TM, 2 * C, NN, 1, 4>(bufG, w2, nb2, acc, r, hh);
    asm volatile("" ::: "memory");                         // keep the epilogue's global loads below the GEMM (register pressure)
    WN_STAMP(6);

    // ---- epilogue: global pieces are issued first, the accumulators reach them through the fp32 exchange tile ---------------
    constexpr int PS = C * 4 + 16;
    static_assert(TM * PS <= 2 * TM * PA, "the exchange tile fits the two buffers");
    constexpr int SCH = C / 4;                             // 16-byte pieces per fp32 row
    constexpr int NSK = TM * SCH / NT;
    char* const sg = (char*)(skip + ((size_t)b * Tn + t0) * C);
    u32x4_t yo[NST];
    {
        const int tq = pin();
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tq + k * NT;
            const int i = idx / CPR, cc = idx - i * CPR;
            const bool in = y_next && t0 + i < Tn;
            yo[k] = *(const u32x4_t*)(yb + (unsigned)(in ? (t0 + i) * (C * 2) + cc * 16 : 0));
        }
    }
    u32x4_t sk[NSK];
    {
        const int tq = pin();
#pragma unroll
        for (int k = 0; k < NSK; ++k) {
            const int idx = tq + k * NT;
            const int row = idx / SCH, cc = idx - row * SCH;
            const bool in = !first && t0 + row < Tn;
            sk[k] = WN_NT_LOAD((const u32x4_t*)(sg + (unsigned)(in ? row * (C * 4) + cc * 16 : 0)));
            if (!in) sk[k] = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    // ---- L2 prefetch for the workgroup that follows this one on the XCD.  A tile's first 11-15 K cycles (of ~88 K) were the latency of its window loads
    // (tools/wn_stamps.py: entry -> "staging issued") with nothing else to run: one workgroup fills the CU.  Workgroups are handed to the XCDs round-robin,
    // so workgroup lin + pf_stride (pf_stride = resident workgroups of the grid, a multiple of 8) runs on THIS XCD next: one dword of each 128-byte line of
    // its windows is requested here, an epilogue (~19 K cycles) before it starts -- early enough to have landed in the XCD's L2, late enough not to be evicted
    // by the ~0.5 MB per tile that flow through it.  The loads are LDS-DMA (no destination registers) into 256 scratch bytes per wave behind the parameters
    // that nobody reads; the kernel waits for them before it ends (the LDS is the next workgroup's by then).
    constexpr int LPR = C * 2 / 128;                       // 128-byte lines per row
    constexpr int NPF = (3 * TM * LPR + NT - 1) / NT;
    if (pf_stride > 0 && lin_next < gridDim.x * gridDim.y) {
        const unsigned scratch = (unsigned)(2 * TM * PA + 6 * C * 4) + (unsigned)wave * 256u;
        const int bn = (int)(ltile_next / gridDim.x), tn0 = (int)(ltile_next % gridDim.x) * TM;
        const char* const ybn = (const char*)(y + (size_t)bn * Tn * C);
        const int tq = pin();
        const bool overlap = 2 * dil <= TM;
        const int nlines = (overlap ? TM + 2 * dil : 3 * TM) * LPR;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int li = tq + u * NT;
            const int row = li / LPR;
            const int t = overlap ? tn0 - dil + row : tn0 + (row % TM) + (row / TM - 1) * dil;
            if (li < nlines && t >= 0 && t < Tn) {
                const char* const ptr = ybn + (size_t)t * (C * 2) + (li % LPR) * 128;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(ptr), "s"(scratch) : "memory");
            }
        }
    }
    __syncthreads();                                       // every wave has left GEMM 2: X and G become the exchange tile
    auto put = [&](int part, int boff) __attribute__((always_inline)) {       // part 0: residual tiles, 1: skip tiles
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = col0 + 32 * g;
            const float bias = prm[boff + col];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) *(float*)(smem + (size_t)row_of(i, q) * PS + col * 4) = acc[part * NG + g][i][q] + bias;
        }
    };
    put(0, 2 * C);
    __syncthreads();
    if (y_next) {
        char* const ob = (char*)(y_next + ((size_t)b * Tn + t0) * C);
        const int tq = pin();
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tq + k * NT;
            const int i = idx / CPR, cc = idx - i * CPR;
            if (t0 + i < Tn) {
                float f[8];
                unpack16<bf16_t>(yo[k], f);
                const float* const rs = (const float*)(smem + (size_t)i * PS + cc * 32);
                const float* const en = prm + 4 * C + cc * 8;
                const float* const en1 = prm + 5 * C + cc * 8;
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = ((f[u] - en[u]) + rs[u]) * 0.70710678118654752440f + en1[u];
                WN_NT_STORE(pack16<bf16_t>(f), (u32x4_t*)(ob + (unsigned)(i * (C * 2) + cc * 16)));
            }

        }
    }
    __syncthreads();
    WN_STAMP(7);
    put(1, 3 * C);
    __syncthreads();
    WN_STAMP(8);
    {
        const int tq = pin();
#pragma unroll
        for (int k = 0; k < NSK; ++k) {
            const int idx = tq + k * NT;
            const int row = idx / SCH, cc = idx - row * SCH;
            if (t0 + row < Tn) {
                const f32x4_vec a = *(const f32x4_vec*)(smem + (size_t)row * PS + cc * 16);
                WN_NT_STORE(a + __builtin_bit_cast(f32x4_vec, sk[k]), (f32x4_vec*)(sg + (unsigned)(row * (C * 4) + cc * 16)));
            }

        }
    }
    WN_STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the prefetch DMAs have landed: the LDS may be handed on
}


const char* launch_wn_layer(const WnIO& io, const WnLayerArgs& a, hipStream_t s) {
    if (io.bf16) {
        // next-tile L2 prefetch of the 128-position kernels (ADF_WN_PREFETCH=1; off by default): the stride to the workgroup that follows on the same XCD = the
        // workgroups resident at once = one per CU (two for the paired 64-position form).  Measured (tools/calls/r04_call22.sh, profiles/r04_wavenet_layer_stamps.txt):
        // it does what it is for -- a tile's window-load latency drops from 11-15 K to 4-5 K cycles, the tile from 88.2 K to 80.3 K cycles at dilation 1 -- and the
        // launch takes the same 3.56-3.60 ms: the clock the chip holds falls from 2.13 to 1.92 GHz (0.47 -> 0.52 ns per cycle).  The layer kernel runs against the
        // power-management loop, not against its own stalls, exactly as the resblock conv kernel: removed wait cycles come back as frequency.
        static const int wn_pf = adf_route_switch("ADF_WN_PREFETCH", 0);
        static int num_cu_dev[kMaxDevices] = {};
        int& ncu = num_cu_dev[current_device()];
        if (ncu == 0 && (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || ncu < 8)) ncu = 256;
        int pf_stride = wn_pf ? ncu / 8 * 8 : 0;
        constexpr int C = 256, TM = 64, TMW = 128;
        if (io.C == 128 || io.C == 64) {                   // other widths: the 128-position kernel with C / 32 waves
            static bool attr_n[kMaxDevices] = {};
            bool& an = attr_n[current_device()];
            if (!an) {
                if (hipFuncSetAttribute((const void*)wn_layer_bf16_wide_kernel<128, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)wn_layer_bf16_wide_kernel<64, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return "wn_layer: hipFuncSetAttribute failed";
                an = true;
            }
            const dim3 grid(ceil_div(io.T, TMW), io.B);
            constexpr size_t lds128 = (size_t)2 * TMW * (128 * 2 + 16) + 6 * 128 * 4 + 2048, lds64 = (size_t)2 * TMW * (64 * 2 + 16) + 6 * 64 * 4 + 2048;
            if (io.C == 128)
                hipLaunchKernelGGL((wn_layer_bf16_wide_kernel<128, TMW>), grid, dim3(256), lds128, s, (const bf16_t*)a.y,
                                   (bf16_t*)a.y_next, a.skip, a.w1, a.b1, a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T, pf_stride);
            else
                hipLaunchKernelGGL((wn_layer_bf16_wide_kernel<64, TMW>), grid, dim3(128), lds64, s, (const bf16_t*)a.y,
                                   (bf16_t*)a.y_next, a.skip, a.w1, a.b1, a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T, pf_stride);
            return WN_LAUNCH_CHECK("wn_layer_bf16_wide");
        }
        if (io.C != C) return "WaveNet bf16 mode: the MFMA layer kernels are built for residual_channels = 64, 128 or 256 (use fp32 for other widths)";
        const size_t lds = (size_t)4 * TM * WnTile<C, TM>::PA + 6 * C * 4;
        static bool attr_done[kMaxDevices] = {};
        bool& attr = attr_done[current_device()];
        if (!attr) {
            if (hipFuncSetAttribute((const void*)wn_layer_bf16_kernel<C, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)wn_layer_bf16_wide_kernel<C, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return "wn_layer: hipFuncSetAttribute failed";
            attr = true;
        }
        // route switch (the parity tests run both): 1 = 128-position tiles, 0 = 64-position tiles
        static const int wide = adf_route_switch("ADF_WN_WIDE", 1);
        if (wide == 2) {
            // 64-position tiles on four waves (64 gate + 64 filter columns each): 74 KB of LDS, so a CU holds TWO workgroups and one's staging / gate /
            // epilogue runs under the other's GEMMs -- at twice the weight stream per position
            static bool attr2[kMaxDevices] = {};
            bool& a2 = attr2[current_device()];
            if (!a2) {
                if (hipFuncSetAttribute((const void*)wn_layer_bf16_wide_kernel<C, TM, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return "wn_layer: hipFuncSetAttribute failed";
                a2 = true;
            }
            const size_t lds2 = (size_t)2 * TM * WnTile<C, TM>::PA + 6 * C * 4 + 2048;
            hipLaunchKernelGGL((wn_layer_bf16_wide_kernel<C, TM, 64>), dim3(ceil_div(io.T, TM), io.B), dim3(256), lds2, s, (const bf16_t*)a.y,
                               (bf16_t*)a.y_next, a.skip, a.w1, a.b1, a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T, 2 * pf_stride);
            return WN_LAUNCH_CHECK("wn_layer_bf16_pair");
        }
        if (wide) {
            const size_t ldsw = (size_t)2 * TMW * WnTile<C, TMW>::PA + 6 * C * 4 + 2048;
            hipLaunchKernelGGL((wn_layer_bf16_wide_kernel<C, TMW>), dim3(ceil_div(io.T, TMW), io.B), dim3(512), ldsw, s, (const bf16_t*)a.y,
                               (bf16_t*)a.y_next, a.skip, a.w1, a.b1, a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T, pf_stride);
            return WN_LAUNCH_CHECK("wn_layer_bf16_wide");
        }
        hipLaunchKernelGGL((wn_layer_bf16_kernel<C, TM>), dim3(ceil_div(io.T, TM), io.B), dim3(512), lds, s, (const bf16_t*)a.y, (bf16_t*)a.y_next,
                           a.skip, a.w1, a.b1, a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T);
        return WN_LAUNCH_CHECK("wn_layer_bf16");
    }
    if (io.C % 32 || io.C > 512) return "WaveNet fp32 mode: residual_channels must be a multiple of 32, at most 512";
    const size_t lds = (size_t)4 * io.C * kWnTP * 4;
    static bool attr_done32[kMaxDevices] = {};
    bool& attr = attr_done32[current_device()];
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wn_layer_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "wn_layer: hipFuncSetAttribute failed";
        attr = true;
    }
    hipLaunchKernelGGL(wn_layer_f32_kernel, dim3(ceil_div(io.T, kWnTP), io.B), dim3(io.C), lds, s, (const float*)a.y, (float*)a.y_next, a.skip,
                       (const float*)a.w1, a.b1, (const float*)a.w2, a.b2, io.e, io.e_bstride, a.n, a.dilation, a.first, io.T, io.C);
    return WN_LAUNCH_CHECK("wn_layer_f32");
}

// ------------------------------------------------------------------------------------------------ skip projection + output conv
__device__ __forceinline__ float wn_edm_out(float F, int mode, const float* x_noisy, const float* coef, int coef_bstride, int b, size_t o) {
    if (mode == 0) return F;
    const float c_skip = coef[(size_t)b * coef_bstride + 2], c_out = coef[(size_t)b * coef_bstride + 3];
    return fminf(fmaxf(fmaf(c_out, F, c_skip * x_noisy[o]), -1.0f), 1.0f);
}

__global__ void __launch_bounds__(512) wn_final_f32_kernel(const float* __restrict__ skip, float scale, const float* __restrict__ w_sp,
                                                            const float* __restrict__ b_sp, const float* __restrict__ w_out,
                                                            const float* __restrict__ b_out, float* __restrict__ out, int mode,
                                                            const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                            int coef_bstride, int Tn, int C) {
    constexpr int TP = kWnTP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const St = (float*)smem;                    // [C][TP]: skip * scale, transposed
    float* const Pt = St + C * TP;                     // [C][TP]: w_out[c] * relu(sp[c])
    const int c = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * TP;
    for (int idx = c; idx < TP * C; idx += C) {
        const int i = idx / C, ch = idx - i * C;
        const int t = t0 + i;
        St[ch * TP + i] = t < Tn ? skip[((size_t)b * Tn + t) * C + ch] * scale : 0.0f;      // wavenet.py:152
    }
    __syncthreads();
    float acc[TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) acc[i] = b_sp[c];
    for (int k = 0; k < C; ++k) {
        const float w = w_sp[(size_t)k * C + c];
        const f32x4_vec* a4 = (const f32x4_vec*)(St + k * TP);
#pragma unroll
        for (int q = 0; q < TP / 4; ++q) {
            const f32x4_vec a = a4[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q * 4 + u] = fmaf(w, a[u], acc[q * 4 + u]);
        }
    }
    const float wo = w_out[c];
#pragma unroll
    for (int i = 0; i < TP; ++i) Pt[c * TP + i] = wo * fmaxf(acc[i], 0.0f);                  // :177-179
    __syncthreads();
    if (c < TP && t0 + c < Tn) {
        float F = b_out[0];
        for (int k = 0; k < C; ++k) F += Pt[k * TP + c];
        const size_t o = (size_t)b * Tn + t0 + c;
        out[o] = wn_edm_out(F, mode, x_noisy, coef, coef_bstride, b, o);
    }
}

template <int C, int TM>
__global__ void __launch_bounds__(2 * C) wn_final_bf16_kernel(const float* __restrict__ skip, float scale, const void* __restrict__ w_sp,
                                                            const float* __restrict__ b_sp, const float* __restrict__ w_out,
                                                            const float* __restrict__ b_out, float* __restrict__ out, int mode,
                                                            const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                            int coef_bstride, int Tn) {
    using Tl = WnTile<C, TM>;
    constexpr int PA = Tl::PA, MT = Tl::MT;
    static_assert(C == 256 || C == 128 || C == 64, "a wave per 32 columns");
    constexpr int NT = 2 * C, NW = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufA = smem;                               // [TM][PA]: bf16(skip * scale)
    float* const prm = (float*)(bufA + TM * PA);           // b_sp (C) | w_out (C)
    float* const red = prm + 2 * C;                        // [NW][TM] per-wave partial dot products
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TM;
    constexpr int CPR = C / 8;
    for (int idx = tid; idx < TM * CPR; idx += NT) {
        const int i = idx / CPR, cc = idx - i * CPR;
        const bool in = t0 + i < Tn;
        const float* src = skip + ((size_t)b * Tn + (in ? t0 + i : 0)) * C + cc * 8;
        float f[8];
        const u32x4_t lo = *(const u32x4_t*)src, hi = *(const u32x4_t*)(src + 4);
        unpack16<float>(lo, f); unpack16<float>(hi, f + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = in ? f[k] * scale : 0.0f;
        *(u32x4_t*)(bufA + (size_t)i * PA + cc * 16) = pack16<bf16_t>(f);
    }
    for (int i = tid; i < 2 * C; i += NT) prm[i] = i < C ? b_sp[i] : w_out[i - C];
    __syncthreads();
    const int col = wave * 32 + r;
    const int nb1[1] = {wave * 32};
    wn_f32x16_t acc[1][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][i][q] = 0.f;
    wn_gemm<C, TM, C, 1, 1, (C >= 128 ? 8 : 4)>(bufA, w_sp, nb1, acc, r, hh);
    const float bs = prm[col], wo = prm[C + col];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = wo * bf16_stored(fmaxf(acc[0][i][q] + bs, 0.0f));
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);       // over the 32 columns of this half-wave
            if (r == 0) red[wave * TM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh] = v;
        }
    __syncthreads();
    static_assert(TM <= NT, "one thread per position in the last step");
    if (tid < TM && t0 + tid < Tn) {
        float F = b_out[0];
#pragma unroll
        for (int w = 0; w < NW; ++w) F += red[w * TM + tid];
        const size_t o = (size_t)b * Tn + t0 + tid;
        out[o] = wn_edm_out(F, mode, x_noisy, coef, coef_bstride, b, o);
    }
}

const char* launch_wn_final(const WnIO& io, const WnFinalArgs& a, hipStream_t s) {
    if (io.bf16) {
        constexpr int C = 256, TM = 64;
        if (io.C == 128 || io.C == 64) {
            static bool attr_n[kMaxDevices] = {};
            bool& an = attr_n[current_device()];
            if (!an) {
                if (hipFuncSetAttribute((const void*)wn_final_bf16_kernel<128, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)wn_final_bf16_kernel<64, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return "wn_final: hipFuncSetAttribute failed";
                an = true;
            }
            const dim3 grid(ceil_div(io.T, TM), io.B);
            constexpr size_t lds128 = (size_t)TM * (128 * 2 + 16) + (2 * 128 + 4 * TM) * 4, lds64 = (size_t)TM * (64 * 2 + 16) + (2 * 64 + 2 * TM) * 4;
            if (io.C == 128)
                hipLaunchKernelGGL((wn_final_bf16_kernel<128, TM>), grid, dim3(256), lds128, s, a.skip, a.skip_scale, a.w_sp,
                                   a.b_sp, a.w_out, a.b_out, a.out, a.mode, a.x_noisy, a.coef, a.coef_bstride, io.T);
            else
                hipLaunchKernelGGL((wn_final_bf16_kernel<64, TM>), grid, dim3(128), lds64, s, a.skip, a.skip_scale, a.w_sp,
                                   a.b_sp, a.w_out, a.b_out, a.out, a.mode, a.x_noisy, a.coef, a.coef_bstride, io.T);
            return WN_LAUNCH_CHECK("wn_final_bf16");
        }
        if (io.C != C) return "WaveNet bf16 mode: built for residual_channels = 64, 128 or 256";
        const size_t lds = (size_t)TM * WnTile<C, TM>::PA + (2 * C + 8 * TM) * 4;
        static bool attr_done[kMaxDevices] = {};
        bool& attr = attr_done[current_device()];
        if (!attr) {
            if (hipFuncSetAttribute((const void*)wn_final_bf16_kernel<C, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return "wn_final: hipFuncSetAttribute failed";
            attr = true;
        }
        hipLaunchKernelGGL((wn_final_bf16_kernel<C, TM>), dim3(ceil_div(io.T, TM), io.B), dim3(512), lds, s, a.skip, a.skip_scale, a.w_sp, a.b_sp,
                           a.w_out, a.b_out, a.out, a.mode, a.x_noisy, a.coef, a.coef_bstride, io.T);
        return WN_LAUNCH_CHECK("wn_final_bf16");
    }
    if (io.C % 32 || io.C > 512) return "WaveNet fp32 mode: residual_channels must be a multiple of 32, at most 512";
    const size_t lds = (size_t)2 * io.C * kWnTP * 4;
    hipLaunchKernelGGL(wn_final_f32_kernel, dim3(ceil_div(io.T, kWnTP), io.B), dim3(io.C), lds, s, a.skip, a.skip_scale, (const float*)a.w_sp, a.b_sp,
                       a.w_out, a.b_out, a.out, a.mode, a.x_noisy, a.coef, a.coef_bstride, io.T, io.C);
    return WN_LAUNCH_CHECK("wn_final_f32");
}

}  // namespace adf
