// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the EDM sampling path.
// wave = 64 lanes, everything here is written for that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace adf {

// ---- element types ------------------------------------------------------------------
// Activations / GEMM weights are stored either as fp32 ("parity" mode) or bf16 ("throughput"
// mode); all accumulation, norms, softmax and sampler state are fp32 (stats in fp64).
struct bf16_t { uint16_t v; };
// native 16-byte vector (HIP's uint4 struct defeats SROA in unrolled staging arrays -> scratch)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_hw_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    // round-to-nearest-even; NaN stays NaN (quiet)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPerChunk = 4;   // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kPerChunk = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->v); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
};

// unpack one 16-byte chunk into floats / pack floats into a chunk
template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& q, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& q, float* f) {
    f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& q, float* f) {
    f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
    f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
    f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* f);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* f) {
    return u32x4_t{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}
// two fp32 -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN kept)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    f32x2_hw_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}
// one fp32 -> bf16 with the hardware converter (the software f32_to_bf16 above is ~8 vector instructions)
__device__ __forceinline__ unsigned short f32_to_bf16_hw(float f) { return (unsigned short)(pack_bf16x2(f, f) & 0xffffu); }
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* f) {
    u32x4_t q;
    q.x = pack_bf16x2(f[0], f[1]);
    q.y = pack_bf16x2(f[2], f[3]);
    q.z = pack_bf16x2(f[4], f[5]);
    q.w = pack_bf16x2(f[6], f[7]);
    return q;
}

// ---- split-bf16 ("f32x3") mode -------------------------------------------------------------------------------------------------------
// Storage is fp32 exactly as in the parity mode (Elem / unpack16 / pack16 below are the float ones); only the GEMM OPERANDS differ: a value x is staged
// as hi = rn_bf16(x) and lo = rn_bf16(x - hi) (x = hi + lo up to 2^-17 |x|) and a product a * b is taken as a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA
// with fp32 accumulation (the dropped a_lo b_lo term is 2^-18 of the product): ~1e-5 relative per dot product against the exact-fp32 chain, at a third of
// the bf16 MFMA rate instead of the sixteenth that v_mfma_f32_32x32x2f32 runs at.  A 128-byte K row of 32 fp32 becomes 64 B of hi + 64 B of lo:
// 16-byte slot s < 4 holds the hi parts of K = 8 s .. 8 s + 7 (the MFMA fragment of K step s >> 1, lane half s & 1), slot 4 + s the lo parts -- the
// SAME row pitch, LDS footprint and staging loop as the fp32 rows.  Weights are split once, at pack time (pack_weight_kernel<f32x3_t>).
struct f32x3_t { float v; };
template <> struct Elem<f32x3_t> {
    static constexpr int kPerChunk = 4;
    __device__ static __forceinline__ float ld(const f32x3_t* p) { return p->v; }
    __device__ static __forceinline__ void st(f32x3_t* p, float v) { p->v = v; }
};
template <> __device__ __forceinline__ void unpack16<f32x3_t>(const u32x4_t& q, float* f) { unpack16<float>(q, f); }
template <> __device__ __forceinline__ u32x4_t pack16<f32x3_t>(const float* f) { return pack16<float>(f); }
template <typename T> struct IsX3 { static constexpr bool value = false; };
template <> struct IsX3<f32x3_t> { static constexpr bool value = true; };
// four fp32 -> their hi parts (8 B) and lo parts (8 B)
__device__ __forceinline__ void split_bf16x4(const float* f, u32x2_t& hi, u32x2_t& lo) {
    hi.x = pack_bf16x2(f[0], f[1]);
    hi.y = pack_bf16x2(f[2], f[3]);
    lo.x = pack_bf16x2(f[0] - __uint_as_float(hi.x << 16), f[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(f[2] - __uint_as_float(hi.y << 16), f[3] - __uint_as_float(hi.y & 0xffff0000u));
}

// Pack a chunk for storing AND leave in f[] the values the tensor then holds: the GroupNorm statistics of a produced tensor are
// reduced from the STORED (bf16-rounded) values, i.e. they are the statistics of the tensor the next layer reads -- what the
// reference's GroupNorm computes on its input -- not of the fp32 accumulators behind it.  fp32 storage: values unchanged.
template <typename T> __device__ __forceinline__ u32x4_t pack16_stored(float* f) {
    const u32x4_t q = pack16<T>(f);
    if constexpr (sizeof(T) == 2) unpack16<T>(q, f);
    return q;
}
__device__ __forceinline__ float bf16_stored(float v) { return bf16_to_f32(f32_to_bf16_hw(v)); }

// GroupNorm(+FiLM) folded to a per-(sample, channel) affine y = A*x + Bc of the (possibly concatenated) input
// [src0 ; scale1*src1]: statistics [B][G][2] (sum, sumsq in fp64) per source tensor, gamma / beta, optional FiLM
// (scale + 1, shift) with an optional second addend.  Shared by gn_finalize_kernel, gn_norm_apply_kernel and the GEMM
// kernels that derive their affine table themselves.
struct GnFinalizeArgs {
    const double* stats0; const double* stats1;  // [B][G][2] per source tensor
    int c0, c1, L, G, B;
    float scale1, eps;
    const float* gamma; const float* beta;        // [c0+c1]
    const float* film; int film_bstride;           // film[b*bstride + c] = scale, film[b*bstride + ctot + c] = shift
    const float* film2; int film2_bstride;         // optional second addend (class-embedding part of the projection)
    float* ab;                                     // [B][c0+c1][2] (gn_finalize only)
};
// The computation is split in two so that a kernel can issue the loads, do other work (its first DMAs), and finish later:
// gn_affine_load only loads, gn_affine_finish only computes.
struct GnRaw {
    double sum, sq;          // statistics of the (coarse) group, summed over its stored fine groups
    float gamma, beta, fs, fh;
};
// EVERY load below is unconditional and independent of the others (optional tensors are read through a valid dummy pointer and discarded): a load
// under a condition, or a select on a loaded value, makes the compiler wait `vmcnt(0)` on the spot, and the statistics -> gamma / beta -> FiLM ->
// class FiLM chain was four to five serialised memory round trips at the head of every kernel that derives its table (ISA of conv_gemm_rb_kernel,
// round 2).  The statistics pointer is formed as stats0 + a selected byte distance: a per-lane select between two kernel-argument pointers is
// otherwise compiled to a load of the pointer itself from the kernarg segment -- one more dependent round trip.
__device__ __forceinline__ const double* gn_select_ptr(bool second, const double* p0, const double* p1) {
    // p0 + (second ? p1 - p0 : 0): arithmetic on p0 keeps the global address space (a select of two integers cast back would make flat loads)
    long long d = (const char*)p1 - (const char*)p0;
    asm volatile("" : "+s"(d));
    return (const double*)((const char*)p0 + (second ? d : 0ll));
}
__device__ __forceinline__ GnRaw gn_affine_load(const GnFinalizeArgs& a, int b, int c) {
    const int ctot = a.c0 + a.c1;
    const int gs = ctot / a.G;
    const int cstart = (c / gs) * gs;
    const bool s1 = cstart >= a.c0;
    const double* st = gn_select_ptr(s1, a.stats0, a.stats1);
    const int csrc = s1 ? a.c1 : a.c0;
    const int lc = s1 ? cstart - a.c0 : cstart;
    const int fg = csrc / a.G;                // channels per stored (fine) group
    const int g0 = lc / fg, g1 = (lc + gs + fg - 1) / fg;
    GnRaw r;
    // one or two fine groups per coarse group in every shape served (one source, or two equal ones): both loads go out together; more in a loop
    const double* p0 = st + ((size_t)b * a.G + g0) * 2;
    const double* p1 = g0 + 1 < g1 ? p0 + 2 : p0;
    const double s0 = p0[0], q0 = p0[1], s1v = p1[0], q1v = p1[1];
    const float gamma = a.gamma[c], beta = a.beta[c];
    const bool hf = a.film != nullptr, hf2 = hf && a.film2 != nullptr;                 // uniform
    int cd = c;                                        // opaque copy of the index: gamma[cd] through the dummy pointer must not be folded into the gamma load
    asm volatile("" : "+v"(cd));                       // above (a copy of that register is a wait for it, and the optional loads end up behind a branch again)
    const float* const f1 = hf ? a.film + (size_t)b * a.film_bstride : a.gamma;
    const float* const f2 = hf2 ? a.film2 + (size_t)b * a.film2_bstride : a.gamma;
    const float f1s = f1[cd], f1h = f1[hf ? ctot + cd : cd], f2s = f2[cd], f2h = f2[hf2 ? ctot + cd : cd];
    r.sum = s0 + (g0 + 1 < g1 ? s1v : 0.0); r.sq = q0 + (g0 + 1 < g1 ? q1v : 0.0);
    for (int g = g0 + 2; g < g1; ++g) { r.sum += st[((size_t)b * a.G + g) * 2]; r.sq += st[((size_t)b * a.G + g) * 2 + 1]; }
    r.gamma = gamma;
    r.beta = beta;
    r.fs = 1.0f; r.fh = 0.0f;
    if (hf) {
        r.fs = f1s + 1.0f;
        r.fh = f1h;
        if (hf2) { r.fs += f2s; r.fh += f2h; }   // class-embedding part of the FiLM projection (precomputed per sampler run)
    }
    return r;
}
// FAST (bf16 throughput mode, inside a GEMM kernel's start-up path): no fp64 division / square root -- the variance is still
// formed in fp64 (cancellation), the reciprocal square root in fp32 with one Newton step (~1e-7 relative).
template <bool FAST = false>
__device__ __forceinline__ void gn_affine_finish(const GnFinalizeArgs& a, int c, const GnRaw& r, float& A, float& Bc) {
    const int ctot = a.c0 + a.c1;
    const int gs = ctot / a.G;
    const bool s1 = (c / gs) * gs >= a.c0;
    const double sc = s1 ? (double)a.scale1 : 1.0;
    const double sum = r.sum * sc, sq = r.sq * sc * sc;
    float rstd, meanf;
    if constexpr (FAST) {
        const double inv_cnt = (double)(1.0f / ((float)a.L * (float)gs));     // L * gs is a small power-of-two multiple: exact in fp32 for the shapes served
        const double mean = sum * inv_cnt;
        double var = sq * inv_cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float vf = (float)var + a.eps;
        float q = __builtin_amdgcn_rsqf(vf);
        q = q * (1.5f - 0.5f * vf * q * q);
        rstd = q; meanf = (float)mean;
    } else {
        const double cnt = (double)a.L * (double)gs;
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        meanf = (float)mean;
    }
    A = rstd * r.gamma;
    Bc = r.beta - meanf * A;
    if (a.film) {
        A *= r.fs;
        Bc = fmaf(Bc, r.fs, r.fh);
    }
    if (s1) A *= a.scale1;
}
template <bool FAST = false>
__device__ __forceinline__ void gn_affine(const GnFinalizeArgs& a, int b, int c, float& A, float& Bc) {
    gn_affine_finish<FAST>(a, c, gn_affine_load(a, b, c), A, Bc);
}

// SiLU = v * sigmoid(v) with the hardware exp2 / rcp (each ~1 ulp): 5 VALU ops, 2 of them transcendental
__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int kRowBytes = 128;      // bytes of K per LDS row / per packed weight row
constexpr int kRowBytesPack = kRowBytes;

// Launch-side state that HIP keeps per device (kernel attributes, CU count, cached device buffers) is indexed by the
// CURRENT device: the C entry points make the handle's device current before anything is launched (adf_api.hip DeviceScope).
constexpr int kMaxDevices = 64;
static inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

// Run-time switches.  ROUTE switches (which of two equivalent kernel routes a layer takes) are read from the environment
// once per process: the parity tests run both sides of each (ADF_GEMM_PP / _RB / _UP / _WS, ADF_RB_FUSED, ADF_TR_FUSED, and
// ADF_GEMM_TRACE, which prints the route of every launch).  TUNING thresholds are compile-time constants in the product
// build; a build with -DADF_EXPERIMENTS reads them from the environment too (A/B runs inside one gpurun call).
static inline int adf_route_switch(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline long long adf_tuning(const char* name, long long dflt) {
#ifdef ADF_EXPERIMENTS
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// XCD-aware tile order.  Workgroups are handed to the 8 XCDs (each with its own L2) round-robin in linear grid order, so tiles that are neighbours in
// the grid -- and share halo rows / dilated windows / an activation tile read by two N tiles -- land in eight different L2s.  With this map every XCD
// works through ONE contiguous range of logical tiles instead: lin = the workgroup's linear id (x fastest), total = workgroups of the grid.
#ifndef ADF_XCD_ORDER
#define ADF_XCD_ORDER 1
#endif
__device__ __forceinline__ unsigned adf_xcd_tile(unsigned lin, unsigned total) {
    return (!ADF_XCD_ORDER || total % 8u) ? lin : (lin % 8u) * (total / 8u) + lin / 8u;
}

}  // namespace adf
