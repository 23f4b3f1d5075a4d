// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the EDM sampling path.
// wave = 64 lanes, everything here is written for that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adf {

// ---- element types ------------------------------------------------------------------
// Activations / GEMM weights are stored either as fp32 ("parity" mode) or bf16 ("throughput"
// mode); all accumulation, norms, softmax and sampler state are fp32 (stats in fp64).
struct bf16_t { uint16_t v; };
// native 16-byte vector (HIP's uint4 struct defeats SROA in unrolled staging arrays -> scratch)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

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
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* f) {
    u32x4_t q;
    q.x = pack_bf16x2(f[0], f[1]);
    q.y = pack_bf16x2(f[2], f[3]);
    q.z = pack_bf16x2(f[4], f[5]);
    q.w = pack_bf16x2(f[6], f[7]);
    return q;
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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace adf
