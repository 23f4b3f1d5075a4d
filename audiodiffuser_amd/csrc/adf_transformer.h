// One launch for a whole TransformerBlock1d (unet1d.py:63-121, attention_utils.py:84-182, FeedForward1d unet1d.py:40-61)
// on the short levels of the net (bf16 throughput mode, 16 or 64 tokens per sample, 256 channels, 8 heads of 32,
// feed-forward width 512): LayerNorm -> q|k|v projection -> attention -> output projection + residual -> LayerNorm1d ->
// 1x1 conv + GELU -> LayerNorm1d -> 1x1 conv + residual, plus the GroupNorm statistics of the result.
//
// Why: at these levels the nine launches of the unfused path (3 x ln_rows, 4 GEMMs, attention, gn_stats) take 60-100 us per
// block (tools/layer_table.py: the GEMMs run at 20-60 TF/s, every launch pays 4-7 us of start-up) -- 0.43 ms of a 3.7 ms
// network pass -- for 17-67 MFLOP per sample.  Here one 512-thread workgroup owns one sample:
//   * activations never leave LDS (row pitch +16 B: conflict-free ds_read_b128 fragments); they are rounded to bf16
//     at exactly the points where the unfused path stores a bf16 tensor, so both paths agree to rounding noise;
//   * every GEMM splits its OUTPUT COLUMNS over the 8 waves, so a wave needs only its own columns of the weights and
//     reads them straight from the packed global layout into MFMA B fragments (each weight byte is fetched once per
//     workgroup, 1 MB in all, 3-8 K steps ahead of use); the A fragments come from LDS;
//   * attention: wave = head; S^T = K Q^T on MFMA with the softmax lane-local, P^T reused as the B operand of
//     O^T = V^T P^T (the scheme of attention_mfma32_kernel, with q / k / v read from LDS);
//   * the residual of the feed-forward stays in the registers of the wave that owns the same 32 columns in both GEMMs;
//   * GroupNorm statistics of the output: a group of 32 channels is one wave's columns -> no atomics.
// The floor is the weight stream: 1 MB per workgroup at ~34 B/clk from L2 = ~15 us.
#pragma once
#include "adf_common.h"
#include <type_traits>

namespace adf {

struct TrFusedArgs {
    const bf16_t* x;      // [B][NTOK][256]
    bf16_t* out;          // [B][NTOK][256]
    const float* ln_w; const float* ln_b;    // nn.LayerNorm(256) before the attention
    const float* g0;                          // LayerNorm1d gain (256) before the first 1x1 conv
    const float* g3;                          // LayerNorm1d gain (512) before the second
    const void* wqkv; const void* wproj; const void* wff1; const void* wff2;   // fragment-major bf16 [K/16][2][n_pad][8] (repack_frag_kernel)
    int npad_qkv, npad_proj, npad_ff1, npad_ff2;
    double* stats;        // [B][8][2] sum / sumsq of the output per group of 32 channels, or nullptr
    float eps;
    const bf16_t* att;    // MODE 2: attention output [rows][256]
    bf16_t* qkv_out;      // MODE 1: q|k|v [rows][768]
    int tiles_per_sample; // MODE 1 / 2: workgroups (64-row tiles) per sample
    int B;                // samples (MODE 0) or row tiles = worker workgroups; the grid may carry helper workgroups beyond (see the kernel)
};

typedef __attribute__((ext_vector_type(8))) __bf16 tr_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float tr_f32x16_t;

// MODE 0: the whole block, one workgroup per sample (16 or 64 tokens).  Longer samples (256 tokens) run as two launches around
// the stand-alone attention kernel, a workgroup per 64-row tile (everything but the attention is row-local):
// MODE 1: LayerNorm -> q|k|v projection -> a.qkv_out;  MODE 2: attention output a.att -> projection + residual -> ... -> out,
// statistics by fp64 atomics (tiles_per_sample workgroups share a sample).
template <int NTOK, int MODE = 0>
__global__ void __launch_bounds__(512) transformer_small_kernel(const TrFusedArgs a) {
    constexpr int C = 256, MID = 512, D = 32;
    constexpr int MR = NTOK < 32 ? 32 : NTOK;           // rows of the MFMA tiles (rows >= NTOK are padding)
    constexpr int MT = MR / 32;
    constexpr int PA = C * 2 + 16;                      // LDS row pitches (bytes)
    constexpr int PQ = 3 * C * 2 + 16;
    constexpr int PF = MID * 2 + 16;
    static_assert(MR * PF <= MR * PQ, "the feed-forward buffer reuses the q|k|v buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufA = smem;                            // [MR][PA]: xn -> att -> x1 -> n1 -> x2
    char* const bufQ = smem + MR * PA;                  // [MR][PQ]: q|k|v, then [MR][PF]: f1 -> n2
    float* const prm = (float*)(smem + MR * PA + MR * PQ + 8 * 1024);   // LayerNorm parameters: ln_w | ln_b | g0 | g3 (after the helpers' scratch)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x;
    const bf16_t* const xb = a.x + (size_t)(b < a.B ? b : 0) * NTOK * C;
    // ---- L2 warm-up by helper workgroups.  The block's 1 MB of weights is cold in L2 when it starts and a CU draws only ~11
    // B/clk of L2 misses however much it keeps in flight.  The grid therefore carries 3 helper workgroups per sample
    // (b >= a.B; they occupy the CUs this launch would leave idle): workgroups b, b + 8, b + 16 ... usually share an XCD
    // (round-robin dispatch; only speed depends on it), and each helper pulls a different 1/(3B/8) of every weight array
    // towards its XCD's L2 by LDS-DMA into a scratch KB per wave (no registers, nobody reads it) and exits, while the
    // workers are in their first LayerNorm: their fragment loads then find the lines in L2 (~34 B/clk).
    if (b >= a.B) {
        const int hidx = b - a.B;
        const int per_xcd = (int)(gridDim.x - a.B) / 8 > 0 ? (int)(gridDim.x - a.B) / 8 : 1;
        const int slice = (hidx >> 3) % per_xcd;
        const unsigned scratch = (unsigned)(MR * PA + MR * PQ) + (unsigned)wave * 1024u;
        auto warm = [&](const void* W, int n_pad, int chunks) __attribute__((always_inline)) {
            const unsigned total = (unsigned)chunks * (unsigned)n_pad * 128u;          // bytes of the array (a multiple of 1 KB)
            const unsigned per = ((total / 1024u + per_xcd - 1) / per_xcd) * 1024u;      // bytes of one slice
            const unsigned lo = (unsigned)slice * per, hi = lo + per < total ? lo + per : total;
            for (unsigned off = lo + (unsigned)wave * 1024u; off < hi; off += 8u * 1024u) {
                const char* g = (const char*)W + off + lane * 16;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(scratch) : "memory");
            }
        };
        if (MODE != 2) warm(a.wqkv, a.npad_qkv, 4);
        if (MODE != 1) {
            warm(a.wproj, a.npad_proj, 4);
            warm(a.wff1, a.npad_ff1, 4);
            warm(a.wff2, a.npad_ff2, 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- row LayerNorm (the arithmetic of ln_rows_kernel: two-pass variance on registers), TPR threads per row ----
    constexpr int TPR = 512 / NTOK;                     // 8 (64 tokens) or 32 (16 tokens)
    const int lrow = tid / TPR, lsub = tid % TPR;
    auto layer_norm = [&](auto cwc, const char* src, int src_pitch, bool src_global, char* dst, int dst_pitch, const float* gamma,
                          const float* beta, bool sync_params) __attribute__((always_inline)) {
        constexpr int CW = decltype(cwc)::value;        // channels of the row
        constexpr int NCH = CW / 8 / TPR;               // 16-byte chunks per thread
        float f[NCH][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int cc = lsub + k * TPR;
            const u32x4_t v = src_global ? *(const u32x4_t*)(src + (size_t)lrow * src_pitch + cc * 16)
                                         : *(const u32x4_t*)(src + lrow * src_pitch + cc * 16);
            unpack16<bf16_t>(v, f[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[k][e];
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.0f / (float)CW);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; sq = fmaf(d, d, sq); }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq * (1.0f / (float)CW) + a.eps);
        if (sync_params) __syncthreads();                // the parameter stores to LDS above (uniform flag)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int cc = lsub + k * TPR;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f[k][e] - mean) * rstd * gamma[cc * 8 + e] + (beta ? beta[cc * 8 + e] : 0.f);   // (LDS)
            *(u32x4_t*)(dst + lrow * dst_pitch + cc * 16) = pack16<bf16_t>(o);
        }
    };

    // ---- one GEMM stage: acc[i][j] += A[rows of M tile i][K] * W[columns n0 + 32 j ..][K]^T, A from LDS, W fragments from
    // the fragment-major global weights.  The ring of fragments lives across the stages: `prefetch` issues the first DEPTH K
    // steps (16 KB per wave) BEFORE the phase that precedes the GEMM (LayerNorm, attention, the previous epilogue), so the
    // weight stream -- the floor of this kernel -- keeps running through the vector-only phases.
    // fragment-major weights (repack_frag_kernel): the 16-byte fragments of K step ks, half hh, for all columns are
    // contiguous, so a wave's load is two 512-byte runs.  (Read from the conv layout -- 128-byte rows -- every lane is
    // its own 64-byte request: 64 requests per KB held all four GEMMs at 9-14 B/clk/CU whatever was in flight.)
    tr_bf16x8_t wf[21];                                  // max over NT of (DEPTH + 1) * NT: 17, 18, 21
    auto depth_of = [](int nt) constexpr -> int { return nt == 1 ? 16 : (nt == 2 ? 8 : 6); };
    auto prefetch = [&](auto ntc, const void* W, int n_pad, int n0) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value, DEPTH = depth_of(NT);
        const char* const wl = (const char*)W + ((size_t)hh * n_pad + n0 + r) * 16;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[d * NT + j] = __builtin_bit_cast(tr_bf16x8_t, *(const u32x4_t*)(wl + ((size_t)d * 2 * n_pad + j * 32) * 16));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gemm = [&](auto ntc, auto ksc, const char* A, int pitch, const void* W, int n_pad, int n0, tr_f32x16_t (&acc)[MT][decltype(ntc)::value])
                    __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value, KS = decltype(ksc)::value;
        constexpr int DEPTH = depth_of(NT), RING = DEPTH + 1;
        static_assert(KS >= DEPTH, "ring depth");
        const char* const wl = (const char*)W + ((size_t)hh * n_pad + n0 + r) * 16;
        auto wfrag = [&](int ks, int j) __attribute__((always_inline)) -> tr_bf16x8_t {
            return __builtin_bit_cast(tr_bf16x8_t, *(const u32x4_t*)(wl + ((size_t)ks * 2 * n_pad + j * 32) * 16));
        };
        __builtin_amdgcn_sched_barrier(0);
        tr_bf16x8_t af[2][MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = *(const tr_bf16x8_t*)(A + (i * 32 + r) * pitch + hh * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + DEPTH < KS) {
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[((ks + DEPTH) % RING) * NT + j] = wfrag(ks + DEPTH, j);
            }
            if (ks + 1 < KS) {                           // A fragments one K step ahead
#pragma unroll
                for (int i = 0; i < MT; ++i) af[(ks + 1) & 1][i] = *(const tr_bf16x8_t*)(A + (i * 32 + r) * pitch + (ks + 1) * 32 + hh * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], wf[(ks % RING) * NT + j], acc[i][j], 0, 0, 0);
        }
    };
    // GELU with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result): a fifth of
    // the instructions of erff(), which made this epilogue cost as much as the GEMM before it
    auto gelu_fast = [&](float v) __attribute__((always_inline)) -> float {
        const float x = fabsf(v) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
        const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
        const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
        return 0.5f * v * (1.0f + copysignf(erf_abs, v));
    };
    auto zero = [&](auto& acc) __attribute__((always_inline)) {
        for (auto& row : acc)
            for (auto& t : row)
#pragma unroll
                for (int e = 0; e < 16; ++e) t[e] = 0.f;
    };
    // accumulator element e of lane (r, hh): row (e & 3) + 8 (e >> 2) + 4 hh of the tile, column r
    auto row_of = [&](int i, int e) __attribute__((always_inline)) -> int { return i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh; };
    const std::integral_constant<int, 1> nt1{};
    const std::integral_constant<int, 2> nt2{};
    const std::integral_constant<int, 3> nt3{};
    const std::integral_constant<int, 16> ks16{};
    const std::integral_constant<int, 32> ks32{};

    if constexpr (MODE != 2) prefetch(nt3, a.wqkv, a.npad_qkv, wave * 96);
    else prefetch(nt1, a.wproj, a.npad_proj, wave * 32);
    // LayerNorm parameters -> LDS, their loads in flight together with the input rows' (one round trip instead of three)
    // (three unconditional loads per thread, the vector chosen by a wave-uniform select: the loop with a branch per vector was waited for trip by trip)
    {
        const bool lo = wave < 4;                                                     // uniform: threads 0..255 / 256..511
        const int c = tid & 255;
        float v0 = (lo ? a.ln_w : a.ln_b)[c], v1 = (lo ? a.g0 : a.g3)[c], v2 = a.g3[256 + c];
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2));       // all three are loaded HERE (the compiler otherwise sinks v2's load into the `if (lo)` below)
        prm[tid] = v0;                    // ln_w | ln_b
        prm[512 + tid] = v1;              // g0 | g3[0..255]
        if (lo) prm[1024 + c] = v2;       // g3[256..511]
    }
    if constexpr (MODE != 2) {
    // ---- S0: LayerNorm of the input rows -> bufA ------------------------------------------------------------------
    layer_norm(std::integral_constant<int, C>{}, (const char*)xb, C * 2, true, bufA, PA, prm, prm + 256, true);
    __syncthreads();
    // ---- S1: q | k | v = xn W^T (768 columns, 96 per wave) -> bufQ -------------------------------------------------
    {
        tr_f32x16_t acc[MT][3];
        zero(acc);
        gemm(nt3, ks16, bufA, PA, a.wqkv, a.npad_qkv, wave * 96, acc);
        if constexpr (MODE == 0) prefetch(nt1, a.wproj, a.npad_proj, wave * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    *(unsigned short*)(bufQ + row_of(i, e) * PQ + (wave * 96 + j * 32 + r) * 2) = f32_to_bf16_hw(acc[i][j][e]);
    }
    __syncthreads();
    }
    if constexpr (MODE == 1) {                          // q|k|v rows -> global, done
        bf16_t* const qo = a.qkv_out + (size_t)b * NTOK * 3 * C;
        for (int idx = tid; idx < NTOK * (3 * C / 8); idx += 512) {
            const int row = idx / (3 * C / 8), cc = idx % (3 * C / 8);
            *(u32x4_t*)(qo + (size_t)row * 3 * C + cc * 8) = *(const u32x4_t*)(bufQ + row * PQ + cc * 16);
        }
        return;
    }
    if constexpr (MODE == 2) {                          // attention output rows -> bufA
        const bf16_t* const ab = a.att + (size_t)b * NTOK * C;
        // (every load of a thread before its first store: as load -> store per trip these were NTOK * C / 8 / 512 serialised round trips)
        constexpr int AT = NTOK * (C / 8) / 512 > 0 ? NTOK * (C / 8) / 512 : 1;
        u32x4_t av[AT];
#pragma unroll
        for (int k = 0; k < AT; ++k) {
            const int idx = tid + k * 512;
            const int row = idx < NTOK * (C / 8) ? idx / (C / 8) : 0, cc = idx % (C / 8);
            av[k] = *(const u32x4_t*)(ab + (size_t)row * C + cc * 8);
        }
#pragma unroll
        for (int k = 0; k < AT; ++k) {
            const int idx = tid + k * 512;
            if (idx < NTOK * (C / 8)) *(u32x4_t*)(bufA + (idx / (C / 8)) * PA + (idx % (C / 8)) * 16) = av[k];
        }
        __syncthreads();
    }
    if constexpr (MODE == 0) {
    // ---- S2: attention, wave = head; output rows -> bufA ---------------------------------------------------------
    {
        const char* const qb = bufQ + wave * D * 2;
        const char* const kb = qb + C * 2;
        const char* const vb = qb + 2 * C * 2;
        const float scale = 0.17677669529663687f;       // 32^-1/2
#pragma unroll 1
        for (int qt = 0; qt < MT; ++qt) {
            const int query = qt * 32 + r;
            tr_bf16x8_t qf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const tr_bf16x8_t*)(qb + query * PQ + ks * 32 + hh * 16);
            tr_f32x16_t o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
            float m = -INFINITY, l = 0.f;
#pragma unroll 1
            for (int kt = 0; kt < MT; ++kt) {
                tr_f32x16_t st;
#pragma unroll
                for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const tr_bf16x8_t kf = *(const tr_bf16x8_t*)(kb + (kt * 32 + r) * PQ + ks * 32 + hh * 16);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st, 0, 0, 0);
                }
                float mt = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const float sv = key < NTOK ? st[e] * scale : -INFINITY;
                    st[e] = sv;
                    mt = fmaxf(mt, sv);
                }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float mn = fmaxf(m, mt);
                const float alpha = __expf(m - mn);
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float pv = __expf(st[e] - mn); st[e] = pv; psum += pv; }
                l = l * alpha + psum;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] *= alpha;
                m = mn;
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    u32x4_t pw, vw;
                    pw.x = pack_bf16x2(st[8 * sgrp + 0], st[8 * sgrp + 1]);
                    pw.y = pack_bf16x2(st[8 * sgrp + 2], st[8 * sgrp + 3]);
                    pw.z = pack_bf16x2(st[8 * sgrp + 4], st[8 * sgrp + 5]);
                    pw.w = pack_bf16x2(st[8 * sgrp + 6], st[8 * sgrp + 7]);
                    unsigned short ve[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        int key = kt * 32 + 16 * sgrp + 8 * (j >> 2) + 4 * hh + (j & 3);
                        key = key < NTOK ? key : NTOK - 1;  // its probability is 0; keep the value finite
                        ve[j] = *(const unsigned short*)(vb + key * PQ + r * 2);
                    }
                    vw.x = (unsigned)ve[0] | ((unsigned)ve[1] << 16);
                    vw.y = (unsigned)ve[2] | ((unsigned)ve[3] << 16);
                    vw.z = (unsigned)ve[4] | ((unsigned)ve[5] << 16);
                    vw.w = (unsigned)ve[6] | ((unsigned)ve[7] << 16);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tr_bf16x8_t, vw), __builtin_bit_cast(tr_bf16x8_t, pw), o, 0, 0, 0);
                }
            }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            if (query < NTOK) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {            // registers 4g .. 4g+3 = head dims 8g + 4hh .. +3 of this query
                    uint2 w;
                    w.x = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv);
                    w.y = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                    *(uint2*)(bufA + query * PA + (wave * D + 8 * g + 4 * hh) * 2) = w;
                }
            }
        }
    }
    __syncthreads();
    }
    // ---- S3: x1 = att Wp^T + x (32 columns per wave); kept (rounded to bf16, as the unfused path stores it) for S7 -----
    float x1r[MT][16];
    {
        tr_f32x16_t acc[MT][1];
        zero(acc);
        gemm(nt1, ks16, bufA, PA, a.wproj, a.npad_proj, wave * 32, acc);
        prefetch(nt2, a.wff1, a.npad_ff1, wave * 64);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_of(i, e);
                const float res = row < NTOK ? bf16_to_f32(xb[(size_t)row * C + wave * 32 + r].v) : 0.f;
                x1r[i][e] = bf16_to_f32(f32_to_bf16_hw(acc[i][0][e] + res));
            }
    }
    __syncthreads();                                    // every wave is done reading att
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) *(unsigned short*)(bufA + row_of(i, e) * PA + (wave * 32 + r) * 2) = f32_to_bf16_hw(x1r[i][e]);
    __syncthreads();
    // ---- S4: n1 = LayerNorm1d(x1), in place ----------------------------------------------------------------------
    layer_norm(std::integral_constant<int, C>{}, bufA, PA, false, bufA, PA, prm + 512, nullptr, false);
    __syncthreads();
    // ---- S5: f1 = gelu(n1 W1^T) (512 columns, 64 per wave) -> bufQ as [MR][512] -----------------------------------
    {
        tr_f32x16_t acc[MT][2];
        zero(acc);
        gemm(nt2, ks16, bufA, PA, a.wff1, a.npad_ff1, wave * 64, acc);
        prefetch(nt1, a.wff2, a.npad_ff2, wave * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    *(unsigned short*)(bufQ + row_of(i, e) * PF + (wave * 64 + j * 32 + r) * 2) = f32_to_bf16_hw(gelu_fast(acc[i][j][e]));
    }
    __syncthreads();
    // ---- S6: n2 = LayerNorm1d(f1), in place ----------------------------------------------------------------------
    layer_norm(std::integral_constant<int, MID>{}, bufQ, PF, false, bufQ, PF, prm + 768, nullptr, false);
    __syncthreads();
    // ---- S7: x2 = n2 W2^T + x1 -> bufA -> global; GroupNorm statistics of x2 -------------------------------------
    {
        tr_f32x16_t acc[MT][1];
        zero(acc);
        gemm(nt1, ks32, bufQ, PF, a.wff2, a.npad_ff2, wave * 32, acc);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_of(i, e);
                const unsigned short q = f32_to_bf16_hw(acc[i][0][e] + x1r[i][e]);
                *(unsigned short*)(bufA + row * PA + (wave * 32 + r) * 2) = q;
                if (row < NTOK) { const float v = bf16_to_f32(q); s1 += v; s2 = fmaf(v, v, s2); }
            }
        if (a.stats) {                                   // group `wave` = this wave's 32 columns, all rows of the sample
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (lane == 0) {
                if constexpr (MODE == 0) {
                    double* sp = a.stats + ((size_t)b * 8 + wave) * 2;
                    sp[0] = (double)s1; sp[1] = (double)s2;
                } else {
                    double* sp = a.stats + ((size_t)(b / a.tiles_per_sample) * 8 + wave) * 2;
                    atomicAdd(sp, (double)s1);
                    atomicAdd(sp + 1, (double)s2);
                }
            }
        }
    }
    __syncthreads();
    {
        bf16_t* const ob = a.out + (size_t)b * NTOK * C;
        for (int idx = tid; idx < NTOK * (C / 8); idx += 512) {
            const int row = idx / (C / 8), cc = idx % (C / 8);
            *(u32x4_t*)(ob + (size_t)row * C + cc * 8) = *(const u32x4_t*)(bufA + row * PA + cc * 16);
        }
    }
}

inline const char* launch_transformer_small(const TrFusedArgs& a, int B, int ntok, hipStream_t s) {
    const int mr = ntok < 32 ? 32 : ntok;
    const size_t lds = (size_t)mr * (256 * 2 + 16) + (size_t)mr * (768 * 2 + 16) + 8 * 1024 + 1280 * 4;   // + the warm-up scratch + LayerNorm parameters
    if (ntok != 64 && ntok != 16) return "transformer_small: 16 or 64 tokens per sample";
    static bool attr_done[kMaxDevices] = {};
    bool& attr = attr_done[current_device()];
    if (!attr) {                                         // both instances need more than the default 64 KB of dynamic LDS
        if (hipFuncSetAttribute((const void*)transformer_small_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)transformer_small_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "transformer_small: hipFuncSetAttribute failed";
        attr = true;
    }
    // 3 helper workgroups per sample, as far as otherwise idle CUs exist (256 CUs, one workgroup each)
    const int helpers = B < 256 ? ((256 - B) / 8 > 3 * B / 8 ? 3 * B / 8 : (256 - B) / 8) * 8 : 0;
    TrFusedArgs aa = a;
    aa.B = B;
    if (ntok == 64) hipLaunchKernelGGL(transformer_small_kernel<64>, dim3(B + helpers), dim3(512), lds, s, aa);
    else hipLaunchKernelGGL(transformer_small_kernel<16>, dim3(B + helpers), dim3(512), lds, s, aa);
    return hipGetLastError() == hipSuccess ? nullptr : "transformer_small: launch failed";
}

// 256-token (or longer) samples: MODE 1 / MODE 2 over 64-row tiles (rows = B * tokens, a multiple of 64)
inline const char* launch_transformer_tiles(const TrFusedArgs& a, int rows, int tokens, int mode, hipStream_t s) {
    if (rows % 64 || tokens % 64 || (mode != 1 && mode != 2)) return "transformer_tiles: unsupported shape";
    static bool attr_done[kMaxDevices] = {};
    bool& attr = attr_done[current_device()];
    if (!attr) {
        if (hipFuncSetAttribute((const void*)transformer_small_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)transformer_small_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "transformer_tiles: hipFuncSetAttribute failed";
        attr = true;
    }
    const size_t lds = (size_t)64 * (256 * 2 + 16) + (size_t)64 * (768 * 2 + 16) + 8 * 1024 + 1280 * 4;
    const int tiles = rows / 64;
    const int helpers = tiles < 256 ? ((256 - tiles) / 8 > 3 * tiles / 8 ? 3 * tiles / 8 : (256 - tiles) / 8) * 8 : 0;
    TrFusedArgs aa = a;
    aa.B = tiles;
    aa.tiles_per_sample = tokens / 64;
    if (mode == 1) hipLaunchKernelGGL((transformer_small_kernel<64, 1>), dim3(tiles + helpers), dim3(512), lds, s, aa);
    else hipLaunchKernelGGL((transformer_small_kernel<64, 2>), dim3(tiles + helpers), dim3(512), lds, s, aa);
    return hipGetLastError() == hipSuccess ? nullptr : "transformer_tiles: launch failed";
}

}  // namespace adf
