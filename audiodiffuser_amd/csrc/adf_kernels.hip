// Non-GEMM kernels of the EDM sampling path: norms, attention, waveform in/out transforms with the
// EDM preconditioning fused, sigma embedding, sampler state updates, weight packing.
// All are bandwidth- or latency-bound; they use 16-byte accesses along the contiguous channel axis
// of the NLC activations (or the sample axis of the fp32 waveform) and 64-lane wave reductions.
#include "adf_kernels.h"

namespace adf {

#define ADF_LAUNCH_CHECK(name) (hipGetLastError() == hipSuccess ? nullptr : name ": launch failed")

// =====================================================================================================
// GroupNorm statistics (reference: torch.nn.GroupNorm inside ConvBlock1d, unet1d.py:178-182,198)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x, int L, int C, int G, int rows_per_block,
                                                       double* __restrict__ stats) {
    constexpr int EPC = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* csum = (float*)smem;       // [C]
    float* csq = csum + C;            // [C]
    const int b = blockIdx.y;
    const int cpr = C / EPC;          // 16-byte chunks per row (power of two, <= 256)
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * C; i += 256) csum[i] = 0.f;
    __syncthreads();
    const int cc = tid % cpr, rsub = tid / cpr, rstep = 256 / cpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(L, r0 + rows_per_block);
    float s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const T* base = x + (size_t)b * L * C + (size_t)cc * EPC;
    for (int r = r0 + rsub; r < r1; r += rstep) {
        const u32x4_t v = *(const u32x4_t*)(base + (size_t)r * C);
        float f[EPC];
        unpack16<T>(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        atomicAdd(&csum[cc * EPC + e], s[e]);
        atomicAdd(&csq[cc * EPC + e], q[e]);
    }
    __syncthreads();
    const int gs = C / G;
    if (tid < G) {
        double a = 0.0, c = 0.0;
        for (int i = 0; i < gs; ++i) { a += (double)csum[tid * gs + i]; c += (double)csq[tid * gs + i]; }
        atomicAdd(&stats[((size_t)b * G + tid) * 2], a);
        atomicAdd(&stats[((size_t)b * G + tid) * 2 + 1], c);
    }
}

const char* launch_gn_stats(const void* x, int bf16, int B, int L, int C, int G, double* stats, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc) return "gn_stats: C must be a multiple of a 16-byte chunk";
    const int cpr = C / epc;
    if (cpr > 256 || (cpr & (cpr - 1))) return "gn_stats: C/chunk must be a power of two <= 256";
    if (C % G || G > 256) return "gn_stats: bad group count";
    const int rstep = 256 / cpr;
    int rows_per_block = rstep * 16;
    while ((long long)ceil_div(L, rows_per_block) * B > 8192) rows_per_block *= 2;
    dim3 grid(ceil_div(L, rows_per_block), B);
    const size_t lds = (size_t)2 * C * sizeof(float);
    if (bf16) hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)x, L, C, G, rows_per_block, stats);
    else hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), lds, s, (const float*)x, L, C, G, rows_per_block, stats);
    return ADF_LAUNCH_CHECK("gn_stats");
}

// GroupNorm apply + FiLM (unet1d.py:160-161, 198-200) folded to y = a*x + b per (sample, channel).
__global__ void __launch_bounds__(256) gn_finalize_kernel(const GnFinalizeArgs a) {
    const int b = blockIdx.x;
    const int ctot = a.c0 + a.c1;
    for (int c = threadIdx.x; c < ctot; c += 256) {
        float A, Bc;
        gn_affine(a, b, c, A, Bc);
        float* o = a.ab + ((size_t)b * ctot + c) * 2;
        o[0] = A; o[1] = Bc;
    }
}

const char* launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s) {
    const int ctot = a.c0 + a.c1;
    if (ctot % a.G) return "gn_finalize: channels not divisible by groups";
    const int gs = ctot / a.G;
    if (a.c0 % gs) return "gn_finalize: a group straddles the two concatenated sources";
    if (a.c0 % a.G || (a.c1 && a.c1 % a.G)) return "gn_finalize: source channels not divisible by groups";
    if (gs % (a.c0 / a.G) || (a.c1 && gs % (a.c1 / a.G))) return "gn_finalize: group size not a multiple of the stored group size";
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B), dim3(256), 0, s, a);
    return ADF_LAUNCH_CHECK("gn_finalize");
}

// Materialise silu(a*x + b) of a (possibly concatenated) input as one NLC tensor.  Used for the short
// (L <= 64) levels, where the GEMM runs on flat multi-sample tiles that take raw inputs only.
template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ s0, const T* __restrict__ s1, int c0, int c1, int L,
                                                       const float* __restrict__ ab, int act, T* __restrict__ out) {
    constexpr int EPC = Elem<T>::kPerChunk;
    const int b = blockIdx.y;
    const int ctot = c0 + c1, cpr = ctot / EPC;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)L * cpr) return;
    const int cc = (int)(idx % cpr);
    const int l = (int)(idx / cpr);
    const int c = cc * EPC;
    const T* src = c < c0 ? s0 + ((size_t)b * L + l) * c0 + c : s1 + ((size_t)b * L + l) * c1 + (c - c0);
    float f[EPC];
    unpack16<T>(*(const u32x4_t*)src, f);
    const float* abp = ab + ((size_t)b * ctot + c) * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float v = fmaf(f[e], abp[2 * e], abp[2 * e + 1]);
        f[e] = act ? silu_f(v) : v;
    }
    *(u32x4_t*)(out + ((size_t)b * L + l) * ctot + c) = pack16<T>(f);
}

// gn_finalize + gn_apply in one launch for the short levels: every thread derives the affine of its own
// channels straight from the statistics (the work is tiny, the point is one launch less per GroupNorm).
template <typename T>
__global__ void __launch_bounds__(256) gn_norm_apply_kernel(const T* __restrict__ s0, const T* __restrict__ s1, const GnFinalizeArgs a,
                                                            int act, T* __restrict__ out) {
    constexpr int EPC = Elem<T>::kPerChunk;
    const int b = blockIdx.y;
    const int ctot = a.c0 + a.c1, cpr = ctot / EPC;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.L * cpr) return;
    const int cc = (int)(idx % cpr);
    const int l = (int)(idx / cpr);
    const int c = cc * EPC;
    const bool from1 = c >= a.c0;
    const T* src = from1 ? s1 + ((size_t)b * a.L + l) * a.c1 + (c - a.c0) : s0 + ((size_t)b * a.L + l) * a.c0 + c;
    float f[EPC];
    unpack16<T>(*(const u32x4_t*)src, f);
    const int gs = ctot / a.G;
    const double* st = from1 ? a.stats1 : a.stats0;
    const int csrc = from1 ? a.c1 : a.c0;
    const int fg = csrc / a.G;
    const double sc = from1 ? (double)a.scale1 : 1.0;
    const double cnt = (double)a.L * (double)gs;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int ch = c + e;
        const int cstart = (ch / gs) * gs;
        const int lc = from1 ? cstart - a.c0 : cstart;
        const int g0 = lc / fg, g1 = (lc + gs + fg - 1) / fg;
        double sum = 0.0, sq = 0.0;
        for (int g = g0; g < g1; ++g) { sum += st[((size_t)b * a.G + g) * 2]; sq += st[((size_t)b * a.G + g) * 2 + 1]; }
        sum *= sc; sq *= sc * sc;
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        float A = rstd * a.gamma[ch];
        float Bc = a.beta[ch] - (float)mean * A;
        if (a.film) {
            float fs = a.film[(size_t)b * a.film_bstride + ch] + 1.0f;
            float fh = a.film[(size_t)b * a.film_bstride + ctot + ch];
            if (a.film2) {
                fs += a.film2[(size_t)b * a.film2_bstride + ch];
                fh += a.film2[(size_t)b * a.film2_bstride + ctot + ch];
            }
            A *= fs;
            Bc = fmaf(Bc, fs, fh);
        }
        if (from1) A *= a.scale1;
        const float v = fmaf(f[e], A, Bc);
        f[e] = act ? silu_f(v) : v;
    }
    *(u32x4_t*)(out + ((size_t)b * a.L + l) * ctot + c) = pack16<T>(f);
}

const char* launch_gn_norm_apply(const void* s0, const void* s1, const GnFinalizeArgs& a, int act, void* out, int bf16, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    const int ctot = a.c0 + a.c1;
    if (a.c0 % epc || a.c1 % epc) return "gn_norm_apply: channel counts must be multiples of a 16-byte chunk";
    if (ctot % a.G) return "gn_norm_apply: channels not divisible by groups";
    const int gs = ctot / a.G;
    if (a.c0 % gs) return "gn_norm_apply: a group straddles the two concatenated sources";
    if (a.c0 % a.G || (a.c1 && a.c1 % a.G)) return "gn_norm_apply: source channels not divisible by groups";
    if (gs % (a.c0 / a.G) || (a.c1 && gs % (a.c1 / a.G))) return "gn_norm_apply: group size not a multiple of the stored group size";
    const long long work = (long long)a.L * (ctot / epc);
    dim3 grid((unsigned)((work + 255) / 256), a.B);
    if (bf16) hipLaunchKernelGGL(gn_norm_apply_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)s0, (const bf16_t*)s1, a, act, (bf16_t*)out);
    else hipLaunchKernelGGL(gn_norm_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)s0, (const float*)s1, a, act, (float*)out);
    return ADF_LAUNCH_CHECK("gn_norm_apply");
}

const char* launch_gn_apply(const void* s0, const void* s1, int c0, int c1, int L, int B, const float* ab, int act, void* out,
                            int bf16, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (c0 % epc || c1 % epc) return "gn_apply: channel counts must be multiples of a 16-byte chunk";
    const long long work = (long long)L * ((c0 + c1) / epc);
    dim3 grid((unsigned)((work + 255) / 256), B);
    if (bf16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)s0, (const bf16_t*)s1, c0, c1, L, ab, act, (bf16_t*)out);
    else hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)s0, (const float*)s1, c0, c1, L, ab, act, (float*)out);
    return ADF_LAUNCH_CHECK("gn_apply");
}

// =====================================================================================================
// Row LayerNorm (nn.LayerNorm in TransformerBlock1d unet1d.py:80,111; LayerNorm1d :31-43 in NLC)
// =====================================================================================================
// LPR lanes share a row (so short rows do not idle half the wave): a wave normalises 64/LPR rows at once, a thread
// holds CPL 16-byte chunks of its row in registers; two-pass variance on the registers (no E[x^2] - mean^2).
// The grid is capped and every wave walks its rows with a stride: gamma / beta of the lane's channels are loaded once.
template <typename T, int LPR, int CPL>
__global__ void __launch_bounds__(256) ln_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;                         // lane inside its row group
    const int cpr = C / EPC;
    float gm[CPL][EPC], bt[CPL][EPC];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int cc = sub + k * LPR;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            gm[k][e] = cc < cpr ? gamma[cc * EPC + e] : 0.f;
            bt[k][e] = (cc < cpr && beta) ? beta[cc * EPC + e] : 0.f;
        }
    }
    const float inv_c = 1.0f / (float)C;
    const long long stride = (long long)gridDim.x * 4 * RPW;
    for (long long rb = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; rb < rows; rb += stride) {   // wave-uniform trip count
        const long long row = rb + lane / LPR;
        const bool live = row < rows;
        const long long r = live ? row : rows - 1;      // out-of-range groups recompute the last row and do not store
        float f[CPL][EPC];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int cc = sub + k * LPR;
            if (cc < cpr) {
                const u32x4_t v = *(const u32x4_t*)(x + r * C + (size_t)cc * EPC);
                unpack16<T>(v, f[k]);
#pragma unroll
                for (int e = 0; e < EPC; ++e) sum += f[k][e];
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int cc = sub + k * LPR;
            if (cc < cpr) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float d = f[k][e] - mean; sq = fmaf(d, d, sq); }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq * inv_c + eps);
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int cc = sub + k * LPR;
            if (cc < cpr && live) {
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = (f[k][e] - mean) * rstd * gm[k][e] + bt[k][e];
                *(u32x4_t*)(y + r * C + (size_t)cc * EPC) = pack16<T>(o);
            }
        }
    }
}

template <typename T>
static const char* ln_rows_dispatch(const T* x, T* y, long long rows, int C, const float* gamma, const float* beta, float eps, hipStream_t s) {
    const int cpr = C / Elem<T>::kPerChunk;
#define ADF_LN(LPR, CPL)                                                                                                          \
    do {                                                                                                                          \
        long long grid = (rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR));                                                          \
        if (grid > 4096) grid = 4096;                                                                                            \
        hipLaunchKernelGGL((ln_rows_kernel<T, LPR, CPL>), dim3((unsigned)grid), dim3(256), 0, s, x, y, rows, C, gamma, beta, eps); \
        return ADF_LAUNCH_CHECK("ln_rows");                                                                                       \
    } while (0)
    if (cpr <= 8) ADF_LN(8, 1);
    if (cpr <= 16) ADF_LN(16, 1);
    if (cpr <= 32) ADF_LN(32, 1);
    if (cpr <= 64) ADF_LN(64, 1);
    if (cpr <= 128) ADF_LN(64, 2);
    if (cpr <= 256) ADF_LN(64, 4);
#undef ADF_LN
    return "ln_rows: C too large";
}

const char* launch_ln_rows(const void* x, void* y, int bf16, long long rows, int C, const float* gamma,
                           const float* beta, float eps, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc) return "ln_rows: C must be a multiple of a 16-byte chunk";
    if (rows < 1) return "ln_rows: no rows";
    if (bf16) return ln_rows_dispatch<bf16_t>((const bf16_t*)x, (bf16_t*)y, rows, C, gamma, beta, eps, s);
    return ln_rows_dispatch<float>((const float*)x, (float*)y, rows, C, gamma, beta, eps, s);
}

// =====================================================================================================
// Self-attention (attention_utils.py:160-182): one query row per lane, online softmax in fp32.
// K/V rows are wave-broadcast loads.  (Round-1 VALU version; MFMA version is the next step.)
// =====================================================================================================
template <typename T, int DH>
__global__ void __launch_bounds__(256) attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int B, int N, int C,
                                                        int heads, float scale) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int NCH = DH / EPC;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * heads * N;
    if (gid >= total) return;
    const int qi = (int)(gid % N);
    const int hh = (int)((gid / N) % heads);
    const int b = (int)(gid / ((long long)N * heads));
    const size_t rowstride = (size_t)3 * C;
    const T* base = qkv + (size_t)b * N * rowstride + (size_t)hh * DH;
    float q[DH], o[DH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const u32x4_t v = *(const u32x4_t*)(base + (size_t)qi * rowstride + k * EPC);
        unpack16<T>(v, q + k * EPC);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] *= scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < N; ++j) {
        const T* kr = base + (size_t)j * rowstride + C;
        float sdot = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            float kf[EPC];
            unpack16<T>(*(const u32x4_t*)(kr + k * EPC), kf);
#pragma unroll
            for (int e = 0; e < EPC; ++e) sdot = fmaf(q[k * EPC + e], kf[e], sdot);
        }
        const float mn = fmaxf(m, sdot);
        const float alpha = __expf(m - mn);
        const float p = __expf(sdot - mn);
        l = l * alpha + p;
        const T* vr = kr + C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            float vf[EPC];
            unpack16<T>(*(const u32x4_t*)(vr + k * EPC), vf);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[k * EPC + e] = fmaf(o[k * EPC + e], alpha, p * vf[e]);
        }
        m = mn;
    }
    const float inv = 1.0f / l;
    T* orow = out + ((size_t)b * N + qi) * C + (size_t)hh * DH;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        float t[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) t[e] = o[k * EPC + e] * inv;
        *(u32x4_t*)(orow + k * EPC) = pack16<T>(t);
    }
}

// MFMA self-attention for bf16, head dim 32 (attention_utils.py:160-182).  One wave = 32 queries of one
// (batch, head) pair.  S^T = K Q^T is computed (keys on the accumulator rows, the query on the lane), so the
// softmax over keys is lane-local: 16 registers + one xor-32 exchange.  The P^T accumulator tile is converted
// to bf16 in place and used directly as the B operand of O^T += V^T P^T (the k order inside a step is
// 16s + 8(j>>2) + 4h + (j&3)).
// Round 2 (profiles/r01_attention_pmc.txt: 67 vector instructions per MFMA, 9.3 % MFMA utilisation at N = 1024):
//   * V is staged TRANSPOSED, vt[d][key] with a row pitch of 2 N + 8 bytes (lane = head dim: 32 rows on 32 different
//     bank pairs), so a V^T fragment is two 8-byte reads (keys 4h .. 4h+3 and 8 + 4h .. 8 + 4h+3 of the 16-key step)
//     instead of sixteen 2-byte gathers and their shifts / ors;
//   * the scores stay unscaled in the accumulator: p = exp2(s * c - m) with c = d^-1/2 log2(e) folded into ONE fma per
//     score (the running maximum lives in that scaled log2 domain), v_exp_f32 directly instead of expf;
//   * the accumulator rescale (16 multiplies + the alpha arithmetic) is skipped when no lane's maximum grew in the tile
//     (wave vote; alpha would be exactly 1, so the results are bit for bit the same);
//   * the key-range mask is only applied in the last, partial key tile.
typedef __attribute__((ext_vector_type(8))) __bf16 att_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float att_f32x16_t;

//   * a wave walks `qrep` query tiles of its pair one after the other: the transposed copy of V is staged once per
//     wpp * qrep query tiles (at N = 1024 the staging was ~1/6 of a block's time with one tile per wave).
// Round 4 (profiles/r04_attention_mfma_utilisation.txt: 42.5 vector instructions per MFMA, 11.8 % at N = 1024; the ISA of the key-tile loop had 136 per tile of 4 MFMAs):
//   * two waves per SIMD asked of the register allocator (__launch_bounds__(256, 2)): with a 512-register budget hipcc keeps MFMA results in AccVGPRs and every
//     key tile paid 16 v_accvgpr_read for the scores plus 16 for the output accumulator (hoisted above the rarely-taken rescale branch);
//   * the tile maximum over the SCALED scores (8 v_pk_mul_f32, then 8 v_max3_f32: a product is a canonical value, so fmaxf needs no v_max(x, x) in front of
//     every score -- 31 instructions before; the maximum of the scaled scores is the scaled maximum, bit for bit), scale / subtract and the row sum as
//     packed-fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: 8 + 9 instead of 16 + 17).  (v_max3_f32 by inline asm on the accumulators themselves is NOT safe: the
//     hazard recognizer does not see an asm operand, and the branch into the block skipped the 11 wait states a VALU read of an MFMA result needs --
//     two runs of one launch differed), K rows addressed as a uniform base + a 32-bit lane offset (3 instead of 9), the two halves of a
//     query's keys exchanged by v_permlane32_swap_b32 instead of ds_bpermute_b32 + its wait.
typedef __attribute__((ext_vector_type(2))) float att_f32x2_t;
//   * NW = 8 waves per workgroup sharing one pair's V^T (N >= 256 at head dim 32): 16 waves per CU instead of 8 under the same LDS footprint, four per SIMD
//     to fill each other's transcendental / MFMA latencies (the loop is bound by its vector instructions: 58 + 16 v_exp_f32 per key tile of 4 MFMAs).
template <int D, int NW = 4>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) attention_mfma32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int B, int N,
                                                                int C, int heads, float scale_log2e, int qrep) {
    constexpr int DT = D / 32, DK = D / 16;      // output tiles of 32 head dims, K steps of the score GEMM (head dim 32 or 64)
    constexpr bool kSplitP = D == 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qtiles = (N + 31) >> 5;
    const int wpp = qtiles >= NW ? NW : (qtiles >= 4 ? 4 : (qtiles >= 2 ? 2 : 1));   // waves that share one (b, head) pair
    const int ppb = NW / wpp;                                   // pairs per block
    const int qgroups = (qtiles + wpp * qrep - 1) / (wpp * qrep);
    const int qg = blockIdx.x % qgroups, pg = blockIdx.x / qgroups;
    const int pair = pg * ppb + wave / wpp;
    const bool pair_ok = pair < B * heads;
    const int pc = pair_ok ? pair : 0;
    const int b = pc / heads, hd = pc - b * heads;
    const size_t rowstride = (size_t)3 * C;
    const bf16_t* base = qkv + (size_t)b * N * rowstride + (size_t)hd * D;
    // ---- V of this pair -> LDS transposed: vt[d][key], keys padded to a multiple of 32, staged by the wpp waves of the pair
    const int npad = qtiles * 32;
    const int pitch = npad * 2 + 8;
    char* vt = smem + (size_t)(wave / wpp) * D * pitch;
    {
        const int tl = (wave % wpp) * 64 + lane, nthr = wpp * 64;
        const bf16_t* vb = base + 2 * C;
        // eight pieces per thread requested before the first is scattered (one load -> eight 2-byte stores per trip was 16 serialised memory round trips
        // per block at N = 1024: about as long as the block's 64 key-tile loops)
        const int total = npad * (D / 8);
        for (int base = tl; base < total; base += 8 * nthr) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * nthr;
                const int key = idx / (D / 8), c = idx % (D / 8);
                const int krow = (idx < total && key < N) ? key : N - 1;    // padded keys: any finite value (their probability is 0)
                v[u] = *(const u32x4_t*)(vb + (size_t)krow * rowstride + c * 8);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * nthr;
                if (idx < total) {
                    const int key = idx / (D / 8), c = idx % (D / 8);
                    const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        *(unsigned short*)(vt + (c * 8 + j) * pitch + key * 2) = (unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
                }
            }
        }
    }
    __syncthreads();
    if (!pair_ok) return;
    const bf16_t* kb = base + C;
    const char* vrow = vt + r * pitch + hh * 8;         // this lane's head dim, first key group of a 16-key step
    for (int rep = 0; rep < qrep; ++rep) {
    const int qt = (qg * qrep + rep) * wpp + wave % wpp;
    if (qt >= qtiles) break;
    // ---- Q^T fragments (B operand), query = qt*32 + r ----------------------------------------------------
    const int query = qt * 32 + r;
    const int qrow = query < N ? query : N - 1;
    att_bf16x8_t qf[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks)
        qf[ks] = __builtin_bit_cast(att_bf16x8_t, *(const u32x4_t*)(base + (size_t)qrow * rowstride + ks * 16 + hh * 8));
    att_f32x16_t o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m = -INFINITY, l = 0.f;                       // running maximum in the scaled log2 domain (s * c), running sum
    // (the running sum on the matrix cores -- ones^T P^T, 2 MFMAs per key tile instead of 12 vector instructions -- was tried: it sums the bf16-ROUNDED
    //  probabilities, which moves one attention output of the C3 net from 1.4e-3 to 1.55e-3 of the bf16-storage oracle, whose normaliser sums the unrounded
    //  ones as the fused transformer kernel does; not kept)
    // K fragments (A operand: lane = key) straight from global, one tile ahead of their use
    const char* const kbase = (const char*)kb + hh * 16;       // (a sample's q | k | v rows are < 4 GB: 32-bit lane offsets)
    const unsigned krowbytes = (unsigned)rowstride * 2u;
    auto load_k = [&](int kt, u32x4_t (&kq)[DK]) __attribute__((always_inline)) {
        const int key_r = kt * 32 + r;
        const unsigned koff = (unsigned)(key_r < N ? key_r : N - 1) * krowbytes;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) kq[ks] = *(const u32x4_t*)(kbase + koff + ks * 32);
    };
    u32x4_t kcur[DK], knext[DK];
    load_k(0, kcur);
    for (int kt = 0; kt < qtiles; ++kt) {
        load_k(kt + 1 < qtiles ? kt + 1 : kt, knext);
        att_f32x16_t st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks)
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8_t, kcur[ks]), qf[ks], st, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) kcur[ks] = knext[ks];
        if (kt == qtiles - 1 && (N & 31)) {               // uniform: the partial key tile
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                st[e] = key < N ? st[e] : -INFINITY;
            }
        }
        const att_f32x2_t c2 = {scale_log2e, scale_log2e};
        float mt;
        {
            att_f32x2_t ts[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ts[e] = att_f32x2_t{st[2 * e], st[2 * e + 1]} * c2;
            mt = fmaxf(fmaxf(ts[0].x, ts[0].y), ts[1].x);
#pragma unroll
            for (int e = 3; e < 15; e += 2) mt = fmaxf(fmaxf(mt, ts[e >> 1][e & 1]), ts[(e + 1) >> 1][(e + 1) & 1]);
            mt = fmaxf(mt, ts[7].y);
            // the other 16 keys of this query sit in lane ^ 32: v_permlane32_swap_b32 (no LDS round trip as ds_bpermute)
            const unsigned mu = __float_as_uint(mt);
            const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (!__all(mt <= m)) {                            // some query's maximum grew: rescale (else alpha == 1 exactly)
            const float mn = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
            m = mn;
        }
        const att_f32x2_t negm2 = {-m, -m};
        att_f32x2_t psum2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const att_f32x2_t x = att_f32x2_t{st[e], st[e + 1]} * c2 + negm2;
            const att_f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            st[e] = pv.x; st[e + 1] = pv.y;
            psum2 += pv;
        }
        l += psum2.x + psum2.y;
#pragma unroll
        for (int sgrp = 0; sgrp < 2; ++sgrp) {
            u32x4_t pw;
            pw.x = pack_bf16x2(st[8 * sgrp + 0], st[8 * sgrp + 1]);
            pw.y = pack_bf16x2(st[8 * sgrp + 2], st[8 * sgrp + 3]);
            pw.z = pack_bf16x2(st[8 * sgrp + 4], st[8 * sgrp + 5]);
            pw.w = pack_bf16x2(st[8 * sgrp + 6], st[8 * sgrp + 7]);
            // head dim 64 (the ADM attention blocks): the probabilities go to the matrix cores as bf16 hi + bf16 lo (two MFMAs), i.e. with ~16 mantissa
            // bits -- a single bf16 P costs ~1e-3 of the output (it flips the bf16 rounding of a quarter of the stored elements), which the head-dim-32
            // path of the 1-D nets accepts and the bf16-storage oracle models; here the oracle keeps its fp32 probabilities
            u32x4_t pl = u32x4_t{0u, 0u, 0u, 0u};
            if constexpr (kSplitP) {
                const unsigned hw[4] = {pw.x, pw.y, pw.z, pw.w};
                unsigned lw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h0 = __uint_as_float(hw[j] << 16), h1 = __uint_as_float(hw[j] & 0xffff0000u);
                    lw[j] = pack_bf16x2(st[8 * sgrp + 2 * j] - h0, st[8 * sgrp + 2 * j + 1] - h1);
                }
                pl = u32x4_t{lw[0], lw[1], lw[2], lw[3]};
            }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                // V^T fragment of head dims 32 t + r: elements j = 0 .. 3 = keys 16 s + 4 h + j, j = 4 .. 7 = keys 16 s + 8 + 4 h + (j - 4)
                u32x4_t vw;
                const u32x2_t v0 = *(const u32x2_t*)(vrow + t * 32 * pitch + (kt * 32 + 16 * sgrp) * 2);
                const u32x2_t v1 = *(const u32x2_t*)(vrow + t * 32 * pitch + (kt * 32 + 16 * sgrp + 8) * 2);
                vw.x = v0.x; vw.y = v0.y; vw.z = v1.x; vw.w = v1.y;
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8_t, vw), __builtin_bit_cast(att_bf16x8_t, pw), o[t], 0, 0, 0);
                if constexpr (kSplitP)
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8_t, vw), __builtin_bit_cast(att_bf16x8_t, pl), o[t], 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (query < N) {
        bf16_t* orow = out + ((size_t)b * N + query) * C + (size_t)hd * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {    // registers 4g..4g+3 of tile t = head dims 32 t + 8g + 4hh .. +3
                uint2 w;
                w.x = pack_bf16x2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
                w.y = pack_bf16x2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                *(uint2*)(orow + t * 32 + 8 * g + 4 * hh) = w;
            }
    }
    }
}

// Self-attention of the split-bf16 mode (ADF_DTYPE_F32X3: fp32 q | k | v rows in, fp32 out), head dim 32, N <= 1024 keys (K from LDS up to 256, from global beyond: template parameter KG).  Same plan as the bf16 kernel above
// -- S^T = K Q^T on the matrix cores with the query on the lane (lane-local softmax), P^T reused as the B operand of O^T += V^T P^T -- with every operand as
// bf16 hi + lo and three MFMAs per product (lo hi, hi lo, hi hi; adf_common.h).  K (rows of 64 B hi / 64 B lo, pitch 80: conflict-free ds_read_b128 with the
// key on the lane) and V^T (vt[d][key], pitch 2 N + 8 as above) of a (sample, head) pair are split ONCE, while they are staged into LDS by the waves that share
// the pair; Q is split in registers, the probabilities after the exp2.  It replaces the one-query-per-lane vector kernel (attention_kernel<float, 32>: 430 us per
// launch at 256 tokens and batch 64 = 11 % of a step of this mode).
// NW waves per workgroup; KG: the K rows come from global memory tile by tile and are split in registers (257 .. 1024 tokens, where K hi | lo of a pair
// (160 B per key) no longer fits beside V^T hi | lo (128 B per key): eight waves then share the 132 KB of one pair's V^T and walk all of its query tiles).
template <int NW, bool KG>
__global__ void __launch_bounds__(NW * 64, 2) attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int N, int C, int heads,
                                                           float scale_log2e, int qrep) {
    constexpr int D = 32, KP = 80;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qtiles = (N + 31) >> 5;
    const int wpp = qtiles >= NW ? NW : (qtiles >= 4 ? 4 : (qtiles >= 2 ? 2 : 1));   // waves that share one (b, head) pair
    const int ppb = NW / wpp;                                   // pairs per block
    const int qgroups = (qtiles + wpp * qrep - 1) / (wpp * qrep);
    const int qg = blockIdx.x % qgroups, pg = blockIdx.x / qgroups;
    const int pair = pg * ppb + wave / wpp;
    const bool pair_ok = pair < B * heads;
    const int pc = pair_ok ? pair : 0;
    const int b = pc / heads, hd = pc - b * heads;
    const size_t rowstride = (size_t)3 * C;
    const float* base = qkv + (size_t)b * N * rowstride + (size_t)hd * D;
    const int npad = qtiles * 32;
    const int vpitch = npad * 2 + 8;
    const size_t kbytes = KG ? 0 : (size_t)npad * KP;
    const size_t pair_bytes = 2 * kbytes + (size_t)2 * D * vpitch;
    char* kh = smem + (size_t)(wave / wpp) * pair_bytes;
    char* kl = kh + kbytes;
    char* vth = kl + kbytes;
    char* vtl = vth + (size_t)D * vpitch;
    {
        const int tl = (wave % wpp) * 64 + lane, nthr = wpp * 64;
        // four pieces per thread requested before the first is split and scattered (see the bf16 kernel: a load per trip serialises the memory round trips)
        const int total = npad * 8;
        for (int base_i = tl; base_i < total; base_i += 4 * nthr) {
            f32x4_hw_t kvv[4], vvv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base_i + u * nthr;
                const int key = idx >> 3, c = idx & 7;              // 16-byte chunk c: head dims 4 c .. 4 c + 3
                const int krow = (idx < total && key < N) ? key : N - 1;    // padded keys: any finite value (their probability is 0)
                if (!KG) kvv[u] = *(const f32x4_hw_t*)(base + (size_t)krow * rowstride + C + c * 4);
                vvv[u] = *(const f32x4_hw_t*)(base + (size_t)krow * rowstride + 2 * C + c * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base_i + u * nthr;
                if (idx >= total) continue;
                const int key = idx >> 3, c = idx & 7;
                u32x2_t hi, lo;
                if (!KG) {
                    const float kf[4] = {kvv[u].x, kvv[u].y, kvv[u].z, kvv[u].w};
                    split_bf16x4(kf, hi, lo);
                    *(u32x2_t*)(kh + key * KP + c * 8) = hi;
                    *(u32x2_t*)(kl + key * KP + c * 8) = lo;
                }
                const float vf[4] = {vvv[u].x, vvv[u].y, vvv[u].z, vvv[u].w};
                split_bf16x4(vf, hi, lo);
                const unsigned hw[2] = {hi.x, hi.y}, lw[2] = {lo.x, lo.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *(unsigned short*)(vth + (c * 4 + j) * vpitch + key * 2) = (unsigned short)((j & 1) ? (hw[j >> 1] >> 16) : (hw[j >> 1] & 0xffffu));
                    *(unsigned short*)(vtl + (c * 4 + j) * vpitch + key * 2) = (unsigned short)((j & 1) ? (lw[j >> 1] >> 16) : (lw[j >> 1] & 0xffffu));
                }
            }
        }
    }
    __syncthreads();
    if (!pair_ok) return;
    const char* vrow_h = vth + r * vpitch + hh * 8;     // this lane's head dim, first key group of a 16-key step
    const char* vrow_l = vtl + r * vpitch + hh * 8;
    for (int rep = 0; rep < qrep; ++rep) {
        const int qt = (qg * qrep + rep) * wpp + wave % wpp;
        if (qt >= qtiles) break;
        const int query = qt * 32 + r;
        const int qrow = query < N ? query : N - 1;
        att_bf16x8_t qh[2], ql[2];                       // Q^T fragments (B operand): head dims 16 ks + 8 hh .. + 7 of this lane's query
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* qp = base + (size_t)qrow * rowstride + ks * 16 + hh * 8;
            const f32x4_hw_t a0 = *(const f32x4_hw_t*)qp, a1 = *(const f32x4_hw_t*)(qp + 4);
            const float f0[4] = {a0.x, a0.y, a0.z, a0.w}, f1[4] = {a1.x, a1.y, a1.z, a1.w};
            u32x2_t h0, l0, h1, l1;
            split_bf16x4(f0, h0, l0);
            split_bf16x4(f1, h1, l1);
            qh[ks] = __builtin_bit_cast(att_bf16x8_t, u32x4_t{h0.x, h0.y, h1.x, h1.y});
            ql[ks] = __builtin_bit_cast(att_bf16x8_t, u32x4_t{l0.x, l0.y, l1.x, l1.y});
        }
        att_f32x16_t o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
        float m = -INFINITY, l = 0.f;                   // running maximum in the scaled log2 domain (s * c), running sum
        // KG: this lane's key row of a tile (head dims 16 ks + 8 hh .. + 7, two K steps = 64 bytes) straight from global, one tile ahead of its use
        f32x4_hw_t kreg[4], knext[4];
        auto load_k = [&](int kt, f32x4_hw_t (&kq)[4]) __attribute__((always_inline)) {
            const int key_r = kt * 32 + r;
            const float* kp = base + (size_t)(key_r < N ? key_r : N - 1) * rowstride + C + hh * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { kq[2 * ks] = *(const f32x4_hw_t*)(kp + ks * 16); kq[2 * ks + 1] = *(const f32x4_hw_t*)(kp + ks * 16 + 4); }
        };
        if (KG) load_k(0, kreg);
        for (int kt = 0; kt < qtiles; ++kt) {
            if (KG) load_k(kt + 1 < qtiles ? kt + 1 : kt, knext);
            att_f32x16_t st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                att_bf16x8_t fh, fl;
                if (KG) {
                    const float f0[4] = {kreg[2 * ks].x, kreg[2 * ks].y, kreg[2 * ks].z, kreg[2 * ks].w};
                    const float f1[4] = {kreg[2 * ks + 1].x, kreg[2 * ks + 1].y, kreg[2 * ks + 1].z, kreg[2 * ks + 1].w};
                    u32x2_t h0, l0, h1, l1;
                    split_bf16x4(f0, h0, l0);
                    split_bf16x4(f1, h1, l1);
                    fh = __builtin_bit_cast(att_bf16x8_t, u32x4_t{h0.x, h0.y, h1.x, h1.y});
                    fl = __builtin_bit_cast(att_bf16x8_t, u32x4_t{l0.x, l0.y, l1.x, l1.y});
                } else {
                    fh = *(const att_bf16x8_t*)(kh + (kt * 32 + r) * KP + (ks * 2 + hh) * 16);
                    fl = *(const att_bf16x8_t*)(kl + (kt * 32 + r) * KP + (ks * 2 + hh) * 16);
                }
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, qh[ks], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, ql[ks], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, qh[ks], st, 0, 0, 0);
            }
            if (KG) {
#pragma unroll
                for (int u = 0; u < 4; ++u) kreg[u] = knext[u];
            }
            if (kt == qtiles - 1 && (N & 31)) {               // uniform: the partial key tile
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    st[e] = key < N ? st[e] : -INFINITY;
                }
            }
            // (the vector-instruction trims of attention_mfma32_kernel: maximum over the scaled scores as v_max3, packed fp32, v_permlane32_swap)
            const att_f32x2_t c2 = {scale_log2e, scale_log2e};
            float mt;
            {
                att_f32x2_t ts[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ts[e] = att_f32x2_t{st[2 * e], st[2 * e + 1]} * c2;
                mt = fmaxf(fmaxf(ts[0].x, ts[0].y), ts[1].x);
#pragma unroll
                for (int e = 3; e < 15; e += 2) mt = fmaxf(fmaxf(mt, ts[e >> 1][e & 1]), ts[(e + 1) >> 1][(e + 1) & 1]);
                mt = fmaxf(mt, ts[7].y);
                const unsigned mu = __float_as_uint(mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            if (!__all(mt <= m)) {                            // some query's maximum grew: rescale (else alpha == 1 exactly)
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                l *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] *= alpha;
                m = mn;
            }
            const att_f32x2_t negm2 = {-m, -m};
            att_f32x2_t psum2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const att_f32x2_t x = att_f32x2_t{st[e], st[e + 1]} * c2 + negm2;
                const att_f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                st[e] = pv.x; st[e + 1] = pv.y;
                psum2 += pv;
            }
            l += psum2.x + psum2.y;
#pragma unroll
            for (int sgrp = 0; sgrp < 2; ++sgrp) {
                float pf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = st[8 * sgrp + j];
                u32x2_t h0, l0, h1, l1;
                split_bf16x4(pf, h0, l0);
                split_bf16x4(pf + 4, h1, l1);
                const att_bf16x8_t ph = __builtin_bit_cast(att_bf16x8_t, u32x4_t{h0.x, h0.y, h1.x, h1.y});
                const att_bf16x8_t pl = __builtin_bit_cast(att_bf16x8_t, u32x4_t{l0.x, l0.y, l1.x, l1.y});
                // V^T fragments of this lane's head dim: elements j = 0 .. 3 = keys 16 s + 4 h + j, j = 4 .. 7 = keys 16 s + 8 + 4 h + (j - 4)
                const int ko = (kt * 32 + 16 * sgrp) * 2;
                const u32x2_t a0 = *(const u32x2_t*)(vrow_h + ko), a1 = *(const u32x2_t*)(vrow_h + ko + 16);
                const u32x2_t b0 = *(const u32x2_t*)(vrow_l + ko), b1 = *(const u32x2_t*)(vrow_l + ko + 16);
                const att_bf16x8_t vh = __builtin_bit_cast(att_bf16x8_t, u32x4_t{a0.x, a0.y, a1.x, a1.y});
                const att_bf16x8_t vl = __builtin_bit_cast(att_bf16x8_t, u32x4_t{b0.x, b0.y, b1.x, b1.y});
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o, 0, 0, 0);
            }
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (query < N) {
            float* orow = out + ((size_t)b * N + query) * C + (size_t)hd * D;
#pragma unroll
            for (int g = 0; g < 4; ++g)      // registers 4g .. 4g+3 = head dims 8g + 4hh .. +3
                *(f32x4_hw_t*)(orow + 8 * g + 4 * hh) = f32x4_hw_t{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
        }
    }
}

template <typename T>
static const char* attention_dispatch(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
    const int dh = C / heads;
    const long long total = (long long)B * heads * N;
    const unsigned grid = (unsigned)((total + 255) / 256);
    const float scale = 1.0f / sqrtf((float)dh);
#define ADF_ATT(D) hipLaunchKernelGGL((attention_kernel<T, D>), dim3(grid), dim3(256), 0, s, (const T*)qkv, (T*)out, B, N, C, heads, scale)
    switch (dh) {
        case 8: ADF_ATT(8); break;
        case 16: ADF_ATT(16); break;
        case 32: ADF_ATT(32); break;
        case 64: ADF_ATT(64); break;
        default: return "attention: head dim must be 8, 16, 32 or 64";
    }
#undef ADF_ATT
    return ADF_LAUNCH_CHECK("attention");
}

const char* launch_attention(const void* qkv, void* out, int bf16, int B, int N, int C, int heads, hipStream_t s) {
    if (C % heads) return "attention: C % heads != 0";
    const int dh_ = C / heads;
    if (bf16 && (dh_ == 32 || dh_ == 64) && N >= 1 && N <= (dh_ == 32 ? 1024 : 512)) {        // V^T of a pair in LDS: dh x (N padded to 32) bf16 <= 72 KB
        const int qtiles = (N + 31) / 32;
        // eight waves per workgroup from 8 query tiles on (head dim 32; ADF_ATT_NW=4 keeps four: A/B)
        static int att_nw = -1;
        if (att_nw < 0) att_nw = adf_route_switch("ADF_ATT_NW", 8);
        const int nw = (dh_ == 32 && att_nw == 8 && qtiles >= 8) ? 8 : 4;      // (256 tokens: one workgroup per pair, 21.9 -> 19.9 us; 1024: 204 -> 183 us)
        const int wpp = qtiles >= nw ? nw : (qtiles >= 4 ? 4 : (qtiles >= 2 ? 2 : 1));
        const int ppb = nw / wpp;
        // query tiles per wave: 2 from 8 tiles on, as long as >= 4 blocks per CU remain
        const int qrep = (qtiles >= 8 && (long long)B * heads * (qtiles / 8) >= 1024) ? 2 : 1;
        const int qgroups = (qtiles + wpp * qrep - 1) / (wpp * qrep);
        const long long blocks = (long long)((B * heads + ppb - 1) / ppb) * qgroups;
        const size_t lds = (size_t)ppb * dh_ * ((size_t)qtiles * 64 + 8);      // vt[dh][keys padded to 32] + 8 bytes of row padding, per pair
        if (lds > 72 * 1024) return "attention_mfma: V^T does not fit LDS";
        static bool attr_done[kMaxDevices][3] = {};
        bool& attr = attr_done[current_device()][dh_ == 64 ? 1 : (nw == 8 ? 2 : 0)];
        const void* fn = dh_ == 32 ? (nw == 8 ? (const void*)attention_mfma32_kernel<32, 8> : (const void*)attention_mfma32_kernel<32>) : (const void*)attention_mfma32_kernel<64>;
        if (!attr) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess)
                return "attention_mfma: hipFuncSetAttribute failed";
            attr = true;
        }
        const float sl2e = (float)(1.4426950408889634 / sqrt((double)dh_));
        if (dh_ == 32 && nw == 8)
            hipLaunchKernelGGL((attention_mfma32_kernel<32, 8>), dim3((unsigned)blocks), dim3(512), lds, s, (const bf16_t*)qkv, (bf16_t*)out, B, N, C, heads, sl2e, qrep);
        else if (dh_ == 32)
            hipLaunchKernelGGL(attention_mfma32_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)out, B, N, C, heads, sl2e, qrep);
        else
            hipLaunchKernelGGL(attention_mfma32_kernel<64>, dim3((unsigned)blocks), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)out, B, N, C, heads, sl2e, qrep);
        return ADF_LAUNCH_CHECK("attention_mfma");
    }
    return bf16 ? attention_dispatch<bf16_t>(qkv, out, B, N, C, heads, s) : attention_dispatch<float>(qkv, out, B, N, C, heads, s);
}

// split-bf16 mode (fp32 tensors): the MFMA kernel above where it applies (head dim 32, <= 1024 tokens), else the fp32 vector kernel
const char* launch_attention_x3(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
    if (C % heads) return "attention: C % heads != 0";
    static int use = -1;
    if (use < 0) use = adf_route_switch("ADF_ATT_X3", 1);        // 0: the vector kernel (route test)
    if (!use || C / heads != 32 || N < 1 || N > 1024 || C % 4) return launch_attention(qkv, out, 0, B, N, C, heads, s);
    const int qtiles = (N + 31) / 32;
    const bool kg = N > 256;                                      // K from global (split in registers), eight waves per workgroup
    const int nw = kg ? 8 : 4;
    const int wpp = qtiles >= nw ? nw : (qtiles >= 4 ? 4 : (qtiles >= 2 ? 2 : 1));
    const int ppb = nw / wpp;
    const int qrep = qtiles > wpp ? (qtiles + wpp - 1) / wpp : 1;          // every pair is staged once: its waves walk all of its query tiles
    const int qgroups = (qtiles + wpp * qrep - 1) / (wpp * qrep);
    const long long blocks = (long long)((B * heads + ppb - 1) / ppb) * qgroups;
    const size_t npad = (size_t)qtiles * 32;
    const size_t lds = (size_t)ppb * ((kg ? 0 : 2 * npad * 80) + 2 * 32 * (npad * 2 + 8));
    if (lds > (kg ? 136 : 80) * 1024 || blocks > 0x7fffffffLL) return launch_attention(qkv, out, 0, B, N, C, heads, s);
    static bool attr_done[kMaxDevices][2] = {};
    bool& attr = attr_done[current_device()][kg];
    if (!attr) {
        const hipError_t e = kg ? hipFuncSetAttribute((const void*)attention_x3_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024)
                                : hipFuncSetAttribute((const void*)attention_x3_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return "attention_x3: hipFuncSetAttribute failed";
        attr = true;
    }
    const float sl2e = (float)(1.4426950408889634 / sqrt(32.0));
    if (kg) hipLaunchKernelGGL((attention_x3_kernel<8, true>), dim3((unsigned)blocks), dim3(512), lds, s, (const float*)qkv, (float*)out, B, N, C, heads, sl2e, qrep);
    else hipLaunchKernelGGL((attention_x3_kernel<4, false>), dim3((unsigned)blocks), dim3(256), lds, s, (const float*)qkv, (float*)out, B, N, C, heads, sl2e, qrep);
    return ADF_LAUNCH_CHECK("attention_x3");
}

// =====================================================================================================
// Waveform -> features: WAVenc1d (unet1d.py:584-591) with c_in * x fused (diffusion.py:50)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) to_in_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ out,
                                                    int in_ch, int L, int nf, int wl, int stride, int pad,
                                                    const float* __restrict__ coef, int coef_bstride) {
    constexpr int EPC = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl_s = (float*)smem;  // [in_ch][wl][nf]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < nf * in_ch * wl; i += 256) {
        const int k = i % wl, ci = (i / wl) % in_ch, co = i / (wl * in_ch);
        wl_s[(ci * wl + k) * nf + co] = w[i];
    }
    __syncthreads();
    const int cpr = nf / EPC;
    const int Lo = L / stride;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Lo * cpr) return;
    const int cc = (int)(idx % cpr);
    const int m = (int)(idx / cpr);
    const float cin = coef ? coef[(size_t)b * coef_bstride] : 1.0f;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < in_ch; ++ci) {
        const float* xr = x + ((size_t)b * in_ch + ci) * L;
        for (int k = 0; k < wl; ++k) {
            const int p = m * stride + k - pad;
            if (p < 0 || p >= L) continue;
            const float xv = cin * xr[p];
            const float* wr = wl_s + (ci * wl + k) * nf + cc * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaf(wr[e], xv, acc[e]);
        }
    }
    *(u32x4_t*)(out + ((size_t)b * Lo + m) * nf + (size_t)cc * EPC) = pack16<T>(acc);
}

// Fast path of the shipped shape (one waveform channel, window 8, stride 2): a thread owns one 16-byte output chunk column
// (its EPC channels x 8 taps of weights stay in registers) and walks ROWS_PER_BLOCK / rows-per-iteration rows of one sample, so
// the weight load is paid once per 1024 rows instead of once per 32 (the generic kernel above spends most of its 40 us per
// call in the global -> LDS weight prologue of its 16 K small blocks); same arithmetic order as the generic kernel.
template <typename T>
__global__ void __launch_bounds__(256) to_in_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ out, int L,
                                                         int nf, int pad, const float* __restrict__ coef, int coef_bstride) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int WL = 8, STRIDE = 2, ROWS = 256;   // output rows per block
    constexpr int WIN = ROWS * STRIDE + 8;           // staged waveform window: x[2 m0 - 4 .. 2 m0 + 2 ROWS + 4)
    __shared__ __attribute__((aligned(16))) float xs[WIN];
    const int b = blockIdx.y;
    const int cpr = nf / EPC;                       // 16-byte chunks per output row (a power of two <= 256, checked by the launcher)
    const int cc = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int rpi = 256 / cpr;                      // rows per iteration
    const int Lo = L / STRIDE;
    const int m0 = blockIdx.x * ROWS;
    const float cin = coef ? coef[(size_t)b * coef_bstride] : 1.0f;
    const float* xr = x + (size_t)b * L;
    // c_in * x of the window, zero outside the waveform (one 16-byte load per thread: the 8 scalar loads per output chunk of a
    // per-thread version made the kernel vector-memory-instruction bound at 30 us)
    for (int i = threadIdx.x; i < WIN / 4; i += 256) {
        const int p0 = m0 * STRIDE - 4 + i * 4;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (p0 >= 0 && p0 + 3 < L) v = *(const float4*)(xr + p0);
        else {
            if (p0 >= 0 && p0 < L) v.x = xr[p0];
            if (p0 + 1 >= 0 && p0 + 1 < L) v.y = xr[p0 + 1];
            if (p0 + 2 >= 0 && p0 + 2 < L) v.z = xr[p0 + 2];
            if (p0 + 3 >= 0 && p0 + 3 < L) v.w = xr[p0 + 3];
        }
        *(float4*)(xs + i * 4) = float4{cin * v.x, cin * v.y, cin * v.z, cin * v.w};
    }
    float wr[WL][EPC];
#pragma unroll
    for (int k = 0; k < WL; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) wr[k][e] = w[(size_t)(cc * EPC + e) * WL + k];
    __syncthreads();
    const int shift = 4 - pad;                      // x[m * STRIDE + k - pad] sits at xs[(m - m0) * STRIDE + k + shift]
#pragma unroll 4
    for (int ml = rsub; ml < ROWS && m0 + ml < Lo; ml += rpi) {
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const float xv = xs[ml * STRIDE + k + shift];
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaf(wr[k][e], xv, acc[e]);
        }
        *(u32x4_t*)(out + ((size_t)b * Lo + m0 + ml) * nf + (size_t)cc * EPC) = pack16<T>(acc);
    }
}

const char* launch_to_in(const float* x, const float* w, void* out, int bf16, int B, int in_ch, int L, int nf,
                         int wl, int stride, int pad, const float* coef, int coef_bstride, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (nf % epc) return "to_in: num_filters must be a multiple of a 16-byte chunk";
    if (L % stride) return "to_in: L % stride != 0";
    const int cpr_fast = nf / epc;
    if (in_ch == 1 && wl == 8 && stride == 2 && pad >= 0 && pad <= 4 && L % 4 == 0 && cpr_fast <= 256 && (cpr_fast & (cpr_fast - 1)) == 0) {
        const int Lo = L / stride;
        dim3 grid((unsigned)((Lo + 255) / 256), B);
        if (bf16) hipLaunchKernelGGL(to_in_rows_kernel<bf16_t>, grid, dim3(256), 0, s, x, w, (bf16_t*)out, L, nf, pad, coef, coef_bstride);
        else hipLaunchKernelGGL(to_in_rows_kernel<float>, grid, dim3(256), 0, s, x, w, (float*)out, L, nf, pad, coef, coef_bstride);
        return ADF_LAUNCH_CHECK("to_in_rows");
    }
    const size_t lds = (size_t)nf * in_ch * wl * sizeof(float);
    if (lds > 60000) return "to_in: weight tile too large for LDS";
    const long long work = (long long)(L / stride) * (nf / epc);
    dim3 grid((unsigned)((work + 255) / 256), B);
    if (bf16) hipLaunchKernelGGL(to_in_kernel<bf16_t>, grid, dim3(256), lds, s, x, w, (bf16_t*)out, in_ch, L, nf, wl, stride, pad, coef, coef_bstride);
    else hipLaunchKernelGGL(to_in_kernel<float>, grid, dim3(256), lds, s, x, w, (float*)out, in_ch, L, nf, wl, stride, pad, coef, coef_bstride);
    return ADF_LAUNCH_CHECK("to_in");
}

// =====================================================================================================
// Features -> waveform: WAVdec1d (unet1d.py:611-622) + EDM combine and clamp (diffusion.py:60-63)
// Each thread turns one feature row into its wl tap products P[row][k]; outputs gather wl/stride of them.
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) to_out_kernel(const T* __restrict__ h, const float* __restrict__ w, float* __restrict__ out,
                                                     int out_ch, int Lh, int nf, int wl, int stride, int pad, int mode,
                                                     const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                     int coef_bstride) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int R = 256;                 // feature rows owned by a block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (wl + stride - 1) / stride;
    float* ws = (float*)smem;              // [nf][wl]   (for this output channel)
    float* P = ws + nf * wl;               // [R + 2*halo][wl]
    const int b = blockIdx.y, oc = blockIdx.z;
    for (int i = threadIdx.x; i < nf * wl; i += 256) {
        const int k = i % wl, ci = i / wl;
        ws[i] = w[((size_t)ci * out_ch + oc) * wl + k];
    }
    __syncthreads();
    const int i0 = blockIdx.x * R - halo;
    const int nrows = R + 2 * halo;
    for (int rr = threadIdx.x; rr < nrows; rr += 256) {
        const int i = i0 + rr;
        float p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = 0.f;
        if (i >= 0 && i < Lh) {
            const T* row = h + ((size_t)b * Lh + i) * nf;
            for (int c = 0; c < nf; c += EPC) {
                float f[EPC];
                unpack16<T>(*(const u32x4_t*)(row + c), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float* wr = ws + (c + e) * wl;
#pragma unroll
                    for (int k = 0; k < 16; ++k) if (k < wl) p[k] = fmaf(f[e], wr[k], p[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < wl) P[rr * wl + k] = p[k];
    }
    __syncthreads();
    const int L = Lh * stride;
    const int l0 = blockIdx.x * R * stride;
    float c_skip = 0.f, c_out = 1.f;
    if (mode == 1) { c_skip = coef[(size_t)b * coef_bstride + 2]; c_out = coef[(size_t)b * coef_bstride + 3]; }
    for (int t = threadIdx.x; t < R * stride; t += 256) {
        const int l = l0 + t;
        if (l >= L) break;
        float acc = 0.f;
        // taps k with (l + pad - k) % stride == 0, feature row i = (l + pad - k) / stride
        const int k0 = (l + pad) % stride;
        for (int k = k0; k < wl; k += stride) {
            const int i = (l + pad - k) / stride;
            if (i >= 0 && i < Lh && (l + pad - k) >= 0) acc += P[(i - i0) * wl + k];
        }
        const size_t o = ((size_t)b * out_ch + oc) * L + l;
        float v = acc;
        if (mode == 1) {
            v = fmaf(c_skip, x_noisy[o], c_out * acc);
            v = fminf(fmaxf(v, -1.0f), 1.0f);
        }
        out[o] = v;
    }
}

// bf16 fast path: the tap products P[row][k] = sum_ci h[row][ci] * w[ci][k] are a skinny GEMM
// (rows x nf) x (nf x wl<=32) done on MFMA with the A fragments loaded straight from global memory
// (lane = row, 16 bytes = 8 consecutive channels: exactly the 32x32x16 A-operand layout, no LDS staging).
typedef __attribute__((ext_vector_type(8))) __bf16 to_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float to_f32x16_t;
__global__ void __launch_bounds__(256) to_out_mfma_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w, float* __restrict__ out,
                                                          int out_ch, int Lh, int nf, int wl, int stride, int pad, int mode,
                                                          const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                          int coef_bstride) {
    constexpr int R = 256;                 // feature rows owned by a block
    constexpr int MAXKS = 8;               // nf <= 128
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (wl + stride - 1) / stride;
    float* P = (float*)smem;               // [ceil32(R + 2*halo)][wl]
    const int b = blockIdx.y, oc = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ksteps = nf / 16;
    // B operand: lane (r = tap column, hh) holds w[ci = 16 ks + 8 hh + j][oc][tap r]
    to_bf16x8_t bf[MAXKS];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks) {
        u32x4_t q = u32x4_t{0u, 0u, 0u, 0u};
        if (ks < ksteps && r < wl) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = w[((size_t)(16 * ks + 8 * hh + j) * out_ch + oc) * wl + r];
            q = pack16<bf16_t>(f);
        }
        bf[ks] = __builtin_bit_cast(to_bf16x8_t, q);
    }
    const int i0 = blockIdx.x * R - halo;
    const int nrows = R + 2 * halo;
    const int ntile = (nrows + 31) / 32;
    // nf == 64 (the benched nets): the 32 rows of a tile are read as four fully coalesced 1 KB pieces (8 rows x 128 bytes per wave
    // instruction) into a wave-private LDS tile and the A fragments come from there; the direct form below issues 64 separate 16-byte
    // requests at a 128-byte stride per instruction (33 us per launch at B = 64)
    const bool staged = nf == 64;
    char* const tl = (char*)(P + (size_t)ntile * 32 * wl) + wave * (32 * 144);
    for (int t = wave; t < ntile; t += 4) {
        const int i = i0 + t * 32 + r;                       // this lane's feature row (A operand row)
        const int ic = i < 0 ? 0 : (i >= Lh ? Lh - 1 : i);   // clamped: out-of-range rows are masked below
        const bf16_t* row = h + ((size_t)b * Lh + ic) * nf + hh * 8;
        to_f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (staged) {
            u32x4_t rv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = q * 8 + (lane >> 3);
                const int ii = i0 + t * 32 + rr;
                const int icc = ii < 0 ? 0 : (ii >= Lh ? Lh - 1 : ii);
                rv[q] = *(const u32x4_t*)(h + ((size_t)b * Lh + icc) * nf + (lane & 7) * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4_t*)(tl + (q * 8 + (lane >> 3)) * 144 + (lane & 7) * 16) = rv[q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const to_bf16x8_t af = *(const to_bf16x8_t*)(tl + r * 144 + (ks * 2 + hh) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[ks], acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();                 // the tile is rewritten in the next trip
        } else {
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                if (ks < ksteps) {
                    const to_bf16x8_t af = __builtin_bit_cast(to_bf16x8_t, *(const u32x4_t*)(row + ks * 16));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[ks], acc, 0, 0, 0);
                }
            }
        }
        if (r < wl) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int ii = i0 + rr;
                if (rr < nrows) P[rr * wl + r] = (ii >= 0 && ii < Lh) ? acc[e] : 0.f;
            }
        }
    }
    __syncthreads();
    const int L = Lh * stride;
    const int l0 = blockIdx.x * R * stride;
    float c_skip = 0.f, c_out = 1.f;
    if (mode == 1) { c_skip = coef[(size_t)b * coef_bstride + 2]; c_out = coef[(size_t)b * coef_bstride + 3]; }
    for (int t = threadIdx.x; t < R * stride; t += 256) {
        const int l = l0 + t;
        if (l >= L) break;
        float acc = 0.f;
        const int k0 = (l + pad) % stride;
        for (int k = k0; k < wl; k += stride) {
            const int i = (l + pad - k) / stride;
            if (i >= 0 && i < Lh && (l + pad - k) >= 0) acc += P[(i - i0) * wl + k];
        }
        const size_t o = ((size_t)b * out_ch + oc) * L + l;
        float v = acc;
        if (mode == 1) {
            v = fmaf(c_skip, x_noisy[o], c_out * acc);
            v = fminf(fmaxf(v, -1.0f), 1.0f);
        }
        out[o] = v;
    }
}

// Split-bf16 form (ADF_DTYPE_F32X3: fp32 rows, each operand as bf16 hi + lo, three MFMAs per product): the same skinny GEMM.  A tile of 32 rows x 64
// channels (8 KB) is read as eight fully coalesced 1 KB pieces (4 rows x 256 bytes per wave instruction) into a wave-private LDS tile (pitch 272 bytes);
// the lane's 8 channels of a K step (32 bytes) are split there into the hi and lo A fragments.  nf a multiple of 64.
__global__ void __launch_bounds__(256) to_out_x3_kernel(const float* __restrict__ h, const float* __restrict__ w, float* __restrict__ out,
                                                        int out_ch, int Lh, int nf, int wl, int stride, int pad, int mode,
                                                        const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                        int coef_bstride) {
    constexpr int R = 256;                 // feature rows owned by a block
    constexpr int MAXKS = 8;               // nf <= 128
    constexpr int PITCH = 272;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (wl + stride - 1) / stride;
    float* P = (float*)smem;               // [ceil32(R + 2*halo)][wl]
    const int b = blockIdx.y, oc = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ksteps = nf / 16;
    // B operand: lane (r = tap column, hh) holds w[ci = 16 ks + 8 hh + j][oc][tap r], split once
    to_bf16x8_t bh[MAXKS], bl[MAXKS];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks) {
        u32x4_t qh = u32x4_t{0u, 0u, 0u, 0u}, ql = qh;
        if (ks < ksteps && r < wl) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = w[((size_t)(16 * ks + 8 * hh + j) * out_ch + oc) * wl + r];
            u32x2_t h0, l0, h1, l1;
            split_bf16x4(f, h0, l0);
            split_bf16x4(f + 4, h1, l1);
            qh = u32x4_t{h0.x, h0.y, h1.x, h1.y};
            ql = u32x4_t{l0.x, l0.y, l1.x, l1.y};
        }
        bh[ks] = __builtin_bit_cast(to_bf16x8_t, qh);
        bl[ks] = __builtin_bit_cast(to_bf16x8_t, ql);
    }
    const int i0 = blockIdx.x * R - halo;
    const int nrows = R + 2 * halo;
    const int ntile = (nrows + 31) / 32;
    char* const tl = (char*)(P + (size_t)ntile * 32 * wl) + wave * (32 * PITCH);
    for (int t = wave; t < ntile; t += 4) {
        to_f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int cc = 0; cc < MAXKS / 4; ++cc) {
            const int c0 = cc * 64;
            if (c0 >= nf) break;
            u32x4_t rv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ii = i0 + t * 32 + q * 4 + (lane >> 4);
                const int icc = ii < 0 ? 0 : (ii >= Lh ? Lh - 1 : ii);     // clamped: out-of-range rows are masked below
                rv[q] = *(const u32x4_t*)(h + ((size_t)b * Lh + icc) * nf + c0 + (lane & 15) * 4);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *(u32x4_t*)(tl + (q * 4 + (lane >> 4)) * PITCH + (lane & 15) * 16) = rv[q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char* ap = tl + r * PITCH + (ks * 2 + hh) * 32;
                const u32x4_t a0 = *(const u32x4_t*)ap, a1 = *(const u32x4_t*)(ap + 16);
                u32x2_t h0, l0, h1, l1;
                split_bf16x4((const float*)&a0, h0, l0);
                split_bf16x4((const float*)&a1, h1, l1);
                const to_bf16x8_t ah = __builtin_bit_cast(to_bf16x8_t, u32x4_t{h0.x, h0.y, h1.x, h1.y});
                const to_bf16x8_t al = __builtin_bit_cast(to_bf16x8_t, u32x4_t{l0.x, l0.y, l1.x, l1.y});
                const int kk = cc * 4 + ks;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kk], acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();                 // the tile is rewritten in the next trip
        }
        if (r < wl) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int ii = i0 + rr;
                if (rr < nrows) P[rr * wl + r] = (ii >= 0 && ii < Lh) ? acc[e] : 0.f;
            }
        }
    }
    __syncthreads();
    const int L = Lh * stride;
    const int l0 = blockIdx.x * R * stride;
    float c_skip = 0.f, c_out = 1.f;
    if (mode == 1) { c_skip = coef[(size_t)b * coef_bstride + 2]; c_out = coef[(size_t)b * coef_bstride + 3]; }
    for (int t = threadIdx.x; t < R * stride; t += 256) {
        const int l = l0 + t;
        if (l >= L) break;
        float acc = 0.f;
        const int k0 = (l + pad) % stride;
        for (int k = k0; k < wl; k += stride) {
            const int i = (l + pad - k) / stride;
            if (i >= 0 && i < Lh && (l + pad - k) >= 0) acc += P[(i - i0) * wl + k];
        }
        const size_t o = ((size_t)b * out_ch + oc) * L + l;
        float v = acc;
        if (mode == 1) {
            v = fmaf(c_skip, x_noisy[o], c_out * acc);
            v = fminf(fmaxf(v, -1.0f), 1.0f);
        }
        out[o] = v;
    }
}

// dtype: 0 = fp32 rows (VALU), 1 = bf16 rows, 2 = fp32 rows with split-bf16 products
const char* launch_to_out(const void* h, const float* w, float* out, int dtype, int B, int out_ch, int Lh, int nf,
                          int wl, int stride, int pad, int mode, const float* x_noisy, const float* coef,
                          int coef_bstride, hipStream_t s) {
    const int bf16 = dtype == 1;
    const int epc = bf16 ? 8 : 4;
    if (nf % epc) return "to_out: num_filters must be a multiple of a 16-byte chunk";
    if (wl > 16) return "to_out: window_length > 16 unsupported";
    const int halo = (wl + stride - 1) / stride;
    dim3 grid(ceil_div(Lh, 256), B, out_ch);
    if (bf16 && nf % 16 == 0 && nf <= 128) {
        const size_t lds = (size_t)round_up(256 + 2 * halo, 32) * wl * sizeof(float) + 4 * 32 * 144;      // P + one staging tile per wave
        hipLaunchKernelGGL(to_out_mfma_kernel, grid, dim3(256), lds, s, (const bf16_t*)h, w, out, out_ch, Lh, nf, wl, stride, pad, mode, x_noisy, coef, coef_bstride);
        return ADF_LAUNCH_CHECK("to_out_mfma");
    }
    if (dtype == 2 && nf % 64 == 0 && nf <= 128) {
        const size_t lds = (size_t)round_up(256 + 2 * halo, 32) * wl * sizeof(float) + 4 * 32 * 272;      // P + one staging tile per wave
        hipLaunchKernelGGL(to_out_x3_kernel, grid, dim3(256), lds, s, (const float*)h, w, out, out_ch, Lh, nf, wl, stride, pad, mode, x_noisy, coef, coef_bstride);
        return ADF_LAUNCH_CHECK("to_out_x3");
    }
    const size_t lds = ((size_t)nf * wl + (size_t)(256 + 2 * halo) * wl) * sizeof(float);
    if (bf16) hipLaunchKernelGGL(to_out_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)h, w, out, out_ch, Lh, nf, wl, stride, pad, mode, x_noisy, coef, coef_bstride);
    else hipLaunchKernelGGL(to_out_kernel<float>, grid, dim3(256), lds, s, (const float*)h, w, out, out_ch, Lh, nf, wl, stride, pad, mode, x_noisy, coef, coef_bstride);
    return ADF_LAUNCH_CHECK("to_out");
}

// =====================================================================================================
// EDM preconditioning scalars (diffusion.py:232-241): coef[b] = (c_in, c_noise, c_skip, c_out)
// =====================================================================================================
__global__ void edm_coef_kernel(const float* __restrict__ sigmas, float sigma_scalar, int nb, float sd, float* __restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const float s = sigmas ? sigmas[b] : sigma_scalar;
    const float s2 = s * s, d2 = sd * sd;
    coef[b * 4 + 0] = 1.0f / sqrtf(s2 + d2);
    coef[b * 4 + 1] = logf(s) * 0.25f;
    coef[b * 4 + 2] = d2 / (s2 + d2);
    coef[b * 4 + 3] = s * sd * (1.0f / sqrtf(d2 + s2));
}

// The same for a list of sigmas known on the host (the sigma of every denoiser evaluation of a sampler run): the values ride in
// the kernel arguments (no host -> device copy inside a captured graph), 256 per launch.
struct SigmaPack { float s[256]; };
__global__ void edm_coef_pack_kernel(const SigmaPack pk, int n, float sd, float* __restrict__ coef) {
    const int b = threadIdx.x;
    if (b >= n) return;
    const float s = pk.s[b];
    const float s2 = s * s, d2 = sd * sd;
    coef[b * 4 + 0] = 1.0f / sqrtf(s2 + d2);
    coef[b * 4 + 1] = logf(s) * 0.25f;
    coef[b * 4 + 2] = d2 / (s2 + d2);
    coef[b * 4 + 3] = s * sd * (1.0f / sqrtf(d2 + s2));
}
const char* launch_edm_coef_list(const float* sigmas_host, int n, float sigma_data, float* coef, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 256) {
        SigmaPack pk;
        const int m = n - i0 < 256 ? n - i0 : 256;
        for (int i = 0; i < 256; ++i) pk.s[i] = i < m ? sigmas_host[i0 + i] : 1.0f;
        hipLaunchKernelGGL(edm_coef_pack_kernel, dim3(1), dim3(256), 0, s, pk, m, sigma_data, coef + (size_t)i0 * 4);
        if (hipGetLastError() != hipSuccess) return "edm_coef_list: launch failed";
    }
    return nullptr;
}

const char* launch_edm_coef(const float* sigmas_dev, float sigma_scalar, int nb, float sigma_data, float* coef, hipStream_t s) {
    hipLaunchKernelGGL(edm_coef_kernel, dim3(ceil_div(nb, 64)), dim3(64), 0, s, sigmas_dev, sigma_scalar, nb, sigma_data, coef);
    return ADF_LAUNCH_CHECK("edm_coef");
}

// =====================================================================================================
// sigma embedding: LearnedPositionalEmbedding + MLP (unet1d.py:128-148, 678-684)
// =====================================================================================================
__global__ void __launch_bounds__(256) time_embed_kernel(const TimeEmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ch = a.ch, td = 4 * ch, half = ch / 2;
    float* feat = (float*)smem;          // [ch + 1]
    float* h1 = feat + (ch + 1);         // [td]
    const int b = blockIdx.x;
    const float t = a.t[(size_t)b * a.t_stride];
    for (int i = threadIdx.x; i < ch + 1; i += 256) {
        float v;
        if (i == 0) v = t;
        else if (i <= half) v = sinf(t * a.fourier[i - 1] * 6.28318530717958647692f);
        else v = cosf(t * a.fourier[i - 1 - half] * 6.28318530717958647692f);
        feat[i] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < td; j += 256) {
        float acc = a.b1[j];
        const float* wr = a.w1 + (size_t)j * (ch + 1);
        for (int i = 0; i < ch + 1; ++i) acc = fmaf(wr[i], feat[i], acc);
        h1[j] = silu_f(acc);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < td; j += 256) {
        float acc = a.b2[j];
        const float* wr = a.w2 + (size_t)j * td;
        for (int i = 0; i < td; ++i) acc = fmaf(wr[i], h1[i], acc);
        a.temb[(size_t)b * td + j] = acc;
    }
}

const char* launch_time_embed(const TimeEmbedArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(a.ch + 1 + 4 * a.ch) * sizeof(float);
    hipLaunchKernelGGL(time_embed_kernel, dim3(a.nb), dim3(256), lds, s, a);
    return ADF_LAUNCH_CHECK("time_embed");
}

// FiLM projections of every resblock in one launch (unet1d.py:269-276, 306-310): one wave per output row.
// film[b][j] = bias[j] + sum_{i < in_dim} W[j][w_col0 + i] * silu(in[b][i]);  W rows are ldw floats long (the
// reference Linear reads cat(time_embed, class_embed): the two column ranges are projected separately).
__global__ void __launch_bounds__(256) film_kernel(const float* __restrict__ in, int in_dim, const float* __restrict__ w, int ldw,
                                                   int w_col0, const float* __restrict__ bias, float* __restrict__ film, int nb, int total) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= total) return;
    const float* wr = w + (size_t)j * ldw + w_col0;
    const float bj = bias ? bias[j] : 0.f;
    for (int b = 0; b < nb; ++b) {
        float acc = 0.f;
        for (int i = lane; i < in_dim; i += 64) acc = fmaf(wr[i], silu_f(in[(size_t)b * in_dim + i]), acc);
        acc = wave_sum(acc);
        if (lane == 0) film[(size_t)b * total + j] = acc + bj;
    }
}

const char* launch_film(const float* in, int in_dim, const float* w, int ldw, int w_col0, const float* bias, float* film, int nb,
                        int total, hipStream_t s) {
    hipLaunchKernelGGL(film_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, s, in, in_dim, w, ldw, w_col0, bias, film, nb, total);
    return ADF_LAUNCH_CHECK("film");
}

// LabelEmbedder (conditioner.py:92-111): row b < nb-1 takes label_emb[classes[b]] (or the null embedding when
// null_all), the last row always the null embedding; then LayerNorm -> Linear -> SiLU -> Linear.  One block per row.
__global__ void __launch_bounds__(256) class_embed_kernel(const long long* __restrict__ classes, int num_classes, int null_all,
                                                          const float* __restrict__ emb, const float* __restrict__ null_emb,
                                                          const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2, int ch, int cdim,
                                                          float* __restrict__ out, int nrows) {
    __shared__ float e[512];
    __shared__ float hdn[2048];
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool use_null = b == nrows - 1 || null_all;
    long long cls = use_null ? 0 : classes[b];
    cls = cls < 0 ? 0 : (cls >= num_classes ? num_classes - 1 : cls);      // the host checks the range; never read outside
    const float* src = use_null ? null_emb : emb + (size_t)cls * ch;
    float v0 = 0.f;
    for (int i = tid; i < ch; i += 256) { e[i] = src[i]; v0 += src[i]; }
    v0 = wave_sum(v0);
    if ((tid & 63) == 0) red[tid >> 6] = v0;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)ch;
    __syncthreads();
    float v1 = 0.f;
    for (int i = tid; i < ch; i += 256) { const float d = e[i] - mean; v1 += d * d; }
    v1 = wave_sum(v1);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = v1;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)ch + 1e-5f);
    for (int i = tid; i < ch; i += 256) e[i] = (e[i] - mean) * rstd * lnw[i] + lnb[i];
    __syncthreads();
    for (int j = tid; j < cdim; j += 256) {
        float acc = b1[j];
        for (int i = 0; i < ch; ++i) acc = fmaf(w1[(size_t)j * ch + i], e[i], acc);
        hdn[j] = silu_f(acc);
    }
    __syncthreads();
    for (int j = tid; j < cdim; j += 256) {
        float acc = b2[j];
        for (int i = 0; i < cdim; ++i) acc = fmaf(w2[(size_t)j * cdim + i], hdn[i], acc);
        out[(size_t)b * cdim + j] = acc;
    }
}

const char* launch_class_embed(const long long* classes, int num_classes, int null_all, const float* emb, const float* null_emb,
                               const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2, const float* b2,
                               int ch, int cdim, float* out, int nrows, hipStream_t s) {
    if (ch > 512 || cdim > 2048) return "class_embed: channels too large";
    hipLaunchKernelGGL(class_embed_kernel, dim3(nrows), dim3(256), 0, s, classes, num_classes, null_all, emb, null_emb, lnw, lnb, w1, b1,
                       w2, b2, ch, cdim, out, nrows);
    return ADF_LAUNCH_CHECK("class_embed");
}

// Classifier-free guidance + EDM preconditioning (diffusion.py:52-59): out = clamp(c_skip x + c_out (n + (c - n) s), -1, 1); clampit = 0 leaves the
// estimate unclipped for the dynamic threshold below (fc == fn, scale 1: the unguided estimate)
__global__ void __launch_bounds__(256) cfg_combine_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ fc,
                                                          const float* __restrict__ fn, const float* __restrict__ coef, int coef_bstride,
                                                          float scale, long long per_sample, long long n, int clampit) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / per_sample;
        const float c_skip = coef[b * coef_bstride + 2], c_out = coef[b * coef_bstride + 3];
        const float nl = fn[i];
        const float pred = nl + (fc[i] - nl) * scale;
        const float v = c_skip * x[i] + c_out * pred;
        out[i] = clampit ? fminf(fmaxf(v, -1.0f), 1.0f) : v;
    }
}

const char* launch_cfg_combine(float* out, const float* x, const float* fc, const float* fn, const float* coef, int coef_bstride,
                               float scale, long long per_sample, long long n, int clampit, hipStream_t s) {
    long long g = (n + 255) / 256;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g))), dim3(256), 0, s, out, x, fc, fn, coef,
                       coef_bstride, scale, per_sample, n, clampit);
    return ADF_LAUNCH_CHECK("cfg_combine");
}

// Dynamic thresholding (components/utils.py:23-33): per sample, scale = max(1, quantile(|x|, q)) with torch.quantile's linear interpolation
// between the two order statistics around rank q (n - 1) (ATen quantile_compute: ranks in fp32, lerp), then x = clamp(x, -scale, scale) / scale.
// The order statistics are EXACT: a radix select over the bit patterns of |x| (monotonic for non-negative floats), four 8-bit passes with a
// 256-bin LDS histogram each, one workgroup per sample; the upper neighbour is the same value when enough elements tie, else the minimum of the
// larger ones (a fifth pass).  NaNs are not expected (torch.quantile would return NaN).
__global__ void __launch_bounds__(1024) dyn_scale_kernel(const float* __restrict__ x, long long per_sample, float q, float* __restrict__ scale_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_k, sh_le, sh_min;
    const int tid = threadIdx.x;
    const float* const xb = x + (size_t)blockIdx.x * per_sample;
    const float ranks = q * (float)(per_sample - 1);                      // fp32, as ATen computes it
    const long long below = (long long)floorf(ranks), above = (long long)ceilf(ranks);
    const float weight = ranks - (float)below;
    unsigned prefix = 0u;                                                // the bytes of the answer found so far (high to low)
    unsigned k = (unsigned)below;                                        // rank of the answer among the elements that match `prefix`
    unsigned eq = 0u;                                                    // elements in the answer's bin (after the last pass: equal to it)
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned mask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
        for (long long i = tid; i < per_sample; i += 1024) {
            const unsigned u = __float_as_uint(xb[i]) & 0x7fffffffu;
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                                   // wave 0: lane l owns bins 4 l .. 4 l + 3
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
            const unsigned exc = inc - tot;                               // matching elements in the bins before this lane's
            if (k >= exc && k < inc) {                                    // exactly one lane
                unsigned c = exc; int bin = 4 * tid;
                if (k >= c + h0) { c += h0; ++bin; if (k >= c + h1) { c += h1; ++bin; if (k >= c + h2) { c += h2; ++bin; } } }
                sh_prefix = prefix | ((unsigned)bin << shift);
                sh_k = k - c;                                             // the c elements of the smaller bins are smaller than the answer
                sh_le = hist[bin];
            }
        }
        __syncthreads();
        prefix = sh_prefix; k = sh_k; eq = sh_le;
        __syncthreads();
    }
    const float v_below = __uint_as_float(prefix);
    float v_above = v_below;
    // (below - k) elements are smaller than v_below, eq equal to it: the element of rank `above` is v_below itself unless above >= that count
    if (above != below && (unsigned long long)above >= (unsigned long long)((unsigned)below - k) + eq) {
        if (tid == 0) sh_min = 0x7f800000u;
        __syncthreads();
        unsigned m = 0x7f800000u;
        for (long long i = tid; i < per_sample; i += 1024) {
            const unsigned u = __float_as_uint(xb[i]) & 0x7fffffffu;
            if (u > prefix && u < m) m = u;
        }
        atomicMin(&sh_min, m);
        __syncthreads();
        v_above = __uint_as_float(sh_min);
    }
    if (tid == 0) {
        const float d = v_above - v_below;
        const float r = fabsf(weight) < 0.5f ? v_below + weight * d : v_above - d * (1.0f - weight);    // at::lerp
        scale_out[blockIdx.x] = fmaxf(r, 1.0f);
    }
}
__global__ void __launch_bounds__(256) dyn_apply_kernel(float* __restrict__ x, const float* __restrict__ scale, long long per_sample, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = scale[i / per_sample];
        x[i] = __fdiv_rn(fminf(fmaxf(x[i], -s), s), s);
    }
}
const char* launch_dyn_threshold(float* x, int B, long long per_sample, float q, float* scale_scratch, hipStream_t s) {
    if (per_sample < 1 || per_sample > 0x7fffffffll) return "dyn_threshold: bad sample size";
    if (!(q > 0.0f) || q > 1.0f) return "dyn_threshold: the quantile must be in (0, 1]";
    hipLaunchKernelGGL(dyn_scale_kernel, dim3((unsigned)B), dim3(1024), 0, s, x, per_sample, q, scale_scratch);
    const long long n = (long long)B * per_sample;
    long long g = (n + 255) / 256;
    hipLaunchKernelGGL(dyn_apply_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, x, scale_scratch, per_sample, n);
    return ADF_LAUNCH_CHECK("dyn_threshold");
}

// =====================================================================================================
// Sampler state updates (sampler_edm.py:333-369 Heun/churn, :251-282 RK2-alpha, :624-690 DPM multistep)
// Operation order follows the reference expressions so fp32 rounding matches as closely as possible.
// =====================================================================================================
static inline unsigned ew_grid(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
#define ADF_EW_LOOP for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)

__global__ void __launch_bounds__(256) scale_kernel(float* o, const float* x, float s, long long n) { ADF_EW_LOOP o[i] = s * x[i]; }
__global__ void __launch_bounds__(256) churn_kernel(float* o, const float* x, const float* eps, float c, float sn, long long n) {
    ADF_EW_LOOP o[i] = x[i] + c * (sn * eps[i]);
}
__global__ void __launch_bounds__(256) euler_kernel(float* xn, float* d, const float* x, const float* den, float sigma, float dt, long long n) {
    ADF_EW_LOOP { const float dd = (x[i] - den[i]) / sigma; d[i] = dd; xn[i] = x[i] + dt * dd; }
}
// x_next = x_base + ((x_eval - den) / sigma) * dt: the update of the DPM2-family steps, whose derivative is taken at
// x_eval but applied to x_base (sampler_edm.py:448-464, stochastic_sampler_edm.py:69-81)
__global__ void __launch_bounds__(256) dstep_kernel(float* xn, const float* xb, const float* xe, const float* den, float sigma, float dt,
                                                    long long n) {
    ADF_EW_LOOP { const float dd = (xe[i] - den[i]) / sigma; xn[i] = xb[i] + dd * dt; }
}
// LMSSampler step (sampler_edm.py:1170-1188): d = (x - den) / sigma is stored as the newest derivative, then
// x += ((c0*d + c1*d1) + c2*d2) + c3*d3 in the reference's order of additions
__global__ void __launch_bounds__(256) lms_kernel(float* x, const float* den, float sigma, const LmsArgs a, long long n) {
    ADF_EW_LOOP {
        const float xv = x[i];
        const float d = (xv - den[i]) / sigma;
        a.dcur[i] = d;
        float acc = a.c[0] * d;
        if (a.order > 1) acc = acc + a.c[1] * a.d1[i];
        if (a.order > 2) acc = acc + a.c[2] * a.d2[i];
        if (a.order > 3) acc = acc + a.c[3] * a.d3[i];
        x[i] = xv + acc;
    }
}
// single-step DPM-Solver combinations (sampler_edm.py:568-622, x0 prediction): out = (a*x - b*e0) + c*(e1 - e0)
__global__ void __launch_bounds__(256) lincomb_kernel(float* out, const float* x, const float* e0, const float* e1, float a, float b,
                                                      float c, int clampit, long long n) {
    ADF_EW_LOOP {
        const float e = e0[i];
        float v = a * x[i] - b * e;
        if (e1) v = v + c * (e1[i] - e);
        if (clampit) v = fminf(fmaxf(v, -1.0f), 1.0f);
        out[i] = v;
    }
}
// noise prediction from the denoised estimate, in place: m = (x - m) / sigma (DPMSampler.model_fn, sampler_edm.py:700-706)
__global__ void __launch_bounds__(256) eps_kernel(float* m, const float* x, float sigma, long long n) { ADF_EW_LOOP m[i] = (x[i] - m[i]) / sigma; }
__global__ void __launch_bounds__(256) reflow_kernel(float* m, const float* x, float sigma, long long n) { ADF_EW_LOOP m[i] = x[i] - m[i] * sigma; }
// DPM-Solver++(2M) update (sampler_edm.py:1096-1107): out = ratio*x - coef*(c1*d - c2*d_old); first step / final sigma 0: d_old null
__global__ void __launch_bounds__(256) dpm2m_kernel(float* out, const float* x, const float* d, const float* d_old, float ratio, float coef,
                                                    float c1, float c2, long long n) {
    ADF_EW_LOOP {
        float v = d[i];
        if (d_old) v = c1 * v - c2 * d_old[i];
        out[i] = ratio * x[i] - coef * v;
    }
}
__global__ void __launch_bounds__(256) rk2_kernel(float* xn, const float* x, const float* d, const float* xe, const float* den2,
                                                  float sigma2, float h, float w1, float w2, long long n) {
    ADF_EW_LOOP { const float d2 = (xe[i] - den2[i]) / sigma2; xn[i] = x[i] + h * (w1 * d[i] + w2 * d2); }
}
__global__ void __launch_bounds__(256) dpm_kernel(float* xo, const float* x, const DpmArgs a, int clampit, long long n) {
    ADF_EW_LOOP {
        const float m0 = a.m0[i];
        float v = a.ratio * x[i] - a.phi1 * m0;
        if (a.order == 2) {
            const float d10 = a.inv_r0 * (m0 - a.m1[i]);
            v = v - 0.5f * a.phi1 * d10;
        } else if (a.order == 3) {
            const float m1 = a.m1[i];
            const float d10 = a.inv_r0 * (m0 - m1);
            const float d11 = a.inv_r1 * (m1 - a.m2[i]);
            const float d1 = d10 + a.r0_frac * (d10 - d11);
            const float d2 = a.inv_r01 * (d10 - d11);
            v = v + a.phi2 * d1 - a.phi3 * d2;
        }
        if (clampit) v = fminf(fmaxf(v, -1.0f), 1.0f);
        xo[i] = v;
    }
}
__global__ void __launch_bounds__(256) clamp_kernel(float* x, long long n) { ADF_EW_LOOP x[i] = fminf(fmaxf(x[i], -1.0f), 1.0f); }

const char* launch_scale(float* out, const float* in, float s, long long n, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, st, out, in, s, n);
    return ADF_LAUNCH_CHECK("scale");
}
const char* launch_churn(float* x_hat, const float* x, const float* eps, float c, float s_noise, long long n, hipStream_t st) {
    hipLaunchKernelGGL(churn_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x_hat, x, eps, c, s_noise, n);
    return ADF_LAUNCH_CHECK("churn");
}
const char* launch_euler(float* x_next, float* d, const float* x, const float* den, float sigma, float dt, long long n, hipStream_t st) {
    hipLaunchKernelGGL(euler_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x_next, d, x, den, sigma, dt, n);
    return ADF_LAUNCH_CHECK("euler");
}
const char* launch_rk2(float* x_next, const float* x, const float* d, const float* x_e, const float* den2, float sigma2,
                       float h, float w1, float w2, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rk2_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x_next, x, d, x_e, den2, sigma2, h, w1, w2, n);
    return ADF_LAUNCH_CHECK("rk2");
}
const char* launch_dpm_update(float* x_out, const float* x, const DpmArgs& a, int clamp, long long n, hipStream_t st) {
    hipLaunchKernelGGL(dpm_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x_out, x, a, clamp, n);
    return ADF_LAUNCH_CHECK("dpm_update");
}
const char* launch_reflow(float* m, const float* x, float sigma, long long n, hipStream_t st) {
    hipLaunchKernelGGL(reflow_kernel, dim3(ew_grid(n)), dim3(256), 0, st, m, x, sigma, n);
    return ADF_LAUNCH_CHECK("reflow");
}
const char* launch_eps(float* m, const float* x, float sigma, long long n, hipStream_t st) {
    hipLaunchKernelGGL(eps_kernel, dim3(ew_grid(n)), dim3(256), 0, st, m, x, sigma, n);
    return ADF_LAUNCH_CHECK("eps");
}
const char* launch_dpm2m(float* out, const float* x, const float* d, const float* d_old, float ratio, float coef, float c1, float c2, long long n,
                         hipStream_t st) {
    hipLaunchKernelGGL(dpm2m_kernel, dim3(ew_grid(n)), dim3(256), 0, st, out, x, d, d_old, ratio, coef, c1, c2, n);
    return ADF_LAUNCH_CHECK("dpm2m");
}
const char* launch_lms(float* x, const float* den, float sigma, const LmsArgs& a, long long n, hipStream_t st) {
    hipLaunchKernelGGL(lms_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, den, sigma, a, n);
    return ADF_LAUNCH_CHECK("lms");
}
const char* launch_lincomb(float* out, const float* x, const float* e0, const float* e1, float a, float b, float c, int clampit,
                           long long n, hipStream_t st) {
    hipLaunchKernelGGL(lincomb_kernel, dim3(ew_grid(n)), dim3(256), 0, st, out, x, e0, e1, a, b, c, clampit, n);
    return ADF_LAUNCH_CHECK("lincomb");
}
const char* launch_dstep(float* xn, const float* xb, const float* xe, const float* den, float sigma, float dt, long long n, hipStream_t st) {
    hipLaunchKernelGGL(dstep_kernel, dim3(ew_grid(n)), dim3(256), 0, st, xn, xb, xe, den, sigma, dt, n);
    return ADF_LAUNCH_CHECK("dstep");
}
__global__ void __launch_bounds__(256) unipc_kernel(float* __restrict__ out, const float* __restrict__ x, const UniPcArgs a, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float m0 = a.m0[i];
        const float xt = a.a * x[i] - a.hp * m0;
        float res = 0.f;
        if (a.K > 0) res = res + a.rho[0] * ((a.m[0][i] - m0) / a.rk[0]);
        if (a.K > 1) res = res + a.rho[1] * ((a.m[1][i] - m0) / a.rk[1]);
        if (a.mt) res = res + a.rho_t * (a.mt[i] - m0);
        out[i] = xt - a.sb * res;
    }
}
const char* launch_unipc(float* out, const float* x, const UniPcArgs& a, long long n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(unipc_kernel, dim3(blocks), dim3(256), 0, s, out, x, a, n);
    return ADF_LAUNCH_CHECK("unipc");
}

const char* launch_clamp(float* x, long long n, hipStream_t st) {
    hipLaunchKernelGGL(clamp_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, n);
    return ADF_LAUNCH_CHECK("clamp");
}

// =====================================================================================================
// NLC (T) -> NCL fp32 copy of an internal activation (debug taps used by the parity tests)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) nlc_to_ncl_kernel(const T* __restrict__ x, float* __restrict__ y, int L, int C, long long n) {
    ADF_EW_LOOP {
        const int l = (int)(i % L);
        const int c = (int)((i / L) % C);
        const long long b = i / ((long long)L * C);
        y[i] = Elem<T>::ld(x + ((size_t)b * L + l) * C + c);
    }
}
const char* launch_nlc_to_ncl_f32(const void* x, float* y, int bf16, int B, int L, int C, hipStream_t st) {
    const long long n = (long long)B * L * C;
    if (bf16) hipLaunchKernelGGL(nlc_to_ncl_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, y, L, C, n);
    else hipLaunchKernelGGL(nlc_to_ncl_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, y, L, C, n);
    return ADF_LAUNCH_CHECK("nlc_to_ncl");
}

// =====================================================================================================
// Weight packing: dst[chunk][tap][n][kc]  (row of 128 bytes = KC elements of K)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int mode, int cout,
                                                          int cin, int K, int f, int n_offset, int n_rows, int n_pad, int nchunk,
                                                          int taps) {
    constexpr int KC = kRowBytesPack / (int)sizeof(T);
    const long long total = (long long)nchunk * taps * n_rows * KC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int kc = (int)(i % KC);
        long long r = i / KC;
        const int nl = (int)(r % n_rows); r /= n_rows;
        const int tap = (int)(r % taps);
        const int chunk = (int)(r / taps);
        const int ci = chunk * KC + kc;
        float v = 0.f;
        if (mode == 2) {
            // strided conv folded to stride 1: the input [L][cin] is read as [L/f][f*cin], channel j*cin + c of folded
            // row p is x[f*p + j][c], so folded tap t' carries the original taps t'*f + j (zero beyond K)
            if (ci < f * cin) {
                const int j = ci / cin, c = ci - j * cin, t = tap * f + j;
                if (t < K) v = src[((size_t)nl * cin + c) * K + t];
            }
        } else if (ci < cin) {
            if (mode == 0) {
                v = src[((size_t)nl * cin + ci) * K + tap];
            } else if (mode == 3) {
                // ConvTranspose1d(K = 2 f, stride f, padding f / 2) as a 3-tap conv over the input rows, column p * cout + co = output row f j + p:
                // x[j - 1] carries kernel tap p + 3 f / 2 (p < f / 2), x[j] tap p + f / 2, x[j + 1] tap p - f / 2 (p >= f / 2)
                const int p = nl / cout, co = nl - p * cout;
                const int k = tap == 1 ? p + f / 2 : (tap == 0 ? (p < f / 2 ? p + f + f / 2 : -1) : (p >= f / 2 ? p - f / 2 : -1));
                if (k >= 0) v = src[((size_t)ci * cout + co) * K + k];
            } else {
                const int p = nl / cout, co = nl - p * cout;
                v = src[((size_t)ci * cout + co) * K + (p + tap * f)];
            }
        }
        const size_t o = (((size_t)chunk * taps + tap) * n_pad + n_offset + nl) * KC + kc;
        if constexpr (IsX3<T>::value) {
            // split-bf16 row (adf_common.h): hi part of K element kc at bf16 index kc of the row, lo part at 32 + kc (slots kc >> 3 and 4 + (kc >> 3))
            unsigned short* row = (unsigned short*)(dst + (o - kc));
            const unsigned short hi = f32_to_bf16_hw(v);
            row[kc] = hi;
            row[32 + kc] = f32_to_bf16_hw(v - bf16_to_f32(hi));
        } else {
            Elem<T>::st(dst + o, v);
        }
    }
}

// dtype: 0 = fp32 rows, 1 = bf16 rows, 2 = split-bf16 rows (hi | lo halves of a 32-element fp32 row, f32x3_t)
const char* launch_pack_weight(const float* src, void* dst, int dtype, int mode, int cout, int cin, int K, int f,
                               int n_offset, int n_pad, int nchunk, hipStream_t s) {
    const int bf16 = dtype == 1;
    const int taps = mode == 0 ? K : (mode == 2 ? (K - 1) / f + 1 : (mode == 3 ? 3 : 2));
    if ((mode == 1 || mode == 3) && K != 2 * f) return "pack_weight: transposed conv needs K == 2*factor";
    if (mode == 3 && f % 2) return "pack_weight: the 3-tap form of a transposed conv needs an even factor";
    if (mode == 2 && (f < 1 || (K - 1) % f)) return "pack_weight: folded strided conv needs K == factor*k + 1";
    // only the real rows are written; the destination is zero-initialised at allocation (row / K padding)
    const int n_rows = (mode == 1 || mode == 3) ? f * cout : cout;
    if (n_offset + n_rows > n_pad) return "pack_weight: rows exceed n_pad";
    const int kc = kRowBytesPack / (bf16 ? 2 : 4);
    const long long total = (long long)nchunk * taps * n_rows * kc;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (bf16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, mode, cout, cin, K, f, n_offset, n_rows, n_pad, nchunk, taps);
    else if (dtype == 2) hipLaunchKernelGGL(pack_weight_kernel<f32x3_t>, dim3(grid), dim3(256), 0, s, src, (f32x3_t*)dst, mode, cout, cin, K, f, n_offset, n_rows, n_pad, nchunk, taps);
    else hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, mode, cout, cin, K, f, n_offset, n_rows, n_pad, nchunk, taps);
    return ADF_LAUNCH_CHECK("pack_weight");
}

// Second copy of a packed bf16 1x1 weight in MFMA-fragment order for adf_transformer.h: dst[K step][half][n][8 channels],
// i.e. the 16-byte B fragments of one K step (16 channels) are contiguous over the output columns.
__global__ void __launch_bounds__(256) repack_frag_kernel(const char* __restrict__ src, char* __restrict__ dst, int n_offset, int n_rows,
                                                          int n_pad, int nchunk) {
    const long long total = (long long)nchunk * n_rows * 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int u = (int)(i & 7);                      // 16-byte unit of the 128-byte row: K step u >> 1 of the chunk, half u & 1
        const int n = n_offset + (int)((i >> 3) % n_rows);
        const int chunk = (int)((i >> 3) / n_rows);
        const u32x4_t v = *(const u32x4_t*)(src + ((size_t)chunk * n_pad + n) * 128 + u * 16);
        *(u32x4_t*)(dst + (((size_t)(chunk * 4 + (u >> 1)) * 2 + (u & 1)) * n_pad + n) * 16) = v;
    }
}
const char* launch_repack_frag(const void* src, void* dst, int n_offset, int n_rows, int n_pad, int nchunk, hipStream_t s) {
    const long long total = (long long)nchunk * n_rows * 8;
    const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(repack_frag_kernel, dim3(grid), dim3(256), 0, s, (const char*)src, (char*)dst, n_offset, n_rows, n_pad, nchunk);
    return ADF_LAUNCH_CHECK("repack_frag");
}

__global__ void __launch_bounds__(256) upsample_nearest_pad_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ out, int L, int cpr, int f, long long total) {
    const int rows = f * L + 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        long long r = i / cpr;
        const int row = (int)(r % rows);
        const long long b = r / rows;
        int ui = row - 1;                                     // row of the upsampled signal; reflection about its first / last sample
        if (ui < 0) ui = 1;
        if (ui >= f * L) ui = f * L - 2;
        out[i] = x[(b * L + ui / f) * cpr + c];
    }
}
const char* launch_upsample_nearest_pad(const void* x, void* out, int bf16, int B, int L, int C, int f, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc || f < 1 || f * L < 2) return "upsample_nearest_pad: channels must be a multiple of a 16-byte chunk, f * L >= 2";
    const int cpr = C / epc;
    const long long total = (long long)B * (f * L + 2) * cpr;
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(upsample_nearest_pad_kernel, dim3(grid), dim3(256), 0, s, (const u32x4_t*)x, (u32x4_t*)out, L, cpr, f, total);
    return ADF_LAUNCH_CHECK("upsample_nearest_pad");
}

}  // namespace adf
