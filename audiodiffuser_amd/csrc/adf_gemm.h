// Fused implicit-GEMM kernel for every conv1d / conv_transpose1d / linear on the U-Net path
// (reference call sites: src/models/backbones/unet1d.py:193-207 ConvBlock1d, :297-316 ResnetBlock1d,
//  :214-225 Downsample1d, :248-255 Upsample1d, :49-61 FeedForward1d, attention_utils.py:117,157,184).
//
// Layout: activations are channels-last  [B][L][C]  ("NLC", the reference's tensors are [B][C][L]);
// only the network's C=1 waveform input/output cross the boundary, so the internal layout is free.
// GEMM view:  out[m, n] = sum_seg sum_tap sum_ci  f(A[m*stride + off0 + tap*step, ci]) * W[seg][tap][n][ci]
//   M = positions of one sample, N = output channels (x phases for transposed conv), K = taps x Cin.
//   f() = fused prologue: per-(sample, channel) affine (GroupNorm apply + FiLM folded) then SiLU,
//   applied while the activation tile is staged into LDS; conv zero padding is applied after f().
//   A second K segment (raw skip-concat input, 1x1) implements ResnetBlock1d.to_out inside the
//   same accumulators.  The channel concat of the up path is never materialised: a segment reads
//   two source tensors.
// Tiling: an M tile either lies inside one sample ("per-sample" mode) or, for short sequences with raw
//   inputs, covers several whole samples ("flat" mode: the LDS image keeps one zero-padded segment per
//   sample so taps never leak across sample boundaries).
// Epilogue: accumulators (+bias) are staged through LDS as fp32, then written with 16-byte stores along
//   the contiguous channel axis (identity residual, GELU and the transposed-conv phase scatter applied
//   there), and the GroupNorm statistics of the produced tensor are reduced from the same LDS image
//   (fp64 atomics into [B][G][2]).
//
// MFMA: 32x32x16 bf16 (throughput mode) or 32x32x2 f32 (parity mode, exact fp32 FMA chain); a wave
// owns an (MT*32) x (NT*32) accumulator tile; K is walked in 128-byte rows (64 bf16 / 32 fp32).
// LDS rows hold 128 B of K padded to a 144-byte pitch: with that pitch the ds_read_b128 fragment reads
// (lane = row, 16 lanes per LDS cycle) hit 16 distinct 4-bank slots, i.e. they are conflict-free on gfx950's
// 64-bank LDS, and every fragment / staging address is "per-thread base + compile-time immediate".
#pragma once
#include "adf_common.h"

namespace adf {

constexpr int kARows = 144;     // activation rows staged per tile (incl. halo / per-sample padding)
constexpr int kTapGroup = 3;    // taps of weights staged per iteration

struct GemmSeg {
    const void* src0;
    const void* src1;   // optional second source, concatenated along channels
    int c0, c1;
    const float* ab;    // [B][c0+c1][2] fused affine, or nullptr = raw input
    GnFinalizeArgs gn;  // gn.gamma != nullptr: `ab` (= gn.ab) has NOT been computed yet.  launch_conv_gemm either lets the
                        // kernel derive the table from the statistics itself (DMA kernel) or launches gn_finalize first.
    float scale1;       // raw input: multiplier of source 1 (skip scale)
    int act;            // 1 = SiLU after the affine
    int taps, stride, off0, step;
    const void* w;      // packed [nchunk][taps][n_pad][row of 128 B]
    const void* wfrag;  // optional second copy in MFMA-fragment order [K step][2][n_pad][8] (bf16; adf_gemm_tile.h)
    int nchunk;
};

struct GemmArgs {
    GemmSeg seg[2];
    int nseg;
    int B, lin, mrows, n, n_pad;
    int gn_ready;        // replay of a recorded launch: the `ab` tables are already filled, do not launch gn_finalize again
    int flat;            // set by the launcher: tiles run over the flattened B*mrows rows
    int seg_rows;        // set by the launcher: rows of one sample inside a tile (flat mode)
    const float* bias0;
    const float* bias1;
    int bias_mod;        // bias index = n % bias_mod
    const void* res;     // identity residual, same layout as out (plain mode only)
    int gelu;
    void* out;
    int out_rows, out_c;
    int scatter_f, scatter_pad;  // transposed conv: n = phase*out_c + co -> row m*f + phase - pad
    int phase_c;                 // > 0: a transposed conv in its 3-tap form (n = f * phase_c phase-major columns, out_c = n): the GroupNorm statistics
                                 // of the output are per group of the phase_c CHANNELS (column n is channel n % phase_c).  Only conv_gemm_rb_kernel
                                 // takes this form: ask conv_gemm_phase_eligible() before building it
    double* stats;       // optional [B][stats_groups][2] (sum, sumsq) of the produced tensor
    int stats_groups;
};

// out[k] = bias0[bi[k]] + bias1[bi[k]] (absent vectors = 0) for ok[k], else 0 -- with EVERY load issued unconditionally and before the first use (absent
// vectors through a valid dummy pointer): as `if (a.bias0) b += a.bias0[bi]; if (a.bias1) ...` per element each load was waited for on the spot, 2 N
// serialised memory round trips (16 in the split-K epilogue).  See "Waits the compiler adds" in DESIGN.md.
template <int N>
__device__ __forceinline__ void gemm_bias_load(const GemmArgs& a, const int (&bi)[N], const bool (&ok)[N], float (&out)[N]) {
    const bool h0 = a.bias0 != nullptr, h1 = a.bias1 != nullptr;                      // uniform
    const float* const dmy = (const float*)a.seg[0].w;
    const float* const p0 = h0 ? a.bias0 : dmy;
    const float* const p1 = h1 ? a.bias1 : dmy;
    float v0[N], v1[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = ok[k] ? bi[k] : 0;
        v0[k] = p0[h0 ? idx : 0]; v1[k] = p1[h1 ? idx : 0];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = ok[k] ? (h0 ? v0[k] : 0.f) + (h1 ? v1[k] : 0.f) : 0.f;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// Staging loads are UNCONDITIONAL: lanes with nothing to load read offset 0 of the same (global) buffer and
// the value is discarded / zeroed afterwards.  A branch around a load makes hipcc serialise the loads behind
// vmcnt(0) waits, and a select between pointers of different address spaces turns them into flat loads.

constexpr int kLdsPitch = 144;   // bytes between LDS rows (128 B of K + 16 B pad)
__device__ __forceinline__ int lds_swz(int row, int c16) { return row * kLdsPitch + (c16 << 4); }

// Asynchronous 16-byte global load that hipcc does not track: no compiler-inserted s_waitcnt, the data is
// only valid after the matching hand-counted wait (ADF_VMWAIT_*), which also names the destination registers so
// that no consumer can be scheduled above it (cdna_hip_programming.md 5.7, form (ii)).
#define ADF_GLOAD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

// A kernarg pointer as a guaranteed scalar: a per-lane select between two struct fields otherwise becomes a
// per-lane select of the fields' ADDRESSES followed by a global_load of the pointer (plus its vmcnt(0) wait).
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

template <int TM, int TN>
constexpr int gemm_lds_bytes() {
    constexpr int main_b = kARows * kLdsPitch + kTapGroup * TN * kLdsPitch;
    constexpr int epi_b = TM * TN * 4;
    return main_b > epi_b ? main_b : epi_b;
}

// Software-pipelined MFMA sequence for one staged K chunk (bf16): the fragments of step s+1 are read from LDS
// while the MFMAs of step s issue (double-buffered fragment registers, fully unrolled so every index is static).
// addrA(tap, i, c16) / addrW(tap, j, c16) return the LDS byte address of a 16-byte fragment chunk.
template <int TAPS, int MT, int NT, typename FA, typename FW>
__device__ __forceinline__ void mfma_chunk_bf16(f32x16_t (&acc)[MT][NT], int h, FA addrA, FW addrW) {
    constexpr int NS = TAPS * 4;
    bf16x8_t fa[2][MT], fb[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[0][i] = *(const bf16x8_t*)addrA(0, i, h);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[0][j] = *(const bf16x8_t*)addrW(0, j, h);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        if (st + 1 < NS) {
            const int tap = (st + 1) >> 2, ks = (st + 1) & 3;
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[nxt][i] = *(const bf16x8_t*)addrA(tap, i, ks * 2 + h);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[nxt][j] = *(const bf16x8_t*)addrW(tap, j, ks * 2 + h);
        }
        // pin the order: hipcc otherwise sinks the reads next to their uses and the LDS latency is exposed again
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, int MT, int NT, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) conv_gemm_kernel(const GemmArgs a) {
    constexpr int TM = 32 * MT * WM, TN = 32 * NT * WN, NTHR = 64 * WM * WN;
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int KC = kRowBytes / (int)sizeof(T);
    constexpr int A_CH = (kARows * 8 + NTHR - 1) / NTHR;
    constexpr int W_CH = (kTapGroup * TN * 8 + NTHR - 1) / NTHR;
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr bool kX3 = IsX3<T>::value;              // fp32 storage, operands staged as bf16 hi | lo (adf_common.h)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsW = smem + kARows * kLdsPitch;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    int bid = blockIdx.x;
    const int tn_i = bid % tiles_n; bid /= tiles_n;
    // per-sample mode: bid = b * tiles_m + tm_i ; flat mode: bid = tile over the flattened rows
    const int seg = a.flat ? a.seg_rows : TM;   // rows of one sample inside the tile
    const int nsegs = TM / seg;
    int b0, m0;                                  // first sample of the tile, first row inside that sample
    long long R0;                                // first flattened output row (b * mrows + m)
    if (a.flat) {
        R0 = (long long)bid * TM;
        b0 = (int)(R0 / a.mrows);
        m0 = 0;
    } else {
        const int tiles_m = (a.mrows + TM - 1) / TM;
        const int tm_i = bid % tiles_m;
        b0 = bid / tiles_m;
        m0 = tm_i * TM;
        R0 = (long long)b0 * a.mrows + m0;
    }
    const int n0 = tn_i * TN;
    const int c16 = tid & 7;  // this thread's 16-byte column within a 128-byte row (NTHR % 8 == 0)

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ntg0 = (a.seg[0].taps + kTapGroup - 1) / kTapGroup;
    const int nit0 = a.seg[0].nchunk * ntg0;
    const int ntg1 = a.nseg > 1 ? (a.seg[1].taps + kTapGroup - 1) / kTapGroup : 1;
    const int nit = nit0 + (a.nseg > 1 ? a.seg[1].nchunk * ntg1 : 0);

    u32x4_t ra[A_CH];
    u32x4_t rw[W_CH];
    f32x4_t abq[EPC / 2];    // (a, b) pairs of this thread's EPC channels, as loaded
    float raw_scale = 1.0f;
    unsigned avalid = 0;

    // ---- staging: global -> registers ---------------------------------------------------
    // All offsets are 32-bit (every tensor / packed weight is < 4 GB, checked by the launcher) and every
    // load is unconditional: a lane with nothing to load re-reads offset 0 and the value is dropped.
    // Weight buffers are over-allocated by kTapGroup tap slabs, so the weight loads need no guard at all.
    // weight slab row of this thread for i = 0: the others are a compile-time number of rows further on
    const int wrow0 = tid >> 3;                           // tap_l * TN + n_l for i = 0 (NTHR/8 rows per step)
    int arow_idx[A_CH];          // flattened input row (bb*lin + p) of each staged chunk, or -1
    auto setup_segment = [&](const GemmSeg& sg) {
        const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
        const int segrows = (seg - 1) * sg.stride + sg.taps;   // staged rows per sample segment
        const int nrows = nsegs * segrows;
        const int p_lo = m0 * sg.stride + off_min;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = (tid + i * NTHR) >> 3;
            const int j = a.flat ? row / segrows : 0;
            const int p = p_lo + (row - j * segrows);
            const int bb = b0 + j;
            const bool ok = row < nrows && p >= 0 && p < a.lin && bb < a.B;
            arow_idx[i] = ok ? bb * a.lin + p : -1;
        }
    };
    auto load_regs = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int local = s1 ? it - nit0 : it;
        const int ntg = s1 ? ntg1 : ntg0;
        const int chunk = local / ntg, tg = local - chunk * ntg;
        if (local == 0) setup_segment(sg);
        if (tg == 0) {
            const int ctot = sg.c0 + sg.c1;
            const int cidx = chunk * KC + c16 * EPC;
            const bool cvalid = cidx < ctot;
            const bool from1 = sg.c1 > 0 && cidx >= sg.c0;   // K-padding lanes of a single-source segment stay on src0
            const char* src0u = uniform_ptr(sg.src0);
            const char* src1u = uniform_ptr(sg.src1);
            const char* src = from1 ? src1u : src0u;
            const int c0u = __builtin_amdgcn_readfirstlane(sg.c0), c1u = __builtin_amdgcn_readfirstlane(sg.c1);
            const unsigned rowbytes = (unsigned)(from1 ? c1u : c0u) * (unsigned)sizeof(T);
            const unsigned colbytes = (unsigned)(from1 ? cidx - c0u : cidx) * (unsigned)sizeof(T);
            avalid = 0;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const bool ok = cvalid && arow_idx[i] >= 0;
                const unsigned off = ok ? (unsigned)arow_idx[i] * rowbytes + colbytes : 0u;
                ra[i] = *(const u32x4_t*)(src + off);
                avalid |= (ok ? 1u : 0u) << i;
            }
            // affine table (or, for raw inputs, any valid global address: the values are then ignored in
            // store_lds).  No branch and no use of the loaded values here, so nothing waits on these loads.
            const float* abp = sg.ab ? sg.ab + (cvalid ? (unsigned)(b0 * ctot + cidx) * 2u : 0u) : (const float*)sg.w;
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) abq[e] = *(const f32x4_t*)(abp + e * 4);
            raw_scale = from1 ? sg.scale1 : 1.0f;
        }
        const char* wp = (const char*)sg.w + (size_t)(chunk * sg.taps + tg * kTapGroup) * a.n_pad * kRowBytes;
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int row = wrow0 + i * (NTHR / 8);
            const int tap_l = row / TN, n_l = row - tap_l * TN;     // TN is a power of two: shifts
            const unsigned woff = (unsigned)((tap_l * a.n_pad + n0 + n_l) * kRowBytes + c16 * 16);
            rw[i] = *(const u32x4_t*)(wp + woff);
        }
    };

    // ---- staging: registers -> (fused prologue) -> LDS -------------------------------------
    // SiLU(v) = v / (1 + 2^(-log2e * v)), v = a*x + b: the exp2 argument is a second affine of x, so the
    // per-element cost is 2 packed FMAs + exp2 + packed add + rcp + packed mul.
    const int lds_row0 = tid >> 3;
    auto store_lds = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int local = s1 ? it - nit0 : it;
        const int ntg = s1 ? ntg1 : ntg0;
        const int tg = local % ntg;
        if (tg == 0) {
            const int nrows = nsegs * ((seg - 1) * sg.stride + sg.taps);
            const bool act = sg.act != 0;
            const bool use_ab = sg.ab != nullptr;
            f32x2_t fa2[EPC / 2], fb2[EPC / 2];
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) {
                fa2[e] = use_ab ? f32x2_t{abq[e].x, abq[e].z} : f32x2_t{raw_scale, raw_scale};
                fb2[e] = use_ab ? f32x2_t{abq[e].y, abq[e].w} : f32x2_t{0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int row = lds_row0 + i * (NTHR / 8);
                if (row < nrows) {
                    u32x4_t q = u32x4_t{0u, 0u, 0u, 0u};
                    if ((avalid >> i) & 1u) {
                        float f[EPC];
                        unpack16<T>(ra[i], f);
                        if (act) {
#pragma unroll
                            for (int e = 0; e < EPC / 2; ++e) {
                                const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                                const f32x2_t v2 = x2 * fa2[e] + fb2[e];
                                const f32x2_t z2 = v2 * -1.4426950408889634f;
                                f32x2_t d2 = {__builtin_amdgcn_exp2f(z2.x), __builtin_amdgcn_exp2f(z2.y)};
                                d2 = d2 + 1.0f;
                                const f32x2_t r2 = {__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
                                const f32x2_t y2 = v2 * r2;
                                f[2 * e] = y2.x; f[2 * e + 1] = y2.y;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < EPC / 2; ++e) {
                                const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                                const f32x2_t v2 = x2 * fa2[e] + fb2[e];
                                f[2 * e] = v2.x; f[2 * e + 1] = v2.y;
                            }
                        }
                        q = pack16<T>(f);
                    }
                    if constexpr (kX3) {
                        // split-bf16 operand: the chunk's four values as hi | lo halves of the fragment slots c16 >> 1 and 4 + (c16 >> 1) (adf_common.h)
                        float f4[4];
                        unpack16<float>(q, f4);
                        u32x2_t hi, lo;
                        split_bf16x4(f4, hi, lo);
                        *(u32x2_t*)(ldsA + lds_swz(row, c16 >> 1) + (c16 & 1) * 8) = hi;
                        *(u32x2_t*)(ldsA + lds_swz(row, 4 + (c16 >> 1)) + (c16 & 1) * 8) = lo;
                    } else {
                        *(u32x4_t*)(ldsA + lds_swz(row, c16)) = q;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int row = lds_row0 + i * (NTHR / 8);
            if (row < kTapGroup * TN) *(u32x4_t*)(ldsW + lds_swz(row, c16)) = rw[i];
        }
    };

    // ---- MFMA over one staged (chunk, tap group) ---------------------------------------------
    auto compute = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int local = s1 ? it - nit0 : it;
        const int ntg = s1 ? ntg1 : ntg0;
        const int tg = local % ntg;
        const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
        const int segrows = (seg - 1) * sg.stride + sg.taps;
        int ntap = sg.taps - tg * kTapGroup;
        ntap = ntap > kTapGroup ? kTapGroup : ntap;
        int abase[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ti = (wm * MT + i) * 32 + r;   // tile row
            const int j = a.flat ? ti / seg : 0;
            abase[i] = j * segrows + (ti - j * seg) * sg.stride;
        }
        for (int tap_l = 0; tap_l < ntap; ++tap_l) {
            const int aoff = sg.off0 + (tg * kTapGroup + tap_l) * sg.step - off_min;
            int arow[MT], wrow[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) arow[i] = abase[i] + aoff;
#pragma unroll
            for (int j = 0; j < NT; ++j) wrow[j] = tap_l * TN + (wn * NT + j) * 32 + r;
            if constexpr (kBf16) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8_t fa_[MT], fb_[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa_[i] = *(const bf16x8_t*)(ldsA + lds_swz(arow[i], ks * 2 + h));
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb_[j] = *(const bf16x8_t*)(ldsW + lds_swz(wrow[j], ks * 2 + h));
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
                }
            } else if constexpr (kX3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t ah_[MT], al_[MT], bh_[NT], bl_[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        ah_[i] = *(const bf16x8_t*)(ldsA + lds_swz(arow[i], ks * 2 + h));
                        al_[i] = *(const bf16x8_t*)(ldsA + lds_swz(arow[i], 4 + ks * 2 + h));
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bh_[j] = *(const bf16x8_t*)(ldsW + lds_swz(wrow[j], ks * 2 + h));
                        bl_[j] = *(const bf16x8_t*)(ldsW + lds_swz(wrow[j], 4 + ks * 2 + h));
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_[i], bh_[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bl_[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bh_[j], acc[i][j], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float4 fa_[MT][2], fb_[NT][2];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        fa_[i][0] = *(const float4*)(ldsA + lds_swz(arow[i], ks * 4 + 2 * h));
                        fa_[i][1] = *(const float4*)(ldsA + lds_swz(arow[i], ks * 4 + 2 * h + 1));
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        fb_[j][0] = *(const float4*)(ldsW + lds_swz(wrow[j], ks * 4 + 2 * h));
                        fb_[j][1] = *(const float4*)(ldsW + lds_swz(wrow[j], ks * 4 + 2 * h + 1));
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].x, fb_[j][0].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].y, fb_[j][0].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].z, fb_[j][0].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].w, fb_[j][0].w, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].x, fb_[j][1].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].y, fb_[j][1].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].z, fb_[j][1].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].w, fb_[j][1].w, acc[i][j], 0, 0, 0);
                        }
                }
            }
        }
    };

    // ---- main loop: loads of iteration it+1 are in flight while iteration it computes --------
    load_regs(0);
    for (int it = 0; it < nit; ++it) {
        if (it > 0) __syncthreads();
        store_lds(it);
        __syncthreads();
        if (it + 1 < nit) load_regs(it + 1);
        compute(it);
    }

    // ---- epilogue phase 1: accumulators (+bias) -> LDS fp32 image [TM][TN] -----------------------
    __syncthreads();
    float* tile = (float*)smem;
    float bias_c[NT];
    {
        int bi[NT]; bool okb[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { const int n = n0 + (wn * NT + j) * 32 + r; okb[j] = n < a.n; bi[j] = n % a.bias_mod; }
        gemm_bias_load<NT>(a, bi, okb, bias_c);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wn * NT + j) * 32 + r;
        const float bias = bias_c[j];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                tile[row * TN + col] = acc[i][j][e] + bias;
            }
    }
    __syncthreads();

    // ---- phase 2: 16-byte stores along the channel axis (+ residual / GELU / phase scatter) ------
    T* out = (T*)a.out;
    const T* res = (const T*)a.res;
    const bool do_stats = a.stats != nullptr;
    const long long rows_total = (long long)a.B * a.mrows;
    constexpr int CPR = TN / EPC;   // 16-byte chunks per tile row
    constexpr int P2 = (TM * CPR) / NTHR;   // chunks per thread (exact: TM*CPR is a multiple of NTHR)
    static_assert((TM * CPR) % NTHR == 0, "tile chunks must divide evenly over the block");
    {
        unsigned off[P2];
        bool okv[P2];
        u32x4_t rres[P2];
#pragma unroll
        for (int k = 0; k < P2; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / CPR, cc = idx - row * CPR;
            const int n = n0 + cc * EPC;
            int bb, m;
            bool ok;
            if (a.flat) {
                const long long R = R0 + row;
                ok = R < rows_total;
                bb = (int)(R / a.mrows);
                m = (int)(R - (long long)bb * a.mrows);
            } else {
                bb = b0; m = m0 + row;
                ok = m < a.mrows;
            }
            ok = ok && n < a.n;
            if (a.scatter_f) {
                const int phase = n / a.out_c, co = n - phase * a.out_c;
                const int orow = m * a.scatter_f + phase - a.scatter_pad;
                ok = ok && orow >= 0 && orow < a.out_rows;
                off[k] = (unsigned)((bb * a.out_rows + orow) * a.out_c + co);
            } else {
                off[k] = (unsigned)((bb * a.out_rows + m) * a.out_c + n);
            }
            okv[k] = ok;
            if (!ok) off[k] = 0;
        }
        if (res) {   // wave-uniform
#pragma unroll
            for (int k = 0; k < P2; ++k) rres[k] = *(const u32x4_t*)(res + off[k]);
        } else {
#pragma unroll
            for (int k = 0; k < P2; ++k) rres[k] = u32x4_t{0u, 0u, 0u, 0u};
        }
        // statistics: this thread's chunks all cover the same EPC channels (cc = tid % CPR) and rows
        // tid/CPR + k*RPK, so per-channel sums accumulate in registers and are flushed once per sample
        // segment: reduce over the channels of a group inside the thread, over the threads of a group and over
        // the row-lanes of the wave with shuffles, then one fp64 atomic pair per (wave, group).
        constexpr int RPK = NTHR / CPR;                 // tile rows covered by one k step
        const bool stats_here = do_stats;
        const int gs = stats_here ? a.out_c / a.stats_groups : EPC;   // channels per group (>= EPC, power of two)
        const int tpg = gs / EPC;                       // threads (adjacent cc) per group
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < P2; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / CPR, cc = idx - row * CPR;
            float v[EPC], rr[EPC];
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                const float4 q = *(const float4*)(tile + row * TN + cc * EPC + e);
                v[e] = q.x; v[e + 1] = q.y; v[e + 2] = q.z; v[e + 3] = q.w;
            }
            unpack16<T>(rres[k], rr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += rr[e];
            if (a.gelu) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = gelu_erf_f(v[e]);
            }
            if (okv[k]) {
                *(u32x4_t*)(out + off[k]) = pack16_stored<T>(v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
            }
            if (stats_here && ((((k + 1) * RPK) % seg) == 0 || k == P2 - 1)) {   // wave-uniform
                for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                for (int o = CPR; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                const int n = n0 + cc * EPC;
                if (lane < CPR && (cc & (tpg - 1)) == 0 && n < a.n) {
                    const int trow = k * RPK;                       // any row of the segment just finished
                    const int bb = a.flat ? (int)((R0 + trow) / a.mrows) : b0;
                    if (bb < a.B) {
                        double* sp = a.stats + ((size_t)bb * a.stats_groups + (a.scatter_f ? n % a.out_c : n) / gs) * 2;   // scatter: n = phase * out_c + channel
                        atomicAdd(sp, (double)s1);
                        atomicAdd(sp + 1, (double)s2);
                    }
                }
                s1 = 0.f; s2 = 0.f;
            }
        }
    }
}

// =====================================================================================================
// Weight-stationary, warp-specialised persistent variant for the large layers (per-sample tiling, TM = 128).
//   * The whole weight operand of the block's N tile (every K chunk x tap of both segments) is copied into LDS
//     ONCE; the block then walks its M tiles, so the steady state streams only activations from HBM.  In the
//     plain kernel every 128-row tile re-reads the weights from L2, 3-4x the HBM bytes of the layer.
//   * block = 8 waves: waves 0-3 "consumers" (MFMA + epilogue), waves 4-7 "producers" (activation loads two K
//     steps ahead in registers, fused GroupNorm/FiLM/SiLU prologue, LDS stores).  Each SIMD hosts one wave of
//     each role, so the prologue's VALU/transcendental work overlaps the matrix pipe.
//   * LDS = weights (<= ~110 KB) + 2 activation stages (2 x 18.7 KB) + 4 x 2 KB wave-private epilogue scratch.
//   * step g: producers fill stage g&1 with K iteration g, consumers compute iteration g-1 from stage (g-1)&1;
//     ONE workgroup barrier per step.  The epilogue is wave-local (each consumer wave transposes its own 64-row
//     slice through its private scratch, 8 rows at a time), so it needs no workgroup barrier at all.
// =====================================================================================================
constexpr int kWsARows = 136;                       // (128-1)*1 + 9 taps max for stride 1
constexpr int kWsScratch = 4 * 2048;                // 4 consumer waves x (8 rows x 64 cols x fp32)

template <typename T, int NT, int WN>   // consumer wave tile = 64 x (NT*32); consumer waves = (4/WN) x WN; TN = NT*WN*32
__global__ void __launch_bounds__(512) conv_gemm_ws_kernel(const GemmArgs a, int tiles_m_total, int blocks_per_n) {
    constexpr int WM = 4 / WN;
    constexpr int MT = 4 / WM;                              // TM = 128 = WM * MT * 32
    constexpr int TM = 128, TN = NT * WN * 32, NTHR = 256;  // NTHR = threads of ONE role
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int KC = kRowBytes / (int)sizeof(T);
    constexpr int A_CH = (kWsARows * 8 + NTHR - 1) / NTHR;
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int ASTAGE = kWsARows * kLdsPitch;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NTHR;   // scalar role branch
    const int tid = threadIdx.x & (NTHR - 1), lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    const int tiles_m = (a.mrows + TM - 1) / TM;
    const int nit0 = a.seg[0].nchunk;                       // one K iteration = one chunk with ALL its taps
    const int nit = nit0 + (a.nseg > 1 ? a.seg[1].nchunk : 0);
    const int tn_i = (int)blockIdx.x % tiles_n;
    const int slot = (int)blockIdx.x / tiles_n;
    const int n0 = tn_i * TN;
    const int my_tiles = slot < tiles_m_total ? (tiles_m_total - 1 - slot) / blocks_per_n + 1 : 0;
    const int G = my_tiles * nit;
    // LDS carve-up: [weights][A stage 0][A stage 1][scratch]
    const int wrows0 = a.seg[0].nchunk * a.seg[0].taps * TN;
    const int wrows = wrows0 + (a.nseg > 1 ? a.seg[1].nchunk * a.seg[1].taps * TN : 0);
    char* ldsWall = smem;
    char* ldsA0 = smem + (size_t)wrows * kLdsPitch;
    char* scratch = ldsA0 + 2 * ASTAGE;
    auto tile_geom = [&](int tseq, int& b0, int& m0) {
        const int t = slot + tseq * blocks_per_n;
        b0 = t / tiles_m;
        m0 = (t - b0 * tiles_m) * TM;
    };

    // ---- one-time weight fill: rows [seg][chunk][tap][n_l] -> LDS (all 512 threads, 8 loads in flight each) ----
    {
        const int t512 = (int)threadIdx.x;
        const int c16w = t512 & 7;
        for (int base = t512 >> 3; base < wrows; base += 64 * 8) {
            u32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int row = base + 64 * k;
                row = row < wrows ? row : wrows - 1;              // clamped: the store below is guarded
                const bool s1 = row >= wrows0;
                const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
                const int rl = s1 ? row - wrows0 : row;
                const int ct = rl / TN, n_l = rl - ct * TN;       // ct = chunk * taps + tap
                // rows beyond n_pad re-read row n_pad-1 (their columns are masked in the epilogue)
                const int nn = (n0 + n_l) < a.n_pad ? (n0 + n_l) : a.n_pad - 1;
                v[k] = *(const u32x4_t*)((const char*)sg.w + ((size_t)ct * a.n_pad + nn) * kRowBytes + c16w * 16);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = base + 64 * k;
                if (row < wrows) *(u32x4_t*)(ldsWall + lds_swz(row, c16w)) = v[k];
            }
        }
    }
    __syncthreads();
    // both roles execute the same, even number of steps: 2 * ceil((G + 1) / 2)
    const int npairs = (G + 2) / 2;

    if (producer) {
        // ================================ producers ==================================================
        const int c16 = tid & 7;
        const int lds_row0 = tid >> 3;
        struct Regs { u32x4_t ra[A_CH]; f32x4_t abq[EPC / 2]; float raw_scale; unsigned avalid; };
        Regs R0s, R1s;
        int arow_idx[A_CH];
        int abq_b0 = 0;
        auto load_a = [&](int q, Regs& R) {
            const int tseq = q / nit, it = q - tseq * nit;
            const bool s1 = it >= nit0;
            const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
            const int chunk = s1 ? it - nit0 : it;
            if (chunk == 0) {
                int b0, m0;
                tile_geom(tseq, b0, m0);
                const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
                const int nrows = (TM - 1) * sg.stride + sg.taps;
                const int p_lo = m0 * sg.stride + off_min;
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    const int row = (tid + i * NTHR) >> 3;
                    const int p = p_lo + row;
                    const bool ok = row < nrows && p >= 0 && p < a.lin;
                    arow_idx[i] = ok ? b0 * a.lin + p : -1;
                }
                abq_b0 = b0;
            }
            const int ctot = sg.c0 + sg.c1;
            const int cidx = chunk * KC + c16 * EPC;
            const bool cvalid = cidx < ctot;
            const bool from1 = sg.c1 > 0 && cidx >= sg.c0;
            const char* src0u = uniform_ptr(sg.src0);
            const char* src1u = uniform_ptr(sg.src1);
            const char* src = from1 ? src1u : src0u;
            const int c0u = __builtin_amdgcn_readfirstlane(sg.c0), c1u = __builtin_amdgcn_readfirstlane(sg.c1);
            const unsigned rowbytes = (unsigned)(from1 ? c1u : c0u) * (unsigned)sizeof(T);
            const unsigned colbytes = (unsigned)(from1 ? cidx - c0u : cidx) * (unsigned)sizeof(T);
            R.avalid = 0;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const bool ok = cvalid && arow_idx[i] >= 0;
                const unsigned off = (ok ? (unsigned)arow_idx[i] : 0u) * rowbytes + (ok ? colbytes : 0u);
                const char* ptr = src + off;
                ADF_GLOAD16(R.ra[i], ptr);
                R.avalid |= (ok ? 1u : 0u) << i;
            }
            const float* abp = sg.ab ? sg.ab + (cvalid ? (unsigned)(abq_b0 * ctot + cidx) * 2u : 0u) : (const float*)sg.w;
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) { const float* pe = abp + e * 4; ADF_GLOAD16(R.abq[e], pe); }
            R.raw_scale = from1 ? sg.scale1 : 1.0f;
        };
        // wait until only the OTHER register set's loads (issued later) are still in flight
        auto wait_set = [&](Regs& R) {
            static_assert(A_CH == 5, "wait_set is written for 5 activation chunks per producer thread");
            if constexpr (EPC == 8)
                asm volatile("s_waitcnt vmcnt(9)" : "+v"(R.ra[0]), "+v"(R.ra[1]), "+v"(R.ra[2]), "+v"(R.ra[3]), "+v"(R.ra[4]),
                             "+v"(R.abq[0]), "+v"(R.abq[1]), "+v"(R.abq[2]), "+v"(R.abq[3]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(R.ra[0]), "+v"(R.ra[1]), "+v"(R.ra[2]), "+v"(R.ra[3]), "+v"(R.ra[4]),
                             "+v"(R.abq[0]), "+v"(R.abq[1]) : : "memory");
        };
        auto store_a = [&](int q, Regs& R, char* ldsA) {
            const int it = q % nit;
            const bool s1 = it >= nit0;
            const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
            const int nrows = (TM - 1) * sg.stride + sg.taps;
            const bool act = sg.act != 0;
            const bool use_ab = sg.ab != nullptr;
            f32x2_t fa2[EPC / 2], fb2[EPC / 2], za2[EPC / 2], zb2[EPC / 2];
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) {
                fa2[e] = use_ab ? f32x2_t{R.abq[e].x, R.abq[e].z} : f32x2_t{R.raw_scale, R.raw_scale};
                fb2[e] = use_ab ? f32x2_t{R.abq[e].y, R.abq[e].w} : f32x2_t{0.f, 0.f};
                za2[e] = fa2[e] * -1.4426950408889634f;
                zb2[e] = fb2[e] * -1.4426950408889634f;
            }
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int row = lds_row0 + i * (NTHR / 8);
                if (row < nrows) {
                    u32x4_t qv = u32x4_t{0u, 0u, 0u, 0u};
                    if ((R.avalid >> i) & 1u) {
                        float f[EPC];
                        unpack16<T>(R.ra[i], f);
                        if (act) {
#pragma unroll
                            for (int e = 0; e < EPC / 2; ++e) {
                                const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                                const f32x2_t v2 = x2 * fa2[e] + fb2[e];
                                const f32x2_t z2 = x2 * za2[e] + zb2[e];
                                f32x2_t d2 = {__builtin_amdgcn_exp2f(z2.x), __builtin_amdgcn_exp2f(z2.y)};
                                d2 = d2 + 1.0f;
                                const f32x2_t r2 = {__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
                                const f32x2_t y2 = v2 * r2;
                                f[2 * e] = y2.x; f[2 * e + 1] = y2.y;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < EPC / 2; ++e) {
                                const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                                const f32x2_t v2 = x2 * fa2[e] + fb2[e];
                                f[2 * e] = v2.x; f[2 * e + 1] = v2.y;
                            }
                        }
                        qv = pack16<T>(f);
                    }
                    *(u32x4_t*)(ldsA + lds_swz(row, c16)) = qv;
                }
            }
        };
        // Straight-line even/odd step pairs with unconditional loads (iteration index clamped to G-1): the
        // compiler then knows exactly which loads are outstanding and keeps two K steps in flight (a runtime
        // even/odd branch or conditional loads make it drain vmcnt(0) every step).
        if (G > 0) {
            const int qmax = G - 1;
            load_a(0, R0s);
            load_a(qmax < 1 ? qmax : 1, R1s);
            for (int pr = 0; pr < npairs; ++pr) {
                const int g = 2 * pr;
                wait_set(R0s);
                store_a(g < qmax ? g : qmax, R0s, ldsA0);            // a step >= G writes a stage nobody reads again
                load_a(g + 2 < qmax ? g + 2 : qmax, R0s);
                __syncthreads();
                wait_set(R1s);
                store_a(g + 1 < qmax ? g + 1 : qmax, R1s, ldsA0 + ASTAGE);
                load_a(g + 3 < qmax ? g + 3 : qmax, R1s);
                __syncthreads();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of ours may be in flight when the wave ends
        } else {
            for (int pr = 0; pr < npairs; ++pr) { __syncthreads(); __syncthreads(); }
        }
        return;
    }

    // ==================================== consumers ====================================================
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    float bias_r[NT];        // bias of this lane's output column(s): the block's N tile never changes
    {
        int bi[NT]; bool okb[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { const int nb = n0 + (wn * NT + j) * 32 + r; okb[j] = nb < a.n; bi[j] = nb % a.bias_mod; }
        gemm_bias_load<NT>(a, bi, okb, bias_r);
    }
    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = bias_r[j];

    for (int g = 0; g < 2 * npairs; ++g) {
        if (g >= 1 && g <= G) {
            const int q = g - 1;
            const int tseq = q / nit, it = q - tseq * nit;
            const char* ldsA = ldsA0 + ((q & 1) ? ASTAGE : 0);
            // geometry of the wave-local epilogue (see below) -- needed here to prefetch the identity residual
            constexpr int WCOLS = NT * 32;                       // columns owned by the wave
            constexpr int CPW = WCOLS / EPC;                     // 16-byte chunks per row
            constexpr int RPP = 64 / CPW;                        // rows handled by the 64 lanes at once
            constexpr int NSUB = RPP >= 8 ? 1 : 8 / RPP;         // sub-steps per 8-row pass
            constexpr int NRES = MT * 4 * NSUB;
            u32x4_t rres[NRES];
            const int cc = lane % CPW, rsub = lane / CPW;        // this lane's chunk column / row inside a sub-step
            const int ncol0 = n0 + wn * WCOLS;
            if (it == nit - 1 && a.res != nullptr) {             // wave-uniform
                int b0, m0;
                tile_geom(tseq, b0, m0);
                const T* res = (const T*)a.res;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4)
#pragma unroll
                        for (int sb = 0; sb < NSUB; ++sb) {
                            const int prow = sb * RPP + rsub;
                            const int m = m0 + (wm * MT + i) * 32 + 8 * p4 + prow;
                            const int n = ncol0 + cc * EPC;
                            const bool ok = prow < 8 && m < a.mrows && n < a.n;
                            const unsigned off = ok ? (unsigned)((b0 * a.out_rows + m) * a.out_c + n) : 0u;
                            rres[(i * 4 + p4) * NSUB + sb] = *(const u32x4_t*)(res + off);
                        }
            }
            {
                const bool s1 = it >= nit0;
                const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
                const int chunk = s1 ? it - nit0 : it;
                const char* ldsW = ldsWall + (size_t)((s1 ? wrows0 : 0) + chunk * sg.taps * TN) * kLdsPitch;
                const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
                int abase[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) abase[i] = ((wm * MT + i) * 32 + r) * sg.stride;
                bool pipelined = false;
                if constexpr (kBf16) {
                    const int aoff0 = sg.off0 - off_min, astep = sg.step;
                    auto addrA = [&](int tap, int i, int c16) { return ldsA + lds_swz(abase[i] + aoff0 + tap * astep, c16); };
                    auto addrW = [&](int tap, int j, int c16) { return ldsW + lds_swz(tap * TN + (wn * NT + j) * 32 + r, c16); };
                    if (sg.taps == 3) { mfma_chunk_bf16<3, MT, NT>(acc, h, addrA, addrW); pipelined = true; }
                    else if (sg.taps == 1) { mfma_chunk_bf16<1, MT, NT>(acc, h, addrA, addrW); pipelined = true; }
                }
                for (int tap = 0; tap < (pipelined ? 0 : sg.taps); ++tap) {
                    const int aoff = sg.off0 + tap * sg.step - off_min;
                    int arow[MT], wrow[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) arow[i] = abase[i] + aoff;
#pragma unroll
                    for (int j = 0; j < NT; ++j) wrow[j] = tap * TN + (wn * NT + j) * 32 + r;
                    if constexpr (kBf16) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            bf16x8_t fa_[MT], fb_[NT];
#pragma unroll
                            for (int i = 0; i < MT; ++i) fa_[i] = *(const bf16x8_t*)(ldsA + lds_swz(arow[i], ks * 2 + h));
#pragma unroll
                            for (int j = 0; j < NT; ++j) fb_[j] = *(const bf16x8_t*)(ldsW + lds_swz(wrow[j], ks * 2 + h));
#pragma unroll
                            for (int i = 0; i < MT; ++i)
#pragma unroll
                                for (int j = 0; j < NT; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            float4 fa_[MT][2], fb_[NT][2];
#pragma unroll
                            for (int i = 0; i < MT; ++i) {
                                fa_[i][0] = *(const float4*)(ldsA + lds_swz(arow[i], ks * 4 + 2 * h));
                                fa_[i][1] = *(const float4*)(ldsA + lds_swz(arow[i], ks * 4 + 2 * h + 1));
                            }
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                fb_[j][0] = *(const float4*)(ldsW + lds_swz(wrow[j], ks * 4 + 2 * h));
                                fb_[j][1] = *(const float4*)(ldsW + lds_swz(wrow[j], ks * 4 + 2 * h + 1));
                            }
#pragma unroll
                            for (int i = 0; i < MT; ++i)
#pragma unroll
                                for (int j = 0; j < NT; ++j) {
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].x, fb_[j][0].x, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].y, fb_[j][0].y, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].z, fb_[j][0].z, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][0].w, fb_[j][0].w, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].x, fb_[j][1].x, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].y, fb_[j][1].y, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].z, fb_[j][1].z, acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i][1].w, fb_[j][1].w, acc[i][j], 0, 0, 0);
                                }
                        }
                    }
                }
            }
            if (it == nit - 1) {
                // ---------------- wave-local epilogue of this tile (no workgroup barrier) ----------------
                // The wave owns rows [wm*MT*32, +MT*32) x cols [wn*NT*32, +NT*32) of the tile.  Pass (i, p4):
                // accumulator registers 4*p4..4*p4+3 of both lane halves are rows 8*p4 .. 8*p4+7 of m-tile i;
                // they go through the wave's private scratch and leave as 16-byte channel chunks.  The bias is
                // already in the accumulators (they are re-initialised with it), offsets advance incrementally.
                int b0, m0;
                tile_geom(tseq, b0, m0);
                float* sc = (float*)(scratch + wave * 2048);         // [8][WCOLS<=64] fp32
                float* scw = sc + (4 * h) * WCOLS + r;               // this lane's write base
                const float* scr = sc + (rsub < 8 ? rsub : 7) * WCOLS + cc * EPC;   // and read base (sub-step 0)
                T* out = (T*)a.out;
                const bool has_res = a.res != nullptr;
                const bool stats_here = a.stats != nullptr;
                const int gs = stats_here ? a.out_c / a.stats_groups : EPC;
                const int tpg = gs / EPC;
                const int n = ncol0 + cc * EPC;
                const int mw0 = m0 + wm * MT * 32;                   // first row owned by the wave
                const bool full = !a.scatter_f && (mw0 + MT * 32 <= a.mrows) && (ncol0 + WCOLS <= a.n) && (RPP <= 8 || rsub < 8);
                f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) {
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const int e = 4 * p4 + e4;
                                scw[e4 * WCOLS + j * 32] = acc[i][j][e];
                                acc[i][j][e] = bias_r[j];
                            }
                        // same-wave LDS traffic is processed in order; only the compiler must not reorder it
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int sb = 0; sb < NSUB; ++sb) {
                            const int prow = sb * RPP + rsub;        // row inside the 8-row pass
                            const int m = mw0 + i * 32 + 8 * p4 + prow;
                            bool ok = true;
                            unsigned off;
                            if (full) {
                                off = (unsigned)((b0 * a.out_rows + m) * a.out_c + n);
                            } else {
                                ok = prow < 8 && m < a.mrows && n < a.n;
                                if (a.scatter_f) {
                                    const int phase = n / a.out_c, co = n - phase * a.out_c;
                                    const int orow = m * a.scatter_f + phase - a.scatter_pad;
                                    ok = ok && orow >= 0 && orow < a.out_rows;
                                    off = (unsigned)((b0 * a.out_rows + orow) * a.out_c + co);
                                } else {
                                    off = (unsigned)((b0 * a.out_rows + m) * a.out_c + n);
                                }
                            }
                            float v[EPC];
#pragma unroll
                            for (int e = 0; e < EPC; e += 4) {
                                const float4 qv = *(const float4*)(scr + sb * RPP * WCOLS + e);
                                v[e] = qv.x; v[e + 1] = qv.y; v[e + 2] = qv.z; v[e + 3] = qv.w;
                            }
                            if (has_res) {
                                float rr[EPC];
                                unpack16<T>(rres[(i * 4 + p4) * NSUB + sb], rr);
#pragma unroll
                                for (int e = 0; e < EPC; ++e) v[e] += rr[e];
                            }
                            if (a.gelu) {
#pragma unroll
                                for (int e = 0; e < EPC; ++e) v[e] = gelu_erf_f(v[e]);
                            }
                            if (ok) {
                                *(u32x4_t*)(out + off) = pack16_stored<T>(v);
#pragma unroll
                                for (int e = 0; e < EPC; e += 2) {
                                    const f32x2_t v2 = {v[e], v[e + 1]};
                                    s1v += v2;
                                    s2v += v2 * v2;
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (stats_here) {
                    float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
                    for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    for (int o = CPW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    if (lane < CPW && (cc & (tpg - 1)) == 0 && n < a.n) {
                        double* sp = a.stats + ((size_t)b0 * a.stats_groups + (a.scatter_f ? n % a.out_c : n) / gs) * 2;
                        atomicAdd(sp, (double)s1);
                        atomicAdd(sp + 1, (double)s2);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// =====================================================================================================
// Intra-block split-K variant for the short levels (L <= 64: few rows, long K, latency-bound).
//   One block = one 32 x 32 output tile; its 4 waves each walk a quarter of the K iterations with a PRIVATE
//   staging region and only wave-level synchronisation (a wave's LDS traffic is processed in order), then
//   the four partial accumulators are summed through LDS.  Cuts the serial K-loop length by 4.
//   Restricted to stride-1 segments with <= 3 taps (3-tap convs and 1x1 ops); flat or per-sample tiles.
// =====================================================================================================
constexpr int ks_a_rows(int mt) { return 32 * mt + 8; }                 // tile rows + taps, rounded to the 8-row staging step
constexpr int ks_wave_lds(int mt, int nt) { return (ks_a_rows(mt) + kTapGroup * 32 * nt) * kLdsPitch; }   // 19.6 KB (32x32) .. 38 KB (64x64)

template <typename T, int MT, int NT>   // tile = (32 MT) x (32 NT); every wave accumulates the whole tile over its K quarter
__global__ void __launch_bounds__(256) conv_gemm_ksplit_kernel(const GemmArgs a) {
    constexpr int TM = 32 * MT, TN = 32 * NT, NW = 64;     // NW = threads of one staging group (a wave)
    constexpr int kKsARows = ks_a_rows(MT);
    constexpr int kKsWaveLds = ks_wave_lds(MT, NT);
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int KC = kRowBytes / (int)sizeof(T);
    constexpr int A_CH = (kKsARows * 8) / NW;              // 5 / 9
    constexpr int W_CH = (kTapGroup * TN * 8) / NW;        // 12 / 24
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr bool kX3 = IsX3<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    char* ldsA = smem + wave * kKsWaveLds;
    char* ldsW = ldsA + kKsARows * kLdsPitch;

    const int tiles_n = (a.n_pad + TN - 1) / TN;
    int bid = blockIdx.x;
    const int tn_i = bid % tiles_n; bid /= tiles_n;
    const int seg = a.flat ? a.seg_rows : TM;
    const int nsegs = TM / seg;
    int b0, m0;
    long long R0;
    if (a.flat) {
        R0 = (long long)bid * TM; b0 = (int)(R0 / a.mrows); m0 = 0;
    } else {
        const int tiles_m = (a.mrows + TM - 1) / TM;
        const int tm_i = bid % tiles_m;
        b0 = bid / tiles_m; m0 = tm_i * TM;
        R0 = (long long)b0 * a.mrows + m0;
    }
    const int n0 = tn_i * TN;
    const int c16 = lane & 7;
    const int lds_row0 = lane >> 3;

    const int nit0 = a.seg[0].nchunk;                      // taps <= kTapGroup: one iteration per chunk
    const int nit = nit0 + (a.nseg > 1 ? a.seg[1].nchunk : 0);
    const int it_begin = (nit * wave) / 4, it_end = (nit * (wave + 1)) / 4;

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    u32x4_t ra[A_CH], rw[W_CH];
    f32x4_t abq[EPC / 2];
    float raw_scale = 1.0f;
    unsigned avalid = 0;
    int arow_idx[A_CH];

    auto load_regs = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int chunk = s1 ? it - nit0 : it;
        if (chunk == 0 || it == it_begin) {
            const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
            const int segrows = (seg - 1) * sg.stride + sg.taps;
            const int nrows = nsegs * segrows;
            const int p_lo = m0 * sg.stride + off_min;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int row = lds_row0 + i * (NW / 8);
                const int j = a.flat ? row / segrows : 0;
                const int p = p_lo + (row - j * segrows);
                const int bb = b0 + j;
                const bool ok = row < nrows && p >= 0 && p < a.lin && bb < a.B;
                arow_idx[i] = ok ? bb * a.lin + p : -1;
            }
        }
        const int ctot = sg.c0 + sg.c1;
        const int cidx = chunk * KC + c16 * EPC;
        const bool cvalid = cidx < ctot;
        const int c0u = __builtin_amdgcn_readfirstlane(sg.c0), c1u = __builtin_amdgcn_readfirstlane(sg.c1);
        const bool from1 = c1u > 0 && cidx >= c0u;
        const char* src0u = uniform_ptr(sg.src0);
        const char* src1u = uniform_ptr(sg.src1);
        const char* src = from1 ? src1u : src0u;
        const unsigned rowbytes = (unsigned)(from1 ? c1u : c0u) * (unsigned)sizeof(T);
        const unsigned colbytes = (unsigned)(from1 ? cidx - c0u : cidx) * (unsigned)sizeof(T);
        avalid = 0;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const bool ok = cvalid && arow_idx[i] >= 0;
            const unsigned off = (ok ? (unsigned)arow_idx[i] : 0u) * rowbytes + (ok ? colbytes : 0u);
            ra[i] = *(const u32x4_t*)(src + off);
            avalid |= (ok ? 1u : 0u) << i;
        }
        const float* abp = sg.ab ? sg.ab + (cvalid ? (unsigned)(b0 * ctot + cidx) * 2u : 0u) : (const float*)sg.w;
#pragma unroll
        for (int e = 0; e < EPC / 2; ++e) abq[e] = *(const f32x4_t*)(abp + e * 4);
        raw_scale = from1 ? sg.scale1 : 1.0f;
        const char* wp = uniform_ptr(sg.w) + (size_t)(chunk * sg.taps) * a.n_pad * kRowBytes;
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int row = lds_row0 + i * (NW / 8);
            const int tap_l = row / TN, n_l = row - tap_l * TN;
            const unsigned woff = (unsigned)((tap_l * a.n_pad + n0 + n_l) * kRowBytes + c16 * 16);
            rw[i] = *(const u32x4_t*)(wp + woff);
        }
    };
    auto store_lds = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int nrows = nsegs * ((seg - 1) * sg.stride + sg.taps);
        const bool act = sg.act != 0;
        const bool use_ab = sg.ab != nullptr;
        f32x2_t fa2[EPC / 2], fb2[EPC / 2], za2[EPC / 2], zb2[EPC / 2];
#pragma unroll
        for (int e = 0; e < EPC / 2; ++e) {
            fa2[e] = use_ab ? f32x2_t{abq[e].x, abq[e].z} : f32x2_t{raw_scale, raw_scale};
            fb2[e] = use_ab ? f32x2_t{abq[e].y, abq[e].w} : f32x2_t{0.f, 0.f};
            za2[e] = fa2[e] * -1.4426950408889634f;
            zb2[e] = fb2[e] * -1.4426950408889634f;
        }
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = lds_row0 + i * (NW / 8);
            if (row < nrows) {
                u32x4_t qv = u32x4_t{0u, 0u, 0u, 0u};
                if ((avalid >> i) & 1u) {
                    float f[EPC];
                    unpack16<T>(ra[i], f);
#pragma unroll
                    for (int e = 0; e < EPC / 2; ++e) {
                        const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                        f32x2_t y2 = x2 * fa2[e] + fb2[e];
                        if (act) {
                            const f32x2_t z2 = x2 * za2[e] + zb2[e];
                            f32x2_t d2 = {__builtin_amdgcn_exp2f(z2.x), __builtin_amdgcn_exp2f(z2.y)};
                            d2 = d2 + 1.0f;
                            const f32x2_t r2 = {__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
                            y2 = y2 * r2;
                        }
                        f[2 * e] = y2.x; f[2 * e + 1] = y2.y;
                    }
                    qv = pack16<T>(f);
                }
                if constexpr (kX3) {
                    float f4[4];
                    unpack16<float>(qv, f4);
                    u32x2_t hi, lo;
                    split_bf16x4(f4, hi, lo);
                    *(u32x2_t*)(ldsA + lds_swz(row, c16 >> 1) + (c16 & 1) * 8) = hi;
                    *(u32x2_t*)(ldsA + lds_swz(row, 4 + (c16 >> 1)) + (c16 & 1) * 8) = lo;
                } else {
                    *(u32x4_t*)(ldsA + lds_swz(row, c16)) = qv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int row = lds_row0 + i * (NW / 8);
            *(u32x4_t*)(ldsW + lds_swz(row, c16)) = rw[i];
        }
    };
    auto compute = [&](int it) {
        const bool s1 = it >= nit0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int off_min = sg.step > 0 ? sg.off0 : sg.off0 - (sg.taps - 1);
        const int segrows = (seg - 1) * sg.stride + sg.taps;
        int abase[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ti = i * 32 + r;
            const int j = a.flat ? ti / seg : 0;
            abase[i] = j * segrows + (ti - j * seg) * sg.stride;
        }
        for (int tap = 0; tap < sg.taps; ++tap) {
            const int aoff = sg.off0 + tap * sg.step - off_min;
            if constexpr (kBf16) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8_t fa_[MT], fb_[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa_[i] = *(const bf16x8_t*)(ldsA + lds_swz(abase[i] + aoff, ks * 2 + h));
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb_[j] = *(const bf16x8_t*)(ldsW + lds_swz(tap * TN + j * 32 + r, ks * 2 + h));
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
                }
            } else if constexpr (kX3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t ah_[MT], al_[MT], bh_[NT], bl_[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        ah_[i] = *(const bf16x8_t*)(ldsA + lds_swz(abase[i] + aoff, ks * 2 + h));
                        al_[i] = *(const bf16x8_t*)(ldsA + lds_swz(abase[i] + aoff, 4 + ks * 2 + h));
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bh_[j] = *(const bf16x8_t*)(ldsW + lds_swz(tap * TN + j * 32 + r, ks * 2 + h));
                        bl_[j] = *(const bf16x8_t*)(ldsW + lds_swz(tap * TN + j * 32 + r, 4 + ks * 2 + h));
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_[i], bh_[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bl_[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bh_[j], acc[i][j], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float4 a0[MT], a1[MT], w0[NT], w1[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        a0[i] = *(const float4*)(ldsA + lds_swz(abase[i] + aoff, ks * 4 + 2 * h));
                        a1[i] = *(const float4*)(ldsA + lds_swz(abase[i] + aoff, ks * 4 + 2 * h + 1));
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        w0[j] = *(const float4*)(ldsW + lds_swz(tap * TN + j * 32 + r, ks * 4 + 2 * h));
                        w1[j] = *(const float4*)(ldsW + lds_swz(tap * TN + j * 32 + r, ks * 4 + 2 * h + 1));
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].x, w0[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].y, w0[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].z, w0[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].w, w0[j].w, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i].x, w1[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i].y, w1[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i].z, w1[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i].w, w1[j].w, acc[i][j], 0, 0, 0);
                        }
                }
            }
        }
    };
    // wave-private pipeline: a wave's own LDS writes are visible to its later reads (in-order LDS queue); the
    // fences only stop the compiler from reordering across the hand-off points
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (it_begin < it_end) {
        load_regs(it_begin);
        for (int it = it_begin; it < it_end; ++it) {
            wave_sync();
            store_lds(it);
            wave_sync();
            if (it + 1 < it_end) load_regs(it + 1);
            compute(it);
        }
    }

    // ---- cross-wave reduction + epilogue ------------------------------------------------------------------
    __syncthreads();
    float* part = (float*)smem;                            // [4][TM][TN] fp32 partial tiles (16 KB .. 64 KB, over the staging area)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                part[(wave * TM + row) * TN + j * 32 + r] = acc[i][j][e];
            }
    __syncthreads();
    constexpr int CPR = TN / EPC;                          // chunks per tile row
    constexpr int NCH = TM * CPR;                          // 16-byte chunks in the tile
    constexpr int NPASS = (NCH + 255) / 256;               // 1 (up to 256 chunks) .. 4 (64x64 fp32)
    T* out = (T*)a.out;
    const T* res = (const T*)a.res;
    const long long rows_total = (long long)a.B * a.mrows;
    const bool want_stats = a.stats != nullptr;
    const int gs = want_stats ? a.out_c / a.stats_groups : EPC;
    const int tpg = gs / EPC;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int idx = tid + ps * 256;
        const bool active = idx < NCH;
        const int row = idx / CPR, cc = idx - (idx / CPR) * CPR;
        const int n = n0 + cc * EPC;
        int bb, m;
        bool ok;
        if (a.flat) {
            const long long R = R0 + row;
            ok = R < rows_total;
            bb = (int)(R / a.mrows);
            m = (int)(R - (long long)bb * a.mrows);
        } else {
            bb = b0; m = m0 + row;
            ok = m < a.mrows;
        }
        ok = ok && active && n < a.n;
        float s1 = 0.f, s2 = 0.f;
        if (active) {
            float v[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const float4 q = *(const float4*)(part + (w * TM + row) * TN + cc * EPC + e);
                    v[e] += q.x; v[e + 1] += q.y; v[e + 2] += q.z; v[e + 3] += q.w;
                }
            if (ok) {
                int bi[EPC]; bool okb[EPC]; float bsum[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) { bi[e] = (n + e) % a.bias_mod; okb[e] = true; }
                gemm_bias_load<EPC>(a, bi, okb, bsum);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += bsum[e];
                const unsigned off = (unsigned)((bb * a.out_rows + m) * a.out_c + n);
                if (res) {
                    float rr[EPC];
                    unpack16<T>(*(const u32x4_t*)(res + off), rr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += rr[e];
                }
                if (a.gelu) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = gelu_erf_f(v[e]);
                }
                *(u32x4_t*)(out + off) = pack16_stored<T>(v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
            }
        }
        if (want_stats && (tid & ~63) + ps * 256 < NCH) {      // wave-uniform: NCH is a multiple of 64
            // a wave covers 64 / CPR consecutive rows, all inside one sample (seg is a multiple of that)
            for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            for (int o = CPR; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (lane < CPR && (cc & (tpg - 1)) == 0 && n < a.n) {
                const int frow = ((tid >> 6) * 64 + ps * 256) / CPR;     // first tile row of this wave in this pass
                const int sb = a.flat ? (int)((R0 + frow) / a.mrows) : b0;
                if (sb < a.B) {
                    double* sp = a.stats + ((size_t)sb * a.stats_groups + n / gs) * 2;
                    atomicAdd(sp, (double)s1);
                    atomicAdd(sp + 1, (double)s2);
                }
            }
        }
    }
}

// host-side launcher (adf_gemm.hip); *stats_fused tells whether the requested statistics were produced
// dtype: 0 = fp32 (exact-fp32 MFMA), 1 = bf16 storage, 2 = fp32 storage with split-bf16 operands (f32x3_t, adf_common.h)
const char* launch_conv_gemm(const GemmArgs& a, int dtype, hipStream_t stream, bool* stats_fused = nullptr);
// would launch_conv_gemm route this phase_c > 0 conv to conv_gemm_rb_kernel (shape, tile count)?  No launch.
bool conv_gemm_phase_eligible(const GemmArgs& a, int dtype);   // dtype as launch_conv_gemm: 1 -> adf_gemm_rb.h, 2 -> adf_gemm_rbx3.h

}  // namespace adf
