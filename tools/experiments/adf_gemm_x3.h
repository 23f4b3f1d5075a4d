// EXPERIMENT, not shipped (round 4).  Built into the product, parity-green (every f32x3 test), measured, taken out: 729.8 / 729.9 ms per bench step against 671.4 /
// 670.7 on the generic kernel (tools/calls/r04_call11.sh), per launch 132.9 us against 114.7 us on the 1024-tile launches.  Knock-out builds (-DADF_X3_KNOCK,
// tools/calls/r04_call12.sh; us per launch of that class): full 137.0 | no MFMA 104.5 | no activation staging 80.2 | no weight DMA 128.4 | no epilogue stores 118.5 |
// none of them 32.3 -- the ACTIVATION path is what a launch waits for, not the matrix pipe: fp32 rows requested ONE chunk (three steps = 1.1 us of MFMAs) ahead arrive
// from HBM / the memory-side cache later than that under full load, every tile pays a cold fill (one workgroup per CU: nothing else to run), and the generic kernel's
// two workgroups per CU hide exactly that.  What would make it win: persistent workgroups (no per-tile fill) with the rows three chunks ahead (an LDS-DMA raw ring:
// 151 KB of LDS with everything else) and an epilogue that does not overlay the stages -- priced at a rewrite, not done.  The two earlier versions (weights through
// registers one step ahead: 146 us; weights and activations in separate producer waves: 206 us) are described below.
// Split-bf16 ("f32x3", adf_common.h) implicit GEMM for the large stride-1 layers: a producer / consumer kernel that STREAMS its weights.
// (reference call sites: src/models/backbones/unet1d.py:193-207 ConvBlock1d, :297-316 ResnetBlock1d; the 1x1 projections of :49-61 and attention_utils.py:117-184)
//
// Why a kernel of its own (round 4): in this mode a product costs three bf16 MFMAs, so the generic kernel (adf_gemm.h: stage -> barrier -> MFMA -> barrier, the
// phases of one workgroup in series, two workgroups per CU overlapping by chance) sat at 59 % of the matrix pipe on its biggest launches and lower elsewhere
// (profiles/r04_f32x3_mode_per_nfe_summary.txt: 5.2 of a 6.9 ms evaluation).  Here the two roles run CONCURRENTLY on every SIMD:
//   * waves 0-3 ("consumers"): one 64 x 64 output tile each of the workgroup's 128 x 128, per step 2 K steps x 4 tiles x 3 = 24 MFMAs on fragments read from LDS
//     (hi | lo halves of a 128-byte row: adf_common.h).  They also ISSUE the weight stream: the slab of a step (one tap of one 32-channel chunk: 128 rows x 128 B,
//     split at pack time) goes global -> LDS by LDS-DMA (four 1 KB pieces per wave), THREE steps ahead, into a ring of four stages (128-byte rows, 16-byte chunk c of
//     row r at slot c ^ ((r >> 1) & 7), applied on the DMA's source side: conflict-free ds_read_b128), waited for with a counted vmcnt;
//   * waves 4-7 ("producers"): once per chunk the activation rows of the NEXT chunk (128 + taps - 1 rows x 32 channels fp32), requested a whole chunk ahead:
//     GroupNorm / FiLM affine + SiLU (or the raw scale), split into bf16 hi + lo, LDS (two stages);
//   * a step = one (chunk, tap); ONE LDS-only barrier per step.
// A SIMD holds one wave of each role (waves w and w + 4), so the producer's loads, vector work and LDS stores issue in the gaps of the consumer's MFMA stream
// without any hand interleaving.  One workgroup per CU.
// Two versions before this one measured SLOWER than the generic kernel (146 / 206 us per launch against 115): the weight slab through registers ONE step ahead
// leaves it a single step (0.4 us of MFMAs) to arrive from L2 under full load (it needs ~1.5 us), so every step waited for its weights; and vmcnt retires in issue
// order per wave, so weights and activations requested by the same wave wait for each other.  Hence: weights three steps ahead, in the consumers' own queue.
// Shapes (launch_conv_gemm): per-sample tiles, mrows = lin = out_rows a multiple of 128, n = n_pad = out_c a multiple of 128, stride 1, step +1, segment 0 with 3 taps
// (off0 -1) or 1 tap (off0 0), optional segment 1 with 1 tap; channels per source multiples of 32; >= 256 tiles.  Everything else stays on adf_gemm.h.
// Epilogue: as adf_gemm.h (bias, identity residual, GELU, 16-byte stores, GroupNorm statistics of the stored tile by fp64 atomics), all eight waves.
#pragma once
#include "../../audiodiffuser_amd/csrc/adf_gemm.h"

namespace adf {

// timing knock-outs of diagnostic builds only (tools/build_variant.sh NAME -DADF_X3_KNOCK=bits; results are wrong by construction):
// 1 no MFMA, 2 no activation staging (loads + prologue + LDS stores), 4 no weight DMA, 8 no epilogue stores / atomics, 16 no fragment reads
#ifndef ADF_X3_KNOCK
#define ADF_X3_KNOCK 0
#endif

constexpr int kX3ARows = 130;                                   // 128 + 2 halo rows
constexpr int kX3AStage = kX3ARows * kLdsPitch;                 // 18,720 B
constexpr int kX3WStage = 128 * 128;                            // 16,384 B: unpadded rows (LDS-DMA writes 1 KB pieces), swizzled
constexpr int kX3WDepth = 4;                                    // weight stages: the slab of step s + 3 is issued in step s
constexpr int kX3Lds = 2 * kX3AStage + kX3WDepth * kX3WStage;   // 102,976 B (the epilogue's 128 x 128 fp32 image, 64 KB, overlays it)
__device__ __forceinline__ int x3_wswz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__global__ void __launch_bounds__(512) conv_gemm_x3p_kernel(const GemmArgs a) {
    constexpr int TM = 128, TN = 128, KC = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsW = smem + 2 * kX3AStage;
    const int tid512 = (int)threadIdx.x;
    const bool producer = __builtin_amdgcn_readfirstlane(tid512) >= 256;          // scalar role branch (waves 4-7)
    const int tid = tid512 & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = a.n_pad / TN, tiles_m = a.mrows / TM;
    int bid = (int)blockIdx.x;
    const int tn_i = bid % tiles_n; bid /= tiles_n;
    const int tm_i = bid % tiles_m;
    const int b0 = bid / tiles_m, m0 = tm_i * TM, n0 = tn_i * TN;

    // the step sequence, identical in both roles: for seg, for chunk, for tap
    const int nch0 = a.seg[0].nchunk, taps0 = a.seg[0].taps;
    const int nch1 = a.nseg > 1 ? a.seg[1].nchunk : 0;
    const int steps0 = nch0 * taps0, nsteps = steps0 + nch1;                    // (segment 1 has one tap)
    const int nchunks = nch0 + nch1;
    // chunk index q (over both segments) -> (segment, chunk inside it)
    auto seg_of = [&](int q) __attribute__((always_inline)) -> const GemmSeg& { return q >= nch0 ? a.seg[1] : a.seg[0]; };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- producer state -----------------------------------------------------------------------------------------------------------------------
    const int c16 = tid & 7;                                   // this thread's 16-byte column of every staged row (4 fp32 channels)
    const int row0 = tid >> 3;                                 // first staged row (then + 32 per item)
    constexpr int NA = 5;                                      // items per thread: 130 rows x 8 columns over 256 threads
    u32x4_t ra[NA];
    f32x4_t abq[2];
    float raw_scale = 1.0f;
    unsigned avalid = 0;
    bool a_act = false, a_ab = false;
    int a_rows = 0;
    // weight slab of `step` -> ring stage step % 4, this consumer wave's four 1 KB pieces (rows 32 wave .. 32 wave + 31 of the slab)
    // lane offsets of the four pieces inside the slab: row = 32 wave + 8 piece + (lane >> 3), LDS slot = lane & 7 <- source chunk (lane & 7) ^ ((row >> 1) & 7)
    unsigned wl[4];
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
        const int row = wave * 32 + pc * 8 + (lane >> 3);
        wl[pc] = (unsigned)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
    auto issue_w = [&](int step) __attribute__((always_inline)) {
        const bool s1 = step >= steps0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int ct = s1 ? step - steps0 : step;                           // chunk * taps + tap inside the segment
        const char* wp = uniform_ptr(sg.w) + ((size_t)ct * a.n_pad + n0) * kRowBytes;
        const unsigned l0 = (unsigned)(2 * kX3AStage + (step & (kX3WDepth - 1)) * kX3WStage + wave * 4096);
        if (ADF_X3_KNOCK & 4) return;
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %0\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %0"
                     :: "s"(wp), "v"(wl[0]), "v"(wl[1]), "v"(wl[2]), "v"(wl[3]), "s"(l0) : "memory", "scc");
    };
    auto load_a = [&](int q) __attribute__((always_inline)) {               // rows of chunk q (over both segments) -> ra, its affine -> abq
        if (ADF_X3_KNOCK & 2) return;
        const GemmSeg& sg = seg_of(q);
        const int chunk = q >= nch0 ? q - nch0 : q;
        const int nrows = TM + sg.taps - 1;
        const int p_lo = m0 + sg.off0;
        const int ctot = sg.c0 + sg.c1;
        const int cidx = chunk * KC + c16 * 4;
        const int c0u = __builtin_amdgcn_readfirstlane(sg.c0), c1u = __builtin_amdgcn_readfirstlane(sg.c1);
        const bool from1 = c1u > 0 && cidx >= c0u;
        const char* src0u = uniform_ptr(sg.src0);
        const char* src1u = uniform_ptr(sg.src1);
        const char* src = from1 ? src1u : src0u;
        const unsigned rowbytes = (unsigned)(from1 ? c1u : c0u) * 4u;
        const unsigned colbytes = (unsigned)(from1 ? cidx - c0u : cidx) * 4u;
        avalid = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = row0 + i * 32;
            const int p = p_lo + row;
            const bool ok = row < nrows && p >= 0 && p < a.lin;
            const unsigned off = ok ? (unsigned)(b0 * a.lin + p) * rowbytes + colbytes : 0u;
            ra[i] = *(const u32x4_t*)(src + off);
            avalid |= (ok ? 1u : 0u) << i;
        }
        const float* abp = sg.ab ? sg.ab + (unsigned)(b0 * ctot + cidx) * 2u : (const float*)sg.w;     // (raw input: any valid address, the values are ignored)
        abq[0] = *(const f32x4_t*)abp;
        abq[1] = *(const f32x4_t*)(abp + 4);
        raw_scale = from1 ? sg.scale1 : 1.0f;
        a_act = sg.act != 0; a_ab = sg.ab != nullptr; a_rows = nrows;
    };
    auto store_a = [&](int stage) __attribute__((always_inline)) {          // prologue + split of the rows in ra -> LDS
        if (ADF_X3_KNOCK & 2) return;
        char* const dst = ldsA + stage * kX3AStage;
        float fa[4], fb[4];
        fa[0] = a_ab ? abq[0].x : raw_scale; fb[0] = a_ab ? abq[0].y : 0.f;
        fa[1] = a_ab ? abq[0].z : raw_scale; fb[1] = a_ab ? abq[0].w : 0.f;
        fa[2] = a_ab ? abq[1].x : raw_scale; fb[2] = a_ab ? abq[1].y : 0.f;
        fa[3] = a_ab ? abq[1].z : raw_scale; fb[3] = a_ab ? abq[1].w : 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = row0 + i * 32;
            if (row < a_rows) {
                float f[4] = {0.f, 0.f, 0.f, 0.f};
                if ((avalid >> i) & 1u) {
                    unpack16<float>(ra[i], f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaf(f[e], fa[e], fb[e]);
                        f[e] = a_act ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)) : v;
                    }
                }
                u32x2_t hi, lo;
                split_bf16x4(f, hi, lo);
                *(u32x2_t*)(dst + lds_swz(row, c16 >> 1) + (c16 & 1) * 8) = hi;
                *(u32x2_t*)(dst + lds_swz(row, 4 + (c16 >> 1)) + (c16 & 1) * 8) = lo;
            }
        }
    };

    // ---- fill: chunk 0 and the first slab into stage 0; chunk 1 and the second slab on their way ---------------------------------------------------
    if (producer) {
        load_a(0);
        store_a(0);
        if (nchunks > 1) load_a(1);
    } else {
        issue_w(0);
        if (nsteps > 1) issue_w(1);
        if (nsteps > 2) issue_w(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- steps ---------------------------------------------------------------------------------------------------------------------------------------
    int q = 0, tap = 0;                                         // chunk (over both segments) and tap of the current step
    for (int s = 0; s < nsteps; ++s) {
        const int taps_q = q >= nch0 ? 1 : taps0;
        if (!producer) {
            const char* const A = ldsA + (q & 1) * kX3AStage;
            const char* const W = ldsW + (s & (kX3WDepth - 1)) * kX3WStage;
            if (s + 3 < nsteps) issue_w(s + 3);                 // (its stage was read last in step s - 1, whose barrier has passed)
            int arow[2], wrow[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) arow[i] = wm * 64 + i * 32 + r + tap;            // staged row = output row + tap (3 taps: off0 -1; 1 tap: tap = 0)
#pragma unroll
            for (int j = 0; j < 2; ++j) wrow[j] = wn * 64 + j * 32 + r;
            bf16x8_t ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[0][i] = *(const bf16x8_t*)(A + lds_swz(arow[i], h));
                al[0][i] = *(const bf16x8_t*)(A + lds_swz(arow[i], 4 + h));
                bh[0][i] = *(const bf16x8_t*)(W + x3_wswz(wrow[i], h));
                bl[0][i] = *(const bf16x8_t*)(W + x3_wswz(wrow[i], 4 + h));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[1][i] = *(const bf16x8_t*)(A + lds_swz(arow[i], 2 + h));
                        al[1][i] = *(const bf16x8_t*)(A + lds_swz(arow[i], 6 + h));
                        bh[1][i] = *(const bf16x8_t*)(W + x3_wswz(wrow[i], 2 + h));
                        bl[1][i] = *(const bf16x8_t*)(W + x3_wswz(wrow[i], 6 + h));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ADF_X3_KNOCK & 1) { asm volatile("" :: "v"(al[ks][i]), "v"(ah[ks][i]), "v"(bh[ks][j]), "v"(bl[ks][j])); continue; }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                    }
            }
            // the slab of the NEXT step has landed (this wave's pieces; the barrier below covers the others'): at most the two slabs behind it still fly
            if (s + 3 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // last step of a chunk: the next chunk's rows (in registers since the previous chunk) -> the other activation stage; the chunk after it starts
            if (tap == taps_q - 1 && q + 1 < nchunks) {
                store_a((q + 1) & 1);
                if (q + 2 < nchunks) load_a(q + 2);
            }
        }
        if (++tap == taps_q) { tap = 0; ++q; }
        // LDS-only barrier: __syncthreads() is a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) as well -- it made the producers wait at EVERY step for the slab
        // (and the rows) they had just requested for later steps: 6 K cycles per step instead of the consumers' 0.8 K (first version: 146 us per launch against
        // 115 us on the generic kernel).  The producers' global loads stay in flight across this barrier; the compiler waits for them where store_a uses them.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();

    // ---- epilogue phase 1: consumers' accumulators (+ bias) -> LDS fp32 image [128][128] (over the stages: every wave is past the last barrier) -------------
    float* const tile = (float*)smem;
    if (!producer) {
        float bias_c[2];
        {
            int bi[2]; bool okb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int n = n0 + wn * 64 + j * 32 + r; okb[j] = n < a.n; bi[j] = n % a.bias_mod; }
            gemm_bias_load<2>(a, bi, okb, bias_c);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    tile[row * TN + col] = acc[i][j][e] + bias_c[j];
                }
        }
    }
    __syncthreads();
    // ---- phase 2: all 512 threads: 16-byte stores along the channel axis (+ residual / GELU), statistics of the stored values ----------------------------------
    {
        float* out = (float*)a.out;
        const float* res = (const float*)a.res;
        constexpr int CPR = TN / 4;                            // 32 chunks per tile row
        constexpr int P2 = TM * CPR / 512;                     // 8 chunks per thread
        constexpr int RPK = 512 / CPR;                         // 16 tile rows per k step
        const int cc = tid512 % CPR, rbase = tid512 / CPR;
        const int n = n0 + cc * 4;
        const bool do_stats = a.stats != nullptr;
        const int gs = do_stats ? a.out_c / a.stats_groups : 4;
        const int tpg = gs / 4;
        unsigned off[P2];
        u32x4_t rres[P2];
#pragma unroll
        for (int k = 0; k < P2; ++k) {
            const int m = m0 + rbase + k * RPK;
            off[k] = (unsigned)((b0 * a.out_rows + m) * a.out_c + n);
        }
        if (res) {
#pragma unroll
            for (int k = 0; k < P2; ++k) rres[k] = *(const u32x4_t*)(res + off[k]);
        } else {
#pragma unroll
            for (int k = 0; k < P2; ++k) rres[k] = u32x4_t{0u, 0u, 0u, 0u};
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < P2; ++k) {
            const int row = rbase + k * RPK;
            const float4 qv = *(const float4*)(tile + row * TN + cc * 4);
            float v[4] = {qv.x, qv.y, qv.z, qv.w}, rr[4];
            unpack16<float>(rres[k], rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
            if (a.gelu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
            }
            if (!(ADF_X3_KNOCK & 8)) *(u32x4_t*)(out + off[k]) = pack16<float>(v);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
        }
        if (do_stats && !(ADF_X3_KNOCK & 8)) {
            // a thread's chunks all cover the same 4 channels; a wave covers 2 tile rows x 32 chunks per k step: reduce over the channels of a group (adjacent
            // cc), then over the two row-lanes of the wave, one fp64 atomic pair per (wave, group)
            for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if ((tid512 & 63) < CPR && (cc & (tpg - 1)) == 0) {
                double* sp = a.stats + ((size_t)b0 * a.stats_groups + n / gs) * 2;
                atomicAdd(sp, (double)s1);
                atomicAdd(sp + 1, (double)s2);
            }
        }
    }
}

}  // namespace adf

// ---- the route in launch_conv_gemm (adf_gemm.hip, ahead of the split-K block) while it was in the product ----
#if 0
    if (x3) {
        // split-bf16 mode, large stride-1 layers: the producer / consumer kernel of adf_gemm_x3.h (ADF_GEMM_X3P=0: the generic kernel, route test / A/B)
        static int use_x3p = -1;
        if (use_x3p < 0) use_x3p = adf_route_switch("ADF_GEMM_X3P", 1);
        bool ok = use_x3p && !a.scatter_f && !flat && a.mrows % 128 == 0 && a.lin == a.mrows && a.out_rows == a.mrows && a.n == a.n_pad && a.out_c == a.n &&
                  a.n % 128 == 0 && a.bias_mod == a.n && (long long)a.B * (a.mrows / 128) * (a.n / 128) >= 256;
        for (int sgi = 0; sgi < a.nseg && ok; ++sgi) {
            const GemmSeg& g = a.seg[sgi];
            if (g.stride != 1 || g.step != 1 || g.c0 % 32 || g.c1 % 32 || g.c0 + g.c1 != g.nchunk * 32) ok = false;
            if (!((g.taps == 3 && g.off0 == -1) || (g.taps == 1 && g.off0 == 0))) ok = false;
            if (sgi == 1 && g.taps != 1) ok = false;
        }
        if (ok) {
            GemmArgs b = a;
            b.stats = nullptr;
            if (a_in.stats) {
                const int gs = a.stats_groups > 0 ? a.out_c / a.stats_groups : 0;
                const bool sok = gs >= 4 && gs <= 128 && gs * a.stats_groups == a.out_c && (gs & (gs - 1)) == 0;
                b.stats = sok ? a_in.stats : nullptr;
                if (stats_fused) *stats_fused = sok;
            }
            if (const char* e = settle_gn(false)) return e;
            b.seg[0].gn.gamma = nullptr;
            static bool attr_done[kMaxDevices] = {};
            bool& attr_set = attr_done[current_device()];
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)conv_gemm_x3p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kX3Lds) != hipSuccess)
                    return "hipFuncSetAttribute(MaxDynamicSharedMemorySize, x3p) failed";
                attr_set = true;
            }
            const long long blocks = (long long)a.B * (a.mrows / 128) * (a.n / 128);
            if (blocks > 0x7fffffffLL) return "conv_gemm_x3p: bad grid";
            trace_route("x3p", b, 128, 128);
            hipLaunchKernelGGL(conv_gemm_x3p_kernel, dim3((unsigned)blocks), dim3(512), kX3Lds, stream, b);
            return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm_x3p: launch failed";
        }
    }
#endif
