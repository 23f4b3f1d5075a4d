// "Tile" implicit-GEMM kernel for the stride-1 3-tap convs of the resblocks (bf16 throughput mode): the construction that made
// the fused short-level kernels (adf_resblock_small.h) fast, as a stand-alone conv.
//   * A persistent 512-thread workgroup takes a tile of TM consecutive positions of one sample: the (concatenated) input rows
//     with their two halo rows are staged into LDS ONCE, through the GroupNorm/FiLM/SiLU prologue (table derived in the
//     kernel from the statistics, gn_affine<FAST>), rounded to bf16 -- a separate phase, no vector work inside the K loop;
//   * every wave owns 32 output columns (all 8 waves side by side for 256 output channels; 4 column groups x 2 row halves
//     for 128) and all its tile rows: the K loop is one 16-byte weight-fragment load per lane (fragment-major copy of the
//     weights, straight from L2 into the MFMA B operand, 16 K steps ahead), MT ds_read_b128 A fragments and MT MFMAs per K
//     step -- no barrier, no LDS traffic for weights, no DMA bookkeeping.  Weights are re-read per tile from L2
//     (~34 B/clk/CU, which 4 row tiles of MFMA work per fragment balance);
//   * the 1x1 residual conv of the raw concat runs first on a raw copy of the tile in the same LDS buffer (accumulators kept),
//     an identity residual initialises the accumulators; bias, GroupNorm statistics (fp64 atomics) and the bf16 rows leave
//     through LDS as 16-byte stores.
// Against the LDS-DMA kernel (adf_gemm_pp.h: ~10 vector + 10 scalar instructions per MFMA, three barriers per 48 MFMAs) this K
// loop issues ~2.5 instructions per MFMA -- but the staging phase and the epilogue do not overlap with it and the weights are
// re-read from L2 for every tile.  MEASURED (profiles/README.md): equal to the DMA kernel at 256 channels (L = 1024: 53-60 vs
// 45-56 us at K = 768, 91 vs 93 us at K = 1536), 25-50 % slower at 128 channels (two waves share every weight fragment); 334 vs
// 319 ms per bench step.  It is therefore an experiment route (ADF_GEMM_TILE=1), not the default.
#pragma once
#include "adf_gemm.h"
#include <type_traits>

namespace adf {

// N = output channels (128 or 256), CIN = input channels of the 3-tap segment (N or 2 N), CRES = channels of the raw concat of
// the 1x1 residual segment (0 = none)
template <int N, int CIN, int CRES>
struct TileCfg {
    static constexpr int WCOLS = N / 32;                 // column groups of 32
    static constexpr int WROWS = 8 / WCOLS;              // row groups
    static constexpr int MT = 4;                          // 32-row MFMA tiles per wave
    static constexpr int TM = 32 * MT * WROWS;            // 128 (N = 256) or 256 (N = 128) positions per tile
    static constexpr int CMAX = CIN > CRES ? CIN : CRES;
    static constexpr int PX = CMAX * 2 + 16;              // LDS row pitch
    static constexpr int RX = TM + 2;
    static constexpr int kTabOfs = RX * PX;               // [CIN][2] floats
    static constexpr int kPrmOfs = kTabOfs + CIN * 8;     // bias [N] floats
    static constexpr int kLds = kPrmOfs + N * 4;
};

template <int N, int CIN, int CRES>
__global__ void __launch_bounds__(512) conv_gemm_tile_kernel(const GemmArgs a, int tiles_total, int tiles_per_sample) {
    typedef TileCfg<N, CIN, CRES> Cfg;
    constexpr int MT = Cfg::MT, TM = Cfg::TM, PX = Cfg::PX, RX = Cfg::RX, WCOLS = Cfg::WCOLS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufX = smem;
    float* const tab = (float*)(smem + Cfg::kTabOfs);
    float* const prm = (float*)(smem + Cfg::kPrmOfs);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave % WCOLS, wr = wave / WCOLS;       // column group, row group
    const int col = wc * 32 + r;
    const int rowbase = wr * 32 * MT;                     // first tile row of this wave
    const GemmSeg& sg = a.seg[0];
    const int L = a.lin;

    auto gemm = [&](auto tapsc, auto chc, const void* W, int row0, f32x16_t (&acc)[MT]) __attribute__((always_inline)) {
        constexpr int TAPS = decltype(tapsc)::value;
        constexpr int KS = (decltype(chc)::value / 64) * TAPS * 4;
        const char* const wl = (const char*)W + ((size_t)hh * N + col) * 16;
        auto wfrag = [&](int ks) __attribute__((always_inline)) -> bf16x8_t {
            return __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(wl + (size_t)ks * 2 * N * 16));
        };
        constexpr int DEPTH = 15 < KS ? 15 : KS;         // 15 KB of weight loads in flight per wave; RING even: A-fragment parity is static
        constexpr int RING = DEPTH + 1;
        static_assert(RING % 2 == 0, "ring");
        bf16x8_t wf[RING];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) wf[d] = wfrag(d);
        __builtin_amdgcn_sched_barrier(0);
        // A fragments one K step ahead (their LDS latency would otherwise sit between every load and its four MFMAs)
        auto afrag = [&](int ks, bf16x8_t (&af)[MT]) __attribute__((always_inline)) {
            const int ct = ks >> 2, q = ks & 3;
            const int chunk = ct / TAPS, tap = ct - chunk * TAPS;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *(const bf16x8_t*)(bufX + (rowbase + i * 32 + r + tap + row0) * PX + chunk * 128 + q * 32 + hh * 16);
        };
        bf16x8_t af[2][MT];
        afrag(0, af[0]);
#pragma unroll 1
        for (int kb = 0; kb < KS; kb += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int ks = kb + u;
                if (ks < KS) {
                    if (ks + DEPTH < KS) wf[(u + DEPTH) % RING] = wfrag(ks + DEPTH);
                    if (ks + 1 < KS) afrag(ks + 1, af[(u + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][i], wf[u], acc[i], 0, 0, 0);
                }
            }
        }
    };
    auto row_of = [&](int i, int e) __attribute__((always_inline)) -> int { return rowbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh; };

    // biases once per workgroup
    for (int n = tid; n < N; n += 512) prm[n] = (a.bias0 ? a.bias0[n] : 0.f) + (a.bias1 ? a.bias1[n] : 0.f);

    const int nblk = (int)gridDim.x, bidx = (int)blockIdx.x;
    const int t_lo = (int)((long long)bidx * tiles_total / nblk), t_hi = (int)((long long)(bidx + 1) * tiles_total / nblk);
    int tab_b = -1;
    for (int t = t_lo; t < t_hi; ++t) {
        const int b = t / tiles_per_sample, m0 = (t - b * tiles_per_sample) * TM;
        // one staged 16-byte chunk of a (concatenated) input: position p of the sample, chunk cc
        auto src_chunk = [&](const GemmSeg& g, int p, int cc) __attribute__((always_inline)) -> u32x4_t {
            const int c = cc * 8;
            return c < g.c0 ? *(const u32x4_t*)((const bf16_t*)g.src0 + ((size_t)b * L + p) * g.c0 + c)
                            : *(const u32x4_t*)((const bf16_t*)g.src1 + ((size_t)b * L + p) * g.c1 + (c - g.c0));
        };
        if (sg.gn.gamma && b != tab_b) {                   // affine table of this sample (uniform branch)
            __syncthreads();
            if (tid < CIN) {
                float A, Bc;
                gn_affine<true>(sg.gn, b, tid, A, Bc);
                tab[2 * tid] = A; tab[2 * tid + 1] = Bc;
            }
            tab_b = b;
        }
        __syncthreads();                                   // previous tile's output rows have left bufX; table / biases visible
        f32x16_t acc[MT];
        if constexpr (CRES > 0) {
            // raw concat rows (source 1 scaled as the unfused path's raw segment does) -> bufX rows 1 .. TM; 1x1 conv into acc
            const GemmSeg& s1 = a.seg[1];
            constexpr int CPRR = CRES / 8;
            for (int idx = tid; idx < TM * CPRR; idx += 512) {
                const int row = idx / CPRR, cc = idx % CPRR;
                u32x4_t v = src_chunk(s1, m0 + row, cc);
                if (cc * 8 >= s1.c0 && s1.scale1 != 1.0f) {
                    float f[8];
                    unpack16<bf16_t>(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= s1.scale1;
                    v = pack16<bf16_t>(f);
                }
                *(u32x4_t*)(bufX + (row + 1) * PX + cc * 16) = v;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            gemm(std::integral_constant<int, 1>{}, std::integral_constant<int, CRES>{}, s1.wfrag, 1, acc);
            __syncthreads();
        } else if (a.res) {
            // identity residual: its tile goes through LDS (16-byte loads), the accumulators start from it
            const bf16_t* const rs = (const bf16_t*)a.res + ((size_t)b * L + m0) * N;
            for (int idx = tid; idx < TM * (N / 8); idx += 512) {
                const int row = idx / (N / 8), cc = idx % (N / 8);
                *(u32x4_t*)(bufX + row * PX + cc * 16) = *(const u32x4_t*)(rs + (size_t)row * N + cc * 8);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = bf16_to_f32(*(const unsigned short*)(bufX + row_of(i, e) * PX + col * 2));
            __syncthreads();
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        }
        // activated rows -1 .. TM of the tile -> bufX rows 0 .. TM + 1 (zeros outside the sample)
        constexpr int CPR = CIN / 8;
        for (int idx = tid; idx < RX * CPR; idx += 512) {
            const int row = idx / CPR, cc = idx % CPR;
            const int p = m0 + row - 1;
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            if (p >= 0 && p < L) {
                float f[8];
                unpack16<bf16_t>(src_chunk(sg, p, cc), f);
                if (sg.gn.gamma) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float w = fmaf(f[e], tab[2 * (cc * 8 + e)], tab[2 * (cc * 8 + e) + 1]);
                        f[e] = sg.act ? silu_f(w) : w;
                    }
                }
                v = pack16<bf16_t>(f);
            }
            *(u32x4_t*)(bufX + row * PX + cc * 16) = v;
        }
        __syncthreads();
        gemm(std::integral_constant<int, 3>{}, std::integral_constant<int, CIN>{}, sg.wfrag, 0, acc);
        __syncthreads();                                   // every wave is done reading the tile
        // epilogue: + bias, statistics, bf16 rows -> bufX -> global
        {
            const float bias = prm[col];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][e] + bias;
                    s1 += v; s2 = fmaf(v, v, s2);
                    *(unsigned short*)(bufX + row_of(i, e) * PX + col * 2) = f32_to_bf16(v);
                }
            if (a.stats) {
                constexpr int GS = N / 8;                  // channels per group: 16 (two groups in a wave's columns) or 32
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
#pragma unroll
                for (int o = GS / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (lane < 32 && (lane % GS) == 0) {
                    double* sp = a.stats + ((size_t)b * 8 + col / GS) * 2;
                    atomicAdd(sp, (double)s1);
                    atomicAdd(sp + 1, (double)s2);
                }
            }
        }
        __syncthreads();
        {
            bf16_t* const ob = (bf16_t*)a.out + ((size_t)b * L + m0) * N;
            for (int idx = tid; idx < TM * (N / 8); idx += 512) {
                const int row = idx / (N / 8), cc = idx % (N / 8);
                *(u32x4_t*)(ob + (size_t)row * N + cc * 8) = *(const u32x4_t*)(bufX + row * PX + cc * 16);
            }
        }
    }
}

}  // namespace adf
