// Transposed convs of the up path (Upsample1d, unet1d.py:230-256: ConvTranspose1d k = 2f, stride f) in bf16 throughput mode,
// built like the fused short-level kernels: as a GEMM over m = 0 .. L with two taps (x[m], x[m-1]) and N = f * Cout columns
// (column n = phase * Cout + co lands in output row m f + phase - f/2), a persistent 512-thread workgroup stages the raw input
// rows of a tile of m into LDS once, every wave owns 32 columns per pass and reads its weight fragments straight from a
// fragment-major copy in L2 (15 K steps ahead, A fragments one step ahead), the 256-column passes reuse the staged tile; the
// output of a pass goes through LDS and leaves as 16-byte stores; GroupNorm statistics by fp64 atomics.
// The plain / weight-stationary kernels ran these six launches at 260-325 TF/s (0.29 ms of a 3.2 ms network pass).
#pragma once
#include "adf_gemm.h"
#include <type_traits>

namespace adf {

template <int CIN, int COUT, int F, int MTP>
struct UpCfg {
    static constexpr int N = F * COUT;
    static constexpr int PASSW = N >= 256 ? 256 : N;      // columns per pass
    static constexpr int NPASS = N / PASSW;
    static constexpr int WCOLS = PASSW / 32, WROWS = 8 / WCOLS;
    static constexpr int MT = MTP;                         // 32-row MFMA tiles per wave (2 or 4)
    static constexpr int TM = 32 * MT * WROWS;             // values of m per tile
    static constexpr int PX = CIN * 2 + 16, PO = PASSW * 2 + 16;
    static constexpr int kOfsO = (TM + 1) * PX;
    static constexpr int kOfsB = kOfsO + TM * PO;          // bias [COUT] floats
    static constexpr int kLds = kOfsB + COUT * 4;
};

// A work item is (sample, tile of m, column pass): the passes of a tile go to different workgroups (each re-stages the few input
// rows), so that the short levels -- one tile of m per sample, 1 MB of weights -- still spread over all CUs.
template <int CIN, int COUT, int F, int MTP>
__global__ void __launch_bounds__(512) conv_gemm_up_kernel(const GemmArgs a, int tiles_total, int tiles_per_sample) {
    typedef UpCfg<CIN, COUT, F, MTP> Cfg;
    constexpr int MT = Cfg::MT, TM = Cfg::TM, PX = Cfg::PX, PO = Cfg::PO, PASSW = Cfg::PASSW, WCOLS = Cfg::WCOLS, N = Cfg::N;
    constexpr int KS = (CIN / 64) * 2 * 4;                 // K steps of 16 channels: chunks x 2 taps x 4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufX = smem;                               // [TM + 1][PX]: input positions m0 - 1 .. m0 + TM - 1 (zeros outside the sample)
    char* const bufO = smem + Cfg::kOfsO;                  // [TM][PO]: one pass of output columns
    float* const bias = (float*)(smem + Cfg::kOfsB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave % WCOLS, wr = wave / WCOLS;
    const int rowbase = wr * 32 * MT;
    const GemmSeg& sg = a.seg[0];
    const int L = a.lin, Lout = a.out_rows;
    const bf16_t* const src = (const bf16_t*)sg.src0;
    for (int n = tid; n < COUT; n += 512) bias[n] = a.bias0 ? a.bias0[n] : 0.f;
    auto row_of = [&](int i, int e) __attribute__((always_inline)) -> int { return rowbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh; };

    const int nblk = (int)gridDim.x, bidx = (int)blockIdx.x;
    const int t_lo = (int)((long long)bidx * tiles_total / nblk), t_hi = (int)((long long)(bidx + 1) * tiles_total / nblk);
    for (int tt = t_lo; tt < t_hi; ++tt) {
        const int t = tt / Cfg::NPASS, pass = tt - t * Cfg::NPASS;
        const int b = t / tiles_per_sample, m0 = (t - b * tiles_per_sample) * TM;
        __syncthreads();                                   // the previous tile is out of bufX / bufO
        constexpr int CPR = CIN / 8;
        // the pieces of a thread are loaded four at a time (unconditionally, rows outside the sample on a clamped address) before their LDS stores:
        // as load -> store per trip under a condition these were (TM + 1) CPR / 512 serialised memory round trips per tile (four at a time: the
        // weight-fragment ring is live here, eight would spill)
        constexpr int XS = ((TM + 1) * CPR + 511) / 512, XG = 4;
#pragma unroll
        for (int g0 = 0; g0 < XS; g0 += XG) {
            u32x4_t xs[XG];
#pragma unroll
            for (int k = 0; k < XG; ++k) {
                const int idx = tid + (g0 + k) * 512;
                const int row = idx < (TM + 1) * CPR ? idx / CPR : 0, cc = idx % CPR;
                const int p = m0 - 1 + row;
                const int pc = p < 0 ? 0 : (p < L ? p : L - 1);
                xs[k] = *(const u32x4_t*)(src + ((size_t)b * L + pc) * CIN + cc * 8);
            }
#pragma unroll
            for (int k = 0; k < XG; ++k) {
                const int idx = tid + (g0 + k) * 512;
                if (g0 + k >= XS || idx >= (TM + 1) * CPR) continue;
                const int row = idx / CPR, cc = idx % CPR;
                const int p = m0 - 1 + row;
                *(u32x4_t*)(bufX + row * PX + cc * 16) = (p >= 0 && p < L) ? xs[k] : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        {
            const int col = pass * PASSW + wc * 32 + r;     // this lane's column n
            f32x16_t acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            {
                const char* const wl = (const char*)sg.wfrag + ((size_t)hh * a.n_pad + col) * 16;
                auto wfrag = [&](int ks) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(wl + (size_t)ks * 2 * a.n_pad * 16));
                };
                constexpr int DEPTH = 15 < KS ? 15 : KS, RING = DEPTH + 1;
                static_assert(RING % 2 == 0 && KS >= DEPTH, "ring");
                bf16x8_t wf[RING];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) wf[d] = wfrag(d);
                __builtin_amdgcn_sched_barrier(0);
                // K step ks = (chunk * 2 + tap) * 4 + q: tap t reads x[m - t] = staged row (m - m0) + 1 - t
                auto afrag = [&](int ks, bf16x8_t (&af)[MT]) __attribute__((always_inline)) {
                    const int ct = ks >> 2, q = ks & 3;
                    const int chunk = ct >> 1, tap = ct & 1;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[i] = *(const bf16x8_t*)(bufX + (rowbase + i * 32 + r + 1 - tap) * PX + chunk * 128 + q * 32 + hh * 16);
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
            }
            // epilogue of the pass: + bias, statistics over the rows that exist, bf16 -> bufO
            {
                const int phase = col / COUT, co = col - phase * COUT;
                const float bv = bias[co];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row_of(i, e);
                        const int orow = (m0 + row) * F + phase - F / 2;
                        const unsigned short qv = f32_to_bf16_hw(acc[i][e] + bv);
                        const float v = bf16_to_f32(qv);                 // statistics of the stored values (adf_common.h pack16_stored)
                        if (m0 + row <= L && orow >= 0 && orow < Lout) { s1 += v; s2 = fmaf(v, v, s2); }
                        *(unsigned short*)(bufO + row * PO + (wc * 32 + r) * 2) = qv;
                    }
                if (a.stats) {
                    constexpr int GS = COUT / 8;           // channels per group (8, 16 or 32): the wave's 32 columns hold 32 / GS groups
                    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
#pragma unroll
                    for (int o = GS / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    if (lane < 32 && (lane % GS) == 0) {
                        double* sp = a.stats + ((size_t)b * 8 + co / GS) * 2;
                        atomicAdd(sp, (double)s1);
                        atomicAdd(sp + 1, (double)s2);
                    }
                }
            }
            __syncthreads();
            // rows of the pass -> global: (m, local phase) -> output row, COUT channels = COUT / 8 chunks of 16 bytes
            {
                constexpr int PPP = PASSW / COUT;            // phases per pass
                constexpr int CPO = COUT / 8;
                bf16_t* const ob = (bf16_t*)a.out + (size_t)b * Lout * COUT;
                for (int idx = tid; idx < TM * PPP * CPO; idx += 512) {
                    const int cc = idx % CPO, rp = idx / CPO;
                    const int pl = rp % PPP, row = rp / PPP;
                    const int phase = pass * PPP + pl;
                    const int orow = (m0 + row) * F + phase - F / 2;
                    if (m0 + row <= L && orow >= 0 && orow < Lout)
                        *(u32x4_t*)(ob + (size_t)orow * COUT + cc * 8) = *(const u32x4_t*)(bufO + row * PO + (pl * COUT + cc * 8) * 2);
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace adf
