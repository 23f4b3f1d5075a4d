// A short-level ResnetBlock1d (unet1d.py:258-317; 16 or 64 positions, 256 output channels, bf16) as TWO launches of four workgroups per
// sample, for batches that leave CUs idle with adf_resblock_small.h's one workgroup per sample (64 samples on 256 CUs).
//
// Why: the one-launch kernel is bound by the weight stream of its workgroup -- every workgroup reads ALL 0.8-1.4 MB of the block's
// weights from L2 at ~34 B/clk for 16 or 64 positions (18-39 us per block; 12 blocks = 0.25 ms of a 2.6 ms evaluation on 64 of 256 CUs).
// A workgroup that owns 64 of the 256 output columns reads a quarter of them.  conv1 and GroupNorm 2 split that way for free: a
// GroupNorm group is 32 consecutive columns, so the statistics of a column quarter are complete inside its workgroup.  conv2 needs ALL
// columns of silu(FiLM(GN2(h1))) as its K dimension: that is the one exchange, and it is the launch boundary --
//   phase 1 (B x 4 workgroups): silu(GN1([x ; skip])) -> conv k=3 for 64 columns -> GroupNorm 2 + FiLM + SiLU -> hact[b][row][64 cols]
//   phase 2 (B x 4 workgroups): conv k=3 over hact (+ the 1x1 residual conv of the raw concat, or the identity) for 64 columns ->
//                               statistics of the stored result, out[b][row][64 cols]
// The arithmetic and the bf16 rounding points are those of adf_resblock_small.h (h1 rounded where the unfused path stores it, the
// statistics from the rounded values, the activated operand rounded once); only the K summation is split over four waves.
// Inside a workgroup: wave = (column tile of 32, K quarter); a wave's weight fragments (12-24 K steps x 16 B per lane) are ALL
// requested at kernel entry, ahead of the parameter loads and the prologue; partial sums meet in LDS.
#pragma once
#include "adf_resblock_small.h"

namespace adf {

struct RbSplitArgs {
    RbFusedArgs f;
    bf16_t* hact;             // [B][N][256]: silu(FiLM(GN2(h1))) as conv2 reads it
};

// the weight fragments of one wave for one phase, requested by rb_split_load_w ahead of everything else (round 4 split the kernel body into load / run so that
// tools/experiments/adf_resblock_chain.h -- several blocks in one launch, measured slower -- could request them ahead of its barriers)
template <int NTOK, int CIN, int PHASE>
struct RbSplitWf {
    static constexpr int CO = 256;
    static constexpr int KS_MAIN = (PHASE == 1 ? CIN / 64 : CO / 64) * 3 * 4;   // K steps of the 3-tap conv
    static constexpr int KS_RES = CIN / 16;                                     // K steps of the 1x1 residual conv (phase 2, when there is one)
    static constexpr int KM = KS_MAIN / 4, KR = KS_RES / 4;                     // per wave
    rb_bf16x8_t wf[KM], wr_[PHASE == 2 ? KR : 1];
};
template <int NTOK, int CIN, int PHASE>
__device__ __forceinline__ void rb_split_load_w(const RbSplitArgs& aa, int bq, RbSplitWf<NTOK, CIN, PHASE>& W) {
    const RbFusedArgs& a = aa.f;
    constexpr int CO = 256, KM = RbSplitWf<NTOK, CIN, PHASE>::KM, KR = RbSplitWf<NTOK, CIN, PHASE>::KR;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5, ct = wave & 1, kq = wave >> 1, q = bq & 3;
    const int col = q * 64 + ct * 32 + r;
    const void* const Wm = PHASE == 1 ? a.w1 : a.w2;
    auto wptr = [&](const void* Wp) __attribute__((always_inline)) -> const char* { return (const char*)Wp + ((size_t)hh * CO + col) * 16; };
    const char* const wl = wptr(Wm);
#pragma unroll
    for (int k = 0; k < KM; ++k) W.wf[k] = __builtin_bit_cast(rb_bf16x8_t, *(const u32x4_t*)(wl + (size_t)(kq * KM + k) * 2 * CO * 16));
    if constexpr (PHASE == 2) {
        const char* const wl2 = wptr(a.wr ? a.wr : a.w2);            // (no residual conv: a valid address, the values are not used)
#pragma unroll
        for (int k = 0; k < KR; ++k) W.wr_[k] = __builtin_bit_cast(rb_bf16x8_t, *(const u32x4_t*)(wl2 + (size_t)(kq * KR + k) * 2 * CO * 16));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// one phase of one block for workgroup `bq` (= sample * 4 + column quarter); smem: resblock_split_lds(NTOK, CIN) bytes
template <int NTOK, int CIN, int PHASE>
__device__ __forceinline__ void rb_split_run(const RbSplitArgs& aa, char* smem, int bq, const RbSplitWf<NTOK, CIN, PHASE>& Wreg) {
    const RbFusedArgs& a = aa.f;
    constexpr int CO = 256;
    constexpr int MR = NTOK < 32 ? 32 : NTOK;
    constexpr int MT = MR / 32;
    constexpr int PX = CIN * 2 + 16, PH = CO * 2 + 16;   // LDS row pitches
    constexpr int RX = MR + 2;                            // rows incl. the two halo rows (row 0 and row NTOK + 1 are zero)
    char* const bufX = smem;                              // [RX][PX]: phase 1: silu(GN1(input)); phase 2: raw concat (1x1 residual conv); then partial sums
    char* const bufH = smem + RX * PX;                    // [RX][PH]: phase 2: hact rows
    float* const tab = (float*)(bufH + RX * PH);          // [CIN][2] affine of GroupNorm 1 (phase 1)
    float* const prm = tab + 2 * CIN;                     // per column of this quarter: bias | gamma2 | beta2 | film scale + 1 | film shift: 5 x 64
    char* const otile = (char*)(prm + 5 * 64);            // [NTOK][64] bf16 (8 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int ct = wave & 1, kq = wave >> 1;              // column tile inside the quarter, K quarter
    const int b = bq >> 2, q = bq & 3;
    const int n0 = q * 64 + ct * 32;                      // first output column of this wave
    const int col = n0 + r;
    // ---- this wave's weight fragments: every K step of its quarter, requested before anything else (rb_split_load_w) ----------------
    constexpr int KM = RbSplitWf<NTOK, CIN, PHASE>::KM, KR = RbSplitWf<NTOK, CIN, PHASE>::KR;
    const rb_bf16x8_t (&wf)[KM] = Wreg.wf;
    const rb_bf16x8_t (&wr_)[PHASE == 2 ? KR : 1] = Wreg.wr_;
    auto row_of = [&](int i, int e) __attribute__((always_inline)) -> int { return i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh; };
    const bf16_t* const xb = a.x + (size_t)b * NTOK * 256;
    const bf16_t* const sb = a.skip ? a.skip + (size_t)b * NTOK * 256 : nullptr;
    constexpr int CPR = CIN / 8;                          // 16-byte chunks per input row
    constexpr int XT = NTOK * CPR / 512 > 0 ? NTOK * CPR / 512 : 1;

    // ---- parameters of this quarter's 64 columns (every load unconditional, absent tensors through a dummy pointer) ----------------
    {
        const int c = tid & 63, which = tid >> 6;                                     // which = 0 .. 7: one parameter vector per wave
        const bool hbr = a.wr && a.br, hf = a.film != nullptr, hf2 = hf && a.film2 != nullptr;
        const float* const dummy = a.b1;
        const int cg = q * 64 + c;
        int cd = cg;
        asm volatile("" : "+v"(cd));
        // vectors: 0 bias (phase 1: b1; phase 2: b2), 1 br (phase 2), 2 gamma2, 3 beta2, 4 film scale, 5 film shift, 6 film2 scale, 7 film2 shift
        const float* p = dummy;
        int idx = cd;
        if (which == 0) { p = PHASE == 1 ? a.b1 : a.b2; }
        else if (which == 1) { p = hbr ? a.br : dummy; }
        else if (which == 2) { p = a.gamma2; }
        else if (which == 3) { p = a.beta2; }
        else if (which == 4) { p = hf ? a.film + (size_t)b * a.film_bstride : dummy; }
        else if (which == 5) { p = hf ? a.film + (size_t)b * a.film_bstride : dummy; idx = hf ? cd + 256 : cd; }
        else if (which == 6) { p = hf2 ? a.film2 + (size_t)b * a.film2_bstride : dummy; }
        else { p = hf2 ? a.film2 + (size_t)b * a.film2_bstride : dummy; idx = hf2 ? cd + 256 : cd; }
        float v = p[idx];
        // staging: [8][64] floats in otile (free until the epilogue), combined below
        ((float*)otile)[which * 64 + c] = v;
    }
    GnRaw gr = {};
    if constexpr (PHASE == 1) gr = gn_affine_load(a.gn1, b, tid < CIN ? tid : 0);
    // ---- the rows this phase multiplies: all pieces of a thread in flight together ------------------------------------------------
    u32x4_t xv[PHASE == 1 ? XT : XT];
    constexpr int HT = NTOK * (CO / 8) / 512 > 0 ? NTOK * (CO / 8) / 512 : 1;
    u32x4_t hv[PHASE == 2 ? HT : 1];
    const bool need_x = PHASE == 1 || a.wr != nullptr;                            // uniform
    if (need_x) {
        const long long dskip = sb ? (const char*)sb - (const char*)xb : 0ll;
#pragma unroll
        for (int k = 0; k < XT; ++k) {
            const int idx = tid + k * 512;
            const int row = idx / CPR, cc = idx % CPR;
            const bool live = idx < NTOK * CPR;
            const long long off = ((long long)(live ? row : 0) * 256 + (cc < 32 ? cc : cc - 32) * 8) * 2 + (cc < 32 ? 0ll : dskip);
            xv[k] = *(const u32x4_t*)((const char*)xb + off);
        }
    }
    if constexpr (PHASE == 2) {
        const bf16_t* const hb = aa.hact + (size_t)b * NTOK * CO;
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const int idx = tid + k * 512;
            const bool live = idx < NTOK * (CO / 8);
            hv[k] = *(const u32x4_t*)(hb + (size_t)(live ? idx : 0) * 8);
        }
    }
    __syncthreads();                                      // the staged parameter vectors are in LDS
    if (tid < 64) {
        const float* const st = (const float*)otile;
        const bool hbr = a.wr && a.br, hf = a.film != nullptr, hf2 = hf && a.film2 != nullptr;
        prm[tid] = st[tid] + (PHASE == 2 && hbr ? st[64 + tid] : 0.f);
        prm[64 + tid] = st[128 + tid];
        prm[128 + tid] = st[192 + tid];
        prm[192 + tid] = (hf ? st[256 + tid] : 0.f) + (hf2 ? st[384 + tid] : 0.f) + 1.0f;
        prm[256 + tid] = (hf ? st[320 + tid] : 0.f) + (hf2 ? st[448 + tid] : 0.f);
    }
    if constexpr (PHASE == 1) {
        if (tid < CIN) {
            float A, Bc;
            gn_affine_finish<true>(a.gn1, tid, gr, A, Bc);
            tab[2 * tid] = A; tab[2 * tid + 1] = Bc;
        }
    }
    __syncthreads();

    // ---- operand rows -> LDS ------------------------------------------------------------------------------------------------------
    if constexpr (PHASE == 1) {
#pragma unroll
        for (int k = 0; k < XT; ++k) {
            const int idx = tid + k * 512;
            if (idx >= NTOK * CPR) continue;
            const int row = idx / CPR, cc = idx % CPR;
            float f[8];
            unpack16<bf16_t>(xv[k], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = silu_f(fmaf(f[e], tab[2 * (cc * 8 + e)], tab[2 * (cc * 8 + e) + 1]));
            *(u32x4_t*)(bufX + (row + 1) * PX + cc * 16) = pack16<bf16_t>(f);
        }
        for (int idx = tid; idx < 2 * CPR; idx += 512) {
            const int row = idx < CPR ? 0 : NTOK + 1, cc = idx % CPR;
            *(u32x4_t*)(bufX + row * PX + cc * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
    } else {
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const int idx = tid + k * 512;
            if (idx >= NTOK * (CO / 8)) continue;
            const int row = idx / (CO / 8), cc = idx % (CO / 8);
            *(u32x4_t*)(bufH + (row + 1) * PH + cc * 16) = hv[k];
        }
        for (int idx = tid; idx < 2 * (CO / 8); idx += 512) {
            const int row = idx < CO / 8 ? 0 : NTOK + 1, cc = idx % (CO / 8);
            *(u32x4_t*)(bufH + row * PH + cc * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
        if (a.wr) {
#pragma unroll
            for (int k = 0; k < XT; ++k) {
                const int idx = tid + k * 512;
                if (idx >= NTOK * CPR) continue;
                const int row = idx / CPR, cc = idx % CPR;
                u32x4_t v = xv[k];
                if (cc >= 32 && a.skip_scale != 1.0f) {        // the residual conv reads the raw concat [x ; skip_scale * skip]
                    float f[8];
                    unpack16<bf16_t>(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= a.skip_scale;
                    v = pack16<bf16_t>(f);
                }
                *(u32x4_t*)(bufX + (row + 1) * PX + cc * 16) = v;
            }
        }
    }
    __syncthreads();

    // ---- this wave's share of the K sum -----------------------------------------------------------------------------------------------
    rb_f32x16_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    {
        const char* const A = PHASE == 1 ? bufX : bufH;
        constexpr int pitch = PHASE == 1 ? PX : PH;
        auto afrag = [&](int ks, rb_bf16x8_t (&af)[MT]) __attribute__((always_inline)) {
            const int cti = ks >> 2, qq = ks & 3;         // (chunk, tap) pair in packed order, K step inside the chunk
            const int chunk = cti / 3, tap = cti - chunk * 3;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *(const rb_bf16x8_t*)(A + (i * 32 + r + tap) * pitch + chunk * 128 + qq * 32 + hh * 16);
        };
        rb_bf16x8_t af[2][MT];
        afrag(kq * KM, af[0]);
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k + 1 < KM) afrag(kq * KM + k + 1, af[(k + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k & 1][i], wf[k], acc[i], 0, 0, 0);
        }
        if constexpr (PHASE == 2) {
            if (a.wr) {
                auto xfrag = [&](int ks, rb_bf16x8_t (&xf)[MT]) __attribute__((always_inline)) {
                    const int chunk = ks >> 2, qq = ks & 3;
#pragma unroll
                    for (int i = 0; i < MT; ++i) xf[i] = *(const rb_bf16x8_t*)(bufX + (i * 32 + r + 1) * PX + chunk * 128 + qq * 32 + hh * 16);
                };
                rb_bf16x8_t xf[2][MT];
                xfrag(kq * KR, xf[0]);
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    if (k + 1 < KR) xfrag(kq * KR + k + 1, xf[(k + 1) & 1]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[k & 1][i], wr_[k], acc[i], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                      // every wave has read its operand rows: bufX holds the partial sums from here on
    float* const red = (float*)bufX;                      // [3 K quarters][2 column tiles][MT][16][64 lanes]
    if (kq > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((((kq - 1) * 2 + ct) * MT + i) * 16 + e) * 64 + lane] = acc[i][e];
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] += red[(((k * 2 + ct) * MT + i) * 16 + e) * 64 + lane];
        const int lc = ct * 32 + r;                       // column inside the quarter
        const float bias = prm[lc];
        if constexpr (PHASE == 1) {
            // h1 as it is stored / read back; GroupNorm 2 over the sample (this wave's 32 columns are one group); silu(FiLM(GN2(h1)))
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = bf16_stored(acc[i][e] + bias);
                    acc[i][e] = v;
                    if (row_of(i, e) < NTOK) { s1 += v; s2 = fmaf(v, v, s2); }
                }
            double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
            const double inv_cnt = 1.0 / (double)(NTOK * 32);
            const double mean = d1 * inv_cnt;
            double var = d2 * inv_cnt - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float vf = (float)var + a.eps;
            float rs = __builtin_amdgcn_rsqf(vf);
            rs = rs * (1.5f - 0.5f * vf * rs * rs);
            float A = rs * prm[64 + lc];
            float Bc = prm[128 + lc] - (float)mean * A;
            if (a.film) { const float fs = prm[192 + lc], fh = prm[256 + lc]; A *= fs; Bc = fmaf(Bc, fs, fh); }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row_of(i, e);
                    const float h1 = bf16_to_f32(f32_to_bf16_hw(acc[i][e]));
                    if (row < NTOK) *(unsigned short*)(otile + row * 128 + lc * 2) = f32_to_bf16_hw(silu_f(fmaf(h1, A, Bc)));
                }
        } else {
            // y = conv2 (+ 1x1 residual conv) + biases (+ identity residual) -> statistics of the stored values, bf16 rows
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row_of(i, e);
                    float y = acc[i][e];
                    if (!a.wr && row < NTOK) y += bf16_to_f32(xb[(size_t)row * 256 + col].v);
                    const unsigned short qv = f32_to_bf16_hw(y + bias);
                    const float v = bf16_to_f32(qv);
                    if (row < NTOK) {
                        s1 += v; s2 = fmaf(v, v, s2);
                        *(unsigned short*)(otile + row * 128 + lc * 2) = qv;
                    }
                }
            if (a.stats) {
                double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
                if (lane == 0) {
                    double* sp = a.stats + ((size_t)b * 8 + q * 2 + ct) * 2;
                    sp[0] = d1; sp[1] = d2;
                }
            }
        }
    }
    __syncthreads();
    {
        bf16_t* const ob = (PHASE == 1 ? aa.hact : a.out) + (size_t)b * NTOK * CO + q * 64;
        for (int idx = tid; idx < NTOK * 8; idx += 512) {
            const int row = idx >> 3, cc = idx & 7;
            *(u32x4_t*)(ob + (size_t)row * CO + cc * 8) = *(const u32x4_t*)(otile + row * 128 + cc * 16);
        }
    }
}

template <int NTOK, int CIN, int PHASE>
__global__ void __launch_bounds__(512) resblock_split_kernel(const RbSplitArgs aa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RbSplitWf<NTOK, CIN, PHASE> W;
    rb_split_load_w<NTOK, CIN, PHASE>(aa, (int)blockIdx.x, W);
    rb_split_run<NTOK, CIN, PHASE>(aa, smem, (int)blockIdx.x, W);
}

inline size_t resblock_split_lds(int ntok, int cin) {
    const int mr = ntok < 32 ? 32 : ntok;
    return (size_t)(mr + 2) * (cin * 2 + 16) + (size_t)(mr + 2) * (256 * 2 + 16) + (size_t)(2 * cin + 5 * 64) * 4 + 8 * 1024;
}

template <int NTOK, int CIN>
inline const char* launch_resblock_split_t(const RbSplitArgs& a, int B, hipStream_t s) {
    static bool attr_done[kMaxDevices] = {};
    bool& attr = attr_done[current_device()];
    if (!attr) {
        if (hipFuncSetAttribute((const void*)resblock_split_kernel<NTOK, CIN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)resblock_split_kernel<NTOK, CIN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "resblock_split: hipFuncSetAttribute failed";
        attr = true;
    }
    const size_t lds = resblock_split_lds(NTOK, CIN);
    if (lds > 160 * 1024) return "resblock_split: LDS budget exceeded";
    hipLaunchKernelGGL((resblock_split_kernel<NTOK, CIN, 1>), dim3(B * 4), dim3(512), lds, s, a);
    hipLaunchKernelGGL((resblock_split_kernel<NTOK, CIN, 2>), dim3(B * 4), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? nullptr : "resblock_split: launch failed";
}

inline const char* launch_resblock_split(const RbSplitArgs& a, int B, int ntok, int cin, hipStream_t s) {
    if ((ntok != 64 && ntok != 16) || (cin != 256 && cin != 512)) return "resblock_split: unsupported shape";
    RbSplitArgs aa = a;
    aa.f.B = B;
    if (ntok == 64 && cin == 256) return launch_resblock_split_t<64, 256>(aa, B, s);
    if (ntok == 64) return launch_resblock_split_t<64, 512>(aa, B, s);
    if (cin == 256) return launch_resblock_split_t<16, 256>(aa, B, s);
    return launch_resblock_split_t<16, 512>(aa, B, s);
}

}  // namespace adf
