// One launch for a whole ResnetBlock1d (unet1d.py:258-317) on the short levels of the net (bf16 throughput mode, 16 or 64
// positions per sample, 256 output channels, input 256 channels or the 256 + 256 skip concat, 8 GroupNorm groups):
//   GroupNorm(+skip scale) -> SiLU -> conv k=3 -> GroupNorm -> FiLM -> SiLU -> conv k=3 -> + residual (identity, or the 1x1
//   conv of the raw concat) and the GroupNorm statistics of the result,
// replacing gn_norm_apply / gn_finalize x 2 + two split-K GEMM launches (45-65 us per block at these levels: the GEMMs run at
// 40-200 TF/s and every launch pays 4-7 us of start-up; 12 of the 29 resblocks of the configs[1] net sit here).
// A GroupNorm needs statistics over the whole sample -- the reason a resblock is otherwise at least two launches -- and here
// the whole sample is in one workgroup.  Same construction as adf_transformer.h: activations in LDS (zero halo rows for the
// 3-tap convs), rounded to bf16 where the unfused path stores or stages a bf16 tensor; each wave owns 32 output columns
// (= one GroupNorm group) of every GEMM and reads its weights once, from a fragment-major copy, 16 KB ahead; helper
// workgroups warm the XCD's L2 with the block's 0.8-1.4 MB of weights.
#pragma once
#include "adf_common.h"
#include <type_traits>

namespace adf {

struct RbFusedArgs {
    const bf16_t* x; const bf16_t* skip;      // [B][N][256] each; skip may be null (then CIN = 256)
    bf16_t* out;                               // [B][N][256]
    GnFinalizeArgs gn1;                        // GroupNorm 1 of [x ; skip_scale * skip] (statistics of the inputs, gamma, beta)
    const float* gamma2; const float* beta2;   // GroupNorm 2
    const float* film; int film_bstride;       // FiLM of GroupNorm 2: film[b*bstride + c] = scale, [.. + 256 + c] = shift
    const float* film2; int film2_bstride;     // optional second addend (class conditioning)
    const void* w1; const void* w2; const void* wr;   // fragment-major bf16 weights [K step][2][256][8]; wr null = identity residual
    const float* b1; const float* b2; const float* br;
    float skip_scale, eps;
    double* stats;                             // [B][8][2] statistics of the output, or nullptr
    int B;
};

typedef __attribute__((ext_vector_type(8))) __bf16 rb_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float rb_f32x16_t;

template <int NTOK, int CIN>
__global__ void __launch_bounds__(512) resblock_small_kernel(const RbFusedArgs a) {
    constexpr int CO = 256;
    constexpr int MR = NTOK < 32 ? 32 : NTOK;
    constexpr int MT = MR / 32;
    constexpr int PX = CIN * 2 + 16, PH = CO * 2 + 16;   // LDS row pitches
    constexpr int RX = MR + 2;                            // rows incl. the two halo rows (row 0 and row NTOK + 1 are zero)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufX = smem;                              // [RX][PX]: raw input (residual conv) -> silu(GN1(input)) -> output rows
    char* const bufH = smem + RX * PX;                    // [RX][PH]: silu(FiLM(GN2(h1)))
    float* const tab = (float*)(bufH + RX * PH);          // [CIN][2] affine of GroupNorm 1
    float* const prm = tab + 2 * CIN;                     // b1 | b2 (+ br) | gamma2 | beta2 | film scale + 1 | film shift: 6 x 256
    const unsigned scratch_ofs = (unsigned)(RX * PX + RX * PH + (2 * CIN + 6 * 256) * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x;

    // ---- helper workgroups: pull the weights towards this XCD's L2 (see adf_transformer.h) ----------------------------
    if (b >= a.B) {
        const int hidx = b - a.B;
        const int per_xcd = (int)(gridDim.x - a.B) / 8 > 0 ? (int)(gridDim.x - a.B) / 8 : 1;
        const int slice = (hidx >> 3) % per_xcd;
        const unsigned scratch = scratch_ofs + (unsigned)wave * 1024u;
        auto warm = [&](const void* W, unsigned total) __attribute__((always_inline)) {
            if (!W) return;
            const unsigned per = ((total / 1024u + per_xcd - 1) / per_xcd) * 1024u;
            const unsigned lo = (unsigned)slice * per, hi = lo + per < total ? lo + per : total;
            for (unsigned off = lo + (unsigned)wave * 1024u; off < hi; off += 8u * 1024u) {
                const char* g = (const char*)W + off + lane * 16;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(scratch) : "memory");
            }
        };
        warm(a.w1, 3u * CIN * CO * 2u);
        warm(a.wr, (unsigned)CIN * CO * 2u);
        warm(a.w2, 3u * CO * CO * 2u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- GEMM stage: acc[i] += sum over taps t and K steps: A[row i*32 + r + t + row0][channels] * W[tap t][columns n0 + r]^T.
    // A from LDS (pitch, halo-shifted rows), W fragments from the fragment-major global copy in packed K order [chunk][tap] --------
    // The ring of weight fragments lives across the stages: `prefetch(W)` issues the first 16 K steps of a GEMM (16 KB per wave)
    // and is called BEFORE the phase that precedes that GEMM (parameter loads, prologue, the previous GEMM's epilogue), so
    // the weight stream -- the floor of this kernel -- keeps running through the vector-only phases.
    constexpr int DEPTH = 15, RING = DEPTH + 1;          // RING even: the parity of the A-fragment double buffer is static
    rb_bf16x8_t wf[RING];
    auto wptr = [&](const void* W, int n0) __attribute__((always_inline)) -> const char* { return (const char*)W + ((size_t)hh * CO + n0 + r) * 16; };
    auto prefetch = [&](const void* W, int n0) __attribute__((always_inline)) {
        const char* const wl = wptr(W, n0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) wf[d] = __builtin_bit_cast(rb_bf16x8_t, *(const u32x4_t*)(wl + (size_t)d * 2 * CO * 16));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gemm = [&](auto tapsc, auto chunksc, const char* A, int pitch, int row0, const void* W, int n0, rb_f32x16_t (&acc)[MT])
                    __attribute__((always_inline)) {
        constexpr int TAPS = decltype(tapsc)::value, CH = decltype(chunksc)::value;     // CH = 64-channel chunks of the input
        constexpr int KS = CH * TAPS * 4;
        static_assert(KS >= DEPTH, "ring depth");
        const char* const wl = wptr(W, n0);
        auto wfrag = [&](int ks) __attribute__((always_inline)) -> rb_bf16x8_t {
            return __builtin_bit_cast(rb_bf16x8_t, *(const u32x4_t*)(wl + (size_t)ks * 2 * CO * 16));
        };
        __builtin_amdgcn_sched_barrier(0);
        // A fragments one K step ahead (their LDS latency would otherwise sit between every load and its MFMAs)
        auto afrag = [&](int ks, rb_bf16x8_t (&af)[MT]) __attribute__((always_inline)) {
            const int ct = ks >> 2, q = ks & 3;           // (chunk, tap) pair in packed order, K step inside the chunk
            const int chunk = ct / TAPS, tap = ct - chunk * TAPS;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *(const rb_bf16x8_t*)(A + (i * 32 + r + tap + row0) * pitch + chunk * 128 + q * 32 + hh * 16);
        };
        rb_bf16x8_t af[2][MT];
        afrag(0, af[0]);
#pragma unroll 1
        for (int kb = 0; kb < KS; kb += RING) {           // RING K steps per trip, so that ring slots are compile-time registers
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
    auto row_of = [&](int i, int e) __attribute__((always_inline)) -> int { return i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh; };
    const std::integral_constant<int, 1> one{};
    const std::integral_constant<int, 3> three{};
    const std::integral_constant<int, CIN / 64> chin{};
    const std::integral_constant<int, CO / 64> chco{};
    const int col = wave * 32 + r;                         // this lane's output column in every GEMM
    const bf16_t* const xb = a.x + (size_t)b * NTOK * 256;
    const bf16_t* const sb = a.skip ? a.skip + (size_t)b * NTOK * 256 : nullptr;

    prefetch(a.wr ? a.wr : a.w1, wave * 32);
    // ---- R0: parameters and the GroupNorm-1 table -> LDS; raw input rows -> bufX (for the 1x1 residual conv) ---------------
    // (every parameter load unconditional and independent -- absent tensors through a dummy pointer, the half of the block that takes which vector
    //  chosen by a wave-uniform select: as a loop with a branch per vector these were five serialised memory round trips, see gn_affine_load)
    {
        const bool lo = wave < 4;                                                     // uniform: threads 0..255 / 256..511
        const int c = tid & 255;
        const bool hbr = a.wr && a.br, hf = a.film != nullptr, hf2 = hf && a.film2 != nullptr;
        const float* const dummy = a.b1;
        int cd = c;                                                                   // opaque index for the dummy reads (see gn_affine_load)
        asm volatile("" : "+v"(cd));
        const float* const p0 = lo ? a.b1 : a.b2;
        const float* const pbr = hbr ? a.br : dummy;
        const float* const p1 = lo ? a.gamma2 : a.beta2;
        const float* const pf = hf ? a.film + (size_t)b * a.film_bstride + (lo ? 0 : 256) : dummy;
        const float* const pf2 = hf2 ? a.film2 + (size_t)b * a.film2_bstride + (lo ? 0 : 256) : dummy;
        float v0 = p0[c], vbr = pbr[cd], v1 = p1[c], vf = pf[cd], vf2 = pf2[cd];
        GnRaw gr = gn_affine_load(a.gn1, b, tid < CIN ? tid : 0);
        // everything is loaded HERE, in one round trip (the compiler otherwise sinks loads towards their uses, behind other loads' waits)
        asm volatile("" : "+v"(v0), "+v"(vbr), "+v"(v1), "+v"(vf), "+v"(vf2), "+v"(gr.gamma), "+v"(gr.beta), "+v"(gr.fs), "+v"(gr.fh), "+v"(gr.sum), "+v"(gr.sq));
        prm[(lo ? 0 : 1) * 256 + c] = v0 + (!lo && hbr ? vbr : 0.f);                  // b1 | b2 (+ br)
        prm[(lo ? 2 : 3) * 256 + c] = v1;                                             // gamma2 | beta2
        const float f = (hf ? vf : 0.f) + (hf2 ? vf2 : 0.f);
        prm[(lo ? 4 : 5) * 256 + c] = lo ? f + 1.0f : f;                              // film scale + 1 | film shift
        if (tid < CIN) {
            float A, Bc;
            gn_affine_finish<true>(a.gn1, tid, gr, A, Bc);
            tab[2 * tid] = A; tab[2 * tid + 1] = Bc;
        }
    }
    constexpr int CPR = CIN / 8;                          // 16-byte chunks per input row
    // The input rows are read ONCE, all pieces of a thread in flight together with the parameter loads above, and kept in registers for both uses
    // (raw rows for the 1x1 residual conv, activated rows for conv1).  As two loops of load -> use -> store per piece they were 2 x NTOK * CPR / 512
    // (16 at 64 tokens x 512 channels) serialised memory round trips.
    constexpr int XT = NTOK * CPR / 512 > 0 ? NTOK * CPR / 512 : 1;
    static_assert(NTOK * CPR % 512 == 0 || NTOK * CPR < 512, "whole sweeps of the workgroup");
    u32x4_t xv[XT];
    {
        const long long dskip = sb ? (const char*)sb - (const char*)xb : 0ll;        // x and skip rows through ONE base pointer (no per-lane pointer select)
#pragma unroll
        for (int k = 0; k < XT; ++k) {
            const int idx = tid + k * 512;
            const int row = idx / CPR, cc = idx % CPR;
            const bool live = idx < NTOK * CPR;
            const long long off = ((long long)(live ? row : 0) * 256 + (cc < 32 ? cc : cc - 32) * 8) * 2 + (cc < 32 ? 0ll : dskip);
            xv[k] = *(const u32x4_t*)((const char*)xb + off);
        }
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
    __syncthreads();

    // ---- R1: the residual into the output accumulators: 1x1 conv of the raw concat, or the input itself -------------------
    rb_f32x16_t accy[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accy[i][e] = 0.f;
    if (a.wr) {
        gemm(one, chin, bufX, PX, 1, a.wr, wave * 32, accy);
        prefetch(a.w1, wave * 32);
        __syncthreads();                                  // bufX is rewritten next
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_of(i, e);
                accy[i][e] = row < NTOK ? bf16_to_f32(xb[(size_t)row * 256 + col].v) : 0.f;
            }
    }
    // ---- R2: silu(GroupNorm1(input)) -> bufX rows 1 .. NTOK, zero halo rows ---------------------------------------------
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
    for (int idx = tid; idx < 2 * (CO / 8); idx += 512) {
        const int row = idx < CO / 8 ? 0 : NTOK + 1, cc = idx % (CO / 8);
        *(u32x4_t*)(bufH + row * PH + cc * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // ---- R3: h1 = conv1 + bias; GroupNorm 2 over the sample (this wave's 32 columns are one group); silu(FiLM(GN2(h1))) -> bufH
    {
        rb_f32x16_t acch[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acch[i][e] = 0.f;
        gemm(three, chin, bufX, PX, 0, a.w1, wave * 32, acch);
        prefetch(a.w2, wave * 32);
        const float bias = prm[col];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = bf16_stored(acch[i][e] + bias);        // h1 as it is stored / read back: statistics of the stored values
                acch[i][e] = v;
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
        float A = rs * prm[512 + col];
        float Bc = prm[768 + col] - (float)mean * A;
        if (a.film) { const float fs = prm[1024 + col], fh = prm[1280 + col]; A *= fs; Bc = fmaf(Bc, fs, fh); }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_of(i, e);
                const float h1 = bf16_to_f32(f32_to_bf16_hw(acch[i][e]));          // as the unfused path stores it
                const unsigned short q = row < NTOK ? f32_to_bf16_hw(silu_f(fmaf(h1, A, Bc))) : (unsigned short)0;
                *(unsigned short*)(bufH + (row + 1) * PH + col * 2) = q;        // rows >= NTOK: zeros (row NTOK is the upper halo row)
            }
    }
    __syncthreads();

    // ---- R4: y = conv2 + residual + biases -> statistics, bf16 rows -> bufX -> global ------------------------------------
    gemm(three, chco, bufH, PH, 0, a.w2, wave * 32, accy);
    {
        const float bias = prm[256 + col];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_of(i, e);
                const unsigned short qv = f32_to_bf16_hw(accy[i][e] + bias);
                const float v = bf16_to_f32(qv);
                if (row < NTOK) { s1 += v; s2 = fmaf(v, v, s2); }
                *(unsigned short*)(bufX + (row + 1) * PX + col * 2) = qv;
            }
        if (a.stats) {
            double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
            if (lane == 0) {
                double* sp = a.stats + ((size_t)b * 8 + wave) * 2;
                sp[0] = d1; sp[1] = d2;
            }
        }
    }
    __syncthreads();
    {
        bf16_t* const ob = a.out + (size_t)b * NTOK * CO;
        for (int idx = tid; idx < NTOK * (CO / 8); idx += 512) {
            const int row = idx / (CO / 8), cc = idx % (CO / 8);
            *(u32x4_t*)(ob + (size_t)row * CO + cc * 8) = *(const u32x4_t*)(bufX + (row + 1) * PX + cc * 16);
        }
    }
}

inline size_t resblock_small_lds(int ntok, int cin) {
    const int mr = ntok < 32 ? 32 : ntok;
    return (size_t)(mr + 2) * (cin * 2 + 16) + (size_t)(mr + 2) * (256 * 2 + 16) + (size_t)(2 * cin + 6 * 256) * 4 + 8 * 1024;
}

inline const char* launch_resblock_small(const RbFusedArgs& a, int B, int ntok, int cin, hipStream_t s) {
    if ((ntok != 64 && ntok != 16) || (cin != 256 && cin != 512)) return "resblock_small: unsupported shape";
    static bool attr_done[kMaxDevices] = {};
    bool& attr = attr_done[current_device()];
    if (!attr) {
        if (hipFuncSetAttribute((const void*)resblock_small_kernel<64, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)resblock_small_kernel<64, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)resblock_small_kernel<16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)resblock_small_kernel<16, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "resblock_small: hipFuncSetAttribute failed";
        attr = true;
    }
    const size_t lds = resblock_small_lds(ntok, cin);
    if (lds > 160 * 1024) return "resblock_small: LDS budget exceeded";
    const int helpers = B < 256 ? ((256 - B) / 8 > 3 * B / 8 ? 3 * B / 8 : (256 - B) / 8) * 8 : 0;
    RbFusedArgs aa = a;
    aa.B = B;
    const dim3 grid(B + helpers), blk(512);
    if (ntok == 64 && cin == 256) hipLaunchKernelGGL((resblock_small_kernel<64, 256>), grid, blk, lds, s, aa);
    else if (ntok == 64) hipLaunchKernelGGL((resblock_small_kernel<64, 512>), grid, blk, lds, s, aa);
    else if (cin == 256) hipLaunchKernelGGL((resblock_small_kernel<16, 256>), grid, blk, lds, s, aa);
    else hipLaunchKernelGGL((resblock_small_kernel<16, 512>), grid, blk, lds, s, aa);
    return hipGetLastError() == hipSuccess ? nullptr : "resblock_small: launch failed";
}

}  // namespace adf
