// Software-pipelined persistent implicit-GEMM kernel for the large stride-1 layers (bf16 throughput mode).
//
// Why it exists (profiles/README.md, "knock-out sweep"): in the plain kernel of adf_gemm.h the phases of a K step
// (global loads -> prologue math -> LDS stores -> barrier -> MFMA -> barrier) add up serially and the weights of a
// 128-row tile are 3/4 of the staged bytes.  Here
//   * the block tile is 256 x 128 (8 waves, 4 x 2, wave tile 64 x 64): weight staging per output row halves;
//   * a K step is 32 channels x all taps of the segment;
//   * everything that is in flight is in flight towards LDS: activations, weights and the affine table are brought
//     in with LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, M0 = LDS base) TWO steps ahead into a
//     ring of three stages, and completion is hand-counted (s_waitcnt vmcnt(N): the DMA returns in order).  No
//     register ever holds data that has not arrived -- with asm loads into compiler-allocated registers the register
//     allocator copies the "value" (v_mov) between the load and the wait once pressure is high, i.e. copies stale data;
//   * LDS rows are 64 B with no padding (DMA writes a wave's 64 x 16 B contiguously); the 16-byte chunk c of row r
//     sits at slot c ^ ((r >> 2) & 3), applied on the SOURCE side of the DMA, which makes the ds_read_b128 fragment
//     reads conflict-free;
//   * the fused GroupNorm/FiLM/SiLU prologue of step s+1 runs IN PLACE on the chunks the wave itself fetched (so it
//     needs only the wave's own vmcnt wait), while the MFMAs of step s read stage s % 3; ONE workgroup barrier per
//     step; raw segments (the 1x1 residual) skip the prologue entirely: their bytes go HBM -> LDS -> MFMA untouched;
//   * the two waves that share a SIMD (w, w+4) run the two halves of a step in opposite order (prologue->MFMA vs
//     MFMA->prologue), so one wave's VALU/LDS work sits beside the other's matrix work;
//   * blocks are persistent (one per CU) and the DMA stream runs across tile boundaries: the epilogue of tile t
//     (wave-local, through a 2 KB scratch per wave) executes while the first steps of tile t+1 are in flight; the
//     affine table is double-buffered per tile and fetched three steps ahead; the bias vector sits in LDS.
// Shapes (checked by the launcher): stride 1, step +1, taps 3 (pad 1) or 1, channels multiple of 32 and sources
// split at a multiple of 32, mrows = lin = out_rows a multiple of 256, n = n_pad = out_c a multiple of 128,
// tile counts powers of two, at least 4 K steps, no phase scatter / GELU / identity residual.
#pragma once
#include "adf_gemm.h"

namespace adf {

constexpr int kPpTM = 256, kPpTN = 128;
constexpr int kPpRow = 64;                                  // bytes of K per staged row (32 bf16)
constexpr int kPpAStage = 17 * 1024;                        // 16 pieces of 16 rows + the halo piece (rows 256, 257)
constexpr int kPpWStage = 3 * kPpTN * kPpRow;               // 24,576 B: 3 taps x 128 rows
constexpr int kPpStages = 3;
constexpr int kPpScratch = 8 * 2048;                        // 8 waves x (8 rows x 64 cols x fp32)
constexpr int kPpTab = 8192;                                // (a, b) of up to 1024 channels, per slot
constexpr int kPpBias = 2048;                               // up to 512 output columns
constexpr int kPpOffW = kPpStages * kPpAStage;              //  52,224
constexpr int kPpOffScr = kPpOffW + kPpStages * kPpWStage;  // 125,952
constexpr int kPpOffTab = kPpOffScr + kPpScratch;           // 142,336
constexpr int kPpOffBias = kPpOffTab + 2 * kPpTab;          // 158,720
constexpr int kPpLds = kPpOffBias + kPpBias;                // 160,768 B
constexpr int kPpMaxCin = kPpTab / 8;
constexpr int kPpMaxN = kPpBias / 4;

// One LDS-DMA piece: the active lanes copy 16 B each from `gsrc` to LDS byte address lds_dst + 16 * lane.
// (Dynamic LDS starts at byte 0: the kernel has no static __shared__.)
__device__ __forceinline__ void pp_dma16(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MFMAs of one K step: TAPS x 2 sub-steps of 16 channels; fragments of sub-step st+1 are read while st issues.
// aoff[tap] / woff: LDS byte address of this lane's fragment chunk for sub-step parity 0 (parity 1 = address ^ 32).
template <int TAPS>
__device__ __forceinline__ void pp_mfma_step(f32x16_t (&acc)[2][2], const char* lds, const unsigned (&aoff)[3], unsigned woff) {
    constexpr int NS = TAPS * 2;
    bf16x8_t fa[2][2], fb[2][2];
    auto rdA = [&](int st, int i) { return *(const bf16x8_t*)(lds + ((st & 1) ? (aoff[st >> 1] ^ 32u) : aoff[st >> 1]) + i * 32 * kPpRow); };
    auto rdW = [&](int st, int j) { return *(const bf16x8_t*)(lds + ((st & 1) ? (woff ^ 32u) : woff) + ((st >> 1) * kPpTN + j * 32) * kPpRow); };
#pragma unroll
    for (int i = 0; i < 2; ++i) { fa[0][i] = rdA(0, i); fb[0][i] = rdW(0, i); }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        if (st + 1 < NS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[nxt][i] = rdA(st + 1, i); fb[nxt][i] = rdW(st + 1, i); }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
    }
}

__global__ void __launch_bounds__(512) conv_gemm_pp_kernel(const GemmArgs a, int tiles_total, int tm_shift, int tn_shift) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsScr = smem + kPpOffScr;
    char* const ldsTab = smem + kPpOffTab;
    float* const ldsBias = (float*)(smem + kPpOffBias);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool early = (wave & 4) == 0;            // waves w and w+4 share a SIMD: opposite phase order
    const int slot = lane & 3;                      // physical 16-byte slot of this lane inside its staged row
    const int lrow = lane >> 2;                     // row inside a 16-row DMA piece
    const int chunk = slot ^ ((lane >> 4) & 3);     // logical chunk stored at that slot (row >> 2 == lane >> 4 mod 4)
    const int srow = wave * 16 + lrow;              // staged row of this lane in unit 0 (+128 in unit 1, 256 + lrow halo)
    const unsigned lane_lds = (unsigned)lane * 16u; // byte position of the lane inside a piece

    const int nblk = (int)gridDim.x, bidx = (int)blockIdx.x;
    const int t_lo = (int)((long long)bidx * tiles_total / nblk);
    const int t_hi = (int)((long long)(bidx + 1) * tiles_total / nblk);
    const int nst0 = (a.seg[0].c0 + a.seg[0].c1) >> 5;
    const int nst1 = a.nseg > 1 ? (a.seg[1].c0 + a.seg[1].c1) >> 5 : 0;
    const int nsteps = nst0 + nst1;
    const int Q = (t_hi - t_lo) * nsteps;
    if (Q <= 0) return;
    const int ctot0 = a.seg[0].c0 + a.seg[0].c1;
    const bool use_tab = a.seg[0].ab != nullptr;

    auto geom = [&](int tseq, int& b0, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_lo + tseq;
        const int tml = t >> tn_shift;
        n0 = (t & ((1 << tn_shift) - 1)) * kPpTN;
        b0 = tml >> tm_shift;
        m0 = (tml & ((1 << tm_shift) - 1)) * kPpTM;
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- block prologue (ordinary loads; the DMA pipeline starts after it) ---------------------------------
    for (int n = tid; n < a.n_pad; n += 512) {
        float bv = 0.f;
        if (n < a.n) {
            const int bi = n % a.bias_mod;
            if (a.bias0) bv += a.bias0[bi];
            if (a.bias1) bv += a.bias1[bi];
        }
        ldsBias[n] = bv;
    }
    if (use_tab) {
        int b0, m0, n0;
        geom(0, b0, m0, n0);
        if (tid * 2 < ctot0) *(f32x4_t*)(ldsTab + tid * 16) = *(const f32x4_t*)(a.seg[0].ab + ((size_t)b0 * ctot0 + tid * 2) * 2);
    }
    __syncthreads();

    // ---- stream state (all wave-uniform) ---------------------------------------------------------------------
    int is_tseq = 0, is_step = 0, is_b0, is_m0, is_n0;      // DMA stream (two steps ahead)
    geom(0, is_b0, is_m0, is_n0);
    int tr_tseq = 0, tr_step = 0, tr_b0, tr_m0, tr_n0;      // prologue stream (one step ahead)
    geom(0, tr_b0, tr_m0, tr_n0);
    int mm_tseq = 0, mm_step = 0;                            // MFMA stream
    int st_is = 0, st_tr = 0, st_mm = 0;                     // ring stage of each stream

    // ---- DMA of one K step into ring stage st_is: 2 activation pieces per wave (+ the halo piece in wave 0 of a
    // 3-tap segment) and one weight piece per tap.  Returns the number of DMA instructions this wave issued.
    auto issue = [&]() __attribute__((always_inline)) -> int {
        const bool s1 = is_step >= nst0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int k32 = s1 ? is_step - nst0 : is_step;
        const bool halo = wave == 0 && sg.taps == 3;
        const int cbase = k32 * 32;
        const bool from1 = sg.c1 > 0 && cbase >= sg.c0;                    // uniform: sources split at a multiple of 32
        const char* src = from1 ? uniform_ptr(sg.src1) : uniform_ptr(sg.src0);
        const unsigned rowbytes = (unsigned)(from1 ? sg.c1 : sg.c0) * 2u;
        const unsigned colbytes = (unsigned)(cbase - (from1 ? sg.c0 : 0)) * 2u + (unsigned)chunk * 16u;
        const int p0 = is_m0 + sg.off0 + srow;                              // input position of this lane's row, unit 0
        const unsigned rowbase = (unsigned)(is_b0 * a.lin);
        const unsigned ldsA = (unsigned)(st_is * kPpAStage) + (unsigned)wave * 1024u;
        int n = 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = p0 + i * 128;
            const bool ok = p >= 0 && p < a.lin;
            unsigned off = ok ? (rowbase + (unsigned)p) * rowbytes + colbytes : 0u;   // padding rows: any address, zeroed later
            if (a.dbg & 16) {   // timing experiment: same bytes, but the 256 x 64 B slab of a step is one contiguous 16 KB block
                const unsigned tile = (rowbase + (unsigned)is_m0) >> 8, cs = (unsigned)(from1 ? sg.c1 : sg.c0) >> 5;
                const unsigned ks = (unsigned)(cbase - (from1 ? sg.c0 : 0)) >> 5;
                off = ((tile * cs + ks) * 256u + (unsigned)(srow + i * 128)) * 64u + (unsigned)chunk * 16u;
            }
            pp_dma16(src + off, ldsA + (unsigned)i * 8192u);
        }
        if (halo) {
            const int p = is_m0 + sg.off0 + 256 + lrow;
            const bool ok = p >= 0 && p < a.lin;
            const unsigned off = ok ? (rowbase + (unsigned)p) * rowbytes + colbytes : 0u;
            if (lane < 8) pp_dma16(src + off, (unsigned)(st_is * kPpAStage) + 16u * 1024u);
            n = 3;
        }
        // weights: wave w copies rows 16w .. 16w+15 of every tap slab (128 rows x 64 B of this K half)
        const unsigned slab = (unsigned)a.n_pad * (unsigned)kRowBytes;
        const char* wp = uniform_ptr(sg.w) + (size_t)((k32 >> 1) * sg.taps) * slab + (size_t)is_n0 * kRowBytes + (k32 & 1) * 64;
        const char* wl = wp + (unsigned)srow * (unsigned)kRowBytes + (unsigned)chunk * 16u;
        const unsigned ldsW = (unsigned)(kPpOffW + st_is * kPpWStage) + (unsigned)wave * 1024u;
        pp_dma16(wl, ldsW);
        n += 1;
        if (sg.taps == 3) {
            pp_dma16(wl + slab, ldsW + 8192u);
            pp_dma16(wl + slab + slab, ldsW + 16384u);
            n += 2;
        }
        if (++is_step == nsteps) { is_step = 0; ++is_tseq; geom(is_tseq, is_b0, is_m0, is_n0); }
        st_is = st_is == kPpStages - 1 ? 0 : st_is + 1;
        return n;
    };
    // wait until at most `n` of this wave's DMA instructions (all issued after the batch about to be used) are in flight
    auto wait_loads = [&](int n) __attribute__((always_inline)) {
        if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- fused prologue, in place on the chunks this wave fetched (stage st_tr); zero padding for rows outside
    // the sample.  Raw segments only need the zero fill (tiles at a sample edge of a 3-tap segment).
    auto transform = [&]() __attribute__((always_inline)) {
        const bool s1 = tr_step >= nst0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int k32 = s1 ? tr_step - nst0 : tr_step;
        const bool with_tab = !s1 && use_tab;
        const bool act = sg.act != 0 && !(a.dbg & 2);
        const bool halo = wave == 0 && sg.taps == 3;
        char* const ldsA = smem + st_tr * kPpAStage + wave * 1024 + lane_lds;
        char* const ldsH = smem + st_tr * kPpAStage + 16 * 1024 + lane_lds;
        const int p0 = tr_m0 + sg.off0 + srow;
        const int ph = tr_m0 + sg.off0 + 256 + lrow;
        const bool from1 = sg.c1 > 0 && k32 * 32 >= sg.c0;
        const float scale = from1 ? sg.scale1 : 1.0f;
        const bool math = with_tab || act || scale != 1.0f;                 // uniform
        const bool edge = sg.taps == 3 && (tr_m0 == 0 || tr_m0 + kPpTM == a.lin);   // uniform: a halo row is padding
        if (math) {
            f32x2_t fa2[4], fb2[4];
            if (with_tab) {
                const f32x4_t* tp = (const f32x4_t*)(ldsTab + (tr_tseq & 1) * kPpTab + (k32 * 32 + chunk * 8) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t t = tp[e];
                    fa2[e] = f32x2_t{t.x, t.z};
                    fb2[e] = f32x2_t{t.y, t.w};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { fa2[e] = f32x2_t{scale, scale}; fb2[e] = f32x2_t{0.f, 0.f}; }
            }
            auto unit = [&](char* addr, bool valid) __attribute__((always_inline)) {
                const u32x4_t raw = *(const u32x4_t*)addr;
                float f[8];
                unpack16<T>(raw, f);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2_t x2 = {f[2 * e], f[2 * e + 1]};
                    f32x2_t v2 = x2 * fa2[e] + fb2[e];
                    if (act) {
                        const f32x2_t z2 = v2 * -1.4426950408889634f;
                        f32x2_t d2 = {__builtin_amdgcn_exp2f(z2.x), __builtin_amdgcn_exp2f(z2.y)};
                        d2 = d2 + 1.0f;
                        const f32x2_t r2 = {__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
                        v2 = v2 * r2;
                    }
                    f[2 * e] = v2.x; f[2 * e + 1] = v2.y;
                }
                u32x4_t qv = pack16<T>(f);
                qv.x = valid ? qv.x : 0u; qv.y = valid ? qv.y : 0u; qv.z = valid ? qv.z : 0u; qv.w = valid ? qv.w : 0u;
                *(u32x4_t*)addr = qv;
            };
            unit(ldsA, p0 >= 0 && p0 < a.lin);
            unit(ldsA + 8192, p0 + 128 >= 0 && p0 + 128 < a.lin);
            if (halo) {
                if (lane < 8) unit(ldsH, ph >= 0 && ph < a.lin);
            }
        } else if (edge) {
            const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
            if (p0 < 0) *(u32x4_t*)ldsA = z;
            if (halo && lane < 8 && ph >= a.lin) *(u32x4_t*)ldsH = z;
        }
        if (++tr_step == nsteps) { tr_step = 0; ++tr_tseq; geom(tr_tseq, tr_b0, tr_m0, tr_n0); }
        st_tr = st_tr == kPpStages - 1 ? 0 : st_tr + 1;
    };

    // ---- accumulators and fragment addresses -------------------------------------------------------------------
    f32x16_t acc[2][2];
    {
        float bias_r[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_r[j] = ldsBias[is_n0 + wn * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = bias_r[j];
    }
    unsigned aoff0[3];                                       // sub-step parity 0 of tap t, row wm*64 + r + t (stage 0)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = wm * 64 + r + t;
        aoff0[t] = (unsigned)(row * kPpRow + ((h ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned woff0 = (unsigned)(kPpOffW + (wn * 64 + r) * kPpRow + ((h ^ ((r >> 2) & 3)) << 4));
    auto mfma = [&]() __attribute__((always_inline)) {
        const int taps = mm_step >= nst0 ? a.seg[1].taps : a.seg[0].taps;
        if (!(a.dbg & 4)) {
            unsigned aoff[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) aoff[t] = aoff0[t] + (unsigned)(st_mm * kPpAStage);
            const unsigned woff = woff0 + (unsigned)(st_mm * kPpWStage);
            if (taps == 3) pp_mfma_step<3>(acc, smem, aoff, woff);
            else pp_mfma_step<1>(acc, smem, aoff, woff);
        }
        st_mm = st_mm == kPpStages - 1 ? 0 : st_mm + 1;
    };

    // ---- wave-local epilogue of one finished tile --------------------------------------------------------------
    const int cc = lane & 7, rsub = lane >> 3;              // this lane's 16-byte chunk column / row inside an 8-row pass
    auto epilogue = [&](int tseq, int next_n0) __attribute__((always_inline)) {
        int b0, m0, n0;
        geom(tseq, b0, m0, n0);
        float* sc = (float*)(ldsScr + wave * 2048);          // [8][64] fp32
        float* scw = sc + (4 * h) * 64 + r;
        const float* scr = sc + rsub * 64 + cc * 8;
        T* out = (T*)a.out;
        const bool stats_here = a.stats != nullptr && !(a.dbg & 8);
        const int gs = stats_here ? a.out_c / a.stats_groups : 8;
        const int tpg = gs / 8;
        const int n = n0 + wn * 64 + cc * 8;
        const int mw0 = m0 + wm * 64;
        f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        float nb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) nb[j] = ldsBias[next_n0 + wn * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 4 * p4 + e4;
                        scw[e4 * 64 + j * 32] = acc[i][j][e];
                        acc[i][j][e] = nb[j];
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int m = mw0 + i * 32 + 8 * p4 + rsub;
                const unsigned off = (unsigned)((b0 * a.out_rows + m) * a.out_c + n);
                float v[8];
                {
                    const float4 q0 = *(const float4*)(scr), q1 = *(const float4*)(scr + 4);
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                }
                if (!(a.dbg & 1)) *(u32x4_t*)(out + off) = pack16<T>(v);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2_t v2 = {v[e], v[e + 1]};
                    s1v += v2;
                    s2v += v2 * v2;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (stats_here) {
            float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
            for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            for (int o = 8; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (lane < 8 && (cc & (tpg - 1)) == 0) {
                double* sp = a.stats + ((size_t)b0 * a.stats_groups + n / gs) * 2;
                atomicAdd(sp, (double)s1);
                atomicAdd(sp + 1, (double)s2);
            }
        }
    };

    // ---- pipeline ---------------------------------------------------------------------------------------------
    {
        (void)issue();
        int n1 = 0;
        if (Q > 1) n1 = issue();
        wait_loads(n1);
        transform();
        lds_barrier();
    }
    for (int s = 0; s < Q; ++s) {
        const bool more2 = s + 2 < Q, more1 = s + 1 < Q;
        int nissued = 0;
        if (more2) nissued = issue();                        // step s+2 -> stage (s+2) % 3, last read by the MFMAs of step s-1
        if (more1) wait_loads(nissued);                      // step s+1 (and any table DMA of the previous step) has landed
        if (use_tab && mm_step + 3 == nsteps && s + 3 < Q) { // table of the tile that starts at step s+3 -> slot (tile & 1)
            int b0, m0, n0;
            geom(mm_tseq + 1, b0, m0, n0);
            const unsigned boff = (unsigned)wave * 1024u + lane_lds;
            if (boff < (unsigned)ctot0 * 8u)
                pp_dma16(uniform_ptr(a.seg[0].ab) + ((size_t)b0 * ctot0) * 8 + boff,
                         (unsigned)(kPpOffTab + ((mm_tseq + 1) & 1) * kPpTab) + (unsigned)wave * 1024u);
        }
        if (mm_step == 0 && mm_tseq > 0) {                   // the previous tile finished with the last step
            int b0, m0, n0;
            geom(mm_tseq, b0, m0, n0);
            epilogue(mm_tseq - 1, n0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (early) {
            if (more1) transform();
            __builtin_amdgcn_sched_barrier(0);
            mfma();
        } else {
            mfma();
            __builtin_amdgcn_sched_barrier(0);
            if (more1) transform();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++mm_step == nsteps) { mm_step = 0; ++mm_tseq; }
        lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epilogue(mm_tseq - 1, 0);
}

}  // namespace adf
