// Persistent LDS-DMA implicit-GEMM kernel for the large stride-1 layers (bf16 throughput mode), 256 x 128 tiles.
//
// What the measurements behind it say (profiles/README.md, tools/micro/l2_bw.hip):
//   * in the plain kernel of adf_gemm.h the phases of a K step (global loads -> prologue math -> LDS stores -> barrier
//     -> MFMA -> barrier) add up serially (knock-out sweep), the weights of a 128-row tile are 3/4 of the bytes a CU
//     pulls in, and a CU keeps only ~40 vector-memory misses in flight;
//   * a CU pulls ~34 B/clk from L2 and ~11.5 B/clk (6.2 TB/s chip-wide) from HBM with 128-byte pieces, and HALF of
//     either with 64-byte pieces: staged rows must be full 128-byte lines.
// Hence
//   * the block tile is 256 x 128 (8 waves, 4 x 2, wave tile 64 x 64): weight bytes per output row halve;
//   * K is walked in blocks of 64 channels (128-byte rows); a 3-tap block is three sub-steps (one per tap) that share
//     the staged activations and stream one 16 KB weight slab each; ONE workgroup barrier per sub-step;
//   * everything that is in flight is in flight towards LDS: activations (two blocks ahead, ring of 3 stages), weights
//     (one sub-step ahead, 2 stages) and the affine table arrive by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave
//     instruction, M0 = LDS base); completion is hand-counted with s_waitcnt vmcnt(N) (DMA returns in order).  No
//     register ever holds data that has not arrived -- with asm loads into compiler-allocated registers the register
//     allocator copies the "value" (v_mov) between the load and the wait once pressure is high, i.e. copies stale data;
//   * LDS rows are 128 B with no padding (DMA writes a wave's 64 x 16 B contiguously); the 16-byte chunk c of row r
//     sits at slot c ^ ((r >> 1) & 7), applied on the SOURCE side of the DMA: conflict-free ds_read_b128 fragments;
//   * the fused GroupNorm/FiLM/SiLU prologue of the next block runs IN PLACE on the chunks the wave itself fetched
//     (so it needs only the wave's own vmcnt wait) during the last sub-step of the current block; raw segments (the
//     1x1 residual conv, the identity residual) skip it: their bytes go HBM -> LDS -> MFMA untouched;
//   * the identity residual of a resblock is a third kind of K segment: the residual tensor times a packed identity
//     matrix (bf16 x 1.0 accumulated in fp32 is exact), so it streams through the same DMA ring instead of needing
//     prefetch registers in the epilogue;
//   * blocks are persistent (one per CU) and the DMA stream runs across tile boundaries: the epilogue of tile t
//     (wave-local, through a 2 KB scratch per wave) executes while the first blocks of tile t+1 are in flight; the
//     affine table is double-buffered per tile; the bias vector sits in LDS.
// Shapes (checked by the launcher): stride 1, step +1, taps 3 (pad 1) or 1, channels multiple of 64 and sources
// split at a multiple of 64, mrows = lin = out_rows a multiple of 256, n = n_pad = out_c a multiple of 128,
// M tile count a power of two, at least 2 K blocks, no phase scatter.
#pragma once
#include "adf_gemm.h"
#include <type_traits>

namespace adf {


constexpr int kPpTM = 256, kPpTN = 128;                    // kPpTM: the larger of the two tile heights (MT = 2); MT = 1 gives 128
constexpr int kPpRow = 128;                                 // bytes of K per staged row (64 bf16)
constexpr int kPpAStage = 33 * 1024;                        // 32 pieces of 8 rows + the halo piece (rows 256, 257)
constexpr int kPpAStages = 3;
constexpr int kPpWStage = kPpTN * kPpRow;                   // 16,384 B: one tap slab
constexpr int kPpWStages = 2;
constexpr int kPpScratch = 8 * 2048;                        // 8 waves x (8 rows x 64 cols x fp32)
constexpr int kPpTab = 4096;                                // (a, b) of up to 512 channels, per slot
constexpr int kPpBias = 2048;                               // up to 512 output columns
constexpr int kPpOffW = kPpAStages * kPpAStage;             // 101,376
constexpr int kPpOffScr = kPpOffW + kPpWStages * kPpWStage; // 134,144
constexpr int kPpOffTab = kPpOffScr + kPpScratch;           // 150,528
constexpr int kPpOffBias = kPpOffTab + 2 * kPpTab;          // 158,720
constexpr int kPpLds = kPpOffBias + kPpBias;                // 160,768 B
constexpr int kPpMaxCin = kPpTab / 8;
constexpr int kPpMaxN = kPpBias / 4;

// One LDS-DMA piece: the active lanes copy 16 B each from `base + voff` (uniform 64-bit base in SGPRs, 32-bit lane
// offset: no per-lane 64-bit address arithmetic) to LDS byte address lds_dst + 16 * lane.
// (Dynamic LDS starts at byte 0: the kernel has no static __shared__.)
__device__ __forceinline__ void pp_dma16(const char* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(voff), "s"(lds_dst) : "memory", "vcc");   // (vcc is not accepted as the address pair)
}

// wave-uniform description of one K block (64 channels of one segment) of one tile
struct PpBlk {
    const char* asrc;     // source tensor + byte offset of the block's first channel
    const char* w;        // packed weights of (block, tap 0), column n0
    unsigned rowbytes;    // bytes per row of the source tensor
    unsigned rowbase;     // b0 * lin
    int p_lo;             // position inside the sample of staged row 0 (m0 + off0)
    int taps;
    int tabofs;           // byte offset of the block's first channel inside the LDS affine table, or -1 = raw input
    int act;
    float scale;
    int tseq, last;       // tile sequence number; last = the block is the last K block of its tile
};

// MT = 32-row accumulator tiles per wave: block tile (128 MT) x 128.  MT = 1 serves the levels whose 256-row tile count
// would leave CUs idle (L = 256 at batch 64): half the MFMAs per sub-step, but still far less per-step latency than
// the plain kernel.
template <int MT>
__global__ void __launch_bounds__(512) conv_gemm_pp_kernel(const GemmArgs a, int tiles_total, int tm_shift, int tiles_n) {
    typedef bf16_t T;
    constexpr int TM = 128 * MT;                    // rows of the block tile
    constexpr int HP = TM / 8;                      // index of the halo piece (rows TM, TM + 1) inside an A stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsScr = smem + kPpOffScr;
    char* const ldsTab = smem + kPpOffTab;
    float* const ldsBias = (float*)(smem + kPpOffBias);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool early = (wave & 4) == 0;            // waves w and w+4 share a SIMD: they run prologue and MFMAs in opposite order
    const int lrow = lane >> 3;                     // row inside an 8-row DMA piece
    // logical 16-byte chunk stored at this lane's slot: slot ^ ((row >> 1) & 7), row = 8 * piece + lrow, piece = wave + 8 i
    const int chunk = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int srow = wave * 8 + lrow;               // staged row of this lane in unit 0 (+64 per unit, 256 + lrow halo)
    const unsigned lane_lds = (unsigned)lane * 16u; // byte position of the lane inside a piece

    const int nblk_grid = (int)gridDim.x, bidx = (int)blockIdx.x;
    const int t_lo = (int)((long long)bidx * tiles_total / nblk_grid);
    const int t_hi = (int)((long long)(bidx + 1) * tiles_total / nblk_grid);
    const int ntiles = t_hi - t_lo;
    if (ntiles <= 0) return;
    const int ctot0 = a.seg[0].c0 + a.seg[0].c1;
    const int nb0 = ctot0 >> 6;                                              // 64-channel blocks of segment 0
    const int nb1 = a.nseg > 1 ? (a.seg[1].c0 + a.seg[1].c1) >> 6 : 0;
    const int nb = nb0 + nb1;                                                // K blocks per tile
    const int GB = ntiles * nb;                                              // K blocks of this thread block
    const bool gn_in = a.seg[0].gn.gamma != nullptr;                         // the affine table is derived here from the GroupNorm statistics
    const bool use_tab = a.seg[0].ab != nullptr || gn_in;
    const unsigned slab = (unsigned)a.n_pad * (unsigned)kRowBytes;           // one tap of packed weights

    const float inv_tiles_n = 1.0f / (float)tiles_n;
    const int tn_shift = 31 - __builtin_clz((unsigned)tiles_n);
    auto geom = [&](int tseq, int& b0, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_lo + tseq;
        // N tile index fastest; the N tile count need not be a power of two (768-wide q|k|v projections): exact for t < 2^22
        const int tml = __builtin_amdgcn_readfirstlane((tiles_n & (tiles_n - 1)) == 0 ? t >> tn_shift : (int)(((float)t + 0.5f) * inv_tiles_n));
        n0 = (t - tml * tiles_n) * kPpTN;
        b0 = tml >> tm_shift;
        m0 = (tml & ((1 << tm_shift) - 1)) * TM;
    };
    int b_first;
    { int m0_, n0_; geom(0, b_first, m0_, n0_); }
    // (a, b) of one sample's input channels -> table slot: two channels per thread, the arithmetic of gn_finalize_kernel
    auto fill_table = [&](int b, int slot) __attribute__((always_inline)) {
        // both channels' loads go out together, unconditionally (lanes past the end on a clamped index): see gn_affine_load
        const int c_ = tid * 2 < ctot0 ? tid * 2 : 0;
        const GnRaw r0 = gn_affine_load(a.seg[0].gn, b, c_), r1 = gn_affine_load(a.seg[0].gn, b, c_ + 1);
        if (tid * 2 < ctot0) {
            float A0, B0, A1, B1;
            gn_affine_finish<true>(a.seg[0].gn, c_, r0, A0, B0);
            gn_affine_finish<true>(a.seg[0].gn, c_ + 1, r1, A1, B1);
            *(f32x4_t*)(ldsTab + slot * kPpTab + tid * 16) = f32x4_t{A0, B0, A1, B1};
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // wait until at most n of this wave's DMA instructions (the youngest ones) are still in flight
    auto wait_dma = [&](int n) __attribute__((always_inline)) {
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // issue_a never returns more than 6
    };
    auto desc = [&](int tseq, int blk) __attribute__((always_inline)) -> PpBlk {
        const bool s1 = blk >= nb0;
        const GemmSeg& sg = s1 ? a.seg[1] : a.seg[0];
        const int bl = s1 ? blk - nb0 : blk;
        const int cbase = bl * 64;
        const bool from1 = sg.c1 > 0 && cbase >= sg.c0;                    // sources split at a multiple of 64
        int b0, m0, n0;
        geom(tseq, b0, m0, n0);
        PpBlk d;
        d.asrc = (from1 ? uniform_ptr(sg.src1) : uniform_ptr(sg.src0)) + (size_t)(cbase - (from1 ? sg.c0 : 0)) * 2;
        d.w = uniform_ptr(sg.w) + ((size_t)(bl * sg.taps) * a.n_pad + n0) * kRowBytes;
        d.rowbytes = (unsigned)(from1 ? sg.c1 : sg.c0) * 2u;
        d.rowbase = (unsigned)(b0 * a.lin);
        d.p_lo = m0 + sg.off0;
        d.taps = sg.taps;
        // table slot: alternates per tile (DMA of a precomputed table) or per sample (derived in the kernel)
        d.tabofs = (!s1 && use_tab) ? ((gn_in ? b0 - b_first : tseq) & 1) * kPpTab + cbase * 8 : -1;
        d.act = sg.act;
        d.scale = from1 ? sg.scale1 : 1.0f;
        d.tseq = tseq;
        d.last = blk == nb - 1;
        return d;
    };

    // ---- DMA of the activations of K block d into ring stage st: 4 pieces per wave (+ the halo piece in wave 0
    // of a 3-tap block).  With the last block of a tile goes the affine table of the NEXT tile (two blocks before
    // its first use is prepared).  Returns the number of DMA instructions this wave issued (0 / 4 / 5 / 6).
    // part: 0 = pieces 0-1, 1 = pieces 2-3, 2 = halo + table, -1 = everything (pipeline fill)
    auto issue_a = [&](const PpBlk& d, int st, int part) __attribute__((always_inline)) -> int {
        const unsigned colbytes = (unsigned)chunk * 16u;
        const unsigned ldsA = (unsigned)(st * kPpAStage) + (unsigned)wave * 1024u;
        // uniform base = row p_lo of the tile (may be row -1 of the sample: only wave 0's first row, which then fetches row
        // 0 instead and is zeroed later); lane offset = staged row * pitch + column bytes
        const char* const base = d.asrc + ((long long)d.rowbase + d.p_lo) * (long long)d.rowbytes;
        unsigned voff = (unsigned)srow * d.rowbytes + colbytes;
        const unsigned step = 64u * d.rowbytes;
        int n = 0;
        if (part <= 0) {
            const unsigned v0 = (d.p_lo < 0 && srow == 0) ? voff + d.rowbytes : voff;
            pp_dma16(base, v0, ldsA);
            pp_dma16(base + step, voff, ldsA + 8192u);
            n = 2;
        }
        if (MT == 2 && (part < 0 || part == 1)) {
            pp_dma16(base + 2 * step, voff, ldsA + 16384u);
            pp_dma16(base + 3 * step, voff, ldsA + 24576u);
            n += 2;
        }
        if (part < 0 || part == 2) {
            if (wave == 0 && d.taps == 3) {
                // rows TM, TM + 1 of the tile; past the end of the sample both lanes rows fetch row TM (zeroed later)
                const bool hi_ok = d.p_lo + TM + 1 < a.lin;
                const unsigned vh = (hi_ok ? (unsigned)lrow * d.rowbytes : 0u) + colbytes;
                if (lane < 16) pp_dma16(base + (unsigned)TM * d.rowbytes, vh, (unsigned)(st * kPpAStage) + (unsigned)HP * 1024u);
                n += 1;
            }
            if (use_tab && !gn_in && d.last && d.tseq + 1 < ntiles && (unsigned)wave * 1024u < (unsigned)ctot0 * 8u) {
                int b0, m0, n0;
                geom(d.tseq + 1, b0, m0, n0);
                const unsigned boff = (unsigned)wave * 1024u + lane_lds;
                if (boff < (unsigned)ctot0 * 8u)
                    pp_dma16(uniform_ptr(a.seg[0].ab) + ((size_t)b0 * ctot0) * 8, boff,
                             (unsigned)(kPpOffTab + ((d.tseq + 1) & 1) * kPpTab) + (unsigned)wave * 1024u);
                n += 1;
            }
        }
        return n;
    };
    // ---- DMA of one weight slab (tap slab at wsrc) into W stage st: 2 pieces per wave ---------------------
    const unsigned wlane = (unsigned)srow * (unsigned)kRowBytes + (unsigned)chunk * 16u;
    auto issue_w = [&](const char* wsrc, int st) __attribute__((always_inline)) {
#ifdef ADF_PP_KNOCK_W
        return;               // timing knock-out of a diagnostic build (results wrong by construction): no weight-slab DMA
#endif
        const unsigned ldsW = (unsigned)(kPpOffW + st * kPpWStage) + (unsigned)wave * 1024u;
        pp_dma16(wsrc, wlane, ldsW);
        pp_dma16(wsrc + 64 * kRowBytes, wlane, ldsW + 8192u);
    };

    // ---- fused prologue of K block d, in place on the chunks this wave fetched (stage st); zero padding for rows
    // outside the sample.  Raw blocks only need the zero fill (sample-edge tiles of a 3-tap block).
    // The work is cut into 8-byte halves of the wave's pieces and done in three parts (P = 0, 1, 2: one per sub-step
    // of the block that computes meanwhile; P = -1: everything) so that every sub-step has vector work for one wave
    // of a SIMD to do while its partner runs MFMAs; halo piece and zero fill go with the last part.
    auto transform = [&](const PpBlk& d, int st, auto partc) __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        constexpr int NH = 4 * MT;                                          // 8-byte halves of this wave's pieces
        constexpr int B1 = (NH * 3 + 7) / 8, B2 = (NH * 6 + 7) / 8;
        constexpr int H0 = P <= 0 ? 0 : (P == 1 ? B1 : B2);
        constexpr int H1 = P < 0 ? NH : (P == 0 ? B1 : (P == 1 ? B2 : NH));    // (P = 3 never gets here)
        constexpr bool kTail = P < 0 || P >= 2;                             // P = 3: only the zero fill
        char* const ldsA = smem + st * kPpAStage + wave * 1024 + lane_lds;
        char* const ldsH = smem + st * kPpAStage + HP * 1024 + lane_lds;
        const bool halo = wave == 0 && d.taps == 3;
        if (P != 3 && (d.tabofs >= 0 || d.act || d.scale != 1.0f)) {       // uniform
            float fa[8], fb[8];
            if (d.tabofs >= 0) {
                const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tabofs + chunk * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t t = tp[e];
                    fa[2 * e] = t.x; fb[2 * e] = t.y; fa[2 * e + 1] = t.z; fb[2 * e + 1] = t.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { fa[e] = d.scale; fb[e] = 0.f; }
            }
            // staged (all exps, then all adds, then all rcps ...) so the transcendentals pipeline instead of forming
            // one serial dependency chain per element
            auto math4 = [&](const u32x2_t& raw, int sub, auto actc) __attribute__((always_inline)) -> u32x2_t {
                constexpr bool kAct = decltype(actc)::value;
                float v[4];
                v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
                v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], fa[4 * sub + e], fb[4 * sub + e]);
                if constexpr (kAct) {
                    float ex[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_exp2f(v[e] * -1.4426950408889634f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_rcpf(ex[e] + 1.0f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= ex[e];
                }
                u32x2_t o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                return o;
            };
            auto run = [&](auto actc) __attribute__((always_inline)) {
                u32x2_t raw[H1 - H0 > 0 ? H1 - H0 : 1];
#pragma unroll
                for (int hh = H0; hh < H1; ++hh) raw[hh - H0] = *(const u32x2_t*)(ldsA + (hh >> 1) * 8192 + (hh & 1) * 8);
#pragma unroll
                for (int hh = H0; hh < H1; ++hh)
                    *(u32x2_t*)(ldsA + (hh >> 1) * 8192 + (hh & 1) * 8) = math4(raw[hh - H0], hh & 1, actc);
                if (kTail && halo && lane < 16) {
                    const u32x4_t hr = *(const u32x4_t*)ldsH;
                    const u32x2_t lo = math4(u32x2_t{hr.x, hr.y}, 0, actc), hi = math4(u32x2_t{hr.z, hr.w}, 1, actc);
                    *(u32x4_t*)ldsH = u32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
            };
            if (d.act) run(std::true_type{});                               // uniform: two straight-line versions
            else run(std::false_type{});
        }
        if constexpr (kTail) {
            const int p0 = d.p_lo + srow;
            const int ph = d.p_lo + TM + lrow;
            const bool edge = d.taps == 3 && (d.p_lo < 0 || d.p_lo + TM + 2 > a.lin);    // uniform: a halo row is padding
            if (edge) {                                                     // conv zero padding applies to the activated tensor
                const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
                if (p0 < 0) *(u32x4_t*)ldsA = z;
                if (halo && lane < 16 && ph >= a.lin) *(u32x4_t*)ldsH = z;
            }
        }
    };

    // ---- accumulators and fragment addresses -------------------------------------------------------------------
    f32x16_t acc[MT][2];
    // fragment chunk (ks*2 + h) of staged row R sits at byte R*128 + (((ks*2 + h) ^ f) << 4), f = (R >> 1) & 7
    //   = (R*128 + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5)) ^ (ks << 5): one base per tap, XOR selects the 16-channel sub-step
    unsigned abase[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = wm * 32 * MT + r + t, f = (row >> 1) & 7;
        abase[t] = (unsigned)(row * kPpRow + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5));
    }
    const int fw = (r >> 1) & 7;
    const unsigned wbase = (unsigned)((wn * 64 + r) * kPpRow + ((h ^ (fw & 1)) << 4) + ((fw >> 1) << 5));
    // `between(ks)` runs after the MFMAs of 16-channel group ks have been issued: the DMA instructions of the step are
    // placed there, one or two per group, so the ~100 cycles each of them holds the wave are covered by queued MFMAs
    // `gap(q)` (q = 0 .. 8 MT - 1) runs after MFMA q of the sub-step has been issued, fenced so that the compiler keeps it there.
    auto mfma_gaps = [&](auto tapc, int stA, int stW, auto between, auto gapped, auto gap) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr bool kGaps = decltype(gapped)::value;
        const unsigned ab = abase[TAP];
        const char* pa = smem + stA * kPpAStage;
        const char* pw = smem + kPpOffW + stW * kPpWStage;
        bf16x8_t fa[2][MT], fb[2][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *(const bf16x8_t*)(pa + ab + i * 32 * kPpRow);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = *(const bf16x8_t*)(pw + wbase + j * 32 * kPpRow);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[nxt][i] = *(const bf16x8_t*)(pa + (ab ^ (unsigned)((ks + 1) << 5)) + i * 32 * kPpRow);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = *(const bf16x8_t*)(pw + (wbase ^ (unsigned)((ks + 1) << 5)) + j * 32 * kPpRow);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
                    if constexpr (kGaps) {
                        __builtin_amdgcn_sched_barrier(0);
                        gap(ks * 2 * MT + i * 2 + j);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            between(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mfma = [&](auto tapc, int stA, int stW, auto between) __attribute__((always_inline)) {
        mfma_gaps(tapc, stA, stW, between, std::false_type{}, [](int) {});
    };

    // ---- one sub-step with the GroupNorm/FiLM/SiLU prologue of part P of block dn (stage stN) woven into the gaps
    // between its MFMAs, one element per gap and lane (two for MT = 1), in two stages one gap apart (exp2 of element e
    // beside the rcp of element e-1) so that no gap holds a serial transcendental chain.  Measured with per-wave
    // s_memtime stamps (tools/pp_stamps.py): a prologue part placed before or after the MFMAs of its sub-step costs
    // ~1500 cycles whatever its size (LDS round trips + dependent chains while the partner wave owns the issue
    // slots), so the vector work has to ride inside the wave's own MFMA stream.  EARLY waves use the first gaps, the
    // others the last ones; the halo piece (wave 0) goes with the part that has the fewest elements.
    auto mfma_fused = [&](auto tapc, auto partc, auto earlyc, int stA, int stW, const PpBlk& dn, int stN, auto fuse, auto between)
                          __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        constexpr bool EARLY = decltype(earlyc)::value;
        // parts 0 / 1 take the even / odd 8-byte halves of the wave's pieces (MT = 2: of the first three), so each needs
        // the (a, b) of 4 channels only; part 2 takes the last piece (MT = 2).  The halo chunk (wave 0) is split the same
        // way over parts 0 / 1 (MT = 2) or goes whole into the otherwise empty part 2 (MT = 1).
        constexpr int NOWN = P < 2 ? (MT == 2 ? 3 : 2) : (MT == 2 ? 2 : 0);   // own halves of this part
        auto half_of = [](int idx) constexpr -> int { return P < 2 ? P + 2 * idx : 6 + idx; };
        constexpr int NHU = !EARLY ? 0 : (MT == 2 ? (P < 2 ? 1 : 0) : (P < 2 ? 0 : 2));   // halo halves of this part (wave 0 is an early wave)
        constexpr int HH0 = MT == 2 ? P : 0;                                // first of them
        constexpr bool kHalo = NHU > 0;
        constexpr int NE = 4 * NOWN, NEH = NE + 4 * NHU;
        constexpr int GAPS = 8 * MT;
        static_assert(NEH <= GAPS, "one element per gap");
        constexpr int G0 = EARLY ? 0 : GAPS - NEH;
        char* const ldsN = smem + stN * kPpAStage + wave * 1024 + lane_lds;
        char* const ldsH = smem + stN * kPpAStage + HP * 1024 + lane_lds;
        const bool halo = kHalo && wave == 0 && dn.taps == 3;               // uniform
        // `fuse` (uniform) = block dn has a GroupNorm + SiLU prologue; otherwise the gaps stay empty (same code path, so
        // that the accumulators keep one register assignment across all kinds of sub-steps)
        float ta[8], tb[8];
        constexpr int NU = NOWN + NHU;                                      // 8-byte units: own halves, then halo halves
        u32x2_t raw[NU > 0 ? NU : 1];
        if (fuse) {
            const f32x4_t* tp = (const f32x4_t*)(ldsTab + dn.tabofs + chunk * 64);
#pragma unroll
            for (int e = (P == 1 ? 2 : 0); e < (P == 0 ? 2 : 4); ++e) {
                const f32x4_t t = tp[e];
                ta[2 * e] = t.x; tb[2 * e] = t.y; ta[2 * e + 1] = t.z; tb[2 * e + 1] = t.w;
            }
#pragma unroll
            for (int i = 0; i < NOWN; ++i) raw[i] = *(const u32x2_t*)(ldsN + (half_of(i) >> 1) * 8192 + (half_of(i) & 1) * 8);
            if (halo) {
#pragma unroll
                for (int u = 0; u < NHU; ++u) raw[NOWN + u] = *(const u32x2_t*)(ldsH + (HH0 + u) * 8);
            }
        }
        float vp[2], tq[2], ow[4];                                           // pipeline registers alternate by element parity: no copies
        auto valid = [&](int e) __attribute__((always_inline)) -> bool { return e >= 0 && e < NEH && (e < NE || halo); };
        auto chan_of = [&](int e) __attribute__((always_inline)) -> int { return e < NE ? ((half_of(e >> 2) & 1) * 4 + (e & 3)) : (HH0 + ((e - NE) >> 2)) * 4 + (e & 3); };
        // One gap: stage b of element eb (1 + exp -> rcp -> product; pack + store when its 8-byte unit is complete) beside
        // stage a of element ea = eb + 1 (unpack, affine, exp2).  The empty asm statements pin the two chains between
        // them (IR passes move pure arithmetic across sched_barrier) and leave their interleaving to the scheduler.
        auto gap = [&](int q) __attribute__((always_inline)) {
            const int ea = q - G0, eb = ea - 1;
            if (fuse) {
                const bool va = valid(ea), vb = valid(eb);
                float told = 0.f, vold = 0.f, x = 0.f, o = 0.f, v = 0.f, t = 0.f;
                if (vb) { told = tq[eb & 1]; vold = vp[eb & 1]; asm volatile("" : "+v"(told)); }
                if (va) {
                    const unsigned w = (ea & 2) ? raw[ea >> 2].y : raw[ea >> 2].x;
                    x = __uint_as_float((ea & 1) ? (w & 0xffff0000u) : (w << 16));
                    asm volatile("" : "+v"(x));
                }
                if (vb) o = vold * __builtin_amdgcn_rcpf(told + 1.0f);
                if (va) {
                    v = fmaf(x, ta[chan_of(ea)], tb[chan_of(ea)]);
                    t = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
                }
                if (vb) asm volatile("" : "+v"(o));
                if (va) { asm volatile("" : "+v"(t)); vp[ea & 1] = v; tq[ea & 1] = t; }
                if (vb) {
                    ow[eb & 3] = o;
                    if ((eb & 3) == 3) {
                        u32x2_t pk;
                        pk.x = pack_bf16x2(ow[0], ow[1]);
                        pk.y = pack_bf16x2(ow[2], ow[3]);
                        if (eb < NE) {
                            const int hh = half_of(eb >> 2);
                            *(u32x2_t*)(ldsN + (hh >> 1) * 8192 + (hh & 1) * 8) = pk;
                        } else if (lane < 16) {
                            *(u32x2_t*)(ldsH + (HH0 + ((eb - NE) >> 2)) * 8) = pk;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        mfma_gaps(tapc, stA, stW, between, std::true_type{}, gap);
        gap(GAPS);                                                           // stage b of the last elements
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
        const bool stats_here = a.stats != nullptr;
        const int gs = stats_here ? a.out_c / a.stats_groups : 8;
        const int tpg = gs / 8;
        const int n = n0 + wn * 64 + cc * 8;
        const int mw0 = m0 + wm * 32 * MT;
        f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        float nb_[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) nb_[j] = ldsBias[next_n0 + wn * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 4 * p4 + e4;
                        scw[e4 * 64 + j * 32] = acc[i][j][e];
                        acc[i][j][e] = nb_[j];
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
                if (a.gelu) {                                            // uniform: FeedForward1d's GELU on the 1x1 "conv" output
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
                }
                *(u32x4_t*)(out + off) = pack16_stored<T>(v);
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

    // ---- pipeline -----------------------------------------------------------------------------------------------
    // K blocks are numbered g = tile * nb + blk over the whole thread block; the activations of block g live in A
    // stage g % 3 (fetched while block g-2 computes, prepared in place during the last sub-step of block g-1), the
    // weight slab of a sub-step lives in W stage (sub-step count) & 1 (fetched during the previous sub-step).
    const std::integral_constant<int, 0> tap0{};
    const std::integral_constant<int, 1> tap1{};
    const std::integral_constant<int, 2> tap2{};
    const std::integral_constant<int, 0> part0{};
    const std::integral_constant<int, 1> part1{};
    const std::integral_constant<int, 2> part2{};
    const std::integral_constant<int, -1> part_all{};
    const std::integral_constant<int, 3> part_tail{};
    int nt = 0, nbk = 0;                                     // (tile, block) cursor of the descriptor stream
    auto next_desc = [&]() __attribute__((always_inline)) -> PpBlk {
        if (++nbk == nb) { nbk = 0; ++nt; }
        return desc(nt < ntiles ? nt : ntiles - 1, nbk);     // past the end: a valid but unused descriptor
    };
    // bias vector and the first tile's affine table: their global loads go out first (one HBM round trip, ~3 us on a cold
    // line, that gates the block's first barrier), then the descriptors and the first DMAs, then the LDS stores
    // (every one of them unconditional -- absent tensors through a dummy pointer, lanes past the end on a clamped index -- and the selects after
    //  the last load: loads under conditions are waited for one by one, see gn_affine_load)
    float bias_v[(kPpMaxN + 511) / 512];
    const bool hb0 = a.bias0 != nullptr, hb1 = a.bias1 != nullptr;                   // uniform
    const float* const dummy_f = (const float*)a.seg[0].w;                           // always there, >= 16 KB
    const float* const pb0 = hb0 ? a.bias0 : dummy_f;
    const float* const pb1 = hb1 ? a.bias1 : dummy_f;
    float b0v[(kPpMaxN + 511) / 512], b1v[(kPpMaxN + 511) / 512];
#pragma unroll
    for (int q = 0; q < (kPpMaxN + 511) / 512; ++q) {
        const int n = tid + q * 512;
        const int bi = n < a.n ? n % a.bias_mod : 0;
        b0v[q] = pb0[hb0 ? bi : 0]; b1v[q] = pb1[hb1 ? bi : 0];
    }
    const bool tab_in = use_tab && !gn_in;                                            // uniform
    const int c_tab = tid * 2 < ctot0 ? tid * 2 : 0;
    const f32x4_t tab_l = *(const f32x4_t*)(tab_in ? (const float*)(a.seg[0].ab + ((size_t)b_first * ctot0 + c_tab) * 2) : dummy_f);
    GnRaw gr0 = {}, gr1 = {};                                                // statistics / gamma / beta / FiLM of this thread's two channels
    if (gn_in) { gr0 = gn_affine_load(a.seg[0].gn, b_first, c_tab); gr1 = gn_affine_load(a.seg[0].gn, b_first, c_tab + 1); }
#pragma unroll
    for (int q = 0; q < (kPpMaxN + 511) / 512; ++q) bias_v[q] = tid + q * 512 < a.n ? (hb0 ? b0v[q] : 0.f) + (hb1 ? b1v[q] : 0.f) : 0.f;
    const f32x4_t tab_v = tab_l;
    PpBlk dc = desc(0, 0);
    (void)issue_a(dc, 0, -1);
    issue_w(dc.w, 0);
    PpBlk d1 = next_desc();
    if (GB > 1) (void)issue_a(d1, 1, -1);
    PpBlk d2 = next_desc();
#pragma unroll
    for (int q = 0; q < (kPpMaxN + 511) / 512; ++q)
        if (tid + q * 512 < a.n_pad) ldsBias[tid + q * 512] = bias_v[q];
    if (gn_in) {
        if (tid * 2 < ctot0) {
            float A0, B0, A1, B1;
            gn_affine_finish<true>(a.seg[0].gn, tid * 2, gr0, A0, B0);
            gn_affine_finish<true>(a.seg[0].gn, tid * 2 + 1, gr1, A1, B1);
            *(f32x4_t*)(ldsTab + tid * 16) = f32x4_t{A0, B0, A1, B1};
        }
    } else if (use_tab && tid * 2 < ctot0) *(f32x4_t*)(ldsTab + tid * 16) = tab_v;
    wait_dma(0);
    __syncthreads();
    {
        int b0, m0, n0;
        geom(0, b0, m0, n0);
        float bias_r[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_r[j] = ldsBias[n0 + wn * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = bias_r[j];
    }
    transform(dc, 0, part_all);
    lds_barrier();

    int stA = 0, stW = 0;
    // the two waves of a SIMD run their own copy of the loop (EARLY decides where the prologue elements sit among the
    // MFMA gaps); both copies reach the same barriers
    auto main_loop = [&](auto earlyc) __attribute__((always_inline)) {
    const bool EARLY = earlyc;
    for (int g = 0; g < GB; ++g) {
        const bool has1 = g + 1 < GB, has2 = g + 2 < GB;
        const int stA1 = stA == kPpAStages - 1 ? 0 : stA + 1;
        const int stA2 = stA1 == kPpAStages - 1 ? 0 : stA1 + 1;
        if (g > 0 && (g % nb) == 0) {                        // first block of a tile: the previous tile is complete
            int b0, m0, n0;
            geom(dc.tseq, b0, m0, n0);
            epilogue(dc.tseq - 1, n0);
        }
        if (dc.taps == 3) {
            // The activations of block g+1 are prepared in three parts beside the three sub-steps of block g; the two waves
            // of a SIMD do (vector part, MFMAs) in opposite order, so one of them always has matrix work for the pipe.
            // ---- tap 0: next slab = tap 1; the activations of block g+2 start their way
            int nA = 0;
            const bool fuse = has1 && d1.tabofs >= 0 && d1.act;               // uniform: the usual case (conv after GroupNorm + SiLU)
            const bool slow = has1 && !fuse && (d1.tabofs >= 0 || d1.act || d1.scale != 1.0f);   // other prologues: beside the MFMAs
            auto sub = [&](auto tapc, auto partc, auto between) __attribute__((always_inline)) {
                if (slow && EARLY) transform(d1, stA1, partc);
                __builtin_amdgcn_sched_barrier(0);
                auto call = [&](auto ec) __attribute__((always_inline)) {
                    if (fuse) mfma_fused(tapc, partc, ec, stA, stW, d1, stA1, std::true_type{}, between);
                    else mfma_fused(tapc, partc, ec, stA, stW, d1, stA1, std::false_type{}, between);
                };
                if (EARLY) call(std::true_type{});
                else call(std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
                if (slow && !EARLY) transform(d1, stA1, partc);
            };
            sub(tap0, part0, [&](int ks) __attribute__((always_inline)) {
                if (ks == 0) issue_w(dc.w + slab, stW ^ 1);
                else if (has2) nA += issue_a(d2, stA2, ks - 1);
            });
            wait_dma(nA);
            stW ^= 1;
            lds_barrier();
            // ---- tap 1
            sub(tap1, part1, [&](int ks) __attribute__((always_inline)) { if (ks == 0) issue_w(dc.w + slab + slab, stW ^ 1); });
            wait_dma(0);
            if (gn_in && (g % nb) == 0 && dc.tseq + 1 < ntiles) {            // first block of a tile: the next tile's sample
                int bc, bn, m0_, n0_;
                geom(dc.tseq, bc, m0_, n0_);
                geom(dc.tseq + 1, bn, m0_, n0_);
                if (bn != bc) fill_table(bn, (bn - b_first) & 1);           // nothing else is in flight here (vmcnt drained)
            }
            stW ^= 1;
            lds_barrier();
            // ---- tap 2: next slab = tap 0 of block g+1; zero padding of block g+1 once its last part is in place
            sub(tap2, part2, [&](int ks) __attribute__((always_inline)) { if (ks == 0 && has1) issue_w(d1.w, stW ^ 1); });
            if (has1 && !slow) transform(d1, stA1, part_tail);
            wait_dma(0);
            stW ^= 1;
            lds_barrier();
        } else {
            // ---- single-tap block: everything of the next block is needed after this one sub-step
            int nA = 0;
            mfma(tap0, stA, stW, [&](int ks) __attribute__((always_inline)) {
                if (ks == 0) { if (has1) issue_w(d1.w, stW ^ 1); }
                else if (has2) nA += issue_a(d2, stA2, ks - 1);
            });
            wait_dma(nA);
            if (has1) transform(d1, stA1, part_all);
            stW ^= 1;
            lds_barrier();
        }
        dc = d1; d1 = d2;
        d2 = next_desc();
        stA = stA1;
    }
    wait_dma(0);
    epilogue(ntiles - 1, 0);
    };
    main_loop(early);
}

}  // namespace adf
