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
// tile counts powers of two, at least 2 K blocks, no phase scatter / GELU.
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

// One LDS-DMA piece: the active lanes copy 16 B each from `gsrc` to LDS byte address lds_dst + 16 * lane.
// (Dynamic LDS starts at byte 0: the kernel has no static __shared__.)
__device__ __forceinline__ void pp_dma16(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Four pieces with one M0 set-up: piece 0 copies from g0, piece i > 0 from g1 + (i-1) * gstep; piece i lands at lds_dst + i * 8 KB.
__device__ __forceinline__ void pp_dma16x4(const char* g0, const char* g1, unsigned gstep, unsigned lds_dst) {
    unsigned keep;
    const char* g2 = g1 + gstep;
    const char* g3 = g2 + gstep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_dst) : "memory", "scc");
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
__global__ void __launch_bounds__(512) conv_gemm_pp_kernel(const GemmArgs a, int tiles_total, int tm_shift, int tn_shift) {
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
    const bool use_tab = a.seg[0].ab != nullptr;
    const unsigned slab = (unsigned)a.n_pad * (unsigned)kRowBytes;           // one tap of packed weights

    auto geom = [&](int tseq, int& b0, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_lo + tseq;
        const int tml = t >> tn_shift;
        n0 = (t & ((1 << tn_shift) - 1)) * kPpTN;
        b0 = tml >> tm_shift;
        m0 = (tml & ((1 << tm_shift) - 1)) * TM;
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
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
        d.tabofs = (!s1 && use_tab) ? (tseq & 1) * kPpTab + cbase * 8 : -1;
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
        const int p0 = d.p_lo + srow;                                       // input position of this lane's row, unit 0
        const unsigned ldsA = (unsigned)(st * kPpAStage) + (unsigned)wave * 1024u;
        // only the first row of a tile (unit 0, p = -1) can be padding here; that lane fetches row 0 instead and is zeroed later
        const char* g0 = d.asrc + ((long long)d.rowbase + p0) * (long long)d.rowbytes + colbytes;
        const unsigned step = 64u * d.rowbytes;
        int n = 0;
        if (part < 0 && MT == 2) { pp_dma16x4(p0 < 0 ? g0 + d.rowbytes : g0, g0 + step, step, ldsA); n = 4; }
        if (part == 0 || (part < 0 && MT == 1)) { pp_dma16(p0 < 0 ? g0 + d.rowbytes : g0, ldsA); pp_dma16(g0 + step, ldsA + 8192u); n = 2; }
        if (part == 1 && MT == 2) { pp_dma16(g0 + 2 * step, ldsA + 16384u); pp_dma16(g0 + 3 * step, ldsA + 24576u); n = 2; }
        if (part < 0 || part == 2) {
            if (wave == 0 && d.taps == 3) {
                const int p = d.p_lo + TM + lrow;
                const bool ok = p >= 0 && p < a.lin;
                const unsigned off = ok ? (d.rowbase + (unsigned)p) * d.rowbytes + colbytes : 0u;
                if (lane < 16) pp_dma16(d.asrc + off, (unsigned)(st * kPpAStage) + (unsigned)HP * 1024u);
                n += 1;
            }
            if (use_tab && d.last && d.tseq + 1 < ntiles && (unsigned)wave * 1024u < (unsigned)ctot0 * 8u) {
                int b0, m0, n0;
                geom(d.tseq + 1, b0, m0, n0);
                const unsigned boff = (unsigned)wave * 1024u + lane_lds;
                if (boff < (unsigned)ctot0 * 8u)
                    pp_dma16(uniform_ptr(a.seg[0].ab) + ((size_t)b0 * ctot0) * 8 + boff,
                             (unsigned)(kPpOffTab + ((d.tseq + 1) & 1) * kPpTab) + (unsigned)wave * 1024u);
                n += 1;
            }
        }
        return n;
    };
    // ---- DMA of one weight slab (tap slab at wsrc) into W stage st: 2 pieces per wave ---------------------
    const unsigned wlane = (unsigned)srow * (unsigned)kRowBytes + (unsigned)chunk * 16u;
    auto issue_w = [&](const char* wsrc, int st) __attribute__((always_inline)) {
        const unsigned ldsW = (unsigned)(kPpOffW + st * kPpWStage) + (unsigned)wave * 1024u;
        pp_dma16(wsrc + wlane, ldsW);
        pp_dma16(wsrc + wlane + 64 * kRowBytes, ldsW + 8192u);
    };

    // ---- fused prologue of K block d, in place on the chunks this wave fetched (stage st); zero padding for rows
    // outside the sample.  Raw blocks only need the zero fill (sample-edge tiles of a 3-tap block).
    auto transform = [&](const PpBlk& d, int st) __attribute__((always_inline)) {
        char* const ldsA = smem + st * kPpAStage + wave * 1024 + lane_lds;
        char* const ldsH = smem + st * kPpAStage + HP * 1024 + lane_lds;
        const bool halo = wave == 0 && d.taps == 3;
        const int p0 = d.p_lo + srow;
        const int ph = d.p_lo + TM + lrow;
        const bool edge = d.taps == 3 && (d.p_lo < 0 || d.p_lo + TM + 2 > a.lin);    // uniform: a halo row is padding
        if (d.tabofs >= 0 || d.act || d.scale != 1.0f) {                   // uniform
            f32x2_t fa2[4], fb2[4];
            if (d.tabofs >= 0) {
                const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tabofs + chunk * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t t = tp[e];
                    fa2[e] = f32x2_t{t.x, t.z};
                    fb2[e] = f32x2_t{t.y, t.w};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { fa2[e] = f32x2_t{d.scale, d.scale}; fb2[e] = f32x2_t{0.f, 0.f}; }
            }
            // 8-wide stages (all exps, then all adds, then all rcps ...) so the transcendentals of a chunk pipeline
            // instead of forming one serial dependency chain per pair
            auto math = [&](const u32x4_t& raw, auto actc) __attribute__((always_inline)) -> u32x4_t {
                constexpr bool kAct = decltype(actc)::value;
                float f[8];
                unpack16<T>(raw, f);
                f32x2_t v2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v2[e] = f32x2_t{f[2 * e], f[2 * e + 1]} * fa2[e] + fb2[e];
                if constexpr (kAct) {
                    float ex[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2_t z2 = v2[e] * -1.4426950408889634f;
                        ex[2 * e] = z2.x; ex[2 * e + 1] = z2.y;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) ex[e] = __builtin_amdgcn_exp2f(ex[e]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ex[e] += 1.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ex[e] = __builtin_amdgcn_rcpf(ex[e]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v2[e] = v2[e] * f32x2_t{ex[2 * e], ex[2 * e + 1]};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[2 * e] = v2[e].x; f[2 * e + 1] = v2[e].y; }
                return pack16<T>(f);
            };
            constexpr int NU = 2 * MT;                                      // this wave's pieces of the block
            u32x4_t raw[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) raw[i] = *(const u32x4_t*)(ldsA + i * 8192);
            if (d.act) {                                                    // uniform: two straight-line versions
#pragma unroll
                for (int i = 0; i < NU; ++i) *(u32x4_t*)(ldsA + i * 8192) = math(raw[i], std::true_type{});
                if (halo && lane < 16) *(u32x4_t*)ldsH = math(*(const u32x4_t*)ldsH, std::true_type{});
            } else {
#pragma unroll
                for (int i = 0; i < NU; ++i) *(u32x4_t*)(ldsA + i * 8192) = math(raw[i], std::false_type{});
                if (halo && lane < 16) *(u32x4_t*)ldsH = math(*(const u32x4_t*)ldsH, std::false_type{});
            }
        }
        if (edge) {                                                         // conv zero padding applies to the activated tensor
            const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
            if (p0 < 0) *(u32x4_t*)ldsA = z;
            if (halo && lane < 16 && ph >= a.lin) *(u32x4_t*)ldsH = z;
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
    auto mfma = [&](auto tapc, int stA, int stW, auto between) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
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
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            between(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
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
                *(u32x4_t*)(out + off) = pack16<T>(v);
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
    int nt = 0, nbk = 0;                                     // (tile, block) cursor of the descriptor stream
    auto next_desc = [&]() __attribute__((always_inline)) -> PpBlk {
        if (++nbk == nb) { nbk = 0; ++nt; }
        return desc(nt < ntiles ? nt : ntiles - 1, nbk);     // past the end: a valid but unused descriptor
    };
    PpBlk dc = desc(0, 0);
    PpBlk d1 = next_desc();
    PpBlk d2 = next_desc();
    (void)issue_a(dc, 0, -1);
    if (GB > 1) (void)issue_a(d1, 1, -1);
    issue_w(dc.w, 0);
    // ---- bias vector and the first tile's affine table -> LDS with ordinary loads, issued while the first DMAs fly
    // (their latency and the DMA latency overlap; the compiler's waits for these loads also cover the older DMAs)
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
    transform(dc, 0);
    lds_barrier();

    int stA = 0, stW = 0;
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
            // ---- tap 0: next slab = tap 1; the activations of block g+2 start their way (in three parts)
            int nA = 0;
            mfma(tap0, stA, stW, [&](int ks) __attribute__((always_inline)) {
                if (ks == 0) issue_w(dc.w + slab, stW ^ 1);
                else if (has2) nA += issue_a(d2, stA2, ks - 1);
            });
            wait_dma(nA);
            stW ^= 1;
            lds_barrier();
            // ---- tap 1
            mfma(tap1, stA, stW, [&](int ks) __attribute__((always_inline)) { if (ks == 0) issue_w(dc.w + slab + slab, stW ^ 1); });
            wait_dma(0);
            stW ^= 1;
            lds_barrier();
            // ---- tap 2: next slab = tap 0 of block g+1, whose activations are prepared beside the MFMAs: the two waves
            // of a SIMD do it in opposite order
            if (has1 && early) transform(d1, stA1);
            __builtin_amdgcn_sched_barrier(0);
            mfma(tap2, stA, stW, [&](int ks) __attribute__((always_inline)) { if (ks == 0 && has1) issue_w(d1.w, stW ^ 1); });
            __builtin_amdgcn_sched_barrier(0);
            if (has1 && !early) transform(d1, stA1);
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
            if (has1) transform(d1, stA1);
            stW ^= 1;
            lds_barrier();
        }
        dc = d1; d1 = d2;
        d2 = next_desc();
        stA = stA1;
    }
    wait_dma(0);
    epilogue(ntiles - 1, 0);
}

}  // namespace adf
