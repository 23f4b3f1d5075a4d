// Resblock conv kernel, SPLIT-bf16 form (ADF_DTYPE_F32X3, adf_common.h): adf_gemm_rb.h's data path -- persistent 256 (128) x 128 tiles, LDS-DMA ring of three activation
// stages and two (three) weight stages, host-written K-block table, prologue in the MFMA gaps, wave-local epilogue -- on fp32 tensors
// (reference: src/models/backbones/unet1d.py:193-207, :297-316).  What changes against the bf16 kernel:
//   * a K block is 32 channels: a staged 128-byte row holds 32 fp32 as it arrives by DMA and, after the prologue, 64 B of bf16 hi parts (16-byte slots 0-3 = the MFMA
//     fragments of K step slot >> 1, lane half slot & 1) + 64 B of lo parts (slots 4-7).  The split is IN PLACE: the eight 16-byte chunks of a row belong to the eight
//     lanes of ONE wave (a DMA piece is 8 rows x 8 chunks of one wave instruction), so the wave reads all raw chunks of its pieces into registers at the head of the
//     block's first sub-step -- every lane of the wave at once -- and writes hi | lo halves afterwards: no lane's write can overtake another lane's read;
//   * every block goes through the prologue (a raw block = affine (scale, 0), no SiLU, by the same instructions): the split is needed either way;
//   * a sub-step is 2 K steps x 4 tiles x 3 MFMAs (lo hi, hi lo, hi hi) = 24 per wave against ~50 vector instructions of prologue: the matrix pipe, not the issue
//     port, bounds it -- the reason this mode wants THIS data path (the generic kernel waits for its fp32 activation rows: tools/experiments/adf_gemm_x3.h);
//   * fragment addresses: with the hi slots at 0-3 and the lo slots at 4-7 the bf16 kernel's four K-step addresses base ^ (ks << 5) ARE (hi K0, hi K1, lo K0, lo K1);
//   * epilogue: fp32 rows (two 16-byte stores per lane and pass), fp32 residual, statistics from the fp32 values.
// Weight slabs are pack_weight_kernel<f32x3_t>'s (hi | lo rows, 32 K elements each), DMA'd with the same source-side swizzle.
// Shapes (try_launch_rbx3): fp32 storage, mrows = lin = out_rows a multiple of 256 (128) with a power-of-two tile count per sample, n = n_pad = out_c in {128, 256}
// (two N tiles), segment 0 = 3 taps (off0 -1) with the GroupNorm table derived in the kernel + SiLU, or raw; channels per source a multiple of 64, at most 512
// with a table, at most 1024 raw; optional segment 1 = 1 tap raw, or an identity residual (epilogue).
#pragma once
#include "adf_gemm_rb.h"

namespace adf {

constexpr int kRbx3MaxBlk = 32;                    // 32-channel K blocks: 512 channels with a table = 16, conv2 + 1x1 residual over a 512-channel concat = 24, raw 1024 = 32
struct Rbx3Args {
    RbHead h;
    RbBlk blk[kRbx3MaxBlk];
};

// MH = 32-row accumulator tiles per wave: 2 -> 256-row block tiles, 1 -> 128-row tiles (as in adf_gemm_rb.h).  One 128-column N tile per workgroup tile (NH = 1): with
// n = 256 a K block is prepared once per N tile -- in this mode the prologue is a small part of a sub-step.
template <int MH>
__global__ void __launch_bounds__(512) conv_gemm_rbx3_kernel(const Rbx3Args a) {
    typedef float T;
    constexpr int NH = 1;
    constexpr int TM = 128 * MH, HP = TM / 8;            // HP: piece index of the halo rows TM, TM + 1 -- they follow row TM - 1 in the stage
    constexpr int TNB = kPpTN * NH;                      // columns of the block tile
    // Weight ring.  256 x 128 tiles: two stages, the slab of sub-step s + 1 fetched during sub-step s, stage = a compile-time parity.
    // 128-row tiles and 256-column tiles (W3): THREE stages (the third in the 16 KB this kernel does not use between the ring and the tables), the slab of
    // sub-step s + 2 fetched during sub-step s.  A 128-row sub-step is 8 MFMAs per wave (~500 matrix cycles per SIMD) and a slab takes 1-1.5 K cycles from
    // L2: with one slab in flight the raw K = 3072 launch ran 1.7 K cycles per sub-step with nothing but the wait in it (50 -> 37 us).  On the 256 x 256 tiles
    // (a K block = six sub-steps of 1.8 K cycles over ONE staged activation block: half the activation DMAs per slab) it is worth 1.6 % of a step
    // (234.3 -> 230.5 ms, A/B), on the 256 x 128 tiles nothing (234.3 / 234.9): there the third slab queues in front of the activation pieces.
    // The stage is then a run-time LDS offset (one v_add per K step of fragment reads).
    constexpr bool W3 = MH == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the argument head and the first three block descriptors: one batch of scalar loads, one wait
    const RbHead H = a.h;
    const RbBlk e_first[3] = {a.blk[0], a.blk[1], a.blk[2]};
    asm volatile("" :: "s"(H.nb3), "s"(H.tiles_total), "s"(H.gn.stats0), "s"(H.gn.gamma), "s"(H.gn.film), "s"(H.bias0), "s"(H.out), "s"(H.stats),
                 "s"(e_first[0].src), "s"(e_first[1].src), "s"(e_first[2].src), "s"(e_first[0].tab), "s"(e_first[1].tab), "s"(e_first[2].tab));
    char* const ldsScr = smem + kPpOffScr;
    char* const ldsTab = smem + kPpOffTab;
    float* const ldsBias = (float*)(smem + kPpOffBias);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);     // logical 16-byte chunk stored at this lane's slot
    const int srow = wave * 8 + lrow;                                            // staged row of this lane in unit 0 (+64 per unit)
    const unsigned lane_lds = (unsigned)lane * 16u;

    __builtin_amdgcn_s_setprio(3);
    const int nblk_grid = (int)gridDim.x, bidx = (int)blockIdx.x;
    // (tiles_total <= 2^22 and the grid <= 256 blocks: the products fit 32 bits)
    const int t_lo = (int)((unsigned)bidx * (unsigned)H.tiles_total / (unsigned)nblk_grid);
    const int t_hi = (int)((unsigned)(bidx + 1) * (unsigned)H.tiles_total / (unsigned)nblk_grid);
    const int ntiles = t_hi - t_lo;
    if (ntiles <= 0) return;
    const int nb3 = H.nb3, nb1 = H.nb1, nb = nb3 + nb1;
    const int ctot0 = H.gn.c0 + H.gn.c1;
    const int tn_shift = H.tiles_n > 1 ? 1 : 0;
    const int tm_mask = (1 << H.tm_shift) - 1;

    // ---- tile geometry: advanced once per tile ------------------------------------------------------------------
    struct Tile { int b0, m0, n0; };
    auto tile_of = [&](int tseq) __attribute__((always_inline)) -> Tile {
        const int t = t_lo + tseq;
        const int tml = t >> tn_shift;
        Tile g;
        g.n0 = (t & (H.tiles_n - 1)) * TNB;
        g.b0 = tml >> H.tm_shift;
        g.m0 = (tml & tm_mask) * TM;
        return g;
    };
    const int b_first = tile_of(0).b0;

    // ---- run-time description of one K block of one tile (all wave-uniform, SGPRs) -------------------------------
    struct Blk {
        const char* abase;    // row p_lo of the tile in the block's source (+ channel offset)
        const char* w;        // weight slab of tap 0 for the tile's N tile
        unsigned pitch;
        int tab;              // byte offset into ldsTab (slot included) or -1
        float scale;
        int edge;             // bit 0: staged row 0 is before the sample; bit 1: staged row TM + 1 is past it (3-tap blocks)
        int taps;
    };
    // Every block goes through the same prologue code: y = act ? silu(a x + b) : a x + b with (a, b) from the GroupNorm table,
    // or (scale, 0) for a raw block (a raw block with scale 1 is split from x * 1.0 + 0 = x exactly).  One code
    // path = one register assignment for the accumulators over the whole loop (an if / else around two copies of a sub-step
    // made the register allocator move accumulator tiles through scratch).
    int d_t = 0, d_k = 0;                              // (tile, block) cursor of the descriptor stream
    Tile d_tile = tile_of(0);
    auto make_desc_of = [&](const RbBlk& e) __attribute__((always_inline)) -> Blk {
        Blk d;
        const int three = d_k < nb3;
        const int p_lo = d_tile.m0 - three;
        d.pitch = e.pitch;
        d.abase = e.src + (long long)(d_tile.b0 * H.L + p_lo) * (long long)e.pitch;
        d.w = e.w + (unsigned)d_tile.n0 * (unsigned)kRowBytes;
        d.tab = e.tab >= 0 ? ((d_tile.b0 - b_first) & 1) * kPpTab + e.tab : -1;
        d.scale = e.scale;
        d.taps = three ? 3 : 1;
        d.edge = three ? ((d_tile.m0 == 0 ? 1 : 0) | (d_tile.m0 + TM >= H.L ? 2 : 0)) : 0;
        return d;
    };
    auto make_desc = [&]() __attribute__((always_inline)) -> Blk { return make_desc_of(a.blk[d_k]); };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++d_k == nb) {
            d_k = 0;
            if (++d_t < ntiles) d_tile = tile_of(d_t);    // past the end: a valid but unused descriptor
        }
    };
    const unsigned slab = (unsigned)H.n * (unsigned)kRowBytes;           // one tap of packed weights

    // ---- DMA ----------------------------------------------------------------------------------------------------
    const unsigned colbytes = (unsigned)chunk * 16u;
    // activations of block d into ring stage offset `st` (bytes): pieces 0-1 / 2-3 / halo
    auto issue_a01 = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        const unsigned v = (unsigned)srow * d.pitch + colbytes;
        const unsigned v0 = ((d.edge & 1) && srow == 0) ? v + d.pitch : v;      // row -1 of the sample: fetch row 0, zeroed later
        const unsigned l = st + (unsigned)wave * 1024u;
        rb_dma2(d.abase, v0, v + 64u * d.pitch, l, l + 8192u);
    };
    auto issue_a23 = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (MH == 1) return;
        const unsigned v = (unsigned)(srow + 128) * d.pitch + colbytes;
        const unsigned l = st + (unsigned)wave * 1024u + 16384u;
        rb_dma2(d.abase, v, v + 64u * d.pitch, l, l + 8192u);
    };
    auto issue_halo = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (wave == 0 && d.taps == 3) {
            // rows TM, TM + 1 of the tile; past the end of the sample both lane rows fetch row TM (zeroed later)
            const unsigned vh = (unsigned)TM * d.pitch + ((d.edge & 2) ? 0u : (unsigned)lrow * d.pitch) + colbytes;
            if (lane < 16) rb_dma1(d.abase, vh, st + (unsigned)HP * 1024u);
        }
    };
    const unsigned wlane = (unsigned)srow * (unsigned)kRowBytes + colbytes;
    auto issue_w = [&](const char* wsrc, int wst) __attribute__((always_inline)) {
        const unsigned l = (unsigned)(kPpOffW + wst * kPpWStage) + (unsigned)wave * 1024u;
        rb_dma2(wsrc, wlane, wlane + 64u * (unsigned)kRowBytes, l, l + 8192u);
    };
    auto issue_w_at = [&](const char* wsrc, unsigned wofs) __attribute__((always_inline)) {       // wofs: stage byte offset from kPpOffW
        const unsigned l = (unsigned)kPpOffW + wofs + (unsigned)wave * 1024u;
        rb_dma2(wsrc, wlane, wlane + 64u * (unsigned)kRowBytes, l, l + 8192u);
    };

    // ---- GroupNorm table of one sample: ONE channel per thread, every load issued before the first use -------------------
    // (the arithmetic of gn_finalize_kernel / gn_affine<true>.  Per-wave stamps showed the two-channels-per-thread form on waves
    //  0-1 -- four dependent memory round trips -- holding the block's first barrier until 12.6 K cycles after entry, three times
    //  the landing time of the first DMAs.)
    struct GnLoaded { double s0, q0, s1, q1; float gamma, beta, f1s, f1h, f2s, f2h; };     // raw loads: nothing is computed from them before gn_store
    auto gn_two = [&](int c) __attribute__((always_inline)) -> bool {
        const GnFinalizeArgs& g = H.gn;
        const int ctot = g.c0 + g.c1, gs = ctot / g.G;
        const bool from1 = (c / gs) * gs >= g.c0;
        return gs > (from1 ? g.c1 : g.c0) / g.G;          // two stored (fine) groups per coarse group: two equal sources
    };
    auto gn_load = [&](int b, int c) __attribute__((always_inline)) -> GnLoaded {
        const GnFinalizeArgs& g = H.gn;
        const int ctot = g.c0 + g.c1;
        const int gs = ctot / g.G;                         // channels per (coarse) group
        const int cstart = (c / gs) * gs;
        const bool from1 = cstart >= g.c0;
        const double* st = gn_select_ptr(from1, g.stats0, g.stats1);
        const int csrc = from1 ? g.c1 : g.c0;
        const int lc = from1 ? cstart - g.c0 : cstart;
        const int fg = csrc / g.G;                         // channels per stored (fine) group: gs = fg (one source) or 2 fg (two equal sources)
        const int g0 = lc / fg;
        const bool two = gs > fg;
        const double* p0 = st + ((size_t)b * g.G + g0) * 2;
        const double* p1 = two ? p0 + 2 : p0;
        // every load unconditional and independent (optional FiLM tensors through a valid dummy pointer), no arithmetic on a loaded value here: see
        // gn_affine_load in adf_common.h -- the conditional form was five serialised round trips before this kernel's first DMA, and three to four
        // in the middle of the pipeline at every change of sample
        const bool hf = g.film != nullptr, hf2 = hf && g.film2 != nullptr;                 // uniform
        int cd = c;                                        // opaque copy of the index, so that gamma[cd] through the dummy pointer is not folded into the
        asm volatile("" : "+v"(cd));                       // gamma load (a wait for it)
        const float* const f1 = hf ? g.film + (size_t)b * g.film_bstride : g.gamma;
        const float* const f2 = hf2 ? g.film2 + (size_t)b * g.film2_bstride : g.gamma;
        GnLoaded r;
        r.s0 = p0[0]; r.q0 = p0[1]; r.s1 = p1[0]; r.q1 = p1[1];
        r.gamma = g.gamma[c]; r.beta = g.beta[c];
        r.f1s = f1[cd]; r.f1h = f1[hf ? ctot + cd : cd]; r.f2s = f2[cd]; r.f2h = f2[hf2 ? ctot + cd : cd];
        return r;
    };
    auto gn_store = [&](int c, const GnLoaded& v, int slot) __attribute__((always_inline)) {
        const bool two = gn_two(c);
        const bool hf = H.gn.film != nullptr, hf2 = hf && H.gn.film2 != nullptr;           // uniform
        GnRaw r;
        r.sum = v.s0 + (two ? v.s1 : 0.0); r.sq = v.q0 + (two ? v.q1 : 0.0);
        r.gamma = v.gamma; r.beta = v.beta;
        r.fs = hf ? v.f1s + 1.0f + (hf2 ? v.f2s : 0.f) : 1.0f;
        r.fh = hf ? v.f1h + (hf2 ? v.f2h : 0.f) : 0.0f;
        float A, Bc;
        gn_affine_finish<true>(H.gn, c, r, A, Bc);
        *(f32x2_t*)(ldsTab + slot * kPpTab + c * 8) = f32x2_t{A, Bc};
    };
    auto fill_table = [&](int b, int slot) __attribute__((always_inline)) {
        // (the thread index goes through an empty asm: otherwise the per-lane 64-bit addresses of gamma / beta / FiLM / statistics
        //  are computed at kernel entry and kept -- spilled -- across the whole tile loop)
        int t1 = tid;
        asm volatile("" : "+v"(t1));
        const GnLoaded v = gn_load(b, t1 < ctot0 ? t1 : ctot0 - 1);
        if (t1 < ctot0) gn_store(t1, v, slot);
    };

    // ---- prologue: one GROUP = the 4 fp32 of a 16-byte chunk this lane fetched -> silu(a x + b) (or a x + b) -> bf16 hi | lo halves, in place ------------------------------
    // The lane's chunk holds channels 4 chunk .. 4 chunk + 3 of the block for every one of its pieces, so (a, b) are loaded once per block.
    // act:  y = v * rcp(1 + exp2(-log2(e) v));  no act: the exponent is the constant -200 instead, exp2 underflows to 0 and y = v * rcp(1) = v exactly.
    // Destinations: logical hi slot chunk >> 1 (half chunk & 1) and lo slot 4 + (chunk >> 1), at physical slot (logical ^ f), f = (row >> 1) & 7 -- the same f for every
    // piece of the lane (rows 64 u + 8 wave + lrow).
    const int fsw = ((wave & 1) << 2) + (lane >> 4);                                   // f of this lane's rows
    const unsigned dst_hi = (unsigned)((((chunk >> 1) ^ fsw) << 4) + ((chunk & 1) << 3));      // byte offsets inside the lane's row
    const unsigned dst_lo = (unsigned)((((4 + (chunk >> 1)) ^ fsw) << 4) + ((chunk & 1) << 3));
    const unsigned row_lds = (unsigned)lrow * 128u;                                   // the lane's row inside a piece
    auto load_ab4 = [&](const Blk& d, float* fa, float* fb) __attribute__((always_inline)) {
        if (d.tab >= 0) {                                                 // uniform
            const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tab + chunk * 32);
            const f32x4_t t0 = tp[0], t1 = tp[1];
            fa[0] = t0.x; fb[0] = t0.y; fa[1] = t0.z; fb[1] = t0.w; fa[2] = t1.x; fb[2] = t1.y; fa[3] = t1.z; fb[3] = t1.w;
        } else {
            for (int e = 0; e < 4; ++e) { fa[e] = d.scale; fb[e] = 0.f; }
        }
    };
    struct Grp { float x[4], u[4]; u32x2_t hi, lo; };
    // stage st (0 .. 10) of one group; raw = the chunk as fetched; piece_base = LDS byte address of the lane's row in that piece
    auto grp_stage = [&](Grp& g, const u32x4_t& raw, const float* fa, const float* fb, float ec, float ed, int st, char* rowp) __attribute__((always_inline)) {
        switch (st) {
            case 0:
                g.x[0] = fmaf(__uint_as_float(raw.x), fa[0], fb[0]); g.x[1] = fmaf(__uint_as_float(raw.y), fa[1], fb[1]);
                g.x[2] = fmaf(__uint_as_float(raw.z), fa[2], fb[2]); g.x[3] = fmaf(__uint_as_float(raw.w), fa[3], fb[3]);
                break;
            case 1: for (int e = 0; e < 4; ++e) g.u[e] = fmaf(g.x[e], ec, ed); break;
            case 2: for (int e = 0; e < 4; ++e) g.u[e] = __builtin_amdgcn_exp2f(g.u[e]); break;
            case 3: for (int e = 0; e < 4; ++e) g.u[e] = g.u[e] + 1.0f; break;
            case 4: for (int e = 0; e < 4; ++e) g.u[e] = __builtin_amdgcn_rcpf(g.u[e]); break;
            case 5: for (int e = 0; e < 4; ++e) g.x[e] = g.x[e] * g.u[e]; break;
            case 6: g.hi.x = pack_bf16x2(g.x[0], g.x[1]); g.hi.y = pack_bf16x2(g.x[2], g.x[3]); break;
            case 7:
                g.u[0] = __uint_as_float(g.hi.x << 16); g.u[1] = __uint_as_float(g.hi.x & 0xffff0000u);
                g.u[2] = __uint_as_float(g.hi.y << 16); g.u[3] = __uint_as_float(g.hi.y & 0xffff0000u);
                break;
            case 8: for (int e = 0; e < 4; ++e) g.x[e] = g.x[e] - g.u[e]; break;
            case 9: g.lo.x = pack_bf16x2(g.x[0], g.x[1]); g.lo.y = pack_bf16x2(g.x[2], g.x[3]); break;
            default:
                *(u32x2_t*)(rowp + dst_hi) = g.hi;
                *(u32x2_t*)(rowp + dst_lo) = g.lo;
                break;
        }
    };
    constexpr int NST = 11;                                            // stages of a group
    constexpr int NU = 2 * MH;                                         // pieces (of 8 rows) per lane and block: rows 64 u + 8 wave + lrow
    // zero padding of the activated tensor: staged row 0 / staged row TM + 1 of an edge tile
    auto zero_fill = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (d.edge) {                                                     // uniform
            const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
            if ((d.edge & 1) && srow == 0) *(u32x4_t*)(smem + st + wave * 1024 + lane_lds) = z;
            if ((d.edge & 2) && wave == 0 && lane >= 8 && lane < 16) *(u32x4_t*)(smem + st + HP * 1024 + lane_lds) = z;
        }
    };
    // whole prologue of block d (stage st) at once: the pipeline fill, and the block after a 1-tap block
    auto transform_all = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        char* const rowbase = smem + st + wave * 1024 + row_lds;
        const bool halo = wave == 0 && d.taps == 3 && lane < 16;
        float fa[4], fb[4];
        load_ab4(d, fa, fb);
        const float ec = d.tab >= 0 ? -1.4426950408889634f : 0.0f, ed = d.tab >= 0 ? 0.0f : -200.0f;
        u32x4_t raw[NU + 1];
#pragma unroll
        for (int u = 0; u < NU; ++u) raw[u] = *(const u32x4_t*)(smem + st + wave * 1024 + lane_lds + u * 8192);
        raw[NU] = *(const u32x4_t*)(smem + st + HP * 1024 + (halo ? lane_lds : 0u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // every lane holds its raw chunks before any hi | lo half is written
        Grp g;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int stg = 0; stg < NST; ++stg) grp_stage(g, raw[u], fa, fb, ec, ed, stg, rowbase + u * 8192);
        if (halo) {
#pragma unroll
            for (int stg = 0; stg < NST; ++stg) grp_stage(g, raw[NU], fa, fb, ec, ed, stg, smem + st + HP * 1024 + row_lds);
        }
        zero_fill(d, st);
    };

    // ---- accumulators and fragment addresses ----------------------------------------------------------------------
    f32x16_t acc[NH][MH][2];
    // fragment chunk (ks*2 + h) of staged row R sits at byte R*128 + (((ks*2 + h) ^ f) << 4), f = (R >> 1) & 7
    //   = (R*128 + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5)) ^ (ks << 5)
    unsigned abase0[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = wm * 32 * MH + r + t, f = (row >> 1) & 7;
        abase0[t] = (unsigned)(row * kPpRow + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5));
    }
    const int fw = (r >> 1) & 7;
    const unsigned wbase0 = (unsigned)((wn * 64 + r) * kPpRow + ((h ^ (fw & 1)) << 4) + ((fw >> 1) << 5));
    unsigned wadr[4];                                   // weight fragment addresses of the 4 K steps (stage 0; + kPpWStage for stage 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wadr[ks] = (unsigned)kPpOffW + (wbase0 ^ (unsigned)(ks << 5));

    // One sub-step: tap TAP over the 32 channels of the block in stage `sa_` with the weight slab in stage WST: 2 K steps x MH x 2 tiles x 3 MFMAs (lo hi, hi lo, hi hi).
    // `work(q)` (q = 0 .. 12 MH - 1) is emitted after MFMA q: one prologue stage of the next block; `mid(m)` (m = 0 .. 3) at the points where adf_gemm_rb.h's K steps
    // end (two per K step here): the DMA instructions of the step.
    constexpr int GAPS = 12 * MH;
    unsigned ws0 = 0u, ws1 = (unsigned)kPpWStage, ws2 = 2u * (unsigned)kPpWStage;     // W3: stage offsets of sub-steps s, s + 1, s + 2
    auto substep = [&](auto tapc, auto wstc, auto nhc, unsigned sa_, auto work, auto mid) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int WST = decltype(wstc)::value;
        const char* const pw = smem + (W3 ? ws0 : (unsigned)(WST * kPpWStage));
        unsigned aadr[4];                               // hi K step 0, hi K step 1, lo K step 0, lo K step 1
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) aadr[ks] = sa_ + (abase0[TAP] ^ (unsigned)(ks << 5));
        bf16x8_t fah[2][MH], fal[2][MH], fbh[2][2], fbl[2][2];
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            fah[0][i] = *(const bf16x8_t*)(smem + aadr[0] + i * 32 * kPpRow);
            fal[0][i] = *(const bf16x8_t*)(smem + aadr[2] + i * 32 * kPpRow);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fbh[0][j] = *(const bf16x8_t*)(pw + wadr[0] + j * 32 * kPpRow);
            fbl[0][j] = *(const bf16x8_t*)(pw + wadr[2] + j * 32 * kPpRow);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < MH; ++i) {
                    fah[1][i] = *(const bf16x8_t*)(smem + aadr[1] + i * 32 * kPpRow);
                    fal[1][i] = *(const bf16x8_t*)(smem + aadr[3] + i * 32 * kPpRow);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fbh[1][j] = *(const bf16x8_t*)(pw + wadr[1] + j * 32 * kPpRow);
                    fbl[1][j] = *(const bf16x8_t*)(pw + wadr[3] + j * 32 * kPpRow);
                }
            }
#pragma unroll
            for (int i = 0; i < MH; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const bf16x8_t& av = m == 0 ? fal[k][i] : fah[k][i];
                        const bf16x8_t& bv = m == 1 ? fbl[k][j] : fbh[k][j];
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[0][i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        work(k * 6 * MH + (i * 2 + j) * 3 + m);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            mid(2 * k);
            __builtin_amdgcn_sched_barrier(0);
            mid(2 * k + 1);
            if (k == 0) __builtin_amdgcn_s_setprio(1);          // progress-based priority (adf_gemm_rb.h, ADF_RB_PRIO = 4): the wave that is behind wins
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(3);
    };

    // Prologue of block dn (stage sn) as per-gap work in the three sub-steps of the block before it: its NU + 1 groups (the lane's chunk of each of its pieces + of the
    // halo piece) x 11 stages, ONE stage (4 independent instructions) per MFMA gap, in order -- 55 (33) of the 72 (36) gaps of a block.  All raw chunks are read at
    // the head of the first sub-step (see the header: in-place split).
    struct Part {
        u32x4_t raw[NU + 1];
        float ta[4], tb[4];
        float ec, ed;          // exponent = ec * v + ed: (-log2 e, 0) with SiLU, (0, -200) without
        Grp g;
    };
    auto part_begin = [&](auto partc, const Blk& dn, unsigned sn, Part& p) __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        if (P != 0) return;
        load_ab4(dn, p.ta, p.tb);
        p.ec = dn.tab >= 0 ? -1.4426950408889634f : 0.0f;
        p.ed = dn.tab >= 0 ? 0.0f : -200.0f;
#pragma unroll
        for (int u = 0; u < NU; ++u) p.raw[u] = *(const u32x4_t*)(smem + sn + wave * 1024 + lane_lds + u * 8192);
        // (stale bytes when the next block has one tap or this lane holds no halo chunk: computed, never stored)
        p.raw[NU] = *(const u32x4_t*)(smem + sn + HP * 1024 + (lane < 16 ? lane_lds : 0u));
    };
    auto part_gap = [&](auto partc, const Blk& dn, unsigned sn, Part& p, int q) __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        const int gq = P * GAPS + q;                   // gap of the block (compile-time after inlining)
        if (gq >= (NU + 1) * NST) return;
        const int grp = gq / NST, stg = gq % NST;
        if (grp < NU) {
            grp_stage(p.g, p.raw[grp], p.ta, p.tb, p.ec, p.ed, stg, smem + sn + wave * 1024 + row_lds + grp * 8192);
        } else if (stg < NST - 1) {
            grp_stage(p.g, p.raw[NU], p.ta, p.tb, p.ec, p.ed, stg, smem);
        } else if (wave == 0 && dn.taps == 3 && lane < 16) {          // the halo chunk's store
            grp_stage(p.g, p.raw[NU], p.ta, p.tb, p.ec, p.ed, stg, smem + sn + HP * 1024 + row_lds);
        }
        // keep the stage in this gap (IR passes move pure arithmetic across sched_barrier)
#pragma unroll
        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(p.g.x[e])); asm volatile("" : "+v"(p.g.u[e])); }
    };
    // what does not ride in the gaps: the zero padding of an edge tile, after the last sub-step of the block
    auto part_end = [&](const Blk& dn, unsigned sn) __attribute__((always_inline)) { zero_fill(dn, sn); };

    // ---- wave-local epilogue of one finished tile ---------------------------------------------------------------
    const int cc = lane & 7, rsub = lane >> 3;
    // Round 3 (profiles/r03_rb_launch_timeline.txt: 6.3 K cycles per tile, a seventh of the launch, as eight serial passes of
    // LDS write -> wait -> LDS read -> wait -> store): the passes are software-pipelined over TWO 2 KB buffers per wave -- while pass p is
    // packed and stored, the rows of pass p + 1 are already on their way back from LDS and the accumulators of pass p + 2 on their way
    // in.  LDS operations of one wave execute in issue order, so "write p + 2 behind read p" is all the ordering the two buffers need.
    // The buffers live in the activation stage that is free at a tile boundary (`scr_stage`: the ring stage of the tile's last block).
    auto epilogue = [&](const Tile& g, int next_n0, unsigned scr_stage) __attribute__((always_inline)) {
        float* const sc0 = (float*)(smem + scr_stage + wave * 4096);          // 2 x [8][64] fp32
        T* out = (T*)H.out;
        const bool stats_here = H.stats != nullptr;
        const int gs = stats_here ? (H.stats_mod ? H.stats_mod : H.n) / H.stats_groups : 8;
        const int tpg = gs / 8;
        const int mw0 = g.m0 + wm * 32 * MH;
        const T* resp = (const T*)H.res;
        const bool has_res = resp != nullptr;                                   // uniform
        constexpr int PH = 4 * MH;                                              // passes per N half
        constexpr int P = NH * PH;                                              // passes: (N half, 32-row half, 8-row quarter)
        auto res_off = [&](int pass) __attribute__((always_inline)) -> unsigned {
            const int hf = pass / PH, q = pass % PH;
            const int m = mw0 + (q >> 2) * 32 + 8 * (q & 3) + rsub;
            return (unsigned)((g.b0 * H.L + m) * H.n + g.n0 + hf * kPpTN + wn * 64 + cc * 8);
        };
        float nb_[NH][2];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
#pragma unroll
            for (int j = 0; j < 2; ++j) nb_[hf][j] = ldsBias[next_n0 + hf * kPpTN + wn * 64 + j * 32 + r];
        // accumulators of pass p -> buffer p & 1 (and the next tile's bias into them)
        auto wr = [&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            constexpr int hf = p / PH, i = (p >> 2) % MH, p4 = p & 3;
            float* const scw = sc0 + (p & 1) * 512 + (4 * h) * 64 + r;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    scw[e4 * 64 + j * 32] = acc[hf][i][j][4 * p4 + e4];
                    acc[hf][i][j][4 * p4 + e4] = nb_[hf][j];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        struct Row { float4 q0, q1; u32x4_t res, res1; };
        auto rd = [&](auto pc, Row& rw) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const float* const scr = sc0 + (p & 1) * 512 + rsub * 64 + cc * 8;
            rw.q0 = *(const float4*)(scr);
            rw.q1 = *(const float4*)(scr + 4);
            if (has_res) { rw.res = *(const u32x4_t*)(resp + res_off(p)); rw.res1 = *(const u32x4_t*)(resp + res_off(p) + 4); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        auto flush_stats = [&](int hf) __attribute__((always_inline)) {
            if (stats_here) {
                const int ncol = g.n0 + hf * kPpTN + wn * 64 + cc * 8;
                const int n = H.stats_mod ? (ncol & (H.stats_mod - 1)) : ncol;          // (several phases add into the same group: atomics)
                float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
                for (int o = 1; o < tpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                for (int o = 8; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (lane < 8 && (cc & (tpg - 1)) == 0) {
                    double* sp = H.stats + ((size_t)g.b0 * H.stats_groups + n / gs) * 2;
                    atomicAdd(sp, (double)s1);
                    atomicAdd(sp + 1, (double)s2);
                }
            }
            s1v = f32x2_t{0.f, 0.f}; s2v = f32x2_t{0.f, 0.f};
        };
        auto process = [&](auto pc, const Row& rw) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            float v[8];
            v[0] = rw.q0.x; v[1] = rw.q0.y; v[2] = rw.q0.z; v[3] = rw.q0.w; v[4] = rw.q1.x; v[5] = rw.q1.y; v[6] = rw.q1.z; v[7] = rw.q1.w;
            if (has_res) {
                float rf[8];
                unpack16<float>(rw.res, rf);
                unpack16<float>(rw.res1, rf + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rf[e];
            }
            *(u32x4_t*)(out + res_off(p)) = pack16<float>(v);
            *(u32x4_t*)(out + res_off(p) + 4) = pack16<float>(v + 4);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2_t v2 = {v[e], v[e + 1]};
                s1v += v2;
                s2v += v2 * v2;
            }
            if (p % PH == PH - 1) flush_stats(p / PH);
        };
        Row rows[2];
        wr(std::integral_constant<int, 0>{});
        wr(std::integral_constant<int, 1>{});
        rd(std::integral_constant<int, 0>{}, rows[0]);
        rb_static_for<0, P>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            if (p > 0 && (p * 4) % P == 0) __builtin_amdgcn_s_setprio(3 - (p * 4) / P);
            if constexpr (p + 1 < P) rd(std::integral_constant<int, p + 1>{}, rows[(p + 1) & 1]);
            if constexpr (p + 2 < P) wr(std::integral_constant<int, p + 2>{});
            process(pc, rows[p & 1]);
        });
        __builtin_amdgcn_s_setprio(3);
    };

    auto kstamp = [&](int) __attribute__((always_inline)) {};
    auto tl = [&](int) __attribute__((always_inline)) {};
    kstamp(11);
    tl(110);                                           // (fine start-up stamps 110-114 of the diagnostic build: argument head here)
    // ---- start-up: bias vector, the first sample's table, first DMAs -------------------------------------------------
    // (all start-up loads unconditional -- absent tensors through a dummy pointer, lanes past the end on a clamped index: two bias loads and the
    //  table loads under conditions were seven serialised round trips ahead of the first DMA)
    const bool hb0 = H.bias0 != nullptr, hb1 = H.bias1 != nullptr;                    // uniform
    const float* const dummy_f = (const float*)a.blk[0].w;                            // always there, >= 1 KB
    const int bidx_l = tid < H.n ? tid : 0;
    const float b0v = (hb0 ? H.bias0 : dummy_f)[bidx_l], b1v = (hb1 ? H.bias1 : dummy_f)[bidx_l];
    GnLoaded gl = {};
    const bool has_tab = H.gn.gamma != nullptr;        // uniform: segment 0 has a GroupNorm table (else every block is raw)
    if (has_tab) gl = gn_load(b_first, tid < ctot0 ? tid : ctot0 - 1);
    tl(111);                                           // statistics / parameter loads issued
    // (blocks 0, 1 and 2 mod nb of the first tile -- or, with two blocks per tile, block 0 of the next one -- from the preloaded entries)
    // Round 4 (profiles/r04_rb_launch_timeline_28_1_before.txt, fine stamps): the first barrier waited for the WHOLE fill -- block 0, its slab, the second
    // slab and block 1: 96 KB per workgroup with every CU starting at once land at ~11 B/clk/CU = 9 K cycles (the older four waves had issued by cycle
    // 2.5 K, the younger four got their DMAs out at 6-8 K behind them) -- although the first prologue and sub-step need only block 0 and one slab
    // (49 KB).  Those go out first and are waited for; the rest of the fill follows and lands under the first block's prologue (3.4 K cycles).
    Blk dc = make_desc_of(e_first[0]);
    advance();
    issue_a01(dc, 0u); issue_a23(dc, 0u); issue_halo(dc, 0u);
    issue_w(dc.w, 0);
    if (W3 && !ADF_RB_FILL2) issue_w(dc.w + (unsigned)(1 / NH) * slab + (unsigned)(1 % NH) * (unsigned)kPpWStage, 1);     // (a tile starts with a 3-tap block: its second sub-step)
    Blk d1 = make_desc_of(e_first[1]);
    advance();
    if (!ADF_RB_FILL2) { issue_a01(d1, (unsigned)kPpAStage); issue_a23(d1, (unsigned)kPpAStage); issue_halo(d1, (unsigned)kPpAStage); }
    Blk d2 = make_desc_of(d_k == 0 ? e_first[0] : e_first[2]);
    advance();
    tl(112);                                           // first DMAs issued
    if (tid < H.n) ldsBias[tid] = (hb0 ? b0v : 0.f) + (hb1 ? b1v : 0.f);
    if (has_tab && tid < ctot0) gn_store(tid, gl, 0);
    kstamp(12);
    tl(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ADF_RB_FILL2) {
        if (W3) issue_w(dc.w + (unsigned)(1 / NH) * slab + (unsigned)(1 % NH) * (unsigned)kPpWStage, 1);
        issue_a01(d1, (unsigned)kPpAStage); issue_a23(d1, (unsigned)kPpAStage); issue_halo(d1, (unsigned)kPpAStage);
    }
    __syncthreads();
    kstamp(13);
    tl(2);
    Tile cur_tile = tile_of(0);
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        float bias_r[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_r[j] = ldsBias[cur_tile.n0 + hf * kPpTN + wn * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[hf][i][j][e] = bias_r[j];
    }
    transform_all(dc, 0u);
    // (FILL2: block 1 and the second slab, issued behind the first wait, are read from the head of the first sub-step on -- part_begin)
    if (ADF_RB_FILL2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    kstamp(14);
    tl(3);

    // ---- pipeline -----------------------------------------------------------------------------------------------------
    // K blocks are numbered over the whole thread block; block g lives in A stage g % 3 (fetched while block g-2 computes,
    // prepared in place while block g-1 computes), the weight slab of a sub-step in W stage (sub-step count) & 1.
    unsigned sa = 0u, sa1 = (unsigned)kPpAStage, sa2 = 2u * (unsigned)kPpAStage;       // stage byte offsets of blocks g, g+1, g+2
    const std::integral_constant<int, 0> c0{};
    const std::integral_constant<int, 1> c1{};
    const std::integral_constant<int, 2> c2{};
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto rotate = [&]() __attribute__((always_inline)) {
        const unsigned t = sa; sa = sa1; sa1 = sa2; sa2 = t;
        dc = d1; d1 = d2; d2 = make_desc();
        advance();
    };
    int tseq = 0, kb = 0;                              // tile / block of the block being computed
    bool prev_one = false;                             // the block before this one was a 1-tap block
    int remaining = ntiles * nb;                       // K blocks still to compute (this one included)

    // The sub-steps of one K block: u = 0 .. TAPS * NH - 1 = (tap u / NH, N half u % NH), reading W stage (WP + u) & 1 and
    // fetching the slab of sub-step u + 1 (the first one of block g+1 at the end).  The activations of block g+2 go out behind
    // the slabs of sub-steps 0 (halo, pieces 0-1) and 1 (pieces 2-3): spread over time, a slab is never queued behind more
    // than two HBM pieces of its own wave (tools/micro/dma_mix.hip: the two streams share the CU's miss slots, they do not
    // overlap), and the end-of-sub-step wait leaves exactly the pieces issued in that sub-step in flight.
    auto next_slab = [&](int u, int taps_) __attribute__((always_inline)) -> const char* {
        return dc.w + (unsigned)(u / NH) * slab + (unsigned)(u % NH) * (unsigned)kPpWStage;
    };
    // ---- W3: the slab TWO sub-steps ahead, and counted waits ----------------------------------------------------------------------
    // v = index of that sub-step counted from the first sub-step of the current block (ucur of them), running on into block g+1 and,
    // behind a one-sub-step block, g+2.  Returns the number of DMA instructions issued (2 or 0).
    auto slab_of = [&](const Blk& d, int v) __attribute__((always_inline)) -> const char* {
        return d.w + (unsigned)(v / NH) * slab + (unsigned)(v % NH) * (unsigned)kPpWStage;
    };
    auto issue_ahead = [&](int v, int ucur, bool has1, bool has2) __attribute__((always_inline)) -> int {
        if (v < ucur) { issue_w_at(slab_of(dc, v), ws2); return 2; }
        v -= ucur;
        if (!has1) return 0;
        const int u1 = d1.taps * NH;                               // uniform
        if (v < u1) { issue_w_at(slab_of(d1, v), ws2); return 2; }
        if (!has2) return 0;
        issue_w_at(slab_of(d2, v - u1), ws2);
        return 2;
    };
    // at most n of this wave's DMA instructions -- the youngest -- still in flight (loads return in order: everything older is in LDS)
    auto wait_n = [&](int n) __attribute__((always_inline)) {
        if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto w_rotate = [&]() __attribute__((always_inline)) { const unsigned t = ws0; ws0 = ws1; ws1 = ws2; ws2 = t; };
    // any count up to 31 (once per tile: a branch tree is fine here); a smaller count than asked for is always safe
    auto wait_upto = [&](int n) __attribute__((always_inline)) {
#define ADF_RB_WCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        switch (n < 0 ? 0 : (n > 31 ? 31 : n)) {
            ADF_RB_WCASE(0) ADF_RB_WCASE(1) ADF_RB_WCASE(2) ADF_RB_WCASE(3) ADF_RB_WCASE(4) ADF_RB_WCASE(5) ADF_RB_WCASE(6) ADF_RB_WCASE(7)
            ADF_RB_WCASE(8) ADF_RB_WCASE(9) ADF_RB_WCASE(10) ADF_RB_WCASE(11) ADF_RB_WCASE(12) ADF_RB_WCASE(13) ADF_RB_WCASE(14) ADF_RB_WCASE(15)
            ADF_RB_WCASE(16) ADF_RB_WCASE(17) ADF_RB_WCASE(18) ADF_RB_WCASE(19) ADF_RB_WCASE(20) ADF_RB_WCASE(21) ADF_RB_WCASE(22) ADF_RB_WCASE(23)
            ADF_RB_WCASE(24) ADF_RB_WCASE(25) ADF_RB_WCASE(26) ADF_RB_WCASE(27) ADF_RB_WCASE(28) ADF_RB_WCASE(29) ADF_RB_WCASE(30)
            default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
        }
#undef ADF_RB_WCASE
    };
    // vector-memory instructions this wave issued in the epilogue that has just run (0 = none pending): P stores, P residual loads, 2 statistics atomics per N half.
    // Consumed by the wait at the end of the tile's first sub-step.
    int epi_vm = 0;
    bool slab1_pre = false;                            // two-stage ring: the slab of the tile's second sub-step went out ahead of the epilogue
    int aq_prev = 0;                                   // activation DMAs this wave issued in the previous sub-step (behind its slab)
    // one 3-tap block whose first sub-step reads W stage WP; the prologue of block g+1 (any kind) rides in its gaps
    auto block3 = [&](auto wpc) __attribute__((always_inline)) {
        constexpr int WP = decltype(wpc)::value;
        constexpr int U = 3 * NH;
        auto stamp = [&](int) __attribute__((always_inline)) {};
        const bool has1 = remaining > 1, has2 = remaining > 2;
        // after a 1-tap block the activations of block g+1 (issued one sub-step ago) may still be in flight: the prologue
        // parts below read them from the head of this block on
        if (prev_one) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); aq_prev = 0; }
        Part part;
        stamp(0);
        rb_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr int TAP = u / NH, HF = u % NH;
            const std::integral_constant<int, TAP> tapc{};
            const std::integral_constant<int, (WP + u) & 1> wstc{};
            const std::integral_constant<int, HF> hfc{};
            if (HF == 0) part_begin(tapc, d1, sa1, part);
            int wq = 0, aq = 0;                                // W3: DMA instructions of this sub-step: slab / activations
            substep(tapc, wstc, hfc, sa, [&](int q) __attribute__((always_inline)) { part_gap(tapc, d1, sa1, part, q); },
                    [&](int ks) __attribute__((always_inline)) {
                        if (ks == 0) {
                            if (W3) wq = issue_ahead(u + 2, U, has1, has2);
                            else if (u + 1 < U) { if (!(u == 0 && slab1_pre)) issue_w(next_slab(u + 1, 3), (WP + u + 1) & 1); }
                            else if (has1) issue_w(d1.w, (WP + u + 1) & 1);
                        } else if (ks == 1 && has2) {
                            if (u == 0) { issue_halo(d2, sa2); issue_a01(d2, sa2); aq = (wave == 0 && d2.taps == 3) ? 3 : 2; }
                            else if (u == 1 && MH == 2) { issue_a23(d2, sa2); aq = 2; }
                        }
                    });
            if (u == U - 1) part_end(d1, sa1);
            stamp(3 * u + 1);
            // the next slab has landed; the activation pieces issued in this sub-step (the 2 youngest) may still fly
            if (u == 0 && epi_vm > 0) {
                // first sub-step of a tile behind an epilogue: the next slab was issued BEFORE the epilogue's stores / residual loads / atomics, which
                // may all still fly (they are younger), as may what this sub-step issued
                wait_upto(epi_vm + (W3 ? aq_prev + wq + aq : aq));
                epi_vm = 0; slab1_pre = false;
                if (W3) { aq_prev = aq; w_rotate(); }
            } else if (W3) {
                // in flight behind the next sub-step's slab: the activations that followed it, this sub-step's slab and activations; the block's
                // last sub-step also ends the flight of block g+2's activations (their prologue starts with the next sub-step)
                wait_n((u == U - 1 ? 0 : aq_prev) + wq + aq);
                aq_prev = aq;
                w_rotate();
            } else if ((u == 0 || (MH == 2 && u == 1)) && has2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(3 * u + 2);
            if (has_tab && u == 2 && kb == 0 && tseq + 1 < ntiles) {    // first block of a tile: the next tile's sample
                const Tile nt = tile_of(tseq + 1);
                if (nt.b0 != cur_tile.b0) fill_table(nt.b0, (nt.b0 - b_first) & 1);     // nothing else is in flight here
            }
            lds_barrier();
            stamp(3 * u + 3);
        });
    };
    // one 1-tap (raw) block reading W stage WP first: everything of the next block is needed after its NH sub-steps
    auto block1 = [&](auto wpc) __attribute__((always_inline)) {
        constexpr int WP = decltype(wpc)::value;
        const bool has1 = remaining > 1, has2 = remaining > 2;
        rb_static_for<0, NH>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const std::integral_constant<int, (WP + u) & 1> wstc{};
            const std::integral_constant<int, u> hfc{};
            int wq = 0, aq = 0;
            substep(c0, wstc, hfc, sa, [](int) __attribute__((always_inline)) {},
                    [&](int ks) __attribute__((always_inline)) {
                        if (ks == 0) {
                            if (W3) wq = issue_ahead(u + 2, NH, has1, has2);
                            else if (u + 1 < NH) issue_w(next_slab(u + 1, 1), (WP + u + 1) & 1);
                            else if (has1) issue_w(d1.w, (WP + u + 1) & 1);
                        } else if (u == NH - 1 && has2) {
                            if (ks == 1) { issue_a01(d2, sa2); aq += 2; }
                            else if (ks == 2) { issue_a23(d2, sa2); if (MH == 2) aq += 2; }
                            else { issue_halo(d2, sa2); if (wave == 0 && d2.taps == 3) aq += 1; }
                        }
                    });
            if (W3) {
                // last sub-step: block g+1's activations (fetched during block g-1, maybe one sub-step ago) are prepared below -- only this
                // sub-step's own DMAs may still fly; before it: as in a 3-tap block
                wait_n((u == NH - 1 ? 0 : aq_prev) + wq + aq);
                aq_prev = aq;
                w_rotate();
                if (u == NH - 1 && has1) transform_all(d1, sa1);
            } else if (u == NH - 1) {
                // block g+1's activations (issued one block ago) and its first slab have landed; block g+2's may still fly
                if (has2) {
                    if (MH == 2) { if (wave == 0 && d2.taps == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                    else { if (wave == 0 && d2.taps == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (has1) transform_all(d1, sa1);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();
        });
    };
    constexpr int kPar3 = (3 * NH) & 1, kPar1 = NH & 1;       // weight stage parity after one 3-tap / 1-tap block

    for (; tseq < ntiles; ++tseq) {
        cur_tile = tile_of(tseq);
        // the previous tile's accumulators leave, this tile's start from its bias
        if (tseq > 0) {
            if (ADF_RB_PRESLAB) {
                // (a tile that ended with 1-tap blocks: the activations of the new tile's SECOND block went out in its last sub-step and are read from the
                //  head of the first block on -- the wait block3 does for that, taken here, ahead of the epilogue's stores)
                if (prev_one) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); aq_prev = 0; prev_one = false; }
                // (two-stage ring: the stage the second sub-step's slab goes to was read last by the finished tile's last sub-step, whose barrier has passed)
                if (!W3) { issue_w(next_slab(1, 3), 1); slab1_pre = true; }
                epi_vm = NH * 4 * MH * (H.res != nullptr ? 2 : 1) + (H.stats != nullptr ? 2 * NH : 0);
            }
            epilogue(tile_of(tseq - 1), cur_tile.n0, sa2);       // (sa2: the stage of the previous tile's last block, not yet refilled)
        }
        // blocks come in pairs (nb3 and nb1 are even): the weight stage parity is a compile-time constant
        for (kb = 0; kb < nb3; kb += 2) {
            block3(c0); rotate(); --remaining; prev_one = false;
            block3(std::integral_constant<int, kPar3>{}); rotate(); --remaining;
        }
        for (int k1 = 0; k1 < nb1; k1 += 2) {
            block1(c0); rotate(); --remaining;
            block1(std::integral_constant<int, kPar1>{}); rotate(); --remaining; prev_one = true;
        }
    }
    {
        const Tile last = tile_of(ntiles - 1);
        tl(108);
        epilogue(last, 0, sa2);
        tl(109);
    }
}

}  // namespace adf
