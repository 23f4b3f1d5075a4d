// Resblock conv kernel (bf16): the GroupNorm + SiLU -> 3-tap conv (+ 1x1 residual conv / identity residual) launches of
// ResnetBlock1d (reference: src/models/backbones/unet1d.py:193-207, :297-316) on 256 x 128 tiles, persistent, all staging by
// LDS-DMA.  Same data path as adf_gemm_pp.h (ring of 3 activation stages of 64 channels, 2 weight stages of one tap slab,
// 128-byte LDS rows with the 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) applied on the DMA source side, prologue in
// place on the chunks a wave fetched itself, wave-local epilogue through LDS with fp64 statistics atomics), rebuilt around
// what the SQ counters of that kernel said (profiles/r01_pp_kernel_sq_counters.txt: 10.3 scalar + 10.1 vector + 1.4 branch
// instructions per MFMA, the matrix pipe busy 24 % of the time): the instruction stream, not bytes, is what bounds it.
//   * the K blocks of a tile (source tensor, channel offset, weight slab, table offset, kind) are the SAME for every tile: the
//     host writes them into the kernel arguments once (RbArgs::blk) and the kernel reads the next one with a scalar load
//     -- no per-block descriptor arithmetic (that and the 64-bit address arithmetic of every DMA were most of the scalar work);
//   * tile geometry (sample, first row, N tile) is advanced once per tile, all byte offsets are 32 bit (tensors < 4 GiB);
//   * a DMA is s_mov m0 + s_nop + global_load_lds with a uniform base in SGPRs and a per-lane 32-bit offset that is computed
//     once per K block; the two weight pieces / the activation pieces of a step go out in one asm statement;
//   * the loop is specialised: 3-tap blocks in a loop unrolled by two (the weight stage parity is then a compile-time constant
//     and the fragment addresses are VGPR + immediate), 1-tap residual blocks in their own loop, no run-time kind tests
//     inside a sub-step; the fragment base addresses of a block are computed once per block (12 adds) instead of per read;
//   * the prologue elements ride between the MFMAs of the wave's own sub-step (as in adf_gemm_pp.h), emitted as
//     straight-line code per (tap, parity) with the loads of a part at the head of the sub-step.
// What bounds it now (per-wave s_memtime stamps and knock-out builds, tools/rb_stamps.py, profiles/README.md round 2): a sub-step
// of two waves on a SIMD costs the SUM of its parts -- 32 MFMAs + fragment reads ~1280 cycles, the 2 x 100 prologue instructions
// ~740, the DMA issues ~300 -- in whatever order they are arranged: the whole vector burst of waves 0-3 before their MFMAs and
// of waves 4-7 after theirs, or the DMA issues of the two halves at different points of the sub-step, measured equal to the
// woven form (those variants are not kept).  The CU's LDS-DMA path does not overlap its two streams either
// (tools/micro/dma_mix.hip: 16 KB weight slabs from L2 alone 1536 cycles per K block, 32 KB of activations from HBM alone
// 3043, together 5156).
// Shapes (checked by launch_rb): bf16, mrows = lin = out_rows a multiple of 256 with a power-of-two tile count per sample,
// n = n_pad = out_c in {128, 256}, segment 0 = 3 taps (off0 -1) over one or two sources with the GroupNorm table derived in
// the kernel and SiLU, channels per source a multiple of 64, at most 512 input channels; optional segment 1 = 1 tap raw over
// one or two sources (source 1 scaled), or an identity residual (becomes a 1-tap segment against a packed identity).
#pragma once
#include "adf_gemm.h"
#include "adf_gemm_pp.h"
#include <type_traits>

namespace adf {

#ifdef ADF_RB_STAMP
// diagnostic build only (tools/build_variant.sh rbstamp -DADF_RB_STAMP; tools/rb_stamps.py): s_memtime of every wave of thread
// block 0 at the phase boundaries of one steady-state 3-tap K block; the product build contains none of this
extern __device__ unsigned long long adf_rb_stamps[8 * 16];
#endif

#ifdef ADF_RB_TL
// diagnostic build only (tools/build_variant.sh rbtl -DADF_RB_TL -fno-slp-vectorize; tools/rb_timeline.py): the WHOLE launch of two thread
// blocks (block 0 and one from the middle of the grid) as 32-bit s_memtime stamps per wave -- entry, first DMAs issued, landed,
// first block ready; per tile: tile start, previous tile's epilogue done, and per sub-step "MFMAs + gap work done" / "barrier
// passed"; last epilogue done; plus s_memrealtime (100 MHz) at entry and exit, which gives the clock the launch ran at.
// The stamps live in 4 KB of LDS beyond the product layout (the launcher asks for kRbLds) and leave at the end of the block.
extern __device__ unsigned adf_rb_tl[2 * 8 * 128];
constexpr int kRbLds = kPpLds + 3072;
#else
constexpr int kRbLds = kPpLds;
#endif
constexpr int kRbMaxBlk = 24;
// timing knock-outs of diagnostic builds only (tools/build_variant.sh NAME -DADF_RB_STAMP -DADF_RB_KNOCK=bits; results are wrong by
// construction): 1 no prologue work in the gaps, 2 no activation DMA, 4 no weight DMA, 8 no MFMA, 16 no fragment reads
#ifndef ADF_RB_KNOCK
#define ADF_RB_KNOCK 0
#endif
// GroupNorm statistics of the stored tile by v_dot2c_f32_bf16 on the packed pairs instead of unpack + packed-fp32 add / fma
#ifndef ADF_RB_DOT2
#define ADF_RB_DOT2 1
#endif
// non-temporal epilogue accesses (A/B builds): 1 = output stores, 2 = residual loads.  Measured (round 3): in the isolated replay of bench.py (operands
// rotated, no consumer) the resblock launches get 6.5 % faster with them (1.458 -> 1.36 ms per pass), end to end nothing moves (236.8 / 237.3 / 236.8 ms per
// step): the consumer of a tensor then reads from HBM what it found in L2 / the memory-side cache before.  Off in the product.
#ifndef ADF_RB_NT
#define ADF_RB_NT 0
#endif
// three-stage weight ring: 1 = the 128-row forms and the 256-column forms (product), 2 = every form, 3 = the 128-row forms only, 4 = 128-row and 128-column forms, 0 = none (A/B builds)
#ifndef ADF_RB_W3
#define ADF_RB_W3 1
#endif
// wave priorities (round 4; product = 4: progress-based, 229.2 / 230.4 -> 227.5 / 229.1 ms per step in two A/B pairs; tools/micro/rb_floor.hip: the sub-step's
// instruction mix with no data dependencies 1901 -> 1786 cycles; 1-3 measured neutral).  The two waves of a SIMD (w and w + 4) are arbitrated by age: per-wave stamps show waves 0-3 finishing a sub-step's
// work in ~1500 cycles and waves 4-7 in ~2150, the older four then waiting at the barrier.  1 = the younger four at priority 1 throughout;
// 2 = the younger four at priority 2 for the second half (K steps 2-3) of every sub-step, 0 again at its barrier; 3 = as 2 with the first half instead;
// 4 = progress-based: every wave at priority 3 - (K step) inside a sub-step (3 - quarter inside an epilogue), so the wave that is behind wins
#ifndef ADF_RB_PRIO
#define ADF_RB_PRIO 4
#endif
// start-up fill in two parts (1, product): block 0 + its first slab are issued and waited for, the second slab and block 1 go out behind that wait; 0 = the whole
// fill before the first barrier (A/B builds)
#ifndef ADF_RB_FILL2
#define ADF_RB_FILL2 1
#endif
// tile boundary (1, product): the slab of the new tile's second sub-step is issued BEFORE the epilogue of the finished tile (two-stage rings; the three-stage
// rings have it in flight anyway) and the wait at the end of the new tile's first sub-step counts the epilogue's own vector-memory instructions (stores,
// residual loads, statistics atomics) as "may still fly": vmcnt retires in issue order, so with vmcnt(2) there the first sub-step of every tile waited for the
// epilogue's stores and fp64 atomics to be acknowledged (an atomic stays counted for ~2.8 K cycles with every CU issuing) -- the "first sub-step after an
// epilogue costs twice a steady one" of profiles/r03_rb_launch_timeline_final.txt.  0 = as before (A/B builds)
#ifndef ADF_RB_PRESLAB
#define ADF_RB_PRESLAB 1
#endif


struct RbBlk {
    const char* src;      // source tensor + byte offset of the block's first channel
    const char* w;        // packed weights of (block, tap 0), N tile 0
    unsigned pitch;       // bytes per row of the source tensor
    int tab;              // byte offset of the block's first channel inside a table slot, or -1 = raw input
    float scale;          // raw input: multiplier (the skip scale of a concatenated source 1)
    int pad_;
};

// Everything but the block table: the kernel copies it to registers in ONE scalar-load round trip at its entry (round 3: read field
// by field where first used, the kernel arguments were ~12 serialised scalar-cache round trips ahead of the first DMA -- ~7 K cycles
// from wave start to the first DMA instruction, profiles/r03_rb_launch_timeline.txt).
struct RbHead {
    int nb3, nb1;             // K blocks of one tile, in order: nb3 three-tap blocks (GroupNorm + SiLU), then nb1 one-tap raw blocks
    int B, L;                 // samples, rows per sample
    int tm_shift;             // log2(block tiles (256 or 128 rows) per sample)
    int tiles_n;              // N tiles (1 or 2), N tile index fastest
    int tiles_total;
    int n;                    // output channels (= n_pad = out_c)
    GnFinalizeArgs gn;        // GroupNorm (+ FiLM) of segment 0: the affine table is derived here from the statistics
    const float* bias0; const float* bias1;
    void* out;
    const void* res;          // identity residual (same layout as out), added in the epilogue in fp32 before the rounding, or nullptr
    double* stats; int stats_groups;
    int stats_mod;            // 0, or the channels of one phase of a transposed conv in its 3-tap form: column n is channel n % stats_mod (a power of two >= 64)
};
struct RbArgs {
    RbHead h;
    RbBlk blk[kRbMaxBlk];
};

// two DMA pieces of 1 KB from one uniform base: lane offsets va / vb, LDS destinations la / lb (+ 16 * lane)
__device__ __forceinline__ void rb_dma2(const char* base, unsigned va, unsigned vb, unsigned la, unsigned lb) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
                 :: "s"(base), "v"(va), "v"(vb), "s"(la), "s"(lb) : "memory");
}
__device__ __forceinline__ void rb_dma1(const char* base, unsigned va, unsigned la) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(base), "v"(va), "s"(la) : "memory");
}

template <int I, int N, typename F>
__device__ __forceinline__ void rb_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rb_static_for<I + 1, N>(f);
    }
}

// NH = 128-column halves of the N tile: 1 -> 256 x 128 tiles (n = 128); 2 -> 256 x 256 tiles (n = 256): the activations of a K
// block are fetched and activated ONCE and multiplied with the two half slabs of each tap in turn (twice the MFMAs per
// prologue element and per activation byte -- the vector issue slots and the DMA path are what bound the kernel)
// RAW: segment 0 has no prologue (the folded strided convs of Downsample1d): its bytes go HBM -> LDS -> MFMA untouched, the gaps
// stay empty and no table is derived; only the zero padding of the edge tiles is applied.
// MH = 32-row accumulator tiles per wave: 2 -> 256-row block tiles, 1 -> 128-row tiles for the levels whose 256-row tile count leaves CUs idle
// (L = 256 at batch 64: round 3, these launches ran on adf_gemm_pp.h's 128-row form before -- profiles/r03_pp_launch_timeline_L256.txt).  The
// LDS layout is the same (a 128-row block uses the first half of a stage, halo piece behind it).
template <int NH, bool RAW, int MH = 2>
__global__ void __launch_bounds__(512) conv_gemm_rb_kernel(const RbArgs a) {
    typedef bf16_t T;
    constexpr int TM = 128 * MH, HP = TM / 8;            // HP: piece index of the halo rows TM, TM + 1 -- they follow row TM - 1 in the stage
    constexpr int TNB = kPpTN * NH;                      // columns of the block tile
    // Weight ring.  256 x 128 tiles: two stages, the slab of sub-step s + 1 fetched during sub-step s, stage = a compile-time parity.
    // 128-row tiles and 256-column tiles (W3): THREE stages (the third in the 16 KB this kernel does not use between the ring and the tables), the slab of
    // sub-step s + 2 fetched during sub-step s.  A 128-row sub-step is 8 MFMAs per wave (~500 matrix cycles per SIMD) and a slab takes 1-1.5 K cycles from
    // L2: with one slab in flight the raw K = 3072 launch ran 1.7 K cycles per sub-step with nothing but the wait in it (50 -> 37 us).  On the 256 x 256 tiles
    // (a K block = six sub-steps of 1.8 K cycles over ONE staged activation block: half the activation DMAs per slab) it is worth 1.6 % of a step
    // (234.3 -> 230.5 ms, A/B), on the 256 x 128 tiles nothing (234.3 / 234.9): there the third slab queues in front of the activation pieces.
    // The stage is then a run-time LDS offset (one v_add per K step of fragment reads).
    constexpr bool W3 = ADF_RB_W3 == 2 || (ADF_RB_W3 == 1 && (MH == 1 || NH == 2)) || (ADF_RB_W3 == 3 && MH == 1) || (ADF_RB_W3 == 4 && (MH == 1 || NH == 1));
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

    if (ADF_RB_PRIO == 4) __builtin_amdgcn_s_setprio(3);
    if (ADF_RB_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (ADF_RB_PRIO == 3 && wave >= 4) __builtin_amdgcn_s_setprio(2);
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
    // or (scale, 0) for a raw block (a raw block with scale 1 comes back bit for bit: bf16 * 1.0 + 0 rounds to itself).  One code
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
        if (ADF_RB_KNOCK & 2) return;
        const unsigned v = (unsigned)srow * d.pitch + colbytes;
        const unsigned v0 = ((d.edge & 1) && srow == 0) ? v + d.pitch : v;      // row -1 of the sample: fetch row 0, zeroed later
        const unsigned l = st + (unsigned)wave * 1024u;
        rb_dma2(d.abase, v0, v + 64u * d.pitch, l, l + 8192u);
    };
    auto issue_a23 = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if ((ADF_RB_KNOCK & 2) || MH == 1) return;
        const unsigned v = (unsigned)(srow + 128) * d.pitch + colbytes;
        const unsigned l = st + (unsigned)wave * 1024u + 16384u;
        rb_dma2(d.abase, v, v + 64u * d.pitch, l, l + 8192u);
    };
    auto issue_halo = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (ADF_RB_KNOCK & 2) return;
        if (wave == 0 && d.taps == 3) {
            // rows TM, TM + 1 of the tile; past the end of the sample both lane rows fetch row TM (zeroed later)
            const unsigned vh = (unsigned)TM * d.pitch + ((d.edge & 2) ? 0u : (unsigned)lrow * d.pitch) + colbytes;
            if (lane < 16) rb_dma1(d.abase, vh, st + (unsigned)HP * 1024u);
        }
    };
    const unsigned wlane = (unsigned)srow * (unsigned)kRowBytes + colbytes;
    auto issue_w = [&](const char* wsrc, int wst) __attribute__((always_inline)) {
        if (ADF_RB_KNOCK & 4) return;
        const unsigned l = (unsigned)(kPpOffW + wst * kPpWStage) + (unsigned)wave * 1024u;
        rb_dma2(wsrc, wlane, wlane + 64u * (unsigned)kRowBytes, l, l + 8192u);
    };
    auto issue_w_at = [&](const char* wsrc, unsigned wofs) __attribute__((always_inline)) {       // wofs: stage byte offset from kPpOffW
        if (ADF_RB_KNOCK & 4) return;
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

    // ---- prologue arithmetic on one 8-byte half (4 elements) of a chunk this lane fetched ----------------------------
    // act:  y = v * rcp(1 + exp2(-log2(e) v));  no act: the exponent is the constant -200 instead, exp2 underflows to 0 and
    // y = v * rcp(1) = v exactly -- the same instructions, no select
    auto silu4 = [&](const u32x2_t& raw, const float* fa, const float* fb, bool act) __attribute__((always_inline)) -> u32x2_t {
        float v[4], ex[4];
        const float ec = act ? -1.4426950408889634f : 0.0f, ed = act ? 0.0f : -200.0f;
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], fa[e], fb[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_exp2f(fmaf(v[e], ec, ed));
#pragma unroll
        for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_rcpf(ex[e] + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= ex[e];
        u32x2_t o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        return o;
    };
    // (a, b) of the lane's channels [c_lo, c_lo + n) of block d: table entries, or (scale, 0) for a raw block
    auto load_ab = [&](const Blk& d, int c_lo, int n, float* fa, float* fb) __attribute__((always_inline)) {
        if (d.tab >= 0) {                                                 // uniform
            const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tab + chunk * 64 + c_lo * 8);
            for (int e = 0; e < n / 2; ++e) {
                const f32x4_t t = tp[e];
                fa[2 * e] = t.x; fb[2 * e] = t.y; fa[2 * e + 1] = t.z; fb[2 * e + 1] = t.w;
            }
        } else {
            for (int e = 0; e < n; ++e) { fa[e] = d.scale; fb[e] = 0.f; }
        }
    };
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
        char* const ldsA = smem + st + wave * 1024 + lane_lds;
        if (d.tab < 0) {                                                 // uniform: raw block
            if (d.scale != 1.0f) {                                       // scaled skip channels; else the bytes go to the MFMAs untouched
                const float sc = d.scale;
#pragma unroll
                for (int u = 0; u < 2 * MH; ++u) {
                    const u32x4_t q = *(const u32x4_t*)(ldsA + u * 8192);
                    u32x4_t o;
                    o.x = pack_bf16x2(__uint_as_float(q.x << 16) * sc, __uint_as_float(q.x & 0xffff0000u) * sc);
                    o.y = pack_bf16x2(__uint_as_float(q.y << 16) * sc, __uint_as_float(q.y & 0xffff0000u) * sc);
                    o.z = pack_bf16x2(__uint_as_float(q.z << 16) * sc, __uint_as_float(q.z & 0xffff0000u) * sc);
                    o.w = pack_bf16x2(__uint_as_float(q.w << 16) * sc, __uint_as_float(q.w & 0xffff0000u) * sc);
                    *(u32x4_t*)(ldsA + u * 8192) = o;
                }
            }
            return;
        }
        char* const ldsH = smem + st + HP * 1024 + lane_lds;
        const bool halo = wave == 0 && d.taps == 3 && lane < 16;
        const bool act = true;
        float fa[8], fb[8];
        load_ab(d, 0, 8, fa, fb);
#pragma unroll
        for (int u = 0; u < 2 * MH; ++u) {
            const u32x4_t q = *(const u32x4_t*)(ldsA + u * 8192);
            const u32x2_t lo = silu4(u32x2_t{q.x, q.y}, fa, fb, act), hi = silu4(u32x2_t{q.z, q.w}, fa + 4, fb + 4, act);
            *(u32x4_t*)(ldsA + u * 8192) = u32x4_t{lo.x, lo.y, hi.x, hi.y};
        }
        if (halo) {
            const u32x4_t q = *(const u32x4_t*)ldsH;
            const u32x2_t lo = silu4(u32x2_t{q.x, q.y}, fa, fb, act), hi = silu4(u32x2_t{q.z, q.w}, fa + 4, fb + 4, act);
            *(u32x4_t*)ldsH = u32x4_t{lo.x, lo.y, hi.x, hi.y};
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

    // One sub-step: the 16 MFMAs of tap TAP over the 64 channels of the block in stage `sa_` with the weight slab in stage
    // WST.  `work(q)` (q = 0 .. 15) is emitted after MFMA q: prologue elements of the next block; `mid(ks)` after the four
    // MFMAs of K step ks: the DMA instructions of the step.
    unsigned ws0 = 0u, ws1 = (unsigned)kPpWStage, ws2 = 2u * (unsigned)kPpWStage;     // W3: stage offsets of sub-steps s, s + 1, s + 2
    auto substep = [&](auto tapc, auto wstc, auto nhc, unsigned sa_, auto work, auto mid) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int WST = decltype(wstc)::value;
        constexpr int H = decltype(nhc)::value;
        const char* const pw = smem + (W3 ? ws0 : (unsigned)(WST * kPpWStage));
        unsigned aadr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) aadr[ks] = sa_ + (abase0[TAP] ^ (unsigned)(ks << 5));
        bf16x8_t fa[2][MH], fb[2][2];
#pragma unroll
        for (int i = 0; i < MH; ++i) fa[0][i] = *(const bf16x8_t*)(smem + aadr[0] + i * 32 * kPpRow);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = *(const bf16x8_t*)(pw + wadr[0] + j * 32 * kPpRow);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MH; ++i) fa[nxt][i] = *(const bf16x8_t*)(smem + aadr[ks + 1] + i * 32 * kPpRow);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = *(const bf16x8_t*)(pw + wadr[ks + 1] + j * 32 * kPpRow);
            }
#pragma unroll
            for (int i = 0; i < MH; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!(ADF_RB_KNOCK & 8)) acc[H][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[H][i][j], 0, 0, 0);
                    else { asm volatile("" :: "v"(fa[cur][i]), "v"(fb[cur][j])); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ADF_RB_KNOCK & 1)) work(ks * 2 * MH + i * 2 + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            mid(ks);
            if (ADF_RB_PRIO == 4) {          // progress-based: the wave that is behind wins the SIMD's arbitration
                if (ks == 0) __builtin_amdgcn_s_setprio(2);
                else if (ks == 1) __builtin_amdgcn_s_setprio(1);
                else if (ks == 2) __builtin_amdgcn_s_setprio(0);
            }
            if (ADF_RB_PRIO == 2 && ks == 1 && wave >= 4) __builtin_amdgcn_s_setprio(2);
            if (ADF_RB_PRIO == 3 && ks == 1 && wave >= 4) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ADF_RB_PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(0);
        if (ADF_RB_PRIO == 3 && wave >= 4) __builtin_amdgcn_s_setprio(2);
        if (ADF_RB_PRIO == 4) __builtin_amdgcn_s_setprio(3);
    };

    // Prologue of block dn (stage sn) as per-gap work in the three sub-steps of the block before it.  Parts 0 / 1 take the even /
    // odd 8-byte halves of the wave's pieces 0-2 (so each needs the (a, b) of 4 channels only), part 2 piece 3 and the
    // wave's sixteen elements of the halo chunk (one per lane, see halo_e).
    // A half = 4 elements = one GROUP; a group goes through 8 stages of 4 INDEPENDENT instructions each (unpack, affine,
    // exponent, exp2, 1 + t, rcp, product, pack + store): 32 slots per group, 6 slots per MFMA gap.  Measured with per-wave
    // stamps (tools/rb_stamps.py): one element per gap as a dependent chain (unpack -> fma -> fma -> exp2 | add -> rcp -> mul)
    // ran at ~9 cycles per instruction -- with two waves per SIMD nothing hides the latency of a dependent vector instruction
    // -- and the 16 MFMAs of a sub-step took 1700-2100 cycles; four independent elements per stage issue back to back.
    struct Part {
        u32x2_t raw[3];
        float ta[8], tb[8];
        float x[4], u[4];
        u32x2_t pk;
        float ec, ed;          // exponent = ec * v + ed: (-log2 e, 0) with SiLU, (0, -200) without (see silu4)
        unsigned hraw;         // the lane's ONE element of the halo chunk (part 2), its (a, b) and its chain
        float ha, hb, hx, hu;
    };
    // The halo chunk (rows TM, TM + 1 = 128 elements) is spread over all eight waves: element wave * 16 + (lane & 15), one dependent chain of
    // eight instructions per wave riding one slot per gap next to the four independent slots of a group.  (Round 2 had wave 0's lanes 0-15 take the
    // whole chunk as two more GROUPS and every other wave run the same 64 slots on a dummy: a fifth of the prologue's vector instructions, and the
    // kernel is bound by vector issue -- profiles/r03_rb_launch_timeline.txt.)
    const unsigned halo_e = (unsigned)wave * 16u + ((unsigned)lane & 15u);
    auto half_of = [](int P, int k) constexpr -> int { return P < 2 ? P + 2 * k : 6 + k; };
    auto part_begin = [&](auto partc, const Blk& dn, unsigned sn, Part& p) __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        char* const ldsN = smem + sn + wave * 1024 + lane_lds;
        if (P == 0) {
            load_ab(dn, 0, 4, p.ta, p.tb);
            p.ec = dn.tab >= 0 ? -1.4426950408889634f : 0.0f;
            p.ed = dn.tab >= 0 ? 0.0f : -200.0f;
        } else if (P == 1) load_ab(dn, 4, 4, p.ta + 4, p.tb + 4);
        constexpr int NOWN = MH == 2 ? (P < 2 ? 3 : 2) : (P < 2 ? 2 : 0);
#pragma unroll
        for (int k = 0; k < NOWN; ++k) {
            const int hh = half_of(P, k);
            p.raw[k] = *(const u32x2_t*)(ldsN + (hh >> 1) * 8192 + (hh & 1) * 8);
        }
        if (P == 2) {
            // (stale bytes when the next block has one tap: computed, never stored)
            p.hraw = *(const unsigned short*)(smem + sn + HP * 1024 + halo_e * 2u);
            if (dn.tab >= 0) {                                            // uniform
                const f32x2_t t = *(const f32x2_t*)(ldsTab + dn.tab + (halo_e & 63u) * 8u);
                p.ha = t.x; p.hb = t.y;
            } else { p.ha = dn.scale; p.hb = 0.f; }
        }
    };
    auto part_gap = [&](auto partc, const Blk& dn, unsigned sn, Part& p, int q) __attribute__((always_inline)) {
        constexpr int P = decltype(partc)::value;
        constexpr int NG = MH == 2 ? (P < 2 ? 3 : 2) : (P < 2 ? 2 : 0);      // groups of the part (MH = 1: the four halves of pieces 0-1 in parts 0 / 1)
        constexpr int OPG = NG * 32 / (8 * MH * NH);  // slots per gap: 8 MH NH gaps x OPG = NG x 32   (q = 0 .. 8 MH NH - 1)
        char* const ldsN = smem + sn + wave * 1024 + lane_lds;
        if (P == 2 && q % NH == 0 && q / NH < 8) {    // the halo element's chain: stage q / NH
            switch (q / NH) {
                case 0: p.hx = __uint_as_float(p.hraw << 16); break;
                case 1: p.hx = fmaf(p.hx, p.ha, p.hb); break;
                case 2: p.hu = fmaf(p.hx, p.ec, p.ed); break;
                case 3: p.hu = __builtin_amdgcn_exp2f(p.hu); break;
                case 4: p.hu = p.hu + 1.0f; break;
                case 5: p.hu = __builtin_amdgcn_rcpf(p.hu); break;
                case 6: p.hx = p.hx * p.hu; break;
                default:
                    // row TM + 1 past the end of the sample is zeroed by wave 0 (zero_fill): its owners (waves 4-7) leave it alone
                    if (dn.taps == 3 && !((dn.edge & 2) && wave >= 4)) {                      // uniform
                        if (lane < 16) *(unsigned short*)(smem + sn + HP * 1024 + halo_e * 2u) = (unsigned short)pack_bf16x2(p.hx, p.hx);
                    }
                    break;
            }
        }
#pragma unroll
        for (int n = q * OPG; n < (q + 1) * OPG; ++n) {
            const int k = n >> 5, st = (n >> 2) & 7, j = n & 3;
            if (k >= NG) continue;
            // channel of element j of group k inside the lane's 8-channel chunk
            const int hh = half_of(P, k);
            const int ch = (hh & 1) * 4 + j;
            switch (st) {
                case 0: {
                    const unsigned w = (j & 2) ? p.raw[k].y : p.raw[k].x;
                    p.x[j] = __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
                    break;
                }
                case 1: p.x[j] = fmaf(p.x[j], p.ta[ch], p.tb[ch]); break;
                case 2: p.u[j] = fmaf(p.x[j], p.ec, p.ed); break;
                case 3: p.u[j] = __builtin_amdgcn_exp2f(p.u[j]); break;
                case 4: p.u[j] = p.u[j] + 1.0f; break;
                case 5: p.u[j] = __builtin_amdgcn_rcpf(p.u[j]); break;
                case 6: p.x[j] = p.x[j] * p.u[j]; break;
                default:
                    if (j == 0) p.pk.x = pack_bf16x2(p.x[0], p.x[1]);
                    else if (j == 1) p.pk.y = pack_bf16x2(p.x[2], p.x[3]);
                    else if (j == 2) *(u32x2_t*)(ldsN + (hh >> 1) * 8192 + (hh & 1) * 8) = p.pk;
                    break;
            }
        }
        // keep the slots of this gap in this gap (IR passes move pure arithmetic across sched_barrier)
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(p.x[j])); asm volatile("" : "+v"(p.u[j])); }
        if (P == 2) { asm volatile("" : "+v"(p.hx)); asm volatile("" : "+v"(p.hu)); }
    };
    // what does not ride in the gaps: the zero padding of an edge tile, after the MFMAs of part 2
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
        struct Row { float4 q0, q1; u32x4_t res; };
        auto rd = [&](auto pc, Row& rw) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const float* const scr = sc0 + (p & 1) * 512 + rsub * 64 + cc * 8;
            rw.q0 = *(const float4*)(scr);
            rw.q1 = *(const float4*)(scr + 4);
            if (has_res) rw.res = (ADF_RB_NT & 2) ? __builtin_nontemporal_load((const u32x4_t*)(resp + res_off(p))) : *(const u32x4_t*)(resp + res_off(p));
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
                unpack16<T>(rw.res, rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rf[e];
            }
#if ADF_RB_DOT2
            // statistics of the STORED values straight from the packed pairs: v_dot2c_f32_bf16 (exact products of the bf16 pair, fp32 accumulate) against (1, 1)
            // and against itself -- no unpacking of the rounded values, no packed-fp32 adds / fmas
            const u32x4_t pk = pack16<T>(v);
            if (ADF_RB_NT & 1) __builtin_nontemporal_store(pk, (u32x4_t*)(out + res_off(p)));
            else *(u32x4_t*)(out + res_off(p)) = pk;
            {
                typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
                const bf2_t one2 = __builtin_bit_cast(bf2_t, 0x3f803f80u);
                const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf2_t q2 = __builtin_bit_cast(bf2_t, w4[e]);
                    if (e & 1) { s1v.y = __builtin_amdgcn_fdot2_f32_bf16(q2, one2, s1v.y, false); s2v.y = __builtin_amdgcn_fdot2_f32_bf16(q2, q2, s2v.y, false); }
                    else { s1v.x = __builtin_amdgcn_fdot2_f32_bf16(q2, one2, s1v.x, false); s2v.x = __builtin_amdgcn_fdot2_f32_bf16(q2, q2, s2v.x, false); }
                }
            }
#else
            if (ADF_RB_NT & 1) __builtin_nontemporal_store(pack16_stored<T>(v), (u32x4_t*)(out + res_off(p)));
            else *(u32x4_t*)(out + res_off(p)) = pack16_stored<T>(v);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2_t v2 = {v[e], v[e + 1]};
                s1v += v2;
                s2v += v2 * v2;
            }
#endif
            if (p % PH == PH - 1) flush_stats(p / PH);
        };
        Row rows[2];
        wr(std::integral_constant<int, 0>{});
        wr(std::integral_constant<int, 1>{});
        rd(std::integral_constant<int, 0>{}, rows[0]);
        rb_static_for<0, P>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            if (ADF_RB_PRIO == 4 && p > 0 && (p * 4) % P == 0) __builtin_amdgcn_s_setprio(3 - (p * 4) / P);
            if constexpr (p + 1 < P) rd(std::integral_constant<int, p + 1>{}, rows[(p + 1) & 1]);
            if constexpr (p + 2 < P) wr(std::integral_constant<int, p + 2>{});
            process(pc, rows[p & 1]);
        });
        if (ADF_RB_PRIO == 4) __builtin_amdgcn_s_setprio(3);
    };

#ifdef ADF_RB_STAMP
    auto kstamp = [&](int id) __attribute__((always_inline)) {
        if (bidx == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) ((unsigned long long*)(smem + kPpOffBias + 1024))[wave * 16 + id] = t;
        }
    };
#else
    auto kstamp = [&](int) __attribute__((always_inline)) {};
#endif
#ifdef ADF_RB_TL
    const int tl_which = bidx == 0 ? 0 : (bidx == (int)gridDim.x / 2 + 3 ? 1 : -1);
    unsigned* const tl_lds = (unsigned*)(smem + kPpOffBias + 1024) + wave * 128;
    auto tl = [&](int id) __attribute__((always_inline)) {
        if (tl_which >= 0 && id < 128) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) tl_lds[id] = (unsigned)t;
        }
    };
    auto tl_real = [&](int id) __attribute__((always_inline)) {
        if (tl_which >= 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) { tl_lds[id] = (unsigned)t; tl_lds[id + 1] = (unsigned)(t >> 32); }
        }
    };
    if (tl_which >= 0) for (int i = lane; i < 128; i += 64) tl_lds[i] = 0u;
    int tl_sub = 0;
    tl_real(120);
    tl(0);
#else
    auto tl = [&](int) __attribute__((always_inline)) {};
#endif
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
    if constexpr (!RAW) gl = gn_load(b_first, tid < ctot0 ? tid : ctot0 - 1);
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
#ifdef ADF_RB_TL
    asm volatile("" :: "v"(gl.gamma), "v"(gl.beta), "v"(gl.f1s), "v"(gl.f1h), "v"(gl.f2s), "v"(gl.f2h), "v"(gl.s0), "v"(gl.q0), "v"(gl.s1), "v"(gl.q1), "v"(b0v), "v"(b1v));
    tl(113);                                           // their values are here
#endif
    if (tid < H.n) ldsBias[tid] = (hb0 ? b0v : 0.f) + (hb1 ? b1v : 0.f);
    if (!RAW && tid < ctot0) gn_store(tid, gl, 0);
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
    if (RAW) zero_fill(dc, 0u); else transform_all(dc, 0u);
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
#ifdef ADF_RB_STAMP
        const bool stamp_on = bidx == 0 && tseq == 1 && kb == 2 && WP == 0;
        auto stamp = [&](int id) __attribute__((always_inline)) {
            if (stamp_on && id < 16) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) ((unsigned long long*)(smem + kPpOffBias + 1024))[wave * 16 + id] = t;
            }
        };
#else
        auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
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
            if (HF == 0 && !RAW) part_begin(tapc, d1, sa1, part);
            int wq = 0, aq = 0;                                // W3: DMA instructions of this sub-step: slab / activations
            substep(tapc, wstc, hfc, sa, [&](int q) __attribute__((always_inline)) { if (!RAW) part_gap(tapc, d1, sa1, part, HF * 8 * MH + q); },
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
#ifdef ADF_RB_TL
            tl(4 + tseq * 26 + 2 + 2 * tl_sub);
#endif
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
            if (!RAW && u == 2 && kb == 0 && tseq + 1 < ntiles) {    // first block of a tile: the next tile's sample
                const Tile nt = tile_of(tseq + 1);
                if (nt.b0 != cur_tile.b0) fill_table(nt.b0, (nt.b0 - b_first) & 1);     // nothing else is in flight here
            }
            lds_barrier();
            stamp(3 * u + 3);
#ifdef ADF_RB_TL
            tl(4 + tseq * 26 + 3 + 2 * tl_sub);
            ++tl_sub;
#endif
        });
#ifdef ADF_RB_STAMP
        if (stamp_on && lane < 16) adf_rb_stamps[wave * 16 + lane] = ((const unsigned long long*)(smem + kPpOffBias + 1024))[wave * 16 + lane];
#endif
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
#ifdef ADF_RB_TL
        tl_sub = 0;
        tl(4 + tseq * 26);
#endif
        // the previous tile's accumulators leave, this tile's start from its bias
        if (tseq > 0) {
#ifdef ADF_RB_STAMP
            if (tseq == 1) kstamp(15);
#endif
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
#ifdef ADF_RB_TL
        tl(4 + tseq * 26 + 1);
#endif
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
#ifdef ADF_RB_TL
    tl_real(122);
    if (tl_which >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 128; i += 64) adf_rb_tl[(tl_which * 8 + wave) * 128 + i] = tl_lds[i];
    }
#endif
}

}  // namespace adf
