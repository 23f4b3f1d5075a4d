// EXPERIMENT, NOT PART OF THE PRODUCT BUILD (round 3).  Kept as the record of a measured negative result: see profiles/README.md
// "Round 3: the role-split resblock kernel" and DESIGN.md section 4.  To try it again: copy next to adf_gemm_rb.h, include it from
// adf_gemm.hip and launch conv_gemm_rs_kernel<RAW> from try_launch_rb for nh == 1 (it takes the same RbArgs and kRsLds bytes of LDS).
// State: the version of this file is the last one built (consumers also activating chunks 6 / 7, v_dot2c statistics, descriptor one
// block ahead), which FAILS parity (relative error 0.1 on the first resblock); the version before those three changes passed
// tests/test_gpu_parity.py::test_config2_batch64_full_length_bf16_vs_bf16_oracle and ran the dominant launch in 83-97 us against
// 84-87 us for adf_gemm_rb.h on the same boxes -- no gain: with eight waves per CU the SIMD's vector issue is the wall either way
// (tools/micro/role_split.hip).
//
// Role-split resblock conv kernel (bf16, n = 128): the launches of adf_gemm_rb.h's <NH = 1> kernel -- GroupNorm + SiLU -> 3-tap conv
// (+ 1x1 residual conv of the raw concat / identity residual) of ResnetBlock1d (reference: src/models/backbones/unet1d.py:193-207,
// :297-316) and the folded strided convs of Downsample1d (:214-225) -- with the eight waves of a workgroup in two ROLES.
//
// Why (profiles/r03_rb_launch_timeline.txt, tools/rb_timeline.py on the rb kernel): in a sub-step of the symmetric kernel every wave
// runs 16 MFMAs with ~110 vector instructions of SiLU(GroupNorm) prologue and its DMA issues woven between them.  A wave issues in
// order, so an MFMA that finds the matrix pipe busy with its SIMD partner's also holds the vector work queued behind it, and the
// SIMD arbitrates by age: waves 0-3 finished a sub-step's work in ~1500 cycles, waves 4-7 in ~2150, and the older four then sat
// ~900 cycles at the barrier -- 2330 cycles per sub-step for 1024 cycles of matrix work per SIMD.
//
// Here waves 0-3 ("consumers", one per SIMD) issue ONLY fragment reads and MFMAs: 64 rows x 128 columns each, 32 MFMAs per sub-step
// back to back.  Waves 4-7 ("producers", their SIMD partners) issue every LDS-DMA piece and run the whole prologue in place on the
// pieces they fetched themselves; they never touch the matrix pipe, so their vector stream only yields the 8 issue cycles of each
// MFMA.  One s_barrier per sub-step hands the stages over (W slab of the next sub-step landed, next K block activated).
//   * the MFMA operands are swapped (srcA = weight fragment, srcB = activation fragment): an accumulator lane then holds ONE position
//     and 4 consecutive channels per register quad, so the epilogue is cvt_pk -> v_permlane32_swap (lanes r and r + 32 hold the two
//     halves of 8 consecutive channels) -> one 16-byte global store per lane: no LDS transposition, no scratch -- the 16 KB it took
//     hold a THIRD weight stage instead (a slab is issued two sub-steps ahead of its use);
//   * the halo rows of a K block cost the producers one group of 4 elements on 32 lanes of one wave (the symmetric kernel ran two
//     groups on every lane of every wave for them: 20 % of its vector work);
//   * GroupNorm statistics of the stored values per lane (8 groups of 16 channels), reduced over the wave by a merging butterfly
//     (17 shuffles for 16 sums), fp64 atomics from 16 lanes.
// Same LDS image as adf_gemm_rb.h (ring of 3 activation stages of 64 channels with XOR-swizzled 128-byte rows, the swizzle applied on
// the DMA source side), same RbArgs, same shapes with n = 128 and 16-channel statistics groups (launch_rb routes them here).
#pragma once
#include "adf_gemm_rb.h"

namespace adf {

constexpr int kRsWStages = 3;
constexpr int kRsOffW = kPpAStages * kPpAStage;                 // 101,376
constexpr int kRsOffTab = kRsOffW + kRsWStages * kPpWStage;     // 150,528
constexpr int kRsOffBias = kRsOffTab + 2 * kPpTab;              // 158,720
#ifdef ADF_RB_TL
constexpr int kRsLds = kRsOffBias + kPpBias + 3072;
#else
constexpr int kRsLds = kRsOffBias + kPpBias;                    // 160,768 B
#endif

#define ADF_RS_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// s_waitcnt vmcnt(n) with a wave-uniform run-time n (the immediate must be a literal)
__device__ __forceinline__ void rs_wait_vm(int n) {
    switch (n) {
        ADF_RS_WAIT_CASE(0) ADF_RS_WAIT_CASE(1) ADF_RS_WAIT_CASE(2) ADF_RS_WAIT_CASE(3) ADF_RS_WAIT_CASE(4) ADF_RS_WAIT_CASE(5)
        ADF_RS_WAIT_CASE(6) ADF_RS_WAIT_CASE(7) ADF_RS_WAIT_CASE(8) ADF_RS_WAIT_CASE(9) ADF_RS_WAIT_CASE(10) ADF_RS_WAIT_CASE(11)
        ADF_RS_WAIT_CASE(12) ADF_RS_WAIT_CASE(13) ADF_RS_WAIT_CASE(14) ADF_RS_WAIT_CASE(15) ADF_RS_WAIT_CASE(16) ADF_RS_WAIT_CASE(17)
        ADF_RS_WAIT_CASE(18) ADF_RS_WAIT_CASE(19) ADF_RS_WAIT_CASE(20) ADF_RS_WAIT_CASE(21) ADF_RS_WAIT_CASE(22) ADF_RS_WAIT_CASE(23)
        ADF_RS_WAIT_CASE(24) ADF_RS_WAIT_CASE(25) ADF_RS_WAIT_CASE(26) ADF_RS_WAIT_CASE(27) ADF_RS_WAIT_CASE(28) ADF_RS_WAIT_CASE(29)
        ADF_RS_WAIT_CASE(30) ADF_RS_WAIT_CASE(31) ADF_RS_WAIT_CASE(32) ADF_RS_WAIT_CASE(33) ADF_RS_WAIT_CASE(34) ADF_RS_WAIT_CASE(35)
        ADF_RS_WAIT_CASE(36) ADF_RS_WAIT_CASE(37) ADF_RS_WAIT_CASE(38) ADF_RS_WAIT_CASE(39) ADF_RS_WAIT_CASE(40) ADF_RS_WAIT_CASE(41)
        ADF_RS_WAIT_CASE(42) ADF_RS_WAIT_CASE(43) ADF_RS_WAIT_CASE(44) ADF_RS_WAIT_CASE(45) ADF_RS_WAIT_CASE(46) ADF_RS_WAIT_CASE(47)
        default: if (n > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // n < 0: nothing to wait for
    }
}
#undef ADF_RS_WAIT_CASE

// The weight slab of the k-th sub-step counted from the head of the current block: blocks (current, next, one after) with t_c / t_1
// taps.  By VALUE: a walk over the block descriptors -- or a lambda capturing them by reference -- is compiled to a select of their
// addresses, which puts the descriptors into scratch memory.
__device__ __forceinline__ const char* rs_slab_at(int k, int t_c, int t_1, const char* w_c, const char* w_1, const char* w_2, unsigned slab) {
    const bool in0 = k < t_c, in1 = k < t_c + t_1;
    const unsigned long long a0 = (unsigned long long)w_c, a1 = (unsigned long long)w_1, a2 = (unsigned long long)w_2;
    const unsigned long long base = in0 ? a0 : (in1 ? a1 : a2);
    const int tap = in0 ? k : (in1 ? k - t_c : k - t_c - t_1);
    return (const char*)(base + (unsigned long long)((unsigned)tap * slab));
}

// RAW: segment 0 has no prologue (the folded strided convs of Downsample1d): the producers only move bytes and zero the padding rows.
template <bool RAW>
__global__ void __launch_bounds__(512) conv_gemm_rs_kernel(const RbArgs a) {
    typedef bf16_t T;
    constexpr int TM = 256, HP = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsTab = smem + kRsOffTab;
    float* const ldsBias = (float*)(smem + kRsOffBias);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave >= 4;                              // uniform: waves w and w + 4 share a SIMD.  The PRODUCERS are the older waves 0-3 (and run at
                                                                  // s_setprio 3): the SIMD arbitrates vector issue by priority, then age, and a younger producer beside an MFMA
                                                                  // stream got one issue slot in three (measured: ~1000 cycles per 8-element chunk, 190 per DMA instruction)
    const int wq = wave & 3;                                      // consumer: 64-row slice of the tile; producer: DMA / prologue lane of the pieces wq + 4 u
    const int r = lane & 31, h = lane >> 5;
    const int lrow = lane >> 3;
    // logical 16-byte chunk stored at this lane's slot of a piece p = wq + 4 u (rows 8 p + lrow): slot ^ ((row >> 1) & 7); p has the parity of wq
    const int chunk = (lane & 7) ^ ((((wq & 1) << 2) + (lane >> 4)) & 7);
    const int srow = wq * 8 + lrow;                               // staged row of this lane in piece u = 0 (+ 32 per u)
    const unsigned lane_lds = (unsigned)lane * 16u;
    const unsigned colbytes = (unsigned)chunk * 16u;

    const int nblk_grid = (int)gridDim.x, bidx = (int)blockIdx.x;
    const int t_lo = (int)((unsigned)bidx * (unsigned)a.tiles_total / (unsigned)nblk_grid);
    const int t_hi = (int)((unsigned)(bidx + 1) * (unsigned)a.tiles_total / (unsigned)nblk_grid);
    const int ntiles = t_hi - t_lo;
    if (ntiles <= 0) return;
    const int nb3 = a.nb3, nb1 = a.nb1, nb = nb3 + nb1;
    const int ctot0 = a.gn.c0 + a.gn.c1;
    const int tm_mask = (1 << a.tm_shift) - 1;

    struct Tile { int b0, m0; };
    auto tile_of = [&](int tseq) __attribute__((always_inline)) -> Tile {
        const int t = t_lo + tseq;                                // n = 128: one N tile
        Tile g;
        g.b0 = t >> a.tm_shift;
        g.m0 = (t & tm_mask) * TM;
        return g;
    };
    const int b_first = tile_of(0).b0;

    struct Blk {
        const char* abase;    // row p_lo of the tile in the block's source (+ channel offset)
        const char* w;        // weight slab of tap 0
        unsigned pitch;
        int tab;              // byte offset into ldsTab (slot included) or -1
        float scale;
        int edge;             // bit 0: staged row 0 is before the sample; bit 1: staged row TM + 1 is past it (3-tap blocks)
        int taps;
    };
    int d_t = 0, d_k = 0;                              // (tile, block) cursor of the descriptor stream
    Tile d_tile = tile_of(0);
    auto make_desc = [&]() __attribute__((always_inline)) -> Blk {
        const RbBlk& e = a.blk[d_k];
        Blk d;
        const int three = d_k < nb3;
        const int p_lo = d_tile.m0 - three;
        d.pitch = e.pitch;
        d.abase = e.src + (long long)(d_tile.b0 * a.L + p_lo) * (long long)e.pitch;
        d.w = e.w;
        d.tab = e.tab >= 0 ? ((d_tile.b0 - b_first) & 1) * kPpTab + e.tab : -1;
        d.scale = e.scale;
        d.taps = three ? 3 : 1;
        d.edge = three ? ((d_tile.m0 == 0 ? 1 : 0) | (d_tile.m0 + TM >= a.L ? 2 : 0)) : 0;
        return d;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++d_k == nb) {
            d_k = 0;
            if (++d_t < ntiles) d_tile = tile_of(d_t);    // past the end: a valid but unused descriptor
        }
    };
    const unsigned slab = (unsigned)a.n * (unsigned)kRowBytes;           // one tap of packed weights: 16 KB

#ifdef ADF_RB_TL
    const int tl_which = bidx == 0 ? 0 : (bidx == (int)gridDim.x / 2 + 3 ? 1 : -1);
    unsigned* const tl_lds = (unsigned*)(smem + kRsOffBias + 1024) + wave * 128;
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

    // =================================================== producer side ===================================================
    // activation pieces u0 .. u0 + 2 * npair - 1 of block d into ring stage offset `st` (bytes); piece p = wq + 4 u holds rows 8 p .. 8 p + 7
    auto issue_a = [&](const Blk& d, unsigned st, int u0, int npair) __attribute__((always_inline)) {
        for (int k = 0; k < npair; ++k) {
            const int u = u0 + 2 * k;
            const unsigned v = (unsigned)(srow + 32 * u) * d.pitch + colbytes;
            const unsigned v0 = (u == 0 && (d.edge & 1) && srow == 0) ? v + d.pitch : v;      // row -1 of the sample: fetch row 0, zeroed later
            const unsigned l = st + (unsigned)(wq + 4 * u) * 1024u;
            rb_dma2(d.abase, v0, v + 32u * d.pitch, l, l + 4096u);
        }
    };
    // rows TM, TM + 1 (3-tap blocks): lanes 0-15 of producer 0; past the end of the sample both rows fetch row TM (zeroed later)
    auto issue_halo = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        const unsigned vh = (unsigned)TM * d.pitch + ((d.edge & 2) ? 0u : (unsigned)lrow * d.pitch) + (unsigned)(lane & 7) * 16u;
        if (lane < 16) rb_dma1(d.abase, vh, st + (unsigned)HP * 1024u);
    };
    const unsigned wlane = (unsigned)srow * (unsigned)kRowBytes + colbytes;
    auto issue_w = [&](const char* wsrc, int wst) __attribute__((always_inline)) {
        const unsigned l = (unsigned)(kRsOffW + wst * kPpWStage) + (unsigned)wq * 1024u;
        rb_dma2(wsrc, wlane, wlane + 32u * (unsigned)kRowBytes, l, l + 4096u);
        rb_dma2(wsrc, wlane + 64u * (unsigned)kRowBytes, wlane + 96u * (unsigned)kRowBytes, l + 8192u, l + 12288u);
    };

    // ---- GroupNorm table of one sample (the arithmetic of gn_finalize_kernel / gn_affine<true>), channels c = t, t + 256 of 256 threads ----
    auto fill_table = [&](int b, int slot, int t) __attribute__((always_inline)) {
        int t1 = t;
        asm volatile("" : "+v"(t1));                  // (keeps the per-lane 64-bit addresses from being computed at kernel entry and held)
        const int c0_ = t1 < ctot0 ? t1 : ctot0 - 1, c1_ = t1 + 256 < ctot0 ? t1 + 256 : ctot0 - 1;
        const GnRaw r0 = gn_affine_load(a.gn, b, c0_), r1 = gn_affine_load(a.gn, b, c1_);
        float A0, B0, A1, B1;
        gn_affine_finish<true>(a.gn, c0_, r0, A0, B0);
        gn_affine_finish<true>(a.gn, c1_, r1, A1, B1);
        if (t1 < ctot0) *(f32x2_t*)(ldsTab + slot * kPpTab + c0_ * 8) = f32x2_t{A0, B0};
        if (t1 + 256 < ctot0) *(f32x2_t*)(ldsTab + slot * kPpTab + c1_ * 8) = f32x2_t{A1, B1};
    };

    // ---- prologue arithmetic: y = v * rcp(1 + exp2(ec * v + ed)) on v = a x + b; (ec, ed) = (-log2 e, 0) with SiLU, (0, -200) without
    // (exp2 underflows to 0 and y = v exactly: one code path for table blocks and raw, scaled blocks) ----
    auto act8 = [&](const u32x4_t& q, const float* fa, const float* fb, float ec, float ed) __attribute__((always_inline)) -> u32x4_t {
        float v[8], ex[8];
        unpack16<T>(q, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], fa[e], fb[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) ex[e] = __builtin_amdgcn_exp2f(fmaf(v[e], ec, ed));
#pragma unroll
        for (int e = 0; e < 8; ++e) ex[e] = __builtin_amdgcn_rcpf(ex[e] + 1.0f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= ex[e];
        return pack16<T>(v);
    };
    struct Tab { float fa[8], fb[8], ec, ed; bool work; };
    // (a, b) of the lane's 8 channels of block d (the chunk of a lane is the same in all its main pieces)
    auto load_tab = [&](const Blk& d, int ch0) __attribute__((always_inline)) -> Tab {
        Tab t;
        if (d.tab >= 0) {                                                 // uniform
            const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tab + ch0 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4_t x = tp[e];
                t.fa[2 * e] = x.x; t.fb[2 * e] = x.y; t.fa[2 * e + 1] = x.z; t.fb[2 * e + 1] = x.w;
            }
            t.ec = -1.4426950408889634f; t.ed = 0.0f; t.work = true;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { t.fa[e] = d.scale; t.fb[e] = 0.f; }
            t.ec = 0.0f; t.ed = -200.0f; t.work = d.scale != 1.0f;          // raw, unscaled: the bytes go to the MFMAs untouched
        }
        return t;
    };
    // pieces u0 .. u0 + N - 1 of the lane's column of pieces (piece wv + 4 u), in place: every chunk is read before the first one is
    // computed (one LDS round trip per call instead of one per chunk: a producer has nothing else to cover it with)
    auto transform_n = [&](auto nc, const Tab& t, unsigned st, int wv, int u0) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
        if (!t.work) return;
        char* const base = smem + st + wv * 1024 + lane_lds + u0 * 4096;
        u32x4_t q[N];
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = *(const u32x4_t*)(base + k * 4096);
#pragma unroll
        for (int k = 0; k < N; ++k) *(u32x4_t*)(base + k * 4096) = act8(q[k], t.fa, t.fb, t.ec, t.ed);
    };
    auto transform = [&](const Tab& t, unsigned st, int wv, int u0, int u1) __attribute__((always_inline)) {
        const int n = u1 - u0;                             // uniform: 2, 3, 4 or 8
        if (n == 3) transform_n(std::integral_constant<int, 3>{}, t, st, wv, u0);
        else if (n == 2) transform_n(std::integral_constant<int, 2>{}, t, st, wv, u0);
        else if (n == 4) transform_n(std::integral_constant<int, 4>{}, t, st, wv, u0);
        else { transform_n(std::integral_constant<int, 4>{}, t, st, wv, u0); transform_n(std::integral_constant<int, 4>{}, t, st, wv, u0 + 4); }
    };
    // the halo rows TM, TM + 1 of a 3-tap block: 16 chunks = 32 halves of 4 elements on lanes 0-31 of producer 0 (rows 256 / 257: swizzle 0)
    auto transform_halo = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (d.taps != 3) return;
        if (d.tab < 0 && d.scale == 1.0f) return;
        const int hc = lane & 15, hf = (lane >> 4) & 1;
        const int ch0 = (hc & 7) * 8 + hf * 4;
        float fa[4], fb[4], ec, ed;
        if (d.tab >= 0) {
            const f32x4_t* tp = (const f32x4_t*)(ldsTab + d.tab + ch0 * 8);
            const f32x4_t x0 = tp[0], x1 = tp[1];
            fa[0] = x0.x; fb[0] = x0.y; fa[1] = x0.z; fb[1] = x0.w; fa[2] = x1.x; fb[2] = x1.y; fa[3] = x1.z; fb[3] = x1.w;
            ec = -1.4426950408889634f; ed = 0.0f;
        } else {
            for (int e = 0; e < 4; ++e) { fa[e] = d.scale; fb[e] = 0.f; }
            ec = 0.0f; ed = -200.0f;
        }
        char* const p = smem + st + HP * 1024 + hc * 16 + hf * 8;
        const u32x2_t q = *(const u32x2_t*)p;
        float v[4], ex[4];
        v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
        v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], fa[e], fb[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(v[e], ec, ed)) + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= ex[e];
        if (lane < 32) *(u32x2_t*)p = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    };
    // zero padding of the activated tensor: staged row 0 / staged row TM + 1 of an edge tile (producer 0)
    auto zero_fill = [&](const Blk& d, unsigned st) __attribute__((always_inline)) {
        if (d.edge) {                                                     // uniform
            const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
            if ((d.edge & 1) && lane < 8) *(u32x4_t*)(smem + st + lane_lds) = z;
            if ((d.edge & 2) && lane >= 8 && lane < 16) *(u32x4_t*)(smem + st + HP * 1024 + lane_lds) = z;
        }
    };

    // =================================================== consumer side ===================================================
    f32x16_t acc[2][4];                                 // [32-row half][32-column tile]: lane = position r, register 4 g + e = channel 8 g + 4 h + e
    // fragment chunk (ks*2 + h) of staged row R sits at byte R*128 + (((ks*2 + h) ^ f) << 4), f = (R >> 1) & 7
    //   = (R*128 + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5)) ^ (ks << 5)
    unsigned abase0[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = wq * 64 + r + t, f = (row >> 1) & 7;
        abase0[t] = (unsigned)(row * kPpRow + ((h ^ (f & 1)) << 4) + ((f >> 1) << 5));
    }
    const int fw = (r >> 1) & 7;
    const unsigned wbase0 = (unsigned)(r * kPpRow + ((h ^ (fw & 1)) << 4) + ((fw >> 1) << 5));

    // one sub-step: the 32 MFMAs of tap TAP over the 64 channels of the block in A stage offset sa_ with the slab in W stage offset sw_.
    // Registers: 128 accumulators leave room for 32 of fragments, not for two full K steps of them (a first version with both operands
    // double-buffered spilled inside this loop): the two activation fragments of a K step serve all four column tiles and are
    // double-buffered, the weight fragment of column tile j is dead after its two MFMAs and its slot is reloaded for the next K
    // step right behind them -- eight MFMAs (256 cycles) ahead of its use.
    auto substep = [&](auto tapc, unsigned sa_, unsigned sw_) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        const unsigned a0 = sa_ + abase0[TAP], w0 = sw_ + wbase0;
        bf16x8_t fa[2][2], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8_t*)(smem + w0 + j * 32 * kPpRow);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *(const bf16x8_t*)(smem + a0 + i * 32 * kPpRow);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const unsigned a1 = a0 ^ (unsigned)((ks + 1) << 5), w1 = w0 ^ (unsigned)((ks + 1) << 5);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[cur][i], acc[i][j], 0, 0, 0);
                if (ks + 1 < 4) {
                    fb[j] = *(const bf16x8_t*)(smem + w1 + j * 32 * kPpRow);
                    if (j == 0) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) fa[nxt][i] = *(const bf16x8_t*)(smem + a1 + i * 32 * kPpRow);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = *(const f32x4_t*)(ldsBias + 32 * j + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 2; ++i) { acc[i][j][4 * g] = b4.x; acc[i][j][4 * g + 1] = b4.y; acc[i][j][4 * g + 2] = b4.z; acc[i][j][4 * g + 3] = b4.w; }
            }
    };

    // ---- epilogue of one finished tile: this wave's 64 rows x 128 channels straight from the accumulators ----
    auto epilogue_t = [&](const Tile& g, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr bool has_res = decltype(resc)::value, stats_here = decltype(statc)::value;      // (compile-time: as run-time flags every quad sat behind two branches)
        T* const out = (T*)a.out;
        const T* const resp = (const T*)a.res;
#ifdef ADF_RB_TL
        tl(110);
#endif
        float s1[8], s2[8];                                                     // per 16-channel group: sum, sum of squares of the STORED values
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = g.m0 + wq * 64 + i * 32 + r;
            const unsigned rowoff = (unsigned)((g.b0 * a.L + m) * a.n);         // elements (tensors < 4 GiB)
            u32x2_t res[2][4];                                                 // identity residual: the quads of column tile j + 1 are fetched while tile j is packed
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) res[0][q] = *(const u32x2_t*)(resp + rowoff + 8 * q + 4 * h);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (has_res && j + 1 < 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) res[(j + 1) & 1][q] = *(const u32x2_t*)(resp + rowoff + 32 * (j + 1) + 8 * q + 4 * h);
                }
                u32x2_t pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
                    if (has_res) {
                        v0 += __uint_as_float(res[j & 1][q].x << 16); v1 += __uint_as_float(res[j & 1][q].x & 0xffff0000u);
                        v2 += __uint_as_float(res[j & 1][q].y << 16); v3 += __uint_as_float(res[j & 1][q].y & 0xffff0000u);
                    }
                    pk[q].x = pack_bf16x2(v0, v1);
                    pk[q].y = pack_bf16x2(v2, v3);
                    if (stats_here) {
                        // sum and sum of squares of the STORED values, two per instruction: v_dot2c_f32_bf16 (products of bf16 pairs, fp32 accumulate)
                        typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
                        const bf2_t one2 = __builtin_bit_cast(bf2_t, 0x3f803f80u);
                        const bf2_t p0 = __builtin_bit_cast(bf2_t, pk[q].x), p1 = __builtin_bit_cast(bf2_t, pk[q].y);
                        const int k = 2 * j + (q >> 1);
                        s1[k] = __builtin_amdgcn_fdot2_f32_bf16(p0, one2, s1[k], false);
                        s1[k] = __builtin_amdgcn_fdot2_f32_bf16(p1, one2, s1[k], false);
                        s2[k] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, s2[k], false);
                        s2[k] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, s2[k], false);
                    }
                }
                // lanes r (h = 0) and r + 32 (h = 1) hold channels 8 q + 0..3 and 8 q + 4..7 of position r: after the swap the lower lane has
                // the 8 consecutive channels of q = q0, the upper lane those of q = q0 + 1 (v_permlane32_swap: vdst[32..63] <-> src[0..31])
#pragma unroll
                for (int q0 = 0; q0 < 4; q0 += 2) {
                    const u32x2_t sx = __builtin_amdgcn_permlane32_swap(pk[q0].x, pk[q0 + 1].x, false, false);
                    const u32x2_t sy = __builtin_amdgcn_permlane32_swap(pk[q0].y, pk[q0 + 1].y, false, false);
                    *(u32x4_t*)(out + rowoff + 32 * j + 8 * (q0 + h)) = u32x4_t{sx.x, sy.x, sx.y, sy.y};
                }
            }
        }
#ifdef ADF_RB_TL
        tl(111);
#endif
        if (stats_here) {
            // 16 sums over 64 lanes without LDS: v_permlane32_swap / v_permlane16_swap ARE the merging butterfly steps of lane bits 5 and 4
            // (swap(a, b): the upper half of a <-> the lower half of b, so a + b is the pair sum of `a` in the lower lanes and of `b` in the
            // upper ones), then four DPP adds sum a value over the 16 lanes of a row
            float x8[8], x4[4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1[k]), __float_as_uint(s2[k]), false, false);
                x8[k] = __uint_as_float(sw.x) + __uint_as_float(sw.y);          // lanes 0-31: sum (which = 0), lanes 32-63: sum of squares, of group k
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x8[2 * k]), __float_as_uint(x8[2 * k + 1]), false, false);
                x4[k] = __uint_as_float(sw.x) + __uint_as_float(sw.y);          // even rows of 16 lanes: group 2 k, odd rows: group 2 k + 1
            }
            auto row_sum = [&](float v) __attribute__((always_inline)) -> float {
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, false));    // row_half_mirror
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, false));    // row_mirror
                return v;
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) x4[k] = row_sum(x4[k]);
            const bool b5 = lane & 32, b4 = lane & 16;
            // every lane of row (b5, b4) holds the four totals of (which = b5, groups 2 k + b4): lanes 0-3 of the row add one each
            const int kk = lane & 3;
            const float x1 = kk == 0 ? x4[0] : (kk == 1 ? x4[1] : (kk == 2 ? x4[2] : x4[3]));
            if ((lane & 15) < 4) {
                const int grp = 2 * kk + (b4 ? 1 : 0);
                atomicAdd(a.stats + ((size_t)g.b0 * a.stats_groups + grp) * 2 + (b5 ? 1 : 0), (double)x1);
            }
        }
#ifdef ADF_RB_TL
        tl(112);
#endif
    };

    auto epilogue = [&](const Tile& g) __attribute__((always_inline)) {
        if (a.res != nullptr) {
            if (a.stats != nullptr) epilogue_t(g, std::true_type{}, std::true_type{}); else epilogue_t(g, std::true_type{}, std::false_type{});
        } else {
            if (a.stats != nullptr) epilogue_t(g, std::false_type{}, std::true_type{}); else epilogue_t(g, std::false_type{}, std::false_type{});
        }
    };

    // =================================================== start-up ===================================================
    Blk dc = make_desc();
    advance();
    Blk d1 = make_desc();
    advance();
    Blk d2 = make_desc();
    advance();
    Blk d3 = make_desc();                              // one block further than anybody reads: its scalar loads of the kernel arguments have
    advance();                                         // a whole block to come back (at the head of a block they cost ~1000 cycles)
    const int total_blocks = ntiles * nb;
    // the weight slab of the k-th sub-step from the head of block dc (k = 0 ..): walks dc, d1, d2 by their tap counts
    // (selected as VALUES from scalar copies of the three slab pointers and tap counts: a walk over the descriptors themselves is
    //  compiled to a select of their addresses, which puts all three structs into scratch memory)
    const char *w_c = dc.w, *w_1 = d1.w, *w_2 = d2.w;
    int t_c = dc.taps, t_1 = d1.taps;
#define slab_at(k) rs_slab_at((k), t_c, t_1, w_c, w_1, w_2, slab)
    // number of sub-steps left from the head of block dc (caps the slab prefetch at the end of the launch)
    int blocks_left = total_blocks;
    auto substeps_left_ge = [&](int k) __attribute__((always_inline)) -> bool {
        int n = dc.taps;
        if (blocks_left > 1) n += d1.taps;
        if (blocks_left > 2) n += d2.taps;
        return k < n;
    };

    // How many DMA instructions this producer wave has issued AFTER ...  (s_waitcnt vmcnt(n) = "all but the n youngest have landed";
    // the counters say how young a piece may be: everything issued behind it may stay in flight)
    int y_a1 = 0;          // ... the last activation piece of block g + 1 (d1): its prologue starts at the head of block g
    int y_a2 = 0;          // ... the last activation piece of block g + 2 (d2)
    auto issued_n = [&](int n) __attribute__((always_inline)) { y_a1 += n; y_a2 += n; };
    if (!consumer) {
        // the activation pieces of blocks 0 and 1 (they need kernel arguments only), then the wait for block 0
        issue_a(dc, 0u, 0, 4);
        if (wq == 0 && dc.taps == 3) issue_halo(dc, 0u);
        int after0 = 0;                                                       // DMA instructions issued after block 0's activations
        if (blocks_left > 1) {
            issue_a(d1, (unsigned)kPpAStage, 0, 4);
            after0 += 8;
            if (wq == 0 && d1.taps == 3) { issue_halo(d1, (unsigned)kPpAStage); after0 += 1; }
            y_a1 = 0;
        }
        rs_wait_vm(after0);
    } else {
        // The consumers issue the weight slabs -- here the first two, in the pipeline the slab of sub-step s + 2 behind the MFMAs of
        // sub-step s: 4 of the 7 DMA instructions a producer issued per sub-step in the first version, ~85 cycles each in its stream.
        issue_w(slab_at(0), 0);
        if (substeps_left_ge(1)) issue_w(slab_at(1), 1);
        // ... and have nothing to multiply yet: they derive the first sample's table and the bias vector meanwhile
        const int t = tid - 256;                                              // 0 .. 255
        const bool hb0 = a.bias0 != nullptr, hb1 = a.bias1 != nullptr;        // uniform
        const float* const dummy_f = (const float*)a.blk[0].w;                // always there, >= 1 KB
        const int bi = t < a.n ? t : 0;
        const float b0v = (hb0 ? a.bias0 : dummy_f)[bi], b1v = (hb1 ? a.bias1 : dummy_f)[bi];
        if constexpr (!RAW) fill_table(b_first, 0, t);
        if (t < a.n) ldsBias[t] = (hb0 ? b0v : 0.f) + (hb1 ? b1v : 0.f);
    }
    tl(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // block 0's activations have landed, the table is there
    tl(2);
    // block 0 is activated by all eight waves: wave w takes pieces (w & 3) + 4 u of the producer column it shares, u = 0-3 (producers) / 4-7 (consumers)
    if constexpr (!RAW) {
        const Tab t0 = load_tab(dc, chunk * 8);
        transform(t0, 0u, wq, consumer ? 4 : 0, consumer ? 8 : 4);
        if (wave == 0) transform_halo(dc, 0u);
    }
    if (wave == 0) zero_fill(dc, 0u);                  // (row 0 and the halo rows were activated by this wave: LDS operations of one wave stay in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (consumer) {
        init_acc();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // both start-up slabs have landed before anybody multiplies
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    tl(3);

    // =================================================== pipeline ===================================================
    // K blocks are numbered over the whole thread block: block g lives in A stage g % 3 (fetched while block g - 2 is multiplied,
    // activated in place while block g - 1 is), the slab of sub-step s in W stage s % 3 (issued during sub-step s - 2).
    unsigned sa = 0u, sa1 = (unsigned)kPpAStage, sa2 = 2u * (unsigned)kPpAStage;       // stage byte offsets of blocks g, g + 1, g + 2
    int ws = 0;                                                                        // W stage of the current sub-step
    const std::integral_constant<int, 0> c0{};
    const std::integral_constant<int, 1> c1{};
    const std::integral_constant<int, 2> c2{};
    auto rotate = [&]() __attribute__((always_inline)) {
        const unsigned t = sa; sa = sa1; sa1 = sa2; sa2 = t;
        dc = d1; d1 = d2; d2 = d3; d3 = make_desc();
        advance();
        --blocks_left;
        w_c = w_1; w_1 = w_2; w_2 = d2.w;
        t_c = t_1; t_1 = d1.taps;
        y_a1 = y_a2;
    };
    Tile cur_tile = tile_of(0);

    // the producer's share of sub-step u of block dc
    Tab tab1;                                              // (a, b) of this lane's 8 channels in block g + 1: loaded at the head of block g
    tab1.work = false;
    auto produce = [&](int u, int tseq, int kb) __attribute__((always_inline)) {
        // activations of block g + 2, spread over the first two sub-steps of a 3-tap block
        if (blocks_left > 2) {
            if (dc.taps == 3) {
                if (u == 0) {
                    issue_a(d2, sa2, 0, 2); issued_n(4);
                    if (wq == 0 && d2.taps == 3) { issue_halo(d2, sa2); issued_n(1); }
                } else if (u == 1) { issue_a(d2, sa2, 4, 2); issued_n(4); y_a2 = 0; }
            } else {
                issue_a(d2, sa2, 0, 4); issued_n(8);
                if (wq == 0 && d2.taps == 3) { issue_halo(d2, sa2); issued_n(1); }
                y_a2 = 0;
            }
        }
#ifdef ADF_RB_TL
        const bool tlp = tseq == 1 && kb == 1;
        if (tlp) tl(110 + 3 * u);
#endif
        if (blocks_left > 1) {
            if (u == 0) rs_wait_vm(y_a1);                  // every piece of block g + 1 has landed
            if constexpr (!RAW) {
                if (u == 0) tab1 = load_tab(d1, chunk * 8);
                if (dc.taps == 3) {
                    // (chunks 6 and 7 of the column are activated by the consumer of this SIMD behind its MFMAs of sub-steps 1 and 2)
                    if (u == 0) transform(tab1, sa1, wq, 0, 2);
                    else if (u == 1) transform(tab1, sa1, wq, 2, 4);
                    else { transform(tab1, sa1, wq, 4, 6); if (wq == 0) transform_halo(d1, sa1); }
                } else {
                    transform(tab1, sa1, wq, 0, 8);
                    if (wq == 0) transform_halo(d1, sa1);
                }
            }
            if (u == dc.taps - 1 && wq == 0) zero_fill(d1, sa1);      // (behind this wave's own stores to those rows: LDS operations of a wave stay in order)
        }
        if constexpr (!RAW) {
            // first block of a tile: the table of the next tile's sample, if it is another one (rare: once per sample)
            if (u == dc.taps - 1 && kb == 0 && tseq + 1 < ntiles) {
                const Tile nt = tile_of(tseq + 1);
                if (nt.b0 != cur_tile.b0) {
                    fill_table(nt.b0, (nt.b0 - b_first) & 1, tid);
                    // The table loads are the only vector-memory loads the COMPILER sees in this loop, and some of them (optional tensors read
                    // through a dummy pointer) are never used: its wait bookkeeping kept them "pending" and protected the registers they
                    // target with s_waitcnt vmcnt(0 / 1) wherever those were reused -- inside the prologue loop of every later sub-step,
                    // i.e. a drain of the activation pieces just issued (the ISA of the first version; ~1000 cycles per 16-byte chunk).
                    // A wait it can see, here, once per sample: nothing of its own is pending afterwards.
                    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), expcnt / lgkmcnt untouched
                }
            }
        }
#ifdef ADF_RB_TL
        if (tlp) { tl(111 + 3 * u); tl(112 + 3 * u); }
#endif
    };
    // the consumer's share behind its MFMAs: the slab of sub-step s + 2 goes out, the slab of sub-step s + 1 (issued one sub-step ago) has landed
    auto consume_tail = [&](int u) __attribute__((always_inline)) {
        if constexpr (!RAW) {
            // The consumer is done with the matrix pipe for this sub-step while its producer still has ~2 chunks to go: it takes one chunk
            // of block g + 1 in sub-steps 1 and 2 of a 3-tap block (every piece of that block has landed and been waited for by the
            // producers before the barrier that opened sub-step 1)
            if (dc.taps == 3 && u > 0 && blocks_left > 1) {
                const Tab tabc = load_tab(d1, chunk * 8);      // (read again in each of the two sub-steps: 16 registers that must not live across the MFMAs)
                transform_n(std::integral_constant<int, 1>{}, tabc, sa1, wq, 5 + u);
            }
        }
        const int wnext = ws + 2 >= 3 ? ws - 1 : ws + 2;
        if (substeps_left_ge(u + 2)) {
            issue_w(slab_at(u + 2), wnext);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    // The two roles walk the same (tile, block, sub-step) sequence in SEPARATE loops -- one s_barrier per sub-step in both -- so that
    // the accumulators are live in the consumers' code only (one loop with a role test inside spilled 232 registers).  Written out
    // twice through a macro: as a generic lambda the block descriptors stayed in scratch memory.
#ifdef ADF_RB_TL
#define ADF_RS_TL(x) x
#else
#define ADF_RS_TL(x)
#endif
    // (3-tap blocks and 1-tap blocks in two loops, the sub-steps of a block in straight line: an if / else around copies of a sub-step
    //  makes the register allocator move the accumulator tiles through scratch)
#define ADF_RS_SUB(U, WORK)                                                                          \
    {                                                                                                \
        constexpr int u = U;                                                                         \
        (void)u;                                                                                     \
        WORK                                                                                         \
        ADF_RS_TL(tl(4 + tseq * 26 + 2 + 2 * tl_sub);)                                               \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                              \
        ADF_RS_TL(tl(4 + tseq * 26 + 3 + 2 * tl_sub); ++tl_sub;)                                     \
        ws = ws == 2 ? 0 : ws + 1;                                                                   \
    }
#define ADF_RS_LOOP(WORK_TILE, WORK0, WORK1, WORK2, WORK_END)                                         \
    for (int tseq = 0; tseq <= ntiles; ++tseq) {                                                     \
        ADF_RS_TL(tl_sub = 0; tl(tseq < ntiles ? 4 + tseq * 26 : 108);)                              \
        WORK_TILE                                                                                    \
        ADF_RS_TL(tl(tseq < ntiles ? 4 + tseq * 26 + 1 : 109);)                                      \
        if (tseq == ntiles) break;                                                                   \
        cur_tile = tile_of(tseq);                                                                    \
        for (int kb = 0; kb < nb3; ++kb) {                                                           \
            ADF_RS_SUB(0, WORK0) ADF_RS_SUB(1, WORK1) ADF_RS_SUB(2, WORK2)                           \
            rotate();                                                                                \
        }                                                                                            \
        for (int kb = nb3; kb < nb; ++kb) {                                                          \
            ADF_RS_SUB(0, WORK0)                                                                     \
            rotate();                                                                                \
        }                                                                                            \
    }                                                                                                \
    WORK_END
    if (!consumer) __builtin_amdgcn_s_setprio(3);
    if (consumer) {
        ADF_RS_LOOP(if (tseq > 0) { epilogue(tile_of(tseq - 1)); if (tseq < ntiles) init_acc(); },
                    substep(c0, sa, (unsigned)(kRsOffW + ws * kPpWStage)); consume_tail(0);,
                    substep(c1, sa, (unsigned)(kRsOffW + ws * kPpWStage)); consume_tail(1);,
                    substep(c2, sa, (unsigned)(kRsOffW + ws * kPpWStage)); consume_tail(2);, )
    } else {
        ADF_RS_LOOP(, produce(u, tseq, kb);, produce(u, tseq, kb);, produce(u, tseq, kb);, )
    }
#undef ADF_RS_SUB
#undef slab_at
#undef ADF_RS_LOOP
#undef ADF_RS_TL
#ifdef ADF_RB_TL
    tl_real(122);
    if (tl_which >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 128; i += 64) adf_rb_tl[(tl_which * 8 + wave) * 128 + i] = tl_lds[i];
    }
#endif
}

}  // namespace adf
