// 2-D kernels of the ADM-style U-Net (BASELINE config 4).
// Reference: src/models/backbones/unet2d_oai.py (GroupNorm32 :10-21, timestep_embedding :31-49, Upsample :102-127, Downsample
// :130-158, ResBlock :162-272, AttentionBlock :274-322, UNetModel :382-635).
//
// conv2d_gemm_kernel: implicit GEMM, 128 (or 64) pixels x 128 output channels per workgroup of 8 (4) waves, K walked as (channel
// chunk of 128 bytes) x (tap); the activation chunk of a tap is GATHERED from the channels-last input with the tap's bounds test
// (zero padding), the GroupNorm (+ scale-shift) affine and SiLU applied on the way into LDS; nearest-x2 upsampling and stride 2 are
// index maps of the gather, so Upsample / Downsample cost no extra pass.  Register prefetch one iteration ahead into the other of
// two LDS stages (one barrier per iteration); the output tile leaves through LDS as 16-byte pieces with the residual added there.
// This is the first correct path for this row (fp32 parity + a bf16 bench line); it is not yet tuned like the 1-D resblock kernels.
#include "adf_conv2d.h"
#include <type_traits>

#ifdef ADF_C2_STAMP
// diagnostic build (tools/build_variant.sh c2stamp -DADF_C2_STAMP; tools/c2_stamps.py): s_memtime at the phase boundaries of the spatial-tile
// kernel, all waves of one workgroup in the middle of the grid
namespace adf { __device__ unsigned long long adf_c2_stamps[8 * 16]; }
extern "C" int adf_debug_c2_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(adf::adf_c2_stamps), sizeof(unsigned long long) * 8 * 16);
}
#define C2_STAMP(i) do { if (stamped && lane == 0) adf_c2_stamps[wave * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define C2_STAMP(i) do { } while (0)
#endif

// output stores and residual loads are pure streams; the weights (re-read by every tile) and the halos (shared by neighbouring tiles) are what L2 should keep
#ifndef ADF_C2_NT
#define ADF_C2_NT 1
#endif
#if ADF_C2_NT
#define C2_NT_LOAD(p) __builtin_nontemporal_load(p)
#define C2_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define C2_NT_LOAD(p) (*(p))
#define C2_NT_STORE(v, p) (*(p) = (v))
#endif

namespace adf {

#define C2_LAUNCH_CHECK(name) (hipGetLastError() == hipSuccess ? nullptr : "launch failed: " name)

typedef __attribute__((ext_vector_type(8))) __bf16 c2_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float c2_f32x16_t;
typedef float c2_f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kC2Pitch = 144;       // LDS row pitch: 128 bytes of K + 16 (conflict-free 16-byte fragment reads, adf_gemm.h)

// Tail of both conv kernels: the output tile sits in LDS ([TM][PO], bias added); every thread walks 16-byte pieces of ONE piece column
// (NT is a multiple of the pieces per row), adds the residual, stores, and -- if asked -- reduces the GroupNorm statistics of what it
// stored: per-thread partial sums -> lanes of a wave that share the piece column (shuffles) -> per-channel sums in LDS -> per-group fp64
// atomics.  pix(row) maps a tile row to its pixel's element offset / cout inside the image.
// The residual pieces a thread adds in c2_store_tile, loaded ahead of the accumulator -> LDS phase so that their latency hides behind it.
template <typename T, int TM, int TN, int NT>
struct C2Res { u32x4_t v[TM * (TN / Elem<T>::kPerChunk) / NT]; };
template <typename T, int TM, int TN, int NT, typename FP>
__device__ __forceinline__ void c2_res_prefetch(const Conv2dArgs& a, int tid, int n0, const T* rg, FP pix, C2Res<T, TM, TN, NT>& r) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int PPR = TN / EPC;
    const int pc = tid % PPR;
    const int col = n0 + pc * EPC;
#pragma unroll
    for (int k = 0; k < TM * PPR / NT; ++k) {
        const int row = tid / PPR + k * (NT / PPR);
        r.v[k] = (rg && col < a.cout) ? C2_NT_LOAD((const u32x4_t*)(rg + pix(row) * a.cout + col)) : u32x4_t{0u, 0u, 0u, 0u};
    }
}

template <typename T, int TM, int TN, int NT, typename FP>
__device__ __forceinline__ void c2_store_tile(const Conv2dArgs& a, char* lds, int PO, int tid, int n0, int b, T* og, const T* rg, FP pix,
                                              const C2Res<T, TM, TN, NT>& pre) {
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int PPR = TN / EPC;
    static_assert(NT % PPR == 0 && PPR <= 64 && 64 % PPR == 0, "a thread keeps one piece column");
    const int pc = tid % PPR;
    const int col = n0 + pc * EPC;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const bool pairs = sizeof(T) == 2 && a.stats && a.stats_groups > 0 && ((a.cout / a.stats_groups) & 1) == 0;       // uniform
    if (col < a.cout) {
#pragma unroll
        for (int k = 0; k < TM * PPR / NT; ++k) {
            const int row = tid / PPR + k * (NT / PPR);
            u32x4_t v = *(const u32x4_t*)(lds + row * PO + pc * 16);
            const size_t o = pix(row) * a.cout + col;
            float f[EPC];
            unpack16<T>(v, f);
            if (rg) {
                float g[EPC];
                unpack16<T>(pre.v[k], g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] += g[e];
            }
            if constexpr (sizeof(T) == 2) {
                if (pairs) {
                    // statistics of the stored values from the packed pairs (v_dot2c_f32_bf16: exact products, fp32 accumulate): the pair sum goes to the
                    // even channel's slot, the odd one stays 0 -- every statistics group holds whole pairs (even group size)
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
                    v = pack16<T>(f);
                    C2_NT_STORE(v, (u32x4_t*)(og + o));
                    const bf2_t one2 = __builtin_bit_cast(bf2_t, 0x3f803f80u);
                    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf2_t q2 = __builtin_bit_cast(bf2_t, w4[e]);
                        s1[2 * e] = __builtin_amdgcn_fdot2_f32_bf16(q2, one2, s1[2 * e], false);
                        s2[2 * e] = __builtin_amdgcn_fdot2_f32_bf16(q2, q2, s2[2 * e], false);
                    }
                    continue;
                }
            }
            v = pack16_stored<T>(f);
            C2_NT_STORE(v, (u32x4_t*)(og + o));
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += f[e]; s2[e] = fmaf(f[e], f[e], s2[e]); }
        }
    }
    if (!a.stats) return;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        if (pairs && (e & 1)) continue;                // (pair sums live in the even slots)
#pragma unroll
        for (int o = PPR; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    __syncthreads();                                   // every thread has read its rows of the tile: LDS is free
    float* const c1 = (float*)lds;                     // [TN] sums, [TN] sums of squares
    float* const c2 = c1 + TN;
    for (int i = tid; i < 2 * TN; i += NT) c1[i] = 0.f;
    __syncthreads();
    if ((tid & 63) < PPR) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { if (pairs && (e & 1)) continue; atomicAdd(&c1[pc * EPC + e], s1[e]); atomicAdd(&c2[pc * EPC + e], s2[e]); }
    }
    __syncthreads();
    const int gs = a.cout / a.stats_groups;
    if (tid < TN / gs && n0 + tid * gs < a.cout) {
        double d1 = 0.0, d2 = 0.0;
        for (int c = tid * gs; c < (tid + 1) * gs; ++c) { d1 += (double)c1[c]; d2 += (double)c2[c]; }
        double* const st = a.stats + ((size_t)b * a.stats_groups + (n0 + tid * gs) / gs) * 2;
        atomicAdd(st, d1);
        atomicAdd(st + 1, d2);
    }
}

template <typename T, int TM>
__global__ void __launch_bounds__(TM * 4) conv2d_gemm_kernel(const Conv2dArgs a) {
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int KC = 128 / (int)sizeof(T);           // channels per K chunk
    constexpr int TN = 128;
    constexpr int NT = TM * 4;                         // threads: TM / 32 wave rows x 2 wave columns, a wave owns 32 pixels x 64 channels
    constexpr int STAGE = (TM + TN) * kC2Pitch;        // one LDS stage: activations, then weights
    constexpr int PO = TN * (int)sizeof(T) + 16;       // row pitch of the output tile in LDS
    static_assert(TM * PO <= 2 * STAGE, "the output tile reuses the two stages");
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    // the (a, b) prologue table of this tile's image: read from global it costs four 16-byte loads per staged activation piece
    // (4x the bytes of the activations themselves, 57 % of the first version's time went there); a tile lies inside one image
    __shared__ __attribute__((aligned(16))) float abs_[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int HW = a.H * a.W;
    const unsigned ltile = adf_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);      // (pixel tile, N tile), N tile fastest, XCD order
    const long long m0 = (long long)(ltile / gridDim.y) * TM;
    const int n0 = (int)(ltile % gridDim.y) * TN;
    const int Hin = a.mode == 1 ? a.H / 2 : (a.mode == 2 ? a.H * 2 : a.H);
    const int Win = a.mode == 1 ? a.W / 2 : (a.mode == 2 ? a.W * 2 : a.W);

    // staging: TM * 8 activation pieces and TN * 8 weight pieces of 16 bytes per iteration; thread -> (row tid >> 3 (+ NT / 8 ...), piece tid & 7)
    constexpr int RPT = NT / 8;                        // rows covered by one sweep of the workgroup
    constexpr int NA = TM / RPT, NW = TN / RPT;        // sweeps: 2 for the activations, 2 (TM = 128) or 4 (TM = 64) for the weights
    const int c16 = tid & 7, srow = tid >> 3;
    int pb[NA], py[NA], px[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const long long p = m0 + srow + RPT * k;
        pb[k] = (int)(p / HW);
        const int pp = (int)(p - (long long)pb[k] * HW);
        py[k] = pp / a.W; px[k] = pp - py[k] * a.W;
    }
    const T* const xg = (const T*)a.x;
    const char* const wg = (const char*)a.w;
    const int nit = a.nchunk * a.taps;
    if (a.ab) {
        const float* const abg = a.ab + (size_t)(m0 / HW) * a.cin * 2;
        // 16-byte pieces: 1024 channels are one trip of the workgroup (float by float they were four load -> store trips, each a memory round trip)
        for (int i = tid; i < a.cin / 2; i += NT) ((u32x4_t*)abs_)[i] = ((const u32x4_t*)abg)[i];
    }

    u32x4_t ra[NA], rw[NW];
    bool va[NA], wok[NW];                              // weight rows past n_pad are zeroed when the slab is stored (a select behind a load is a wait behind it)
    unsigned woff[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int row = srow + RPT * k;
        wok[k] = n0 + row < a.n_pad;
        woff[k] = (unsigned)((wok[k] ? n0 + row : 0) * 128 + c16 * 16);
    }
    auto load_regs = [&](int it) __attribute__((always_inline)) {
        const int ck = it / a.taps, tap = it - ck * a.taps;
        const int dy = a.taps == 9 ? tap / 3 : 1, dx = a.taps == 9 ? tap - (tap / 3) * 3 : 1;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            int sy, sx;
            bool ok;
            if (a.mode == 2) {
                sy = 2 * py[k] + dy - 1; sx = 2 * px[k] + dx - 1;
                ok = sy >= 0 && sy < Hin && sx >= 0 && sx < Win;
            } else {
                const int oy = py[k] + dy - 1, ox = px[k] + dx - 1;
                ok = oy >= 0 && oy < a.H && ox >= 0 && ox < a.W;
                sy = a.mode == 1 ? oy >> 1 : oy; sx = a.mode == 1 ? ox >> 1 : ox;
            }
            va[k] = ok;
            const bool s1 = ck * KC >= a.c0;                              // uniform: this chunk comes from the second source
            const int cs = s1 ? a.cin - a.c0 : a.c0, cb = s1 ? ck * KC - a.c0 : ck * KC;
            const size_t off = ok ? (((size_t)pb[k] * Hin + sy) * Win + sx) * cs + (size_t)cb + c16 * EPC : 0;
            ra[k] = *(const u32x4_t*)((s1 ? (const T*)a.x1 : xg) + off);
        }
        const char* const wbase = wg + (size_t)it * a.n_pad * 128;          // packed order [chunk][tap]
#pragma unroll
        for (int k = 0; k < NW; ++k) rw[k] = *(const u32x4_t*)(wbase + woff[k]);
    };
    auto store_lds = [&](int it) __attribute__((always_inline)) {
        const int ck = it / a.taps;
        char* const st = lds + (it & 1) * STAGE;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            u32x4_t v = ra[k];
            if (!va[k]) v = u32x4_t{0u, 0u, 0u, 0u};                     // zero padding, applied after the activation
            else if (a.ab) {
                float f[EPC];
                unpack16<T>(v, f);
                const float* const ab = abs_ + (ck * KC + c16 * EPC) * 2;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float t = fmaf(f[e], ab[2 * e], ab[2 * e + 1]);
                    f[e] = a.act ? (kBf16 ? silu_f(t) : t / (1.0f + expf(-t))) : t;
                }
                v = pack16<T>(f);
            }
            *(u32x4_t*)(st + (srow + RPT * k) * kC2Pitch + c16 * 16) = v;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) *(u32x4_t*)(st + (TM + srow + RPT * k) * kC2Pitch + c16 * 16) = wok[k] ? rw[k] : u32x4_t{0u, 0u, 0u, 0u};
    };

    c2_f32x16_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    // one barrier per iteration: while iteration `it` computes from stage it & 1, the registers of iteration it + 1 (loaded before
    // the MFMAs) are written to the other stage afterwards.  The loads are issued on every path (the last iteration fetches its own operands
    // again into the stage nobody reads): under a condition the compiler waits for them where they are issued (see conv2d_tile_kernel)
    load_regs(0);
    __syncthreads();                                   // the prologue table is in LDS
    store_lds(0);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        load_regs(it + 1 < nit ? it + 1 : it);
        const char* const aRow = lds + (it & 1) * STAGE + (wm * 32 + r) * kC2Pitch;
        const char* const wRow = lds + (it & 1) * STAGE + (TM + wn * 64 + r) * kC2Pitch;
        if constexpr (kBf16) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const c2_bf16x8_t fa = *(const c2_bf16x8_t*)(aRow + (ks * 2 + h) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const c2_bf16x8_t fb = *(const c2_bf16x8_t*)(wRow + j * 32 * kC2Pitch + (ks * 2 + h) * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                c2_f32x4_t fa[2], fb[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) fa[u] = *(const c2_f32x4_t*)(aRow + (ks * 4 + 2 * h + u) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) fb[j][u] = *(const c2_f32x4_t*)(wRow + j * 32 * kC2Pitch + (ks * 4 + 2 * h + u) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][e], fb[j][u][e], acc[j], 0, 0, 0);
            }
        }
        store_lds(it + 1 < nit ? it + 1 : it);
        __syncthreads();
    }

    // ---- epilogue: accumulators (+ bias) -> LDS tile -> 16-byte pieces (+ residual) -> global ---------------------------------
    // (accumulator layout: column r, rows (q & 3) + 8 (q >> 2) + 4 h)
    auto pixg = [&](int row) { return (size_t)(m0 + row); };
    C2Res<T, TM, TN, NT> pre;
    c2_res_prefetch<T, TM, TN, NT>(a, tid, n0, (const T*)a.res, pixg, pre);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lc = wn * 64 + j * 32 + r;
        float bias = (a.bias && n0 + lc < a.cout) ? a.bias[n0 + lc] : 0.f;
        if (a.bias_b && n0 + lc < a.cout) bias += a.bias_b[(size_t)(m0 / HW) * a.bias_bstride + n0 + lc];       // (a tile lies inside one image)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if constexpr (kBf16) *(unsigned short*)(lds + row * PO + lc * 2) = f32_to_bf16_hw(acc[j][q] + bias);
            else *(float*)(lds + row * PO + lc * 4) = acc[j][q] + bias;
        }
    }
    __syncthreads();
    c2_store_tile<T, TM, TN, NT>(a, lds, PO, tid, n0, (int)(m0 / HW), (T*)a.out, (const T*)a.res, pixg, pre);
}

// The same-size 3x3 convs (the two convs of every ResBlock: > 90 % of the network's flops) on SPATIAL tiles: TH x 32 output pixels
// per workgroup.  A K chunk of the (TH + 2) x 34 halo is gathered, activated and staged ONCE and serves all nine taps -- a tap is a
// row offset into the halo -- so the GroupNorm / SiLU prologue and the gather's address arithmetic run once per halo pixel and chunk
// instead of once per pixel, chunk AND tap (the per-tap gather above spends ~4x the MFMA cycles on vector work); the weight slab of a
// (chunk, tap) is the only thing staged per iteration.  A wave owns one tile row (32 pixels) x 64 channels.
// WR = tile rows per wave: with one row a wave's 32 x 64 tile reads 3 KB of fragments from LDS per two MFMAs and the kernel runs at the
// LDS read bandwidth (128 B/clk/CU) at a third of the matrix rate; two rows (64 x 64 per wave, four MFMAs per 4 KB) halve that.
template <int I, int N, typename F>
__device__ __forceinline__ void c2_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        c2_static_for<I + 1, N>(f);
    }
}
// timing knock-outs of diagnostic builds only (tools/build_variant.sh NAME -DADF_C2_KNOCK=bits; results are wrong by construction):
// 1 weight slabs loaded once, 2 halo chunks loaded once, 4 no MFMA, 8 no per-iteration barrier, 16 no prologue arithmetic, 32 no fragment reads
#ifndef ADF_C2_KNOCK
#define ADF_C2_KNOCK 0
#endif
// (TH = 5, ten waves, takes 112 registers, so a CU holds one workgroup; bounding it to 96 for two -- amdgpu_waves_per_eu(5) -- spills ~5 dwords per
// iteration and measured 6 % slower: 379 against 404 TF/s on the 512-channel convs of config 4's coarsest level)
template <typename T, int TH, int WR>
__global__ void __launch_bounds__(TH / WR * 128) conv2d_tile_kernel(const Conv2dArgs a) {
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int KC = 128 / (int)sizeof(T);
    constexpr int TW = 32, TM = TH * TW, TN = 128;
    constexpr int NT = TH / WR * 128;                  // 2 TH / WR waves
    constexpr int HR = (TH + 2) * (TW + 2);            // halo rows
    constexpr int ASTAGE = HR * kC2Pitch, WSTAGE = TN * kC2Pitch;
    constexpr int PO = TN * (int)sizeof(T) + 16;
    // (the output tile of the epilogue reuses the stages and the table; the launcher sizes the allocation for the larger of the two)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const ldsA = lds;                            // ONE halo stage (restaged between chunks behind a barrier): with two the
                                                       // workgroup takes 104 KB and a CU holds a single one, whose start-up, barriers
                                                       // and epilogue nothing overlaps (K is only 18 iterations at 128 channels)
    char* const ldsW = lds + ASTAGE;                   // 2 weight stages
    float* const abs_ = (float*)(ldsW + 2 * WSTAGE);   // [cin][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    // logical tile = (spatial tile, N tile) with the N tile fastest, in XCD order: the N tiles of a spatial tile (same halo) and its spatial neighbours
    // (shared halo rows / columns) run close in time on ONE XCD's L2
    const unsigned ltile = adf_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    int bid = (int)(ltile / gridDim.y);
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TH;
    const int b = bid / tiles_y;
    const int n0 = (int)(ltile % gridDim.y) * TN;
    // mode 1 (nearest x2 upsampling fused): the halo keeps its OUTPUT geometry in LDS, each of its pixels is fetched from input pixel (y >> 1, x >> 1)
    // (four halo pixels share a source pixel: the repeats are cache hits), so the taps stay row offsets into the halo
    const int Hin = a.mode == 1 ? a.H >> 1 : a.H, Win = a.mode == 1 ? a.W >> 1 : a.W;
    const T* const xg = (const T*)a.x + (size_t)b * Hin * Win * a.c0;
    const T* const xg1 = a.x1 ? (const T*)a.x1 + (size_t)b * Hin * Win * (a.cin - a.c0) : nullptr;
    const char* const wg = (const char*)a.w;
#ifdef ADF_C2_STAMP
    const bool stamped = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;
#endif
    C2_STAMP(0);
    if (a.ab) {
        const float* const abg = a.ab + (size_t)b * a.cin * 2;
        // 16-byte pieces: 1024 channels are one trip of the workgroup (float by float they were four load -> store trips, each a memory round trip)
        for (int i = tid; i < a.cin / 2; i += NT) ((u32x4_t*)abs_)[i] = ((const u32x4_t*)abg)[i];
    }

    // halo pieces of this thread: piece ids tid + k NT over HR * 8 pieces; (pixel offset, validity) are fixed for the whole kernel
    constexpr int NHP = (HR * 8 + NT - 1) / NT;
    int hoff[NHP];                                     // pixel index of the halo piece inside the image, or -1 = outside the image / no piece
#pragma unroll
    for (int k = 0; k < NHP; ++k) {
        const int id = tid + k * NT;
        const int row = id >> 3;
        const int hy = row / (TW + 2), hx = row - hy * (TW + 2);
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        hoff[k] = (id < HR * 8 && y >= 0 && y < a.H && x >= 0 && x < a.W) ? (a.mode == 1 ? (y >> 1) * Win + (x >> 1) : y * a.W + x) : -1;   // source pixel index; the channel stride depends on the source
    }
    u32x4_t ra[NHP], rw[2];
    auto load_a = [&](int ck) __attribute__((always_inline)) {
        const bool s1 = ck * KC >= a.c0;                                  // uniform
        const int cs = s1 ? a.cin - a.c0 : a.c0, cb = (s1 ? ck * KC - a.c0 : ck * KC) + (tid & 7) * EPC;
        const T* const src = s1 ? xg1 : xg;
#pragma unroll
        for (int k = 0; k < NHP; ++k) ra[k] = *(const u32x4_t*)(src + (hoff[k] >= 0 ? hoff[k] * cs + cb : 0));
    };
    // the prologue (GroupNorm affine + SiLU) of a chunk's halo pieces, in place in their registers.  For the chunks after the first it runs in
    // the MIDDLE of the previous chunk's nine taps, where the waves have drifted apart and its vector work overlaps other waves' MFMAs; done at the
    // restage point (tap 8, between two barriers) every wave of the workgroup did it at the same time, with the matrix pipes idle
    auto transform_a = [&](int ck) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
            const int id = tid + k * NT;
            if (id >= HR * 8) continue;
            u32x4_t v = ra[k];
            if (hoff[k] < 0) v = u32x4_t{0u, 0u, 0u, 0u};                 // zero padding, applied after the activation
            else if (a.ab && !(ADF_C2_KNOCK & 16)) {
                float f[EPC];
                unpack16<T>(v, f);
                const float* const ab = abs_ + (ck * KC + (id & 7) * EPC) * 2;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float t = fmaf(f[e], ab[2 * e], ab[2 * e + 1]);
                    f[e] = a.act ? (kBf16 ? silu_f(t) : t / (1.0f + expf(-t))) : t;
                }
                v = pack16<T>(f);
            }
            ra[k] = v;
        }
    };
    auto store_a = [&]() __attribute__((always_inline)) {
        char* const st = ldsA;
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
            const int id = tid + k * NT;
            if (id >= HR * 8) continue;
            *(u32x4_t*)(st + (id >> 3) * kC2Pitch + (id & 7) * 16) = ra[k];
        }
    };
    // weight slabs travel TWO iterations ahead (one iteration of 8 MFMAs per wave does not cover an L2 round trip), slab j in register set j % 3:
    // nine taps per chunk, so the set of a tap is a compile-time constant
    constexpr int NWP = (TN * 8 + NT - 1) / NT;        // weight pieces per thread and iteration: 2 (TH = 4, 5: the last sweep of TH = 5 is partial) or 4 (TH = 2)
    constexpr bool kWPartial = (TN * 8) % NT != 0;
    u32x4_t rwv[3][NWP];
    // rows past n_pad (and the pieces past the slab in a partial last sweep) are zeroed when the slab is STORED: a select right behind the load
    // makes the compiler wait for it on the spot (s_waitcnt vmcnt(0) behind every weight load: the slab that was meant to travel for two
    // iterations was waited for at once -- found with the knock-out builds, -DADF_C2_KNOCK=1: 407 -> 719 TF/s on the 512-channel convs)
    bool wok[NWP];
    unsigned woff[NWP];
#pragma unroll
    for (int k = 0; k < NWP; ++k) {
        const int id = tid + k * NT, row = id >> 3;
        wok[k] = n0 + row < a.n_pad && (!kWPartial || id < TN * 8);
        woff[k] = (unsigned)((wok[k] ? n0 + row : 0) * 128 + (id & 7) * 16);
    }
    const size_t wslab = (size_t)a.n_pad * 128;
    auto load_w = [&](int it, u32x4_t (&rv)[NWP]) __attribute__((always_inline)) {     // it = ck * 9 + tap: the packed order [chunk][tap]
        const char* const base = wg + (size_t)it * wslab;
#pragma unroll
        for (int k = 0; k < NWP; ++k) rv[k] = *(const u32x4_t*)(base + woff[k]);
    };
    auto store_w = [&](int it, const u32x4_t (&rv)[NWP]) __attribute__((always_inline)) {
        char* const st = ldsW + (it & 1) * WSTAGE;
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int id = tid + k * NT;
            if (!kWPartial || id < TN * 8) *(u32x4_t*)(st + (id >> 3) * kC2Pitch + (id & 7) * 16) = wok[k] ? rv[k] : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    (void)rw;

    c2_f32x16_t acc[WR][2];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nit = a.nchunk * 9;
    load_a(0);
    load_w(0, rwv[0]);
    load_w(nit > 1 ? 1 : 0, rwv[1]);
    __syncthreads();                                   // the prologue table is in LDS
    C2_STAMP(1);
    transform_a(0);
    store_a();
    store_w(0, rwv[0]);
    __syncthreads();
    C2_STAMP(2);
    // iteration it = 9 ck + tap: issue W(it + 2), compute from LDS stage it & 1, store W(it + 1) into the other stage.  EVERY load of the loop body is
    // issued on every path (indices clamped at the end: the last slab / chunk is fetched again and never used): with a load under a condition the
    // compiler's vmcnt bookkeeping assumes the fewest younger loads, i.e. it waits for ALL loads in flight before every store_w -- the two-iteration
    // prefetch was a zero-iteration one (knock-out builds: 407 -> 719 TF/s without the weight loads on the 512-channel convs)
    auto iteration = [&](int ck, auto tapc) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int dy = tap / 3, dx = tap - dy * 3;
        const int it = ck * 9 + tap;
        if (!(ADF_C2_KNOCK & 1)) load_w(it + 2 < nit ? it + 2 : nit - 1, rwv[(tap + 2) % 3]);
        if (!(ADF_C2_KNOCK & 2) && tap == 0) load_a(ck + 1 < a.nchunk ? ck + 1 : a.nchunk - 1);     // the next chunk's halo travels during this chunk's nine taps
        const char* const aRow = ldsA + ((wm * WR + dy) * (TW + 2) + r + dx) * kC2Pitch;      // tile row wm * WR (+ i): one halo row further
        const char* const wRow = ldsW + (it & 1) * WSTAGE + (wn * 64 + r) * kC2Pitch;
        if constexpr (kBf16) {
            // fragments of K step ks + 1 are read while the MFMAs of step ks issue (the LDS latency would otherwise sit between every read
            // and its MFMAs: with 4 - 8 MFMAs per step nothing else covers it)
            c2_bf16x8_t fa[2][WR], fb[2][2];
            auto frags = [&](int ks, c2_bf16x8_t (&xa)[WR], c2_bf16x8_t (&xb)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < WR; ++i) xa[i] = *(const c2_bf16x8_t*)(aRow + i * (TW + 2) * kC2Pitch + (ks * 2 + h) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) xb[j] = *(const c2_bf16x8_t*)(wRow + j * 32 * kC2Pitch + (ks * 2 + h) * 16);
            };
            if ((ADF_C2_KNOCK & 32)) {
#pragma unroll
                for (int z = 0; z < 2; ++z) {
#pragma unroll
                    for (int i = 0; i < WR; ++i) fa[z][i] = __builtin_bit_cast(c2_bf16x8_t, ra[0]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[z][j] = __builtin_bit_cast(c2_bf16x8_t, rwv[(tap + 1) % 3][0]);
                }
            } else
            frags(0, fa[0], fb[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (!(ADF_C2_KNOCK & 32) && ks + 1 < 4) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (!(ADF_C2_KNOCK & 4))
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                if ((ADF_C2_KNOCK & 4)) { asm volatile("" :: "v"(fa[ks & 1][0]), "v"(fb[ks & 1][0]), "v"(fb[ks & 1][1])); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                c2_f32x4_t fa[WR][2], fb[2][2];
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int u = 0; u < 2; ++u) fa[i][u] = *(const c2_f32x4_t*)(aRow + i * (TW + 2) * kC2Pitch + (ks * 4 + 2 * h + u) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) fb[j][u] = *(const c2_f32x4_t*)(wRow + j * 32 * kC2Pitch + (ks * 4 + 2 * h + u) * 16);
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][u][e], fb[j][u][e], acc[i][j], 0, 0, 0);
            }
        }
        store_w(it + 1, rwv[(tap + 1) % 3]);                          // (after the last iteration: a stage nobody reads any more)
        if (tap == 4 && ck + 1 < a.nchunk) transform_a(ck + 1);       // the halo loaded at tap 0 has landed; registers only
        if (tap == 8 && ck + 1 < a.nchunk) {                          // every wave must have left this chunk's halo before it is replaced
            __syncthreads();
            store_a();
        }
        if (!(ADF_C2_KNOCK & 8)) __syncthreads();
    };
    for (int ck = 0; ck < a.nchunk; ++ck) {
        c2_static_for<0, 9>([&](auto tapc) __attribute__((always_inline)) { iteration(ck, tapc); });
        if (ck == 0) C2_STAMP(3);
        if (ck == 1) C2_STAMP(5);
    }
    C2_STAMP(6);

    // ---- epilogue through LDS (as conv2d_gemm_kernel): tile rows wm * WR + i, pixel = accumulator row -------------------------------
    T* const og = (T*)a.out + (size_t)b * a.H * a.W * a.cout;
    const T* const rg = a.res ? (const T*)a.res + (size_t)b * a.H * a.W * a.cout : nullptr;
    auto pixt = [&](int row) { return (size_t)(ty0 + row / TW) * a.W + tx0 + (row % TW); };
    C2Res<T, TM, TN, NT> pre;
    c2_res_prefetch<T, TM, TN, NT>(a, tid, n0, rg, pixt, pre);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lc = wn * 64 + j * 32 + r;
        float bias = (a.bias && n0 + lc < a.cout) ? a.bias[n0 + lc] : 0.f;
        if (a.bias_b && n0 + lc < a.cout) bias += a.bias_b[(size_t)b * a.bias_bstride + n0 + lc];
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (wm * WR + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if constexpr (kBf16) *(unsigned short*)(lds + row * PO + lc * 2) = f32_to_bf16_hw(acc[i][j][q] + bias);
                else *(float*)(lds + row * PO + lc * 4) = acc[i][j][q] + bias;
            }
    }
    __syncthreads();
    C2_STAMP(7);
    c2_store_tile<T, TM, TN, NT>(a, lds, PO, tid, n0, b, og, rg, pixt, pre);
    C2_STAMP(8);
}

template <typename T, int TH, int WR>
static const char* launch_conv2d_tile(const Conv2dArgs& a, hipStream_t s) {
    constexpr int HR = (TH + 2) * 34;
    const size_t stages = (size_t)HR * kC2Pitch + 2 * 128 * kC2Pitch + (size_t)2 * a.cin * 4;
    const size_t otile = (size_t)TH * 32 * (128 * sizeof(T) + 16);
    const size_t lds = stages > otile ? stages : otile;
    static bool attr_done[kMaxDevices] = {};
    bool& attr = attr_done[current_device()];
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv2d_tile_kernel<T, TH, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "conv2d_tile: hipFuncSetAttribute failed";
        attr = true;
    }
    const dim3 grid((unsigned)(a.B * (a.H / TH) * (a.W / 32)), (unsigned)ceil_div(a.cout, 128)), blk(TH / WR * 128);
    hipLaunchKernelGGL((conv2d_tile_kernel<T, TH, WR>), grid, blk, lds, s, a);
    return C2_LAUNCH_CHECK("conv2d_tile");
}

static inline bool NT_OK(const Conv2dArgs&) { return true; }
// ADF_C2_TRACE=1: one stderr line per conv2d launch (tools/adm_layer_table.py matches them with a kernel trace)
static void c2_trace(const char* route, const Conv2dArgs& a) {
    static const int on = adf_route_switch("ADF_C2_TRACE", 0);
    if (on) fprintf(stderr, "[adf conv2d] %-6s B=%d H=%d W=%d cin=%d c0=%d cout=%d taps=%d mode=%d ab=%d act=%d res=%d stats=%d\n", route, a.B, a.H, a.W, a.cin, a.c0,
                    a.cout, a.taps, a.mode, a.ab != nullptr, a.act, a.res != nullptr, a.stats != nullptr);
}

const char* launch_conv2d(const Conv2dArgs& a, int bf16, hipStream_t s) {
    const int kc = bf16 ? 64 : 32;
    if (a.taps != 9 && a.taps != 1) return "conv2d: taps must be 9 or 1";
    if (a.taps == 1 && a.mode != 0) return "conv2d: a 1x1 conv has no resampling mode";
    if (a.cin % kc || a.cin > 1024) return "conv2d: input channels must be a multiple of the 128-byte K chunk, at most 1024";
    if (a.c0 < 1 || a.c0 > a.cin || a.c0 % kc || (a.c0 < a.cin) != (a.x1 != nullptr)) return "conv2d: bad source split";
    if (a.nchunk * kc != a.cin) return "conv2d: packed weight chunk count does not match the input channels";
    if (a.cout % (bf16 ? 8 : 4)) return "conv2d: output channels must be a multiple of a 16-byte piece";
    if (a.stats) {
        const int gs = a.stats_groups > 0 ? a.cout / a.stats_groups : 0;
        if (gs < 1 || a.cout % a.stats_groups || 128 % gs || (a.cout > 128 && a.cout % 128)) return "conv2d: statistics need a group size dividing the 128-channel tile";
    }
    const long long px = (long long)a.B * a.H * a.W;
    if (((long long)a.H * a.W) % 64) return "conv2d: H*W must be a multiple of 64";
    if (a.mode == 1 && ((a.H | a.W) & 1)) return "conv2d: upsampled output must have even height and width";
    const unsigned ny = (unsigned)ceil_div(a.cout, 128);
    // same-size and upsampling 3x3: spatial tiles (the halo staged once per chunk serves all nine taps); route switch for the parity tests
    static const int tile_route = adf_route_switch("ADF_CONV2D_TILE", 1);
    if (tile_route && a.taps == 9 && a.mode != 2 && a.W % 32 == 0 && (long long)a.H * a.W * a.cin < (1ll << 31) && NT_OK(a)) {
        // a workgroup re-reads every weight slab from L2: at 128 pixels per workgroup that stream (16 KB per iteration against 64 MFMAs) runs at
        // the L2 -> CU rate and bounds the kernel; 256 pixels per workgroup halve it
        static const int big = (int)adf_tuning("ADF_CONV2D_TH8", 0);     // measured slower than two 128-pixel workgroups per CU: kept for A/B runs only
        if (big && a.H % 8 == 0 && px / 256 * ny >= 512) return c2_trace("t8x2", a), bf16 ? launch_conv2d_tile<bf16_t, 8, 2>(a, s) : launch_conv2d_tile<float, 8, 2>(a, s);
        static const int th8w1 = (int)adf_tuning("ADF_CONV2D_TH8W1", 0); // 8 x 32 pixels on sixteen waves, one workgroup per CU (A/B runs)
        if (th8w1 && a.H % 8 == 0 && px / 256 * ny >= 256) return c2_trace("t8", a), bf16 ? launch_conv2d_tile<bf16_t, 8, 1>(a, s) : launch_conv2d_tile<float, 8, 1>(a, s);
        static const int wr2 = (int)adf_tuning("ADF_CONV2D_WR2", 0);     // 64 x 64 wave tiles on four waves (A/B runs)
        if (wr2 && a.H % 4 == 0 && px / 128 * ny >= 128) return c2_trace("t4x2", a), bf16 ? launch_conv2d_tile<bf16_t, 4, 2>(a, s) : launch_conv2d_tile<float, 4, 2>(a, s);
        // (A/B: 160-pixel tiles also where 128-pixel ones divide the image, for convs of at most this many input channels -- few K iterations per tile, so the
        //  tile's start-up and epilogue weigh most)
        static const int prefer5 = (int)adf_tuning("ADF_CONV2D_PREFER5", 0);
        if (prefer5 && a.cin <= prefer5 && a.H % 5 == 0 && px / 160 * ny >= 128)
            return c2_trace("t5", a), bf16 ? launch_conv2d_tile<bf16_t, 5, 1>(a, s) : launch_conv2d_tile<float, 5, 1>(a, s);
        if (a.H % 4 == 0 && px / 128 * ny >= 128) return c2_trace("t4", a), bf16 ? launch_conv2d_tile<bf16_t, 4, 1>(a, s) : launch_conv2d_tile<float, 4, 1>(a, s);
        // heights such as 10 (the 80-row mel block three levels down): 160-pixel workgroups instead of 64-pixel ones -- 2.5 x fewer passes over the weights
        static const int th5 = (int)adf_tuning("ADF_CONV2D_TH5", 1);
        if (th5 && a.H % 5 == 0 && px / 160 * ny >= 128) return c2_trace("t5", a), bf16 ? launch_conv2d_tile<bf16_t, 5, 1>(a, s) : launch_conv2d_tile<float, 5, 1>(a, s);
        if (a.H % 2 == 0) return c2_trace("t2", a), bf16 ? launch_conv2d_tile<bf16_t, 2, 1>(a, s) : launch_conv2d_tile<float, 2, 1>(a, s);
    }
    // 128-pixel tiles (each weight piece staged once per 128 pixels) when the image divides and the grid still fills the chip
    if (((long long)a.H * a.W) % 128 == 0 && px / 128 * ny >= 512) {
        const dim3 grid((unsigned)(px / 128), ny), blk(512);
        c2_trace("g128", a);
        if (bf16) hipLaunchKernelGGL((conv2d_gemm_kernel<bf16_t, 128>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((conv2d_gemm_kernel<float, 128>), grid, blk, 0, s, a);
    } else {
        const dim3 grid((unsigned)(px / 64), ny), blk(256);
        c2_trace("g64", a);
        if (bf16) hipLaunchKernelGGL((conv2d_gemm_kernel<bf16_t, 64>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((conv2d_gemm_kernel<float, 64>), grid, blk, 0, s, a);
    }
    return C2_LAUNCH_CHECK("conv2d");
}

// ------------------------------------------------------------------------------------------------ first / last conv
constexpr int kC2InSweeps = 16;
constexpr int kC2InRounds = 4;      // stagings per workgroup (weights staged and statistics flushed once for all of them)
// A workgroup covers kC2InSweeps sweeps of 256 / cpr consecutive pixels.  The 9 * cin input values of each of its pixels (c_in applied, zero outside the
// image) are staged ONCE into LDS by one batch of global loads per thread; every thread of a pixel (one per 16-byte piece of the output row) then
// reads them from there.  (Read straight from global they were 9 loads per thread and sweep, 16 threads fetching the same values, and the kernel
// sat at their latency: 0.37 ms for 64 blocks of 80 x 256 against 0.06 ms at the write rate; two and four pixels in flight per thread gave 0.33 / 0.18.)
template <typename T, int CIN>     // CIN = 1, 2: the thread's weights (its 16-byte piece of the output row x 9 taps x CIN) live in registers; 0: read from LDS per use
__global__ void __launch_bounds__(256) conv2d_in_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, int B, int cin, int H, int W, int cout,
                                                        const float* __restrict__ coef, int coef_bstride, double* __restrict__ stats, int fg) {
    constexpr int EPC = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ws = (float*)smem;                    // [cin][9][cout] + bias [cout]
    float* const xs = ws + cin * 9 * cout + cout;      // [pixels of this workgroup][cin][9]
    const int cpr = cout / EPC, ppb = 256 / cpr;
    const int npb = ppb * kC2InSweeps;                 // pixels per workgroup
    float* const st = xs + npb * cin * 9;              // [2][cout]: sums / sums of squares of this workgroup's stored outputs (stats != nullptr)
    if (stats)
        for (int i = threadIdx.x; i < 2 * cout; i += 256) st[i] = 0.f;
    const int npix = B * H * W, HW = H * W;
    const int p00 = blockIdx.x * npb * kC2InRounds;
    for (int i = threadIdx.x; i < cout * cin * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % cin, co = i / (9 * cin);
        ws[(ci * 9 + t) * cout + co] = w[i];
    }
    for (int i = threadIdx.x; i < cout; i += 256) ws[cin * 9 * cout + i] = bias[i];
    __syncthreads();
    const int cc = threadIdx.x % cpr, lp = threadIdx.x / cpr;
    const int nsw = lp < ppb ? kC2InSweeps : 0;        // (no early return: the statistics tail below has a barrier)
    float bias_r[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) bias_r[e] = ws[cin * 9 * cout + cc * EPC + e];
    // (with the weights read from LDS for every pixel -- 18 ds_read_b128 per thread and sweep -- the LDS reads bounded the kernel at 0.35 ms)
    constexpr int WR = CIN > 0 ? CIN : 1;
    float wreg[WR][9][EPC];
    if constexpr (CIN > 0) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < EPC; ++e) wreg[ci][t][e] = ws[(ci * 9 + t) * cout + cc * EPC + e];
    }
    float a1[EPC], a2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    for (int rd = 0; rd < kC2InRounds; ++rd) {         // the workgroup's weights and statistics serve kC2InRounds stagings
        const int p0 = p00 + rd * npb;
        if (p0 >= npix) break;
        if (rd) __syncthreads();                       // every thread is done with the previous round's inputs
        // stage the inputs: item i = (pixel, channel, tap); all loads of a thread before its first LDS store
        const int nitem = npb * cin * 9;
        constexpr int kMaxItems = 12;                  // per thread: 256 pixels x 9 taps x cin / 256 threads = 9 cin (cin = 1: 9)
        float xv[kMaxItems];
        for (int base = 0; base < nitem; base += 256 * kMaxItems) {
#pragma unroll
            for (int k = 0; k < kMaxItems; ++k) {
                const int i = base + threadIdx.x + k * 256;
                const int t = i % 9, ci = (i / 9) % cin, lpix = i / (9 * cin);
                const int p = p0 + lpix;
                const int pc = (i < nitem && p < npix) ? p : 0;
                const int b = pc / HW, pp = pc - b * HW;
                const int y = pp / W, xx = pp - y * W;
                const int sy = y + t / 3 - 1, sx = xx + t % 3 - 1;
                const bool ok = i < nitem && p < npix && sy >= 0 && sy < H && sx >= 0 && sx < W;
                const float sc = coef ? coef[(size_t)b * coef_bstride] : 1.0f;
                const float v = x[((size_t)b * cin + ci) * HW + (ok ? sy * W + sx : 0)];
                xv[k] = ok ? v * sc : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kMaxItems; ++k) {
                const int i = base + threadIdx.x + k * 256;
                if (i < nitem) xs[i] = xv[k];
            }
        }
        __syncthreads();
        for (int sw = 0; sw < nsw; ++sw) {
            const int lpix = sw * ppb + lp;
            const int p = p0 + lpix;
            if (p >= npix) break;
            float f[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = bias_r[e];
            if constexpr (CIN > 0) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float* const xp = xs + (lpix * CIN + ci) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float vs = xp[t];
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = fmaf(wreg[ci][t][e], vs, f[e]);
                    }
                }
            } else {
                for (int ci = 0; ci < cin; ++ci) {
                    const float* const xp = xs + (lpix * cin + ci) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float vs = xp[t];
                        const float* const wp = ws + (ci * 9 + t) * cout + cc * EPC;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = fmaf(wp[e], vs, f[e]);
                    }
                }
            }
            const u32x4_t pk = pack16<T>(f);
            *(u32x4_t*)(out + (size_t)p * cout + cc * EPC) = pk;
            if (stats) {                                   // of the STORED values, as the separate statistics pass reads them
                float g[EPC];
                unpack16<T>(pk, g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { a1[e] += g[e]; a2[e] = fmaf(g[e], g[e], a2[e]); }
            }
        }
    }
    // FINE GroupNorm statistics of the output (sum, sum of squares per fg channels of a sample), what launch_gn_stats_any would reduce in a second
    // pass over the tensor (0.12 ms at 64 x 80 x 256 x 192): the launcher passes stats only when every workgroup's pixels lie in ONE sample
    if (stats) {
        if (nsw) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { atomicAdd(&st[cc * EPC + e], a1[e]); atomicAdd(&st[cout + cc * EPC + e], a2[e]); }
        }
        __syncthreads();
        const int b = p00 / HW, G = cout / fg;
        for (int g = threadIdx.x; g < G; g += 256) {
            double d1 = 0.0, d2 = 0.0;
            for (int c = g * fg; c < (g + 1) * fg; ++c) { d1 += (double)st[c]; d2 += (double)st[cout + c]; }
            atomicAdd(&stats[((size_t)b * G + g) * 2], d1);
            atomicAdd(&stats[((size_t)b * G + g) * 2 + 1], d2);
        }
    }
}
const char* launch_conv2d_in(const float* x, const float* w, const float* bias, void* out, int bf16, int B, int cin, int H, int W, int cout,
                             const float* coef, int coef_bstride, double* stats, int fg, hipStream_t s) {
    if (cout % (bf16 ? 8 : 4)) return "conv2d_in: output channels must be a multiple of a 16-byte chunk";
    if (stats && (fg < 1 || cout % fg)) return "conv2d_in: the statistics group must divide the output channels";
    const int cpr = cout / (bf16 ? 8 : 4);
    if (cpr > 256) return "conv2d_in: more than 256 16-byte pieces per output pixel";
    const long long npix = (long long)B * H * W;
    if (npix >= (1ll << 31)) return "conv2d_in: more than 2^31 pixels";
    const int per_block = (256 / cpr) * kC2InSweeps;
    const int blocks = (int)((npix + (long long)per_block * kC2InRounds - 1) / ((long long)per_block * kC2InRounds));
    const size_t lds = ((size_t)cout * cin * 9 + cout + (size_t)per_block * cin * 9 + 2 * (size_t)cout) * 4;
    if (lds > 64 * 1024) return "conv2d_in: weights and the staged inputs do not fit LDS";
    // the statistics ride in the store when no workgroup straddles two samples; otherwise the separate pass over the stored tensor follows
    double* const fused = stats && ((long long)H * W) % ((long long)per_block * kC2InRounds) == 0 ? stats : nullptr;
#define ADF_C2IN(T_, CI_) hipLaunchKernelGGL((conv2d_in_kernel<T_, CI_>), dim3(blocks), dim3(256), lds, s, x, w, bias, (T_*)out, B, cin, H, W, cout, coef, coef_bstride, fused, fg)
    if (bf16) { if (cin == 1) ADF_C2IN(bf16_t, 1); else if (cin == 2) ADF_C2IN(bf16_t, 2); else ADF_C2IN(bf16_t, 0); }
    else { if (cin == 1) ADF_C2IN(float, 1); else if (cin == 2) ADF_C2IN(float, 2); else ADF_C2IN(float, 0); }
#undef ADF_C2IN
    if (const char* e = C2_LAUNCH_CHECK("conv2d_in")) return e;
    if (stats && !fused) return launch_gn_stats_any(out, bf16, B, H * W, cout, cout / fg, stats, s);
    return nullptr;
}

constexpr int kC2OutMax = 4;        // output channels of the last conv served by the vector kernel
// Every INPUT pixel of the tile's halo region is activated once (SiLU(GroupNorm): ten instructions per element against one FMA per tap) and
// leaves its 9 * cout partial dot products -- pixel x tap -- in LDS; an output pixel then sums nine neighbours' partials.  (The first version
// activated every pixel nine times, once per tap of every neighbour: 0.74 ms for 64 blocks of 80 x 256, 4 % of a pass.)
template <typename T, int TH, int TW, int CO>
__global__ void __launch_bounds__(256) conv2d_out_kernel(const T* __restrict__ hx, const float* __restrict__ ab, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int B, int cin, int H, int W,
                                                         int mode, const float* __restrict__ x_noisy, const float* __restrict__ coef,
                                                         int coef_bstride) {
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int HW_ = TW + 2, HR = (TH + 2) * HW_;
    constexpr int cout = CO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ws = (float*)smem;                    // [cout][9][cin]
    float* const part = ws + cout * 9 * cin;           // [HR][9 * cout]
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % cin, co = i / (9 * cin);
        ws[(co * 9 + t) * cin + ci] = w[i];
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TH;
    const int b = bid / tiles_y;
    const int nq = 9 * cout;
    __syncthreads();
    const int sub = tid & 7;                           // 8 lanes share an input pixel, each walks every 8th 16-byte piece of the channels
    const int cpr = cin / EPC;
    const float* const abb = ab + (size_t)b * cin * 2;
    for (int hp = tid >> 3; hp < HR + 31 - (HR + 31) % 32; hp += 32) {      // whole sweeps: the shuffles below need all lanes
        const int hy = hp / HW_, hxx = hp - hy * HW_;
        const int y = ty0 + hy - 1, x = tx0 + hxx - 1;
        const bool live = hp < HR && y >= 0 && y < H && x >= 0 && x < W;
        float acc[CO][9];
#pragma unroll
        for (int co = 0; co < CO; ++co)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[co][t] = 0.f;
        if (live) {
            const T* const row = hx + (((size_t)b * H + y) * W + x) * cin;
            for (int cc = sub; cc < cpr; cc += 8) {
                float f[EPC];
                unpack16<T>(*(const u32x4_t*)(row + cc * EPC), f);
                const float* const abp = abb + cc * EPC * 2;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float tt = fmaf(f[e], abp[2 * e], abp[2 * e + 1]);
                    f[e] = kBf16 ? silu_f(tt) : tt / (1.0f + expf(-tt));
                }
#pragma unroll
                for (int co = 0; co < CO; ++co) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float* const wp = ws + (co * 9 + t) * cin + cc * EPC;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[co][t] = fmaf(wp[e], f[e], acc[co][t]);
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = acc[co][t];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                if (hp < HR && sub == (t & 7)) part[hp * nq + co * 9 + t] = v;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < TH * TW * cout; i += 256) {
        const int co = i / (TH * TW), pix = i - co * (TH * TW);
        const int ly = pix / TW, lx = pix - ly * TW;
        const int y = ty0 + ly, x = tx0 + lx;
        if (y >= H || x >= W) continue;
        float F = bias[co];
#pragma unroll
        for (int t = 0; t < 9; ++t) F += part[((ly + t / 3) * HW_ + lx + t % 3) * nq + co * 9 + t];
        const size_t o = (((size_t)b * cout + co) * H + y) * W + x;
        if (mode == 0) out[o] = F;
        else {
            const float c_skip = coef[(size_t)b * coef_bstride + 2], c_out = coef[(size_t)b * coef_bstride + 3];
            out[o] = fminf(fmaxf(fmaf(c_out, F, c_skip * x_noisy[o]), -1.0f), 1.0f);
        }
    }
}
template <typename T, int TH, int TW, int CO>
static const char* launch_conv2d_out_t(const void* h, const float* ab, const float* w, const float* bias, float* out, int B, int cin, int H, int W,
                                       int mode, const float* x_noisy, const float* coef, int coef_bstride, hipStream_t s) {
    const size_t lds = ((size_t)CO * cin * 9 + (size_t)(TH + 2) * (TW + 2) * 9 * CO) * 4;
    if (lds > 64 * 1024) return "conv2d_out: weights and partial sums do not fit LDS";
    const long long blocks = (long long)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (blocks > 0x7fffffffll) return "conv2d_out: grid too large";
    hipLaunchKernelGGL((conv2d_out_kernel<T, TH, TW, CO>), dim3((unsigned)blocks), dim3(256), lds, s, (const T*)h, ab, w, bias, out, B, cin, H, W, mode,
                       x_noisy, coef, coef_bstride);
    return C2_LAUNCH_CHECK("conv2d_out");
}
const char* launch_conv2d_out(const void* h, const float* ab, const float* w, const float* bias, float* out, int bf16, int B, int cin, int H,
                              int W, int cout, int mode, const float* x_noisy, const float* coef, int coef_bstride, hipStream_t s) {
    if (cout < 1 || cout > kC2OutMax) return "conv2d_out: more output channels than the vector kernel serves";
    if (cin % (bf16 ? 8 : 4)) return "conv2d_out: input channels must be a multiple of a 16-byte chunk";
    if (!ab) return "conv2d_out: the last conv always follows a GroupNorm";
    // 8 x 32 output pixels per workgroup (halo 1.33); 4 x 32 (1.59) when the partial sums of 3 - 4 output channels would not fit 64 KB
#define ADF_C2OUT(T_, TH_, CO_) launch_conv2d_out_t<T_, TH_, 32, CO_>(h, ab, w, bias, out, B, cin, H, W, mode, x_noisy, coef, coef_bstride, s)
    switch (cout) {
        case 1: return bf16 ? ADF_C2OUT(bf16_t, 8, 1) : ADF_C2OUT(float, 8, 1);
        case 2: return bf16 ? ADF_C2OUT(bf16_t, 8, 2) : ADF_C2OUT(float, 8, 2);
        case 3: return bf16 ? ADF_C2OUT(bf16_t, 4, 3) : ADF_C2OUT(float, 4, 3);
        default: return bf16 ? ADF_C2OUT(bf16_t, 4, 4) : ADF_C2OUT(float, 4, 4);
    }
#undef ADF_C2OUT
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void __launch_bounds__(256) concat2_kernel(const u32x4_t* __restrict__ s0, const u32x4_t* __restrict__ s1, int q0, int q1,
                                                      long long rows, u32x4_t* __restrict__ out) {
    const int q = q0 + q1;
    const long long total = rows * q;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long row = idx / q;
        const int c = (int)(idx - row * q);
        out[idx] = c < q0 ? s0[row * q0 + c] : s1[row * q1 + (c - q0)];
    }
}
const char* launch_concat2(const void* s0, const void* s1, int c0, int c1, long long rows, void* out, int bf16, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (c0 % epc || c1 % epc) return "concat2: channel counts must be multiples of a 16-byte chunk";
    const long long total = rows * ((c0 + c1) / epc);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(concat2_kernel, dim3(blocks), dim3(256), 0, s, (const u32x4_t*)s0, (const u32x4_t*)s1, c0 / epc, c1 / epc, rows, (u32x4_t*)out);
    return C2_LAUNCH_CHECK("concat2");
}

template <typename T>
__global__ void __launch_bounds__(256) gn_stats_any_kernel(const T* __restrict__ x, int L, int C, int G, int rows_per_block, double* __restrict__ stats) {
    constexpr int EPC = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const s1 = (float*)smem;                    // [C] sums, [C] sums of squares of this block's rows
    float* const s2 = s1 + C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) s1[i] = 0.f;
    __syncthreads();
    const int b = blockIdx.y, cpr = C / EPC;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < L ? r0 + rows_per_block : L;
    // a thread keeps ONE channel piece and strides over rows, so its partial sums stay in registers
    for (int cc = threadIdx.x % cpr, lane_row = threadIdx.x / cpr, step = 256 / cpr > 0 ? 256 / cpr : 1; cc < cpr && lane_row < step; cc += cpr) {
        float a1[EPC], a2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
        for (int row = r0 + lane_row; row < r1; row += step) {
            float f[EPC];
            unpack16<T>(*(const u32x4_t*)(x + ((size_t)b * L + row) * C + cc * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { a1[e] += f[e]; a2[e] = fmaf(f[e], f[e], a2[e]); }
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { atomicAdd(&s1[cc * EPC + e], a1[e]); atomicAdd(&s2[cc * EPC + e], a2[e]); }
    }
    __syncthreads();
    const int gs = C / G;
    for (int g = threadIdx.x; g < G; g += 256) {
        double d1 = 0.0, d2 = 0.0;
        for (int c = g * gs; c < (g + 1) * gs; ++c) { d1 += (double)s1[c]; d2 += (double)s2[c]; }
        atomicAdd(&stats[((size_t)b * G + g) * 2], d1);
        atomicAdd(&stats[((size_t)b * G + g) * 2 + 1], d2);
    }
}
const char* launch_gn_stats_any(const void* x, int bf16, int B, int L, int C, int G, double* stats, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc || C % G) return "gn_stats_any: C must be a multiple of a 16-byte chunk and of the group count";
    if (C / epc > 256) return "gn_stats_any: more than 256 channel pieces per row";
    int rows_per_block = 64;
    while ((long long)ceil_div(L, rows_per_block) * B > 4096) rows_per_block *= 2;
    const dim3 grid(ceil_div(L, rows_per_block), B);
    const size_t lds = (size_t)2 * C * 4;
    if (bf16) hipLaunchKernelGGL(gn_stats_any_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)x, L, C, G, rows_per_block, stats);
    else hipLaunchKernelGGL(gn_stats_any_kernel<float>, grid, dim3(256), lds, s, (const float*)x, L, C, G, rows_per_block, stats);
    return C2_LAUNCH_CHECK("gn_stats_any");
}

__global__ void __launch_bounds__(256) gn_finalize_fine_kernel(const GnFineArgs a) {
    const int b = blockIdx.x, ctot = a.c0 + a.c1, gs = ctot / a.G;
    for (int c = threadIdx.x; c < ctot; c += 256) {
        const int g = c / gs;
        double sum = 0.0, sq = 0.0;
        for (int k = g * gs / a.fg; k < (g + 1) * gs / a.fg; ++k) {     // fine groups of fg channels; a group may straddle the two sources
            const double* st = k * a.fg < a.c0 ? a.stats0 + ((size_t)b * (a.c0 / a.fg) + k) * 2 : a.stats1 + ((size_t)b * (a.c1 / a.fg) + (k - a.c0 / a.fg)) * 2;
            sum += st[0]; sq += st[1];
        }
        const double cnt = (double)a.L * (double)gs;
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        float A = rstd * a.gamma[c];
        float Bc = a.beta[c] - (float)mean * A;
        if (a.film) {
            const float fs = a.film[(size_t)b * a.film_bstride + c] + 1.0f, fh = a.film[(size_t)b * a.film_bstride + ctot + c];
            A *= fs;
            Bc = fmaf(Bc, fs, fh);
        }
        float* o = a.ab + ((size_t)b * ctot + c) * 2;
        o[0] = A; o[1] = Bc;
    }
}
const char* launch_gn_finalize_fine(const GnFineArgs& a, hipStream_t s) {
    const int ctot = a.c0 + a.c1;
    if (a.fg < 1 || ctot % a.G || (ctot / a.G) % a.fg || a.c0 % a.fg || a.c1 % a.fg) return "gn_finalize_fine: group size and source widths must be multiples of the fine group";
    hipLaunchKernelGGL(gn_finalize_fine_kernel, dim3(a.B), dim3(256), 0, s, a);
    return C2_LAUNCH_CHECK("gn_finalize_fine");
}

__global__ void __launch_bounds__(256) adm_time_embed_kernel(const float* __restrict__ t, int t_stride, int mc, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, int dim_out, float* __restrict__ emb) {
    __shared__ float f[1024];
    __shared__ float hdn[2048];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float tv = t[(size_t)b * t_stride];
    const int half = mc / 2;
    for (int i = tid; i < half; i += 256) {                           // :41-46, cosines first
        const float ang = tv * expf(-9.210340371976184f * (float)i / (float)half);
        f[i] = cosf(ang);
        f[half + i] = sinf(ang);
    }
    if (tid == 0 && (mc & 1)) f[mc - 1] = 0.f;                         // :47-48
    __syncthreads();
    for (int j = tid; j < dim_out; j += 256) {
        float acc = b1[j];
        for (int i = 0; i < mc; ++i) acc = fmaf(w1[(size_t)j * mc + i], f[i], acc);
        hdn[j] = acc / (1.0f + expf(-acc));                            // nn.SiLU, :457
    }
    __syncthreads();
    for (int j = tid; j < dim_out; j += 256) {
        float acc = b2[j];
        for (int i = 0; i < dim_out; ++i) acc = fmaf(w2[(size_t)j * dim_out + i], hdn[i], acc);
        emb[(size_t)b * dim_out + j] = acc;
    }
}
const char* launch_adm_time_embed(const float* t, int t_stride, int nb, int mc, const float* w1, const float* b1, const float* w2,
                                  const float* b2, int dim_out, float* emb, hipStream_t s) {
    if (mc > 1024 || dim_out > 2048 || mc < 2) return "adm_time_embed: unsupported widths";
    hipLaunchKernelGGL(adm_time_embed_kernel, dim3(nb), dim3(256), 0, s, t, t_stride, mc, w1, b1, w2, b2, dim_out, emb);
    return C2_LAUNCH_CHECK("adm_time_embed");
}

__global__ void __launch_bounds__(256) add_rows_kernel(float* __restrict__ out, const float* __restrict__ a, int a_bs, const float* __restrict__ c, int c_bs, int n) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[(size_t)b * n + i] = a[(size_t)b * a_bs + i] + c[(size_t)b * c_bs + i];
}
const char* launch_add_rows(float* out, const float* a, int a_bstride, const float* c, int c_bstride, int B, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_rows_kernel, dim3(ceil_div(n, 256), B), dim3(256), 0, s, out, a, a_bstride, c, c_bstride, n);
    return C2_LAUNCH_CHECK("add_rows");
}

__global__ void __launch_bounds__(256) permute_qkv_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int heads, int d, int cols) {
    const long long total = (long long)3 * heads * d * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int col = (int)(idx % cols);
        const int row = (int)(idx / cols);                 // destination row = (which * heads + h) * d + c
        const int c = row % d, hh = (row / d) % heads, which = row / (d * heads);
        dst[idx] = src[((size_t)(hh * 3 + which) * d + c) * cols + col];
    }
}
const char* launch_permute_qkv_rows(const float* src, float* dst, int heads, int d, int cols, hipStream_t s) {
    const long long total = (long long)3 * heads * d * cols;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(permute_qkv_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, heads, d, cols);
    return C2_LAUNCH_CHECK("permute_qkv_rows");
}

template <typename T>
__global__ void __launch_bounds__(256) avgpool2_kernel(const T* __restrict__ x, const float* __restrict__ ab, int act, T* __restrict__ out, int H, int W, int C,
                                                       long long total) {
    constexpr int EPC = Elem<T>::kPerChunk;
    const int cpr = C / EPC, Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cpr) * EPC;
        long long r = i / cpr;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho);
        const long long b = r / Ho;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const u32x4_t v = *(const u32x4_t*)(x + ((b * H + 2 * yo + (d >> 1)) * W + 2 * xo + (d & 1)) * C + c);
            float f[EPC];
            unpack16<T>(v, f);
            if (ab) {
                const float* const t = ab + ((size_t)b * C + c) * 2;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float u = fmaf(f[e], t[2 * e], t[2 * e + 1]);
                    f[e] = act ? (sizeof(T) == 2 ? silu_f(u) : u / (1.0f + expf(-u))) : u;
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] *= 0.25f;
        *(u32x4_t*)(out + ((b * Ho + yo) * Wo + xo) * C + c) = pack16<T>(acc);
    }
}
const char* launch_avgpool2(const void* x, const float* ab, int act, void* out, int bf16, int B, int H, int W, int C, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc || H % 2 || W % 2 || H < 2 || W < 2) return "avgpool2: even H and W, channels a multiple of a 16-byte chunk";
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / epc);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (bf16) hipLaunchKernelGGL(avgpool2_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ab, act, (bf16_t*)out, H, W, C, total);
    else hipLaunchKernelGGL(avgpool2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ab, act, (float*)out, H, W, C, total);
    return C2_LAUNCH_CHECK("avgpool2");
}

__global__ void __launch_bounds__(256) nearest_up2_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ out, int H, int W, int cpr, long long total) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        long long r = i / cpr;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho);
        const long long b = r / Ho;
        out[i] = x[((b * H + (yo >> 1)) * W + (xo >> 1)) * cpr + c];
    }
}
const char* launch_nearest_up2(const void* x, void* out, int bf16, int B, int H, int W, int C, hipStream_t s) {
    const int epc = bf16 ? 8 : 4;
    if (C % epc) return "nearest_up2: channels must be a multiple of a 16-byte chunk";
    const long long total = (long long)B * (2 * H) * (2 * W) * (C / epc);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(nearest_up2_kernel, dim3(grid), dim3(256), 0, s, (const u32x4_t*)x, (u32x4_t*)out, H, W, C / epc, total);
    return C2_LAUNCH_CHECK("nearest_up2");
}

}  // namespace adf
