// Tile-shape dispatch for the fused implicit-GEMM kernel (see adf_gemm.h).
#include "adf_gemm.h"
#include "adf_gemm_pp.h"
#include "adf_gemm_rb.h"
#include "adf_gemm_rbx3.h"
#include "adf_gemm_up.h"
#include "adf_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>


#ifdef ADF_RB_STAMP
namespace adf { __device__ unsigned long long adf_rb_stamps[8 * 16]; }
extern "C" int adf_debug_rb_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(adf::adf_rb_stamps), sizeof(unsigned long long) * 8 * 16);
}
#endif

#ifdef ADF_RB_TL
namespace adf { __device__ unsigned adf_rb_tl[2 * 8 * 128]; }
extern "C" int adf_debug_rb_timeline(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(adf::adf_rb_tl), sizeof(unsigned) * 2 * 8 * 128);
}
#endif

namespace adf {

namespace {

template <typename T, int MT, int NT, int WM, int WN>
const char* launch_variant(const GemmArgs& a, hipStream_t stream) {
    constexpr int TM = 32 * MT * WM, TN = 32 * NT * WN, NTHR = 64 * WM * WN;
    constexpr int lds = gemm_lds_bytes<TM, TN>();
    static bool attr_done[kMaxDevices] = {};
    bool& attr_set = attr_done[current_device()];
    auto kern = conv_gemm_kernel<T, MT, NT, WM, WN>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        attr_set = true;
    }
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    const long long tiles_m = a.flat ? ((long long)a.B * a.mrows + TM - 1) / TM : (long long)((a.mrows + TM - 1) / TM) * a.B;
    const long long blocks = (long long)tiles_n * tiles_m;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return "conv_gemm: bad grid";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), lds, stream, a);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm: launch failed";
}

// Weight-stationary warp-specialised kernel (adf_gemm.h): persistent 512-thread blocks, one per CU.
int ws_lds_bytes(const GemmArgs& a, int tn) {
    long long wrows = 0;
    for (int s = 0; s < a.nseg; ++s) wrows += (long long)a.seg[s].nchunk * a.seg[s].taps * tn;
    const long long b = wrows * kLdsPitch + 2LL * kWsARows * kLdsPitch + kWsScratch;
    return b > 0x7fffffff ? 0x7fffffff : (int)b;
}

template <typename T, int NT, int WN>
const char* launch_ws_variant(const GemmArgs& a, hipStream_t stream) {
    constexpr int TN = NT * WN * 32;
    static bool attr_done[kMaxDevices] = {};
    bool& attr_set = attr_done[current_device()];
    static int num_cu_dev[kMaxDevices] = {};
    int& num_cu = num_cu_dev[current_device()];
    auto kern = conv_gemm_ws_kernel<T, NT, WN>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize, ws) failed";
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu < 1)
            num_cu = 256;
        attr_set = true;
    }
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    const long long tiles_m_total = (long long)((a.mrows + 127) / 128) * a.B;
    if (tiles_m_total <= 0 || tiles_m_total > 0x7fffffffLL) return "conv_gemm_ws: bad tile count";
    long long bpn = num_cu / tiles_n;
    if (bpn < 1) bpn = 1;
    if (bpn > tiles_m_total) bpn = tiles_m_total;
    hipLaunchKernelGGL(kern, dim3((unsigned)(bpn * tiles_n)), dim3(512), (size_t)ws_lds_bytes(a, TN), stream, a, (int)tiles_m_total, (int)bpn);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm_ws: launch failed";
}

// Transposed-conv kernel (adf_gemm_up.h): persistent 512-thread workgroups over tiles of m = 0 .. L
template <int CIN, int COUT, int F, int MTP>
const char* launch_up_mt(const GemmArgs& a, hipStream_t stream) {
    typedef UpCfg<CIN, COUT, F, MTP> Cfg;
    static bool attr_done[kMaxDevices] = {};
    bool& attr_set = attr_done[current_device()];
    auto kern = conv_gemm_up_kernel<CIN, COUT, F, MTP>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize, up) failed";
        attr_set = true;
    }
    static_assert(Cfg::kLds <= 160 * 1024, "up kernel LDS budget");
    const int tps = (a.mrows + Cfg::TM - 1) / Cfg::TM;
    const long long items = (long long)a.B * tps * Cfg::NPASS;
    if (items <= 0 || items > 0x7fffffffLL) return "conv_gemm_up: bad tile count";
    const long long grid = items < 256 ? items : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), Cfg::kLds, stream, a, (int)items, tps);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm_up: launch failed";
}
// tile height: the one that needs fewer rounds of 256 workgroups; ties go to the taller tile (weights are re-read per tile)
template <int CIN, int COUT, int F>
const char* launch_up(const GemmArgs& a, hipStream_t stream) {
    auto rounds = [&](int tm, int npass) { return ((long long)a.B * ((a.mrows + tm - 1) / tm) * npass + 255) / 256; };
    typedef UpCfg<CIN, COUT, F, 2> C2;
    typedef UpCfg<CIN, COUT, F, 4> C4;
    if (rounds(C4::TM, C4::NPASS) <= rounds(C2::TM, C2::NPASS)) return launch_up_mt<CIN, COUT, F, 4>(a, stream);
    return launch_up_mt<CIN, COUT, F, 2>(a, stream);
}

// Persistent LDS-DMA kernel (adf_gemm_pp.h): one 512-thread block per CU, block tile (128 MT) x 128.
template <int MT>
const char* launch_pp(const GemmArgs& a, hipStream_t stream) {
    static bool attr_done[kMaxDevices] = {};
    bool& attr_set = attr_done[current_device()];
    static int num_cu_dev[kMaxDevices] = {};
    int& num_cu = num_cu_dev[current_device()];
    auto kern = conv_gemm_pp_kernel<MT>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kPpLds) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize, pp) failed";
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu < 1)
            num_cu = 256;
        attr_set = true;
    }
    constexpr int TM = 128 * MT;
    const int tiles_m = a.mrows / TM, tiles_n = a.n_pad / kPpTN;
    int tm_shift = 0;
    while ((1 << tm_shift) < tiles_m) ++tm_shift;
    const long long tiles_total = (long long)a.B * tiles_m * tiles_n;
    if (tiles_total <= 0 || tiles_total > (1 << 22)) return "conv_gemm_pp: bad tile count";
    const long long grid = tiles_total < num_cu ? tiles_total : num_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), kPpLds, stream, a, (int)tiles_total, tm_shift, tiles_n);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm_pp: launch failed";
}

// What the two forms of the resblock conv kernel (adf_gemm_rb.h: bf16 rows, 64-channel K blocks; adf_gemm_rbx3.h: fp32 rows, 32-channel K blocks) share on the
// host: the shape checks, the K-block table of one tile and the argument head -- everything but the tile shape.  false: not a shape the kernel is written for.
struct RbForm {
    int esz;            // bytes per stored element
    int blk_ch;         // channels of a K block (a 128-byte row)
    int max_blk;        // blocks of the argument table
    bool even_pairs;    // bf16 form: an even number of 64-channel blocks per tile (the ring parity of the first block of a tile is a compile-time constant)
};
template <typename ArgsT>
static bool rb_build_args(const GemmArgs& a, const RbForm& f, ArgsT& r, bool& raw0) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const GemmSeg& g0 = a.seg[0];
    if (a.scatter_f || a.gelu || a.flat || a.mrows % 128 || a.lin != a.mrows || a.out_rows != a.mrows) return false;
    if (a.n != a.n_pad || a.out_c != a.n || (a.n != 128 && a.n != 256) || a.bias_mod != a.n) return false;
    if (!pow2(a.mrows / 128)) return false;
    // segment 0: GroupNorm + SiLU with the table derived in the kernel, or raw (the folded down convs, the 3-tap form of a transposed conv)
    raw0 = !g0.gn.gamma && !g0.ab && !g0.act && g0.c1 == 0;
    if (!raw0 && (!g0.gn.gamma || !g0.act)) return false;
    if (g0.taps != 3 || g0.off0 != -1 || g0.stride != 1 || g0.step != 1) return false;
    auto seg_ok = [&](int c0, int c1, int nb_before) {
        if (c0 % 64 || c1 % 64 || nb_before + (c0 + c1) / f.blk_ch > f.max_blk) return false;
        return !(f.even_pairs && ((c0 + c1) / 64) % 2);
    };
    if (!seg_ok(g0.c0, g0.c1, 0)) return false;
    if (!raw0 && g0.c0 + g0.c1 > kPpMaxCin) return false;
    if (!raw0) {
        // the kernel's table fill sums one or two stored (fine) statistics groups per GroupNorm group: one source, or two equal ones
        const GnFinalizeArgs& gn = g0.gn;
        if (gn.G < 1 || (gn.c0 + gn.c1) % gn.G || gn.c0 % gn.G || gn.c1 % gn.G) return false;
        const int gs = (gn.c0 + gn.c1) / gn.G;
        if (gn.c0 % gs) return false;                                       // a group must not straddle the two sources
        for (int cs : {gn.c0, gn.c1}) {
            if (cs == 0) continue;
            const int fg = cs / gn.G;
            if (gs != fg && gs != 2 * fg) return false;
        }
    }
    if (raw0 && (a.nseg > 1 || a.res)) return false;
    if (a.res && a.nseg > 1) return false;
    memset(&r, 0, sizeof(r));
    int nb = 0;
    auto add_seg = [&](const void* s0, const void* s1, int c0, int c1, const void* w, int taps, bool table, float scale1) {
        for (int c = 0; c < c0 + c1; c += f.blk_ch, ++nb) {
            const bool from1 = c >= c0;
            RbBlk& e = r.blk[nb];
            e.src = (const char*)(from1 ? s1 : s0) + (size_t)(from1 ? c - c0 : c) * f.esz;
            e.pitch = (unsigned)(from1 ? c1 : c0) * (unsigned)f.esz;
            e.w = (const char*)w + (size_t)(c / f.blk_ch) * taps * a.n_pad * kRowBytes;
            e.tab = table ? c * 8 : -1;
            e.scale = from1 ? scale1 : 1.0f;
        }
    };
    add_seg(g0.src0, g0.src1, g0.c0, g0.c1, g0.w, 3, !raw0, 1.0f);
    r.h.nb3 = nb;
    if (a.nseg > 1) {
        const GemmSeg& g1 = a.seg[1];
        if (g1.taps != 1 || g1.off0 != 0 || g1.stride != 1 || g1.step != 1 || g1.ab || g1.gn.gamma || g1.act) return false;
        if (!seg_ok(g1.c0, g1.c1, nb)) return false;
        add_seg(g1.src0, g1.src1, g1.c0, g1.c1, g1.w, 1, false, g1.scale1);
    }
    r.h.res = a.nseg == 1 ? a.res : nullptr;                             // identity residual: added in the epilogue (fp32, before the rounding)
    r.h.nb1 = nb - r.h.nb3;
    r.h.B = a.B; r.h.L = a.mrows;
    r.h.n = a.n;
    r.h.gn = g0.gn;
    r.h.bias0 = a.bias0; r.h.bias1 = a.bias1;
    r.h.out = a.out;
    r.h.stats = nullptr; r.h.stats_groups = 0;
    if (a.phase_c && (!raw0 || (a.phase_c & (a.phase_c - 1)) || a.phase_c < 64 || a.n % a.phase_c)) return false;
    if (a.stats) {
        const int sc = a.phase_c ? a.phase_c : a.out_c;                      // channels the statistics are over
        const int gs = a.stats_groups > 0 ? sc / a.stats_groups : 0;
        if (!(gs > 0 && gs * a.stats_groups == sc && (gs & (gs - 1)) == 0 && gs >= 8 && gs <= 64)) return false;
        r.h.stats = a.stats; r.h.stats_groups = a.stats_groups;
        r.h.stats_mod = a.phase_c;
    }
    return true;
}
static int rb_num_cus() {
    static int num_cu_dev[kMaxDevices] = {};
    int& n = num_cu_dev[current_device()];
    if (n == 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || n < 1)) n = 256;
    return n;
}

// Resblock conv kernel (adf_gemm_rb.h): fills the K-block table of one tile and launches one 512-thread block per CU.
// Returns false (and launches nothing) when the shape is not one the kernel is written for.
bool try_launch_rb(const GemmArgs& a, const void* ident, long long min_tiles, hipStream_t stream, const char** err, bool dry = false) {
    *err = nullptr;
    (void)ident;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    RbArgs r;
    bool raw0 = false;
    if (!rb_build_args(a, RbForm{2, 64, kRbMaxBlk, true}, r, raw0)) return false;
    // Tile shape.  n = 256: one 256 x 256 tile per 256 rows (the activations are fetched and activated once) when that still gives
    // every CU a tile, else two 256 x 128 tiles; when even those leave CUs idle (L = 256 at batch 64), 128-row tiles.
    // ADF_RB_M128_NH=2 (A/B): 128 x 256 tiles there -- half as many thread blocks, each activation prepared once.
    static long long m128_nh = -1;
    if (m128_nh < 0) m128_nh = adf_tuning("ADF_RB_M128_NH", 1);
    auto shape = [&](int tm_, int& nh_) -> long long {       // thread-block tiles at tile height tm_ (0: the rows do not divide)
        if (a.mrows % tm_ || !pow2(a.mrows / tm_)) return 0;
        const long long tiles_m = (long long)a.B * (a.mrows / tm_);
        nh_ = (a.n == 256 && tiles_m >= min_tiles) ? 2 : 1;
        return tiles_m * (a.n / (kPpTN * nh_));
    };
    int tm = 256, nh = 1;
    long long tiles_total = shape(256, nh);
    bool wide128 = false;
    if (tiles_total < min_tiles) {
        tm = 128;
        tiles_total = shape(128, nh);
        if (m128_nh == 2 && a.n == 256 && nh == 1 && tiles_total >= min_tiles) { nh = 2; tiles_total /= 2; wide128 = true; }
    }
    if (tiles_total > (1 << 22) || tiles_total < (wide128 ? min_tiles / 2 : min_tiles)) return false;     // fewer tiles than CUs even on 128-row tiles: the other routes
    r.h.tiles_n = a.n / (kPpTN * nh);
    r.h.tm_shift = 0;
    while ((1 << r.h.tm_shift) < a.mrows / tm) ++r.h.tm_shift;
    r.h.tiles_total = (int)tiles_total;
    if (dry) return true;
    static bool attr_done[kMaxDevices] = {};
    const int dev = current_device();
    if (!attr_done[dev]) {
        bool ok = true;
        for (const void* k : {(const void*)conv_gemm_rb_kernel<1, false, 2>, (const void*)conv_gemm_rb_kernel<2, false, 2>, (const void*)conv_gemm_rb_kernel<1, true, 2>,
                              (const void*)conv_gemm_rb_kernel<2, true, 2>, (const void*)conv_gemm_rb_kernel<1, false, 1>, (const void*)conv_gemm_rb_kernel<2, false, 1>,
                              (const void*)conv_gemm_rb_kernel<1, true, 1>, (const void*)conv_gemm_rb_kernel<2, true, 1>})
            ok = ok && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kRbLds) == hipSuccess;
        if (!ok) {
            *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize, rb) failed";
            return true;
        }
        attr_done[dev] = true;
    }
    const int ncu = rb_num_cus();
    const long long grid = tiles_total < ncu ? tiles_total : ncu;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), kRbLds, stream, r); };
    if (tm == 256) {
        if (nh == 2 && raw0) go(conv_gemm_rb_kernel<2, true, 2>);
        else if (nh == 2) go(conv_gemm_rb_kernel<2, false, 2>);
        else if (raw0) go(conv_gemm_rb_kernel<1, true, 2>);
        else go(conv_gemm_rb_kernel<1, false, 2>);
    } else {
        if (nh == 2 && raw0) go(conv_gemm_rb_kernel<2, true, 1>);
        else if (nh == 2) go(conv_gemm_rb_kernel<2, false, 1>);
        else if (raw0) go(conv_gemm_rb_kernel<1, true, 1>);
        else go(conv_gemm_rb_kernel<1, false, 1>);
    }
    if (hipGetLastError() != hipSuccess) *err = "conv_gemm_rb: launch failed";
    return true;
}

// Split-bf16 form of the resblock conv kernel (adf_gemm_rbx3.h): fp32 storage, 32-channel K blocks, one 128-column N tile per workgroup tile.
// Same shapes as try_launch_rb; returns false (and launches nothing) when the shape is not one the kernel is written for.
bool try_launch_rbx3(const GemmArgs& a, long long min_tiles, hipStream_t stream, const char** err, bool dry = false) {
    *err = nullptr;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    Rbx3Args r;
    bool raw0 = false;
    if (!rb_build_args(a, RbForm{4, 32, kRbx3MaxBlk, false}, r, raw0)) return false;
    auto tiles_at = [&](int tm_) -> long long {
        if (a.mrows % tm_ || !pow2(a.mrows / tm_)) return 0;
        return (long long)a.B * (a.mrows / tm_) * (a.n / kPpTN);
    };
    int tm = 256;
    long long tiles_total = tiles_at(256);
    if (tiles_total < min_tiles) { tm = 128; tiles_total = tiles_at(128); }
    if (tiles_total > (1 << 22) || tiles_total < min_tiles) return false;
    r.h.tiles_n = a.n / kPpTN;
    r.h.tm_shift = 0;
    while ((1 << r.h.tm_shift) < a.mrows / tm) ++r.h.tm_shift;
    r.h.tiles_total = (int)tiles_total;
    if (dry) return true;
    static bool attr_done[kMaxDevices] = {};
    const int dev = current_device();
    if (!attr_done[dev]) {
        bool ok = true;
        for (const void* k : {(const void*)conv_gemm_rbx3_kernel<2>, (const void*)conv_gemm_rbx3_kernel<1>})
            ok = ok && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kRbLds) == hipSuccess;
        if (!ok) {
            *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize, rbx3) failed";
            return true;
        }
        attr_done[dev] = true;
    }
    const int ncu = rb_num_cus();
    const long long grid = tiles_total < ncu ? tiles_total : ncu;
    if (tm == 256) hipLaunchKernelGGL(conv_gemm_rbx3_kernel<2>, dim3((unsigned)grid), dim3(512), kRbLds, stream, r);
    else hipLaunchKernelGGL(conv_gemm_rbx3_kernel<1>, dim3((unsigned)grid), dim3(512), kRbLds, stream, r);
    if (hipGetLastError() != hipSuccess) *err = "conv_gemm_rbx3: launch failed";
    return true;
}

// shapes the pipelined kernel is written for (see the header of adf_gemm_pp.h); an identity residual counts as a
// second segment
bool pp_eligible(const GemmArgs& a, int tm) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (a.scatter_f || a.mrows % tm || a.lin != a.mrows || a.out_rows != a.mrows) return false;
    if (a.n != a.n_pad || a.out_c != a.n || a.n_pad % kPpTN || a.n_pad > kPpMaxN) return false;
    if (!pow2(a.mrows / tm)) return false;
    if (a.res && a.nseg > 1) return false;
    int nb = a.res ? a.n / 64 : 0;
    for (int s = 0; s < a.nseg; ++s) {
        const GemmSeg& g = a.seg[s];
        if (g.stride != 1 || g.step != 1) return false;
        if (!((g.taps == 3 && g.off0 == -1) || (g.taps == 1 && g.off0 == 0))) return false;
        if (g.c0 % 64 || g.c1 % 64 || (g.ab && g.c0 + g.c1 > kPpMaxCin)) return false;   // the LDS affine table holds kPpMaxCin channels
        if (s == 1 && (g.ab || g.act)) return false;
        nb += (g.c0 + g.c1) / 64;
    }
    return nb >= 2;
}

// packed bf16 identity [n/64 chunks][1 tap][n rows][64 channels]: row r of chunk c holds 1.0 at channel r - 64 c
const void* pp_identity(int n, hipStream_t stream, const char** err) {
    static void* cache_dev[kMaxDevices][kPpMaxN / 64 + 1] = {};     // one copy per device (freed at process exit)
    void** cache = cache_dev[current_device()];
    const int idx = n / 64;
    if (cache[idx]) return cache[idx];
    const size_t elems = (size_t)(n / 64 + kTapGroup) * n * 64;     // over-allocated like every packed weight
    std::vector<uint16_t> hbuf(elems, 0);
    for (int r = 0; r < n; ++r) hbuf[((size_t)(r / 64) * n + r) * 64 + (r % 64)] = 0x3F80;
    void* d = nullptr;
    if (hipMalloc(&d, elems * 2) != hipSuccess) { *err = "conv_gemm_pp: hipMalloc(identity) failed"; return nullptr; }
    if (hipMemcpyAsync(d, hbuf.data(), elems * 2, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        *err = "conv_gemm_pp: identity upload failed";
        return nullptr;
    }
    cache[idx] = d;
    return d;
}

template <typename T, int MT, int NT>
const char* launch_ksplit(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = 4 * ks_wave_lds(MT, NT);
    constexpr int TM = 32 * MT, TN = 32 * NT;
    static bool attr_done[kMaxDevices] = {};
    bool& attr_set = attr_done[current_device()];
    auto kern = conv_gemm_ksplit_kernel<T, MT, NT>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize, ksplit) failed";
        attr_set = true;
    }
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    const long long tiles_m = a.flat ? ((long long)a.B * a.mrows + TM - 1) / TM : (long long)((a.mrows + TM - 1) / TM) * a.B;
    const long long blocks = tiles_m * tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return "conv_gemm_ksplit: bad grid";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm_ksplit: launch failed";
}

template <typename T>
const char* dispatch(const GemmArgs& a, int tm, int tn, hipStream_t s) {
    if (tm == 128 && tn == 128) return launch_variant<T, 2, 2, 2, 2>(a, s);
    if (tm == 128 && tn == 64) return launch_variant<T, 2, 1, 2, 2>(a, s);
    if (tm == 128 && tn == 32) return launch_variant<T, 1, 1, 4, 1>(a, s);
    if (tm == 64 && tn == 128) return launch_variant<T, 1, 2, 2, 2>(a, s);
    if (tm == 64 && tn == 64) return launch_variant<T, 1, 1, 2, 2>(a, s);
    if (tm == 64 && tn == 32) return launch_variant<T, 1, 1, 2, 1>(a, s);
    if (tm == 32 && tn == 128) return launch_variant<T, 1, 1, 1, 4>(a, s);
    if (tm == 32 && tn == 64) return launch_variant<T, 1, 1, 1, 2>(a, s);
    return launch_variant<T, 1, 1, 1, 1>(a, s);
}

}  // namespace

// ADF_GEMM_TRACE=1: print the kernel chosen for every launch (stderr)
static void trace_route(const char* route, const GemmArgs& a, int tm, int tn) {
    static int on = -1;
    if (on < 0) on = adf_route_switch("ADF_GEMM_TRACE", 0);
    if (!on) return;
    fprintf(stderr, "[adf gemm] %-6s B=%d lin=%d mrows=%d n=%d/%d nseg=%d seg0(c=%d+%d taps=%d stride=%d off0=%d step=%d ab=%d act=%d)", route, a.B, a.lin, a.mrows,
            a.n, a.n_pad, a.nseg, a.seg[0].c0, a.seg[0].c1, a.seg[0].taps, a.seg[0].stride, a.seg[0].off0, a.seg[0].step, a.seg[0].ab != nullptr, a.seg[0].act);
    if (a.nseg > 1) fprintf(stderr, " seg1(c=%d+%d taps=%d ab=%d)", a.seg[1].c0, a.seg[1].c1, a.seg[1].taps, a.seg[1].ab != nullptr);
    fprintf(stderr, " res=%d gelu=%d scatter=%d out=%dx%d stats=%d flat=%d tile=%dx%d\n", a.res != nullptr, a.gelu, a.scatter_f, a.out_rows, a.out_c, a.stats != nullptr,
            a.flat, tm, tn);
}

static long long rb_min_tiles() {
    static int use_rb = -1;
    if (use_rb < 0) use_rb = adf_route_switch("ADF_GEMM_RB", 1);
    return use_rb == 0 ? -1 : (use_rb >= 2 ? 32 : 256);        // ADF_GEMM_RB=2: also small batches (tests); 0: the route is off
}
static long long rbx3_min_tiles() {
    static int use_rbx3 = -1;
    if (use_rbx3 < 0) use_rbx3 = adf_route_switch("ADF_GEMM_RBX3", 1);
    return use_rbx3 == 0 ? -1 : (use_rbx3 >= 2 ? 32 : 256);    // ADF_GEMM_RBX3=2: also small batches (tests); 0: the route is off
}
bool conv_gemm_phase_eligible(const GemmArgs& a, int dtype) {
    const long long mt = dtype == 2 ? rbx3_min_tiles() : rb_min_tiles();
    if (dtype == 0 || mt < 0 || !a.phase_c || a.flat) return false;
    const char* err = nullptr;
    return dtype == 2 ? try_launch_rbx3(a, mt, nullptr, &err, true) : try_launch_rb(a, nullptr, mt, nullptr, &err, true);
}

const char* launch_conv_gemm(const GemmArgs& a_in, int dtype, hipStream_t stream, bool* stats_fused) {
    // dtype: 0 = fp32 storage + exact-fp32 MFMA, 1 = bf16 storage, 2 = fp32 storage + split-bf16 operands (f32x3_t: the generic and split-K kernels only)
    const bool dtype_bf16 = dtype == 1, x3 = dtype == 2;
    GemmArgs a = a_in;
    if (stats_fused) *stats_fused = false;
    if (a.phase_c) {                 // only the resblock conv kernel's raw form knows the phase-major statistics
        // what try_launch_rb accepts drives the answer: declined with a statistics request (group size), it is asked again without one and the caller
        // runs the separate statistics pass (stats_fused stays false)
        const long long mt = x3 ? rbx3_min_tiles() : rb_min_tiles();
        if (dtype == 0 || mt < 0 || a.flat) return "conv_gemm: a 3-tap-form transposed conv that conv_gemm_rb_kernel does not take";
        const char* err = nullptr;
        auto go = [&]() { return x3 ? try_launch_rbx3(a, mt, stream, &err) : try_launch_rb(a, nullptr, mt, stream, &err); };
        bool taken = go();
        if (!taken && a.stats) { a.stats = nullptr; taken = go(); }
        if (!taken) return "conv_gemm: a 3-tap-form transposed conv that conv_gemm_rb_kernel does not take";
        if (stats_fused) *stats_fused = a.stats != nullptr;
        trace_route(x3 ? "rbx3" : "rb", a, 256, 128);
        return err;
    }
    if (a.nseg < 1 || a.nseg > 2) return "conv_gemm: nseg must be 1 or 2";
    if (a.n_pad % 32) return "conv_gemm: n_pad must be a multiple of 32";
    const int epc = dtype_bf16 ? 8 : 4;
    if (a.n % epc || a.out_c % epc) return "conv_gemm: output channels must be a multiple of a 16-byte chunk";
    const long long esz = dtype_bf16 ? 2 : 4;
    if ((long long)a.B * a.out_rows * a.out_c * esz >= (1LL << 32)) return "conv_gemm: output tensor must be < 4 GiB";
    // GroupNorm affine of segment 0 still to be derived from the statistics: the DMA kernel does it itself (one launch
    // less per GroupNorm: 4.7 us each, 44 per network pass before), every other route gets gn_finalize launched here
    const bool gn_pending = a.seg[0].gn.gamma != nullptr;
    static int gn_in_kernel = -1;     // ADF_GEMM_GN=0: always launch gn_finalize (A/B)
    if (gn_in_kernel < 0) gn_in_kernel = (int)adf_tuning("ADF_GEMM_GN", 1);
    auto settle_gn = [&](bool in_kernel) -> const char* {
        if (!gn_pending) return nullptr;
        in_kernel = in_kernel && gn_in_kernel;
        const char* err = nullptr;
        if (!in_kernel) {
            if (!a.gn_ready) err = launch_gn_finalize(a.seg[0].gn, stream);
            a.seg[0].gn.gamma = nullptr;
        }
        return err;
    };
    if (a.nseg > 1) a.seg[1].gn.gamma = nullptr;
    bool raw = true;
    for (int s = 0; s < a.nseg; ++s) {
        const GemmSeg& g = a.seg[s];
        if (g.c0 % epc || g.c1 % epc) return "conv_gemm: channel counts must be multiples of a 16-byte chunk";
        if (g.step != 1 && g.step != -1) return "conv_gemm: step must be +-1";
        if (g.taps < 1 || g.stride < 1) return "conv_gemm: bad taps/stride";
        if (g.ab) raw = false;
        if ((long long)a.B * a.lin * (g.c0 > g.c1 ? g.c0 : g.c1) * esz >= (1LL << 32)) return "conv_gemm: input tensor must be < 4 GiB";
    }
    // Tile selection.  Per-sample tiles need (TM-1)*stride + taps staged rows; flat tiles (several whole
    // samples per tile, raw inputs only) need (TM/mrows) * ((mrows-1)*stride + taps).
    const bool can_flat = raw && !a.scatter_f && a.lin == a.mrows && (a.mrows & (a.mrows - 1)) == 0;
    int tm = 0, flat = 0;
    for (int cand = 128; cand >= 32 && !tm; cand >>= 1) {
        if (can_flat && a.mrows < cand) {
            bool fits = true;
            for (int s = 0; s < a.nseg; ++s)
                if ((cand / a.mrows) * ((a.mrows - 1) * a.seg[s].stride + a.seg[s].taps) > kARows) fits = false;
            if (fits) { tm = cand; flat = 1; continue; }
        }
        bool fits = true;
        for (int s = 0; s < a.nseg; ++s)
            if ((cand - 1) * a.seg[s].stride + a.seg[s].taps > kARows) fits = false;
        if (!fits) continue;
        if (cand > 32 && a.mrows <= cand / 2) continue;
        tm = cand;
    }
    if (!tm) return "conv_gemm: no tile shape fits (stride/taps too large)";
    int tn = a.n_pad >= 128 ? 128 : (a.n_pad >= 64 ? 64 : 32);
    if (a.n_pad % tn && a.n_pad % 64 == 0) tn = 64;
    // keep the 256 CUs busy when the problem is small: prefer narrower N tiles, then shorter M tiles
    auto nblocks = [&](int tm_, int tn_) {
        const long long tmn = flat ? ((long long)a.B * a.mrows + tm_ - 1) / tm_ : (long long)((a.mrows + tm_ - 1) / tm_) * a.B;
        return tmn * ((a.n_pad + tn_ - 1) / tn_);
    };
    static int min_blocks = -1;
    if (min_blocks < 0) min_blocks = (int)adf_tuning("ADF_GEMM_MINBLOCKS", 512);
    while (nblocks(tm, tn) < min_blocks && tn > 32) tn >>= 1;
    while (nblocks(tm, tn) < min_blocks && tm > 32) {
        tm >>= 1;
        if (flat && a.mrows >= tm) flat = 0;   // a tile now lies inside one sample again
    }
    a.flat = flat;
    a.seg_rows = flat ? a.mrows : tm;
    if (a.stats) {
        // the epilogue reduces statistics per thread-column and wave: see adf_gemm.h phase 2
        const int nthr = (tm == 64 && tn == 32) || (tm == 32 && tn == 64) ? 128 : ((tm == 32 && tn == 32) ? 64 : 256);
        const int rpk = nthr * epc / tn;
        const int gs = a.stats_groups > 0 ? a.out_c / a.stats_groups : 0;
        const bool ok = gs > 0 && gs * a.stats_groups == a.out_c && (gs & (gs - 1)) == 0 && gs >= epc && gs <= tn &&
                        (!flat || a.mrows % rpk == 0);
        if (!ok) a.stats = nullptr;
        else if (stats_fused) *stats_fused = true;
    }
    {
        // transposed convs of the up path in bf16 (adf_gemm_up.h); ADF_GEMM_UP=0 leaves them to the plain / weight-stationary kernels
        static int use_up = -1;
        if (use_up < 0) use_up = adf_route_switch("ADF_GEMM_UP", 1);
        const GemmSeg& g = a.seg[0];
        const int f = a.scatter_f, cout = a.out_c, cin = g.c0;
        if (use_up && dtype_bf16 && (f == 2 || f == 4) && a.nseg == 1 && g.taps == 2 && g.stride == 1 && g.off0 == 0 && g.step == -1 && !g.ab &&
            !gn_pending && !g.act && g.scale1 == 1.0f && g.wfrag && g.c1 == 0 && a.n == f * cout && a.n == a.n_pad && a.mrows == a.lin + 1 &&
            a.out_rows == a.lin * f && a.scatter_pad == f / 2 && a.bias_mod == cout && !a.res && !a.gelu && !a.bias1 &&
            (!a_in.stats || a.stats_groups == 8)) {
            int cfg = 0;
            // measured (us per launch, this kernel vs the plain / weight-stationary route): 256 -> 256 x4 at L = 16 / 64 / 256: 28 each vs
            // 17 / 27 / 65; 256 -> 128 x2 at L = 1024: 47 vs 66; 128 -> 128 x2 at L = 2048 and 128 -> 64 x2 at L = 4096: 57 / 53 vs 57 / 53
            // (their tiles are bound by the per-CU HBM fetch rate, which this kernel does not overlap with the MFMAs): those two
            // stay on the old routes unless ADF_GEMM_UP=2
            if (cin == 256 && cout == 256 && f == 4) cfg = 1;
            else if (cin == 256 && cout == 128 && f == 2) cfg = 2;
            else if (use_up >= 2 && cin == 128 && cout == 128 && f == 2) cfg = 3;
            else if (use_up >= 2 && cin == 128 && cout == 64 && f == 2) cfg = 4;
            if (cfg) {
                a.stats = a_in.stats;
                if (stats_fused && a_in.stats) *stats_fused = true;
                trace_route("up", a, 64, 256);
                if (cfg == 1) return launch_up<256, 256, 4>(a, stream);
                if (cfg == 2) return launch_up<256, 128, 2>(a, stream);
                if (cfg == 3) return launch_up<128, 128, 2>(a, stream);
                return launch_up<128, 64, 2>(a, stream);
            }
        }
    }
    {
        // short levels (few rows, long K): intra-block split-K, 32 x 32 tiles, 4 waves x K/4 each
        static int use_ks = -1;
        if (use_ks < 0) use_ks = (int)adf_tuning("ADF_GEMM_KSPLIT", 1);
        int nit_total = 0;
        static long long ks_rows = -1;      // ADF_GEMM_KSPLIT_ROWS: largest B * rows the split-K kernel takes
        if (ks_rows < 0) ks_rows = adf_tuning("ADF_GEMM_KSPLIT_ROWS", 4096);
        bool ks_ok = use_ks && !a.scatter_f && (long long)a.B * a.mrows <= ks_rows;
        const int ks_flat = (can_flat && a.mrows < 32 && 32 % a.mrows == 0) ? 1 : 0;
        const int ks_seg = ks_flat ? a.mrows : 32;
        if (!ks_flat && !raw && a.mrows < 32) ks_ok = false;     // per-sample tiles of a tiny sample: leave to the plain kernel
        for (int s = 0; s < a.nseg; ++s) {
            const GemmSeg& g = a.seg[s];
            if (g.stride != 1 || g.taps > kTapGroup) ks_ok = false;
            if ((32 / ks_seg) * ((ks_seg - 1) + g.taps) > ks_a_rows(1)) ks_ok = false;
            nit_total += g.nchunk;
        }
        static int ks_minit = -1;           // ADF_GEMM_KSPLIT_MINIT: fewest 64-channel K chunks (all segments) for which K is split over the waves
        if (ks_minit < 0) ks_minit = (int)adf_tuning("ADF_GEMM_KSPLIT_MINIT", 4);
        if (ks_ok && nit_total >= ks_minit) {
            // 64 x 64 tiles when they still give >= 128 blocks: every tile row re-reads all weights and every tile column
            // all activations (from L2), so the bytes a CU pulls halve against 32 x 32 (ADF_GEMM_KSPLIT=32 forces the small tile)
            const bool big = use_ks != 32 && !ks_flat && a.mrows % 64 == 0 && a.n_pad % 64 == 0 &&
                             (long long)a.B * (a.mrows / 64) * (a.n_pad / 64) >= 128;
            const int tile = big ? 64 : 32;
            a.flat = ks_flat;
            a.seg_rows = big ? 64 : ks_seg;
            if (a_in.stats) {
                const int gs = a.stats_groups > 0 ? a.out_c / a.stats_groups : 0;
                const int rows_per_wave = 64 / (tile / epc);
                const bool ok = gs > 0 && gs * a.stats_groups == a.out_c && (gs & (gs - 1)) == 0 && gs >= epc && gs <= tile &&
                                (a.seg_rows % rows_per_wave == 0);
                a.stats = ok ? a_in.stats : nullptr;
                if (stats_fused) *stats_fused = ok;
            }
            if (const char* e = settle_gn(false)) return e;
            trace_route("ksplit", a, tile, tile);
            if (big) return dtype_bf16 ? launch_ksplit<bf16_t, 2, 2>(a, stream) : (x3 ? launch_ksplit<f32x3_t, 2, 2>(a, stream) : launch_ksplit<float, 2, 2>(a, stream));
            return dtype_bf16 ? launch_ksplit<bf16_t, 1, 1>(a, stream) : (x3 ? launch_ksplit<f32x3_t, 1, 1>(a, stream) : launch_ksplit<float, 1, 1>(a, stream));
        }
    }
    {
        // large stride-1 bf16 layers: persistent LDS-DMA 256 x 128 kernel (adf_gemm_pp.h).  Measured on MI355X
        // (profiles/README.md): 12-20 % faster than the other routes where it applies by default -- >= 256 tiles and
        // no identity residual (those layers keep their weights resident in the weight-stationary kernel, which wins).
        // ADF_GEMM_PP=0 disables it, =2 also takes identity-residual layers and >= 128 tiles (used by the tests).
        static int use_pp = -1;
        if (use_pp < 0) use_pp = adf_route_switch("ADF_GEMM_PP", 1);
        // tile height: 256 rows when that gives every CU a tile, else 128 rows (the L = 256 level at batch 64)
        int ptm = 0;
        if (use_pp && dtype_bf16 && !flat && tm == 128) {
            const long long t256 = pp_eligible(a, 256) ? (long long)a.B * (a.mrows / 256) * (a.n_pad / kPpTN) : 0;
            const long long t128 = pp_eligible(a, 128) ? (long long)a.B * (a.mrows / 128) * (a.n_pad / kPpTN) : 0;
            const long long need = use_pp >= 2 ? 128 : 256;
            if (t256 >= need) ptm = 256;
            else if (t128 >= need && use_pp != 3) ptm = 128;       // ADF_GEMM_PP=3: 256-row tiles only (A/B)
        }
        {
            // resblock convs (GroupNorm + SiLU prologue derived in the kernel) on 256-row tiles: adf_gemm_rb.h.  ADF_GEMM_RB=0
            // leaves them to the routes below (A/B; the tests compare the two).
            static int use_rb = -1;
            if (use_rb < 0) use_rb = adf_route_switch("ADF_GEMM_RB", 1);
            const bool rb_raw = !gn_pending && !a.seg[0].ab && !a.seg[0].act && a.nseg == 1 && !a.res && a.seg[0].taps == 3;
            // fp32 storage, split-bf16 products: the same data path on 32-channel blocks (adf_gemm_rbx3.h).  ADF_GEMM_RBX3=0: the generic kernel (A/B)
            const long long rbx3_mt = rbx3_min_tiles();
            if (rbx3_mt >= 0 && x3 && !flat && ((gn_pending && gn_in_kernel && !a.gn_ready) || rb_raw)) {
                const char* err = nullptr;
                GemmArgs b = a;
                b.stats = a_in.stats;
                if (try_launch_rbx3(b, rbx3_mt, stream, &err)) {
                    if (stats_fused) *stats_fused = a_in.stats != nullptr;
                    trace_route("rbx3", a, 256, 128);
                    return err;
                }
            }
            if (use_rb && dtype_bf16 && !flat && ((gn_pending && gn_in_kernel && !a.gn_ready) || rb_raw)) {
                const char* err = nullptr;
                const void* ident = nullptr;
                GemmArgs b = a;
                b.stats = a_in.stats;
                if (try_launch_rb(b, ident, use_rb >= 2 ? 32 : 256, stream, &err)) {     // ADF_GEMM_RB=2: also small batches (tests)
                    if (stats_fused) *stats_fused = a_in.stats != nullptr;
                    trace_route("rb", a, 256, 128);
                    return err;
                }
            }
        }
        if (ptm) {
            // identity-residual 3-tap layers stay with the weight-stationary kernel; the transformer's 1x1 projections
            // can come here with ADF_GEMM_PP_LINEAR=1 (residual as an identity K segment, GELU in the epilogue): measured
            // equal to the plain kernel end to end (388.6-393.9 vs 389.6-390.9 ms), so they stay there by default
            static int pp_linear = -1;
            if (pp_linear < 0) pp_linear = (int)adf_tuning("ADF_GEMM_PP_LINEAR", 0);
            const bool linear = a.seg[0].taps == 1 && a.nseg == 1;
            const bool take = linear ? pp_linear != 0 : (use_pp == 2 || !a.res);
            if (take) {
                if (a_in.stats) {
                    const int gs = a.stats_groups > 0 ? a.out_c / a.stats_groups : 0;
                    const bool ok = gs > 0 && gs * a.stats_groups == a.out_c && (gs & (gs - 1)) == 0 && gs >= 8 && gs <= 64;
                    a.stats = ok ? a_in.stats : nullptr;
                    if (stats_fused) *stats_fused = ok;
                }
                if (a.res) {                     // identity residual = a raw 1-tap segment against the packed identity
                    const char* err = nullptr;
                    const void* ident = pp_identity(a.n, stream, &err);
                    if (!ident) return err;
                    GemmSeg& g = a.seg[1];
                    g = GemmSeg{};
                    g.src0 = a.res; g.src1 = nullptr; g.c0 = a.n; g.c1 = 0; g.ab = nullptr; g.scale1 = 1.0f; g.act = 0;
                    g.taps = 1; g.stride = 1; g.off0 = 0; g.step = 1; g.w = ident; g.nchunk = a.n / 64;
                    a.nseg = 2;
                    a.res = nullptr;
                }
                if (const char* e = settle_gn(a.seg[0].taps == 3)) return e;
                trace_route("pp", a, ptm, 128);
                return ptm == 256 ? launch_pp<2>(a, stream) : launch_pp<1>(a, stream);
            }
        }
    }
    {
        // large stride-1 layers: weight-stationary persistent kernel when the weights of an N tile fit in LDS
        static int use_ws = -1;
        if (use_ws < 0) use_ws = adf_route_switch("ADF_GEMM_WS", 1);
        bool ws_ok = use_ws && !x3 && !flat && tm == 128 && a.n_pad >= 64;
        for (int s = 0; s < a.nseg; ++s)
            if (a.seg[s].stride != 1 || 127 + a.seg[s].taps > kWsARows) ws_ok = false;
        const long long tiles_m_total = (long long)((a.mrows + 127) / 128) * a.B;
        static long long ws_min_m = -1;     // ADF_GEMM_WS_MINM: fewest 128-row tiles for the weight-stationary kernel
        if (ws_min_m < 0) ws_min_m = adf_tuning("ADF_GEMM_WS_MINM", 256);
        if (ws_ok && tiles_m_total >= ws_min_m) {
            // measured on MI355X: the weight-stationary kernel wins with 128-wide N tiles (Cin = Cout = 128 layers);
            // with 64-wide tiles (ADF_GEMM_WS=64 to force) the doubled activation staging loses to the plain kernel
            int wtn = 0;
            if (a.n_pad >= 128 && ws_lds_bytes(a, 128) <= 160 * 1024) wtn = 128;
            else if (use_ws == 64 && ws_lds_bytes(a, 64) <= 160 * 1024) wtn = 64;
            if (wtn) {
                if (a_in.stats) {
                    const int gs = a.stats_groups > 0 ? a.out_c / a.stats_groups : 0;
                    const int wcols = wtn / 2;    // columns owned by one consumer wave
                    const bool ok = gs > 0 && gs * a.stats_groups == a.out_c && (gs & (gs - 1)) == 0 && gs >= epc && gs <= wcols;
                    a.stats = ok ? a_in.stats : nullptr;
                    if (stats_fused) *stats_fused = ok;
                }
                if (const char* e = settle_gn(false)) return e;
                trace_route("ws", a, 128, wtn);
                if (wtn == 128) return dtype_bf16 ? launch_ws_variant<bf16_t, 2, 2>(a, stream) : launch_ws_variant<float, 2, 2>(a, stream);
                return dtype_bf16 ? launch_ws_variant<bf16_t, 1, 2>(a, stream) : launch_ws_variant<float, 1, 2>(a, stream);
            }
            // weights too large to stay resident (identity-residual 256 -> 256 layers): the plain kernel; a variant that streamed
            // the weights through an LDS-DMA ring beside the producer / consumer waves measured 1 % slower end to end
        }
    }
    if (const char* e = settle_gn(false)) return e;
    trace_route("plain", a, tm, tn);
    return dtype_bf16 ? dispatch<bf16_t>(a, tm, tn, stream) : (x3 ? dispatch<f32x3_t>(a, tm, tn, stream) : dispatch<float>(a, tm, tn, stream));
}

}  // namespace adf
