// Tile-shape dispatch for the fused implicit-GEMM kernel (see adf_gemm.h).
#include "adf_gemm.h"

namespace adf {

namespace {

template <typename T, int MT, int NT, int WM, int WN>
const char* launch_variant(const GemmArgs& a, hipStream_t stream) {
    constexpr int TM = 32 * MT * WM, TN = 32 * NT * WN, NTHR = 64 * WM * WN;
    constexpr int lds = kARows * kRowBytes + kTapGroup * TN * kRowBytes;
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<T, MT, NT, WM, WN>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        attr_set = true;
    }
    const int tiles_n = (a.n_pad + TN - 1) / TN;
    const int tiles_m = (a.mrows + TM - 1) / TM;
    const long long blocks = (long long)tiles_n * tiles_m * a.B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return "conv_gemm: bad grid";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), lds, stream, a);
    return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm: launch failed";
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

const char* launch_conv_gemm(const GemmArgs& a, int dtype_bf16, hipStream_t stream) {
    if (a.nseg < 1 || a.nseg > 2) return "conv_gemm: nseg must be 1 or 2";
    if (a.n_pad % 32) return "conv_gemm: n_pad must be a multiple of 32";
    const int epc = dtype_bf16 ? 8 : 4;
    for (int s = 0; s < a.nseg; ++s) {
        const GemmSeg& g = a.seg[s];
        if (g.c0 % epc || g.c1 % epc) return "conv_gemm: channel counts must be multiples of a 16-byte chunk";
        if (g.step != 1 && g.step != -1) return "conv_gemm: step must be +-1";
        if (g.taps < 1 || g.stride < 1) return "conv_gemm: bad taps/stride";
    }
    if (a.stats) {
        if (a.scatter_f) return "conv_gemm: fused stats unsupported with phase scatter";
        const int gs = a.out_c / a.stats_groups;
        if (gs * a.stats_groups != a.out_c || (gs & (gs - 1))) return "conv_gemm: fused stats need power-of-two group size";
    }
    // tile selection: largest M tile whose staged rows (incl. halo) fit, not much larger than M
    int tm = 0;
    for (int cand = 128; cand >= 32; cand >>= 1) {
        bool fits = true;
        for (int s = 0; s < a.nseg; ++s)
            if ((cand - 1) * a.seg[s].stride + a.seg[s].taps > kARows) fits = false;
        if (!fits) continue;
        if (cand > 32 && a.mrows <= cand / 2) continue;
        tm = cand;
        break;
    }
    if (!tm) return "conv_gemm: no tile shape fits (stride/taps too large)";
    int tn = a.n_pad >= 128 ? 128 : (a.n_pad >= 64 ? 64 : 32);
    if (a.n_pad % tn && a.n_pad % 64 == 0) tn = 64;
    // keep >= ~2 waves of blocks on the 256 CUs when the problem is small
    auto nblocks = [&](int tm_, int tn_) { return (long long)ceil_div(a.mrows, tm_) * ceil_div(a.n_pad, tn_) * a.B; };
    while (nblocks(tm, tn) < 512 && tn > 32) tn >>= 1;
    while (nblocks(tm, tn) < 512 && tm > 32) tm >>= 1;
    return dtype_bf16 ? dispatch<bf16_t>(a, tm, tn, stream) : dispatch<float>(a, tm, tn, stream);
}

}  // namespace adf
