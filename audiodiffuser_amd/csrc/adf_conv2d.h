// Launchers of the 2-D kernels (adf_conv2d.hip) behind the ADM-style U-Net of BASELINE config 4
// (reference: src/models/backbones/unet2d_oai.py).  Same rules as adf_kernels.h.
//
// Layout: activations are channels-last [B][H*W][C] ("NHWC", fp32 or bf16); the boundary tensors keep the reference's
// [B][C][H][W] fp32.  A 3x3 / 1x1 convolution is an implicit GEMM over pixels (M) x output channels (N) x (taps x Cin) (K) on
// MFMA (32x32x16 bf16 / exact-fp32 32x32x2 f32), with the weights in the packed layout of launch_pack_weight
// ([64- or 32-channel chunk][tap][Cout padded to 32][128-byte row]: a (Cout, Cin, 3, 3) tensor is the (Cout, Cin, 9) conv1d case).
#pragma once
#include "adf_common.h"

namespace adf {

// out[b][p][n] = bias[n] + sum_{tap, ci} f(x[b][src(p, tap)][ci]) * w[n][ci][tap]  (+ res[b][p][n])
//   f = per-(sample, channel) affine a*x + b from `ab` ([B][Cin][2]; GroupNorm (+ scale-shift) folded), then SiLU if act;
//       zero padding is applied AFTER f (the reference pads the activated tensor); ab null = raw input
//   src: mode 0 same size (pad 1 for 3x3); mode 1 nearest x2 upsampling of the input fused (Upsample :122-127: the conv reads pixel
//        ((y + dy - 1) >> 1, (x + dx - 1) >> 1) of an (H/2) x (W/2) input); mode 2 stride 2 (Downsample :146-158: input is 2H x 2W)
//   two sources: x holds channels [0, c0), x1 (optional) channels [c0, cin) of the same pixels -- the skip concat of the output blocks (:629) is
//   never materialised (c0 a multiple of the 128-byte K chunk); the prologue table covers the concatenated channels
struct Conv2dArgs {
    const void* x; const void* x1; int c0; const float* ab; int act;
    int B, H, W;            // OUTPUT height / width
    int cin, cout, n_pad;
    int taps;               // 9 (3x3) or 1
    int mode;
    const void* w; int nchunk;
    const float* bias;
    const float* bias_b; int bias_bstride;   // optional per-SAMPLE addend to the bias: bias_b[b * bias_bstride + co] (ResBlock's additive conditioning, unet2d_oai.py:268-270)
    const void* res;        // optional, same layout / type as out
    void* out;
    double* stats;          // optional [B][stats_groups][2]: (sum, sumsq) of the STORED output (after bias and residual) per GroupNorm group,
    int stats_groups;       // accumulated with fp64 atomics into a pre-zeroed buffer -- the statistics the next GroupNorm reads
};
const char* launch_conv2d(const Conv2dArgs& a, int bf16, hipStream_t s);

// First conv (:467-469) straight from the fp32 [B][Cin][H][W] input with the EDM c_in scaling fused (Cin small: vector kernel).  stats (optional,
// pre-zeroed [B][cout / fg][2]): the FINE GroupNorm statistics of the stored output, reduced in the store when a workgroup's pixels lie in one
// sample, by launch_gn_stats_any behind it otherwise.
const char* launch_conv2d_in(const float* x, const float* w, const float* bias, void* out, int bf16, int B, int cin, int H, int W, int cout,
                             const float* coef, int coef_bstride, double* stats, int fg, hipStream_t s);
// Last conv (:596-600): SiLU(GroupNorm(h)) -> 3x3 conv to a few channels, written as fp32 [B][Cout][H][W], + the EDM epilogue
// (mode 1: clamp(c_skip * x_noisy + c_out * F, -1, 1)).
const char* launch_conv2d_out(const void* h, const float* ab, const float* w, const float* bias, float* out, int bf16, int B, int cin, int H,
                              int W, int cout, int mode, const float* x_noisy, const float* coef, int coef_bstride, hipStream_t s);

// 2 x 2 average pool of a channels-last [B][H][W][C] tensor (H, W even) -> [B][H/2][W/2][C] (Downsample(use_conv=False) :153-156, ResBlock(down=True)'s h_upd / x_upd
// :249-254); with ab (the GroupNorm table [B][C][2]) every input element goes through a v + b (and SiLU with act) first: avg_pool(in_rest(x)) in one pass
const char* launch_avgpool2(const void* x, const float* ab, int act, void* out, int bf16, int B, int H, int W, int C, hipStream_t s);
// nearest x 2 upsampling of a channels-last tensor: [B][H][W][C] -> [B][2H][2W][C] (Upsample(use_conv=False) :122-125, ResBlock(up=True)'s x_upd)
const char* launch_nearest_up2(const void* x, void* out, int bf16, int B, int H, int W, int C, hipStream_t s);
// out[b][p][0:c0] = s0, out[b][p][c0:c0+c1] = s1 (the skip concat of the output blocks, :629)
const char* launch_concat2(const void* s0, const void* s1, int c0, int c1, long long rows, void* out, int bf16, hipStream_t s);
// GroupNorm statistics of a channels-last tensor for any C that is a multiple of a 16-byte chunk: stats[b][g][2] += (sum, sumsq)
const char* launch_gn_stats_any(const void* x, int bf16, int B, int L, int C, int G, double* stats, hipStream_t s);
// GroupNorm32 (+ scale-shift) of the (virtual) concat [x0 (c0) ; x1 (c1)] folded to the per-(sample, channel) table ab[b][c0 + c1][2], from
// FINE statistics of each source: stats[b][c / fg][2] = (sum, sumsq) over fg consecutive channels (what the conv epilogues emit; fg = the greatest
// common divisor of every group size of the net: 4 for the config-4 net).  A group of the concat may straddle the two sources (768 / 32 = 24 channels
// per group against c0 = 512): any range that is a multiple of fg channels can be summed.
struct GnFineArgs {
    const double* stats0; const double* stats1; int c0, c1, L, G, B, fg; float eps;      // fg = channels per fine statistics group
    const float* gamma; const float* beta; const float* film; int film_bstride; float* ab;
};
const char* launch_gn_finalize_fine(const GnFineArgs& a, hipStream_t s);
// timestep_embedding (:31-49): cos | sin features of t[b * t_stride] -> Linear -> SiLU -> Linear (time_embed :455-459): emb[b][dim_out]
const char* launch_adm_time_embed(const float* t, int t_stride, int nb, int mc, const float* w1, const float* b1, const float* w2,
                                  const float* b2, int dim_out, float* emb, hipStream_t s);
// out[b][i] = a[b * a_bstride + i] + c[b * c_bstride + i]   (time embedding + class embedding rows, unet2d_oai.py:623)
const char* launch_add_rows(float* out, const float* a, int a_bstride, const float* c, int c_bstride, int B, int n, hipStream_t s);
// dst row (which * heads + h) * d + c  <-  src row (h * 3 + which) * d + c   (QKVAttentionLegacy :338-340 -> the q | k | v layout of
// launch_attention); cols floats per row
const char* launch_permute_qkv_rows(const float* src, float* dst, int heads, int d, int cols, hipStream_t s);

}  // namespace adf
