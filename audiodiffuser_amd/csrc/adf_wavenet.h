// Launchers of the WaveNetNoise kernels (adf_wavenet.hip): the DiffWave-style network of BASELINE config 5
// (reference: src/models/backbones/wavenet.py:94-180).  Same rules as adf_kernels.h: a launcher returns nullptr or a static
// error string, never allocates or synchronises, and is safe inside a stream capture.
//
// Layout: the residual stream is channels-last [B][T][C] (fp32 or bf16) and holds y_n = h_n + e_n, the layer input INCLUDING
// the layer's diffusion-step addend (wavenet.py:109-110): a dilated-conv operand is then a plain copy of stored rows (zero
// outside [0, T), which is the conv's zero padding of y), and the layer recovers h_n = y_n - e_n for its residual.  The skip
// sum is fp32 [B][T][C] in both modes.
#pragma once
#include "adf_common.h"

namespace adf {

// sum of squares of a tensor (fp64 accumulate) -> out[0]: the whole-tensor norm of WeightNorm (wavenet.py:29, :50)
const char* launch_wn_sumsq(const float* v, long long numel, double* out, hipStream_t s);

// Effective weight w = v * g / ||v|| (wavenet.py:44-51) of a Conv1d (cout, cin, K), written as a GEMM operand:
//   layout 0 (fp32 kernels): [K][cin][cout] fp32
//   layout 1 (bf16 MFMA kernels): fragment-major [K step of 16 input channels][half][cout][8] bf16, K steps ordered (tap, channel)
//   layout 2: fp32 copy in the tensor's own order (the 1-input-channel input projection)
const char* launch_wn_pack(const float* v, const float* g, const double* sumsq, void* dst, int layout, int cout, int cin, int K,
                           hipStream_t s);

// Diffusion-step embedding (wavenet.py:88-92, :141-142): t[b * t_stride] -> sin | cos features (dim_in) -> Linear -> swish ->
// Linear, stored BEFORE the second swish (launch_film applies it): pre[b][dim_out].
const char* launch_wn_step_embed(const float* t, int t_stride, int nb, const float* w1, const float* b1, const float* w2,
                                 const float* b2, int dim_in, int dim_mid, int dim_out, float* pre, hipStream_t s);

struct WnIO {
    int B, T, C, bf16;
    const float* e; int e_bstride;       // addends of every layer for this pass: e[b * e_bstride + n * C + c]
};

// y0 = relu(w_in * (c_in * x) + b_in) + e_0      (wavenet.py:171-173; c_in from coef[b * coef_bstride] or 1)
const char* launch_wn_input(const WnIO& io, const float* x, const float* coef, int coef_bstride, const float* w_in, const float* b_in,
                            void* y0, hipStream_t s);

// One residual layer n (wavenet.py:108-116), ONE launch: dilated conv (k = 3) -> sigmoid * tanh -> 1x1 conv -> residual half
// into y_next = ((y - e_n) + res) / sqrt(2) + e_{n+1}, skip half accumulated into skip (first layer: stored).
// y_next may be null for the last layer (its residual output is never used).
struct WnLayerArgs {
    const void* y; void* y_next; float* skip;
    const void* w1; const float* b1;     // dilated conv, packed (layout 0 / 1), bias [2C]
    const void* w2; const float* b2;     // output projection, packed, bias [2C]
    int n, dilation, first;
};
const char* launch_wn_layer(const WnIO& io, const WnLayerArgs& a, hipStream_t s);

// out = w_out . relu(W_sp (skip * sqrt(1 / layers)) + b_sp) + b_out   (wavenet.py:152, :177-179), then the EDM epilogue:
//   mode 0: out = F;  mode 1: out = clamp(c_skip * x_noisy + c_out * F, -1, 1) with coef rows (c_in, c_noise, c_skip, c_out)
struct WnFinalArgs {
    const float* skip; float skip_scale;
    const void* w_sp; const float* b_sp; const float* w_out; const float* b_out;
    float* out; int mode; const float* x_noisy; const float* coef; int coef_bstride;
};
const char* launch_wn_final(const WnIO& io, const WnFinalArgs& a, hipStream_t s);

}  // namespace adf
