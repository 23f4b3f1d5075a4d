// Launchers of the non-GEMM kernels of the EDM sampling path (adf_kernels.hip).
// Every launcher returns nullptr on success or a static error string; none of them allocates,
// synchronises or copies, so they are safe inside hipGraph capture.
#pragma once
#include "adf_common.h"

namespace adf {

// GroupNorm statistics of an NLC tensor: stats[b][g][2] += (sum, sumsq) in fp64 (buffer pre-zeroed).
const char* launch_gn_stats(const void* x, int bf16, int B, int L, int C, int G, double* stats, hipStream_t s);

// Fold GroupNorm (+ optional FiLM scale/shift) into a per-(sample, channel) affine y = a*x + b
// for a (possibly concatenated) input [src0 (c0) ; scale1 * src1 (c1)].
// GnFinalizeArgs: adf_common.h
const char* launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s);

// out[b][l][c0+c1] = act(a*x + b) of the concatenated input [src0 ; src1] (affine from launch_gn_finalize).
const char* launch_gn_apply(const void* s0, const void* s1, int c0, int c1, int L, int B, const float* ab, int act, void* out,
                            int bf16, hipStream_t s);

// gn_finalize + gn_apply fused (short levels): out = act(GroupNorm/FiLM affine of [src0 ; scale1*src1]).
const char* launch_gn_norm_apply(const void* s0, const void* s1, const GnFinalizeArgs& a, int act, void* out, int bf16, hipStream_t s);

// Row LayerNorm over the channel (last) dim of an NLC tensor; beta may be null (gain only).
const char* launch_ln_rows(const void* x, void* y, int bf16, long long rows, int C, const float* gamma,
                           const float* beta, float eps, hipStream_t s);

// Multi-head self-attention on a fused qkv tensor [B][N][3C] (q | k | v), out [B][N][C].
const char* launch_attention(const void* qkv, void* out, int bf16, int B, int N, int C, int heads, hipStream_t s);
// split-bf16 mode (fp32 tensors; ADF_DTYPE_F32X3): MFMA attention with bf16 hi + lo operands where it applies, else launch_attention(.., 0, ..)
const char* launch_attention_x3(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s);

// Input transform Conv1d(in_ch -> nf, k=wl, stride, pad) on the fp32 waveform x[B][in_ch][L], with the
// EDM c_in scaling fused: out NLC [B][L/stride][nf].  coef may be null (c_in = 1).
// nn.Upsample(scale_factor = f, mode = "nearest") + nn.ReflectionPad1d(1) of a channels-last [B][L][C] tensor: out [B][f L + 2][C], row i + 1 = x[i / f],
// row 0 = upsampled row 1, row f L + 1 = upsampled row f L - 2 (unet1d.py:236-239); C a multiple of a 16-byte chunk
const char* launch_upsample_nearest_pad(const void* x, void* out, int bf16, int B, int L, int C, int f, hipStream_t s);
const char* launch_to_in(const float* x, const float* w, void* out, int bf16, int B, int in_ch, int L, int nf,
                         int wl, int stride, int pad, const float* coef, int coef_bstride, hipStream_t s);

// Output transform ConvTranspose1d(nf -> out_ch, k=wl, stride, pad) + EDM epilogue
//   mode 0: out = F;  mode 1: out = clamp(c_skip*x_noisy + c_out*F, -1, 1).   out / x_noisy: fp32 [B][out_ch][L]
const char* launch_to_out(const void* h, const float* w, float* out, int dtype, int B, int out_ch, int Lh, int nf,
                          int wl, int stride, int pad, int mode, const float* x_noisy, const float* coef,
                          int coef_bstride, hipStream_t s);

// coef[b][4] = (c_in, c_noise, c_skip, c_out) from sigma (EluDiffusion.get_scale_weights).
const char* launch_edm_coef(const float* sigmas_dev, float sigma_scalar, int nb, float sigma_data, float* coef,
                            hipStream_t s);
// coef[i] for a host-side list of sigmas (values passed in the kernel arguments: usable inside a stream capture)
const char* launch_edm_coef_list(const float* sigmas_host, int n, float sigma_data, float* coef, hipStream_t s);

// Time embedding MLP: t[b] -> temb[b][4*ch];  t read as t[b*t_stride].
struct TimeEmbedArgs {
    const float* t; int t_stride; int nb; int ch;
    const float* fourier; const float* w1; const float* b1; const float* w2; const float* b2;
    float* temb;
};
const char* launch_time_embed(const TimeEmbedArgs& a, hipStream_t s);

// All resblocks' FiLM projections at once: film[b][j] = bias[j] + sum_i W[j][w_col0 + i] * silu(in[b][i]), rows of W are ldw long.
const char* launch_film(const float* in, int in_dim, const float* w, int ldw, int w_col0, const float* bias, float* film, int nb,
                        int total, hipStream_t s);
// LabelEmbedder: class embeddings [nrows][cdim], last row = null embedding (conditioner.py:92-111).
const char* launch_class_embed(const long long* classes, int num_classes, int null_all, const float* emb, const float* null_emb,
                               const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2, const float* b2,
                               int ch, int cdim, float* out, int nrows, hipStream_t s);
// out = clamp(c_skip x + c_out (fn + (fc - fn) scale), -1, 1); coef rows are (c_in, c_noise, c_skip, c_out).
const char* launch_cfg_combine(float* out, const float* x, const float* fc, const float* fn, const float* coef, int coef_bstride,
                               float scale, long long per_sample, long long n, int clampit, hipStream_t s);
// Dynamic thresholding of EluDiffusion (components/utils.py:23-33) in place on x [B][per_sample]: per sample scale = max(1, quantile(|x|, q)) (torch's
// linear interpolation, exact order statistics by radix select), x = clamp(x, -scale, scale) / scale.  scale_scratch: B floats.
const char* launch_dyn_threshold(float* x, int B, long long per_sample, float q, float* scale_scratch, hipStream_t s);

// ---- sampler state updates (fp32, flat arrays of n elements) --------------------------------
const char* launch_scale(float* out, const float* in, float s, long long n, hipStream_t st);
// x_hat = x + c * s_noise * eps
const char* launch_churn(float* x_hat, const float* x, const float* eps, float c, float s_noise, long long n, hipStream_t st);
// d = (x - den) / sigma ; x_next = x + dt * d
const char* launch_euler(float* x_next, float* d, const float* x, const float* den, float sigma, float dt, long long n, hipStream_t st);
// d2 = (x_e - den2) / sigma2 ; x_next = x + h * (w1 * d + w2 * d2)
const char* launch_rk2(float* x_next, const float* x, const float* d, const float* x_e, const float* den2,
                       float sigma2, float h, float w1, float w2, long long n, hipStream_t st);
// DPM-Solver++ multistep update (orders 1..3), x0-prediction form.
struct DpmArgs {
    int order; float ratio, phi1, phi2, phi3, inv_r0, inv_r1, r0_frac, inv_r01;
    const float* m0; const float* m1; const float* m2;
};
const char* launch_dpm_update(float* x_out, const float* x, const DpmArgs& a, int clamp, long long n, hipStream_t st);
// x_next = x_base + ((x_eval - den) / sigma) * dt  (DPM2 / ancestral DPM2 steps)
// m = (x - m) / sigma in place: the noise-prediction model value of DPMSampler(x0_pred=False)
const char* launch_eps(float* m, const float* x, float sigma, long long n, hipStream_t s);
// m = x - m * sigma (the reflow reading of DPM2MSampler, stochastic_sampler_edm.py:214-215)
const char* launch_reflow(float* m, const float* x, float sigma, long long n, hipStream_t s);
// DPM2MSampler update: out = ratio*x - coef*(c1*d - c2*d_old) (d_old may be null: out = ratio*x - coef*d)
const char* launch_dpm2m(float* out, const float* x, const float* d, const float* d_old, float ratio, float coef, float c1, float c2, long long n,
                         hipStream_t s);
// LMSSampler: newest derivative d = (x - den) / sigma -> dcur; x += c[0]*d + c[1]*d1 + c[2]*d2 + c[3]*d3 (first `order` terms)
struct LmsArgs { float* dcur; const float* d1; const float* d2; const float* d3; float c[4]; int order; };
const char* launch_lms(float* x, const float* den, float sigma, const LmsArgs& a, long long n, hipStream_t s);
// out = (a*x - b*e0) + c*(e1 - e0) (e1 may be null), optionally clamped to [-1, 1]: the single-step DPM-Solver updates
const char* launch_lincomb(float* out, const float* x, const float* e0, const float* e1, float a, float b, float c, int clampit,
                           long long n, hipStream_t s);
// UniPCSampler.multistep_uni_pc_update (sampler_edm.py:872-994), one predictor or corrector formula per launch, the reference's order
// of operations:  xt_ = a*x - hp*m0;  res = sum_{k<K} rho[k] * ((m[k] - m0) / rk[k])  (+ rho_t * (mt - m0) if mt);  out = xt_ - sb*res
struct UniPcArgs { float a, hp, sb; int K; float rk[2], rho[2], rho_t; const float* m0; const float* m[2]; const float* mt; };
const char* launch_unipc(float* out, const float* x, const UniPcArgs& a, long long n, hipStream_t s);
const char* launch_dstep(float* x_next, const float* x_base, const float* x_eval, const float* den, float sigma, float dt, long long n,
                         hipStream_t st);
const char* launch_clamp(float* x, long long n, hipStream_t st);

const char* launch_nlc_to_ncl_f32(const void* x, float* y, int bf16, int B, int L, int C, hipStream_t st);

// ---- weight packing (fp32 state-dict tensors -> GEMM operand layout) ---------------------------
// mode 0: Conv1d / Linear weight (Cout, Cin, K) ; mode 1: ConvTranspose1d weight (Cin, Cout, K=2f) as 2-tap phases
// bf16 1x1 weights of the transformer blocks: second copy in MFMA-fragment order (rows [n_offset, n_offset + n_rows) of a packed buffer)
const char* launch_repack_frag(const void* src_packed, void* dst, int n_offset, int n_rows, int n_pad, int nchunk, hipStream_t s);
const char* launch_pack_weight(const float* src, void* dst, int dtype, int mode, int cout, int cin, int K, int f,
                               int n_offset, int n_pad, int nchunk, hipStream_t s);

}  // namespace adf
