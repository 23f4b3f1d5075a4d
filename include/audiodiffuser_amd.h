/* C ABI of libadf_hip.so -- the MI355X (gfx950) implementation of AudioDiffuser's EDM sampling hot path.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *   adf_create / adf_load_weight   <- UNet1dBase.__init__ + Lightning strict state_dict load
 *                                     (src/models/backbones/unet1d.py:818-862, src/eval.py:73)
 *   adf_net_forward                <- UNet1dBase.forward            (src/models/backbones/unet1d.py:864-893, :771-816)
 *   adf_denoise                    <- Diffusion.denoise_fn + EluDiffusion.get_scale_weights + clip
 *                                     (src/models/components/diffusion.py:32-63, :232-241; components/utils.py:20-22)
 *   adf_sampler_run                <- EDMSampler.forward / EDMAlphaSampler.forward / DPMSampler.forward (multistep) /
 *                                     DPM2Sampler.forward (src/models/components/sampler_edm.py:371-397, :284-300,
 *                                     :710-768, :470-493), ADPM2Sampler.forward (stochastic_sampler_edm.py:85-100), UniPCSampler.forward (:996-1053)
 *   the call site all of them sit behind: src/models/diffunet_complex_module.py:86-89
 *   adf_adm_create                 <- UNetModel.__init__ (src/models/backbones/unet2d_oai.py:410-601); adf_net_forward on that handle
 *                                     <- UNetModel.forward (:603-634)
 *   adf_wavenet_create             <- WaveNetNoise.__init__ (src/models/backbones/wavenet.py:153-167); the handle it returns goes
 *                                     through the same entry points: adf_load_weight (the module's state_dict keys, incl. the
 *                                     custom WeightNorm's weight_g / weight_v, wavenet.py:15-55), adf_net_forward
 *                                     <- WaveNetNoise.forward (wavenet.py:169-180), adf_denoise / adf_sampler_run as above
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; adf_last_error() gives the message.
 *     No C++ exception crosses this boundary.
 *   - tensors are caller-owned DEVICE buffers (fp32, contiguous, reference layout [B][C][L]); the library
 *     never frees them and only owns its weights / workspace / graphs.
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream); all work
 *     is enqueued there, nothing synchronises the device except adf_load_weight's final packing, which is
 *     also stream-ordered.
 *   - one handle is not thread-safe; distinct handles are independent.
 *   - a handle belongs to the device that was current at adf_create: every entry point makes that device current for
 *     its duration (and restores the caller's), so pointers and the stream must belong to that device.
 *   - memory the library keeps grows only up to a cap: at most 4 (B, L) workspaces per handle and 8 captured sampler
 *     graphs per workspace are retained (least recently used ones are released).
 */
#ifndef AUDIODIFFUSER_AMD_H
#define AUDIODIFFUSER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header grows or an entry point's argument list changes; a binding compares it with
 * adf_abi_version() of the library it loaded before the first call (audiodiffuser_amd/_lib.py does; INTEGRATION.md's C example too).
 * 3: adf_sampler_run(n_injected), adf_sampler_desc.reflow, adf_get_counters.
 * 4: ADF_FLAG_NEAREST_UPSAMPLE (adf_net_config.flags bit 1; state-dict keys ...upsample.2.weight / .bias), WaveNetNoise in bf16 at 64 / 128
 *    residual channels, adf_debug_tap on a WaveNet handle keeps every layer while all of them fit 512 MiB (256 MiB before).
 * 5: ADF_DTYPE_F32X3 (a third value of adf_net_config.dtype: fp32 storage, every GEMM operand split into bf16 hi + lo, three bf16 MFMAs per product). */
#define ADF_ABI_VERSION 5
int adf_abi_version(void);

#define ADF_MAX_LAYERS 12

#define ADF_DTYPE_F32 0  /* parity mode: fp32 storage, exact-fp32 MFMA */
#define ADF_DTYPE_BF16 1 /* throughput mode: bf16 storage, fp32 accumulate */
#define ADF_DTYPE_F32X3 2 /* split-bf16 mode (UNet1dBase only): fp32 storage as ADF_DTYPE_F32; a GEMM operand x is staged as bf16 hi = rn(x) and
                             lo = rn(x - hi) (x = hi + lo to 2^-17) and a product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation:
                             ~1e-5 relative per layer against the exact-fp32 mode, inside the 1e-3 bar of the reference comparison, at the bf16 MFMA rate / 3 */

/* Hyper-parameters of UNet1dBase; same meaning as the reference kwargs. */
typedef struct adf_net_config {
    int32_t channels, num_filters, window_length, stride, in_channels, out_channels;
    int32_t resnet_groups, kernel_multiplier_downsample;
    int32_t num_layers;                     /* len(multipliers) - 1 */
    int32_t multipliers[ADF_MAX_LAYERS + 1];
    int32_t factors[ADF_MAX_LAYERS];
    int32_t num_blocks[ADF_MAX_LAYERS];
    int32_t attentions[ADF_MAX_LAYERS];
    int32_t attention_heads, attention_multiplier;
    int32_t use_skip_scale, use_attention_bottleneck;
    int32_t dtype;                          /* ADF_DTYPE_* */
    int32_t flags;                          /* ADF_FLAG_* */
    int32_t num_classes;                    /* > 0: class_cond=True with that many labels (LabelEmbedder); 0: unconditional */
} adf_net_config;

#define ADF_FLAG_SEPARATE_GN_STATS 1 /* compute GroupNorm statistics in a separate pass instead of the GEMM epilogue */
#define ADF_FLAG_NEAREST_UPSAMPLE 2  /* UNet1dBase(use_nearest_upsample=True): Upsample1d = nearest x f -> ReflectionPad1d(1) -> Conv1d(k = 3)
                                        (unet1d.py:236-246; state-dict keys ...upsample.2.weight / .bias); every factor >= 2 */

#define ADF_SAMPLER_EDM 0       /* EDMSampler: Heun + optional churn      */
#define ADF_SAMPLER_EDM_ALPHA 1 /* EDMAlphaSampler: generalised RK2        */
#define ADF_SAMPLER_DPM_MULTISTEP 2 /* DPMSampler(multisteps=True, x0_pred=True, log_time_spacing=False) */
#define ADF_SAMPLER_DPM2 3      /* DPM2Sampler ("DPM2 Karras", optional churn)   sampler_edm.py:401-493 */
#define ADF_SAMPLER_ADPM2 4     /* ADPM2Sampler ("DPM2 a Karras", ancestral)     stochastic_sampler_edm.py:35-100 */
#define ADF_SAMPLER_LMS 5       /* LMSSampler ("LMS Karras"), order 1..4         sampler_edm.py:1134-1190 */
#define ADF_SAMPLER_DPM2M 7     /* DPM2MSampler ("DPM-Solver++(2M) Karras"); needs num_steps + 1 sigmas   sampler_edm.py:1056-1131;
                                   with `reflow` the class of the same name in stochastic_sampler_edm.py:180-259 */
#define ADF_SAMPLER_UNIPC 8     /* UniPCSampler (variant bh2), order 1..3, x0 or noise prediction, both spacings   sampler_edm.py:807-1053 */
#define ADF_SAMPLER_DPM_SINGLESTEP 6 /* DPMSampler(multisteps=False, x0_pred=True) sampler_edm.py:568-622, :769-805 */
#define ADF_SAMPLER_ADPMPP2S 9  /* ADPMPP2SSampler ("DPM++ 2S a Karras", ancestral; one draw per step with sigma_next > 0)   stochastic_sampler_edm.py:102-178 */

typedef struct adf_sampler_desc {
    int32_t kind;
    int32_t num_steps;   /* the reference constructor's num_steps */
    float s_tmin, s_tmax, s_churn, s_noise;  /* EDM */
    int32_t use_heun;    /* EDM, EDM_ALPHA */
    float alpha;         /* EDM_ALPHA */
    int32_t order;       /* DPM, UniPC: 1..3; LMS: 1..4 */
    float sigma_data;    /* EluDiffusion.sigma_data */
    int32_t use_graph;   /* capture the whole step loop into one hipGraph and replay it */
    float rho, eta;      /* ADPM2 */
    int32_t log_time_spacing;  /* DPM (both kinds), UniPC: the reference's log_time_spacing flag (sampler_edm.py:518, :546-556, :824) */
    int32_t eps_pred;          /* DPM (both kinds), UniPC: 1 = the reference's x0_pred=False (noise prediction, :700-706, :841-846) */
    int32_t reflow;            /* DPM2M: 1 = read the denoiser output as a velocity, denoised = x - output * sigma (stochastic_sampler_edm.py:214-215) */
} adf_sampler_desc;

/* Hyper-parameters of WaveNetNoise (wavenet.py:154-157) and of the ResidualGroup it builds (:120: dim_in 128, dim_mid 512,
 * dim_out 512; ResidualBlock's Linear(512, C) fixes dim_out = 512 in the reference). */
typedef struct adf_wavenet_config {
    int32_t residual_channels, residual_layers, dilation_cycle;
    int32_t dim_in, dim_mid, dim_out;
    int32_t dtype;                          /* ADF_DTYPE_*; BF16 (MFMA kernels) needs residual_channels = 64, 128 or 256 */
} adf_wavenet_config;

/* Hyper-parameters of the ADM-style UNetModel (unet2d_oai.py:410-430).  attention_ds holds the downsample factors at which
 * attention runs -- what the constructor derives from `attention_resolutions` and `image_size` (:433-436). */
#define ADF_ADM_MAX_LEVELS 8
typedef struct adf_adm_config {
    int32_t in_channels, model_channels, out_channels, num_res_blocks;
    int32_t n_mult; int32_t channel_mult[ADF_ADM_MAX_LEVELS];
    int32_t n_attention_ds; int32_t attention_ds[ADF_ADM_MAX_LEVELS];
    int32_t conv_resample, num_heads, num_head_channels, use_scale_shift_norm, resblock_updown, use_new_attention_order;
    int32_t num_classes;                    /* 0 = unconditional; > 0: LabelEmbedder + adf_set_condition (labels, guidance), as for adf_create */
    int32_t dtype;
} adf_adm_config;

typedef struct adf_handle adf_handle;

int adf_create(const adf_net_config* cfg, adf_handle** out);
/* A UNetModel (ADM) handle: x / out of adf_net_forward, adf_denoise, adf_sampler_run are [B][C][H][W] fp32 with L = H * W, the
 * shape given by adf_set_image_shape before the call.  On the device: scale-shift or additive conditioning, conv / pooled resampling, resblock up/down, either
 * attention order, unconditional (BASELINE config 4) or class-conditional.  Debug taps: "input_blocks.<i>", "middle_block",
 * "output_blocks.<i>" (the outputs of the reference's blocks). */
int adf_adm_create(const adf_adm_config* cfg, adf_handle** out);
int adf_set_image_shape(adf_handle* h, int H, int W);

/* Clipping of every denoiser evaluation from here on (adf_denoise, adf_sampler_run): 0 = clamp to [-1, 1] (the default), q in (0, 1] = the dynamic
 * thresholding of EluDiffusion(dynamic_threshold = q) -- per sample, scale = max(1, quantile(|x|, q)) (torch.quantile's linear interpolation),
 * x = clamp(x, -scale, scale) / scale.  Replaces src/models/components/utils.py:19-33 (`clip`) as called from diffusion.py:61. */
int adf_set_dynamic_threshold(adf_handle* h, float quantile);
/* A WaveNetNoise handle.  x / out of adf_net_forward, adf_denoise, adf_sampler_run are [B][1][T] (the reference's forward takes
 * audio [B][T] and returns [B][1][T]: same memory); any T >= 1.  Debug taps: "y<n>" = input of residual layer n including its
 * diffusion-step addend (kept while all of them fit 512 MiB), "skip" = the normalised skip sum. */
int adf_wavenet_create(const adf_wavenet_config* cfg, adf_handle** out);
void adf_destroy(adf_handle* h);
const char* adf_last_error(const adf_handle* h);   /* h may be NULL: error of the last failed adf_create */

/* state_dict interface: names/shapes are exactly the reference UNet1dBase.state_dict() keys */
int adf_num_weights(const adf_handle* h);
const char* adf_weight_name(const adf_handle* h, int index);
int64_t adf_weight_numel(const adf_handle* h, int index);
int adf_load_weight(adf_handle* h, const char* name, const float* dev_fp32, int64_t numel, void* stream);
int adf_weights_missing(const adf_handle* h);      /* number of tensors not loaded yet */

/* Class conditioning / classifier-free guidance for the calls that follow with the same B
 * (UNet1dBase.forward(classes=, cond_drop_prob=) unet1d.py:864-893, LabelEmbedder conditioner.py:59-111, the CFG branch
 * of Diffusion.denoise_fn diffusion.py:49-54).  classes_dev: int64 [B] labels on the device, or NULL to clear the
 * condition.  null_labels != 0: every sample uses the null embedding (cond_drop_prob = 1).  cond_scale applies to
 * adf_denoise / adf_sampler_run only: != 1 runs the network twice per evaluation (labels, null) and combines
 * null + (cond - null) * cond_scale before the EDM preconditioning.  A class-conditional network needs a condition. */
int adf_set_condition(adf_handle* h, const int64_t* classes_dev, int B, int null_labels, float cond_scale, void* stream);

/* out = net(x, t):  x [B][in_channels][L], t [B], out [B][out_channels][L] */
int adf_net_forward(adf_handle* h, const float* x, const float* t, float* out, int B, int L, void* stream);

/* out = clamp(c_skip*x + c_out*net(c_in*x, c_noise), -1, 1); sigmas_dev [B] or NULL (then `sigma` for all) */
int adf_denoise(adf_handle* h, const float* x_noisy, const float* sigmas_dev, float sigma, float sigma_data,
                float* out, int B, int L, void* stream);

/* Full sampling loop.  sigmas_host: the schedule tensor (host, n_sigmas entries) the reference passes as
 * `sigmas`; noise: unit-variance [B][C][L]; injected_noise: [n_injected][B][C][L] draws replacing
 * randn_like, one per step in the reference's order (required when the EDM / DPM2 sampler churns and always for
 * ADPM2, which adds noise every step; may be NULL otherwise).  n_injected = number of [B][C][L] draws behind the
 * pointer: fewer than the sampler consumes (num_steps for EDM, num_steps - 1 for DPM2 / ADPM2) is an error. */
int adf_sampler_run(adf_handle* h, const adf_sampler_desc* desc, const float* sigmas_host, int n_sigmas,
                    const float* noise, const float* injected_noise, int n_injected, float* out, int B, int L, void* stream);
int adf_sampler_nfe(const adf_sampler_desc* desc, const float* sigmas_host, int n_sigmas);

/* Parity-test support: copy an internal activation of the LAST adf_net_forward/adf_denoise call, converted to
 * fp32 [B][C][L].  Names follow oracle/unet1d.py taps ("to_in", "down0.conv", "down0.block1", "mid.attn", ...). */
int adf_debug_tap_shape(adf_handle* h, const char* name, int* C, int* L);
int adf_debug_tap_copy(adf_handle* h, const char* name, float* out_fp32, void* stream);
/* The dynamic threshold alone, in place on x_dev [B][per_sample] (tests: exactness of the order statistics against torch.quantile). */
int adf_debug_dyn_threshold(adf_handle* h, float* x_dev, int B, long long per_sample, float quantile, void* stream);
int adf_debug_tap_count(adf_handle* h);
const char* adf_debug_tap_name(adf_handle* h, int index);

/* What the handle has enqueued so far.  The plugin classes fall back to a host-side loop of tensor ops around a foreign `fn`
 * / `net` (interface compatibility); these counters are how a caller -- and every GPU sampler test -- tells that the device loop
 * of adf_sampler_run ran instead.  net_passes counts network passes enqueued by host code: a captured loop counts once at
 * capture, its replays are graph_replays (sampler_evals adds the evaluations of every run, replayed or not). */
typedef struct adf_run_counters {
    int64_t sampler_runs;     /* successful adf_sampler_run calls */
    int64_t sampler_evals;    /* denoiser evaluations those runs performed on the device */
    int64_t graph_captures;   /* step loops captured into a hipGraph */
    int64_t graph_replays;    /* adf_sampler_run calls served by hipGraphLaunch */
    int64_t denoise_calls;    /* adf_denoise calls */
    int64_t net_passes;       /* network passes enqueued by host code (eager, warm-up, capture) */
} adf_run_counters;
int adf_get_counters(const adf_handle* h, adf_run_counters* out);

/* Bytes of device memory held by the handle (weights + workspaces). */
int64_t adf_device_bytes(const adf_handle* h);

/* Instrumentation for bench.py: the two GEMM launches of resblock `level` of the last forward are replayed `iters` times
 * with HIP events on `stream`, exactly as the network pass issues them (GroupNorm table from the input statistics,
 * output statistics in the epilogue), each iteration on another copy of the operands (>= 3 copies, >= 320 MiB in
 * rotation) so that the 256 MiB Infinity Cache cannot serve them.  *_block1 = conv1, *_block2 = conv2 (+ residual). */
int adf_bench_resblock(adf_handle* h, int B, int L, int level, int iters, float* ms_block1, float* ms_block2,
                       double* algo_bytes_block1, double* algo_bytes_block2, double* flops_block1,
                       double* flops_block2, void* stream);
/* One of the two launches only (conv = 1 | 2); *copies = number of operand copies in rotation = number of untimed warm-up
 * launches that precede the `iters` timed ones (lets a profiler pass pick out the timed dispatches). */
int adf_bench_layer(adf_handle* h, int B, int L, int level, int conv, int iters, float* ms, double* algo_bytes, double* flops,
                    int* copies, void* stream);

/* WaveNetNoise handles: residual layer `layer` of the last pass replayed `iters` times (after 2 untimed launches) with HIP events
 * on `stream`.  The working set of one launch at bench sizes (y, y_next, fp32 skip sum: > 5 GB at B = 128, T = 22050) is far beyond
 * the 256 MiB Infinity Cache, so no operand rotation is needed. */
int adf_bench_wavenet_layer(adf_handle* h, int B, int T, int layer, int iters, float* ms, double* algo_bytes, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif
