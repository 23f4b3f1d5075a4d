// C-ABI implementation (include/audiodiffuser_amd.h): weight registry keyed by the reference state_dict
// names, per-(B, L) workspace, the U-Net walk that launches the fused kernels, and the sampler drivers
// (EDM Heun/churn, EDM-alpha, DPM-Solver multistep) with whole-loop hipGraph capture.
//
// Host logic mirrors (does not copy) the reference control flow:
//   UNet1d.forward            src/models/backbones/unet1d.py:771-816
//   Down/UpsampleBlock1d      src/models/backbones/unet1d.py:441-468, :542-566
//   Diffusion.denoise_fn      src/models/components/diffusion.py:32-63
//   EDMSampler / Alpha / DPM  src/models/components/sampler_edm.py:333-397, :251-300, :624-768
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

namespace adf_api {

std::string g_create_error;

int get_plan(adf_handle* h, int B, int L, hipStream_t s, Plan** out) {
    const adf_net_config& c = h->cfg;
    int total = c.stride;
    for (int i = 0; i < c.num_layers; ++i) total *= c.factors[i];
    if (B < 1 || L < 1 || L % total) return fail(h, "length must be a positive multiple of the total down-sampling factor");
    if (adf_weights_missing(h)) return fail(h, "weights are not fully loaded");
    if (h->wn && !h->wn->packed && wn_pack_weights(h, s)) return 1;
    if (h->adm) {
        const AdmW& a = *h->adm;
        int f = 1;
        for (int i = 1; i < a.cfg.n_mult; ++i) f *= 2;
        if (a.H < 1 || a.W < 1 || (long long)a.H * a.W != L) return fail(h, "UNetModel: call adf_set_image_shape(H, W) with H * W equal to the length argument first");
        if (a.H % f || a.W % f || ((a.H / f) * (a.W / f)) % 64)
            return fail(h, "UNetModel: H and W must be multiples of 2^(levels-1) and the coarsest level a multiple of 64 pixels");
    }
    const std::tuple<int, int, int> pkey{B, L, h->adm ? h->adm->H : 0};
    auto it = h->plans.find(pkey);
    if (it != h->plans.end()) { *out = it->second; h->last_plan = it->second; it->second->last_use = ++h->use_clock; return 0; }
    Plan* p = new Plan();
    p->B = B; p->L = L;
    p->dry = true;
    FwdIO io;
    io.nb = B;
    if (forward(h, p, io, s)) { delete p; return 1; }
    // a long-running caller with varying batch sizes / lengths must not accumulate workspaces: release the least recently
    // used plan first (its graphs and buffers may still be referenced by queued work: drain the device before freeing)
    while (h->plans.size() >= kMaxPlans) {
        auto victim = h->plans.begin();
        for (auto i2 = h->plans.begin(); i2 != h->plans.end(); ++i2)
            if (i2->second->last_use < victim->second->last_use) victim = i2;
        (void)hipDeviceSynchronize();
        destroy_plan(h, victim->second);
        h->plans.erase(victim);
    }
    p->arena_bytes = p->arena_off; p->stats_bytes = p->stats_off;
    p->arena = (char*)dalloc(h, p->arena_bytes, p);
    p->stats = (char*)dalloc(h, p->stats_bytes ? p->stats_bytes : 256, p);
    const size_t wave = (size_t)B * c.out_channels * L;
    p->temb = (float*)dalloc(h, (size_t)B * 4 * c.channels * 4, p);
    p->film = (float*)dalloc(h, (size_t)B * h->film_total * 4, p);
    p->coef = (float*)dalloc(h, (size_t)B * 4 * 4, p);
    bool ok = p->arena && p->stats && p->temb && p->film && p->coef;
    for (int i = 0; i < 10; ++i) { p->sb[i] = (float*)dalloc(h, wave * 4, p); ok = ok && p->sb[i]; }
    p->noise_stage = (float*)dalloc(h, wave * 4, p);
    p->out_stage = (float*)dalloc(h, wave * 4, p);
    ok = ok && p->noise_stage && p->out_stage;
    if (!ok) { destroy_plan(h, p); return fail(h, "device allocation failed for the workspace"); }
    p->dry = false;
    // eager warm-up (loads code objects, sets kernel attributes) so a later graph capture is clean
    io.x = p->noise_stage; io.out = p->out_stage; io.t = p->coef; io.t_stride = 1; io.nb = B; io.mode = 0;
    if (forward(h, p, io, s)) { destroy_plan(h, p); return 1; }
    if (hipStreamSynchronize(s) != hipSuccess) {
        destroy_plan(h, p);
        return fail(h, std::string("warm-up forward failed: ") + hipGetErrorString(hipGetLastError()));
    }
    h->plans[pkey] = p;
    h->last_plan = p;
    p->last_use = ++h->use_clock;
    *out = p;
    return 0;
}

}  // namespace adf_api

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int adf_create(const adf_net_config* cfg, adf_handle** out) {
    if (!cfg || !out) { g_create_error = "adf_create: null argument"; return 1; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "adf_create: no HIP device available"; return 1; }
    const adf_net_config& c = *cfg;
    if (c.num_layers < 1 || c.num_layers > ADF_MAX_LAYERS) { g_create_error = "adf_create: bad num_layers"; return 1; }
    if (c.num_filters != c.channels * c.multipliers[0]) { g_create_error = "adf_create: num_filters must equal channels*multipliers[0]"; return 1; }
    if (c.channels % 2 || c.channels < 2) { g_create_error = "adf_create: channels must be even"; return 1; }
    if (c.dtype != ADF_DTYPE_F32 && c.dtype != ADF_DTYPE_BF16 && c.dtype != ADF_DTYPE_F32X3) { g_create_error = "adf_create: bad dtype"; return 1; }
    adf_handle* h = new adf_handle();
    h->cfg = c;
    if (hipGetDevice(&h->device) != hipSuccess) { g_create_error = "adf_create: hipGetDevice failed"; delete h; return 1; }
    h->bf16 = c.dtype == ADF_DTYPE_BF16;
    h->x3 = c.dtype == ADF_DTYPE_F32X3;             // fp32 storage (esz 4, 32 K elements per row) with split-bf16 GEMM operands
    h->esz = h->bf16 ? 2 : 4;
    h->kc = kRowBytes / h->esz;
    if (build_weights(h)) { g_create_error = h->err; adf_destroy(h); return 1; }
    *out = h;
    return 0;
}

int adf_wavenet_create(const adf_wavenet_config* cfg, adf_handle** out) {
    if (!cfg || !out) { g_create_error = "adf_wavenet_create: null argument"; return 1; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "adf_wavenet_create: no HIP device available"; return 1; }
    const adf_wavenet_config& c = *cfg;
    if (c.residual_channels < 32 || c.residual_channels % 32 || c.residual_channels > 512) { g_create_error = "adf_wavenet_create: residual_channels must be a multiple of 32 in [32, 512]"; return 1; }
    if (c.residual_layers < 1 || c.residual_layers > 1024 || c.dilation_cycle < 1 || c.dilation_cycle > 24) { g_create_error = "adf_wavenet_create: bad residual_layers / dilation_cycle"; return 1; }
    if (c.dim_in < 4 || c.dim_in % 2 || c.dim_in > 1024 || c.dim_mid < 1 || c.dim_mid > 1024 || c.dim_out < 4 || c.dim_out % 4 || c.dim_out > 1024) { g_create_error = "adf_wavenet_create: bad embedding widths"; return 1; }
    if (c.dtype != ADF_DTYPE_F32 && c.dtype != ADF_DTYPE_BF16) { g_create_error = "adf_wavenet_create: bad dtype"; return 1; }
    if (c.dtype == ADF_DTYPE_BF16 && c.residual_channels != 256 && c.residual_channels != 128 && c.residual_channels != 64) { g_create_error = "adf_wavenet_create: the bf16 (MFMA) kernels are built for residual_channels = 64, 128 or 256; use ADF_DTYPE_F32 for other widths"; return 1; }
    adf_handle* h = new adf_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    // the fields of the U-Net config the shared plan / sampler code reads: one waveform channel in and out, no length
    // constraint, embedding width 4 * channels = dim_out
    h->cfg.in_channels = 1; h->cfg.out_channels = 1; h->cfg.stride = 1; h->cfg.num_layers = 0; h->cfg.channels = c.dim_out / 4;
    h->cfg.dtype = c.dtype; h->cfg.resnet_groups = 1;
    if (hipGetDevice(&h->device) != hipSuccess) { g_create_error = "adf_wavenet_create: hipGetDevice failed"; delete h; return 1; }
    h->bf16 = c.dtype == ADF_DTYPE_BF16;
    h->esz = h->bf16 ? 2 : 4;
    h->kc = kRowBytes / h->esz;
    h->wn = new WnW();
    h->wn->cfg = c;
    if (wn_build_weights(h)) { g_create_error = h->err; adf_destroy(h); return 1; }
    *out = h;
    return 0;
}

int adf_adm_create(const adf_adm_config* cfg, adf_handle** out) {
    if (!cfg || !out) { g_create_error = "adf_adm_create: null argument"; return 1; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "adf_adm_create: no HIP device available"; return 1; }
    const adf_adm_config& c = *cfg;
    const int kc = c.dtype == ADF_DTYPE_BF16 ? 64 : 32;
    if (c.dtype != ADF_DTYPE_F32 && c.dtype != ADF_DTYPE_BF16) { g_create_error = "adf_adm_create: bad dtype"; return 1; }
    if (c.n_mult < 1 || c.n_mult > ADF_ADM_MAX_LEVELS || c.num_res_blocks < 1 || c.n_attention_ds < 0 || c.n_attention_ds > ADF_ADM_MAX_LEVELS) { g_create_error = "adf_adm_create: bad level / block counts"; return 1; }
    if (c.model_channels < 32 || c.model_channels % 32 || c.model_channels % kc || c.model_channels > 256) { g_create_error = "adf_adm_create: model_channels must be a multiple of 32 (fp32) / 64 (bf16), at most 256"; return 1; }
    if (c.in_channels < 1 || c.out_channels < 1 || c.out_channels > 4) { g_create_error = "adf_adm_create: in_channels >= 1, 1 <= out_channels <= 4"; return 1; }
    if (c.num_classes < 0) { g_create_error = "adf_adm_create: num_classes must be >= 0"; return 1; }
    for (int i = 0; i < c.n_mult; ++i) if (c.channel_mult[i] < 1) { g_create_error = "adf_adm_create: bad channel_mult"; return 1; }
    adf_handle* h = new adf_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    // the fields of the U-Net config the shared plan / sampler code reads (embedding width 4 * channels = 4 * model_channels)
    h->cfg.in_channels = c.in_channels; h->cfg.out_channels = c.out_channels; h->cfg.stride = 1; h->cfg.num_layers = 0;
    h->cfg.channels = c.model_channels; h->cfg.dtype = c.dtype; h->cfg.resnet_groups = 32;
    if (hipGetDevice(&h->device) != hipSuccess) { g_create_error = "adf_adm_create: hipGetDevice failed"; delete h; return 1; }
    h->bf16 = c.dtype == ADF_DTYPE_BF16;
    h->esz = h->bf16 ? 2 : 4;
    h->kc = kRowBytes / h->esz;
    h->adm = new AdmW();
    h->adm->cfg = c;
    if (adm_build_weights(h)) { g_create_error = h->err; adf_destroy(h); return 1; }
    *out = h;
    return 0;
}

int adf_set_image_shape(adf_handle* h, int H, int W) {
    if (!h || !h->adm) return h ? fail(h, "adf_set_image_shape: not a UNetModel handle") : 1;
    if (H < 1 || W < 1) return fail(h, "adf_set_image_shape: bad shape");
    h->adm->H = H; h->adm->W = W;
    return 0;
}

void adf_destroy(adf_handle* h) {
    if (!h) return;
    DeviceScope scope(h);
    for (auto& kv : h->plans) destroy_plan(h, kv.second);
    h->plans.clear();
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    delete h->wn;
    delete h->adm;
    delete h;
}

const char* adf_last_error(const adf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int adf_num_weights(const adf_handle* h) { return (int)h->names.size(); }
const char* adf_weight_name(const adf_handle* h, int i) { return (i >= 0 && i < (int)h->names.size()) ? h->names[i].c_str() : nullptr; }
int64_t adf_weight_numel(const adf_handle* h, int i) {
    if (i < 0 || i >= (int)h->names.size()) return -1;
    return h->slots.at(h->names[i]).numel;
}

int adf_load_weight(adf_handle* h, const char* name, const float* dev, int64_t numel, void* stream) {
    ADF_ON_DEVICE(h);
    auto it = h->slots.find(name ? name : "");
    if (it == h->slots.end()) return fail(h, std::string("unexpected state_dict key: ") + (name ? name : "(null)"));
    Slot& sl = it->second;
    if (numel != sl.numel) return fail(h, std::string("size mismatch for ") + name + ": expected " + std::to_string(sl.numel) + " got " + std::to_string(numel));
    hipStream_t s = (hipStream_t)stream;
    if (sl.kind == 0) {
        if (hipMemcpyAsync(sl.dst, dev, (size_t)numel * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "hipMemcpyAsync failed");
        for (int p = 0; p < (sl.rep ? sl.rep_n : 0); ++p)
            if (hipMemcpyAsync(sl.rep + (size_t)p * numel, dev, (size_t)numel * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "hipMemcpyAsync failed");
    } else if (sl.kind == 5) {               // qkv bias with the legacy head-major rows -> q | k | v rows
        if (const char* e = launch_permute_qkv_rows(dev, (float*)sl.dst, sl.f, sl.cout / (3 * sl.f), 1, s)) return fail(h, e);
    } else if (sl.kind == 4) {               // qkv weight: permute the rows into a scratch copy, then pack that
        if (const char* e = launch_permute_qkv_rows(dev, (float*)sl.frag, sl.f, sl.cout / (3 * sl.f), sl.cin, s)) return fail(h, e);
        if (const char* e = launch_pack_weight((const float*)sl.frag, sl.dst, h->gemm_dtype(), 0, sl.cout, sl.cin, sl.K, 0, sl.n_offset, sl.n_pad, sl.nchunk, s))
            return fail(h, e);
    } else {
        const char* e = launch_pack_weight(dev, sl.dst, h->gemm_dtype(), sl.kind == 2 ? 1 : (sl.kind == 3 ? 2 : 0), sl.cout, sl.cin, sl.K, sl.f, sl.n_offset,
                                           sl.n_pad, sl.nchunk, s);
        if (e) return fail(h, e);
        if (sl.kind == 2 && sl.dst3) {
            e = launch_pack_weight(dev, sl.dst3, h->gemm_dtype(), 3, sl.cout, sl.cin, sl.K, sl.f, 0, sl.n_pad3, sl.nchunk, s);
            if (e) return fail(h, e);
        }
        if (sl.frag) {
            // rows of this tensor: cout (conv / linear) or f * cout phase-major rows (transposed conv)
            e = launch_repack_frag(sl.dst, sl.frag, sl.n_offset, sl.kind == 2 ? sl.f * sl.cout : sl.cout, sl.n_pad, sl.nchunk * sl.taps, s);
            if (e) return fail(h, e);
        }
    }
    sl.loaded = true;
    if (h->wn) h->wn->packed = false;      // the effective weights (v * g / ||v||) are rebuilt before the next pass
    return 0;
}

int adf_weights_missing(const adf_handle* h) {
    int m = 0;
    for (const auto& kv : h->slots) m += kv.second.loaded ? 0 : 1;
    return m;
}

int adf_net_forward(adf_handle* h, const float* x, const float* t, float* out, int B, int L, void* stream) {
    ADF_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Plan* p;
    if (get_plan(h, B, L, s, &p)) return 1;
    FwdIO io;
    io.x = x; io.out = out; io.t = t; io.t_stride = 1; io.nb = B;
    if (cond_rows(h, B, false, io)) return 1;
    return forward(h, p, io, s);
}

int adf_set_condition(adf_handle* h, const int64_t* classes_dev, int B, int null_labels, float cond_scale, void* stream) {
    ADF_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    if (!classes_dev) { h->cond_on = false; h->cond_scale = 1.0f; return 0; }
    if (h->cdim == 0) return fail(h, "adf_set_condition: the network was built without class conditioning (num_classes = 0)");
    if (B < 1) return fail(h, "adf_set_condition: bad batch size");
    if (adf_weights_missing(h)) return fail(h, "weights are not fully loaded");
    if (B > h->cond_cap) {
        // graphs captured earlier hold the old buffer addresses; drain them before the buffers go
        (void)hipDeviceSynchronize();
        for (auto& kv : h->plans) drop_graphs(kv.second);
        dfree(h, h->cond_classes, (size_t)h->cond_cap * 8);
        dfree(h, h->cond_emb, (size_t)(h->cond_cap + 1) * h->cdim * 4);
        dfree(h, h->cond_film, (size_t)(h->cond_cap + 1) * h->film_total * 4);
        h->cond_cap = 0;
        h->cond_classes = (long long*)dalloc(h, (size_t)B * 8);
        h->cond_emb = (float*)dalloc(h, (size_t)(B + 1) * h->cdim * 4);
        h->cond_film = (float*)dalloc(h, (size_t)(B + 1) * h->film_total * 4);
        if (!h->cond_classes || !h->cond_emb || !h->cond_film) return fail(h, "device allocation failed for the class condition");
        h->cond_cap = B;
    }
    if (hipMemcpyAsync(h->cond_classes, classes_dev, (size_t)B * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "class label copy failed");
    const adf_net_config& c = h->cfg;
    if (const char* e = launch_class_embed(h->cond_classes, h->adm ? h->adm->cfg.num_classes : c.num_classes, null_labels ? 1 : 0, h->lab_emb, h->lab_null, h->lab_lnw, h->lab_lnb,
                                           h->lab_w1, h->lab_b1, h->lab_w2, h->lab_b2, c.channels, h->cdim, h->cond_emb, B + 1, s))
        return fail(h, e);
    // class part of every FiLM projection: columns [tdim, tdim + cdim) of the concatenated weight, no bias (it is in the time part)
    // (not for the ADM net: its embeddings are ADDED before the SiLU of emb_layers, unet2d_oai.py:621-623, so nothing separates)
    if (!h->adm)
        if (const char* e = launch_film(h->cond_emb, h->cdim, h->film_w, 4 * c.channels + h->cdim, 4 * c.channels, nullptr, h->cond_film, B + 1,
                                        h->film_total, s))
            return fail(h, e);
    h->cond_on = true; h->cond_B = B; h->cond_scale = cond_scale;
    return 0;
}

int adf_set_dynamic_threshold(adf_handle* h, float quantile) {
    if (!h) return 1;
    if (!(quantile >= 0.0f) || quantile > 1.0f) return fail(h, "adf_set_dynamic_threshold: the quantile must be in [0, 1] (0 = clamp to [-1, 1])");
    h->dyn_q = quantile;
    return 0;
}

int adf_denoise(adf_handle* h, const float* x_noisy, const float* sigmas_dev, float sigma, float sigma_data, float* out, int B,
                int L, void* stream) {
    ADF_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Plan* p;
    if (get_plan(h, B, L, s, &p)) return 1;
    ++h->ctr.denoise_calls;
    if (!sigmas_dev) return denoise_scalar(h, p, x_noisy, sigma, sigma_data, out, s);
    if (const char* e = launch_edm_coef(sigmas_dev, 0.f, B, sigma_data, p->coef, s)) return fail(h, e);
    FwdIO io;
    io.x = x_noisy; io.t = p->coef + 1; io.t_stride = 4; io.nb = B;
    io.coef = p->coef; io.coef_bstride = 4; io.x_noisy = x_noisy;
    return denoise_io(h, p, io, out, s);
}

int adf_sampler_nfe(const adf_sampler_desc* desc, const float* sigmas_host, int n_sigmas) {
    SamplerCtx c{nullptr, nullptr, desc, sigmas_host, n_sigmas, nullptr, 0};
    c.count_only = true;
    float* r = nullptr;
    if (run_sampler(c, &r)) return -1;
    return c.nfe;
}

int adf_sampler_run(adf_handle* h, const adf_sampler_desc* desc, const float* sigmas_host, int n_sigmas, const float* noise,
                    const float* injected_noise, int n_injected, float* out, int B, int L, void* stream) {
    ADF_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    if (!desc || !sigmas_host || n_sigmas < 1) return fail(h, "adf_sampler_run: bad arguments");
    Plan* p;
    if (get_plan(h, B, L, s, &p)) return 1;
    const long long n = (long long)B * h->cfg.out_channels * L;
    if (h->cfg.in_channels != h->cfg.out_channels) return fail(h, "sampler needs in_channels == out_channels");
    if (h->cdim > 0) {
        if (!h->cond_on || h->cond_B != B) return fail(h, "class-conditional network: call adf_set_condition with the labels of this batch first");
        if (h->cond_scale != 1.0f && ensure_cfg_buffers(h, p)) return 1;
    }
    if (h->dyn_q > 0.0f) {                               // buffers of the dynamic threshold: allocated before any capture starts
        if (ensure_cfg_buffers(h, p)) return 1;
        if (!p->dyn_scale) {
            p->dyn_scale = (float*)dalloc(h, (size_t)B * 4, p);
            if (!p->dyn_scale) return fail(h, "device allocation failed for the dynamic-threshold scales");
        }
    }
    if (hipMemcpyAsync(p->noise_stage, noise, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "noise copy failed");
    if (injected_noise) {
        // one draw per step: the EDM sampler steps num_steps times, the DPM2 family num_steps - 1 times
        const int ndraws = desc->kind == ADF_SAMPLER_ADPMPP2S ? adpmpp2s_draws(sigmas_host, n_sigmas, desc->num_steps)
                           : (desc->kind == ADF_SAMPLER_DPM2 || desc->kind == ADF_SAMPLER_ADPM2) ? desc->num_steps - 1 : desc->num_steps;
        const size_t need = (size_t)(ndraws > 0 ? ndraws : 0) * n;
        // the ABI carries the number of [B][C][L] draws behind the pointer: a short buffer is an error, not an over-read
        if (n_injected < ndraws)
            return fail(h, "injected_noise holds " + std::to_string(n_injected) + " draws of [B][C][L], this sampler consumes " + std::to_string(ndraws));
        if (p->inj_cap < need) {
            if (p->inj_stage) {                      // captured graphs read the old staging buffer: drain and drop them with it
                (void)hipDeviceSynchronize();
                drop_graphs(p);
                dfree(h, p->inj_stage, p->inj_cap * 4, p);
                p->inj_stage = nullptr; p->inj_cap = 0;
            }
            p->inj_stage = (float*)dalloc(h, need * 4, p);
            if (!p->inj_stage) return fail(h, "device allocation failed for injected noise");
            p->inj_cap = need;
        }
        if (hipMemcpyAsync(p->inj_stage, injected_noise, need * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "injected-noise copy failed");
    } else if ((desc->kind == ADF_SAMPLER_EDM || desc->kind == ADF_SAMPLER_DPM2) && desc->s_churn > 0.f) {
        return fail(h, "a sampler with s_churn > 0 needs injected_noise (pre-drawn randn_like tensors)");
    } else if (desc->kind == ADF_SAMPLER_ADPM2) {
        return fail(h, "ADPM2Sampler needs injected_noise (one pre-drawn randn_like tensor per step)");
    } else if (desc->kind == ADF_SAMPLER_ADPMPP2S && adpmpp2s_draws(sigmas_host, n_sigmas, desc->num_steps) > 0) {
        return fail(h, "ADPMPP2SSampler needs injected_noise (one pre-drawn randn_like tensor per step with sigma_next > 0)");
    }
    // the sigma of every denoiser evaluation of this run (host logic only), then the buffers of the per-run table -- sized
    // before any capture starts
    std::vector<float> eval_sigmas;
    {
        SamplerCtx cc{nullptr, nullptr, desc, sigmas_host, n_sigmas, nullptr, 0};
        cc.count_only = true;
        cc.collect = &eval_sigmas;
        float* r0 = nullptr;
        if (run_sampler(cc, &r0)) eval_sigmas.clear();     // a schedule the sampler rejects: the real pass below reports why
    }
    const int n_eval = (int)eval_sigmas.size();
    if (n_eval > p->pre_cap) {
        if (p->pre_cap) {
            (void)hipDeviceSynchronize();
            drop_graphs(p);
            dfree(h, p->coef_all, (size_t)p->pre_cap * 4 * 4, p);
            dfree(h, p->temb_all, (size_t)p->pre_cap * 4 * h->cfg.channels * 4, p);
            dfree(h, p->film_all, (size_t)p->pre_cap * h->film_total * 4, p);
            p->pre_cap = 0;
        }
        p->coef_all = (float*)dalloc(h, (size_t)n_eval * 4 * 4, p);
        p->temb_all = (float*)dalloc(h, (size_t)n_eval * 4 * h->cfg.channels * 4, p);
        p->film_all = (float*)dalloc(h, (size_t)n_eval * h->film_total * 4, p);
        if (!p->coef_all || !p->temb_all || !p->film_all) return fail(h, "device allocation failed for the per-run sigma table");
        p->pre_cap = n_eval;
    }
    // head of the loop (inside the captured graph when there is one): coefficients, sigma embeddings and FiLM projections of all
    // evaluations in three launches
    auto sigma_table = [&](hipStream_t st) -> int {
        if (n_eval == 0) return 0;
        const adf_net_config& cfg = h->cfg;
        if (const char* e = launch_edm_coef_list(eval_sigmas.data(), n_eval, desc->sigma_data, p->coef_all, st)) return fail(h, e);
        if (h->wn) {
            const adf_wavenet_config& wc = h->wn->cfg;
            if (const char* e = launch_wn_step_embed(p->coef_all + 1, 4, n_eval, h->wn->fc1w, h->wn->fc1b, h->wn->fc2w, h->wn->fc2b, wc.dim_in,
                                                     wc.dim_mid, wc.dim_out, p->temb_all, st))
                return fail(h, e);
        } else if (h->adm) {
            const AdmW& am = *h->adm;
            if (const char* e = launch_adm_time_embed(p->coef_all + 1, 4, n_eval, am.cfg.model_channels, am.t_w1, am.t_b1, am.t_w2, am.t_b2,
                                                      4 * am.cfg.model_channels, p->temb_all, st))
                return fail(h, e);
        } else {
            TimeEmbedArgs te;
            te.t = p->coef_all + 1; te.t_stride = 4; te.nb = n_eval; te.ch = cfg.channels;
            te.fourier = h->fourier; te.w1 = h->t_w1; te.b1 = h->t_b1; te.w2 = h->t_w2; te.b2 = h->t_b2; te.temb = p->temb_all;
            if (const char* e = launch_time_embed(te, st)) return fail(h, e);
        }
        if (h->adm && h->cdim > 0) return 0;       // per-sample FiLM rows (time + class embedding): projected inside each pass
        if (const char* e = launch_film(p->temb_all, 4 * cfg.channels, h->film_w, 4 * cfg.channels + h->cdim, 0, h->film_b, p->film_all, n_eval,
                                        h->film_total, st))
            return fail(h, e);
        return 0;
    };
    SamplerCtx c{h, p, desc, sigmas_host, n_sigmas, s, n};
    c.precomputed = n_eval > 0;
    float* result = nullptr;
    if (!desc->use_graph) {
        if (sigma_table(s)) return 1;
        if (run_sampler(c, &result)) return 1;
        if (hipMemcpyAsync(out, result, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "result copy failed");
        ++h->ctr.sampler_runs; h->ctr.sampler_evals += n_eval;
        return 0;
    }
    if (!h->gstream) {
        if (hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess)
            return fail(h, "could not create the graph stream / events");
    }
    hipStream_t gs = h->gstream;
    c.s = gs;
    // inputs staged on the caller's stream must be visible to the graph stream
    if (hipEventRecord(h->ev_in, s) != hipSuccess || hipStreamWaitEvent(gs, h->ev_in, 0) != hipSuccess)
        return fail(h, "stream fence (in) failed");
    std::string key((const char*)desc, sizeof(*desc));
    key.append((const char*)sigmas_host, (size_t)n_sigmas * 4);
    key.push_back(injected_noise ? 'i' : 'n');
    key.push_back(h->cond_on ? 'c' : 'u');                   // the guidance branch structure is part of the captured graph
    key.append((const char*)&h->cond_scale, sizeof(float));
    key.append((const char*)&h->dyn_q, sizeof(float));        // the clipping of every evaluation is part of the captured graph
    auto it = std::find_if(p->graphs.begin(), p->graphs.end(), [&](const std::pair<std::string, hipGraphExec_t>& g) { return g.first == key; });
    if (it != p->graphs.end() && it != p->graphs.begin()) {      // most recently used first
        std::rotate(p->graphs.begin(), it, it + 1);
        it = p->graphs.begin();
    }
    if (it == p->graphs.end()) {
        const hipError_t be = hipStreamBeginCapture(gs, hipStreamCaptureModeRelaxed);
        if (be != hipSuccess) return fail(h, std::string("hipStreamBeginCapture failed: ") + hipGetErrorString(be));
        int rc = sigma_table(gs);
        if (!rc) rc = run_sampler(c, &result);
        if (!rc && hipMemcpyAsync(p->out_stage, result, (size_t)n * 4, hipMemcpyDeviceToDevice, gs) != hipSuccess) { rc = 1; h->err = "result copy failed (capture)"; }
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(gs, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
        if (ee != hipSuccess || !graph) return fail(h, std::string("hipStreamEndCapture failed: ") + hipGetErrorString(ee));
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail(h, std::string("hipGraphInstantiate failed: ") + hipGetErrorString(ie));
        if (p->graphs.size() >= kMaxGraphsPerPlan) {               // every distinct (sampler, schedule, guidance) adds one: cap it
            if (hipStreamSynchronize(gs) != hipSuccess) { (void)hipGraphExecDestroy(exec); return fail(h, "graph stream sync failed"); }
            (void)hipGraphExecDestroy(p->graphs.back().second);
            p->graphs.pop_back();
        }
        p->graphs.insert(p->graphs.begin(), std::make_pair(key, exec));
        it = p->graphs.begin();
        ++h->ctr.graph_captures;
    }
    if (hipGraphLaunch(it->second, gs) != hipSuccess) return fail(h, "hipGraphLaunch failed");
    if (hipEventRecord(h->ev_out, gs) != hipSuccess || hipStreamWaitEvent(s, h->ev_out, 0) != hipSuccess)
        return fail(h, "stream fence (out) failed");
    if (hipMemcpyAsync(out, p->out_stage, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "result copy failed");
    ++h->ctr.sampler_runs; ++h->ctr.graph_replays; h->ctr.sampler_evals += n_eval;
    return 0;
}

int adf_abi_version(void) { return ADF_ABI_VERSION; }

int adf_get_counters(const adf_handle* h, adf_run_counters* out) {
    if (!h || !out) return 1;
    *out = h->ctr;
    return 0;
}

int adf_debug_dyn_threshold(adf_handle* h, float* x_dev, int B, long long per_sample, float quantile, void* stream) {
    ADF_ON_DEVICE(h);
    if (!x_dev || B < 1) return fail(h, "adf_debug_dyn_threshold: bad arguments");
    float* sc = (float*)dalloc(h, (size_t)B * 4);
    if (!sc) return fail(h, "device allocation failed");
    const char* e = launch_dyn_threshold(x_dev, B, per_sample, quantile, sc, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    dfree(h, sc, (size_t)B * 4);
    return e ? fail(h, e) : 0;
}

int adf_debug_tap_count(adf_handle* h) { return h->last_plan ? (int)h->last_plan->taps.size() : 0; }
const char* adf_debug_tap_name(adf_handle* h, int i) {
    if (!h->last_plan || i < 0 || i >= (int)h->last_plan->taps.size()) return nullptr;
    return h->last_plan->taps[i].name.c_str();
}
int adf_debug_tap_shape(adf_handle* h, const char* name, int* C, int* L) {
    if (!h->last_plan) return fail(h, "no forward has run yet");
    for (const auto& t : h->last_plan->taps)
        if (t.name == name) { *C = t.C; *L = t.L; return 0; }
    return fail(h, std::string("unknown tap ") + name);
}
int adf_debug_tap_copy(adf_handle* h, const char* name, float* out, void* stream) {
    ADF_ON_DEVICE(h);
    if (!h->last_plan) return fail(h, "no forward has run yet");
    for (const auto& t : h->last_plan->taps)
        if (t.name == name) {
            const char* e = launch_nlc_to_ncl_f32(t.p, out, t.f32 ? 0 : h->bf16, h->last_plan->B, t.L, t.C, (hipStream_t)stream);
            if (!e && t.scale != 1.0f) e = launch_scale(out, out, t.scale, (long long)h->last_plan->B * t.L * t.C, (hipStream_t)stream);
            return e ? fail(h, e) : 0;
        }
    return fail(h, std::string("unknown tap ") + name);
}

int64_t adf_device_bytes(const adf_handle* h) { return h->bytes; }

}  // extern "C"
