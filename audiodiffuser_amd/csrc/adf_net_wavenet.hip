// WaveNetNoise behind the C ABI (reference: src/models/backbones/wavenet.py:153-180): registry, weight-norm packing, the layer walk.
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

namespace adf_api {

// ---- WaveNetNoise ----------------------------------------------------------------------------------------------------
// Registration order = the reference module's state_dict order (wavenet.py:158-167; the custom WeightNorm re-registers
// g and v after the bias, :37-42).
int wn_build_weights(adf_handle* h) {
    WnW& w = *h->wn;
    const adf_wavenet_config& c = w.cfg;
    Registrar R{h};
    const int C = c.residual_channels;
    auto conv = [&](const std::string& pre, WnConv& cv, int cout, int cin, int K, int layout) {
        cv.cout = cout; cv.cin = cin; cv.K = K;
        cv.bias = R.reg_f32(pre + ".conv.module.bias", cout);
        cv.g = R.reg_f32(pre + ".conv.module.weight_g", 1);
        cv.v = R.reg_f32(pre + ".conv.module.weight_v", (int64_t)cout * cin * K);
        cv.packed = dalloc(h, (size_t)cout * cin * K * (layout == 1 ? 2 : 4));
        if (!cv.packed) R.ok = false;
    };
    const int lay = h->bf16 ? 1 : 0;
    conv("input_projection", w.in, C, 1, 1, 2);
    w.fc1w = R.reg_f32("residual_layer.fc_t1.weight", (int64_t)c.dim_mid * c.dim_in);
    w.fc1b = R.reg_f32("residual_layer.fc_t1.bias", c.dim_mid);
    w.fc2w = R.reg_f32("residual_layer.fc_t2.weight", (int64_t)c.dim_out * c.dim_mid);
    w.fc2b = R.reg_f32("residual_layer.fc_t2.bias", c.dim_out);
    // the per-layer diffusion projections, concatenated: one launch_film call computes every layer's addend
    h->film_total = c.residual_layers * C;
    h->film_w = (float*)dalloc(h, (size_t)h->film_total * c.dim_out * 4);
    h->film_b = (float*)dalloc(h, (size_t)h->film_total * 4);
    if (!h->film_w || !h->film_b) R.ok = false;
    w.dil.resize(c.residual_layers);
    w.outp.resize(c.residual_layers);
    for (int n = 0; n < c.residual_layers && R.ok; ++n) {
        const std::string pre = "residual_layer.residual_blocks." + std::to_string(n);
        conv(pre + ".dilated_conv", w.dil[n], 2 * C, C, 3, lay);
        R.reg_f32(pre + ".diffusion_projection.weight", (int64_t)C * c.dim_out, h->film_w + (size_t)n * C * c.dim_out);
        R.reg_f32(pre + ".diffusion_projection.bias", C, h->film_b + (size_t)n * C);
        conv(pre + ".output_projection", w.outp[n], 2 * C, C, 1, lay);
    }
    conv("skip_projection", w.sp, C, C, 1, lay);
    w.out_w = R.reg_f32("output_projection.conv.weight", C);
    w.out_b = R.reg_f32("output_projection.conv.bias", 1);
    w.sumsq = (double*)dalloc(h, 256);
    if (!w.sumsq) R.ok = false;
    return R.ok ? 0 : fail(h, "device allocation failed while building the weight registry");
}

// effective weights of every weight-normed conv, as GEMM operands (stream-ordered; the one sumsq scratch is reused in order)
int wn_pack_weights(adf_handle* h, hipStream_t s) {
    WnW& w = *h->wn;
    const int lay = h->bf16 ? 1 : 0;
    auto one = [&](const WnConv& cv, int layout) -> int {
        if (const char* e = launch_wn_sumsq(cv.v, (long long)cv.cout * cv.cin * cv.K, w.sumsq, s)) return fail(h, e);
        if (const char* e = launch_wn_pack(cv.v, cv.g, w.sumsq, cv.packed, layout, cv.cout, cv.cin, cv.K, s)) return fail(h, e);
        return 0;
    };
    if (one(w.in, 2) || one(w.sp, lay)) return 1;
    for (size_t n = 0; n < w.dil.size(); ++n)
        if (one(w.dil[n], lay) || one(w.outp[n], lay)) return 1;
    w.packed = true;
    return 0;
}

// WaveNetNoise.forward (wavenet.py:169-180) for x [B][1][T]; io as for the U-Net (EDM scalars fused into the first and last kernel)
int wn_forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s) {
    WnW& w = *h->wn;
    const adf_wavenet_config& c = w.cfg;
    Walker W{h, p, s};
    p->arena_off = 0; p->stats_off = 0;
    p->taps.clear(); p->rbs.clear(); p->wn_layers.clear();
    const int B = p->B, T = p->L, C = c.residual_channels, NL = c.residual_layers;
    const size_t act = (size_t)B * T * C * h->esz;
    // every layer input stays resident when that is small (the parity taps y<n>); otherwise two buffers alternate
    const bool keep = act * (size_t)NL <= ((size_t)512 << 20);   // (one bf16 waveform of 22050 samples x 36 layers = 406 MB: the full-size parity test)
    std::vector<void*> ys(keep ? NL : 2);
    for (auto& q : ys) q = W.alloc(act);
    float* const skip = (float*)W.alloc((size_t)B * T * C * 4);
    if (p->dry) return 0;
    const float* film = io.film_pre ? io.film_pre : p->film;
    if (!io.film_pre) {
        W.check(launch_wn_step_embed(io.t, io.t_stride, io.nb, w.fc1w, w.fc1b, w.fc2w, w.fc2b, c.dim_in, c.dim_mid, c.dim_out, p->temb, s));
        W.check(launch_film(p->temb, c.dim_out, h->film_w, c.dim_out, 0, h->film_b, p->film, io.nb, h->film_total, s));
    }
    WnIO wio;
    wio.B = B; wio.T = T; wio.C = C; wio.bf16 = h->bf16 ? 1 : 0;
    wio.e = film; wio.e_bstride = io.nb > 1 ? h->film_total : 0;
    W.check(launch_wn_input(wio, io.x, io.coef, io.coef_bstride, (const float*)w.in.packed, w.in.bias, ys[0], s));
    for (int n = 0; n < NL && !W.bad; ++n) {
        WnLayerArgs a;
        a.y = ys[keep ? n : (n & 1)];
        a.y_next = n + 1 < NL ? ys[keep ? n + 1 : ((n + 1) & 1)] : nullptr;
        a.skip = skip;
        a.w1 = w.dil[n].packed; a.b1 = w.dil[n].bias;
        a.w2 = w.outp[n].packed; a.b2 = w.outp[n].bias;
        a.n = n; a.first = n == 0 ? 1 : 0;
        a.dilation = 1 << (n % c.dilation_cycle);
        if (keep) p->taps.push_back({"y" + std::to_string(n), (void*)a.y, C, T});
        p->wn_layers.push_back(a);
        W.check(launch_wn_layer(wio, a, s));
    }
    p->wn_io = wio;
    WnFinalArgs f;
    f.skip = skip; f.skip_scale = (float)std::sqrt(1.0 / (double)NL);
    f.w_sp = w.sp.packed; f.b_sp = w.sp.bias; f.w_out = w.out_w; f.b_out = w.out_b;
    f.out = io.out; f.mode = io.mode; f.x_noisy = io.x_noisy; f.coef = io.coef; f.coef_bstride = io.coef_bstride;
    TapRec sk{"skip", (void*)skip, C, T};
    sk.f32 = 1; sk.scale = f.skip_scale;
    p->taps.push_back(sk);
    W.check(launch_wn_final(wio, f, s));
    return W.bad ? 1 : 0;
}

}  // namespace adf_api
