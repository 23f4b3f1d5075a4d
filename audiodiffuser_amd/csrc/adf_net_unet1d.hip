// UNet1dBase behind the C ABI: the weight registry in the reference's state_dict order and the network walk that launches the
// fused kernels (reference: src/models/backbones/unet1d.py:771-816 UNet1d.forward, :441-468 / :542-566 Down / UpsampleBlock1d).
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

namespace adf_api {

int build_weights(adf_handle* h) {
    const adf_net_config& c = h->cfg;
    Registrar R{h};
    const int ch = c.channels, tdim = 4 * ch, n = c.num_layers;
    h->cdim = c.num_classes > 0 ? 4 * ch : 0;
    const int temb = tdim + h->cdim;     // every FiLM Linear reads cat(time_embed, class_embed) (unet1d.py:272)
    if (c.num_classes > 0) {             // conditioner.py:64-90, registered before the U-Net
        h->lab_null = R.reg_f32("label_conditioner.null_classes_emb", ch);
        h->lab_emb = R.reg_f32("label_conditioner.label_emb.weight", (int64_t)c.num_classes * ch);
        h->lab_lnw = R.reg_f32("label_conditioner.class_to_cond.0.weight", ch);
        h->lab_lnb = R.reg_f32("label_conditioner.class_to_cond.0.bias", ch);
        h->lab_w1 = R.reg_f32("label_conditioner.class_to_cond.1.weight", (int64_t)h->cdim * ch);
        h->lab_b1 = R.reg_f32("label_conditioner.class_to_cond.1.bias", h->cdim);
        h->lab_w2 = R.reg_f32("label_conditioner.class_to_cond.3.weight", (int64_t)h->cdim * h->cdim);
        h->lab_b2 = R.reg_f32("label_conditioner.class_to_cond.3.bias", h->cdim);
    }
    h->to_in_w = R.reg_f32("unet.to_in.to_in.weight", (int64_t)c.num_filters * c.in_channels * c.window_length);
    h->to_out_w = R.reg_f32("unet.to_out.to_out.weight", (int64_t)c.num_filters * c.out_channels * c.window_length);
    h->fourier = R.reg_f32("unet.to_time.0.0.weights", ch / 2);
    h->t_w1 = R.reg_f32("unet.to_time.0.1.weight", (int64_t)tdim * (ch + 1));
    h->t_b1 = R.reg_f32("unet.to_time.0.1.bias", tdim);
    h->t_w2 = R.reg_f32("unet.to_time.2.weight", (int64_t)tdim * tdim);
    h->t_b2 = R.reg_f32("unet.to_time.2.bias", tdim);
    h->downs.resize(n);
    for (int i = 0; i < n; ++i) {
        DownW& d = h->downs[i];
        d.cin = ch * c.multipliers[i]; d.cout = ch * c.multipliers[i + 1]; d.factor = c.factors[i];
        const std::string pre = "unet.downsamples." + std::to_string(i);
        R.conv_folded(pre + ".downsample", d.down, d.cout, d.cin, d.factor * c.kernel_multiplier_downsample + 1, d.factor);
        d.blocks.resize(c.num_blocks[i]);
        for (int j = 0; j < c.num_blocks[i]; ++j) R.resblock(pre + ".blocks." + std::to_string(j), d.blocks[j], d.cout, d.cout, temb);
        d.attn = c.attentions[i] != 0;
        if (d.attn) R.transformer(pre + ".transformer", d.tr, d.cout, c.attention_multiplier);
    }
    const int cb = ch * c.multipliers[n];
    R.resblock("unet.bottleneck.pre_block", h->mid_pre, cb, cb, temb);
    if (c.use_attention_bottleneck) R.transformer("unet.bottleneck.transformer", h->mid_tr, cb, c.attention_multiplier);
    R.resblock("unet.bottleneck.post_block", h->mid_post, cb, cb, temb);
    h->ups.resize(n);
    for (int u = 0; u < n; ++u) {
        const int i = n - 1 - u;
        UpW& up = h->ups[u];
        up.cin = ch * c.multipliers[i + 1]; up.cout = ch * c.multipliers[i]; up.factor = c.factors[i];
        const std::string pre = "unet.upsamples." + std::to_string(u);
        const int nb = c.num_blocks[i] + (c.attentions[i] ? 1 : 0);
        up.blocks.resize(nb);
        for (int j = 0; j < nb; ++j) R.resblock(pre + ".blocks." + std::to_string(j), up.blocks[j], 2 * up.cin, up.cin, temb);
        up.attn = c.attentions[i] != 0;
        if (up.attn) R.transformer(pre + ".transformer", up.tr, up.cin, c.attention_multiplier);
        const int f = up.factor;
        if (c.flags & ADF_FLAG_NEAREST_UPSAMPLE) {          // nn.Sequential(Upsample(nearest), ReflectionPad1d(1), Conv1d(k = 3)): unet1d.py:236-246
            up.nearest = true;
            R.conv(pre + ".upsample.2", up.up, up.cout, up.cin, 3, true);
            continue;
        }
        R.reg_pack(pre + ".upsample.weight", up.up, up.cout, up.cin, 2 * f, 0, f * up.cout, true, f);
        if (h->bf16 && up.up.w && (up.cin == 128 || up.cin == 256) && (up.cout == 64 || up.cout == 128 || up.cout == 256) && (f == 2 || f == 4)) {
            up.up.wfrag = dalloc(h, (size_t)up.up.nchunk * up.up.taps * up.up.n_pad * kRowBytes);   // fragment-major copy: adf_gemm_up.h
            if (!up.up.wfrag) R.ok = false;
            else h->slots[pre + ".upsample.weight"].frag = up.up.wfrag;
        }
        up.up.bias = R.reg_f32(pre + ".upsample.bias", up.cout);
        // (f = 2 only: the packing is written for any even factor, but no configuration with f = 4 and f * cout <= 256 is in the tests)
        if ((h->bf16 || h->x3) && up.up.w && f == 2 && (f * up.cout == 128 || f * up.cout == 256) && up.cin % 128 == 0 && up.cout % 64 == 0) {
            ConvW& w3 = up.up3;
            w3.cin = up.cin; w3.K = 2 * f; w3.f = f; w3.taps = 3;
            w3.n = f * up.cout; w3.n_pad = w3.n; w3.cout = w3.n;
            w3.nchunk = ceil_div(up.cin, h->kc);
            w3.w = dalloc(h, ((size_t)w3.nchunk * 3 * w3.n_pad + (size_t)kTapGroup * (w3.n_pad + 128)) * kRowBytes);
            w3.bias = (float*)dalloc(h, (size_t)w3.n * 4);
            if (!w3.w || !w3.bias) R.ok = false;
            else {
                Slot& sw = h->slots[pre + ".upsample.weight"];
                sw.dst3 = w3.w; sw.n_pad3 = w3.n_pad;
                Slot& sb = h->slots[pre + ".upsample.bias"];
                sb.rep = w3.bias; sb.rep_n = f;
            }
        }
    }
    // one concatenated FiLM projection for all resblocks
    h->film_w = (float*)dalloc(h, (size_t)h->film_total * temb * 4);
    h->film_b = (float*)dalloc(h, (size_t)h->film_total * 4);
    if (!h->film_w || !h->film_b) R.ok = false;
    for (const auto& fn : R.film_names) {
        R.reg_f32(fn.pre + ".to_cond_embedding.1.weight", (int64_t)fn.rows * temb, h->film_w + (size_t)fn.off * temb);
        R.reg_f32(fn.pre + ".to_cond_embedding.1.bias", fn.rows, h->film_b + fn.off);
    }
    return R.ok ? 0 : fail(h, "device allocation failed while building the weight registry");
}

int forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s) {
    if (!p->dry) ++h->ctr.net_passes;
    if (h->wn) return wn_forward(h, p, io, s);
    if (h->adm) return adm_forward(h, p, io, s);
    const adf_net_config& c = h->cfg;
    Walker W{h, p, s};
    W.film2 = io.film2; W.film2_bstride = io.film2_bstride;
    W.film = io.film_pre ? io.film_pre : p->film;
    p->arena_off = 0; p->stats_off = 0;
    p->taps.clear(); p->rbs.clear();
    const int B = p->B, L = p->L, n = c.num_layers;
    const int pad = c.window_length / 2 - c.stride / 2;
    const int tdim = 4 * c.channels;
    if (!p->dry && p->stats_bytes) {
        if (hipMemsetAsync(p->stats, 0, p->stats_bytes, s) != hipSuccess) return fail(h, "hipMemsetAsync(stats) failed");
    }
    // sigma embedding + every resblock's FiLM projection (unless the sampler computed them for the whole run already)
    if (W.live() && !io.film_pre) {
        TimeEmbedArgs te;
        te.t = io.t; te.t_stride = io.t_stride; te.nb = io.nb; te.ch = c.channels;
        te.fourier = h->fourier; te.w1 = h->t_w1; te.b1 = h->t_b1; te.w2 = h->t_w2; te.b2 = h->t_b2; te.temb = p->temb;
        W.check(launch_time_embed(te, s));
        W.check(launch_film(p->temb, tdim, h->film_w, tdim + h->cdim, 0, h->film_b, p->film, io.nb, h->film_total, s));
    }
    Act x = W.new_act(c.num_filters, L / c.stride);
    if (W.live())
        W.check(launch_to_in(io.x, h->to_in_w, x.p, h->bf16, B, c.in_channels, L, c.num_filters, c.window_length, c.stride, pad,
                             io.coef, io.coef_bstride, s));
    W.tap("to_in", x);
    std::vector<std::vector<Act>> skips_list;
    for (int i = 0; i < n; ++i) {
        const DownW& d = h->downs[i];
        const int f = d.factor, km = c.kernel_multiplier_downsample;
        Act y = W.new_act(d.cout, x.L / f);
        // Downsample1d (unet1d.py:214-225) as a stride-1 conv over the row-folded view [L/f][f*C] (Registrar::conv_folded)
        if (x.L % f) return fail(h, "downsample: length not divisible by the factor");
        Act xv = x;
        xv.C = x.C * f; xv.L = x.L / f; xv.stats = nullptr;
        GemmArgs g = W.gemm_base(y, xv.L, y.L, d.down);
        g.seg[0] = Walker::seg_of(xv, nullptr, nullptr, 1.f, 0, km + 1, 1, -(km / 2), 1, d.down);
        W.run_gemm(g, y, true);
        W.tap("down" + std::to_string(i) + ".conv", y);
        x = y;
        std::vector<Act> skips;
        for (size_t j = 0; j < d.blocks.size(); ++j) {
            x = W.resblock("down" + std::to_string(i) + ".block" + std::to_string(j), x, nullptr, d.blocks[j], io.nb);
            skips.push_back(x);
        }
        if (d.attn) {
            x = W.transformer("down" + std::to_string(i) + ".attn", x, d.tr);
            skips.push_back(x);
        }
        skips_list.push_back(skips);
    }
    x = W.resblock("mid.pre", x, nullptr, h->mid_pre, io.nb);
    if (c.use_attention_bottleneck) x = W.transformer("mid.attn", x, h->mid_tr);
    x = W.resblock("mid.post", x, nullptr, h->mid_post, io.nb);
    for (int u = 0; u < n; ++u) {
        const UpW& up = h->ups[u];
        std::vector<Act>& skips = skips_list.back();
        for (size_t j = 0; j < up.blocks.size(); ++j) {
            if (skips.empty()) { W.check("upsample: skip stack underflow"); break; }
            Act sk = skips.back();
            skips.pop_back();
            x = W.resblock("up" + std::to_string(u) + ".block" + std::to_string(j), x, &sk, up.blocks[j], io.nb);
        }
        skips_list.pop_back();
        if (up.attn) x = W.transformer("up" + std::to_string(u) + ".attn", x, up.tr);
        const int f = up.factor;
        Act y = W.new_act(up.cout, x.L * f);
        bool done = false;
        if (up.nearest) {
            // the upsampled, reflection-padded rows are written once ([f L + 2][cin]: row i + 1 = x[i / f], rows 0 and f L + 1 the reflected ones), the
            // 3-tap conv then runs over them without padding (lin = mrows + 2, first tap at the output row)
            Act u0 = W.new_act(up.cin, x.L * f + 2);
            if (W.live()) W.check(launch_upsample_nearest_pad(x.p, u0.p, h->bf16, B, x.L, up.cin, f, s));
            GemmArgs g = W.gemm_base(y, x.L * f + 2, x.L * f, up.up);
            g.seg[0] = Walker::seg_of(u0, nullptr, nullptr, 1.f, 0, 3, 1, 0, 1, up.up);
            W.run_gemm(g, y, u + 1 < n);
            done = true;
        }
        if (up.up3.w) {
            // ConvTranspose1d(kernel 2 f, stride f, padding f / 2), f even, as a 3-tap conv over the INPUT rows with f * cout columns: column p * cout + co of
            // row j is output row f j + p -- the same bytes -- and uses x[j] (tap p + f / 2), x[j - 1] (tap p + 3 f / 2, p < f / 2) or x[j + 1] (tap p - f / 2,
            // p >= f / 2); the other third of the packed weights is zero.  No scatter, no L + 1-th row: the resblock conv kernel's raw form takes it.
            Act y3 = y;
            y3.C = f * up.cout; y3.L = x.L;
            GemmArgs g3 = W.gemm_base(y3, x.L, x.L, up.up3);
            g3.seg[0] = Walker::seg_of(x, nullptr, nullptr, 1.f, 0, 3, 1, -1, 1, up.up3);
            g3.phase_c = up.cout;
            // (eligibility is asked with the statistics request attached -- the launcher's shape checks include the group size the epilogue can
            //  reduce -- and, if that is what it declines, again without: the statistics then come from the separate pass.  ADVICE r3: asked without
            //  and launched with, a group size outside 8 .. 64 channels was a hard error instead of the fallback every other route has.)
            const bool ask = u + 1 < n && W.can_fuse_stats(up.cout);
            g3.stats = ask ? (double*)(uintptr_t)256 : nullptr;                 // (a dry check: never dereferenced)
            g3.stats_groups = ask ? h->cfg.resnet_groups : 0;
            bool elig = conv_gemm_phase_eligible(g3, h->gemm_dtype());
            if (!elig && ask) { g3.stats = nullptr; g3.stats_groups = 0; elig = conv_gemm_phase_eligible(g3, h->gemm_dtype()); }
            if (elig) {
                if (ask) { y.stats = W.alloc_stats(); if (g3.stats) g3.stats = y.stats; }
                if (W.live()) {
                    bool fused = false;
                    W.check(launch_conv_gemm(g3, h->gemm_dtype(), s, &fused));
                    if (ask && !fused) W.check(launch_gn_stats(y.p, h->bf16, B, y.L, y.C, h->cfg.resnet_groups, y.stats, s));
                }
                done = true;
            }
        }
        if (!done) {
            GemmArgs g = W.gemm_base(y, x.L, x.L + 1, up.up);
            g.seg[0] = Walker::seg_of(x, nullptr, nullptr, 1.f, 0, 2, 1, 0, -1, up.up);
            g.bias_mod = up.cout;
            g.scatter_f = f; g.scatter_pad = f / 2 + f % 2;
            W.run_gemm(g, y, u + 1 < n);
        }
        W.tap("up" + std::to_string(u) + ".conv", y);
        x = y;
    }
    if (W.live())
        W.check(launch_to_out(x.p, h->to_out_w, io.out, h->gemm_dtype(), B, c.out_channels, x.L, c.num_filters, c.window_length, c.stride, pad,
                              io.mode, io.x_noisy, io.coef, io.coef_bstride, s));
    return W.bad ? 1 : 0;
}

}  // namespace adf_api
