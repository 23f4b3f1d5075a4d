// ADM-style UNetModel behind the C ABI (reference: src/models/backbones/unet2d_oai.py:382-635): registry and the block walk.
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

namespace adf_api {

// ---- ADM-style 2-D U-Net ---------------------------------------------------------------------------------------------
// The module list UNetModel.__init__ builds (unet2d_oai.py:467-594), registered in state_dict order.
int adm_build_weights(adf_handle* h) {
    AdmW& a = *h->adm;
    const adf_adm_config& c = a.cfg;
    const int mc = c.model_channels, ted = 4 * mc;
    auto has_att = [&](int ds) { for (int i = 0; i < c.n_attention_ds; ++i) if (c.attention_ds[i] == ds) return true; return false; };
    auto heads_of = [&](int ch) { return c.num_head_channels == -1 ? c.num_heads : ch / c.num_head_channels; };
    // pass 1: structure
    auto new_res = [&](int cin, int cout) { AdmRes r; r.cin = cin; r.cout = cout; r.has_skip = cin != cout; r.film_off = h->film_total; h->film_total += (c.use_scale_shift_norm ? 2 : 1) * cout;
                                            a.res.push_back(r); return AdmLayer{1, (int)a.res.size() - 1}; };
    auto new_attn = [&](int ch) { AdmAttn t; t.c = ch; t.heads = heads_of(ch); a.attn.push_back(t); return AdmLayer{2, (int)a.attn.size() - 1}; };
    int ch = a.input_ch = c.channel_mult[0] * mc;
    a.input_blocks.push_back({AdmLayer{0, 0}});
    std::vector<int> chans{ch};
    int ds = 1;
    for (int level = 0; level < c.n_mult; ++level) {
        for (int k = 0; k < c.num_res_blocks; ++k) {
            std::vector<AdmLayer> ls{new_res(ch, c.channel_mult[level] * mc)};
            ch = c.channel_mult[level] * mc;
            if (has_att(ds)) ls.push_back(new_attn(ch));
            a.input_blocks.push_back(ls);
            chans.push_back(ch);
        }
        if (level != c.n_mult - 1) {
            if (c.resblock_updown) {                       // ResBlock(ch, ch, down=True) instead of Downsample (:515-528)
                AdmLayer l = new_res(ch, ch);
                a.res[l.idx].updown = 2;
                a.input_blocks.push_back({l});
            } else if (c.conv_resample) {
                a.resample.emplace_back();
                a.resample.back().cin = ch; a.resample.back().cout = ch;
                a.input_blocks.push_back({AdmLayer{3, (int)a.resample.size() - 1}});
            } else a.input_blocks.push_back({AdmLayer{5, 0}});
            chans.push_back(ch);
            ds *= 2;
        }
    }
    a.skip_ch = chans;
    a.middle = {new_res(ch, ch), new_attn(ch), new_res(ch, ch)};
    for (int level = c.n_mult - 1; level >= 0; --level) {
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            const int ich = chans.back(); chans.pop_back();
            std::vector<AdmLayer> ls{new_res(ch + ich, mc * c.channel_mult[level])};
            ch = mc * c.channel_mult[level];
            if (has_att(ds)) ls.push_back(new_attn(ch));
            if (level && i == c.num_res_blocks) {
                if (c.resblock_updown) {                   // ResBlock(ch, ch, up=True) instead of Upsample (:575-588)
                    AdmLayer l = new_res(ch, ch);
                    a.res[l.idx].updown = 1;
                    ls.push_back(l);
                } else if (c.conv_resample) {
                    a.resample.emplace_back();
                    a.resample.back().cin = ch; a.resample.back().cout = ch;
                    ls.push_back(AdmLayer{4, (int)a.resample.size() - 1});
                } else ls.push_back(AdmLayer{6, 0});
                ds /= 2;
            }
            a.output_blocks.push_back(ls);
        }
    }
    a.final_ch = ch;
    if (a.final_ch != a.input_ch) return fail(h, "UNetModel: the last level's width must equal the first's (out conv, unet2d_oai.py:599)");
    {
        auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
        int g = a.final_ch / 32;
        for (const AdmRes& r : a.res) { g = gcd(g, r.cin / 32); g = gcd(g, r.cout / 32); }
        for (const AdmAttn& t : a.attn) g = gcd(g, t.c / 32);
        for (int sc : a.skip_ch) g = gcd(g, sc);       // a concat splits at the skip's width
        a.fg = g < 1 ? 1 : (g > 4 ? 4 : g);
        while (128 % a.fg) --a.fg;
    }
    // pass 2: registry, in the module's registration order
    Registrar R{h};
    h->film_w = (float*)dalloc(h, (size_t)h->film_total * ted * 4);
    h->film_b = (float*)dalloc(h, (size_t)h->film_total * 4);
    if (!h->film_w || !h->film_b) R.ok = false;
    a.t_w1 = R.reg_f32("time_embed.0.weight", (int64_t)ted * mc);
    a.t_b1 = R.reg_f32("time_embed.0.bias", ted);
    a.t_w2 = R.reg_f32("time_embed.2.weight", (int64_t)ted * ted);
    a.t_b2 = R.reg_f32("time_embed.2.bias", ted);
    if (c.num_classes > 0) {             // LabelEmbedder(num_classes, None, model_channels, 4 * model_channels), conditioner.py:64-90; unet2d_oai.py:461-468
        h->cdim = ted;
        h->lab_null = R.reg_f32("label_conditioner.null_classes_emb", mc);
        h->lab_emb = R.reg_f32("label_conditioner.label_emb.weight", (int64_t)c.num_classes * mc);
        h->lab_lnw = R.reg_f32("label_conditioner.class_to_cond.0.weight", mc);
        h->lab_lnb = R.reg_f32("label_conditioner.class_to_cond.0.bias", mc);
        h->lab_w1 = R.reg_f32("label_conditioner.class_to_cond.1.weight", (int64_t)ted * mc);
        h->lab_b1 = R.reg_f32("label_conditioner.class_to_cond.1.bias", ted);
        h->lab_w2 = R.reg_f32("label_conditioner.class_to_cond.3.weight", (int64_t)ted * ted);
        h->lab_b2 = R.reg_f32("label_conditioner.class_to_cond.3.bias", ted);
    }
    auto reg_layer = [&](const AdmLayer& l, const std::string& pre) {
        if (l.kind == 0) {
            a.in_w = R.reg_f32(pre + ".weight", (int64_t)a.input_ch * c.in_channels * 9);
            a.in_b = R.reg_f32(pre + ".bias", a.input_ch);
        } else if (l.kind == 1) {
            AdmRes& r = a.res[l.idx];
            r.g1w = R.reg_f32(pre + ".in_layers.0.weight", r.cin);
            r.g1b = R.reg_f32(pre + ".in_layers.0.bias", r.cin);
            R.conv(pre + ".in_layers.2", r.c1, r.cout, r.cin, 9, true);
            const int ew = (c.use_scale_shift_norm ? 2 : 1) * r.cout;             // Linear(4 mc, 2 cout) or, additive conditioning, Linear(4 mc, cout) (:214-220)
            R.reg_f32(pre + ".emb_layers.1.weight", (int64_t)ew * ted, h->film_w + (size_t)r.film_off * ted);
            R.reg_f32(pre + ".emb_layers.1.bias", ew, h->film_b + r.film_off);
            r.g2w = R.reg_f32(pre + ".out_layers.0.weight", r.cout);
            r.g2b = R.reg_f32(pre + ".out_layers.0.bias", r.cout);
            R.conv(pre + ".out_layers.3", r.c2, r.cout, r.cout, 9, true);
            if (r.has_skip) R.conv(pre + ".skip_connection", r.skip, r.cout, r.cin, 1, true);
        } else if (l.kind == 2) {
            AdmAttn& t = a.attn[l.idx];
            t.gw = R.reg_f32(pre + ".norm.weight", t.c);
            t.gb = R.reg_f32(pre + ".norm.bias", t.c);
            R.conv(pre + ".qkv", t.qkv, 3 * t.c, t.c, 1, true);
            R.conv(pre + ".proj_out", t.proj, t.c, t.c, 1, true);
            if (!c.use_new_attention_order) {
                // QKVAttentionLegacy (:338-340) keeps each head's q | k | v rows together; the attention kernel reads q | k | v blocks:
                // the rows of the weight and of the bias are permuted once at load (slot kinds 4 / 5)
                t.qkv_tmp = (float*)dalloc(h, (size_t)3 * t.c * t.c * 4);
                if (!t.qkv_tmp) R.ok = false;
                Slot& sw = h->slots[pre + ".qkv.weight"]; sw.kind = 4; sw.frag = t.qkv_tmp; sw.f = t.heads;
                Slot& sb = h->slots[pre + ".qkv.bias"]; sb.kind = 5; sb.f = t.heads; sb.cout = 3 * t.c;
            }
        } else if (l.kind == 3 || l.kind == 4) {
            ConvW& w = a.resample[l.idx];
            R.conv(pre + (l.kind == 3 ? ".op" : ".conv"), w, w.cout, w.cin, 9, true);
        }                                                   // (5 / 6: AvgPool2d / nearest interpolation, no parameters)
    };
    for (size_t i = 0; i < a.input_blocks.size(); ++i)
        for (size_t j = 0; j < a.input_blocks[i].size(); ++j) reg_layer(a.input_blocks[i][j], "input_blocks." + std::to_string(i) + "." + std::to_string(j));
    for (size_t j = 0; j < a.middle.size(); ++j) reg_layer(a.middle[j], "middle_block." + std::to_string(j));
    for (size_t i = 0; i < a.output_blocks.size(); ++i)
        for (size_t j = 0; j < a.output_blocks[i].size(); ++j) reg_layer(a.output_blocks[i][j], "output_blocks." + std::to_string(i) + "." + std::to_string(j));
    a.out_gw = R.reg_f32("out.0.weight", a.final_ch);
    a.out_gb = R.reg_f32("out.0.bias", a.final_ch);
    a.out_w = R.reg_f32("out.2.weight", (int64_t)c.out_channels * a.input_ch * 9);
    a.out_b = R.reg_f32("out.2.bias", c.out_channels);
    return R.ok ? 0 : fail(h, "device allocation failed while building the weight registry");
}

// UNetModel.forward (unet2d_oai.py:603-634) on channels-last activations; x / out are the reference's [B][C][H][W] fp32
int adm_forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s) {
    AdmW& a = *h->adm;
    const adf_adm_config& c = a.cfg;
    Walker W{h, p, s};
    p->arena_off = 0; p->stats_off = 0;
    p->taps.clear(); p->rbs.clear();
    const int B = p->B, ted = 4 * c.model_channels;
    if (!p->dry && p->stats_bytes && hipMemsetAsync(p->stats, 0, p->stats_bytes, s) != hipSuccess) return fail(h, "hipMemsetAsync(stats) failed");
    const float* film = io.film_pre ? io.film_pre : p->film;
    int film_bs = io.nb > 1 ? h->film_total : 0;
    if (h->cdim > 0) {
        // class-conditional: emb[b] = time_embed(t) + label_conditioner(classes[b]) (unet2d_oai.py:619-623), so every sample has its own FiLM rows
        float* emb_b = (float*)W.alloc((size_t)B * ted * 4);
        film = p->film; film_bs = h->film_total;
        if (W.live()) {
            const float* te = io.temb_pre;
            int te_bs = 0;
            if (!te) {
                W.check(launch_adm_time_embed(io.t, io.t_stride, io.nb, c.model_channels, a.t_w1, a.t_b1, a.t_w2, a.t_b2, ted, p->temb, s));
                te = p->temb; te_bs = io.nb > 1 ? ted : 0;
            }
            const float* ce = io.null_cond ? h->cond_emb + (size_t)B * ted : h->cond_emb;       // last row = the null embedding
            W.check(launch_add_rows(emb_b, te, te_bs, ce, io.null_cond ? 0 : ted, B, ted, s));
            W.check(launch_film(emb_b, ted, h->film_w, ted, 0, h->film_b, p->film, B, h->film_total, s));
        }
    } else if (W.live() && !io.film_pre) {
        W.check(launch_adm_time_embed(io.t, io.t_stride, io.nb, c.model_channels, a.t_w1, a.t_b1, a.t_w2, a.t_b2, ted, p->temb, s));
        W.check(launch_film(p->temb, ted, h->film_w, ted, 0, h->film_b, p->film, io.nb, h->film_total, s));
    }
    // st: FINE GroupNorm statistics of the tensor ([B][C / fg][2]), when its producer reduced them; t1 / st1: the second source of a virtual
    // concat (the skip of an output block, unet2d_oai.py:629: never materialised -- convs and the GroupNorm table read both sources)
    struct T2 { Act t; int H, W; double* st = nullptr; Act t1; double* st1 = nullptr; };
    const int fg = a.fg;
    auto alloc_fine = [&](int C) -> double* {
        const size_t bytes = ((size_t)B * (C / fg) * 2 * sizeof(double) + 255) & ~(size_t)255;
        const size_t off = p->stats_off;
        p->stats_off += bytes;
        if (p->dry) return (double*)(uintptr_t)(off + 256);
        if (p->stats_off > p->stats_bytes) { W.check("stats arena overflow"); return nullptr; }
        return (double*)(p->stats + off);
    };
    auto ensure_stats = [&](const Act& t, double*& st) {
        if (st) return;
        st = alloc_fine(t.C);
        if (W.live()) W.check(launch_gn_stats_any(t.p, h->bf16, B, t.L, t.C, t.C / fg, st, s));
    };
    // GroupNorm32 (:10-21) (+ scale-shift, :262-267) of a tensor (or a virtual concat) folded to the per-(sample, channel) table a conv prologue reads
    auto gn_table = [&](T2& x, const float* gamma, const float* beta, const float* fl) -> float* {
        ensure_stats(x.t, x.st);
        if (x.t1.p || x.t1.C) ensure_stats(x.t1, x.st1);
        const int ctot = x.t.C + x.t1.C;
        float* ab = (float*)W.alloc((size_t)B * ctot * 2 * 4);
        if (W.live()) {
            GnFineArgs g;
            memset(&g, 0, sizeof(g));
            g.stats0 = x.st; g.stats1 = x.st1; g.c0 = x.t.C; g.c1 = x.t1.C; g.L = x.t.L; g.G = 32; g.B = B; g.fg = fg; g.eps = 1e-5f;
            g.gamma = gamma; g.beta = beta; g.film = fl; g.film_bstride = film_bs; g.ab = ab;
            W.check(launch_gn_finalize_fine(g, s));
        }
        return ab;
    };
    // stats: also reduce the (fine) GroupNorm statistics of the output in the epilogue (where a GroupNorm reads this tensor next)
    auto conv = [&](const T2& x, const ConvW& w, const float* ab, int act, int mode, const void* res, bool stats, const float* bias_b = nullptr) -> T2 {
        T2 y;
        if (stats && w.cout % fg == 0 && (w.cout <= 128 || w.cout % 128 == 0)) y.st = alloc_fine(w.cout);
        y.H = mode == 1 ? x.H * 2 : (mode == 2 ? x.H / 2 : x.H);
        y.W = mode == 1 ? x.W * 2 : (mode == 2 ? x.W / 2 : x.W);
        y.t = W.new_act(w.cout, y.H * y.W);
        if (W.live()) {
            Conv2dArgs g;
            g.x = x.t.p; g.x1 = x.t1.C ? x.t1.p : nullptr; g.c0 = x.t.C;
            g.ab = ab; g.act = act; g.B = B; g.H = y.H; g.W = y.W; g.cin = x.t.C + x.t1.C; g.cout = w.cout; g.n_pad = w.n_pad;
            g.taps = w.taps; g.mode = mode; g.w = w.w; g.nchunk = w.nchunk; g.bias = w.bias; g.res = res; g.out = y.t.p;
            g.bias_b = bias_b; g.bias_bstride = film_bs;
            g.stats = y.st; g.stats_groups = w.cout / fg;
            W.check(launch_conv2d(g, h->bf16, s));
        }
        return y;
    };
    auto run = [&](const std::vector<AdmLayer>& ls, T2 x, const std::string& bname) -> T2 {
        int lj = -1;
        for (const AdmLayer& l : ls) {
            ++lj;
            const std::string ln = bname + "." + std::to_string(lj);
            if (W.bad) break;
            if (l.kind == 0) {
                T2 y; y.H = x.H; y.W = x.W; y.t = W.new_act(a.input_ch, x.H * x.W);
                y.st = alloc_fine(a.input_ch);   // here, so that the copy pushed on the skip stack carries them (the last output block reads them again)
                if (W.live()) W.check(launch_conv2d_in(io.x, a.in_w, a.in_b, y.t.p, h->bf16, B, c.in_channels, x.H, x.W, a.input_ch, io.coef, io.coef_bstride, y.st, fg, s));
                x = y;
                W.tap(ln, x.t);
            } else if (l.kind == 1) {                                  // ResBlock._forward, :248-272
                const AdmRes& r = a.res[l.idx];
                const float* ab1 = gn_table(x, r.g1w, r.g1b, nullptr);
                // scale-shift form (:262-267): the embedding enters the out_norm table; additive form (:268-270, h = out_norm(h + emb_out)): it is a
                // per-sample addend to conv1's bias, so that the stored tensor (and the statistics reduced from it) is h + emb_out
                const bool ss = c.use_scale_shift_norm != 0;
                const float* emb_b = ss ? nullptr : film + r.film_off;
                T2 hh;
                if (r.updown == 2) {
                    // ResBlock(down=True) (:249-254): h = in_conv(avg_pool(in_rest(x))), x = avg_pool(x).  The pooled activation and the pooled
                    // input are written once each (GroupNorm + SiLU fused into the first pool); conv1 then takes its input raw
                    if (x.t1.C) { W.check("ResBlock(down=True) on a skip concat"); break; }
                    T2 p1; p1.H = x.H / 2; p1.W = x.W / 2; p1.t = W.new_act(r.cin, p1.H * p1.W);
                    T2 p2 = p1; p2.t = W.new_act(r.cin, p1.H * p1.W);
                    if (W.live()) {
                        W.check(launch_avgpool2(x.t.p, ab1, 1, p1.t.p, h->bf16, B, x.H, x.W, r.cin, s));
                        W.check(launch_avgpool2(x.t.p, nullptr, 0, p2.t.p, h->bf16, B, x.H, x.W, r.cin, s));
                    }
                    hh = conv(p1, r.c1, nullptr, 0, 0, nullptr, true, emb_b);
                    x = p2;
                } else if (r.updown == 1) {
                    // ResBlock(up=True): h = in_conv(nearest x 2 (in_rest(x))) -- the upsampling is an index map of the conv's gather (mode 1),
                    // x = nearest x 2 (x) is written once (it is the block's residual)
                    if (x.t1.C) { W.check("ResBlock(up=True) on a skip concat"); break; }
                    hh = conv(x, r.c1, ab1, 1, 1, nullptr, true, emb_b);
                    T2 u2; u2.H = x.H * 2; u2.W = x.W * 2; u2.t = W.new_act(r.cin, u2.H * u2.W);
                    if (W.live()) W.check(launch_nearest_up2(x.t.p, u2.t.p, h->bf16, B, x.H, x.W, r.cin, s));
                    x = u2;
                } else hh = conv(x, r.c1, ab1, 1, 0, nullptr, true, emb_b);
                W.tap(ln + ".h1", hh.t);
                const float* ab2 = gn_table(hh, r.g2w, r.g2b, ss ? film + r.film_off : nullptr);
                const void* skip = x.t.p;
                if (r.has_skip) { T2 sk2 = conv(x, r.skip, nullptr, 0, 0, nullptr, false); W.tap(ln + ".skip", sk2.t); skip = sk2.t.p; }
                x = conv(hh, r.c2, ab2, 1, 0, skip, true);
                W.tap(ln, x.t);
            } else if (l.kind == 2) {                                   // AttentionBlock._forward, :316-322
                const AdmAttn& t = a.attn[l.idx];
                const float* ab = gn_table(x, t.gw, t.gb, nullptr);
                T2 xn; xn.H = x.H; xn.W = x.W; xn.t = W.new_act(t.c, x.t.L);
                if (W.live()) W.check(launch_gn_apply(x.t.p, nullptr, t.c, 0, x.t.L, B, ab, 0, xn.t.p, h->bf16, s));
                W.tap(ln + ".xn", xn.t);
                T2 qkv = conv(xn, t.qkv, nullptr, 0, 0, nullptr, false);
                W.tap(ln + ".qkv", qkv.t);         // q | k | v blocks (the rows were permuted at load for the legacy order)
                T2 att; att.H = x.H; att.W = x.W; att.t = W.new_act(t.c, x.t.L);
                if (W.live()) W.check(launch_attention(qkv.t.p, att.t.p, h->bf16, B, x.t.L, t.c, t.heads, s));
                W.tap(ln + ".att", att.t);
                x = conv(att, t.proj, nullptr, 0, 0, xn.t.p, true);    // the residual is the NORMALISED input (:318-322)
                W.tap(ln, x.t);
            } else if (l.kind == 5 || l.kind == 6) {                     // Downsample / Upsample without a conv (conv_resample=False, :122-125, :153-156)
                if (x.t1.C) { W.check("pooled resampling on a skip concat"); break; }
                T2 y; y.H = l.kind == 5 ? x.H / 2 : x.H * 2; y.W = l.kind == 5 ? x.W / 2 : x.W * 2;
                y.t = W.new_act(x.t.C, y.H * y.W);
                if (W.live()) W.check(l.kind == 5 ? launch_avgpool2(x.t.p, nullptr, 0, y.t.p, h->bf16, B, x.H, x.W, x.t.C, s)
                                                  : launch_nearest_up2(x.t.p, y.t.p, h->bf16, B, x.H, x.W, x.t.C, s));
                x = y;
                W.tap(ln, x.t);
            } else {
                x = conv(x, a.resample[l.idx], nullptr, 0, l.kind == 3 ? 2 : 1, nullptr, true);
                W.tap(ln, x.t);
            }
        }
        return x;
    };
    T2 x; x.H = a.H; x.W = a.W; x.t = Act{};
    std::vector<T2> hs;
    for (size_t i = 0; i < a.input_blocks.size() && !W.bad; ++i) {
        x = run(a.input_blocks[i], x, "input_blocks." + std::to_string(i));
        W.tap("input_blocks." + std::to_string(i), x.t);
        hs.push_back(x);
    }
    x = run(a.middle, x, "middle_block");
    W.tap("middle_block", x.t);
    for (size_t i = 0; i < a.output_blocks.size() && !W.bad; ++i) {
        const T2 sk = hs.back(); hs.pop_back();
        if (sk.H != x.H || sk.W != x.W) { W.check("UNetModel: skip shape mismatch"); break; }
        T2 cat = x;                                                    // [x ; skip] along channels, by reference
        cat.t1 = sk.t; cat.st1 = sk.st;
        x = run(a.output_blocks[i], cat, "output_blocks." + std::to_string(i));
        W.tap("output_blocks." + std::to_string(i), x.t);
    }
    const float* abo = gn_table(x, a.out_gw, a.out_gb, nullptr);
    if (W.live())
        W.check(launch_conv2d_out(x.t.p, abo, a.out_w, a.out_b, io.out, h->bf16, B, a.final_ch, x.H, x.W, c.out_channels, io.mode, io.x_noisy, io.coef,
                                  io.coef_bstride, s));
    return W.bad ? 1 : 0;
}

}  // namespace adf_api
