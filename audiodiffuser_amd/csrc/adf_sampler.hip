// Denoise wrappers (Diffusion.denoise_fn, src/models/components/diffusion.py:32-63; classifier-free guidance :49-54; dynamic threshold
// components/utils.py:19-33) and the sampler state machines: every branch of the reference loops depends on the sigma schedule only, so
// they are resolved on the host and the whole loop is enqueued (and captured) at once.  References per function below
// (src/models/components/sampler_edm.py, stochastic_sampler_edm.py).
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

namespace adf_api {

// class part of the FiLM projections for one network pass: the per-sample rows, or the null row for every sample
int cond_rows(adf_handle* h, int B, bool null_branch, FwdIO& io) {
    if (h->cdim == 0) return 0;
    if (!h->cond_on || h->cond_B != B)
        return fail(h, "class-conditional network: call adf_set_condition with the labels of this batch first");
    io.null_cond = null_branch;
    if (h->adm) return 0;            // the ADM net adds the class embedding to the time embedding before the FiLM projections (adm_forward)
    if (null_branch) { io.film2 = h->cond_film + (size_t)B * h->film_total; io.film2_bstride = 0; }
    else { io.film2 = h->cond_film; io.film2_bstride = h->film_total; }
    return 0;
}

// (allocated outside graph capture: adf_sampler_run calls this before it starts capturing)
int ensure_cfg_buffers(adf_handle* h, Plan* p) {
    if (p->cfg_c) return 0;
    const size_t wave = (size_t)p->B * h->cfg.out_channels * p->L;
    p->cfg_c = (float*)dalloc(h, wave * 4, p);
    p->cfg_n = (float*)dalloc(h, wave * 4, p);
    if (!p->cfg_c || !p->cfg_n) return fail(h, "device allocation failed for the guidance buffers");
    return 0;
}

// One denoiser evaluation.  io carries x / t / coef (preconditioning scalars already in p->coef); with classifier-free
// guidance the network runs twice (labels, null labels) in raw mode and cfg_combine applies guidance + preconditioning.
// Dynamic thresholding (EluDiffusion(dynamic_threshold = q), components/utils.py:23-33): the estimate leaves the combine kernel unclipped and is
// rescaled in place by its per-sample quantile.
int denoise_io(adf_handle* h, Plan* p, FwdIO io, float* out, hipStream_t s) {
    const bool cfg = h->cdim > 0 && h->cond_on && h->cond_scale != 1.0f;
    const bool dyn = h->dyn_q > 0.0f;
    if (!cfg && !dyn) {
        if (cond_rows(h, p->B, false, io)) return 1;
        io.out = out; io.mode = 1;
        return forward(h, p, io, s);
    }
    const long long per_sample = (long long)h->cfg.out_channels * p->L;
    const size_t wave = (size_t)p->B * per_sample;
    if (ensure_cfg_buffers(h, p)) return 1;
    if (dyn && !p->dyn_scale) {
        p->dyn_scale = (float*)dalloc(h, (size_t)p->B * 4, p);
        if (!p->dyn_scale) return fail(h, "device allocation failed for the dynamic-threshold scales");
    }
    io.mode = 0;                                     // raw network output; c_in is still applied by to_in
    io.out = p->cfg_c;
    if (cond_rows(h, p->B, false, io) || forward(h, p, io, s)) return 1;
    if (cfg) {
        io.out = p->cfg_n;
        if (cond_rows(h, p->B, true, io) || forward(h, p, io, s)) return 1;
    }
    if (const char* e = launch_cfg_combine(out, io.x_noisy, p->cfg_c, cfg ? p->cfg_n : p->cfg_c, io.coef, io.coef_bstride, cfg ? h->cond_scale : 1.0f,
                                           per_sample, (long long)wave, dyn ? 0 : 1, s))
        return fail(h, e);
    if (dyn)
        if (const char* e = launch_dyn_threshold(out, p->B, per_sample, h->dyn_q, p->dyn_scale, s)) return fail(h, e);
    return 0;
}

int denoise_scalar(adf_handle* h, Plan* p, const float* x, float sigma, float sigma_data, float* out, hipStream_t s) {
    if (const char* e = launch_edm_coef(nullptr, sigma, 1, sigma_data, p->coef, s)) return fail(h, e);
    FwdIO io;
    io.x = x; io.t = p->coef + 1; io.t_stride = 4; io.nb = 1;
    io.coef = p->coef; io.coef_bstride = 0; io.x_noisy = x;
    return denoise_io(h, p, io, out, s);
}

// ---- sampler drivers -----------------------------------------------------------------------------------

// returns the buffer holding the final sample through *result
int run_edm(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (c.nsig < N) return c.count_only ? 1 : fail(c.h, "EDMSampler: need at least num_steps sigmas");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* XH = c.count_only ? nullptr : p->sb[2];
    float* XE = c.count_only ? nullptr : p->sb[3];
    float* D = c.count_only ? nullptr : p->sb[4];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    const float gmax = fminf(d.s_churn / (float)N, (float)(std::sqrt(2.0) - 1.0));
    for (int i = 0; i < N; ++i) {
        const float sg = c.sig[i];
        const float sn = (i + 1 < c.nsig) ? c.sig[i + 1] : 0.0f;
        const float gamma = (sg >= d.s_tmin && sg <= d.s_tmax) ? gmax : 0.0f;
        float s_hat = sg;
        const float* xh = X;
        if (gamma > 0.f) {
            s_hat = sg + gamma * sg;
            const float cc = sqrtf(s_hat * s_hat - sg * sg);
            if (!c.count_only) {
                if (!p->inj_stage) return fail(c.h, "EDMSampler with churn needs injected_noise");
                if (c.ck(launch_churn(XH, X, p->inj_stage + (size_t)i * c.n, cc, d.s_noise, c.n, c.s))) return 1;
            }
            xh = XH;
        }
        if (c.den(xh, s_hat, DEN)) return 1;
        const float dt = sn - s_hat;
        if (!c.count_only && c.ck(launch_euler(XE, D, xh, DEN, s_hat, dt, c.n, c.s))) return 1;
        if (sn != 0.f && d.use_heun) {
            if (c.den(XE, sn, DEN)) return 1;
            if (!c.count_only && c.ck(launch_rk2(XN, xh, D, XE, DEN, sn, 0.5f * dt, 1.0f, 1.0f, c.n, c.s))) return 1;
            std::swap(X, XN);
        } else {
            std::swap(X, XE);
        }
    }
    *result = X;
    return 0;
}

int run_edm_alpha(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (c.nsig < N) return c.count_only ? 1 : fail(c.h, "EDMAlphaSampler: need at least num_steps sigmas");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* XE = c.count_only ? nullptr : p->sb[3];
    float* D = c.count_only ? nullptr : p->sb[4];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    const float alpha = d.alpha;
    for (int i = 0; i + 1 < N; ++i) {
        const float sg = c.sig[i], sn = c.sig[i + 1];
        const float hh = sn - sg;
        if (c.den(X, sg, DEN)) return 1;
        const float sp = sg + alpha * hh;
        if (sp != 0.f && d.use_heun) {
            if (!c.count_only && c.ck(launch_euler(XE, D, X, DEN, sg, alpha * hh, c.n, c.s))) return 1;
            if (c.den(XE, sp, DEN)) return 1;
            const float w1 = (float)(1.0 - 0.5 / (double)alpha), w2 = (float)(0.5 / (double)alpha);
            if (!c.count_only && c.ck(launch_rk2(XN, X, D, XE, DEN, sp, hh, w1, w2, c.n, c.s))) return 1;
            std::swap(X, XN);
        } else {
            if (!c.count_only && c.ck(launch_euler(XE, D, X, DEN, sg, hh, c.n, c.s))) return 1;
            std::swap(X, XE);
        }
    }
    *result = X;
    return 0;
}

// DPMSampler.get_lambda / lambd / sigma / inv_lambd (sampler_edm.py:528-556) on host fp32 scalars.  log spacing: the
// grid holds lambda = -log sigma, a torch.linspace over n + 1 points between the first and the last sigma; otherwise the
// grid IS the sigma list.
struct DpmGrid {
    std::vector<float> g;
    bool logsp;
    float lam(float v) const { return logsp ? v : -logf(v); }
    float sig(float v) const { return logsp ? expf(-v) : v; }
    float inv(float v) const { return logsp ? v : expf(-v); }
};
DpmGrid dpm_grid(const float* sig, int nsig, int n, bool logsp) {
    DpmGrid r;
    r.logsp = logsp;
    if (!logsp) { r.g.assign(sig, sig + nsig); return r; }
    const float start = -logf(sig[0]), end = -logf(sig[nsig - 1]);
    const int steps = n + 1;
    const float step = (end - start) / (float)(steps - 1);
    r.g.resize(steps);
    for (int i = 0; i < steps; ++i)                                   // torch.linspace: from the start in the first half, from the end in the second
        r.g[i] = i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
    return r;
}

int run_dpm(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const bool logsp = d.log_time_spacing != 0;
    const int steps = logsp ? d.num_steps : d.num_steps - 1;  // sampler_edm.py:526
    const int order = d.order;
    if (order < 1 || order > 3) return c.count_only ? 1 : fail(c.h, "DPMSampler: order must be 1, 2 or 3");
    if (steps < order || c.nsig < 2 || (!logsp && c.nsig < steps + 1)) return c.count_only ? 1 : fail(c.h, "DPMSampler: not enough steps / sigmas");
    const DpmGrid G = dpm_grid(c.sig, c.nsig, steps, logsp);
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* M[3] = {c.count_only ? nullptr : p->sb[6], c.count_only ? nullptr : p->sb[7], c.count_only ? nullptr : p->sb[8]};
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    // history of grid values: index 0 = most recent
    float sh[3] = {G.g[0], 0.f, 0.f};
    const bool eps = d.eps_pred != 0;
    if (c.model(X, G.sig(G.g[0]), M[0])) return 1;
    for (int step = 1; step <= steps; ++step) {
        const int ord = step < order ? step : std::min(order, steps + 1 - step);
        const float sc = G.g[step];
        const float hcur = G.lam(sc) - G.lam(sh[0]);
        DpmArgs a;
        memset(&a, 0, sizeof(a));
        a.order = ord;
        a.ratio = eps ? 1.0f : G.sig(sc) / G.sig(sh[0]);
        const float scur = G.sig(sc);
        // noise-prediction forms (:640-645, :660-662, :685-689): x - (s phi1) m0 - 0.5 (s phi1) D1_0, resp. - (s phi2) D1 - (s phi3) D2
        const float e1 = expm1f(hcur);
        a.phi1 = eps ? scur * e1 : expm1f(-hcur);
        a.m0 = M[0]; a.m1 = M[1]; a.m2 = M[2];
        if (ord == 2) {
            const float h1 = G.lam(sh[0]) - G.lam(sh[1]);
            const float r0 = h1 / hcur;
            a.inv_r0 = 1.0f / r0;
        } else if (ord == 3) {
            const float h1 = G.lam(sh[1]) - G.lam(sh[2]);
            const float h0 = G.lam(sh[0]) - G.lam(sh[1]);
            const float r0 = h0 / hcur, r1 = h1 / hcur;
            a.inv_r0 = 1.0f / r0; a.inv_r1 = 1.0f / r1;
            a.r0_frac = r0 / (r0 + r1);
            a.inv_r01 = 1.0f / (r0 + r1);
            if (eps) {
                const float p2 = e1 / hcur - 1.0f, p3 = p2 / hcur - 0.5f;
                a.phi2 = -(scur * p2);                      // the kernel forms v + phi2 D1 - phi3 D2
                a.phi3 = scur * p3;
            } else {
                a.phi2 = a.phi1 / hcur + 1.0f;
                a.phi3 = a.phi2 / hcur - 0.5f;
            }
        }
        const int last = step == steps;
        if (!c.count_only && c.ck(launch_dpm_update(XN, X, a, last, c.n, c.s))) return 1;
        std::swap(X, XN);
        sh[2] = sh[1]; sh[1] = sh[0]; sh[0] = sc;
        if (!last) {
            float* oldest = M[2];
            M[2] = M[1]; M[1] = M[0]; M[0] = oldest;
            if (c.model(X, G.sig(sc), M[0])) return 1;
        }
    }
    *result = X;
    return 0;
}

// DPMSampler with multisteps=False, x0_pred=True ("DPM-Solver-fast"): sampler_edm.py:769-805 + :568-622.  Kept as
// written: with log_time_spacing=False the grid is the whole sigma list but only len(orders) intervals are walked (the
// run stops early), and the intermediate points add a lambda-space step to a sigma before inv_lambd (:584, :604).
int run_dpm_single(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const bool logsp = d.log_time_spacing != 0;
    const int n_eff = logsp ? d.num_steps : d.num_steps - 1;
    const int order = d.order;
    if (order < 1 || order > 3) return c.count_only ? 1 : fail(c.h, "DPMSampler: order must be 1, 2 or 3");
    if (n_eff < 1 || c.nsig < 2) return c.count_only ? 1 : fail(c.h, "DPMSampler: not enough steps / sigmas");
    std::vector<int> orders;
    int K;
    if (order == 3) {
        K = n_eff / 3 + 1;
        if (n_eff % 3 == 0) { orders.assign(std::max(K - 2, 0), 3); orders.push_back(2); orders.push_back(1); }
        else { orders.assign(K - 1, 3); orders.push_back(n_eff % 3); }
    } else if (order == 2) {
        K = (n_eff + 1) / 2;
        orders.assign(n_eff / 2, 2);
        if (n_eff % 2) orders.push_back(1);
    } else {
        K = n_eff;
        orders.assign(n_eff, 1);
    }
    if (!logsp && c.nsig < (int)orders.size() + 1) return c.count_only ? 1 : fail(c.h, "DPMSampler: fewer sigmas than solver intervals");
    const DpmGrid G = dpm_grid(c.sig, c.nsig, K, logsp);
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* U = c.count_only ? nullptr : p->sb[2];
    float* E0 = c.count_only ? nullptr : p->sb[6];
    float* E1 = c.count_only ? nullptr : p->sb[7];
    float* E2 = c.count_only ? nullptr : p->sb[8];
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    auto comb = [&](float* out, const float* e1, float a, float b, float cc, int clampit) -> int {
        return c.count_only ? 0 : c.ck(launch_lincomb(out, X, E0, e1, a, b, cc, clampit, c.n, c.s));
    };
    for (size_t i = 0; i < orders.size(); ++i) {
        const float cur = G.g[i], nxt = G.g[i + 1];
        const float h = G.lam(nxt) - G.lam(cur);
        const float ratio = G.sig(nxt) / G.sig(cur);
        const int last = i + 1 == orders.size();
        if (c.model(X, G.sig(cur), E0)) return 1;
        if (d.eps_pred) {
            // noise-prediction forms (:578-579, :594-597, :617-621): every update is x - b eps + c (eps' - eps)
            const float sn = G.sig(nxt), eh = expm1f(h);
            if (orders[i] == 1) {
                if (comb(XN, nullptr, 1.0f, sn * eh, 0.f, last)) return 1;
            } else if (orders[i] == 2) {
                const float r1 = 0.5f;
                const float s1 = G.inv(cur + r1 * h);
                if (comb(U, nullptr, 1.0f, G.sig(s1) * expm1f(r1 * h), 0.f, 0)) return 1;
                if (c.model(U, G.sig(s1), E1)) return 1;
                if (comb(XN, E1, 1.0f, sn * eh, -(sn / (float)(2.0 * 0.5) * eh), last)) return 1;
            } else {
                const double r1d = 1.0 / 3.0, r2d = 2.0 / 3.0;
                const float r1 = (float)r1d, r2 = (float)r2d;
                const float s1 = G.inv(cur + r1 * h), s2 = G.inv(cur + r2 * h);
                if (comb(U, nullptr, 1.0f, G.sig(s1) * expm1f(r1 * h), 0.f, 0)) return 1;
                if (c.model(U, G.sig(s1), E1)) return 1;
                const float cu2 = -(G.sig(s2) * (float)(r2d / r1d) * (expm1f(r2 * h) / (r2 * h) - 1.0f));
                if (comb(U, E1, 1.0f, G.sig(s2) * expm1f(r2 * h), cu2, 0)) return 1;
                if (c.model(U, G.sig(s2), E2)) return 1;
                const float cx3 = -(sn / (float)r2d * (eh / h - 1.0f));
                if (comb(XN, E2, 1.0f, sn * eh, cx3, last)) return 1;
            }
        } else if (orders[i] == 1) {
            if (comb(XN, nullptr, ratio, expm1f(-h), 0.f, last)) return 1;
        } else if (orders[i] == 2) {
            const float r1 = 0.5f;
            const float s1 = G.inv(cur + r1 * h);
            if (comb(U, nullptr, G.sig(s1) / G.sig(cur), expm1f(-r1 * h), 0.f, 0)) return 1;
            if (c.model(U, G.sig(s1), E1)) return 1;
            if (comb(XN, E1, ratio, expm1f(-h), -((float)(1.0 / (2.0 * 0.5)) * expm1f(-h)), last)) return 1;
        } else {
            const double r1d = 1.0 / 3.0, r2d = 2.0 / 3.0;
            const float r1 = (float)r1d, r2 = (float)r2d;
            const float s1 = G.inv(cur + r1 * h), s2 = G.inv(cur + r2 * h);
            if (comb(U, nullptr, G.sig(s1) / G.sig(cur), expm1f(-r1 * h), 0.f, 0)) return 1;
            if (c.model(U, G.sig(s1), E1)) return 1;
            const float cu2 = (float)(r2d / r1d) * (expm1f(-r2 * h) / (r2 * h) + 1.0f);
            if (comb(U, E1, G.sig(s2) / G.sig(cur), expm1f(-r2 * h), cu2, 0)) return 1;
            if (c.model(U, G.sig(s2), E2)) return 1;
            const float cx3 = (float)(1.0 / r2d) * (expm1f(-h) / h + 1.0f);
            if (comb(XN, E2, ratio, expm1f(-h), cx3, last)) return 1;
        }
        std::swap(X, XN);
    }
    *result = X;
    return 0;
}

// DPM2MSampler: sampler_edm.py:1111-1131 (num_steps updates over sigmas[i], sigmas[i + 1]; the schedule must hold num_steps + 1
// entries -- with fewer the reference raises IndexError), :1072-1109 (step), fp32 scalars on the host
int run_dpm2m(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (N < 1 || c.nsig < N + 1) return c.count_only ? 1 : fail(c.h, "DPM2MSampler: the schedule must hold num_steps + 1 sigmas (the reference indexes sigmas[i + 1])");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* D[2] = {c.count_only ? nullptr : p->sb[5], c.count_only ? nullptr : p->sb[6]};
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    for (int i = 0; i < N; ++i) {
        const float sg = c.sig[i], sn = c.sig[i + 1];
        float* den = D[i & 1];
        const float* old = i > 0 ? D[(i + 1) & 1] : nullptr;
        if (c.den(X, sg, den)) return 1;
        if (d.reflow && !c.count_only && c.ck(launch_reflow(den, X, sg, c.n, c.s))) return 1;     // stochastic_sampler_edm.py:214-215
        const float t = -logf(sg), tn = -logf(sn);
        const float h = tn - t;
        const float ratio = fminf(expf(-tn), expf(-t)) / fmaxf(expf(-tn), expf(-t));
        if (!old || sn == 0.0f) {
            if (!c.count_only && c.ck(launch_dpm2m(XN, X, den, nullptr, ratio, expm1f(-h), 1.f, 0.f, c.n, c.s))) return 1;
        } else {
            const float h_last = t - (-logf(c.sig[i - 1]));
            const float h_min = fminf(h_last, h), h_max = fmaxf(h_last, h);
            const float r = h_max / h_min;
            const float h_d = (h_max + h_min) / 2.0f;
            const float c2 = 1.0f / (2.0f * r);
            if (!c.count_only && c.ck(launch_dpm2m(XN, X, den, old, ratio, expm1f(-h_d), 1.0f + c2, c2, c.n, c.s))) return 1;
        }
        std::swap(X, XN);
    }
    if (!c.count_only && c.ck(launch_clamp(X, c.n, c.s))) return 1;
    *result = X;
    return 0;
}

// LMSSampler.linear_multistep_coeff (sampler_edm.py:1149-1160): the integral over [t_i, t_{i+1}] of the Lagrange basis
// polynomial of node t_{i-j} among t_i .. t_{i-order+1}.  Degree <= 3, so 3-point Gauss-Legendre in double is exact (the
// reference integrates numerically with scipy quad to 1e-4 relative).
double lms_coeff(int order, const float* t, int i, int j) {
    static const double gx[3] = {-0.7745966692414834, 0.0, 0.7745966692414834};
    static const double gw[3] = {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0};
    const double a = t[i], b = t[i + 1], half = 0.5 * (b - a), mid = 0.5 * (a + b);
    double s = 0.0;
    for (int q = 0; q < 3; ++q) {
        const double tau = mid + half * gx[q];
        double prod = 1.0;
        for (int k = 0; k < order; ++k) {
            if (k == j) continue;
            prod *= (tau - (double)t[i - k]) / ((double)t[i - j] - (double)t[i - k]);
        }
        s += gw[q] * prod;
    }
    return s * half;
}

// LMSSampler.forward: sampler_edm.py:1162-1190 (num_steps - 1 evaluations, history of `order` derivatives, final clamp)
int run_lms(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps, order = d.order;
    if (order < 1 || order > 4) return c.count_only ? 1 : fail(c.h, "LMSSampler: order must be 1..4");
    if (N < 2 || c.nsig < N) return c.count_only ? 1 : fail(c.h, "LMSSampler: need at least num_steps (>= 2) sigmas");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    float* D[4] = {nullptr, nullptr, nullptr, nullptr};
    if (!c.count_only) { D[0] = p->sb[1]; D[1] = p->sb[2]; D[2] = p->sb[3]; D[3] = p->sb[4]; }
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    for (int i = 0; i + 1 < N; ++i) {
        if (c.den(X, c.sig[i], DEN)) return 1;
        const int cur = std::min(i + 1, order);
        LmsArgs a;
        memset(&a, 0, sizeof(a));
        a.order = cur;
        for (int j = 0; j < cur; ++j) a.c[j] = (float)lms_coeff(cur, c.sig, i, j);
        a.dcur = D[i & 3];
        a.d1 = D[(i + 3) & 3]; a.d2 = D[(i + 2) & 3]; a.d3 = D[(i + 1) & 3];
        if (!c.count_only && c.ck(launch_lms(X, DEN, c.sig[i], a, c.n, c.s))) return 1;
    }
    if (!c.count_only && c.ck(launch_clamp(X, c.n, c.s))) return 1;
    *result = X;
    return 0;
}

// DPM2Sampler: sampler_edm.py:470-493 (loop over num_steps-1 steps, final clamp), :428-468 (step).  As written in
// the reference the churned point only feeds the first derivative; both updates start from the un-churned x.
int run_dpm2(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (N < 2 || c.nsig < N) return c.count_only ? 1 : fail(c.h, "DPM2Sampler: need at least num_steps (>= 2) sigmas");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* XH = c.count_only ? nullptr : p->sb[2];
    float* X2 = c.count_only ? nullptr : p->sb[3];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    const float gmax = fminf(d.s_churn / (float)N, (float)(std::sqrt(2.0) - 1.0));
    for (int i = 0; i + 1 < N; ++i) {
        const float sg = c.sig[i], sn = c.sig[i + 1];
        const float gamma = (sg >= d.s_tmin && sg <= d.s_tmax) ? gmax : 0.0f;
        const float s_hat = sg + gamma * sg;
        const float* xh = X;
        if (gamma > 0.f) {
            if (!c.count_only) {
                if (!p->inj_stage) return fail(c.h, "DPM2Sampler with churn needs injected_noise");
                const float cc = sqrtf(s_hat * s_hat - sg * sg);
                if (c.ck(launch_churn(XH, X, p->inj_stage + (size_t)i * c.n, cc, d.s_noise, c.n, c.s))) return 1;
            }
            xh = XH;
        }
        if (c.den(xh, s_hat, DEN)) return 1;
        if (sn == 0.0f) {
            if (!c.count_only && c.ck(launch_dstep(XN, X, xh, DEN, s_hat, sn - s_hat, c.n, c.s))) return 1;
        } else {
            const float lh = logf(s_hat), ln = logf(sn);
            const float s_mid = expf(lh + 0.5f * (ln - lh));                 // log().lerp(log(), 0.5).exp() in fp32
            if (!c.count_only && c.ck(launch_dstep(X2, X, xh, DEN, s_hat, s_mid - s_hat, c.n, c.s))) return 1;
            if (c.den(X2, s_mid, DEN)) return 1;
            if (!c.count_only && c.ck(launch_dstep(XN, X, X2, DEN, s_mid, sn - s_hat, c.n, c.s))) return 1;
        }
        std::swap(X, XN);
    }
    if (!c.count_only && c.ck(launch_clamp(X, c.n, c.s))) return 1;
    *result = X;
    return 0;
}

// ADPM2Sampler: stochastic_sampler_edm.py:85-100 (loop, final clamp), :53-83 (step), :29-32 (get_sigmas); fp32 scalars
int run_adpm2(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (N < 2 || c.nsig < N) return c.count_only ? 1 : fail(c.h, "ADPM2Sampler: need at least num_steps (>= 2) sigmas");
    if (!(d.rho > 0.f)) return c.count_only ? 1 : fail(c.h, "ADPM2Sampler: rho must be positive");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* XM = c.count_only ? nullptr : p->sb[3];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    if (!c.count_only && !p->inj_stage) return fail(c.h, "ADPM2Sampler needs injected_noise (one draw per step)");
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    for (int i = 0; i + 1 < N; ++i) {
        const float sg = c.sig[i], sn = c.sig[i + 1];
        const float up_raw = d.eta * sqrtf(sn * sn * (sg * sg - sn * sn) / (sg * sg));
        const float s_up = sn < up_raw ? sn : up_raw;                          // python min(sigma_next, ...)
        const float s_down = sqrtf(sn * sn - s_up * s_up);
        const float inv = 1.0f / d.rho;
        const float s_mid = powf((powf(sg, inv) + powf(s_down, inv)) / 2.0f, d.rho);
        if (c.den(X, sg, DEN)) return 1;
        if (!c.count_only && c.ck(launch_dstep(XM, X, X, DEN, sg, s_mid - sg, c.n, c.s))) return 1;
        if (c.den(XM, s_mid, DEN)) return 1;
        if (!c.count_only) {
            if (c.ck(launch_dstep(XN, X, XM, DEN, s_mid, s_down - sg, c.n, c.s))) return 1;
            if (c.ck(launch_churn(X, XN, p->inj_stage + (size_t)i * c.n, s_up, 1.0f, c.n, c.s))) return 1;   // x + sigma_up * randn
        }
    }
    if (!c.count_only && c.ck(launch_clamp(X, c.n, c.s))) return 1;
    *result = X;
    return 0;
}

// ADPMPP2SSampler: stochastic_sampler_edm.py:162-178 (loop, final clamp), :117-160 (step), :29-32 (get_sigmas); fp32 scalars.  A draw is
// consumed only by a step whose sigma_next is positive (:158): adpmpp2s_draws() counts them for the injected-noise check.
int adpmpp2s_draws(const float* sig, int nsig, int N) {
    int k = 0;
    for (int i = 0; i + 1 < N && i + 1 < nsig; ++i) k += sig[i + 1] > 0.0f;
    return k;
}
int run_adpmpp2s(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const int N = d.num_steps;
    if (N < 2 || c.nsig < N) return c.count_only ? 1 : fail(c.h, "ADPMPP2SSampler: need at least num_steps (>= 2) sigmas");
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* X2 = c.count_only ? nullptr : p->sb[3];
    float* DEN = c.count_only ? nullptr : p->sb[5];
    if (!c.count_only && !p->inj_stage && adpmpp2s_draws(c.sig, c.nsig, N) > 0) return fail(c.h, "ADPMPP2SSampler needs injected_noise (one draw per step with sigma_next > 0)");
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    int k = 0;
    for (int i = 0; i + 1 < N; ++i) {
        const float sg = c.sig[i], sn = c.sig[i + 1];
        const float up_raw = d.eta * sqrtf(sn * sn * (sg * sg - sn * sn) / (sg * sg));
        const float s_up = sn < up_raw ? sn : up_raw;                          // python min(sigma_next, ...)
        const float s_down = sqrtf(sn * sn - s_up * s_up);
        if (c.den(X, sg, DEN)) return 1;
        if (s_down == 0.0f) {                                                  // Euler step to sigma_down (:136-140)
            if (!c.count_only && c.ck(launch_dstep(XN, X, X, DEN, sg, s_down - sg, c.n, c.s))) return 1;
        } else {
            const float t = -logf(sg), tn = -logf(s_down);
            const float h = tn - t;
            const float sm = t + 0.5f * h;
            const float sig_mid = expf(-sm);
            if (!c.count_only && c.ck(launch_dpm2m(X2, X, DEN, nullptr, sig_mid / expf(-t), expm1f(-h * 0.5f), 1.f, 0.f, c.n, c.s))) return 1;
            if (c.den(X2, sig_mid, DEN)) return 1;
            if (!c.count_only && c.ck(launch_dpm2m(XN, X, DEN, nullptr, expf(-tn) / expf(-t), expm1f(-h), 1.f, 0.f, c.n, c.s))) return 1;
        }
        if (sn > 0.0f) {
            if (!c.count_only && c.ck(launch_churn(X, XN, p->inj_stage + (size_t)k * c.n, s_up, 1.0f, c.n, c.s))) return 1;   // x + sigma_up * randn
            ++k;
        } else {
            std::swap(X, XN);
        }
    }
    if (!c.count_only && c.ck(launch_clamp(X, c.n, c.s))) return 1;
    *result = X;
    return 0;
}

// UniPCSampler.forward (sampler_edm.py:996-1053, variant 'bh2').  Every coefficient depends on the grid only: computed on the host
// in fp32 in the reference's order of operations (the small solves of :934, :942 by Gaussian elimination with partial pivoting, as
// LAPACK's gesv does); one launch per predictor / corrector formula.
static void unipc_solve(int n, float A[3][3], float* b, float* x) {
    int piv[3] = {0, 1, 2};
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabsf(A[piv[i]][k]) > fabsf(A[piv[p]][k])) p = i;
        std::swap(piv[k], piv[p]);
        for (int i = k + 1; i < n; ++i) {
            const float f = A[piv[i]][k] / A[piv[k]][k];
            for (int j = k; j < n; ++j) A[piv[i]][j] -= f * A[piv[k]][j];
            b[piv[i]] -= f * b[piv[k]];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        float acc = b[piv[k]];
        for (int j = k + 1; j < n; ++j) acc -= A[piv[k]][j] * x[j];
        x[k] = acc / A[piv[k]][k];
    }
}

int run_unipc(SamplerCtx& c, float** result) {
    const adf_sampler_desc& d = *c.d;
    const bool logsp = d.log_time_spacing != 0, eps = d.eps_pred != 0;
    const int steps = logsp ? d.num_steps : d.num_steps - 1;          // :828
    const int order = d.order;
    if (order < 1 || order > 3) return c.count_only ? 1 : fail(c.h, "UniPCSampler: order must be 1, 2 or 3");
    if (steps < order || c.nsig < 2 || (!logsp && c.nsig < steps + 1)) return c.count_only ? 1 : fail(c.h, "UniPCSampler: not enough steps / sigmas");
    const DpmGrid G = dpm_grid(c.sig, c.nsig, steps, logsp);
    Plan* p = c.p;
    float* X = c.count_only ? nullptr : p->sb[0];
    float* XN = c.count_only ? nullptr : p->sb[1];
    float* XT = c.count_only ? nullptr : p->sb[2];
    float* MB[4] = {c.count_only ? nullptr : p->sb[6], c.count_only ? nullptr : p->sb[7], c.count_only ? nullptr : p->sb[8], c.count_only ? nullptr : p->sb[9]};
    if (!c.count_only && c.ck(launch_scale(X, p->noise_stage, c.sig[0], c.n, c.s))) return 1;
    // history, oldest first (as the reference's lists); a free buffer of MB receives the next model value
    std::vector<float*> ml; std::vector<float> gl;
    auto free_buf = [&]() -> float* { for (float* b : MB) if (std::find(ml.begin(), ml.end(), b) == ml.end()) return b; return MB[0]; };
    float* m_first = free_buf();
    if (c.model(X, G.sig(G.g[0]), m_first)) return 1;
    ml.push_back(m_first); gl.push_back(G.g[0]);
    auto update = [&](float g_cur, int ord, bool corr, float** x_io, float** m_out) -> int {
        const float g0 = gl.back();
        const float h = G.lam(g_cur) - G.lam(g0);
        float rks[3]; int K = 0;
        const float* mk[2] = {nullptr, nullptr};
        for (int i = 1; i < ord; ++i) { rks[K] = (G.lam(gl[gl.size() - 1 - i]) - G.lam(g0)) / h; mk[K] = ml[ml.size() - 1 - i]; ++K; }
        rks[K] = 1.0f;
        const float hh = eps ? h : -h;
        const float h_phi_1 = expm1f(hh);
        float h_phi_k = h_phi_1 / hh - 1.0f;
        const float B_h = expm1f(hh);
        float R[3][3], bb[3];
        float fact = 1.0f;
        for (int i = 1; i <= ord; ++i) {
            for (int j = 0; j < ord; ++j) R[i - 1][j] = i == 1 ? 1.0f : (i == 2 ? rks[j] : rks[j] * rks[j]);
            bb[i - 1] = h_phi_k * fact / B_h;
            fact *= (float)(i + 1);
            h_phi_k = h_phi_k / hh - 1.0f / fact;
        }
        float rhos_p[3] = {0.f, 0.f, 0.f}, rhos_c[3] = {0.f, 0.f, 0.f};
        if (K > 0) {
            if (ord == 2) rhos_p[0] = 0.5f;
            else { float A2[3][3], b2[3]; for (int i = 0; i < ord - 1; ++i) { b2[i] = bb[i]; for (int j = 0; j < ord - 1; ++j) A2[i][j] = R[i][j]; } unipc_solve(ord - 1, A2, b2, rhos_p); }
        }
        if (corr) {
            if (ord == 1) rhos_c[0] = 0.5f;
            else { float A2[3][3], b2[3]; for (int i = 0; i < ord; ++i) { b2[i] = bb[i]; for (int j = 0; j < ord; ++j) A2[i][j] = R[i][j]; } unipc_solve(ord, A2, b2, rhos_c); }
        }
        const float sc = G.sig(g_cur);
        UniPcArgs u;
        memset(&u, 0, sizeof(u));
        u.a = eps ? 1.0f : sc / G.sig(g0);
        u.hp = eps ? sc * h_phi_1 : h_phi_1;
        u.sb = eps ? sc * B_h : B_h;
        u.K = K; u.m0 = ml.back(); u.m[0] = mk[0]; u.m[1] = mk[1];
        for (int k = 0; k < K; ++k) { u.rk[k] = rks[k]; u.rho[k] = rhos_p[k]; }
        u.mt = nullptr;
        float* xin = *x_io;
        float* xt = corr ? XT : (xin == X ? XN : X);
        if (!c.count_only && c.ck(launch_unipc(xt, xin, u, c.n, c.s))) return 1;      // predictor (:951-957 / :973-979)
        *m_out = nullptr;
        if (corr) {
            float* mt = free_buf();
            if (c.model(xt, sc, mt)) return 1;
            for (int k = 0; k < K; ++k) u.rho[k] = rhos_c[k];
            u.rho_t = rhos_c[ord - 1]; u.mt = mt;
            float* xo = xin == X ? XN : X;
            if (!c.count_only && c.ck(launch_unipc(xo, xin, u, c.n, c.s))) return 1;  // corrector (:959-967 / :981-990)
            *m_out = mt; *x_io = xo;
        } else {
            *x_io = xt;
        }
        return 0;
    };
    float* x = X;
    for (int step = 1; step < order; ++step) {                         // :1013-1022
        float* m = nullptr;
        if (update(G.g[step], step, true, &x, &m)) return 1;
        gl.push_back(G.g[step]); ml.push_back(m);
    }
    for (int step = order; step <= steps; ++step) {                    // :1025-1051
        float* m = nullptr;
        const int so = order < steps + 1 - step ? order : steps + 1 - step;
        if (update(G.g[step], so, step != steps, &x, &m)) return 1;
        for (int i = 0; i + 1 < order; ++i) { gl[i] = gl[i + 1]; ml[i] = ml[i + 1]; }
        gl.back() = G.g[step];
        if (step < steps) ml.back() = m;
    }
    if (!c.count_only && c.ck(launch_clamp(x, c.n, c.s))) return 1;
    *result = x;
    return 0;
}

int run_sampler(SamplerCtx& c, float** result) {
    switch (c.d->kind) {
        case ADF_SAMPLER_DPM2: return run_dpm2(c, result);
        case ADF_SAMPLER_ADPM2: return run_adpm2(c, result);
        case ADF_SAMPLER_EDM: return run_edm(c, result);
        case ADF_SAMPLER_EDM_ALPHA: return run_edm_alpha(c, result);
        case ADF_SAMPLER_DPM_MULTISTEP: return run_dpm(c, result);
        case ADF_SAMPLER_DPM_SINGLESTEP: return run_dpm_single(c, result);
        case ADF_SAMPLER_LMS: return run_lms(c, result);
        case ADF_SAMPLER_DPM2M: return run_dpm2m(c, result);
        case ADF_SAMPLER_UNIPC: return run_unipc(c, result);
        case ADF_SAMPLER_ADPMPP2S: return run_adpmpp2s(c, result);
        default: return c.count_only ? 1 : fail(c.h, "unknown sampler kind");
    }
}

}  // namespace adf_api
