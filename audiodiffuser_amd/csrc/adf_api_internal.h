// Internals shared by the translation units behind the C ABI (include/audiodiffuser_amd.h): the handle, the per-(B, L) workspace
// ("plan"), the weight registry, the network walker.  Round 3 split of what was one 2,400-line adf_api.hip:
//   adf_api.hip            handle life cycle, weights, workspaces, the extern "C" entry points
//   adf_net_unet1d.hip     UNet1dBase: registry + walk (unet1d.py:771-816)
//   adf_net_wavenet.hip    WaveNetNoise (wavenet.py:153-180)
//   adf_net_adm.hip        ADM UNetModel (unet2d_oai.py:382-635)
//   adf_sampler.hip        denoise wrappers and the sampler state machines (sampler_edm.py, stochastic_sampler_edm.py)
//   adf_bench_replay.hip   adf_bench_* instrumentation
#pragma once
#include "../../include/audiodiffuser_amd.h"
#include "adf_gemm.h"
#include "adf_kernels.h"
#include "adf_wavenet.h"
#include "adf_conv2d.h"
#include "adf_transformer.h"
#include "adf_resblock_small.h"
#include "adf_resblock_split.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>


namespace adf_api {
using namespace adf;

extern std::string g_create_error;

struct ConvW {
    void* w = nullptr;
    void* wfrag = nullptr;   // transformer 1x1 weights (bf16): second copy in MFMA-fragment order (adf_transformer.h)
    float* bias = nullptr;
    int cout = 0, cin = 0, K = 0, n = 0, n_pad = 0, nchunk = 0, taps = 0, f = 0;
};
struct ResW {
    int cin = 0, cout = 0, film_off = 0;
    float *g1w = nullptr, *g1b = nullptr, *g2w = nullptr, *g2b = nullptr;
    ConvW c1, c2, cr;
    bool has_res = false;
};
struct TrW {
    int c = 0, mid = 0;
    float *lnw = nullptr, *lnb = nullptr, *g0 = nullptr, *g3 = nullptr;
    ConvW qkv, proj, ff1, ff2;
};
struct DownW { ConvW down; std::vector<ResW> blocks; bool attn = false; TrW tr; int factor = 1, cin = 0, cout = 0; };
// up3: the same transposed conv as a 3-tap stride-1 conv with f * cout output columns (phase-major; [B][f L][cout] IS [B][L][f cout] in memory) -- the shape
// conv_gemm_rb_kernel<.., RAW> is written for; packed beside `up` when the factor is even and f * cout is 128 or 256 (bf16 mode)
struct UpW { std::vector<ResW> blocks; bool attn = false; TrW tr; ConvW up; ConvW up3; int factor = 1, cin = 0, cout = 0; bool nearest = false; };

struct Slot {
    int kind = 0;  // 0 = fp32 copy, 1 = pack conv/linear, 2 = pack transposed conv
    void* dst = nullptr;
    void* frag = nullptr;    // also repacked to ConvW::wfrag after packing
    void* dst3 = nullptr; int n_pad3 = 0;   // kind 2: also packed in the 3-tap phase form (UpW::up3)
    float* rep = nullptr; int rep_n = 0;    // kind 0: also copied rep_n times back to back (the bias of that form)
    int64_t numel = 0;
    bool loaded = false;
    int cout = 0, cin = 0, K = 0, f = 0, n_offset = 0, n_pad = 0, nchunk = 0, taps = 0;
};

struct Act { void* p = nullptr; int C = 0, L = 0; double* stats = nullptr; };
struct TapRec { std::string name; void* p; int C, L; int f32 = 0; float scale = 1.0f; };   // f32: an fp32 buffer whatever the storage mode

// WaveNetNoise (wavenet.py:153-180): a weight-normed conv keeps the state-dict tensors (bias, 0-dim g, v) in fp32 and a packed
// GEMM operand of the effective weight v * g / ||v||, rebuilt when a tensor was (re)loaded
struct WnConv {
    float *bias = nullptr, *g = nullptr, *v = nullptr;
    void* packed = nullptr;
    int cout = 0, cin = 0, K = 0;
};
// ADM-style 2-D U-Net (unet2d_oai.py:382-635): the module list of UNetModel.__init__ as data
struct AdmRes { int cin = 0, cout = 0, film_off = 0; float *g1w = nullptr, *g1b = nullptr, *g2w = nullptr, *g2b = nullptr; ConvW c1, c2, skip; bool has_skip = false;
                int updown = 0; };        // 1: ResBlock(up=True), 2: ResBlock(down=True) (resblock_updown, unet2d_oai.py:197-207, :249-254)
struct AdmAttn { int c = 0, heads = 0; float *gw = nullptr, *gb = nullptr; ConvW qkv, proj; float* qkv_tmp = nullptr; };
struct AdmLayer { int kind; int idx; };       // kind: 0 input conv, 1 ResBlock, 2 AttentionBlock, 3 Downsample (conv), 4 Upsample (conv), 5 average pool, 6 nearest x 2
struct AdmW {
    adf_adm_config cfg;
    int H = 0, W = 0;                    // image shape of the calls that follow (adf_set_image_shape)
    std::vector<AdmRes> res;
    std::vector<AdmAttn> attn;
    std::vector<ConvW> resample;
    std::vector<std::vector<AdmLayer>> input_blocks, output_blocks;
    std::vector<AdmLayer> middle;
    std::vector<int> skip_ch;            // channels of the input-block outputs, in push order
    float *in_w = nullptr, *in_b = nullptr, *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr;
    float *out_gw = nullptr, *out_gb = nullptr, *out_w = nullptr, *out_b = nullptr;
    int input_ch = 0, final_ch = 0;
    int fg = 4;                          // channels per fine statistics group: gcd of every GroupNorm group size of the net (incl. the skip concats)
};

struct WnW {
    adf_wavenet_config cfg;
    WnConv in, sp;
    std::vector<WnConv> dil, outp;
    float *fc1w = nullptr, *fc1b = nullptr, *fc2w = nullptr, *fc2b = nullptr, *out_w = nullptr, *out_b = nullptr;
    double* sumsq = nullptr;             // scratch of the norm reduction
    bool packed = false;
};
struct RbRec { std::string name; GemmArgs g1, g2; int cin, cout, L; };

struct Plan {
    int B = 0, L = 0;
    char* arena = nullptr; size_t arena_bytes = 0, arena_off = 0;
    char* stats = nullptr; size_t stats_bytes = 0, stats_off = 0;
    bool dry = false;
    std::vector<TapRec> taps;
    std::vector<RbRec> rbs;
    float *temb = nullptr, *film = nullptr, *coef = nullptr;
    // per sampler run: (c_in, c_noise, c_skip, c_out), sigma embedding and the FiLM projections of EVERY denoiser evaluation of
    // the run, computed by three launches at the head of the loop (sigma is uniform over the batch and the whole schedule is
    // known on the host) instead of three launches per evaluation
    float *coef_all = nullptr, *temb_all = nullptr, *film_all = nullptr;
    int pre_cap = 0;
    // sampler state (fp32 [B][C][L] each)
    float* sb[10] = {nullptr};
    float* noise_stage = nullptr; float* out_stage = nullptr; float* inj_stage = nullptr; size_t inj_cap = 0;
    float* cfg_c = nullptr; float* cfg_n = nullptr;      // raw network outputs of the two CFG branches
    float* dyn_scale = nullptr;                          // [B] per-sample scales of the dynamic threshold
    // captured sampler loops, most recently used first; at most kMaxGraphsPerPlan are kept (the oldest is destroyed)
    std::vector<std::pair<std::string, hipGraphExec_t>> graphs;
    std::vector<void*> allocs;                            // device memory owned by this plan (released when the plan is evicted)
    int64_t bytes = 0;
    unsigned long long last_use = 0;
    // timing replay buffers of adf_bench_resblock (rotating copies of one layer's operands), sized on first use
    char* bench_buf = nullptr; size_t bench_cap = 0;
    // WaveNetNoise: the layer launches of the last pass, for adf_bench_wavenet_layer
    WnIO wn_io; std::vector<WnLayerArgs> wn_layers;
};
constexpr size_t kMaxGraphsPerPlan = 8;
constexpr size_t kMaxPlans = 4;      // (B, L) workspaces kept per handle; the least recently used one is released beyond that


}  // namespace adf_api

using namespace adf_api;      // (this header is private to the translation units behind the C ABI)

struct adf_handle {
    adf_net_config cfg;
    int device = 0;                     // the device that was current at adf_create: every entry point runs on it
    unsigned long long use_clock = 0;
    bool bf16 = false;
    bool x3 = false;                    // ADF_DTYPE_F32X3: storage and every non-GEMM kernel as fp32 (bf16 == false), GEMM operands split into bf16 hi + lo
    int gemm_dtype() const { return bf16 ? 1 : (x3 ? 2 : 0); }      // the `dtype` of launch_conv_gemm / launch_pack_weight
    int esz = 4, kc = 32;
    std::string err;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    std::vector<std::string> names;
    std::map<std::string, Slot> slots;
    float *to_in_w = nullptr, *to_out_w = nullptr, *fourier = nullptr, *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr,
          *t_b2 = nullptr, *film_w = nullptr, *film_b = nullptr;
    int film_total = 0;
    // class conditioning (LabelEmbedder) and the state set by adf_set_condition
    float *lab_null = nullptr, *lab_emb = nullptr, *lab_lnw = nullptr, *lab_lnb = nullptr, *lab_w1 = nullptr, *lab_b1 = nullptr,
          *lab_w2 = nullptr, *lab_b2 = nullptr;
    int cdim = 0;                       // width of the class embedding (4 * channels) or 0
    bool cond_on = false;
    int cond_B = 0;
    float cond_scale = 1.0f;
    float dyn_q = 0.0f;                                  // > 0: dynamic thresholding at this quantile instead of clamp(-1, 1) (adf_set_dynamic_threshold)
    long long* cond_classes = nullptr;  // [cond_B]
    float* cond_emb = nullptr;          // [cond_B + 1][cdim], last row = null embedding
    float* cond_film = nullptr;         // [cond_B + 1][film_total]: class part of every FiLM projection
    int cond_cap = 0;
    std::vector<DownW> downs;
    ResW mid_pre, mid_post;
    TrW mid_tr;
    std::vector<UpW> ups;
    // (B, L, H): H = image height of a UNetModel handle (W = L / H), 0 otherwise -- two image shapes with equal H * W must not share
    // a workspace: captured graphs and tap shapes carry the conv2d geometry
    std::map<std::tuple<int, int, int>, Plan*> plans;
    Plan* last_plan = nullptr;
    WnW* wn = nullptr;                  // non-null: the handle is a WaveNetNoise (adf_wavenet_create), not a UNet1dBase
    AdmW* adm = nullptr;                // non-null: the handle is an ADM-style UNetModel (adf_adm_create)
    // graphs are captured and replayed on a library-owned stream (the caller's stream may be the legacy
    // default stream, which cannot be captured); it is fenced against the caller's stream with events
    hipStream_t gstream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    adf_run_counters ctr{};             // what the device loop has done so far (adf_get_counters): lets a test tell it from a host-side loop
};

namespace adf_api {

inline int fail(adf_handle* h, const std::string& m) { h->err = m; return 1; }

// Makes the handle's device current for the duration of a C entry point (and restores the caller's afterwards): buffers,
// kernel attributes and launches of one handle all belong to the device it was created on, whatever is current in the caller.
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(const adf_handle* h) {
        if (!h) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (cur != h->device) {
            if (hipSetDevice(h->device) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ADF_ON_DEVICE(h)                                                            \
    DeviceScope adf_scope_(h);                                                      \
    if (!adf_scope_.ok) return fail(h, "could not make the handle's device current")

// device memory owned by the handle (weights, condition buffers) or, with `owner`, by one (B, L) plan
inline void* dalloc(adf_handle* h, size_t bytes, Plan* owner = nullptr) {
    void* p = nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, bytes);
    (owner ? owner->allocs : h->allocs).push_back(p);
    if (owner) owner->bytes += (int64_t)bytes;
    h->bytes += (int64_t)bytes;
    return p;
}
inline void dfree(adf_handle* h, void* ptr, size_t bytes, Plan* owner = nullptr) {
    if (!ptr) return;
    std::vector<void*>& v = owner ? owner->allocs : h->allocs;
    auto it = std::find(v.begin(), v.end(), ptr);
    if (it != v.end()) v.erase(it);
    bytes = (bytes + 255) & ~(size_t)255;
    if (owner) owner->bytes -= (int64_t)bytes;
    h->bytes -= (int64_t)bytes;
    (void)hipFree(ptr);
}
inline void drop_graphs(Plan* p) {
    for (auto& g : p->graphs) (void)hipGraphExecDestroy(g.second);
    p->graphs.clear();
}
// (the caller has synchronised the device if work of this plan may still be in flight)
inline void destroy_plan(adf_handle* h, Plan* p) {
    drop_graphs(p);
    for (void* q : p->allocs) (void)hipFree(q);
    h->bytes -= p->bytes;
    if (h->last_plan == p) h->last_plan = nullptr;
    delete p;
}
// ---- weight registry ---------------------------------------------------------------------------
struct Registrar {
    adf_handle* h;
    bool ok = true;
    float* reg_f32(const std::string& name, int64_t numel, float* dst = nullptr) {
        if (!dst) dst = (float*)dalloc(h, (size_t)numel * 4);
        if (!dst) { ok = false; return nullptr; }
        Slot s; s.kind = 0; s.dst = dst; s.numel = numel;
        h->names.push_back(name); h->slots[name] = s;
        return dst;
    }
    // Conv1d / Linear weight (cout, cin, K) packed as GEMM operand; several tensors may share one packed
    // buffer at different row offsets (fused qkv).
    void reg_pack(const std::string& name, ConvW& w, int cout, int cin, int K, int n_offset, int n_total, bool transposed, int f) {
        if (!w.w) {
            w.cin = cin; w.K = K; w.f = f;
            w.taps = transposed ? 2 : K;
            w.n = n_total; w.n_pad = round_up(n_total, 32);
            w.nchunk = ceil_div(cin, h->kc);
            // + kTapGroup slabs of 128 rows: the kernel's weight staging loads are unguarded (adf_gemm.h)
            w.w = dalloc(h, ((size_t)w.nchunk * w.taps * w.n_pad + (size_t)kTapGroup * (w.n_pad + 128)) * kRowBytes);
            if (!w.w) { ok = false; return; }
        }
        w.cout = cout;
        Slot s; s.kind = transposed ? 2 : 1; s.dst = w.w; s.numel = (int64_t)cout * cin * K;
        s.cout = cout; s.cin = cin; s.K = K; s.f = f; s.n_offset = n_offset; s.n_pad = w.n_pad; s.nchunk = w.nchunk; s.taps = w.taps;
        h->names.push_back(name); h->slots[name] = s;
    }
    void conv(const std::string& pre, ConvW& w, int cout, int cin, int K, bool bias) {
        reg_pack(pre + ".weight", w, cout, cin, K, 0, cout, false, 0);
        if (bias) w.bias = reg_f32(pre + ".bias", cout);
    }
    // Strided Conv1d (kernel f*km + 1, stride f, pad f*(km/2)) folded to a stride-1 conv with km + 1 taps over
    // f*cin channels: the contiguous [L][cin] input is the same memory as [L/f][f*cin], so the fast stride-1 GEMM
    // kernels apply unchanged (the folded taps beyond the real kernel length are zero weights)
    void conv_folded(const std::string& pre, ConvW& w, int cout, int cin, int K, int f) {
        w.cin = f * cin; w.K = K; w.f = f;
        w.taps = (K - 1) / f + 1;
        w.n = cout; w.n_pad = round_up(cout, 32);
        w.nchunk = ceil_div(f * cin, h->kc);
        w.w = dalloc(h, ((size_t)w.nchunk * w.taps * w.n_pad + (size_t)kTapGroup * (w.n_pad + 128)) * kRowBytes);
        if (!w.w) { ok = false; return; }
        w.cout = cout;
        Slot s; s.kind = 3; s.dst = w.w; s.numel = (int64_t)cout * cin * K;
        s.cout = cout; s.cin = cin; s.K = K; s.f = f; s.n_offset = 0; s.n_pad = w.n_pad; s.nchunk = w.nchunk; s.taps = w.taps;
        h->names.push_back(pre + ".weight"); h->slots[pre + ".weight"] = s;
        w.bias = reg_f32(pre + ".bias", cout);
    }
    void resblock(const std::string& pre, ResW& r, int cin, int cout, int temb) {
        r.cin = cin; r.cout = cout;
        r.film_off = h->film_total;
        h->film_total += 2 * cout;
        // FiLM weights are registered later (one concatenated matrix), remember the order via names
        film_names.push_back({pre, r.film_off, 2 * cout});
        r.g1w = reg_f32(pre + ".block1.groupnorm.weight", cin);
        r.g1b = reg_f32(pre + ".block1.groupnorm.bias", cin);
        conv(pre + ".block1.project", r.c1, cout, cin, 3, true);
        r.g2w = reg_f32(pre + ".block2.groupnorm.weight", cout);
        r.g2b = reg_f32(pre + ".block2.groupnorm.bias", cout);
        conv(pre + ".block2.project", r.c2, cout, cout, 3, true);
        r.has_res = cin != cout;
        if (r.has_res) conv(pre + ".to_out", r.cr, cout, cin, 1, true);
        if (h->bf16 && (cout == 256 || cout == 128) && (cin == cout || cin == 2 * cout)) {   // fragment-major copies: adf_resblock_small.h, adf_gemm_tile.h
            const std::pair<const char*, ConvW*> m[] = {{".block1.project.weight", &r.c1}, {".block2.project.weight", &r.c2}, {".to_out.weight", &r.cr}};
            for (const auto& kv : m) {
                ConvW* w = kv.second;
                if (!w->w) continue;
                w->wfrag = dalloc(h, (size_t)w->nchunk * w->taps * w->n_pad * kRowBytes);
                if (!w->wfrag) { ok = false; return; }
                h->slots[pre + kv.first].frag = w->wfrag;
            }
        }
        (void)temb;
    }
    void transformer(const std::string& pre, TrW& t, int c, int mult) {
        t.c = c; t.mid = c * mult;
        t.lnw = reg_f32(pre + ".norm.weight", c);
        t.lnb = reg_f32(pre + ".norm.bias", c);
        reg_pack(pre + ".attention.to_q.weight", t.qkv, c, c, 1, 0, 3 * c, false, 0);
        reg_pack(pre + ".attention.to_kv.weight", t.qkv, 2 * c, c, 1, c, 3 * c, false, 0);
        t.qkv.cout = 3 * c;
        conv(pre + ".attention.to_out", t.proj, c, c, 1, false);
        t.g0 = reg_f32(pre + ".feed_forward.0.g", c);
        conv(pre + ".feed_forward.1", t.ff1, t.mid, c, 1, false);
        t.g3 = reg_f32(pre + ".feed_forward.3.g", t.mid);
        conv(pre + ".feed_forward.4", t.ff2, c, t.mid, 1, false);
        if (h->bf16) {
            for (ConvW* w : {&t.qkv, &t.proj, &t.ff1, &t.ff2}) {
                w->wfrag = dalloc(h, (size_t)w->nchunk * w->n_pad * kRowBytes);
                if (!w->wfrag) { ok = false; return; }
            }
            const std::pair<const char*, ConvW*> m[] = {{".attention.to_q.weight", &t.qkv}, {".attention.to_kv.weight", &t.qkv},
                                                       {".attention.to_out.weight", &t.proj}, {".feed_forward.1.weight", &t.ff1},
                                                       {".feed_forward.4.weight", &t.ff2}};
            for (const auto& kv : m) h->slots[pre + kv.first].frag = kv.second->wfrag;
        }
    }
    struct FilmName { std::string pre; int off, rows; };
    std::vector<FilmName> film_names;
};


int build_weights(adf_handle* h);
// ---- per-(B, L) plan -----------------------------------------------------------------------------
struct Walker {
    adf_handle* h;
    Plan* p;
    hipStream_t s;
    bool bad = false;
    const float* film2 = nullptr;       // class part of the FiLM projections for this pass (FwdIO::film2)
    int film2_bstride = 0;
    const float* film = nullptr;        // time part: Plan::film, or the evaluation's row of Plan::film_all

    void check(const char* e) { if (e && !bad) { bad = true; h->err = e; } }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        const size_t off = p->arena_off;
        p->arena_off += bytes;
        if (p->dry) return nullptr;
        if (p->arena_off > p->arena_bytes) { check("arena overflow"); return nullptr; }
        return p->arena + off;
    }
    double* alloc_stats() {
        const size_t bytes = ((size_t)p->B * h->cfg.resnet_groups * 2 * sizeof(double) + 255) & ~(size_t)255;
        const size_t off = p->stats_off;
        p->stats_off += bytes;
        if (p->dry) return (double*)(uintptr_t)(off + 256);  // non-null marker
        if (p->stats_off > p->stats_bytes) { check("stats arena overflow"); return nullptr; }
        return (double*)(p->stats + off);
    }
    Act new_act(int C, int L) { Act a; a.C = C; a.L = L; a.p = alloc((size_t)p->B * L * C * h->esz); return a; }
    void tap(const std::string& name, const Act& a) { p->taps.push_back({name, a.p, a.C, a.L}); }
    bool live() const { return !p->dry && !bad; }

    bool can_fuse_stats(int C) const {
        if (h->cfg.flags & ADF_FLAG_SEPARATE_GN_STATS) return false;
        const int G = h->cfg.resnet_groups;
        if (C % G) return false;
        const int gs = C / G;
        return (gs & (gs - 1)) == 0;
    }
    double* ensure_stats(Act& t) {
        if (!t.stats) {
            t.stats = alloc_stats();
            if (live()) check(launch_gn_stats(t.p, h->bf16, p->B, t.L, t.C, h->cfg.resnet_groups, t.stats, s));
        }
        return t.stats;
    }

    GemmArgs gemm_base(const Act& out, int lin, int mrows, const ConvW& w) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.nseg = 1; g.B = p->B; g.lin = lin; g.mrows = mrows; g.n = w.n; g.n_pad = w.n_pad;
        g.bias0 = w.bias; g.bias_mod = w.n > 0 ? w.n : 1;
        g.out = out.p; g.out_rows = out.L; g.out_c = out.C;
        return g;
    }
    static GemmSeg seg_of(const Act& x, const Act* skip, const float* ab, float scale1, int act, int taps, int stride, int off0,
                          int step, const ConvW& w) {
        GemmSeg sg;
        memset(&sg, 0, sizeof(sg));
        sg.src0 = x.p; sg.c0 = x.C;
        sg.src1 = skip ? skip->p : nullptr; sg.c1 = skip ? skip->C : 0;
        sg.ab = ab; sg.scale1 = scale1; sg.act = act;
        sg.taps = taps; sg.stride = stride; sg.off0 = off0; sg.step = step;
        sg.w = w.w; sg.wfrag = w.wfrag; sg.nchunk = w.nchunk;
        return sg;
    }
    void run_gemm(GemmArgs& g, Act& out, bool want_stats) {
        const bool ask = want_stats && can_fuse_stats(out.C);      // also for the phase-scattered transposed convs
        if (ask) {
            out.stats = alloc_stats();
            g.stats = out.stats; g.stats_groups = h->cfg.resnet_groups;
        }
        if (live()) {
            bool fused = false;
            check(launch_conv_gemm(g, h->gemm_dtype(), s, &fused));
            // the launcher may decline (tile shape / group size): fill the same buffer with the separate pass
            if (ask && !fused) check(launch_gn_stats(out.p, h->bf16, p->B, out.L, out.C, h->cfg.resnet_groups, out.stats, s));
        }
    }

    Act linear(const Act& x, const ConvW& w, const void* res, int gelu, bool want_stats) {
        // a 1x1 op has no halo: run it over the flattened [B*L] rows as one long sample
        Act out = new_act(w.n, x.L);
        const int rows = p->B * x.L;
        GemmArgs g = gemm_base(out, rows, rows, w);
        g.B = 1; g.out_rows = rows;
        g.seg[0] = seg_of(x, nullptr, nullptr, 1.f, 0, 1, 1, 0, 1, w);
        g.res = res; g.gelu = gelu;
        run_gemm(g, out, false);
        (void)want_stats;   // per-sample statistics come from the separate pass (ensure_stats) when needed
        return out;
    }

    Act resblock(const std::string& name, Act& x, Act* skip, const ResW& r, int nb) {
        const int B = p->B, G = h->cfg.resnet_groups;
        const float sscale = h->cfg.use_skip_scale ? 0.70710678118654752440f : 1.0f;
        const int ctot = x.C + (skip ? skip->C : 0);
        if (ctot != r.cin) check("resblock: channel mismatch");
        double* s0 = ensure_stats(x);
        double* s1 = skip ? ensure_stats(*skip) : nullptr;
        static int short_max = -1;       // ADF_SHORT_LEVEL: longest level that materialises silu(GN(x)) for flat GEMM tiles
        if (short_max < 0) short_max = (int)adf_tuning("ADF_SHORT_LEVEL", 32);
        const bool short_level = x.L <= short_max && (x.L & (x.L - 1)) == 0;
        float* ab1 = (float*)alloc((size_t)B * ctot * 2 * 4);
        GnFinalizeArgs f1;
        memset(&f1, 0, sizeof(f1));
        f1.stats0 = s0; f1.stats1 = s1; f1.c0 = x.C; f1.c1 = skip ? skip->C : 0; f1.L = x.L; f1.G = G; f1.B = B;
        f1.scale1 = sscale; f1.eps = 1e-5f; f1.gamma = r.g1w; f1.beta = r.g1b; f1.film = nullptr; f1.ab = ab1;
        // short levels in bf16 mode: the whole resblock in one launch (adf_resblock_small.h); ADF_RB_FUSED=0 keeps the separate launches
        // (2 = four workgroups per sample in two launches when the batch leaves CUs idle, adf_resblock_split.h; 1 = always the one-launch kernel)
        static int rb_fused = -1;
        if (rb_fused < 0) rb_fused = adf_route_switch("ADF_RB_FUSED", 2);
        if (rb_fused && h->bf16 && (x.L == 16 || x.L == 64) && r.cout == 256 && x.C == 256 && (!skip || skip->C == 256) && G == 8 &&
            r.c1.wfrag && r.c2.wfrag && (!r.has_res || r.cr.wfrag) && r.c1.n_pad == 256 && !(h->cfg.flags & ADF_FLAG_SEPARATE_GN_STATS)) {
            Act y = new_act(r.cout, x.L);
            RbFusedArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.x = (const bf16_t*)x.p; fa.skip = skip ? (const bf16_t*)skip->p : nullptr; fa.out = (bf16_t*)y.p;
            fa.gn1 = f1;
            fa.gamma2 = r.g2w; fa.beta2 = r.g2b;
            fa.film = film + r.film_off; fa.film_bstride = nb == 1 ? 0 : h->film_total;
            if (film2) { fa.film2 = film2 + r.film_off; fa.film2_bstride = film2_bstride; }
            fa.w1 = r.c1.wfrag; fa.w2 = r.c2.wfrag; fa.wr = r.has_res ? r.cr.wfrag : nullptr;
            fa.b1 = r.c1.bias; fa.b2 = r.c2.bias; fa.br = r.has_res ? r.cr.bias : nullptr;
            fa.skip_scale = sscale; fa.eps = 1e-5f;
            y.stats = alloc_stats(); fa.stats = y.stats;
            if (rb_fused >= 2 && B * 4 <= 256) {
                Act hact = new_act(r.cout, x.L);
                RbSplitArgs sa;
                sa.f = fa; sa.hact = (bf16_t*)hact.p;
                if (live()) check(launch_resblock_split(sa, B, x.L, ctot, s));
            } else if (live()) check(launch_resblock_small(fa, B, x.L, ctot, s));
            RbRec rec{name, GemmArgs{}, GemmArgs{}, r.cin, r.cout, x.L};
            rec.g1.nseg = 0;                               // marks a fused block for adf_bench_resblock (keeps the block numbering)
            p->rbs.push_back(rec);
            tap(name, y);
            return y;
        }
        Act h1 = new_act(r.cout, x.L);
        GemmArgs g1 = gemm_base(h1, x.L, x.L, r.c1);
        if (short_level) {
            // short levels: one launch normalises + activates the (concatenated) input; the GEMM then takes raw tiles
            Act a1 = new_act(ctot, x.L);
            if (live()) check(launch_gn_norm_apply(x.p, skip ? skip->p : nullptr, f1, 1, a1.p, h->bf16, s));
            g1.seg[0] = seg_of(a1, nullptr, nullptr, 1.f, 0, 3, 1, -1, 1, r.c1);
        } else {
            g1.seg[0] = seg_of(x, skip, ab1, sscale, 1, 3, 1, -1, 1, r.c1);
            g1.seg[0].gn = f1;           // launch_conv_gemm derives the table (in the DMA kernel) or launches gn_finalize
        }
        run_gemm(g1, h1, true);
        tap(name + ".h1", h1);        // the block's stored intermediate (not there when the whole block is one launch)
        double* sh = ensure_stats(h1);
        float* ab2 = (float*)alloc((size_t)B * r.cout * 2 * 4);
        GnFinalizeArgs f2;
        memset(&f2, 0, sizeof(f2));
        f2.stats0 = sh; f2.c0 = r.cout; f2.L = x.L; f2.G = G; f2.B = B; f2.scale1 = 1.f; f2.eps = 1e-5f;
        f2.gamma = r.g2w; f2.beta = r.g2b;
        f2.film = film + r.film_off; f2.film_bstride = nb == 1 ? 0 : h->film_total; f2.ab = ab2;
        if (film2) { f2.film2 = film2 + r.film_off; f2.film2_bstride = film2_bstride; }
        Act y = new_act(r.cout, x.L);
        GemmArgs g2 = gemm_base(y, x.L, x.L, r.c2);
        if (short_level) {
            Act a2 = new_act(r.cout, x.L);
            if (live()) check(launch_gn_norm_apply(h1.p, nullptr, f2, 1, a2.p, h->bf16, s));
            g2.seg[0] = seg_of(a2, nullptr, nullptr, 1.f, 0, 3, 1, -1, 1, r.c2);
        } else {
            g2.seg[0] = seg_of(h1, nullptr, ab2, 1.f, 1, 3, 1, -1, 1, r.c2);
            g2.seg[0].gn = f2;
        }
        if (r.has_res) {
            g2.nseg = 2;
            g2.seg[1] = seg_of(x, skip, nullptr, sscale, 0, 1, 1, 0, 1, r.cr);
            g2.bias1 = r.cr.bias;
        } else {
            if (skip) check("resblock: identity residual with a skip input");
            g2.res = x.p;
        }
        run_gemm(g2, y, true);
        p->rbs.push_back({name, g1, g2, r.cin, r.cout, x.L});
        tap(name, y);
        return y;
    }

    Act transformer(const std::string& name, Act& x, const TrW& t) {
        const long long rows = (long long)p->B * x.L;
        // short levels in bf16 mode: the whole block in one launch (adf_transformer.h); ADF_TR_FUSED=0 keeps the nine launches
        static int tr_fused = -1;
        if (tr_fused < 0) tr_fused = adf_route_switch("ADF_TR_FUSED", 2);
        if (tr_fused && h->bf16 && t.c == 256 && t.mid == 512 && h->cfg.attention_heads == 8 && (x.L == 16 || x.L == 64) &&
            x.C == 256 && t.qkv.nchunk == 4 && t.ff2.nchunk == 8 && t.qkv.wfrag) {
            Act x2 = new_act(t.c, x.L);
            TrFusedArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.x = (const bf16_t*)x.p; fa.out = (bf16_t*)x2.p;
            fa.ln_w = t.lnw; fa.ln_b = t.lnb; fa.g0 = t.g0; fa.g3 = t.g3;
            fa.wqkv = t.qkv.wfrag; fa.wproj = t.proj.wfrag; fa.wff1 = t.ff1.wfrag; fa.wff2 = t.ff2.wfrag;
            fa.npad_qkv = t.qkv.n_pad; fa.npad_proj = t.proj.n_pad; fa.npad_ff1 = t.ff1.n_pad; fa.npad_ff2 = t.ff2.n_pad;
            fa.eps = 1e-5f;
            if (h->cfg.resnet_groups == 8 && !(h->cfg.flags & ADF_FLAG_SEPARATE_GN_STATS)) { x2.stats = alloc_stats(); fa.stats = x2.stats; }
            if (live()) check(launch_transformer_small(fa, p->B, x.L, s));
            tap(name, x2);
            return x2;
        }
        // longer samples (256 tokens): two fused launches around the attention kernel (ADF_TR_FUSED=1 keeps these unfused)
        if (tr_fused >= 2 && h->bf16 && t.c == 256 && t.mid == 512 && h->cfg.attention_heads == 8 && x.L % 64 == 0 && x.L > 64 &&
            x.C == 256 && t.qkv.nchunk == 4 && t.ff2.nchunk == 8 && t.qkv.wfrag && h->cfg.resnet_groups == 8 &&
            !(h->cfg.flags & ADF_FLAG_SEPARATE_GN_STATS)) {
            Act qkv = new_act(3 * t.c, x.L), att = new_act(t.c, x.L), x2 = new_act(t.c, x.L);
            TrFusedArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.x = (const bf16_t*)x.p; fa.out = (bf16_t*)x2.p; fa.qkv_out = (bf16_t*)qkv.p; fa.att = (const bf16_t*)att.p;
            fa.ln_w = t.lnw; fa.ln_b = t.lnb; fa.g0 = t.g0; fa.g3 = t.g3;
            fa.wqkv = t.qkv.wfrag; fa.wproj = t.proj.wfrag; fa.wff1 = t.ff1.wfrag; fa.wff2 = t.ff2.wfrag;
            fa.npad_qkv = t.qkv.n_pad; fa.npad_proj = t.proj.n_pad; fa.npad_ff1 = t.ff1.n_pad; fa.npad_ff2 = t.ff2.n_pad;
            fa.eps = 1e-5f;
            x2.stats = alloc_stats(); fa.stats = x2.stats;
            if (live()) {
                check(launch_transformer_tiles(fa, (int)rows, x.L, 1, s));
                check(launch_attention(qkv.p, att.p, h->bf16, p->B, x.L, t.c, h->cfg.attention_heads, s));
                check(launch_transformer_tiles(fa, (int)rows, x.L, 2, s));
            }
            tap(name + ".qkv", qkv);
            tap(name + ".att", att);
            tap(name, x2);
            return x2;
        }
        // the nine-launch path: every stored tensor of the block is a recorded activation (the parity tests hold each launch to
        // the oracle on its own; the fused kernels above are then held to this path)
        Act xn = new_act(t.c, x.L);
        if (live()) check(launch_ln_rows(x.p, xn.p, h->bf16, rows, t.c, t.lnw, t.lnb, 1e-5f, s));
        tap(name + ".ln", xn);
        Act qkv = linear(xn, t.qkv, nullptr, 0, false);
        tap(name + ".qkv", qkv);
        Act att = new_act(t.c, x.L);
        if (live()) check(h->x3 ? launch_attention_x3(qkv.p, att.p, p->B, x.L, t.c, h->cfg.attention_heads, s)
                                : launch_attention(qkv.p, att.p, h->bf16, p->B, x.L, t.c, h->cfg.attention_heads, s));
        tap(name + ".att", att);
        Act x1 = linear(att, t.proj, x.p, 0, false);
        tap(name + ".x1", x1);
        Act n1 = new_act(t.c, x.L);
        if (live()) check(launch_ln_rows(x1.p, n1.p, h->bf16, rows, t.c, t.g0, nullptr, 1e-5f, s));
        tap(name + ".n1", n1);
        Act f1 = linear(n1, t.ff1, nullptr, 1, false);
        tap(name + ".f1", f1);
        Act n2 = new_act(t.mid, x.L);
        if (live()) check(launch_ln_rows(f1.p, n2.p, h->bf16, rows, t.mid, t.g3, nullptr, 1e-5f, s));
        tap(name + ".n2", n2);
        Act x2 = linear(n2, t.ff2, x1.p, 0, true);
        tap(name, x2);
        return x2;
    }
};

struct FwdIO {
    const float* x = nullptr; float* out = nullptr;
    const float* t = nullptr; int t_stride = 0; int nb = 0;
    const float* coef = nullptr; int coef_bstride = 0; int mode = 0; const float* x_noisy = nullptr;
    const float* film2 = nullptr; int film2_bstride = 0;   // class part of the FiLM projections (rows of adf_handle::cond_film)
    const float* film_pre = nullptr;                       // this evaluation's row of Plan::film_all: sigma embedding + FiLM already computed
    const float* temb_pre = nullptr;                       // this evaluation's row of Plan::temb_all (class-conditional ADM net: the FiLM rows are per sample)
    bool null_cond = false;                                // class-conditional ADM net: every sample takes the null class embedding (guidance branch)
};

int wn_forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s);
int adm_forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s);

int forward(adf_handle* h, Plan* p, const FwdIO& io, hipStream_t s);
int wn_pack_weights(adf_handle* h, hipStream_t s);
int wn_build_weights(adf_handle* h);
int adm_build_weights(adf_handle* h);
int get_plan(adf_handle* h, int B, int L, hipStream_t s, Plan** out);
int cond_rows(adf_handle* h, int B, bool null_branch, FwdIO& io);
int ensure_cfg_buffers(adf_handle* h, Plan* p);
int denoise_io(adf_handle* h, Plan* p, FwdIO io, float* out, hipStream_t s);
int denoise_scalar(adf_handle* h, Plan* p, const float* x, float sigma, float sigma_data, float* out, hipStream_t s);

struct SamplerCtx {
    adf_handle* h; Plan* p; const adf_sampler_desc* d; const float* sig; int nsig; hipStream_t s; long long n;
    int nfe = 0;
    bool count_only = false;
    std::vector<float>* collect = nullptr;     // count_only pass: the sigma of every evaluation, in order
    bool precomputed = false;                  // real pass: evaluation k reads row k of Plan::coef_all / film_all
    int den(const float* x, float sigma, float* out) {
        const int k = nfe++;
        if (count_only) { if (collect) collect->push_back(sigma); return 0; }
        if (precomputed) {
            FwdIO io;
            io.x = x; io.t = p->coef_all + (size_t)k * 4 + 1; io.t_stride = 4; io.nb = 1;
            io.coef = p->coef_all + (size_t)k * 4; io.coef_bstride = 0; io.x_noisy = x;
            if (h->adm && h->cdim > 0) io.temb_pre = p->temb_all + (size_t)k * 4 * h->cfg.channels;
            else io.film_pre = p->film_all + (size_t)k * h->film_total;
            return denoise_io(h, p, io, out, s);
        }
        return denoise_scalar(h, p, x, sigma, d->sigma_data, out, s);
    }
    int ck(const char* e) { if (e) { h->err = e; return 1; } return 0; }
    // DPMSampler.model_fn (sampler_edm.py:692-708): the denoised estimate, or with eps_pred the noise prediction (x - D) / sigma
    int model(const float* x, float sigma, float* out) {
        if (den(x, sigma, out)) return 1;
        if (d->eps_pred && !count_only) return ck(launch_eps(out, x, sigma, n, s));
        return 0;
    }
};
int run_sampler(SamplerCtx& c, float** result);
int adpmpp2s_draws(const float* sig, int nsig, int N);      // randn_like draws ADPMPP2SSampler consumes on this schedule

}  // namespace adf_api
