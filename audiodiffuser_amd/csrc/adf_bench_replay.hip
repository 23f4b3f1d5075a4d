// adf_bench_* of include/audiodiffuser_amd.h: HIP-event replays of single launches of the last network pass (bench.py's roofline object).
#include "adf_api_internal.h"

using namespace adf;
using namespace adf_api;

extern "C" {


// conv = 0: both launches of the block; 1 / 2: only conv1 / conv2 (the other's outputs are zero)
static int bench_resblock_impl(adf_handle* h, int B, int L, int level, int conv, int iters, float* ms1, float* ms2, double* bytes1,
                               double* bytes2, double* flops1, double* flops2, int* ncopies, void* stream) {
    ADF_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Plan* p;
    if (get_plan(h, B, L, s, &p)) return 1;
    if (p->rbs.empty()) return fail(h, "no resblock recorded; run a forward first");
    if (level < 0 || level >= (int)p->rbs.size()) return fail(h, "resblock index out of range");
    if (iters < 1) return fail(h, "bench_resblock: iters must be >= 1");
    const RbRec& r = p->rbs[level];
    if (r.g1.nseg == 0) {                                  // fused short-level block: no separate conv launches to replay
        *ms1 = *ms2 = 0.f; *bytes1 = *bytes2 = *flops1 = *flops2 = 0.0;
        return 0;
    }
    // The replay must cost what the launch costs inside a network pass: (1) the launch is the real one -- GroupNorm table
    // derived from the input statistics (in the kernel, or by the gn_finalize launch the route needs), statistics of the
    // output reduced in the epilogue (into a scratch buffer); (2) its operands are NOT served by the 256 MiB Infinity Cache:
    // every iteration works on another copy of (inputs, residual, output), >= 3 copies and >= 320 MiB in rotation.
    const size_t esz = (size_t)h->esz;
    struct Op { const void** ptr; size_t bytes; };
    auto operands = [&](GemmArgs& g, std::vector<Op>& ops) {
        for (int k = 0; k < g.nseg; ++k) {
            if (g.seg[k].src0) ops.push_back({&g.seg[k].src0, (size_t)g.B * g.lin * g.seg[k].c0 * esz});
            if (g.seg[k].src1) ops.push_back({&g.seg[k].src1, (size_t)g.B * g.lin * g.seg[k].c1 * esz});
        }
        if (g.res) ops.push_back({&g.res, (size_t)g.B * g.out_rows * g.out_c * esz});
        ops.push_back({(const void**)&g.out, (size_t)g.B * g.out_rows * g.out_c * esz});
    };
    auto set_bytes = [&](const GemmArgs& gc) {
        GemmArgs g = gc;
        std::vector<Op> ops;
        operands(g, ops);
        size_t t = 0;
        for (const Op& o : ops) t += (o.bytes + 255) & ~(size_t)255;
        return t;
    };
    const size_t rot_min = (size_t)320 << 20;
    auto copies = [&](const GemmArgs& g) { const size_t sb = set_bytes(g); size_t n = (rot_min + sb - 1) / sb; return n < 3 ? (size_t)3 : n; };
    const size_t stats_bytes = ((size_t)B * h->cfg.resnet_groups * 2 * sizeof(double) + 255) & ~(size_t)255;
    const size_t need = std::max(set_bytes(r.g1) * copies(r.g1), set_bytes(r.g2) * copies(r.g2)) + stats_bytes;
    if (p->bench_cap < need) {
        if (hipStreamSynchronize(s) != hipSuccess) return fail(h, "bench_resblock: stream sync failed");
        dfree(h, p->bench_buf, p->bench_cap, p);
        p->bench_buf = (char*)dalloc(h, need, p);
        p->bench_cap = p->bench_buf ? need : 0;
        if (!p->bench_buf) return fail(h, "bench_resblock: device allocation failed for the rotating operand copies");
    }
    double* scratch_stats = (double*)p->bench_buf;
    auto run = [&](const GemmArgs& g0, float* ms) -> int {
        const size_t R = copies(g0);
        if (ncopies) *ncopies = (int)R;
        std::vector<GemmArgs> sets(R, g0);
        char* cur = p->bench_buf + stats_bytes;
        for (size_t k = 0; k < R; ++k) {
            std::vector<Op> ops;
            operands(sets[k], ops);
            for (Op& o : ops) {
                const bool is_out = (const void**)&sets[k].out == o.ptr;
                if (!is_out && hipMemcpyAsync(cur, *o.ptr, o.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(h, "bench_resblock: operand copy failed");
                *o.ptr = cur;
                cur += (o.bytes + 255) & ~(size_t)255;
            }
            if (sets[k].stats) sets[k].stats = scratch_stats;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(h, "bench_resblock: hipEventCreate failed");
        int rc = 0;
        for (size_t k = 0; k < R && !rc; ++k)
            if (const char* e = launch_conv_gemm(sets[k], h->gemm_dtype(), s)) rc = fail(h, e);          // warm-up: code, attributes, TLBs
        if (!rc && hipEventRecord(e0, s) != hipSuccess) rc = fail(h, "bench_resblock: hipEventRecord failed");
        for (int i = 0; i < iters && !rc; ++i)
            if (const char* e = launch_conv_gemm(sets[(size_t)i % R], h->gemm_dtype(), s)) rc = fail(h, e);
        if (!rc && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(h, "bench_resblock: event record / sync failed");
        float t = 0.f;
        if (!rc && hipEventElapsedTime(&t, e0, e1) != hipSuccess) rc = fail(h, "bench_resblock: hipEventElapsedTime failed");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms = t / (float)iters;
        return rc;
    };
    *ms1 = *ms2 = 0.f;
    if ((conv != 2 && run(r.g1, ms1)) || (conv != 1 && run(r.g2, ms2))) return 1;
    const double es = h->esz, BL = (double)B * r.L, ci = r.cin, co = r.cout;
    // x read for conv1; x read again for the residual; h1 written and re-read; y written; weights once
    *bytes1 = BL * es * (ci + co) + es * 3.0 * ci * co;
    *bytes2 = BL * es * (co + ci + co) + es * (3.0 * co * co + (ci != co ? ci * co : 0.0));
    *flops1 = 2.0 * BL * 3.0 * ci * co;
    *flops2 = 2.0 * BL * (3.0 * co * co + (ci != co ? ci * co : 0.0));
    return 0;
}

int adf_bench_resblock(adf_handle* h, int B, int L, int level, int iters, float* ms1, float* ms2, double* bytes1, double* bytes2,
                       double* flops1, double* flops2, void* stream) {
    return bench_resblock_impl(h, B, L, level, 0, iters, ms1, ms2, bytes1, bytes2, flops1, flops2, nullptr, stream);
}

int adf_bench_layer(adf_handle* h, int B, int L, int level, int conv, int iters, float* ms, double* algo_bytes, double* flops,
                    int* copies, void* stream) {
    if (conv != 1 && conv != 2) return fail(h, "adf_bench_layer: conv must be 1 or 2");
    float m1 = 0.f, m2 = 0.f;
    double b1 = 0, b2 = 0, f1 = 0, f2 = 0;
    if (bench_resblock_impl(h, B, L, level, conv, iters, &m1, &m2, &b1, &b2, &f1, &f2, copies, stream)) return 1;
    *ms = conv == 1 ? m1 : m2; *algo_bytes = conv == 1 ? b1 : b2; *flops = conv == 1 ? f1 : f2;
    return 0;
}

int adf_bench_wavenet_layer(adf_handle* h, int B, int T, int layer, int iters, float* ms, double* algo_bytes, double* flops, void* stream) {
    ADF_ON_DEVICE(h);
    if (!h->wn) return fail(h, "adf_bench_wavenet_layer: not a WaveNetNoise handle");
    hipStream_t s = (hipStream_t)stream;
    Plan* p;
    if (get_plan(h, B, T, s, &p)) return 1;
    if (layer < 0 || layer >= (int)p->wn_layers.size()) return fail(h, "adf_bench_wavenet_layer: layer out of range (run a forward first)");
    if (iters < 1) return fail(h, "adf_bench_wavenet_layer: iters must be positive");
    const WnLayerArgs& a = p->wn_layers[layer];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(h, "hipEventCreate failed");
    const char* err = nullptr;
    for (int i = 0; i < 2 && !err; ++i) err = launch_wn_layer(p->wn_io, a, s);
    if (!err && hipEventRecord(e0, s) != hipSuccess) err = "hipEventRecord failed";
    for (int i = 0; i < iters && !err; ++i) err = launch_wn_layer(p->wn_io, a, s);
    if (!err && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) err = "event sync failed";
    float t = 0.f;
    if (!err && hipEventElapsedTime(&t, e0, e1) != hipSuccess) err = "hipEventElapsedTime failed";
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err) return fail(h, err);
    const double C = h->wn->cfg.residual_channels, pos = (double)B * T, esz = h->esz;
    *ms = t / (float)iters;
    // per position: read y, write y_next (not for the last layer), skip read-modify-write in fp32 (first layer: write only);
    // per launch: both weight matrices once.  Flops: the K = 3C and K = C GEMMs onto 2C columns each.
    *algo_bytes = pos * C * (esz + (a.y_next ? esz : 0.0) + (a.first ? 4.0 : 8.0)) + 8.0 * C * C * esz;
    *flops = pos * 2.0 * 8.0 * C * C;
    return 0;
}


}  // extern "C"
