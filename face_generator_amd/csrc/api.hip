// C-ABI entry points of libfacegen_hip.so (declared in include/facegen_hip.h): context, memory,
// optimizer / criterion / RNG and the module-level (nn.Module protocol) operator entries.
#include "fg_internal.h"
#include "conv_ops.h"
#include "../../include/facegen_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static char g_err[512] = "";
bool fg_attr_first(fg_ctx* ctx, const void* key) {
    for (const void* k : ctx->attr_keys) if (k == key) return false;
    ctx->attr_keys.push_back(key);
    return true;
}
bool g_fg_launch_log = false;
void fg_log_launch_line(const char* name, dim3 grid, dim3 block, size_t lds) {
    fprintf(stderr, "fg-launch %s grid=%u,%u,%u block=%u lds=%zu\n", name, grid.x, grid.y, grid.z, block.x, lds);
}
bool g_fg_dry = false;      // fg_internal.h: a planning-only context exists in this process

int fg_set_err(fg_ctx* c, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_err, 512, fmt, ap);
    va_end(ap);
    return code;
}

const char* fg_intern(fg_ctx* ctx, const char* s) {
    if (!ctx || !ctx->prof) return "";
    for (auto* n : ctx->names) if (*n == s) return n->c_str();
    ctx->names.push_back(new std::string(s));
    return ctx->names.back()->c_str();
}
FgProfScope::FgProfScope(fg_ctx* c, const char* name, double alg, double exec, double bytes) : ctx(c), idx(-1) {
    if (!c || !c->prof) return;
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
        if (!c->prof_pool.empty()) { e[i] = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else if (hipEventCreate(&e[i]) != hipSuccess) return;
    }
    FgProfRec r{name, e[0], e[1], alg, exec, bytes};
    (void)hipEventRecord(r.e0, c->stream);
    c->prof_recs.push_back(r);
    idx = (int)c->prof_recs.size() - 1;
}
FgProfScope::~FgProfScope() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof_recs[idx].e1, ctx->stream);
}

#define NEED(ctx, cond, msg) \
    do { if (!(cond)) return fg_set_err((ctx), FG_ERR_INVALID, "%s: %s", __func__, msg); } while (0)

extern "C" {
#pragma GCC visibility push(default)

const char* fg_version(void) { return "facegen_hip 0.1 (gfx950)"; }

static int g_real_ctx = 0;       // live contexts bound to a device / planning-only ones: never both in one process
static int g_dry_ctx = 0;

int fg_ctx_create(int device, fg_ctx** out) {
    if (!out) return fg_set_err(nullptr, FG_ERR_INVALID, "fg_ctx_create: null out");
    if (const char* m = getenv("FG_LAUNCH_LOG")) g_fg_launch_log = atoi(m) != 0;
    fg_plan_env_init();
    if (device == FG_DEVICE_NONE) {
        if (g_real_ctx) return fg_set_err(nullptr, FG_ERR_INVALID, "fg_ctx_create: a planning-only context cannot join a process that holds a device context");
        fg_ctx* c = new fg_ctx();
        c->device = FG_DEVICE_NONE; c->stream = nullptr; c->err[0] = 0; c->sm_count = 256;     // MI355X: 256 CUs
        g_fg_dry = true; ++g_dry_ctx;
        *out = c;
        return FG_OK;
    }
    if (g_dry_ctx) return fg_set_err(nullptr, FG_ERR_INVALID, "fg_ctx_create: this process holds a planning-only context (FG_DEVICE_NONE)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fg_set_err(nullptr, FG_ERR_HIP, "fg_ctx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fg_set_err(nullptr, FG_ERR_INVALID, "fg_ctx_create: device %d of %d", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fg_set_err(nullptr, FG_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fg_set_err(nullptr, FG_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fg_set_err(nullptr, FG_ERR_UNSUPPORTED, "fg_ctx_create: built for gfx950, device is %s", prop.gcnArchName);
    fg_ctx* c = new fg_ctx();
    c->device = device; c->stream = nullptr; c->err[0] = 0; c->sm_count = prop.multiProcessorCount;
    if (const char* m = getenv("FG_MATH")) c->math = atoi(m) == 6 ? 6 : 0;   // default arithmetic, see fg_set_math
    if (const char* m = getenv("FG_FUSE_PRELU")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_PRELU;
    if (const char* m = getenv("FG_THIN_SLAB")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_THIN_SLAB;
    if (const char* m = getenv("FG_DEFER_WFINISH")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_WFINISH_BATCH;
    if (const char* m = getenv("FG_THIN_BIAS")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_THIN_BIAS;
    if (const char* m = getenv("FG_WINO")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_WINOGRAD;
    if (const char* m = getenv("FG_WINO_UP")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_WINOGRAD_UP;
    if (const char* m = getenv("FG_WINO_5X5")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_WINOGRAD_5X5;
    if (const char* m = getenv("FG_WINO_WGRAD")) if (atoi(m) == 0) c->fusion &= ~FG_FUSE_WINOGRAD_WGRAD;
    c->fusion &= ~FG_FUSE_ADAM_PACK;        // measured slower than the two launches: opt-in
    if (const char* m = getenv("FG_ADAM_PACK")) if (atoi(m) != 0) c->fusion |= FG_FUSE_ADAM_PACK;
    ++g_real_ctx;
    *out = c;
    return FG_OK;
}
int fg_ctx_destroy(fg_ctx* ctx) {
    if (!ctx) return FG_OK;
    if (ctx->device == FG_DEVICE_NONE) { if (--g_dry_ctx == 0) g_fg_dry = false; }
    else --g_real_ctx;
    if (ctx->clk_stream) { (void)hipStreamSynchronize(ctx->clk_stream); (void)hipStreamDestroy(ctx->clk_stream); (void)hipFree(ctx->clk_dev); }
    delete ctx;
    return FG_OK;
}
int fg_set_math(fg_ctx* ctx, int mode) {
    if (!ctx) return FG_ERR_INVALID;
    if (mode != 0 && mode != 6) return fg_set_err(ctx, FG_ERR_INVALID, "fg_set_math: mode %d (0 = fp32 MFMA, 6 = bf16x6)", mode);
    ctx->math = mode;
    return FG_OK;
}
int fg_get_math(fg_ctx* ctx) { return ctx ? ctx->math : -1; }
int fg_set_fusion(fg_ctx* ctx, int flags) {
    if (!ctx) return FG_ERR_INVALID;
    if (flags & ~FG_FUSE_ALL) return fg_set_err(ctx, FG_ERR_INVALID, "fg_set_fusion: unknown bits in %d", flags);
    ctx->fusion = flags;
    return FG_OK;
}
int fg_get_fusion(fg_ctx* ctx) { return ctx ? ctx->fusion : -1; }
int fg_scale_bilinear(fg_ctx* ctx, const float* src, float* dst, int n, int c, int hs, int ws, int hd, int wd, int layout) {
    NEED(ctx, ctx, "null ctx");
    if (!src || !dst || n < 0 || c < 1 || hs < 1 || ws < 1 || hd < 1 || wd < 1 || (layout != 0 && layout != 1))
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_scale_bilinear: bad argument (n=%d c=%d %dx%d -> %dx%d layout=%d)", n, c, hs, ws, hd, wd, layout);
    return fg_launch_scale_bilinear(ctx, src, dst, n, c, hs, ws, hd, wd, layout, nullptr, nullptr);
}
int fg_c2f_coarse_diff(fg_ctx* ctx, const float* fine, float* coarse, float* diff, float* tmp, int n, int c, int s, int cs, int layout) {
    NEED(ctx, ctx, "null ctx");
    if (!fine || !coarse || !diff || !tmp || n < 0 || c < 1 || s < 1 || cs < 1 || (layout != 0 && layout != 1))
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_c2f_coarse_diff: bad argument (n=%d c=%d s=%d cs=%d layout=%d)", n, c, s, cs, layout);
    int rc = fg_launch_scale_bilinear(ctx, fine, tmp, n, c, s, s, cs, cs, layout, nullptr, nullptr);       // dataset_c2f.lua:54
    if (rc) return rc;
    return fg_launch_scale_bilinear(ctx, tmp, coarse, n, c, cs, cs, s, s, layout, fine, diff);              // :55 and :59-61
}
int fg_test_set_wino_wgrad_thresholds(fg_ctx* ctx, long long min_chunks, long long min_blocks) {
    if (!ctx) return FG_ERR_INVALID;
    fg_plan_set_wino_wgrad_thresholds(min_chunks, min_blocks);
    return FG_OK;
}

// one wave: sleeps in 4096-cycle naps until `ticks` of the 100 MHz counter have passed (bounded by max_naps)
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks, long long max_naps) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r = r0;
    for (long long i = 0; i < max_naps && r - r0 < ticks; ++i) {
        __builtin_amdgcn_s_sleep(64);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r - r0; }
}
int fg_prof_clock_start(fg_ctx* ctx, double ms) {
    NEED(ctx, ctx && ms > 0 && ms < 60000, "bad argument");
    if (g_fg_dry) return FG_OK;
    if (!ctx->clk_stream) {
        FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->clk_stream, hipStreamNonBlocking));
        FG_HIP(ctx, hipMalloc((void**)&ctx->clk_dev, 16));
    }
    FG_HIP(ctx, hipMemsetAsync(ctx->clk_dev, 0, 16, ctx->clk_stream));
    const unsigned long long ticks = (unsigned long long)(ms * 1e5);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ctx->clk_stream, ctx->clk_dev, ticks, (long long)(ms * 2.4e6 / 4096.0) + 16);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_prof_clock_read(fg_ctx* ctx, double* ghz, double* covered_ms) {
    NEED(ctx, ctx && ghz, "bad argument");
    *ghz = 0.0;
    if (covered_ms) *covered_ms = 0.0;
    if (g_fg_dry) return FG_OK;
    NEED(ctx, ctx->clk_stream, "fg_prof_clock_read before fg_prof_clock_start");
    unsigned long long h[2] = {0, 0};
    FG_HIP(ctx, hipStreamSynchronize(ctx->clk_stream));
    FG_HIP(ctx, hipMemcpy(h, ctx->clk_dev, 16, hipMemcpyDeviceToHost));
    if (h[1]) *ghz = (double)h[0] / (double)h[1] * 0.1;
    if (covered_ms) *covered_ms = (double)h[1] * 1e-5;
    return FG_OK;
}
int fg_prof_enable(fg_ctx* ctx, int on) { NEED(ctx, ctx, "null ctx"); ctx->prof = on != 0; return FG_OK; }
// Synchronises, then writes one line per kernel label: "name calls total_ms alg_flops exec_flops bytes\n".
int fg_prof_report(fg_ctx* ctx, char* buf, size_t len, int reset) {
    NEED(ctx, ctx && buf && len > 0, "bad argument");
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct Agg { const char* name; long calls; double ms, alg, exec, bytes; };
    std::vector<Agg> aggs;
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        Agg* a = nullptr;
        for (auto& x : aggs) if (strcmp(x.name, r.name) == 0) { a = &x; break; }
        if (!a) { aggs.push_back(Agg{r.name, 0, 0, 0, 0, 0}); a = &aggs.back(); }
        a->calls++; a->ms += ms; a->alg += r.alg_flops; a->exec += r.exec_flops; a->bytes += r.bytes;
    }
    size_t off = 0;
    buf[0] = 0;
    for (auto& a : aggs) {
        int n = snprintf(buf + off, len - off, "%s %ld %.6f %.6e %.6e %.6e\n", a.name, a.calls, a.ms, a.alg, a.exec, a.bytes);
        if (n < 0 || (size_t)n >= len - off) break;
        off += n;
    }
    if (reset) {
        for (auto& r : ctx->prof_recs) { ctx->prof_pool.push_back(r.e0); ctx->prof_pool.push_back(r.e1); }
        ctx->prof_recs.clear();
    }
    return FG_OK;
}
int fg_ctx_set_stream(fg_ctx* ctx, void* s) { NEED(ctx, ctx, "null ctx"); ctx->stream = (hipStream_t)s; return FG_OK; }
const char* fg_last_error(const fg_ctx* ctx) { return ctx ? ctx->err : g_err; }
int fg_stream_sync(fg_ctx* ctx) { NEED(ctx, ctx, "null ctx"); FG_HIP(ctx, hipStreamSynchronize(ctx->stream)); return FG_OK; }
int fg_malloc(fg_ctx* ctx, size_t bytes, void** out) {
    NEED(ctx, ctx && out, "null argument");
    if (fg_dev_alloc(out, bytes) != hipSuccess) return fg_set_err(ctx, FG_ERR_NOMEM, "fg_malloc(%zu)", bytes);
    return FG_OK;
}
int fg_free(fg_ctx* ctx, void* p) { NEED(ctx, ctx, "null ctx"); if (g_fg_dry) { free(p); return FG_OK; } FG_HIP(ctx, hipFree(p)); return FG_OK; }
int fg_h2d(fg_ctx* ctx, void* d, const void* s, size_t n) {
    NEED(ctx, ctx && d && s, "null argument");
    if (g_fg_dry) { memmove(d, s, n); return FG_OK; }
    FG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, ctx->stream));
    return FG_OK;
}
int fg_d2h(fg_ctx* ctx, void* d, const void* s, size_t n) {
    NEED(ctx, ctx && d && s, "null argument");
    if (g_fg_dry) { memmove(d, s, n); return FG_OK; }
    FG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FG_OK;
}
int fg_d2d(fg_ctx* ctx, void* d, const void* s, size_t n) {
    NEED(ctx, ctx && d && s, "null argument");
    if (g_fg_dry) { memmove(d, s, n); return FG_OK; }
    FG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, ctx->stream));
    return FG_OK;
}
int fg_fill(fg_ctx* ctx, float* p, float v, long long n) { NEED(ctx, ctx && p && n >= 0, "bad argument"); return fg_launch_fill(ctx, p, v, n); }
int fg_axpby(fg_ctx* ctx, float a, const float* x, float b, float* y, long long n) {
    NEED(ctx, ctx && x && y && n >= 0, "bad argument");
    return fg_launch_axpby(ctx, a, x, b, y, n);
}
int fg_nchw_to_nhwc(fg_ctx* ctx, const float* s, float* d, int n, int c, int h, int w) {
    NEED(ctx, ctx && s && d && n >= 0 && c > 0 && h > 0 && w > 0, "bad argument");
    return fg_launch_nchw_to_nhwc(ctx, s, d, n, c, h, w);
}
int fg_nhwc_to_nchw(fg_ctx* ctx, const float* s, float* d, int n, int c, int h, int w) {
    NEED(ctx, ctx && s && d && n >= 0 && c > 0 && h > 0 && w > 0, "bad argument");
    return fg_launch_nhwc_to_nchw(ctx, s, d, n, c, h, w);
}
int fg_rng_uniform(fg_ctx* ctx, uint64_t seed, uint64_t off, float* o, long long n, float lo, float hi) {
    NEED(ctx, ctx && o && n >= 0, "bad argument");
    return fg_launch_rng_uniform(ctx, seed, off, o, n, lo, hi);
}
int fg_rng_bernoulli(fg_ctx* ctx, uint64_t seed, uint64_t off, float* o, long long n, float keep) {
    NEED(ctx, ctx && o && n >= 0 && keep >= 0.f && keep <= 1.f, "bad argument");
    return fg_launch_rng_bernoulli(ctx, seed, off, o, n, keep);
}
int fg_rng_normal(fg_ctx* ctx, uint64_t seed, uint64_t off, float* o, long long n, float mean, float std) {
    NEED(ctx, ctx && o && n >= 0, "bad argument");
    return fg_launch_rng_normal(ctx, seed, off, o, n, mean, std);
}

int fg_bce_forward_backward(fg_ctx* ctx, const float* prob, const float* target, int n, float* loss, float* grad, int* conf) {
    NEED(ctx, ctx && prob && target && n > 0, "bad argument");
    return fg_launch_bce(ctx, prob, target, loss, grad, conf, n);
}

int fg_adam_fused(fg_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float gscale, float l1_mul,
                  float l2, float clamp, double lr, double beta1, double beta2, double eps, int t, float* g_out) {
    NEED(ctx, ctx && p && g && m && v && n >= 0 && t >= 1, "bad argument");
    AdamArgs a; a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.gscale = gscale; a.l1 = 0.f; a.l1_mul = l1_mul; a.l2 = l2;
    a.clamp = clamp; a.lr_d = lr; a.beta1_d = beta1; a.beta2_d = beta2; a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.eps = (float)eps; a.t = t; a.gout = g_out;
    return fg_launch_adam(ctx, a);
}
int fg_sgd_fused(fg_ctx* ctx, float* p, const float* g, float* mom, long long n, float gscale, float l1_mul, float l2,
                 float clamp, double lr, double momentum, double dampening, double wd, int nesterov, int first) {
    NEED(ctx, ctx && p && g && n >= 0 && (momentum == 0.0 || mom), "bad argument");
    return fg_launch_sgd(ctx, p, g, mom, n, gscale, l1_mul, l2, clamp, (float)lr, (float)momentum, (float)(1.0 - dampening),
                         (float)wd, nesterov, first);
}
int fg_adagrad_fused(fg_ctx* ctx, float* p, const float* g, float* var, long long n, float gscale, float l1_mul, float l2,
                     float clamp, double clr) {
    NEED(ctx, ctx && p && g && var && n >= 0, "bad argument");
    return fg_launch_adagrad(ctx, p, g, var, n, gscale, l1_mul, l2, clamp, (float)clr);
}
int fg_norms(fg_ctx* ctx, const float* p, long long n, float* out2, float* scratch) {
    NEED(ctx, ctx && p && out2 && scratch && n >= 0, "bad argument");
    return fg_launch_norms(ctx, p, n, out2, scratch);
}

// ---------------------------------------------------------------- module-level conv / linear
static bool thin_in(int cin, int cout) { return cin <= 4 && cout % 64 == 0; }
static bool thin_out(int cin, int cout) { return cout <= 4 && cin % 64 == 0; }
static ConvGeom mk_geom(int b, int h, int w, int cin, int cout, int k, int pad, int up, int fusion = 0) {
    ConvGeom g; memset(&g, 0, sizeof(g));
    g.B = b; g.H = h; g.W = w; g.Cin = cin; g.Cout = cout; g.k = k; g.pad = pad; g.fold = up;
    fg_geom_set_wino(g, fusion);  // 3x3 layers: Winograd F(2x2, 3x3) forward / data gradient (fg_set_fusion bit FG_FUSE_WINOGRAD)
    return g;
}
static inline long long a64(long long v) { return (v + 63) / 64 * 64; }

size_t fg_conv2d_workspace_bytes(int batch, int h, int w, int cin, int cout, int k, int up) {
    if (thin_in(cin, cout) || thin_out(cin, cout)) {
        const long long na = (long long)k * k * (cin <= 4 ? cin : cout), cw = cin <= 4 ? cout : cin;
        return (size_t)(a64((long long)cin * cout * k * k) + (FG_THIN_WGRAD_BLOCKS + 1) * na * cw + 258LL * (cout > 64 ? cout : 64) + 256) * 4;
    }
    // (no context here: the bound covers both settings of FG_FUSE_WINOGRAD)
    size_t need = 0;
    for (int wn = 0; wn < 2; ++wn) {
        ConvGeom g = mk_geom(batch, h, w, cin, cout, k, (k - 1) / 2, up, wn ? FG_FUSE_ALL : 0);
        if (wn && !g.wino) break;
        long long pf = fg_geom_pack_floats(g, 0), pb = fg_geom_pack_floats(g, 1);
        const size_t n = (size_t)(a64(pf > pb ? pf : pb) + fg_conv_scratch_floats(g) + 64) * sizeof(float);
        if (n > need) need = n;
    }
    return need;
}
static int conv_check(fg_ctx* ctx, int cin, int cout, int k, int pad, int up) {
    if (k % 2 != 1 || pad != (k - 1) / 2) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "conv2d: only odd-k 'same' stride-1");
    if (thin_in(cin, cout) || thin_out(cin, cout)) {
        if (up) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "conv2d: upsample fold on a thin conv");
        return FG_OK;
    }
    // (any channel count: a ragged one -- nInputPlane / nOutputPlane % 4 != 0 -- takes a zero-padded copy of the operand, conv_ops.hip)
    if (up) {
        int T, rmin;
        fg_fold_window(k, pad, &T, &rmin);
        if (4 * T * T > FG_MAX_GROUPS) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "conv2d: fold groups");
    } else if (k * k > FG_MAX_GROUPS) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "conv2d: k too large");
    return FG_OK;
}

int fg_conv2d_forward(fg_ctx* ctx, const float* x, const float* wt, const float* bias, float* y, int batch, int h,
                      int w, int cin, int cout, int k, int pad, int up, void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && x && wt && y && wsv, "null argument");
    int rc = conv_check(ctx, cin, cout, k, pad, up);
    if (rc) return rc;
    if (ws_bytes < fg_conv2d_workspace_bytes(batch, h, w, cin, cout, k, up)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv2d: workspace");
    float* ws = (float*)wsv;
    if (thin_in(cin, cout)) {
        if ((rc = fg_launch_thin_pack(ctx, wt, ws, cout, cin, k, 0))) return rc;
        return fg_launch_thin_in_conv(ctx, x, ws, bias, y, batch, h, w, cin, cout, k, 0);
    }
    if (thin_out(cin, cout)) {
        if ((rc = fg_launch_thin_pack(ctx, wt, ws, cout, cin, k, 1))) return rc;
        return fg_launch_thin_out_conv(ctx, x, ws, bias, y, batch, h, w, cin, cout, k, 0, 0);
    }
    ConvGeom g = mk_geom(batch, h, w, cin, cout, k, pad, up, ctx->fusion);
    const long long pf = a64(fg_geom_pack_floats(g, 0));
    if ((rc = fg_conv_pack(ctx, g, wt, ws, nullptr))) return rc;
    return fg_conv_forward_run(ctx, g, x, ws, bias, y, ws + pf, (long long)(ws_bytes / 4) - pf);
}
int fg_conv2d_backward_data(fg_ctx* ctx, const float* gy, const float* wt, float* gx, int batch, int h, int w, int cin,
                            int cout, int k, int pad, int up, void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && gy && wt && gx && wsv, "null argument");
    int rc = conv_check(ctx, cin, cout, k, pad, up);
    if (rc) return rc;
    if (ws_bytes < fg_conv2d_workspace_bytes(batch, h, w, cin, cout, k, up)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv2d: workspace");
    float* ws = (float*)wsv;
    if (thin_in(cin, cout)) {  // dX (thin) from dY (wide)
        if ((rc = fg_launch_thin_pack(ctx, wt, ws, cout, cin, k, 0))) return rc;
        return fg_launch_thin_out_conv(ctx, gy, ws, nullptr, gx, batch, h, w, cout, cin, k, 1, 0);
    }
    if (thin_out(cin, cout)) {
        if ((rc = fg_launch_thin_pack(ctx, wt, ws, cout, cin, k, 1))) return rc;
        return fg_launch_thin_in_conv(ctx, gy, ws, nullptr, gx, batch, h, w, cout, cin, k, 1);
    }
    ConvGeom g = mk_geom(batch, h, w, cin, cout, k, pad, up, ctx->fusion);
    const long long pb = a64(fg_geom_pack_floats(g, 1));
    if ((rc = fg_conv_pack(ctx, g, wt, nullptr, ws))) return rc;
    return fg_conv_dgrad_run(ctx, g, gy, ws, gx, ws + pb, (long long)(ws_bytes / 4) - pb);
}
int fg_conv2d_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, float beta, int batch,
                              int h, int w, int cin, int cout, int k, int pad, int up, void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && x && gy && gw && wsv, "null argument");
    int rc = conv_check(ctx, cin, cout, k, pad, up);
    if (rc) return rc;
    if (ws_bytes < fg_conv2d_workspace_bytes(batch, h, w, cin, cout, k, up)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv2d: workspace");
    float* ws = (float*)wsv;
    if (thin_in(cin, cout) || thin_out(cin, cout)) {
        const bool tin = thin_in(cin, cout);
        const int cs = tin ? cin : cout, cw = tin ? cout : cin;
        float* gwt = ws + (long long)FG_THIN_WGRAD_BLOCKS * k * k * cs * cw;
        rc = tin ? fg_launch_thin_wgrad(ctx, x, gy, gwt, batch, h, w, cs, cw, k, +1, ws)
                 : fg_launch_thin_wgrad(ctx, gy, x, gwt, batch, h, w, cs, cw, k, -1, ws);
        if (rc) return rc;
        if ((rc = fg_launch_thin_unpack_grad(ctx, gwt, gw, cout, cin, k, tin ? 0 : 1, beta))) return rc;
        if (gb) return fg_launch_colsum(ctx, gy, (long long)batch * h * w, cout, beta, gb, ws);
        return FG_OK;
    }
    ConvGeom g = mk_geom(batch, h, w, cin, cout, k, pad, up, ctx->fusion);
    return fg_conv_wgrad_run(ctx, g, x, gy, gw, gb, beta, ws, (long long)(ws_bytes / 4));
}

size_t fg_linear_workspace_bytes(int batch, int in_f, int out_f) {
    ConvGeom g = mk_geom(batch, 1, 1, in_f, out_f, 1, 0, 0);
    long long pf = fg_geom_pack_floats(g, 0), pb = fg_geom_pack_floats(g, 1);
    return (size_t)(a64(pf > pb ? pf : pb) + fg_conv_scratch_floats(g) + 64) * sizeof(float);
}
int fg_linear_forward(fg_ctx* ctx, const float* x, const float* wt, const float* bias, float* y, int batch, int in_f,
                      int out_f, void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && x && wt && y && wsv, "null argument");
    if (ws_bytes < fg_linear_workspace_bytes(batch, in_f, out_f)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "linear: workspace");
    if (out_f == 1) return fg_launch_gemv_forward(ctx, x, wt, bias, y, batch, in_f, 0);
    ConvGeom g = mk_geom(batch, 1, 1, in_f, out_f, 1, 0, 0);
    float* ws = (float*)wsv;
    const long long pf = a64(fg_geom_pack_floats(g, 0));
    int rc = fg_conv_pack(ctx, g, wt, ws, nullptr);
    if (rc) return rc;
    return fg_conv_forward_run(ctx, g, x, ws, bias, y, ws + pf, (long long)(ws_bytes / 4) - pf);
}
int fg_linear_backward_data(fg_ctx* ctx, const float* gy, const float* wt, float* gx, int batch, int in_f, int out_f,
                            void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && gy && wt && gx && wsv, "null argument");
    if (out_f == 1) return fg_launch_gemv_backward(ctx, nullptr, wt, nullptr, gy, gx, nullptr, nullptr, 0.f, batch, in_f, 0);
    if (ws_bytes < fg_linear_workspace_bytes(batch, in_f, out_f)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "linear: workspace");
    ConvGeom g = mk_geom(batch, 1, 1, in_f, out_f, 1, 0, 0);
    float* ws = (float*)wsv;
    const long long pb = a64(fg_geom_pack_floats(g, 1));
    int rc = fg_conv_pack(ctx, g, wt, nullptr, ws);
    if (rc) return rc;
    return fg_conv_dgrad_run(ctx, g, gy, ws, gx, ws + pb, (long long)(ws_bytes / 4) - pb);
}
int fg_linear_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, float beta, int batch,
                              int in_f, int out_f, void* wsv, size_t ws_bytes) {
    NEED(ctx, ctx && x && gy && gw && wsv, "null argument");
    if (out_f == 1) return fg_launch_gemv_backward(ctx, x, gw /*unused as w*/, nullptr, gy, nullptr, gw, gb, beta, batch, in_f, 0);
    if (ws_bytes < fg_linear_workspace_bytes(batch, in_f, out_f)) return fg_set_err(ctx, FG_ERR_WORKSPACE, "linear: workspace");
    ConvGeom g = mk_geom(batch, 1, 1, in_f, out_f, 1, 0, 0);
    return fg_conv_wgrad_run(ctx, g, x, gy, gw, gb, beta, (float*)wsv, (long long)(ws_bytes / 4));
}

// ---------------------------------------------------------------- module-level pointwise
long long fg_bn_scratch_floats(int c) { return (long long)3 * CR_ROWBLOCKS_MAX * c + 2 * c + 64; }
int fg_batchnorm_forward(fg_ctx* ctx, const float* x, float* y, long long rows, int c, const float* gamma,
                         const float* beta, const float* slope, float* save_mean, float* save_invstd, float* rmean,
                         float* rvar, float eps, float momentum, int train, float* scratch) {
    NEED(ctx, ctx && x && y && gamma && beta && save_mean && save_invstd && scratch && rows > 0, "bad argument");
    NEED(ctx, train || (rmean && rvar), "evaluate mode needs running stats");
    BnArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.M = rows; a.C = c; a.gamma = gamma; a.beta = beta; a.slope = slope; a.mean = save_mean;
    a.invstd = save_invstd; a.running_mean = rmean; a.running_var = rvar; a.eps = eps; a.momentum = momentum;
    a.train = train; a.scratch = scratch;
    return fg_launch_bn_forward(ctx, a);
}
int fg_batchnorm_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, long long rows, int c,
                          const float* gamma, const float* beta, const float* slope, const float* save_mean,
                          const float* save_invstd, float* ggamma, float* gbeta, float* gslope, float acc, float* scratch) {
    NEED(ctx, ctx && x && gy && gamma && beta && save_mean && save_invstd && scratch && rows > 0, "bad argument");
    BnBwdArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.gy = gy; a.gx = gx; a.M = rows; a.C = c; a.gamma = gamma; a.beta = beta; a.slope = slope;
    a.mean = save_mean; a.invstd = save_invstd; a.ggamma = ggamma; a.gbeta = gbeta; a.gslope = gslope; a.gbeta_acc = acc;
    a.scratch = scratch;
    return fg_launch_bn_backward(ctx, a);
}
int fg_prelu_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y, long long n) {
    NEED(ctx, ctx && x && slope && y && n >= 0, "bad argument");
    return fg_launch_prelu_forward(ctx, x, slope, mask, mscale, y, n);
}
int fg_prelu_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask, float mscale,
                      float* gx, float* gslope, float acc, long long n, float* scratch) {
    NEED(ctx, ctx && x && gy && slope && scratch && n >= 0, "bad argument");
    return fg_launch_prelu_backward(ctx, x, gy, slope, mask, mscale, gx, gslope, acc, n, scratch);
}
int fg_actpool_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y,
                       int batch, int h, int w, int c) {
    NEED(ctx, ctx && x && y, "null argument");
    return fg_launch_actpool_forward(ctx, x, slope, mask, mscale, y, batch, h, w, c);
}
int fg_actpool_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask, float mscale,
                        float* gx, float* gslope, float acc, int batch, int h, int w, int c, float* scratch) {
    NEED(ctx, ctx && x && gy && scratch, "null argument");
    return fg_launch_actpool_backward(ctx, x, gy, slope, mask, mscale, gx, gslope, acc, batch, h, w, c, scratch);
}
int fg_spatial_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float mscale, float* y, int batch, int hw, int c) {
    NEED(ctx, ctx && x && y, "null argument");
    return fg_launch_scale_mask_nc(ctx, x, mask, mscale, y, batch, hw, c);
}
int fg_avgpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int b, int h, int w, int c) {
    NEED(ctx, ctx && x && y && h % 2 == 0 && w % 2 == 0, "bad argument");
    return fg_launch_avgpool_forward(ctx, x, y, b, h, w, c);
}
int fg_avgpool2x2_backward(fg_ctx* ctx, const float* gy, float* gx, int b, int h, int w, int c) {
    NEED(ctx, ctx && gy && gx && h % 2 == 0 && w % 2 == 0, "bad argument");
    return fg_launch_avgpool_backward(ctx, gy, gx, b, h, w, c);
}
int fg_upsample_nearest2x_forward(fg_ctx* ctx, const float* x, float* y, int b, int h, int w, int c) {
    NEED(ctx, ctx && x && y, "null argument");
    return fg_launch_upsample_forward(ctx, x, y, b, h, w, c);
}
static int review_args_ok(fg_ctx* ctx, const void* a, const void* b_, int b, int h, int w, int c, int f) {
    if (!ctx || !a || !b_ || b <= 0 || h <= 0 || w <= 0 || c <= 0 || f < 1 || c % (f * f))
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_conv_upsample_view: bad argument (c must be a multiple of factor^2)");
    return FG_OK;
}
int fg_conv_upsample_view_forward(fg_ctx* ctx, const float* v, float* u, int b, int h, int w, int c, int f) {
    int rc = review_args_ok(ctx, v, u, b, h, w, c, f);
    return rc ? rc : fg_launch_nchw_review(ctx, v, u, b, h, w, c, f, 0);
}
int fg_conv_upsample_view_backward(fg_ctx* ctx, const float* gu, float* gv, int b, int h, int w, int c, int f) {
    int rc = review_args_ok(ctx, gu, gv, b, h, w, c, f);
    return rc ? rc : fg_launch_nchw_review(ctx, gu, gv, b, h, w, c, f, 1);
}
int fg_upsample_nearest2x_backward(fg_ctx* ctx, const float* gy, float* gx, int b, int h, int w, int c) {
    NEED(ctx, ctx && gy && gx, "null argument");
    return fg_launch_upsample_backward(ctx, gy, gx, b, h, w, c);
}
int fg_maxpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int b, int h, int w, int c) {
    NEED(ctx, ctx && x && y, "null argument");
    return fg_launch_maxpool_forward(ctx, x, y, b, h, w, c);
}
int fg_maxpool2x2_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, int b, int h, int w, int c) {
    NEED(ctx, ctx && x && gy && gx, "null argument");
    return fg_launch_maxpool_backward(ctx, x, gy, gx, b, h, w, c);
}
int fg_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float scale, float* y, long long n) {
    NEED(ctx, ctx && x && y && n >= 0, "bad argument");
    return fg_launch_mul_mask(ctx, x, mask, scale, y, n);
}
int fg_concat_channels(fg_ctx* ctx, const float* a, const float* b, float* out, long long npix, int ca, int cb) {
    NEED(ctx, ctx && a && b && out && npix >= 0 && ca > 0 && cb > 0, "bad argument");
    return fg_launch_concat(ctx, a, b, out, npix, ca, cb);
}
int fg_split_channels(fg_ctx* ctx, const float* g, float* ga, float* gb, long long npix, int ca, int cb) {
    NEED(ctx, ctx && g && npix >= 0 && ca > 0 && cb > 0, "bad argument");
    return fg_launch_split(ctx, g, ga, gb, npix, ca, cb);
}
int fg_add(fg_ctx* ctx, const float* a, const float* b, float* out, long long n) {
    NEED(ctx, ctx && a && b && out && n >= 0, "bad argument");
    return fg_launch_add(ctx, a, b, out, n);
}
int fg_sigmoid_forward(fg_ctx* ctx, const float* x, float* y, long long n) { NEED(ctx, ctx && x && y, "null argument"); return fg_launch_sigmoid_forward(ctx, x, y, n); }
int fg_sigmoid_backward(fg_ctx* ctx, const float* y, const float* gy, float* gx, long long n) {
    NEED(ctx, ctx && y && gy && gx, "null argument");
    return fg_launch_sigmoid_backward(ctx, y, gy, gx, n);
}
int fg_leakyrelu_forward(fg_ctx* ctx, const float* x, float s, float* y, long long n) { NEED(ctx, ctx && x && y, "null argument"); return fg_launch_leakyrelu_forward(ctx, x, s, y, n); }
int fg_leakyrelu_backward(fg_ctx* ctx, const float* x, const float* gy, float s, float* gx, long long n) {
    NEED(ctx, ctx && x && gy && gx, "null argument");
    return fg_launch_leakyrelu_backward(ctx, x, gy, s, gx, n);
}

#pragma GCC visibility pop
}  // extern "C"
