// Step-level entries of libfacegen_hip.so: the exchange step of data parallelism behind the C ABI (fg_comm_* : RCCL over
// xGMI, bound at run time), and the evaluation reduction of adversarial.approxParzen.
#include "fg_internal.h"
#include "../../include/facegen_hip.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>      // types and enums only: the library is bound with dlopen/dlsym (see rccl_bind)
#include <vector>

// ---------------------------------------------------------------------------------------------------------------------
// adversarial.approxParzen (adversarial_c2f.lua:305-344): distance of the ground-truth fine image to each generation
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// one block per generation i: dist[i] = || (gen[i] + cond) - fine ||_2, squares accumulated in fp64 like THTensor_(dist)
__global__ __launch_bounds__(256) void parzen_dist_kernel(const float* __restrict__ gen, const float* __restrict__ cond,
                                                          const float* __restrict__ fine, long long elems,
                                                          float* __restrict__ dist) {
    __shared__ double sh[4];
    const float* g = gen + (long long)blockIdx.x * elems;
    double acc = 0.0;
    for (long long e = threadIdx.x; e < elems; e += 256) {
        const float v = g[e] + cond[e];          // neighbors:add(condInputs) rounds to fp32 first (:322)
        const float d = v - fine[e];
        acc += (double)d * (double)d;
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dist[blockIdx.x] = (float)sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void min_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    float m = 1e10f;                             // `local dist = 1e10` (:324)
    for (int i = threadIdx.x; i < n; i += 64) m = fminf(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) out[0] = m;
}

// ---------------------------------------------------------------------------------------------------------------------
// fg_comm: RCCL communicator (one rank per process / GPU).  librccl is bound at run time so that libfacegen_hip.so keeps
// linking against libamdhip64 only: a single-GPU Lua host needs no RCCL, and a host that already carries a librccl
// (PyTorch wheels bundle one) shares that instance instead of loading a second one.
// ---------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char where[256] = "";
};
static RcclApi g_rccl;

static int rccl_bind(fg_ctx* ctx) {
    if (g_rccl.handle) return FG_OK;
    // an instance that is already in the process first (RTLD_NOLOAD matches by soname), then the usual search path
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    const char* env = getenv("FG_RCCL_LIB");
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { snprintf(g_rccl.where, sizeof(g_rccl.where), "%s%s", n, pass == 0 ? " (already loaded)" : ""); break; }
        }
    if (!h) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "fg_comm: librccl.so not found (%s); set FG_RCCL_LIB", dlerror());
    RcclApi a;
    a.handle = h;
    bool ok = true;
#define FG_SYM(field, name) do { *(void**)(&a.field) = dlsym(h, name); ok = ok && a.field != nullptr; } while (0)
    FG_SYM(GetUniqueId, "ncclGetUniqueId"); FG_SYM(CommInitRank, "ncclCommInitRank"); FG_SYM(CommDestroy, "ncclCommDestroy");
    FG_SYM(AllReduce, "ncclAllReduce"); FG_SYM(Broadcast, "ncclBroadcast"); FG_SYM(GroupStart, "ncclGroupStart");
    FG_SYM(GroupEnd, "ncclGroupEnd"); FG_SYM(GetErrorString, "ncclGetErrorString");
#undef FG_SYM
    if (!ok) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "fg_comm: librccl.so lacks a required symbol");
    memcpy(a.where, g_rccl.where, sizeof(a.where));
    g_rccl = a;
    return FG_OK;
}

struct CommOp { char op; char dtype; char stream; size_t count; };     // op a/b/w, dtype f/d/i/-, stream c/s
struct fg_comm {
    fg_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool dry = false;                  // fg_comm_create_dry: no transport, the schedule is all there is
    bool trace = false;                // fg_comm_set_trace
    std::vector<CommOp> ops;
    hipStream_t side = nullptr;        // the exchange runs here, ordered against the compute stream with events
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    int pending = 0;                   // all-reduces issued on `side` since the last fg_comm_wait
};

#define FG_NCCL(c, call)                                                                                            \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return fg_set_err((c)->ctx, FG_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

extern "C" {
#pragma GCC visibility push(default)

int fg_parzen_min_dist(fg_ctx* ctx, const float* gen, const float* cond, const float* fine, int n, long long elems,
                       float* dist, float* min_out) {
    if (!ctx || !gen || !cond || !fine || !dist || n <= 0 || elems <= 0)
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_parzen_min_dist: bad argument");
    hipLaunchKernelGGL(parzen_dist_kernel, dim3(n), dim3(256), 0, ctx->stream, gen, cond, fine, elems, dist);
    FG_CHECK_LAUNCH(ctx);
    if (min_out) {
        hipLaunchKernelGGL(min_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)dist, n, min_out);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}

int fg_comm_unique_id(fg_ctx* ctx, char* id_out, size_t len) {
    if (!ctx || !id_out || len < FG_COMM_ID_BYTES) return fg_set_err(ctx, FG_ERR_INVALID, "fg_comm_unique_id: buffer of %d bytes required", FG_COMM_ID_BYTES);
    int rc = rccl_bind(ctx);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fg_set_err(ctx, FG_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    static_assert(sizeof(id) == FG_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return FG_OK;
}

int fg_comm_create(fg_ctx* ctx, const char* id, size_t len, int rank, int world, fg_comm** out) {
    if (!ctx || !id || !out || len < FG_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world)
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_comm_create: bad argument");
    int rc = rccl_bind(ctx);
    if (rc) return rc;
    FG_HIP(ctx, hipSetDevice(ctx->device));
    fg_comm* c = new fg_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fg_set_err(ctx, FG_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
        fg_comm_destroy(c);
        return fg_set_err(ctx, FG_ERR_HIP, "fg_comm_create: stream / events");
    }
    *out = c;
    return FG_OK;
}

int fg_comm_create_dry(fg_ctx* ctx, int rank, int world, fg_comm** out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return fg_set_err(ctx, FG_ERR_INVALID, "fg_comm_create_dry: bad argument");
    fg_comm* c = new fg_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->dry = true; c->trace = true;
    *out = c;
    return FG_OK;
}
int fg_comm_set_trace(fg_comm* c, int on) {
    if (!c) return FG_ERR_INVALID;
    c->trace = on != 0 || c->dry;
    return FG_OK;
}
int fg_comm_schedule(fg_comm* c, char* buf, size_t len, int reset) {
    if (!c || !buf || len == 0) return fg_set_err(c ? c->ctx : nullptr, FG_ERR_INVALID, "fg_comm_schedule: bad argument");
    size_t off = 0;
    buf[0] = 0;
    int seq = 0;
    for (const CommOp& o : c->ops) {
        const char* op = o.op == 'a' ? "allreduce" : o.op == 'b' ? "broadcast" : "wait";
        const char* dt = o.dtype == 'f' ? "f32" : o.dtype == 'd' ? "f64" : o.dtype == 'i' ? "i32" : "-";
        int n = snprintf(buf + off, len - off, "%d %s %s %zu %s\n", seq++, op, dt, o.count, o.stream == 's' ? "side" : "compute");
        if (n < 0 || (size_t)n >= len - off) return fg_set_err(c->ctx, FG_ERR_WORKSPACE, "fg_comm_schedule: buffer of %zu bytes too small", len);
        off += n;
    }
    if (reset) c->ops.clear();
    return FG_OK;
}

int fg_comm_destroy(fg_comm* c) {
    if (!c) return FG_OK;
    if (c->dry) { delete c; return FG_OK; }
    if (c->side) { (void)hipStreamSynchronize(c->side); }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return FG_OK;
}

int fg_comm_rank(const fg_comm* c) { return c ? c->rank : -1; }
int fg_comm_world(const fg_comm* c) { return c ? c->world : 0; }
const char* fg_comm_library(void) { return g_rccl.handle ? g_rccl.where : ""; }

static int allreduce_typed(fg_comm* c, void* buf, size_t n, ncclDataType_t dt, int async) {
    if (!c || !buf) return fg_set_err(c ? c->ctx : nullptr, FG_ERR_INVALID, "fg_allreduce: null argument");
    if (n == 0) return FG_OK;
    fg_ctx* ctx = c->ctx;
    if (c->trace) c->ops.push_back(CommOp{'a', dt == ncclFloat32 ? 'f' : dt == ncclFloat64 ? 'd' : 'i', async ? 's' : 'c', n});
    if (c->dry) { if (async) c->pending++; return FG_OK; }
    if (!async) {
        FG_NCCL(c, g_rccl.AllReduce(buf, buf, n, dt, ncclSum, c->comm, ctx->stream));
        return FG_OK;
    }
    // overlapped: everything enqueued on the compute stream so far (the gradients) happens-before the exchange, which
    // runs on the side stream while the compute stream goes on; fg_comm_wait joins the two again
    FG_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
    FG_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_ready, 0));
    FG_NCCL(c, g_rccl.AllReduce(buf, buf, n, dt, ncclSum, c->comm, c->side));
    c->pending++;
    return FG_OK;
}

int fg_allreduce_sum(fg_comm* c, float* buf, size_t n) { return allreduce_typed(c, buf, n, ncclFloat32, 0); }
int fg_allreduce_sum_async(fg_comm* c, float* buf, size_t n) { return allreduce_typed(c, buf, n, ncclFloat32, 1); }
int fg_allreduce_sum_f64(fg_comm* c, double* buf, size_t n) { return allreduce_typed(c, buf, n, ncclFloat64, 0); }
int fg_allreduce_sum_i32(fg_comm* c, int* buf, size_t n) { return allreduce_typed(c, buf, n, ncclInt32, 0); }

int fg_comm_wait(fg_comm* c) {
    if (!c) return FG_ERR_INVALID;
    if (!c->pending) return FG_OK;
    if (c->trace) c->ops.push_back(CommOp{'w', '-', 'c', (size_t)c->pending});
    if (c->dry) { c->pending = 0; return FG_OK; }
    FG_HIP(c->ctx, hipEventRecord(c->ev_done, c->side));
    FG_HIP(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
    c->pending = 0;
    return FG_OK;
}

int fg_broadcast(fg_comm* c, float* buf, size_t n, int root) {
    if (!c || !buf || root < 0 || root >= c->world) return fg_set_err(c ? c->ctx : nullptr, FG_ERR_INVALID, "fg_broadcast: bad argument");
    if (n == 0) return FG_OK;
    if (c->trace) c->ops.push_back(CommOp{'b', 'f', 'c', n});
    if (c->dry) return FG_OK;
    FG_NCCL(c, g_rccl.Broadcast(buf, buf, n, ncclFloat32, root, c->comm, c->ctx->stream));
    return FG_OK;
}

#pragma GCC visibility pop
}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// fg_gan: the two closures of the training loop as device-resident C entries (SURVEY.md 8(b) level (ii)).
//   fg_step_D = adversarial.lua:240-268 + fevalD (:83-179) + interruptableAdam/Sgd/Adagrad on D
//   fg_step_G = adversarial.lua:275-288 + fevalG_on_D (:187-231) + the optimizer on G
//   table mode = adversarial_c2f.lua:40-187 (G{noise, cond} through JoinTable, D{x, cond} through CAddTable)
// One host call enqueues a whole closure: the noise batch and every dropout mask come from ONE Philox launch, G's last
// stage writes straight into D's batch, criterion / gradients / penalty + clamp + optimizer never leave HBM, and (N > 1)
// the gradient all-reduce runs on the communicator's stream -- D's under the next generator forward (its update is
// deferred until D is next evaluated), G's in buckets under its own backward.
// ---------------------------------------------------------------------------------------------------------------------
#include <vector>

struct OptCfg {
    int method = 0;                    // 0 adam, 1 sgd, 2 adagrad
    double lr = -1.0, beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
    double momentum = 0.0, dampening = -1.0, weight_decay = 0.0, lr_decay = 0.0;
    int nesterov = 0;
    int steps = 0;                     // Adam's t / optim.sgd's evalCounter
    bool mom_init = false;             // sgd: the momentum buffer holds a gradient already
};
struct GBucket { int s_from, s_to; long long lo, hi; };

struct fg_gan {
    fg_ctx* ctx = nullptr;
    fg_net *G = nullptr, *D = nullptr;
    fg_comm* comm = nullptr;
    int table = 0, maxB = 0;
    float *wsG = nullptr, *wsD = nullptr, *ws = nullptr;
    size_t wsG_bytes = 0, wsD_bytes = 0, ws_bytes = 0;
    int gc = 0, gh = 0, gw = 0, ic = 0, ih = 0, iw = 0;
    long long img = 0, gin = 0, nz_elems = 0, nP[2] = {0, 0};      // per-sample sizes; parameter counts [D, G]
    long long o_dinput = 0, o_ginput = 0, o_noise = 0, o_dsum = 0, o_gx = 0, o_targets = 0, o_dprob = 0,
              o_loss = 0, o_conf = 0, o_opt[2] = {0, 0}, o_sync = 0, total = 0;
    std::vector<long long> o_mask[2];  // [D, G] dropout masks
    float l1[2] = {0.f, 0.f}, l2[2] = {1e-4f, 0.f}, clamp[2] = {1.f, 5.f};    // train.lua:29-37
    OptCfg opt[2];
    uint64_t noise_seed = 1, noise_off = 0, mask_seed = 1000, mask_off = 0;
    int targets_B[2] = {-1, -1};
    int pendingD = 0;                  // D's gradient all-reduce is in flight, its update is deferred
    int grads_local[2] = {0, 0};       // a NO_UPDATE step left un-reduced gradients for fg_gan_update
    long long d_out_off = 0;
    int last_B[2] = {0, 0};
    std::vector<GBucket> buckets;
    bool sync_bn = false;
    int overlap = 1;                   // 2 (test hook): take the N > 1 exchange path even on a one-rank communicator
};

static inline long long al64(long long v) { return (v + 63) / 64 * 64; }

static void gan_layout(fg_gan* g) {
    const long long B = g->maxB;
    long long off = 0;
    auto take = [&](long long n) { const long long o = off; off += al64(n); return o; };
    g->o_dinput = take(B * g->img);
    g->o_ginput = take(B * g->gin);
    g->o_noise = take(B * g->nz_elems);
    g->o_dsum = g->table ? take(B * g->img) : 0;
    g->o_gx = take(B * g->img);
    for (int w = 0; w < 2; ++w) {
        const fg_net* n = w == 0 ? g->D : g->G;
        const int nm = fg_net_num_masks(n);
        g->o_mask[w].resize(nm);
        for (int i = 0; i < nm; ++i) g->o_mask[w][i] = take(fg_net_mask_elems(n, i, (int)B));
    }
    g->o_targets = take(2 * B);
    g->o_dprob = take(B);
    g->o_loss = take(4);
    g->o_conf = take(8);
    g->o_opt[0] = take(2 * g->nP[0]);
    g->o_opt[1] = take(2 * g->nP[1]);
    int cmax = fg_net_max_bn_channels(g->G);
    if (fg_net_max_bn_channels(g->D) > cmax) cmax = fg_net_max_bn_channels(g->D);
    g->o_sync = take(2 * (2LL * cmax + 2));     // doubles
    g->total = off;
}

static void gan_buckets(fg_gan* g, long long target) {
    g->buckets.clear();
    const int ns = fg_net_num_stages(g->G);
    int cur_from = ns - 1;
    long long lo_acc = -1, hi_acc = -1;
    for (int st = ns - 1; st >= 0; --st) {
        long long lo = 0, hi = 0;
        fg_net_stage_params(g->G, st, &lo, &hi);
        if (hi > lo) {
            lo_acc = lo_acc < 0 ? lo : (lo < lo_acc ? lo : lo_acc);
            hi_acc = hi_acc < 0 ? hi : (hi > hi_acc ? hi : hi_acc);
        }
        if ((hi_acc >= 0 && hi_acc - lo_acc >= target && st > 0) || st == 0) {
            g->buckets.push_back(GBucket{cur_from, st, lo_acc < 0 ? 0 : lo_acc, hi_acc < 0 ? 0 : hi_acc});
            cur_from = st - 1; lo_acc = hi_acc = -1;
        }
    }
}

static int gan_world(const fg_gan* g) { return g->comm ? g->comm->world : 1; }
static bool gan_exchange(const fg_gan* g) { return g->comm && (g->comm->world > 1 || g->overlap == 2); }

// sync-BN: run a forward / backward to completion through its pauses (one fp64 all-reduce per BatchNorm)
static int gan_drain(fg_gan* g, fg_net* n, int rc, bool fwd) {
    while (rc == FG_PAUSED_SYNC) {
        if (!g->comm) return fg_set_err(g->ctx, FG_ERR_INVALID, "fg_gan: sync-BN without a communicator");
        int r = fg_allreduce_sum_f64(g->comm, (double*)(g->ws + g->o_sync), (size_t)fg_net_sync_count(n));
        if (r) return r;
        long long off = 0;
        rc = fwd ? fg_net_forward_resume(n, &off) : fg_net_backward_resume(n);
    }
    return rc;
}

static int gan_optimize(fg_gan* g, int w) {
    fg_ctx* ctx = g->ctx;
    fg_net* net = w == 0 ? g->D : g->G;
    float *p = nullptr, *gr = nullptr;
    fg_net_vectors(net, &p, &gr, nullptr);
    const long long n = g->nP[w];
    OptCfg& o = g->opt[w];
    const float gscale = 1.f / (float)gan_world(g);   // BCE averages over the local batch: global mean = all-reduced sum / world
    const bool pen = g->l1[w] != 0.f || g->l2[w] != 0.f;
    // adversarial.lua:109 (D: sign * D_L1) vs :223 (G: sign * G_L2 -- quirk C4, preserved)
    const float l1mul = pen ? (w == 0 ? g->l1[w] : g->l2[w]) : 0.f, l2 = pen ? g->l2[w] : 0.f;
    float* s0 = g->ws + g->o_opt[w];
    int rc;
    if (o.method == 0) {
        AdamArgs a; memset(&a, 0, sizeof(a));
        o.steps += 1;
        a.p = p; a.g = gr; a.m = s0; a.v = s0 + n; a.n = n; a.gscale = gscale; a.l1_mul = l1mul; a.l2 = l2; a.clamp = g->clamp[w];
        a.lr_d = o.lr < 0 ? 1e-3 : o.lr; a.beta1_d = o.beta1; a.beta2_d = o.beta2;
        a.beta1 = (float)o.beta1; a.beta2 = (float)o.beta2; a.eps = (float)o.eps; a.t = o.steps; a.gout = nullptr;
        return fg_net_adam_step(net, a);          // one launch: penalty + clamp + Adam + the re-pack of every layer
    } else if (o.method == 1) {
        const double lr = o.lr < 0 ? 1e-3 : o.lr, damp = o.dampening < 0 ? o.momentum : o.dampening;
        if (o.nesterov && (o.momentum <= 0 || damp != 0)) return fg_set_err(ctx, FG_ERR_INVALID, "Nesterov momentum requires a momentum and zero dampening");
        const double clr = lr / (1.0 + o.steps * o.lr_decay);
        const int first = (o.momentum != 0 && !o.mom_init) ? 1 : 0;
        rc = fg_launch_sgd(ctx, p, gr, o.momentum != 0 ? s0 : nullptr, n, gscale, l1mul, l2, g->clamp[w], (float)clr,
                           (float)o.momentum, (float)(1.0 - damp), (float)o.weight_decay, o.nesterov, first);
        if (o.momentum != 0) o.mom_init = true;
        o.steps += 1;
    } else {
        const double lr = o.lr < 0 ? 1e-3 : o.lr;
        const double clr = lr / (1.0 + o.steps * o.lr_decay);
        rc = fg_launch_adagrad(ctx, p, gr, s0, n, gscale, l1mul, l2, g->clamp[w], (float)clr);
        o.steps += 1;
    }
    if (rc) return rc;
    return fg_net_params_changed(net);
}

static int gan_finish_pending(fg_gan* g) {
    if (!g->pendingD) return FG_OK;
    g->pendingD = 0;
    int rc = fg_comm_wait(g->comm);
    if (rc) return rc;
    return gan_optimize(g, 0);
}

// D's update: with N > 1 the 11 MB all-reduce is started on the communicator's stream and the optimizer step waits until D
// is next needed (the generator forward that follows does not read D)
static int gan_schedule_update_D(fg_gan* g) {
    if (gan_exchange(g)) {
        float* gr = nullptr;
        fg_net_vectors(g->D, nullptr, &gr, nullptr);
        if (g->overlap) {
            int rc = fg_allreduce_sum_async(g->comm, gr, (size_t)g->nP[0]);
            if (rc) return rc;
            g->pendingD = 1;
            return FG_OK;
        }
        int rc = fg_allreduce_sum(g->comm, gr, (size_t)g->nP[0]);
        if (rc) return rc;
    }
    return gan_optimize(g, 0);
}

static int gan_targets(fg_gan* g, int w, int B) {
    if (g->targets_B[w] == B) return FG_OK;
    float* t = g->ws + g->o_targets + (w ? g->maxB : 0);
    int rc;
    if (w == 0) {       // Y_NOT_GENERATOR = 1 for the real half (adversarial.lua:247), fakes 0 (:255)
        if ((rc = fg_launch_fill(g->ctx, t, 1.f, B / 2))) return rc;
        if ((rc = fg_launch_fill(g->ctx, t + B / 2, 0.f, B / 2))) return rc;
    } else if ((rc = fg_launch_fill(g->ctx, t, 1.f, B))) return rc;     // targets:fill(Y_NOT_GENERATOR) (:277)
    g->targets_B[w] = B;
    return FG_OK;
}

// noise batch (when the caller passes none) + every dropout mask (ditto) of one closure in a single Philox launch
static int gan_draw(fg_gan* g, int n_noise_rows, const float* noise_in, int B, const float* const* masks_in,
                    const float** noise_out, std::vector<const float*>& dmasks) {
    RngMulti m; memset(&m, 0, sizeof(m));
    if (noise_in) *noise_out = noise_in;
    else {
        RngSeg& s = m.seg[m.n++];
        s.out = g->ws + g->o_noise; s.n = (long long)n_noise_rows * g->nz_elems; s.seed = g->noise_seed; s.offset = g->noise_off;
        s.lo = -1.f; s.hi = 1.f; s.mode = 0;                       // NN_UTILS.createNoiseInputs: uniform(-1, 1) (nn_utils.lua:37)
        g->noise_off += (uint64_t)((s.n + 3) / 4);
        *noise_out = s.out;
    }
    const int nm = fg_net_num_masks(g->D);
    dmasks.assign(nm, nullptr);
    for (int i = 0; i < nm; ++i) {
        if (masks_in) { dmasks[i] = masks_in[i]; continue; }
        if (m.n >= FG_RNG_MAX_SEGS) return fg_set_err(g->ctx, FG_ERR_UNSUPPORTED, "fg_gan: more than %d random segments per step", FG_RNG_MAX_SEGS);
        RngSeg& s = m.seg[m.n++];
        s.out = g->ws + g->o_mask[0][i]; s.n = fg_net_mask_elems(g->D, i, B); s.seed = g->mask_seed; s.offset = g->mask_off;
        s.lo = fg_net_mask_keep(g->D, i); s.hi = 0.f; s.mode = 1;
        g->mask_off += (uint64_t)((s.n + 3) / 4);
        dmasks[i] = s.out;
    }
    return fg_launch_rng_multi(g->ctx, m);
}

extern "C" {
#pragma GCC visibility push(default)

size_t fg_gan_workspace_bytes(const fg_net* G, const fg_net* D, int table_inputs, int max_batch) {
    if (!G || !D || max_batch <= 0) return 0;
    fg_gan t;
    t.G = (fg_net*)G; t.D = (fg_net*)D; t.table = table_inputs; t.maxB = max_batch;
    fg_net_in_dims(G, &t.gc, &t.gh, &t.gw); fg_net_in_dims(D, &t.ic, &t.ih, &t.iw);
    t.img = (long long)t.ic * t.ih * t.iw; t.gin = (long long)t.gc * t.gh * t.gw;
    t.nz_elems = table_inputs ? (long long)t.gh * t.gw : t.gin;
    t.nP[0] = fg_net_num_params(D); t.nP[1] = fg_net_num_params(G);
    gan_layout(&t);
    return (size_t)t.total * sizeof(float) + 256;
}

int fg_gan_create(fg_ctx* ctx, fg_net* G, fg_net* D, int table_inputs, int max_batch, void* ws, size_t ws_bytes, fg_gan** out) {
    if (!ctx || !G || !D || !ws || !out || max_batch < 2) return fg_set_err(ctx, FG_ERR_INVALID, "fg_gan_create: bad argument");
    if ((uintptr_t)ws & 255) return fg_set_err(ctx, FG_ERR_INVALID, "fg_gan_create: workspace must be 256-byte aligned");
    fg_gan* g = new fg_gan();
    g->ctx = ctx; g->G = G; g->D = D; g->table = table_inputs ? 1 : 0; g->maxB = max_batch; g->ws = (float*)ws; g->ws_bytes = ws_bytes;
    fg_net_in_dims(G, &g->gc, &g->gh, &g->gw); fg_net_in_dims(D, &g->ic, &g->ih, &g->iw);
    g->img = (long long)g->ic * g->ih * g->iw; g->gin = (long long)g->gc * g->gh * g->gw;
    g->nz_elems = g->table ? (long long)g->gh * g->gw : g->gin;
    g->nP[0] = fg_net_num_params(D); g->nP[1] = fg_net_num_params(G);
    int oc = 0, oh = 0, ow = 0;
    fg_net_out_dims(G, &oc, &oh, &ow);
    int rc = FG_OK;
    if ((long long)oc * oh * ow != g->img) rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_gan_create: G produces %dx%dx%d, D takes %dx%dx%d", oc, oh, ow, g->ic, g->ih, g->iw);
    if (!rc && g->table && (g->gh != g->ih || g->gw != g->iw || g->gc != g->ic + 1))
        rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_gan_create: table mode expects G{noise[1], cond[C]} at D's resolution");
    fg_net_out_dims(D, &oc, &oh, &ow);
    if (!rc && oc * oh * ow != 1) rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_gan_create: D must end in one probability");
    if (!rc && fg_net_num_masks(G) != 0) rc = fg_set_err(ctx, FG_ERR_UNSUPPORTED, "fg_gan_create: dropout inside G is not built");
    if (!rc) {
        gan_layout(g);
        if ((size_t)g->total * sizeof(float) > ws_bytes) rc = fg_set_err(ctx, FG_ERR_WORKSPACE, "fg_gan_create: workspace %zu < %lld bytes", ws_bytes, g->total * 4LL);
    }
    if (!rc && !g_fg_dry && hipMemsetAsync(g->ws + g->o_opt[0], 0, (size_t)(2 * g->nP[0]) * 4, ctx->stream) != hipSuccess) rc = fg_set_err(ctx, FG_ERR_HIP, "fg_gan_create: memset");
    if (!rc && !g_fg_dry && hipMemsetAsync(g->ws + g->o_opt[1], 0, (size_t)(2 * g->nP[1]) * 4, ctx->stream) != hipSuccess) rc = fg_set_err(ctx, FG_ERR_HIP, "fg_gan_create: memset");
    if (rc) { delete g; return rc; }
    gan_buckets(g, 900000);
    *out = g;
    return FG_OK;
}

int fg_gan_destroy(fg_gan* g) { delete g; return FG_OK; }

/* the nets' own workspaces (fg_net_workspace_bytes(net, max_batch)); re-bind whenever the caller re-allocates them */
int fg_gan_bind_workspaces(fg_gan* g, void* wsG, size_t wsG_bytes, void* wsD, size_t wsD_bytes) {
    if (!g || !wsG || !wsD) return fg_set_err(g ? g->ctx : nullptr, FG_ERR_INVALID, "fg_gan_bind_workspaces: null argument");
    g->wsG = (float*)wsG; g->wsG_bytes = wsG_bytes; g->wsD = (float*)wsD; g->wsD_bytes = wsD_bytes;
    return FG_OK;
}

int fg_gan_set_comm(fg_gan* g, fg_comm* comm, int sync_bn, int overlap) {
    if (!g) return FG_ERR_INVALID;
    int rc = gan_finish_pending(g);
    if (rc) return rc;
    g->comm = comm; g->overlap = overlap;
    g->sync_bn = comm && comm->world > 1 && sync_bn;
    const int cmax = fg_net_max_bn_channels(g->G) > fg_net_max_bn_channels(g->D) ? fg_net_max_bn_channels(g->G) : fg_net_max_bn_channels(g->D);
    for (fg_net* n : {g->G, g->D})
        if ((rc = fg_net_set_sync_bn(n, g->sync_bn ? 1 : 0, g->sync_bn ? (double*)(g->ws + g->o_sync) : nullptr, 2LL * cmax + 2))) return rc;
    return FG_OK;
}

int fg_gan_set_seeds(fg_gan* g, uint64_t noise_seed, uint64_t noise_offset, uint64_t mask_seed, uint64_t mask_offset) {
    if (!g) return FG_ERR_INVALID;
    g->noise_seed = noise_seed; g->noise_off = noise_offset; g->mask_seed = mask_seed; g->mask_off = mask_offset;
    return FG_OK;
}

/* which: 0 = D, 1 = G.  OPT.{D,G}_L1 / _L2 / _clamp (train.lua:29-37) */
int fg_gan_set_penalty(fg_gan* g, int which, float l1, float l2, float clamp) {
    if (!g || which < 0 || which > 1) return FG_ERR_INVALID;
    g->l1[which] = l1; g->l2[which] = l2; g->clamp[which] = clamp;
    return FG_OK;
}

int fg_gan_set_optimizer(fg_gan* g, int which, int method, double lr, double beta1, double beta2, double eps, double momentum,
                         double dampening, double weight_decay, double lr_decay, int nesterov) {
    if (!g || which < 0 || which > 1 || method < 0 || method > 2) return fg_set_err(g ? g->ctx : nullptr, FG_ERR_INVALID, "fg_gan_set_optimizer: bad argument");
    OptCfg& o = g->opt[which];
    if (o.method != method) {      // a different rule starts from a clean state (OPTSTATE.<method>.<net> are separate tables)
        if (which == 0) { int rc = gan_finish_pending(g); if (rc) return rc; }
        FG_HIP(g->ctx, hipMemsetAsync(g->ws + g->o_opt[which], 0, (size_t)(2 * g->nP[which]) * 4, g->ctx->stream));
        o.steps = 0; o.mom_init = false;
    }
    o.method = method; o.lr = lr; o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.momentum = momentum; o.dampening = dampening;
    o.weight_decay = weight_decay; o.lr_decay = lr_decay; o.nesterov = nesterov;
    return FG_OK;
}
int fg_gan_optimizer_steps(const fg_gan* g, int which) { return (g && which >= 0 && which <= 1) ? g->opt[which].steps : -1; }
int fg_gan_set_optimizer_steps(fg_gan* g, int which, int steps) {
    if (!g || which < 0 || which > 1 || steps < 0) return FG_ERR_INVALID;
    g->opt[which].steps = steps; g->opt[which].mom_init = steps > 0;
    return FG_OK;
}

int fg_gan_buffer(const fg_gan* g, int what, long long* offset_floats, long long* count) {
    if (!g) return FG_ERR_INVALID;
    long long o = -1, c = 0;
    switch (what) {
        case FG_GAN_D_INPUT: o = g->o_dinput; c = g->maxB * g->img; break;
        case FG_GAN_NOISE: o = g->o_noise; c = g->maxB * g->nz_elems; break;
        case FG_GAN_D_GRAD_INPUT: o = g->o_gx; c = g->maxB * g->img; break;
        case FG_GAN_LOSS: o = g->o_loss; c = 2; break;
        case FG_GAN_CONFUSION: o = g->o_conf; c = 8; break;
        case FG_GAN_OPT_STATE_D: o = g->o_opt[0]; c = 2 * g->nP[0]; break;
        case FG_GAN_OPT_STATE_G: o = g->o_opt[1]; c = 2 * g->nP[1]; break;
        case FG_GAN_D_OUTPUT: o = g->d_out_off; c = g->last_B[0] > g->last_B[1] ? g->last_B[0] : g->last_B[1]; break;
        case FG_GAN_D_MASKS: o = g->o_mask[0].empty() ? 0 : g->o_mask[0][0]; c = g->o_targets - o; break;
        case FG_GAN_SYNC_BUF: o = g->o_sync; c = g->total - g->o_sync; break;
        default: return fg_set_err(g->ctx, FG_ERR_INVALID, "fg_gan_buffer: unknown buffer %d", what);
    }
    if (offset_floats) *offset_floats = o;
    if (count) *count = c;
    return FG_OK;
}
long long fg_gan_mask_offset(const fg_gan* g, int mask_index) {
    return (g && mask_index >= 0 && mask_index < (int)g->o_mask[0].size()) ? g->o_mask[0][mask_index] : -1;
}

int fg_gan_finish_pending(fg_gan* g) { return g ? gan_finish_pending(g) : FG_ERR_INVALID; }
int fg_gan_pending(const fg_gan* g) { return g ? g->pendingD : 0; }

static int gan_check_step(fg_gan* g, int B, const char* who) {
    if (!g) return FG_ERR_INVALID;
    if (!g->wsG || !g->wsD) return fg_set_err(g->ctx, FG_ERR_INVALID, "%s: fg_gan_bind_workspaces first", who);
    if (B < 2 || (B & 1) || B > g->maxB) return fg_set_err(g->ctx, FG_ERR_INVALID, "%s: batch %d (even, 2..%d)", who, B, g->maxB);
    return FG_OK;
}

int fg_step_D(fg_gan* g, int B, const float* real, const float* cond_real, const float* cond_fake, const float* noise,
              const float* const* masks, int flags) {
    int rc = gan_check_step(g, B, "fg_step_D");
    if (rc) return rc;
    fg_ctx* ctx = g->ctx;
    if (!real || (g->table && (!cond_real || !cond_fake))) return fg_set_err(ctx, FG_ERR_INVALID, "fg_step_D: null input");
    const int h = B / 2;
    if ((rc = gan_finish_pending(g))) return rc;
    const float* nz = nullptr;
    std::vector<const float*> dm;
    if ((rc = gan_draw(g, h, noise, B, masks, &nz, dm))) return rc;
    float* dinput = g->ws + g->o_dinput;
    const float* gin = nz;
    if (g->table) {                      // nn.JoinTable(2, 2){noise, cond} (models_c2f.lua:116)
        if ((rc = fg_launch_concat(ctx, nz, cond_fake, g->ws + g->o_ginput, (long long)h * g->gh * g->gw, 1, g->ic))) return rc;
        gin = g->ws + g->o_ginput;
    }
    long long off = 0;
    // C5: the fakes come from G in TRAIN mode (BatchNorm batch statistics over B/2; running statistics move)
    rc = fg_net_forward_to(g->G, h, gin, g->wsG, g->wsG_bytes, 1, nullptr, 0, &off, dinput + (long long)h * g->img);
    if ((rc = gan_drain(g, g->G, rc, true))) return rc;
    const float* din = dinput;
    if (g->table) {                      // nn.CAddTable{x, cond} (models_c2f.lua:240) over [real | fake] x [cond_real | cond_fake]
        if ((rc = fg_launch_add_halves(ctx, real, cond_real, dinput + (long long)h * g->img, cond_fake, g->ws + g->o_dsum,
                                       (long long)h * g->img))) return rc;
        din = g->ws + g->o_dsum;
    } else if ((rc = fg_launch_copy(ctx, real, dinput, (long long)h * g->img))) return rc;
    if ((rc = gan_targets(g, 0, B))) return rc;
    rc = fg_net_forward(g->D, B, din, g->wsD, g->wsD_bytes, 1, dm.empty() ? nullptr : dm.data(), (int)dm.size(), &off);
    if ((rc = gan_drain(g, g->D, rc, true))) return rc;
    g->d_out_off = off; g->last_B[0] = B;
    int* conf = (int*)(g->ws + g->o_conf);
    if ((rc = fg_launch_bce(ctx, g->wsD + off, g->ws + g->o_targets, g->ws + g->o_loss, g->ws + g->o_dprob, conf, B))) return rc;
    rc = fg_net_backward(g->D, B, din, g->ws + g->o_dprob, g->wsD, g->wsD_bytes, FG_BWD_PARAM_GRADS, nullptr);
    if ((rc = gan_drain(g, g->D, rc, false))) return rc;
    if (flags & FG_STEP_NO_UPDATE) {
        // the maxAccuracyD gate (adversarial.lua:124-178) is decided on the host from the confusion counts; with N > 1 it
        // must be the same decision on every rank: conf[4..7] = counts of the GLOBAL batch
        if ((rc = fg_launch_copy(ctx, (const float*)conf, (float*)(conf + 4), 4))) return rc;
        if (gan_exchange(g) && (rc = fg_allreduce_sum_i32(g->comm, conf + 4, 4))) return rc;
        g->grads_local[0] = 1;
        return FG_OK;
    }
    return gan_schedule_update_D(g);
}

int fg_step_G(fg_gan* g, int B, const float* cond, const float* noise, const float* const* masks, int flags) {
    int rc = gan_check_step(g, B, "fg_step_G");
    if (rc) return rc;
    fg_ctx* ctx = g->ctx;
    if (g->table && !cond) return fg_set_err(ctx, FG_ERR_INVALID, "fg_step_G: null input");
    const float* nz = nullptr;
    std::vector<const float*> dm;
    if ((rc = gan_draw(g, B, noise, B, masks, &nz, dm))) return rc;
    float* dinput = g->ws + g->o_dinput;          // `samples` (adversarial.lua:202): G's output IS D's batch
    const float* gin = nz;
    if (g->table) {
        if ((rc = fg_launch_concat(ctx, nz, cond, g->ws + g->o_ginput, (long long)B * g->gh * g->gw, 1, g->ic))) return rc;
        gin = g->ws + g->o_ginput;
    }
    long long off = 0;
    rc = fg_net_forward_to(g->G, B, gin, g->wsG, g->wsG_bytes, 1, nullptr, 0, &off, dinput);
    if ((rc = gan_drain(g, g->G, rc, true))) return rc;
    if ((rc = gan_finish_pending(g))) return rc;   // D's deferred update must land before D is evaluated
    const float* din = dinput;
    if (g->table) {
        if ((rc = fg_launch_add(ctx, dinput, cond, g->ws + g->o_dsum, (long long)B * g->img))) return rc;
        din = g->ws + g->o_dsum;
    }
    if ((rc = gan_targets(g, 1, B))) return rc;
    rc = fg_net_forward(g->D, B, din, g->wsD, g->wsD_bytes, 1, dm.empty() ? nullptr : dm.data(), (int)dm.size(), &off);
    if ((rc = gan_drain(g, g->D, rc, true))) return rc;
    g->d_out_off = off; g->last_B[1] = B;
    if ((rc = fg_launch_bce(ctx, g->wsD + off, g->ws + g->o_targets + g->maxB, g->ws + g->o_loss + 1, g->ws + g->o_dprob, nullptr, B))) return rc;
    // MODEL_D.modules[1].gradInput (adversarial.lua:210); D's weight gradients are not formed (quirk C6)
    float* gx = g->ws + g->o_gx;
    rc = fg_net_backward(g->D, B, din, g->ws + g->o_dprob, g->wsD, g->wsD_bytes, FG_BWD_INPUT_GRAD, gx);
    if ((rc = gan_drain(g, g->D, rc, false))) return rc;
    float* gG = nullptr;
    fg_net_vectors(g->G, nullptr, &gG, nullptr);
    const bool update = !(flags & FG_STEP_NO_UPDATE);
    if (update && gan_exchange(g) && g->overlap) {
        // bucketed all-reduce overlapped with backward: G's gradients are produced output -> input; each finished ~1 M-parameter
        // range of the flat vector goes onto the communicator's stream while the earlier layers still compute
        const int last = fg_net_num_stages(g->G) - 1;
        for (const GBucket& b : g->buckets) {
            rc = fg_net_backward_range(g->G, B, gin, b.s_from == last ? gx : nullptr, g->wsG, g->wsG_bytes, FG_BWD_PARAM_GRADS, nullptr,
                                       b.s_from, b.s_to);
            if ((rc = gan_drain(g, g->G, rc, false))) return rc;
            if (b.hi > b.lo && (rc = fg_allreduce_sum_async(g->comm, gG + b.lo, (size_t)(b.hi - b.lo)))) return rc;
        }
        if ((rc = fg_comm_wait(g->comm))) return rc;
        return gan_optimize(g, 1);
    }
    rc = fg_net_backward(g->G, B, gin, gx, g->wsG, g->wsG_bytes, FG_BWD_PARAM_GRADS, nullptr);
    if ((rc = gan_drain(g, g->G, rc, false))) return rc;
    if (!update) { g->grads_local[1] = 1; return FG_OK; }
    if (gan_exchange(g) && (rc = fg_allreduce_sum(g->comm, gG, (size_t)g->nP[1]))) return rc;
    return gan_optimize(g, 1);
}

/* the optimizer step a FG_STEP_NO_UPDATE closure left out (interruptable_optimizers.lua:60-66: not calling it IS the
 * false,false return of fevalD -- no update, no step count) */
int fg_gan_update(fg_gan* g, int which) {
    if (!g || which < 0 || which > 1) return FG_ERR_INVALID;
    if (!g->grads_local[which]) return fg_set_err(g->ctx, FG_ERR_INVALID, "fg_gan_update: no gradients pending for %s", which ? "G" : "D");
    g->grads_local[which] = 0;
    if (which == 0) return gan_schedule_update_D(g);
    if (gan_exchange(g)) {
        float* gG = nullptr;
        fg_net_vectors(g->G, nullptr, &gG, nullptr);
        int rc = fg_allreduce_sum(g->comm, gG, (size_t)g->nP[1]);
        if (rc) return rc;
    }
    return gan_optimize(g, 1);
}

#pragma GCC visibility pop
}  // extern "C"
