// Step-level entries of libfacegen_hip.so: the exchange step of data parallelism behind the C ABI (fg_comm_* : RCCL over
// xGMI, bound at run time), and the evaluation reduction of adversarial.approxParzen.
#include "fg_internal.h"
#include "../../include/facegen_hip.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>      // types and enums only: the library is bound with dlopen/dlsym (see rccl_bind)

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

struct fg_comm {
    fg_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
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

int fg_comm_destroy(fg_comm* c) {
    if (!c) return FG_OK;
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
    FG_HIP(c->ctx, hipEventRecord(c->ev_done, c->side));
    FG_HIP(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
    c->pending = 0;
    return FG_OK;
}

int fg_broadcast(fg_comm* c, float* buf, size_t n, int root) {
    if (!c || !buf || root < 0 || root >= c->world) return fg_set_err(c ? c->ctx : nullptr, FG_ERR_INVALID, "fg_broadcast: bad argument");
    if (n == 0) return FG_OK;
    FG_NCCL(c, g_rccl.Broadcast(buf, buf, n, ncclFloat32, root, c->comm, c->ctx->stream));
    return FG_OK;
}

#pragma GCC visibility pop
}  // extern "C"
