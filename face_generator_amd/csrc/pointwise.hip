// HBM-bound pointwise / reduction kernels of the GAN step for gfx950 (wave = 64):
// layout conversion, column sums, SpatialBatchNormalization(+PReLU), PReLU(+Dropout),
// PReLU+SpatialDropout+AvgPool, nearest upsample, sigmoid, LeakyReLU, Linear(K->1)+Sigmoid,
// BCECriterion, fused penalty+clamp+Adam, norms, Philox RNG.
// Reference semantics: SURVEY.md Appendix A (Torch7 nn modules; call sites models.lua:57-81, 382-416,
// train.lua:148, interruptable_optimizers.lua:49-94, adversarial.lua:103-123).
#include <string.h>
#include "fg_internal.h"

#define FG_GRID(n, bs) dim3((unsigned)((((n) + (bs)-1) / (bs)) < 4096 ? (((n) + (bs)-1) / (bs)) : 4096))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// block sum (256 threads), result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}

// ------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ s, float* __restrict__ d, int N, int C, int H, int W) {
    long long total = (long long)N * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long t = i / C;
        int w = (int)(t % W); t /= W;
        int h = (int)(t % H);
        int n = (int)(t / H);
        d[i] = s[(((long long)n * C + c) * H + h) * W + w];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ s, float* __restrict__ d, int N, int C, int H, int W) {
    long long total = (long long)N * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long long t = i / W;
        int h = (int)(t % H); t /= H;
        int c = (int)(t % C);
        int n = (int)(t / C);
        d[i] = s[(((long long)n * H + h) * W + w) * C + c];
    }
}
int fg_launch_nchw_to_nhwc(fg_ctx* ctx, const float* s, float* d, int N, int C, int H, int W) {
    long long n = (long long)N * C * H * W;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, s, d, N, C, H, W);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_nhwc_to_nchw(fg_ctx* ctx, const float* s, float* d, int N, int C, int H, int W) {
    long long n = (long long)N * C * H * W;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, s, d, N, C, H, W);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

__global__ void fill_kernel(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}
int fg_launch_fill(fg_ctx* ctx, float* p, float v, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(fill_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, p, v, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = (b == 0.f) ? a * x[i] : a * x[i] + b * y[i];
}
int fg_launch_axpby(fg_ctx* ctx, float a, const float* x, float b, float* y, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(axpby_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, a, x, b, y, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ column reductions over [M][C]
// block = (64 columns, 4 row lanes); grid = (row blocks, column blocks). K partial sums per element.
#define CR_ROWBLOCKS 256
static inline int cr_rowblocks(long long M) { return (int)((M + 63) / 64 < CR_ROWBLOCKS ? (M + 63) / 64 : CR_ROWBLOCKS); }

template <int K, class F>
__device__ __forceinline__ void colreduce_body(long long M, int C, float* __restrict__ part, F f) {
    __shared__ float sh[K][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    const int nrb = gridDim.x;
    const long long rows_per = (M + nrb - 1) / nrb;
    const long long r0 = blockIdx.x * rows_per;
    const long long r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    if (c < C)
        for (long long r = r0 + ty; r < r1; r += 4) f(r, c, acc);
#pragma unroll
    for (int k = 0; k < K; ++k) sh[k][ty][tx] = acc[k];
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            part[((size_t)k * nrb + blockIdx.x) * C + c] = (sh[k][0][tx] + sh[k][1][tx]) + (sh[k][2][tx] + sh[k][3][tx]);
    }
}

// float4 variant: thread = 4 consecutive channels x one row lane; needs C % 4 == 0 and 256 % (C/4) == 0 (C <= 1024).
// 16 B per lane per load (1 KiB per wave instruction), rows of a block reduced through LDS.
// Blocks of CR4_NT = 1024 threads: the number of partial rows (and with it the cost of the final pass) stays at
// CR_ROWBLOCKS while four times as many loads are in flight per CU (256-thread blocks left one wave per SIMD and ran at
// ~2 TB/s).
#define CR4_NT 1024
static inline bool cr4_ok(int C) { return C % 4 == 0 && C >= 4 && C <= 1024 && 256 % (C / 4) == 0; }
template <int K, class F>
__device__ __forceinline__ void colreduce4_body(long long M, int C, float* __restrict__ part, F f) {
    __shared__ float4 sh4[K][CR4_NT];
    const int c4n = C >> 2;
    const int cx = threadIdx.x % c4n, ry = threadIdx.x / c4n, RY = CR4_NT / c4n;
    const int nrb = gridDim.x;
    const long long rows_per = (M + nrb - 1) / nrb;
    const long long r0 = blockIdx.x * rows_per;
    const long long r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
    float4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0 + ry; r < r1; r += RY) f(r, cx, acc);
#pragma unroll
    for (int k = 0; k < K; ++k) sh4[k][threadIdx.x] = acc[k];
    __syncthreads();
    if (ry == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 t = sh4[k][cx];
            for (int j = 1; j < RY; ++j) {
                const float4 u = sh4[k][j * c4n + cx];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *(float4*)(part + ((size_t)k * nrb + blockIdx.x) * C + cx * 4) = t;
        }
    }
}

__global__ __launch_bounds__(CR4_NT) void colsum4_partial_kernel(const float* __restrict__ x, long long M, int C,
                                                              float* __restrict__ part) {
    colreduce4_body<1>(M, C, part, [&](long long r, int cx, float4* acc) {
        const float4 v = *((const float4*)(x + r * C) + cx);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    });
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long long M, int C,
                                                             float* __restrict__ part) {
    colreduce_body<1>(M, C, part, [&](long long r, int c, float* acc) { acc[0] += x[r * C + c]; });
}
// final stage of every column reduction: block = 64 columns x 16 partial lanes, fp64 accumulate
// block = 16 columns x 64 partial lanes (round 4; was 64 x 16: the 512 slabs of a thin weight gradient were walked 32 rows per lane, one
// dependent-latency chain each: 17.7 us for 3.5 MB); fp64 sums of <= a few thousand fp32 values are exact to 1e-16, so the association
// does not show in the rounded result.  store(c, t) writes column c's sum.
template <class Store>
__device__ __forceinline__ void colsum_final_body(const float* __restrict__ part, int nrb, int C, Store store) {
    __shared__ double sh[64][16];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    double s = 0.0;
    if (c < C) {
        int b = ry;
        for (; b + 192 < nrb; b += 256) {
            const float v0 = part[(size_t)b * C + c], v1 = part[(size_t)(b + 64) * C + c];
            const float v2 = part[(size_t)(b + 128) * C + c], v3 = part[(size_t)(b + 192) * C + c];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nrb; b += 64) s += (double)part[(size_t)b * C + c];
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += sh[i][cx];
        store(c, (float)t);
    }
}
// (columns >= C1 go to out2[c - C1] when out2 is given: one launch finishes two results that share their partial rows)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nrb, int C, float beta,
                                                            float* __restrict__ out, int C1 = 0, float* __restrict__ out2 = nullptr) {
    colsum_final_body(part, nrb, C, [&](int c, float t) {
        float* o = (out2 && c >= C1) ? out2 + (c - C1) : out + c;
        *o = (beta == 0.f) ? t : beta * (*o) + t;
    });
}
// the slabs of a thin weight gradient, [tap][s][c] columns, finished straight into the reference layout gradW[O][I][k][k]
// (thin_unpack_grad_kernel's index map: mode 0 = thin-input layer, s = I, c = O; mode 1 = thin-output layer, s = O, c = I), beta = 0;
// columns >= C1 (the ones-column row: a bias gradient) go to out2
__global__ __launch_bounds__(1024) void colsum_final_thin_kernel(const float* __restrict__ part, int nrb, int C, float* __restrict__ gradW,
                                                                 int O, int I, int kk, int mode, int C1, float* __restrict__ out2) {
    colsum_final_body(part, nrb, C, [&](int col, float t) {
        if (col >= C1) { out2[col - C1] = t; return; }
        const int Cs = mode == 0 ? I : O, Cw = mode == 0 ? O : I;
        const int c = col % Cw, ts = col / Cw, s = ts % Cs, tap = ts / Cs;
        const int o = mode == 0 ? c : s, i = mode == 0 ? s : c;
        gradW[((size_t)o * I + i) * kk + tap] = t;
    });
}
// ---- deferred finals (see FgDefer): all jobs of a backward pass in one launch; block -> job by a scan over <= 48 entries
struct FgFinalBatch { FgFinalJob jobs[FG_DEFER_MAX]; int n; };
__global__ __launch_bounds__(1024) void multi_final_kernel(const FgFinalBatch b) {
    __shared__ double sh[16][64];
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.jobs[j + 1].blk0) ++j;
    const FgFinalJob jb = b.jobs[j];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (jb.C == 1) {     // scalar jobs (PReLU slope gradients, up to a few thousand partials): all 1024 threads over the rows
        double s1 = 0.0;
        for (int r = threadIdx.x; r < jb.nrb; r += 1024) s1 += (double)jb.part[r];
        s1 = wave_sum_d(s1);
        if (tx == 0) sh[0][ty] = s1;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += sh[0][i];
            jb.out[0] = (jb.beta == 0.f) ? (float)t : jb.beta * jb.out[0] + (float)t;
        }
        return;
    }
    // round 4: 16 columns x 64 row lanes per block (was 64 x 16: a bias-gradient job of the wave-specialised weight gradient has up
    // to 2 048 partial rows and 128 columns -- two blocks walked 128 rows each, one dependent-latency chain per lane: 42 us of a
    // 4.3 ms step for 1 MB of partials; as the BatchNorm finals of round 2b).  fp64 sums of <= 2 048 fp32 values are exact up to
    // 1e-16, so the result does not depend on the association.
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = ((int)blockIdx.x - jb.blk0) * 16 + cx;
    double s = 0.0;
    if (c < jb.C) {
        int r = ry;
        for (; r + 192 < jb.nrb; r += 256) {       // four independent loads in flight
            const float v0 = jb.part[(size_t)r * jb.C + c], v1 = jb.part[(size_t)(r + 64) * jb.C + c];
            const float v2 = jb.part[(size_t)(r + 128) * jb.C + c], v3 = jb.part[(size_t)(r + 192) * jb.C + c];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; r < jb.nrb; r += 64) s += (double)jb.part[(size_t)r * jb.C + c];
    }
    double* shf = &sh[0][0];                        // [64 row lanes][16 columns]
    shf[ry * 16 + cx] = s;
    __syncthreads();
    if (ry == 0 && c < jb.C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += shf[i * 16 + cx];
        jb.out[c] = (jb.beta == 0.f) ? (float)t : jb.beta * jb.out[c] + (float)t;
    }
}
float* fg_defer_alloc(fg_ctx* ctx, long long floats) {
    FgDefer* d = ctx->defer;
    if (!d || d->n >= FG_DEFER_MAX) return nullptr;
    const long long need = (floats + 3) / 4 * 4;
    if (d->used + need > d->cap) return nullptr;
    float* p = d->arena + d->used;
    d->used += need;
    return p;
}
void fg_defer_push(fg_ctx* ctx, const float* part, int nrb, int C, float beta, float* out) {
    FgDefer* d = ctx->defer;
    FgFinalJob& j = d->jobs[d->n++];
    j.part = part; j.out = out; j.nrb = nrb; j.C = C; j.beta = beta; j.blk0 = d->blocks;
    d->blocks += fg_cdiv(C, 16);
}
int fg_defer_flush(fg_ctx* ctx) {
    FgDefer* d = ctx->defer;
    if (d && d->wn > 0) {
        const int rc = fg_launch_wgrad_finish_jobs(ctx, d->wjobs, d->wn, d->wblocks);
        d->wn = 0; d->wblocks = 0;
        if (rc) { d->n = 0; d->used = 0; d->blocks = 0; return rc; }
    }
    if (!d || d->n == 0) { if (d) { d->used = 0; d->blocks = 0; } return FG_OK; }
    FgFinalBatch b;
    memcpy(b.jobs, d->jobs, sizeof(FgFinalJob) * d->n);
    b.n = d->n;
    hipLaunchKernelGGL(multi_final_kernel, dim3(d->blocks), dim3(1024), 0, ctx->stream, b);
    d->n = 0; d->used = 0; d->blocks = 0;
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_colsum_final(fg_ctx* ctx, const float* part, int nrb, int C, float beta, float* out) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(fg_cdiv(C, 16)), dim3(1024), 0, ctx->stream, part, nrb, C, beta, out, 0, (float*)nullptr);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// bias gradient of a Linear whose output is viewed as [o_c][o_hw] (models.lua:58-59): column sums of gy [M][C] in the device's NHWC
// feature order j = hw * o_c + c, written in the reference's order c * o_hw + hw -- one launch for a few-hundred-row reduction
// (was: partial sums, final, transposition: three launches for 8 192 floats).  fp64 accumulation in row order.
// Block = 64 columns x 4 row lanes (one thread per column walked all M rows: 32 blocks for 8 192 columns, one dependent chain of 128
// loads each -- 11 us for 4 MB): the four lanes' fp64 partial sums (rows r = lane mod 4, ascending) meet in LDS, added in lane order.
__global__ __launch_bounds__(256) void colsum_perm_kernel(const float* __restrict__ x, int M, int C, int o_c, int o_hw, float* __restrict__ out) {
    __shared__ double sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (j < C) {
        int r = ty;
        for (; r + 12 < M; r += 16) {
            const float v0 = x[(size_t)r * C + j], v1 = x[(size_t)(r + 4) * C + j], v2 = x[(size_t)(r + 8) * C + j], v3 = x[(size_t)(r + 12) * C + j];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; r < M; r += 4) s += (double)x[(size_t)r * C + j];
    }
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < C) {
        const double t = ((sh[0][tx] + sh[1][tx]) + sh[2][tx]) + sh[3][tx];
        const int hw = j / o_c, c = j - hw * o_c;
        out[(size_t)c * o_hw + hw] = (float)t;
    }
}
int fg_launch_colsum_perm(fg_ctx* ctx, const float* x, int M, int C, int o_c, int o_hw, float* out) {
    if (C == 0) return FG_OK;
    if ((long long)o_c * o_hw != C) return fg_set_err(ctx, FG_ERR_INVALID, "colsum_perm: %d x %d != %d", o_c, o_hw, C);
    hipLaunchKernelGGL(colsum_perm_kernel, dim3(fg_cdiv(C, 64)), dim3(256), 0, ctx->stream, x, M, C, o_c, o_hw, out);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_colsum_final_thin(fg_ctx* ctx, const float* part, int nrb, float* gradW, int O, int I, int k, int mode, int C2, float* out2) {
    const int C1 = O * I * k * k;
    hipLaunchKernelGGL(colsum_final_thin_kernel, dim3(fg_cdiv(C1 + C2, 16)), dim3(1024), 0, ctx->stream, part, nrb, C1 + C2, gradW, O, I,
                       k * k, mode, C1, out2);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_colsum_final2(fg_ctx* ctx, const float* part, int nrb, int C1, float* out1, int C2, float* out2) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(fg_cdiv(C1 + C2, 16)), dim3(1024), 0, ctx->stream, part, nrb, C1 + C2, 0.f, out1, C1, out2);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// N <= 4 columns (bias gradients of the 3- / 1-channel convolutions): lanes over ROWS, the N sums in registers
template <int N>
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ x, long long M, float* __restrict__ part) {
    __shared__ float sh[N][256];
    float acc[N];
#pragma unroll
    for (int c = 0; c < N; ++c) acc[c] = 0.f;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < N; ++c) acc[c] += x[r * N + c];
    }
#pragma unroll
    for (int c = 0; c < N; ++c) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {          // fixed-shape tree: deterministic
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int c = 0; c < N; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < N) part[(size_t)blockIdx.x * N + threadIdx.x] = sh[threadIdx.x][0];
}
int fg_launch_colsum(fg_ctx* ctx, const float* x, long long M, int N, float beta, float* out, float* scratch) {
    const int nrb = cr_rowblocks(M);
    float* dpart = fg_defer_alloc(ctx, (long long)nrb * N);        // inside fg_net backward: final batched at the end
    if (dpart) scratch = dpart;
    if (N <= 4 && N >= 1) {
        switch (N) {
            case 1: hipLaunchKernelGGL(colsum_small_kernel<1>, dim3(nrb), dim3(256), 0, ctx->stream, x, M, scratch); break;
            case 2: hipLaunchKernelGGL(colsum_small_kernel<2>, dim3(nrb), dim3(256), 0, ctx->stream, x, M, scratch); break;
            case 3: hipLaunchKernelGGL(colsum_small_kernel<3>, dim3(nrb), dim3(256), 0, ctx->stream, x, M, scratch); break;
            default: hipLaunchKernelGGL(colsum_small_kernel<4>, dim3(nrb), dim3(256), 0, ctx->stream, x, M, scratch); break;
        }
    } else
    if (cr4_ok(N))
        hipLaunchKernelGGL(colsum4_partial_kernel, dim3(nrb), dim3(CR4_NT), 0, ctx->stream, x, M, N, scratch);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nrb, fg_cdiv(N, 64)), dim3(256), 0, ctx->stream, x, M, N, scratch);
    FG_CHECK_LAUNCH(ctx);
    if (dpart) { fg_defer_push(ctx, dpart, nrb, N, beta, out); return FG_OK; }
    hipLaunchKernelGGL(colsum_final_kernel, dim3(fg_cdiv(N, 16)), dim3(1024), 0, ctx->stream, scratch, nrb, N, beta, out);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ SpatialBatchNormalization (+PReLU)
// z = ((x - mean) * invstd) * gamma + beta with every operation rounded separately (no FMA contraction): the sign of
// z decides the PReLU branch in forward AND backward, so it must not depend on how the compiler fuses the expression.
__device__ __forceinline__ float bn_xhat(float x, float mu, float is) { return __fmul_rn(__fsub_rn(x, mu), is); }
__device__ __forceinline__ float bn_z(float x, float mu, float is, float g, float b) {
    return __fadd_rn(__fmul_rn(bn_xhat(x, mu, is), g), b);
}
// stats: shifted single pass (pivot = x[0][c]) -> S1 = sum(x-K), S2 = sum((x-K)^2); fp64 finalize.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long long M, int C,
                                                               float* __restrict__ part) {
    colreduce_body<2>(M, C, part, [&](long long r, int c, float* acc) {
        const float d = x[r * C + c] - x[c];
        acc[0] += d;
        acc[1] = fmaf(d, d, acc[1]);
    });
}
__global__ __launch_bounds__(CR4_NT) void bn_stats4_partial_kernel(const float* __restrict__ x, long long M, int C,
                                                                float* __restrict__ part) {
    const float4 piv = *((const float4*)x + threadIdx.x % (C >> 2));
    colreduce4_body<2>(M, C, part, [&](long long r, int cx, float4* acc) {
        const float4 v = *((const float4*)(x + r * C) + cx);
        const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
        acc[0].x += dx; acc[0].y += dy; acc[0].z += dz; acc[0].w += dw;
        acc[1].x = fmaf(dx, dx, acc[1].x); acc[1].y = fmaf(dy, dy, acc[1].y);
        acc[1].z = fmaf(dz, dz, acc[1].z); acc[1].w = fmaf(dw, dw, acc[1].w);
    });
}
// Final stage of the BatchNorm reductions: per-channel fp64 sums over the nrb rows of K partial planes ([K][nrb][C]).
// Block = 16 channels x 64 row lanes (a 64-column block left 2-4 blocks walking up to 1024 rows 16 at a time: 13-15 us for
// 2 MB of partials); a wave holds 4 row lanes of the 16 channels: shuffle-reduce those, then 16 waves through LDS.
// Result valid in threads 0..15 (channel = blockIdx.x * 16 + threadIdx.x).
template <int K>
__device__ __forceinline__ void bn_final_sums(const float* __restrict__ part, int nrb, int C, double (&s)[K]) {
    __shared__ double sh[K][16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    if (c < C)
        for (int b = ty; b < nrb; b += 64) {
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] += (double)part[((size_t)k * nrb + b) * C + c];
        }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        s[k] += __shfl_xor(s[k], 16, 64);
        s[k] += __shfl_xor(s[k], 32, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[k][w][lane] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += sh[k][i][threadIdx.x];
            s[k] = t;
        }
    }
}
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const float* __restrict__ part, const float* __restrict__ x,
                                                              int nrb, long long M, int C, float eps, float momentum,
                                                              float* __restrict__ mean, float* __restrict__ invstd,
                                                              float* __restrict__ rmean, float* __restrict__ rvar) {
    double s[2];
    bn_final_sums<2>(part, nrb, C, s);
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x >= 16 || c >= C) return;
    const double s1 = s[0], s2 = s[1];
    const double n = (double)M;
    const double mu = (double)x[c] + s1 / n;
    double var = (s2 - s1 * s1 / n) / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = var * n / (n > 1.0 ? n - 1.0 : 1.0);
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = (float)(1.0 / sqrt((double)rvar[c] + (double)eps));
}
// apply: the grid is sized so that (gridDim * 256 * 4) % C == 0 -> a thread always owns the same 4 channels and keeps
// their statistics / affine parameters in registers
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       long long total4, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ slope,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd) {
    const float a = slope ? slope[0] : 1.f;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)((i0 * 4) % C);
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    for (long long i = i0; i < total4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = ((const float4*)x)[i];
        float z;
        z = bn_z(v.x, mu.x, is.x, g.x, b.x); v.x = z > 0.f ? z : a * z;
        z = bn_z(v.y, mu.y, is.y, g.y, b.y); v.y = z > 0.f ? z : a * z;
        z = bn_z(v.z, mu.z, is.z, g.z, b.z); v.z = z > 0.f ? z : a * z;
        z = bn_z(v.w, mu.w, is.w, g.w, b.w); v.w = z > 0.f ? z : a * z;
        ((float4*)y)[i] = v;
    }
}
// grid with (blocks * 1024) % C == 0 (C % 4 == 0): blocks = multiple of C / gcd(C, 1024)
static inline int bn_apply_blocks(long long total4, int C) {
    int g = 1024, t = C;
    while (t) { int r = g % t; g = t; t = r; }       // g = gcd(1024, C)
    const int unit = C / g;
    long long want = (total4 + 255) / 256;
    if (want > 4096) want = 4096;
    long long blocks = (want + unit - 1) / unit * unit;
    return (int)(blocks < unit ? unit : blocks);
}
int fg_launch_bn_forward(fg_ctx* ctx, const BnArgs& a) {
    if (a.C % 4) return fg_set_err(ctx, FG_ERR_INVALID, "bn: C %% 4");
    if (a.train && a.stats_part) {
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3(fg_cdiv(a.C, 16)), dim3(1024), 0, ctx->stream, a.stats_part, a.stats_pivot,
                           a.stats_rows, a.M, a.C, a.eps, a.momentum, a.mean, a.invstd, a.running_mean, a.running_var);
        FG_CHECK_LAUNCH(ctx);
    } else if (a.train) {
        const int nrb = cr_rowblocks(a.M);
        if (cr4_ok(a.C))
            hipLaunchKernelGGL(bn_stats4_partial_kernel, dim3(nrb), dim3(CR4_NT), 0, ctx->stream, a.x, a.M, a.C, a.scratch);
        else
            hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nrb, fg_cdiv(a.C, 64)), dim3(256), 0, ctx->stream, a.x, a.M,
                               a.C, a.scratch);
        FG_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3(fg_cdiv(a.C, 16)), dim3(1024), 0, ctx->stream, a.scratch, a.x,
                           nrb, a.M, a.C, a.eps, a.momentum, a.mean, a.invstd, a.running_mean, a.running_var);
        FG_CHECK_LAUNCH(ctx);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(fg_cdiv(a.C, 256)), dim3(256), 0, ctx->stream, a.running_mean,
                           a.running_var, a.C, a.eps, a.mean, a.invstd);
        FG_CHECK_LAUNCH(ctx);
    }
    const long long t4 = a.M * a.C / 4;
    {
        FgProfScope prof(ctx, fg_intern(ctx, "bn_apply_kernel"), 0.0, 0.0, 32.0 * (double)t4);      // read x, write y
        hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_apply_blocks(t4, a.C)), dim3(256), 0, ctx->stream, a.x, a.y, t4, a.C,
                           a.gamma, a.beta, a.slope, a.mean, a.invstd);
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---- sync-BN: per-channel sums leave the GPU as fp64 so that the cross-rank all-reduce + the global
// mean / variance are exact to fp64 rounding.  sync = [sum x (C)] [sum x^2 (C)] [rows (1)]
__global__ __launch_bounds__(1024) void bn_sync_local_kernel(const float* __restrict__ part, const float* __restrict__ x,
                                                             int nrb, long long M, int C, double* __restrict__ sync) {
    __shared__ double sh[2][16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int b = ty; b < nrb; b += 16) {
            s1 += (double)part[(size_t)b * C + c];
            s2 += (double)part[((size_t)nrb + b) * C + c];
        }
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) sync[2 * C] = (double)M;
    if (ty != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1 += sh[0][i][tx]; s2 += sh[1][i][tx]; }
    const double K = (double)x[c], n = (double)M;     // un-shift: sum x = S1 + nK ; sum x^2 = S2 + 2K S1 + n K^2
    sync[c] = s1 + n * K;
    sync[C + c] = s2 + 2.0 * K * s1 + n * K * K;
}
__global__ void bn_sync_global_kernel(const double* __restrict__ sync, int C, float eps, float momentum,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                      float* __restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = sync[2 * C];
    const double mu = sync[c] / n;
    double var = sync[C + c] / n - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = var * n / (n > 1.0 ? n - 1.0 : 1.0);
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}
static int bn_launch_stats_partial(fg_ctx* ctx, const BnArgs& a, int nrb) {
    if (cr4_ok(a.C))
        hipLaunchKernelGGL(bn_stats4_partial_kernel, dim3(nrb), dim3(CR4_NT), 0, ctx->stream, a.x, a.M, a.C, a.scratch);
    else
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nrb, fg_cdiv(a.C, 64)), dim3(256), 0, ctx->stream, a.x, a.M,
                           a.C, a.scratch);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_bn_forward_sync1(fg_ctx* ctx, const BnArgs& a, double* sync) {
    const int nrb = cr_rowblocks(a.M);
    int rc = bn_launch_stats_partial(ctx, a, nrb);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_sync_local_kernel, dim3(fg_cdiv(a.C, 64)), dim3(1024), 0, ctx->stream, a.scratch, a.x, nrb, a.M,
                       a.C, sync);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_bn_forward_sync2(fg_ctx* ctx, const BnArgs& a, const double* sync) {
    hipLaunchKernelGGL(bn_sync_global_kernel, dim3(fg_cdiv(a.C, 256)), dim3(256), 0, ctx->stream, sync, a.C, a.eps,
                       a.momentum, a.mean, a.invstd, a.running_mean, a.running_var);
    FG_CHECK_LAUNCH(ctx);
    const long long t4 = a.M * a.C / 4;
    {
        FgProfScope prof(ctx, fg_intern(ctx, "bn_apply_kernel"), 0.0, 0.0, 32.0 * (double)t4);      // read x, write y
        hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_apply_blocks(t4, a.C)), dim3(256), 0, ctx->stream, a.x, a.y, t4, a.C,
                           a.gamma, a.beta, a.slope, a.mean, a.invstd);
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// backward: dz = PReLU'(z) * gy with z = gamma*xhat + beta; sums: S_dz, S_dz_xhat per channel, S_slope global
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             long long M, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ slope,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ part) {
    const float a = slope ? slope[0] : 1.f;
    colreduce_body<3>(M, C, part, [&](long long r, int c, float* acc) {
        const float xh = bn_xhat(x[r * C + c], mean[c], invstd[c]);
        const float z = __fadd_rn(__fmul_rn(xh, gamma[c]), beta[c]);
        const float g = gy[r * C + c];
        const float dz = z > 0.f ? g : a * g;
        acc[0] += dz;
        acc[1] = fmaf(dz, xh, acc[1]);
        if (!(z > 0.f)) acc[2] = fmaf(z, g, acc[2]);
    });
}
__global__ __launch_bounds__(CR4_NT) void bn_bwd4_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              long long M, int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ slope,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ part) {
    const float a = slope ? slope[0] : 1.f;
    const int cq = (threadIdx.x % (C >> 2)) * 4;
    const float4 mu = *(const float4*)(mean + cq), is = *(const float4*)(invstd + cq);
    const float4 g4 = *(const float4*)(gamma + cq), b4 = *(const float4*)(beta + cq);
    colreduce4_body<3>(M, C, part, [&](long long r, int cx, float4* acc) {
        const float4 xv = *((const float4*)(x + r * C) + cx), gv = *((const float4*)(gy + r * C) + cx);
#define FG_BNB(f)                                                                       \
        {                                                                               \
            const float xh = bn_xhat(xv.f, mu.f, is.f);                                 \
            const float z = __fadd_rn(__fmul_rn(xh, g4.f), b4.f);                       \
            const float dz = z > 0.f ? gv.f : a * gv.f;                                 \
            acc[0].f += dz;                                                             \
            acc[1].f = fmaf(dz, xh, acc[1].f);                                          \
            if (!(z > 0.f)) acc[2].f = fmaf(z, gv.f, acc[2].f);                         \
        }
        FG_BNB(x) FG_BNB(y) FG_BNB(z) FG_BNB(w)
#undef FG_BNB
    });
}
// scratch layout: part[3][nrb][C], then coef[2][C], then slope_part[ceil(C/16)]
// (when slope_part is given every block also leaves the sum of ITS 16 channels' PReLU-slope partials there; the few block
// values are finished by the batched deferred final of the backward pass, or by scalar_final_kernel)
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float* __restrict__ part, int nrb, long long M, int C,
                                                            float* __restrict__ coef, float* __restrict__ ggamma,
                                                            float* __restrict__ gbeta, float acc, float* __restrict__ slope_part) {
    double s[3];
    if (slope_part) bn_final_sums<3>(part, nrb, C, s);
    else { double t[2]; bn_final_sums<2>(part, nrb, C, t); s[0] = t[0]; s[1] = t[1]; s[2] = 0.0; }
    if (threadIdx.x >= 64) return;
    const int c = blockIdx.x * 16 + threadIdx.x;
    const bool own = threadIdx.x < 16 && c < C;
    if (slope_part) {                       // wave 0: lanes 0..15 hold the channel sums
        double v = own ? s[2] : 0.0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) slope_part[blockIdx.x] = (float)v;
    }
    if (!own) return;
    coef[c] = (float)(s[0] / (double)M);
    coef[C + c] = (float)(s[1] / (double)M);
    if (ggamma) ggamma[c] = (acc == 0.f ? 0.f : acc * ggamma[c]) + (float)s[1];
    if (gbeta) gbeta[c] = (acc == 0.f ? 0.f : acc * gbeta[c]) + (float)s[0];
}
__global__ __launch_bounds__(1024) void scalar_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out,
                                                            float acc) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
        out[0] = (acc == 0.f ? 0.f : acc * out[0]) + (float)t;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ gx, long long total4, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ slope, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ coef,
                                                           int train) {
    const float a = slope ? slope[0] : 1.f;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)((i0 * 4) % C);
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const float4 m1 = *(const float4*)(coef + c), m2 = *(const float4*)(coef + C + c);
    for (long long i = i0; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const float4 xv = ((const float4*)x)[i], gv = ((const float4*)gy)[i];
        float4 o;
#define FG_BNA(f)                                                                                   \
        {                                                                                           \
            const float xh = bn_xhat(xv.f, mu.f, is.f);                                             \
            const float z = __fadd_rn(__fmul_rn(xh, g.f), b.f);                                     \
            const float dz = z > 0.f ? gv.f : a * gv.f;                                             \
            o.f = train ? g.f * is.f * (dz - m1.f - xh * m2.f) : g.f * is.f * dz;                   \
        }
        FG_BNA(x) FG_BNA(y) FG_BNA(z) FG_BNA(w)
#undef FG_BNA
        ((float4*)gx)[i] = o;
    }
}
// sync-BN backward: local sums -> fp64 sync buffer [sum dz (C)] [sum dz*xhat (C)] [rows]; parameter gradients are written
// from the LOCAL sums (the later gradient all-reduce makes them global); the coefficients use the reduced sums.
__global__ __launch_bounds__(1024) void bn_bwd_sync_local_kernel(const float* __restrict__ part, int nrb, long long M, int C,
                                                                 double* __restrict__ sync, float* __restrict__ ggamma,
                                                                 float* __restrict__ gbeta, float acc) {
    __shared__ double sh[2][16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int b = ty; b < nrb; b += 16) {
            s1 += (double)part[(size_t)b * C + c];
            s2 += (double)part[((size_t)nrb + b) * C + c];
        }
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) sync[2 * C] = (double)M;
    if (ty != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1 += sh[0][i][tx]; s2 += sh[1][i][tx]; }
    sync[c] = s1;
    sync[C + c] = s2;
    if (ggamma) ggamma[c] = (acc == 0.f ? 0.f : acc * ggamma[c]) + (float)s2;
    if (gbeta) gbeta[c] = (acc == 0.f ? 0.f : acc * gbeta[c]) + (float)s1;
}
__global__ void bn_bwd_sync_global_kernel(const double* __restrict__ sync, int C, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = sync[2 * C];
    coef[c] = (float)(sync[c] / n);
    coef[C + c] = (float)(sync[C + c] / n);
}
int fg_launch_bn_backward_sync1(fg_ctx* ctx, const BnBwdArgs& a, double* sync) {
    const int nrb = cr_rowblocks(a.M);
    float* part = a.scratch;
    if (cr4_ok(a.C))
        hipLaunchKernelGGL(bn_bwd4_partial_kernel, dim3(nrb), dim3(CR4_NT), 0, ctx->stream, a.x, a.gy, a.M, a.C, a.gamma,
                           a.beta, a.slope, a.mean, a.invstd, part);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nrb, fg_cdiv(a.C, 64)), dim3(256), 0, ctx->stream, a.x, a.gy, a.M,
                           a.C, a.gamma, a.beta, a.slope, a.mean, a.invstd, part);
    FG_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(bn_bwd_sync_local_kernel, dim3(fg_cdiv(a.C, 64)), dim3(1024), 0, ctx->stream, part, nrb, a.M, a.C,
                       sync, a.ggamma, a.gbeta, a.gbeta_acc);
    FG_CHECK_LAUNCH(ctx);
    if (a.slope && a.gslope) {
        hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, part + (size_t)2 * nrb * a.C,
                           nrb * a.C, a.gslope, a.gbeta_acc);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}
int fg_launch_bn_backward_sync2(fg_ctx* ctx, const BnBwdArgs& a, const double* sync) {
    const int nrb = cr_rowblocks(a.M);
    float* coef = a.scratch + (size_t)3 * nrb * a.C;
    hipLaunchKernelGGL(bn_bwd_sync_global_kernel, dim3(fg_cdiv(a.C, 256)), dim3(256), 0, ctx->stream, sync, a.C, coef);
    FG_CHECK_LAUNCH(ctx);
    if (a.gx) {
        const long long t4 = a.M * a.C / 4;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_apply_blocks(t4, a.C)), dim3(256), 0, ctx->stream, a.x, a.gy, a.gx,
                           t4, a.C, a.gamma, a.beta, a.slope, a.mean, a.invstd, coef, 1);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}

int fg_launch_bn_backward(fg_ctx* ctx, const BnBwdArgs& a) {
    if (a.C % 4) return fg_set_err(ctx, FG_ERR_INVALID, "bn: C %% 4");
    const int nrb = cr_rowblocks(a.M);
    const int ncb = fg_cdiv(a.C, 64);
    float* part = a.scratch;
    float* coef = a.scratch + (size_t)3 * nrb * a.C;
    {
        FgProfScope prof(ctx, fg_intern(ctx, "bn_bwd_partial_kernel"), 0.0, 0.0, 8.0 * (double)a.M * a.C);     // read x, gy
        if (cr4_ok(a.C))
            hipLaunchKernelGGL(bn_bwd4_partial_kernel, dim3(nrb), dim3(CR4_NT), 0, ctx->stream, a.x, a.gy, a.M, a.C, a.gamma,
                               a.beta, a.slope, a.mean, a.invstd, part);
        else
            hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nrb, ncb), dim3(256), 0, ctx->stream, a.x, a.gy, a.M, a.C, a.gamma,
                               a.beta, a.slope, a.mean, a.invstd, part);
    }
    FG_CHECK_LAUNCH(ctx);
    float* gs = (a.slope && a.gslope) ? a.gslope : nullptr;
    const int nfb = fg_cdiv(a.C, 16);
    float* dpart = gs ? fg_defer_alloc(ctx, nfb) : nullptr;          // inside fg_net backward: finished by the batched final
    float* spart = gs ? (dpart ? dpart : coef + (size_t)2 * a.C) : nullptr;
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(nfb), dim3(1024), 0, ctx->stream, part, nrb, a.M, a.C,
                       coef, a.ggamma, a.gbeta, a.gbeta_acc, spart);
    FG_CHECK_LAUNCH(ctx);
    if (dpart) fg_defer_push(ctx, dpart, nfb, 1, a.gbeta_acc, gs);
    else if (gs) {
        hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, spart, nfb, gs, a.gbeta_acc);
        FG_CHECK_LAUNCH(ctx);
    }
    if (a.gx) {
        const long long t4 = a.M * a.C / 4;
        {
            FgProfScope prof(ctx, fg_intern(ctx, "bn_bwd_apply_kernel"), 0.0, 0.0, 48.0 * (double)t4);      // read x, gy; write gx
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_apply_blocks(t4, a.C)), dim3(256), 0, ctx->stream, a.x, a.gy, a.gx, t4, a.C,
                               a.gamma, a.beta, a.slope, a.mean, a.invstd, coef, 1);
        }
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}

// ------------------------------------------------------------------ PReLU (+ same-shape mask)
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                        const float* __restrict__ mask, float mscale,
                                                        float* __restrict__ y, long long n) {
    const float a = slope[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r = v > 0.f ? v : a * v;
        if (mask) r *= mask[i] * mscale;
        y[i] = r;
    }
}
int fg_launch_prelu_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y,
                            long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(prelu_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, slope, mask, mscale, y, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ slope, const float* __restrict__ mask,
                                                        float mscale, float* __restrict__ gx, float* __restrict__ part,
                                                        long long n) {
    __shared__ float sh[4];
    const float a = slope[0];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float g = gy[i];
        if (mask) g *= mask[i] * mscale;
        if (gx) gx[i] = v > 0.f ? g : a * g;
        if (!(v > 0.f)) s = fmaf(v, g, s);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
int fg_launch_prelu_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask,
                             float mscale, float* gx, float* gslope, float acc, long long n, float* scratch) {
    if (n == 0) return FG_OK;
    dim3 grid = FG_GRID(n, 256);
    if (grid.x > 1024) grid.x = 1024;
    float* dpart = gslope ? fg_defer_alloc(ctx, grid.x) : nullptr;
    if (dpart) scratch = dpart;
    hipLaunchKernelGGL(prelu_bwd_kernel, grid, dim3(256), 0, ctx->stream, x, gy, slope, mask, mscale, gx, scratch, n);
    FG_CHECK_LAUNCH(ctx);
    if (dpart) { fg_defer_push(ctx, dpart, (int)grid.x, 1, acc, gslope); return FG_OK; }
    if (gslope) {
        hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, scratch, (int)grid.x, gslope, acc);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}

// ------------------------------------------------------------------ PReLU -> SpatialDropout -> AvgPool(2,2,2,2)
// element index -> (channel quad, x, y, sample) of an [B][H2][W2][C4] walk.  32-bit unsigned divisions wherever the index fits (every
// size of both workloads): a 64-bit division by a run-time divisor is ~100 instructions, and the pooling kernels did six per 16-byte
// result (scripts/ubench/store_width.hip's first version measured exactly that: a "2 TB/s store pattern" that was its index arithmetic)
__device__ __forceinline__ void fg_decode_quad(long long i, int C4, int W2, int H2, int& c4, int& w2, int& h2, int& b) {
    if (i <= 0xFFFFFFFFLL) {
        unsigned t = (unsigned)i;
        c4 = (int)(t % (unsigned)C4); t /= (unsigned)C4;
        w2 = (int)(t % (unsigned)W2); t /= (unsigned)W2;
        h2 = (int)(t % (unsigned)H2);
        b = (int)(t / (unsigned)H2);
    } else {
        c4 = (int)(i % C4);
        long long t = i / C4;
        w2 = (int)(t % W2); t /= W2;
        h2 = (int)(t % H2);
        b = (int)(t / H2);
    }
}
__device__ __forceinline__ float4 fg_sum_parts4(const FgSplitParts& sp, size_t i4, int c) {   // i4: float4 index, c: its channel
    float4 v = ((const float4*)sp.part)[i4];
    int k = 1;
    for (; k + 2 < sp.splits; k += 3) {          // three loads in flight; the additions stay in split order (bit-identical)
        const float4 t0 = ((const float4*)(sp.part + k * sp.stride))[i4], t1 = ((const float4*)(sp.part + (k + 1) * sp.stride))[i4];
        const float4 t2 = ((const float4*)(sp.part + (k + 2) * sp.stride))[i4];
        v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w;
        v.x += t1.x; v.y += t1.y; v.z += t1.z; v.w += t1.w;
        v.x += t2.x; v.y += t2.y; v.z += t2.z; v.w += t2.w;
    }
    for (; k < sp.splits; ++k) {
        const float4 t = ((const float4*)(sp.part + k * sp.stride))[i4];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (sp.bias) { v.x += sp.bias[c]; v.y += sp.bias[c + 1]; v.z += sp.bias[c + 2]; v.w += sp.bias[c + 3]; }
    return v;
}
__global__ __launch_bounds__(256) void actpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                          const float* __restrict__ mask, float mscale,
                                                          float* __restrict__ y, int B, int H, int W, int C,
                                                          const FgSplitParts sp, float* __restrict__ xout) {
    const float a = slope ? slope[0] : 1.f;
    const int H2 = H >> 1, W2 = W >> 1, C4 = C >> 2;
    const long long total = (long long)B * H2 * W2 * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4, w2, h2, b;
        fg_decode_quad(i, C4, W2, H2, c4, w2, h2, b);
        const float4* px = (const float4*)(x + (((size_t)b * H + 2 * h2) * W + 2 * w2) * C) + c4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float4 v;
                if (sp.splits) {       // the producing layer's split-K partials: finish the pre-activation here (and keep it)
                    const size_t i4 = ((((size_t)b * H + 2 * h2 + dy) * W + 2 * w2 + dx) * C4) + c4;
                    v = fg_sum_parts4(sp, i4, c4 * 4);
                    ((float4*)xout)[i4] = v;
                } else v = px[((size_t)dy * W + dx) * C4];
                s.x += v.x > 0.f ? v.x : a * v.x;
                s.y += v.y > 0.f ? v.y : a * v.y;
                s.z += v.z > 0.f ? v.z : a * v.z;
                s.w += v.w > 0.f ? v.w : a * v.w;
            }
        float4 m = make_float4(mscale, mscale, mscale, mscale);
        if (mask) {
            const float4 mk = *(const float4*)(mask + (size_t)b * C + c4 * 4);
            m.x *= mk.x; m.y *= mk.y; m.z *= mk.z; m.w *= mk.w;
        }
        s.x *= 0.25f * m.x; s.y *= 0.25f * m.y; s.z *= 0.25f * m.z; s.w *= 0.25f * m.w;
        ((float4*)y)[i] = s;
    }
}
int fg_launch_actpool_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale,
                              float* y, int B, int H, int W, int C, const FgSplitParts* sp, float* xout) {
    if (C % 4 || H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "actpool: C%%4, even H/W");
    long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
    if (n == 0) return FG_OK;
    FgSplitParts none; memset(&none, 0, sizeof(none));
    if (sp && sp->splits && (!xout || sp->stride % 4 || sp->N != C)) return fg_set_err(ctx, FG_ERR_INVALID, "actpool: bad split partials");
    {
        // read x (or `splits` partials, then also write x), write the pooled y
        const double nx = 4.0 * B * H * W * C, sx = (sp && sp->splits) ? (double)sp->splits + 1.0 : 1.0;
        FgProfScope prof(ctx, fg_intern(ctx, "actpool_fwd_kernel"), 0.0, 0.0, nx * sx + nx / 4);
        hipLaunchKernelGGL(actpool_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, slope, mask, mscale, y, B, H,
                           W, C, (sp && sp->splits) ? *sp : none, xout);
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ __launch_bounds__(256) void actpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ slope, const float* __restrict__ mask,
                                                          float mscale, float* __restrict__ gx, float* __restrict__ part,
                                                          int B, int H, int W, int C, const FgSplitParts sp) {
    __shared__ float sh[4];
    const float a = slope ? slope[0] : 1.f;
    const int H2 = H >> 1, W2 = W >> 1, C4 = C >> 2;
    const long long total = (long long)B * H * W * C4;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4, w, h, b;
        fg_decode_quad(i, C4, W, H, c4, w, h, b);
        const float4 v = ((const float4*)x)[i];
        const size_t gi4 = (((size_t)b * H2 + (h >> 1)) * W2 + (w >> 1)) * C4 + c4;
        float4 g = sp.splits ? fg_sum_parts4(sp, gi4, c4 * 4) : ((const float4*)gy)[gi4];      // pooled gradient (or its partials)
        float4 m = make_float4(mscale, mscale, mscale, mscale);
        if (mask) {
            const float4 mk = *(const float4*)(mask + (size_t)b * C + c4 * 4);
            m.x *= mk.x; m.y *= mk.y; m.z *= mk.z; m.w *= mk.w;
        }
        g.x *= 0.25f * m.x; g.y *= 0.25f * m.y; g.z *= 0.25f * m.z; g.w *= 0.25f * m.w;
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x; if (!(v.x > 0.f)) s = fmaf(v.x, g.x, s);
        o.y = v.y > 0.f ? g.y : a * g.y; if (!(v.y > 0.f)) s = fmaf(v.y, g.y, s);
        o.z = v.z > 0.f ? g.z : a * g.z; if (!(v.z > 0.f)) s = fmaf(v.z, g.z, s);
        o.w = v.w > 0.f ? g.w : a * g.w; if (!(v.w > 0.f)) s = fmaf(v.w, g.w, s);
        if (gx) ((float4*)gx)[i] = o;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
int fg_launch_actpool_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask,
                               float mscale, float* gx, float* gslope, float acc, int B, int H, int W, int C,
                               float* scratch, const FgSplitParts* sp) {
    FgSplitParts spv; memset(&spv, 0, sizeof(spv));
    if (sp && sp->splits) { spv = *sp; if (spv.stride % 4 || spv.bias) return fg_set_err(ctx, FG_ERR_INVALID, "actpool: bad split partials"); }
    if (C % 4 || H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "actpool: C%%4, even H/W");
    long long n = (long long)B * H * W * (C / 4);
    if (n == 0) return FG_OK;
    dim3 grid = FG_GRID(n, 256);
    if (grid.x > 1024) grid.x = 1024;
    float* dpart = (slope && gslope) ? fg_defer_alloc(ctx, grid.x) : nullptr;
    if (dpart) scratch = dpart;
    {
        // read x and the pooled gradient (or its `splits` partials), write gx
        const double nx = 4.0 * B * H * W * C;
        FgProfScope prof(ctx, fg_intern(ctx, "actpool_bwd_kernel"), 0.0, 0.0, nx * (gx ? 2.0 : 1.0) + nx / 4 * (spv.splits ? spv.splits : 1));
        hipLaunchKernelGGL(actpool_bwd_kernel, grid, dim3(256), 0, ctx->stream, x, gy, slope, mask, mscale, gx, scratch, B,
                           H, W, C, spv);
    }
    FG_CHECK_LAUNCH(ctx);
    if (dpart) { fg_defer_push(ctx, dpart, (int)grid.x, 1, acc, gslope); return FG_OK; }
    if (slope && gslope) {
        hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, scratch, (int)grid.x, gslope, acc);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}

// ------------------------------------------------------------------ standalone module pieces
__global__ void scale_mask_nc_kernel(const float* __restrict__ x, const float* __restrict__ mask, float mscale,
                                     float* __restrict__ y, int B, int HW, int C) {
    const long long total = (long long)B * HW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long long)HW * C));
        y[i] = x[i] * (mask ? mask[(size_t)b * C + c] * mscale : mscale);
    }
}
int fg_launch_scale_mask_nc(fg_ctx* ctx, const float* x, const float* mask, float mscale, float* y, int B, int HW,
                            int C) {
    long long n = (long long)B * HW * C;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(scale_mask_nc_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, mask, mscale, y, B, HW, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)B * H2 * W2 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w2 = (int)(t % W2); t /= W2;
        const int h2 = (int)(t % H2);
        const int b = (int)(t / H2);
        const float* p = x + (((size_t)b * H + 2 * h2) * W + 2 * w2) * C + c;
        y[i] = 0.25f * ((p[0] + p[C]) + (p[(size_t)W * C] + p[(size_t)W * C + C]));
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B, int H, int W, int C) {
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)B * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        gx[i] = 0.25f * gy[(((size_t)b * H2 + (h >> 1)) * W2 + (w >> 1)) * C + c];
    }
}
int fg_launch_avgpool_forward(fg_ctx* ctx, const float* x, float* y, int B, int H, int W, int C) {
    long long n = (long long)B * (H / 2) * (W / 2) * C;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(avgpool_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, y, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_avgpool_backward(fg_ctx* ctx, const float* gy, float* gx, int B, int H, int W, int C) {
    long long n = (long long)B * H * W * C;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(avgpool_bwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, gy, gx, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const int H2 = H * 2, W2 = W * 2;
    const long long total = (long long)B * H2 * W2 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % W2); t /= W2;
        const int h = (int)(t % H2);
        const int b = (int)(t / H2);
        y[i] = x[(((size_t)b * H + (h >> 1)) * W + (w >> 1)) * C + c];
    }
}
__global__ void upsample_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B, int H, int W, int C) {
    const int W2 = W * 2;
    const long long total = (long long)B * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        const float* p = gy + (((size_t)b * 2 * H + 2 * h) * W2 + 2 * w) * C + c;
        gx[i] = (p[0] + p[C]) + (p[(size_t)W2 * C] + p[(size_t)W2 * C + C]);
    }
}
int fg_launch_upsample_forward(fg_ctx* ctx, const float* x, float* y, int B, int H, int W, int C) {
    long long n = (long long)B * H * W * C * 4;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(upsample_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, y, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_upsample_backward(fg_ctx* ctx, const float* gy, float* gx, int B, int H, int W, int C) {
    long long n = (long long)B * H * W * C;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(upsample_bwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, gy, gx, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// cudnn.SpatialConvolutionUpsample with factor f > 1 (layers/cudnnSpatialConvolutionUpsample.lua:19-31): the convolution to
// nOut * f * f planes is followed by a FLAT re-view of its contiguous NCHW output [N][nOut*f*f][h][w] as [N][nOut][h*f][w*f]
// (Tensor:view -- not a pixel shuffle).  With NHWC activations inside, the view is an index map: element j of a sample's flat
// NCHW order is (co, y, x) = (j / hw, (j % hw) / w, j % w) in the convolution's output and (cu, Y, X) = (j / (hf*wf), ...) in
// the viewed tensor.  dir = 0: V[n][y][x][co] -> U[n][Y][X][cu] (forward); dir = 1: gU -> gV (updateGradInput /
// accGradParameters view the gradient back, :34-58).  One thread per DESTINATION element.
__global__ void nchw_review_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int h, int w, int C, int f, int dir) {
    const int cu_n = C / (f * f), hf = h * f, wf = w * f;
    const long long per = (long long)C * h * w, total = (long long)B * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per;
        long long r = i % per;
        long long j, so;
        if (dir == 0) {               // destination U, NHWC [hf][wf][cu]
            const int cu = (int)(r % cu_n); r /= cu_n;
            const int X = (int)(r % wf), Y = (int)(r / wf);
            j = ((long long)cu * hf + Y) * wf + X;
            const int co = (int)(j / ((long long)h * w)), q = (int)(j % ((long long)h * w));
            so = ((long long)q) * C + co;                              // V NHWC: [y*w + x][co]
        } else {                      // destination gV, NHWC [h][w][co]
            const int co = (int)(r % C); r /= C;
            j = (long long)co * h * w + r;                             // r = y*w + x
            const int cu = (int)(j / ((long long)hf * wf)), q = (int)(j % ((long long)hf * wf));
            so = ((long long)q) * cu_n + cu;                           // gU NHWC: [Y*wf + X][cu]
        }
        dst[i] = src[n * per + so];
    }
}
int fg_launch_nchw_review(fg_ctx* ctx, const float* src, float* dst, int B, int h, int w, int C, int f, int dir) {
    const long long n = (long long)B * C * h * w;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(nchw_review_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, src, dst, B, h, w, C, f, dir);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx,
                                   long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = y[i];
        gx[i] = gy[i] * v * (1.f - v);
    }
}
int fg_launch_sigmoid_forward(fg_ctx* ctx, const float* x, float* y, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(sigmoid_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, y, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_sigmoid_backward(fg_ctx* ctx, const float* y, const float* gy, float* gx, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, y, gy, gx, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// LeakyReLU.lua:13-31: y = (|x|+x)/2 + (-s/2)(|x|-x); backward: x >= 0 -> gy, else s*gy
__global__ void leakyrelu_fwd_kernel(const float* __restrict__ x, float s, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i], av = fabsf(v);
        y[i] = (av + v) * 0.5f + (av - v) * (-s * 0.5f);
    }
}
__global__ void leakyrelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float s,
                                     float* __restrict__ gx, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        gx[i] = x[i] >= 0.f ? gy[i] : s * gy[i];
}
int fg_launch_leakyrelu_forward(fg_ctx* ctx, const float* x, float s, float* y, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(leakyrelu_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, s, y, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_leakyrelu_backward(fg_ctx* ctx, const float* x, const float* gy, float s, float* gx, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(leakyrelu_bwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, gy, s, gx, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ SpatialMaxPooling(2,2) (models_c2f.lua:251, 256)
// ties: first max in scan order (dy, dx); backward recomputes the argmax from the saved input (no index tensor)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                          int W, int C) {
    const int H2 = H >> 1, W2 = W >> 1, C4 = C >> 2;
    const long long total = (long long)B * H2 * W2 * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4, w2, h2, b;
        fg_decode_quad(i, C4, W2, H2, c4, w2, h2, b);
        const float4* px = (const float4*)(x + (((size_t)b * H + 2 * h2) * W + 2 * w2) * C) + c4;
        float4 m = px[0];
        const float4 v1 = px[C4], v2 = px[(size_t)W * C4], v3 = px[(size_t)W * C4 + C4];
        m.x = fmaxf(fmaxf(m.x, v1.x), fmaxf(v2.x, v3.x)); m.y = fmaxf(fmaxf(m.y, v1.y), fmaxf(v2.y, v3.y));
        m.z = fmaxf(fmaxf(m.z, v1.z), fmaxf(v2.z, v3.z)); m.w = fmaxf(fmaxf(m.w, v1.w), fmaxf(v2.w, v3.w));
        ((float4*)y)[i] = m;
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gx, int B, int H, int W, int C) {
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)B * H2 * W2 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w2 = (int)(t % W2); t /= W2;
        const int h2 = (int)(t % H2);
        const int b = (int)(t / H2);
        const size_t base = (((size_t)b * H + 2 * h2) * W + 2 * w2) * C + c;
        const size_t o[4] = {base, base + C, base + (size_t)W * C, base + (size_t)W * C + C};
        int am = 0;
        float m = x[o[0]];
#pragma unroll
        for (int k = 1; k < 4; ++k) { const float v = x[o[k]]; if (v > m) { m = v; am = k; } }
        const float g = gy[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) gx[o[k]] = (k == am) ? g : 0.f;
    }
}
// maxpool backward + the backward of the PReLU in front of it: the pooled tensor was prelu(xpre) (re-evaluated with
// prelu_fwd_kernel's expression, so the argmax is the forward's), gx = gradient wrt xpre, slope-gradient partials per block
// (mask / mscale: the nn.Dropout BEHIND the pool, on the pooled tensor: its backward is one multiply of the incoming gradient)
__global__ __launch_bounds__(256) void maxpool_prelu_bwd_kernel(const float* __restrict__ xpre, const float* __restrict__ gy,
                                                                const float* __restrict__ slope, float* __restrict__ gx,
                                                                float* __restrict__ part, int B, int H, int W, int C,
                                                                const float* __restrict__ mask, float mscale) {
    __shared__ float sh[4];
    const float a = slope[0];
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)B * H2 * W2 * C;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w2 = (int)(t % W2); t /= W2;
        const int h2 = (int)(t % H2);
        const int b = (int)(t / H2);
        const size_t base = (((size_t)b * H + 2 * h2) * W + 2 * w2) * C + c;
        const size_t o[4] = {base, base + C, base + (size_t)W * C, base + (size_t)W * C + C};
        float xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = xpre[o[k]];
        int am = 0;
        float m = xv[0] > 0.f ? xv[0] : a * xv[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) { const float v = xv[k] > 0.f ? xv[k] : a * xv[k]; if (v > m) { m = v; am = k; } }
        const float g = mask ? gy[i] * (mask[i] * mscale) : gy[i];           // mul_mask_kernel's expression
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool pos = xv[k] > 0.f;
            const float gk = (k == am) ? g : 0.f;
            if (gx) gx[o[k]] = pos ? gk : a * gk;           // (null: the stage is the net's first and no input gradient was asked for)
            if (k == am && !pos) s = fmaf(xv[k], g, s);
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0 && part) part[blockIdx.x] = s;
}
int fg_launch_maxpool_prelu_backward(fg_ctx* ctx, const float* xpre, const float* gy, const float* slope, float* gx,
                                     float* gslope, int B, int H, int W, int C, float* scratch, const float* mask, float mscale) {
    if (H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "maxpool: even H/W");
    long long n = (long long)B * (H / 2) * (W / 2) * C;
    if (n == 0) return FG_OK;
    dim3 grid = FG_GRID(n, 256);
    if (grid.x > 1024) grid.x = 1024;
    float* dpart = gslope ? fg_defer_alloc(ctx, grid.x) : nullptr;
    float* part = dpart ? dpart : (gslope ? scratch : nullptr);
    hipLaunchKernelGGL(maxpool_prelu_bwd_kernel, grid, dim3(256), 0, ctx->stream, xpre, gy, slope, gx, part, B, H, W, C, mask, mscale);
    FG_CHECK_LAUNCH(ctx);
    if (dpart) { fg_defer_push(ctx, dpart, (int)grid.x, 1, 0.f, gslope); return FG_OK; }
    if (gslope) {
        hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, (int)grid.x, gslope, 0.f);
        FG_CHECK_LAUNCH(ctx);
    }
    return FG_OK;
}
// PReLU -> SpatialMaxPooling(2, 2) [-> Dropout] forward in one pass over the pre-activation: y = max over the window of
// prelu_fwd_kernel's expression (the same one maxpool_prelu_bwd_kernel re-evaluates for its argmax), times mul_mask_kernel's
// mask[i] * mscale.  Replaces a second full-resolution store in the producing layer's epilogue + maxpool_fwd + mul_mask.
__global__ __launch_bounds__(256) void actmaxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                             const float* __restrict__ mask, float mscale, float* __restrict__ y,
                                                             int B, int H, int W, int C) {
    const float a = slope[0];
    const int H2 = H >> 1, W2 = W >> 1, C4 = C >> 2;
    const long long total = (long long)B * H2 * W2 * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c4, w2, h2, b;
        fg_decode_quad(i, C4, W2, H2, c4, w2, h2, b);
        const float4* px = (const float4*)(x + (((size_t)b * H + 2 * h2) * W + 2 * w2) * C) + c4;
        const float4 v0 = px[0], v1 = px[C4], v2 = px[(size_t)W * C4], v3 = px[(size_t)W * C4 + C4];
        auto act = [a](float v) { return v > 0.f ? v : a * v; };
        float4 m;
        m.x = fmaxf(fmaxf(act(v0.x), act(v1.x)), fmaxf(act(v2.x), act(v3.x)));
        m.y = fmaxf(fmaxf(act(v0.y), act(v1.y)), fmaxf(act(v2.y), act(v3.y)));
        m.z = fmaxf(fmaxf(act(v0.z), act(v1.z)), fmaxf(act(v2.z), act(v3.z)));
        m.w = fmaxf(fmaxf(act(v0.w), act(v1.w)), fmaxf(act(v2.w), act(v3.w)));
        if (mask) {
            const float4 k = ((const float4*)mask)[i];
            m.x = m.x * (k.x * mscale); m.y = m.y * (k.y * mscale); m.z = m.z * (k.z * mscale); m.w = m.w * (k.w * mscale);
        }
        ((float4*)y)[i] = m;
    }
}
int fg_launch_actmaxpool_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y, int B,
                                 int H, int W, int C) {
    if (C % 4 || H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "actmaxpool: C%%4, even H/W");
    const long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
    if (n == 0) return FG_OK;
    {
        const double nx = 4.0 * B * H * W * C;
        FgProfScope prof(ctx, fg_intern(ctx, "actmaxpool_fwd_kernel"), 0.0, 0.0, nx + nx / 4 * (mask ? 2.0 : 1.0));
        hipLaunchKernelGGL(actmaxpool_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, slope, mask, mscale, y, B, H, W, C);
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_maxpool_forward(fg_ctx* ctx, const float* x, float* y, int B, int H, int W, int C) {
    if (C % 4 || H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "maxpool: C%%4, even H/W");
    long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(maxpool_fwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, y, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_maxpool_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, int B, int H, int W, int C) {
    if (H % 2 || W % 2) return fg_set_err(ctx, FG_ERR_INVALID, "maxpool: even H/W");
    long long n = (long long)B * (H / 2) * (W / 2) * C;
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(maxpool_bwd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, gy, gx, B, H, W, C);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// y = x * mask * scale (elementwise, nn.Dropout on any shape; mask nullptr -> y = x * scale)
__global__ void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, float scale,
                                float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = x[i] * (mask ? mask[i] * scale : scale);
}
int fg_launch_mul_mask(fg_ctx* ctx, const float* x, const float* mask, float scale, float* y, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(mul_mask_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, x, mask, scale, y, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// nn.JoinTable(2,2) on NHWC: out[pix][0..ca) = a, [ca..ca+cb) = b ; nn.CAddTable: out = a + b
__global__ void concat_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                              long long npix, int ca, int cb) {
    const int c = ca + cb;
    const long long total = npix * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % c);
        const long long p = i / c;
        out[i] = k < ca ? a[p * ca + k] : b[p * cb + (k - ca)];
    }
}
__global__ void split_kernel(const float* __restrict__ g, float* __restrict__ ga, float* __restrict__ gb,
                             long long npix, int ca, int cb) {
    const int c = ca + cb;
    const long long total = npix * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % c);
        const long long p = i / c;
        if (k < ca) { if (ga) ga[p * ca + k] = g[i]; }
        else if (gb) gb[p * cb + (k - ca)] = g[i];
    }
}
int fg_launch_concat(fg_ctx* ctx, const float* a, const float* b, float* out, long long npix, int ca, int cb) {
    if (npix == 0) return FG_OK;
    hipLaunchKernelGGL(concat_kernel, FG_GRID(npix * (ca + cb), 256), dim3(256), 0, ctx->stream, a, b, out, npix, ca, cb);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_split(fg_ctx* ctx, const float* g, float* ga, float* gb, long long npix, int ca, int cb) {
    if (npix == 0) return FG_OK;
    hipLaunchKernelGGL(split_kernel, FG_GRID(npix * (ca + cb), 256), dim3(256), 0, ctx->stream, g, ga, gb, npix, ca, cb);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
int fg_launch_add(fg_ctx* ctx, const float* a, const float* b, float* out, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(add_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, a, b, out, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// plain device copy (hipMemcpyAsync's blit kernel took ~0.4 ms for a 3 MB batch half on this stack: a 7 % tax on the c2f step)
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long n4 = n >> 2;
    const bool al = (((size_t)src | (size_t)dst) & 15) == 0;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (al) {
        for (long long i = i0; i < n4; i += step) ((float4*)dst)[i] = ((const float4*)src)[i];
        for (long long i = (n4 << 2) + i0; i < n; i += step) dst[i] = src[i];
    } else
        for (long long i = i0; i < n; i += step) dst[i] = src[i];
}
int fg_launch_copy(fg_ctx* ctx, const float* src, float* dst, long long n) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(copy_kernel, FG_GRID((n + 3) / 4, 256), dim3(256), 0, ctx->stream, src, dst, n);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// out = [a0 + b0 | a1 + b1]: nn.CAddTable over a batch assembled from two halves (adversarial_c2f.lua:125-150)
__global__ __launch_bounds__(256) void add_halves_kernel(const float* __restrict__ a0, const float* __restrict__ b0,
                                                         const float* __restrict__ a1, const float* __restrict__ b1,
                                                         float* __restrict__ out, long long nh) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nh; i += (long long)gridDim.x * blockDim.x)
        out[i] = i < nh ? a0[i] + b0[i] : a1[i - nh] + b1[i - nh];
}
int fg_launch_add_halves(fg_ctx* ctx, const float* a0, const float* b0, const float* a1, const float* b1, float* out, long long nh) {
    if (nh == 0) return FG_OK;
    hipLaunchKernelGGL(add_halves_kernel, FG_GRID(2 * nh, 256), dim3(256), 0, ctx->stream, a0, b0, a1, b1, out, nh);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// stride-2 data gradient helper: out[b][2y][2x][c] = g[b][y][x][c], every other position 0 (out is [B][2H][2W][C])
__global__ void zero_insert2_kernel(const float* __restrict__ g, float* __restrict__ out, int H, int W, int C4, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int X = (int)(t % (2 * W)); t /= 2 * W;
    const int Y = (int)(t % (2 * H));
    const long long b = t / (2 * H);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((X | Y) & 1) == 0) v = ((const float4*)g)[((b * H + (Y >> 1)) * W + (X >> 1)) * C4 + c];
    ((float4*)out)[i] = v;
}
int fg_launch_zero_insert2(fg_ctx* ctx, const float* g, float* out, int B, int H, int W, int C) {
    if (C % 4) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "zero_insert2: C %% 4");
    const long long n4 = (long long)B * 2 * H * 2 * W * (C / 4);
    if (n4 == 0) return FG_OK;
    hipLaunchKernelGGL(zero_insert2_kernel, FG_GRID(n4, 256), dim3(256), 0, ctx->stream, g, out, H, W, C / 4, n4);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ Linear(K -> 1) [+ Sigmoid]
__global__ __launch_bounds__(64) void gemv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y, int B, int K,
                                                      int sigmoid) {
    const int row = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) s = fmaf(x[(size_t)row * K + k], w[k], s);
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        s += b[0];
        y[row] = sigmoid ? 1.f / (1.f + expf(-s)) : s;
    }
}
int fg_launch_gemv_forward(fg_ctx* ctx, const float* x, const float* w, const float* b, float* y, int B, int K,
                           int sigmoid) {
    if (B == 0) return FG_OK;
    hipLaunchKernelGGL(gemv_fwd_kernel, dim3(B), dim3(64), 0, ctx->stream, x, w, b, y, B, K, sigmoid);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
// dl[b] = gy[b]*y(1-y) (or gy); gx[b][k] = dl[b] w[k]; gw[k] = sum_b dl[b] x[b][k]; gb = sum dl.
// block = 64 columns x 4 batch lanes, grid over K; block 0 also reduces the bias gradient.
__global__ __launch_bounds__(256) void gemv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ y, const float* __restrict__ gy,
                                                       float* __restrict__ gx, float* __restrict__ gw,
                                                       float* __restrict__ gb, float acc, int B, int K, int sigmoid) {
    extern __shared__ float dl[];  // [B] then [4][64] partials
    float* red = dl + B;
    __shared__ float sh[4];
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float v = sigmoid ? y[b] : 0.f;
        dl[b] = sigmoid ? gy[b] * v * (1.f - v) : gy[b];
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (k < K) {
        const float wk = gx ? w[k] : 0.f;
        for (int b = ty; b < B; b += 4) {
            if (gw) s = fmaf(dl[b], x[(size_t)b * K + k], s);
            if (gx) gx[(size_t)b * K + k] = dl[b] * wk;
        }
    }
    red[ty * 64 + tx] = s;
    __syncthreads();
    if (ty == 0 && k < K && gw) {
        const float t = (red[tx] + red[64 + tx]) + (red[128 + tx] + red[192 + tx]);
        gw[k] = (acc == 0.f ? 0.f : acc * gw[k]) + t;
    }
    if (blockIdx.x == 0 && gb) {
        float sb = 0.f;
        for (int b = threadIdx.x; b < B; b += blockDim.x) sb += dl[b];
        sb = block_sum(sb, sh);
        if (threadIdx.x == 0) gb[0] = (acc == 0.f ? 0.f : acc * gb[0]) + sb;
    }
}
// The same backward inside an fg_net backward pass, with the nn.PReLU [+ nn.Dropout] in FRONT of the Linear(K -> 1) folded in
// (models.lua:410-412): grid (K / 64, row slices of 16 samples) instead of K / 64 blocks that each walk the whole batch (8 blocks for
// K = 512: latency-bound), the stored gradient is the one wrt the PReLU's input, g * mask * mscale * (xp > 0 ? 1 : slope) -- the
// expressions of prelu_bwd_kernel -- and every sum (weight gradient rows, bias gradient, slope gradient) leaves per-block partials
// for the deferred final of the pass (fixed order).  One launch instead of two, 64 blocks instead of 8.
#define GEMV_BWD_ROWS 16
__global__ __launch_bounds__(256) void gemv_bwd_act_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ y, const float* __restrict__ gy,
                                                           float* __restrict__ gx, float* __restrict__ gwp, float* __restrict__ gbp,
                                                           int B, int K, int sigmoid, const float* __restrict__ xp,
                                                           const float* __restrict__ slope, const float* __restrict__ mask, float mscale,
                                                           float* __restrict__ apart) {
    __shared__ float dl[GEMV_BWD_ROWS];
    __shared__ float red[4 * 64];
    __shared__ float sh[4];
    const int b0 = blockIdx.y * GEMV_BWD_ROWS, nb = min(GEMV_BWD_ROWS, B - b0);
    if ((int)threadIdx.x < nb) {
        const int b = b0 + threadIdx.x;
        const float v = sigmoid ? y[b] : 0.f;
        dl[threadIdx.x] = sigmoid ? gy[b] * v * (1.f - v) : gy[b];
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + tx;
    float s = 0.f, ss = 0.f;
    const float sl = xp ? slope[0] : 0.f;
    if (k < K) {
        const float wk = w[k];
        for (int r = ty; r < nb; r += 4) {
            const size_t i = (size_t)(b0 + r) * K + k;
            if (gwp) s = fmaf(dl[r], x[i], s);
            float g = dl[r] * wk;
            if (xp) {
                if (mask) g = g * (mask[i] * mscale);
                const float xv = xp[i];
                const bool pos = xv > 0.f;
                ss = fmaf(pos ? 0.f : xv, g, ss);
                g = pos ? g : sl * g;
            }
            gx[i] = g;
        }
    }
    red[ty * 64 + tx] = s;
    __syncthreads();
    if (ty == 0 && k < K && gwp) gwp[(size_t)blockIdx.y * K + k] = (red[tx] + red[64 + tx]) + (red[128 + tx] + red[192 + tx]);
    if (apart) {
        ss = block_sum(ss, sh);
        if (threadIdx.x == 0) apart[blockIdx.y * gridDim.x + blockIdx.x] = ss;
    }
    if (blockIdx.x == 0 && gbp && threadIdx.x == 0) {
        float sb = 0.f;
        for (int r = 0; r < nb; ++r) sb += dl[r];
        gbp[blockIdx.y] = sb;
    }
}
int fg_launch_gemv_backward(fg_ctx* ctx, const float* x, const float* w, const float* y, const float* gy, float* gx,
                            float* gw, float* gb, float acc, int B, int K, int sigmoid, const FgActBwd* actb) {
    if (actb) actb->applied = 0;
    if (B == 0) return FG_OK;
    // inside an fg_net backward pass (deferred finals available), input gradient wanted, nothing to accumulate onto: the sliced
    // kernel, with the PReLU [+ Dropout] in front folded in when the caller describes one
    if (gx && acc == 0.f && ctx->defer && ctx->defer->n + 3 <= FG_DEFER_MAX) {
        const dim3 grid(fg_cdiv(K, 64), fg_cdiv(B, GEMV_BWD_ROWS));
        const bool act = actb && actb->x;
        float* gwp = gw ? fg_defer_alloc(ctx, (long long)grid.y * K) : nullptr;
        float* gbp = gb ? fg_defer_alloc(ctx, grid.y) : nullptr;
        float* ap = (act && actb->gslope) ? fg_defer_alloc(ctx, (long long)grid.x * grid.y) : nullptr;
        if ((!gw || gwp) && (!gb || gbp) && (!(act && actb->gslope) || ap)) {
            hipLaunchKernelGGL(gemv_bwd_act_kernel, grid, dim3(256), 0, ctx->stream, x, w, y, gy, gx, gwp, gbp, B, K, sigmoid,
                               act ? actb->x : (const float*)nullptr, act ? actb->slope : (const float*)nullptr,
                               act ? actb->mask : (const float*)nullptr, act ? actb->mscale : 1.f, ap);
            FG_CHECK_LAUNCH(ctx);
            if (gwp) fg_defer_push(ctx, gwp, (int)grid.y, K, 0.f, gw);
            if (gbp) fg_defer_push(ctx, gbp, (int)grid.y, 1, 0.f, gb);
            if (ap) fg_defer_push(ctx, ap, (int)(grid.x * grid.y), 1, 0.f, actb->gslope);
            if (act) actb->applied = 1;
            return FG_OK;
        }
    }
    hipLaunchKernelGGL(gemv_bwd_kernel, dim3(fg_cdiv(K, 64)), dim3(256), (B + 256) * sizeof(float), ctx->stream, x, w, y,
                       gy, gx, gw, gb, acc, B, K, sigmoid);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ BCECriterion (train.lua:148), eps = 1e-12
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ prob, const float* __restrict__ target,
                                                  float* __restrict__ loss, float* __restrict__ grad,
                                                  int* __restrict__ conf, int B) {
    __shared__ double shd[4];
    __shared__ int shc[4];
    const float eps = 1e-12f;
    double s = 0.0;
    if (threadIdx.x < 4) shc[threadIdx.x] = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float x = prob[b], t = target[b];
        s += (double)(t * logf(x + eps) + (1.f - t) * logf((1.f - x) + eps));
        if (grad) grad[b] = -(t - x) / (((1.f - x) + eps) * (x + eps)) / (float)B;
        if (conf) atomicAdd(&shc[(x > 0.5f ? 2 : 0) + (t > 0.5f ? 1 : 0)], 1);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss) loss[0] = (float)(-(shd[0] + shd[1] + shd[2] + shd[3]) / (double)B);
    if (threadIdx.x < 4 && conf) conf[threadIdx.x] = shc[threadIdx.x];
}
int fg_launch_bce(fg_ctx* ctx, const float* prob, const float* target, float* loss, float* grad, int* confusion,
                  int B) {
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, ctx->stream, prob, target, loss, grad, confusion, B);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ optimizers over the flat vector
__device__ __forceinline__ float sgnf(float v) { return fg_sgnf(v); }
__device__ __forceinline__ float prep_grad(float g, float p, float gscale, float l1mul, float l2, float clamp) {
    return fg_prep_grad(g, p, gscale, l1mul, l2, clamp);
}
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a, const AdamScalars k, int vec4) {
    if (vec4) {       // 16-byte form: n / 4 quads, then the <= 3 trailing elements
        const long long nq = a.n >> 2;
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x)
            fg_adam_elem4(a, k, q << 2);
        if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) (void)fg_adam_elem(a, k, (nq << 2) + threadIdx.x);
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x)
        (void)fg_adam_elem(a, k, i);
}
AdamScalars fg_adam_scalars(const AdamArgs& a) {
    const double bc1 = 1.0 - pow(a.beta1_d, (double)a.t);
    const double bc2 = 1.0 - pow(a.beta2_d, (double)a.t);
    AdamScalars k;
    k.step = (float)(a.lr_d * sqrt(bc2) / bc1); k.ob1 = (float)(1.0 - a.beta1_d); k.ob2 = (float)(1.0 - a.beta2_d);
    return k;
}
int fg_launch_adam(fg_ctx* ctx, const AdamArgs& a) {
    if (a.n == 0) return FG_OK;
    {
        FgProfScope prof(ctx, fg_intern(ctx, "adam_kernel"), 0.0, 0.0, 28.0 * (double)a.n);   // read p, g, m, v; write p, m, v
        const int vec4 = a.n >= 4;
        hipLaunchKernelGGL(adam_kernel, FG_GRID(vec4 ? (a.n + 3) / 4 : a.n, 256), dim3(256), 0, ctx->stream, a, fg_adam_scalars(a), vec4);
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void sgd_kernel(float* p, const float* g, float* mom, long long n, float gscale, float l1mul, float l2,
                           float clamp, float lr, float momentum, float dampening, float wd, int nesterov, int first) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float d = prep_grad(g[i], pv, gscale, l1mul, l2, clamp);
        if (wd != 0.f) d += wd * pv;
        if (momentum != 0.f) {
            float mv = first ? d : mom[i] * momentum + dampening * d;   // `dampening` holds (1 - dampening)
            mom[i] = mv;
            d = nesterov ? d + momentum * mv : mv;
        }
        p[i] = pv - lr * d;
    }
}
int fg_launch_sgd(fg_ctx* ctx, float* p, const float* g, float* mom, long long n, float gscale, float l1mul, float l2,
                  float clamp, float lr, float momentum, float dampening, float wd, int nesterov, int first) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(sgd_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, p, g, mom, n, gscale, l1mul, l2, clamp,
                       lr, momentum, dampening, wd, nesterov, first);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void adagrad_kernel(float* p, const float* g, float* var, long long n, float gscale, float l1mul, float l2,
                               float clamp, float clr) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float d = prep_grad(g[i], pv, gscale, l1mul, l2, clamp);
        const float v = var[i] + d * d;
        var[i] = v;
        p[i] = pv - clr * (d / (sqrtf(v) + 1e-10f));
    }
}
int fg_launch_adagrad(fg_ctx* ctx, float* p, const float* g, float* var, long long n, float gscale, float l1mul,
                      float l2, float clamp, float clr) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(adagrad_kernel, FG_GRID(n, 256), dim3(256), 0, ctx->stream, p, g, var, n, gscale, l1mul, l2,
                       clamp, clr);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ __launch_bounds__(256) void norms_partial_kernel(const float* __restrict__ p, long long n,
                                                            float* __restrict__ part) {
    __shared__ float sh[4];
    float s1 = 0.f, s2 = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = p[i];
        s1 += fabsf(v);
        s2 = fmaf(v, v, s2);
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s1;
        part[gridDim.x + blockIdx.x] = s2;
    }
}
__global__ void norms_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += (double)part[threadIdx.x * nb + i];
        out[threadIdx.x] = (float)s;
    }
}
int fg_launch_norms(fg_ctx* ctx, const float* p, long long n, float* out2, float* scratch) {
    dim3 grid = FG_GRID(n > 0 ? n : 1, 256);
    if (grid.x > 512) grid.x = 512;
    hipLaunchKernelGGL(norms_partial_kernel, grid, dim3(256), 0, ctx->stream, p, n, scratch);
    FG_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(norms_final_kernel, dim3(1), dim3(64), 0, ctx->stream, scratch, (int)grid.x, out2);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// ---------------------------------------------------------------------------------------------------------------------------------
// image.scale(src, width, height) of the Torch7 `image` package, default mode 'bilinear' (dataset_c2f.lua:53-56): the width pass
// of image_(Main_scaleLinear_rowcol) over every row, then its height pass over every column of the (float) intermediate.  Per axis:
// up-scaling interpolates linearly with scale = (src - 1) / (dst - 1) (end points onto end points), down-scaling is a box mean with
// fractional coverage of the two end pixels (scale = src / dst), equal lengths copy.  One thread per output element; every product,
// sum and quotient is rounded on its own, in the reference's order (no FMA contraction) -- bit-for-bit the C loop (provenance of the
// algorithm: include/facegen_hip.h fg_scale_bilinear).  sub_from (optional): also writes diff = sub_from - result (dataset_c2f.lua:59-61).
// ---------------------------------------------------------------------------------------------------------------------------------
struct ScaleTerm { int i0, i1; float w0, w1, n; int has_last, div; };
__device__ __forceinline__ ScaleTerm fg_scale_plan(int src_len, int dst_len, int di) {
#pragma clang fp contract(off)
    ScaleTerm t;
    t.w0 = 1.f; t.w1 = 0.f; t.n = 1.f; t.has_last = 0; t.div = 0;
    if (dst_len > src_len) {
        if (src_len == 1 || di == dst_len - 1) { t.i0 = src_len - 1; t.i1 = t.i0 + 1; return t; }
        const float scale = (float)(src_len - 1) / (float)(dst_len - 1);
        float f = (float)di * scale;
        const int i = (int)f;
        f = f - (float)i;
        t.i0 = i; t.i1 = i + 1; t.w0 = 1.f - f; t.w1 = f; t.has_last = 1;
        return t;
    }
    if (dst_len < src_len) {
        const float scale = (float)src_len / (float)dst_len;
        const float s0 = (float)di * scale, s1 = (float)(di + 1) * scale;
        t.i0 = (int)s0; t.i1 = (int)s1;
        const float f0 = s0 - (float)t.i0, f1 = s1 - (float)t.i1;
        t.w0 = 1.f - f0;
        float n = t.w0;
        for (int si = t.i0 + 1; si < t.i1; ++si) n = n + 1.f;
        t.has_last = t.i1 < src_len;
        t.w1 = f1;
        if (t.has_last) n = n + f1;
        t.n = n; t.div = 1;
        return t;
    }
    t.i0 = di; t.i1 = di + 1;
    return t;
}
template <typename F>
__device__ __forceinline__ float fg_scale_eval(const ScaleTerm& t, F fetch) {
#pragma clang fp contract(off)
    float acc = t.w0 * fetch(t.i0);
    for (int si = t.i0 + 1; si < t.i1; ++si) acc = acc + fetch(si);
    if (t.has_last) { const float v = t.w1 * fetch(t.i1); acc = acc + v; }
    if (t.div) acc = acc / t.n;
    return acc;
}
__global__ __launch_bounds__(256) void scale_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int C, int Hs,
                                                             int Ws, int Hd, int Wd, int nchw, const float* __restrict__ sub_from,
                                                             float* __restrict__ diff) {
#pragma clang fp contract(off)
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    long long r = idx;
    int c, x, y;
    if (nchw) { x = (int)(r % Wd); r /= Wd; y = (int)(r % Hd); r /= Hd; c = (int)(r % C); r /= C; }
    else { c = (int)(r % C); r /= C; x = (int)(r % Wd); r /= Wd; y = (int)(r % Hd); r /= Hd; }
    const long long n = r;
    const long long sy = nchw ? Ws : (long long)Ws * C, sx = nchw ? 1 : C;
    const float* img = src + (nchw ? (n * C + c) * (long long)Hs * Ws : n * (long long)Hs * Ws * C + c);
    const ScaleTerm th = fg_scale_plan(Ws, Wd, x), tv = fg_scale_plan(Hs, Hd, y);
    const float v = fg_scale_eval(tv, [&](int j) {
        const float* row = img + j * sy;
        return fg_scale_eval(th, [&](int i) { return row[i * sx]; });
    });
    dst[idx] = v;
    if (diff) diff[idx] = sub_from[idx] - v;
    }
}
int fg_launch_scale_bilinear(fg_ctx* ctx, const float* src, float* dst, int N, int C, int Hs, int Ws, int Hd, int Wd, int nchw,
                             const float* sub_from, float* diff) {
    const long long total = (long long)N * C * Hd * Wd;
    if (total == 0) return FG_OK;
    hipLaunchKernelGGL(scale_bilinear_kernel, FG_GRID(total, 256), dim3(256), 0, ctx->stream, src, dst, total, C, Hs, Ws, Hd, Wd, nchw,
                       sub_from, diff);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// mode 0: uniform(lo,hi)  1: bernoulli(keep=lo)  2: normal(mean=lo, std=hi)
__global__ void rng_kernel(uint64_t seed, uint64_t offset, float* __restrict__ out, long long n, float lo, float hi,
                           int mode) {
    const long long nq = (n + 3) / 4;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)q;
        uint32_t r[4];
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        float v[4];
        if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const float u1 = ((r[j] >> 8) + 1) * (1.f / 16777216.f), u2 = (r[j + 1] >> 8) * (1.f / 16777216.f);
                const float rad = sqrtf(-2.f * logf(u1));
                v[j] = lo + hi * rad * cosf(6.28318530718f * u2);
                v[j + 1] = lo + hi * rad * sinf(6.28318530718f * u2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = (r[j] >> 8) * (1.f / 16777216.f);
                v[j] = mode == 0 ? lo + u * (hi - lo) : (u < lo ? 1.f : 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = v[j];
    }
}
// several independent draws (the noise batch and every dropout mask of a training step) in ONE launch: segment k owns the
// quads [q0_k, q0_{k+1}) of the launch and is bit-identical to rng_kernel(seed_k, offset_k, out_k, n_k, ...)
__global__ __launch_bounds__(256) void rng_multi_kernel(const RngMulti m) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < m.total_quads; q += (long long)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < FG_RNG_MAX_SEGS; ++j) k += (j < m.n && q >= m.seg[j].q0) ? 1 : 0;
        const RngSeg sg = m.seg[k];
        const long long ql = q - sg.q0;
        const uint64_t ctr = sg.offset + (uint64_t)ql;
        uint32_t r[4];
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sg.seed, (uint32_t)(sg.seed >> 32), r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = (r[j] >> 8) * (1.f / 16777216.f);
            const float v = sg.mode == 0 ? sg.lo + u * (sg.hi - sg.lo) : (u < sg.lo ? 1.f : 0.f);
            if (ql * 4 + j < sg.n) sg.out[ql * 4 + j] = v;
        }
    }
}
int fg_launch_rng_multi(fg_ctx* ctx, RngMulti& m) {
    long long q = 0;
    for (int i = 0; i < m.n; ++i) {
        if (m.seg[i].mode != 0 && m.seg[i].mode != 1) return fg_set_err(ctx, FG_ERR_INVALID, "rng_multi: uniform / bernoulli only");
        m.seg[i].q0 = q;
        q += (m.seg[i].n + 3) / 4;
    }
    m.total_quads = q;
    if (q == 0) return FG_OK;
    hipLaunchKernelGGL(rng_multi_kernel, FG_GRID(q, 256), dim3(256), 0, ctx->stream, m);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
static int launch_rng(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi,
                      int mode) {
    if (n == 0) return FG_OK;
    hipLaunchKernelGGL(rng_kernel, FG_GRID((n + 3) / 4, 256), dim3(256), 0, ctx->stream, seed, offset, out, n, lo, hi,
                       mode);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_rng_uniform(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi) {
    return launch_rng(ctx, seed, offset, out, n, lo, hi, 0);
}
int fg_launch_rng_bernoulli(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float keep_prob) {
    return launch_rng(ctx, seed, offset, out, n, keep_prob, 0.f, 1);
}
int fg_launch_rng_normal(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float mean, float std) {
    return launch_rng(ctx, seed, offset, out, n, mean, std, 2);
}
