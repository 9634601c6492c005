// Internal declarations shared by the HIP translation units of libfacegen_hip.so.
// gfx950 (MI355X / CDNA4) only.  All device tensors are fp32, activations NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/facegen_hip.h"

#define FG_THIN_WGRAD_BLOCKS 512   // max partial slabs of the thin weight-gradient kernel
#define FG_MAX_GROUPS 64   // (parity,tap) groups per launch table (7x7 plain = 49, folded dgrad = 36)

#include <vector>
#include <string>
struct FgProfRec { const char* name; hipEvent_t e0, e1; double alg_flops, exec_flops, bytes; };
// Deferred final reductions (fg_net backward): the second stage of the bias-gradient / PReLU-slope reductions of a whole
// backward pass is collected here and run by ONE launch (fg_defer_flush) instead of one ~6 us launch per layer.
#define FG_DEFER_MAX 48
struct FgFinalJob { const float* part; float* out; int nrb, C; float beta; int blk0; };
struct FgDefer {
    float* arena = nullptr;          // partial sums must outlive the shared scratch: dedicated workspace region
    long long cap = 0, used = 0;
    FgFinalJob jobs[FG_DEFER_MAX];
    int n = 0, blocks = 0;
    // the split-K / parity sums of the weight gradients (wgrad_finish): same idea, one launch for all layers of the pass
    struct FgWFinishJob* wjobs = nullptr;    // FG_DEFER_WMAX entries (owned by the net)
    int wn = 0;
    long long wblocks = 0;
};
struct fg_ctx {
    FgDefer* defer = nullptr;        // non-null only inside fg_net backward
    int device;
    hipStream_t stream;
    char err[512];
    int sm_count;
    int math = 0;        // 0: native fp32 MFMA; 6: fp32 emulated with six split-bf16 plane products (fg_set_math)
    std::vector<const void*> attr_keys;   // fg_attr_first: call sites whose kernels had their dynamic-LDS limit raised on THIS context's device
    int fusion = FG_FUSE_DEFAULT;   // fg_set_fusion: which optional kernel fusions / variants are on (default from the environment)
    // optional per-launch HIP-event timing of the contraction kernels (bench.py roofline leg)
    bool prof = false;
    std::vector<FgProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::vector<std::string*> names;
    // fg_prof_clock_*: a one-wave probe on its own stream (s_memtime against the 100 MHz s_memrealtime)
    hipStream_t clk_stream = nullptr;
    unsigned long long* clk_dev = nullptr;
};
// true the first time `key` (the address of a call site's static) is seen on this context: hipFuncSetAttribute is per device, so
// the "done" flag belongs to the context, not to a process-wide static (one host thread may drive several devices)
bool fg_attr_first(fg_ctx* ctx, const void* key);
const char* fg_intern(fg_ctx* ctx, const char* s);  // stable pointer for a profile label
// RAII helper: records an event pair around one launch when profiling is on
struct FgProfScope {
    fg_ctx* ctx; int idx;
    FgProfScope(fg_ctx* c, const char* name, double alg, double exec, double bytes);
    ~FgProfScope();
};


int fg_set_err(fg_ctx* c, int code, const char* fmt, ...);
// Planning-only mode (fg_ctx_create(FG_DEVICE_NONE), include/facegen_hip.h): the process holds no HIP device.  Every kernel launch
// and every HIP runtime call of the library becomes a no-op while ALL host-side control flow -- plan building, stage walks,
// sync-BN pauses, bucket boundaries, the order / size / stream of every collective -- runs unchanged; "device" buffers are
// host allocations nobody dereferences.  A process is either planning-only or real, never both.
extern bool g_fg_dry;
// FG_LAUNCH_LOG=1 (read at fg_ctx_create): one stderr line per kernel launch -- name, grid, block, dynamic LDS.  Works in planning-only
// mode too, which makes the dispatch list of a whole training iteration readable on a machine without a GPU.
extern bool g_fg_launch_log;
void fg_log_launch_line(const char* name, dim3 grid, dim3 block, size_t lds);
template <typename... A>
static inline void fg_log_launch(const char* name, dim3 grid, dim3 block, size_t lds, hipStream_t, A&&...) { fg_log_launch_line(name, grid, block, lds); }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                            \
    do {                                                                               \
        if (g_fg_launch_log) fg_log_launch(#kernelName, __VA_ARGS__);                  \
        if (!g_fg_dry) hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);          \
    } while (0)
#define FG_HIP(ctx, call)                                                              \
    do {                                                                               \
        if (g_fg_dry) break;                                                           \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fg_set_err((ctx), FG_ERR_HIP, "%s: %s (%s:%d)", #call,              \
                              hipGetErrorString(e_), __FILE__, __LINE__);              \
    } while (0)
// library-owned device allocations (packed weights, job tables): host memory in planning-only mode
static inline hipError_t fg_dev_alloc(void** p, size_t bytes) {
    if (g_fg_dry) { *p = calloc(1, bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
static inline void fg_dev_free(void* p) { if (g_fg_dry) free(p); else (void)hipFree(p); }
#define FG_CHECK_LAUNCH(ctx) FG_HIP(ctx, hipGetLastError())

static inline int fg_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int fg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------
// Implicit-GEMM (gather-A) contraction:  Out[m][n] = bias[n] + sum_{g,c} A[pix(m,g)][c] * Bp[g][n][c]
//   m runs over an "M-space" of pixels (Nb x Hm x Wm); group g = one (parity,tap) with an
//   integer pixel offset into the A image; out-of-range A pixels read as zero.
//   Covers conv forward, conv data-grad, nearest-x2-folded conv forward / data-grad and
//   Linear forward / data-grad (Hm = Wm = 1, G = 1).
// ---------------------------------------------------------------------------------
struct IgemmArgs {
    const float* A;      // NHWC [Nb][Ha][Wa][Ca]
    const float* Bp;     // packed [P][G][Npad][Kpad]  (row = output channel, contiguous k)
    const float* bias;   // [N] or nullptr
    const void* A6;      // bf16x6 mode: A as split planes [pixel][Ca/16][3][16] bf16 (nullptr = fp32 path)
    const void* B6;      // bf16x6 mode: packed weights as split planes [P][G][Npad][Kpad/16][3][16] bf16
    float* Out;          // NHWC [Nb][Ho][Wo][N]  (or [splits][...] partials)
    float* stats_part;   // optional (splits == 1): per-channel sum / sum of squares of the RAW accumulators (= output - bias) of
    int stats_rows;      //   every wave's rows, [2][stats_rows][N] -- BatchNorm statistics without re-reading the output
    int Nb, Hm, Wm, M;   // M = Nb*Hm*Wm
    int lgH, lgW;        // log2(Hm), log2(Wm) if both are powers of two, else -1
    int Ha, Wa, Ca, Kpad;
    int asy, asx;        // A coord = y*asy + aoy[p][g]
    int Ho, Wo, N;
    int osy, osx;        // out coord = y*osy + ooy[p]
    int G, Npad, P;      // P = output parities (1 plain, 4 nearest-x2 folded forward)
    int splits;          // split-K over groups (gridDim.y); partials at Out + s*split_stride
    long long split_stride;
    int goff[4][FG_MAX_GROUPS];   // per (parity, group): (oy & 0xffff) | (ox << 16) pixel offsets into A
    signed char ooy[4], oox[4];
    long long a_bytes;   // size of the A tensor in bytes (< 2 GiB: raw-buffer addressing)
    // an nn.PReLU folded into the epilogue (splits == 1, fp32 kernels).  Forward: the PReLU BEHIND the layer -- the
    // pre-activation still goes to Out (backward needs it) and prelu(Out) to act_y (same layout).  Data gradient: the PReLU
    // in FRONT of the layer -- act_x is its input (same layout as Out), the stored value is the gradient wrt that input,
    // acc * (x > 0 ? 1 : slope), and every MFMA wave leaves its part of the slope gradient sum_{x <= 0} acc * x in
    // act_part[block * 4 + wave] (optional).  Same expressions as prelu_fwd_kernel / prelu_bwd_kernel.
    const float* act_slope;
    float* act_y;
    const float* act_x;
    float* act_part;
    double alg_flops;    // host-side bookkeeping only: reference-formulation FLOPs of this launch
    const char* tag;     // host-side: profile label
    unsigned long long* dbg_trace;   // measurement only (FG_WS_TRACE=1): s_memtime rows of igemm_ws_trace_kernel
    int dbg_nostore;     // measurement only (FG_DEBUG_NOSTORE=1): the epilogue's stores go to a zero-sized buffer (dropped by the hardware)
};
// tile: 0 = 128x128, 1 = 128x64, 2 = 64x64.  P = gridDim.z parities.
int fg_launch_igemm(fg_ctx* ctx, const IgemmArgs& a, int P, int tile);
// blocks along x of that launch (act_part holds 4 floats per block); 0 = bad tile
long long fg_igemm_blocks(const IgemmArgs& a, int P, int tile);
// sums split partials (+bias) : out[i] = bias[i % N] + sum_s part[s*stride + i]
// fp32 rows [rows][C] (C % 16 == 0) -> split-bf16 planes [rows][C/16][3][16]: x = h + m + l exactly
int fg_launch_split_planes(fg_ctx* ctx, const float* src, long long rows, int C, void* dst);
// the nn.PReLU [+ nn.Dropout] behind a layer, folded into the pass that finishes the layer's output (split-K sum)
struct FgActFuse { const float* slope; const float* mask; float mscale; float* y; mutable int applied; };
// the nn.PReLU in front of a layer, folded into the epilogue of the kernel that produces the gradient wrt the layer's input:
// x = the PReLU's input, gslope = its slope gradient (nullptr: not wanted); applied tells whether the launch folded it in
// mask / mscale: the nn.Dropout between that PReLU and the layer (gradient * mask * mscale first) -- only the pass that sums
// split-K partials folds a masked PReLU, the contraction epilogues take plain ones
struct FgActBwd { const float* x; const float* slope; float* gslope; const float* mask; float mscale; mutable int applied; };
// fg_set_fusion bit FG_FUSE_PRELU (default on; FG_FUSE_PRELU=0 in the environment clears it at context creation): off keeps
// every PReLU a pass of its own -- the A/B switch for measurements and for the parity tests, which run both ways
static inline bool fg_fuse_prelu(const fg_ctx* ctx) { return (ctx->fusion & FG_FUSE_PRELU) != 0; }
int fg_launch_sum_splits(fg_ctx* ctx, const float* part, int splits, long long stride, const float* bias,
                         int N, float* out, long long count, const FgActFuse* act = nullptr);
// data-gradient form: out = PReLU'(x) * (sum_s part_s [* mask * mscale]), slope-gradient partials through the deferred final
// (needs the arena of an fg_net backward pass when actb->gslope is set; actb->applied = 0 and a plain sum otherwise)
int fg_launch_sum_splits_actbwd(fg_ctx* ctx, const float* part, int splits, long long stride, float* out, long long count,
                                const FgActBwd* actb);

// ---------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) contraction (wino.hip): 3x3 / pad 1 / stride 1 layers, folded nearest-x2 up-convolutions (each output parity
// is a 3x3 convolution of the source) and 5x5 layers (four 3x3 sub-kernels), forward and data gradient.
//   M-space = B x TH x TW tiles of 2 x 2 "output elements"; tile (ty, tx), element (a, b), parity p is output pixel
//     (osy (2 ty + a) + ooy[p], osx (2 tx + b) + oox[p]);
//   K = KG groups x C channels; group g reads patch element (i, j), 0 <= i, j < 4, of tile (ty, tx) at input pixel
//     (isy (2 ty + i) + goy[g], isx (2 tx + j) + gox[g]), zero outside the input;
//   Out = bias + sum_{g, c} A^T [ U[p][.][g][c] (.) B^T patch_g,c B ] A;  U = G taps G^T comes pre-transformed from the re-pack
//   launch (WeightMap::wino, fg_wino_pack_at).
// ---------------------------------------------------------------------------------
struct WinoArgs {
    const float* X;      // NHWC [B][Hi][Wi][C]
    const float* U;      // packed [P][Npad / 64][KG][C / 8][pos 16][k half 2][64][4]  (fg_wino_pack_index)
    const float* bias;   // [N] or nullptr (added only when splits == 1)
    float* Out;          // NHWC [B][Ho][Wo][N]  (or [splits][...] partials)
    int B, Hi, Wi, C;
    int TH, TW, T;       // tiles per column / row / in the batch (T = B * TH * TW)
    int lgTH, lgTW;      // log2 if both are powers of two, else -1
    int isy, isx, KG;
    signed char goy[4], gox[4];
    int Ho, Wo, N, Npad; // Npad: output channels per parity, padded to 64
    int P, osy, osx;
    signed char ooy[4], oox[4];
    int splits;          // split over the K chunks (gridDim.y); partials at Out + s * split_stride
    long long split_stride;
    long long x_bytes;   // size of X in bytes (< 2 GiB: raw-buffer addressing)
    // an nn.PReLU folded into the epilogue (splits == 1), same meaning as IgemmArgs::act_*
    const float* act_slope;
    float* act_y;
    const float* act_x;
    float* act_part;     // 4 floats per block (fg_wino_blocks)
    float* stats_part;   // optional (splits == 1, forward): BatchNorm partial sums of the raw accumulators, [2][stats_rows][N],
    int stats_rows;      //   one row per (tile block, parity, wave row): stats_rows = 2 * P * ceil(T / 64)
    double alg_flops;    // host-side bookkeeping: reference-formulation FLOPs of this launch
    const char* tag;
    unsigned long long* dbg_trace;   // measurement only (FG_WINO_TRACE=1): s_memtime rows of wino_trace_kernel
};
int fg_launch_wino(fg_ctx* ctx, const WinoArgs& a);
long long fg_wino_blocks(const WinoArgs& a);
// packed position of U[parity p][group g][pos][n][k] -- n: output channel of the contraction (< Npad, a multiple of 64), k: its
// reduction channel (< Kpad, a multiple of 8): the (p, channel block n / 64, g) triple owns Kpad / 8 consecutive chunk images of
// [pos 16][k half 2][64][4] floats, the order wino_kernel's LDS stage wants
__host__ __device__ static inline size_t fg_wino_pack_at(int p, int g, int KG, int Npad, int Kpad, int pos, int n, int k) {
    const size_t img = (((size_t)p * (Npad >> 6) + (n >> 6)) * KG + g) * (Kpad >> 3) + (k >> 3);
    return ((img * 16 + pos) * 2 + ((k >> 2) & 1)) * 256 + (n & 63) * 4 + (k & 3);
}
// parities / K groups of the forward (bwd = 0) and data-gradient (bwd = 1) Winograd packs of a WeightMap with wino != 0
__host__ __device__ static inline void fg_wino_pack_shape(int kind, int wino, int bwd, int* P, int* KG) {
    const int par = kind == 1 ? 4 : 1, grp = wino == 2 ? 4 : 1;
    *P = bwd ? 1 : par;
    *KG = bwd ? par * grp : grp;
}

// ---------------------------------------------------------------------------------
// Weight-gradient contraction: Part[pg][s][n][c] = sum_{m in split s} dY[pixd(m,p)][n] * X[pixx(m,pg)][c]
// ---------------------------------------------------------------------------------
struct WgradArgs {
    const float* dY;     // NHWC [Nb][Hd][Wd][Nd]
    const float* X;      // NHWC [Nb][Hx][Wx][Cx]
    float* Part;         // [P*G][S][Npad][Cpad]
    float* bias_part;    // optional [P][S][Nd]: per-channel sums of dY over each (parity, split) -- the bias gradient's partials
    const void* D6;      // bf16x6 mode: dY / X as split planes [pixel][C/16][3][16] bf16 (nullptr = fp32 path)
    const void* X6;
    int Nb, Hm, Wm, M, lgH, lgW;
    int Hd, Wd, Nd, dsy, dsx;
    int Hx, Wx, Cx, xsy, xsx;
    int G, S, Npad, Cpad;
    int m_per_split;     // multiple of 32
    signed char doy[4], dox[4];
    signed char xoy[4][FG_MAX_GROUPS], xox[4][FG_MAX_GROUPS];
    long long d_bytes, x_bytes;   // operand sizes (< 2 GiB: raw-buffer addressing)
    double alg_flops;
    const char* tag;
};
int fg_launch_wgrad(fg_ctx* ctx, const WgradArgs& a, int P, int tile);  // tile: 0 = 128x128, 2 = 64x64
// wave-specialised fp32 weight gradient; cfg 0: block tile 256 dY-channels x 128 X-channels, cfg 1: 128 x 256
int fg_launch_wgrad_ws(fg_ctx* ctx, const WgradArgs& a, int P, int cfg);
bool fg_wgrad_ws_shape_ok(const WgradArgs& a, int cfg);
int fg_wgrad_ws_bias_rows(const WgradArgs& a, int cfg);      // bias_part rows per (parity, split): [P][S][rows][Nd]
// bf16x6 weight-gradient contraction; cfg 0: block tile 256 dY-channels x 128 X-channels, cfg 1: 128 x 256
int fg_launch_wgrad6(fg_ctx* ctx, const WgradArgs& a, int P, int cfg);

// ---------------------------------------------------------------------------------
// Winograd-domain weight gradient (wino_wgrad.hip) of a WinoArgs layer:
//   Part[unit = parity * KG + group][split][Npad][Cpad][pos 16] = sum over the split's tiles of dM'[pos][n] * V[pos][c]
//   (dM' = A' dY A'^T of the tile's 2x2 gradient block with the signs of A's last row left out, V = B^T patch B);
//   fg_launch_wgrad_finish with WeightMap::wino set sums the splits, applies the signs and G^T . G and scatters the sub-kernel
//   gradients into the reference taps.  Geometry fields as in WinoArgs (forward orientation); the tile grid must be 2^a x 2^b, b >= 1.
// ---------------------------------------------------------------------------------
struct WinoWgradArgs {
    const float* X;      // NHWC [B][Hi][Wi][Cx]   the layer's input
    const float* dY;     // NHWC [B][Ho][Wo][Nd]   the gradient of its output
    float* Part;
    float* bias_part;    // optional [P][S][Nd]: per-channel sums of dY over each (parity, split)
    int B, Hi, Wi, Cx, Ho, Wo, Nd;
    int TH, TW, T, lgTH, lgTW;
    int isy, isx, KG;
    signed char goy[4], gox[4];
    int P, osy, osx;
    signed char ooy[4], oox[4];
    int S, chunks_per_split;      // a chunk = 8 consecutive tiles
    int Npad, Cpad;               // multiples of 64
    long long x_bytes, d_bytes;
    double alg_flops;
    const char* tag;
    unsigned long long* dbg_trace;   // measurement only (FG_WINO_WGRAD_TRACE): s_memtime rows of wino_wgrad_trace_kernel
};
int fg_launch_wino_wgrad(fg_ctx* ctx, const WinoWgradArgs& a);

// Reference-layout <-> packed-layout description of one weight tensor.
struct WeightMap {
    int kind;            // 0 = plain conv / linear (k=1), 1 = nearest-x2 folded conv
    int wino;            // packs hold Winograd F(2x2, 3x3) transforms U = G t G^T of 3x3 sub-kernels t (wino.hip):
                         //   1 = one sub-kernel per output parity (kind 0: the 3x3 taps; kind 1: the folded 3x3 taps of each parity),
                         //   2 = a 5x5 layer as four sub-kernels at tap offsets (0 | 3, 0 | 3) of the zero-extended 6x6 window.
                         // forward pack: P parities x KG groups; data-gradient pack: one parity, the forward parities become groups
    int O, I, k, pad;    // reference W[O][I][k][k]
    int T, rmin;         // folded: window T x T, r = t + rmin
    int G;               // groups per parity (k*k plain, T*T folded)
    int P;               // parities (1 plain, 4 folded)
    // feature permutations for Linear next to a View (NCHW flatten <-> NHWC memory):
    int o_c, o_hw;       // if o_hw > 1: ref row o = c*o_hw + hw  <->  packed row = hw*o_c + c
    int i_c, i_hw;       // same for columns
};
// mode 0: forward pack  Bp[p][g][O_pad][I_pad]; mode 1: data-grad pack Bp[g'][I_pad][O_pad]
int fg_launch_pack_weights(fg_ctx* ctx, const WeightMap& wm, int mode, const float* W, float* Bp, int rows_pad,
                           int cols_pad);
// gradW_ref = beta*gradW_ref + sum over splits / parities of Part ([P*G][S][Npad][Cpad], n = O, c = I)
int fg_launch_wgrad_finish(fg_ctx* ctx, const WeightMap& wm, const float* Part, int S, int Npad, int Cpad, float beta,
                           float* gradW);
// the same reduction queued for the end of the backward pass (fg_defer_flush); false = not inside fg_net backward / table full
#define FG_DEFER_WMAX 16
struct FgWFinishJob { WeightMap wm; const float* part; float* gradW; int S, Npad, Cpad, ib; float beta; long long blk0; };
long long fg_conv_wgrad_part_floats(const struct ConvGeom& g);
long long fg_conv_wgrad_bias_part_floats(const struct ConvGeom& g);   // its bias-gradient partial rows (deferred final)
bool fg_defer_push_wfinish(fg_ctx* ctx, const WeightMap& wm, const float* Part, int S, int Npad, int Cpad, float beta, float* gradW);
// true inside the backward pass of a net whose workspace reserved room for parked weight-gradient partials (FG_FUSE_WFINISH_BATCH on
// at its creation AND now, job table not full)
static inline bool fg_defer_parks_w(const fg_ctx* ctx) {
    return (ctx->fusion & FG_FUSE_WFINISH_BATCH) && ctx->defer && ctx->defer->wjobs && ctx->defer->wn < FG_DEFER_WMAX;
}
int fg_launch_wgrad_finish_jobs(fg_ctx* ctx, const FgWFinishJob* jobs, int n, long long blocks);
// One launch re-packs every layer of a net after an optimizer step.
struct PackJob {
    WeightMap wm;
    int mode;             // 0 forward pack, 1 data-grad pack (one thread per packed element: Linear + View permutations),
                          // 7 both packs of a conv layer (one block per 16 x 16 channel patch, LDS-staged),
                          // 2 thin pack [tap][I][O], 3 thin pack [tap][O][I], 4 bias NCHW->NHWC perm,
                          // 8 both packs of a Linear layer, 9 no pack: optimizer update only (fused launch)
    long long src_off;    // offset into the flat parameter vector
    float* dst;
    int rows, cols;       // padded tile dims (modes 0/1)
    long long start, count;
    // mode 7: a conv layer's forward AND data-gradient packs from one LDS-staged pass over 16 x 16 (out, in) channel patches
    float* dst2;          // data-gradient pack
    int rows2, cols2;     // its padded dims (rows = in-channels, cols = out-channels)
    int npo, npi;         // patches along the out / in channel axis (over the padded extents)
};
// prof_bytes: algorithmic bytes of the launch for the profile (every parameter read once, every packed element written once)
// lds_floats: dynamic shared memory of the launch = max over the jobs of fg_pack_lds_floats(mode, k)
int fg_launch_pack_jobs(fg_ctx* ctx, const PackJob* jobs_dev, int njobs, long long total, const float* params, double prof_bytes,
                        long long lds_floats);
long long fg_pack_lds_floats(int mode, int k);
void fg_fold_window(int k, int pad, int* T, int* rmin);
static inline int fg_fold_r(int parity, int d, int pad) {  // floor((parity + d - pad)/2)
    int v = parity + d - pad;
    return (v >= 0) ? v / 2 : -((-v + 1) / 2);
}

// ---------------------------------------------------------------------------------
// pointwise / reduction kernels (pointwise.hip)
// ---------------------------------------------------------------------------------
int fg_launch_nchw_to_nhwc(fg_ctx*, const float* src, float* dst, int N, int C, int H, int W);
int fg_launch_nhwc_to_nchw(fg_ctx*, const float* src, float* dst, int N, int C, int H, int W);
int fg_launch_fill(fg_ctx*, float* p, float v, long long n);
// column sums of [M][N] -> out[N] = beta*out + sum (two-stage, deterministic). scratch >= 256*N floats
int fg_launch_colsum(fg_ctx*, const float* x, long long M, int N, float beta, float* out, float* scratch);

// BatchNorm (+ optional PReLU) over NHWC [M][C]
struct BnArgs {
    const float* x; float* y; long long M; int C;
    const float* gamma; const float* beta; const float* slope;  // slope nullptr -> no PReLU
    float* mean; float* invstd;        // [C] saved stats (train)
    float* running_mean; float* running_var; float eps, momentum; int train;
    float* scratch;                    // >= (2*nblk*C + 4*C) floats, nblk <= 1024
    // train mode, optional: partial sums left by the producing convolution's epilogue ([2][stats_rows][C], shifted by
    // stats_pivot[c] = the conv bias) -- the statistics pass over x is skipped
    const float* stats_part; int stats_rows; const float* stats_pivot;
};
int fg_launch_bn_forward(fg_ctx*, const BnArgs& a);
struct BnBwdArgs {
    const float* x; const float* gy; float* gx; long long M; int C;
    const float* gamma; const float* beta; const float* slope; const float* mean; const float* invstd;
    float* ggamma; float* gbeta; float* gslope; float gbeta_acc;  // grads = acc*old + new
    float* scratch;
};
int fg_launch_bn_backward(fg_ctx*, const BnBwdArgs& a);
// sync-BN halves: *_sync1 leaves fp64 per-channel sums [2C+1] in `sync` for the cross-rank all-reduce, *_sync2 finishes
int fg_launch_bn_forward_sync1(fg_ctx*, const BnArgs& a, double* sync);
int fg_launch_bn_forward_sync2(fg_ctx*, const BnArgs& a, const double* sync);
int fg_launch_bn_backward_sync1(fg_ctx*, const BnBwdArgs& a, double* sync);
int fg_launch_bn_backward_sync2(fg_ctx*, const BnBwdArgs& a, const double* sync);

// PReLU [+ dropout mask (scaled)] elementwise over n elements; mask index = i (same shape) or nullptr
int fg_launch_prelu_forward(fg_ctx*, const float* x, const float* slope, const float* mask, float mscale, float* y,
                            long long n);
int fg_launch_prelu_backward(fg_ctx*, const float* x, const float* gy, const float* slope, const float* mask,
                             float mscale, float* gx, float* gslope, float acc, long long n, float* scratch);
// fused PReLU -> SpatialDropout(mask[B][C], unscaled in train / (1-p) in eval) -> AvgPool2x2 on NHWC [B][H][W][C]
// Split-K partials a contraction left un-summed for the pointwise pass behind it: value[i] = bias[i % N] + sum_s part[s * stride + i]
// (fixed order = sum_splits_kernel's).  splits == 0: nothing pending.
struct FgSplitParts { const float* part; int splits; long long stride; const float* bias; int N; };
// sp (optional): x (forward) / gy (backward) is not materialised yet -- the kernel sums the partials itself; forward also
// writes the finished pre-activation to xout (the backward pass needs it)
int fg_launch_actpool_forward(fg_ctx*, const float* x, const float* slope, const float* mask, float mscale, float* y,
                              int B, int H, int W, int C, const FgSplitParts* sp = nullptr, float* xout = nullptr);
int fg_launch_actpool_backward(fg_ctx*, const float* x, const float* gy, const float* slope, const float* mask,
                               float mscale, float* gx, float* gslope, float acc, int B, int H, int W, int C,
                               float* scratch, const FgSplitParts* sp = nullptr);
// standalone pieces (module-level API)
int fg_launch_scale_mask_nc(fg_ctx*, const float* x, const float* mask, float mscale, float* y, int B, int HW, int C);
int fg_launch_avgpool_forward(fg_ctx*, const float* x, float* y, int B, int H, int W, int C);
int fg_launch_avgpool_backward(fg_ctx*, const float* gy, float* gx, int B, int H, int W, int C);
int fg_launch_upsample_forward(fg_ctx*, const float* x, float* y, int B, int H, int W, int C);
int fg_launch_upsample_backward(fg_ctx*, const float* gy, float* gx, int B, int H, int W, int C);
// SpatialConvolutionUpsample factor > 1: flat-NCHW re-view on NHWC tensors; src / dst hold B * C * h * w floats, C = nOut * f * f
int fg_launch_nchw_review(fg_ctx*, const float* src, float* dst, int B, int h, int w, int C, int f, int dir);
int fg_launch_sigmoid_forward(fg_ctx*, const float* x, float* y, long long n);
int fg_launch_sigmoid_backward(fg_ctx*, const float* y, const float* gy, float* gx, long long n);
int fg_launch_leakyrelu_forward(fg_ctx*, const float* x, float s, float* y, long long n);
int fg_launch_leakyrelu_backward(fg_ctx*, const float* x, const float* gy, float s, float* gx, long long n);
int fg_launch_axpby(fg_ctx*, float a, const float* x, float b, float* y, long long n);  // y = a*x + b*y
int fg_launch_maxpool_forward(fg_ctx*, const float* x, float* y, int B, int H, int W, int C);
int fg_launch_maxpool_backward(fg_ctx*, const float* x, const float* gy, float* gx, int B, int H, int W, int C);
// SpatialMaxPooling backward + the backward of the nn.PReLU in front of it in one pass: xpre = the PReLU's input (the
// pooled tensor prelu(xpre) is re-evaluated, bit-identical to the forward), gx = gradient wrt xpre
// mask / mscale (optional): the nn.Dropout behind the pool -- gy is multiplied by mask[i] * mscale first
int fg_launch_maxpool_prelu_backward(fg_ctx*, const float* xpre, const float* gy, const float* slope, float* gx,
                                     float* gslope, int B, int H, int W, int C, float* scratch, const float* mask = nullptr, float mscale = 1.f);
// PReLU -> SpatialMaxPooling(2, 2) [-> Dropout] forward from the pre-activation in one pass (mask: on the pooled tensor)
int fg_launch_actmaxpool_forward(fg_ctx*, const float* x, const float* slope, const float* mask, float mscale, float* y, int B, int H,
                                 int W, int C);
int fg_launch_mul_mask(fg_ctx*, const float* x, const float* mask, float scale, float* y, long long n);
int fg_launch_concat(fg_ctx*, const float* a, const float* b, float* out, long long npix, int ca, int cb);
int fg_launch_split(fg_ctx*, const float* g, float* ga, float* gb, long long npix, int ca, int cb);
int fg_launch_add(fg_ctx*, const float* a, const float* b, float* out, long long n);
int fg_launch_copy(fg_ctx*, const float* src, float* dst, long long n);
int fg_launch_add_halves(fg_ctx*, const float* a0, const float* b0, const float* a1, const float* b1, float* out, long long nh);
// deferred finals: partial buffer of `floats` floats (nullptr = not deferring / arena full -> caller uses its scratch and an
// immediate final); registration of one final job; flush = run all registered jobs in one launch
int fg_launch_colsum_final(fg_ctx* ctx, const float* part, int nrb, int C, float beta, float* out);   // out[c] = beta*out[c] + sum_r part[r][c]
// the same over C1 + C2 columns: the first C1 sums go to out1, the other C2 to out2 (beta = 0)
// out[c * o_hw + hw] = sum_r x[r][hw * o_c + c]: the bias gradient of a Linear behind a View, reference order, one launch (M small)
int fg_launch_colsum_perm(fg_ctx* ctx, const float* x, int M, int C, int o_c, int o_hw, float* out);
int fg_launch_colsum_final2(fg_ctx* ctx, const float* part, int nrb, int C1, float* out1, int C2, float* out2);
// the slabs of a thin weight gradient ([tap][s][c] columns) summed straight into gradW[O][I][k][k] (mode: fg_launch_thin_unpack_grad's);
// C2 further columns (the bias row) go to out2
int fg_launch_colsum_final_thin(fg_ctx* ctx, const float* part, int nrb, float* gradW, int O, int I, int k, int mode, int C2, float* out2);
float* fg_defer_alloc(fg_ctx* ctx, long long floats);
void fg_defer_push(fg_ctx* ctx, const float* part, int nrb, int C, float beta, float* out);
int fg_defer_flush(fg_ctx* ctx);
// image.scale (Torch7 `image`, bilinear): src [N][Hs][Ws][C] (nchw: [N][C][Hs][Ws]) -> dst at Hd x Wd; diff = sub_from - dst (optional)
int fg_launch_scale_bilinear(fg_ctx*, const float* src, float* dst, int N, int C, int Hs, int Ws, int Hd, int Wd, int nchw,
                             const float* sub_from, float* diff);
int fg_launch_zero_insert2(fg_ctx*, const float* g, float* out, int B, int H, int W, int C);   // g [B][H][W][C] -> out [B][2H][2W][C]

// thin convolutions (3 <-> wide channels), NHWC, stride 1, "same" pad, odd k <= 7
// thin-in : out[pix][c<Cw] = bias[c] + sum_{tap, s<Cs} in[pix+off(tap)][s] * Wp[tap][s][c]
// actf / actb (optional, MFMA variants only; ->applied says whether it happened): the PReLU behind (forward) / in front of
// (data gradient of a thin-output layer) the layer, folded into the epilogue like IgemmArgs::act_*
int fg_launch_thin_in_conv(fg_ctx*, const float* in, const float* Wp, const float* bias, float* out, int B, int H,
                           int W, int Cs, int Cw, int k, int flip, const FgActFuse* actf = nullptr,
                           const FgActBwd* actb = nullptr, float* padbuf = nullptr, long long padbuf_floats = 0)
                           /* padbuf: >= B*(H+k-1)*(W+k-1)*Cs floats enables the zero-bordered 5x5 / 7x7 MFMA path */;
// thin-out: out[pix][s<Cs] = act(bias[s] + sum_{tap, c<Cw} in[pix+off(tap)][c] * Wp[tap][s][c])
int fg_launch_thin_out_conv(fg_ctx*, const float* in, const float* Wp, const float* bias, float* out, int B, int H,
                            int W, int Cw, int Cs, int k, int flip, int sigmoid, float* rbuf = nullptr, long long rbuf_floats = 0)   /* rbuf: >= B*H*W*32 floats enables the two-pass 5x5/7x7 MFMA path */;
// thin wgrad: gw[tap][s][c] (partials reduced) = sum_pix thin[pix + sgn*off(tap)][s] * wide[pix][c]
// wide_colsum / colsum_done (optional, both or neither): sum_pix wide[pix][c] from the ones column of the matrix-pipe kernels
// (*colsum_done = 1 when produced); the slabs then have k*k*Cs + 1 rows -- scratch >= (FG_THIN_WGRAD_BLOCKS + 1) * (k*k*Cs + 1) * Cw
int fg_launch_thin_wgrad(fg_ctx*, const float* thin, const float* wide, float* gw_tsc, int B, int H, int W, int Cs,
                         int Cw, int k, int shift_thin, float* scratch, float* wide_colsum = nullptr, int* colsum_done = nullptr,
                         float* gradW_ref = nullptr, int unpack_mode = 0, int* unpacked = nullptr);
// repack between reference [O][I][k][k] and thin layouts
// mode 0: Wp[tap][s=I][c=O] (thin-in fwd, I small)      mode 1: Wp[tap][s=O][c=I] (thin-out fwd, O small)
int fg_launch_thin_pack(fg_ctx*, const float* W, float* Wp, int O, int I, int k, int mode);
// gradW_ref[O][I][k][k] = beta*gradW + gw[tap][s][c]  (mode as above: s = I (0) or s = O (1))
int fg_launch_thin_unpack_grad(fg_ctx*, const float* gw, float* gradW, int O, int I, int k, int mode, float beta);

// Linear(K -> 1) head: y[b] = act(x[b].w + b0)
int fg_launch_gemv_forward(fg_ctx*, const float* x, const float* w, const float* b, float* y, int B, int K,
                           int sigmoid);
// gy is grad wrt y (post-sigmoid if sigmoid) ; writes gx [B][K] (optional), gradw[K], gradb[1] (acc*old + new)
// actb (optional): the nn.PReLU [+ nn.Dropout] in front of the layer, folded into this launch (actb->applied; slope-gradient partials
// through the deferred final of an fg_net backward pass, like the contraction epilogues)
int fg_launch_gemv_backward(fg_ctx*, const float* x, const float* w, const float* y, const float* gy, float* gx,
                            float* gw, float* gb, float acc, int B, int K, int sigmoid, const FgActBwd* actb = nullptr);

// BCECriterion forward+backward fused: loss (device scalar), grad[B], confusion[4] = [pred][target] counts
int fg_launch_bce(fg_ctx*, const float* prob, const float* target, float* loss, float* grad, int* confusion, int B);

// fused penalty + clamp + Torch7-Adam over a flat vector
struct AdamArgs {
    float* p; const float* g; float* m; float* v; long long n;
    float gscale;            // gradient pre-scale (1/world after an all-reduce sum)
    float l1, l1_mul, l2;    // g += l1_mul*sign(p) + l2*p   (l1_mul: quirk C4 lets G use G_L2 here)
    float clamp;             // 0 = off
    float beta1, beta2, eps; int t;      // t already incremented; float copies used inside the kernel
    double lr_d, beta1_d, beta2_d;       // Lua-number (double) hyper-parameters for the host-side scalars
    float* gout;             // optional: write the penalised+clamped gradient back (feval's return value)
};
int fg_launch_adam(fg_ctx*, const AdamArgs& a);
// One element of penalty + clamp + Torch7-Adam, shared by adam_kernel and the fused optimizer + re-pack launch.  No product is
// contracted into an FMA (`#pragma clang fp contract(off)`; HIP's __fmul_rn & co. are plain operators and do get contracted): the
// two kernels must give the same bits, and the reference (TH's cmul / cadd / addcmul loops) rounds every product and sum as well.
__device__ __forceinline__ float fg_sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float fg_prep_grad(float g, float p, float gscale, float l1mul, float l2, float clamp) {
#pragma clang fp contract(off)
    g = g * gscale;
    if (l1mul != 0.f || l2 != 0.f) {                                  // adversarial.lua:109 / :223
        const float a = fg_sgnf(p) * l1mul, b = p * l2;
        g = g + (a + b);
    }
    if (clamp != 0.f) g = fminf(fmaxf(g, -clamp), clamp);             // adversarial.lua:121-123
    return g;
}
struct AdamScalars { float step, ob1, ob2; };      // lr * sqrt(1 - b2^t) / (1 - b1^t), 1 - b1, 1 - b2 (host side, in double)
AdamScalars fg_adam_scalars(const AdamArgs& a);
// one element's arithmetic (registers in, registers out): shared by the scalar and the 16-byte forms below
__device__ __forceinline__ void fg_adam_math(const AdamArgs& a, const AdamScalars& k, float p, float graw, float mo, float vo,
                                             float* pn, float* mn, float* vn, float* gp) {
#pragma clang fp contract(off)
    const float g = fg_prep_grad(graw, p, a.gscale, a.l1_mul, a.l2, a.clamp);
    // interruptable_optimizers.lua:78-90 : m = b1*m + (1-b1) g ; v = b2*v + (1-b2) g*g ; denom = sqrt(v)+eps
    const float m1 = mo * a.beta1, m2 = k.ob1 * g;
    const float m = m1 + m2;
    const float v1 = vo * a.beta2, v2 = (k.ob2 * g) * g;
    const float v = v1 + v2;
    const float denom = sqrtf(v) + a.eps;
    const float q = m / denom;
    const float u = k.step * q;
    *pn = p - u; *mn = m; *vn = v; *gp = g;
}
__device__ __forceinline__ float fg_adam_elem(const AdamArgs& a, const AdamScalars& k, long long i) {
    float pn, m, v, g;
    fg_adam_math(a, k, a.p[i], a.g[i], a.m[i], a.v[i], &pn, &m, &v, &g);
    a.m[i] = m;
    a.v[i] = v;
    a.p[i] = pn;
    if (a.gout) a.gout[i] = g;
    return pn;
}
// (dword-aligned 16-byte vector: gfx950 takes global_load / store_dwordx4 at any dword address -- the second moment of a net with an
// odd parameter count, v = m + n, is not 16-byte aligned)
typedef float fg_f4u __attribute__((ext_vector_type(4), aligned(4)));
// four consecutive elements through 16-byte loads and stores: a dword-per-lane store makes the
// L2 fetch the line it is about to overwrite (profiles/r06_pmc_tail.txt: adam_kernel fetched its 43 MB of operands PLUS ~its 32 MB of
// results), a 16-byte-per-lane store does not (bn_apply: fetch = operands)
__device__ __forceinline__ void fg_adam_elem4(const AdamArgs& a, const AdamScalars& k, long long i) {
    const fg_f4u p = *(const fg_f4u*)(a.p + i), g = *(const fg_f4u*)(a.g + i), m = *(const fg_f4u*)(a.m + i), v = *(const fg_f4u*)(a.v + i);
    float pn[4], mn[4], vn[4], gp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fg_adam_math(a, k, p[e], g[e], m[e], v[e], &pn[e], &mn[e], &vn[e], &gp[e]);
    *(fg_f4u*)(a.m + i) = fg_f4u{mn[0], mn[1], mn[2], mn[3]};
    *(fg_f4u*)(a.v + i) = fg_f4u{vn[0], vn[1], vn[2], vn[3]};
    *(fg_f4u*)(a.p + i) = fg_f4u{pn[0], pn[1], pn[2], pn[3]};
    if (a.gout) *(fg_f4u*)(a.gout + i) = fg_f4u{gp[0], gp[1], gp[2], gp[3]};
}
// the optimizer step and the re-pack in ONE pass: every pack job takes its weights from the Adam update of that element (each
// parameter is read by exactly one job; mode 9 jobs update the parameters no pack reads)
struct fg_net;
int fg_net_adam_step(fg_net* n, const AdamArgs& a);
int fg_launch_adam_pack_jobs(fg_ctx* ctx, const PackJob* jobs_dev, int njobs, long long total, const AdamArgs& a, long long lds_floats);
int fg_launch_sgd(fg_ctx*, float* p, const float* g, float* mom, long long n, float gscale, float l1mul, float l2,
                  float clamp, float lr, float momentum, float dampening, float wd, int nesterov, int first);
int fg_launch_adagrad(fg_ctx*, float* p, const float* g, float* var, long long n, float gscale, float l1mul, float l2,
                      float clamp, float clr);
// out[0] = sum|p|, out[1] = sum p^2 (fp64 accumulate in the final stage)
int fg_launch_norms(fg_ctx*, const float* p, long long n, float* out2, float* scratch);

// Philox4x32-10 counter RNG
#define FG_RNG_MAX_SEGS 12
struct RngSeg { float* out; long long n; uint64_t seed, offset; float lo, hi; int mode; long long q0; };   // mode 0 uniform(lo,hi), 1 bernoulli(keep = lo)
struct RngMulti { RngSeg seg[FG_RNG_MAX_SEGS]; int n; long long total_quads; };
int fg_launch_rng_multi(fg_ctx*, RngMulti& m);   // fills q0 / total_quads; segment i == fg_launch_rng_*(seed_i, offset_i, ...)
int fg_launch_rng_uniform(fg_ctx*, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi);
int fg_launch_rng_bernoulli(fg_ctx*, uint64_t seed, uint64_t offset, float* out, long long n, float keep_prob);
int fg_launch_rng_normal(fg_ctx*, uint64_t seed, uint64_t offset, float* out, long long n, float mean, float std);
