// "Thin" convolutions for gfx950: one side of the contraction has <= 4 channels (image side of the nets:
// D's first conv 3->64 and G's last conv 128->3, models.lua:385 / :73, and their data / weight gradients).
// N or K of the GEMM view is 3 (or 27), so MFMA tiles would be > 90 % padding: these are HBM/VALU-bound and
// run on the vector ALU with coalesced NHWC accesses and weights held in registers.
#include "fg_internal.h"
// FG_THIN_DBG (drop the gathers / the stores of thin_in_mfma_kernel: timing experiments, WRONG results) exists only in builds with
// -DFG_MEASURE; the production kernel carries no such branch
#ifdef FG_MEASURE
#define FG_THIN_DBG_BIT(v, b) ((v) & (b))
#else
#define FG_THIN_DBG_BIT(v, b) 0
#endif
#include <string.h>
#include <stdlib.h>
typedef float tw_f32x16 __attribute__((ext_vector_type(16)));   // MFMA 32x32 accumulator
typedef float tw_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_x(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------
// thin-in: out[pix][c] = bias[c] + sum_{tap,s} in[pix + off(tap)][s] * Wp[tap][s][c]     (Cs small, Cw wide)
// block = 256 threads = CBLK channels x PL pixel lanes; weights for channel c live in registers.
// ---------------------------------------------------------------------------------
// One wave per pixel (wave-uniform pixel index): the CS-channel input taps are the same for all 64 lanes, so they
// are fetched through the scalar cache (s_load) and feed v_fmac as SGPR operands; each lane owns CJ output channels
// whose weights stay in registers.  Stores are 256 B coalesced per wave.
template <int K, int CS, int CJ>
__global__ __launch_bounds__(256) void thin_in_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B,
                                                      int H, int W, int flip, int Cw) {
    constexpr int PAD = (K - 1) / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.y * (CJ * 64) + lane;       // this lane's first output channel
    float w[K * K * CS][CJ];
#pragma unroll
    for (int t = 0; t < K * K * CS; ++t)
#pragma unroll
        for (int j = 0; j < CJ; ++j) w[t][j] = Wp[(size_t)t * Cw + cb + 64 * j];
    float bv[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) bv[j] = bias ? bias[cb + 64 * j] : 0.f;
    const int npix = B * H * W;
    const int nwaves = gridDim.x * 4;
    // two pixels per iteration; every tap is loaded from a clamped (always valid) wave-uniform address and zeroed by a
    // scalar select, so all 2*K*K scalar loads of an iteration are issued before the first use (one latency, not K)
    for (int pix0 = blockIdx.x * 4 + wave; pix0 < npix; pix0 += 2 * nwaves) {
        float v[2][K * K * CS];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = min(pix0 + u * nwaves, npix - 1);
            const int x = pix % W;
            const int t = pix / W;
            const int y = t % H;
            const int b = t / H;
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
                const int yy = y + (flip ? PAD - dy : dy - PAD);
                const int yc = min(max(yy, 0), H - 1);
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const int xx = x + (flip ? PAD - dx : dx - PAD);
                    const int xc = min(max(xx, 0), W - 1);
                    const bool ok = yy == yc && xx == xc;
                    const float* ip = in + ((size_t)(b * H + yc) * W + xc) * CS;   // wave-uniform address
#pragma unroll
                    for (int s = 0; s < CS; ++s) {
                        const float tv = ip[s];
                        v[u][(dy * K + dx) * CS + s] = ok ? tv : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = pix0 + u * nwaves;
            if (pix >= npix) break;
            float acc[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) acc[j] = bv[j];
#pragma unroll
            for (int t = 0; t < K * K * CS; ++t)
#pragma unroll
                for (int j = 0; j < CJ; ++j) acc[j] = fmaf(v[u][t], w[t][j], acc[j]);
#pragma unroll
            for (int j = 0; j < CJ; ++j) out[(size_t)pix * Cw + cb + 64 * j] = acc[j];
        }
    }
}

// 3x3 thin-in, one wave per image ROW: the wave walks x with a sliding 3x3xCS window held in scalar registers (one new
// column = 3 scalar loads per pixel, no per-pixel index arithmetic); the x loop is unrolled by 3 so the window slots
// rotate statically.  Weights are pre-flipped for the data-grad form, so the loop body is always a correlation.
template <int CS, int CJ>
__global__ __launch_bounds__(256) void thin_in_row3_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int B, int H, int W, int flip, int Cw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.y * (CJ * 64) + lane;
    float w[9 * CS][CJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
            for (int j = 0; j < CJ; ++j) w[t * CS + s][j] = Wp[(size_t)((flip ? 8 - t : t) * CS + s) * Cw + cb + 64 * j];
    float bv[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) bv[j] = bias ? bias[cb + 64 * j] : 0.f;
    const int row = blockIdx.x * 4 + wave;            // (b, y), wave-uniform
    if (row >= B * H) return;
    const int y = row % H;
    const float* rp[3];
    bool rok[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        rok[dy] = (unsigned)yy < (unsigned)H;
        rp[dy] = in + (size_t)(row + (rok[dy] ? dy - 1 : 0)) * W * CS;
    }
    float win[3][3][CS];                               // [column slot][row][channel]
#define FG_LOADCOL(slot, xc)                                                                        \
    {                                                                                               \
        const bool cok = (unsigned)(xc) < (unsigned)W;                                              \
        const int xcl = cok ? (xc) : 0;                                                             \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                            \
            _Pragma("unroll") for (int s = 0; s < CS; ++s) {                                        \
                const float tv = rp[dy][xcl * CS + s];                                              \
                win[slot][dy][s] = (cok && rok[dy]) ? tv : 0.f;                                     \
            }                                                                                       \
    }
    FG_LOADCOL(2, -1)                                  // column -1 lives in slot (-1 mod 3) = 2
    FG_LOADCOL(0, 0)
    float* orow = out + (size_t)row * W * Cw + cb;
    for (int x0 = 0; x0 < W; x0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int x = x0 + u;
            if (x >= W) break;
            FG_LOADCOL((u + 1) % 3, x + 1)
            float acc[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) acc[j] = bv[j];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < CS; ++s)
#pragma unroll
                        for (int j = 0; j < CJ; ++j)
                            acc[j] = fmaf(win[(u + dx + 2) % 3][dy][s], w[(dy * 3 + dx) * CS + s][j], acc[j]);
#pragma unroll
            for (int j = 0; j < CJ; ++j) orow[(size_t)x * Cw + 64 * j] = acc[j];
        }
    }
#undef FG_LOADCOL
}

// Large-kernel thin-in (c2f generator head data-grad: 7x7, 3 -> 256): a strip of TR image rows (+ halo, zero padded) of
// the thin operand is staged in LDS; each wave computes NPX consecutive output pixels per step for its 64 channels,
// reading every halo row segment once (uniform-address LDS broadcast) and re-using it for the NPX pixels; the K*K*CS
// weights of the lane's channel stay in registers (pre-flipped for the data-grad form).
template <int K, int CS>
__global__ __launch_bounds__(256) void thin_in_rows_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int B, int H, int W, int flip, int Cw) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int TR = 4, NPX = 4;
    constexpr int SEG = (NPX + K - 1) * CS;            // floats of one halo row segment
    constexpr int SEG4 = (SEG + 3) / 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int WP = (W + 2 * PAD + 3) / 4 * 4 + 4;      // padded row length in pixels (room for the last float4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    float w[K * K * CS];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int s = 0; s < CS; ++s) w[t * CS + s] = Wp[(size_t)((flip ? K * K - 1 - t : t) * CS + s) * Cw + c];
    const float bv = bias ? bias[c] : 0.f;
    const int strips_per_img = (H + TR - 1) / TR;
    const int b = blockIdx.x / strips_per_img, y0 = (blockIdx.x - b * strips_per_img) * TR;
    for (int i = threadIdx.x; i < (TR + 2 * PAD) * WP * CS; i += 256) {
        const int s = i % CS;
        const int t = i / CS;
        const int xx = t % WP - PAD, yy = y0 + t / WP - PAD;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = in[((size_t)(b * H + yy) * W + xx) * CS + s];
        tile[i] = v;
    }
    __syncthreads();
    const int rows = min(TR, H - y0);
    const int gpr = (W + NPX - 1) / NPX;               // pixel groups per row
    for (int grp = wave; grp < rows * gpr; grp += 4) {
        const int r = grp / gpr, x0 = (grp - r * gpr) * NPX;
        float acc[NPX];
#pragma unroll
        for (int q = 0; q < NPX; ++q) acc[q] = bv;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            float seg[SEG4];
            const float* sp = tile + ((r + dy) * WP + x0) * CS;     // wave-uniform address
#pragma unroll
            for (int i = 0; i < SEG4; ++i) seg[i] = sp[i];
#pragma unroll
            for (int dx = 0; dx < K; ++dx)
#pragma unroll
                for (int s = 0; s < CS; ++s)
#pragma unroll
                    for (int q = 0; q < NPX; ++q) acc[q] = fmaf(seg[(q + dx) * CS + s], w[(dy * K + dx) * CS + s], acc[q]);
            __builtin_amdgcn_sched_barrier(0);   // keep one row segment live at a time (else 7 x 32 VGPRs get hoisted)
        }
#pragma unroll
        for (int q = 0; q < NPX; ++q)
            if (x0 + q < W) out[((size_t)(b * H + y0 + r) * W + x0 + q) * Cw + c] = acc[q];
    }
}

// generic fallback: runtime k / Cs, weights re-read through L1
__global__ __launch_bounds__(256) void thin_in_generic_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int B, int H, int W, int Cs, int Cw, int K, int flip,
                                                              int cblk) {
    const int PAD = (K - 1) / 2;
    const int pl = threadIdx.x / cblk, tc = threadIdx.x - pl * cblk;
    const int PL = 256 / cblk;
    const int c = blockIdx.y * cblk + tc;
    const float bv = bias ? bias[c] : 0.f;
    const int npix = B * H * W;
    const int p0 = blockIdx.x * 128;
    for (int j = pl; j < 128; j += PL) {
        const int pix = p0 + j;
        if (pix >= npix) break;
        const int x = pix % W;
        const int t = pix / W;
        const int y = t % H;
        const int b = t / H;
        float acc = bv;
        for (int dy = 0; dy < K; ++dy) {
            const int yy = y + (flip ? PAD - dy : dy - PAD);
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = 0; dx < K; ++dx) {
                const int xx = x + (flip ? PAD - dx : dx - PAD);
                if ((unsigned)xx >= (unsigned)W) continue;
                const float* ip = in + ((size_t)(b * H + yy) * W + xx) * Cs;
                for (int s = 0; s < Cs; ++s) acc = fmaf(ip[s], Wp[(size_t)((dy * K + dx) * Cs + s) * Cw + c], acc);
            }
        }
        out[(size_t)pix * Cw + c] = acc;
    }
}

// An nn.PReLU folded into the epilogue of the MFMA thin-input kernels (the same contract as IgemmArgs::act_*): y = also
// write prelu(out) there (forward); x = the PReLU's input, the stored value becomes acc * (x > 0 ? 1 : slope) and the wave's
// share of the slope gradient goes to part[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] (data gradient of a thin-OUTPUT layer).
struct ThinEpi { const float* slope; float* y; const float* x; float* part; };
template <int EPI>
__device__ __forceinline__ void thin_epi_store(const ThinEpi& e, float sl, float* __restrict__ out, size_t idx, float v0, float v1,
                                               float x0, float x1, float& s) {
    if constexpr (EPI == 2) {          // x0 / x1: the PReLU input at idx / idx + 32, fetched before the tile's MFMA loop
        s = fmaf(x0 > 0.f ? 0.f : x0, v0, s);
        s = fmaf(x1 > 0.f ? 0.f : x1, v1, s);
        out[idx] = x0 > 0.f ? v0 : sl * v0;
        out[idx + 32] = x1 > 0.f ? v1 : sl * v1;
    } else {
        out[idx] = v0;
        out[idx + 32] = v1;
        if constexpr (EPI == 1) {
            e.y[idx] = v0 > 0.f ? v0 : sl * v0;
            e.y[idx + 32] = v1 > 0.f ? v1 : sl * v1;
        }
    }
}
// EPI == 2: the 2 x 16 values of x a lane needs for one 32-pixel tile, requested BEFORE the tile's MFMA loop so the loads
// are hidden behind it (issued in the epilogue they doubled the kernel's exposed memory time)
template <int EPI>
__device__ __forceinline__ void thin_epi_prefetch(const ThinEpi& e, int tile, int h, int npix, int Cw, int cbj, float (&x0)[16],
                                                  float (&x1)[16]) {
    if constexpr (EPI == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const size_t idx = (size_t)(p < npix ? p : 0) * Cw + cbj;
            x0[r] = e.x[idx];
            x1[r] = e.x[idx + 32];
        }
    }
}
__device__ __forceinline__ void thin_epi_finish(const ThinEpi& e, float s, int lane, int wave) {
    if (!e.part) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) e.part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] = s;
}

// 3x3 thin-input convolution (3 or 1 channels -> 64 / 128) on the fp32 matrix pipe: a 32-pixel x 32-channel tile is
// ceil(9*CS / 2) v_mfma_f32_32x32x2_f32 with K = the (tap, channel) index; A = the shifted input values of the 32 pixels
// (per-lane gather from the tiny input, zero outside the image), B = the packed weights held in registers for the whole
// kernel, bias folded into the accumulator init.  Replaces the row-walking VALU kernel (26 us for a 33 MB output).
template <int CS, int EPI>
__global__ __launch_bounds__(256) void thin_in_mfma_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int npix, int H, int W, int flip, int Cw, int lgH, int lgW,
                                                           const ThinEpi epi, int nt_store) {
    constexpr int NA = 9 * CS;
    constexpr int KS = (NA + 1) / 2;
    constexpr int TS_LD = 68;            // floats per staged pixel row (64 channels + 4: 16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) float tsm[EPI == 0 ? 4 : 1][EPI == 0 ? 32 * TS_LD : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = blockIdx.y * 64;
    const int h = lane >> 5, j = lane & 31;
    float wb[KS][2];
    int kdesc[KS];                       // (oy + 1) | (ox + 1) << 2 | s << 4 | valid << 8
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 2 * ks + h;
        const bool valid = k < NA;
        const int tap = valid ? k / CS : 0, sc = valid ? k - tap * CS : 0;
        const size_t widx = (size_t)((flip ? 8 - tap : tap) * CS + sc) * Cw + cb + j;
        wb[ks][0] = valid ? Wp[widx] : 0.f;
        wb[ks][1] = valid ? Wp[widx + 32] : 0.f;
        kdesc[ks] = (tap / 3) | ((tap % 3) << 2) | (sc << 4) | ((valid ? 1 : 0) << 8);
    }
    const float b0 = bias ? bias[cb + j] : 0.f, b1 = bias ? bias[cb + 32 + j] : 0.f;
    const bool p2 = lgW >= 0 && lgH >= 0;
    const int ntiles = (npix + 31) / 32;
    const float esl = EPI ? epi.slope[0] : 1.f;
    float es = 0.f;
    // A fragments of one tile: lane (pixel j, k half h) gathers its KS shifted input values (zero outside the image)
    auto load_a = [&](int tile, float (&av)[KS]) {
        const int pix = tile * 32 + j;
        const bool ok = pix < npix;
        int x, y, t;
        if (p2) { x = pix & (W - 1); t = pix >> lgW; y = t & (H - 1); }
        else { t = pix / W; x = pix - t * W; y = t % H; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = kdesc[ks];
            const int yy = y + (d & 3) - 1, xx = x + ((d >> 2) & 3) - 1;
            float a = 0.f;
            if (ok && (d >> 8) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && !FG_THIN_DBG_BIT(nt_store, 4))
                a = in[(size_t)((t - y + yy) * W + xx) * CS + ((d >> 4) & 15)];
            av[ks] = a;
        }
    };
    const int tstep = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    float a_cur[KS], a_nxt[KS];
    if (tile < ntiles) load_a(tile, a_cur);
    for (; tile < ntiles; tile += tstep) {
        // a wave owns several tiles: the next tile's gather is in flight while this one multiplies and stores
        if (tile + tstep < ntiles) load_a(tile + tstep, a_nxt);
        tw_f32x16 acc0, acc1;
        float px0[16], px1[16];
        thin_epi_prefetch<EPI>(epi, tile, h, npix, Cw, cb + j, px0, px1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ks], wb[ks][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ks], wb[ks][1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
        if constexpr (EPI == 0) {
            // the kernel is store-bound: transpose the wave's 32-pixel x 64-channel tile through LDS so that one store
            // instruction writes four whole 256-byte channel rows as float4 (the accumulator layout gives 128-byte half rows,
            // 32 store instructions per tile: 1.9 TB/s)
            float* ts = tsm[wave];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pl = (r & 3) + 8 * (r >> 2) + 4 * h;
                ts[pl * TS_LD + j] = acc0[r];
                ts[pl * TS_LD + 32 + j] = acc1[r];
            }
            __builtin_amdgcn_wave_barrier();        // same wave, in-order LDS: only the compiler must not reorder
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pl = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
                const tw_f32x4 v = *(const tw_f32x4*)(ts + pl * TS_LD + c4);
                const int p = tile * 32 + pl;
                if (p < npix && !FG_THIN_DBG_BIT(nt_store, 2)) {
                    // measurement switches: FG_THIN_NT=1 streaming stores that do not allocate in L2; FG_THIN_DBG bit 0 drops the
                    // stores, bit 1 the gathers (what is left of the kernel without them)
                    if (nt_store & 1) __builtin_nontemporal_store(v, (tw_f32x4*)(out + (size_t)p * Cw + cb + c4));
                    else *(tw_f32x4*)(out + (size_t)p * Cw + cb + c4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < npix) thin_epi_store<EPI>(epi, esl, out, (size_t)p * Cw + cb + j, acc0[r], acc1[r], px0[r], px1[r], es);
            }
        }
    }
    if constexpr (EPI == 2) thin_epi_finish(epi, es, lane, wave);
}

// The same contraction for the larger thin layers (5x5 / 7x7 kernels, 4-channel input: the c2f nets): K = K*K*CS is up to
// 147, too many weight fragments for registers, so the packed weights of the block's 64 output channels live in LDS
// ([K*K*CS][64] floats, one conflict-free ds_read_b32 per MFMA).
template <int K, int CS, int EPI>
__global__ __launch_bounds__(256) void thin_in_mfma_lds_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int npix, int H, int W, int flip, int Cw, int lgH, int lgW,
                                                               const ThinEpi epi) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NA = K * K * CS;
    constexpr int KS = (NA + 1) / 2;
    __shared__ float wsh[2 * KS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = blockIdx.y * 64;
    const int h = lane >> 5, j = lane & 31;
    for (int e = threadIdx.x; e < 2 * KS * 64; e += 256) {
        const int kk = e >> 6, c = e & 63;
        float v = 0.f;
        if (kk < NA) {
            const int tap = kk / CS, sc = kk - tap * CS;
            v = Wp[(size_t)((flip ? K * K - 1 - tap : tap) * CS + sc) * Cw + cb + c];
        }
        wsh[kk][c] = v;
    }
    __syncthreads();
    const float b0 = bias ? bias[cb + j] : 0.f, b1 = bias ? bias[cb + 32 + j] : 0.f;
    const bool p2 = lgW >= 0 && lgH >= 0;
    const int ntiles = (npix + 31) / 32;
    const float esl = EPI ? epi.slope[0] : 1.f;
    float es = 0.f;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int pix = tile * 32 + j;
        const bool ok = pix < npix;
        int x, y, t;
        if (p2) { x = pix & (W - 1); t = pix >> lgW; y = t & (H - 1); }
        else { t = pix / W; x = pix - t * W; y = t % H; }
        tw_f32x16 acc0, acc1;
        float px0[16], px1[16];
        thin_epi_prefetch<EPI>(epi, tile, h, npix, Cw, cb + j, px0, px1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
        // the gather of a tile in groups of GQ values BEFORE their MFMAs (a load per MFMA pair behind its own
        // `s_waitcnt vmcnt(0)` left the matrix pipe waiting one cache round trip per pair: 67 TFLOP/s on the 7x7 layers)
        constexpr int GQ = 16;
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += GQ) {
            float av[GQ];
#pragma unroll
            for (int u = 0; u < GQ; ++u) {
                const int k = 2 * (ks0 + u) + h;
                const int tap = k / CS, sc = k - tap * CS;
                const int dy = tap / K, dx = tap - dy * K;
                const int yy = y + dy - PAD, xx = x + dx - PAD;
                float a = 0.f;
                if (ok && k < NA && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    a = in[(size_t)((t - y + yy) * W + xx) * CS + sc];
                av[u] = a;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GQ; ++u) {
                const int k = 2 * (ks0 + u) + h;
                if (ks0 + u < KS) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wsh[k][j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wsh[k][32 + j], acc1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (p < npix) thin_epi_store<EPI>(epi, esl, out, (size_t)p * Cw + cb + j, acc0[r], acc1[r], px0[r], px1[r], es);
        }
    }
    if constexpr (EPI == 2) thin_epi_finish(epi, es, lane, wave);
}

// zero-bordered copy of a thin tensor: [B][H + 2 PAD][W + 2 PAD][CS] (shared with the thin weight gradient below)
__global__ __launch_bounds__(256) void thin_pad_kernel(const float* __restrict__ thin, float* __restrict__ padded, int B, int H, int W,
                                                       int CS, int PAD) {
    const int Hp = H + 2 * PAD, Wp = W + 2 * PAD;
    const long long n = (long long)B * Hp * Wp * CS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i % CS);
        long long t = i / CS;
        const int xp = (int)(t % Wp); t /= Wp;
        const int yp = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const int y = yp - PAD, x = xp - PAD;
        padded[i] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? thin[((size_t)(b * H + y) * W + x) * CS + s] : 0.f;
    }
}
static int fg_launch_thin_pad(fg_ctx* ctx, const float* thin, float* padded, int B, int H, int W, int Cs, int pad) {
    const long long padf = (long long)B * (H + 2 * pad) * (W + 2 * pad) * Cs;
    hipLaunchKernelGGL(thin_pad_kernel, dim3((unsigned)((padf + 255) / 256 < 4096 ? (padf + 255) / 256 : 4096)), dim3(256), 0, ctx->stream,
                       thin, padded, B, H, W, Cs, pad);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// 5x5 / 7x7 thin-input convolution from a ZERO-BORDERED copy of the input (power-of-two maps; the data gradient of the c2f
// generator head, models_c2f.lua:131 backward).  thin_in_mfma_lds_kernel spends ~10 VALU instructions per MFMA on its gather
// (tap -> (dy, dx, s) divisions by constants that differ between the two lane halves, four bounds, the address, a select) and
// every instruction issued beside an MFMA costs the pipe 6-9 cycles: 69 TFLOP/s.  Here the K axis is ordered so that
// NO per-gather arithmetic is left: the K x K window is split into an upper and a lower half of RH = (K + 1) / 2 rows; the k-pair
// q = (d, dx, s) of a v_mfma_f32_32x32x2_f32 multiplies tap (d, dx) in lanes 0-31 and tap (d + RH, dx) in lanes 32-63 (zero
// weights for the row past the window), so a lane's address is (per-tile row base of row d, already including its half's RH rows)
// + a COMPILE-TIME immediate (dx * CS + s) * 4: one raw buffer load per k-pair, nothing else.  (K + 1) / K more MFMAs (7x7: +14 %).
template <int K, int CS, int EPI>
__global__ __launch_bounds__(256) void thin_in_mfma_pad_kernel(const float* __restrict__ inp, const float* __restrict__ Wp,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int npix, int B, int H, int W, int flip, int Cw, int lgH, int lgW,
                                                               const ThinEpi epi) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int RH = (K + 1) / 2;
    constexpr int NQ = RH * K * CS;                       // k-pairs
    constexpr int TI_OOB = 0x7FFFFFF0;
    __shared__ float wsh[2 * NQ][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = blockIdx.y * 64;
    const int h = lane >> 5, j = lane & 31;
    for (int e = threadIdx.x; e < 2 * NQ * 64; e += 256) {
        const int kk = e >> 6, c = e & 63;
        const int q = kk >> 1, hh = kk & 1;
        const int d = q / (K * CS), rem = q - d * (K * CS), dx = rem / CS, sc = rem - dx * CS;
        const int dy = d + hh * RH;
        float v = 0.f;
        if (dy < K) {
            const int tap = dy * K + dx;
            v = Wp[(size_t)((flip ? K * K - 1 - tap : tap) * CS + sc) * Cw + cb + c];
        }
        wsh[kk][c] = v;
    }
    __syncthreads();
    const int Wpd = W + 2 * PAD;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)inp, 0, B * (H + 2 * PAD) * Wpd * CS * 4, 0x00020000);
    const int obytes = npix * Cw * 4;                      // (the launcher guarantees < 2 GiB)
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? epi.x : out), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 1 ? epi.y : out), 0, obytes, 0x00020000);
    const float b0 = bias ? bias[cb + j] : 0.f, b1 = bias ? bias[cb + 32 + j] : 0.f;
    const int ntiles = (npix + 31) / 32;
    const float esl = EPI ? epi.slope[0] : 1.f;
    float es = 0.f;
    constexpr int GQ = K * CS;                            // one window row (both halves) per group: 21 loads, 42 MFMAs (7x7x3)
    auto rowbase = [&](int tile, int d) -> int {          // byte offset of padded row (y + d + h RH), column x, of the lane's pixel
        const int pix = tile * 32 + j;
        if (pix >= npix) return TI_OOB;
        const int x = pix & (W - 1), row = pix >> lgW, b = row >> lgH;      // row = b * H + y; padded row of (y + dy - PAD) = row + 2 PAD b + dy
        return ((row + b * (2 * PAD) + d + h * RH) * Wpd + x) * (CS * 4);
    };
    auto gather = [&](int rb, float (&av)[GQ]) {
#pragma unroll
        for (int u = 0; u < GQ; ++u) av[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irs, rb + u * 4, 0, 0));
    };
    // the window rows' gathers run one row ahead of the MFMAs -- ACROSS tiles when RH is even (the ping-pong parity then carries
    // over): row 0 of the next tile is requested during the last row of this one, i.e. BEFORE this tile's 32 stores, so that its
    // MFMAs do not wait behind them on the in-order vmcnt (575 -> 495 us came from the address-free gathers; this is the rest)
    constexpr bool XTILE = (RH % 2) == 0;
    float avA[GQ], avB[GQ];
    if (XTILE) gather(rowbase(blockIdx.x * 4 + wave, 0), avA);
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        tw_f32x16 acc0, acc1;
        float px0[16], px1[16];
        // epilogue addressing without per-element arithmetic: element r of the tile sits at (lane part) + (wave-uniform part of r),
        // so the 2 x 16 loads of x and the 2 x 16 (+ 2 x 16) stores take the lane part as voffset and the r part as SCALAR offset
        // (64-bit pointer arithmetic per element was ~3 VALU instructions each: more issue slots than the tile's MFMAs)
        const bool fullt = tile * 32 + 32 <= npix;                                    // wave-uniform
        const int vlane = ((tile * 32 + 4 * h) * Cw + cb + j) * 4;
        if constexpr (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ((r & 3) + 8 * (r >> 2)) * Cw * 4;
                const int vo = (fullt || tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < npix) ? vlane : TI_OOB;
                px0[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vo, so, 0));
                px1[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, vo + 128, so, 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
        // window rows d = 0 .. RH-1, software-pipelined: the loads of row d + 1 are in flight behind the MFMAs of row d
        if (!XTILE) gather(rowbase(tile, 0), avA);
#pragma unroll
        for (int d = 0; d < RH; ++d) {
            float (&cur)[GQ] = (d & 1) ? avB : avA;
            float (&nxt)[GQ] = (d & 1) ? avA : avB;
            if (d + 1 < RH) gather(rowbase(tile, d + 1), nxt);
            else if (XTILE) gather(rowbase(tile + gridDim.x * 4, 0), nxt);     // (past the last tile: out-of-range offsets, zeros)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GQ; ++u) {
                const int kk = 2 * (d * GQ + u) + h;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u], wsh[kk][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u], wsh[kk][32 + j], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = ((r & 3) + 8 * (r >> 2)) * Cw * 4;
            const int vo = (fullt || tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < npix) ? vlane : TI_OOB;
            const float v0 = acc0[r], v1 = acc1[r];
            if constexpr (EPI == 2) {
                const float x0 = px0[r], x1 = px1[r];
                es = fmaf(x0 > 0.f ? 0.f : x0, v0, es);                 // (an element past the end read x = 0 and v is finite)
                es = fmaf(x1 > 0.f ? 0.f : x1, v1, es);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x0 > 0.f ? v0 : esl * v0), ors, vo, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x1 > 0.f ? v1 : esl * v1), ors, vo + 128, so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), ors, vo, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), ors, vo + 128, so, 0);
                if constexpr (EPI == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0 > 0.f ? v0 : esl * v0), yrs, vo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1 > 0.f ? v1 : esl * v1), yrs, vo + 128, so, 0);
                }
            }
        }
    }
    if constexpr (EPI == 2) thin_epi_finish(epi, es, lane, wave);
}
// A/B switch (round 3; default on): FG_THIN_PADDED=0 keeps the bounds-checked gathers of the 5x5 / 7x7 thin kernels
static bool fg_thin_padded_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FG_THIN_PADDED"); v = e ? atoi(e) : 1; }
    return v != 0;
}

int fg_launch_thin_in_conv(fg_ctx* ctx, const float* in, const float* Wp, const float* bias, float* out, int B, int H,
                           int W, int Cs, int Cw, int k, int flip, const FgActFuse* actf, const FgActBwd* actb,
                           float* padbuf, long long padbuf_floats) {
    if (actf) actf->applied = 0;
    if (actb) actb->applied = 0;
    // profile label "thin_in<k,Cs>": algorithmic bytes = the thin input read once + the wide output written once (a fused PReLU
    // forward writes it twice, a fused PReLU backward also reads the PReLU's input)
    char plabel[48] = "";
    if (ctx->prof) snprintf(plabel, sizeof(plabel), "thin_in<%d,%d>", k, Cs);
    // (alg_flops = the layer's 2 x MACs: the 5x5 / 7x7 instances are bound by the padded matrix pipe, not by these bytes -- bench.py
    // files them under roofline.mfma_padded with their useful-FLOP fraction)
    FgProfScope prof(ctx, fg_intern(ctx, plabel), 2.0 * B * H * W * (double)k * k * Cs * Cw, 0.0,
                     4.0 * B * H * W * ((double)Cs + (double)Cw * (1 + ((actf && actf->y) || (actb && actb->x) ? 1 : 0))));
    ThinEpi epi; memset(&epi, 0, sizeof(epi));
    // fold the neighbouring PReLU into the epilogue (MFMA variants): forward = plain PReLU only (no same-shape mask)
    const bool want_f = actf && actf->y && actf->slope && !actf->mask && fg_fuse_prelu(ctx);
    const bool want_b = !want_f && actb && actb->x && actb->slope && !actb->mask && fg_fuse_prelu(ctx);
    auto arm = [&](dim3 g) -> bool {      // false: the slope-gradient partials do not fit the deferred arena -> not folded
        if (want_f) { epi.slope = actf->slope; epi.y = actf->y; return true; }
        if (want_b) {
            const long long np = (long long)g.x * g.y * 4;
            float* dp = actb->gslope ? fg_defer_alloc(ctx, np) : nullptr;
            if (actb->gslope && !dp) return false;
            epi.slope = actb->slope; epi.x = actb->x; epi.part = dp;
            if (dp) fg_defer_push(ctx, dp, (int)np, 1, 0.f, actb->gslope);
        }
        return true;
    };
    auto armed = [&]() { if (epi.y) actf->applied = 1; if (epi.x) actb->applied = 1; };
    if (Cw % 64) return fg_set_err(ctx, FG_ERR_INVALID, "thin_in: Cw %% 64");
    const int cblk = Cw >= 256 ? 256 : Cw;
    if (Cw % cblk || 256 % cblk) return fg_set_err(ctx, FG_ERR_INVALID, "thin_in: Cw=%d unsupported", Cw);
    const int npix = B * H * W;
    if (npix == 0) return FG_OK;
    dim3 grid(fg_cdiv(npix, 128), Cw / cblk);
    if (k == 3 && (Cs == 1 || Cs == 3)) {
        int lgH = -1, lgW = -1;
        for (int q = 0; q < 15; ++q) { if ((1 << q) == H) lgH = q; if ((1 << q) == W) lgW = q; }
        // tiles per wave: every wave first fetches its 2 x 14 weight fragments, so one tile per wave is all set-up latency
        // (1 -> 4 tiles per wave with the next tile's gather in flight: 17.6 -> 15.5 us for the 33 / 67 MB outputs)
#ifdef FG_MEASURE       // measurement build only: tiles per wave, non-temporal stores, and the (wrong-result) DBG variants
        static int tpw_env = -1, nt_env = -1;
        if (tpw_env < 0) { const char* e = getenv("FG_THIN_TPW"); tpw_env = e ? atoi(e) : 0; }
        if (nt_env < 0) {
            const char* e = getenv("FG_THIN_NT"); nt_env = e ? (atoi(e) & 1) : 0;
            const char* dbg = getenv("FG_THIN_DBG"); if (dbg) nt_env |= (atoi(dbg) & 3) << 1;
        }
        const int tpw = tpw_env > 0 ? (tpw_env > 16 ? 16 : tpw_env) : 4;
        const int nt_store = nt_env;
#else
        const int tpw = 4, nt_store = 0;
#endif
        int nb = fg_cdiv(fg_cdiv(npix, 32), 4 * tpw);
        if (nb > 2048) nb = 2048;
        if (nb < 1) nb = 1;
        dim3 mgrid(nb, Cw / 64);
        arm(mgrid);
        const int em = epi.x ? 2 : (epi.y ? 1 : 0);
#define TIM(CC, EE) hipLaunchKernelGGL((thin_in_mfma_kernel<CC, EE>), mgrid, dim3(256), 0, ctx->stream, in, Wp, bias, out, npix, H, W, flip, Cw, lgH, lgW, epi, nt_store)
        if (Cs == 3) { if (em == 2) TIM(3, 2); else if (em == 1) TIM(3, 1); else TIM(3, 0); }
        else { if (em == 2) TIM(1, 2); else if (em == 1) TIM(1, 1); else TIM(1, 0); }
#undef TIM
        FG_CHECK_LAUNCH(ctx);
        armed();
        return FG_OK;
    }
    if ((k == 3 && Cs == 4) || ((k == 5 || k == 7) && (Cs == 1 || Cs == 3))) {      // exactly the TIL(...) instances below
        int lgH = -1, lgW = -1;
        for (int q = 0; q < 15; ++q) { if ((1 << q) == H) lgH = q; if ((1 << q) == W) lgW = q; }
        int nb = fg_cdiv(fg_cdiv(npix, 32), 4);
        if (nb > 1024) nb = 1024;
        dim3 mgrid(nb, Cw / 64);
        arm(mgrid);
        const int em = epi.x ? 2 : (epi.y ? 1 : 0);
        {   // 5x5 / 7x7 on power-of-two maps with room for a zero-bordered copy of the input: no gather arithmetic at all
            const int pad = (k - 1) / 2;
            const long long padf = (long long)B * (H + 2 * pad) * (W + 2 * pad) * Cs;
            if (k >= 5 && lgH >= 0 && lgW >= 0 && padbuf && padf <= padbuf_floats && padf < (1LL << 28) && (long long)npix * Cw * 4 < 0x7FFFFFF0LL &&
                fg_thin_padded_on()) {
                const int rcp = fg_launch_thin_pad(ctx, in, padbuf, B, H, W, Cs, pad);
                if (rcp) return rcp;
#define TIP(KK, CC)                                                                                                  \
                if (k == KK && Cs == CC) {                                                                           \
                    if (em == 2) hipLaunchKernelGGL((thin_in_mfma_pad_kernel<KK, CC, 2>), mgrid, dim3(256), 0, ctx->stream, padbuf, Wp, bias, out, npix, B, H, W, flip, Cw, lgH, lgW, epi); \
                    else if (em == 1) hipLaunchKernelGGL((thin_in_mfma_pad_kernel<KK, CC, 1>), mgrid, dim3(256), 0, ctx->stream, padbuf, Wp, bias, out, npix, B, H, W, flip, Cw, lgH, lgW, epi); \
                    else hipLaunchKernelGGL((thin_in_mfma_pad_kernel<KK, CC, 0>), mgrid, dim3(256), 0, ctx->stream, padbuf, Wp, bias, out, npix, B, H, W, flip, Cw, lgH, lgW, epi); \
                    FG_CHECK_LAUNCH(ctx);                                                                            \
                    armed();                                                                                         \
                    return FG_OK;                                                                                    \
                }
                TIP(5, 1) TIP(5, 3) TIP(7, 1) TIP(7, 3)
#undef TIP
            }
        }
#define TIL(KK, CC)                                                                                                  \
        if (k == KK && Cs == CC) {                                                                                   \
            if (em == 2) hipLaunchKernelGGL((thin_in_mfma_lds_kernel<KK, CC, 2>), mgrid, dim3(256), 0, ctx->stream, in, Wp, bias, out, npix, H, W, flip, Cw, lgH, lgW, epi); \
            else if (em == 1) hipLaunchKernelGGL((thin_in_mfma_lds_kernel<KK, CC, 1>), mgrid, dim3(256), 0, ctx->stream, in, Wp, bias, out, npix, H, W, flip, Cw, lgH, lgW, epi); \
            else hipLaunchKernelGGL((thin_in_mfma_lds_kernel<KK, CC, 0>), mgrid, dim3(256), 0, ctx->stream, in, Wp, bias, out, npix, H, W, flip, Cw, lgH, lgW, epi); \
            FG_CHECK_LAUNCH(ctx);                                                                                    \
            armed();                                                                                                 \
            return FG_OK;                                                                                            \
        }
        TIL(3, 4) TIL(5, 1) TIL(5, 3) TIL(7, 1) TIL(7, 3)
#undef TIL
    }
    if (k == 3 && (Cs == 1 || Cs == 3 || Cs == 4) && (Cw == 64 || Cw == 128)) {
        dim3 rgrid(fg_cdiv(B * H, 4), 1);
#define TIR3(CC, JJ)                                                                                                 \
    if (Cs == CC && Cw == JJ * 64) {                                                                                 \
        hipLaunchKernelGGL((thin_in_row3_kernel<CC, JJ>), rgrid, dim3(256), 0, ctx->stream, in, Wp, bias, out, B, H, W, \
                           flip, Cw);                                                                                \
        FG_CHECK_LAUNCH(ctx);                                                                                        \
        return FG_OK;                                                                                                \
    }
        TIR3(1, 1) TIR3(3, 1) TIR3(4, 1) TIR3(1, 2) TIR3(3, 2) TIR3(4, 2)
#undef TIR3
    }
    {
        int nblk = fg_cdiv(npix, 4);
        if (nblk > 4096) nblk = 4096;
#define TI(KK, CC, JJ)                                                                                               \
    if (k == KK && Cs == CC && Cw % (JJ * 64) == 0 && (JJ == 1 || Cw == JJ * 64)) {                                 \
        hipLaunchKernelGGL((thin_in_kernel<KK, CC, JJ>), dim3(nblk, Cw / (JJ * 64)), dim3(256), 0, ctx->stream, in, Wp, \
                           bias, out, B, H, W, flip, Cw);                                                           \
        FG_CHECK_LAUNCH(ctx);                                                                                       \
        return FG_OK;                                                                                               \
    }
        TI(3, 1, 2) TI(3, 3, 2) TI(3, 4, 2) TI(3, 1, 1) TI(3, 3, 1) TI(3, 4, 1)
    }
    if ((k == 5 || k == 7) && (Cs == 1 || Cs == 3)) {
        const int pad = (k - 1) / 2;
        const int wp = (W + 2 * pad + 3) / 4 * 4 + 4;
        const size_t lds = (size_t)(4 + 2 * pad) * wp * Cs * sizeof(float) + 64;
        dim3 rgrid(B * ((H + 3) / 4), Cw / 64);
#define TIR(KK, CC)                                                                                                  \
    if (k == KK && Cs == CC) {                                                                                       \
        hipLaunchKernelGGL((thin_in_rows_kernel<KK, CC>), rgrid, dim3(256), lds, ctx->stream, in, Wp, bias, out, B, H, W, \
                           flip, Cw);                                                                                \
        FG_CHECK_LAUNCH(ctx);                                                                                        \
        return FG_OK;                                                                                                \
    }
        TIR(5, 1) TIR(5, 3) TIR(7, 1) TIR(7, 3)
#undef TIR
    }
#undef TI
    hipLaunchKernelGGL(thin_in_generic_kernel, grid, dim3(256), 0, ctx->stream, in, Wp, bias, out, B, H, W, Cs, Cw, k,
                       flip, cblk);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---------------------------------------------------------------------------------
// thin-out: out[pix][s] = act(bias[s] + sum_{tap,c} in[pix + off(tap)][c] * Wp[tap][s][c])   (Cw wide, Cs small)
// one wave per pixel; lane owns channels lane + 64 j; butterfly reduction of the Cs partials.
// ---------------------------------------------------------------------------------
#define FG_OOB_T 0x7FFFFFF0
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// NP pixels per wave iteration; every tap is fetched with a raw-buffer load whose offset is pushed out of range for
// padding taps (hardware returns 0), so the NP*K*K*CJ loads of one iteration are branch-free and all in flight.
template <int K, int CJ, int CS, int NP>
__global__ __launch_bounds__(256) void thin_out_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B,
                                                       int H, int W, int flip, int sigmoid, int in_bytes) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int Cw = CJ * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    float w[K * K][CS][CJ];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
            for (int j = 0; j < CJ; ++j) w[t][s][j] = Wp[((size_t)t * CS + s) * Cw + lane + 64 * j];
    const int npix = B * H * W;
    const int ngrp = (npix + NP - 1) / NP;
    const int nwaves = gridDim.x * 4;
    for (int grp = blockIdx.x * 4 + wave; grp < ngrp; grp += nwaves) {
        float acc[NP][CS];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s = 0; s < CS; ++s) acc[q][s] = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int pix = grp * NP + q;
            const bool pv = pix < npix;
            const int x = pix % W;
            const int t = pix / W;
            const int y = t % H;
            const int b = t / H;
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
                const int yy = y + (flip ? PAD - dy : dy - PAD);
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const int xx = x + (flip ? PAD - dx : dx - PAD);
                    const bool ok = pv && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                    const int base = ok ? (((b * H + yy) * W + xx) * Cw + lane) * 4 : FG_OOB_T;
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 256 * j, 0, 0));
#pragma unroll
                        for (int s = 0; s < CS; ++s) acc[q][s] = fmaf(v, w[dy * K + dx][s][j], acc[q][s]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s = 0; s < CS; ++s) acc[q][s] = wave_sum_x(acc[q][s]);
        if (lane < NP * CS) {
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int s = 0; s < CS; ++s) r = (lane == q * CS + s) ? acc[q][s] : r;
            const int s = lane % CS;
            r += bias ? bias[s] : 0.f;
            if (sigmoid) r = 1.f / (1.f + expf(-r));
            const int pix = grp * NP + lane / CS;
            if (pix < npix) out[(size_t)pix * CS + s] = r;     // NP*CS consecutive floats per group
        }
    }
}
// LDS-tiled thin-out for large kernels / many input channels (c2f generator head: 7x7, 256 -> 3, models_c2f.lua:131).
// Block = 16x16 output pixels (4 waves x 8x8), lane = pixel.  Per 32-channel chunk the (16+K-1)^2 input halo tile is
// staged in LDS (rows padded to 36 floats); each lane reads its taps as ds_read_b128 (4 channels) and multiplies them
// with wave-uniform weights (scalar loads -> SGPR operands): 4*CS FMAs per LDS read.
template <int K, int CS>
__global__ __launch_bounds__(256) void thin_out_tiled_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int H, int W, int Cw, int flip, int sigmoid) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int TP = 16 + K - 1;   // tile edge incl. halo
    constexpr int LDC = 36;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [TP*TP][LDC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    int bid = blockIdx.x;
    const int txi = bid % tiles_x; bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = txi * 16, y0 = tyi * 16;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    float acc[CS];
#pragma unroll
    for (int s = 0; s < CS; ++s) acc[s] = 0.f;
    for (int c0 = 0; c0 < Cw; c0 += 32) {
        __syncthreads();
        for (int i = threadIdx.x; i < TP * TP * 8; i += 256) {
            const int c4 = i & 7, pp = i >> 3;
            const int px = pp % TP, py = pp / TP;
            const int xx = x0 + px - PAD, yy = y0 + py - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H)
                v = *(const float4*)(in + ((size_t)(b * H + yy) * W + xx) * Cw + c0 + c4 * 4);
            *(float4*)(tile + pp * LDC + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int dy = 0; dy < K; ++dy) {
#pragma unroll 1
            for (int dx = 0; dx < K; ++dx) {
                const int ty = ly + (flip ? 2 * PAD - dy : dy), tx = lx + (flip ? 2 * PAD - dx : dx);
                const float* tp = tile + (ty * TP + tx) * LDC;
                const float* wp = Wp + (size_t)(dy * K + dx) * CS * Cw + c0;   // wave-uniform
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const float4 v = *(const float4*)(tp + c4 * 4);
#pragma unroll
                    for (int s = 0; s < CS; ++s) {
                        const float* ws = wp + (size_t)s * Cw + c4 * 4;
                        acc[s] = fmaf(v.x, ws[0], acc[s]);
                        acc[s] = fmaf(v.y, ws[1], acc[s]);
                        acc[s] = fmaf(v.z, ws[2], acc[s]);
                        acc[s] = fmaf(v.w, ws[3], acc[s]);
                    }
                }
            }
        }
    }
    const int ox = x0 + lx, oy = y0 + ly;
    if (ox < W && oy < H) {
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            float r = acc[s] + (bias ? bias[s] : 0.f);
            if (sigmoid) r = 1.f / (1.f + expf(-r));
            out[((size_t)(b * H + oy) * W + ox) * CS + s] = r;
        }
    }
}

__global__ __launch_bounds__(256) void thin_out_generic_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int B, int H, int W, int Cw, int Cs, int K, int flip,
                                                               int sigmoid) {
    const int PAD = (K - 1) / 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int npix = B * H * W;
    const int p0 = blockIdx.x * 64;
    for (int q = wid; q < 64; q += 4) {
        const int pix = p0 + q;
        if (pix >= npix) break;
        const int x = pix % W;
        const int t = pix / W;
        const int y = t % H;
        const int b = t / H;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int dy = 0; dy < K; ++dy) {
            const int yy = y + (flip ? PAD - dy : dy - PAD);
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = 0; dx < K; ++dx) {
                const int xx = x + (flip ? PAD - dx : dx - PAD);
                if ((unsigned)xx >= (unsigned)W) continue;
                const float* ip = in + ((size_t)(b * H + yy) * W + xx) * Cw;
                const float* wp = Wp + (size_t)(dy * K + dx) * Cs * Cw;
                for (int c = lane; c < Cw; c += 64) {
                    const float v = ip[c];
                    for (int s = 0; s < Cs; ++s) acc[s] = fmaf(v, wp[(size_t)s * Cw + c], acc[s]);
                }
            }
        }
        for (int s = 0; s < Cs; ++s) {
            float r = wave_sum_x(acc[s]);
            if (lane == 0) {
                r += bias ? bias[s] : 0.f;
                if (sigmoid) r = 1.f / (1.f + expf(-r));
                out[(size_t)pix * Cs + s] = r;
            }
        }
    }
}

// 3x3 thin-output convolution, sliding window: one wave computes NP = 4 horizontally adjacent output pixels from the
// (NP + 2) x 3 input pixels they share -- 18 coalesced 256-byte channel-row loads instead of 36 (thin_out_kernel loads all
// nine taps of every pixel separately and is bound by the vector-memory pipe).  Needs W % 4 == 0.
template <int CJ, int CS>
__global__ __launch_bounds__(256) void thin_out_win_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out, int B,
                                                           int H, int W, int flip, int sigmoid, int in_bytes) {
    constexpr int NP = 4;
    constexpr int Cw = CJ * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    float w[9][CS][CJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
            for (int j = 0; j < CJ; ++j) w[t][s][j] = Wp[((size_t)(flip ? 8 - t : t) * CS + s) * Cw + lane + 64 * j];
    const int ngrp = B * H * (W / NP);
    const int nwaves = gridDim.x * 4;
    for (int grp = blockIdx.x * 4 + wave; grp < ngrp; grp += nwaves) {
        const int xg = grp % (W / NP);
        const int t = grp / (W / NP);              // b * H + y
        const int y = t % H;
        const int x0 = xg * NP;
        float acc[NP][CS];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s = 0; s < CS; ++s) acc[q][s] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool rok = (unsigned)yy < (unsigned)H;
            float v[NP + 2][CJ];
#pragma unroll
            for (int c = 0; c < NP + 2; ++c) {
                const int xx = x0 + c - 1;
                const bool ok = rok && (unsigned)xx < (unsigned)W;
                const int base = ok ? (((t - y + yy) * W + xx) * Cw + lane) * 4 : FG_OOB_T;
#pragma unroll
                for (int j = 0; j < CJ; ++j) v[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 256 * j, 0, 0));
            }
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int j = 0; j < CJ; ++j)
#pragma unroll
                        for (int s = 0; s < CS; ++s) acc[q][s] = fmaf(v[q + dx][j], w[dy * 3 + dx][s][j], acc[q][s]);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s = 0; s < CS; ++s) acc[q][s] = wave_sum_x(acc[q][s]);
        if (lane < NP * CS) {
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int s = 0; s < CS; ++s) r = (lane == q * CS + s) ? acc[q][s] : r;
            const int s = lane % CS;
            r += bias ? bias[s] : 0.f;
            if (sigmoid) r = 1.f / (1.f + expf(-r));
            out[((size_t)t * W + x0) * CS + lane] = r;       // NP*CS consecutive floats per group
        }
    }
}

// 3x3 thin-output convolution on the matrix pipe, one pass (replaces thin_out_win_kernel where the shape allows: that kernel
// re-loads every input pixel 4.5 times and is VALU-bound, 2.4 TB/s): a block owns R full-width output rows of one sample.
//   phase 1: Z[q][n] = sum_c in[q][c] * Wt[c][n] for the (R + 2) x W input pixels q of the slab and the 9*CS <= 27 columns
//            n = (tap, s) -- one 32-pixel x 32-column MFMA tile per wave pass, K = Cw.  A is read straight from global with
//            the K permutation of the igemm kernels (lane (pixel i, half h) loads the float4 of channels 8jj + 4h .. + 3, so a
//            pixel's 512-byte channel row is consumed whole by the two half-waves over the jj steps); B = the packed
//            weights in registers; rows outside the image load zeros (raw-buffer range check) and give Z = 0.
//   phase 2: out[y][x][s] = act(bias[s] + sum_tap Z[(y + dy - 1, x + dx - 1)][tap * CS + s]) gathered from the slab's Z in
//            LDS (row stride 29 floats: conflict-free column reads).
// Every input row is read once per block that needs it (1 + 2/R times from L2, once from HBM).
template <int CJ, int CS>
__global__ __launch_bounds__(256) void thin_out_slab_mfma_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int H, int W, int R, int flip, int sigmoid, int in_bytes) {
    constexpr int Cw = CJ * 64;
    constexpr int NJ = Cw / 8;                    // float4 loads per lane and tile
    constexpr int ZLD = 29;
    extern __shared__ __attribute__((aligned(16))) float tos_sm[];     // Z [(R + 2) * W][ZLD]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nrb = (H + R - 1) / R;
    const int b = blockIdx.x / nrb, y0 = (blockIdx.x - b * nrb) * R;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    // B fragments: lane (column n = i, half h) holds Wt[c = 8jj + 4h + t][n] = Wp[tap'][s][c], n = tap * CS + s
    float wreg[NJ][4];
    {
        const int tap = i / CS, sc = i - tap * CS;
        const bool live = i < 9 * CS;
        const float* wp = Wp + ((size_t)(flip ? 8 - tap : tap) * CS + sc) * Cw + 4 * h;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const tw_f32x4 v = live ? *(const tw_f32x4*)(wp + 8 * jj) : tw_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) wreg[jj][t] = v[t];
        }
    }
    const int ntile = (R + 2) * W / 32;
    auto load_a = [&](int tile, tw_f32x4 (&a)[NJ]) {
        const int q = tile * 32 + i;                   // slab pixel of this lane's A row
        const int qr = q / W, qx = q - qr * W;
        const int yy = y0 - 1 + qr;
        const int voff = ((unsigned)yy < (unsigned)H) ? (((b * H + yy) * W + qx) * Cw + 4 * h) * 4 : FG_OOB_T;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
            a[jj] = __builtin_bit_cast(tw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, jj * 32, 0));
    };
    tw_f32x4 av[NJ], an[NJ];
    if (wave < ntile) load_a(wave, av);
    for (int tile = wave; tile < ntile; tile += 4) {
        if (tile + 4 < ntile) load_a(tile + 4, an);    // the next tile's rows stream in behind this tile's 8 * CJ * 4 MFMAs
        tw_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj][t], wreg[jj][t], acc, 0, 0, 0);
        if (i < 9 * CS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tos_sm[(tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * ZLD + i] = acc[r];
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) av[jj] = an[jj];
    }
    __syncthreads();
    const int nout = R * W;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int ry = o / W, x = o - ry * W;
        const int y = y0 + ry;
        if (y >= H) continue;
        float r[CS];
#pragma unroll
        for (int sc = 0; sc < CS; ++sc) r[sc] = bias ? bias[sc] : 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                if ((unsigned)xx >= (unsigned)W) continue;
                const float* z = tos_sm + ((ry + dy) * W + xx) * ZLD + (dy * 3 + dx) * CS;
#pragma unroll
                for (int sc = 0; sc < CS; ++sc) r[sc] += z[sc];
            }
        float* op = out + ((size_t)(b * H + y) * W + x) * CS;
#pragma unroll
        for (int sc = 0; sc < CS; ++sc) op[sc] = sigmoid ? 1.f / (1.f + expf(-r[sc])) : r[sc];
    }
}

// 5x5 / 7x7 thin-OUTPUT convolution (e.g. 256 -> 3, 7x7: the c2f generator head, models_c2f.lua:131) in two passes:
//   (1) R[pix][(dx, s)] = sum_{dy, c} in[y + dy - PAD][x][c] * W[dy][dx][c][s]   -- only the VERTICAL taps are folded into
//       the contraction (K = K*Cw), so the N axis is the K*CS <= 21 (dx, s) columns of one 32-wide MFMA tile instead of 3;
//       a block owns 4 output rows x 32 columns, stages the (4 + K - 1) input rows and the weights 32 channels at a time
//       in LDS (conflict-free b128 / b32 fragment reads), one wave per output row;
//   (2) out[y][x][s] = bias[s] + sum_dx R[y][x + dx - PAD][(dx, s)]   -- a K-tap horizontal gather of the small R.
// The VALU kernel (thin_out_tiled_kernel) needs a scalar weight load and an LDS read per 4 FMAs and ran at 25 TFLOP/s.
template <int K, int CS>
__global__ __launch_bounds__(256) void thin_out_rows_mfma_kernel(const float* __restrict__ in, const float* __restrict__ Wp,
                                                                 float* __restrict__ R, int B, int H, int W, int Cw) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NR = 4 + K - 1;                 // staged input rows
    constexpr int LDC = 36;                       // floats per staged pixel (32 channels + 4 pad)
    constexpr int NJ = K * CS;                    // live columns of the 32-wide tile
    extern __shared__ __attribute__((aligned(16))) float tor_sm[];
    float* xs = tor_sm;                            // [NR][32][LDC]
    float* wsm = tor_sm + NR * 32 * LDC;           // [dy][c][j]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 3) / 4;
    int bid = blockIdx.x;
    const int txi = bid % tiles_x; bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = txi * 32, y0 = tyi * 4;
    // Round 2b: (i) the chunk of step c0 + 32 is fetched into registers while the chunk of step c0 multiplies (load and
    // multiply phases used to alternate behind the block barriers); (ii) the weights are read with the channel fastest (128-byte
    // runs instead of 4-byte words Cw floats apart) into [dy][c][33] (conflict-free both ways); (iii) two accumulators, so
    // that consecutive MFMAs do not wait for each other's result.
    constexpr int WLD = 33;
    constexpr int NX = (NR * 32 * 8 + 255) / 256;     // float4 per thread and chunk of the input rows
    constexpr int NWV = K * 32 * 32 / 256;            // weight floats per thread and chunk
    tw_f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float4 px[NX];
    float pw[NWV];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int c4 = e & 7, pp = e >> 3;
            const int pxx = pp & 31, py = pp >> 5;
            const int xx = x0 + pxx, yy = y0 + py - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NR * 32 * 8 && xx < W && (unsigned)yy < (unsigned)H)
                v = *(const float4*)(in + ((size_t)(b * H + yy) * W + xx) * Cw + c0 + c4 * 4);
            px[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int c = e & 31, j = (e >> 5) & 31, dy = e >> 10;
            pw[u] = (j < NJ) ? Wp[(size_t)((dy * K + j / CS) * CS + j % CS) * Cw + c0 + c] : 0.f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cw; c0 += 32) {
        __syncthreads();                                   // the previous chunk's fragment reads are done
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < NR * 32 * 8) *(float4*)(xs + (e >> 3) * LDC + (e & 7) * 4) = px[u];
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int c = e & 31, j = (e >> 5) & 31, dy = e >> 10;
            wsm[(dy * 32 + c) * WLD + j] = pw[u];
        }
        __syncthreads();
        if (c0 + 32 < Cw) fetch(c0 + 32);                  // in flight behind this chunk's K * 16 MFMAs
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            const float* xrow = xs + ((wave + dy) * 32 + i) * LDC + 4 * h;
            const float* wrow = wsm + dy * 32 * WLD + i;                // + c * WLD
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xv = *(const float4*)(xrow + 8 * q);       // channels 8q + 4h + {0..3}
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.x, wrow[(8 * q + 4 * h + 0) * WLD], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.y, wrow[(8 * q + 4 * h + 1) * WLD], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.z, wrow[(8 * q + 4 * h + 2) * WLD], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.w, wrow[(8 * q + 4 * h + 3) * WLD], acc2, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    const int y = y0 + wave;
    if (y < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (x < W) R[((size_t)(b * H + y) * W + x) * 32 + i] = acc[r];
        }
    }
}
template <int K, int CS>
__global__ __launch_bounds__(256) void thin_out_rows_gather_kernel(const float* __restrict__ R, const float* __restrict__ bias,
                                                                   float* __restrict__ out, int npix, int W, int sigmoid) {
    constexpr int PAD = (K - 1) / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix * CS) return;
    const int pix = idx / CS, s = idx - pix * CS;
    const int x = pix % W;
    float r = bias ? bias[s] : 0.f;
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
        const int xx = x + dx - PAD;
        if ((unsigned)xx < (unsigned)W) r += R[(size_t)(pix + dx - PAD) * 32 + dx * CS + s];
    }
    if (sigmoid) r = 1.f / (1.f + expf(-r));
    out[idx] = r;
}

int fg_launch_thin_out_conv(fg_ctx* ctx, const float* in, const float* Wp, const float* bias, float* out, int B, int H,
                            int W, int Cw, int Cs, int k, int flip, int sigmoid, float* rbuf, long long rbuf_floats) {
    if (Cs > 4) return fg_set_err(ctx, FG_ERR_INVALID, "thin_out: Cs > 4");
    const int npix = B * H * W;
    if (npix == 0) return FG_OK;
    // profile label "thin_out<k,Cs>": algorithmic bytes = the wide input read once + the thin output written once
    char plabel[48] = "";
    if (ctx->prof) snprintf(plabel, sizeof(plabel), "thin_out<%d,%d>", k, Cs);
    FgProfScope prof(ctx, fg_intern(ctx, plabel), 2.0 * npix * (double)k * k * Cs * Cw, 0.0, 4.0 * npix * ((double)Cw + Cs));
    if (rbuf && !flip && (k == 5 || k == 7) && (Cs == 1 || Cs == 3) && Cw % 32 == 0 && (long long)npix * 32 <= rbuf_floats) {
        dim3 grid(B * ((H + 3) / 4) * ((W + 31) / 32));
#define TOR(KK, CC)                                                                                                  \
        if (k == KK && Cs == CC) {                                                                                   \
            const size_t lds = (size_t)((4 + KK - 1) * 32 * 36 + KK * 32 * 33) * sizeof(float);                      \
            static bool attr = false;                                                                                \
            if (!attr) {                                                                                             \
                FG_HIP(ctx, hipFuncSetAttribute((const void*)thin_out_rows_mfma_kernel<KK, CC>,                      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
                attr = true;                                                                                         \
            }                                                                                                        \
            hipLaunchKernelGGL((thin_out_rows_mfma_kernel<KK, CC>), grid, dim3(256), lds, ctx->stream, in, Wp, rbuf, B, H, W, Cw); \
            FG_CHECK_LAUNCH(ctx);                                                                                    \
            hipLaunchKernelGGL((thin_out_rows_gather_kernel<KK, CC>), dim3(fg_cdiv(npix * CC, 256)), dim3(256), 0,    \
                               ctx->stream, rbuf, bias, out, npix, W, sigmoid);                                      \
            FG_CHECK_LAUNCH(ctx);                                                                                    \
            return FG_OK;                                                                                            \
        }
        TOR(5, 1) TOR(5, 3) TOR(7, 1) TOR(7, 3)
#undef TOR
    }
    dim3 grid(fg_cdiv(npix, 64));
    const long long in_bytes = (long long)npix * Cw * 4;
    {   // 3x3 on the matrix pipe (fg_set_fusion bit FG_FUSE_THIN_SLAB; off: the sliding-window VALU kernel)
        const bool use_slab = (ctx->fusion & FG_FUSE_THIN_SLAB) != 0;
        // rows per block: ONE round of ~256 blocks fills the chip (measured at B = 128, 32 x 32: R = 16 -> 256 blocks 23 us;
        // R = 8 -> 512 blocks 32 us; R = 6 -> 768 blocks = 1.5 rounds of resident blocks 36 us), Z must fit 80 KB of LDS,
        // (R + 2) * W must be whole 32-pixel tiles; prefer an R that divides H (equal blocks)
        int R = 0;
        if (W >= 4 && W <= 128) {
            const int rmax = (80 * 1024 / (29 * 4)) / W - 2;
            int rt = (int)(((long long)H * B + 255) / 256);
            if (rt > rmax) rt = rmax;
            if (rt > H) rt = H;
            if (rt < 2) rt = 2;
            for (int pass = 0; pass < 2 && !R; ++pass)
                for (int r = rt; r >= 2; --r)
                    if (((r + 2) * W) % 32 == 0 && (pass == 1 || H % r == 0)) { R = r; break; }
            if (!R && rmax >= 2)
                for (int r = rt + 1; r <= rmax; ++r) if (((r + 2) * W) % 32 == 0) { R = r; break; }
        }
        if (use_slab && in_bytes < (long long)FG_OOB_T && k == 3 && R >= 2 && (Cw == 64 || Cw == 128) && Cs <= 3 &&
            ((R + 2) * W) % 32 == 0) {
            const int nblk = B * ((H + R - 1) / R);
            const size_t lds = (size_t)(R + 2) * W * 29 * sizeof(float);
#define TOS(JJ, CC)                                                                                                  \
            if (Cw == JJ * 64 && Cs == CC) {                                                                         \
                static bool attr = false;                                                                            \
                if (!attr) {                                                                                         \
                    FG_HIP(ctx, hipFuncSetAttribute((const void*)thin_out_slab_mfma_kernel<JJ, CC>,                  \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));        \
                    attr = true;                                                                                     \
                }                                                                                                    \
                hipLaunchKernelGGL((thin_out_slab_mfma_kernel<JJ, CC>), dim3(nblk), dim3(256), lds, ctx->stream, in, Wp, bias, \
                                   out, H, W, R, flip, sigmoid, (int)in_bytes);                                      \
                FG_CHECK_LAUNCH(ctx);                                                                                \
                return FG_OK;                                                                                        \
            }
            TOS(1, 1) TOS(1, 3) TOS(2, 1) TOS(2, 3)
#undef TOS
        }
    }
    if (in_bytes < (long long)FG_OOB_T && k == 3 && W % 4 == 0) {
        int nblk = fg_cdiv(fg_cdiv(npix, 4), 4);
        if (nblk > 4096) nblk = 4096;
#define TOW(JJ, CC)                                                                                                  \
        if (Cw == JJ * 64 && Cs == CC) {                                                                             \
            hipLaunchKernelGGL((thin_out_win_kernel<JJ, CC>), dim3(nblk), dim3(256), 0, ctx->stream, in, Wp, bias, out, B, H, \
                               W, flip, sigmoid, (int)in_bytes);                                                     \
            FG_CHECK_LAUNCH(ctx);                                                                                    \
            return FG_OK;                                                                                            \
        }
        TOW(1, 1) TOW(1, 3) TOW(2, 1) TOW(2, 3)
#undef TOW
    }
    if (in_bytes < (long long)FG_OOB_T) {
        int nblk = fg_cdiv(fg_cdiv(npix, 4), 4);
        if (nblk > 4096) nblk = 4096;
#define TO(KK, JJ, CC)                                                                                              \
    if (k == KK && Cw == JJ * 64 && Cs == CC) {                                                                     \
        hipLaunchKernelGGL((thin_out_kernel<KK, JJ, CC, 4>), dim3(nblk), dim3(256), 0, ctx->stream, in, Wp, bias, out, B, \
                           H, W, flip, sigmoid, (int)in_bytes);                                                     \
        FG_CHECK_LAUNCH(ctx);                                                                                       \
        return FG_OK;                                                                                               \
    }
        TO(3, 1, 1) TO(3, 1, 3) TO(3, 2, 1) TO(3, 2, 3)
    }
#undef TO
    if (Cw % 32 == 0 && (k == 3 || k == 5 || k == 7) && (Cs == 1 || Cs == 3)) {
        const int tp = 16 + k - 1;
        const size_t lds = (size_t)tp * tp * 36 * sizeof(float);
        dim3 tgrid(B * ((H + 15) / 16) * ((W + 15) / 16));
#define TOT(KK, CC)                                                                                                  \
    if (k == KK && Cs == CC) {                                                                                       \
        static bool attr = false;                                                                                    \
        if (!attr) {                                                                                                 \
            FG_HIP(ctx, hipFuncSetAttribute((const void*)thin_out_tiled_kernel<KK, CC>,                              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            attr = true;                                                                                             \
        }                                                                                                            \
        hipLaunchKernelGGL((thin_out_tiled_kernel<KK, CC>), tgrid, dim3(256), lds, ctx->stream, in, Wp, bias, out, B, H, \
                           W, Cw, flip, sigmoid);                                                                    \
        FG_CHECK_LAUNCH(ctx);                                                                                        \
        return FG_OK;                                                                                                \
    }
        TOT(3, 1) TOT(3, 3) TOT(5, 1) TOT(5, 3) TOT(7, 1) TOT(7, 3)
#undef TOT
    }
    hipLaunchKernelGGL(thin_out_generic_kernel, grid, dim3(256), 0, ctx->stream, in, Wp, bias, out, B, H, W, Cw, Cs, k,
                       flip, sigmoid);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---------------------------------------------------------------------------------
// thin wgrad: gw[tap][s][c] = sum_pix thin[pix + sgn*off(tap)][s] * wide[pix][c]
// ---------------------------------------------------------------------------------
#define TW_BLOCKS FG_THIN_WGRAD_BLOCKS
#define TW_ROWS 4   // image rows per staged strip
// Block = cblk wide channels x PL pixel lanes.  A strip of TW_ROWS image rows of the THIN operand (+ halo, zero
// padded) is staged in LDS, so the 27 thin reads per pixel are conflict-free LDS broadcasts without bounds checks;
// the wide operand streams from HBM, coalesced over channels.  Strips are distributed grid-stride; every block
// writes one partial [taps*CS][Cw] slab, reduced by colsum_final_kernel (fp64).
template <int K, int CS>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* __restrict__ thin, const float* __restrict__ wide,
                                                         float* __restrict__ part, int B, int H, int W, int Cw, int sgn,
                                                         int cblk) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NA = K * K * CS;
    extern __shared__ float sh[];
    const int WP = W + 2 * PAD;
    float* tile = sh;                                  // [(TW_ROWS + 2 PAD)][WP][CS]
    float* red = sh + (TW_ROWS + 2 * PAD) * WP * CS;   // [PL][NA][cblk] (PL > 1 only)
    const int pl = threadIdx.x / cblk, tc = threadIdx.x - pl * cblk;
    const int PL = 256 / cblk;
    const int c = blockIdx.y * cblk + tc;
    const int strips_per_img = (H + TW_ROWS - 1) / TW_ROWS;
    const int nstrips = B * strips_per_img;
    float acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = 0.f;
    for (int st = blockIdx.x; st < nstrips; st += gridDim.x) {
        const int b = st / strips_per_img, y0 = (st - b * strips_per_img) * TW_ROWS;
        __syncthreads();
        for (int i = threadIdx.x; i < (TW_ROWS + 2 * PAD) * WP * CS; i += 256) {
            const int s = i % CS;
            const int t = i / CS;
            const int xx = t % WP - PAD, yy = y0 + t / WP - PAD;
            float v = 0.f;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = thin[((size_t)(b * H + yy) * W + xx) * CS + s];
            tile[i] = v;
        }
        __syncthreads();
        const int rows = min(TW_ROWS, H - y0);
        const int npx = rows * W;
        for (int j0 = pl; j0 < npx; j0 += 4 * PL) {
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // 4 independent coalesced loads in flight
                const int j = j0 + u * PL;
                wv[u] = j < npx ? wide[((size_t)(b * H + y0) * W + j) * Cw + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * PL;
                if (j >= npx) break;
                const int ry = j / W, rx = j - ry * W;
#pragma unroll
                for (int dy = 0; dy < K; ++dy)
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) {
                        const float* tp = tile + ((ry + PAD + sgn * (dy - PAD)) * WP + rx + PAD + sgn * (dx - PAD)) * CS;
#pragma unroll
                        for (int s = 0; s < CS; ++s)
                            acc[(dy * K + dx) * CS + s] = fmaf(tp[s], wv[u], acc[(dy * K + dx) * CS + s]);
                    }
            }
        }
    }
    float* dst = part + (size_t)blockIdx.x * NA * Cw;
    if (PL == 1) {
#pragma unroll
        for (int t = 0; t < NA; ++t) dst[(size_t)t * Cw + c] = acc[t];
    } else {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NA; ++t) red[(pl * NA + t) * cblk + tc] = acc[t];
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                float s = 0.f;
                for (int q = 0; q < PL; ++q) s += red[(q * NA + t) * cblk + tc];
                dst[(size_t)t * Cw + c] = s;
            }
        }
    }
}

// 3x3 thin weight gradient on the fp32 matrix pipe: gw[(tap,s)][c] = sum_pix thin[pix + sgn*(tap - 1)][s] * wide[pix][c].
// One v_mfma_f32_32x32x2_f32 per pixel PAIR and 32 wide channels: A = wide[pix][c] straight from global (every element is
// used exactly once, so there is nothing to stage), B = the <= 27 shifted thin values of the pair (a per-lane gather from
// the tiny thin tensor, zero outside the image), columns 27..31 idle.  The VALU kernel above needs one LDS read per FMA
// and ran 10x over the HBM time of its 33 MB stream.
// PADDED (5x5 / 7x7 layers, power-of-two maps): `thin` is the zero-bordered copy above, so a shifted value is ONE add (pixel base +
// per-lane constant) and ONE load -- no bounds, no select (the 5 gathers of a 7x7x3 pixel pair cost ~50 VALU instructions per 10
// MFMAs before, and every instruction issued beside an MFMA costs its pipe 6-9 cycles: DESIGN 4.7); columns >= NA read tap 0 and are
// never stored.
// ONES (round 4): column NA of the (tap, s) axis -- idle in every instance, NA is never a multiple of 32 -- multiplies the constant
// 1, so row NA of the slab is sum_pix wide[pix][c]: with wide = the output gradient of a thin-INPUT convolution that is its bias
// gradient, from the pass that streams the tensor anyway (the separate column-sum pass re-read all of it: 134 MB per layer in c2f).
// (a template parameter: as a run-time flag it cost the 7x7 instance -- 160 accumulator registers, a hand-scheduled load / MFMA
// pipeline -- 417 -> 507 us although that instance never uses it)
template <int K, int CS, int PADDED = 0, int ONES = 0>
__global__ __launch_bounds__(256) void thin_wgrad_mfma_kernel(const float* __restrict__ thin, const float* __restrict__ wide,
                                                              float* __restrict__ part, int B, int H, int W, int Cw, int sgn, int lgH, int lgW) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NA = K * K * CS;
    constexpr int NCT = (NA + 31) / 32;          // 32-column tiles of the (tap, s) axis: 1 (3x3) ... 5 (7x7x3)
    static_assert(NA % 32 != 0, "the ones column needs an idle column");
    constexpr int CTO = NA / 32;                  // the tile that holds column NA
    __shared__ float red[4][2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cbase = blockIdx.y * 64;
    const long long npix = (long long)B * H * W, npairs = (npix + 1) / 2;
    const int j = lane & 31, k = lane >> 5;
    const bool ones_lane = ONES && (CTO * 32 + j == NA);
    constexpr int NR = NA + ONES;                 // slab rows
    int oy[NCT], ox[NCT], sch[NCT];
    bool jok[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int jj = ct * 32 + j;
        jok[ct] = jj < NA;
        const int tap = jok[ct] ? jj / CS : 0;
        sch[ct] = jok[ct] ? jj - tap * CS : 0;
        oy[ct] = sgn * (tap / K - PAD); ox[ct] = sgn * (tap % K - PAD);
    }
    const int Wp = W + 2 * PAD;
    int poff[NCT];                                // PADDED: offset of this lane's (tap, s) from the pixel's own padded position
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) poff[ct] = (oy[ct] * Wp + ox[ct]) * CS + sch[ct];
    tw_f32x16 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[ct][0][r] = 0.f; acc[ct][1][r] = 0.f; }
    // 32-bit pixel indices (the launcher guarantees npix * Cw < 2^31); shifts instead of divisions for power-of-two maps
    const int npx = (int)npix, npr = (int)npairs;
    const int stride = gridDim.x * 4;
    const bool p2 = lgW >= 0 && lgH >= 0;
    constexpr int U = NCT <= 2 ? 2 : 1;           // independent pixel pairs in flight (register budget: NCT accumulators)
    if constexpr (PADDED) {
        // software-pipelined: the 2 + NCT loads of the NEXT pixel pair are issued before the 2 NCT MFMAs of the current one (with
        // two waves per SIMD and ~2 000 cycles of load latency against 640 cycles of MFMAs per pair the plain loop was latency-bound)
        // raw buffer loads: 32-bit byte offsets (no 64-bit address arithmetic), and a pair past the end gets an out-of-range
        // offset -- the hardware returns zeros, no select behind the load
        constexpr int TW_OOB = 0x7FFFFFF0;
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wide, 0, npx * Cw * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)thin, 0, B * (H + 2 * PAD) * Wp * CS * 4, 0x00020000);
        const int wlane = (cbase + j) * 4, wrow = Cw * 4;
        int poffB[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) poffB[ct] = poff[ct] * 4;
        auto fetch = [&](int pp, float& a0, float& a1, float (&bv)[NCT]) {
            const int pix = 2 * pp + k;
            const bool ok = pp < npr && pix < npx;
            const int wo = ok ? pix * wrow + wlane : TW_OOB;
            a0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, wo, 0, 0));
            a1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, wo + 128, 0, 0));
            const int pc = ok ? pix : 0;
            const int x = pc & (W - 1), row = pc >> lgW, b = row >> lgH;              // row = b * H + y
            const int base = (((row + b * (2 * PAD) + PAD) * Wp + x + PAD) * CS) * 4;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bv[ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(trs, base + poffB[ct], 0, 0));
            if constexpr (ONES) { if (ones_lane) bv[CTO] = 1.f; }      // (a pair past the end multiplies zeros of the wide tensor)
        };
        auto mul = [&](float a0, float a1, const float (&bv)[NCT]) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[ct], acc[ct][0], 0, 0, 0);
                acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[ct], acc[ct][1], 0, 0, 0);
            }
        };
        float a0A, a1A, bA[NCT], a0B, a1B, bB[NCT];
        int pp = blockIdx.x * 4 + wave;
        fetch(pp, a0A, a1A, bA);
        for (; pp < npr; pp += 2 * stride) {
            fetch(pp + stride, a0B, a1B, bB);
            __builtin_amdgcn_sched_barrier(0);          // keep the next pair's loads IN FRONT of this pair's MFMAs
            mul(a0A, a1A, bA);
            __builtin_amdgcn_sched_barrier(0);
            fetch(pp + 2 * stride, a0A, a1A, bA);
            __builtin_amdgcn_sched_barrier(0);
            mul(a0B, a1B, bB);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int pp0 = blockIdx.x * 4 + wave; pp0 < npr; pp0 += U * stride) {
        float a0[U], a1[U], bv[U][NCT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = pp0 + u * stride;
            const int pix = 2 * pp + k;
            a0[u] = 0.f; a1[u] = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bv[u][ct] = 0.f;
            if (pp < npr && pix < npx) {
                const float* wp = wide + (size_t)pix * Cw + cbase + j;
                a0[u] = wp[0]; a1[u] = wp[32];
                int x, y, t;
                if (p2) { x = pix & (W - 1); t = pix >> lgW; y = t & (H - 1); }
                else { t = pix / W; x = pix - t * W; y = t % H; }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int yy = y + oy[ct], xx = x + ox[ct];
                    if (jok[ct] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                        bv[u][ct] = thin[(size_t)((t - y + yy) * W + xx) * CS + sch[ct]];
                }
                if constexpr (ONES) { if (ones_lane) bv[u][CTO] = 1.f; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bv[u][ct], acc[ct][0], 0, 0, 0);
                acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], bv[u][ct], acc[ct][1], 0, 0, 0);
            }
    }
    // block reduce (fixed wave order) -> one slab [NA][Cw] per block, like thin_wgrad_kernel
    float* dst = part + (size_t)blockIdx.x * NR * Cw;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (ct) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) { red[wave][0][r][lane] = acc[ct][0][r]; red[wave][1][r][lane] = acc[ct][1][r]; }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * 16 * 64; e += 256) {
            const int tile = e >> 10, r = (e >> 6) & 15, l = e & 63;
            const int col = ct * 32 + (l & 31), row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            if (col < NR)
                dst[(size_t)col * Cw + cbase + tile * 32 + row] = (red[0][tile][r][l] + red[1][tile][r][l]) + (red[2][tile][r][l] + red[3][tile][r][l]);
        }
    }
}

static bool fg_thin_wgrad_padded_on() { return fg_thin_padded_on(); }

// wide_colsum (optional): receives sum_pix wide[pix][c] (beta = 0) from the ones column of the matrix-pipe kernels; *colsum_done
// tells the caller whether it was produced (the VALU fallback kernels do not).  With it the slabs have k*k*Cs + 1 rows: `scratch`
// holds FG_THIN_WGRAD_BLOCKS * (k*k*Cs + 1) * Cw floats for them.
// gradW_ref / unpack_mode (optional, matrix-pipe kernels only; *unpacked = 1 when used): the final sum writes the reference layout
// gradW[O][I][k][k] itself (beta = 0) instead of gw_tsc -- no thin_unpack_grad launch behind it
int fg_launch_thin_wgrad(fg_ctx* ctx, const float* thin, const float* wide, float* gw_tsc, int B, int H, int W, int Cs,
                         int Cw, int k, int shift_thin, float* scratch, float* wide_colsum, int* colsum_done, float* gradW_ref,
                         int unpack_mode, int* unpacked) {
    if (colsum_done) *colsum_done = 0;
    if (unpacked) *unpacked = 0;
    if (Cw % 64) return fg_set_err(ctx, FG_ERR_INVALID, "thin_wgrad: Cw %% 64");
    // (3x3 layers only: the thin-INPUT convolutions of models.lua:385 / models_c2f.lua:123, 244)
    const int ones = (wide_colsum && colsum_done && k == 3 && (ctx->fusion & FG_FUSE_THIN_BIAS)) ? 1 : 0;      // fg_set_fusion
    const int NR = k * k * Cs + ones;
    auto finish = [&](int nblocks) -> int {       // fp64 sum of the slabs; row k*k*Cs (the ones column) goes to wide_colsum
        if (gradW_ref && unpacked) {
            *unpacked = 1;
            if (ones) *colsum_done = 1;
            const int O = unpack_mode == 0 ? Cw : Cs, I = unpack_mode == 0 ? Cs : Cw;
            return fg_launch_colsum_final_thin(ctx, scratch, nblocks, gradW_ref, O, I, k, unpack_mode, ones ? Cw : 0, wide_colsum);
        }
        if (!ones) return fg_launch_colsum_final(ctx, scratch, nblocks, NR * Cw, 0.f, gw_tsc);
        *colsum_done = 1;
        return fg_launch_colsum_final2(ctx, scratch, nblocks, (NR - 1) * Cw, gw_tsc, Cw, wide_colsum);
    };
    // profile label "thin_wgrad<k,Cs>": algorithmic bytes = both tensors read once (the per-block slabs are small)
    char plabel[48] = "";
    if (ctx->prof) snprintf(plabel, sizeof(plabel), "thin_wgrad<%d,%d>", k, Cs);
    FgProfScope prof(ctx, fg_intern(ctx, plabel), 2.0 * B * H * W * (double)k * k * Cs * Cw, 0.0, 4.0 * B * H * W * ((double)Cw + Cs));
    {
        const long long npairs = ((long long)B * H * W + 1) / 2;
        int nb = (int)((npairs + 3) / 4 < TW_BLOCKS ? (npairs + 3) / 4 : TW_BLOCKS);
        if (nb < 1) nb = 1;
        dim3 grid(nb, Cw / 64);
        int lgH = -1, lgW = -1;
        for (int q = 0; q < 15; ++q) { if ((1 << q) == H) lgH = q; if ((1 << q) == W) lgW = q; }
        const bool fits = (long long)B * H * W * Cw < (1LL << 31);
        // 5x5 / 7x7 on power-of-two maps: gather from a zero-bordered copy of the thin tensor kept in the tail of the slab area
        // (the layer's slabs are large: the copy displaces a few of the TW_BLOCKS blocks)
        if (fits && k >= 5 && lgH >= 0 && lgW >= 0 && (Cs == 1 || Cs == 3) && fg_thin_wgrad_padded_on()) {
            const int pad = (k - 1) / 2;
            const long long slab = (long long)NR * Cw, padf = (long long)B * (H + 2 * pad) * (W + 2 * pad) * Cs;
            const long long take = (padf + slab - 1) / slab;
            if (take <= TW_BLOCKS / 4 && padf < (1LL << 28) && (long long)B * H * W * Cw * 4 < 0x7FFFFFF0LL) {
                const int nbp = nb < TW_BLOCKS - (int)take ? nb : TW_BLOCKS - (int)take;
                float* padded = scratch + (long long)(TW_BLOCKS - take) * slab;
{ const int rcp = fg_launch_thin_pad(ctx, thin, padded, B, H, W, Cs, pad); if (rcp) return rcp; }
                dim3 gridp(nbp, Cw / 64);
#define TWP(KK, CC)                                                                                                  \
                if (k == KK && Cs == CC) {                                                                           \
                    hipLaunchKernelGGL((thin_wgrad_mfma_kernel<KK, CC, 1>), gridp, dim3(256), 0, ctx->stream, padded, wide, scratch, B, \
                                       H, W, Cw, shift_thin, lgH, lgW);                                              \
                    FG_CHECK_LAUNCH(ctx);                                                                            \
                    return finish(nbp);                                                                              \
                }
                TWP(5, 1) TWP(5, 3) TWP(7, 1) TWP(7, 3)
#undef TWP
            }
        }
#define TWM(KK, CC)                                                                                                  \
        if (fits && k == KK && Cs == CC) {                                                                           \
            if (KK == 3 && ones)                                                                                     \
                hipLaunchKernelGGL((thin_wgrad_mfma_kernel<KK, CC, 0, (KK == 3 ? 1 : 0)>), grid, dim3(256), 0, ctx->stream, thin, wide, \
                                   scratch, B, H, W, Cw, shift_thin, lgH, lgW);                                      \
            else                                                                                                     \
                hipLaunchKernelGGL((thin_wgrad_mfma_kernel<KK, CC>), grid, dim3(256), 0, ctx->stream, thin, wide, scratch, B, H, \
                                   W, Cw, shift_thin, lgH, lgW);                                                     \
            FG_CHECK_LAUNCH(ctx);                                                                                    \
            return finish(nb);                                                                                       \
        }
        TWM(3, 1) TWM(3, 3) TWM(3, 4) TWM(5, 1) TWM(5, 3) TWM(7, 1) TWM(7, 3)
#undef TWM
    }
    const int cblk = Cw >= 256 ? 256 : Cw;
    if (Cw % cblk || 256 % cblk) return fg_set_err(ctx, FG_ERR_INVALID, "thin_wgrad: Cw=%d unsupported", Cw);
    const int nstrips = B * ((H + TW_ROWS - 1) / TW_ROWS);
    int nb = nstrips < TW_BLOCKS ? nstrips : TW_BLOCKS;
    if (nb < 1) nb = 1;
    const int PL = 256 / cblk;
    const int NA = k * k * Cs;
    const int pad = (k - 1) / 2;
    dim3 grid(nb, Cw / cblk);
    const size_t lds = ((size_t)(TW_ROWS + 2 * pad) * (W + 2 * pad) * Cs + (PL > 1 ? (size_t)PL * NA * cblk : 0)) * sizeof(float);
#define TWG(KK, CC)                                                                                                  \
    if (k == KK && Cs == CC) {                                                                                       \
        hipLaunchKernelGGL((thin_wgrad_kernel<KK, CC>), grid, dim3(256), lds, ctx->stream, thin, wide, scratch, B, H, W, \
                           Cw, shift_thin, cblk);                                                                    \
        FG_CHECK_LAUNCH(ctx);                                                                                        \
        return fg_launch_colsum_final(ctx, scratch, nb, NA * Cw, 0.f, gw_tsc);                                       \
    }
    TWG(3, 1) TWG(3, 3) TWG(3, 4) TWG(5, 3) TWG(7, 3) TWG(7, 1)
#undef TWG
    return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "thin_wgrad: k=%d Cs=%d not built", k, Cs);
}

// ---------------------------------------------------------------------------------
// reference [O][I][k][k] <-> thin layouts
// ---------------------------------------------------------------------------------
__global__ void thin_pack_kernel(const float* __restrict__ Wr, float* __restrict__ Wp, int O, int I, int kk, int mode) {
    const int total = O * I * kk;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // idx enumerates packed [tap][s][c]
    const int Cs = mode == 0 ? I : O, Cw = mode == 0 ? O : I;
    const int c = idx % Cw;
    const int t = idx / Cw;
    const int s = t % Cs;
    const int tap = t / Cs;
    const int o = mode == 0 ? c : s, i = mode == 0 ? s : c;
    Wp[idx] = Wr[((size_t)o * I + i) * kk + tap];
}
int fg_launch_thin_pack(fg_ctx* ctx, const float* W, float* Wp, int O, int I, int k, int mode) {
    const int total = O * I * k * k;
    hipLaunchKernelGGL(thin_pack_kernel, dim3(fg_cdiv(total, 256)), dim3(256), 0, ctx->stream, W, Wp, O, I, k * k, mode);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
__global__ void thin_unpack_grad_kernel(const float* __restrict__ gw, float* __restrict__ gradW, int O, int I, int kk,
                                        int mode, float beta) {
    const int total = O * I * kk;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Cs = mode == 0 ? I : O, Cw = mode == 0 ? O : I;
    const int c = idx % Cw;
    const int t = idx / Cw;
    const int s = t % Cs;
    const int tap = t / Cs;
    const int o = mode == 0 ? c : s, i = mode == 0 ? s : c;
    const size_t r = ((size_t)o * I + i) * kk + tap;
    gradW[r] = (beta == 0.f ? 0.f : beta * gradW[r]) + gw[idx];
}
int fg_launch_thin_unpack_grad(fg_ctx* ctx, const float* gw, float* gradW, int O, int I, int k, int mode, float beta) {
    const int total = O * I * k * k;
    hipLaunchKernelGGL(thin_unpack_grad_kernel, dim3(fg_cdiv(total, 256)), dim3(256), 0, ctx->stream, gw, gradW, O, I,
                       k * k, mode, beta);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
