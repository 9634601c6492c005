// fp32-MFMA implicit-GEMM kernels for gfx950 (MI355X):
//   igemm_kernel  : gather-A contraction  (conv fwd / dgrad, nearest-x2-folded conv fwd / dgrad, Linear fwd / dgrad)
//   wgrad_kernel  : pixel-reduction contraction (conv / Linear weight gradients), split over pixels
//   pack / finish : reference [O][I][kH][kW] <-> packed tile layouts (incl. the nearest-x2 tap folding)
//
// Replaces, for the hot path, what the reference dispatches to THNN SpatialConvolutionMM / cuDNN v3 /
// cuBLAS sgemm (models.lua:59-73, 385-412).  Exact fp32: v_mfma_f32_32x32x2_f32 is
// bit-for-bit a k-ordered fmaf chain.
//
// Tiling: 256 threads = 4 waves (2x2), block tile BMxBN, BK = 32, register-staged double-buffered LDS,
// rows padded to 36 floats so the ds_read_b128 fragment reads are conflict-free (16 distinct 16-B slots
// per 16-lane group).  Lane l of an MFMA holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; one ds_read_b128 per
// 32-row fragment feeds 4 MFMAs (k = kk+j for lanes<32, kk+4+j for lanes>=32 -- same permutation on A and B).
#include "fg_internal.h"
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void fg_decode_m(int m, int lgH, int lgW, int Hm, int Wm, int& n, int& y, int& x) {
    if (lgW >= 0) {
        x = m & (Wm - 1);
        y = (m >> lgW) & (Hm - 1);
        n = m >> (lgW + lgH);
    } else {
        x = m % Wm;
        int t = m / Wm;
        y = t % Hm;
        n = t / Hm;
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define FG_OOB 0x7FFFFFF0   // voffset marker: beyond any buffer (< 2 GiB) -> the buffer load returns zeros
// EPI == 2 epilogue: row offset (in floats) of a masked row -- (FG_ROW_MASKED + col) * 4 lands in [2 GiB, 4 GiB), out of range
// of every buffer resource, without a per-element compare (128 lane masks per wave spilled the scalar registers)
#define FG_ROW_MASKED 0x20000000

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP's float4 struct array went to scratch)
__device__ __forceinline__ f32x4 fg_buffer_load4(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// Epilogue store of one 32x32 accumulator tile: branch-free raw-buffer stores (masked rows / padded columns get an
// out-of-range voffset, which the hardware drops).  The bias value is made available BEFORE the store sequence: with a
// conditional bias load the compiler put `s_waitcnt vmcnt(0)` in front of every store, which on gfx9 also waits for
// all earlier STORES -- the 64..128 stores of a wave ran one memory round trip at a time.
__device__ __forceinline__ void fg_store_acc_tile(__amdgpu_buffer_rsrc_t orsrc, const int* rowoff, int row0, int col,
                                                  bool colok, float bv, const f32x16& acc, int lane) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int4 ro = *(const int4*)(rowoff + row0 + 8 * r4 + 4 * (lane >> 5));
        const int offs[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int voff = (offs[q] >= 0 && colok) ? (offs[q] + col) * 4 : FG_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r4 * 4 + q] + bv), orsrc, voff, 0, 0);
        }
    }
}

// Epilogue of the fp32 kernels for one wave's MI x NI accumulator tiles (rows row_base + 32 mi, columns col_base + 32 ni):
// plain store (+ bias), store + the PReLU behind the layer (act_y), or the backward of the PReLU in front of the layer
// (act_x: 16 loads of x per tile in flight, then 16 stores; the wave's share of the slope gradient goes to
// act_part[part_idx]).  EPI selects the variant at compile time (0 plain, 1 act_y, 2 act_x): the kernels without a fused
// PReLU keep their register allocation.
#ifdef FG_MEASURE
#define FG_STORE_RANGE(a) ((a).dbg_nostore ? 0 : FG_OOB)     // FG_DEBUG_NOSTORE=1: the stores go to a zero-sized buffer (results are WRONG)
#else
#define FG_STORE_RANGE(a) FG_OOB
#endif
template <int MI, int NI, int EPI>
__device__ __forceinline__ void fg_epilogue(const IgemmArgs& a, float* outp, const int* rowoff, int row_base, int col_base,
                                            const f32x16 (&acc)[MI][NI], int lane, int part_idx) {
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, FG_STORE_RANGE(a), 0x00020000);
    if constexpr (EPI == 2) {
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.act_x, 0, FG_OOB, 0x00020000);
        const float sl = a.act_slope[0];
        float s = 0.f;
        // rows outermost: the 16 row offsets of a 32-row tile are read once and serve both column tiles; one tile's 16
        // loads of x are in flight at a time (the accumulators already hold half of the register file)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int ro[16];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int4 t = *(const int4*)(rowoff + row_base + mi * 32 + 8 * r4 + 4 * (lane >> 5));
                ro[r4 * 4 + 0] = t.x; ro[r4 * 4 + 1] = t.y; ro[r4 * 4 + 2] = t.z; ro[r4 * 4 + 3] = t.w;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = col_base + ni * 32 + (lane & 31);
                const bool colok = col < a.N;
                float xv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int vo = colok ? (int)(((unsigned)ro[i] + (unsigned)col) * 4u) : FG_OOB;   // unsigned: a masked row wraps past 2 GiB
                    xv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, vo, 0, 0));
                }
                float t4[4] = {0.f, 0.f, 0.f, 0.f};       // four short chains per tile instead of one 128-long serial chain
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int vo = colok ? (int)(((unsigned)ro[i] + (unsigned)col) * 4u) : FG_OOB;   // unsigned: a masked row wraps past 2 GiB
                    const float g = acc[mi][ni][i];
                    const bool pos = xv[i] > 0.f;
                    t4[i & 3] = fmaf(pos ? 0.f : xv[i], g, t4[i & 3]);   // masked elements read x = 0: they add 0 * g
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pos ? g : sl * g), orsrc, vo, 0, 0);
                }
                s += (t4[0] + t4[1]) + (t4[2] + t4[3]);
                // pin the sum here: LLVM otherwise sinks the whole reduction into the `if (act_part)` block below and keeps
                // all 128 x values and lane masks alive until then (spills)
                asm volatile("" : "+v"(s));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.act_part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) a.act_part[part_idx] = s;
        }
        return;
    }
    const bool add_bias = (a.bias != nullptr) && (a.splits == 1);
    if constexpr (EPI == 1) {
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.act_y, 0, FG_OOB, 0x00020000);
        const float sl = a.act_slope[0];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = col_base + ni * 32 + (lane & 31);
            const bool colok = col < a.N;
            float bv = add_bias ? a.bias[colok ? col : 0] : 0.f;
            asm volatile("" : "+v"(bv));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int4 ro = *(const int4*)(rowoff + row_base + mi * 32 + 8 * r4 + 4 * (lane >> 5));
                    const int offs[4] = {ro.x, ro.y, ro.z, ro.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int voff = (offs[q] >= 0 && colok) ? (offs[q] + col) * 4 : FG_OOB;
                        const float v = acc[mi][ni][r4 * 4 + q] + bv;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, voff, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v > 0.f ? v : sl * v), yrsrc, voff, 0, 0);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = col_base + ni * 32 + (lane & 31);
        const bool colok = col < a.N;
        float bv = add_bias ? a.bias[colok ? col : 0] : 0.f;
        asm volatile("" : "+v"(bv));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            fg_store_acc_tile(orsrc, rowoff, row_base + mi * 32, col, colok, bv, acc[mi][ni], lane);
    }
}

// BatchNorm statistics from the accumulators (IgemmArgs::stats_part): lane l of a 32x32 tile holds 16 rows of column l & 31;
// a wave's MI tiles x 16 rows x the two lane halves are summed in registers, so every wave leaves ONE partial row.
template <int MI, int NI>
__device__ __forceinline__ void fg_store_stats(const IgemmArgs& a, const f32x16 (&acc)[MI][NI], int row, int col0, int lane) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[mi][ni][r]; s1 += v; s2 = fmaf(v, v, s2); }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const int col = col0 + ni * 32 + (lane & 31);
        if (lane < 32 && col < a.N) {
            a.stats_part[(size_t)row * a.N + col] = s1;
            a.stats_part[((size_t)a.stats_rows + row) * a.N + col] = s2;
        }
    }
}

template <int BM, int BN, int BK, int EPI>
__device__ __forceinline__ void igemm_body(const IgemmArgs& a) {
    constexpr int LDK = BK + 4;                 // padded row: conflict-free ds_read_b128 fragment reads
    constexpr int LPR = BK / 4;                 // lanes (float4) per tile row
    constexpr int RPP = 256 / LPR;              // rows per load pass
    constexpr int RA = BM / RPP, RB = BN / RPP; // float4 loads per thread per K-step
    constexpr int MI = BM / 64, NI = BN / 64;  // 32x32 MFMA tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;
    int* rowoff = (int*)(smem + 2 * (BM + BN) * LDK);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware block -> (tile, parity) map.  The dispatcher places block b on XCD b % 8 (speed only, never relied on
    // for correctness): give every XCD a contiguous range of M-tiles and make the N-tile / output parity the fastest
    // index, so the blocks that re-read one input neighbourhood (all parities, all N-tiles, 3x3 halo of adjacent
    // M-tiles) share one L2 instead of pulling the activation through all eight.
    const int ntn = a.Npad / BN;
    const int np = a.P;
    const int per_m = ntn * np;                       // blocks sharing one M-tile
    const int nmt = (a.M + BM - 1) / BM;
    int lin = blockIdx.x;
    if ((nmt & 7) == 0) {
        const int xcd = lin & 7, loc = lin >> 3;
        const int mt_per_xcd = nmt >> 3;
        lin = (xcd * mt_per_xcd + loc / per_m) * per_m + loc % per_m;
    }
    const int tile_m = lin / per_m;
    const int rem = lin - tile_m * per_m;
    const int tile_n = rem / np, p = rem - tile_n * np;
    const int split = blockIdx.y;

    if (tid < BM) {
        int m = tile_m * BM + tid, off = EPI == 2 ? FG_ROW_MASKED : -1;
        if (m < a.M) {
            int n, y, x;
            fg_decode_m(m, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
            off = ((n * a.Ho + y * a.osy + a.ooy[p]) * a.Wo + x * a.osx + a.oox[p]) * a.N;
        }
        rowoff[tid] = off;
    }

    // A operand through a raw buffer resource: out-of-image taps / ragged rows / K tail read as hardware zeros
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)a.a_bytes, 0x00020000);
    const int lrow = tid / LPR, lk = (tid % LPR) * 4;
    int ry[RA], rx[RA], rn[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = tile_m * BM + lrow + RPP * i;
        int n, y, x;
        fg_decode_m(m < a.M ? m : 0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
        rn[i] = n * a.Ha * a.Wa;
        ry[i] = m < a.M ? y * a.asy : -(1 << 20);   // ragged rows: never in range
        rx[i] = x * a.asx;
    }
    const int kc = a.Kpad / BK;
    const int kt_all = a.G * kc;                       // K-steps of the whole contraction
    const int kt_per = (kt_all + a.splits - 1) / a.splits;
    const int kt0 = split * kt_per;                    // this split's K-step range
    const int KT = max(0, min(kt_all, kt0 + kt_per) - kt0);
    const bool ktail = a.Ca != a.Kpad;

    int g = kt0 / kc;
    int col0 = (kt0 - g * kc) * BK;
    int voff[RA];
#define FG_SET_GROUP()                                                                                   \
    {                                                                                                    \
        const int go = a.goff[p][g < a.G ? g : 0];                                                       \
        const int oy = (int)(short)(go & 0xffff), ox = go >> 16;                                         \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                                 \
            const int ya = ry[i] + oy, xa = rx[i] + ox;                                                  \
            const bool ok = (unsigned)ya < (unsigned)a.Ha && (unsigned)xa < (unsigned)a.Wa;              \
            voff[i] = ok ? ((rn[i] + ya * a.Wa + xa) * a.Ca + lk) * 4 : FG_OOB;                          \
        }                                                                                                \
    }
    FG_SET_GROUP();
    const float* bptr = a.Bp + ((size_t)(p * a.G + g) * a.Npad + tile_n * BN + lrow) * a.Kpad + col0 + lk;
    const size_t brow = (size_t)RPP * a.Kpad;
    const size_t bjump = (size_t)(a.Npad - 1) * a.Kpad;

    f32x4 ra[RA], rb[RB];
#define FG_LOAD_TILE()                                                                                   \
    {                                                                                                    \
        const int cb = col0 * 4;                                                                         \
        const bool kin = !ktail || (col0 + lk < a.Ca);                                                   \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                   \
            ra[i] = fg_buffer_load4(arsrc, kin ? voff[i] + cb : FG_OOB);                                 \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) rb[i] = *(const f32x4*)(bptr + i * brow);         \
        col0 += BK; bptr += BK;                                                                          \
        if (col0 == a.Kpad) { col0 = 0; ++g; bptr += bjump; FG_SET_GROUP(); }                            \
    }
#define FG_STORE_TILE(buf)                                                                               \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                   \
            *(f32x4*)(As + (buf) * BM * LDK + (lrow + RPP * i) * LDK + lk) = ra[i];                       \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                   \
            *(f32x4*)(Bs + (buf) * BN * LDK + (lrow + RPP * i) * LDK + lk) = rb[i];                       \
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int a_lds = (wm * (BM / 2) + (lane & 31)) * LDK + (lane >> 5) * 4;
    const int b_lds = (wn * (BN / 2) + (lane & 31)) * LDK + (lane >> 5) * 4;

    if (KT > 0) {
        FG_LOAD_TILE();
        FG_STORE_TILE(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) FG_LOAD_TILE();
        {
            const float* Ab = As + cur * BM * LDK + a_lds;
            const float* Bb = Bs + cur * BN * LDK + b_lds;
            f32x4 af[2][MI], bf[2][NI];   // fragment double buffer: chunk kk+8 is read while chunk kk multiplies
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[0][mi] = *(const f32x4*)(Ab + mi * 32 * LDK);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *(const f32x4*)(Bb + ni * 32 * LDK);
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                if (c < BK / 8 - 1) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) af[(c + 1) & 1][mi] = *(const f32x4*)(Ab + mi * 32 * LDK + (c + 1) * 8);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bf[(c + 1) & 1][ni] = *(const f32x4*)(Bb + ni * 32 * LDK + (c + 1) * 8);
                }
                // (round 4) keep the NEXT chunk's fragment reads in front of THIS chunk's MFMAs: without the fence the compiler folds
                // the two-deep buffer into read -> lgkmcnt(0) -> MFMAs on one register set, and a wave with a single accumulator
                // (the 64 x 64 tile: one wave per SIMD in a one-round launch) leaves the matrix pipe idle for every LDS round trip
                if constexpr (MI * NI == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][mi][j], bf[c & 1][ni][j],
                                                                               acc[mi][ni], 0, 0, 0);
                if constexpr (MI * NI == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) FG_STORE_TILE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef FG_SET_GROUP
#undef FG_LOAD_TILE
#undef FG_STORE_TILE

    if (a.stats_part) fg_store_stats<MI, NI>(a, acc, (tile_m * np + p) * 2 + wm, tile_n * BN + wn * (BN / 2), lane);
    fg_epilogue<MI, NI, EPI>(a, a.Out + (size_t)split * a.split_stride, rowoff, wm * (BM / 2), tile_n * BN + wn * (BN / 2), acc,
                             lane, blockIdx.x * 4 + wid);
}
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) { igemm_body<BM, BN, BK, 0>(a); }
// the same kernel with an nn.PReLU folded into the epilogue: EPI 1 = behind the layer (forward), 2 = in front (data gradient)
template <int BM, int BN, int BK, int EPI>
__global__ __launch_bounds__(256, 2) void igemm_act_kernel(const IgemmArgs a) { igemm_body<BM, BN, BK, EPI>(a); }

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for the large layers: ONE 512-thread block per CU; waves 0-3 only multiply (one per SIMD,
// 128x64 outputs each = 8 accumulator tiles, fragments double-buffered across the K-step barrier), waves 4-7 only move
// data (gather address math, raw-buffer loads, ds_write) two K-steps ahead into a 4-stage LDS ring.  The MFMA waves
// issue nothing but ds_read_b128 + v_mfma, so the matrix pipe never waits for a staging phase.
// Block tile 256 x 128, BK = 16, rows padded to 20 floats (conflict-free b128 fragment reads).
// ---------------------------------------------------------------------------------------------------------------
#define WS_BM 256
#define WS_BN 128
#define WS_BK 16
#define WS_LDK 20
#define WS_NS 4
#define WS_STAGE ((WS_BM + WS_BN) * WS_LDK)
// BN = 128: MFMA waves 2x2, each 128x64; BN = 64 (layers with 64 output channels): waves 4x1, each 64x64
// BM / NS are template parameters for the BN = 64 variants (NS = 3: see igemm_ws64x3_kernel).
template <int BN, int EPI, int BM = WS_BM, int NS = WS_NS, int TRACE = 0>
__device__ __forceinline__ void igemm_ws_body(const IgemmArgs& a) {
    constexpr int WNW = BN / 64, MI = BM / ((4 / WNW) * 32), NI = 2, NB = BN / 64, AR = BM / 64;
    constexpr int STAGE = (BM + BN) * WS_LDK;
    static_assert(NS == 4 || NS == 3, "ring of 3 or 4 stages");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* rowoff = (int*)(smem + NS * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = a.Npad / BN;
    const int np = a.P;
    const int per_m = ntn * np;
    const int nmt = (a.M + BM - 1) / BM;
    int lin = blockIdx.x;
    if ((nmt & 7) == 0) {
        const int xcd = lin & 7, loc = lin >> 3;
        lin = (xcd * (nmt >> 3) + loc / per_m) * per_m + loc % per_m;
    }
    const int tile_m = lin / per_m;
    const int rem = lin - tile_m * per_m;
    const int tile_n = rem / np, p = rem - tile_n * np;
    const int split = blockIdx.y;

    if (tid < BM) {
        int m = tile_m * BM + tid, off = EPI == 2 ? FG_ROW_MASKED : -1;
        if (m < a.M) {
            int n, y, x;
            fg_decode_m(m, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
            off = ((n * a.Ho + y * a.osy + a.ooy[p]) * a.Wo + x * a.osx + a.oox[p]) * a.N;
        }
        rowoff[tid] = off;
    }
    const int kc = a.Kpad / WS_BK;
    const int kt_all = a.G * kc;
    const int kt_per = (kt_all + a.splits - 1) / a.splits;
    const int kt0 = split * kt_per;
    const int KT = max(0, min(kt_all, kt0 + kt_per) - kt0);

    if (wid >= 4) {
        // ------------------------------------------------------------------ loader waves (256 threads)
        const int lt = tid - 256;
        const int lrow = lt >> 2, lk = (lt & 3) * 4;           // 4 lanes per 16-float row, 64 rows per pass
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)a.a_bytes, 0x00020000);
        int ry[AR], rx[AR], rn[AR];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = tile_m * BM + lrow + 64 * i;
            int n, y, x;
            fg_decode_m(m < a.M ? m : 0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
            rn[i] = n * a.Ha * a.Wa;
            ry[i] = m < a.M ? y * a.asy : -(1 << 20);
            rx[i] = x * a.asx;
        }
        const bool ktail = a.Ca != a.Kpad;
        int g = kt0 / kc;
        int col0 = (kt0 - g * kc) * WS_BK;
        int voff[AR];
#define WS_SET_GROUP()                                                                                   \
        {                                                                                                \
            const int go = a.goff[p][g < a.G ? g : 0];                                                   \
            const int oy = (int)(short)(go & 0xffff), ox = go >> 16;                                     \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                             \
                const int ya = ry[i] + oy, xa = rx[i] + ox;                                              \
                const bool ok = (unsigned)ya < (unsigned)a.Ha && (unsigned)xa < (unsigned)a.Wa;          \
                voff[i] = ok ? ((rn[i] + ya * a.Wa + xa) * a.Ca + lk) * 4 : FG_OOB;                      \
            }                                                                                            \
        }
        WS_SET_GROUP();
        // packed weights [P][G][Npad][Kpad]: < 2 GiB (the 33.5 M-weight Linear packs to 134 MB), raw-buffer addressing
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.Bp, 0, FG_OOB, 0x00020000);
        int boff = (int)((((size_t)(p * a.G + g) * a.Npad + tile_n * BN + lrow) * a.Kpad + col0 + lk) * 4);
        const int browB = 64 * a.Kpad * 4;
        const int bjumpB = (a.Npad - 1) * a.Kpad * 4;
        f32x4 xa[AR], xb[NB], ya[AR], yb[NB];     // two tiles in flight: a load has two full K-steps to land
        // (instruction count matters here -- every instruction a loader wave issues costs the SIMD's matrix pipe 6-9 cycles, DESIGN
        // 4.7: the K tail is a wave-uniform branch, not a select per load; the packed weights come through a raw buffer with a
        // 32-bit per-lane offset, not a 64-bit pointer; one wait for all staged loads in front of the LDS stores)
#define WS_LOAD(ra, rb)                                                                                  \
        {                                                                                                \
            const int cb = col0 * 4;                                                                     \
            if (ktail) {                                                                                 \
                const bool kin = col0 + lk < a.Ca;                                                       \
                _Pragma("unroll") for (int i = 0; i < AR; ++i)                                           \
                    ra[i] = fg_buffer_load4(arsrc, kin ? voff[i] + cb : FG_OOB);                         \
            } else {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < AR; ++i) ra[i] = fg_buffer_load4(arsrc, voff[i] + cb); \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = fg_buffer_load4(brsrc, boff + i * browB); \
            col0 += WS_BK; boff += WS_BK * 4;                                                            \
            if (col0 == a.Kpad) { col0 = 0; ++g; boff += bjumpB; WS_SET_GROUP(); }                      \
        }
#define WS_STORE(st, ra, rb)                                                                             \
        {                                                                                                \
            __builtin_amdgcn_s_waitcnt(0x0f70);       /* vmcnt(0): these loads were issued two K-steps ago */ \
            float* As = smem + (st) * STAGE;                                                          \
            float* Bs = As + BM * WS_LDK;                                                                \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                               \
                *(f32x4*)(As + (lrow + 64 * i) * WS_LDK + lk) = ra[i];                                   \
            _Pragma("unroll") for (int i = 0; i < NB; ++i)                                               \
                *(f32x4*)(Bs + (lrow + 64 * i) * WS_LDK + lk) = rb[i];                                   \
        }
        // prologue: tiles 0 and 1 resident, tiles 2 and 3 in flight
        if (KT > 0) { WS_LOAD(xa, xb); WS_STORE(0, xa, xb); }
        if (KT > 1) { WS_LOAD(xa, xb); WS_STORE(1, xa, xb); }
        if (KT > 2) { WS_LOAD(xa, xb); }
        if (KT > 3) { WS_LOAD(ya, yb); }
        __syncthreads();
        int sw = 2;                                  // ring stage tile kt + 2 goes to (NS = 3: a counter; NS = 4: a mask)
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 2 < KT) {
                WS_STORE(NS == 4 ? ((kt + 2) & 3) : sw, xa, xb);
                if (kt + 4 < KT) { WS_LOAD(xa, xb); }
            }
            if (NS == 3) sw = sw == 2 ? 0 : sw + 1;
            __syncthreads();
            if (kt + 1 < KT) {
                if (kt + 3 < KT) {
                    WS_STORE(NS == 4 ? ((kt + 3) & 3) : sw, ya, yb);
                    if (kt + 5 < KT) { WS_LOAD(ya, yb); }
                }
                if (NS == 3) sw = sw == 2 ? 0 : sw + 1;
                __syncthreads();
            }
        }
#undef WS_SET_GROUP
#undef WS_LOAD
#undef WS_STORE
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves (one per SIMD)
    const int wm = wid / WNW, wn = wid - wm * WNW;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int a_lds = (wm * MI * 32 + (lane & 31)) * WS_LDK + (lane >> 5) * 4;
    const int b_lds = BM * WS_LDK + (wn * 64 + (lane & 31)) * WS_LDK + (lane >> 5) * 4;
    f32x4 af[2][MI], bf[2][NI];     // [chunk parity][tile]
    // TRACE (measurement kernel only): s_memtime of wave 0 at kernel entry, after the ring's first barrier, after every K-step
    // barrier and after the epilogue -> dbg_trace[block][0 .. KT + 3]
    unsigned long long* trc = nullptr;
    if (TRACE) {
        if (wid == 0 && a.dbg_trace) trc = a.dbg_trace + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 128;
        if (trc && lane == 0) trc[0] = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();              // tiles 0 and 1 are in the ring
    if (TRACE && trc && lane == 0) trc[1] = __builtin_amdgcn_s_memtime();
    if (KT > 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[0][mi] = *(const f32x4*)(smem + a_lds + mi * 32 * WS_LDK);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *(const f32x4*)(smem + b_lds + ni * 32 * WS_LDK);
    }
    int sr = 0;                                      // NS = 3: ring stage of tile kt
    for (int kt = 0; kt < KT; ++kt) {
        const int sr1 = sr == 2 ? 0 : sr + 1;
        const float* St = smem + (NS == 4 ? (kt & 3) : sr) * STAGE;
        const float* Sn = smem + (NS == 4 ? ((kt + 1) & 3) : sr1) * STAGE;
        if (NS == 3) sr = sr1;
        // chunk 1 fragments of this tile
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[1][mi] = *(const f32x4*)(St + a_lds + mi * 32 * WS_LDK + 8);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[1][ni] = *(const f32x4*)(St + b_lds + ni * 32 * WS_LDK + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][mi][j], bf[0][ni][j], acc[mi][ni], 0, 0, 0);
        // chunk 0 fragments of the NEXT tile (already in the ring: it was stored one K-step ago)
        if (kt + 1 < KT) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[0][mi] = *(const f32x4*)(Sn + a_lds + mi * 32 * WS_LDK);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *(const f32x4*)(Sn + b_lds + ni * 32 * WS_LDK);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][mi][j], bf[1][ni][j], acc[mi][ni], 0, 0, 0);
        __syncthreads();
        if (TRACE && trc && lane == 0) trc[2 + (kt < 119 ? kt : 119)] = __builtin_amdgcn_s_memtime();   // KT > 120: slot 121 ends up with the LAST step
    }

    if (a.stats_part) fg_store_stats<MI, NI>(a, acc, (tile_m * np + p) * (4 / WNW) + wm, tile_n * BN + wn * 64, lane);
    fg_epilogue<MI, NI, EPI>(a, a.Out + (size_t)split * a.split_stride, rowoff, wm * MI * 32, tile_n * BN + wn * 64, acc, lane,
                             blockIdx.x * 4 + wid);
    if (TRACE && trc && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);        // the stores have left the wave
        trc[(KT < 120 ? KT : 120) + 2] = __builtin_amdgcn_s_memtime();
        trc[127] = (unsigned long long)__builtin_amdgcn_s_getreg(((3 - 1) << 11) | (0 << 6) | 20) | ((unsigned long long)KT << 32);   // XCC_ID, KT
        trc[126] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);                                    // HW_ID (cu / sh / se)
    }
}
template <int BN>
__global__ __launch_bounds__(512, 2) void igemm_ws_kernel(const IgemmArgs a) { igemm_ws_body<BN, 0>(a); }
template <int BN, int EPI>
__global__ __launch_bounds__(512, 2) void igemm_ws_act_kernel(const IgemmArgs a) { igemm_ws_body<BN, EPI>(a); }
// BN = 64 on a 3-stage ring: 77 KB of LDS and <= 128 VGPRs, so TWO blocks share a CU.  The layers with 64 output channels have
// short K loops (3x3 x 64 channels = 36 sixteen-channel steps per tile): with one block per CU the matrix pipe idles through every
// tile's prologue and epilogue; with two, one block's epilogue overlaps the other's K loop.  (A 512 x 64 tile that halves the
// loader instructions per MFMA was measured first and changed nothing: 118.2 vs 118.5 TFLOP/s -- these layers are not issue-bound.)
template <int EPI>
__global__ __launch_bounds__(512, 4) void igemm_ws64x3_kernel(const IgemmArgs a) { igemm_ws_body<64, EPI, WS_BM, 3>(a); }
#ifdef FG_MEASURE       // measurement build only (libfacegen_hip_measure.so): trace kernels, their calibration and launcher
template <int BN>
__global__ __launch_bounds__(512, 2) void igemm_ws_trace_kernel(const IgemmArgs a) { igemm_ws_body<BN, 0, WS_BM, (BN == 64 ? 3 : WS_NS), 1>(a); }
// FG_WS_TRACE=1 (measurement only): EPI-0 launches run the trace kernel; the per-block s_memtime rows are copied back after the
// launch (synchronously) and written to the file FG_WS_TRACE_FILE
// calibration for the trace: 4 waves per CU x 2 blocks issue nothing but MFMAs; s_memtime cycles per MFMA and the wall time of the
// launch give what the counter counts and the clock the chip grants a pure matrix loop
template <int MODE>
__global__ __launch_bounds__(256) void trace_calib_kernel(unsigned long long* out, int n, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = seed;
    float x = seed + threadIdx.x, y = seed - threadIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    } else if (MODE == 1) {
        for (int k = 0; k < n; ++k) {                       // 16 x (s_nop 15) = 256 cycles of sequencer time per trip
            asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
                         "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15" ::: "memory");
        }
    } else {
        for (int k = 0; k < n; ++k) {                       // 16 dependent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\n"
                         "v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\n"
                         "v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\n"
                         "v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1"
                         : "+v"(x) : "v"(y));
        }
    }
    float r = x;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (r1 - r0) + (unsigned long long)(r == 12345.f); }
}
static void fg_trace_calibrate(fg_ctx* ctx, FILE* f) {
    unsigned long long* dev = nullptr;
    if (hipMalloc((void**)&dev, 512 * 16) != hipSuccess) return;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* what[3] = {"80000 MFMA 32x32x2 f32 per wave", "320000 x (s_nop 15) per wave", "320000 dependent v_fma_f32 per wave"};
    for (int mode = 0; mode < 3; ++mode) for (int nblk = 256; nblk <= 512; nblk *= 2) for (int it = 0; it < 2; ++it) {
        const int n = 20000;
        (void)hipEventRecord(e0, ctx->stream);
        if (mode == 0) hipLaunchKernelGGL(trace_calib_kernel<0>, dim3(nblk), dim3(256), 0, ctx->stream, dev, n, 1.0f);
        if (mode == 1) hipLaunchKernelGGL(trace_calib_kernel<1>, dim3(nblk), dim3(256), 0, ctx->stream, dev, n, 1.0f);
        if (mode == 2) hipLaunchKernelGGL(trace_calib_kernel<2>, dim3(nblk), dim3(256), 0, ctx->stream, dev, n, 1.0f);
        (void)hipEventRecord(e1, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nblk * 2);
        (void)hipMemcpy(h.data(), dev, nblk * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0; for (int b = 0; b < nblk; ++b) { cyc += (double)h[b * 2]; rt += (double)h[b * 2 + 1]; }
        cyc /= nblk; rt /= nblk;
        const double per = mode == 0 ? cyc / (n * 4.0) : cyc / (n * 16.0);
        fprintf(f, "# calib %s, %d wave(s) per SIMD: %.0f s_memtime ticks per block = %.2f per instruction of one wave; s_memrealtime %.0f ticks "
                   "(100 MHz -> s_memtime at %.3f GHz); wall %.1f us", what[mode], nblk / 256, cyc, per, rt, cyc / rt * 0.1, ms * 1e3);
        if (mode == 0) fprintf(f, " -> %.1f TFLOP/s", (double)nblk * 4 * n * 4.0 * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12);
        fprintf(f, "\n");
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(dev);
}
static int fg_ws_trace_launch(fg_ctx* ctx, const IgemmArgs& a_in, int BN, dim3 grid, size_t lds) {
    IgemmArgs a = a_in;
    const size_t nblk = (size_t)grid.x * grid.y;
    unsigned long long* dev = nullptr;
    if (hipMalloc((void**)&dev, nblk * 128 * 8) != hipSuccess) return fg_set_err(ctx, FG_ERR_NOMEM, "trace buffer");
    (void)hipMemset(dev, 0, nblk * 128 * 8);
    a.dbg_trace = dev;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float wall_ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {                   // three to settle the clocks, the fourth is the one reported
        (void)hipEventRecord(e0, ctx->stream);
        if (BN == 64) {
            FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws_trace_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(igemm_ws_trace_kernel<64>, grid, dim3(512), lds, ctx->stream, a);
        } else {
            FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws_trace_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(igemm_ws_trace_kernel<128>, grid, dim3(512), lds, ctx->stream, a);
        }
        (void)hipEventRecord(e1, ctx->stream);
    }
    FG_CHECK_LAUNCH(ctx);
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipEventElapsedTime(&wall_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::vector<unsigned long long> host(nblk * 128);
    FG_HIP(ctx, hipMemcpy(host.data(), dev, nblk * 128 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    const char* path = getenv("FG_WS_TRACE_FILE");
    FILE* f = fopen(path ? path : "/tmp/fg_ws_trace.txt", "a");
    if (f) {
        static bool calibrated = false;
        if (!calibrated) { calibrated = true; fg_trace_calibrate(ctx, f); }
        fprintf(f, "# launch %s BN=%d blocks=%zu M=%d Npad=%d G=%d Kpad=%d wall_us=%.1f\n", a.tag ? a.tag : "?", BN, nblk, a.M, a.Npad, a.G, a.Kpad, wall_ms * 1e3);
        for (size_t b = 0; b < nblk; ++b) {
            const unsigned long long* r = host.data() + b * 128;
            const int kt = (int)(r[127] >> 32), xcc = (int)(r[127] & 0xffffffff);
            fprintf(f, "%zu %d %d %llu", b, xcc, kt, r[126]);
            for (int i = 0; i < (kt < 120 ? kt : 120) + 3; ++i) fprintf(f, " %llu", r[i]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return FG_OK;
}
static bool fg_ws_trace_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("FG_WS_TRACE"); on = e ? atoi(e) : 0; }
    return on != 0;
}
#endif   // FG_MEASURE
static bool fg_ws64_ns3() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("FG_IGEMM_WS64_NS3"); on = e ? atoi(e) : 1; }
    return on != 0;
}
static int launch_igemm_ws64x3(fg_ctx* ctx, const IgemmArgs& a, int P) {
    const size_t lds = (size_t)(3 * (WS_BM + 64) * WS_LDK + WS_BM) * sizeof(float);
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws64x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws64x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws64x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid(fg_cdiv(a.M, WS_BM) * (a.Npad / 64) * P, a.splits, 1);
    const double exec = 2.0 * (double)grid.x * WS_BM * 64 * (double)a.G * a.Kpad;
    const int epi = a.act_x ? 2 : (a.act_y ? 1 : 0);
#ifdef FG_MEASURE
    if (fg_ws_trace_on() && epi == 0 && !a.stats_part) return fg_ws_trace_launch(ctx, a, 64, grid, lds);
#endif
    char label[96];
    snprintf(label, sizeof(label), "igemm_ws64x3_kernel<%d>/%s", epi, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    if (epi == 2) hipLaunchKernelGGL(igemm_ws64x3_kernel<2>, grid, dim3(512), lds, ctx->stream, a);
    else if (epi == 1) hipLaunchKernelGGL(igemm_ws64x3_kernel<1>, grid, dim3(512), lds, ctx->stream, a);
    else hipLaunchKernelGGL(igemm_ws64x3_kernel<0>, grid, dim3(512), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
template <int BN>
static int launch_igemm_ws(fg_ctx* ctx, const IgemmArgs& a, int P) {
    if (BN == 64 && fg_ws64_ns3()) return launch_igemm_ws64x3(ctx, a, P);
    const size_t lds = (size_t)(WS_NS * (WS_BM + BN) * WS_LDK + WS_BM) * sizeof(float);
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws_act_kernel<BN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws_act_kernel<BN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid(fg_cdiv(a.M, WS_BM) * (a.Npad / BN) * P, a.splits, 1);
    const double exec = 2.0 * (double)grid.x * WS_BM * BN * (double)a.G * a.Kpad;
    const int epi = a.act_x ? 2 : (a.act_y ? 1 : 0);
#ifdef FG_MEASURE
    if (BN == 128 && fg_ws_trace_on() && epi == 0 && !a.stats_part) return fg_ws_trace_launch(ctx, a, 128, grid, lds);
#endif
    char label[96];
    if (epi) snprintf(label, sizeof(label), "igemm_ws_act_kernel<%d,%d>/%s", BN, epi, a.tag ? a.tag : "?");
    else snprintf(label, sizeof(label), "igemm_ws_kernel<%d>/%s", BN, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    if (epi == 2) hipLaunchKernelGGL((igemm_ws_act_kernel<BN, 2>), grid, dim3(512), lds, ctx->stream, a);
    else if (epi == 1) hipLaunchKernelGGL((igemm_ws_act_kernel<BN, 1>), grid, dim3(512), lds, ctx->stream, a);
    else hipLaunchKernelGGL(igemm_ws_kernel<BN>, grid, dim3(512), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 contraction emulated on the bf16 matrix pipe ("bf16x6"): every fp32 operand is pre-split into three bf16
// planes x = h + m + l (exact: 8+8+8 significand bits), and a*b is formed from the six plane products whose weight is
// >= 2^-16 (hh, hm, mh, mm, hl, lh); each bf16 x bf16 product is exact in the fp32 accumulator and the dropped products
// are <= 2^-24 relative, i.e. below one fp32 rounding.  Measured against fp64 the result is more accurate than the
// native fp32 MFMA (scripts/bf16x_probe.hip, profiles/r01_bf16x_probe.md) while v_mfma_f32_32x32x16_bf16 runs at 16x
// the fp32 MFMA rate: 6 products = 2.67x fewer matrix-pipe cycles.
// Same block structure as igemm_ws_kernel (256x128 tile, 4 MFMA + 4 loader waves); K-step = 16 channels = one 96-byte
// plane row per tile row, 3-stage LDS ring with 112-byte rows (conflict-free b128 fragment reads).
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define W6_ROWB 112
#define W6_NS 3
__device__ __forceinline__ f32x16 fg_mfma_bf16(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// BN = 128: MFMA waves 2x2, each 128x64 (4x2 tiles); BN = 64 (layers with 64 output channels): waves 4x1, each 64x64
template <int BN>
__global__ __launch_bounds__(512, 2) void igemm_ws6_kernel(const IgemmArgs a) {
    constexpr int WNW = BN / 64;                 // MFMA waves along N
    constexpr int WMW = 4 / WNW;                 // ... along M
    constexpr int MI = WS_BM / (WMW * 32);       // 32x32 tiles per wave along M (4 | 2)
    constexpr int NI = 2;
    constexpr int STAGE = (WS_BM + BN) * W6_ROWB;
    constexpr int NBCH = (BN * 6 + 255) / 256;   // B chunks per loader thread (3 | 2, the last partly filled for BN = 64)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    int* rowoff = (int*)(smem6 + W6_NS * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = a.Npad / BN;
    const int np = a.P;
    const int per_m = ntn * np;
    const int nmt = (a.M + WS_BM - 1) / WS_BM;
    int lin = blockIdx.x;
    if ((nmt & 7) == 0) {
        const int xcd = lin & 7, loc = lin >> 3;
        lin = (xcd * (nmt >> 3) + loc / per_m) * per_m + loc % per_m;
    }
    const int tile_m = lin / per_m;
    const int rem = lin - tile_m * per_m;
    const int tile_n = rem / np, p = rem - tile_n * np;
    const int split = blockIdx.y;

    if (tid < WS_BM) {
        int m = tile_m * WS_BM + tid, off = -1;
        if (m < a.M) {
            int n, y, x;
            fg_decode_m(m, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
            off = ((n * a.Ho + y * a.osy + a.ooy[p]) * a.Wo + x * a.osx + a.oox[p]) * a.N;
        }
        rowoff[tid] = off;
    }
    const int kc = a.Kpad / 16;          // K-steps (16-channel plane rows) per tap group
    const int cgA = a.Ca / 16;           // plane rows per pixel actually present in A6
    const int kt_all = a.G * kc;
    const int kt_per = (kt_all + a.splits - 1) / a.splits;
    const int kt0 = split * kt_per;
    const int KT = max(0, min(kt_all, kt0 + kt_per) - kt0);

    if (wid >= 4) {
        // ------------------------------------------------------------------ loader waves: 16-byte chunk c = lt + 256*i,
        // i < 6: A rows (c / 6, part c % 6), i >= 6: B rows
        const int lt = tid - 256;
        const long long a6_bytes = a.a_bytes / 4 * 6;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.A6, 0, (int)a6_bytes, 0x00020000);
        int ry[6], rx[6], rn[6], lds_a[6], part_a[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = lt + 256 * i;
            const int row = c / 6, part = c - row * 6;
            const int m = tile_m * WS_BM + row;
            int n, y, x;
            fg_decode_m(m < a.M ? m : 0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);
            rn[i] = n * a.Ha * a.Wa;
            ry[i] = m < a.M ? y * a.asy : -(1 << 20);
            rx[i] = x * a.asx;
            lds_a[i] = row * W6_ROWB + part * 16;
            part_a[i] = part * 16;
        }
        int lds_b[NBCH];
        bool okb[NBCH];
        const unsigned char* bptr[NBCH];
        int g = kt0 / kc;
        int cg = kt0 - g * kc;
#pragma unroll
        for (int i = 0; i < NBCH; ++i) {
            const int c = lt + 256 * i;            // 0 .. BN*6-1 over the B rows
            okb[i] = c < BN * 6;
            const int row = okb[i] ? c / 6 : 0, part = c - (c / 6) * 6;
            lds_b[i] = (WS_BM + row) * W6_ROWB + part * 16;
            bptr[i] = (const unsigned char*)a.B6 +
                      (((size_t)(p * a.G + g) * a.Npad + tile_n * BN + row) * kc + cg) * 96 + part * 16;
        }
        const size_t bjump = (size_t)(a.Npad - 1) * kc * 96;
        int voff[6];
#define W6_SET_GROUP()                                                                                   \
        {                                                                                                \
            const int go = a.goff[p][g < a.G ? g : 0];                                                   \
            const int oy = (int)(short)(go & 0xffff), ox = go >> 16;                                     \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                              \
                const int ya = ry[i] + oy, xa = rx[i] + ox;                                              \
                const bool ok = (unsigned)ya < (unsigned)a.Ha && (unsigned)xa < (unsigned)a.Wa;          \
                voff[i] = ok ? (rn[i] + ya * a.Wa + xa) * cgA * 96 + part_a[i] : FG_OOB;                 \
            }                                                                                            \
        }
        W6_SET_GROUP();
        f32x4 xa_[6], xb_[NBCH], ya_[6], yb_[NBCH], za_[6], zb_[NBCH];   // three tiles in flight (a K-step is only ~1.2 us)
#define W6_LOAD(ra, rb)                                                                                  \
        {                                                                                                \
            const bool kin = cg < cgA;                                                                   \
            _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                \
                ra[i] = fg_buffer_load4(arsrc, kin ? voff[i] + cg * 96 : FG_OOB);                        \
            _Pragma("unroll") for (int i = 0; i < NBCH; ++i) { rb[i] = *(const f32x4*)bptr[i]; bptr[i] += 96; } \
            if (++cg == kc) {                                                                            \
                cg = 0; ++g;                                                                             \
                _Pragma("unroll") for (int i = 0; i < NBCH; ++i) bptr[i] += bjump;                       \
                W6_SET_GROUP();                                                                          \
            }                                                                                            \
        }
#define W6_STORE(st, ra, rb)                                                                             \
        {                                                                                                \
            unsigned char* S = smem6 + (st) * STAGE;                                                     \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) *(f32x4*)(S + lds_a[i]) = ra[i];               \
            _Pragma("unroll") for (int i = 0; i < NBCH; ++i) if (okb[i]) *(f32x4*)(S + lds_b[i]) = rb[i]; \
        }
        // prologue: tiles 0 and 1 resident, tiles 2, 3, 4 in flight; tile t lives in stage t % 3
        if (KT > 0) { W6_LOAD(xa_, xb_); W6_STORE(0, xa_, xb_); }
        if (KT > 1) { W6_LOAD(xa_, xb_); W6_STORE(1, xa_, xb_); }
        if (KT > 2) { W6_LOAD(xa_, xb_); }
        if (KT > 3) { W6_LOAD(ya_, yb_); }
        if (KT > 4) { W6_LOAD(za_, zb_); }
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 3) {
            if (kt + 2 < KT) {
                W6_STORE(2, xa_, xb_);
                if (kt + 5 < KT) { W6_LOAD(xa_, xb_); }
            }
            __syncthreads();
            if (kt + 1 < KT) {
                if (kt + 3 < KT) {
                    W6_STORE(0, ya_, yb_);
                    if (kt + 6 < KT) { W6_LOAD(ya_, yb_); }
                }
                __syncthreads();
            }
            if (kt + 2 < KT) {
                if (kt + 4 < KT) {
                    W6_STORE(1, za_, zb_);
                    if (kt + 7 < KT) { W6_LOAD(za_, zb_); }
                }
                __syncthreads();
            }
        }
#undef W6_SET_GROUP
#undef W6_LOAD
#undef W6_STORE
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves (one per SIMD)
    const int wm = wid / WNW, wn = wid - wm * WNW;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int a_off = (wm * MI * 32 + (lane & 31)) * W6_ROWB + (lane >> 5) * 16;
    const int b_off = (WS_BM + wn * 64 + (lane & 31)) * W6_ROWB + (lane >> 5) * 16;
    s16x8 A0[MI], A1[MI], A2[MI], B0[NI], B1[NI], B2[NI];
#define W6_LDA(dst, S, pl) _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) dst[mi] = *(const s16x8*)((S) + a_off + mi * 32 * W6_ROWB + (pl) * 32);
#define W6_LDB(dst, S, pl) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) dst[ni] = *(const s16x8*)((S) + b_off + ni * 32 * W6_ROWB + (pl) * 32);
#define W6_PROD(Ax, Bx)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = fg_mfma_bf16(Ax[mi], Bx[ni], acc[mi][ni]);
    __syncthreads();              // tiles 0 and 1 are in the ring
    if (KT > 0) {
        W6_LDA(A0, smem6, 0) W6_LDA(A1, smem6, 1) W6_LDA(A2, smem6, 2)
        W6_LDB(B0, smem6, 0) W6_LDB(B1, smem6, 1) W6_LDB(B2, smem6, 2)
    }
    int sn = 1;
    for (int kt = 0; kt < KT; ++kt) {
        // every read of this iteration targets the NEXT tile's stage (sealed by the previous barrier); each fragment
        // register set is re-filled right after its last use, ordered so the first products of the next step find theirs
        const unsigned char* Sn = smem6 + sn * STAGE;
        const bool nxt = kt + 1 < KT;
        W6_PROD(A1, B1)
        W6_PROD(A2, B0)  if (nxt) { W6_LDA(A2, Sn, 2) }
        W6_PROD(A0, B2)  if (nxt) { W6_LDB(B2, Sn, 2) }
        W6_PROD(A1, B0)  if (nxt) { W6_LDA(A1, Sn, 1) }
        W6_PROD(A0, B1)  if (nxt) { W6_LDB(B1, Sn, 1) }
        W6_PROD(A0, B0)  if (nxt) { W6_LDA(A0, Sn, 0) W6_LDB(B0, Sn, 0) }
        sn = sn == W6_NS - 1 ? 0 : sn + 1;
        // bare barrier: these waves never write LDS and the stage the loaders overwrite next was last READ one full
        // iteration ago, so the fragment loads just issued may stay in flight across it
        asm volatile("s_barrier" ::: "memory");
    }
#undef W6_LDA
#undef W6_LDB
#undef W6_PROD

    float* outp = a.Out + (size_t)split * a.split_stride;
    const bool add_bias = (a.bias != nullptr) && (a.splits == 1);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, FG_OOB, 0x00020000);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = tile_n * BN + wn * 64 + ni * 32 + (lane & 31);
        const bool colok = col < a.N;
        float bv = add_bias ? a.bias[colok ? col : 0] : 0.f;
        asm volatile("" : "+v"(bv));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            fg_store_acc_tile(orsrc, rowoff, wm * MI * 32 + mi * 32, col, colok, bv, acc[mi][ni], lane);
    }
}

template <int BN>
static int launch_igemm_ws6(fg_ctx* ctx, const IgemmArgs& a, int P) {
    const size_t lds = (size_t)W6_NS * (WS_BM + BN) * W6_ROWB + WS_BM * sizeof(int);
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_ws6_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (a.a_bytes / 4 * 6 >= (long long)FG_OOB) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "igemm bf16x6: A planes must be < 2 GiB");
    dim3 grid(fg_cdiv(a.M, WS_BM) * (a.Npad / BN) * P, a.splits, 1);
    const double exec = 2.0 * (double)grid.x * WS_BM * BN * (double)a.G * a.Kpad;     // fp32-equivalent FLOPs
    char label[96];
    snprintf(label, sizeof(label), "igemm_ws6_kernel<%d>/%s", BN, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    hipLaunchKernelGGL(igemm_ws6_kernel<BN>, grid, dim3(512), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// fp32 rows -> split planes.  One thread per 8 consecutive channels (half a 16-channel plane row).
__device__ __forceinline__ unsigned fg_bf16_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void fg_split3(float v, unsigned& h, unsigned& m, unsigned& l) {
    h = fg_bf16_rn(v);
    const float r1 = v - __uint_as_float(h << 16);
    m = fg_bf16_rn(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = fg_bf16_rn(r2);
}
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long long halves, unsigned char* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < halves; i += (long long)gridDim.x * 256) {
        const f32x4 v0 = *(const f32x4*)(src + i * 8), v1 = *(const f32x4*)(src + i * 8 + 4);
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { fg_split3(v0[e], h[e], m[e], l[e]); fg_split3(v1[e], h[4 + e], m[4 + e], l[4 + e]); }
        unsigned char* o = dst + (i >> 1) * 96 + (i & 1) * 16;
        *(uint4*)(o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        *(uint4*)(o + 32) = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
        *(uint4*)(o + 64) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
}
int fg_launch_split_planes(fg_ctx* ctx, const float* src, long long rows, int C, void* dst) {
    if (C % 16) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "split planes: C=%d is not a multiple of 16", C);
    const long long halves = rows * (C / 8);
    if (halves == 0) return FG_OK;
    long long blocks = (halves + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    FgProfScope prof(ctx, fg_intern(ctx, "split_planes_kernel"), 0.0, 0.0, (double)halves * 8 * 10);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, src, halves, (unsigned char*)dst);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

template <int BM, int BN, int BK>
static int launch_igemm_t(fg_ctx* ctx, const IgemmArgs& a, int P) {
    const size_t lds = (size_t)(2 * (BM + BN) * (BK + 4) + BM) * sizeof(float);
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_act_kernel<BM, BN, BK, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)igemm_act_kernel<BM, BN, BK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    }
    dim3 grid(fg_cdiv(a.M, BM) * (a.Npad / BN) * P, a.splits, 1);
    // executed FLOPs: every tile runs the full padded contraction
    const double exec = 2.0 * (double)grid.x * BM * BN * (double)a.G * a.Kpad;
    const int epi = a.act_x ? 2 : (a.act_y ? 1 : 0);
    char label[96];
    if (epi) snprintf(label, sizeof(label), "igemm_act_kernel<%d,%d,%d,%d>/%s", BM, BN, BK, epi, a.tag ? a.tag : "?");
    else snprintf(label, sizeof(label), "igemm_kernel<%d,%d,%d>/%s", BM, BN, BK, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    if (epi == 2) hipLaunchKernelGGL((igemm_act_kernel<BM, BN, BK, 2>), grid, dim3(256), lds, ctx->stream, a);
    else if (epi == 1) hipLaunchKernelGGL((igemm_act_kernel<BM, BN, BK, 1>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((igemm_kernel<BM, BN, BK>), grid, dim3(256), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

long long fg_igemm_blocks(const IgemmArgs& a, int P, int tile) {
    switch (tile) {
        case 0: return (long long)fg_cdiv(a.M, 128) * (a.Npad / 128) * P;
        case 1: return (long long)fg_cdiv(a.M, 128) * (a.Npad / 64) * P;
        case 2: return (long long)fg_cdiv(a.M, 64) * (a.Npad / 64) * P;
        case 5: return (long long)fg_cdiv(a.M, WS_BM) * (a.Npad / 64) * P;
        case 4: return (long long)fg_cdiv(a.M, WS_BM) * (a.Npad / ((a.Npad % 128 == 0) ? 128 : 64)) * P;
    }
    return 0;
}

int fg_launch_igemm(fg_ctx* ctx, const IgemmArgs& a_in, int P, int tile) {
    IgemmArgs a = a_in;
    a.P = P;
#ifdef FG_MEASURE
    {   // measurement build only: how much of a launch is its output stores?
        static int ns = -1;
        if (ns < 0) { const char* e = getenv("FG_DEBUG_NOSTORE"); ns = e ? atoi(e) : 0; }
        a.dbg_nostore = ns;
    }
#else
    a.dbg_nostore = 0;
#endif
    if ((a.act_y || a.act_x) && (a.splits != 1 || a.A6 || !a.act_slope || (a.act_y && a.act_x)))
        return fg_set_err(ctx, FG_ERR_INVALID, "igemm: a fused PReLU needs splits == 1, the fp32 path and its slope");
    if (a.Ca % 4 != 0 || a.Kpad % 32 != 0) return fg_set_err(ctx, FG_ERR_INVALID, "igemm: Ca %% 4 / Kpad %% 32");
    if ((long long)a.Nb * a.Ho * a.Wo * a.N * 4 >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "igemm: output tensor must be < 2 GiB per launch");
    if (a.a_bytes <= 0 || a.a_bytes >= (long long)FG_OOB) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "igemm: A operand of %lld bytes (must be < 2 GiB per launch)", a.a_bytes);
    if (a.G > FG_MAX_GROUPS || P > 4 || a.splits < 1) return fg_set_err(ctx, FG_ERR_INVALID, "igemm: G/P/splits");
    switch (tile) {
        case 0: if (a.Npad % 128) break; return launch_igemm_t<128, 128, 32>(ctx, a, P);
        case 1: if (a.Npad % 64) break; return launch_igemm_t<128, 64, 32>(ctx, a, P);
        case 2: {
            if (a.Npad % 64) break;
            // 64-deep K-steps (half the barriers, 70 KB of LDS: still two blocks per CU) where the padded K allows: D's mid-size
            // convolutions 56.7 -> 54.7 us, G's first data gradient 199.8 -> 193.8 (FG_IGEMM_BK64=0 switches back)
            static int bk64 = -1;
            if (bk64 < 0) { const char* e = getenv("FG_IGEMM_BK64"); bk64 = e ? atoi(e) : 1; }
            if (bk64 && a.Kpad % 64 == 0 && ((a.G * (a.Kpad / 64)) % a.splits == 0)) return launch_igemm_t<64, 64, 64>(ctx, a, P);
            return launch_igemm_t<64, 64, 32>(ctx, a, P);
        }
        case 5: if (a.Npad % 64 || a.A6) break; return launch_igemm_ws<64>(ctx, a, P);
        case 4:
            if (a.A6) { if (a.Npad % 64) break; return (a.Npad % 128 == 0) ? launch_igemm_ws6<128>(ctx, a, P) : launch_igemm_ws6<64>(ctx, a, P); }
            if (a.Npad % 64) break;
            return (a.Npad % 128 == 0) ? launch_igemm_ws<128>(ctx, a, P) : launch_igemm_ws<64>(ctx, a, P);
    }
    return fg_set_err(ctx, FG_ERR_INVALID, "igemm: bad tile %d for Npad %d", tile, a.Npad);
}

// act (optional): the PReLU [+ Dropout] that follows the layer, applied to the finished sum in the same pass --
// act_y = prelu(out) [* mask * mscale], exactly prelu_fwd_kernel's expression; `out` (the pre-activation) is still written,
// the backward pass needs it
__global__ void sum_splits_kernel(const float* __restrict__ part, int splits, long long stride,
                                  const float* __restrict__ bias, int N, float* __restrict__ out, long long count4,
                                  const float* __restrict__ act_slope, const float* __restrict__ act_mask, float act_mscale,
                                  float* __restrict__ act_y) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    const float a = act_slope ? act_slope[0] : 1.f;
    for (; i < count4; i += step) {
        float4 s = ((const float4*)part)[i];
        int k = 1;
        for (; k + 2 < splits; k += 3) {             // three loads in flight; the additions stay in split order (bit-identical)
            const float4 t0 = ((const float4*)(part + k * stride))[i], t1 = ((const float4*)(part + (k + 1) * stride))[i];
            const float4 t2 = ((const float4*)(part + (k + 2) * stride))[i];
            s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
            s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
            s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
        }
        for (; k < splits; ++k) {
            float4 t = ((const float4*)(part + k * stride))[i];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (bias) {
            int c = (int)((i * 4) % N);
            s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3];
        }
        ((float4*)out)[i] = s;
        if (act_y) {
            float4 r;
            r.x = s.x > 0.f ? s.x : a * s.x; r.y = s.y > 0.f ? s.y : a * s.y;
            r.z = s.z > 0.f ? s.z : a * s.z; r.w = s.w > 0.f ? s.w : a * s.w;
            if (act_mask) {
                const float4 m = ((const float4*)act_mask)[i];
                r.x *= m.x * act_mscale; r.y *= m.y * act_mscale; r.z *= m.z * act_mscale; r.w *= m.w * act_mscale;
            }
            ((float4*)act_y)[i] = r;
        }
    }
}

__global__ void sum_splits_scalar_kernel(const float* __restrict__ part, int splits, long long stride, const float* __restrict__ bias,
                                         int N, float* __restrict__ out, long long count) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        float s = part[i];
        for (int k = 1; k < splits; ++k) s += part[k * stride + i];
        if (bias) s += bias[(int)(i % N)];
        out[i] = s;
    }
}

int fg_launch_sum_splits(fg_ctx* ctx, const float* part, int splits, long long stride, const float* bias, int N,
                         float* out, long long count, const FgActFuse* act) {
    if (count % 4 || N % 4 || stride % 4 || (((uintptr_t)part | (uintptr_t)out) & 15)) {
        // ragged output widths (a Linear with 10 outputs, ...): one element per thread, same order of additions; the PReLU that
        // follows is left to its own pass (act->applied = 0)
        const int blocks = (int)min((long long)4096, (count + 255) / 256);
        hipLaunchKernelGGL(sum_splits_scalar_kernel, dim3(blocks), dim3(256), 0, ctx->stream, part, splits, stride, bias, N, out, count);
        if (act) act->applied = 0;
        FG_CHECK_LAUNCH(ctx);
        return FG_OK;
    }
    long long c4 = count / 4;
    int blocks = (int)min((long long)2048, (c4 + 255) / 256);
    const bool fuse = act && act->y && act->slope && (!act->mask || ((uintptr_t)act->mask & 15) == 0);
    hipLaunchKernelGGL(sum_splits_kernel, dim3(blocks), dim3(256), 0, ctx->stream, part, splits, stride, bias, N, out,
                       c4, fuse ? act->slope : nullptr, fuse ? act->mask : nullptr, fuse ? act->mscale : 1.f, fuse ? act->y : nullptr);
    if (act) act->applied = fuse ? 1 : 0;
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// Split-K data gradient whose input is the output of an nn.PReLU [+ nn.Dropout]: the pass that sums the partials also runs that
// PReLU's backward (prelu_bwd_kernel's expressions: g = sum [* mask * mscale]; gx = x > 0 ? g : slope * g; slope gradient
// += x * g where x <= 0), one slope-gradient partial per block.
__global__ __launch_bounds__(256) void sum_splits_actbwd_kernel(const float* __restrict__ part, int splits, long long stride,
                                                                float* __restrict__ out, long long count4,
                                                                const float* __restrict__ x, const float* __restrict__ slope,
                                                                const float* __restrict__ mask, float mscale,
                                                                float* __restrict__ spart) {
    __shared__ float sh[4];
    const float a = slope[0];
    float acc = 0.f;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += step) {
        float4 g = ((const float4*)part)[i];
        for (int k = 1; k < splits; ++k) {
            const float4 t = ((const float4*)(part + k * stride))[i];
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
        }
        if (mask) {
            const float4 m = ((const float4*)mask)[i];
            g.x *= m.x * mscale; g.y *= m.y * mscale; g.z *= m.z * mscale; g.w *= m.w * mscale;
        }
        const float4 v = ((const float4*)x)[i];
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x; if (!(v.x > 0.f)) acc = fmaf(v.x, g.x, acc);
        o.y = v.y > 0.f ? g.y : a * g.y; if (!(v.y > 0.f)) acc = fmaf(v.y, g.y, acc);
        o.z = v.z > 0.f ? g.z : a * g.z; if (!(v.z > 0.f)) acc = fmaf(v.z, g.z, acc);
        o.w = v.w > 0.f ? g.w : a * g.w; if (!(v.w > 0.f)) acc = fmaf(v.w, g.w, acc);
        ((float4*)out)[i] = o;
    }
    // block sum (fixed shape: deterministic)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && spart) spart[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int fg_launch_sum_splits_actbwd(fg_ctx* ctx, const float* part, int splits, long long stride, float* out, long long count,
                                const FgActBwd* actb) {
    if (actb) actb->applied = 0;
    const bool ok = actb && actb->x && actb->slope && fg_fuse_prelu(ctx) && count % 4 == 0 && stride % 4 == 0 &&
                    ((uintptr_t)actb->x & 15) == 0 && (!actb->mask || ((uintptr_t)actb->mask & 15) == 0);
    const long long c4 = count / 4;
    const int blocks = (int)min((long long)1024, (c4 + 255) / 256);
    float* dp = (ok && actb->gslope) ? fg_defer_alloc(ctx, blocks) : nullptr;
    if (!ok || (actb->gslope && !dp)) return fg_launch_sum_splits(ctx, part, splits, stride, nullptr, 4, out, count, nullptr);
    hipLaunchKernelGGL(sum_splits_actbwd_kernel, dim3(blocks), dim3(256), 0, ctx->stream, part, splits, stride, out, c4,
                       actb->x, actb->slope, actb->mask, actb->mscale, dp);
    FG_CHECK_LAUNCH(ctx);
    if (dp) fg_defer_push(ctx, dp, blocks, 1, 0.f, actb->gslope);
    actb->applied = 1;
    return FG_OK;
}

// ---------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------
// XCD-aware block order for the pixel-split weight-gradient kernels: blocks of one pixel split (all tiles / taps /
// parities) read the same dY / X rows, so they should share an XCD's L2.  Hardware deals blocks round-robin over the 8
// XCDs in dispatch order; hand XCD k the k-th eighth of the split-major logical order instead.
__device__ __forceinline__ void fg_wgrad_block(int& tile, int& s, int& pg) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int n = gx * gy * gz;
    int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    {   // XCD k owns the dispatch slots k, k+8, ...: n/8 of them, one more for k < n%8
        const int k = lin & 7, q = n >> 3, rem = n & 7;
        lin = k * q + (k < rem ? k : rem) + (lin >> 3);
    }
    const int per = gx * gz;
    s = lin / per;
    const int r = lin - s * per;
    pg = r / gx;
    tile = r - pg * gx;
}

// Bias gradient: the blocks with tc == 0, g == 0 see every dY pixel of their split exactly once and also leave its
// per-channel sums in `bias_part` [P][S][Nd] (no separate column-sum pass over dY).
template <int BT, int BK, int OCC>  // square tile BT x BT (rows = dY channels, cols = X channels), K-step of BK pixels
__global__ __launch_bounds__(256, OCC) void wgrad_kernel(const WgradArgs a) {
    constexpr int R = BT * BK / 1024;   // float4 loads per thread per operand per K-step
    constexpr int F4 = BT / 4;   // float4 per pixel row
    constexpr int PSTEP = 256 / F4;
    constexpr int MI = BT / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                // [2][BK*BT]
    float* Xs = smem + 2 * BK * BT;  // [2][BK*BT]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntc = a.Cpad / BT;
    int bt_, s, pg;
    fg_wgrad_block(bt_, s, pg);
    const int tn = bt_ / ntc, tc = bt_ - tn * ntc;
    const int p = pg / a.G, g = pg - p * a.G;
    const int m0 = s * a.m_per_split;
    const int m1 = min(a.M, m0 + a.m_per_split);
    const int KT = (m1 > m0) ? (m1 - m0 + BK - 1) / BK : 0;

    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)a.d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
    const int lpix = tid / F4, lc = (tid - lpix * F4) * 4;
    const int chD = tn * BT + lc, chX = tc * BT + lc;
    const bool okD = chD < a.Nd, okX = chX < a.Cx;
    const int doy = a.doy[p], dox = a.dox[p], xoy = a.xoy[p][g], xox = a.xox[p][g];

    const bool want_bias = a.bias_part != nullptr && tc == 0 && g == 0;   // block-uniform
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 rd[R], rx[R];
    int mcur = m0;
    // Fast path (all hot-path shapes): Hm, Wm powers of two and a 32-pixel K-step never straddles two samples.  Then a
    // row's offsets are  (wave-uniform base of this K-step) + (per-thread constant), so the per-step address math is
    // scalar except one add and one bounds compare per row.
    const bool fast = a.lgW >= 0 && ((a.Hm * a.Wm) & (BK - 1)) == 0;
    int cD[R], cX[R], cy[R], cxx[R];
    if (fast) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int j = lpix + PSTEP * i;
            const int x = j & (a.Wm - 1), yr = j >> a.lgW;          // pixel inside the K-step (Wm < 32: several rows)
            const int xx = x * a.xsx + xox;
            cxx[i] = xx;
            cy[i] = yr * a.xsy + xoy;
            cD[i] = (((yr * a.dsy + doy) * a.Wd + x * a.dsx + dox) * a.Nd + chD) * 4;
            cX[i] = (((yr * a.xsy + xoy) * a.Wx + xx) * a.Cx + chX) * 4;
        }
    }
    const int sDn = a.Hd * a.Wd * a.Nd * 4, sDy = a.dsy * a.Wd * a.Nd * 4;   // byte strides per sample / per M-space row
    const int sXn = a.Hx * a.Wx * a.Cx * 4, sXy = a.xsy * a.Wx * a.Cx * 4;
#define FG_WLOAD()                                                                                          \
    {                                                                                                       \
        if (fast) {                                                                                         \
            const int rowi = mcur >> a.lgW;                 /* wave-uniform */                              \
            const int yb = rowi & (a.Hm - 1), nb = rowi >> a.lgH;                                           \
            const int x0 = mcur & (a.Wm - 1);               /* Wm > 32: the K-step starts inside a row */  \
            const int x0x = x0 * a.xsx;                                                                     \
            const int bD = nb * sDn + yb * sDy + x0 * a.dsx * a.Nd * 4;                                     \
            const int bX = nb * sXn + yb * sXy + x0x * a.Cx * 4, ybx = yb * a.xsy;                          \
            _Pragma("unroll") for (int i = 0; i < R; ++i) {                                                 \
                const bool ok = mcur + lpix + PSTEP * i < m1;                                               \
                const bool iny = (unsigned)(ybx + cy[i]) < (unsigned)a.Hx;                                  \
                const bool inx = okX && (unsigned)(cxx[i] + x0x) < (unsigned)a.Wx;                          \
                rd[i] = fg_buffer_load4(drsrc, (ok && okD) ? bD + cD[i] : FG_OOB);                          \
                rx[i] = fg_buffer_load4(xrsrc, (ok && inx && iny) ? bX + cX[i] : FG_OOB);                   \
            }                                                                                               \
        } else {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < R; ++i) {                                                 \
                const int m = mcur + lpix + PSTEP * i;                                                      \
                const bool ok = m < m1;                                                                     \
                int n, y, x;                                                                                \
                fg_decode_m(ok ? m : 0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);                                 \
                const int yd = y * a.dsy + doy, xd = x * a.dsx + dox;                                       \
                const int yx = y * a.xsy + xoy, xx = x * a.xsx + xox;                                       \
                const int od = (((n * a.Hd + yd) * a.Wd + xd) * a.Nd + chD) * 4;                            \
                const int ox = (((n * a.Hx + yx) * a.Wx + xx) * a.Cx + chX) * 4;                            \
                const bool inx = (unsigned)yx < (unsigned)a.Hx && (unsigned)xx < (unsigned)a.Wx;            \
                rd[i] = fg_buffer_load4(drsrc, (ok && okD) ? od : FG_OOB);                                  \
                rx[i] = fg_buffer_load4(xrsrc, (ok && okX && inx) ? ox : FG_OOB);                           \
            }                                                                                               \
        }                                                                                                   \
        mcur += BK;                                                                                         \
    }
#define FG_WSTORE(buf)                                                                                      \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < R; ++i) {                                                     \
            if (want_bias) bsum += rd[i];                                                                   \
            *(f32x4*)(Ds + (buf) * BK * BT + (lpix + PSTEP * i) * BT + lc) = rd[i];                         \
            *(f32x4*)(Xs + (buf) * BK * BT + (lpix + PSTEP * i) * BT + lc) = rx[i];                         \
        }                                                                                                   \
    }

    // BT == 64 (the 64-channel layers): intra-block split-K.  Each wave multiplies the WHOLE 64 x 64 tile over its own quarter of
    // the K-step's pixels and the four partial tiles are summed through LDS at the end.  A lane reads TWO channels of a pixel with
    // one ds_read_b64 -- lane (i, k) at [pixel 2 kp + k][channels 2 i, 2 i + 1] -- so register t of the read is an MFMA fragment
    // whose 32 rows are the channels 2 i + t (the channel <-> accumulator-row assignment is free; undone in the epilogue): 2 LDS
    // reads per 4 MFMAs instead of the 2 per MFMA of the one-32x32-tile-per-wave layout (4.9 instructions per MFMA, 113 TFLOP/s).
    constexpr bool SK = (BT == 64 && BK == 64);
    constexpr int AT = SK ? 2 : MI;
    f32x16 acc[AT][AT];
#pragma unroll
    for (int mi = 0; mi < AT; ++mi)
#pragma unroll
        for (int ni = 0; ni < AT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int d_lds = SK ? (16 * wid + (lane >> 5)) * BT + 2 * (lane & 31) : (lane >> 5) * BT + wm * (BT / 2) + (lane & 31);
    const int x_lds = SK ? d_lds : (lane >> 5) * BT + wn * (BT / 2) + (lane & 31);

    if (KT > 0) {
        FG_WLOAD();
        FG_WSTORE(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) FG_WLOAD();
        if constexpr (SK) {
            const float* Db = Ds + cur * BK * BT + d_lds;
            const float* Xb = Xs + cur * BK * BT + x_lds;
            f32x2 af[2], bf[2];           // fragment double buffer
            af[0] = *(const f32x2*)Db; bf[0] = *(const f32x2*)Xb;
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                const int c = kp & 1;
                if (kp + 1 < 8) {
                    af[c ^ 1] = *(const f32x2*)(Db + (2 * kp + 2) * BT);
                    bf[c ^ 1] = *(const f32x2*)(Xb + (2 * kp + 2) * BT);
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0], bf[c][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0], bf[c][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1], bf[c][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1], bf[c][1], acc[1][1], 0, 0, 0);
            }
        } else {
            const float* Db = Ds + cur * BK * BT + d_lds;
            const float* Xb = Xs + cur * BK * BT + x_lds;
            float af[2][MI], bf[2][MI];   // fragment double buffer
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { af[0][mi] = Db[mi * 32]; bf[0][mi] = Xb[mi * 32]; }
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int c = (kk >> 1) & 1;
                if (kk + 2 < BK) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        af[c ^ 1][mi] = Db[(kk + 2) * BT + mi * 32];
                        bf[c ^ 1][mi] = Xb[(kk + 2) * BT + mi * 32];
                    }
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < MI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][mi], bf[c][ni], acc[mi][ni], 0, 0, 0);
            }
        }
        if (more) FG_WSTORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef FG_WLOAD
#undef FG_WSTORE

    float* part = a.Part + ((size_t)pg * a.S + s) * a.Npad * a.Cpad;
    if constexpr (SK) {
        // the four waves' partial tiles -> LDS [wave][tile t,u][r][lane] (exactly the 64 KB of the two operand buffers; the K loop
        // ended with a barrier), summed in a fixed order, stored with the channel permutation undone
        float* red = smem;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wid * 4 + t * 2 + u) * 16 + r) * 64 + lane] = acc[t][u][r];
        __syncthreads();
        const int t = wid >> 1, u = wid & 1;                 // this wave finishes tile (t, u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = ((t * 2 + u) * 16 + r) * 64 + lane;
            const float v = (red[e] + red[e + 4096]) + (red[e + 8192] + red[e + 12288]);
            const int row = tn * BT + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + t;
            const int col = tc * BT + 2 * (lane & 31) + u;
            part[(size_t)row * a.Cpad + col] = v;
        }
        if (want_bias) __syncthreads();                      // `red` is read; the bias sums below reuse the same LDS
    } else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) {
            const int col = tc * BT + wn * (BT / 2) + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tn * BT + wm * (BT / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                part[(size_t)row * a.Cpad + col] = acc[mi][ni][r];
            }
        }
    if (want_bias) {                                         // per-channel sums of this split's dY pixels
        float* sh = smem;                                    // [PSTEP pixel rows][BT channels] (the K loop is finished)
        *(f32x4*)(sh + lpix * BT + lc) = bsum;
        __syncthreads();
        if (tid < BT && tn * BT + tid < a.Nd) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < PSTEP; ++q) t += sh[q * BT + tid];             // fixed order: deterministic
            a.bias_part[((size_t)p * a.S + s) * a.Nd + tn * BT + tid] = t;
        }
    }
}

template <int BT, int BK, int OCC>
static int launch_wgrad_t(fg_ctx* ctx, const WgradArgs& a, int P) {
    const size_t lds = (size_t)(4 * BK * BT) * sizeof(float);
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wgrad_kernel<BT, BK, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    }
    dim3 grid((a.Npad / BT) * (a.Cpad / BT), a.S, P * a.G);
    const double exec = 2.0 * (double)a.Npad * a.Cpad * (double)P * a.G * (double)a.S * fg_round_up(a.m_per_split, 32);
    char label[96];
    snprintf(label, sizeof(label), "wgrad_kernel<%d>/%s", BT, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    hipLaunchKernelGGL((wgrad_kernel<BT, BK, OCC>), grid, dim3(256), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

int fg_launch_wgrad(fg_ctx* ctx, const WgradArgs& a, int P, int tile) {
    if (a.Nd % 4 || a.Cx % 4 || a.m_per_split % 32) return fg_set_err(ctx, FG_ERR_INVALID, "wgrad: alignment");
    if (a.G > FG_MAX_GROUPS || P > 4) return fg_set_err(ctx, FG_ERR_INVALID, "wgrad: G/P");
    if (a.d_bytes <= 0 || a.x_bytes <= 0 || a.d_bytes >= (long long)FG_OOB || a.x_bytes >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad: operands must be < 2 GiB per launch");
    // K-steps of 16 pixels at four blocks per CU (16 waves): measured 4-7 % faster than 32-pixel steps at two blocks per CU --
    // the kernel's limiter is load latency, and the extra resident waves hide it better than a longer step does
    // (the 64-tile is the other way round: 89 vs 106 TFLOP/s at 16 vs 32 pixels, and better again at 64)
    if (tile == 0 && a.Npad % 128 == 0 && a.Cpad % 128 == 0) return launch_wgrad_t<128, 16, 4>(ctx, a, P);
    // the 64-tile (one accumulator tile per wave) wants LONG K-steps: 64 pixels 687 us, 32 pixels 737 us, 16 pixels slower
    // still on the c2f 64-channel layers (FG_WGRAD64_BK=32 switches back)
    static int w64 = -1;
    if (w64 < 0) { const char* e = getenv("FG_WGRAD64_BK"); w64 = e ? atoi(e) : 64; }
    if (tile == 2 && a.Npad % 64 == 0 && a.Cpad % 64 == 0 && w64 == 64 && a.m_per_split % 64 == 0) return launch_wgrad_t<64, 64, 2>(ctx, a, P);
    if (tile == 2 && a.Npad % 64 == 0 && a.Cpad % 64 == 0) return launch_wgrad_t<64, 32, 2>(ctx, a, P);
    return fg_set_err(ctx, FG_ERR_INVALID, "wgrad: bad tile %d for %dx%d", tile, a.Npad, a.Cpad);
}

// ---------------------------------------------------------------------------------
// reference <-> packed weight layouts
// ---------------------------------------------------------------------------------
void fg_fold_window(int k, int pad, int* T, int* rmin) {
    int lo = fg_fold_r(0, 0, pad), hi = fg_fold_r(1, k - 1, pad);
    *rmin = lo;
    *T = hi - lo + 1;
}

__device__ __forceinline__ int dev_fold_r(int parity, int d, int pad) {
    int v = parity + d - pad;
    return (v >= 0) ? (v >> 1) : -((-v + 1) >> 1);
}

// Winograd F(2x2, 3x3) weight transform (WeightMap::wino, wino.hip): U[i][j] = (G t G^T)[i][j] of a 3x3 sub-kernel t,
// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1].
__device__ __forceinline__ float fg_wino_u(const float* t, int i, int j) {
    float r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float w0 = t[a * 3], w1 = t[a * 3 + 1], w2 = t[a * 3 + 2];
        r[a] = j == 0 ? w0 : (j == 3 ? w2 : 0.5f * ((w0 + w2) + (j == 1 ? w1 : -w1)));
    }
    return i == 0 ? r[0] : (i == 3 ? r[2] : 0.5f * ((r[0] + r[2]) + (i == 1 ? r[1] : -r[1])));
}
// all 16 positions at once: the row pass once (12 values), then the column pass -- the SAME expressions as fg_wino_u position by
// position (bit-identical), ~60 operations instead of 16 x 20 with run-time selects (the re-pack launch was VALU-bound: 28 us)
__device__ __forceinline__ void fg_wino_u16(const float* t, float* u) {
    float r[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float w0 = t[a * 3], w1 = t[a * 3 + 1], w2 = t[a * 3 + 2];
        const float e = w0 + w2;
        r[a][0] = w0; r[a][1] = 0.5f * (e + w1); r[a][2] = 0.5f * (e + (-w1)); r[a][3] = w2;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float e = r[0][j] + r[2][j];
        u[0 + j] = r[0][j]; u[4 + j] = 0.5f * (e + r[1][j]); u[8 + j] = 0.5f * (e + (-r[1][j])); u[12 + j] = r[2][j];
    }
}
__device__ __forceinline__ float packed_from_taps(const WeightMap& wm, const float* w, int p, int g);
__device__ __forceinline__ float folded_tap_at(const WeightMap& wm, const float* w, int py, int px, int ty, int tx);
// The 3x3 sub-kernel t of one (out, in) pair whose transform goes to (parity p, group g) of the forward (bwd = 0) or data-gradient
// (bwd = 1) Winograd pack; w = the pair's k*k reference taps.
//   forward: kind 0, k = 3: t = w;  kind 1 (folded nearest-x2, 3x3 window): t = the folded taps of parity p;
//            wino 2 (5x5): group g = (a, b): t[dy][dx] = w[3a + dy][3b + dx] (0 beyond the 5x5 window)
//   data gradient = the same convolution with the taps flipped and in / out exchanged: kind 0: t[dy][dx] = w[2-dy][2-dx];
//            kind 1: group g = the forward parity, its folded taps flipped; wino 2: the sub-kernels of the FLIPPED 5x5 kernel,
//            t[dy][dx] = w[4 - 3a - dy][4 - 3b - dx]
__device__ __forceinline__ void fg_wino_subkernel(const WeightMap& wm, const float* w, int bwd, int p, int g, float* t) {
    if (wm.wino == 2) {
        const int a = g >> 1, b = g & 1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int u = 3 * a + dy, v = 3 * b + dx;
                const int uu = bwd ? 4 - u : u, vv = bwd ? 4 - v : v;
                t[dy * 3 + dx] = (u < 5 && v < 5) ? w[uu * 5 + vv] : 0.f;
            }
        return;
    }
    const int par = wm.kind == 1 ? (bwd ? g : p) : 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int src = bwd ? 8 - q : q;
        // (a Winograd pack of a folded layer has the 3 x 3 window T = 3 -- fg_geom_set_wino --: window coordinates as constants, not
        // src / wm.T at run time: the nine integer divisions per sub-kernel were a third of the re-pack launch's instructions)
        t[q] = wm.kind == 1 ? folded_tap_at(wm, w, par >> 1, par & 1, src / 3, src % 3) : w[src];
    }
}

// value of the (possibly tap-folded) weight for parity p, group g, reference out-channel o, in-channel i
__device__ __forceinline__ float packed_weight_value(const WeightMap& wm, const float* __restrict__ W, int p, int g,
                                                     int o, int i) {
    const int kk = wm.k * wm.k;
    const float* w = W + ((size_t)o * wm.I + i) * kk;
    if (wm.kind == 0) return w[g];
    const int py = p >> 1, px = p & 1, ty = g / wm.T, tx = g - ty * wm.T;
    // source offset r = t + rmin collects the taps d with floor((parity + d - pad)/2) == r, i.e. d in {2r-parity+pad, +1}
    const int dy0 = 2 * (ty + wm.rmin) - py + wm.pad, dx0 = 2 * (tx + wm.rmin) - px + wm.pad;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int dy = dy0 + a;
        if ((unsigned)dy >= (unsigned)wm.k) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int dx = dx0 + b;
            if ((unsigned)dx < (unsigned)wm.k) s += w[dy * wm.k + dx];
        }
    }
    return s;
}

__global__ void pack_weights_kernel(const WeightMap wm, int mode, const float* __restrict__ W, float* __restrict__ Bp,
                                    int rows_pad, int cols_pad, long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % cols_pad);
    long long t = idx / cols_pad;
    const int row = (int)(t % rows_pad);
    const int pg = (int)(t / rows_pad);
    const int p = pg / wm.G, g = pg - p * wm.G;
    // packed (row, col) -> packed (o, i)
    const int po = mode == 0 ? row : col, pi = mode == 0 ? col : row;
    float v = 0.f;
    if (po < wm.O && pi < wm.I) {
        int o = po, i = pi;
        if (wm.o_hw > 1) { int hw = po / wm.o_c, c = po - hw * wm.o_c; o = c * wm.o_hw + hw; }
        if (wm.i_hw > 1) { int hw = pi / wm.i_c, c = pi - hw * wm.i_c; i = c * wm.i_hw + hw; }
        if (wm.wino) {       // pg = (parity, group, position) of the Winograd pack
            int PP, KG; fg_wino_pack_shape(wm.kind, wm.wino, mode, &PP, &KG);
            const int pos = pg & 15, gg = (pg >> 4) % KG, pp = (pg >> 4) / KG;
            float t[9];
            fg_wino_subkernel(wm, W + ((size_t)o * wm.I + i) * wm.k * wm.k, mode, pp, gg, t);
            v = fg_wino_u(t, pos >> 2, pos & 3);
        } else
        v = packed_weight_value(wm, W, p, g, o, i);
    }
    if (wm.wino) {
        int PP, KG; fg_wino_pack_shape(wm.kind, wm.wino, mode, &PP, &KG);
        Bp[fg_wino_pack_at((pg >> 4) / KG, (pg >> 4) % KG, KG, rows_pad, cols_pad, pg & 15, row, col)] = v;      // the order wino_kernel's LDS stage wants
    } else Bp[idx] = v;
}

int fg_launch_pack_weights(fg_ctx* ctx, const WeightMap& wm, int mode, const float* W, float* Bp, int rows_pad,
                           int cols_pad) {
    long long total = (long long)wm.P * wm.G * rows_pad * cols_pad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(fg_cdiv(total, 256)), dim3(256), 0, ctx->stream, wm, mode, W, Bp,
                       rows_pad, cols_pad, total);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// packed value from a pair's k*k taps held in LDS (same arithmetic and summation order as packed_weight_value)
__device__ __forceinline__ float packed_from_taps(const WeightMap& wm, const float* w, int p, int g) {
    if (wm.kind == 0) return w[g];
    const int ty = g / wm.T;
    return folded_tap_at(wm, w, p >> 1, p & 1, ty, g - ty * wm.T);
}
// the folded tap at window position (ty, tx) of output parity (py, px): the sum of the <= 2 x 2 reference taps that land on it
__device__ __forceinline__ float folded_tap_at(const WeightMap& wm, const float* w, int py, int px, int ty, int tx) {
    const int dy0 = 2 * (ty + wm.rmin) - py + wm.pad, dx0 = 2 * (tx + wm.rmin) - px + wm.pad;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int dy = dy0 + a;
        if ((unsigned)dy >= (unsigned)wm.k) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int dx = dx0 + b;
            if ((unsigned)dx < (unsigned)wm.k) s += w[dy * wm.k + dx];
        }
    }
    return s;
}

// LDS row of one out-channel in a mode-7 job: 16 in-channels x k*k taps, +1 so rows fall on distinct banks.  Round 4: the staging
// area is DYNAMIC shared memory sized for the largest job of the net (fg_pack_lds_floats) -- as a static 16 x (16 x 49 + 1) array it
// took 50 KB in every block, three blocks per CU, whatever the net's kernels were (3x3 needs 9 KB, the 16^3 brick of mode 10 17 KB)
#define PK_ROW_OF(kk) (16 * (kk) + 1)
long long fg_pack_lds_floats(int mode, int k) {
    if (mode == 7) return 16LL * PK_ROW_OF(k * k);
    if (mode == 8) return 32 * 33;
    if (mode == 10) return 16 * 273;
    return 0;
}
// ADAM = true: the fused optimizer + re-pack launch.  Every weight a job reads is the freshly UPDATED value of that element
// (fg_adam_elem: penalty, clamp, Adam; parameter and moments written back) -- each parameter is read by exactly one thread of
// exactly one job, so the update happens once.  `params` is the flat parameter vector (ADAM: == ad.p).
template <bool ADAM>
__device__ __forceinline__ float pk_load(const float* __restrict__ params, const AdamArgs& ad, const AdamScalars& ak, long long idx) {
    if constexpr (ADAM) return fg_adam_elem(ad, ak, idx);
    else return params[idx];
}
template <bool ADAM>
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, int njobs, long long total,
                                                        const float* __restrict__ params, const AdamArgs ad, const AdamScalars ak) {
    extern __shared__ float taps[];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int j = 0;
    {   // block-uniform job lookup (every job's count is a multiple of 256)
        const long long b0 = (long long)blockIdx.x * blockDim.x;
        if (b0 >= total) return;
        while (j + 1 < njobs && b0 >= jobs[j + 1].start) ++j;
    }
    const PackJob& jb = jobs[j];
    const long long loc = idx - jb.start;
    const WeightMap& wm = jb.wm;
    const long long w0 = jb.src_off;         // this job's weights start here in the flat parameter vector
    if (jb.mode == 7) {
        // conv layer: both packs from one pass.  A block owns a 16 x 16 patch of (out, in) channel pairs: the reference
        // weights of one out-channel and 16 consecutive in-channels are ONE contiguous run of 16*k*k floats, so the patch is
        // 16 coalesced runs into LDS; every thread then forms the (tap-folded) values of its pair from LDS and writes them
        // with the in-channel fastest (forward pack) and, in a second role, with the out-channel fastest (data-gradient pack).
        const int kk = wm.k * wm.k, run = 16 * kk;
        const int PK_ROW = PK_ROW_OF(kk);
        const int patch = (int)(loc >> 8), t = (int)(loc & 255);
        const int po0 = (patch / jb.npi) * 16, pi0 = (patch % jb.npi) * 16;
        // (16 runs of 16 * kk floats; rows outside the layer read as zero.  No run-time division: the in-channel bound is a bound on
        // the offset inside the run)
        // Four rows' loads are issued before their LDS stores: with one load -> wait -> store per iteration the 256 blocks of G's two
        // up-convolutions (one per CU, four waves) spent 33 us walking 32 dependent memory latencies.
        const int off_end = min(run, (wm.I - pi0) * kk);
        const int nj = (run + 255) >> 8;                     // <= 4 (k <= 7)
        for (int a0 = 0; a0 < 16; a0 += 4) {
            float v[4][4];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const int po = po0 + a0 + aa;
                const long long src = w0 + ((long long)po * wm.I + pi0) * kk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = t + 256 * j;
                    v[aa][j] = (j < nj && po < wm.O && off < off_end) ? pk_load<ADAM>(params, ad, ak, src + off) : 0.f;
                }
            }
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = t + 256 * j;
                    if (j < nj && off < run) taps[(a0 + aa) * PK_ROW + off] = v[aa][j];
                }
        }
        __syncthreads();
        const int ng = wm.P * wm.G;
        {   // forward pack [p][g][O_pad][I_pad]: lanes run over the in-channel.  Winograd pack (chunk images [pos][k half][64 out][4 in]):
            // lanes run over the OUT-channel and a wave holds one quad of in-channels, so that each of its stores is one contiguous
            // 256-byte run of an image (lanes over the in-channel wrote four 64-byte pieces per store: 28 us per re-pack, round 5)
            const int a = wm.wino ? (t & 15) : (t >> 4), b = wm.wino ? (t >> 4) : (t & 15), po = po0 + a, pi = pi0 + b;
            if (po < jb.rows && pi < jb.cols) {
                const float* w = taps + a * PK_ROW + b * kk;
                float* d = jb.dst + (size_t)po * jb.cols + pi;
                const size_t tile = (size_t)jb.rows * jb.cols;
                if (wm.wino) {           // Winograd: U[parity][group][pos][out][in] in the order of wino_kernel's LDS stage
                    int PP, KG; fg_wino_pack_shape(wm.kind, wm.wino, 0, &PP, &KG);
                    for (int pp = 0; pp < PP; ++pp)
                        for (int gg = 0; gg < KG; ++gg) {
                            float tt[9], uu[16];
                            fg_wino_subkernel(wm, w, 0, pp, gg, tt);
                            fg_wino_u16(tt, uu);
                            float* d0 = jb.dst + fg_wino_pack_at(pp, gg, KG, jb.rows, jb.cols, 0, po, pi);      // position stride: 512 floats
#pragma unroll
                            for (int pos = 0; pos < 16; ++pos) d0[pos * 512] = uu[pos];
                        }
                } else
                for (int pg = 0; pg < ng; ++pg) d[(size_t)pg * tile] = packed_from_taps(wm, w, pg / wm.G, pg % wm.G);
            }
        }
        {   // data-gradient pack [p][g][I_pad][O_pad]: lanes run over the out-channel (Winograd: over the in-channel, which is the
            // 64-wide axis of ITS images; see above)
            const int a = wm.wino ? (t >> 4) : (t & 15), b = wm.wino ? (t & 15) : (t >> 4), po = po0 + a, pi = pi0 + b;
            if (pi < jb.rows2 && po < jb.cols2) {
                const float* w = taps + a * PK_ROW + b * kk;
                float* d = jb.dst2 + (size_t)pi * jb.cols2 + po;
                const size_t tile = (size_t)jb.rows2 * jb.cols2;
                if (wm.wino) {           // data gradient: roles of out / in exchanged, taps flipped, parities become K groups
                    int PP, KG; fg_wino_pack_shape(wm.kind, wm.wino, 1, &PP, &KG);
                    for (int gg = 0; gg < KG; ++gg) {
                        float tt[9], uu[16];
                        fg_wino_subkernel(wm, w, 1, 0, gg, tt);
                        fg_wino_u16(tt, uu);
                        float* d0 = jb.dst2 + fg_wino_pack_at(0, gg, KG, jb.rows2, jb.cols2, 0, pi, po);
#pragma unroll
                        for (int pos = 0; pos < 16; ++pos) d0[pos * 512] = uu[pos];
                    }
                } else
                for (int pg = 0; pg < ng; ++pg) d[(size_t)pg * tile] = packed_from_taps(wm, w, pg / wm.G, pg % wm.G);
            }
        }
        return;
    }
    if (jb.mode == 8) {
        // Linear layer (k = 1, optional View permutations on either axis): both packs from one pass over a 32 x 32 patch.
        // The forward pack keeps the reference's in-feature-fastest order (coalesced both ways); the data-gradient pack is its
        // transpose, written out-feature-fastest from an LDS tile (a thread-per-element transpose read one 4-byte word per
        // cache line: 47 us per re-pack launch).
        float* tl = taps;                                   // [32][33]
        const int patch = (int)(loc >> 8), t = (int)(loc & 255);
        const int po0 = (patch / jb.npi) * 32, pi0 = (patch % jb.npi) * 32;
        const int c = t & 31, r0 = t >> 5;
        for (int rr = r0; rr < 32; rr += 8) {
            const int po = po0 + rr, pi = pi0 + c;
            float v = 0.f;
            if (po < wm.O && pi < wm.I) {
                int o = po, i = pi;
                if (wm.o_hw > 1) { int hw = po / wm.o_c, cc = po - hw * wm.o_c; o = cc * wm.o_hw + hw; }
                if (wm.i_hw > 1) { int hw = pi / wm.i_c, cc = pi - hw * wm.i_c; i = cc * wm.i_hw + hw; }
                v = pk_load<ADAM>(params, ad, ak, w0 + (long long)o * wm.I + i);
            }
            tl[rr * 33 + c] = v;
            if (po < jb.rows && pi < jb.cols) jb.dst[(size_t)po * jb.cols + pi] = v;
        }
        __syncthreads();
        for (int rr = r0; rr < 32; rr += 8) {
            const int pi = pi0 + rr, po = po0 + c;
            if (pi < jb.rows2 && po < jb.cols2) jb.dst2[(size_t)pi * jb.cols2 + po] = tl[c * 33 + rr];
        }
        return;
    }
    if (jb.mode == 10) {
        // Linear behind a View (in-features permuted: reference column i = c*HW + hw <-> packed column pi = hw*C + c): three arrays
        // with three different fastest axes -- the reference weights run along hw, the forward pack along c, the data-gradient
        // pack along the out-feature.  A block owns a 16 (out) x 16 (c) x 16 (hw) brick, read and written in 64-byte runs along
        // each array's own fastest axis through LDS (the 32 x 32 patch of mode 8 read this weight one word per cache line:
        // 270 us per re-pack of the 65536 -> 512 layer of models_c2f.lua:262, and seven such streams in the fused optimizer launch).
        float* br = taps;                                    // [16 o][16 c][16 hw], strides 273 / 17 / 1
        const int brick = (int)(loc >> 8), t = (int)(loc & 255);
        const int C = wm.i_c, HW = wm.i_hw;
        const int nc = (C + 15) / 16, nhw = jb.npi;          // bricks along c / hw (hw over the padded column extent)
        const int bo = brick / (nc * nhw), rem = brick - bo * (nc * nhw), bc = rem / nhw, bh = rem - bc * nhw;
        const int o0 = bo * 16, c0 = bc * 16, h0 = bh * 16;
        {   // read: lanes along hw
            const int hw = h0 + (t & 15), c = c0 + (t >> 4);
#pragma unroll 4
            for (int oo = 0; oo < 16; ++oo) {
                const int po = o0 + oo;
                float v = 0.f;
                if (po < wm.O && c < C && hw < HW) {
                    int o = po;
                    if (wm.o_hw > 1) { int ohw = po / wm.o_c, cc = po - ohw * wm.o_c; o = cc * wm.o_hw + ohw; }
                    v = pk_load<ADAM>(params, ad, ak, w0 + (long long)o * wm.I + (long long)c * HW + hw);
                }
                br[oo * 273 + (t >> 4) * 17 + (t & 15)] = v;
            }
        }
        __syncthreads();
        {   // forward pack [po][hw*C + c]: lanes along c
            const int c = c0 + (t & 15), hw = h0 + (t >> 4);
            const long long pi = (long long)hw * C + c;
            if (c < C && pi < jb.cols)
#pragma unroll 4
                for (int oo = 0; oo < 16; ++oo)
                    if (o0 + oo < jb.rows) jb.dst[(size_t)(o0 + oo) * jb.cols + pi] = br[oo * 273 + (t & 15) * 17 + (t >> 4)];
        }
        {   // data-gradient pack [hw*C + c][po]: lanes along the out-feature
            const int po = o0 + (t & 15), cc = t >> 4;
            if (po < jb.cols2)
#pragma unroll 4
                for (int hh = 0; hh < 16; ++hh) {
                    const long long pi = (long long)(h0 + hh) * C + c0 + cc;
                    if (c0 + cc < C && pi < jb.rows2) jb.dst2[(size_t)pi * jb.cols2 + po] = br[(t & 15) * 273 + cc * 17 + hh];
                }
        }
        return;
    }
    if (idx >= total || loc >= jb.count) return;
    if (jb.mode == 9) {          // parameters no pack reads (BatchNorm, PReLU slopes, unpacked biases): the update alone
        if (ADAM) (void)fg_adam_elem(ad, ak, w0 + loc);
        return;
    }
    if (jb.mode <= 1) {
        const float* W = params + w0;        // (never in a fused launch: these jobs read a weight more than once)
        const int col = (int)(loc % jb.cols);
        long long t = loc / jb.cols;
        const int row = (int)(t % jb.rows);
        const int pg = (int)(t / jb.rows);
        const int p = pg / wm.G, g = pg - p * wm.G;
        const int po = jb.mode == 0 ? row : col, pi = jb.mode == 0 ? col : row;
        float v = 0.f;
        if (po < wm.O && pi < wm.I) {
            int o = po, i = pi;
            if (wm.o_hw > 1) { int hw = po / wm.o_c, c = po - hw * wm.o_c; o = c * wm.o_hw + hw; }
            if (wm.i_hw > 1) { int hw = pi / wm.i_c, c = pi - hw * wm.i_c; i = c * wm.i_hw + hw; }
            v = packed_weight_value(wm, W, p, g, o, i);
        }
        jb.dst[loc] = v;
    } else if (jb.mode <= 3) {   // thin layouts [tap][s][c]
        const int kk = wm.k * wm.k;
        const int Cs = jb.mode == 2 ? wm.I : wm.O, Cw = jb.mode == 2 ? wm.O : wm.I;
        const int c = (int)(loc % Cw);
        const int t = (int)(loc / Cw);
        const int sidx = t % Cs, tap = t / Cs;
        const int o = jb.mode == 2 ? c : sidx, i = jb.mode == 2 ? sidx : c;
        jb.dst[loc] = pk_load<ADAM>(params, ad, ak, w0 + ((long long)o * wm.I + i) * kk + tap);
    } else {                     // bias: packed[hw*C + c] = ref[c*HW + hw]
        const int c = (int)(loc % wm.o_c), hw = (int)(loc / wm.o_c);
        jb.dst[loc] = pk_load<ADAM>(params, ad, ak, w0 + (long long)c * wm.o_hw + hw);
    }
}
int fg_launch_pack_jobs(fg_ctx* ctx, const PackJob* jobs_dev, int njobs, long long total, const float* params, double prof_bytes,
                        long long lds_floats) {
    if (total == 0 || njobs == 0) return FG_OK;
    AdamArgs none = AdamArgs();
    {
        FgProfScope prof(ctx, fg_intern(ctx, "pack_jobs_kernel"), 0.0, 0.0, prof_bytes);
        hipLaunchKernelGGL(pack_jobs_kernel<false>, dim3(fg_cdiv(total, 256)), dim3(256), (size_t)lds_floats * sizeof(float), ctx->stream,
                           jobs_dev, njobs, total, params, none, AdamScalars{0.f, 0.f, 0.f});
    }
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_adam_pack_jobs(fg_ctx* ctx, const PackJob* jobs_dev, int njobs, long long total, const AdamArgs& a, long long lds_floats) {
    if (total == 0 || njobs == 0) return FG_OK;
    hipLaunchKernelGGL(pack_jobs_kernel<true>, dim3(fg_cdiv(total, 256)), dim3(256), (size_t)lds_floats * sizeof(float), ctx->stream, jobs_dev, njobs, total,
                       (const float*)a.p, a, fg_adam_scalars(a));
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

__global__ void wgrad_finish_kernel(const WeightMap wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                    float beta, float* __restrict__ gradW);

// one element of the reduction (shared by the per-layer launch and the batched one: same order of additions)
__device__ __forceinline__ void wgrad_finish_one(const WeightMap& wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                                 float beta, float* __restrict__ gradW, int pi, int po, int wi) {
    if (pi >= wm.I || po >= wm.O) return;
    int o = po, i = pi;
    if (wm.o_hw > 1) { int hw = po / wm.o_c, c = po - hw * wm.o_c; o = c * wm.o_hw + hw; }
    if (wm.i_hw > 1) { int hw = pi / wm.i_c, c = pi - hw * wm.i_c; i = c * wm.i_hw + hw; }
    const size_t tile = (size_t)Npad * Cpad;
    const size_t e = (size_t)po * Cpad + pi;
    const int dy = wi / wm.k, dx = wi - dy * wm.k;
    float sum = 0.f;
    // the S partials of one (parity, tap): loads issued four at a time, ADDED in the same order as ever (round 4: as one load per
    // iteration every thread walked 7 ... 51 dependent memory latencies -- 44 us per launch for ~90 MB out of L2)
    auto run = [&](const float* __restrict__ b) {
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            const float v0 = b[(size_t)s * tile], v1 = b[(size_t)(s + 1) * tile], v2 = b[(size_t)(s + 2) * tile], v3 = b[(size_t)(s + 3) * tile];
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; s < S; ++s) sum += b[(size_t)s * tile];
    };
    if (wm.kind == 0) {
        run(Part + (size_t)wi * S * tile + e);
    } else {
        for (int p = 0; p < 4; ++p) {
            const int ty = dev_fold_r(p >> 1, dy, wm.pad) - wm.rmin;
            const int tx = dev_fold_r(p & 1, dx, wm.pad) - wm.rmin;
            const int pg = p * wm.G + ty * wm.T + tx;
            run(Part + (size_t)pg * S * tile + e);
        }
    }
    float* gw = gradW + ((size_t)o * wm.I + i) * wm.k * wm.k + wi;
    *gw = (beta == 0.f) ? sum : beta * (*gw) + sum;
}
// The same reduction for a plain k x k layer (k = 3 | 5), one block per (out-channel, 128 in-channels) and ALL taps: a thread keeps one
// running sum per tap and walks the splits two at a time -- 2 k^2 independent loads in flight (one thread per (channel pair, tap)
// walked its S = 7 ... 51 partials four at a time: the launch was bound by memory latency x occupancy, 39 us for D's ~85 MB).  Every
// tap's additions stay in ascending split order (bit-identical to wgrad_finish_one); the block then writes the [in-channel][tap] run
// of the reference layout contiguously through LDS instead of one word per 36- / 100-byte stride.
__host__ __device__ static inline bool fg_finish_all_taps(const WeightMap& wm) { return !wm.wino && wm.kind == 0 && (wm.k == 3 || wm.k == 5); }
// in-channels per block: 128 (one thread each), or -- long reductions, S >= 12 -- 32 with the splits dealt to FOUR thread groups
// (group q sums the contiguous range [q * ceil(S / 4), ...) in ascending order, the four group sums are added in group order): D's
// 128 -> 256 layer has 27 splits x 9 taps = 243 partials per channel pair and only 256 x 128 pairs -- one thread per pair left 256
// blocks walking 14 dependent load batches each while the rest of the launch had long finished (35 us for 83 MB)
__host__ __device__ static inline int fg_finish_taps_tpb(int S) { return S >= 12 ? 32 : 128; }
template <int K>
__device__ __forceinline__ void wgrad_finish_taps(const WeightMap& wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                                  float beta, float* __restrict__ gradW, int bx, int po, float* sh) {
    constexpr int kk = K * K;
    const int tpb = fg_finish_taps_tpb(S), ngrp = 128 / tpb;
    const int il = (int)threadIdx.x % tpb, q = (int)threadIdx.x / tpb;
    const int pi = bx * tpb + il;
    const size_t tile = (size_t)Npad * Cpad;
    const int per = (S + ngrp - 1) / ngrp, s0 = q * per, s1 = min(S, s0 + per);
    float sum[kk];
#pragma unroll
    for (int wi = 0; wi < kk; ++wi) sum[wi] = 0.f;
    if (pi < wm.I && po < wm.O) {
        const float* __restrict__ b = Part + (size_t)po * Cpad + pi;       // tap wi, split s at b[(wi * S + s) * tile]
        int s = s0;
        for (; s + 2 <= s1; s += 2) {
            float v0[kk], v1[kk];
#pragma unroll
            for (int wi = 0; wi < kk; ++wi) {
                v0[wi] = b[((size_t)wi * S + s) * tile];
                v1[wi] = b[((size_t)wi * S + s + 1) * tile];
            }
#pragma unroll
            for (int wi = 0; wi < kk; ++wi) { sum[wi] += v0[wi]; sum[wi] += v1[wi]; }
        }
        if (s < s1) {
#pragma unroll
            for (int wi = 0; wi < kk; ++wi) sum[wi] += b[((size_t)wi * S + s) * tile];
        }
    }
    // sh: [group][in-channel][tap] -- 128 * kk floats in either shape
#pragma unroll
    for (int wi = 0; wi < kk; ++wi) sh[(q * tpb + il) * kk + wi] = sum[wi];
    __syncthreads();
    if (po >= wm.O) return;
    const int nI = min(tpb, wm.I - bx * tpb);
    float* __restrict__ base = gradW + ((size_t)po * wm.I + (size_t)bx * tpb) * kk;
    for (int idx = (int)threadIdx.x; idx < nI * kk; idx += 128) {
        float t = sh[idx];
        for (int g = 1; g < ngrp; ++g) t += sh[g * tpb * kk + idx];
        base[idx] = (beta == 0.f) ? t : beta * base[idx] + t;
    }
}
__device__ __forceinline__ void wgrad_finish_taps_any(const WeightMap& wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                                      float beta, float* __restrict__ gradW, int bx, int po, float* sh) {
    if (wm.k == 5) wgrad_finish_taps<5>(wm, Part, S, Npad, Cpad, beta, gradW, bx, po, sh);
    else wgrad_finish_taps<3>(wm, Part, S, Npad, Cpad, beta, gradW, bx, po, sh);
}
// Winograd-domain partials (WeightMap::wino, wino_wgrad.hip): Part[unit][split][Npad][Cpad][pos 16].  One thread per (out, in) pair:
// sum the splits per position, apply the signs wino_wgrad_kernel left out of A's last row (s_i s_j, s = (1, 1, 1, -1)),
// t = G^T (dL/dU) G, and scatter the sub-kernel gradient into the reference taps -- the exact adjoint of fg_wino_subkernel:
//   wino 1, kind 0 (3x3): taps = t;   wino 1, kind 1 (folded): tap (dy, dx) collects t_p[fold(py, dy)][fold(px, dx)] of every parity p;
//   wino 2 (5x5): tap (3a + dy, 3b + dx) = t_(a, b)[dy][dx].
// Threads: a block of 128 owns one out-channel and 128 in-channels of a one-unit layer, or 32 in-channels x the 4 units (parities /
// groups) of a folded / 5x5 layer (thread = (unit, in-channel): the 16 position sums of a split are 16 independent loads in flight,
// the units' tap contributions meet in LDS).
#define FG_WINO_FINISH_LDS (25 * 128)
__host__ __device__ static inline int fg_wino_finish_tpu(const WeightMap& wm) { return (wm.kind == 1 || wm.wino == 2) ? 32 : 128; }
template <int K>
__device__ __forceinline__ void wino_wgrad_finish_block(const WeightMap& wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                                        float beta, float* __restrict__ gradW, int bx, int o, float* sh) {
    const int units = (wm.kind == 1 ? 4 : 1) * (wm.wino == 2 ? 4 : 1);
    const int tpu = fg_wino_finish_tpu(wm);
    const int tid = threadIdx.x, u = units == 1 ? 0 : tid / tpu, il = units == 1 ? tid : tid - u * tpu;
    const int i = bx * tpu + il;
    const bool live = i < wm.I && o < wm.O;
    const size_t tile = (size_t)Npad * Cpad;
    float g[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) g[q] = 0.f;
    if (live) {
        float du[16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) du[pos] = 0.f;
        const float4* __restrict__ b = (const float4*)(Part + ((size_t)u * S * tile + (size_t)o * Cpad + i) * 16);
        int s = 0;
        for (; s + 4 <= S; s += 4) {          // four splits = sixteen 16-byte loads in flight; the additions stay in split order
            float4 v[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[e][q] = b[(size_t)(s + e) * tile * 4 + q];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) { du[4 * q] += v[e][q].x; du[4 * q + 1] += v[e][q].y; du[4 * q + 2] += v[e][q].z; du[4 * q + 3] += v[e][q].w; }
        }
        for (; s < S; ++s) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = b[(size_t)s * tile * 4 + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) { du[4 * q] += v[q].x; du[4 * q + 1] += v[q].y; du[4 * q + 2] += v[q].z; du[4 * q + 3] += v[q].w; }
        }
#pragma unroll
        for (int pos = 0; pos < 16; ++pos)
            if (((pos >> 2) == 3) != ((pos & 3) == 3)) du[pos] = -du[pos];
        float r[4][3], t[9];      // G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const float d0 = du[ii * 4], d1 = du[ii * 4 + 1], d2 = du[ii * 4 + 2], d3 = du[ii * 4 + 3];
            r[ii][0] = d0 + 0.5f * (d1 + d2);
            r[ii][1] = 0.5f * (d1 - d2);
            r[ii][2] = 0.5f * (d1 + d2) + d3;
        }
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            t[0 * 3 + bb] = r[0][bb] + 0.5f * (r[1][bb] + r[2][bb]);
            t[1 * 3 + bb] = 0.5f * (r[1][bb] - r[2][bb]);
            t[2 * 3 + bb] = 0.5f * (r[1][bb] + r[2][bb]) + r[3][bb];
        }
        if (K == 5 && wm.wino == 2) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int q = 0; q < K * K; ++q)      // (select chains, not indexed accesses: g and t stay in registers)
                        g[q] = (q == (3 * (u >> 1) + dy) * K + 3 * (u & 1) + dx && 3 * (u >> 1) + dy < K && 3 * (u & 1) + dx < K) ? t[dy * 3 + dx] : g[q];
        } else if (wm.kind == 1) {
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const int ty = dev_fold_r(u >> 1, dy, wm.pad) - wm.rmin, tx = dev_fold_r(u & 1, dx, wm.pad) - wm.rmin;
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 9; ++q) v = (ty * 3 + tx == q) ? t[q] : v;
                    g[dy * K + dx] = v;
                }
        } else if (K == 3) {
#pragma unroll
            for (int q = 0; q < 9; ++q) g[q] = t[q];
        }
    }
    if (units > 1) {        // the units' contributions to one (out, in) pair: summed in unit order
#pragma unroll
        for (int q = 0; q < K * K; ++q) sh[q * 128 + tid] = g[q];
        __syncthreads();
        if (u != 0 || !live) return;
#pragma unroll
        for (int q = 0; q < K * K; ++q) g[q] = ((sh[q * 128 + il] + sh[q * 128 + tpu + il]) + sh[q * 128 + 2 * tpu + il]) + sh[q * 128 + 3 * tpu + il];
    } else if (!live) return;
    float* gw = gradW + ((size_t)o * wm.I + i) * (K * K);
#pragma unroll
    for (int q = 0; q < K * K; ++q) gw[q] = (beta == 0.f) ? g[q] : beta * gw[q] + g[q];
}
__device__ __forceinline__ void wino_wgrad_finish_any(const WeightMap& wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                                      float beta, float* __restrict__ gradW, int bx, int o, float* sh) {
    if (wm.k == 5) wino_wgrad_finish_block<5>(wm, Part, S, Npad, Cpad, beta, gradW, bx, o, sh);
    else wino_wgrad_finish_block<3>(wm, Part, S, Npad, Cpad, beta, gradW, bx, o, sh);
}
// all weight-gradient reductions of a backward pass in one launch: block -> job by a scan over <= FG_DEFER_WMAX entries; inside
// a job the blocks run (in-channel block, out-channel, tap) exactly as the grid of wgrad_finish_kernel does
struct FgWFinishBatch { FgWFinishJob jobs[FG_DEFER_WMAX]; int n; };
__global__ __launch_bounds__(128) void wgrad_finish_jobs_kernel(const FgWFinishBatch b) {
    __shared__ float tl[32][33];
    __shared__ float wsh[FG_WINO_FINISH_LDS];
    int j = 0;
    while (j + 1 < b.n && (long long)blockIdx.x >= b.jobs[j + 1].blk0) ++j;
    const FgWFinishJob& jb = b.jobs[j];
    long long l = (long long)blockIdx.x - jb.blk0;
    if (jb.ib < 0) {
        // Linear behind a View (k = 1, in-features permuted: partial column pi = hw*C + c <-> reference column i = c*HW + hw): a block
        // owns one out-feature and a 32 (c) x 32 (hw) patch -- partials read along c, the gradient written along hw, both in 128-byte
        // runs through LDS (one thread per element wrote the 134 MB gradient of models_c2f.lua:262 one word per cache line)
        const WeightMap& wm = jb.wm;
        const int C = wm.i_c, HW = wm.i_hw, nct = (C + 31) / 32, nht = (HW + 31) / 32;
        const int ht = (int)(l % nht); l /= nht;
        const int ct = (int)(l % nct);
        const int po = (int)(l / nct);
        int o = po;
        if (wm.o_hw > 1) { int hw = po / wm.o_c, c = po - hw * wm.o_c; o = c * wm.o_hw + hw; }
        const size_t tile = (size_t)jb.Npad * jb.Cpad;
        const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
        for (int r = ly; r < 32; r += 4) {                     // row r = hw, lanes = c
            const int hw = ht * 32 + r, c = ct * 32 + lx;
            float sum = 0.f;
            if (hw < HW && c < C) {
                const size_t e = (size_t)po * jb.Cpad + (size_t)hw * C + c;
                const float* __restrict__ b = jb.part + e;
                int s = 0;
                for (; s + 4 <= jb.S; s += 4) {                // four loads in flight, the additions in the same order
                    const float v0 = b[(size_t)s * tile], v1 = b[(size_t)(s + 1) * tile], v2 = b[(size_t)(s + 2) * tile], v3 = b[(size_t)(s + 3) * tile];
                    sum += v0; sum += v1; sum += v2; sum += v3;
                }
                for (; s < jb.S; ++s) sum += b[(size_t)s * tile];
            }
            tl[r][lx] = sum;
        }
        __syncthreads();
        for (int r = ly; r < 32; r += 4) {                     // row r = c, lanes = hw
            const int c = ct * 32 + r, hw = ht * 32 + lx;
            if (hw < HW && c < C) {
                float* gw = jb.gradW + (size_t)o * wm.I + (size_t)c * HW + hw;
                const float sum = tl[lx][r];
                *gw = (jb.beta == 0.f) ? sum : jb.beta * (*gw) + sum;
            }
        }
        return;
    }
    const int bx = (int)(l % jb.ib); l /= jb.ib;
    const int po = (int)(l % jb.wm.O), wi = (int)(l / jb.wm.O);
    if (jb.wm.wino) { wino_wgrad_finish_any(jb.wm, jb.part, jb.S, jb.Npad, jb.Cpad, jb.beta, jb.gradW, bx, po, wsh); return; }
    if (fg_finish_all_taps(jb.wm)) { wgrad_finish_taps_any(jb.wm, jb.part, jb.S, jb.Npad, jb.Cpad, jb.beta, jb.gradW, bx, po, wsh); return; }
    wgrad_finish_one(jb.wm, jb.part, jb.S, jb.Npad, jb.Cpad, jb.beta, jb.gradW, bx * 128 + (int)threadIdx.x, po, wi);
}
int fg_launch_wgrad_finish_jobs(fg_ctx* ctx, const FgWFinishJob* jobs, int n, long long blocks) {
    if (n == 0) return FG_OK;
    FgWFinishBatch b;
    for (int i = 0; i < n; ++i) b.jobs[i] = jobs[i];
    b.n = n;
    hipLaunchKernelGGL(wgrad_finish_jobs_kernel, dim3((unsigned)blocks), dim3(128), 0, ctx->stream, b);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
bool fg_defer_push_wfinish(fg_ctx* ctx, const WeightMap& wm, const float* Part, int S, int Npad, int Cpad, float beta, float* gradW) {
    FgDefer* d = ctx->defer;
    if (!d || !d->wjobs || d->wn >= FG_DEFER_WMAX) return false;
    const bool brick = wm.kind == 0 && wm.k == 1 && wm.i_hw > 1 && wm.i_c > 0;     // Linear behind a View: 32 x 32 patches (ib = -1)
    const long long nb = brick ? (long long)wm.O * fg_cdiv(wm.i_c, 32) * fg_cdiv(wm.i_hw, 32)
                               : (wm.wino ? (long long)fg_cdiv(wm.I, fg_wino_finish_tpu(wm)) * wm.O       // (Winograd partials: wino_wgrad_finish_block)
                                          : (fg_finish_all_taps(wm) ? (long long)fg_cdiv(wm.I, fg_finish_taps_tpb(S)) * wm.O
                                                                    : (long long)fg_cdiv(wm.I, 128) * wm.O * wm.k * wm.k));
    if (d->wblocks + nb > 0x7fffffffLL) return false;
    FgWFinishJob& j = d->wjobs[d->wn++];
    j.wm = wm; j.part = Part; j.gradW = gradW; j.S = S; j.Npad = Npad; j.Cpad = Cpad; j.ib = brick ? -1 : fg_cdiv(wm.I, wm.wino ? fg_wino_finish_tpu(wm) : (fg_finish_all_taps(wm) ? fg_finish_taps_tpb(S) : 128)); j.beta = beta;
    j.blk0 = d->wblocks;
    d->wblocks += nb;
    return true;
}

__global__ void wgrad_finish_kernel(const WeightMap wm, const float* __restrict__ Part, int S, int Npad, int Cpad,
                                    float beta, float* __restrict__ gradW) {
    // x: packed in-channel (coalesced partial reads), y: packed out-channel, z: tap dy*k+dx
    __shared__ float wsh[FG_WINO_FINISH_LDS];
    if (wm.wino) { wino_wgrad_finish_any(wm, Part, S, Npad, Cpad, beta, gradW, blockIdx.x, blockIdx.y, wsh); return; }
    if (fg_finish_all_taps(wm)) { wgrad_finish_taps_any(wm, Part, S, Npad, Cpad, beta, gradW, blockIdx.x, blockIdx.y, wsh); return; }
    wgrad_finish_one(wm, Part, S, Npad, Cpad, beta, gradW, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, blockIdx.z);
}
int fg_launch_wgrad_finish(fg_ctx* ctx, const WeightMap& wm, const float* Part, int S, int Npad, int Cpad, float beta,
                           float* gradW) {
    if (wm.wino && wm.k != 3 && wm.k != 5) return fg_set_err(ctx, FG_ERR_INVALID, "winograd weight-gradient finish: k = %d", wm.k);
    dim3 grid(fg_cdiv(wm.I, wm.wino ? fg_wino_finish_tpu(wm) : (fg_finish_all_taps(wm) ? fg_finish_taps_tpb(S) : 128)), wm.O,
              (wm.wino || fg_finish_all_taps(wm)) ? 1 : wm.k * wm.k);
    hipLaunchKernelGGL(wgrad_finish_kernel, grid, dim3(128), 0, ctx->stream, wm, Part, S, Npad, Cpad, beta, gradW);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised fp32 weight gradient (round 3): the block structure of igemm_ws_kernel on the PIXEL reduction -- one
// 512-thread block per CU, waves 0-3 only multiply (one per SIMD, 128 x 64 channel pairs each = 8 accumulator tiles), waves 4-7
// only move data (plain 16-byte loads of pixel rows, no transposes) three K-steps ahead into a 3-stage LDS ring; K-step = 16
// pixels, LDS image pixel-major [pixel][channel] exactly as in HBM.
// What makes it different from the two wave-specialised attempts of round 2: the MFMA fragments are read with ONE
// ds_read_b128 (row operand) and ONE ds_read_b64 (column operand) per pixel pair and eight MFMAs.  v_mfma_f32_32x32x2_f32 takes
// A[i][k] from lane 32 k + i: with lane (i, k) reading the 16 bytes at [pixel 2 kp + k][channel 4 i .. 4 i + 3], register j of
// the read IS an A fragment whose 32 rows are the channels 4 i + j -- the channel <-> row assignment of an accumulator tile is
// free, it only has to be undone when the tile is stored.  No per-MFMA ds_read_b32, no channel-major staging.
// Block tile RT (row operand) x QT (column operand) = 256 x 128 channels; SWAP = 0: rows = dY channels, columns = X channels;
// SWAP = 1: rows = X channels, columns = dY channels (layers whose input has the 256).  Part stays [pg][s][dY ch][X ch].
// ---------------------------------------------------------------------------------------------------------------
// RT x QT = the block's channel tile, KS = pixels per K-step, WK = 1: the four MFMA waves split the TILE 2 x 2 (256 x 128, KS = 16);
// WK = 4 (round 3, the 64-channel layers: RT x QT = 128 x 64, KS = 64): every MFMA wave multiplies the WHOLE tile (the same 4 x 2
// accumulator tiles, the same b128 / b64 fragment reads) over its own quarter of the K-step's pixels, and the four partial tiles
// are summed through LDS at the end in a fixed order -- a small tile without the two LDS reads per MFMA of wgrad_kernel<64>.
template <int SWAP, int RT = 256, int QT = 128, int KS = 16, int WK = 1>
__global__ __launch_bounds__(512, 2) void wgrad_ws_kernel(const WgradArgs a) {
    constexpr int MI = 4, NI = 2;
    static_assert((WK == 1 && RT == 256 && QT == 128 && KS == 16) || (WK == 4 && RT == 128 && QT == 64 && KS == 64), "tile / wave split");
    constexpr int LGKS = KS == 16 ? 4 : 6;
    constexpr int STAGE = KS * (RT + QT);                       // floats per ring stage
    extern __shared__ __attribute__((aligned(16))) float smemw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nd_t = SWAP ? a.Npad / QT : a.Npad / RT;          // tiles along dY channels
    const int nx_t = SWAP ? a.Cpad / RT : a.Cpad / QT;          // tiles along X channels
    int bt_, s, pg;
    fg_wgrad_block(bt_, s, pg);
    const int td = bt_ / nx_t, tx = bt_ - td * nx_t;
    (void)nd_t;
    const int p = pg / a.G, g = pg - p * a.G;
    const int m0 = s * a.m_per_split;
    const int m1 = min(a.M, m0 + a.m_per_split);
    const int KT = (m1 > m0) ? (m1 - m0) >> LGKS : 0;      // whole KS-pixel steps only (see the launcher)

    if (wid >= 4) {
        // ------------------------------------------------------------------ loader waves (256 threads)
        const int lt = tid - 256;
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)a.d_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
        const int doy = a.doy[p], dox = a.dox[p], xoy = a.xoy[p][g], xox = a.xox[p][g];
        // dY tile: DT channels, X tile: XT channels; each loader lane owns one 16-byte chunk of 4 (resp. 2) pixel rows
        constexpr int DT = SWAP ? QT : RT, XT = SWAP ? RT : QT;
        constexpr int ND = KS * DT / 4 / 256, NX = KS * XT / 4 / 256;       // chunks per lane and K-step
        constexpr int DPP = 256 / (DT / 4), XPP = 256 / (XT / 4);           // pixel rows covered by one pass of 256 lanes
        const int dpix = lt / (DT / 4), dch = (lt - dpix * (DT / 4)) * 4;
        const int xpix = lt / (XT / 4), xch = (lt - xpix * (XT / 4)) * 4;
        const int d_lds0 = (SWAP ? KS * RT : 0) + dpix * DT + dch;          // the row operand's image comes first
        const int x_lds0 = (SWAP ? 0 : KS * RT) + xpix * XT + xch;
        const int gD = (td * DT + dch) * 4, gX = (tx * XT + xch) * 4;
        const int dpixB = a.Nd * 4, xpixB = a.Cx * 4;
        // Address math (the launcher only selects this kernel when Hm, Wm are powers of two, 16 | Hm*Wm and 16 | m_per_split, so a
        // K-step is 16 consecutive pixels of ONE sample -- part of a row, or 16 / Wm whole rows -- and no step is partial): a row's
        // offsets are (wave-uniform base of the step, SALU) + (per-lane constant), one add per dY load, two adds + two compares
        // per X load.  Every non-MFMA instruction issued on a SIMD costs its matrix pipe 6-9 idle cycles (PMC passes of round 3,
        // profiles/r03_pmc_wgrad.md): the loader waves are written for instruction COUNT.
        int cD[ND], cX[NX], cyy[NX], cxx[NX];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int j = dpix + DPP * i, yr = j >> a.lgW, xr = j & (a.Wm - 1);
            cD[i] = (yr * a.dsy * a.Wd + xr * a.dsx) * dpixB + gD;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int j = xpix + XPP * i, yr = j >> a.lgW, xr = j & (a.Wm - 1);
            cX[i] = (yr * a.xsy * a.Wx + xr * a.xsx) * xpixB + gX;
            cyy[i] = yr * a.xsy + xoy;
            cxx[i] = xr * a.xsx + xox;
        }
        // bias gradient = per-channel sums of dY over all pixels.  Every (tap, X tile) block of a (parity, split, dY tile) streams
        // the SAME dY rows, so the nb = G * nx_t of them share the work: block `mine` sums the K-steps kt = mine (mod nb) and leaves
        // one partial row per loader pixel lane, bias_part [P][S][nb][DPP][Nd], finished by the batched final.  (Letting the tap-0
        // blocks sum everything made THEM 18 % slower -- packed fp32 adds beside the MFMAs -- and a launch of one round of blocks
        // lasts as long as its slowest block: 6.66 -> 7.44 ms on the c2f layers.)
        const int nb = a.G * nx_t, mine = g * nx_t + tx;
        const int bperiod = a.bias_part ? nb - 1 : -1;           // -1: never
        int bcount = a.bias_part ? mine : -1;
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
        int mcur = m0;
        // wave-uniform strides (bytes), hoisted: the per-step base is three multiply-adds per operand on the scalar unit
        const int sDn = a.Hd * a.Wd * dpixB, sDy = a.dsy * a.Wd * dpixB, sDx = a.dsx * dpixB, sD0 = (doy * a.Wd + dox) * dpixB;
        const int sXn = a.Hx * a.Wx * xpixB, sXy = a.xsy * a.Wx * xpixB, sXx = a.xsx * xpixB, sX0 = (xoy * a.Wx + xox) * xpixB;
        const int lgW = a.lgW, lgH = a.lgH, wmask = a.Wm - 1, hmask = a.Hm - 1, xsy = a.xsy, xsx = a.xsx;
        const unsigned Hx = (unsigned)a.Hx, Wx = (unsigned)a.Wx;
        f32x4 xd[ND], xx[NX], yd[ND], yx[NX], zd[ND], zx[NX];
#define WW_LOAD(rd_, rx_)                                                                                \
        {                                                                                                \
            const int rowi = mcur >> lgW, x0 = mcur & wmask;                                             \
            const int yb = rowi & hmask, nb = rowi >> lgH;                                               \
            const int bD = nb * sDn + yb * sDy + x0 * sDx + sD0;                                         \
            const int bX = nb * sXn + yb * sXy + x0 * sXx + sX0;                                         \
            const int ybs = yb * xsy, xbs = x0 * xsx;                                                    \
            _Pragma("unroll") for (int i = 0; i < ND; ++i) rd_[i] = fg_buffer_load4(drsrc, bD + cD[i]);  \
            _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                             \
                const bool ok = (unsigned)(ybs + cyy[i]) < Hx && (unsigned)(xbs + cxx[i]) < Wx;          \
                rx_[i] = fg_buffer_load4(xrsrc, ok ? bX + cX[i] : FG_OOB);                               \
            }                                                                                            \
            mcur += KS;                                                                                  \
        }
#define WW_STORE(st, rd_, rx_)                                                                           \
        {                                                                                                \
            float* S = smemw + (st) * STAGE;                                                             \
            if (bcount == 0) { _Pragma("unroll") for (int i = 0; i < ND; ++i) bsum += rd_[i]; }          \
            bcount = bcount <= 0 ? bperiod : bcount - 1;                                                 \
            _Pragma("unroll") for (int i = 0; i < ND; ++i) *(f32x4*)(S + d_lds0 + DPP * i * DT) = rd_[i]; \
            _Pragma("unroll") for (int i = 0; i < NX; ++i) *(f32x4*)(S + x_lds0 + XPP * i * XT) = rx_[i]; \
        }
        if (KT > 0) { WW_LOAD(xd, xx); WW_STORE(0, xd, xx); }
        if (KT > 1) { WW_LOAD(xd, xx); WW_STORE(1, xd, xx); }
        if (KT > 2) { WW_LOAD(xd, xx); }
        if (KT > 3) { WW_LOAD(yd, yx); }
        if (KT > 4) { WW_LOAD(zd, zx); }
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 3) {
            if (kt + 2 < KT) {
                WW_STORE(2, xd, xx);
                if (kt + 5 < KT) { WW_LOAD(xd, xx); }
            }
            __syncthreads();
            if (kt + 1 < KT) {
                if (kt + 3 < KT) {
                    WW_STORE(0, yd, yx);
                    if (kt + 6 < KT) { WW_LOAD(yd, yx); }
                }
                __syncthreads();
            }
            if (kt + 2 < KT) {
                if (kt + 4 < KT) {
                    WW_STORE(1, zd, zx);
                    if (kt + 7 < KT) { WW_LOAD(zd, zx); }
                }
                __syncthreads();
            }
        }
#undef WW_LOAD
#undef WW_STORE
        if (a.bias_part)
            *(f32x4*)(a.bias_part + ((((size_t)p * a.S + s) * nb + mine) * DPP + dpix) * a.Nd + td * DT + dch) = bsum;
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves (one per SIMD)
    const int wm = WK == 1 ? wid >> 1 : 0, wn = WK == 1 ? wid & 1 : 0;      // WK = 1: 2 x 2 waves, 128 row x 64 column channels each
    const int wk = WK == 1 ? 0 : wid;                           // WK = 4: the wave's quarter of the K-step's pixels
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    // lane (i = lane & 31, k = lane >> 5): rows = channels wm*128 + 4 i + j (j = register of the b128), columns = wn*64 + 2 n + j'
    const int r_off = (16 * wk + (lane >> 5)) * RT + wm * 128 + (lane & 31) * 4;
    const int q_off = KS * RT + (16 * wk + (lane >> 5)) * QT + wn * 64 + (lane & 31) * 2;
    f32x4 af[8];
    f32x2 bf[8];
    __syncthreads();                                            // tiles 0 and 1 are in the ring
    if (KT > 0) {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
            af[kp] = *(const f32x4*)(smemw + r_off + 2 * kp * RT);
            bf[kp] = *(const f32x2*)(smemw + q_off + 2 * kp * QT);
        }
    }
    int sn = 1;
    for (int kt = 0; kt < KT; ++kt) {
        const float* Sn = smemw + sn * STAGE;
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kp][mi], bf[kp][ni], acc[mi][ni], 0, 0, 0);
            // this pair's registers are free: fetch the same pair of the next tile (already in the ring; after the last tile the
            // read returns a stale stage nobody uses -- unconditional, a branch per pair is an instruction the matrix pipe pays for)
            af[kp] = *(const f32x4*)(Sn + r_off + 2 * kp * RT);
            bf[kp] = *(const f32x2*)(Sn + q_off + 2 * kp * QT);
        }
        sn = sn == 2 ? 0 : sn + 1;
        asm volatile("s_barrier" ::: "memory");                 // bare: the fragment reads above may still be in flight
    }

    float* part = a.Part + ((size_t)pg * a.S + s) * a.Npad * a.Cpad;
    if constexpr (WK == 4) {
        // four partial 128 x 64 tiles -> LDS [wave][mi][ni][r][lane] (128 KB of the ring; every wave is past the last barrier of the
        // K loop, the loader waves are gone), summed (w0 + w1) + (w2 + w3); wave w finishes the row tiles mi = w
        float* red = smemw;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wid * MI + mi) * NI + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
        __syncthreads();
        constexpr int WS = MI * NI * 16 * 64;                   // floats per wave
        const int mi = wid;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int cq = 2 * (lane & 31) + ni;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = ((mi * NI + ni) * 16 + r) * 64 + lane;
                const float v = (red[e] + red[e + WS]) + (red[e + 2 * WS] + red[e + 3 * WS]);
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int cr = 4 * i + mi;
                const int o = SWAP ? td * QT + cq : td * RT + cr;  // dY channel
                const int c = SWAP ? tx * RT + cr : tx * QT + cq;  // X channel
                part[(size_t)o * a.Cpad + c] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int cq = wn * 64 + 2 * (lane & 31) + ni;      // column-operand channel inside the block tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int cr = wm * 128 + 4 * i + mi;             // row-operand channel inside the block tile
                const int o = SWAP ? td * QT + cq : td * RT + cr;  // dY channel
                const int c = SWAP ? tx * RT + cr : tx * QT + cq;  // X channel
                part[(size_t)o * a.Cpad + c] = acc[mi][ni][r];
            }
        }
}

// the kernel's address math assumes whole 16-pixel steps inside one sample
bool fg_wgrad_ws_shape_ok(const WgradArgs& a, int cfg) {
    const int ks = cfg == 2 ? 63 : 15;
    return a.lgW >= 0 && a.lgH >= 0 && ((a.Hm * a.Wm) & ks) == 0 && (a.m_per_split & ks) == 0 && (a.M & ks) == 0;
}
// partial rows per (parity, split) the kernel leaves in bias_part: (taps x X tiles) blocks x the pixel rows one pass of the 256
// loader lanes covers
int fg_wgrad_ws_bias_rows(const WgradArgs& a, int cfg) { return a.G * (a.Cpad / (cfg == 0 ? 128 : (cfg == 1 ? 256 : 64))) * (cfg == 0 ? 4 : 8); }
// cfg 0: 256 dY channels x 128 X channels per block, cfg 1: 128 x 256, cfg 2: 128 x 64 with the K-step split over the MFMA waves
int fg_launch_wgrad_ws(fg_ctx* ctx, const WgradArgs& a, int P, int cfg) {
    const int RTd = cfg == 0 ? 256 : 128, QTx = cfg == 0 ? 128 : (cfg == 1 ? 256 : 64);
    const size_t lds = cfg == 2 ? (size_t)3 * 64 * (128 + 64) * sizeof(float) : (size_t)3 * 16 * (256 + 128) * sizeof(float);
    static char attr_key[3];
    if (fg_attr_first(ctx, &attr_key[cfg])) {
        if (cfg == 0) FG_HIP(ctx, hipFuncSetAttribute((const void*)wgrad_ws_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else if (cfg == 1) FG_HIP(ctx, hipFuncSetAttribute((const void*)wgrad_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else FG_HIP(ctx, hipFuncSetAttribute((const void*)wgrad_ws_kernel<0, 128, 64, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if ((a.Nd % RTd) || (a.Cx % QTx) || a.Npad != a.Nd || a.Cpad != a.Cx)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad_ws: %d x %d channels do not tile %d x %d", a.Nd, a.Cx, RTd, QTx);
    if (a.d_bytes <= 0 || a.x_bytes <= 0 || a.d_bytes >= (long long)FG_OOB || a.x_bytes >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad_ws: operands must be < 2 GiB per launch");
    if (!fg_wgrad_ws_shape_ok(a, cfg))
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad_ws: needs power-of-two Hm, Wm and whole K-steps (16 / 64 pixels) per sample and split");
    dim3 grid((a.Npad / RTd) * (a.Cpad / QTx), a.S, a.G * P);
    const double exec = 2.0 * (double)a.Npad * a.Cpad * (double)P * a.G * (double)a.M;
    char label[96];
    snprintf(label, sizeof(label), "wgrad_ws_kernel<%d>/%s", cfg, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    if (cfg == 0) hipLaunchKernelGGL(wgrad_ws_kernel<0>, grid, dim3(512), lds, ctx->stream, a);
    else if (cfg == 1) hipLaunchKernelGGL(wgrad_ws_kernel<1>, grid, dim3(512), lds, ctx->stream, a);
    else hipLaunchKernelGGL((wgrad_ws_kernel<0, 128, 64, 64, 4>), grid, dim3(512), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x6 weight-gradient contraction: Part[pg][s][o][c] = sum over the pixels of split s of dY[pixd][o] * X[pixx(g)][c],
// both operands as split-bf16 planes.  The reduction runs over PIXELS, so MFMA fragments need 8 consecutive pixels of
// one channel per lane while memory (and the LDS image) is pixel-major; the fragments are therefore read with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: within 16 lanes, lane c receives element c%4 of the 8-byte slots supplied
// by lanes c/4 + 4j), two reads per 8-pixel fragment.  Same wave-specialised block as igemm_ws6_kernel: 4 MFMA waves
// (2x2, each MI x NI 32x32 tiles) + 4 loader waves, K-step = 16 pixels, 3-stage LDS ring; pixel rows are padded so
// that (row stride mod 256) = 64, which keeps the 8-byte slots of a 32-lane half on distinct banks.
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x8 fg_tr_frag(const unsigned char* lo_addr, int hi_delta) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo_addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo_addr + hi_delta));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int MI, int NI>
__global__ __launch_bounds__(512, 2) void wgrad_ws6_kernel(const WgradArgs a) {
    constexpr int RT = MI * 64, QT = NI * 64;                 // dY channels x X channels per block
    constexpr int RCH = RT / 16 * 6, QCH = QT / 16 * 6;       // 16-byte chunks per pixel row
    constexpr int ROW_R = RT / 16 * 96 + 64, ROW_Q = QT / 16 * 96 + 64;
    constexpr int STAGE = 16 * (ROW_R + ROW_Q);
    constexpr int NR = (16 * RCH + 255) / 256, NQ = (16 * QCH + 255) / 256;   // chunks per loader thread and K-step (last may be partial)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntq = a.Cpad / QT;
    int bt_, s, pg;
    fg_wgrad_block(bt_, s, pg);
    const int tn = bt_ / ntq, tq = bt_ - tn * ntq;
    const int p = pg / a.G, g = pg - p * a.G;
    const int m0 = s * a.m_per_split;
    const int m1 = min(a.M, m0 + a.m_per_split);
    const int KT = (m1 > m0) ? (m1 - m0 + 15) / 16 : 0;

    if (wid >= 4) {
        const int lt = tid - 256;
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.D6, 0, (int)(a.d_bytes / 4 * 6), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.X6, 0, (int)(a.x_bytes / 4 * 6), 0x00020000);
        const int dpixB = a.Nd / 16 * 96, xpixB = a.Cx / 16 * 96;
        const int doy = a.doy[p], dox = a.dox[p], xoy = a.xoy[p][g], xox = a.xox[p][g];
        int pr[NR], ldsR[NR], gR[NR], pq[NQ], ldsQ[NQ], gQ[NQ];
        bool okr[NR], okq[NQ];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lt + 256 * i;
            okr[i] = c < 16 * RCH;
            pr[i] = okr[i] ? c / RCH : 0;
            const int w = okr[i] ? c - pr[i] * RCH : 0;
            ldsR[i] = pr[i] * ROW_R + w * 16;
            gR[i] = tn * (RT / 16) * 96 + w * 16;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = lt + 256 * i;
            okq[i] = c < 16 * QCH;
            pq[i] = okq[i] ? c / QCH : 0;
            const int w = okq[i] ? c - pq[i] * QCH : 0;
            ldsQ[i] = 16 * ROW_R + pq[i] * ROW_Q + w * 16;
            gQ[i] = tq * (QT / 16) * 96 + w * 16;
        }
        int mcur = m0;
        f32x4 xr[NR], xq[NQ], yr[NR], yq[NQ], zr[NR], zq[NQ];
#define G6_LOAD(rr, rq)                                                                                  \
        {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                             \
                const int m = mcur + pr[i];                                                              \
                int n, y, x;                                                                             \
                fg_decode_m(m < m1 ? m : m0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);                         \
                const int off = ((n * a.Hd + y * a.dsy + doy) * a.Wd + x * a.dsx + dox) * dpixB + gR[i]; \
                rr[i] = fg_buffer_load4(drsrc, m < m1 ? off : FG_OOB);                                   \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < NQ; ++i) {                                             \
                const int m = mcur + pq[i];                                                              \
                int n, y, x;                                                                             \
                fg_decode_m(m < m1 ? m : m0, a.lgH, a.lgW, a.Hm, a.Wm, n, y, x);                         \
                const int yy = y * a.xsy + xoy, xx = x * a.xsx + xox;                                    \
                const bool ok = m < m1 && (unsigned)yy < (unsigned)a.Hx && (unsigned)xx < (unsigned)a.Wx;\
                rq[i] = fg_buffer_load4(xrsrc, ok ? ((n * a.Hx + yy) * a.Wx + xx) * xpixB + gQ[i] : FG_OOB); \
            }                                                                                            \
            mcur += 16;                                                                                  \
        }
#define G6_STORE(st, rr, rq)                                                                             \
        {                                                                                                \
            unsigned char* S = smem6 + (st) * STAGE;                                                     \
            _Pragma("unroll") for (int i = 0; i < NR; ++i) if (okr[i]) *(f32x4*)(S + ldsR[i]) = rr[i];   \
            _Pragma("unroll") for (int i = 0; i < NQ; ++i) if (okq[i]) *(f32x4*)(S + ldsQ[i]) = rq[i];   \
        }
        if (KT > 0) { G6_LOAD(xr, xq); G6_STORE(0, xr, xq); }
        if (KT > 1) { G6_LOAD(xr, xq); G6_STORE(1, xr, xq); }
        if (KT > 2) { G6_LOAD(xr, xq); }
        if (KT > 3) { G6_LOAD(yr, yq); }
        if (KT > 4) { G6_LOAD(zr, zq); }
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 3) {
            if (kt + 2 < KT) {
                G6_STORE(2, xr, xq);
                if (kt + 5 < KT) { G6_LOAD(xr, xq); }
            }
            __syncthreads();
            if (kt + 1 < KT) {
                if (kt + 3 < KT) {
                    G6_STORE(0, yr, yq);
                    if (kt + 6 < KT) { G6_LOAD(yr, yq); }
                }
                __syncthreads();
            }
            if (kt + 2 < KT) {
                if (kt + 4 < KT) {
                    G6_STORE(1, zr, zq);
                    if (kt + 7 < KT) { G6_LOAD(zr, zq); }
                }
                __syncthreads();
            }
        }
#undef G6_LOAD
#undef G6_STORE
        return;
    }

    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kg = lane >> 5;
    const int r_off = (8 * kg + (i16 >> 2)) * ROW_R + (wm * MI * 2 + cb) * 96 + 8 * (i16 & 3);
    const int q_off = 16 * ROW_R + (8 * kg + (i16 >> 2)) * ROW_Q + (wn * NI * 2 + cb) * 96 + 8 * (i16 & 3);
    s16x8 A0[MI], A1[MI], A2[MI], B0[NI], B1[NI], B2[NI];
#define G6_LDA(dst, S, pl) _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) dst[mi] = fg_tr_frag((S) + r_off + mi * 192 + (pl) * 32, 4 * ROW_R);
#define G6_LDB(dst, S, pl) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) dst[ni] = fg_tr_frag((S) + q_off + ni * 192 + (pl) * 32, 4 * ROW_Q);
#define G6_PROD(Ax, Bx)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = fg_mfma_bf16(Ax[mi], Bx[ni], acc[mi][ni]);
    __syncthreads();
    if (KT > 0) {
        G6_LDA(A0, smem6, 0) G6_LDA(A1, smem6, 1) G6_LDA(A2, smem6, 2)
        G6_LDB(B0, smem6, 0) G6_LDB(B1, smem6, 1) G6_LDB(B2, smem6, 2)
    }
    int sn = 1;
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* Sn = smem6 + sn * STAGE;
        const bool nxt = kt + 1 < KT;
        G6_PROD(A1, B1)
        G6_PROD(A2, B0)  if (nxt) { G6_LDA(A2, Sn, 2) }
        G6_PROD(A0, B2)  if (nxt) { G6_LDB(B2, Sn, 2) }
        G6_PROD(A1, B0)  if (nxt) { G6_LDA(A1, Sn, 1) }
        G6_PROD(A0, B1)  if (nxt) { G6_LDB(B1, Sn, 1) }
        G6_PROD(A0, B0)  if (nxt) { G6_LDA(A0, Sn, 0) G6_LDB(B0, Sn, 0) }
        sn = sn == 2 ? 0 : sn + 1;
        asm volatile("s_barrier" ::: "memory");     // see igemm_ws6_kernel
    }
#undef G6_LDA
#undef G6_LDB
#undef G6_PROD

    float* part = a.Part + ((size_t)pg * a.S + s) * a.Npad * a.Cpad;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = tq * QT + wn * NI * 32 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tn * RT + wm * MI * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                part[(size_t)row * a.Cpad + col] = acc[mi][ni][r];
            }
        }
}

template <int MI, int NI>
static int launch_wgrad6_t(fg_ctx* ctx, const WgradArgs& a, int P) {
    constexpr int RT = MI * 64, QT = NI * 64;
    const size_t lds = (size_t)3 * 16 * ((RT / 16 * 96 + 64) + (QT / 16 * 96 + 64));
    static char attr_key;                      // one key per call site (and template instance); the flag lives in the context
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wgrad_ws6_kernel<MI, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if ((a.Nd % RT) || (a.Cx % QT) || a.Npad != a.Nd || a.Cpad != a.Cx)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad bf16x6: %d x %d channels do not tile %d x %d", a.Nd, a.Cx, RT, QT);
    if (a.d_bytes / 4 * 6 >= (long long)FG_OOB || a.x_bytes / 4 * 6 >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "wgrad bf16x6: operand planes must be < 2 GiB");
    dim3 grid((a.Npad / RT) * (a.Cpad / QT), a.S, a.G * P);
    const double exec = 2.0 * (double)a.Npad * a.Cpad * (double)a.M * a.G * P;
    char label[96];
    snprintf(label, sizeof(label), "wgrad_ws6_kernel<%d,%d>/%s", MI, NI, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    hipLaunchKernelGGL((wgrad_ws6_kernel<MI, NI>), grid, dim3(512), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
int fg_launch_wgrad6(fg_ctx* ctx, const WgradArgs& a, int P, int cfg) {
    return cfg == 0 ? launch_wgrad6_t<4, 2>(ctx, a, P)      // 256 dY-channels x 128 X-channels
                    : launch_wgrad6_t<2, 4>(ctx, a, P);     // 128 x 256
}
