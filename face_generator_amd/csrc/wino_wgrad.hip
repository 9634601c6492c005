// Winograd F(2x2, 3x3) WEIGHT gradient on the fp32 matrix pipe of gfx950 (MI355X): the accGradParameters of the layers whose
// forward / data gradient run in wino.hip -- the nearest-x2 + 5x5 up-convolutions of G (models.lua:63-64, 68-69), the 3x3 and 5x5
// layers of the coarse-to-fine nets (models_c2f.lua:124-126, 247-254); the reference hands these to cuDNN v3 / THNN.
//
//   Y = A^T [ U (.) V ] A,  U = G g G^T,  V = B^T d B        (wino.hip)
//   dL/dU[pos] = sum over tiles of  dM[pos] (.) V[pos],   dM = A dY A^T  (the 2x2 output-gradient block of a tile, 4x4)
//   dL/dg      = G^T (dL/dU) G                                (wino_wgrad_finish_block in igemm.hip, after the split sums)
// 16 multiplies per tile and (out, in) pair instead of 36: 2.25 x fewer MFMAs than the tap-by-tap contraction (wgrad_ws_kernel).
//
// Shape: per position an [out-channels x tiles] x [tiles x in-channels] contraction, the reduction running over the TILES of the
// batch.  A block owns 64 out-channels x 64 in-channels of ONE (parity, group) unit for all 16 positions (a wave: 32 x 32 x 16 =
// the 256 accumulator registers, as in wino_kernel) and a contiguous range of tile chunks (gridDim.y splits); the partial
// dL/dU of every split goes to HBM and the finish pass (igemm.hip: wino_wgrad_finish_block) sums the splits, applies G^T . G and
// scatters the 3x3 sub-kernel gradients into the reference [O][I][k][k] taps (5x5: four sub-kernels; folded up-convolution: each 5x5 tap collects
// the folded tap it was summed into, one per output parity).
//
// K chunk = 8 consecutive tiles; wave m transforms the tile PAIR (2m, 2m + 1) of the chunk for its lane's channel: lane = channel
// (64 consecutive channels of one pixel = one coalesced 256-byte load), the two tiles of the pair are the two halves of a
// register pair, so both transforms run as packed adds and a value pair is one ds_write_b64.  All tile arithmetic is
// wave-uniform (scalar unit); a lane's offset is its channel + the row of its tile pair, re-computed once per chunk.  Signs: A = [1 0; 1 1; 1 -1; 0 -1] -- the minus signs of its last
// row are left out here (dM'[i][j] = s_i s_j dM[i][j], s = (1, 1, 1, -1)) and applied by the finish kernel.
// LDS stage: dM'[pos 16][k half 2][out-channel 64][4 tiles] | V[pos 16][k half 2][in-channel 64][4 tiles]  (k = tile in chunk).
#include "fg_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FG_OOB 0x7FFFFFF0
#define WW_STAGE 16384

__device__ __forceinline__ float ww_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 ww_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 ww_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// TRACE (measurement kernels only, FG_WINO_WGRAD_TRACE=1 | 2): s_memtime rows in wino.hip's formats -- 1: wave 0 at entry, after the
// prologue's barrier, after every K chunk and after the epilogue; 2: every 8 MFMA slots of the first 14 chunks
template <int TRACE>
__device__ __forceinline__ void ww_body(const WinoWgradArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nbc = a.Cpad >> 6;
    const int tn = blockIdx.x / nbc, tc = blockIdx.x - tn * nbc;
    const int pg = blockIdx.z, par = pg / a.KG, grp = pg - par * a.KG;
    const int nct = (a.T + 7) >> 3;
    const int c0 = blockIdx.y * a.chunks_per_split;
    const int NC = max(0, min(nct, c0 + a.chunks_per_split) - c0);
    const int NCE = (NC + 1) & ~1;
    if (NC <= 0) return;             // (never: fg_launch_wino_wgrad refuses a split without chunks; keeps the zero accumulators of that path out of scratch)
    unsigned long long* trc = nullptr;
    if (TRACE) {
        if (wid == 0 && a.dbg_trace) trc = a.dbg_trace + ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 128;
        if (trc && lane == 0) trc[0] = __builtin_amdgcn_s_memtime();
    }
    int ci = 0;

    // A lane's offset = its channel + the row of its wave's tile pair (4 patch rows / 2 gradient rows, re-computed per chunk, row
    // validity baked in: an invalid row's offset is out of range -> the load returns 0 without touching memory); the COLUMN of a
    // request is a wave-uniform scalar offset (soffset is not range-checked, so the patch descriptor starts 2 pixels before X:
    // every in-range row offset stays >= 0 for patch columns down to -2), its validity one v_cndmask on a per-chunk scalar bit.
    const int dRow = a.osy * a.Wo * a.Nd * 4, dCol = a.osx * a.Nd * 4;
    const int xRow = a.isy * a.Wi * a.Cx * 4, xCol = a.isx * a.Cx * 4;
    const int xShift = 2 * a.Cx * 4;
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)a.d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.X - xShift), 0, (int)a.x_bytes + xShift, 0x00020000);
    const int chD = tn * 64 + lane, chX = tc * 64 + lane;
    const int vD = chD < a.Nd ? chD * 4 : FG_OOB, vX = chX < a.Cx ? chX * 4 : FG_OOB;
    const int goy = a.goy[grp], gox = a.gox[grp], ooy = a.ooy[par], oox = a.oox[par];
    const int cend = c0 + NC;

    // store offsets: [pos][k half][channel][4]: this wave's tile pair is k = 2 wid, 2 wid + 1
    const int sw = (wid >> 1) * 256 + lane * 4 + (wid & 1) * 2;
    const int a_rd = (lane >> 5) * 256 + (wm * 32 + (lane & 31)) * 4;
    const int b_rd = 8192 + (lane >> 5) * 256 + (wn * 32 + (lane & 31)) * 4;

    // the load cursor: chunk cl's requests use (vrow, vdr, colmask)
    int cl = c0;
    int vrow[4], vdr[2], vo[24], colmask = 0;
    int cur_b = 0, cur_y0 = 0, cur_x0 = 0, cur_dy = 0, cur_dx = 0, cur_tv = 0;
    // (in pieces, so that the K loop can give each a slot of its own: a burst of n vector-ALU instructions behind an MFMA costs the
    // wave ~14 + 4.3 (n - 1) cycles of matrix-pipe time, a lone one 14 -- scripts/ubench/issue.hip)
#define WW_CURSOR_A()                                                                                      \
    {                                                                                                      \
        const int t0 = cl * 8 + 2 * wid;                        /* first tile of this wave's pair */      \
        cur_tv = (t0 < a.T && cl < cend) ? 1 : 0;                                                          \
        const int tt = cur_tv ? t0 : 0;                                                                    \
        const int tx = tt & (a.TW - 1), ty = (tt >> a.lgTW) & (a.TH - 1);                                  \
        cur_b = tt >> (a.lgTW + a.lgTH);                                                                   \
        cur_y0 = a.isy * 2 * ty + goy; cur_x0 = a.isx * 2 * tx + gox;                                      \
        cur_dy = a.osy * 2 * ty + ooy; cur_dx = a.osx * 2 * tx + oox;                                      \
    }
#define WW_CURSOR_B()                                                                                      \
    {                                                                                                      \
        const int rowbase = ((cur_b * a.Hi + cur_y0) * a.Wi + cur_x0) * a.Cx * 4 + xShift;                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            vrow[i] = (cur_tv && (unsigned)(cur_y0 + a.isy * i) < (unsigned)a.Hi) ? vX + (rowbase + i * xRow) : FG_OOB; \
        colmask = 0;                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) if ((unsigned)(cur_x0 + a.isx * j) < (unsigned)a.Wi) colmask |= 1 << j; \
        const int dbase = ((cur_b * a.Ho + cur_dy) * a.Wo + cur_dx) * a.Nd * 4;                            \
        vdr[0] = cur_tv ? vD + dbase : FG_OOB;                                                             \
        vdr[1] = cur_tv ? vD + (dbase + dRow) : FG_OOB;                                                    \
    }
    // the offset of patch element (row i, column j of the pair's 4 x 6 window): rows I0 .. I0 + 1
#define WW_CURSOR_C(I0)                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int i = (I0); i < (I0) + 2; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) vo[i * 6 + j] = ((colmask >> j) & 1) ? vrow[i] : FG_OOB; \
    }
#define WW_CURSOR() { WW_CURSOR_A(); WW_CURSOR_B(); WW_CURSOR_C(0); WW_CURSOR_C(2); }
    // one value of the pair (tile half h: 0 = tile 2m, 1 = tile 2m + 1): patch element (i, j) / gradient element (r, e)
#define WW_LDX(i, j, h) ww_load(xrsrc, vo[(i) * 6 + (j) + 2 * (h)], ((j) + 2 * (h)) * xCol)
#define WW_LDD(r, e, h) ww_load(drsrc, vdr[r], ((e) + 2 * (h)) * dCol)

    f32x2 xa[16], xb[16], da[4], db[4];
#define WW_LOAD_ALL(xs, ds)                                                                                \
    {                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) { ds[q].x = WW_LDD(q >> 1, q & 1, 0); ds[q].y = WW_LDD(q >> 1, q & 1, 1); } \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) { xs[q].x = WW_LDX(q >> 2, q & 3, 0); xs[q].y = WW_LDX(q >> 2, q & 3, 1); } \
    }
    // the transforms of a whole chunk and its stores, not interleaved with anything (prologue).  dM' = A' dY A'^T with
    // A' = [1 0; 1 1; 1 -1; 0 1] (signs applied by the finish kernel); V = B^T d B
#define WW_XFORM_STORE(S, xs, ds)                                                                          \
    {                                                                                                      \
        float* Ms = (S) + sw;                                                                              \
        float* Vs = (S) + 8192 + sw;                                                                       \
        f32x2 u_[8];                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                    \
            u_[0 + e] = ds[0 + e]; u_[2 + e] = ds[0 + e] + ds[2 + e]; u_[4 + e] = ds[0 + e] - ds[2 + e]; u_[6 + e] = ds[2 + e]; \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            *(f32x2*)(Ms + (i * 4 + 0) * 512) = u_[i * 2];                                                 \
            *(f32x2*)(Ms + (i * 4 + 1) * 512) = u_[i * 2] + u_[i * 2 + 1];                                 \
            *(f32x2*)(Ms + (i * 4 + 2) * 512) = u_[i * 2] - u_[i * 2 + 1];                                 \
            *(f32x2*)(Ms + (i * 4 + 3) * 512) = u_[i * 2 + 1];                                             \
        }                                                                                                  \
        f32x2 w_[16];                                                                                      \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                    \
            w_[0 + x] = xs[0 + x] - xs[8 + x];                                                             \
            w_[4 + x] = xs[4 + x] + xs[8 + x];                                                             \
            w_[8 + x] = xs[8 + x] - xs[4 + x];                                                             \
            w_[12 + x] = xs[4 + x] - xs[12 + x];                                                           \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            *(f32x2*)(Vs + (i * 4 + 0) * 512) = w_[i * 4 + 0] - w_[i * 4 + 2];                             \
            *(f32x2*)(Vs + (i * 4 + 1) * 512) = w_[i * 4 + 1] + w_[i * 4 + 2];                             \
            *(f32x2*)(Vs + (i * 4 + 2) * 512) = w_[i * 4 + 2] - w_[i * 4 + 1];                             \
            *(f32x2*)(Vs + (i * 4 + 3) * 512) = w_[i * 4 + 1] - w_[i * 4 + 3];                             \
        }                                                                                                  \
    }

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    f32x2 bsum = {0.f, 0.f};          // bias gradient: the sum of the four gradient elements of every tile = position (1, 1) of dM'

    f32x4 fa[2][2], fb[2][2];
    if (NC > 0) {
        WW_CURSOR(); WW_LOAD_ALL(xa, da); ++cl;
        WW_CURSOR(); WW_LOAD_ALL(xb, db); ++cl;
        WW_CURSOR();                                     // chunk c0 + 2: requested by the first chunk of the K loop
    }
    // (the accumulators are written under the latency of the requests above; hipcc would sink their initialisation behind the barrier)
#pragma unroll
    for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[p]));
    __builtin_amdgcn_sched_barrier(0);
    if (NC > 0) {
        bsum += (da[0] + da[1]) + (da[2] + da[3]);
        WW_XFORM_STORE(smem, xa, da);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (NC > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fa[0][h] = *(const f32x4*)(smem + a_rd + h * 512);
            fb[0][h] = *(const f32x4*)(smem + b_rd + h * 512);
        }
    }
    if (TRACE && trc && lane == 0) trc[1] = __builtin_amdgcn_s_memtime();
#define WW_STAMP() { if (TRACE == 1 && trc && lane == 0) trc[2 + (ci < 119 ? ci : 119)] = __builtin_amdgcn_s_memtime(); ++ci; }

    // One K chunk = 64 MFMA slots, as in wino_kernel.  HAS1: the next chunk's values sit in (XS, DS), requested a chunk ago:
    //   slot 0        dM' (16 packed adds in one burst; the copies are free)           slots 2..17  dM' stores
    //   slot 19       V column pass, in place (16 packed adds)      slot 21  V row pass        slots 23..38 V stores
    //   end of slot 55: lgkmcnt(0) + barrier
    // (vector-ALU work in few bursts: a lone VALU instruction behind an MFMA costs the wave 14 matrix-pipe cycles, each further one
    //  of a burst 4.3 -- scripts/ubench/issue.hip; LDS and buffer instructions issue for 0 .. 5)
    // HAS2: the chunk after that is requested into (XL, DL): the 8 gradient values in slots 1..4, the 32 patch values in slots
    //   5..36 (one request per slot); the cursor of the chunk after THAT in slots 57 (scalar), 58, 60, 61 (30 row / element offsets)
#define WW_CHUNK(HAS1, HAS2, XS, DS, XL, DL)                                                               \
    {                                                                                                      \
        const float* Sc = smem + s * WW_STAGE;                                                             \
        float* Sn = smem + (s ^ 1) * WW_STAGE;                                                             \
        float* Ms = Sn + sw;                                                                               \
        float* Vs = Sn + 8192 + sw;                                                                        \
        f32x2 t_, u_[8], m_[16];                                                                           \
        _Pragma("unroll") for (int pr = 0; pr < 8; ++pr) {                                                 \
            _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                \
                const int sl = pr * 8 + m, h = m & 1, j = m >> 1, pos = 2 * pr + h;                        \
                if (TRACE == 2 && (sl & 7) == 0 && ci < 14) {                                              \
                    if (trc && lane == 0) trc[8 + ci * 8 + (sl >> 3)] = __builtin_amdgcn_s_memtime();       \
                }                                                                                          \
                acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pr & 1][h][j], fb[pr & 1][h][j], acc[pos], 0, 0, 0); \
                if (m < 4 && pr < 7) {                                                                     \
                    const int np = 2 * (pr + 1) + (m >> 1);                                                \
                    if ((m & 1) == 0) fa[(pr + 1) & 1][m >> 1] = *(const f32x4*)(Sc + a_rd + np * 512);    \
                    else fb[(pr + 1) & 1][m >> 1] = *(const f32x4*)(Sc + b_rd + np * 512);                 \
                }                                                                                          \
                if (HAS1) {                                                                                \
                    if (m < 4 && pr == 7) {                                                                \
                        if ((m & 1) == 0) fa[0][m >> 1] = *(const f32x4*)(Sn + a_rd + (m >> 1) * 512);     \
                        else fb[0][m >> 1] = *(const f32x4*)(Sn + b_rd + (m >> 1) * 512);                  \
                    }                                                                                      \
                    if (HAS2) {                                                                            \
                        if (sl >= 1 && sl < 5) {                                                           \
                            const int q = sl - 1;                                                          \
                            DL[q].x = WW_LDD(q >> 1, q & 1, 0); DL[q].y = WW_LDD(q >> 1, q & 1, 1);        \
                        }                                                                                  \
                        if (sl >= 5 && sl < 37) {                                                          \
                            const int q = (sl - 5) >> 1, hh = (sl - 5) & 1;                                \
                            if (hh == 0) XL[q].x = WW_LDX(q >> 2, q & 3, 0); else XL[q].y = WW_LDX(q >> 2, q & 3, 1); \
                        }                                                                                  \
                    }                                                                                      \
                    if (HAS2 && sl == 57) { ++cl; WW_CURSOR_A(); }     /* the cursor of the chunk after: slots without other work */ \
                    if (HAS2 && sl == 58) { WW_CURSOR_B(); }                                               \
                    if (HAS2 && sl == 60) { WW_CURSOR_C(0); }                                              \
                    if (HAS2 && sl == 61) { WW_CURSOR_C(2); }                                              \
                    /* dM': u = A' d (columns e = 0, 1), then m = u A'^T -- one burst */                     \
                    if (sl == 0) {                                                                         \
                        bsum = ww_add(bsum, ww_add(ww_add(DS[0], DS[1]), ww_add(DS[2], DS[3])));           \
                        u_[0] = DS[0]; u_[1] = DS[1]; u_[6] = DS[2]; u_[7] = DS[3];                        \
                        u_[2] = ww_add(DS[0], DS[2]); u_[3] = ww_add(DS[1], DS[3]);                        \
                        u_[4] = ww_sub(DS[0], DS[2]); u_[5] = ww_sub(DS[1], DS[3]);                        \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
                            m_[i * 4 + 0] = u_[i * 2]; m_[i * 4 + 3] = u_[i * 2 + 1];                      \
                            m_[i * 4 + 1] = ww_add(u_[i * 2], u_[i * 2 + 1]);                              \
                            m_[i * 4 + 2] = ww_sub(u_[i * 2], u_[i * 2 + 1]);                              \
                        }                                                                                  \
                    }                                                                                      \
                    if (sl >= 2 && sl < 18) *(f32x2*)(Ms + (sl - 2) * 512) = m_[sl - 2];                   \
                    /* V = B^T d B in place in XS: the column pass and the row pass, one burst each */      \
                    if (sl == 19) {                                                                        \
                        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                    \
                            XS[0 + x] = ww_sub(XS[0 + x], XS[8 + x]);                                      \
                            XS[12 + x] = ww_sub(XS[4 + x], XS[12 + x]);                                    \
                            t_ = ww_add(XS[4 + x], XS[8 + x]);                                             \
                            XS[8 + x] = ww_sub(XS[8 + x], XS[4 + x]); XS[4 + x] = t_;                      \
                        }                                                                                  \
                    }                                                                                      \
                    if (sl == 21) {                                                                        \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
                            XS[i * 4 + 0] = ww_sub(XS[i * 4 + 0], XS[i * 4 + 2]);                          \
                            XS[i * 4 + 3] = ww_sub(XS[i * 4 + 1], XS[i * 4 + 3]);                          \
                            t_ = ww_add(XS[i * 4 + 1], XS[i * 4 + 2]);                                     \
                            XS[i * 4 + 2] = ww_sub(XS[i * 4 + 2], XS[i * 4 + 1]); XS[i * 4 + 1] = t_;      \
                        }                                                                                  \
                    }                                                                                      \
                    if (sl >= 23 && sl < 39) *(f32x2*)(Vs + (sl - 23) * 512) = XS[sl - 23];                \
                    if (sl == 55) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
                }                                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
        }                                                                                                  \
    }
    int s = 0;
    if (NCE > 0) {
        for (; ci + 2 < NCE;) {
            WW_CHUNK(true, true, xb, db, xa, da);
            WW_STAMP();
            s ^= 1;
            WW_CHUNK(true, true, xa, da, xb, db);
            WW_STAMP();
            s ^= 1;
        }
        WW_CHUNK(true, false, xb, db, xa, da);
        WW_STAMP();
        s ^= 1;
        WW_CHUNK(false, false, xa, da, xb, db);
        WW_STAMP();
    }
#undef WW_CURSOR
#undef WW_CURSOR_A
#undef WW_CURSOR_B
#undef WW_LDX
#undef WW_LDD
#undef WW_LOAD_ALL
#undef WW_XFORM_STORE
#undef WW_CHUNK

    // ---- partial dL/dU' of this split: Part[pg][split][out][in][pos 16] -- the 16 positions of a channel pair are 64 contiguous bytes
    // (four 16-byte stores per accumulator row here, four 16-byte loads per split in the finish pass).  Lane l holds column (l & 31) =
    // in-channel; rows = out-channels
    {
        float* outp = a.Part + (((size_t)pg * a.S + blockIdx.y) * 16) * (size_t)a.Npad * a.Cpad;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, FG_OOB, 0x00020000);
        const int col = tc * 64 + wn * 32 + (lane & 31);
        const int row0 = tn * 64 + wm * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            const int vo = (row * a.Cpad + col) * 64;
            __builtin_amdgcn_sched_barrier(0);           // one accumulator row at a time (16 reads of the accumulator file, 4 stores)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[4 * q][r], acc[4 * q + 1][r], acc[4 * q + 2][r], acc[4 * q + 3][r]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), orsrc, vo, q * 16, 0);
            }
        }
    }
    // ---- bias-gradient partial: the four waves' sums of this block's channels (only the blocks of in-channel block 0, group 0)
    if (a.bias_part && tc == 0 && grp == 0) {
        __syncthreads();
        float* red = smem;
        red[wid * 64 + lane] = bsum.x + bsum.y;
        __syncthreads();
        if (wid == 0 && chD < a.Nd)
            a.bias_part[((size_t)par * a.S + blockIdx.y) * a.Nd + chD] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    }
    if (TRACE && trc && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE == 1) trc[(NC < 120 ? NC : 120) + 2] = __builtin_amdgcn_s_memtime();
        trc[127] = (unsigned long long)__builtin_amdgcn_s_getreg(((3 - 1) << 11) | (0 << 6) | 20) | ((unsigned long long)NC << 32);   // XCC_ID, NC
        trc[126] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);                                    // HW_ID
    }
#undef WW_STAMP
}
__global__ __launch_bounds__(256) void wino_wgrad_kernel(const WinoWgradArgs a) { ww_body<0>(a); }
#ifdef FG_MEASURE       // measurement build only (libfacegen_hip_measure.so)
template <int TRACE>
__global__ __launch_bounds__(256) void wino_wgrad_trace_kernel(const WinoWgradArgs a) { ww_body<TRACE>(a); }

// FG_WINO_WGRAD_TRACE=1 | 2 (measurement only): the launch runs the trace kernel four times (three to settle the clocks); the per-block
// s_memtime rows of the fourth are appended to FG_WS_TRACE_FILE (row formats of wino.hip: scripts/ws_trace_report.py,
// scripts/wino_trace2_report.py)
static int ww_trace_launch(fg_ctx* ctx, const WinoWgradArgs& a_in, dim3 grid, size_t lds, int mode) {
    WinoWgradArgs a = a_in;
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* dvc = nullptr;
    if (hipMalloc((void**)&dvc, nblk * 128 * 8) != hipSuccess) return fg_set_err(ctx, FG_ERR_NOMEM, "trace buffer");
    (void)hipMemset(dvc, 0, nblk * 128 * 8);
    a.dbg_trace = dvc;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)wino_wgrad_trace_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)wino_wgrad_trace_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, ctx->stream);
        if (mode == 2) hipLaunchKernelGGL(wino_wgrad_trace_kernel<2>, grid, dim3(256), lds, ctx->stream, a);
        else hipLaunchKernelGGL(wino_wgrad_trace_kernel<1>, grid, dim3(256), lds, ctx->stream, a);
        (void)hipEventRecord(e1, ctx->stream);
    }
    FG_CHECK_LAUNCH(ctx);
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float wall_ms = 0.f;
    (void)hipEventElapsedTime(&wall_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::vector<unsigned long long> host(nblk * 128);
    FG_HIP(ctx, hipMemcpy(host.data(), dvc, nblk * 128 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dvc);
    const char* path = getenv("FG_WS_TRACE_FILE");
    FILE* f = fopen(path ? path : "/tmp/fg_ws_trace.txt", "a");
    if (f) {
        fprintf(f, "# launch wino_wgrad(dbg=%d)/%s BN=128 blocks=%zu T=%d Npad=%d Cpad=%d units=%d S=%d wall_us=%.1f\n", mode == 2 ? 100 : 0, a.tag ? a.tag : "?", nblk,
                a.T, a.Npad, a.Cpad, a.P * a.KG, a.S, wall_ms * 1e3);
        for (size_t b = 0; b < nblk; ++b) {
            const unsigned long long* r = host.data() + b * 128;
            const int kt = (int)(r[127] >> 32), xcc = (int)(r[127] & 0xffffffff);
            fprintf(f, "%zu %d %d %llu", b, xcc, kt, r[126]);
            if (mode == 2) { for (int i = 0; i < 126; ++i) fprintf(f, " %llu", r[i]); }
            else for (int i = 0; i < (kt < 120 ? kt : 120) + 3; ++i) fprintf(f, " %llu", r[i]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return FG_OK;
}
#endif   // FG_MEASURE

int fg_launch_wino_wgrad(fg_ctx* ctx, const WinoWgradArgs& a) {
    if (a.chunks_per_split < 1 || (long long)(a.S - 1) * a.chunks_per_split >= (a.T + 7) / 8)
        return fg_set_err(ctx, FG_ERR_INVALID, "winograd wgrad: a split without chunks (S = %d x %d chunks, %d tiles)", a.S, a.chunks_per_split, a.T);
    if (a.Npad % 64 || a.Cpad % 64 || a.S < 1 || a.lgTW < 1 || a.lgTH < 0 || a.P < 1 || a.KG < 1 || a.P * a.KG > 4)
        return fg_set_err(ctx, FG_ERR_INVALID, "winograd wgrad: padded channels %% 64, power-of-two tile grid (>= 2 wide), <= 4 units");
    // (the patch descriptor starts 2 pixels in front of X and is that much longer: the out-of-range marker must stay out of range)
    if (a.x_bytes <= 0 || a.x_bytes + 8LL * a.Cx >= (long long)FG_OOB || a.d_bytes <= 0 || a.d_bytes >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "winograd wgrad: operands must be < 2 GiB per launch");
    const size_t lds = (size_t)(2 * WW_STAGE) * sizeof(float);
    static char attr_key;
    if (fg_attr_first(ctx, &attr_key)) {
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wino_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid((a.Npad / 64) * (a.Cpad / 64), a.S, a.P * a.KG);
    const double exec = 2.0 * (double)a.Nd * a.Cx * grid.z * 16.0 * (double)a.T;     // live channel pairs x live tiles (see fg_launch_wino)
#ifdef FG_MEASURE
    {
        static int tr = -1;
        if (tr < 0) { const char* e = getenv("FG_WINO_WGRAD_TRACE"); tr = e ? atoi(e) : 0; }
        if (tr) return ww_trace_launch(ctx, a, grid, lds, tr);
    }
#endif
    char label[96];
    snprintf(label, sizeof(label), "wino_wgrad_kernel/%s", a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    hipLaunchKernelGGL(wino_wgrad_kernel, grid, dim3(256), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
