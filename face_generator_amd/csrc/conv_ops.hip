// Host-side drivers that turn a convolution / Linear layer (reference geometry) into launches of the
// igemm / wgrad kernels: group-offset tables, nearest-x2 tap folding, tile and split heuristics.
#include "fg_internal.h"
#include "conv_ops.h"
#include <string.h>
#include <stdlib.h>

static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

void fg_geom_weightmap(const ConvGeom& g, WeightMap* wm) {
    memset(wm, 0, sizeof(*wm));
    wm->O = g.Cout; wm->I = g.Cin; wm->k = g.k; wm->pad = g.pad;
    wm->o_c = g.o_c; wm->o_hw = g.o_hw; wm->i_c = g.i_c; wm->i_hw = g.i_hw;
    if (g.fold) {
        wm->kind = 1;
        fg_fold_window(g.k, g.pad, &wm->T, &wm->rmin);
        wm->G = wm->T * wm->T;
        wm->P = 4;
    } else {
        wm->kind = 0;
        wm->T = g.k; wm->rmin = -g.pad;
        wm->G = g.k * g.k;
        wm->P = 1;
    }
}

void fg_geom_packmap(const ConvGeom& g, WeightMap* wm) {
    fg_geom_weightmap(g, wm);
    if (g.wino) {
        // 16 positions x (parities x groups) transformed sub-kernels per (out, in) pair; the same count in both packs
        int P, KG; fg_wino_pack_shape(wm->kind, g.wino, 0, &P, &KG);
        wm->wino = g.wino; wm->G = 16 * P * KG; wm->P = 1;
    }
}
// g.wino: 1 = 3x3 / pad 1 layers, and nearest-x2-folded layers whose folded window is 3x3 at offset -1 (5x5 / pad 2, 3x3 / pad 1):
// every output parity is a 3x3 / pad 1 convolution of the source; 2 = 5x5 / pad 2 layers as four 3x3 sub-kernels
void fg_geom_set_wino(ConvGeom& g, int fusion) {
    g.wino = 0;
    if (g.stride == 2 || g.H < 2 || g.W < 2 || (g.H & 1) || (g.W & 1) || g.Cin % 8 || g.Cout % 8 || g.Cin <= 4 || g.Cout <= 4) return;
    if (g.fold) {
        int T, rmin; fg_fold_window(g.k, g.pad, &T, &rmin);
        if ((fusion & FG_FUSE_WINOGRAD_UP) && T == 3 && rmin == -1) g.wino = 1;
    } else if (g.k == 3 && g.pad == 1) { if (fusion & FG_FUSE_WINOGRAD) g.wino = 1; }
    else if (g.k == 5 && g.pad == 2) { if (fusion & FG_FUSE_WINOGRAD_5X5) g.wino = 2; }
}

static inline int pad_rows(int n) { return n < 128 ? fg_round_up(n, 64) : fg_round_up(n, 128); }

void fg_geom_pack_dims(const ConvGeom& g, int* rows_f, int* cols_f, int* rows_b, int* cols_b) {
    if (g.wino) {        // 64 output channels per block, K chunks of 8 channels (wino.hip)
        *rows_f = fg_round_up(g.Cout, 64); *cols_f = fg_round_up(g.Cin, 8);
        *rows_b = fg_round_up(g.Cin, 64);  *cols_b = fg_round_up(g.Cout, 8);
        return;
    }
    *rows_f = pad_rows(g.Cout); *cols_f = fg_round_up(g.Cin, 32);
    *rows_b = pad_rows(g.Cin);  *cols_b = fg_round_up(g.Cout, 32);
}
long long fg_geom_pack_floats(const ConvGeom& g, int bwd) {
    WeightMap wm; fg_geom_packmap(g, &wm);
    int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
    return (long long)wm.P * wm.G * (bwd ? (long long)rb * cb : (long long)rf * cf);      // (Winograd packs: G = 16 x parities x groups)
}

int fg_conv_pack(fg_ctx* ctx, const ConvGeom& g, const float* W, float* wp_fwd, float* wp_bwd) {
    WeightMap wm; fg_geom_packmap(g, &wm);
    int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
    int rc;
    if (wp_fwd && (rc = fg_launch_pack_weights(ctx, wm, 0, W, wp_fwd, rf, cf))) return rc;
    if (wp_bwd && (rc = fg_launch_pack_weights(ctx, wm, 1, W, wp_bwd, rb, cb))) return rc;
    return FG_OK;
}

// ---- tile / split heuristics: fill >= 2 blocks per CU (256 CUs) where the problem allows ----
static void choose_igemm(long long M, int Npad, int ksteps, int P, int math, int* tile, int* splits) {
    const long long target = 512;
    long long b0 = (Npad % 128 == 0) ? (long long)fg_cdiv(M, 128) * (Npad / 128) * P : 0;
    long long b1 = (long long)fg_cdiv(M, 128) * (Npad / 64) * P;
    long long b2 = (long long)fg_cdiv(M, 64) * (Npad / 64) * P;
    *splits = 1;
    {   // large layers: wave-specialised 256x128 kernel when it fills the chip with whole rounds of 256 blocks
        static int use_ws = -1;
        if (use_ws < 0) { const char* e = getenv("FG_IGEMM_WS"); use_ws = e ? atoi(e) : 1; }
        const long long bw = (Npad % 64 == 0) ? (long long)fg_cdiv(M, 256) * (Npad / ((Npad % 128 == 0) ? 128 : 64)) * P : 0;
        if (use_ws && bw >= 256 && bw % 256 == 0 && M % 256 == 0) { *tile = 4; return; }
        // fp32: layers whose 256x128 tiling leaves part of the chip idle but whose 256x64 tiling gives whole rounds of 256
        // blocks (D's first mid-size convolution, G's first up-convolution at half batch): tile 5 = the same kernel, BN = 64
        if (use_ws && math != 6 && Npad % 128 == 0 && M % 256 == 0) {
            const long long bw64 = (long long)(M / 256) * (Npad / 64) * P;
            if (bw64 >= 256 && bw64 % 256 == 0) { *tile = 5; return; }
        }
        // bf16x6: a K-step is short and cheap, so mid-size layers also use the wave-specialised kernel (256x64 tiles for
        // layers with 64 output channels), split over K so that exactly one round of 256 blocks fills the chip
        // (>= 12 sixteen-channel steps per block)
        if (use_ws && math == 6 && Npad % 64 == 0 && M % 256 == 0) {
            const int bn = (Npad % 128 == 0) ? 128 : 64;
            const long long b6 = (long long)(M / 256) * (Npad / bn) * P;
            if (b6 >= 256 && b6 % 256 == 0) { *tile = 4; return; }
            if (b6 > 0 && b6 < 256 && 256 % b6 == 0 && (2 * ksteps) / (256 / b6) >= 12) { *tile = 4; *splits = (int)(256 / b6); return; }
        }
    }
    if (b0 >= target) { *tile = 0; return; }
    if (b1 >= target) { *tile = 1; return; }
    {   // a Linear over very many input features and few samples (65536 -> 512 at M = 128, models_c2f.lua:262: 8.6 GFLOP): 128 x 128
        // tiles split over K until one round of 256 blocks fills the chip, instead of 64 x 64 tiles capped at 16 splits (round 4;
        // FG_LINEAR_FWD128=0 switches back)
        static int on = -1;
        if (on < 0) { const char* e = getenv("FG_LINEAR_FWD128"); on = e ? atoi(e) : 1; }
        if (on && math != 6 && b0 > 0 && b0 <= 16 && ksteps >= 1024) {
            int s = (int)(256 / b0);
            if (s > ksteps / 16) s = ksteps / 16;
            const int per = (ksteps + s - 1) / s;
            *tile = 0; *splits = (ksteps + per - 1) / per;
            return;
        }
    }
    *tile = 2;
    // (a contraction of <= 4 K-steps over >= 128 tiles -- G's first Linear, 100 -> 8192 -- is shorter than the extra pass that
    // would sum its partials)
    if (b2 < 384 && ksteps >= 4 && !(ksteps <= 4 && b2 >= 128)) {      // split-K over K-steps, >= 2 steps per split
        int s = fg_cdiv(target, b2 > 0 ? b2 : 1);
        if (s > ksteps / 2) s = ksteps / 2;
        if (s > 16) s = 16;
        int per = (ksteps + s - 1) / s;
        s = (ksteps + per - 1) / per;   // every split non-empty
        *splits = s < 1 ? 1 : s;
    }
}
static void fill_wino(WinoArgs& w, const ConvGeom& g, int bwd);
// Winograd launch: blocks of 64 tiles x 64 output channels; when they do not fill the chip the K chunks (8 channels each) are
// split over gridDim.y, >= 4 chunks per split (the partials are summed by the pass behind the layer, like every split-K launch)
static int choose_wino_splits(long long T, int Npad, int C) {      // Npad: all parities' channel blocks x 64; C: all groups' channels
    const long long blocks = ((T + 63) / 64) * (Npad / 64);
    const int nch = C / 8;
    if (blocks >= 192 || nch < 8) return 1;
    int s = (int)(256 / blocks);
    if (s > nch / 4) s = nch / 4;
    if (s < 1) s = 1;
    const int per = (nch + s - 1) / s;
    return (nch + per - 1) / per;          // every split non-empty
}
// Winograd-domain weight gradient (wino_wgrad.hip): blocks of 64 x 64 channel pairs per (parity, group) unit, the reduction over
// the tiles (chunks of 8) split S ways so that one round of ~256 blocks fills the chip.  Taken where the tile grid is a power of
// two each way, both channel counts fill whole 64-blocks and every block still reduces over >= 24 chunks with >= 3/4 of the chip
// busy (D's 64 -> 128 ... 256 -> 512 layers on 16x16 ... 4x4 maps do not: 2 ... 32 channel blocks; they keep the tap-by-tap
// kernels).  false = not taken.
static long long g_ww_min_chunks = 24, g_ww_min_blocks = 192;
void fg_plan_env_init() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = getenv("FG_WINO_WGRAD_MIN_CHUNKS")) g_ww_min_chunks = atoll(e) > 0 ? atoll(e) : 1;
    if (const char* e = getenv("FG_WINO_WGRAD_MIN_BLOCKS")) g_ww_min_blocks = atoll(e);
}
void fg_plan_set_wino_wgrad_thresholds(long long min_chunks, long long min_blocks) {
    g_ww_min_chunks = min_chunks > 0 ? min_chunks : 24;
    g_ww_min_blocks = min_blocks > 0 ? min_blocks : 192;
}
static bool choose_wino_wgrad(const ConvGeom& g, int* S, int* cps) {
    if (!g.wino || g.stride == 2 || (g.Cout % 64) || (g.Cin % 64)) return false;
    const int TH = g.H / 2, TW = g.W / 2;
    if (ilog2_exact(TH) < 0 || ilog2_exact(TW) < 1) return false;
    WeightMap wm; fg_geom_weightmap(g, &wm);
    int P, KG; fg_wino_pack_shape(wm.kind, g.wino, 0, &P, &KG);
    const long long T = (long long)g.B * TH * TW, nct = (T + 7) / 8;
    const long long base = (long long)(g.Cout / 64) * (g.Cin / 64) * P * KG;
    // (planning thresholds of the process: the environment is read once, at the first fg_ctx_create; the parity tests reach the
    // small-shape corners through fg_test_set_wino_wgrad_thresholds)
    const long long min_chunks = g_ww_min_chunks, min_blocks = g_ww_min_blocks;
    long long s = base >= 256 ? 1 : 256 / base;
    if (s > nct / min_chunks) s = nct / min_chunks;
    if (s < 1 || base * s < min_blocks) return false;
    const long long per = (nct + s - 1) / s;
    *cps = (int)per;
    *S = (int)((nct + per - 1) / per);          // every split non-empty
    return true;
}
static long long wino_wgrad_part_floats(const ConvGeom& g, int S) {
    WeightMap wm; fg_geom_weightmap(g, &wm);
    int P, KG; fg_wino_pack_shape(wm.kind, g.wino, 0, &P, &KG);
    return (long long)P * KG * S * 16 * g.Cout * g.Cin;
}
static void choose_wgrad(long long M, int Cout, int Cin, int G, int P, int* tile, int* S, int* mper, int* Npad, int* Cpad) {
    int bt = (Cout >= 128 && Cin >= 128) ? 128 : 64;
    {   // a Linear layer reduces over the B samples only (M = 128): with 128 x 128 tiles Linear(2048, 512) is 64 blocks of 8 K-steps
        // on a quarter of the chip, all prologue and epilogue (24 us for 0.27 GFLOP); 64 x 64 tiles give 256 blocks (round 4;
        // FG_LINEAR_WGRAD64=0 switches back)
        static int on = -1;
        if (on < 0) { const char* e = getenv("FG_LINEAR_WGRAD64"); on = e ? atoi(e) : 1; }
        if (on && G == 1 && P == 1 && M <= 256 && (long long)Cout * Cin <= (1LL << 21)) bt = 64;     // (not the 65536 x 512 layer of models_c2f.lua:262: 8.6 GFLOP)
    }
    *tile = bt == 128 ? 0 : 2;
    *Npad = fg_round_up(Cout, bt);
    *Cpad = fg_round_up(Cin, bt);
    const long long base = (long long)(*Npad / bt) * (*Cpad / bt) * G * P;
    // pick the split count whose grid fills whole waves of 512 resident blocks (2 per CU) best
    const long long maxs = (M + 63) / 64 < 64 ? (M + 63) / 64 : 64;   // >= 2 K-steps per split
    long long best = 1;
    double best_eff = 0.0;
    for (long long s = 1; s <= maxs; ++s) {
        const long long blocks = base * s;
        if (blocks > 2048 && s > 1) break;
        const long long waves = (blocks + 511) / 512;
        const double eff = (double)blocks / (double)(waves * 512);
        // prefer fuller waves; among equals prefer fewer splits (less partial traffic)
        if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    // whole K-steps per split: 64 pixels for the 64-tile kernel (its fast address path takes the wave-uniform part of a step
    // from the step's first pixel), 32 otherwise
    int mp = fg_round_up((int)((M + best - 1) / best), bt == 64 ? 64 : 32);
    *mper = mp;
    *S = (int)((M + mp - 1) / mp);
}


// One block per CU and `base` blocks per pixel split: s splits fill base*s / (256 * rounds) of the chip -- 25 taps x 10 splits = 250
// blocks leave 6 CUs idle for the whole launch (2.3 % of the 6.8 ms 5x5 weight gradient of models_c2f.lua:122).  When every block
// would still run >= 512 K-steps, take the number of ROUNDS (<= 6) whose last round is fullest: 25 x 51 = 1275 blocks = 4.98 rounds.
// More splits are more partial sums to write and add up, so a later round count must win by 1 % (FG_WGRAD_ROUNDS=0: one round).
static long long fg_wgrad_rounds(long long base, long long s1, long long M, int kstep) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("FG_WGRAD_ROUNDS"); on = e ? atoi(e) : 1; }
    if (!on || base <= 0) return s1;
    long long best = s1;
    double beff = (double)(base * s1) / (double)(256 * ((base * s1 + 255) / 256));
    for (int r = 2; r <= 6; ++r) {
        const long long s = 256LL * r / base;
        if (s <= best || M / s / kstep < 512) continue;
        const double eff = (double)(base * s) / (256.0 * r);
        if (eff > beff + 0.01) { beff = eff; best = s; }
    }
    return best;
}

// bf16x6 weight gradient: block tile (dY-channels x X-channels) 256x128 or 128x256; -1 = not tileable.
// S pixel-splits so that one round of ~256 blocks fills the chip with >= 12 sixteen-pixel steps per block.
static int choose_wgrad6(long long M, int Cout, int Cin, int G, int P, int* S, int* mper) {
    // (128x64 / 64x64 tiles were measured slower than the fp32 kernel: too few MFMAs per transposing fragment read)
    static const int RT[2] = {256, 128}, QT[2] = {128, 256};
    int cfg = -1;
    for (int c = 0; c < 2 && cfg < 0; ++c)
        if (Cout % RT[c] == 0 && Cin % QT[c] == 0) cfg = c;
    if (cfg < 0) return -1;
    const long long base = (long long)(Cout / RT[cfg]) * (Cin / QT[cfg]) * G * P;
    long long s = (256 + base / 2) / base;
    if (base * s > 256 && s > 1 && base * s - 256 < base / 2) s -= 1;      // a few blocks over one round would double the launch
    if (s < 1) s = 1;
    const long long maxs = M / 192 > 0 ? M / 192 : 1;
    if (s > maxs) s = maxs;
    s = fg_wgrad_rounds(base, s, M, 16);
    int mp = fg_round_up((int)((M + s - 1) / s), 16);
    *mper = mp;
    *S = (int)((M + mp - 1) / mp);
    return cfg;
}

// fp32 wave-specialised weight gradient: the tilings of choose_wgrad6, else (round 3) 128 dY x 64 X channels with the K-step split
// over the MFMA waves (cfg 2: the 64 -> 128 layers of models_c2f.lua; 64-pixel K-steps, FG_WGRAD_WS64=0 switches it off)
static int choose_wgrad_ws(long long M, int Cout, int Cin, int G, int P, int* S, int* mper) {
    const int c = choose_wgrad6(M, Cout, Cin, G, P, S, mper);
    if (c >= 0) return c;
    static int on = -1;
    if (on < 0) { const char* e = getenv("FG_WGRAD_WS64"); on = e ? atoi(e) : 1; }
    if (!on || Cout % 128 || Cin % 64) return -1;
    const long long base = (long long)(Cout / 128) * (Cin / 64) * G * P;
    long long s = (256 + base / 2) / base;
    if (base * s > 256 && s > 1 && base * s - 256 < base / 2) s -= 1;
    if (s < 1) s = 1;
    const long long maxs = M / 512 > 0 ? M / 512 : 1;          // >= 8 sixty-four-pixel steps per block
    if (s > maxs) s = maxs;
    s = fg_wgrad_rounds(base, s, M, 64);
    const int mp = fg_round_up((int)((M + s - 1) / s), 64);
    *mper = mp;
    *S = (int)((M + mp - 1) / mp);
    return 2;
}

// A/B switch (round 3; default on): FG_WGRAD_WS=0 keeps the symmetric wgrad_kernel for the layers that tile 256 x 128 /
// 128 x 256 channels instead of the wave-specialised wgrad_ws_kernel
// smallest pixel count the wave-specialised weight gradient takes (measurement knob: FG_WGRAD_WS_MINM; default 4096)
static long long fg_wgrad_ws_minm() {
    static long long v = -1;
    if (v < 0) { const char* e = getenv("FG_WGRAD_WS_MINM"); v = e ? atoll(e) : 4096; if (v < 256) v = 256; }
    return v;
}
static bool fg_wgrad_ws_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FG_WGRAD_WS"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// floats of split-K / parity partials the fp32 weight gradient of this layer leaves (upper bound over its two kernels): what the
// deferred-finals arena of a net reserves per convolution so that the partials survive until the end of the backward pass
long long fg_conv_wgrad_part_floats(const ConvGeom& g) {
    WeightMap wm; fg_geom_weightmap(g, &wm);
    const int st = g.stride == 2 ? 2 : 1;
    const long long M = (long long)g.B * (g.H / st) * (g.W / st);
    int wt, S, mper, Np, Cp, S6, mper6;
    choose_wgrad(M, g.Cout, g.Cin, wm.G, wm.P, &wt, &S, &mper, &Np, &Cp);
    long long n = (long long)wm.P * wm.G * S * Np * Cp;
    if (M >= fg_wgrad_ws_minm() && choose_wgrad_ws(M, g.Cout, g.Cin, wm.G, wm.P, &S6, &mper6) >= 0) {
        const long long n6 = (long long)wm.P * wm.G * S6 * g.Cout * g.Cin;
        if (n6 > n) n = n6;
    }
    int Sw, cpsw;
    if (choose_wino_wgrad(g, &Sw, &cpsw)) {      // (either setting of FG_FUSE_WINOGRAD_WGRAD: the bit is read per call)
        const long long nw = wino_wgrad_part_floats(g, Sw);
        if (nw > n) n = nw;
    }
    return n + 64;
}

// floats of bias-gradient partial rows the fp32 weight gradient of this layer leaves for the deferred final (0: it takes the
// separate column-sum pass) -- the same decisions as fg_conv_wgrad_run, for the arena of a net's workspace
long long fg_conv_wgrad_bias_part_floats(const ConvGeom& g) {
    WeightMap wm; fg_geom_weightmap(g, &wm);
    const int st = g.stride == 2 ? 2 : 1;
    const long long M = (long long)g.B * (g.H / st) * (g.W / st);
    int wt, S, mper, Np, Cp;
    choose_wgrad(M, g.Cout, g.Cin, wm.G, wm.P, &wt, &S, &mper, &Np, &Cp);
    long long rows = (long long)wm.P * S;
    {
        int Sw, cpsw;
        if (choose_wino_wgrad(g, &Sw, &cpsw) && (long long)wm.P * Sw > rows) rows = (long long)wm.P * Sw;
    }
    int S6, mper6;
    const int cfg = M >= fg_wgrad_ws_minm() ? choose_wgrad_ws(M, g.Cout, g.Cin, wm.G, wm.P, &S6, &mper6) : -1;
    if (cfg >= 0) {
        WgradArgs a; memset(&a, 0, sizeof(a));
        a.G = wm.G; a.Cpad = g.Cin;
        const long long r = (long long)wm.P * S6 * fg_wgrad_ws_bias_rows(a, cfg);
        if (r <= FG_WS_BIAS_ROWS_MAX && r > rows) rows = r;
    }
    return rows * g.Cout + 64;
}

static long long scratch_for_math(const ConvGeom& g, int math) {
    WeightMap wm; fg_geom_weightmap(g, &wm);
    const long long M = (long long)g.B * g.H * g.W;  // source-resolution M-space
    long long need = 0;
    int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
    const long long outM = g.fold ? M * 4 : M;
    int tile, splits;
    if (g.wino) {
        // forward / data gradient run the Winograd kernel in every math mode: their split partials; the weight gradient below
        // (a smaller run-time batch may split further: splits x tiles is bounded by one round of 256 blocks of 64 tiles)
        WinoArgs wf, wb; memset(&wf, 0, sizeof(wf)); memset(&wb, 0, sizeof(wb));
        fill_wino(wf, g, 0); fill_wino(wb, g, 1);
        const int sf = choose_wino_splits(M / 4, wf.P * wf.Npad, wf.KG * wf.C), sb = choose_wino_splits(M / 4, wb.P * wb.Npad, wb.KG * wb.C);
        const long long nf = sf > 1 ? (long long)sf * outM * g.Cout : 0, nb = sb > 1 ? (long long)sb * M * g.Cin : 0;
        need = nf > nb ? nf : nb;
        // (256 blocks x 64 tiles x 4 outputs x 64 channels: the block count already includes the four parities of a folded layer)
        if (need < 256LL * 64 * 4 * 64 + 64) need = 256LL * 64 * 4 * 64 + 64;
    } else {
    choose_igemm(M, rf, wm.G * (cf / 32), wm.P, math, &tile, &splits);
    {
        long long n = splits > 1 ? (long long)splits * outM * g.Cout : 0;
        // bf16x6: a smaller run-time batch may pick split-K where the full batch does not; its partials are bounded by
        // 256 blocks x one 256x128 tile each
        if (math == 6 && rf % 64 == 0 && n < 256LL * 256 * 128 + 64) n = 256LL * 256 * 128 + 64;
        //  the 128 x 128 split-K branch of choose_igemm (a Linear over >= 32768 input features) splits until ONE round
        // of 256 blocks is full: a SMALLER run-time batch has fewer row blocks and therefore more splits -- its partials are
        // bounded by 256 tiles of 128 x 128, not by the planned batch's split count
        if (math != 6 && rf % 128 == 0 && wm.G * (cf / 32) >= 1024 && n < 256LL * 128 * 128 + 64) n = 256LL * 128 * 128 + 64;
        if (math == 6 && rf % 64 == 0) n += ((M * g.Cin + (long long)wm.P * wm.G * rf * cf) * 3 + 1) / 2 + 64;   // split planes
        if (n > need) need = n;
    }
    choose_igemm(M, rb, wm.G * wm.P * (cb / 32), 1, math, &tile, &splits);
    {
        long long n = splits > 1 ? (long long)splits * M * g.Cin : 0;
        if (math == 6 && rb % 64 == 0 && n < 256LL * 256 * 128 + 64) n = 256LL * 256 * 128 + 64;
        if (math == 6 && rb % 64 == 0) n += ((outM * g.Cout + (long long)wm.P * wm.G * rb * cb) * 3 + 1) / 2 + 64;
        if (n > need) need = n;
    }
    }
    int wt, S, mper, Np, Cp;
    choose_wgrad(M, g.Cout, g.Cin, wm.G, wm.P, &wt, &S, &mper, &Np, &Cp);
    long long n3 = (long long)wm.P * wm.G * S * Np * Cp + (long long)wm.P * S * g.Cout;   // partials + bias-gradient partials
    if (math == 6) {
        int S6, mper6;
        if (M >= 1024 && choose_wgrad6(M, g.Cout, g.Cin, wm.G, wm.P, &S6, &mper6) >= 0)
            // Part + planes of gy and x, then (shared-plane backward) the data-gradient's split-K partials + weight planes
            n3 = (long long)wm.P * wm.G * S6 * g.Cout * g.Cin + 4 + ((outM * g.Cout + M * g.Cin) * 3 + 1) / 2 + 64 +
                 256LL * 256 * 128 + 64 + ((long long)wm.P * wm.G * rb * cb * 3 + 1) / 2 + 64;
    }
    if (math != 6) {
        // the default fp32 path runs wgrad_ws_kernel where it tiles (choose_wgrad_ws: its own split counts, up to 6 rounds of blocks):
        // partials + the bias-gradient rows its loader waves leave (<= FG_WS_BIAS_ROWS_MAX rows) -- sized here explicitly, not by
        // the accident that fg_conv_scratch_floats takes the maximum with the bf16x6 bound
        const long long nws = fg_conv_wgrad_part_floats(g) + fg_conv_wgrad_bias_part_floats(g);
        if (nws > n3) n3 = nws;
    }
    if (n3 > need) need = n3;
    long long n4 = (long long)(CR_ROWBLOCKS_MAX + 2) * g.Cout;
    if (n4 > need) need = n4;
    // ragged channel counts: the zero-padded copies of the operands (pad_operand) at the tail of the scratch
    if (g.Cin % 4) need += M * fg_round_up(g.Cin, 4) + 8;
    if (g.Cout % 4) need += outM * fg_round_up(g.Cout, 4) + 8;
    return need + 64;
}
// upper bound over the math modes, so fg_set_math can be toggled on a live net
long long fg_conv_scratch_floats(const ConvGeom& g) {
    ConvGeom g1 = g; g1.stride = 1;        // sized as the stride-1 layer of the same input (an upper bound) ...
    const long long n0 = scratch_for_math(g1, 0), n6 = scratch_for_math(g1, 6);
    const long long nz = g.stride == 2 ? (long long)g.B * g.H * g.W * g.Cout + 64 : 0;   // ... + the zero-inserted gradient
    return (n0 > n6 ? n0 : n6) + nz;
}

// reference-formulation FLOPs of one pass over this layer (2 x MACs of the un-folded convolution)
static double alg_flops(const ConvGeom& g) {
    const double outpix = (double)g.B * g.H * g.W * (g.fold ? 4.0 : 1.0) / (g.stride == 2 ? 4.0 : 1.0);
    return 2.0 * outpix * g.Cout * g.Cin * g.k * g.k;
}
static const char* tag_of(const ConvGeom& g, int pass) {
    static const char* names[3][3] = {{"linear_fwd", "linear_dgrad", "linear_wgrad"},
                                      {"conv_fwd", "conv_dgrad", "conv_wgrad"},
                                      {"convup_fwd", "convup_dgrad", "convup_wgrad"}};
    const int kind = (g.k == 1 && g.H == 1 && g.W == 1) ? 0 : (g.fold ? 2 : 1);
    return names[kind][pass];
}

static inline int pack_off(int oy, int ox) { return (oy & 0xffff) | (ox << 16); }
static void fill_mspace(IgemmArgs& a, int B, int H, int W) {
    a.Nb = B; a.Hm = H; a.Wm = W; a.M = B * H * W;
    a.lgH = ilog2_exact(H); a.lgW = ilog2_exact(W);
    if (a.lgH < 0 || a.lgW < 0) a.lgH = a.lgW = -1;
}


// Channel counts that are not a multiple of 4 (a `--noiseDim 50` Linear, a 6-channel conv: train.lua's CLI accepts any, nn_utils.lua:35-39).
// The contraction kernels gather their A operand in 16-byte pieces, so such a tensor is first copied into rows of round_up(C, 4)
// floats with a zero tail (the packed weights are zero there as well); the copy lives at the TAIL of the caller's scratch.
__global__ void pad_channels_kernel(const float* __restrict__ src, long long rows, int C, int Cp, float* __restrict__ dst) {
    const long long n = rows * Cp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / Cp;
        const int c = (int)(i - r * Cp);
        dst[i] = c < C ? src[r * C + c] : 0.f;
    }
}
// returns the operand to use (src itself when C % 4 == 0); shrinks *scratch_floats by what the copy takes; *rc on error
static const float* pad_operand(fg_ctx* ctx, const float* src, long long rows, int C, float* scratch, long long* scratch_floats, int* Cp,
                                int* rc) {
    *rc = FG_OK; *Cp = C;
    if (C % 4 == 0) return src;
    *Cp = fg_round_up(C, 4);
    const long long need = (rows * *Cp + 3) / 4 * 4;
    const long long at = (*scratch_floats - need) / 4 * 4;          // 16-byte aligned inside a 16-byte aligned scratch
    if (at < 0) { *rc = fg_set_err(ctx, FG_ERR_WORKSPACE, "ragged channel count %d: scratch %lld < %lld", C, *scratch_floats, need); return nullptr; }
    float* dst = scratch + at;
    const long long n = rows * *Cp;
    const int blocks = (int)(n + 255) / 256 > 4096 ? 4096 : (int)((n + 255) / 256);
    hipLaunchKernelGGL(pad_channels_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, ctx->stream, src, rows, C, *Cp, dst);
    if (!g_fg_dry && hipGetLastError() != hipSuccess) { *rc = fg_set_err(ctx, FG_ERR_HIP, "pad_channels_kernel"); return nullptr; }
    *scratch_floats = at;
    return dst;
}

// Winograd geometry of a g.wino layer (WinoArgs, fg_internal.h); the tile grid is the 2 x 2 tiling of the H x W SOURCE map.
//   forward: 3x3: one group at (-1, -1);  5x5: four groups at (3a - 2, 3b - 2);  folded: one group, FOUR output parities
//            (parity (py, px) of tile element (a, b) is output pixel (2 (2 ty + a) + py, 2 (2 tx + b) + px) of the 2H x 2W map)
//   data gradient (the same convolution, taps flipped): 3x3 / 5x5: the same groups;  folded: four groups, group p reads the
//            parity-p sub-grid of the 2H x 2W gradient (stride 2, offset p - 2), one output parity on the source map
static void fill_wino(WinoArgs& w, const ConvGeom& g, int bwd) {
    const int f = g.fold ? 2 : 1;
    w.B = g.B;
    w.TH = g.H / 2; w.TW = g.W / 2; w.T = g.B * w.TH * w.TW;
    w.lgTH = ilog2_exact(w.TH); w.lgTW = ilog2_exact(w.TW);
    if (w.lgTH < 0 || w.lgTW < 0) w.lgTH = w.lgTW = -1;
    w.isy = w.isx = 1; w.KG = 1; w.goy[0] = w.gox[0] = -1;
    w.P = 1; w.osy = w.osx = 1; w.ooy[0] = w.oox[0] = 0;
    if (g.wino == 2) {
        w.KG = 4;
        for (int q = 0; q < 4; ++q) { w.goy[q] = (signed char)(3 * (q >> 1) - 2); w.gox[q] = (signed char)(3 * (q & 1) - 2); }
    }
    if (!bwd) {
        w.Hi = g.H; w.Wi = g.W; w.C = g.Cin; w.N = g.Cout; w.Ho = f * g.H; w.Wo = f * g.W;
        if (g.fold) {
            w.P = 4; w.osy = w.osx = 2;
            for (int q = 0; q < 4; ++q) { w.ooy[q] = (signed char)(q >> 1); w.oox[q] = (signed char)(q & 1); }
        }
    } else {
        w.Hi = f * g.H; w.Wi = f * g.W; w.C = g.Cout; w.N = g.Cin; w.Ho = g.H; w.Wo = g.W;
        if (g.fold) {
            w.KG = 4; w.isy = w.isx = 2;
            for (int q = 0; q < 4; ++q) { w.goy[q] = (signed char)((q >> 1) - 2); w.gox[q] = (signed char)((q & 1) - 2); }
        }
    }
    w.Npad = fg_round_up(w.N, 64);
    w.x_bytes = (long long)g.B * w.Hi * w.Wi * w.C * 4;
}

// bf16x6 math mode (fg_set_math): the wave-specialised kernel reads both operands as split-bf16 planes; build them in
// the (otherwise unused: tile 4 never splits K) scratch.  Weight planes are rebuilt per call -- a few MB, ~5 us.
static int maybe_split_operands(fg_ctx* ctx, IgemmArgs& a, int tile, long long packed_floats, float* scratch,
                                long long scratch_floats, const void* wp6, const void* a6_have, void* a6_dst,
                                int* a6_written) {
    if (a6_written) *a6_written = 0;
    if (ctx->math != 6 || tile != 4 || (a.Ca % 16) != 0 || (a.Kpad % 16) != 0) return FG_OK;
    if (a.splits > 1) {      // the split-K partials occupy the head of the scratch
        const long long used = (a.split_stride * a.splits + 3) / 4 * 4;
        scratch += used; scratch_floats -= used;
    }
    const long long a_floats = a.a_bytes / 4;
    const long long a6 = (a6_have || a6_dst) ? 0 : (a_floats * 3 + 1) / 2;
    const long long b6 = wp6 ? 0 : (packed_floats * 3 + 1) / 2;
    const long long a6_al = (a6 + 3) / 4 * 4;
    if (a6_al + b6 > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "bf16x6 planes: scratch %lld > %lld", a6_al + b6, scratch_floats);
    int rc;
    void* adst = a6_dst ? a6_dst : (void*)scratch;
    if (!a6_have && (rc = fg_launch_split_planes(ctx, a.A, a_floats / a.Ca, a.Ca, adst))) return rc;
    if (!wp6 && (rc = fg_launch_split_planes(ctx, a.Bp, packed_floats / a.Kpad, a.Kpad, scratch + a6_al))) return rc;
    a.A6 = a6_have ? a6_have : adst; a.B6 = wp6 ? wp6 : (const void*)(scratch + a6_al);
    if (a6_written && a6_dst) *a6_written = 1;
    return FG_OK;
}

int fg_conv_forward_run(fg_ctx* ctx, const ConvGeom& g, const float* x, const float* wp_fwd, const float* bias,
                        float* y, float* scratch, long long scratch_floats, const void* wp6, void* x6_dst, int* x6_written,
                        float* stats_part, long long stats_cap, int* stats_rows, const FgActFuse* act, FgSplitParts* leave) {
    if (leave) memset(leave, 0, sizeof(*leave));
    if (act) act->applied = 0;
    if (x6_written) *x6_written = 0;
    if (stats_rows) *stats_rows = 0;
    if (g.B == 0) return FG_OK;
    WeightMap wm; fg_geom_weightmap(g, &wm);
    int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
    IgemmArgs a; memset(&a, 0, sizeof(a));
    const int st = g.stride == 2 ? 2 : 1;
    if (st == 2 && (g.fold || (g.H & 1) || (g.W & 1))) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "stride-2 conv: even H/W, no folded upsample");
    if (g.wino) {
        // Winograd F(2x2, 3x3) (wino.hip): un-split launches take the bias, the PReLU behind the layer and the BatchNorm statistics in
        // their epilogue, split ones leave partials for the same passes as the implicit GEMM's
        WinoArgs w; memset(&w, 0, sizeof(w));
        fill_wino(w, g, 0);
        w.X = x; w.U = wp_fwd; w.bias = bias; w.Out = y;
        w.alg_flops = alg_flops(g); w.tag = tag_of(g, 0);
        const int splits = choose_wino_splits(w.T, w.P * w.Npad, w.KG * w.C);
        const long long out_count = (long long)g.B * w.Ho * w.Wo * g.Cout;
        w.splits = splits;
        if (splits > 1) {
            if ((long long)splits * out_count > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv fwd (winograd): scratch");
            w.Out = scratch; w.split_stride = out_count;
        }
        if (act && splits == 1 && act->y && act->slope && !act->mask && fg_fuse_prelu(ctx)) { w.act_y = act->y; w.act_slope = act->slope; }
        if (stats_part && stats_rows && splits == 1) {      // one partial row per (tile block, parity, wave row)
            const long long rows = 2LL * w.P * fg_cdiv(w.T, 64);
            if (2 * rows * g.Cout <= stats_cap) { w.stats_part = stats_part; w.stats_rows = (int)rows; *stats_rows = (int)rows; }
        }
        int rc = fg_launch_wino(ctx, w);
        if (rc) return rc;
        if (w.act_y) act->applied = 1;
        if (splits > 1 && leave && !act && out_count % 4 == 0 && g.Cout % 4 == 0) {
            leave->part = scratch; leave->splits = splits; leave->stride = out_count; leave->bias = bias; leave->N = g.Cout;
            return FG_OK;
        }
        if (splits > 1) return fg_launch_sum_splits(ctx, scratch, splits, out_count, bias, g.Cout, y, out_count, act);
        return FG_OK;
    }
    fill_mspace(a, g.B, g.H / st, g.W / st);                 // M-space = output pixels
    int Ca = g.Cin, rcp;
    x = pad_operand(ctx, x, (long long)g.B * g.H * g.W, g.Cin, scratch, &scratch_floats, &Ca, &rcp);
    if (rcp) return rcp;
    a.A = x; a.Bp = wp_fwd; a.bias = bias; a.Out = y;
    a.alg_flops = alg_flops(g); a.tag = tag_of(g, 0);
    a.Ha = g.H; a.Wa = g.W; a.Ca = Ca; a.Kpad = cf; a.asy = a.asx = 1;
    a.N = g.Cout; a.G = wm.G; a.Npad = rf;
    if (g.fold) {
        a.Ho = 2 * g.H; a.Wo = 2 * g.W; a.osy = a.osx = 2;
        for (int p = 0; p < 4; ++p) {
            a.ooy[p] = (signed char)(p >> 1); a.oox[p] = (signed char)(p & 1);
            for (int t = 0; t < wm.G; ++t) {
                a.goff[p][t] = pack_off(t / wm.T + wm.rmin, t % wm.T + wm.rmin);
            }
        }
    } else {
        a.Ho = g.H / st; a.Wo = g.W / st; a.osy = a.osx = 1;
        a.asy = a.asx = st;                                  // input pixel = st * output pixel + tap - pad
        for (int t = 0; t < wm.G; ++t) {
            a.goff[0][t] = pack_off(t / g.k - g.pad, t % g.k - g.pad);
        }
    }
    a.a_bytes = (long long)g.B * g.H * g.W * Ca * 4;
    int tile, splits;
    choose_igemm(a.M, rf, wm.G * (cf / 32), wm.P, ctx->math, &tile, &splits);
    if (tile == 4 && ctx->math == 6 && ((a.Ca % 16) || (a.Kpad % 16))) choose_igemm(a.M, rf, wm.G * (cf / 32), wm.P, 0, &tile, &splits);
    const long long out_count = (long long)a.M * (g.fold ? 4 : 1) * g.Cout;
    a.splits = splits;
    if (splits > 1) {
        if ((long long)splits * out_count > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv fwd: scratch");
        a.Out = scratch; a.split_stride = out_count;
    }
    int rc;
    if ((rc = maybe_split_operands(ctx, a, tile, (long long)wm.P * wm.G * rf * cf, scratch, scratch_floats, wp6, nullptr, x6_dst, x6_written))) return rc;
    if (stats_part && stats_rows && splits == 1 && !a.A6) {
        // BatchNorm statistics in the epilogue: one partial row per wave row of every (M-tile, parity)
        const int bm = tile >= 4 ? 256 : (tile == 2 ? 64 : 128);
        const int wrows = (tile == 5 || (tile == 4 && rf % 128 != 0)) ? 4 : 2;
        const long long rows = (long long)fg_cdiv(a.M, bm) * wm.P * wrows;
        if (2 * rows * g.Cout <= stats_cap) { a.stats_part = stats_part; a.stats_rows = (int)rows; *stats_rows = (int)rows; }
    }
    // the PReLU behind an un-split layer rides on the kernel's epilogue (a same-shape mask does not: split-K layers only)
    if (act && splits == 1 && !a.A6 && act->y && act->slope && !act->mask && fg_fuse_prelu(ctx)) {
        a.act_y = act->y; a.act_slope = act->slope;
    }
    if ((rc = fg_launch_igemm(ctx, a, wm.P, tile))) return rc;
    if (a.act_y) act->applied = 1;
    if (splits > 1 && leave && !act && out_count % 4 == 0 && g.Cout % 4 == 0) {
        leave->part = scratch; leave->splits = splits; leave->stride = out_count; leave->bias = bias; leave->N = g.Cout;
        return FG_OK;
    }
    if (splits > 1) return fg_launch_sum_splits(ctx, scratch, splits, out_count, bias, g.Cout, y, out_count, act);
    return FG_OK;
}

int fg_conv_dgrad_run(fg_ctx* ctx, const ConvGeom& g, const float* gy, const float* wp_bwd, float* gx, float* scratch,
                      long long scratch_floats, const void* wp6, const void* gy6, const FgActBwd* actb, FgSplitParts* leave) {
    if (leave) memset(leave, 0, sizeof(*leave));
    if (actb) actb->applied = 0;
    if (g.B == 0) return FG_OK;
    if (g.stride == 2) {
        // stride-2 data gradient = stride-1 data gradient of the zero-inserted output gradient (the layers that use it are
        // a few MFLOP: models.lua:289-291)
        const long long nz = ((long long)g.B * g.H * g.W * g.Cout + 3) / 4 * 4;
        if (nz > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv dgrad (stride 2): scratch");
        int rc0 = fg_launch_zero_insert2(ctx, gy, scratch, g.B, g.H / 2, g.W / 2, g.Cout);
        if (rc0) return rc0;
        ConvGeom g1 = g; g1.stride = 1;
        return fg_conv_dgrad_run(ctx, g1, scratch, wp_bwd, gx, scratch + nz, scratch_floats - nz, wp6, nullptr);
    }
    WeightMap wm; fg_geom_weightmap(g, &wm);
    int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
    if (g.wino) {
        // the data gradient of a 'same'-padded stride-1 layer is the same convolution with flipped, transposed taps: the same Winograd
        // kernel on the data-gradient pack (a folded layer: the four output parities become four K groups on the stride-2 sub-grids)
        WinoArgs w; memset(&w, 0, sizeof(w));
        fill_wino(w, g, 1);
        w.X = gy; w.U = wp_bwd; w.bias = nullptr; w.Out = gx;
        w.alg_flops = alg_flops(g); w.tag = tag_of(g, 1);
        const int splits = choose_wino_splits(w.T, w.P * w.Npad, w.KG * w.C);
        const long long out_count = (long long)g.B * g.H * g.W * g.Cin;
        w.splits = splits;
        if (splits > 1) {
            if ((long long)splits * out_count > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv dgrad (winograd): scratch");
            w.Out = scratch; w.split_stride = out_count;
        }
        long long nparts = 0;
        if (actb && actb->x && actb->slope && !actb->mask && splits == 1 && fg_fuse_prelu(ctx)) {
            nparts = 4 * fg_wino_blocks(w);
            float* dp = actb->gslope ? fg_defer_alloc(ctx, nparts) : nullptr;
            if (!actb->gslope || dp) { w.act_x = actb->x; w.act_slope = actb->slope; w.act_part = dp; }
        }
        int rc = fg_launch_wino(ctx, w);
        if (rc) return rc;
        if (w.act_x) {
            actb->applied = 1;
            if (w.act_part) fg_defer_push(ctx, w.act_part, (int)nparts, 1, 0.f, actb->gslope);
        }
        if (splits > 1 && leave && !actb && out_count % 4 == 0 && g.Cin % 4 == 0) {
            leave->part = scratch; leave->splits = splits; leave->stride = out_count; leave->bias = nullptr; leave->N = g.Cin;
            return FG_OK;
        }
        if (splits > 1) {
            if (actb && g.Cin % 4 == 0) return fg_launch_sum_splits_actbwd(ctx, scratch, splits, out_count, gx, out_count, actb);
            return fg_launch_sum_splits(ctx, scratch, splits, out_count, nullptr, g.Cin, gx, out_count);
        }
        return FG_OK;
    }
    IgemmArgs a; memset(&a, 0, sizeof(a));
    fill_mspace(a, g.B, g.H, g.W);
    int Ca = g.Cout, rcp;
    gy = pad_operand(ctx, gy, (long long)g.B * g.H * g.W * (g.fold ? 4 : 1), g.Cout, scratch, &scratch_floats, &Ca, &rcp);
    if (rcp) return rcp;
    a.A = gy; a.Bp = wp_bwd; a.bias = nullptr; a.Out = gx;
    a.alg_flops = alg_flops(g); a.tag = tag_of(g, 1);
    a.Ca = Ca; a.Kpad = cb;
    a.Ho = g.H; a.Wo = g.W; a.osy = a.osx = 1; a.N = g.Cin; a.Npad = rb;
    a.G = wm.G * wm.P;
    if (a.G > FG_MAX_GROUPS) return fg_set_err(ctx, FG_ERR_UNSUPPORTED, "conv dgrad: %d groups", a.G);
    if (g.fold) {
        a.Ha = 2 * g.H; a.Wa = 2 * g.W; a.asy = a.asx = 2;
        for (int p = 0; p < 4; ++p)
            for (int t = 0; t < wm.G; ++t) {
                const int ry = t / wm.T + wm.rmin, rx = t % wm.T + wm.rmin;
                a.goff[0][p * wm.G + t] = pack_off((p >> 1) - 2 * ry, (p & 1) - 2 * rx);
            }
    } else {
        a.Ha = g.H; a.Wa = g.W; a.asy = a.asx = 1;
        for (int t = 0; t < wm.G; ++t) {
            a.goff[0][t] = pack_off(g.pad - t / g.k, g.pad - t % g.k);
        }
    }
    a.a_bytes = (long long)g.B * g.H * g.W * (g.fold ? 4 : 1) * Ca * 4;
    int tile, splits;
    choose_igemm(a.M, rb, a.G * (cb / 32), 1, ctx->math, &tile, &splits);
    if (tile == 4 && ctx->math == 6 && ((a.Ca % 16) || (a.Kpad % 16))) choose_igemm(a.M, rb, a.G * (cb / 32), 1, 0, &tile, &splits);
    const long long out_count = (long long)a.M * g.Cin;
    a.splits = splits;
    if (splits > 1) {
        if ((long long)splits * out_count > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv dgrad: scratch");
        a.Out = scratch; a.split_stride = out_count;
    }
    int rc;
    if ((rc = maybe_split_operands(ctx, a, tile, (long long)a.G * rb * cb, scratch, scratch_floats, wp6, gy6, nullptr, nullptr))) return rc;
    // the backward of the PReLU in front of the layer in the epilogue (un-split fp32 launches); its slope-gradient partials
    // (4 per block) go through the batched deferred finals, so this needs the arena of an fg_net backward pass
    long long nparts = 0;
    if (actb && actb->x && actb->slope && !actb->mask && splits == 1 && !a.A6 && fg_fuse_prelu(ctx)) {
        nparts = 4 * fg_igemm_blocks(a, 1, tile);
        float* dp = actb->gslope ? fg_defer_alloc(ctx, nparts) : nullptr;
        if (!actb->gslope || dp) { a.act_x = actb->x; a.act_slope = actb->slope; a.act_part = dp; }
    }
    if ((rc = fg_launch_igemm(ctx, a, 1, tile))) return rc;
    if (a.act_x) {
        actb->applied = 1;
        if (a.act_part) fg_defer_push(ctx, a.act_part, (int)nparts, 1, 0.f, actb->gslope);
    }
    if (splits > 1 && leave && !actb && out_count % 4 == 0 && g.Cin % 4 == 0) {
        leave->part = scratch; leave->splits = splits; leave->stride = out_count; leave->bias = nullptr; leave->N = g.Cin;
        return FG_OK;
    }
    if (splits > 1) {
        // the pass that sums the partials also runs the backward of the PReLU [+ Dropout] in front of the layer
        if (actb && g.Cin % 4 == 0) return fg_launch_sum_splits_actbwd(ctx, scratch, splits, out_count, gx, out_count, actb);
        return fg_launch_sum_splits(ctx, scratch, splits, out_count, nullptr, g.Cin, gx, out_count);
    }
    return FG_OK;
}

int fg_conv_wgrad_run(fg_ctx* ctx, const ConvGeom& g, const float* x, const float* gy, float* gradW, float* gradb,
                      float beta, float* scratch, long long scratch_floats, const void* x6, const void** gy6_out,
                      long long* used_out) {
    if (gy6_out) *gy6_out = nullptr;
    if (used_out) *used_out = 0;
    if (g.B == 0) return FG_OK;
    WeightMap wm; fg_geom_weightmap(g, &wm);
    WgradArgs a; memset(&a, 0, sizeof(a));
    const int st = g.stride == 2 ? 2 : 1;
    const float* gy_ref = gy;                    // the bias gradient sums the caller's tensor (any channel count)
    int Nd = g.Cout, Cx = g.Cin, rcp;
    gy = pad_operand(ctx, gy, (long long)g.B * (g.H / st) * (g.W / st) * (g.fold ? 4 : 1), g.Cout, scratch, &scratch_floats, &Nd, &rcp);
    if (rcp) return rcp;
    x = pad_operand(ctx, x, (long long)g.B * g.H * g.W, g.Cin, scratch, &scratch_floats, &Cx, &rcp);
    if (rcp) return rcp;
    const bool ragged = Nd != g.Cout || Cx != g.Cin;
    a.dY = gy; a.X = x; a.Part = scratch;
    a.alg_flops = alg_flops(g); a.tag = tag_of(g, 2);
    a.Nb = g.B; a.Hm = g.H / st; a.Wm = g.W / st; a.M = g.B * a.Hm * a.Wm;          // M-space = output pixels
    a.lgH = ilog2_exact(a.Hm); a.lgW = ilog2_exact(a.Wm);
    if (a.lgH < 0 || a.lgW < 0) a.lgH = a.lgW = -1;
    a.Nd = Nd; a.Cx = Cx; a.Hx = g.H; a.Wx = g.W; a.xsy = a.xsx = st;
    a.G = wm.G;
    if (g.fold) {
        a.Hd = 2 * g.H; a.Wd = 2 * g.W; a.dsy = a.dsx = 2;
        for (int p = 0; p < 4; ++p) {
            a.doy[p] = (signed char)(p >> 1); a.dox[p] = (signed char)(p & 1);
            for (int t = 0; t < wm.G; ++t) {
                a.xoy[p][t] = (signed char)(t / wm.T + wm.rmin);
                a.xox[p][t] = (signed char)(t % wm.T + wm.rmin);
            }
        }
    } else {
        a.Hd = g.H / st; a.Wd = g.W / st; a.dsy = a.dsx = 1;
        for (int t = 0; t < wm.G; ++t) {
            a.xoy[0][t] = (signed char)(t / g.k - g.pad);
            a.xox[0][t] = (signed char)(t % g.k - g.pad);
        }
    }
    a.d_bytes = (long long)g.B * a.Hd * a.Wd * Nd * 4;
    a.x_bytes = (long long)g.B * g.H * g.W * Cx * 4;
    int tile, rc;
    int Sw, cpsw;
    if ((ctx->fusion & FG_FUSE_WINOGRAD_WGRAD) && ctx->math != 6 && !ragged && choose_wino_wgrad(g, &Sw, &cpsw)) {
        // Winograd-domain weight gradient (wino_wgrad.hip): 16 instead of 36 / 9 / 25-of-36 multiplies per tile and channel pair
        WinoArgs wf; memset(&wf, 0, sizeof(wf));
        fill_wino(wf, g, 0);
        WinoWgradArgs w; memset(&w, 0, sizeof(w));
        w.X = x; w.dY = gy; w.B = g.B; w.Hi = wf.Hi; w.Wi = wf.Wi; w.Cx = g.Cin; w.Ho = wf.Ho; w.Wo = wf.Wo; w.Nd = g.Cout;
        w.TH = wf.TH; w.TW = wf.TW; w.T = wf.T; w.lgTH = wf.lgTH; w.lgTW = wf.lgTW;
        w.isy = wf.isy; w.isx = wf.isx; w.KG = wf.KG; w.P = wf.P; w.osy = wf.osy; w.osx = wf.osx;
        memcpy(w.goy, wf.goy, 4); memcpy(w.gox, wf.gox, 4); memcpy(w.ooy, wf.ooy, 4); memcpy(w.oox, wf.oox, 4);
        w.S = Sw; w.chunks_per_split = cpsw; w.Npad = g.Cout; w.Cpad = g.Cin;
        w.x_bytes = a.x_bytes; w.d_bytes = a.d_bytes; w.alg_flops = a.alg_flops; w.tag = a.tag;
        const long long need = wino_wgrad_part_floats(g, Sw);
        if (need > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv wgrad (winograd): scratch %lld > %lld", need, scratch_floats);
        const int nrb = w.P * Sw;
        const long long nb = (long long)nrb * g.Cout;
        bool deferred = false;
        if (gradb) {
            float* dp = fg_defer_alloc(ctx, nb);          // inside fg_net backward: final batched at the end
            if (dp) { w.bias_part = dp; deferred = true; }
            else if (need + nb <= scratch_floats) w.bias_part = scratch + need;
        }
        float* wp = fg_defer_parks_w(ctx) ? fg_defer_alloc(ctx, need) : nullptr;    // inside fg_net backward: summed at the end
        w.Part = wp ? wp : scratch;
        if ((rc = fg_launch_wino_wgrad(ctx, w))) return rc;
        wm.wino = g.wino;                                  // the partials are Winograd-domain: the finish applies G^T . G and the tap scatter
        if (!(wp && fg_defer_push_wfinish(ctx, wm, w.Part, Sw, w.Npad, w.Cpad, beta, gradW)) &&
            (rc = fg_launch_wgrad_finish(ctx, wm, w.Part, Sw, w.Npad, w.Cpad, beta, gradW))) return rc;
        if (gradb && w.bias_part) {
            if (deferred) { fg_defer_push(ctx, w.bias_part, nrb, g.Cout, beta, gradb); return FG_OK; }
            return fg_launch_colsum_final(ctx, w.bias_part, nrb, g.Cout, beta, gradb);
        }
        if (gradb) {
            if ((long long)CR_ROWBLOCKS_MAX * g.Cout > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "bias grad: scratch");
            return fg_launch_colsum(ctx, gy_ref, (long long)a.M * (g.fold ? 4 : 1), g.Cout, beta, gradb, scratch);
        }
        return FG_OK;
    }
    int cfg6 = -1;
    if (ctx->math == 6 && g.Cout % 16 == 0 && g.Cin % 16 == 0 && a.M >= 1024)   // Linear / tiny maps: too few pixels to reduce over
        cfg6 = choose_wgrad6(a.M, g.Cout, g.Cin, wm.G, wm.P, &a.S, &a.m_per_split);
    if (cfg6 >= 0) {
        a.Npad = g.Cout; a.Cpad = g.Cin;
        const long long part = ((long long)wm.P * wm.G * a.S * a.Npad * a.Cpad + 3) / 4 * 4;
        const long long d_fl = a.d_bytes / 4, x_fl = a.x_bytes / 4;
        const long long d6 = ((d_fl * 3 + 1) / 2 + 3) / 4 * 4, x6n = x6 ? 0 : (x_fl * 3 + 1) / 2;
        if (part + d6 + x6n > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv wgrad (bf16x6): scratch %lld > %lld", part + d6 + x6n, scratch_floats);
        if ((rc = fg_launch_split_planes(ctx, gy, d_fl / g.Cout, g.Cout, scratch + part))) return rc;
        if (!x6 && (rc = fg_launch_split_planes(ctx, x, x_fl / g.Cin, g.Cin, scratch + part + d6))) return rc;
        a.D6 = scratch + part; a.X6 = x6 ? x6 : (const void*)(scratch + part + d6);
        if ((rc = fg_launch_wgrad6(ctx, a, wm.P, cfg6))) return rc;
        if (gy6_out) *gy6_out = a.D6;
        if (used_out) *used_out = part + d6;
    } else if (fg_wgrad_ws_on() && ctx->math != 6 && a.M >= fg_wgrad_ws_minm() && choose_wgrad_ws(a.M, g.Cout, g.Cin, wm.G, wm.P, &a.S, &a.m_per_split) >= 0 &&
               fg_wgrad_ws_shape_ok(a, choose_wgrad_ws(a.M, g.Cout, g.Cin, wm.G, wm.P, &a.S, &a.m_per_split))) {
        // wave-specialised fp32 weight gradient (256 x 128 / 128 x 256 channel tiles, or 128 x 64 with the K-step split over the
        // MFMA waves; one round of ~256 blocks); the bias gradient takes the separate column-sum pass at the end of this function
        const int cfgw = choose_wgrad_ws(a.M, g.Cout, g.Cin, wm.G, wm.P, &a.S, &a.m_per_split);
        a.Npad = g.Cout; a.Cpad = g.Cin;
        const long long need = (long long)wm.P * wm.G * a.S * a.Npad * a.Cpad;
        if (need > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv wgrad (ws): scratch %lld > %lld", need, scratch_floats);
        // bias gradient: the loader waves of the (X tile 0, tap 0) blocks leave per-channel sums of their dY rows
        const int nrb = wm.P * a.S * fg_wgrad_ws_bias_rows(a, cfgw);
        const long long nb = (long long)nrb * g.Cout;
        bool deferred = false;
        if (gradb && nrb <= FG_WS_BIAS_ROWS_MAX) {
            float* dp = fg_defer_alloc(ctx, nb);          // inside fg_net backward: final batched at the end
            if (dp) { a.bias_part = dp; deferred = true; }
            else if (need + nb <= scratch_floats) a.bias_part = scratch + need;
        }
        float* wp = fg_defer_parks_w(ctx) ? fg_defer_alloc(ctx, need) : nullptr;    // inside fg_net backward: summed at the end
        if (wp) a.Part = wp;
        if ((rc = fg_launch_wgrad_ws(ctx, a, wm.P, cfgw))) return rc;
        if (!(wp && fg_defer_push_wfinish(ctx, wm, a.Part, a.S, a.Npad, a.Cpad, beta, gradW)) &&
            (rc = fg_launch_wgrad_finish(ctx, wm, a.Part, a.S, a.Npad, a.Cpad, beta, gradW))) return rc;
        if (gradb && a.bias_part) {
            if (deferred) { fg_defer_push(ctx, a.bias_part, nrb, g.Cout, beta, gradb); return FG_OK; }
            return fg_launch_colsum_final(ctx, a.bias_part, nrb, g.Cout, beta, gradb);
        }
    } else {
        choose_wgrad(a.M, g.Cout, g.Cin, wm.G, wm.P, &tile, &a.S, &a.m_per_split, &a.Npad, &a.Cpad);
        const long long need = (long long)wm.P * wm.G * a.S * a.Npad * a.Cpad;
        if (need > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "conv wgrad: scratch %lld > %lld", need, scratch_floats);
        // bias gradient = column sums of gy over all output pixels: the contraction kernel leaves them per (parity, split)
        const int nrb = wm.P * a.S;
        const long long nb = (long long)nrb * g.Cout;
        bool deferred = false;
        if (gradb && !ragged) {                          // (ragged: the partial rows would be Nd wide -- the column-sum pass below)
            float* dp = fg_defer_alloc(ctx, nb);          // inside fg_net backward: final batched at the end
            if (dp) { a.bias_part = dp; deferred = true; }
            else if (need + nb <= scratch_floats) a.bias_part = scratch + need;
        }
        float* wp = fg_defer_parks_w(ctx) ? fg_defer_alloc(ctx, need) : nullptr;    // inside fg_net backward: summed at the end
        if (wp) a.Part = wp;
        if ((rc = fg_launch_wgrad(ctx, a, wm.P, tile))) return rc;
        if (!(wp && fg_defer_push_wfinish(ctx, wm, a.Part, a.S, a.Npad, a.Cpad, beta, gradW)) &&
            (rc = fg_launch_wgrad_finish(ctx, wm, a.Part, a.S, a.Npad, a.Cpad, beta, gradW))) return rc;
        if (gradb && a.bias_part) {
            if (deferred) { fg_defer_push(ctx, a.bias_part, nrb, g.Cout, beta, gradb); return FG_OK; }
            return fg_launch_colsum_final(ctx, a.bias_part, nrb, g.Cout, beta, gradb);
        }
    }
    if (cfg6 >= 0 && (rc = fg_launch_wgrad_finish(ctx, wm, scratch, a.S, a.Npad, a.Cpad, beta, gradW))) return rc;
    if (gradb) {
        // bias grad = column sums of gy over all output pixels
        const long long rows = (long long)a.M * (g.fold ? 4 : 1);
        if ((long long)CR_ROWBLOCKS_MAX * g.Cout > scratch_floats) return fg_set_err(ctx, FG_ERR_WORKSPACE, "bias grad: scratch");
        if ((rc = fg_launch_colsum(ctx, gy_ref, rows, g.Cout, beta, gradb, scratch))) return rc;
    }
    return FG_OK;
}
