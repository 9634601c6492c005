// fg_net: an nn.Sequential (models.lua:57-81 G, :382-416 D) compiled into a device plan.
// The reference walks its module list calling one or more native kernels per module
// (nn.Sequential:updateOutput / :backward).  Here adjacent modules are fused into stages:
//   UpSamplingNearest(2)+Conv5x5  -> one tap-folded MFMA conv (4 output parities x 3x3 taps on the source)
//   BatchNorm+PReLU               -> stats + one apply pass;  PReLU+SpatialDropout+AvgPool -> one pass
//   PReLU+Dropout, Conv(->3)+Sigmoid, Linear(->1)+Sigmoid
// Activations are NHWC and stay in the caller's workspace between forward and backward.
#include "fg_internal.h"
#include "conv_ops.h"
#include "../../include/facegen_hip.h"
#include <vector>
#include <algorithm>
#include <utility>
#include <string.h>

enum StageKind {
    ST_CONV = 1,      // MFMA conv (fold optional) or Linear (H=W=1)
    ST_GEMV,          // Linear(K->1) [+Sigmoid]
    ST_THIN_IN,       // conv with <=4 input channels
    ST_THIN_OUT,      // conv with <=4 output channels [+Sigmoid]
    ST_BNPRELU,       // SpatialBatchNormalization [+PReLU]
    ST_PRELU,         // PReLU [+Dropout]
    ST_ACTPOOL,       // PReLU + SpatialDropout + AvgPool2
    ST_ACTMAXPOOL,    // PReLU + SpatialMaxPooling(2, 2) [+ Dropout] (models_c2f.lua:245-246, 251-253): one pass forward, one backward
    ST_SIGMOID,
    ST_LEAKYRELU,
    ST_UPSAMPLE,
    ST_AVGPOOL,
    ST_SDROPOUT,
    ST_MAXPOOL,
    ST_DROPOUT,       // standalone nn.Dropout (any shape)
    ST_REVIEW         // the flat NCHW re-view behind a SpatialConvolutionUpsample with factor > 1
};

struct Stage {
    int kind = 0;
    int first_layer = 0, last_layer = 0;
    int ic = 0, ih = 0, iw = 0, oc = 0, oh = 0, ow = 0;
    long long w_off = -1, w_n = 0, b_off = -1, b_n = 0;
    long long slope_off = -1, gamma_off = -1, beta_off = -1, buf_off = -1;
    ConvGeom geom{};
    float* wp_fwd = nullptr;
    float* wp_bwd = nullptr;
    const void* wp_fwd6 = nullptr;   // the same packed weights as split-bf16 planes (bf16x6 math mode), views into fg_net::planes_all
    const void* wp_bwd6 = nullptr;
    float* bias_packed = nullptr;  // Linear followed by View(C,H,W): bias in NHWC feature order
    int has_prelu = 0, has_sigmoid = 0;
    int mask_idx = -1, mask_kind = 0;  // 1 spatial [B][C], 2 elementwise
    float p = 0.f, eps = 1e-5f, momentum = 0.1f, negslope = 0.333f;
    int factor = 1;           // ST_REVIEW
    // per-forward plan
    long long out_off = 0, aux_off = 0;
    long long x6_off = -1;    // ST_CONV: split-bf16 planes of the stage input, kept from forward for the weight gradient
    int x6_valid = 0;
    long long stats_off = -1, stats_cap = 0;   // ST_BNPRELU behind an ST_CONV: statistics partials left by the conv's epilogue
    int stats_rows = 0;       //   rows the last forward's conv actually wrote (0: the statistics pass over x runs)
    int act_done = 0;         // ST_PRELU: this forward's producer already applied it (folded into its split-K sum)
};

struct LayerInfo {
    long long w_off = -1, w_n = 0, b_off = -1, b_n = 0;
    int stage = -1;
    int ends_stage = 0;
};

struct fg_net {
    fg_ctx* ctx = nullptr;
    std::vector<Stage> st;
    std::vector<LayerInfo> layers;
    int in_c = 0, in_h = 0, in_w = 0;
    long long n_params = 0, n_buffers = 0;
    int n_masks = 0;
    std::vector<int> mask_stage;
    float *params = nullptr, *grads = nullptr, *buffers = nullptr;
    bool dirty = true;
    // plan of the last forward
    int plan_batch = -1;
    long long grad_off[2] = {0, 0}, tmp_off = 0, scratch_off = 0, scratch_floats = 0, total_floats = 0;
    const float* const* masks = nullptr;
    std::vector<const float*> mask_ptrs;
    int last_train = 1;
    // ranged backward state (fg_net_backward_range)
    const float* bwd_gcur = nullptr;
    int bwd_pp = 0, bwd_next = -1;
    // sync-BN: forward / backward pause at every BatchNorm for a cross-rank all-reduce of its fp64 sums
    bool sync_bn = false;
    double* sync_buf = nullptr;
    long long sync_cap = 0, sync_count = 0;
    int run_stage = 0, run_phase = 0;          // where a paused pass resumes
    int run_B = 0, run_train = 0, run_to = 0, run_flags = 0;
    const float* run_x = nullptr;
    const float* run_cur = nullptr;
    float* run_ws = nullptr;
    float* run_gx = nullptr;
    int run_pp = 0;
    // one-launch weight re-pack
    PackJob* jobs_dev = nullptr;
    FgDefer defer;                    // deferred final reductions of a backward pass (arena inside the workspace)
    FgWFinishJob wjobs[FG_DEFER_WMAX];  // its queued weight-gradient reductions
    long long defer_off = 0, defer_floats = 0;
    float* packed_all = nullptr;      // packed weights of every contraction stage, contiguous (one split launch)
    unsigned char* planes_all = nullptr;
    long long packed_total = 0;
    bool planes_valid = false;
    float* out_override = nullptr;    // fg_net_forward_to: the last stage writes here instead of into the workspace
    FgSplitParts pend{};              // split-K partials the previous stage left for this one (ST_ACTPOOL sums them itself)
    int n_jobs = 0;                   // pack jobs
    long long jobs_total = 0;
    int n_jobs_all = 0;               // ... + the update-only jobs of the fused optimizer launch
    long long jobs_total_all = 0;
    bool adam_fusable = false;
    long long pack_lds_floats = 64;   // dynamic shared memory of the re-pack launch: the largest staging area any of its jobs needs
    bool park_w = true;               // FG_FUSE_WFINISH_BATCH at creation: the workspace reserves room for parked weight-gradient partials
};

static inline long long align64(long long v) { return (v + 63) / 64 * 64; }

static long long stage_scratch(const Stage& s, int B) {
    long long need = 4096;
    switch (s.kind) {
        case ST_CONV: { ConvGeom g = s.geom; g.B = B; need = fg_conv_scratch_floats(g); break; }
        case ST_BNPRELU: need = (long long)3 * CR_ROWBLOCKS_MAX * s.oc + 2 * s.oc + 64; break;
        case ST_THIN_IN: case ST_THIN_OUT: {
            const int cw = s.kind == ST_THIN_IN ? s.oc : s.ic, cs = s.kind == ST_THIN_IN ? s.ic : s.oc;
            const long long na = (long long)s.geom.k * s.geom.k * cs;
            need = (FG_THIN_WGRAD_BLOCKS + 1) * (na + 1) * cw + (long long)CR_ROWBLOCKS_MAX * (s.oc > 64 ? s.oc : 64) + 64;   // na + 1: the bias row
            const long long nr = (long long)B * s.ih * s.iw * 32 + 64;      // R of the two-pass 5x5 / 7x7 thin-output forward
            if (s.kind == ST_THIN_OUT && s.geom.k >= 5 && nr > need) need = nr;
            break;
        }
        default: break;
    }
    return need;
}

static void make_plan(fg_net* n, int B) {
    long long off = 0, maxact = (long long)B * n->in_c * n->in_h * n->in_w, maxscr = 4096;
    for (auto& s : n->st) {
        const long long osz = (long long)B * s.oc * s.oh * s.ow;
        s.out_off = off; off += align64(osz);
        s.aux_off = off;
        if (s.kind == ST_BNPRELU) off += align64(2 * s.oc);
        if (s.kind == ST_BNPRELU && &s != &n->st[0] && (&s)[-1].kind == ST_CONV) {
            s.stats_cap = 2 * ((long long)B * s.ih * s.iw / 32 + 16) * s.ic;     // >= 2 x (wave rows of any tiling) x C
            s.stats_off = off; off += align64(s.stats_cap);
        }
        if (s.kind == ST_CONV && s.ic % 16 == 0) { s.x6_off = off; off += align64(((long long)B * s.ic * s.ih * s.iw * 3 + 1) / 2); }
        if (osz > maxact) maxact = osz;
        const long long isz = (long long)B * s.ic * s.ih * s.iw;
        if (isz > maxact) maxact = isz;
        const long long sc = stage_scratch(s, B);
        if (sc > maxscr) maxscr = sc;
    }
    {   // arena for the deferred finals: [row blocks][C] partials of every bias-gradient / slope-gradient reduction
        long long dn = 0;
        //  the weight-gradient partials are only parked here while FG_FUSE_WFINISH_BATCH is on, and only the first
        // FG_DEFER_WMAX layers of a pass can defer their reduction: nothing is reserved beyond that (134 MB for the 65536 -> 512
        // Linear of models_c2f.lua:262 alone).  Decided ONCE, when the net is created (fg_net::park_w), so that the workspace size a
        // host queried stays valid whatever fg_set_fusion does later: a bit switched on afterwards finds no arena and falls back to
        // the per-layer reduction (fg_defer_alloc returns null), it never overruns and never asks for more.
        const bool park_w = n->park_w;
        //  the backward pass walks the stages in REVERSE, so the layers that park are the LAST FG_DEFER_WMAX
        // parametrised convolutions / Linears of the plan, not the first
        int nconv = 0, seen = 0;
        for (auto& s : n->st) if (s.kind == ST_CONV && s.w_n > 0) ++nconv;
        for (auto& s : n->st) {
            const bool parks = s.kind == ST_CONV && s.w_n > 0 && park_w && (nconv - seen++) <= FG_DEFER_WMAX;
            // (ST_CONV: the wave-specialised weight gradient leaves one bias partial row per (parity, split, tap, X tile, loader
            // pixel lane) -- up to FG_WS_BIAS_ROWS_MAX rows; sized exactly)
            if (s.kind == ST_CONV) { ConvGeom g = s.geom; g.B = B; dn += (fg_conv_wgrad_bias_part_floats(g) + 63) / 64 * 64; }
            // ... and its split-K / parity partials stay until the batched weight-gradient reduction at the end of the pass
            if (parks) { ConvGeom g = s.geom; g.B = B; dn += (fg_conv_wgrad_part_floats(g) + 63) / 64 * 64; }
            else if (s.kind == ST_THIN_IN || s.kind == ST_THIN_OUT || s.kind == ST_GEMV) dn += (long long)CR_ROWBLOCKS_MAX * s.oc + 64;
            // (the sliced Linear(K -> 1) backward: weight-gradient rows, bias and slope partials per 16-sample slice -- fg_launch_gemv_backward)
            if (s.kind == ST_GEMV) dn += (long long)fg_cdiv(B, 16) * (s.ic + 4 + fg_cdiv(s.ic, 64)) + 192;
            if (s.has_prelu || s.kind == ST_PRELU || s.kind == ST_ACTPOOL) dn += 1024 + 64;
            // a PReLU whose backward rides on the epilogue of the neighbouring contraction leaves 4 partials per block
            if (s.kind == ST_PRELU && s.mask_kind == 0)
                dn += 4LL * fg_cdiv((long long)B * s.ih * s.iw, 64) * fg_cdiv(s.ic, 64) + 8192LL * fg_cdiv(s.ic, 64) + 64;
        }
        n->defer_off = off; n->defer_floats = dn; off += align64(dn);
    }
    n->grad_off[0] = off; off += align64(maxact);
    n->grad_off[1] = off; off += align64(maxact);
    n->tmp_off = off; off += align64(maxact);
    n->scratch_off = off; off += align64(maxscr);
    n->scratch_floats = maxscr;
    n->total_floats = off;
    n->plan_batch = B;
}

static int build_pack_jobs(fg_net* n);
static int ensure_packed(fg_net* n);

static int backward_run_stages(fg_net* n) {
    fg_ctx* ctx = n->ctx;
    const int B = n->run_B, flags = n->run_flags, stage_to = n->run_to;
    const float* x = n->run_x;
    float* gx = n->run_gx;
    const bool want_p = (flags & FG_BWD_PARAM_GRADS) != 0, want_x = (flags & FG_BWD_INPUT_GRAD) != 0;
    float* ws = n->run_ws;
    float* scratch = ws + n->scratch_off;
    float* tmp = ws + n->tmp_off;
    const float* P = n->params;
    float* Gp = n->grads;
    const float* gcur = n->bwd_gcur;
    int pp = n->bwd_pp, rc = FG_OK;
    bool prelu_folded = false;       // the previous (= next deeper) stage's kernel already ran this PReLU's backward
    for (int si = n->run_stage; si >= stage_to; --si) {
        Stage& s = n->st[si];
        if (prelu_folded) { prelu_folded = false; continue; }      // gcur already is the gradient wrt the PReLU's input
        if (s.kind != ST_ACTPOOL) n->pend.splits = 0;
        // a plain PReLU directly in front of this stage (inside this call's range, not the net's first stage): its backward
        // can ride on the epilogue of the kernel that produces this stage's input gradient
        FgActBwd actb; memset(&actb, 0, sizeof(actb));
        const bool pf = si >= 2 && si - 1 >= stage_to && n->st[si - 1].kind == ST_PRELU && n->st[si - 1].mask_kind != 1 &&
                        fg_fuse_prelu(ctx);
        if (pf) {
            const Stage& ps = n->st[si - 1];
            actb.x = ws + n->st[si - 2].out_off; actb.slope = P + ps.slope_off;
            actb.gslope = want_p ? Gp + ps.slope_off : nullptr;
            if (ps.mask_kind == 2) { actb.mask = n->mask_ptrs[ps.mask_idx]; actb.mscale = 1.f / (1.f - ps.p); }   // PReLU + Dropout
        }
        const float* xin = si == 0 ? x : ws + n->st[si - 1].out_off;
        const float* yout = (si + 1 == (int)n->st.size() && n->out_override) ? n->out_override : ws + s.out_off;
        const bool need_gx = si > 0 || want_x;
        float* gxb = si == 0 ? gx : ws + n->grad_off[pp];
        if (!need_gx) gxb = nullptr;
        const float* mask = (s.mask_idx >= 0) ? n->mask_ptrs[s.mask_idx] : nullptr;
        switch (s.kind) {
            case ST_CONV: {
                ConvGeom g = s.geom; g.B = B;
                const void* gy6 = nullptr;       // bf16x6: planes of gcur shared by the weight- and data-gradient
                long long gy6_used = 0;
                if (want_p) {
                    rc = fg_conv_wgrad_run(ctx, g, xin, gcur, Gp + s.w_off, s.bias_packed ? nullptr : Gp + s.b_off, 0.f,
                                           scratch, n->scratch_floats, s.x6_valid ? (const void*)(ws + s.x6_off) : nullptr,
                                           &gy6, &gy6_used);
                    if (!rc && s.bias_packed) {  // bias grad in NHWC feature order -> reference order
                        if (B <= 1024) rc = fg_launch_colsum_perm(ctx, gcur, B, g.Cout, g.o_c, g.o_hw, Gp + s.b_off);
                        else {
                            float* tb = scratch + (long long)CR_ROWBLOCKS_MAX * g.Cout;
                            FgDefer* dsv = ctx->defer; ctx->defer = nullptr;      // `tb` is consumed right away: no deferral
                            rc = fg_launch_colsum(ctx, gcur, (long long)B, g.Cout, 0.f, tb, scratch);
                            ctx->defer = dsv;
                            if (!rc) rc = fg_launch_nhwc_to_nchw(ctx, tb, Gp + s.b_off, 1, g.o_c, g.o_hw, 1);
                        }
                    }
                }
                // the fused PReLU + SpatialDropout + AvgPool backward in front of this layer sums split-K partials itself
                const bool ap = si >= 1 && si - 1 >= stage_to && n->st[si - 1].kind == ST_ACTPOOL && fg_fuse_prelu(ctx);
                n->pend.splits = 0;
                if (!rc && need_gx) rc = fg_conv_dgrad_run(ctx, g, gcur, s.wp_bwd, gxb, scratch + gy6_used, n->scratch_floats - gy6_used,
                                                         n->planes_valid ? s.wp_bwd6 : nullptr, gy6, pf ? &actb : nullptr,
                                                         ap ? &n->pend : nullptr);
                prelu_folded = pf && !rc && actb.applied;
                break;
            }
            case ST_GEMV:
                rc = fg_launch_gemv_backward(ctx, xin, P + s.w_off, yout, gcur, gxb, want_p ? Gp + s.w_off : nullptr,
                                             want_p ? Gp + s.b_off : nullptr, 0.f, B, s.ic, s.has_sigmoid, pf ? &actb : nullptr);
                prelu_folded = pf && !rc && actb.applied;
                break;
            case ST_THIN_IN: {
                const int k = s.geom.k;
                if (want_p) {
                    // slabs of k*k*Cin + 1 rows: the extra row is the bias gradient (the ones column of the matrix-pipe kernel)
                    float* gw = scratch + (long long)FG_THIN_WGRAD_BLOCKS * (k * k * s.ic + 1) * s.oc;
                    int bias_done = 0, unpacked = 0;
                    rc = fg_launch_thin_wgrad(ctx, xin, gcur, gw, B, s.ih, s.iw, s.ic, s.oc, k, +1, scratch, Gp + s.b_off, &bias_done,
                                              Gp + s.w_off, 0, &unpacked);
                    if (!rc && !unpacked) rc = fg_launch_thin_unpack_grad(ctx, gw, Gp + s.w_off, s.geom.Cout, s.geom.Cin, k, 0, 0.f);
                    if (!rc && !bias_done) rc = fg_launch_colsum(ctx, gcur, (long long)B * s.oh * s.ow, s.oc, 0.f, Gp + s.b_off, scratch);
                }
                if (!rc && need_gx)
                    rc = fg_launch_thin_out_conv(ctx, gcur, s.wp_fwd, nullptr, gxb, B, s.ih, s.iw, s.oc, s.ic, k, 1, 0);
                break;
            }
            case ST_THIN_OUT: {
                const int k = s.geom.k;
                const float* gpre = gcur;
                if (s.has_sigmoid) {
                    rc = fg_launch_sigmoid_backward(ctx, yout, gcur, tmp, (long long)B * s.oc * s.oh * s.ow);
                    gpre = tmp;
                }
                if (!rc && want_p) {
                    float* gw = scratch + (long long)FG_THIN_WGRAD_BLOCKS * k * k * s.oc * s.ic;
                    int unpacked = 0;
                    rc = fg_launch_thin_wgrad(ctx, gpre, xin, gw, B, s.ih, s.iw, s.oc, s.ic, k, -1, scratch, nullptr, nullptr,
                                              Gp + s.w_off, 1, &unpacked);
                    if (!rc && !unpacked) rc = fg_launch_thin_unpack_grad(ctx, gw, Gp + s.w_off, s.geom.Cout, s.geom.Cin, k, 1, 0.f);
                    if (!rc) rc = fg_launch_colsum(ctx, gpre, (long long)B * s.oh * s.ow, s.oc, 0.f, Gp + s.b_off, scratch);
                }
                if (!rc && need_gx) {
                    rc = fg_launch_thin_in_conv(ctx, gpre, s.wp_fwd, nullptr, gxb, B, s.ih, s.iw, s.oc, s.ic, k, 1, nullptr,
                                                pf ? &actb : nullptr, scratch, n->scratch_floats);
                    prelu_folded = pf && !rc && actb.applied;
                }
                break;
            }
            case ST_BNPRELU: {
                BnBwdArgs a; memset(&a, 0, sizeof(a));
                a.x = xin; a.gy = gcur; a.gx = gxb; a.M = (long long)B * s.ih * s.iw; a.C = s.ic;
                a.gamma = P + s.gamma_off; a.beta = P + s.beta_off; a.slope = s.has_prelu ? P + s.slope_off : nullptr;
                a.mean = ws + s.aux_off; a.invstd = ws + s.aux_off + s.ic;
                a.ggamma = want_p ? Gp + s.gamma_off : nullptr; a.gbeta = want_p ? Gp + s.beta_off : nullptr;
                a.gslope = (want_p && s.has_prelu) ? Gp + s.slope_off : nullptr; a.gbeta_acc = 0.f; a.scratch = scratch;
                if (!n->sync_bn) { rc = fg_launch_bn_backward(ctx, a); break; }
                if (n->run_phase == 0) {           // local fp64 sums -> pause for the cross-rank all-reduce
                    if (2LL * s.ic + 1 > n->sync_cap) return fg_set_err(ctx, FG_ERR_WORKSPACE, "sync-BN buffer too small");
                    if ((rc = fg_launch_bn_backward_sync1(ctx, a, n->sync_buf))) return rc;
                    n->run_stage = si; n->run_phase = 2; n->bwd_gcur = gcur; n->bwd_pp = pp; n->sync_count = 2LL * s.ic + 1;
                    return FG_PAUSED_SYNC;
                }
                n->run_phase = 0;
                rc = fg_launch_bn_backward_sync2(ctx, a, n->sync_buf);
                break;
            }
            case ST_PRELU: {
                const long long cnt = (long long)B * s.ic * s.ih * s.iw;
                const float sc = (s.mask_kind == 2) ? 1.f / (1.f - s.p) : 1.f;
                rc = fg_launch_prelu_backward(ctx, xin, gcur, P + s.slope_off, mask, sc, gxb,
                                              want_p ? Gp + s.slope_off : nullptr, 0.f, cnt, scratch);
                break;
            }
            case ST_ACTPOOL: {
                // (pending partials sit at the head of the scratch: the slope-gradient partials go behind them)
                const long long used = n->pend.splits ? ((n->pend.part - scratch) + (long long)n->pend.splits * n->pend.stride + 3) / 4 * 4 : 0;
                if (used + 1024 > n->scratch_floats) { rc = fg_set_err(ctx, FG_ERR_WORKSPACE, "actpool backward: scratch"); break; }
                rc = fg_launch_actpool_backward(ctx, xin, gcur, P + s.slope_off, mask, 1.f, gxb,
                                                want_p ? Gp + s.slope_off : nullptr, 0.f, B, s.ih, s.iw, s.ic, scratch + used,
                                                n->pend.splits ? &n->pend : nullptr);
                n->pend.splits = 0;
                break;
            }
            case ST_SIGMOID:
                if (need_gx) rc = fg_launch_sigmoid_backward(ctx, yout, gcur, gxb, (long long)B * s.ic * s.ih * s.iw);
                break;
            case ST_LEAKYRELU:
                if (need_gx) rc = fg_launch_leakyrelu_backward(ctx, xin, gcur, s.negslope, gxb, (long long)B * s.ic * s.ih * s.iw);
                break;
            case ST_UPSAMPLE: if (need_gx) rc = fg_launch_upsample_backward(ctx, gcur, gxb, B, s.ih, s.iw, s.ic); break;
            case ST_REVIEW: if (need_gx) rc = fg_launch_nchw_review(ctx, gcur, gxb, B, s.ih, s.iw, s.ic, s.factor, 1); break;
            case ST_AVGPOOL: if (need_gx) rc = fg_launch_avgpool_backward(ctx, gcur, gxb, B, s.ih, s.iw, s.ic); break;
            case ST_SDROPOUT: if (need_gx) rc = fg_launch_scale_mask_nc(ctx, gcur, mask, 1.f, gxb, B, s.ih * s.iw, s.ic); break;
            case ST_ACTMAXPOOL:
                // (also when only the slope gradient is wanted -- the fused stage can be the net's first one behind a
                //  no-op View, where no input gradient exists)
                if (need_gx || want_p)
                    rc = fg_launch_maxpool_prelu_backward(ctx, xin, gcur, P + s.slope_off, need_gx ? gxb : nullptr, want_p ? Gp + s.slope_off : nullptr, B, s.ih,
                                                          s.iw, s.ic, scratch, mask, s.mask_kind ? 1.f / (1.f - s.p) : 1.f);
                break;
            case ST_MAXPOOL:
                if (need_gx && pf && !actb.mask) {     // pooled tensor = prelu(actb.x): both backward passes in one
                    rc = fg_launch_maxpool_prelu_backward(ctx, actb.x, gcur, actb.slope, gxb, actb.gslope, B, s.ih, s.iw, s.ic, scratch);
                    prelu_folded = !rc;
                } else if (need_gx) rc = fg_launch_maxpool_backward(ctx, xin, gcur, gxb, B, s.ih, s.iw, s.ic);
                break;
            case ST_DROPOUT:
                if (need_gx) rc = fg_launch_mul_mask(ctx, gcur, mask, 1.f / (1.f - s.p), gxb, (long long)B * s.ic * s.ih * s.iw);
                break;
            default: rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: bad stage kind %d", s.kind);
        }
        if (rc) return rc;
        gcur = gxb;
        if (si > 0) pp ^= 1;
    }
    n->bwd_gcur = gcur; n->bwd_pp = pp; n->bwd_next = stage_to - 1; n->run_phase = 0;
    return FG_OK;
}

static int forward_run(fg_net* n, long long* out_offset) {
    fg_ctx* ctx = n->ctx;
    const int B = n->run_B, train = n->run_train;
    float* ws = n->run_ws;
    float* scratch = ws + n->scratch_off;
    const float* cur = n->run_cur;
    const float* P = n->params;
    int rc = FG_OK;
    for (int si = n->run_stage; si < (int)n->st.size(); ++si) {
        Stage& s = n->st[si];
        if (s.kind != ST_ACTPOOL) n->pend.splits = 0;
        float* y = (si + 1 == (int)n->st.size() && n->out_override) ? n->out_override : ws + s.out_off;
        const float* mask = (s.mask_idx >= 0) ? n->mask_ptrs[s.mask_idx] : nullptr;
        switch (s.kind) {
            case ST_CONV: {
                ConvGeom g = s.geom; g.B = B;
                s.x6_valid = 0;
                // a BatchNorm directly behind this convolution takes its batch statistics from the epilogue
                Stage* bn = (train && !n->sync_bn && si + 1 < (int)n->st.size() && n->st[si + 1].kind == ST_BNPRELU &&
                             n->st[si + 1].stats_off >= 0) ? &n->st[si + 1] : nullptr;
                if (bn) bn->stats_rows = 0;
                // a PReLU [+ Dropout] directly behind a split-K layer (the Linear layers) rides on the pass that sums the splits
                FgActFuse act; memset(&act, 0, sizeof(act));
                Stage* pr = (si + 1 < (int)n->st.size() && n->st[si + 1].kind == ST_PRELU) ? &n->st[si + 1] : nullptr;
                if (pr) {
                    const bool dm = pr->mask_kind == 2 && train;
                    act.slope = P + pr->slope_off;
                    act.y = (si + 2 == (int)n->st.size() && n->out_override) ? n->out_override : ws + pr->out_off;
                    act.mask = dm ? n->mask_ptrs[pr->mask_idx] : nullptr; act.mscale = dm ? 1.f / (1.f - pr->p) : 1.f;
                    pr->act_done = 0;
                }
                rc = fg_conv_forward_run(ctx, g, cur, s.wp_fwd, s.bias_packed ? s.bias_packed : P + s.b_off, y, scratch,
                                         n->scratch_floats, n->planes_valid ? s.wp_fwd6 : nullptr,
                                         (train && s.x6_off >= 0) ? (void*)(ws + s.x6_off) : nullptr, &s.x6_valid,
                                         bn ? ws + bn->stats_off : nullptr, bn ? bn->stats_cap : 0, bn ? &bn->stats_rows : nullptr,
                                         pr ? &act : nullptr,
                                         (si + 1 < (int)n->st.size() && n->st[si + 1].kind == ST_ACTPOOL && fg_fuse_prelu(ctx)) ? &n->pend : nullptr);
                if (pr && !rc) pr->act_done = act.applied;
                break;
            }
            case ST_GEMV:
                rc = fg_launch_gemv_forward(ctx, cur, P + s.w_off, P + s.b_off, y, B, s.ic, s.has_sigmoid);
                break;
            case ST_THIN_IN: {
                // a plain PReLU directly behind rides on the MFMA kernel's epilogue
                FgActFuse act; memset(&act, 0, sizeof(act));
                Stage* pr = (si + 1 < (int)n->st.size() && n->st[si + 1].kind == ST_PRELU && n->st[si + 1].mask_kind == 0)
                                ? &n->st[si + 1] : nullptr;
                if (pr) {
                    act.slope = P + pr->slope_off; act.mscale = 1.f; pr->act_done = 0;
                    act.y = (si + 2 == (int)n->st.size() && n->out_override) ? n->out_override : ws + pr->out_off;
                }
                rc = fg_launch_thin_in_conv(ctx, cur, s.wp_fwd, P + s.b_off, y, B, s.ih, s.iw, s.ic, s.oc, s.geom.k, 0,
                                            pr ? &act : nullptr, nullptr);
                if (pr && !rc) pr->act_done = act.applied;
                break;
            }
            case ST_THIN_OUT:
                rc = fg_launch_thin_out_conv(ctx, cur, s.wp_fwd, P + s.b_off, y, B, s.ih, s.iw, s.ic, s.oc, s.geom.k, 0,
                                             s.has_sigmoid, scratch, n->scratch_floats);
                break;
            case ST_BNPRELU: {
                const bool sync = n->sync_bn && train;
                BnArgs a; memset(&a, 0, sizeof(a));
                a.x = cur; a.y = y; a.M = (long long)B * s.ih * s.iw; a.C = s.ic;
                a.gamma = P + s.gamma_off; a.beta = P + s.beta_off; a.slope = s.has_prelu ? P + s.slope_off : nullptr;
                a.mean = ws + s.aux_off; a.invstd = ws + s.aux_off + s.ic;
                a.running_mean = n->buffers + s.buf_off; a.running_var = n->buffers + s.buf_off + s.ic;
                a.eps = s.eps; a.momentum = s.momentum; a.train = train; a.scratch = scratch;
                if (train && !sync && s.stats_rows > 0 && si > 0) {
                    const Stage& cv = n->st[si - 1];
                    a.stats_part = ws + s.stats_off; a.stats_rows = s.stats_rows;
                    a.stats_pivot = cv.bias_packed ? cv.bias_packed : P + cv.b_off;
                }
                s.stats_rows = 0;
                if (!sync) { rc = fg_launch_bn_forward(ctx, a); break; }
                if (n->run_phase == 0) {           // local fp64 sums -> pause for the cross-rank all-reduce
                    if (2LL * s.ic + 1 > n->sync_cap) return fg_set_err(ctx, FG_ERR_WORKSPACE, "sync-BN buffer too small");
                    if ((rc = fg_launch_bn_forward_sync1(ctx, a, n->sync_buf))) return rc;
                    n->run_stage = si; n->run_phase = 1; n->run_cur = cur; n->sync_count = 2LL * s.ic + 1;
                    return FG_PAUSED_SYNC;
                }
                n->run_phase = 0;
                rc = fg_launch_bn_forward_sync2(ctx, a, n->sync_buf);
                break;
            }
            case ST_PRELU: {
                if (s.act_done) { s.act_done = 0; break; }      // written by the producing layer's split-K sum
                const long long cnt = (long long)B * s.ic * s.ih * s.iw;
                const float sc = (s.mask_kind == 2 && train) ? 1.f / (1.f - s.p) : 1.f;
                rc = fg_launch_prelu_forward(ctx, cur, P + s.slope_off, train ? mask : nullptr, sc, y, cnt);
                break;
            }
            case ST_ACTPOOL: {
                const float sc = train ? 1.f : (1.f - s.p);
                // pending: the producing layer split K and left its partials -- this pass finishes the pre-activation into
                // `cur` (the conv stage's output buffer, kept for backward) and pools it
                rc = fg_launch_actpool_forward(ctx, cur, P + s.slope_off, train ? mask : nullptr, sc, y, B, s.ih, s.iw, s.ic,
                                               n->pend.splits ? &n->pend : nullptr, n->pend.splits ? (float*)cur : nullptr);
                n->pend.splits = 0;
                break;
            }
            case ST_SIGMOID: rc = fg_launch_sigmoid_forward(ctx, cur, y, (long long)B * s.ic * s.ih * s.iw); break;
            case ST_LEAKYRELU: rc = fg_launch_leakyrelu_forward(ctx, cur, s.negslope, y, (long long)B * s.ic * s.ih * s.iw); break;
            case ST_UPSAMPLE: rc = fg_launch_upsample_forward(ctx, cur, y, B, s.ih, s.iw, s.ic); break;
            case ST_REVIEW: rc = fg_launch_nchw_review(ctx, cur, y, B, s.ih, s.iw, s.ic, s.factor, 0); break;
            case ST_AVGPOOL: rc = fg_launch_avgpool_forward(ctx, cur, y, B, s.ih, s.iw, s.ic); break;
            case ST_SDROPOUT:
                rc = fg_launch_scale_mask_nc(ctx, cur, train ? mask : nullptr, train ? 1.f : 1.f - s.p, y, B, s.ih * s.iw, s.ic);
                break;
            case ST_MAXPOOL: rc = fg_launch_maxpool_forward(ctx, cur, y, B, s.ih, s.iw, s.ic); break;
            case ST_ACTMAXPOOL: {
                const float* m = (s.mask_kind && n->run_train) ? n->mask_ptrs[s.mask_idx] : nullptr;     // evaluate(): Dropout is the identity
                rc = fg_launch_actmaxpool_forward(ctx, cur, P + s.slope_off, m, s.mask_kind ? 1.f / (1.f - s.p) : 1.f, y, B, s.ih, s.iw, s.ic);
                break;
            }
            case ST_DROPOUT:
                rc = fg_launch_mul_mask(ctx, cur, train ? mask : nullptr, train ? 1.f / (1.f - s.p) : 1.f, y,
                                        (long long)B * s.ic * s.ih * s.iw);
                break;
            default: rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: bad stage kind %d", s.kind);
        }
        if (rc) return rc;
        cur = y;
    }
    n->run_phase = 0;
    if (out_offset) *out_offset = n->st.back().out_off;
    return FG_OK;
}


extern "C" {
#pragma GCC visibility push(default)

int fg_net_create(fg_ctx* ctx, const fg_layer_spec* L, int nl, int in_c, int in_h, int in_w, fg_net** out) {
    if (!ctx || !L || !out || nl <= 0) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_create: null/empty");
    fg_net* n = new fg_net();
    n->ctx = ctx; n->in_c = in_c; n->in_h = in_h; n->in_w = in_w;
    n->park_w = (ctx->fusion & FG_FUSE_WFINISH_BATCH) != 0;
    n->layers.resize(nl);
    int c = in_c, h = in_h, w = in_w;
    int perm_c = 0, perm_hw = 0;  // pending NCHW-flatten permutation for the next Linear
    int review_factor = 0;        // the FG_CONV just parsed is a SpatialConvolutionUpsample with factor > 1
    long long poff = 0, boff = 0;
    int rc = FG_OK;
    auto fail = [&](int code, const char* msg, int i) {
        rc = fg_set_err(ctx, code, "fg_net_create: layer %d: %s", i, msg);
    };
    for (int i = 0; i < nl && rc == FG_OK;) {
        Stage s; s.first_layer = i; s.ic = c; s.ih = h; s.iw = w;
        const fg_layer_spec& l = L[i];
        int consumed = 1;
        switch (l.type) {
            case FG_LINEAR: {
                if (h != 1 || w != 1 || l.a != c) { fail(FG_ERR_INVALID, "Linear input size mismatch", i); break; }
                s.w_off = poff; s.w_n = (long long)l.a * l.b; s.b_off = poff + s.w_n; s.b_n = l.b;
                poff += s.w_n + s.b_n;
                n->layers[i].w_off = s.w_off; n->layers[i].w_n = s.w_n; n->layers[i].b_off = s.b_off; n->layers[i].b_n = s.b_n;
                if (l.b == 1) {
                    if (perm_hw > 1) { fail(FG_ERR_UNSUPPORTED, "Linear(->1) after a spatial flatten", i); break; }
                    s.kind = ST_GEMV; s.oc = 1; s.oh = s.ow = 1;
                    if (i + 1 < nl && L[i + 1].type == FG_SIGMOID) { s.has_sigmoid = 1; consumed = 2; }
                } else {
                    s.kind = ST_CONV;
                    ConvGeom& g = s.geom; g.H = g.W = 1; g.Cin = l.a; g.Cout = l.b; g.k = 1; g.pad = 0; g.fold = 0;
                    if (perm_hw > 1) { g.i_c = perm_c; g.i_hw = perm_hw; }
                    s.oc = l.b; s.oh = s.ow = 1;
                    if (i + 1 < nl && L[i + 1].type == FG_VIEW && L[i + 1].b > 0) {
                        const fg_layer_spec& v = L[i + 1];
                        if ((long long)v.a * v.b * v.c != l.b) { fail(FG_ERR_INVALID, "View size mismatch", i + 1); break; }
                        if (v.b * v.c > 1) { g.o_c = v.a; g.o_hw = v.b * v.c; }
                        s.oc = v.a; s.oh = v.b; s.ow = v.c;
                        consumed = 2;
                    }
                }
                perm_c = perm_hw = 0;
                break;
            }
            case FG_VIEW: {
                if (l.b > 0) {
                    if ((long long)l.a * l.b * l.c != (long long)c * h * w) { fail(FG_ERR_INVALID, "View size mismatch", i); break; }
                    if (l.a == c && l.b == h && l.c == w) { s.kind = 0; break; }
                    fail(FG_ERR_UNSUPPORTED, "View(C,H,W) must directly follow a Linear", i);
                } else {
                    if ((long long)l.a != (long long)c * h * w) { fail(FG_ERR_INVALID, "View size mismatch", i); break; }
                    if (h * w > 1) { perm_c = c; perm_hw = h * w; }
                    c = c * h * w; h = w = 1;
                    s.kind = 0;
                }
                break;
            }
            case FG_UPSAMPLE2X: {
                if (i + 1 < nl && L[i + 1].type == FG_CONV && L[i + 1].a == c && L[i + 1].a > 4 && L[i + 1].b > 4 &&
                    L[i + 1].c % 2 == 1 && L[i + 1].d == (L[i + 1].c - 1) / 2 && L[i + 1].a % 4 == 0 &&
                    !(L[i + 1].q > 1.f) && L[i + 1].p != 2.f) {     // a factor > 1 view or a stride-2 conv behind: not folded
                    const fg_layer_spec& cv = L[i + 1];
                    int T, rmin; fg_fold_window(cv.c, cv.d, &T, &rmin);
                    if (4 * T * T <= FG_MAX_GROUPS) {
                        s.kind = ST_CONV;
                        ConvGeom& g = s.geom; g.H = h; g.W = w; g.Cin = cv.a; g.Cout = cv.b; g.k = cv.c; g.pad = cv.d; g.fold = 1;
                        fg_geom_set_wino(g, ctx->fusion);       // every output parity is a 3x3 convolution of the source: Winograd
                        s.w_n = (long long)cv.a * cv.b * cv.c * cv.c; s.b_n = cv.b;
                        s.w_off = poff; s.b_off = poff + s.w_n; poff += s.w_n + s.b_n;
                        n->layers[i + 1].w_off = s.w_off; n->layers[i + 1].w_n = s.w_n;
                        n->layers[i + 1].b_off = s.b_off; n->layers[i + 1].b_n = s.b_n;
                        s.oc = cv.b; s.oh = 2 * h; s.ow = 2 * w;
                        consumed = 2;
                        break;
                    }
                }
                s.kind = ST_UPSAMPLE; s.oc = c; s.oh = 2 * h; s.ow = 2 * w;
                break;
            }
            case FG_CONV: {
                if (l.a != c) { fail(FG_ERR_INVALID, "conv nInputPlane mismatch", i); break; }
                if (l.c % 2 != 1 || l.d != (l.c - 1) / 2) { fail(FG_ERR_UNSUPPORTED, "only odd-k 'same'-pad convs (stride 1 or 2)", i); break; }
                s.w_n = (long long)l.a * l.b * l.c * l.c; s.b_n = l.b;
                s.w_off = poff; s.b_off = poff + s.w_n; poff += s.w_n + s.b_n;
                n->layers[i].w_off = s.w_off; n->layers[i].w_n = s.w_n; n->layers[i].b_off = s.b_off; n->layers[i].b_n = s.b_n;
                ConvGeom& g = s.geom; g.H = h; g.W = w; g.Cin = l.a; g.Cout = l.b; g.k = l.c; g.pad = l.d; g.fold = 0;
                g.stride = (l.p == 2.f) ? 2 : 1;
                s.oc = l.b; s.oh = h; s.ow = w;
                if (l.q > 1.f) {       // cudnn.SpatialConvolutionUpsample(nIn, nOut, k, k, f): b = nOut * f * f planes, viewed behind
                    const int f = (int)l.q;
                    if ((float)f != l.q || l.b % (f * f) || g.stride != 1) { fail(FG_ERR_INVALID, "SpatialConvolutionUpsample: nOutputPlane must be a multiple of factor^2 (stride 1)", i); break; }
                    review_factor = f;
                }
                if (g.stride == 2) {       // 3x3 stride-2 'same'-pad convs of create_D16_d (models.lua:289-291)
                    if ((h & 1) || (w & 1) || l.c * l.c > FG_MAX_GROUPS) { fail(FG_ERR_UNSUPPORTED, "stride-2 conv: even H/W", i); break; }
                    s.oh = h / 2; s.ow = w / 2; s.kind = ST_CONV;
                } else
                if (l.a <= 4 && l.b % 64 == 0) s.kind = ST_THIN_IN;
                else if (l.b <= 4 && l.a % 64 == 0) {
                    s.kind = ST_THIN_OUT;
                    if (i + 1 < nl && L[i + 1].type == FG_SIGMOID && !(l.q > 1.f)) { s.has_sigmoid = 1; consumed = 2; }
                } else if (l.c * l.c <= FG_MAX_GROUPS) { s.kind = ST_CONV; fg_geom_set_wino(g, ctx->fusion); }   // 3x3: Winograd F(2x2, 3x3)
                else fail(FG_ERR_UNSUPPORTED, "conv channel counts not supported", i);
                break;
            }
            case FG_BATCHNORM: {
                if (l.a != c || c % 4) { fail(FG_ERR_INVALID, "BatchNorm nFeature mismatch / % 4", i); break; }
                s.kind = ST_BNPRELU; s.oc = c; s.oh = h; s.ow = w;
                s.gamma_off = poff; s.beta_off = poff + c; poff += 2 * c;
                n->layers[i].w_off = s.gamma_off; n->layers[i].w_n = c; n->layers[i].b_off = s.beta_off; n->layers[i].b_n = c;
                s.buf_off = boff; boff += 2 * c;
                s.eps = l.p > 0.f ? l.p : 1e-5f; s.momentum = l.q > 0.f ? l.q : 0.1f;
                if (i + 1 < nl && L[i + 1].type == FG_PRELU) {
                    s.has_prelu = 1; s.slope_off = poff; poff += 1;
                    n->layers[i + 1].w_off = s.slope_off; n->layers[i + 1].w_n = 1;
                    consumed = 2;
                }
                break;
            }
            case FG_PRELU: {
                s.slope_off = poff; poff += 1;
                n->layers[i].w_off = s.slope_off; n->layers[i].w_n = 1;
                s.has_prelu = 1; s.oc = c; s.oh = h; s.ow = w;
                static int amp_on = -1;      // A/B switch (round 4; default on): FG_ACTMAXPOOL=0 keeps PReLU / MaxPool / Dropout apart
                if (amp_on < 0) { const char* e = getenv("FG_ACTMAXPOOL"); amp_on = e ? atoi(e) : 1; }
                if (i + 2 < nl && L[i + 1].type == FG_SPATIAL_DROPOUT && L[i + 2].type == FG_AVGPOOL2 && c % 4 == 0 &&
                    h % 2 == 0 && w % 2 == 0) {
                    s.kind = ST_ACTPOOL; s.p = L[i + 1].p; s.mask_kind = 1; s.oh = h / 2; s.ow = w / 2; consumed = 3;
                } else if (amp_on && i + 1 < nl && L[i + 1].type == FG_MAXPOOL2 && c % 4 == 0 && h % 2 == 0 && w % 2 == 0 && i > 0) {
                    // PReLU -> MaxPool [-> Dropout]: the pooled (and masked) tensor is all the next layer reads, prelu(x) at full
                    // resolution is never materialised (the producing layer stores x alone); backward re-evaluates it from x
                    s.kind = ST_ACTMAXPOOL; s.oh = h / 2; s.ow = w / 2; consumed = 2;
                    if (i + 2 < nl && L[i + 2].type == FG_DROPOUT) { s.p = L[i + 2].p; s.mask_kind = 2; consumed = 3; }
                } else if (i + 1 < nl && L[i + 1].type == FG_DROPOUT) {
                    s.kind = ST_PRELU; s.p = L[i + 1].p; s.mask_kind = 2; consumed = 2;
                } else s.kind = ST_PRELU;
                if (s.mask_kind) { s.mask_idx = n->n_masks++; }
                break;
            }
            case FG_SIGMOID: s.kind = ST_SIGMOID; s.oc = c; s.oh = h; s.ow = w; break;
            case FG_LEAKYRELU: s.kind = ST_LEAKYRELU; s.negslope = l.p; s.oc = c; s.oh = h; s.ow = w; break;
            case FG_AVGPOOL2:
                if (h % 2 || w % 2) { fail(FG_ERR_INVALID, "AvgPool on odd size", i); break; }
                s.kind = ST_AVGPOOL; s.oc = c; s.oh = h / 2; s.ow = w / 2; break;
            case FG_SPATIAL_DROPOUT:
                s.kind = ST_SDROPOUT; s.p = l.p; s.mask_kind = 1; s.mask_idx = n->n_masks++; s.oc = c; s.oh = h; s.ow = w; break;
            case FG_DROPOUT:
                s.kind = ST_DROPOUT; s.p = l.p; s.mask_kind = 2; s.mask_idx = n->n_masks++; s.oc = c; s.oh = h; s.ow = w; break;
            case FG_MAXPOOL2:
                if (h % 2 || w % 2 || c % 4) { fail(FG_ERR_INVALID, "MaxPool needs even H/W and C % 4 == 0", i); break; }
                s.kind = ST_MAXPOOL; s.oc = c; s.oh = h / 2; s.ow = w / 2; break;
            default: fail(FG_ERR_INVALID, "unknown layer type", i); break;
        }
        if (rc != FG_OK) break;
        s.last_layer = i + consumed - 1;
        if (s.kind != 0) {
            for (int j = i; j <= s.last_layer; ++j) n->layers[j].stage = (int)n->st.size();
            n->layers[s.last_layer].ends_stage = 1;
            if (s.mask_idx >= 0) n->mask_stage.push_back((int)n->st.size());
            c = s.oc; h = s.oh; w = s.ow;
            if (s.kind != ST_CONV && s.kind != ST_GEMV && perm_hw > 1 && l.type != FG_VIEW) {
                // elementwise stages between the flatten and the Linear keep the pending permutation
            }
            n->st.push_back(s);
            if (review_factor > 1) {
                // the view stage: [c][h][w] (NCHW) re-read as [c / f^2][h f][w f]; layer i now ends HERE (fg_net_layer_output(i)
                // is the module's output, like the reference's self.output), and nothing is fused across it
                const int f = review_factor;
                review_factor = 0;
                if (s.has_sigmoid) { rc = fg_set_err(ctx, FG_ERR_UNSUPPORTED, "fg_net_create: layer %d: Sigmoid fused in front of the view", i); break; }
                Stage v; v.kind = ST_REVIEW; v.first_layer = v.last_layer = i; v.factor = f;
                v.ic = c; v.ih = h; v.iw = w; v.oc = c / (f * f); v.oh = h * f; v.ow = w * f;
                n->layers[i].stage = (int)n->st.size();
                c = v.oc; h = v.oh; w = v.ow;
                n->st.push_back(v);
            }
        } else if (!n->st.empty()) {
            n->layers[i].stage = (int)n->st.size() - 1;
            n->layers[i].ends_stage = 1;
        }
        i += consumed;
    }
    if (rc == FG_OK && n->st.empty()) rc = fg_set_err(ctx, FG_ERR_INVALID, "fg_net_create: no compute stages");
    n->n_params = poff; n->n_buffers = boff;
    // packed-weight storage (owned by the net; allocated here, never on the hot path).  The contraction stages share one
    // allocation so that the bf16x6 planes of all of them are rebuilt by a single launch.
    {
        long long tot = 0;
        for (auto& s : n->st)
            if (s.kind == ST_CONV) {
                ConvGeom g = s.geom; g.B = 1;
                tot += fg_geom_pack_floats(g, 0) + fg_geom_pack_floats(g, 1);
            }
        n->packed_total = tot;
        if (rc == FG_OK && tot > 0 &&
            (fg_dev_alloc((void**)&n->packed_all, tot * sizeof(float)) != hipSuccess ||
             fg_dev_alloc((void**)&n->planes_all, tot / 16 * 96) != hipSuccess))
            rc = fg_set_err(ctx, FG_ERR_NOMEM, "fg_net_create: packed weights");
    }
    long long pk_off = 0;
    for (auto& s : n->st) {
        if (rc != FG_OK) break;
        if (s.kind == ST_CONV) {
            ConvGeom g = s.geom; g.B = 1;
            long long nf = fg_geom_pack_floats(g, 0), nb = fg_geom_pack_floats(g, 1);
            s.wp_fwd = n->packed_all + pk_off; s.wp_fwd6 = n->planes_all + pk_off / 16 * 96; pk_off += nf;
            s.wp_bwd = n->packed_all + pk_off; s.wp_bwd6 = n->planes_all + pk_off / 16 * 96; pk_off += nb;
            if (rc == FG_OK && g.o_hw > 1 && fg_dev_alloc((void**)&s.bias_packed, s.b_n * sizeof(float)) != hipSuccess)
                rc = fg_set_err(ctx, FG_ERR_NOMEM, "fg_net_create: packed bias");
        } else if (s.kind == ST_THIN_IN || s.kind == ST_THIN_OUT) {
            if (fg_dev_alloc((void**)&s.wp_fwd, s.w_n * sizeof(float)) != hipSuccess)
                rc = fg_set_err(ctx, FG_ERR_NOMEM, "fg_net_create: packed weights");
        }
    }
    if (rc == FG_OK) rc = build_pack_jobs(n);
    if (rc != FG_OK) { fg_net_destroy(n); return rc; }
    *out = n;
    return FG_OK;
}

int fg_net_destroy(fg_net* n) {
    if (!n) return FG_OK;
    for (auto& s : n->st) {
        if (s.kind != ST_CONV && s.wp_fwd) fg_dev_free(s.wp_fwd);
        if (s.bias_packed) fg_dev_free(s.bias_packed);
    }
    if (n->packed_all) fg_dev_free(n->packed_all);
    if (n->planes_all) fg_dev_free(n->planes_all);
    if (n->jobs_dev) fg_dev_free(n->jobs_dev);
    delete n;
    return FG_OK;
}

long long fg_net_num_params(const fg_net* n) { return n ? n->n_params : 0; }
int fg_net_in_dims(const fg_net* n, int* c, int* h, int* w) {
    if (!n) return FG_ERR_INVALID;
    if (c) *c = n->in_c; if (h) *h = n->in_h; if (w) *w = n->in_w;
    return FG_OK;
}
int fg_net_vectors(const fg_net* n, float** params, float** grads, float** buffers) {
    if (!n) return FG_ERR_INVALID;
    if (params) *params = n->params; if (grads) *grads = n->grads; if (buffers) *buffers = n->buffers;
    return FG_OK;
}
int fg_net_max_bn_channels(const fg_net* n) {
    int c = 0;
    if (n) for (auto& s : n->st) if (s.kind == ST_BNPRELU && s.ic > c) c = s.ic;
    return c;
}
long long fg_net_num_buffers(const fg_net* n) { return n ? n->n_buffers : 0; }
int fg_net_num_masks(const fg_net* n) { return n ? n->n_masks : 0; }
long long fg_net_mask_elems(const fg_net* n, int mi, int batch) {
    if (!n || mi < 0 || mi >= n->n_masks) return -1;
    const Stage& s = n->st[n->mask_stage[mi]];
    if (s.kind == ST_ACTMAXPOOL) return (long long)batch * s.oc * s.oh * s.ow;       // the Dropout acts on the POOLED tensor
    return s.mask_kind == 1 ? (long long)batch * s.ic : (long long)batch * s.ic * s.ih * s.iw;
}
float fg_net_mask_keep(const fg_net* n, int mi) {
    if (!n || mi < 0 || mi >= n->n_masks) return -1.f;
    return 1.f - n->st[n->mask_stage[mi]].p;
}
int fg_net_out_dims(const fg_net* n, int* c, int* h, int* w) {
    if (!n) return FG_ERR_INVALID;
    const Stage& s = n->st.back();
    if (c) *c = s.oc; if (h) *h = s.oh; if (w) *w = s.ow;
    return FG_OK;
}
size_t fg_net_workspace_bytes(const fg_net* n, int max_batch) {
    if (!n) return 0;
    fg_net tmp = *n;  // plan on a copy: const query
    make_plan(&tmp, max_batch);
    return (size_t)tmp.total_floats * sizeof(float) + 256;
}
int fg_net_param_offset(const fg_net* n, int li, long long* wo, long long* wn, long long* bo, long long* bn) {
    if (!n || li < 0 || li >= (int)n->layers.size()) return FG_ERR_INVALID;
    const LayerInfo& l = n->layers[li];
    if (wo) *wo = l.w_off; if (wn) *wn = l.w_n; if (bo) *bo = l.b_off; if (bn) *bn = l.b_n;
    return FG_OK;
}
int fg_net_bind(fg_net* n, float* params, float* grads, float* buffers) {
    if (!n || !params) return fg_set_err(n ? n->ctx : nullptr, FG_ERR_INVALID, "fg_net_bind: params required");
    n->params = params; n->grads = grads; n->buffers = buffers; n->dirty = true;
    return FG_OK;
}
int fg_net_params_changed(fg_net* n) {
    if (!n) return FG_ERR_INVALID;
    n->dirty = true;
    return FG_OK;
}

// Adam over the net's flat parameter vector + the re-pack of every layer, one launch (a.p must be the net's own vector)
extern "C++" int fg_net_adam_step(fg_net* n, const AdamArgs& a) {
    if (!n) return FG_ERR_INVALID;
    if (!n->adam_fusable || !(n->ctx->fusion & FG_FUSE_ADAM_PACK) || a.p != n->params || a.n != n->n_params || a.gout || n->n_jobs_all == 0) {
        const int rc = fg_launch_adam(n->ctx, a);
        n->dirty = true;
        return rc;
    }
    const int rc = fg_launch_adam_pack_jobs(n->ctx, n->jobs_dev, n->n_jobs_all, n->jobs_total_all, a, n->pack_lds_floats);
    if (rc) { n->dirty = true; return rc; }
    n->dirty = false;
    n->planes_valid = false;
    return FG_OK;
}

// backward with the per-layer final reductions batched into one launch per call (also on a sync-BN pause or an error:
// nothing deferred may survive the return to the caller)
static int backward_run(fg_net* n) {
    fg_ctx* ctx = n->ctx;
    n->defer.arena = n->run_ws + n->defer_off; n->defer.cap = n->defer_floats;
    n->defer.used = 0; n->defer.n = 0; n->defer.blocks = 0;
    //  a net created with FG_FUSE_WFINISH_BATCH off reserved no room for parked weight-gradient partials: no job table,
    // so fg_conv_wgrad_run does not eat the arena of the bias / slope partials when the bit is switched on later
    n->defer.wjobs = n->park_w ? n->wjobs : nullptr; n->defer.wn = 0; n->defer.wblocks = 0;
    ctx->defer = &n->defer;
    const int rc = backward_run_stages(n);
    const int rf = fg_defer_flush(ctx);
    ctx->defer = nullptr;
    return rc ? rc : rf;
}

static int build_pack_jobs(fg_net* n) {
    std::vector<PackJob> jobs;
    long long start = 0;
    auto add = [&](const WeightMap& wm, int mode, long long src, float* dst, int rows, int cols, long long count) {
        PackJob j; memset(&j, 0, sizeof(j));
        j.wm = wm; j.mode = mode; j.src_off = src; j.dst = dst; j.rows = rows; j.cols = cols; j.start = start; j.count = count;
        jobs.push_back(j);
        start += (count + 255) / 256 * 256;        // job spans are whole blocks: the kernel looks its job up per block
    };
    for (auto& s : n->st) {
        if (s.kind == ST_CONV) {
            ConvGeom g = s.geom; g.B = 1;
            WeightMap wm; fg_geom_packmap(g, &wm);
            int rf, cf, rb, cb; fg_geom_pack_dims(g, &rf, &cf, &rb, &cb);
            if (wm.k > 1 && wm.k <= 7 && wm.o_hw <= 1 && wm.i_hw <= 1) {   // convolutions: one LDS-staged job makes both packs
                PackJob j; memset(&j, 0, sizeof(j));
                j.wm = wm; j.mode = 7; j.src_off = s.w_off; j.dst = s.wp_fwd; j.rows = rf; j.cols = cf;
                j.dst2 = s.wp_bwd; j.rows2 = rb; j.cols2 = cb;
                const int opad = rf > cb ? rf : cb, ipad = cf > rb ? cf : rb;
                j.npo = (opad + 15) / 16; j.npi = (ipad + 15) / 16;
                j.start = start; j.count = (long long)j.npo * j.npi * 256;
                jobs.push_back(j);
                start += j.count;
            } else if (wm.k == 1 && wm.i_hw > 1 && wm.i_c > 0) {   // Linear behind a View: 16 x 16 x 16 bricks (mode 10)
                PackJob j; memset(&j, 0, sizeof(j));
                j.wm = wm; j.mode = 10; j.src_off = s.w_off; j.dst = s.wp_fwd; j.rows = rf; j.cols = cf;
                j.dst2 = s.wp_bwd; j.rows2 = rb; j.cols2 = cb;
                const int opad = rf > cb ? rf : cb, ipad = cf > rb ? cf : rb;
                j.npo = (opad + 15) / 16;
                j.npi = ((ipad + wm.i_c - 1) / wm.i_c + 15) / 16;            // bricks along hw, over the padded column extent
                j.start = start; j.count = (long long)j.npo * ((wm.i_c + 15) / 16) * j.npi * 256;
                jobs.push_back(j);
                start += j.count;
            } else if (wm.k == 1) {                           // Linear: both packs from one LDS-tiled pass (mode 8)
                PackJob j; memset(&j, 0, sizeof(j));
                j.wm = wm; j.mode = 8; j.src_off = s.w_off; j.dst = s.wp_fwd; j.rows = rf; j.cols = cf;
                j.dst2 = s.wp_bwd; j.rows2 = rb; j.cols2 = cb;
                const int opad = rf > cb ? rf : cb, ipad = cf > rb ? cf : rb;
                j.npo = (opad + 31) / 32; j.npi = (ipad + 31) / 32;
                j.start = start; j.count = (long long)j.npo * j.npi * 256;
                jobs.push_back(j);
                start += j.count;
            } else {
                add(wm, 0, s.w_off, s.wp_fwd, rf, cf, (long long)wm.P * wm.G * rf * cf);
                add(wm, 1, s.w_off, s.wp_bwd, rb, cb, (long long)wm.P * wm.G * rb * cb);
            }
            if (s.bias_packed) add(wm, 4, s.b_off, s.bias_packed, 0, 0, s.b_n);
        } else if (s.kind == ST_THIN_IN || s.kind == ST_THIN_OUT) {
            WeightMap wm; memset(&wm, 0, sizeof(wm));
            wm.O = s.geom.Cout; wm.I = s.geom.Cin; wm.k = s.geom.k;
            add(wm, s.kind == ST_THIN_IN ? 2 : 3, s.w_off, s.wp_fwd, 0, 0, s.w_n);
        }
    }
    n->n_jobs = (int)jobs.size();
    n->jobs_total = start;
    n->pack_lds_floats = 64;
    for (auto& j : jobs) { const long long f = fg_pack_lds_floats(j.mode, j.wm.k); if (f > n->pack_lds_floats) n->pack_lds_floats = f; }
    // the fused optimizer + re-pack launch (fg_net_adam_step): possible when every pack job reads each of its weights once
    // (all but the generic modes 0 / 1); the parameters NO job reads get update-only jobs (mode 9) behind the pack jobs
    n->adam_fusable = true;
    std::vector<std::pair<long long, long long>> cov;          // [from, to) of the flat parameter vector read by a pack job
    for (auto& j : jobs) {
        if (j.mode <= 1) n->adam_fusable = false;
        const long long cnt = (j.mode == 7 || j.mode == 8 || j.mode == 10) ? (long long)j.wm.O * j.wm.I * j.wm.k * j.wm.k : j.count;
        cov.push_back({j.src_off, j.src_off + cnt});
    }
    std::sort(cov.begin(), cov.end());
    {
        WeightMap none; memset(&none, 0, sizeof(none));
        long long at = 0;
        auto gap = [&](long long from, long long to) { if (to > from) add(none, 9, from, nullptr, 0, 0, to - from); };
        for (auto& c : cov) {
            if (c.first < at) n->adam_fusable = false;          // two jobs over one weight: not fusable
            gap(at, c.first);
            if (c.second > at) at = c.second;
        }
        gap(at, n->n_params);
    }
    n->n_jobs_all = (int)jobs.size();
    n->jobs_total_all = start;
    if (jobs.empty()) return FG_OK;
    if (fg_dev_alloc((void**)&n->jobs_dev, jobs.size() * sizeof(PackJob)) != hipSuccess)
        return fg_set_err(n->ctx, FG_ERR_NOMEM, "fg_net_create: pack jobs");
    FG_HIP(n->ctx, hipMemcpy(n->jobs_dev, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    return FG_OK;
}

static int pack_all(fg_net* n) {
    int rc = fg_launch_pack_jobs(n->ctx, n->jobs_dev, n->n_jobs, n->jobs_total, n->params, 4.0 * ((double)n->n_params + (double)n->packed_total),
                                 n->pack_lds_floats);
    if (rc) return rc;
    n->dirty = false;
    n->planes_valid = false;
    return FG_OK;
}
// packed weights current (+ their bf16x6 planes when that math mode is on)
static int ensure_packed(fg_net* n) {
    int rc;
    if (n->dirty && (rc = pack_all(n))) return rc;
    if (n->ctx->math == 6 && !n->planes_valid && n->packed_total > 0) {
        if ((rc = fg_launch_split_planes(n->ctx, n->packed_all, n->packed_total / 16, 16, n->planes_all))) return rc;
        n->planes_valid = true;
    }
    return FG_OK;
}

int fg_net_forward(fg_net* n, int B, const float* x, void* wsv, size_t ws_bytes, int train, const float* const* masks,
                   int n_masks, long long* out_offset) {
    return fg_net_forward_to(n, B, x, wsv, ws_bytes, train, masks, n_masks, out_offset, nullptr);
}

int fg_net_forward_to(fg_net* n, int B, const float* x, void* wsv, size_t ws_bytes, int train, const float* const* masks,
                      int n_masks, long long* out_offset, float* out_override) {
    if (!n || !x || !wsv) return fg_set_err(n ? n->ctx : nullptr, FG_ERR_INVALID, "fg_net_forward: null argument");
    if (out_override && ((uintptr_t)out_override & 15)) return fg_set_err(n->ctx, FG_ERR_INVALID, "fg_net_forward_to: 16-byte alignment");
    n->out_override = out_override;
    fg_ctx* ctx = n->ctx;
    if (!n->params) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: fg_net_bind first");
    if (B <= 0) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: batch %d", B);
    if (((uintptr_t)wsv & 15) || ((uintptr_t)x & 15)) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: 16-byte alignment");
    if (train && n->n_masks > 0 && (!masks || n_masks != n->n_masks))
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: %d dropout masks required in training mode", n->n_masks);
    if (n->n_buffers > 0 && !n->buffers) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_forward: BN buffers not bound");
    make_plan(n, B);
    if ((size_t)n->total_floats * sizeof(float) > ws_bytes)
        return fg_set_err(ctx, FG_ERR_WORKSPACE, "fg_net_forward: workspace %zu < %lld bytes", ws_bytes,
                          n->total_floats * (long long)sizeof(float));
    int rc;
    if ((rc = ensure_packed(n))) return rc;
    float* ws = (float*)wsv;
    n->mask_ptrs.assign(n->n_masks, nullptr);
    if (train) for (int i = 0; i < n->n_masks; ++i) n->mask_ptrs[i] = masks[i];
    n->last_train = train;
    n->run_stage = 0; n->run_phase = 0; n->run_B = B; n->run_train = train; n->run_x = x; n->run_cur = x; n->run_ws = ws;
    return forward_run(n, out_offset);
}

int fg_net_forward_resume(fg_net* n, long long* out_offset) {
    if (!n || n->run_phase != 1) return fg_set_err(n ? n->ctx : nullptr, FG_ERR_INVALID, "fg_net_forward_resume: not paused");
    return forward_run(n, out_offset);
}

int fg_net_num_stages(const fg_net* n) { return n ? (int)n->st.size() : 0; }

int fg_net_stage_params(const fg_net* n, int stage, long long* lo, long long* hi) {
    if (!n || stage < 0 || stage >= (int)n->st.size()) return FG_ERR_INVALID;
    const Stage& s = n->st[stage];
    long long a = -1, b = -1;
    auto take = [&](long long off, long long cnt) {
        if (off < 0 || cnt <= 0) return;
        if (a < 0 || off < a) a = off;
        if (off + cnt > b) b = off + cnt;
    };
    take(s.w_off, s.w_n); take(s.b_off, s.b_n); take(s.gamma_off, s.ic); take(s.beta_off, s.ic); take(s.slope_off, 1);
    if (a < 0) a = b = 0;
    if (lo) *lo = a;
    if (hi) *hi = b;
    return FG_OK;
}

int fg_net_backward(fg_net* n, int B, const float* x, const float* gy, void* wsv, size_t ws_bytes, int flags, float* gx) {
    if (!n) return FG_ERR_INVALID;
    return fg_net_backward_range(n, B, x, gy, wsv, ws_bytes, flags, gx, (int)n->st.size() - 1, 0);
}

int fg_net_backward_range(fg_net* n, int B, const float* x, const float* gy, void* wsv, size_t ws_bytes, int flags,
                          float* gx, int stage_from, int stage_to) {
    if (!n || !x || !wsv) return fg_set_err(n ? n->ctx : nullptr, FG_ERR_INVALID, "fg_net_backward: null argument");
    fg_ctx* ctx = n->ctx;
    const int last = (int)n->st.size() - 1;
    if (stage_from > last || stage_to < 0 || stage_from < stage_to) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward_range: bad range");
    if (n->plan_batch != B) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: batch %d != forward batch %d", B, n->plan_batch);
    if (!n->last_train) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: last forward was in evaluate mode");
    if ((size_t)n->total_floats * sizeof(float) > ws_bytes) return fg_set_err(ctx, FG_ERR_WORKSPACE, "fg_net_backward: workspace");
    if ((flags & FG_BWD_PARAM_GRADS) && !n->grads) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: gradient vector not bound");
    if ((flags & FG_BWD_INPUT_GRAD) && !gx) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: gx required");
    if (stage_from == last) {            // a new backward pass starts at the output
        if (!gy) return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward: gy required");
        n->bwd_gcur = gy; n->bwd_pp = 0;
    } else if (n->bwd_next != stage_from) {
        return fg_set_err(ctx, FG_ERR_INVALID, "fg_net_backward_range: expected stage %d next", n->bwd_next);
    }
    n->run_stage = stage_from; n->run_phase = 0; n->run_to = stage_to; n->run_flags = flags; n->run_B = B;
    n->run_x = x; n->run_ws = (float*)wsv; n->run_gx = gx;
    { const int rc0 = ensure_packed(n); if (rc0) return rc0; }
    return backward_run(n);
}

int fg_net_backward_resume(fg_net* n) {
    if (!n || n->run_phase != 2) return fg_set_err(n ? n->ctx : nullptr, FG_ERR_INVALID, "fg_net_backward_resume: not paused");
    return backward_run(n);
}

int fg_net_set_sync_bn(fg_net* n, int on, double* sync_buf, long long capacity_doubles) {
    if (!n) return FG_ERR_INVALID;
    if (on && (!sync_buf || capacity_doubles <= 0)) return fg_set_err(n->ctx, FG_ERR_INVALID, "fg_net_set_sync_bn: buffer required");
    n->sync_bn = on != 0; n->sync_buf = sync_buf; n->sync_cap = capacity_doubles;
    return FG_OK;
}
long long fg_net_sync_count(const fg_net* n) { return n ? n->sync_count : 0; }

int fg_net_layer_output(const fg_net* n, int li, long long* off, int* c, int* h, int* w) {
    if (!n || li < 0 || li >= (int)n->layers.size()) return FG_ERR_INVALID;
    const LayerInfo& l = n->layers[li];
    if (l.stage < 0 || !l.ends_stage) return fg_set_err(n->ctx, FG_ERR_INVALID, "layer %d is fused inside a stage", li);
    const Stage& s = n->st[l.stage];
    if (l.stage + 1 == (int)n->st.size() && n->out_override)
        return fg_set_err(n->ctx, FG_ERR_INVALID, "layer %d: the net's output was redirected by fg_net_forward_to", li);
    if (off) *off = s.out_off;
    if (c) *c = s.oc; if (h) *h = s.oh; if (w) *w = s.ow;
    return FG_OK;
}

int fg_net_bn_saved_stats(const fg_net* n, int li, long long* mean_off, long long* invstd_off, int* c) {
    if (!n || li < 0 || li >= (int)n->layers.size()) return FG_ERR_INVALID;
    const LayerInfo& l = n->layers[li];
    if (l.stage < 0 || n->st[l.stage].kind != ST_BNPRELU || n->st[l.stage].first_layer != li)
        return fg_set_err(n->ctx, FG_ERR_INVALID, "layer %d is not a SpatialBatchNormalization", li);
    const Stage& s = n->st[l.stage];
    if (mean_off) *mean_off = s.aux_off;
    if (invstd_off) *invstd_off = s.aux_off + s.ic;
    if (c) *c = s.ic;
    return FG_OK;
}

#pragma GCC visibility pop
}  // extern "C"
