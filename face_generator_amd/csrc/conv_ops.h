// Reference-geometry description of one conv / Linear layer and the host drivers around the MFMA kernels.
#pragma once
#include "fg_internal.h"

#define CR_ROWBLOCKS_MAX 256
// most bias-gradient partial rows a wave-specialised weight gradient may leave (parities x splits x taps x X tiles x loader pixel
// lanes): 25 taps x 51 splits x 4 = 5 100 for the 5x5 128 -> 256 layer of models_c2f.lua:126 since the multi-round split counts of
// round 3 -- with the old bound of 2 048 that layer silently took the separate column-sum pass over its 537 MB output gradient
#define FG_WS_BIAS_ROWS_MAX (32 * CR_ROWBLOCKS_MAX)

struct ConvGeom {
    int B;          // batch
    int H, W;       // input spatial size (source resolution: BEFORE the folded nearest-x2 upsample). Linear: 1,1
    int Cin, Cout;  // reference nInputPlane / nOutputPlane (Linear: in / out features)
    int k, pad;     // odd k, "same" pad = (k-1)/2. Linear: 1, 0
    int fold;       // 1: an nn.SpatialUpSamplingNearest(2) in front of the conv is folded into its taps
    int stride;     // 0 / 1: stride 1; 2: stride 2 (output (H/2) x (W/2); models.lua:289-291 create_D16_d), never with fold
    int wino;       // 1: forward and data gradient run as Winograd F(2x2, 3x3) (wino.hip); the packs hold U = G g G^T (fg_geom_set_wino)
    // Linear next to an nn.View: NCHW-flatten <-> NHWC-memory feature permutation (0 = none)
    int o_c, o_hw, i_c, i_hw;
};

// where the Winograd-domain weight gradient is taken (chunks per block, blocks per launch): process-wide planning thresholds
void fg_plan_env_init();                 // FG_WINO_WGRAD_MIN_CHUNKS / _MIN_BLOCKS, read ONCE (first fg_ctx_create)
void fg_plan_set_wino_wgrad_thresholds(long long min_chunks, long long min_blocks);   // <= 0: the defaults (24, 192)
void fg_geom_weightmap(const ConvGeom& g, WeightMap* wm);
// the map the PACKS of the layer are built with: fg_geom_weightmap, or kind 2 (16 Winograd positions) for a g.wino layer.
// (The weight gradient always uses fg_geom_weightmap: it is computed tap by tap whatever the forward algorithm.)
void fg_geom_packmap(const ConvGeom& g, WeightMap* wm);
// decides g.wino from the geometry (3x3, pad 1, stride 1, no folded upsample, even H / W, channel counts % 8 == 0 and > 4)
// and the context's fg_set_fusion bit FG_FUSE_WINOGRAD (default on; off keeps the implicit GEMM: the A/B and parity switch)
void fg_geom_set_wino(ConvGeom& g, int fusion);
void fg_geom_pack_dims(const ConvGeom& g, int* rows_f, int* cols_f, int* rows_b, int* cols_b);
long long fg_geom_pack_floats(const ConvGeom& g, int bwd);
long long fg_conv_scratch_floats(const ConvGeom& g);
int fg_conv_pack(fg_ctx* ctx, const ConvGeom& g, const float* W, float* wp_fwd, float* wp_bwd);
int fg_conv_forward_run(fg_ctx* ctx, const ConvGeom& g, const float* x, const float* wp_fwd, const float* bias,
                        float* y, float* scratch, long long scratch_floats, const void* wp6 = nullptr,
                        void* x6_dst = nullptr, int* x6_written = nullptr, float* stats_part = nullptr,
                        long long stats_cap = 0, int* stats_rows = nullptr, const FgActFuse* act = nullptr,
                        FgSplitParts* leave = nullptr);
// leave (optional, forward and data gradient): when the launch splits K, do NOT run the pass that sums the partials -- describe
// them in *leave (they sit at the head of `scratch`; leave->splits = 0 if the launch did not split) for the pointwise kernel
// behind the layer, which sums them itself (PReLU + SpatialDropout + AvgPool forward / backward)
// act (optional): the PReLU [+ Dropout] behind the layer; act->applied tells whether this launch folded it in (split-K layers)
// stats_part (optional, capacity stats_cap floats): the kernel's epilogue leaves per-channel sum / sum-of-squares partials of
// the raw accumulators there ([2][*stats_rows][Cout]; *stats_rows = 0 if this launch could not: split-K, bf16x6)
// x6_dst (forward): write the planes of x there instead of into the scratch (the caller keeps them for the weight
// gradient); *x6_written tells whether the bf16x6 path ran.  gy6 (dgrad): planes of gy left by fg_conv_wgrad_run.
// wp6: the packed weights as split-bf16 planes if the caller keeps them (bf16x6 mode), else they are built per call
// actb (optional): the nn.PReLU in front of the layer -- the kernel's epilogue turns the gradient wrt the layer's input into
// the gradient wrt the PReLU's input (actb->applied; un-split fp32 launches inside an fg_net backward pass)
int fg_conv_dgrad_run(fg_ctx* ctx, const ConvGeom& g, const float* gy, const float* wp_bwd, float* gx, float* scratch,
                      long long scratch_floats, const void* wp6 = nullptr, const void* gy6 = nullptr,
                      const FgActBwd* actb = nullptr, FgSplitParts* leave = nullptr);
// bf16x6 plane sharing (all optional): x6 = planes of x kept from the forward pass; *gy6_out = where this call left the
// planes of gy (nullptr if it ran in fp32) and *used_out = scratch floats that must stay untouched while they are used
int fg_conv_wgrad_run(fg_ctx* ctx, const ConvGeom& g, const float* x, const float* gy, float* gradW, float* gradb,
                      float beta, float* scratch, long long scratch_floats, const void* x6 = nullptr,
                      const void** gy6_out = nullptr, long long* used_out = nullptr);
