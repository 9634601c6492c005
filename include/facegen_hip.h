/* libfacegen_hip.so -- C ABI of the MI355X-native (gfx950) GAN training hot path of aleju/face-generator.
 *
 * This is the drop-in boundary: the reference has no native code of its own; its Lua
 * modules dispatch into Torch7's THNN/THCUNN/cuDNN through LuaJIT FFI (upstream convention:
 *   THNN_(SpatialConvolutionMM_updateOutput)(state, input, output, weight, bias, finput, ...)
 * called as input.THNN.X(input:cdata(), ...)).  The entry points below are what a LuaJIT `ffi.cdef` (or the
 * ctypes mirror in face_generator_amd/_lib.py) binds instead.  Plain C: opaque context, caller-owned device
 * buffers (raw pointers + explicit workspace), status-code returns, message via fg_last_error().
 * No C++ exceptions cross the ABI; nothing here names a torch type.
 *
 * Conventions
 *   - all tensors fp32; device activations are NHWC ("internal layout"); the reference's NCHW appears only at
 *     the boundary (fg_nchw_to_nhwc / fg_nhwc_to_nchw = the nn.Copy modules of nn_utils.lua:355-362).
 *   - parameter and gradient vectors are ONE flat fp32 vector per net in REFERENCE order and layout
 *     (module order, weight then bias; conv [O][I][kH][kW], linear [out][in]) == Module:getParameters()
 *     (train.lua:151-152).  Tile-packed / tap-folded weight copies are internal.
 *   - every entry is asynchronous on the context's stream; only fg_d2h / fg_stream_sync block.
 *   - a context is bound to one device and is not thread-safe.
 */
#ifndef FACEGEN_HIP_H
#define FACEGEN_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fg_ctx fg_ctx;
typedef struct fg_net fg_net;
typedef struct fg_comm fg_comm;
typedef struct fg_gan fg_gan;

enum {
    FG_OK = 0,
    FG_ERR_INVALID = -1,      /* bad argument / shape (the Lua shim turns this into error()) */
    FG_ERR_HIP = -2,          /* HIP runtime error, text in fg_last_error */
    FG_ERR_NOMEM = -3,
    FG_ERR_UNSUPPORTED = -4,  /* layer pattern or kernel variant not built */
    FG_ERR_WORKSPACE = -5     /* caller workspace too small */
};

/* Arithmetic of the large convolution contractions (forward / data-gradient of layers that fill the chip with 256x128
 * tiles).  mode 0 (default): native fp32 MFMA (v_mfma_f32_32x32x2_f32).  mode 6: fp32 emulated on the bf16 matrix pipe --
 * operands split exactly into three bf16 planes, six exact plane products accumulated in fp32 (dropped terms <= 2^-24
 * relative); measured more accurate than mode 0 against fp64 and ~1.7x faster.  Domain: finite operands below 2^127 in
 * magnitude (an infinity, or a value that rounds to the bf16 infinity, splits into inf - inf = NaN where mode 0 would
 * propagate inf); operands below 2^-110 lose their low plane to underflow (relative error up to 2^-16 on those).  Replaces nothing in the reference (its
 * cudnn backend picks algorithms internally, models.lua:1-2); exposed so parity can be run in both modes.
 * PRECEDENCE: the Winograd bits of fg_set_fusion win over this switch -- the forward / data gradient of every layer that
 * FG_FUSE_WINOGRAD / _UP / _5X5 cover (3x3, folded up-convolutions, 5x5: most of the FLOPs of both workloads) runs the fp32 Winograd
 * kernels in either mode, and those round ~1.7x worse than the direct fp32 contraction.  A true mode-6 run (the accuracy contract
 * above on every large contraction) needs fg_set_fusion(flags & ~(FG_FUSE_WINOGRAD | FG_FUSE_WINOGRAD_UP | FG_FUSE_WINOGRAD_5X5 |
 * FG_FUSE_WINOGRAD_WGRAD)) before the nets are created. */
int fg_set_math(fg_ctx* ctx, int mode);
int fg_get_math(fg_ctx* ctx);

/* Optional kernel fusions / variants (results equal to the un-fused path up to the summation order of a reduction; exposed so
 * the parity tests and the bench can run both ways).  Default: FG_FUSE_DEFAULT; FG_FUSE_PRELU=0 / FG_THIN_SLAB=0 / FG_DEFER_WFINISH=0 /
 * FG_THIN_BIAS=0 / FG_WINO=0 / FG_WINO_UP=0 / FG_WINO_5X5=0 / FG_WINO_WGRAD=0 in the environment clear a bit at fg_ctx_create, FG_ADAM_PACK=1 sets that one.  Replaces nothing in the reference. */
enum {
    FG_FUSE_PRELU = 1,      /* an nn.PReLU between two contraction layers (models_c2f.lua:118-130, 242-255) rides on their
                             * epilogues: forward copy behind the producing layer, backward (+ slope-gradient partials) in
                             * the kernel that produces its output gradient (data gradient / max-pool backward) */
    FG_FUSE_THIN_SLAB = 2,  /* 3x3 convolutions with <= 3 output channels (models.lua:73, 385 backward) on the matrix pipe
                             * in one pass instead of the sliding-window VALU kernel */
    FG_FUSE_WFINISH_BATCH = 4, /* the split-K / parity sums of ALL weight gradients of a backward pass in one launch at its end
                                * (the partials stay in the net's workspace until then) instead of one launch per layer; same
                                * order of additions, bit-identical gradients */
    FG_FUSE_ADAM_PACK = 8,  /* fg_step_D / fg_step_G / fg_gan_update with Adam: penalty + clamp + Adam + the re-pack of every
                             * layer's weights into the kernels' layouts in ONE launch (each pack job takes its weights from the
                             * update of that element) instead of two; bit-identical parameters.  OFF by default: the pack's
                             * patch-wise access pattern slows the seven streams of the update down by more than the saved
                             * pass over the weights (cfg2 4.31 vs 4.27 ms, c2f 38.8 vs 38.6 ms per step) */
    FG_FUSE_THIN_BIAS = 16, /* the bias gradient of a convolution with <= 4 input channels (models.lua:385; models_c2f.lua:123, 244)
                             * from the weight-gradient kernel itself: one idle column of its (tap, channel) axis multiplies the
                             * constant 1, so the pass that streams the output gradient anyway also leaves its per-channel sums;
                             * off: a separate column-sum pass re-reads the tensor (134 MB per layer at 64x64, B = 128).  Same
                             * fp64 final reduction; the fp32 partial sums are formed in a different order (FG_THIN_BIAS=0 clears it) */
    FG_FUSE_WINOGRAD = 32,  /* 3x3 / pad 1 / stride 1 convolutions with channel counts % 8 == 0 on even-sized maps (models.lua:390-400,
                             * models_c2f.lua:124, 247-254): forward and data gradient as Winograd F(2x2, 3x3) on the fp32 matrix pipe
                             * (16 multiplies per 2x2 outputs instead of 36; transforms in fp32, results equal to the direct
                             * convolution to a few fp32 roundings) instead of the 9-tap implicit GEMM.  Read when a net is created
                             * (fg_net_create: its packed weights hold the transformed taps) and per call by the module-level
                             * fg_conv2d_* entries; FG_WINO=0 clears it.  The weight gradient: FG_FUSE_WINOGRAD_WGRAD */
    FG_FUSE_WINOGRAD_UP = 64,   /* the same for nn.SpatialUpSamplingNearest(2) + 5x5 / pad 2 convolutions (models.lua:63-64, 68-69): after
                                 * the tap fold every output parity is a 3x3 / pad 1 convolution of the source image, so forward (four
                                 * parities sharing one input transform) and data gradient (four stride-2 input groups) are Winograd
                                 * contractions as well; FG_WINO_UP=0 clears it */
    FG_FUSE_WINOGRAD_5X5 = 128, /* 5x5 / pad 2 / stride 1 convolutions (models_c2f.lua:125-126) as four 3x3 sub-kernels at tap offsets
                                 * (0 | 3, 0 | 3) of the zero-extended 6x6 window: 64 instead of 100 multiplies per 2x2 outputs;
                                 * FG_WINO_5X5=0 clears it */
    FG_FUSE_WINOGRAD_WGRAD = 256, /* the WEIGHT gradient of the layers the three bits above select, in the Winograd domain as well
                                   * (dL/dU = sum over tiles of A dY A^T (.) B^T d B, 16 instead of 36 multiplies per tile and
                                   * channel pair; G^T . G and the tap scatter in the pass that sums the split partials), where the
                                   * tile grid is 2^a x 2^b and the layer has enough tiles to reduce over; else, and with the bit
                                   * cleared (FG_WINO_WGRAD=0), the tap-by-tap contraction.  Read per call */
    FG_FUSE_ALL = 511,
    FG_FUSE_DEFAULT = 503
};
int fg_set_fusion(fg_ctx* ctx, int flags);
int fg_get_fusion(fg_ctx* ctx);
/* Test hook (parity tests only; replaces nothing in the reference): the two planning thresholds that decide where the Winograd-domain
 * weight gradient is taken -- at least `min_chunks` eight-tile chunks per block and `min_blocks` blocks per launch (defaults 24, 192;
 * <= 0 restores a default).  Process-wide; also read ONCE from FG_WINO_WGRAD_MIN_CHUNKS / FG_WINO_WGRAD_MIN_BLOCKS at the first
 * fg_ctx_create.  Lets small shapes reach the kernel's corners (one chunk, ragged last chunk, a single channel block). */
int fg_test_set_wino_wgrad_thresholds(fg_ctx* ctx, long long min_chunks, long long min_blocks);

/* ---- context / memory (replaces cutorch.setDevice / cutorch streams, train.lua:79-80) ----
 * fg_ctx_create(FG_DEVICE_NONE, ...) makes a PLANNING-ONLY context: the process needs no HIP device, no kernel is launched and no
 * HIP runtime call is made, while every host-side decision of the library -- plan building, stage walks, sync-BN pauses, gradient
 * buckets, and the order / size / stream of every collective a step issues (fg_comm_create_dry + fg_comm_schedule) -- runs
 * unchanged.  Device pointers are then opaque tokens (host allocations of the right size that nothing dereferences); fg_malloc
 * returns host memory, fg_h2d / fg_d2h / fg_d2d are plain copies.  A process holds either planning-only contexts or real ones. */
enum { FG_DEVICE_NONE = -1 };
int fg_ctx_create(int device, fg_ctx** out);
int fg_ctx_destroy(fg_ctx* ctx);
int fg_ctx_set_stream(fg_ctx* ctx, void* hip_stream); /* NULL = default stream */
const char* fg_last_error(const fg_ctx* ctx);
const char* fg_version(void);
int fg_stream_sync(fg_ctx* ctx);
/* per-launch HIP-event timing of the contraction kernels on the context's stream (measurement only).
 * fg_prof_report synchronises and writes "label calls total_ms algorithmic_flops executed_flops bytes" lines. */
int fg_prof_enable(fg_ctx* ctx, int on);
int fg_prof_report(fg_ctx* ctx, char* buf, size_t len, int reset);
/* the shader clock the chip grants WHILE other work runs (measurement only): fg_prof_clock_start puts a one-wave probe on a
 * stream of its own that sleeps for about `ms` milliseconds between two readings of the shader-cycle counter (s_memtime) and of
 * the constant 100 MHz counter (s_memrealtime); fg_prof_clock_read waits for it and returns cycles / time in GHz and the time
 * the probe covered.  gfx950 clocks to its power budget: a contraction loop is granted 1.9 - 2.2 GHz of the nominal 2.4. */
int fg_prof_clock_start(fg_ctx* ctx, double ms);
int fg_prof_clock_read(fg_ctx* ctx, double* ghz, double* covered_ms);
int fg_malloc(fg_ctx* ctx, size_t bytes, void** out);
int fg_free(fg_ctx* ctx, void* p);
int fg_h2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_d2h(fg_ctx* ctx, void* dst, const void* src, size_t bytes); /* synchronises */
int fg_d2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_fill(fg_ctx* ctx, float* p, float value, long long n);
int fg_axpby(fg_ctx* ctx, float a, const float* x, float b, float* y, long long n); /* y = a*x + b*y */

/* ---- layout boundary: nn.Copy Float<->device of NN_UTILS.activateCuda (nn_utils.lua:355-362) ---- */
int fg_nchw_to_nhwc(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);
int fg_nhwc_to_nchw(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);

/* ---- RNG: torch.Tensor:uniform / :bernoulli / randn (nn_utils.lua:9, 37; nn.Dropout) -- Philox4x32-10 ---- */
int fg_rng_uniform(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi);
int fg_rng_bernoulli(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float keep_prob);
int fg_rng_normal(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float mean, float std);

/* ---- net-level: an nn.Sequential compiled to a device plan (models.lua:57-81, 382-416) ---- */
enum fg_layer_type {
    FG_LINEAR = 1,          /* nn.Linear(a=in, b=out) */
    FG_VIEW = 2,            /* nn.View(a=C, b=H, c=W)  or nn.View(a=features) with b=c=0 */
    FG_PRELU = 3,           /* nn.PReLU() -- one shared slope */
    FG_UPSAMPLE2X = 4,      /* nn.SpatialUpSamplingNearest(2) */
    FG_CONV = 5,            /* (cudnn|nn).SpatialConvolution(a=nIn, b=nOut, c=k, k, dW, dH, d=pad, pad); p = stride dW=dH (0/1: 1, 2: 2);
                             * q = factor f of cudnn.SpatialConvolutionUpsample (0/1: none; f > 1: b = nOutputPlane * f * f planes, the
                             * NCHW output re-viewed as [b / f^2][H * f][W * f], layers/cudnnSpatialConvolutionUpsample.lua:14-31) */
    FG_BATCHNORM = 6,       /* nn.SpatialBatchNormalization(a=nF), p=eps, q=momentum */
    FG_SPATIAL_DROPOUT = 7, /* nn.SpatialDropout(p) */
    FG_AVGPOOL2 = 8,        /* nn.SpatialAveragePooling(2,2,2,2) */
    FG_DROPOUT = 9,         /* nn.Dropout(p) (v2) */
    FG_SIGMOID = 10,        /* nn.Sigmoid */
    FG_LEAKYRELU = 11,      /* LeakyReLU.lua, p = negative slope */
    FG_MAXPOOL2 = 12        /* nn.SpatialMaxPooling(2,2) (models_c2f.lua:251, 256) */
};
typedef struct fg_layer_spec {
    int type;
    int a, b, c, d;
    float p, q;
} fg_layer_spec;

int fg_net_create(fg_ctx* ctx, const fg_layer_spec* layers, int n_layers, int in_c, int in_h, int in_w, fg_net** out);
int fg_net_destroy(fg_net* net);
long long fg_net_num_params(const fg_net* net);   /* length of the flat parameter vector (reference order) */
long long fg_net_num_buffers(const fg_net* net);  /* BN running_mean||running_var per BN layer, module order */
int fg_net_num_masks(const fg_net* net);          /* dropout layers, module order */
long long fg_net_mask_elems(const fg_net* net, int mask_index, int batch);
float fg_net_mask_keep(const fg_net* net, int mask_index);   /* Bernoulli keep probability 1 - p of that dropout layer */
int fg_net_out_dims(const fg_net* net, int* c, int* h, int* w);
size_t fg_net_workspace_bytes(const fg_net* net, int max_batch);
/* offsets (in floats) of layer `layer_index`'s weight / bias inside the flat vector; -1 if it has none */
int fg_net_param_offset(const fg_net* net, int layer_index, long long* weight_off, long long* weight_n,
                        long long* bias_off, long long* bias_n);
/* persistent device pointers: flat params, flat grads (may be NULL for inference), BN buffers (may be NULL) */
int fg_net_bind(fg_net* net, float* params, float* grads, float* buffers);
int fg_net_params_changed(fg_net* net); /* call after the optimizer touched `params` (re-packs lazily) */
/* Module:forward.  x: NHWC [batch][in_h][in_w][in_c].  masks: one device pointer per dropout layer
 * (0/1 keep masks: SpatialDropout [batch][C]; Dropout [batch][features in internal NHWC order]); ignored when
 * train == 0.  Activations stay in `ws` for fg_net_backward.  *out_offset = float offset of the output
 * (NHWC [batch][out_h][out_w][out_c]) inside ws. */
int fg_net_forward(fg_net* net, int batch, const float* x, void* ws, size_t ws_bytes, int train,
                   const float* const* masks, int n_masks, long long* out_offset);
/* fg_net_forward whose LAST stage writes its output to `out` (device, NHWC, 16-byte aligned) instead of into the workspace:
 * a producer net hands its result straight to the consumer's input buffer (G -> D's batch, adversarial.lua:252-256). */
int fg_net_forward_to(fg_net* net, int batch, const float* x, void* ws, size_t ws_bytes, int train,
                      const float* const* masks, int n_masks, long long* out_offset, float* out);
int fg_net_in_dims(const fg_net* net, int* c, int* h, int* w);
int fg_net_vectors(const fg_net* net, float** params, float** grads, float** buffers);   /* what fg_net_bind attached */
int fg_net_max_bn_channels(const fg_net* net);
enum { FG_BWD_PARAM_GRADS = 1, FG_BWD_INPUT_GRAD = 2 };
/* Module:backward after a train-mode forward with the same (batch, x, ws).  gy: grad wrt the output.
 * FG_BWD_PARAM_GRADS writes (overwrites: the reference zeroes before every feval, adversarial.lua:92, 193)
 * the flat gradient vector; FG_BWD_INPUT_GRAD writes gx (NHWC like x). */
int fg_net_backward(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags,
                    float* gx);
/* Bucketed backward for data parallelism: the plan's stages run output -> input; fg_net_backward_range executes stages
 * [stage_from .. stage_to] (descending; the first call starts at fg_net_num_stages()-1 with gy, later calls continue
 * where the previous one stopped and ignore gy), so the host can start the RCCL all-reduce of the gradient range
 * fg_net_stage_params() reports for the finished stages while the remaining stages still compute. */
int fg_net_num_stages(const fg_net* net);
int fg_net_stage_params(const fg_net* net, int stage, long long* param_lo, long long* param_hi);
int fg_net_backward_range(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags,
                          float* gx, int stage_from, int stage_to);
/* sync-BN (data parallelism with exact global-batch BatchNorm statistics): with sync on, fg_net_forward /
 * fg_net_backward[_range] return FG_PAUSED_SYNC (= 1) at every SpatialBatchNormalization after leaving
 * fg_net_sync_count() fp64 values [sum x | sum x^2 | rows] (backward: [sum dz | sum dz*xhat | rows]) in `sync_buf`;
 * the host sum-all-reduces that buffer across ranks and calls the matching *_resume, until FG_OK. */
enum { FG_PAUSED_SYNC = 1 };
int fg_net_set_sync_bn(fg_net* net, int on, double* sync_buf_dev, long long capacity_doubles);
long long fg_net_sync_count(const fg_net* net);
int fg_net_forward_resume(fg_net* net, long long* out_offset);
int fg_net_backward_resume(fg_net* net);
/* debugging / parity: activation after reference layer `layer_index` of the last forward (must end a stage) */
int fg_net_layer_output(const fg_net* net, int layer_index, long long* ws_offset, int* c, int* h, int* w);
/* debugging / parity: where the last train-mode forward left the batch mean / 1/sqrt(var + eps) of BatchNorm layer
 * `layer_index` inside ws (c floats each) -- what the backward pass recomputes the PReLU branch from */
int fg_net_bn_saved_stats(const fg_net* net, int layer_index, long long* mean_offset, long long* invstd_offset, int* c);

/* ---- criterion: nn.BCECriterion() (train.lua:148) fused forward+backward, plus D's confusion counts
 *      (adversarial.lua:112-117): confusion[pred*2 + target] ---- */
int fg_bce_forward_backward(fg_ctx* ctx, const float* prob, const float* target, int n, float* loss_dev,
                            float* grad_dev, int* confusion_dev);

/* ---- optimizers on the flat vectors (interruptable_optimizers.lua:7-167) with the penalty and clamp of
 *      adversarial.lua:103-123 / 218-228 fused in:  g' = clamp(gscale*g + l1_mul*sign(p) + l2*p, +-clamp).
 *      Hyper-parameters are doubles (Lua numbers): 1-beta, the bias corrections and the step size are evaluated in
 *      double on the host exactly like interruptable_optimizers.lua:78-88, then rounded to fp32 once. ---- */
int fg_adam_fused(fg_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float gscale, float l1_mul,
                  float l2, float clamp, double lr, double beta1, double beta2, double eps, int t, float* g_out);
int fg_sgd_fused(fg_ctx* ctx, float* p, const float* g, float* mom_buf, long long n, float gscale, float l1_mul,
                 float l2, float clamp, double lr, double momentum, double dampening, double weight_decay, int nesterov,
                 int first_step);
int fg_adagrad_fused(fg_ctx* ctx, float* p, const float* g, float* variance, long long n, float gscale, float l1_mul,
                     float l2, float clamp, double clr);
/* out2[0] = ||p||_1, out2[1] = ||p||_2^2 (torch.norm of adversarial.lua:105-106); scratch >= 1024 floats */
int fg_norms(fg_ctx* ctx, const float* p, long long n, float* out2_dev, float* scratch);

/* ---- data parallelism: one process (rank) per GPU, RCCL over xGMI (SURVEY.md 8(b) `fg_allreduce_sum(fg_comm*, ...)`, 8(e)).
 *      The reference has no multi-GPU path (train.lua:79 selects one device); BASELINE configs 3 and 5 shard the batch
 *      and sum-all-reduce the flat gradient vector of the net being updated.  librccl is bound at run time (dlopen; an
 *      instance already loaded by the host process is shared; FG_RCCL_LIB overrides the search).
 *      Bootstrap: rank 0 calls fg_comm_unique_id and hands the FG_COMM_ID_BYTES bytes to every rank through any host
 *      channel (file, environment, socket, torch.distributed store); every rank then calls fg_comm_create, which is
 *      collective.  fg_allreduce_sum* / fg_broadcast are in-place and stream-ordered on the context's stream;
 *      fg_allreduce_sum_async runs the exchange on the communicator's own stream after everything enqueued so far and
 *      returns at once -- fg_comm_wait makes the context's stream wait for every exchange issued that way. ---- */
enum { FG_COMM_ID_BYTES = 128 };
int fg_comm_unique_id(fg_ctx* ctx, char* id_out, size_t len);
int fg_comm_create(fg_ctx* ctx, const char* id, size_t len, int rank, int world, fg_comm** out);
int fg_comm_destroy(fg_comm* comm);
int fg_comm_rank(const fg_comm* comm);
int fg_comm_world(const fg_comm* comm);
const char* fg_comm_library(void);   /* which librccl was bound ("" before the first fg_comm_* call) */
int fg_allreduce_sum(fg_comm* comm, float* buf, size_t n);
int fg_allreduce_sum_async(fg_comm* comm, float* buf, size_t n);
int fg_comm_wait(fg_comm* comm);
int fg_allreduce_sum_f64(fg_comm* comm, double* buf, size_t n);   /* sync-BN sums */
int fg_allreduce_sum_i32(fg_comm* comm, int* buf, size_t n);      /* confusion counts of the global batch */
int fg_broadcast(fg_comm* comm, float* buf, size_t n, int root);  /* identical initial replicas */
/* The collective schedule.  fg_comm_set_trace(comm, 1) records every exchange call made on the communicator from then on;
 * fg_comm_schedule writes one line per call, "<seq> <op> <dtype> <count> <stream>" -- op allreduce / broadcast / wait, stream
 * "compute" (the context's stream) or "side" (the communicator's own) -- and optionally clears the record.  Every rank of a job
 * must produce the same text: a mismatch is a hang on real hardware.
 * fg_comm_create_dry builds a communicator of `world` ranks with NO transport underneath: its collectives are recorded (tracing is
 * on from the start) and otherwise skipped, so one process -- with a planning-only context not even a GPU -- can walk the exact
 * exchange path rank `rank` of an N-GPU job takes inside fg_step_D / fg_step_G (bench.py --dry-collective). */
int fg_comm_create_dry(fg_ctx* ctx, int rank, int world, fg_comm** out);
int fg_comm_set_trace(fg_comm* comm, int on);
int fg_comm_schedule(fg_comm* comm, char* buf, size_t len, int reset);

/* ---- step level (SURVEY.md 8(b) level (ii)): the two closures of the training loop as device-resident entries.
 *      fg_step_D = adversarial.lua:240-268 (batch assembly: B/2 real || B/2 fakes from G in TRAIN mode) + fevalD (:83-179:
 *                  D forward, BCECriterion, D backward, confusion counts) + penalty / clamp (:103-123) + the optimizer
 *                  (interruptable_optimizers.lua) on D's flat vectors;
 *      fg_step_G = adversarial.lua:275-288 + fevalG_on_D (:187-231): G forward on B noises, D forward, BCE against all
 *                  ones, D backward to its input only (MODEL_D.modules[1].gradInput, :210), G backward, optimizer on G.
 *      table_inputs = 1: adversarial_c2f.lua:40-187 -- G{noise[B,S,S,1], cond[B,S,S,C]} (JoinTable) -> diff image,
 *                  D{x, cond} (CAddTable); cond_* are device NHWC.
 *      The nets are fg_net objects bound to their flat vectors (fg_net_bind); their workspaces are the caller's
 *      (fg_gan_bind_workspaces: fg_net_workspace_bytes(net, max_batch) each) plus one workspace of the step object itself.
 *      noise / masks == NULL: drawn by the library, all of a closure in one Philox launch (fg_gan_set_seeds); otherwise
 *      noise = device [rows][noiseDim] (table mode [rows][S][S][1]), masks = fg_net_num_masks(D) device pointers.
 *      FG_STEP_NO_UPDATE: stop before the optimizer (gradients stay in the bound gradient vector, un-reduced);
 *      fg_gan_update applies it later -- the maxAccuracyD gate of adversarial.lua:124-178 reads the confusion counts in
 *      between (FG_GAN_CONFUSION: int[0..3] this rank, int[4..7] the global batch, both [pred*2 + target]); never calling
 *      fg_gan_update IS interruptableAdam's false,false path.
 *      fg_gan_set_comm: data parallelism -- gradients are sum-all-reduced over the communicator, scaled by 1/world inside
 *      the optimizer pass, clamp after the reduce; D's exchange overlaps the next generator forward (its update is deferred
 *      until D is next evaluated or fg_gan_finish_pending), G's is bucketed under its own backward; sync_bn = exact
 *      global-batch BatchNorm statistics (fp64 sums reduced at every BatchNorm).  overlap: 0 = blocking exchange, 1 =
 *      overlapped, 2 = overlapped even on a one-rank communicator (exercises the N > 1 path on one GPU). ---- */
enum { FG_STEP_NO_UPDATE = 1 };
enum fg_gan_buffer_id {
    FG_GAN_D_INPUT = 0,       /* D's batch, NHWC [B][H][W][C]: real || fake (D-step), G's samples (G-step)            */
    FG_GAN_NOISE = 1,         /* the noise the last closure used when the library drew it                              */
    FG_GAN_D_GRAD_INPUT = 2,  /* G-step: d loss / d samples                                                           */
    FG_GAN_LOSS = 3,          /* float[2]: BCE of the last D-step, of the last G-step                                  */
    FG_GAN_CONFUSION = 4,     /* int[8] (same storage as floats): local counts, global counts                         */
    FG_GAN_OPT_STATE_D = 5,   /* 2 * nparams(D): Adam m | v   (SGD: momentum buffer; Adagrad: variance)                */
    FG_GAN_OPT_STATE_G = 6,
    FG_GAN_D_OUTPUT = 7,      /* D's probabilities [B] of the last closure; offset relative to D's OWN workspace       */
    FG_GAN_D_MASKS = 8,       /* dropout masks the library drew (fg_gan_mask_offset per mask)                          */
    FG_GAN_SYNC_BUF = 9       /* sync-BN exchange buffer (doubles; count is in floats): what fg_gan_set_comm bound to both nets */
};
size_t fg_gan_workspace_bytes(const fg_net* G, const fg_net* D, int table_inputs, int max_batch);
int fg_gan_create(fg_ctx* ctx, fg_net* G, fg_net* D, int table_inputs, int max_batch, void* ws, size_t ws_bytes,
                  fg_gan** out);
int fg_gan_destroy(fg_gan* gan);
int fg_gan_bind_workspaces(fg_gan* gan, void* wsG, size_t wsG_bytes, void* wsD, size_t wsD_bytes);
int fg_gan_set_comm(fg_gan* gan, fg_comm* comm, int sync_bn, int overlap);
int fg_gan_set_seeds(fg_gan* gan, uint64_t noise_seed, uint64_t noise_offset, uint64_t mask_seed, uint64_t mask_offset);
int fg_gan_set_penalty(fg_gan* gan, int which, float l1, float l2, float clamp);             /* which: 0 = D, 1 = G */
/* method 0 adam / 1 sgd / 2 adagrad (OPT.*_optmethod); lr < 0: the rule's default (1e-3); dampening < 0: = momentum */
int fg_gan_set_optimizer(fg_gan* gan, int which, int method, double lr, double beta1, double beta2, double eps,
                         double momentum, double dampening, double weight_decay, double lr_decay, int nesterov);
int fg_gan_optimizer_steps(const fg_gan* gan, int which);          /* Adam's t / optim.sgd's evalCounter */
int fg_gan_set_optimizer_steps(fg_gan* gan, int which, int steps);
/* offset (in floats, relative to the step workspace unless noted) and length of a result / state buffer */
int fg_gan_buffer(const fg_gan* gan, int what, long long* offset_floats, long long* count);
long long fg_gan_mask_offset(const fg_gan* gan, int mask_index);
int fg_step_D(fg_gan* gan, int batch, const float* real, const float* cond_real, const float* cond_fake,
              const float* noise, const float* const* masks, int flags);
int fg_step_G(fg_gan* gan, int batch, const float* cond, const float* noise, const float* const* masks, int flags);
int fg_gan_update(fg_gan* gan, int which);
int fg_gan_finish_pending(fg_gan* gan);
int fg_gan_pending(const fg_gan* gan);   /* 1 while D's update is deferred behind its gradient all-reduce */

/* ---- adversarial.approxParzen (adversarial_c2f.lua:305-344): dist[i] = || (gen[i] + cond) - fine ||_2 for the n
 *      generations of one example (torch.dist: squares accumulated in double), min_out[0] = min(1e10, min_i dist[i]).
 *      gen [n][elems]; cond, fine [elems]; all three in the same element order. ---- */
int fg_parzen_min_dist(fg_ctx* ctx, const float* gen, const float* cond, const float* fine, int n, long long elems,
                       float* dist, float* min_out);

/* ---- the c2f data step in front of the train closure (dataset_c2f.lua:49-61, `dataset._toResult`) ----
 * fg_scale_bilinear = image.scale(src, width, height) of the Torch7 `image` package (default mode 'bilinear'; un-vendored luarocks
 * dependency, restated in oracle/image_scale.py): width pass then height pass; per axis up-scaling interpolates with
 * (src - 1) / (dst - 1) ("corners onto corners"), down-scaling is a box mean with fractional end coverage, equal sizes copy.
 * Bit-for-bit the C float loop.  src [n][hs][ws][c] (layout 0 = NHWC, the library's activation layout) or [n][c][hs][ws]
 * (layout 1 = NCHW, the reference's image tensors); dst likewise at hd x wd.
 * fg_c2f_coarse_diff = the whole step: coarse = scale(scale(fine, cs, cs), s, s), diff = fine - coarse, for n square images of
 * side s; tmp holds n * c * cs * cs floats. */
int fg_scale_bilinear(fg_ctx* ctx, const float* src, float* dst, int n, int c, int hs, int ws, int hd, int wd, int layout);
int fg_c2f_coarse_diff(fg_ctx* ctx, const float* fine, float* coarse, float* diff, float* tmp, int n, int c, int s, int cs,
                       int layout);

/* ---- module-level ops (nn.Module protocol: updateOutput / updateGradInput / accGradParameters), NHWC ----
 * conv / linear take REFERENCE-layout weights and pack them into `ws` on the fly. */
size_t fg_conv2d_workspace_bytes(int batch, int h, int w, int cin, int cout, int k, int upsample2x);
int fg_conv2d_forward(fg_ctx* ctx, const float* x, const float* w_oihw, const float* bias, float* y, int batch, int h,
                      int w, int cin, int cout, int k, int pad, int upsample2x, void* ws, size_t ws_bytes);
int fg_conv2d_backward_data(fg_ctx* ctx, const float* gy, const float* w_oihw, float* gx, int batch, int h, int w,
                            int cin, int cout, int k, int pad, int upsample2x, void* ws, size_t ws_bytes);
/* gw = beta*gw + dW, gb = beta*gb + db (beta = 1: Torch's accumulate semantics) */
int fg_conv2d_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw_oihw, float* gb, float beta,
                              int batch, int h, int w, int cin, int cout, int k, int pad, int upsample2x, void* ws,
                              size_t ws_bytes);
size_t fg_linear_workspace_bytes(int batch, int in_f, int out_f);
int fg_linear_forward(fg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int batch, int in_f,
                      int out_f, void* ws, size_t ws_bytes);
int fg_linear_backward_data(fg_ctx* ctx, const float* gy, const float* w, float* gx, int batch, int in_f, int out_f,
                            void* ws, size_t ws_bytes);
int fg_linear_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, float beta,
                              int batch, int in_f, int out_f, void* ws, size_t ws_bytes);
/* SpatialBatchNormalization; slope != NULL fuses the following PReLU.  scratch >= fg_bn_scratch_floats(c) floats */
long long fg_bn_scratch_floats(int c);
int fg_batchnorm_forward(fg_ctx* ctx, const float* x, float* y, long long rows, int c, const float* gamma,
                         const float* beta, const float* slope, float* save_mean, float* save_invstd,
                         float* running_mean, float* running_var, float eps, float momentum, int train,
                         float* scratch);
int fg_batchnorm_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, long long rows, int c,
                          const float* gamma, const float* beta, const float* slope, const float* save_mean,
                          const float* save_invstd, float* ggamma, float* gbeta, float* gslope, float acc,
                          float* scratch);
/* PReLU with an optional same-shape keep mask (x mscale): nn.PReLU [+ nn.Dropout]; scratch >= 1024 floats */
int fg_prelu_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y,
                     long long n);
int fg_prelu_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask,
                      float mscale, float* gx, float* gslope, float acc, long long n, float* scratch);
/* fused nn.PReLU -> nn.SpatialDropout(mask [batch][c]) -> nn.SpatialAveragePooling(2,2,2,2) (models.lua:386-388) */
int fg_actpool_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y,
                       int batch, int h, int w, int c);
int fg_actpool_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask,
                        float mscale, float* gx, float* gslope, float acc, int batch, int h, int w, int c,
                        float* scratch);
int fg_spatial_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float mscale, float* y, int batch, int hw,
                             int c);
int fg_avgpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_avgpool2x2_backward(fg_ctx* ctx, const float* gy, float* gx, int batch, int h, int w, int c);
int fg_upsample_nearest2x_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_upsample_nearest2x_backward(fg_ctx* ctx, const float* gy, float* gx, int batch, int h, int w, int c);
/* cudnn.SpatialConvolutionUpsample, factor f > 1 (layers/cudnnSpatialConvolutionUpsample.lua:19-58): the flat re-view of the
 * convolution's NCHW output [batch][c][h][w] (c = nOutputPlane * f * f) as [batch][c / f^2][h * f][w * f], on NHWC tensors.
 * forward: conv output -> viewed output; backward: gradient wrt the viewed output -> gradient wrt the conv output. */
int fg_conv_upsample_view_forward(fg_ctx* ctx, const float* conv_out, float* viewed, int batch, int h, int w, int c, int factor);
int fg_conv_upsample_view_backward(fg_ctx* ctx, const float* g_viewed, float* g_conv_out, int batch, int h, int w, int c, int factor);
/* nn.SpatialMaxPooling(2,2): backward recomputes the argmax (first max in scan order) from the saved input */
int fg_maxpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_maxpool2x2_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, int batch, int h, int w, int c);
/* nn.Dropout on any shape: y = x * mask * scale (mask NULL: y = x * scale) */
int fg_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float scale, float* y, long long n);
/* nn.JoinTable(2,2) / its backward split, nn.CAddTable on NHWC tensors (models_c2f.lua:116, 240) */
int fg_concat_channels(fg_ctx* ctx, const float* a, const float* b, float* out, long long npix, int ca, int cb);
int fg_split_channels(fg_ctx* ctx, const float* g, float* ga, float* gb, long long npix, int ca, int cb);
int fg_add(fg_ctx* ctx, const float* a, const float* b, float* out, long long n);
int fg_sigmoid_forward(fg_ctx* ctx, const float* x, float* y, long long n);
int fg_sigmoid_backward(fg_ctx* ctx, const float* y, const float* gy, float* gx, long long n);
int fg_leakyrelu_forward(fg_ctx* ctx, const float* x, float negslope, float* y, long long n);
int fg_leakyrelu_backward(fg_ctx* ctx, const float* x, const float* gy, float negslope, float* gx, long long n);

#ifdef __cplusplus
}
#endif
#endif /* FACEGEN_HIP_H */
