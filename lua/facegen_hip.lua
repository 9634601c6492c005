--[[ facegen_hip.lua -- LuaJIT-FFI binding of libfacegen_hip.so for the reference's Lua/Torch7 host.

NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (the build image has no Lua/LuaJIT/Torch7; SURVEY.md F6).  What IS checked
mechanically (tests/test_lua_binding.py): the ffi.cdef block below is generated from include/facegen_hip.h
(scripts/gen_lua_cdef.py) and must be up to date; every `C.fg_*(...)` call in lua/*.lua names a declared entry with the
declared number of arguments; every `C.FG_*` constant exists in the header; every optimizer path marks the packed weights
stale (fg_net_params_changed / the step object does it itself).  The Python package face_generator_amd/ drives the SAME
entry points through ctypes and is what the GPU tests execute.

Two levels, as SURVEY.md 8(b) asks:
  (i)  module level  -- M.attach(seq, dims, batch) turns the nn.Sequential inside NN_UTILS.activateCuda's
       {Copy, net, Copy} into a device plan: net:forward / :backward / :training / :evaluate / :getParameters keep their
       nn protocol on host FloatTensors, so the reference's adversarial.lua closures and interruptable_optimizers.lua run
       unchanged (parameters are re-uploaded before a forward whenever the host copy may have changed).
  (ii) step level    -- M.Gan(...) wraps fg_gan_*: one call per closure (fg_step_D / fg_step_G); lua/adversarial_hip.lua
       is adversarial.train re-hosted on it (same signature, same globals).

Replaces, on the hot path only: cutorch.setDevice/manualSeed (train.lua:79-80), :cuda() / nn.Copy
(nn_utils.lua:355-362), MODEL:forward/backward (adversarial.lua:95-100, 202-214), nn.BCECriterion (train.lua:148),
the optimizer tensor math (interruptable_optimizers.lua:78-90) and the penalty/clamp (adversarial.lua:103-123).
]]
local ffi = require 'ffi'

-- BEGIN GENERATED CDEF (scripts/gen_lua_cdef.py)
ffi.cdef[[
typedef struct fg_ctx fg_ctx;
typedef struct fg_net fg_net;
typedef struct fg_comm fg_comm;
typedef struct fg_gan fg_gan;
enum { FG_OK = 0, FG_ERR_INVALID = -1, FG_ERR_HIP = -2, FG_ERR_NOMEM = -3, FG_ERR_UNSUPPORTED = -4, FG_ERR_WORKSPACE = -5 };
int fg_set_math(fg_ctx* ctx, int mode);
int fg_get_math(fg_ctx* ctx);
enum { FG_FUSE_PRELU = 1, FG_FUSE_THIN_SLAB = 2, FG_FUSE_WFINISH_BATCH = 4, FG_FUSE_ADAM_PACK = 8, FG_FUSE_THIN_BIAS = 16, FG_FUSE_WINOGRAD = 32, FG_FUSE_WINOGRAD_UP = 64, FG_FUSE_WINOGRAD_5X5 = 128, FG_FUSE_WINOGRAD_WGRAD = 256, FG_FUSE_ALL = 511, FG_FUSE_DEFAULT = 503 };
int fg_set_fusion(fg_ctx* ctx, int flags);
int fg_get_fusion(fg_ctx* ctx);
int fg_test_set_wino_wgrad_thresholds(fg_ctx* ctx, long long min_chunks, long long min_blocks);
enum { FG_DEVICE_NONE = -1 };
int fg_ctx_create(int device, fg_ctx** out);
int fg_ctx_destroy(fg_ctx* ctx);
int fg_ctx_set_stream(fg_ctx* ctx, void* hip_stream);
const char* fg_last_error(const fg_ctx* ctx);
const char* fg_version(void);
int fg_stream_sync(fg_ctx* ctx);
int fg_prof_enable(fg_ctx* ctx, int on);
int fg_prof_report(fg_ctx* ctx, char* buf, size_t len, int reset);
int fg_prof_clock_start(fg_ctx* ctx, double ms);
int fg_prof_clock_read(fg_ctx* ctx, double* ghz, double* covered_ms);
int fg_malloc(fg_ctx* ctx, size_t bytes, void** out);
int fg_free(fg_ctx* ctx, void* p);
int fg_h2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_d2h(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_d2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_fill(fg_ctx* ctx, float* p, float value, long long n);
int fg_axpby(fg_ctx* ctx, float a, const float* x, float b, float* y, long long n);
int fg_nchw_to_nhwc(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);
int fg_nhwc_to_nchw(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);
int fg_rng_uniform(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi);
int fg_rng_bernoulli(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float keep_prob);
int fg_rng_normal(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float mean, float std);
enum fg_layer_type { FG_LINEAR = 1, FG_VIEW = 2, FG_PRELU = 3, FG_UPSAMPLE2X = 4, FG_CONV = 5, FG_BATCHNORM = 6, FG_SPATIAL_DROPOUT = 7, FG_AVGPOOL2 = 8, FG_DROPOUT = 9, FG_SIGMOID = 10, FG_LEAKYRELU = 11, FG_MAXPOOL2 = 12 };
typedef struct fg_layer_spec { int type; int a, b, c, d; float p, q; } fg_layer_spec;
int fg_net_create(fg_ctx* ctx, const fg_layer_spec* layers, int n_layers, int in_c, int in_h, int in_w, fg_net** out);
int fg_net_destroy(fg_net* net);
long long fg_net_num_params(const fg_net* net);
long long fg_net_num_buffers(const fg_net* net);
int fg_net_num_masks(const fg_net* net);
long long fg_net_mask_elems(const fg_net* net, int mask_index, int batch);
float fg_net_mask_keep(const fg_net* net, int mask_index);
int fg_net_out_dims(const fg_net* net, int* c, int* h, int* w);
size_t fg_net_workspace_bytes(const fg_net* net, int max_batch);
int fg_net_param_offset(const fg_net* net, int layer_index, long long* weight_off, long long* weight_n, long long* bias_off, long long* bias_n);
int fg_net_bind(fg_net* net, float* params, float* grads, float* buffers);
int fg_net_params_changed(fg_net* net);
int fg_net_forward(fg_net* net, int batch, const float* x, void* ws, size_t ws_bytes, int train, const float* const* masks, int n_masks, long long* out_offset);
int fg_net_forward_to(fg_net* net, int batch, const float* x, void* ws, size_t ws_bytes, int train, const float* const* masks, int n_masks, long long* out_offset, float* out);
int fg_net_in_dims(const fg_net* net, int* c, int* h, int* w);
int fg_net_vectors(const fg_net* net, float** params, float** grads, float** buffers);
int fg_net_max_bn_channels(const fg_net* net);
enum { FG_BWD_PARAM_GRADS = 1, FG_BWD_INPUT_GRAD = 2 };
int fg_net_backward(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags, float* gx);
int fg_net_num_stages(const fg_net* net);
int fg_net_stage_params(const fg_net* net, int stage, long long* param_lo, long long* param_hi);
int fg_net_backward_range(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags, float* gx, int stage_from, int stage_to);
enum { FG_PAUSED_SYNC = 1 };
int fg_net_set_sync_bn(fg_net* net, int on, double* sync_buf_dev, long long capacity_doubles);
long long fg_net_sync_count(const fg_net* net);
int fg_net_forward_resume(fg_net* net, long long* out_offset);
int fg_net_backward_resume(fg_net* net);
int fg_net_layer_output(const fg_net* net, int layer_index, long long* ws_offset, int* c, int* h, int* w);
int fg_net_bn_saved_stats(const fg_net* net, int layer_index, long long* mean_offset, long long* invstd_offset, int* c);
int fg_bce_forward_backward(fg_ctx* ctx, const float* prob, const float* target, int n, float* loss_dev, float* grad_dev, int* confusion_dev);
int fg_adam_fused(fg_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float gscale, float l1_mul, float l2, float clamp, double lr, double beta1, double beta2, double eps, int t, float* g_out);
int fg_sgd_fused(fg_ctx* ctx, float* p, const float* g, float* mom_buf, long long n, float gscale, float l1_mul, float l2, float clamp, double lr, double momentum, double dampening, double weight_decay, int nesterov, int first_step);
int fg_adagrad_fused(fg_ctx* ctx, float* p, const float* g, float* variance, long long n, float gscale, float l1_mul, float l2, float clamp, double clr);
int fg_norms(fg_ctx* ctx, const float* p, long long n, float* out2_dev, float* scratch);
enum { FG_COMM_ID_BYTES = 128 };
int fg_comm_unique_id(fg_ctx* ctx, char* id_out, size_t len);
int fg_comm_create(fg_ctx* ctx, const char* id, size_t len, int rank, int world, fg_comm** out);
int fg_comm_destroy(fg_comm* comm);
int fg_comm_rank(const fg_comm* comm);
int fg_comm_world(const fg_comm* comm);
const char* fg_comm_library(void);
int fg_allreduce_sum(fg_comm* comm, float* buf, size_t n);
int fg_allreduce_sum_async(fg_comm* comm, float* buf, size_t n);
int fg_comm_wait(fg_comm* comm);
int fg_allreduce_sum_f64(fg_comm* comm, double* buf, size_t n);
int fg_allreduce_sum_i32(fg_comm* comm, int* buf, size_t n);
int fg_broadcast(fg_comm* comm, float* buf, size_t n, int root);
int fg_comm_create_dry(fg_ctx* ctx, int rank, int world, fg_comm** out);
int fg_comm_set_trace(fg_comm* comm, int on);
int fg_comm_schedule(fg_comm* comm, char* buf, size_t len, int reset);
enum { FG_STEP_NO_UPDATE = 1 };
enum fg_gan_buffer_id { FG_GAN_D_INPUT = 0, FG_GAN_NOISE = 1, FG_GAN_D_GRAD_INPUT = 2, FG_GAN_LOSS = 3, FG_GAN_CONFUSION = 4, FG_GAN_OPT_STATE_D = 5, FG_GAN_OPT_STATE_G = 6, FG_GAN_D_OUTPUT = 7, FG_GAN_D_MASKS = 8, FG_GAN_SYNC_BUF = 9 };
size_t fg_gan_workspace_bytes(const fg_net* G, const fg_net* D, int table_inputs, int max_batch);
int fg_gan_create(fg_ctx* ctx, fg_net* G, fg_net* D, int table_inputs, int max_batch, void* ws, size_t ws_bytes, fg_gan** out);
int fg_gan_destroy(fg_gan* gan);
int fg_gan_bind_workspaces(fg_gan* gan, void* wsG, size_t wsG_bytes, void* wsD, size_t wsD_bytes);
int fg_gan_set_comm(fg_gan* gan, fg_comm* comm, int sync_bn, int overlap);
int fg_gan_set_seeds(fg_gan* gan, uint64_t noise_seed, uint64_t noise_offset, uint64_t mask_seed, uint64_t mask_offset);
int fg_gan_set_penalty(fg_gan* gan, int which, float l1, float l2, float clamp);
int fg_gan_set_optimizer(fg_gan* gan, int which, int method, double lr, double beta1, double beta2, double eps, double momentum, double dampening, double weight_decay, double lr_decay, int nesterov);
int fg_gan_optimizer_steps(const fg_gan* gan, int which);
int fg_gan_set_optimizer_steps(fg_gan* gan, int which, int steps);
int fg_gan_buffer(const fg_gan* gan, int what, long long* offset_floats, long long* count);
long long fg_gan_mask_offset(const fg_gan* gan, int mask_index);
int fg_step_D(fg_gan* gan, int batch, const float* real, const float* cond_real, const float* cond_fake, const float* noise, const float* const* masks, int flags);
int fg_step_G(fg_gan* gan, int batch, const float* cond, const float* noise, const float* const* masks, int flags);
int fg_gan_update(fg_gan* gan, int which);
int fg_gan_finish_pending(fg_gan* gan);
int fg_gan_pending(const fg_gan* gan);
int fg_parzen_min_dist(fg_ctx* ctx, const float* gen, const float* cond, const float* fine, int n, long long elems, float* dist, float* min_out);
int fg_scale_bilinear(fg_ctx* ctx, const float* src, float* dst, int n, int c, int hs, int ws, int hd, int wd, int layout);
int fg_c2f_coarse_diff(fg_ctx* ctx, const float* fine, float* coarse, float* diff, float* tmp, int n, int c, int s, int cs, int layout);
size_t fg_conv2d_workspace_bytes(int batch, int h, int w, int cin, int cout, int k, int upsample2x);
int fg_conv2d_forward(fg_ctx* ctx, const float* x, const float* w_oihw, const float* bias, float* y, int batch, int h, int w, int cin, int cout, int k, int pad, int upsample2x, void* ws, size_t ws_bytes);
int fg_conv2d_backward_data(fg_ctx* ctx, const float* gy, const float* w_oihw, float* gx, int batch, int h, int w, int cin, int cout, int k, int pad, int upsample2x, void* ws, size_t ws_bytes);
int fg_conv2d_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw_oihw, float* gb, float beta, int batch, int h, int w, int cin, int cout, int k, int pad, int upsample2x, void* ws, size_t ws_bytes);
size_t fg_linear_workspace_bytes(int batch, int in_f, int out_f);
int fg_linear_forward(fg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int batch, int in_f, int out_f, void* ws, size_t ws_bytes);
int fg_linear_backward_data(fg_ctx* ctx, const float* gy, const float* w, float* gx, int batch, int in_f, int out_f, void* ws, size_t ws_bytes);
int fg_linear_backward_weight(fg_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, float beta, int batch, int in_f, int out_f, void* ws, size_t ws_bytes);
long long fg_bn_scratch_floats(int c);
int fg_batchnorm_forward(fg_ctx* ctx, const float* x, float* y, long long rows, int c, const float* gamma, const float* beta, const float* slope, float* save_mean, float* save_invstd, float* running_mean, float* running_var, float eps, float momentum, int train, float* scratch);
int fg_batchnorm_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, long long rows, int c, const float* gamma, const float* beta, const float* slope, const float* save_mean, const float* save_invstd, float* ggamma, float* gbeta, float* gslope, float acc, float* scratch);
int fg_prelu_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y, long long n);
int fg_prelu_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask, float mscale, float* gx, float* gslope, float acc, long long n, float* scratch);
int fg_actpool_forward(fg_ctx* ctx, const float* x, const float* slope, const float* mask, float mscale, float* y, int batch, int h, int w, int c);
int fg_actpool_backward(fg_ctx* ctx, const float* x, const float* gy, const float* slope, const float* mask, float mscale, float* gx, float* gslope, float acc, int batch, int h, int w, int c, float* scratch);
int fg_spatial_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float mscale, float* y, int batch, int hw, int c);
int fg_avgpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_avgpool2x2_backward(fg_ctx* ctx, const float* gy, float* gx, int batch, int h, int w, int c);
int fg_upsample_nearest2x_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_upsample_nearest2x_backward(fg_ctx* ctx, const float* gy, float* gx, int batch, int h, int w, int c);
int fg_conv_upsample_view_forward(fg_ctx* ctx, const float* conv_out, float* viewed, int batch, int h, int w, int c, int factor);
int fg_conv_upsample_view_backward(fg_ctx* ctx, const float* g_viewed, float* g_conv_out, int batch, int h, int w, int c, int factor);
int fg_maxpool2x2_forward(fg_ctx* ctx, const float* x, float* y, int batch, int h, int w, int c);
int fg_maxpool2x2_backward(fg_ctx* ctx, const float* x, const float* gy, float* gx, int batch, int h, int w, int c);
int fg_dropout_apply(fg_ctx* ctx, const float* x, const float* mask, float scale, float* y, long long n);
int fg_concat_channels(fg_ctx* ctx, const float* a, const float* b, float* out, long long npix, int ca, int cb);
int fg_split_channels(fg_ctx* ctx, const float* g, float* ga, float* gb, long long npix, int ca, int cb);
int fg_add(fg_ctx* ctx, const float* a, const float* b, float* out, long long n);
int fg_sigmoid_forward(fg_ctx* ctx, const float* x, float* y, long long n);
int fg_sigmoid_backward(fg_ctx* ctx, const float* y, const float* gy, float* gx, long long n);
int fg_leakyrelu_forward(fg_ctx* ctx, const float* x, float negslope, float* y, long long n);
int fg_leakyrelu_backward(fg_ctx* ctx, const float* x, const float* gy, float negslope, float* gx, long long n);
]]
-- END GENERATED CDEF

local C = ffi.load('facegen_hip')      -- libfacegen_hip.so on LD_LIBRARY_PATH
local M = {C = C}

local ctx = nil
local function check(rc)                -- error convention: status code -> Lua error() (SURVEY 8(b)); 1 = FG_PAUSED_SYNC
    if rc ~= 0 and rc ~= C.FG_PAUSED_SYNC then
        error(string.format('libfacegen_hip error %d: %s', rc, ffi.string(C.fg_last_error(ctx))), 2)
    end
    return rc
end
M.check = check

-- cutorch.setDevice(OPT.gpu + 1) replacement (train.lua:79): 1-based like cutorch
function M.setDevice(dev1)
    local out = ffi.new('fg_ctx*[1]')
    check(C.fg_ctx_create(dev1 - 1, out))
    ctx = out[0]
    M.ctx = ctx
end
-- cutorch.manualSeed(OPT.seed) replacement (train.lua:80): seeds of the device noise / dropout streams
M.seed = 1
function M.manualSeed(seed) M.seed = seed end

-- opaque device buffer of n floats -------------------------------------------------------------------------------------
local DeviceTensor = {}
DeviceTensor.__index = DeviceTensor
function M.DeviceTensor(n)
    local p = ffi.new('void*[1]')
    check(C.fg_malloc(ctx, n * 4, p))
    local t = setmetatable({ptr = ffi.cast('float*', p[0]), n = n}, DeviceTensor)
    ffi.gc(t.ptr, function(q) C.fg_free(ctx, q) end)
    return t
end
function DeviceTensor:copy(src)         -- FloatTensor or DeviceTensor -> self
    if torch.isTensor(src) then check(C.fg_h2d(ctx, self.ptr, src:contiguous():data(), self.n * 4))
    else check(C.fg_d2d(ctx, self.ptr, src.ptr, self.n * 4)) end
    return self
end
function DeviceTensor:float()           -- -> torch.FloatTensor (synchronises)
    local t = torch.FloatTensor(self.n)
    check(C.fg_d2h(ctx, t:data(), self.ptr, self.n * 4))
    return t
end
function DeviceTensor:clone() return M.DeviceTensor(self.n):copy(self) end
function DeviceTensor:zero() check(C.fg_fill(ctx, self.ptr, 0, self.n)); return self end
function DeviceTensor:size() return self.n end

-- host NCHW FloatTensor -> device NHWC (the nn.Copy Float -> device of nn_utils.lua:355-362) and back
local function to_device_nhwc(x)
    local n = x:nElement()
    local raw, out = M.DeviceTensor(n):copy(x), nil
    if x:dim() == 4 then
        out = M.DeviceTensor(n)
        check(C.fg_nchw_to_nhwc(ctx, raw.ptr, out.ptr, x:size(1), x:size(2), x:size(3), x:size(4)))
    else
        out = raw
    end
    return out
end
local function to_host_nchw(ptr, b, c, h, w)
    local n = b * c * h * w
    local t = torch.FloatTensor(n)
    if h * w > 1 then
        local tmp = M.DeviceTensor(n)
        check(C.fg_nhwc_to_nchw(ctx, ptr, tmp.ptr, b, c, h, w))
        check(C.fg_d2h(ctx, t:data(), tmp.ptr, n * 4))
        return t:view(b, c, h, w)
    end
    check(C.fg_d2h(ctx, t:data(), ptr, n * 4))
    return t:view(b, c)
end
M.to_device_nhwc, M.to_host_nchw = to_device_nhwc, to_host_nchw

-- image.scale(src, width, height) (dataset_c2f.lua:54-55) on the device: src a host FloatTensor [C][H][W] or [N][C][H][W] in the
-- reference's layout (layout 1 of fg_scale_bilinear); the `image` package's bilinear algorithm bit for bit.  -> host FloatTensor
function M.scale(src, width, height)
    local four = src:dim() == 4
    local n = four and src:size(1) or 1
    local c, hs, ws = src:size(four and 2 or 1), src:size(four and 3 or 2), src:size(four and 4 or 3)
    local inp = M.DeviceTensor(src:nElement()):copy(src)
    local out = M.DeviceTensor(n * c * height * width)
    check(C.fg_scale_bilinear(ctx, inp.ptr, out.ptr, n, c, hs, ws, height, width, 1))
    local t = out:float()
    if four then return t:view(n, c, height, width) end
    return t:view(c, height, width)
end
-- the arithmetic of dataset._toResult (dataset_c2f.lua:49-61) for a batch [N][C][S][S]: -> coarse, diff (host FloatTensors)
function M.coarseDiff(fineImages, coarseScale)
    local n, c, s = fineImages:size(1), fineImages:size(2), fineImages:size(3)
    local cnt = fineImages:nElement()
    local fine = M.DeviceTensor(cnt):copy(fineImages)
    local coarse, diff, tmp = M.DeviceTensor(cnt), M.DeviceTensor(cnt), M.DeviceTensor(n * c * coarseScale * coarseScale)
    check(C.fg_c2f_coarse_diff(ctx, fine.ptr, coarse.ptr, diff.ptr, tmp.ptr, n, c, s, coarseScale, 1))
    return coarse:float():view(n, c, s, s), diff:float():view(n, c, s, s)
end

-- nn module -> fg_layer_spec (typename dispatch like weight-init.lua:52-73) ---------------------------------------------
local function spec_of(m)
    local tn = torch.typename(m)
    if tn == 'nn.Linear' then return {C.FG_LINEAR, m.weight:size(2), m.weight:size(1)}
    elseif tn == 'nn.View' then
        local s = m.size
        if s:size() == 3 then return {C.FG_VIEW, s[1], s[2], s[3]} else return {C.FG_VIEW, s[1], 0, 0} end
    elseif tn == 'nn.PReLU' then return {C.FG_PRELU}
    elseif tn == 'nn.SpatialUpSamplingNearest' then assert(m.scale_factor == 2); return {C.FG_UPSAMPLE2X}
    elseif tn == 'nn.SpatialConvolution' or tn == 'cudnn.SpatialConvolution' or tn == 'cudnn.SpatialConvolutionUpsample' then
        assert(m.kW == m.kH and m.dW == m.dH and (m.dW == 1 or m.dW == 2) and m.padW == m.padH and m.padW == (m.kW - 1) / 2,
               'only odd-k same-pad convolutions with stride 1 or 2 are built')
        -- fg_layer_spec: a = nIn, b = nOut (the parent convolution's: nOutputPlaneU * factor^2), c = k, d = pad,
        -- p = stride (0 / 1: 1, 2: 2), q = factor of cudnn.SpatialConvolutionUpsample (the flat NCHW re-view runs as its own stage)
        return {C.FG_CONV, m.nInputPlane, m.nOutputPlane, m.kW, m.padW, m.dW, m.factor or 1}
    elseif tn == 'nn.SpatialBatchNormalization' then return {C.FG_BATCHNORM, m.running_mean:size(1), 0, 0, 0, m.eps, m.momentum}
    elseif tn == 'nn.SpatialDropout' then return {C.FG_SPATIAL_DROPOUT, 0, 0, 0, 0, m.p}
    elseif tn == 'nn.SpatialAveragePooling' then return {C.FG_AVGPOOL2}
    elseif tn == 'nn.SpatialMaxPooling' then assert(m.kW == 2 and m.kH == 2); return {C.FG_MAXPOOL2}
    elseif tn == 'nn.Dropout' then return {C.FG_DROPOUT, 0, 0, 0, 0, m.p}
    elseif tn == 'nn.Sigmoid' then return {C.FG_SIGMOID}
    elseif tn == 'nn.LeakyReLU' then return {C.FG_LEAKYRELU, 0, 0, 0, 0, m.negval or 0.333}
    end
    error('facegen_hip: module ' .. tostring(tn) .. ' is not on the hot path')
end

-- DeviceNet: what `net:cuda()` becomes inside NN_UTILS.activateCuda (nn_utils.lua:328-363) --------------------------------
local DeviceNet = {}
DeviceNet.__index = DeviceNet
function M.compile(seq, in_c, in_h, in_w, max_batch)
    local n = #seq.modules
    local specs = ffi.new('fg_layer_spec[?]', n)
    for i, m in ipairs(seq.modules) do
        local s = spec_of(m)
        specs[i - 1].type = s[1]; specs[i - 1].a = s[2] or 0; specs[i - 1].b = s[3] or 0
        specs[i - 1].c = s[4] or 0; specs[i - 1].d = s[5] or 0; specs[i - 1].p = s[6] or 0; specs[i - 1].q = s[7] or 0
    end
    local out = ffi.new('fg_net*[1]')
    check(C.fg_net_create(ctx, specs, n, in_c, in_h, in_w, out))
    local net = setmetatable({h = out[0], seq = seq, in_dims = {in_c, in_h, in_w}, max_batch = max_batch, train = true}, DeviceNet)
    net.nparams = tonumber(C.fg_net_num_params(net.h))
    net.nbuffers = tonumber(C.fg_net_num_buffers(net.h))
    net.nmasks = C.fg_net_num_masks(net.h)
    net.params, net.grads = M.DeviceTensor(net.nparams), M.DeviceTensor(net.nparams):zero()
    net.buffers = M.DeviceTensor(math.max(1, net.nbuffers))
    local oc, oh, ow = ffi.new('int[1]'), ffi.new('int[1]'), ffi.new('int[1]')
    check(C.fg_net_out_dims(net.h, oc, oh, ow))
    net.out_dims = {oc[0], oh[0], ow[0]}
    local wsb = tonumber(C.fg_net_workspace_bytes(net.h, max_batch))
    net.ws, net.ws_bytes = M.DeviceTensor(math.ceil(wsb / 4)), wsb
    check(C.fg_net_bind(net.h, net.params.ptr, net.grads.ptr, net.buffers.ptr))
    net:upload()
    return net
end
-- host modules -> flat device vectors, Module:parameters() order (weight then bias) == getParameters() (train.lua:151);
-- BN running statistics are buffers outside the flat vector: [running_mean | running_var] per BN layer, module order
function DeviceNet:upload()
    local flat, off = torch.FloatTensor(self.nparams), 1
    local buf, boff = torch.FloatTensor(math.max(1, self.nbuffers)):zero(), 1
    for _, m in ipairs(self.seq.modules) do
        for _, name in ipairs({'weight', 'bias'}) do
            if m[name] then
                local k = m[name]:nElement()
                flat:narrow(1, off, k):copy(m[name]:float():view(-1)); off = off + k
            end
        end
        if torch.typename(m) == 'nn.SpatialBatchNormalization' then
            local k = m.running_mean:nElement()
            buf:narrow(1, boff, k):copy(m.running_mean:float()); buf:narrow(1, boff + k, k):copy(m.running_var:float())
            boff = boff + 2 * k
        end
    end
    self.params:copy(flat)
    if self.nbuffers > 0 then self.buffers:copy(buf) end
    check(C.fg_net_params_changed(self.h))
    self.device_newer = false                                -- host modules == device vectors
end
-- flat device vectors -> host modules (gradWeight / gradBias too when want_grads), e.g. before torch.save
function DeviceNet:download(want_grads)
    local flat, gflat, off = self.params:float(), want_grads and self.grads:float() or nil, 1
    local buf, boff = self.nbuffers > 0 and self.buffers:float() or nil, 1
    for _, m in ipairs(self.seq.modules) do
        for _, name in ipairs({'weight', 'bias'}) do
            if m[name] then
                local k = m[name]:nElement()
                m[name]:copy(flat:narrow(1, off, k):viewAs(m[name]))
                local gname = name == 'weight' and 'gradWeight' or 'gradBias'
                if gflat and m[gname] then m[gname]:copy(gflat:narrow(1, off, k):viewAs(m[gname])) end
                off = off + k
            end
        end
        if buf and torch.typename(m) == 'nn.SpatialBatchNormalization' then
            local k = m.running_mean:nElement()
            m.running_mean:copy(buf:narrow(1, boff, k)); m.running_var:copy(buf:narrow(1, boff + k, k))
            boff = boff + 2 * k
        end
    end
    self.device_newer = false                                -- host modules == device vectors
end
-- Which side holds the current parameters?  The module-level protocol (seq:forward / :backward below) treats the HOST modules as
-- authoritative -- the reference's optimizers write the host flat vector of getParameters() -- and uploads them before a forward.
-- The step-level entries (Gan:stepD / stepG / update) and the device optimizers (M.interruptable*) move only the DEVICE vectors
-- (parameters, BatchNorm running statistics): they mark the net `device_newer`, and the next module-level forward then DOWNLOADS
-- instead of uploading -- NN_UTILS.visualizeProgress runs MODEL_G:forward / MODEL_D:forward at the start of every epoch
-- (train.lua:204, nn_utils.lua:52, 96) and must see, not overwrite, what adversarial.train learned.
function DeviceNet:markDeviceNewer() self.device_newer = true end
function DeviceNet:syncForModuleCall()
    if self.device_newer then self:download(false) else self:upload() end
end
function DeviceNet:getParameters() return self.params, self.grads end
function DeviceNet:paramsChanged() check(C.fg_net_params_changed(self.h)) end
-- Bernoulli keep masks for every dropout layer (nn.Dropout / nn.SpatialDropout draw theirs from cutorch's generator)
function DeviceNet:drawMasks(batch)
    self.mask_offset = self.mask_offset or 0
    local masks = {}
    for i = 0, self.nmasks - 1 do
        local n = tonumber(C.fg_net_mask_elems(self.h, i, batch))
        local t = M.DeviceTensor(n)
        check(C.fg_rng_bernoulli(ctx, 1000 + M.seed, self.mask_offset, t.ptr, n, C.fg_net_mask_keep(self.h, i)))
        self.mask_offset = self.mask_offset + math.ceil(n / 4)
        masks[i + 1] = t
    end
    return masks
end
function DeviceNet:forward(x_dev, batch, masks)             -- x_dev: DeviceTensor, NHWC; mode = self.train
    local off = ffi.new('long long[1]')
    local mp, nm = nil, 0
    if self.train and self.nmasks > 0 then
        masks = masks or self:drawMasks(batch)
        nm = #masks; mp = ffi.new('const float*[?]', nm)
        for i = 1, nm do mp[i - 1] = masks[i].ptr end
        self.masks = masks                                   -- keep alive until backward
    end
    check(C.fg_net_forward(self.h, batch, x_dev.ptr, self.ws.ptr, self.ws_bytes, self.train and 1 or 0, mp, nm, off))
    self.last = {x = x_dev, batch = batch}
    return self.ws.ptr + off[0]                              -- NHWC output inside the workspace
end
function DeviceNet:backward(gy_ptr, want_params, gx_dev)
    local flags = (want_params and C.FG_BWD_PARAM_GRADS or 0) + (gx_dev and C.FG_BWD_INPUT_GRAD or 0)
    check(C.fg_net_backward(self.h, self.last.batch, self.last.x.ptr, gy_ptr, self.ws.ptr, self.ws_bytes, flags,
                            gx_dev and gx_dev.ptr or nil))
end

-- (i) module level: per-instance overrides on the nn.Sequential that sits between the two nn.Copy modules ------------------
-- `first` (optional): the nn.JoinTable / nn.CAddTable in front of the c2f nets (models_c2f.lua:116, 240) -- the table input
-- {a, b} is combined on the device before the plan runs.
local function infer_input_dims(seq)      -- G starts with nn.Linear(noiseDim, ...), D with a convolution over IMG_DIMENSIONS
    local m = seq.modules[1]
    if torch.typename(m) == 'nn.Linear' then return {m.weight:size(2), 1, 1} end
    return {m.nInputPlane or IMG_DIMENSIONS[1], IMG_DIMENSIONS[2], IMG_DIMENSIONS[3]}
end
function M.attach(seq, dims, max_batch, first)
    dims = dims or infer_input_dims(seq)
    local dn = M.compile(seq, dims[1], dims[2], dims[3], max_batch)
    seq.fg = dn
    local function combine(input)
        if not first then return to_device_nhwc(input), input:size(1) end
        local a, b = to_device_nhwc(input[1]), to_device_nhwc(input[2])
        local B, out = input[1]:size(1), nil
        if torch.typename(first) == 'nn.JoinTable' then
            out = M.DeviceTensor(a.n + b.n)
            check(C.fg_concat_channels(ctx, a.ptr, b.ptr, out.ptr, B * input[1]:size(3) * input[1]:size(4), input[1]:size(2), input[2]:size(2)))
        else
            out = M.DeviceTensor(a.n)
            check(C.fg_add(ctx, a.ptr, b.ptr, out.ptr, a.n))
        end
        return out, B
    end
    function seq:updateOutput(input)
        dn:syncForModuleCall()                               -- host copy authoritative unless a device-side update is newer
        local x, B = combine(input)
        local y = dn:forward(x, B)
        self.output = to_host_nchw(y, B, dn.out_dims[1], dn.out_dims[2], dn.out_dims[3])
        if dn.train and dn.nbuffers > 0 then dn:download(false) end     -- BN running statistics moved
        return self.output
    end
    function seq:backward(input, gradOutput, scale)
        assert(scale == nil or scale == 1, 'facegen_hip: gradient scale ~= 1 is not built')
        local gy = to_device_nhwc(gradOutput)
        local gx = M.DeviceTensor(dn.last.x.n)
        dn:backward(gy.ptr, true, gx)
        dn:download(true)                                    -- accGradParameters: gradWeight / gradBias of every module
        local c, h, w = dims[1], dims[2], dims[3]
        local g = to_host_nchw(gx.ptr, dn.last.batch, c, h, w)
        if first and torch.typename(first) == 'nn.JoinTable' then
            local ca = input[1]:size(2)
            self.gradInput = {g:narrow(2, 1, ca), g:narrow(2, ca + 1, c - ca)}
        elseif first then
            self.gradInput = {g, g}                          -- nn.CAddTable: the same gradient for both inputs
        else
            self.gradInput = g
        end
        return self.gradInput
    end
    seq.updateGradInput = function(self, input, gradOutput) return self:backward(input, gradOutput) end
    seq.accGradParameters = function() end                   -- done inside backward
    local training, evaluate = seq.training, seq.evaluate
    function seq:training() dn.train = true; return training(self) end
    function seq:evaluate() dn.train = false; return evaluate(self) end
    return seq
end

-- torch.save / Module:clone serialise every field of a module, closures with their upvalues included (File:writeObject), and cannot
-- write FFI cdata: the device plan (seq.fg) and the per-instance overrides installed by M.attach (their upvalues are `dn`, `C`, `ctx`)
-- must be taken off the module around a save or a clone.  detach() returns what it removed, reattach() puts it back.
local ATTACHED = {'fg', 'updateOutput', 'backward', 'updateGradInput', 'accGradParameters', 'training', 'evaluate'}
function M.detach(seq)
    local saved = {}
    for _, k in ipairs(ATTACHED) do saved[k] = rawget(seq, k); rawset(seq, k, nil) end
    return saved
end
function M.reattach(seq, saved)
    for _, k in ipairs(ATTACHED) do rawset(seq, k, saved[k]) end
    return seq
end

-- interruptable_optimizers.lua on DEVICE vectors (x, dfdx = DeviceTensor); `fused` carries penalty / clamp / 1/world.
-- `net` (a DeviceNet) is told that its parameters moved: the packed / tap-folded weights are rebuilt before its next use.
local function fused_args(fused) fused = fused or {}; return fused.gscale or 1, fused.l1_mul or 0, fused.l2 or 0, fused.clamp or 0 end
function M.interruptableAdam(opfunc, x, config, state, fused, net)
    config = config or {}; state = state or config
    local fx, dfdx = opfunc(x)
    if fx == false then return false end                     -- interruptable_optimizers.lua:60-66
    state.t = (state.t or 0) + 1
    state.m = state.m or M.DeviceTensor(x.n):zero()
    state.v = state.v or M.DeviceTensor(x.n):zero()
    local gs, l1, l2, cl = fused_args(fused)
    check(C.fg_adam_fused(ctx, x.ptr, dfdx.ptr, state.m.ptr, state.v.ptr, x.n, gs, l1, l2, cl, config.learningRate or 0.001,
                          config.beta1 or 0.9, config.beta2 or 0.999, config.epsilon or 1e-8, state.t, nil))
    if net then net:paramsChanged(); net:markDeviceNewer() end
    return x, {fx}
end
function M.interruptableSgd(opfunc, x, config, state, fused, net)
    config = config or {}; state = state or config
    local lr, lrd, wd = config.learningRate or 1e-3, config.learningRateDecay or 0, config.weightDecay or 0
    local mom = config.momentum or 0
    local damp, nesterov = config.dampening or mom, config.nesterov or false
    state.evalCounter = state.evalCounter or 0
    local fx, dfdx = opfunc(x)
    if fx == false then return false end
    local first = 0
    if mom ~= 0 and not state.dfdx then state.dfdx = M.DeviceTensor(x.n):zero(); first = 1 end
    local gs, l1, l2, cl = fused_args(fused)
    check(C.fg_sgd_fused(ctx, x.ptr, dfdx.ptr, mom ~= 0 and state.dfdx.ptr or nil, x.n, gs, l1, l2, cl,
                         lr / (1 + state.evalCounter * lrd), mom, damp, wd, nesterov and 1 or 0, first))
    state.evalCounter = state.evalCounter + 1
    if net then net:paramsChanged(); net:markDeviceNewer() end
    return x, {fx}
end
function M.interruptableAdagrad(opfunc, x, config, state, fused, net)
    config = config or {}; state = state or config
    local lr, lrd = config.learningRate or 1e-3, config.learningRateDecay or 0
    state.evalCounter = state.evalCounter or 0
    local fx, dfdx = opfunc(x)
    if fx == false then return false end
    state.paramVariance = state.paramVariance or M.DeviceTensor(x.n):zero()
    local gs, l1, l2, cl = fused_args(fused)
    check(C.fg_adagrad_fused(ctx, x.ptr, dfdx.ptr, state.paramVariance.ptr, x.n, gs, l1, l2, cl, lr / (1 + state.evalCounter * lrd)))
    state.evalCounter = state.evalCounter + 1
    if net then net:paramsChanged(); net:markDeviceNewer() end
    return x, {fx}
end

-- data parallelism: fg_comm_* (RCCL bound by the library).  The 128-byte id travels through a file all ranks can read.
function M.Comm(rank, world, id_path)
    local id = ffi.new('char[?]', C.FG_COMM_ID_BYTES)
    if rank == 0 then
        check(C.fg_comm_unique_id(ctx, id, C.FG_COMM_ID_BYTES))
        local f = assert(io.open(id_path .. '.tmp', 'wb')); f:write(ffi.string(id, C.FG_COMM_ID_BYTES)); f:close()
        os.rename(id_path .. '.tmp', id_path)
    else
        local f = io.open(id_path, 'rb')
        while not f do os.execute('sleep 0.1'); f = io.open(id_path, 'rb') end
        local s = f:read(C.FG_COMM_ID_BYTES); f:close()
        ffi.copy(id, s, C.FG_COMM_ID_BYTES)
    end
    local out = ffi.new('fg_comm*[1]')
    check(C.fg_comm_create(ctx, id, C.FG_COMM_ID_BYTES, rank, world, out))
    local comm = {h = out[0], rank = rank, world = world}
    function comm:allreduce(t) check(C.fg_allreduce_sum(self.h, t.ptr, t.n)) end
    function comm:broadcast(t, root) check(C.fg_broadcast(self.h, t.ptr, t.n, root or 0)) end
    function comm:destroy() check(C.fg_comm_destroy(self.h)) end
    return comm
end

-- (ii) step level: fg_gan_* -- one C call per closure of adversarial.lua / adversarial_c2f.lua ------------------------------
local Gan = {}
Gan.__index = Gan
local METHOD = {adam = 0, sgd = 1, adagrad = 2}
function M.Gan(dnG, dnD, table_inputs, max_batch)
    local t = table_inputs and 1 or 0
    local bytes = tonumber(C.fg_gan_workspace_bytes(dnG.h, dnD.h, t, max_batch))
    local ws = M.DeviceTensor(math.ceil(bytes / 4) + 64)
    local base = ffi.cast('char*', ws.ptr)
    local skip = (256 - tonumber(ffi.cast('uintptr_t', base) % 256)) % 256
    local out = ffi.new('fg_gan*[1]')
    check(C.fg_gan_create(ctx, dnG.h, dnD.h, t, max_batch, base + skip, bytes, out))
    local g = setmetatable({h = out[0], ws = ws, base = ffi.cast('float*', base + skip), G = dnG, D = dnD, max_batch = max_batch}, Gan)
    check(C.fg_gan_bind_workspaces(g.h, dnG.ws.ptr, dnG.ws_bytes, dnD.ws.ptr, dnD.ws_bytes))
    check(C.fg_gan_set_seeds(g.h, M.seed, 0, 1000 + M.seed, 0))
    ffi.gc(g.h, C.fg_gan_destroy)
    return g
end
function Gan:setComm(comm, sync_bn) check(C.fg_gan_set_comm(self.h, comm and comm.h or nil, sync_bn and 1 or 0, 1)) end
-- OPT.{D,G}_L1 / _L2 / _clamp (train.lua:29-37) and OPTSTATE.<method>.<net> (train.lua:180-191): which = 'D' | 'G'
function Gan:configure(which, OPT, OPTSTATE)
    local w = which == 'D' and 0 or 1
    check(C.fg_gan_set_penalty(self.h, w, OPT[which .. '_L1'], OPT[which .. '_L2'], OPT[which .. '_clamp']))
    local method = OPT[which .. '_optmethod']
    local cfg = OPTSTATE[method][which]
    check(C.fg_gan_set_optimizer(self.h, w, METHOD[method], cfg.learningRate or -1, cfg.beta1 or 0.9, cfg.beta2 or 0.999,
                                 cfg.epsilon or 1e-8, cfg.momentum or 0, cfg.dampening or -1, cfg.weightDecay or 0,
                                 cfg.learningRateDecay or 0, cfg.nesterov and 1 or 0))
end
function Gan:buffer(what, n)            -- host copy of a result / state buffer (synchronises)
    local off, cnt = ffi.new('long long[1]'), ffi.new('long long[1]')
    check(C.fg_gan_buffer(self.h, what, off, cnt))
    n = n or tonumber(cnt[0])
    local src = (what == C.FG_GAN_D_OUTPUT) and (self.D.ws.ptr + off[0]) or (self.base + off[0])
    local t = torch.FloatTensor(n)
    check(C.fg_d2h(ctx, t:data(), src, n * 4))
    return t
end
function Gan:confusion()                -- {local = {c00, c01, c10, c11}, global = {...}}, [pred * 2 + target]
    local off, cnt = ffi.new('long long[1]'), ffi.new('long long[1]')
    check(C.fg_gan_buffer(self.h, C.FG_GAN_CONFUSION, off, cnt))
    local host = ffi.new('int[8]')
    check(C.fg_d2h(ctx, host, self.base + off[0], 32))
    return {host[0], host[1], host[2], host[3]}, {host[4], host[5], host[6], host[7]}
end
-- the same 8 counts left on the device (a stream-ordered 32-byte copy, no host sync): read later with Gan.readConfusion
function Gan:confusionDevice()
    local off, cnt = ffi.new('long long[1]'), ffi.new('long long[1]')
    check(C.fg_gan_buffer(self.h, C.FG_GAN_CONFUSION, off, cnt))
    local t = M.DeviceTensor(8)
    check(C.fg_d2d(ctx, t.ptr, self.base + off[0], 32))
    return t
end
-- ... or into slot `slot` (0-based) of ONE per-epoch DeviceTensor(8 * closures): no allocation inside the epoch loop
function Gan:confusionInto(t, slot)
    local off, cnt = ffi.new('long long[1]'), ffi.new('long long[1]')
    check(C.fg_gan_buffer(self.h, C.FG_GAN_CONFUSION, off, cnt))
    assert(8 * (slot + 1) <= t.n, 'confusionInto: slot out of range')
    check(C.fg_d2d(ctx, t.ptr + 8 * slot, self.base + off[0], 32))
end
function M.readConfusion(t, slot)
    local host = ffi.new('int[8]')
    check(C.fg_d2h(ctx, host, t.ptr + 8 * (slot or 0), 32))
    return {host[0], host[1], host[2], host[3]}, {host[4], host[5], host[6], host[7]}
end
-- real / cond_*: DeviceTensor NHWC (cond_* only for the table nets); noise / masks nil: drawn by the library
function Gan:stepD(batch, real, cond_real, cond_fake, hold)
    check(C.fg_step_D(self.h, batch, real.ptr, cond_real and cond_real.ptr or nil, cond_fake and cond_fake.ptr or nil,
                      nil, nil, hold and C.FG_STEP_NO_UPDATE or 0))
    self.G:markDeviceNewer()                                 -- G ran in TRAIN mode: its BatchNorm running statistics moved
    if not hold then self.D:markDeviceNewer() end
end
function Gan:stepG(batch, cond, hold)
    check(C.fg_step_G(self.h, batch, cond and cond.ptr or nil, nil, nil, hold and C.FG_STEP_NO_UPDATE or 0))
    self.G:markDeviceNewer(); self.D:markDeviceNewer()       -- a deferred D update lands inside the G-step (N > 1)
end
function Gan:update(which)
    check(C.fg_gan_update(self.h, which == 'D' and 0 or 1))
    if which == 'D' then self.D:markDeviceNewer() else self.G:markDeviceNewer() end
end
function Gan:finishPending() check(C.fg_gan_finish_pending(self.h)) end

return M
