--[[ facegen_hip.lua -- LuaJIT-FFI binding of libfacegen_hip.so for the reference's Lua/Torch7 host.

NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (the build image has no Lua/LuaJIT/Torch7; SURVEY.md F6).  It is the
binding a maintainer adds next to models.lua / adversarial.lua; the Python package face_generator_amd/ drives the
SAME entry points through ctypes and is what the tests execute.  Host tensors cross the ABI as raw float* from
tensor:data(); device memory is an opaque handle owned by this shim.

Replaces, on the hot path only: cutorch.setDevice/manualSeed (train.lua:79-80), :cuda() / nn.Copy
(nn_utils.lua:355-362), MODEL:forward/backward (adversarial.lua:95-100, 202-214), nn.BCECriterion (train.lua:148),
the optimizer tensor math (interruptable_optimizers.lua:78-90) and the penalty/clamp (adversarial.lua:103-123).
]]
local ffi = require 'ffi'

ffi.cdef[[
typedef struct fg_ctx fg_ctx;
typedef struct fg_net fg_net;
typedef struct fg_layer_spec { int type; int a, b, c, d; float p, q; } fg_layer_spec;
int fg_ctx_create(int device, fg_ctx** out);
int fg_set_math(fg_ctx* ctx, int mode);   /* 0 = fp32 MFMA, 6 = fp32 emulated with six split-bf16 plane products */
int fg_get_math(fg_ctx* ctx);
int fg_ctx_destroy(fg_ctx* ctx);
const char* fg_last_error(const fg_ctx* ctx);
int fg_stream_sync(fg_ctx* ctx);
int fg_malloc(fg_ctx* ctx, size_t bytes, void** out);
int fg_free(fg_ctx* ctx, void* p);
int fg_h2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_d2h(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_d2d(fg_ctx* ctx, void* dst, const void* src, size_t bytes);
int fg_fill(fg_ctx* ctx, float* p, float value, long long n);
int fg_nchw_to_nhwc(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);
int fg_nhwc_to_nchw(fg_ctx* ctx, const float* src, float* dst, int n, int c, int h, int w);
int fg_rng_uniform(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float lo, float hi);
int fg_rng_bernoulli(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float keep_prob);
int fg_rng_normal(fg_ctx* ctx, uint64_t seed, uint64_t offset, float* out, long long n, float mean, float std);
int fg_net_create(fg_ctx* ctx, const fg_layer_spec* layers, int n_layers, int in_c, int in_h, int in_w, fg_net** out);
int fg_net_destroy(fg_net* net);
long long fg_net_num_params(const fg_net* net);
long long fg_net_num_buffers(const fg_net* net);
int fg_net_num_masks(const fg_net* net);
long long fg_net_mask_elems(const fg_net* net, int mask_index, int batch);
int fg_net_out_dims(const fg_net* net, int* c, int* h, int* w);
size_t fg_net_workspace_bytes(const fg_net* net, int max_batch);
int fg_net_param_offset(const fg_net* net, int layer_index, long long* wo, long long* wn, long long* bo, long long* bn);
int fg_net_bind(fg_net* net, float* params, float* grads, float* buffers);
int fg_net_params_changed(fg_net* net);
int fg_net_forward(fg_net* net, int batch, const float* x, void* ws, size_t ws_bytes, int train,
                   const float* const* masks, int n_masks, long long* out_offset);
int fg_net_backward(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags, float* gx);
int fg_net_num_stages(const fg_net* net);
int fg_net_stage_params(const fg_net* net, int stage, long long* param_lo, long long* param_hi);
int fg_net_backward_range(fg_net* net, int batch, const float* x, const float* gy, void* ws, size_t ws_bytes, int flags,
                          float* gx, int stage_from, int stage_to);
int fg_net_set_sync_bn(fg_net* net, int on, double* sync_buf_dev, long long capacity_doubles);
long long fg_net_sync_count(const fg_net* net);
int fg_net_forward_resume(fg_net* net, long long* out_offset);
int fg_net_backward_resume(fg_net* net);
int fg_bce_forward_backward(fg_ctx* ctx, const float* prob, const float* target, int n, float* loss_dev,
                            float* grad_dev, int* confusion_dev);
int fg_adam_fused(fg_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float gscale, float l1_mul,
                  float l2, float clamp, double lr, double beta1, double beta2, double eps, int t, float* g_out);
int fg_sgd_fused(fg_ctx* ctx, float* p, const float* g, float* mom_buf, long long n, float gscale, float l1_mul,
                 float l2, float clamp, double lr, double momentum, double dampening, double weight_decay, int nesterov,
                 int first_step);
int fg_adagrad_fused(fg_ctx* ctx, float* p, const float* g, float* variance, long long n, float gscale, float l1_mul,
                     float l2, float clamp, double clr);
int fg_norms(fg_ctx* ctx, const float* p, long long n, float* out2_dev, float* scratch);
]]

local C = ffi.load('facegen_hip')      -- libfacegen_hip.so on LD_LIBRARY_PATH
local M = {C = C}
local FG = {LINEAR = 1, VIEW = 2, PRELU = 3, UPSAMPLE2X = 4, CONV = 5, BATCHNORM = 6, SPATIAL_DROPOUT = 7,
            AVGPOOL2 = 8, DROPOUT = 9, SIGMOID = 10, LEAKYRELU = 11}
M.FG = FG

local ctx = nil
local function check(rc)                -- error convention: status code -> Lua error() (SURVEY 8(b))
    if rc ~= 0 then error(string.format('libfacegen_hip error %d: %s', rc, ffi.string(C.fg_last_error(ctx))), 2) end
end
M.check = check

-- cutorch.setDevice(OPT.gpu + 1) replacement (train.lua:79): 1-based like cutorch
function M.setDevice(dev1)
    local out = ffi.new('fg_ctx*[1]')
    check(C.fg_ctx_create(dev1 - 1, out))
    ctx = out[0]
    M.ctx = ctx
end

-- opaque device buffer of n floats
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

-- nn module -> fg_layer_spec (typename dispatch like weight-init.lua:52-73)
local function spec_of(m)
    local tn = torch.typename(m)
    if tn == 'nn.Linear' then return {FG.LINEAR, m.weight:size(2), m.weight:size(1)}
    elseif tn == 'nn.View' then
        local s = m.size
        if s:size() == 3 then return {FG.VIEW, s[1], s[2], s[3]} else return {FG.VIEW, s[1], 0, 0} end
    elseif tn == 'nn.PReLU' then return {FG.PRELU}
    elseif tn == 'nn.SpatialUpSamplingNearest' then assert(m.scale_factor == 2); return {FG.UPSAMPLE2X}
    elseif tn == 'nn.SpatialConvolution' or tn == 'cudnn.SpatialConvolution' then
        assert(m.kW == m.kH and m.dW == 1 and m.dH == 1 and m.padW == m.padH and m.padW == (m.kW - 1) / 2,
               'only odd-k stride-1 same-pad convolutions are built')
        return {FG.CONV, m.nInputPlane, m.nOutputPlane, m.kW, m.padW}
    elseif tn == 'nn.SpatialBatchNormalization' then return {FG.BATCHNORM, m.running_mean:size(1), 0, 0, 0, m.eps, m.momentum}
    elseif tn == 'nn.SpatialDropout' then return {FG.SPATIAL_DROPOUT, 0, 0, 0, 0, m.p}
    elseif tn == 'nn.SpatialAveragePooling' then return {FG.AVGPOOL2}
    elseif tn == 'nn.Dropout' then return {FG.DROPOUT, 0, 0, 0, 0, m.p}
    elseif tn == 'nn.Sigmoid' then return {FG.SIGMOID}
    elseif tn == 'nn.LeakyReLU' then return {FG.LEAKYRELU, 0, 0, 0, 0, m.negval or 0.333}
    end
    error('facegen_hip: module ' .. tostring(tn) .. ' is not on the hot path')
end

-- DeviceNet: what `net:cuda()` becomes inside NN_UTILS.activateCuda (nn_utils.lua:328-363)
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
    local net = setmetatable({h = out[0], seq = seq, in_dims = {in_c, in_h, in_w}}, DeviceNet)
    net.nparams = tonumber(C.fg_net_num_params(net.h))
    net.params, net.grads = M.DeviceTensor(net.nparams), M.DeviceTensor(net.nparams)
    net.buffers = M.DeviceTensor(math.max(1, tonumber(C.fg_net_num_buffers(net.h))))
    -- flatten host parameters in Module:parameters() order (weight then bias) == getParameters() (train.lua:151)
    local flat, off = torch.FloatTensor(net.nparams), 1
    for _, m in ipairs(seq.modules) do
        for _, name in ipairs({'weight', 'bias'}) do
            if m[name] then
                local k = m[name]:nElement()
                flat:narrow(1, off, k):copy(m[name]:float():view(-1)); off = off + k
            end
        end
    end
    net.params:copy(flat)
    local wsb = tonumber(C.fg_net_workspace_bytes(net.h, max_batch))
    net.ws, net.ws_bytes = M.DeviceTensor(math.ceil(wsb / 4)), wsb
    check(C.fg_net_bind(net.h, net.params.ptr, net.grads.ptr, net.buffers.ptr))
    return net
end
function DeviceNet:getParameters() return self.params, self.grads end
function DeviceNet:forward(x_dev, batch, train, masks)      -- x_dev: DeviceTensor, NHWC
    local off = ffi.new('long long[1]')
    local mp, nm = nil, 0
    if masks then
        nm = #masks; mp = ffi.new('const float*[?]', nm)
        for i = 1, nm do mp[i - 1] = masks[i].ptr end
    end
    check(C.fg_net_forward(self.h, batch, x_dev.ptr, self.ws.ptr, self.ws_bytes, train and 1 or 0, mp, nm, off))
    self.last = {x = x_dev, batch = batch}
    return self.ws.ptr + off[0]                              -- NHWC output inside the workspace
end
function DeviceNet:backward(gy_ptr, want_params, gx_dev)
    local flags = (want_params and 1 or 0) + (gx_dev and 2 or 0)
    check(C.fg_net_backward(self.h, self.last.batch, self.last.x.ptr, gy_ptr, self.ws.ptr, self.ws_bytes, flags,
                            gx_dev and gx_dev.ptr or nil))
end

-- interruptableAdam on device vectors (interruptable_optimizers.lua:49-94); `fused` carries penalty/clamp/1/world
function M.interruptableAdam(opfunc, x, config, state, fused)
    config = config or {}; state = state or config; fused = fused or {}
    local fx, dfdx = opfunc(x)
    if fx == false then return false end
    state.t = (state.t or 0) + 1
    state.m = state.m or M.DeviceTensor(x.n):zero()
    state.v = state.v or M.DeviceTensor(x.n):zero()
    check(C.fg_adam_fused(ctx, x.ptr, dfdx.ptr, state.m.ptr, state.v.ptr, x.n, fused.gscale or 1, fused.l1_mul or 0,
                          fused.l2 or 0, fused.clamp or 0, config.learningRate or 0.001, config.beta1 or 0.9,
                          config.beta2 or 0.999, config.epsilon or 1e-8, state.t, nil))
    return x, {fx}
end

return M
