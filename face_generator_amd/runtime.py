"""Device runtime glue: one fg_ctx per process/GPU, torch tensors as device memory, DeviceNet = fg_net + buffers.

Replaces the reference's `cutorch.setDevice` / `:cuda()` / `nn.Copy` plumbing (train.lua:79-80,
nn_utils.lua:328-363).  Everything numerical happens inside libfacegen_hip.so.
"""
import ctypes

import torch

from . import _lib
from ._lib import FgError, LayerSpec, LAYER_TYPES

_CTX = {}


class Context:
    """fg_ctx bound to one HIP device; launches go to torch's current stream of that device."""

    def __init__(self, device=0):
        self.dry = device == -1
        if self.dry:
            # FG_DEVICE_NONE (include/facegen_hip.h): a PLANNING-ONLY context.  No kernel runs and nothing is computed -- this is
            # not a CPU path -- but every host-side decision of the library (plans, stage walks, sync-BN pauses, gradient buckets,
            # the collective schedule) is taken as on the GPU.  "Device" tensors are host tensors whose contents nobody reads.
            self.lib = _lib.load_library()
            self.device = torch.device("cpu")
            h = ctypes.c_void_p()
            rc = self.lib.fg_ctx_create(-1, ctypes.byref(h))
            if rc != 0:
                raise FgError("fg_ctx_create: %s" % self.lib.fg_last_error(None).decode())
            self.h = h
            return
        if not torch.cuda.is_available():
            raise FgError("no HIP device visible: the face_generator_amd compute path needs an MI355X "
                          "(there is no CPU fallback)")
        self.lib = _lib.load_library()
        self.device = torch.device("cuda", device)
        h = ctypes.c_void_p()
        rc = self.lib.fg_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise FgError("fg_ctx_create: %s" % self.lib.fg_last_error(None).decode())
        self.h = h
        torch.cuda.set_device(self.device)
        self.bind_stream()

    def bind_stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.check(self.lib.fg_ctx_set_stream(self.h, ctypes.c_void_p(s.cuda_stream)))

    def check(self, rc):
        if rc < 0 or rc > 1:      # 1 == FG_PAUSED_SYNC (handled by the caller)
            raise FgError("libfacegen_hip error %d: %s" % (rc, self.lib.fg_last_error(self.h).decode()))

    def set_math(self, mode):
        """0: native fp32 MFMA (default); 6: fp32 emulated with six split-bf16 plane products (include/facegen_hip.h)."""
        self.check(self.lib.fg_set_math(self.h, int(mode)))

    def get_math(self):
        return self.lib.fg_get_math(self.h)

    def set_fusion(self, flags):
        """Optional kernel fusions (include/facegen_hip.h FG_FUSE_*): 1 = PReLU in the neighbouring contraction's epilogue,
        2 = one-pass matrix-pipe 3x3 thin-output convolution, 4 = all weight-gradient split-K sums of a backward pass in one launch,
        8 = Adam + the re-pack of every layer in one launch (measured slower: off by default), 16 = the bias gradient of a thin-input
        convolution from its weight-gradient kernel (no separate column-sum pass), 32 / 64 / 128 = 3x3 / nearest-x2-folded 5x5 / plain 5x5
        convolutions as Winograd F(2x2, 3x3) (read when a net is created), 256 = their weight gradients in the Winograd domain as
        well; default 503."""
        self.check(self.lib.fg_set_fusion(self.h, int(flags)))

    def get_fusion(self):
        return self.lib.fg_get_fusion(self.h)

    def empty(self, *shape):
        if self.dry:
            return torch.zeros(*shape, dtype=torch.float32)          # planning-only: deterministic "results" (all zero)
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def sync(self):
        self.check(self.lib.fg_stream_sync(self.h))

    # ---- boundary helpers (the nn.Copy Float<->device modules) ----
    def to_device_nhwc(self, x_nchw):
        """host/any NCHW float tensor -> device NHWC tensor."""
        x = torch.as_tensor(x_nchw, dtype=torch.float32)
        if x.dim() == 2:
            return x.to(self.device).contiguous()
        n, c, h, w = x.shape
        xd = x.to(self.device).contiguous()
        out = self.empty(n, h, w, c)
        self.check(self.lib.fg_nchw_to_nhwc(self.h, xd.data_ptr(), out.data_ptr(), n, c, h, w))
        return out

    def to_nchw(self, x_nhwc):
        if x_nhwc.dim() == 2:
            return x_nhwc.clone()
        n, h, w, c = x_nhwc.shape
        out = self.empty(n, c, h, w)
        self.check(self.lib.fg_nhwc_to_nchw(self.h, x_nhwc.contiguous().data_ptr(), out.data_ptr(), n, c, h, w))
        return out

    def uniform(self, shape, lo, hi, seed, offset=0):
        out = self.empty(*shape)
        self.check(self.lib.fg_rng_uniform(self.h, seed, offset, out.data_ptr(), out.numel(), lo, hi))
        return out

    def bernoulli(self, shape, keep, seed, offset=0):
        out = self.empty(*shape)
        self.check(self.lib.fg_rng_bernoulli(self.h, seed, offset, out.data_ptr(), out.numel(), keep))
        return out

    def normal(self, shape, mean, std, seed, offset=0):
        out = self.empty(*shape)
        self.check(self.lib.fg_rng_normal(self.h, seed, offset, out.data_ptr(), out.numel(), mean, std))
        return out


def get_context(device=None):
    """device = -1: the planning-only context (a process holds either that one or real ones)."""
    if device is None:
        device = -1 if -1 in _CTX else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if device not in _CTX:
        _CTX[device] = Context(device)
    return _CTX[device]


def make_specs(layers):
    """layers: list of tuples (TYPE, a, b, c, d, p, q) -> ctypes array of fg_layer_spec."""
    arr = (LayerSpec * len(layers))()
    for i, l in enumerate(layers):
        l = tuple(l) + (0,) * (7 - len(l))
        arr[i].type = LAYER_TYPES[l[0]] if isinstance(l[0], str) else int(l[0])
        arr[i].a, arr[i].b, arr[i].c, arr[i].d = int(l[1]), int(l[2]), int(l[3]), int(l[4])
        arr[i].p, arr[i].q = float(l[5]), float(l[6])
    return arr


class DeviceNet:
    """An nn.Sequential compiled by fg_net_create, with its flat parameter / gradient vectors
    (== Module:getParameters(), train.lua:151-152), BN running stats and workspace as torch device tensors."""
    _count = 0          # nets built by this process
    _count_by_structure = {}      # crc32(structure) -> nets of that structure built so far (salt of the default dropout-mask key)

    def __init__(self, ctx, layers, in_dims, max_batch, params=None, grads=None):
        """params / grads: optional slices of a larger flat vector shared by several nets (nn.ConcatSequential)."""
        self.ctx, self.lib = ctx, ctx.lib
        self.layers = list(layers)
        self.in_c, self.in_h, self.in_w = in_dims
        self.specs = make_specs(self.layers)
        h = ctypes.c_void_p()
        ctx.check(self.lib.fg_net_create(ctx.h, self.specs, len(self.layers), self.in_c, self.in_h, self.in_w,
                                         ctypes.byref(h)))
        self.h = h
        self.n_params = self.lib.fg_net_num_params(h)
        self.n_buffers = self.lib.fg_net_num_buffers(h)
        self.n_masks = self.lib.fg_net_num_masks(h)
        c, hh, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.lib.fg_net_out_dims(h, ctypes.byref(c), ctypes.byref(hh), ctypes.byref(w))
        self.out_c, self.out_h, self.out_w = c.value, hh.value, w.value
        if params is not None and (params.numel() != self.n_params or grads.numel() != self.n_params):
            raise FgError("DeviceNet: the shared parameter slice has %d elements, the net %d" % (params.numel(), self.n_params))
        self.params = params if params is not None else ctx.zeros(self.n_params)
        self.grads = grads if grads is not None else ctx.zeros(self.n_params)
        self.buffers = ctx.zeros(max(self.n_buffers, 1))
        self._init_bn_buffers()
        self.max_batch = 0
        self.ws = None
        self.reserve(max_batch)
        ctx.check(self.lib.fg_net_bind(h, self.params.data_ptr(), self.grads.data_ptr(), self.buffers.data_ptr()))
        self.train = True
        self.sync_buf, self._sync_reduce = None, None
        # Philox streams are keyed by (seed, counter) only: the dropout masks must not share the noise stream's key (S.noise_seed:
        # small integers, seed + rank), and two nets must not share one either -- not even two with the same structure (two
        # identical D's, the branches of a CompositeDeviceNet).  Default key = a tag in the upper 32 bits (a namespace no
        # noise seed reaches) + crc32 of the net's STRUCTURE (layer specs, input dims) + its ordinal among the nets of that very
        # structure built so far: building, dropping or re-ordering OTHER nets does not move it.  A training run does
        # not rely on the default: S.set_dist / S.seed re-key every net from OPT.seed and the rank (state.py), which is also what
        # makes the masks of a resumed run repeat.
        import zlib
        crc = zlib.crc32(repr(([tuple(l) for l in self.layers], tuple(in_dims))).encode()) & 0xFFFFFFFF
        ordinal = DeviceNet._count_by_structure.get(crc, 0) + 1
        DeviceNet._count_by_structure[crc] = ordinal
        DeviceNet._count += 1
        # 16-bit tag | the full 32-bit crc | 16-bit ordinal (8 bits wrapped at the 257th net of one structure and the
        # 1st and 257th then shared a Philox stream); past 65 535 nets of one structure the caller has to key them itself
        if ordinal > 0xFFFF:
            raise FgError("more than 65535 nets of one structure in this process: assign net.mask_seed explicitly (state.S.seed does)")
        self.mask_seed, self.mask_offset = (0x4D4B << 48) + (crc << 16) + ordinal, 0
        self._masks = None
        self._batch = 0
        self._x = None

    def __del__(self):
        try:
            self.lib.fg_net_destroy(self.h)
        except Exception:
            pass

    def _init_bn_buffers(self):
        off = 0
        for l in self.layers:
            if l[0] == "BATCHNORM":
                nf = l[1]
                self.buffers[off + nf: off + 2 * nf] = 1.0  # running_var = 1
                off += 2 * nf

    def reserve(self, batch):
        if batch > self.max_batch:
            nbytes = self.lib.fg_net_workspace_bytes(self.h, batch)
            self.ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.ctx.device)
            self.max_batch = batch

    def param_offsets(self, layer_index):
        wo, wn, bo, bn = (ctypes.c_longlong() for _ in range(4))
        self.ctx.check(self.lib.fg_net_param_offset(self.h, layer_index, ctypes.byref(wo), ctypes.byref(wn),
                                                    ctypes.byref(bo), ctypes.byref(bn)))
        return wo.value, wn.value, bo.value, bn.value

    def params_changed(self):
        self.ctx.check(self.lib.fg_net_params_changed(self.h))

    def mask_shape(self, i, batch):
        return (self.lib.fg_net_mask_elems(self.h, i, batch),)

    def draw_masks(self, batch):
        """Bernoulli keep masks for every dropout layer: ONE Philox launch per distinct keep probability (the masks
        of equal-p layers are slices of one buffer); seed/offset advance per call."""
        sizes, keeps = [], []
        k = 0
        for l in self.layers:
            if l[0] in ("SPATIAL_DROPOUT", "DROPOUT"):
                sizes.append(self.lib.fg_net_mask_elems(self.h, k, batch))
                keeps.append(1.0 - float(l[5]))
                k += 1
        masks = [None] * len(sizes)
        for kp in sorted(set(keeps)):
            idx = [i for i, v in enumerate(keeps) if v == kp]
            padded = [(sizes[i] + 3) // 4 * 4 for i in idx]       # 16-byte aligned slices
            buf = self.ctx.bernoulli((sum(padded),), kp, self.mask_seed, self.mask_offset)
            self.mask_offset += sum(padded) // 4
            off = 0
            for i, n in zip(idx, padded):
                masks[i] = buf[off:off + sizes[i]]
                off += n
        return masks

    # ---- sync-BN -------------------------------------------------------------------------------------------------
    def enable_sync_bn(self, reduce_fn):
        """Exact global-batch BatchNorm under data parallelism: `reduce_fn(t)` must sum the fp64 tensor `t` in place
        across ranks (torch.distributed.all_reduce); forward/backward pause at every BatchNorm for it."""
        cmax = max([l[1] for l in self.layers if l[0] == "BATCHNORM"] + [1])
        self.sync_buf = torch.zeros(2 * cmax + 1, dtype=torch.float64, device=self.ctx.device)
        self._sync_reduce = reduce_fn
        self.ctx.check(self.lib.fg_net_set_sync_bn(self.h, 1, self.sync_buf.data_ptr(), self.sync_buf.numel()))

    def disable_sync_bn(self):
        self._sync_reduce = None
        self.ctx.check(self.lib.fg_net_set_sync_bn(self.h, 0, None, 0))

    def _sync_slice(self):
        return self.sync_buf[: self.lib.fg_net_sync_count(self.h)]

    def _drain(self, rc, resume):
        """Run the pause / all-reduce / resume protocol until the pass completes."""
        while rc == 1:
            self._sync_reduce(self._sync_slice())
            rc = resume()
        self.ctx.check(rc)

    # ---- forward / backward ---------------------------------------------------------------------------------------
    def _forward_call(self, x, masks, train):
        train = self.train if train is None else train
        B = x.shape[0]
        self.reserve(B)
        x = x.contiguous()
        if train and self.n_masks:
            if masks is None:
                masks = self.draw_masks(B)
            if len(masks) != self.n_masks:
                raise FgError("forward: %d dropout masks given, the net has %d dropout layers" % (len(masks), self.n_masks))
            self._masks = [m.contiguous() for m in masks]
            mp = (ctypes.c_void_p * self.n_masks)(*[m.data_ptr() for m in self._masks])
        else:
            self._masks, mp = None, None
        self._off = ctypes.c_longlong()
        self._batch, self._x = B, x
        return self.lib.fg_net_forward(self.h, B, x.data_ptr(), self.ws.data_ptr(), self.ws.numel() * 4,
                                       1 if train else 0, mp, self.n_masks if mp is not None else 0,
                                       ctypes.byref(self._off))

    def _forward_resume(self):
        return self.lib.fg_net_forward_resume(self.h, ctypes.byref(self._off))

    def _output_view(self):
        B = self._batch
        n = B * self.out_c * self.out_h * self.out_w
        out = self.ws[self._off.value: self._off.value + n]
        if self.out_h * self.out_w == 1:
            return out.view(B, self.out_c)
        return out.view(B, self.out_h, self.out_w, self.out_c)

    def forward(self, x, masks=None, train=None):
        """x: device NHWC [B,H,W,C] (or [B,F]).  Returns a VIEW into the workspace (NHWC)."""
        self._drain(self._forward_call(x, masks, train), self._forward_resume)
        return self._output_view()

    def forward_steps(self, x, masks=None, train=None):
        """Generator form of forward() for externally driven sync-BN: yields the fp64 sum buffer at every pause (the
        driver reduces it in place across ranks, then advances the generator); the output view is self._output_view()."""
        rc = self._forward_call(x, masks, train)
        while rc == 1:
            yield self._sync_slice()
            rc = self._forward_resume()
        self.ctx.check(rc)

    def backward(self, gy, param_grads=True, input_grad=False):
        """gy: device grad wrt the output (NHWC).  Returns gx (NHWC) if input_grad."""
        gx = self.ctx.empty(*self._x.shape) if input_grad else None
        self._drain(self._backward_call(gy, param_grads, gx), lambda: self.lib.fg_net_backward_resume(self.h))
        return gx

    def _backward_call(self, gy, param_grads, gx, stage_from=None, stage_to=0):
        flags = (_lib.FG_BWD_PARAM_GRADS if param_grads else 0) | (_lib.FG_BWD_INPUT_GRAD if gx is not None else 0)
        gyp = gy.contiguous() if gy is not None else None
        self._gy_keepalive = gyp
        if stage_from is None:
            stage_from = self.lib.fg_net_num_stages(self.h) - 1
        return self.lib.fg_net_backward_range(self.h, self._batch, self._x.data_ptr(),
                                              gyp.data_ptr() if gyp is not None else None, self.ws.data_ptr(),
                                              self.ws.numel() * 4, flags, gx.data_ptr() if gx is not None else None,
                                              stage_from, stage_to)

    def backward_steps(self, gy, param_grads=True):
        rc = self._backward_call(gy, param_grads, None)
        while rc == 1:
            yield self._sync_slice()
            rc = self.lib.fg_net_backward_resume(self.h)
        self.ctx.check(rc)

    def grad_buckets(self, target=900000):
        """Split the plan's stages (output -> input order) into buckets of roughly `target` parameters for the bucketed
        gradient all-reduce: [(stage_from, stage_to, param_lo, param_hi), ...] in execution order."""
        ns = self.lib.fg_net_num_stages(self.h)
        buckets, cur_from, lo_acc, hi_acc = [], ns - 1, None, None
        for st in range(ns - 1, -1, -1):
            lo, hi = ctypes.c_longlong(), ctypes.c_longlong()
            self.ctx.check(self.lib.fg_net_stage_params(self.h, st, ctypes.byref(lo), ctypes.byref(hi)))
            if hi.value > lo.value:
                lo_acc = lo.value if lo_acc is None else min(lo_acc, lo.value)
                hi_acc = hi.value if hi_acc is None else max(hi_acc, hi.value)
            if (hi_acc is not None and hi_acc - lo_acc >= target and st > 0) or st == 0:
                buckets.append((cur_from, st, lo_acc or 0, hi_acc or 0))
                cur_from, lo_acc, hi_acc = st - 1, None, None
        return buckets

    def backward_range(self, gy, stage_from, stage_to, param_grads=True):
        """Stages [stage_from .. stage_to] of the backward pass (see fg_net_backward_range)."""
        self._drain(self._backward_call(gy, param_grads, None, stage_from, stage_to),
                    lambda: self.lib.fg_net_backward_resume(self.h))

    def layer_output(self, layer_index):
        off, c, h, w = ctypes.c_longlong(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.ctx.check(self.lib.fg_net_layer_output(self.h, layer_index, ctypes.byref(off), ctypes.byref(c),
                                                    ctypes.byref(h), ctypes.byref(w)))
        n = self._batch * c.value * h.value * w.value
        t = self.ws[off.value: off.value + n]
        return t.view(self._batch, h.value, w.value, c.value)


class CompositeDeviceNet:
    """ConcatTable{branch nets} -> JoinTable(2) -> tail net (models.lua:305-312), with the DeviceNet surface the trainer
    uses: forward / backward on device tensors, the shared flat params / grads, dropout masks in module order."""

    def __init__(self, ctx, branches, tail, params, grads):
        self.ctx, self.lib = ctx, ctx.lib
        self.branches, self.tail = list(branches), tail
        self.params, self.grads = params, grads
        self.n_params = params.numel()
        b0 = self.branches[0]
        self.in_c, self.in_h, self.in_w = b0.in_c, b0.in_h, b0.in_w
        self.out_c, self.out_h, self.out_w = tail.out_c, tail.out_h, tail.out_w
        self.n_masks = sum(n.n_masks for n in self.branches) + tail.n_masks
        self.n_buffers = 0
        self._train = True
        self._widths = [n.out_c * n.out_h * n.out_w for n in self.branches]

    @property
    def train(self):
        return self._train

    @train.setter
    def train(self, v):
        self._train = bool(v)
        for n in self.branches + [self.tail]:
            n.train = bool(v)

    def _nets(self):
        return self.branches + [self.tail]

    def params_changed(self):
        for n in self._nets():
            n.params_changed()

    def draw_masks(self, batch):
        out = []
        for n in self._nets():
            if n.n_masks:
                out.extend(n.draw_masks(batch))
        return out

    def forward(self, x, masks=None, train=None):
        train = self._train if train is None else train
        if train and self.n_masks and masks is None:
            masks = self.draw_masks(x.shape[0])
        B, k = x.shape[0], 0
        outs = []
        for n in self.branches:
            mk = masks[k:k + n.n_masks] if (masks is not None and n.n_masks) else None
            k += n.n_masks
            outs.append(n.forward(x, masks=mk, train=train).reshape(B, -1))
        joined = self.ctx.empty(B, sum(self._widths))
        if len(outs) != 2:
            raise FgError("CompositeDeviceNet: two branches are built (models.lua:305-309)")
        self.ctx.check(self.lib.fg_concat_channels(self.ctx.h, outs[0].data_ptr(), outs[1].data_ptr(), joined.data_ptr(), B,
                                                   self._widths[0], self._widths[1]))
        self._joined = joined
        mk = masks[k:k + self.tail.n_masks] if (masks is not None and self.tail.n_masks) else None
        return self.tail.forward(joined, masks=mk, train=train)

    def backward(self, gy, param_grads=True, input_grad=False):
        B = self._joined.shape[0]
        gj = self.tail.backward(gy, param_grads=param_grads, input_grad=True)
        g0, g1 = self.ctx.empty(B, self._widths[0]), self.ctx.empty(B, self._widths[1])
        self.ctx.check(self.lib.fg_split_channels(self.ctx.h, gj.data_ptr(), g0.data_ptr(), g1.data_ptr(), B,
                                                  self._widths[0], self._widths[1]))
        gx0 = self.branches[0].backward(g0, param_grads=param_grads, input_grad=input_grad)
        gx1 = self.branches[1].backward(g1, param_grads=param_grads, input_grad=input_grad)
        if not input_grad:
            return None
        gx = torch.empty_like(gx0)
        self.ctx.check(self.lib.fg_add(self.ctx.h, gx0.data_ptr(), gx1.data_ptr(), gx.data_ptr(), gx.numel()))
        return gx


class FusedGan:
    """fg_gan (include/facegen_hip.h, step level): one C call per closure of adversarial.lua / adversarial_c2f.lua.  The step
    workspace is a torch tensor owned here; results are views into it (valid until the next closure of the same kind)."""
    BUF = dict(D_INPUT=0, NOISE=1, D_GRAD_INPUT=2, LOSS=3, CONFUSION=4, OPT_STATE_D=5, OPT_STATE_G=6, D_OUTPUT=7, D_MASKS=8, SYNC_BUF=9)
    NO_UPDATE = 1

    def __init__(self, ctx, dnG, dnD, table_inputs, max_batch):
        self.ctx, self.lib = ctx, ctx.lib
        self.dnG, self.dnD = dnG, dnD
        self.table = 1 if table_inputs else 0
        self.max_batch = int(max_batch)
        dnG.reserve(self.max_batch)
        dnD.reserve(self.max_batch)
        nbytes = self.lib.fg_gan_workspace_bytes(dnG.h, dnD.h, self.table, self.max_batch)
        self.ws = (torch.zeros if ctx.dry else torch.empty)((nbytes + 3) // 4 + 64, dtype=torch.float32, device=ctx.device)
        base = self.ws.data_ptr()
        self._skip = ((-base) % 256) // 4                       # 256-byte aligned start inside the tensor
        h = ctypes.c_void_p()
        ctx.check(self.lib.fg_gan_create(ctx.h, dnG.h, dnD.h, self.table, self.max_batch, base + 4 * self._skip,
                                         nbytes, ctypes.byref(h)))
        self.h = h
        self._bound = None
        self.n_masks = dnD.n_masks

    def __del__(self):
        try:
            self.lib.fg_gan_destroy(self.h)
        except Exception:
            pass

    def _bind(self):
        key = (self.dnG.ws.data_ptr(), self.dnG.ws.numel(), self.dnD.ws.data_ptr(), self.dnD.ws.numel())
        if key != self._bound:
            self.ctx.check(self.lib.fg_gan_bind_workspaces(self.h, key[0], key[1] * 4, key[2], key[3] * 4))
            self._bound = key

    def view(self, what, n=None):
        off, cnt = ctypes.c_longlong(), ctypes.c_longlong()
        self.ctx.check(self.lib.fg_gan_buffer(self.h, self.BUF[what], ctypes.byref(off), ctypes.byref(cnt)))
        n = cnt.value if n is None else n
        if what == "D_OUTPUT":                                   # lives in D's own workspace
            return self.dnD.ws[off.value: off.value + n]
        o = self._skip + off.value
        return self.ws[o: o + n]

    def mask_view(self, i, batch):
        off = self.lib.fg_gan_mask_offset(self.h, i)
        n = self.lib.fg_net_mask_elems(self.dnD.h, i, batch)
        return self.ws[self._skip + off: self._skip + off + n]

    def set_comm(self, coll, sync_bn=False, overlap=1):
        self.ctx.check(self.lib.fg_gan_set_comm(self.h, coll.h if coll is not None else None, 1 if sync_bn else 0, int(overlap)))
        # fg_gan_set_comm (re)bound the sync-BN exchange buffer of BOTH nets to the step workspace: a stand-alone
        # DeviceNet.forward / backward afterwards (createImages, approxParzen run in training mode) must drain through that
        # buffer and this carrier, not through a buffer of its own the library no longer looks at
        on = bool(sync_bn) and coll is not None and coll.get_world_size() > 1
        for dn in (self.dnG, self.dnD):
            if on:
                dn.sync_buf = self.view("SYNC_BUF").view(torch.float64)
                dn._sync_reduce = coll.allreduce_sum_
            else:
                dn.sync_buf, dn._sync_reduce = None, None

    def set_seeds(self, noise_seed, noise_offset, mask_seed, mask_offset):
        self.ctx.check(self.lib.fg_gan_set_seeds(self.h, noise_seed, noise_offset, mask_seed, mask_offset))

    def set_penalty(self, which, l1, l2, clamp):
        self.ctx.check(self.lib.fg_gan_set_penalty(self.h, which, l1, l2, clamp))

    def set_optimizer(self, which, method, cfg):
        m = dict(adam=0, sgd=1, adagrad=2)[method]
        mom = cfg.get("momentum", 0)
        self.ctx.check(self.lib.fg_gan_set_optimizer(
            self.h, which, m, cfg.get("learningRate", -1.0), cfg.get("beta1", 0.9), cfg.get("beta2", 0.999),
            cfg.get("epsilon", 1e-8), mom, cfg.get("dampening", -1.0), cfg.get("weightDecay", 0.0),
            cfg.get("learningRateDecay", 0.0), 1 if cfg.get("nesterov", False) else 0))

    def steps(self, which):
        return self.lib.fg_gan_optimizer_steps(self.h, which)

    def pending(self):
        return bool(self.lib.fg_gan_pending(self.h))

    def finish_pending(self):
        self.ctx.check(self.lib.fg_gan_finish_pending(self.h))

    def update(self, which):
        self.ctx.check(self.lib.fg_gan_update(self.h, which))

    def _mask_array(self, masks):
        if masks is None:
            return None, None
        if len(masks) != self.n_masks:
            raise FgError("step: %d dropout masks given, D has %d dropout layers" % (len(masks), self.n_masks))
        keep = [m.contiguous() for m in masks]
        return (ctypes.c_void_p * self.n_masks)(*[m.data_ptr() for m in keep]), keep

    @staticmethod
    def _p(t):
        return t.data_ptr() if t is not None else None

    def step_D(self, B, real, cond_real=None, cond_fake=None, noise=None, masks=None, flags=0):
        if B > self.max_batch:
            raise FgError("step_D: batch %d exceeds the %d the step object was built for" % (B, self.max_batch))
        self._bind()
        mp, keep = self._mask_array(masks)
        args = [t.contiguous() if t is not None else None for t in (real, cond_real, cond_fake, noise)]
        self.ctx.check(self.lib.fg_step_D(self.h, B, self._p(args[0]), self._p(args[1]), self._p(args[2]), self._p(args[3]),
                                          mp, flags))
        self._after(B, B // 2)

    def step_G(self, B, cond=None, noise=None, masks=None, flags=0):
        if B > self.max_batch:
            raise FgError("step_G: batch %d exceeds the %d the step object was built for" % (B, self.max_batch))
        self._bind()
        mp, keep = self._mask_array(masks)
        args = [t.contiguous() if t is not None else None for t in (cond, noise)]
        self.ctx.check(self.lib.fg_step_G(self.h, B, self._p(args[0]), self._p(args[1]), mp, flags))
        self._after(B, B)

    def _after(self, B, Bg):
        # keep the DeviceNets' own bookkeeping in step (layer_output / a later stand-alone backward read it)
        self.dnD._batch, self.dnG._batch = B, Bg
