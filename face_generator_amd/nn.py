"""Host-side mirror of the Torch7 `nn` surface the reference scripts touch (SURVEY.md 8(b) conformance list).

Modules are descriptors holding host (torch CPU float) parameters in REFERENCE layout until the net is moved to
the device (`:cuda()` / NN_UTILS.activateCuda), where one fg_net plan + one flat parameter vector take over and
`module.weight` etc. become views into that flat device vector (== Module:getParameters(), train.lua:151-152).
Compute always runs in libfacegen_hip.so; there is no CPU execution path (forward on a non-device net raises).
"""
import math

import torch

from ._lib import FgError


class Module:
    _typename = "nn.Module"

    def __init__(self):
        self.train = True
        self.output = None
        self.gradInput = None

    # parameters in reference order: weight then bias (Module.lua)
    def param_names(self):
        return [n for n in ("weight", "bias") if getattr(self, n, None) is not None]

    def spec(self):
        raise NotImplementedError

    def training(self):
        self.train = True
        return self

    def evaluate(self):
        self.train = False
        return self

    # ---- nn.Module protocol on DEVICE tensors (internal NHWC layout), one C entry per call (SURVEY 8(b) level i) ----
    def _dev(self, x):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise FgError("%s: module-level calls take device (NHWC) tensors -- go through nn.Copy / "
                          "Sequential:forward for host tensors; there is no CPU path" % self._typename)
        return x.contiguous()

    def updateOutput(self, input):
        raise FgError("%s:updateOutput is not built" % self._typename)

    def updateGradInput(self, input, gradOutput):
        raise FgError("%s:updateGradInput is not built" % self._typename)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        pass

    def forward(self, x):
        return self.updateOutput(x)

    def backward(self, input, gradOutput, scale=1.0):
        gi = self.updateGradInput(input, gradOutput)
        self.accGradParameters(input, gradOutput, scale)
        return gi

    def zeroGradParameters(self):
        for n in ("gradWeight", "gradBias"):
            if getattr(self, n, None) is not None:
                getattr(self, n).zero_()

    def __repr__(self):
        return self._typename


def _uniform(shape, s, gen):
    return (torch.rand(shape, generator=gen) * 2 - 1) * s


class Linear(Module):
    _typename = "nn.Linear"

    def __init__(self, inputSize, outputSize, gen=None):
        super().__init__()
        s = 1.0 / math.sqrt(inputSize)
        self.weight = _uniform((outputSize, inputSize), s, gen)
        self.bias = _uniform((outputSize,), s, gen)
        self.gradWeight = torch.zeros_like(self.weight)
        self.gradBias = torch.zeros_like(self.bias)

    def reset(self, stdv=None, gen=None):
        """Linear.lua reset(stdv): an explicit stdv is scaled by sqrt(3) (so that U(-s, s) has that standard deviation), the
        default is 1 / sqrt(inputSize); weight and bias are both re-drawn from U(-s, s)."""
        s = stdv * math.sqrt(3.0) if stdv else 1.0 / math.sqrt(self.weight.shape[1])
        self.weight.copy_(_uniform(tuple(self.weight.shape), s, gen))
        self.bias.copy_(_uniform(tuple(self.bias.shape), s, gen))
        return self

    def spec(self):
        return ("LINEAR", self.weight.shape[1], self.weight.shape[0])

    def updateOutput(self, input):
        from . import ops
        self.output = ops.linear_forward(self._dev(input), self.weight, self.bias)
        return self.output

    def updateGradInput(self, input, gradOutput):
        from . import ops
        self.gradInput = ops.linear_backward_data(self._dev(gradOutput), self.weight)
        return self.gradInput

    def accGradParameters(self, input, gradOutput, scale=1.0):
        from . import ops
        gy = self._dev(gradOutput) if scale == 1.0 else self._dev(gradOutput) * scale
        ops.linear_backward_weight(self._dev(input), gy, gw=self.gradWeight, gb=self.gradBias, beta=1.0)

    def __repr__(self):
        return "nn.Linear(%d -> %d)" % (self.weight.shape[1], self.weight.shape[0])


class View(Module):
    _typename = "nn.View"

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = tuple(int(s) for s in sizes)

    def spec(self):
        if len(self.sizes) == 3:
            return ("VIEW",) + self.sizes
        if len(self.sizes) == 1:
            return ("VIEW", self.sizes[0], 0, 0)
        raise FgError("nn.View: only View(C,H,W) and View(features) are supported")

    def updateOutput(self, input):
        """The reference reshapes NCHW-contiguous memory; device tensors are NHWC, so a View between a feature
        vector (NCHW flatten order) and a spatial tensor is a real permutation here."""
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        B = x.shape[0]
        if len(self.sizes) == 3 and x.dim() == 2:
            c, h, w = self.sizes
            self.output = ctx.to_device_nhwc(x.view(B, c, h, w))
        elif len(self.sizes) == 1 and x.dim() == 4:
            self.output = ctx.to_nchw(x).view(B, -1)
        else:
            self.output = x.view((B,) + self.sizes)
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        g = self._dev(gradOutput)
        if input.dim() == 2 and g.dim() == 4:
            self.gradInput = ctx.to_nchw(g).view(input.shape)
        elif input.dim() == 4 and g.dim() == 2:
            B, h, w, c = input.shape
            self.gradInput = ctx.to_device_nhwc(g.view(B, c, h, w))
        else:
            self.gradInput = g.view(input.shape)
        return self.gradInput

    def __repr__(self):
        return "nn.View(%s)" % ", ".join(map(str, self.sizes))


class PReLU(Module):
    _typename = "nn.PReLU"

    def __init__(self, *ignored):  # models.lua passes (nil, nil, true); upstream ignores extra ctor args
        super().__init__()
        self.weight = torch.full((1,), 0.25)
        self.gradWeight = torch.zeros(1)

    def spec(self):
        return ("PRELU",)

    def updateOutput(self, input):
        from . import ops
        self.output = ops.prelu_forward(self._dev(input), self.weight)
        return self.output

    def updateGradInput(self, input, gradOutput):
        from . import ops
        self.gradInput, self._gs = ops.prelu_backward(self._dev(input), self._dev(gradOutput), self.weight)
        return self.gradInput

    def accGradParameters(self, input, gradOutput, scale=1.0):
        if getattr(self, "_gs", None) is None:
            self.updateGradInput(input, gradOutput)
        self.gradWeight.add_(self._gs, alpha=scale)
        self._gs = None


class LeakyReLU(Module):
    """LeakyReLU.lua:7-31."""
    _typename = "nn.LeakyReLU"

    def __init__(self, negval=0.333):
        super().__init__()
        self.negval = negval

    def spec(self):
        return ("LEAKYRELU", 0, 0, 0, 0, self.negval)

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        self.output = torch.empty_like(x)
        ctx.check(ctx.lib.fg_leakyrelu_forward(ctx.h, x.data_ptr(), self.negval, self.output.data_ptr(), x.numel()))
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x, g = self._dev(input), self._dev(gradOutput)
        self.gradInput = torch.empty_like(x)
        ctx.check(ctx.lib.fg_leakyrelu_backward(ctx.h, x.data_ptr(), g.data_ptr(), self.negval, self.gradInput.data_ptr(), x.numel()))
        return self.gradInput


class SpatialUpSamplingNearest(Module):
    _typename = "nn.SpatialUpSamplingNearest"

    def __init__(self, scale):
        super().__init__()
        if scale != 2:
            raise FgError("nn.SpatialUpSamplingNearest: only scale 2 is built (models.lua:63, 68)")

    def spec(self):
        return ("UPSAMPLE2X",)

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        B, H, W, C = x.shape
        self.output = ctx.empty(B, 2 * H, 2 * W, C)
        ctx.check(ctx.lib.fg_upsample_nearest2x_forward(ctx.h, x.data_ptr(), self.output.data_ptr(), B, H, W, C))
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        B, H, W, C = input.shape
        g = self._dev(gradOutput)
        self.gradInput = ctx.empty(B, H, W, C)
        ctx.check(ctx.lib.fg_upsample_nearest2x_backward(ctx.h, g.data_ptr(), self.gradInput.data_ptr(), B, H, W, C))
        return self.gradInput


class SpatialConvolution(Module):
    """nn.SpatialConvolution / cudnn.SpatialConvolution(nIn, nOut, kW, kH, dW, dH, padW[, padH])."""
    _typename = "nn.SpatialConvolution"

    def __init__(self, nInputPlane, nOutputPlane, kW, kH, dW=1, dH=1, padW=0, padH=None, gen=None):
        super().__init__()
        padH = padW if padH is None else padH
        if kW != kH or dW != dH or dW not in (1, 2) or padW != padH or kW % 2 != 1 or padW != (kW - 1) // 2:
            raise FgError("SpatialConvolution: only square odd kernels, stride 1 or 2, 'same' padding are built")
        self.nInputPlane, self.nOutputPlane, self.kW, self.padW = nInputPlane, nOutputPlane, kW, int(padW)
        self.dW = int(dW)          # 2: the stride-2 convs of create_D16_d (models.lua:289-291), plan level only
        s = 1.0 / math.sqrt(kW * kH * nInputPlane)
        self.weight = _uniform((nOutputPlane, nInputPlane, kH, kW), s, gen)
        self.bias = _uniform((nOutputPlane,), s, gen)
        self.gradWeight = torch.zeros_like(self.weight)
        self.gradBias = torch.zeros_like(self.bias)

    def reset(self, stdv=None, gen=None):
        """SpatialConvolution.lua reset(stdv): stdv * sqrt(3), default 1 / sqrt(kW * kH * nInputPlane); weight and bias ~ U(-s, s)."""
        s = stdv * math.sqrt(3.0) if stdv else 1.0 / math.sqrt(self.kW * self.kW * self.nInputPlane)
        self.weight.copy_(_uniform(tuple(self.weight.shape), s, gen))
        self.bias.copy_(_uniform(tuple(self.bias.shape), s, gen))
        return self

    def spec(self):
        return ("CONV", self.nInputPlane, self.nOutputPlane, self.kW, self.padW, float(getattr(self, "dW", 1)))

    def updateOutput(self, input):
        from . import ops
        if getattr(self, "dW", 1) != 1:
            raise FgError("SpatialConvolution (stride 2): module-level entry not built; run it through the compiled net")
        self.output = ops.conv2d_forward(self._dev(input), self.weight, self.bias, pad=self.padW)
        return self.output

    def updateGradInput(self, input, gradOutput):
        from . import ops
        self.gradInput = ops.conv2d_backward_data(self._dev(gradOutput), self.weight, tuple(input.shape[1:3]), pad=self.padW)
        return self.gradInput

    def accGradParameters(self, input, gradOutput, scale=1.0):
        from . import ops
        gy = self._dev(gradOutput) if scale == 1.0 else self._dev(gradOutput) * scale
        ops.conv2d_backward_weight(self._dev(input), gy, self.kW, pad=self.padW, gw=self.gradWeight, gb=self.gradBias, beta=1.0)

    def __repr__(self):
        return "%s(%d -> %d, %dx%d, 1,1, %d,%d)" % (self._typename, self.nInputPlane, self.nOutputPlane, self.kW,
                                                    self.kW, self.padW, self.padW)


def CudnnSpatialConvolution(*args, **kw):
    """cudnn.SpatialConvolution (models.lua:63, 68, 73): the same module under the cudnn rock's type name.  The name matters to
    callers that dispatch on `__typename` -- weight-init.lua:56-71 resets `nn.SpatialConvolution` but NOT `cudnn.SpatialConvolution`
    (it only zeroes its bias) -- and to the checkpoint writer (t7_checkpoint.py)."""
    m = SpatialConvolution(*args, **kw)
    m._typename = "cudnn.SpatialConvolution"
    return m


class SpatialBatchNormalization(Module):
    _typename = "nn.SpatialBatchNormalization"

    def __init__(self, nFeature, eps=1e-5, momentum=0.1, gen=None):
        super().__init__()
        self.nFeature, self.eps, self.momentum = nFeature, eps, momentum
        self.weight = torch.rand(nFeature, generator=gen)   # reset(): gamma ~ U(0,1)
        self.bias = torch.zeros(nFeature)
        self.gradWeight = torch.zeros(nFeature)
        self.gradBias = torch.zeros(nFeature)
        self.running_mean = torch.zeros(nFeature)
        self.running_var = torch.ones(nFeature)

    def spec(self):
        return ("BATCHNORM", self.nFeature, 0, 0, 0, self.eps, self.momentum)

    def updateOutput(self, input):
        from . import ops
        self.output, self.save_mean, self.save_invstd = ops.batchnorm_forward(
            self._dev(input), self.weight, self.bias, None, self.running_mean, self.running_var, self.eps, self.momentum,
            train=self.train)
        return self.output

    def updateGradInput(self, input, gradOutput):
        from . import ops
        if not self.train:
            raise FgError("nn.SpatialBatchNormalization: backward in evaluate mode is not built")
        self.gradInput, self._gg, self._gb, _ = ops.batchnorm_backward(self._dev(input), self._dev(gradOutput), self.weight,
                                                                        self.bias, self.save_mean, self.save_invstd)
        return self.gradInput

    def accGradParameters(self, input, gradOutput, scale=1.0):
        if getattr(self, "_gg", None) is None:
            self.updateGradInput(input, gradOutput)
        self.gradWeight.add_(self._gg, alpha=scale)
        self.gradBias.add_(self._gb, alpha=scale)
        self._gg = None


class SpatialDropout(Module):
    _typename = "nn.SpatialDropout"

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def spec(self):
        return ("SPATIAL_DROPOUT", 0, 0, 0, 0, self.p)

    def _apply(self, x, train):
        from .runtime import get_context
        ctx = get_context(x.device.index)
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        ctx.check(ctx.lib.fg_spatial_dropout_apply(ctx.h, x.data_ptr(), self.noise.data_ptr() if train else None,
                                                   1.0 if train else 1.0 - self.p, y.data_ptr(), B, H * W, C))
        return y

    def updateOutput(self, input):
        from .runtime import get_context
        x = self._dev(input)
        if self.train and getattr(self, "_injected", None) is None:
            ctx = get_context(x.device.index)
            self._seed = getattr(self, "_seed", 0) + 1
            self.noise = ctx.bernoulli((x.shape[0] * x.shape[3],), 1.0 - self.p, 7919, self._seed * 65536)
        elif self.train:
            self.noise = self._injected
        self.output = self._apply(x, self.train)
        return self.output

    def updateGradInput(self, input, gradOutput):
        if not self.train:
            raise FgError("nn.SpatialDropout: backward is an error in evaluate mode (upstream)")
        self.gradInput = self._apply(self._dev(gradOutput), True)
        return self.gradInput

    def set_mask(self, mask_dev):
        """Inject the [B][C] keep mask (reproducible runs / parity tests)."""
        self._injected = mask_dev.contiguous().reshape(-1)


class Dropout(Module):
    _typename = "nn.Dropout"

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def spec(self):
        return ("DROPOUT", 0, 0, 0, 0, self.p)

    def _apply(self, x, train):
        from .runtime import get_context
        ctx = get_context(x.device.index)
        y = torch.empty_like(x)
        ctx.check(ctx.lib.fg_dropout_apply(ctx.h, x.data_ptr(), self.noise.data_ptr() if train else None,
                                           1.0 / (1.0 - self.p) if train else 1.0, y.data_ptr(), x.numel()))
        return y

    def updateOutput(self, input):
        from .runtime import get_context
        x = self._dev(input)
        if self.train and getattr(self, "_injected", None) is None:
            ctx = get_context(x.device.index)
            self._seed = getattr(self, "_seed", 0) + 1
            self.noise = ctx.bernoulli((x.numel(),), 1.0 - self.p, 7907, self._seed * 1048576)
        elif self.train:
            self.noise = self._injected
        self.output = self._apply(x, self.train)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = self._apply(self._dev(gradOutput), self.train)
        return self.gradInput

    def set_mask(self, mask_dev):
        self._injected = mask_dev.contiguous().reshape(-1)


class SpatialAveragePooling(Module):
    _typename = "nn.SpatialAveragePooling"

    def __init__(self, kW, kH, dW=None, dH=None):
        super().__init__()
        if (kW, kH, dW or kW, dH or kH) != (2, 2, 2, 2):
            raise FgError("nn.SpatialAveragePooling: only (2,2,2,2) is built (models.lua:388-403)")

    def spec(self):
        return ("AVGPOOL2",)

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        B, H, W, C = x.shape
        self.output = ctx.empty(B, H // 2, W // 2, C)
        ctx.check(ctx.lib.fg_avgpool2x2_forward(ctx.h, x.data_ptr(), self.output.data_ptr(), B, H, W, C))
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        B, H, W, C = input.shape
        self.gradInput = ctx.empty(B, H, W, C)
        ctx.check(ctx.lib.fg_avgpool2x2_backward(ctx.h, self._dev(gradOutput).data_ptr(), self.gradInput.data_ptr(), B, H, W, C))
        return self.gradInput


class SpatialMaxPooling(Module):
    """nn.SpatialMaxPooling(2,2) (models_c2f.lua:251, 256)."""
    _typename = "nn.SpatialMaxPooling"

    def __init__(self, kW, kH, dW=None, dH=None):
        super().__init__()
        if (kW, kH, dW or kW, dH or kH) != (2, 2, 2, 2):
            raise FgError("nn.SpatialMaxPooling: only (2,2) is built")

    def spec(self):
        return ("MAXPOOL2",)

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        B, H, W, C = x.shape
        self.output = ctx.empty(B, H // 2, W // 2, C)
        ctx.check(ctx.lib.fg_maxpool2x2_forward(ctx.h, x.data_ptr(), self.output.data_ptr(), B, H, W, C))
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        B, H, W, C = x.shape
        self.gradInput = ctx.empty(B, H, W, C)
        ctx.check(ctx.lib.fg_maxpool2x2_backward(ctx.h, x.data_ptr(), self._dev(gradOutput).data_ptr(),
                                                 self.gradInput.data_ptr(), B, H, W, C))
        return self.gradInput


class SpatialConvolutionUpsample(SpatialConvolution):
    """layers/cudnnSpatialConvolutionUpsample.lua:4-58: a 'same' convolution to nOutputPlane * factor^2 planes whose contiguous
    NCHW output is RE-VIEWED (Tensor:view -- a flat reinterpretation, not a pixel shuffle) as [N][nOutputPlane][h*f][w*f];
    updateGradInput / accGradParameters view gradOutput back.  The reference's nets always pass factor = 1
    (models_c2f.lua:123-131), where it is exactly a 'same' convolution; factor > 1 runs the same convolution kernels followed by
    an index-mapped copy (fg_conv_upsample_view_*; inside a compiled net: one extra stage)."""
    _typename = "cudnn.SpatialConvolutionUpsample"

    def __init__(self, nInputPlane, nOutputPlane, kW, kH, factor=2, groups=None, gen=None):
        if kW % 2 != 1 or kH % 2 != 1:
            raise FgError("kW has to be odd / kH has to be odd")          # cudnnSpatialConvolutionUpsample.lua:7-8
        factor = int(factor)
        if factor < 1:
            raise FgError("SpatialConvolutionUpsample: factor must be >= 1")
        super().__init__(nInputPlane, nOutputPlane * factor * factor, kW, kH, 1, 1, (kW - 1) // 2, (kH - 1) // 2, gen=gen)
        self.factor = factor
        self.nInputPlaneU, self.nOutputPlaneU = nInputPlane, nOutputPlane

    def spec(self):
        return ("CONV", self.nInputPlane, self.nOutputPlane, self.kW, self.padW, 1.0, float(self.factor))

    def _view(self, t, backward):
        """conv output NHWC [B][h][w][nOut f^2] <-> viewed NHWC [B][h f][w f][nOut] (gradients: the other way)."""
        from .runtime import get_context
        ctx = get_context(t.device.index)
        f, C = self.factor, self.nOutputPlane
        if backward:
            B, hf, wf, _ = t.shape
            h, w = hf // f, wf // f
            out = ctx.empty(B, h, w, C)
            ctx.check(ctx.lib.fg_conv_upsample_view_backward(ctx.h, t.contiguous().data_ptr(), out.data_ptr(), B, h, w, C, f))
        else:
            B, h, w, _ = t.shape
            out = ctx.empty(B, h * f, w * f, C // (f * f))
            ctx.check(ctx.lib.fg_conv_upsample_view_forward(ctx.h, t.contiguous().data_ptr(), out.data_ptr(), B, h, w, C, f))
        return out

    def updateOutput(self, input):
        y = super().updateOutput(input)
        self.output = y if self.factor == 1 else self._view(y, False)
        return self.output

    def updateGradInput(self, input, gradOutput):
        g = self._dev(gradOutput)
        return super().updateGradInput(input, g if self.factor == 1 else self._view(g, True))

    def accGradParameters(self, input, gradOutput, scale=1.0):
        g = self._dev(gradOutput)
        return super().accGradParameters(input, g if self.factor == 1 else self._view(g, True), scale)


class JoinTable(Module):
    """nn.JoinTable(2, 2): concat along channels (models_c2f.lua:116)."""
    _typename = "nn.JoinTable"

    def __init__(self, dimension=2, nInputDims=None):
        super().__init__()
        if dimension != 2:
            raise FgError("nn.JoinTable: only dimension 2 (channels / features) is built")


class ConcatTable(Module):
    """nn.ConcatTable (models.lua:307-309): every branch gets the same input; output = table of the branch outputs."""
    _typename = "nn.ConcatTable"

    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    def __repr__(self):
        return "nn.ConcatTable {\n" + "\n".join("  |-- %s" % repr(m).replace("\n", "\n  |   ") for m in self.modules) + "\n}"


class CAddTable(Module):
    """nn.CAddTable (models_c2f.lua:240)."""
    _typename = "nn.CAddTable"


class Sigmoid(Module):
    _typename = "nn.Sigmoid"

    def spec(self):
        return ("SIGMOID",)

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        x = self._dev(input)
        self.output = torch.empty_like(x)
        ctx.check(ctx.lib.fg_sigmoid_forward(ctx.h, x.data_ptr(), self.output.data_ptr(), x.numel()))
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context(input.device.index)
        g = self._dev(gradOutput)
        self.gradInput = torch.empty_like(g)
        ctx.check(ctx.lib.fg_sigmoid_backward(ctx.h, self.output.data_ptr(), g.data_ptr(), self.gradInput.data_ptr(), g.numel()))
        return self.gradInput


class Copy(Module):
    """nn.Copy(inType, outType): the host<->device (and NCHW<->NHWC) boundary of NN_UTILS.activateCuda."""
    _typename = "nn.Copy"

    def __init__(self, intype, outtype):
        super().__init__()
        self.intype, self.outtype = intype, outtype

    def updateOutput(self, input):
        from .runtime import get_context
        ctx = get_context()
        if self.outtype.startswith("hip"):
            self.output = ctx.to_device_nhwc(input)
        else:
            self.output = ctx.to_nchw(input).cpu()
        return self.output

    def updateGradInput(self, input, gradOutput):
        from .runtime import get_context
        ctx = get_context()
        self.gradInput = ctx.to_nchw(gradOutput).cpu() if self.outtype.startswith("hip") else ctx.to_device_nhwc(gradOutput)
        return self.gradInput

    def __repr__(self):
        return "nn.Copy(%s -> %s)" % (self.intype, self.outtype)


class Sequential(Module):
    _typename = "nn.Sequential"

    def __init__(self):
        super().__init__()
        self.modules = []
        self.input_dims = None     # (C, H, W) of one input sample; set by MODELS.create_*
        self.device_net = None     # runtime.DeviceNet once on the device
        self._flat = None

    def add(self, m):
        self.modules.append(m)
        return self

    def get(self, i):
        return self.modules[i - 1]   # 1-based like Lua

    def size(self):
        return len(self.modules)

    def listModules(self):
        out = [self]
        for m in self.modules:
            out.extend(m.listModules() if isinstance(m, Sequential) else [m])
        return out

    def training(self):
        self.train = True
        for m in self.modules:
            m.training()
        if self.device_net is not None:
            self.device_net.train = True
        return self

    def evaluate(self):
        self.train = False
        for m in self.modules:
            m.evaluate()
        if self.device_net is not None:
            self.device_net.train = False
        return self

    def _inner(self):
        """The compute Sequential: self, or the middle of {Copy, net, Copy}."""
        if len(self.modules) == 3 and isinstance(self.modules[0], Copy) and isinstance(self.modules[1], Sequential):
            return self.modules[1]
        return self

    def layer_specs(self):
        return [m.spec() for m in self.modules]

    def parameter_list(self):
        """[(module, 'weight'|'bias')...] in Module:parameters() order."""
        out = []
        for m in self.modules:
            if isinstance(m, Sequential):
                out.extend(m.parameter_list())
            elif not isinstance(m, Copy):
                out.extend((m, n) for n in m.param_names())
        return out

    def is_cuda(self):
        return self._inner().device_net is not None

    def cuda(self, ctx=None, max_batch=32):
        """:cuda() -- compile to an fg_net and move parameters / BN buffers to the device."""
        from . import runtime
        inner = self._inner()
        if inner.device_net is not None:
            return self
        if inner.input_dims is None:
            raise FgError("Sequential:cuda(): input_dims unknown (build the net through MODELS.create_*)")
        ctx = ctx or runtime.get_context()
        dn = runtime.DeviceNet(ctx, inner.layer_specs(), inner.input_dims, max_batch)
        off = 0
        boff = 0
        for (m, name) in inner.parameter_list():
            w = getattr(m, name)
            n = w.numel()
            dn.params[off:off + n] = w.reshape(-1).to(ctx.device)
            setattr(m, name, dn.params[off:off + n].view(w.shape))
            gname = "gradWeight" if name == "weight" else "gradBias"
            setattr(m, gname, dn.grads[off:off + n].view(w.shape))
            off += n
        assert off == dn.n_params, (off, dn.n_params)
        for m in inner.modules:
            if isinstance(m, SpatialBatchNormalization):
                nf = m.nFeature
                dn.buffers[boff:boff + nf] = m.running_mean.to(ctx.device)
                dn.buffers[boff + nf:boff + 2 * nf] = m.running_var.to(ctx.device)
                m.running_mean = dn.buffers[boff:boff + nf]
                m.running_var = dn.buffers[boff + nf:boff + 2 * nf]
                boff += 2 * nf
        dn.train = inner.train
        dn.params_changed()
        inner.device_net = dn
        return self

    def float(self):
        """:float() -- pull parameters back to host tensors and drop the device plan."""
        inner = self._inner()
        if inner.device_net is None:
            return self
        for (m, name) in inner.parameter_list():
            setattr(m, name, getattr(m, name).detach().cpu().clone())
            gname = "gradWeight" if name == "weight" else "gradBias"
            setattr(m, gname, getattr(m, gname).detach().cpu().clone())
        for m in inner.modules:
            if isinstance(m, SpatialBatchNormalization):
                m.running_mean = m.running_mean.cpu().clone()
                m.running_var = m.running_var.cpu().clone()
        inner.device_net = None
        return self

    def getParameters(self):
        """-> (flatParameters, flatGradParameters): the net's two flat vectors (device tensors)."""
        inner = self._inner()
        if inner.device_net is None:
            raise FgError("getParameters(): move the net to the device first (NN_UTILS.activateCuda / :cuda())")
        return inner.device_net.params, inner.device_net.grads

    # ---- Module:forward / :backward on host NCHW tensors (the reference's GPU mode returns Float tensors too)
    def forward(self, x, masks=None):
        inner = self._inner()
        dn = inner.device_net
        if dn is None:
            raise FgError("Sequential:forward: no CPU path -- call NN_UTILS.activateCuda(net) / net:cuda() first")
        ctx = dn.ctx
        xd = ctx.to_device_nhwc(x)
        y = dn.forward(xd, masks=masks, train=inner.train)
        self.output = ctx.to_nchw(y).cpu()
        return self.output

    def backward(self, x, gradOutput):
        inner = self._inner()
        dn = inner.device_net
        if dn is None:
            raise FgError("Sequential:backward: no CPU path")
        ctx = dn.ctx
        gy = ctx.to_device_nhwc(gradOutput)
        gx = dn.backward(gy, param_grads=True, input_grad=True)
        self.gradInput = ctx.to_nchw(gx).cpu()
        if self is not inner:
            self.modules[0].gradInput = self.gradInput   # adversarial.lua:210 reads MODEL_D.modules[1].gradInput
        return self.gradInput

    # ---- module-by-module execution (nn.Sequential:updateOutput / :backward of upstream nn): one C entry per module
    #      call, no fusion.  Same results as the compiled plan; used for per-module access and as a cross-check.
    def forward_modules(self, x_dev):
        self._mod_inputs = []
        for m in self.modules:
            self._mod_inputs.append(x_dev)
            x_dev = m.updateOutput(x_dev)
        self.output = x_dev
        return x_dev

    def backward_modules(self, gy_dev, scale=1.0):
        for m, xi in zip(reversed(self.modules), reversed(self._mod_inputs)):
            gy_dev = m.backward(xi, gy_dev, scale)
        self.gradInput = gy_dev
        return gy_dev

    def zeroGradParameters(self):
        for m in self.modules:
            m.zeroGradParameters()

    def __repr__(self):
        lines = ["nn.Sequential {"]
        for i, m in enumerate(self.modules):
            lines.append("  (%d): %s" % (i + 1, repr(m).replace("\n", "\n  ")))
        lines.append("}")
        return "\n".join(lines)


class TableSequential(Sequential):
    """models_c2f.lua nets: {JoinTable | CAddTable, [Copy], inner Sequential, [Copy]} taking a table of two tensors.
    `modules` mirrors the reference's outer container; the compute net is `inner`."""

    def __init__(self, first, inner):
        super().__init__()
        self.first, self.inner = first, inner
        self.modules = [first, inner]

    def _inner(self):
        return self.inner

    def cuda(self, ctx=None, max_batch=32):
        self.inner.cuda(ctx, max_batch)
        if not isinstance(self.modules[1], Copy):
            self.modules = [self.first, Copy("torch.FloatTensor", "hip.NHWC"), self.inner, Copy("hip.NHWC", "torch.FloatTensor")]
        return self

    def is_cuda(self):
        return self.inner.device_net is not None

    def combine_device(self, ctx, a, b):
        """JoinTable / CAddTable on device NHWC tensors."""
        if isinstance(self.first, JoinTable):
            out = ctx.empty(a.shape[0], a.shape[1], a.shape[2], a.shape[3] + b.shape[3])
            ctx.check(ctx.lib.fg_concat_channels(ctx.h, a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                 a.shape[0] * a.shape[1] * a.shape[2], a.shape[3], b.shape[3]))
        else:
            out = torch.empty_like(a)
            ctx.check(ctx.lib.fg_add(ctx.h, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel()))
        return out

    def forward(self, xs, masks=None):
        dn = self.inner.device_net
        if dn is None:
            raise FgError("TableSequential:forward: no CPU path -- build with cuda=true / call :cuda() first")
        ctx = dn.ctx
        a, b = ctx.to_device_nhwc(xs[0]), ctx.to_device_nhwc(xs[1])
        y = dn.forward(self.combine_device(ctx, a.contiguous(), b.contiguous()), masks=masks, train=self.inner.train)
        self.output = ctx.to_nchw(y).cpu()
        return self.output

    def backward(self, xs, gradOutput):
        dn = self.inner.device_net
        ctx = dn.ctx
        g = dn.backward(ctx.to_device_nhwc(gradOutput), param_grads=True, input_grad=True)
        if isinstance(self.first, JoinTable):
            ca = torch.as_tensor(xs[0]).shape[1]
            cb = g.shape[3] - ca
            ga, gb = ctx.empty(*g.shape[:3], ca), ctx.empty(*g.shape[:3], cb)
            ctx.check(ctx.lib.fg_split_channels(ctx.h, g.data_ptr(), ga.data_ptr(), gb.data_ptr(),
                                                g.shape[0] * g.shape[1] * g.shape[2], ca, cb))
            self.gradInput = [ctx.to_nchw(ga).cpu(), ctx.to_nchw(gb).cpu()]
        else:
            gh = ctx.to_nchw(g).cpu()
            self.gradInput = [gh, gh]           # CAddTable: the same gradient for both inputs
        return self.gradInput

    def training(self):
        self.train = True
        self.inner.training()
        return self

    def evaluate(self):
        self.train = False
        self.inner.evaluate()
        return self


class BCECriterion:
    """nn.BCECriterion() (train.lua:148) on device tensors; forward returns a python float like Torch."""

    def __init__(self):
        self.sizeAverage = True
        self.output = None
        self.gradInput = None

    def forward_backward_device(self, ctx, prob, target, want_confusion=True):
        """prob, target: device [B].  -> (loss device scalar, grad device [B], confusion device int32[4])."""
        B = prob.numel()
        loss = ctx.empty(1)
        grad = ctx.empty(B)
        conf = torch.empty(4, dtype=torch.int32, device=ctx.device)   # fully overwritten by the kernel
        ctx.check(ctx.lib.fg_bce_forward_backward(ctx.h, prob.data_ptr(), target.data_ptr(), B, loss.data_ptr(),
                                                  grad.data_ptr(), conf.data_ptr() if want_confusion else None))
        return loss, grad, conf

    def forward(self, input, target):
        from . import runtime
        ctx = runtime.get_context()
        p = torch.as_tensor(input, dtype=torch.float32).reshape(-1).to(ctx.device)
        t = torch.as_tensor(target, dtype=torch.float32).reshape(-1).to(ctx.device)
        loss, grad, _ = self.forward_backward_device(ctx, p, t, False)
        self.output = float(loss.item())
        self._grad = grad.cpu().reshape(torch.as_tensor(input).shape)
        return self.output

    def backward(self, input, target):
        if getattr(self, "_grad", None) is None:
            self.forward(input, target)
        self.gradInput = self._grad
        return self.gradInput


class ConcatSequential(Sequential):
    """models.lua:279-316 create_D16_d: nn.Sequential{ConcatTable{branch...}, JoinTable(2), tail...}.  Every branch and
    the tail is compiled to its own fg_net; they share ONE flat parameter / gradient vector in the reference's order
    (branch 1, branch 2, ..., tail), so getParameters(), the optimizers and the all-reduce see a single net."""

    def __init__(self, branches, tail):
        super().__init__()
        self.branches, self.tail = list(branches), tail
        ct = ConcatTable()
        for b in self.branches:
            ct.add(b)
        self.modules = [ct, JoinTable(2)] + list(tail.modules)

    def _inner(self):
        return self

    def _parts(self):
        return self.branches + [self.tail]

    def parameter_list(self):
        out = []
        for p in self._parts():
            out.extend(p.parameter_list())
        return out

    def listModules(self):
        out = [self, self.modules[0]]
        for b in self.branches:
            out.extend(b.listModules())
        out.append(self.modules[1])
        out.extend(self.tail.modules)
        return out

    def training(self):
        self.train = True
        for p in self._parts():
            p.training()
        if self.device_net is not None:
            self.device_net.train = True
        return self

    def evaluate(self):
        self.train = False
        for p in self._parts():
            p.evaluate()
        if self.device_net is not None:
            self.device_net.train = False
        return self

    def zeroGradParameters(self):
        for p in self._parts():
            p.zeroGradParameters()

    def cuda(self, ctx=None, max_batch=32):
        from . import runtime
        if self.device_net is not None:
            return self
        ctx = ctx or runtime.get_context()
        plist = self.parameter_list()
        total = sum(getattr(m, n).numel() for m, n in plist)
        params, grads = ctx.zeros(total), ctx.zeros(total)
        nets, off = [], 0
        for i, part in enumerate(self._parts()):
            dims = self.input_dims if i < len(self.branches) else (sum(n.out_c * n.out_h * n.out_w for n in nets), 1, 1)
            n_part = sum(getattr(m, n).numel() for m, n in part.parameter_list())
            dn = runtime.DeviceNet(ctx, part.layer_specs(), dims, max_batch, params=params[off:off + n_part],
                                   grads=grads[off:off + n_part])
            o = 0
            for (m, name) in part.parameter_list():
                w = getattr(m, name)
                k = w.numel()
                dn.params[o:o + k] = w.reshape(-1).to(ctx.device)
                setattr(m, name, dn.params[o:o + k].view(w.shape))
                setattr(m, "gradWeight" if name == "weight" else "gradBias", dn.grads[o:o + k].view(w.shape))
                o += k
            dn.train = self.train
            dn.params_changed()
            part.device_net = dn
            nets.append(dn)
            off += n_part
        self.device_net = runtime.CompositeDeviceNet(ctx, nets[:-1], nets[-1], params, grads)
        self.device_net.train = self.train
        return self

    def float(self):
        if self.device_net is None:
            return self
        for (m, name) in self.parameter_list():
            setattr(m, name, getattr(m, name).detach().cpu().clone())
            gname = "gradWeight" if name == "weight" else "gradBias"
            setattr(m, gname, getattr(m, gname).detach().cpu().clone())
        for p in self._parts():
            p.device_net = None
        self.device_net = None
        return self

    def forward_modules(self, x_dev):
        raise FgError("ConcatSequential: module-by-module execution is not built; use the compiled net")

    def __repr__(self):
        lines = ["nn.Sequential {"]
        for i, m in enumerate(self.modules):
            lines.append("  (%d): %s" % (i + 1, repr(m).replace("\n", "\n  ")))
        lines.append("}")
        return "\n".join(lines)


class _CudnnNamespace:
    """`cudnn.*` as the reference scripts spell it (models.lua, models_c2f.lua, layers/cudnnSpatialConvolutionUpsample.lua)."""
    SpatialConvolution = staticmethod(CudnnSpatialConvolution)
    SpatialConvolutionUpsample = SpatialConvolutionUpsample


cudnn = _CudnnNamespace
