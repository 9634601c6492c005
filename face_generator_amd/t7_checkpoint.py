"""Reference checkpoints in Torch7's own serialisation: `{D = net, G = net, opt = OPT, epoch = n}`
written by adversarial.lua:319-329 / adversarial_c2f.lua:206-217 after `NN_UTILS.prepareNetworkForSave`
(nn_utils.lua:246-279: CUDA deactivated, `output` / `gradInput` / `finput` emptied) and read by train.lua:114-129 `--network`
and sample.lua:251-258.  This module maps between those `nn.*` / `cudnn.*` object graphs (torch7_file.T7Object) and the host
module descriptors of face_generator_amd.nn, so a reference-trained net loads into this package and a net trained here can be
written in the layout Torch7's `torch.load` expects.

Field names follow the upstream torch/nn modules of the reference's era (Dec 2015 - early 2016; not under /root/reference,
SURVEY Appendix A).  Version-sensitive points handled on load: BatchNormalization `running_std` (older, = 1/sqrt(var+eps))
vs `running_var`; SpatialConvolution weight stored 4-D or as the 2-D `[nOut][nIn*kH*kW]` matrix of SpatialConvolutionMM;
nets wrapped as `{nn.Copy, net, nn.Copy}` by NN_UTILS.activateCuda (nn_utils.lua:328-363).
PARITY UNPINNED (see torch7_file.py): round-trip and structure tests only.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import nn
from .torch7_file import T7Object, T7Error, LongStorage, lua_array
from . import torch7_file


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy()) if isinstance(t, torch.Tensor) else np.asarray(t)


def _empty():
    return np.zeros((0,), np.float32)


def _base_fields(m):
    f = OrderedDict()
    f["gradInput"] = _empty()
    f["output"] = _empty()
    f["_type"] = "torch.FloatTensor"
    f["train"] = bool(getattr(m, "train", True))
    return f


def _param_fields(f, m, names=("weight", "bias")):
    for n in names:
        f[n] = _np(getattr(m, n)).astype(np.float32)
    for n in names:
        g = "grad" + n[0].upper() + n[1:]
        f[g] = np.zeros_like(f[n])      # prepareNetworkForSave keeps the gradient buffers; their content is not state


# ------------------------------------------------------------------------------------------------ modules -> T7Object
def module_to_t7(m, cudnn_convs=False):
    """One host module descriptor -> the T7Object Torch7 would have serialised for it."""
    f = _base_fields(m)
    t = m._typename
    if isinstance(m, nn.ConcatSequential):      # models.lua:305-312: Sequential{ConcatTable{branches}, JoinTable(2), tail...}
        ct = _base_fields(m.modules[0])
        ct["modules"] = [module_to_t7(b, cudnn_convs) for b in m.branches]
        jt = module_to_t7(m.modules[1], cudnn_convs)
        f["modules"] = [T7Object("nn.ConcatTable", ct), jt] + [module_to_t7(x, cudnn_convs) for x in m.tail.modules]
        return T7Object("nn.Sequential", f)
    if isinstance(m, nn.TableSequential):
        f["modules"] = [module_to_t7(x, cudnn_convs) for x in (m.first, m.inner)]
        return T7Object("nn.Sequential", f)
    if isinstance(m, nn.Sequential):
        f["modules"] = [module_to_t7(x, cudnn_convs) for x in m.modules]
        return T7Object("nn.Sequential", f)
    if isinstance(m, nn.Linear):
        _param_fields(f, m)
    elif isinstance(m, nn.View):
        f["size"] = LongStorage(m.sizes)
        f["numElements"] = int(np.prod(m.sizes))
    elif isinstance(m, nn.PReLU):
        f["nOutputPlane"] = 0
        _param_fields(f, m, ("weight",))
    elif isinstance(m, nn.LeakyReLU):
        f["negval"] = float(m.negval)
    elif isinstance(m, nn.SpatialUpSamplingNearest):
        f["scale_factor"] = 2
        f["inputSize"] = LongStorage((0, 0, 0, 0))
        f["outputSize"] = LongStorage((0, 0, 0, 0))
    elif isinstance(m, nn.SpatialConvolutionUpsample):
        _conv_fields(f, m)
        f["factor"] = int(getattr(m, "factor", 1))
        f["groups"] = 1
        t = "cudnn.SpatialConvolutionUpsample"
    elif isinstance(m, nn.SpatialConvolution):
        _conv_fields(f, m)
        if cudnn_convs or getattr(m, "_typename", "") == "cudnn.SpatialConvolution":
            f["groups"] = 1
            t = "cudnn.SpatialConvolution"
    elif isinstance(m, nn.SpatialBatchNormalization):
        _param_fields(f, m)
        f["running_mean"] = _np(m.running_mean).astype(np.float32)
        f["running_var"] = _np(m.running_var).astype(np.float32)
        f["eps"] = float(m.eps)
        f["momentum"] = float(m.momentum)
        f["affine"] = True
        f["nDim"] = 4
    elif isinstance(m, (nn.SpatialDropout, nn.Dropout)):
        f["p"] = float(m.p)
        f["noise"] = _empty()
        if isinstance(m, nn.Dropout):
            f["v2"] = True
    elif isinstance(m, (nn.SpatialAveragePooling, nn.SpatialMaxPooling)):
        for k in ("kW", "kH", "dW", "dH"):
            f[k] = 2
        f["padW"] = f["padH"] = 0
        f["ceil_mode"] = False
        if isinstance(m, nn.SpatialAveragePooling):
            f["count_include_pad"] = True
            f["divide"] = True
        else:
            f["indices"] = _empty()
    elif isinstance(m, nn.JoinTable):
        f["dimension"] = 2
        f["nInputDims"] = 2
        f["size"] = LongStorage(())
        f["gradInput"] = OrderedDict()
    elif isinstance(m, nn.CAddTable):
        f["gradInput"] = OrderedDict()
    elif isinstance(m, nn.Copy):
        f["intype"], f["outtype"] = "torch.FloatTensor", "torch.FloatTensor"
        f["dontCast"] = False
    elif isinstance(m, nn.Sigmoid):
        pass
    else:
        raise T7Error("module_to_t7: %s is not mapped" % t)
    return T7Object(t, f)


def _conv_fields(f, m):
    f["nInputPlane"], f["nOutputPlane"] = int(m.nInputPlane), int(m.nOutputPlane)
    f["kW"] = f["kH"] = int(m.kW)
    f["dW"] = f["dH"] = int(getattr(m, "dW", 1))
    f["padW"] = f["padH"] = int(m.padW)
    _param_fields(f, m)


# ------------------------------------------------------------------------------------------------ T7Object -> modules
def _num(o, k, default=None):
    v = o.get(k, default)
    return default if v is None else v


def module_from_t7(o):
    """T7Object graph -> host module descriptors (the inverse of module_to_t7, tolerant of era differences)."""
    if not isinstance(o, T7Object):
        raise T7Error("expected a torch object, got %r" % type(o))
    t = o.typename
    f = o.fields

    def setp(m, names):
        for n in names:
            a = np.asarray(f[n], dtype=np.float32)
            dst = getattr(m, n)
            dst.copy_(torch.from_numpy(a.reshape(tuple(dst.shape)).copy()))

    if t in ("nn.Sequential",):
        mods = [module_from_t7(x) for x in lua_array(f["modules"])]
        # NN_UTILS.activateCuda wrapper {Copy, net, Copy} (nn_utils.lua:328-363): the net is module 2
        if len(mods) == 3 and isinstance(mods[0], nn.Copy) and isinstance(mods[2], nn.Copy) and isinstance(mods[1], nn.Sequential):
            return mods[1]
        if mods and isinstance(mods[0], (nn.JoinTable, nn.CAddTable)):     # c2f nets: {JoinTable | CAddTable, inner}
            inner = [x for x in mods[1:] if not isinstance(x, nn.Copy)]
            if len(inner) == 1 and isinstance(inner[0], nn.Sequential):
                return nn.TableSequential(mods[0], inner[0])
        if mods and isinstance(mods[0], _LoadedConcat):                    # create_D16_d layout
            tail = nn.Sequential()
            for x in mods[2:]:
                tail.add(x)
            return nn.ConcatSequential(mods[0].branches, tail)
        s = nn.Sequential()
        for x in mods:
            s.add(x)
        return s
    if t == "nn.ConcatTable":
        return _LoadedConcat([module_from_t7(x) for x in lua_array(f["modules"])])
    if t == "nn.Linear":
        w = np.asarray(f["weight"])
        m = nn.Linear(w.shape[1], w.shape[0])
        setp(m, ("weight", "bias"))
        return m
    if t == "nn.View":
        return nn.View(*[int(v) for v in np.asarray(f["size"]).reshape(-1)])
    if t == "nn.PReLU":
        m = nn.PReLU()
        setp(m, ("weight",))
        return m
    if t == "nn.LeakyReLU":
        return nn.LeakyReLU(float(_num(o, "negval", 0.333)))
    if t == "nn.SpatialUpSamplingNearest":
        return nn.SpatialUpSamplingNearest(int(_num(o, "scale_factor", 2)))
    if t in ("nn.SpatialConvolution", "nn.SpatialConvolutionMM", "cudnn.SpatialConvolution", "cudnn.SpatialConvolutionUpsample"):
        nI, nO, kW = int(f["nInputPlane"]), int(f["nOutputPlane"]), int(f["kW"])
        pad = int(_num(o, "padW", _num(o, "padding", 0)))
        if t == "cudnn.SpatialConvolutionUpsample":
            m = nn.SpatialConvolutionUpsample(nI, nO, kW, int(f["kH"]), int(_num(o, "factor", 1)))
        else:
            cls = nn.cudnn.SpatialConvolution if t == "cudnn.SpatialConvolution" else nn.SpatialConvolution
            m = cls(nI, nO, kW, int(f["kH"]), int(_num(o, "dW", 1)), int(_num(o, "dH", 1)), pad, int(_num(o, "padH", pad)))
        setp(m, ("weight", "bias"))       # 2-D SpatialConvolutionMM weights reshape to [O][I][kH][kW] (same memory order)
        return m
    if t in ("nn.SpatialBatchNormalization", "nn.BatchNormalization", "cudnn.SpatialBatchNormalization"):
        w = np.asarray(f["weight"])
        m = nn.SpatialBatchNormalization(w.shape[0], float(_num(o, "eps", 1e-5)), float(_num(o, "momentum", 0.1)))
        setp(m, ("weight", "bias"))
        m.running_mean.copy_(torch.from_numpy(np.asarray(f["running_mean"], np.float32).copy()))
        if f.get("running_var") is not None:
            m.running_var.copy_(torch.from_numpy(np.asarray(f["running_var"], np.float32).copy()))
        elif f.get("running_std") is not None:        # 2015 pure-Lua version: running_std = 1/sqrt(var + eps)
            rs = np.asarray(f["running_std"], np.float64)
            m.running_var.copy_(torch.from_numpy((1.0 / (rs * rs) - m.eps).astype(np.float32)))
        return m
    if t == "nn.SpatialDropout":
        return nn.SpatialDropout(float(_num(o, "p", 0.5)))
    if t == "nn.Dropout":
        return nn.Dropout(float(_num(o, "p", 0.5)))
    if t == "nn.SpatialAveragePooling":
        return nn.SpatialAveragePooling(int(f["kW"]), int(f["kH"]), int(_num(o, "dW", 2)), int(_num(o, "dH", 2)))
    if t == "nn.SpatialMaxPooling":
        return nn.SpatialMaxPooling(int(f["kW"]), int(f["kH"]), int(_num(o, "dW", 2)), int(_num(o, "dH", 2)))
    if t == "nn.JoinTable":
        return nn.JoinTable(int(_num(o, "dimension", 2)), int(_num(o, "nInputDims", 2)))
    if t == "nn.CAddTable":
        return nn.CAddTable()
    if t == "nn.Copy":
        return nn.Copy(_num(o, "intype", "torch.FloatTensor"), _num(o, "outtype", "torch.FloatTensor"))
    if t == "nn.Sigmoid":
        return nn.Sigmoid()
    raise T7Error("module_from_t7: %s is not on the hot path (not mapped)" % t)


class _LoadedConcat:
    """nn.ConcatTable read from a file, before its parent Sequential turns it into an nn.ConcatSequential"""

    def __init__(self, branches):
        self.branches = branches


def _set_input_dims(net):
    """MODELS.create_* record the per-sample input shape; recover it from the first compute module."""
    inner = net._inner() if hasattr(net, "_inner") else net
    if getattr(inner, "input_dims", None) is not None or isinstance(inner, nn.ConcatSequential):
        return          # (a ConcatSequential takes images: the caller's image_dims)
    for m in inner.modules:
        if isinstance(m, nn.Linear):
            inner.input_dims = (m.weight.shape[1], 1, 1)
            return
        if isinstance(m, nn.SpatialConvolution):
            inner.input_dims = None      # spatial size is not stored in the file: the caller passes it (load_checkpoint)
            return


# ------------------------------------------------------------------------------------------------ checkpoints
def save_checkpoint(filename, D, G, opt, epoch, cudnn_convs_in_G=True):
    """adversarial.lua:319-329 in Torch7's format.  G's convolutions are written as `cudnn.SpatialConvolution` like
    models.lua:64-73 builds them, D's as `nn.SpatialConvolution` (models.lua:385-400)."""
    tab = OrderedDict()
    tab["D"] = module_to_t7(D, False)
    tab["G"] = module_to_t7(G, cudnn_convs_in_G)
    tab["opt"] = OrderedDict((k, v) for k, v in opt.items() if isinstance(v, (int, float, str, bool)) or v is None)
    tab["epoch"] = epoch
    torch7_file.save(filename, tab)


def load_checkpoint(filename, image_dims=None):
    """train.lua:114-129 `--network` / sample.lua:251-258: -> {"D": net, "G": net, "opt": dict, "epoch": n}.
    image_dims = (C, H, W) of the training images (taken from opt.scale / opt.grayscale when present)."""
    tab = torch7_file.load(filename)
    if not isinstance(tab, dict) or "G" not in tab or "D" not in tab:
        raise T7Error("%s is not a {D, G, opt, epoch} checkpoint" % filename)
    out = {"D": module_from_t7(tab["D"]), "G": module_from_t7(tab["G"]), "opt": dict(tab.get("opt") or {}),
           "epoch": tab.get("epoch")}
    if image_dims is None:
        o = out["opt"]
        if o.get("scale") is not None:
            s = int(o["scale"])
            image_dims = (1 if o.get("grayscale") else 3, s, s)
    for k in ("D", "G"):
        _set_input_dims(out[k])
    dn = out["D"]._inner() if hasattr(out["D"], "_inner") else out["D"]
    if getattr(dn, "input_dims", None) is None and image_dims is not None:
        dn.input_dims = tuple(image_dims)
    return out
