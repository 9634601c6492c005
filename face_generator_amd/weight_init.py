"""weight-init.lua: `model = require('weight-init')(model, 'heuristic')` (models.lua:48, 78; models_c2f.lua:138, 271).

Walks the TOP-LEVEL modules of `net` only (weight-init.lua:52: `for i = 1, #net.modules`, no recursion): modules whose
`__typename` is `nn.SpatialConvolution` / `nn.SpatialConvolutionMM` / `nn.Linear` are re-drawn with `m:reset(method(fan_in,
fan_out))` (:55-69), then EVERY top-level module that has a bias gets it zeroed (:71-73).  Consequences the mirror keeps:
`cudnn.SpatialConvolution` (G's convolutions) is not in the list, so only its bias is zeroed; BatchNorm's beta is zeroed; PReLU has
no bias; on the c2f nets the top level is {JoinTable | CAddTable, [Copy], Sequential, [Copy]} -- nothing matches (a no-op).
"""
import math


def w_init_heuristic(fan_in, fan_out):
    """weight-init.lua:15-17 ("Efficient backprop", LeCun 1998)."""
    return math.sqrt(1.0 / (3.0 * fan_in))


def w_init_xavier(fan_in, fan_out):
    """weight-init.lua:22-24."""
    return math.sqrt(2.0 / (fan_in + fan_out))


def w_init_xavier_caffe(fan_in, fan_out):
    """weight-init.lua:29-31."""
    return math.sqrt(1.0 / fan_in)


def w_init_kaiming(fan_in, fan_out):
    """weight-init.lua:36-38."""
    return math.sqrt(4.0 / (fan_in + fan_out))


_METHODS = {"heuristic": w_init_heuristic, "xavier": w_init_xavier, "xavier_caffe": w_init_xavier_caffe,
            "kaiming": w_init_kaiming}


def w_init(net, arg, gen=None):
    """weight-init.lua:41-76.  `gen`: an optional torch.Generator for the re-draws (Torch7 uses its global RNG)."""
    assert arg in _METHODS, arg                          # weight-init.lua:48-49: assert(false) on an unknown method
    method = _METHODS[arg]
    for m in getattr(net, "modules", []):
        t = getattr(m, "_typename", "")
        if t in ("nn.SpatialConvolution", "nn.SpatialConvolutionMM"):
            m.reset(method(m.nInputPlane * m.kW * m.kW, m.nOutputPlane * m.kW * m.kW), gen=gen)
        elif t == "nn.Linear":
            m.reset(method(m.weight.shape[1], m.weight.shape[0]), gen=gen)
        if getattr(m, "bias", None) is not None:
            m.bias.zero_()
    return net
