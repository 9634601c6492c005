"""interruptable_optimizers.lua on the flat device vectors: same names, same (opfunc, x, config, state) contract,
same skip-on-false gate (interruptable_optimizers.lua:60-66) -- the tensor math is one fused HIP kernel.

opfunc(x) -> (f, dfdx) with dfdx a device tensor, or (False, False) to skip the update (no `t` increment).
`fused` (extension used by adversarial.py) folds the caller's penalty/clamp/all-reduce scale into the same pass:
  dict(gscale=1.0, l1_mul=0.0, l2=0.0, clamp=0.0).
"""
import torch

from .runtime import get_context

_NOFUSE = dict(gscale=1.0, l1_mul=0.0, l2=0.0, clamp=0.0)


def _ctx_of(x):
    return get_context(x.device.index)


def interruptableAdam(opfunc, x, config=None, state=None, fused=None):
    """interruptable_optimizers.lua:49-94 (Torch7 Adam: eps added before bias correction)."""
    config = {} if config is None else config
    state = config if state is None else state
    lr = config.get("learningRate", 0.001)
    beta1 = config.get("beta1", 0.9)
    beta2 = config.get("beta2", 0.999)
    epsilon = config.get("epsilon", 1e-8)
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    f = dict(_NOFUSE, **(fused or {}))
    state["t"] = state.get("t", 0)
    if "m" not in state:
        state["m"] = torch.zeros_like(dfdx)
        state["v"] = torch.zeros_like(dfdx)
    state["t"] += 1
    ctx = _ctx_of(x)
    ctx.check(ctx.lib.fg_adam_fused(ctx.h, x.data_ptr(), dfdx.data_ptr(), state["m"].data_ptr(), state["v"].data_ptr(),
                                    x.numel(), f["gscale"], f["l1_mul"], f["l2"], f["clamp"], lr, beta1, beta2, epsilon,
                                    state["t"], None))
    return x, [fx]


def interruptableSgd(opfunc, x, config=None, state=None, fused=None):
    """interruptable_optimizers.lua:97-167."""
    config = {} if config is None else config
    state = config if state is None else state
    lr = config.get("learningRate", 1e-3)
    lrd = config.get("learningRateDecay", 0)
    wd = config.get("weightDecay", 0)
    mom = config.get("momentum", 0)
    damp = config.get("dampening", mom)
    nesterov = config.get("nesterov", False)
    state["evalCounter"] = state.get("evalCounter", 0)
    nevals = state["evalCounter"]
    if nesterov and (mom <= 0 or damp != 0):
        raise ValueError("Nesterov momentum requires a momentum and zero dampening")
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    f = dict(_NOFUSE, **(fused or {}))
    first = 0
    if mom != 0 and "dfdx" not in state:
        state["dfdx"] = torch.zeros_like(dfdx)
        first = 1
    clr = lr / (1 + nevals * lrd)
    ctx = _ctx_of(x)
    ctx.check(ctx.lib.fg_sgd_fused(ctx.h, x.data_ptr(), dfdx.data_ptr(),
                                   state["dfdx"].data_ptr() if mom != 0 else None, x.numel(), f["gscale"], f["l1_mul"],
                                   f["l2"], f["clamp"], clr, mom, damp, wd, int(bool(nesterov)), first))
    state["evalCounter"] += 1
    return x, [fx]


def interruptableAdagrad(opfunc, x, config=None, state=None, fused=None):
    """interruptable_optimizers.lua:7-46."""
    config = {} if config is None else config
    state = config if state is None else state
    lr = config.get("learningRate", 1e-3)
    lrd = config.get("learningRateDecay", 0)
    state["evalCounter"] = state.get("evalCounter", 0)
    nevals = state["evalCounter"]
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    f = dict(_NOFUSE, **(fused or {}))
    clr = lr / (1 + nevals * lrd)
    if "paramVariance" not in state:
        state["paramVariance"] = torch.zeros_like(dfdx)
    ctx = _ctx_of(x)
    ctx.check(ctx.lib.fg_adagrad_fused(ctx.h, x.data_ptr(), dfdx.data_ptr(), state["paramVariance"].data_ptr(),
                                       x.numel(), f["gscale"], f["l1_mul"], f["l2"], f["clamp"], clr))
    state["evalCounter"] += 1
    return x, [fx]
