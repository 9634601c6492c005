"""NN_UTILS: utils/nn_utils.lua re-hosted (same function names; globals live in face_generator_amd.state.S)."""
import math
import os

import torch

from . import nn
from .runtime import get_context
from .state import S


def setWeights(weights, range_, gen=None):
    """nn_utils.lua:8-11: weights:randn():mul(range)."""
    weights.copy_(torch.randn(weights.shape, generator=gen) * range_)


def initializeWeights(model, rangeWeights=0.005, rangeBias=0.001, gen=None):
    """nn_utils.lua:17-29: top-level modules only; every .weight (incl. BN gamma, PReLU slope) and .bias."""
    for m in model.modules:
        if getattr(m, "weight", None) is not None:
            setWeights(m.weight, rangeWeights, gen)
        if getattr(m, "bias", None) is not None:
            setWeights(m.bias, rangeBias, gen)


def createNoiseInputs(N):
    """nn_utils.lua:35-39 -> host FloatTensor [N, noiseDim] ~ U(-1,1) (device Philox, copied back)."""
    return S.next_noise(get_context(), N, S.OPT["noiseDim"]).cpu()


def createImagesFromNoise(noiseInputs, outputAsList=False, refineWithG=None):
    """nn_utils.lua:45-69: G forward in chunks of OPT.batchSize (3rd arg accepted and ignored, quirk C2)."""
    N = noiseInputs.shape[0]
    bs = S.OPT["batchSize"]
    images = None
    for i in range(math.ceil(N / bs)):
        gen = S.MODEL_G.forward(noiseInputs[i * bs:min((i + 1) * bs, N)]).clone()
        if images is None:
            images = torch.empty((N,) + tuple(gen.shape[1:]))
        images[i * bs:min((i + 1) * bs, N)] = gen
    return [images[i] for i in range(N)] if outputAsList else images


def createImages(N, outputAsList=False, refineWithG=None):
    """nn_utils.lua:76-78."""
    return createImagesFromNoise(createNoiseInputs(N), outputAsList, refineWithG)


def sortImagesByPrediction(images, ascending=False, nbMaxOut=None):
    """nn_utils.lua:90-118 (forward-only D ranking; evaluate-mode semantics come from the caller)."""
    imgs = torch.stack(list(images)) if isinstance(images, (list, tuple)) else images
    preds = []
    bs = S.OPT["batchSize"]
    for i in range(0, imgs.shape[0], bs):
        preds.append(S.MODEL_D.forward(imgs[i:i + bs]).reshape(-1))
    preds = torch.cat(preds)
    order = torch.argsort(preds, descending=not ascending)
    if nbMaxOut:
        order = order[:nbMaxOut]
    return [imgs[i] for i in order.tolist()], [float(preds[i]) for i in order.tolist()]


def switchToTrainingMode():
    """nn_utils.lua:207-213."""
    S.MODEL_G.training()
    S.MODEL_D.training()


def switchToEvaluationMode():
    """nn_utils.lua:216-222."""
    S.MODEL_G.evaluate()
    S.MODEL_D.evaluate()


def getNumberOfParameters(net):
    """nn_utils.lua:281-290: counts .weight elements only (biases excluded)."""
    inner = net._inner() if isinstance(net, nn.Sequential) else net
    return sum(m.weight.numel() for m in inner.modules if getattr(m, "weight", None) is not None)


def activateCuda(net, max_batch=None):
    """nn_utils.lua:328-363: wrap as Sequential{Copy F->device, net:cuda(), Copy device->F}; idempotent."""
    if isInCudaMode(net):
        return net
    newNet = nn.Sequential()
    newNet.add(nn.Copy("torch.FloatTensor", "hip.NHWC"))
    net.cuda(get_context(), max_batch or S.OPT["batchSize"])
    newNet.add(net)
    newNet.add(nn.Copy("hip.NHWC", "torch.FloatTensor"))
    newNet.train = net.train
    return newNet


def deactivateCuda(net):
    """nn_utils.lua:292-321: unwrap and :float()."""
    if not isInCudaMode(net):
        return net
    inner = net.get(2)
    inner.float()
    return inner


def isInCudaMode(net):
    """nn_utils.lua:397-403 (a global function in the reference, quirk C16)."""
    return isinstance(net, nn.Sequential) and len(net.modules) > 0 and isinstance(net.modules[0], nn.Copy)


def prepareNetworkForSave(net):
    """nn_utils.lua:246-279: drop output/gradInput before serialising."""
    for m in net.listModules():
        m.output = None
        m.gradInput = None


def _bn_modules(inner):
    """Every SpatialBatchNormalization of the net in module order, descending into nested containers (the branches of
    models.lua:279-316 create_D16_d are Sequentials inside a ConcatTable)."""
    return [m for m in inner.listModules() if isinstance(m, nn.SpatialBatchNormalization)]


def state_dict(net):
    inner = net._inner()
    if isinstance(inner, nn.ConcatSequential):     # per-part layer specs: ConcatTable / JoinTable have no fg_layer_spec
        layers = {"branches": [b.layer_specs() for b in inner.branches], "tail": inner.tail.layer_specs()}
    else:
        layers = inner.layer_specs()
    sd = {"layers": layers, "input_dims": inner.input_dims, "params": [], "bn": []}
    for (m, name) in inner.parameter_list():
        sd["params"].append(getattr(m, name).detach().cpu().clone())
    for m in _bn_modules(inner):
        sd["bn"].append((m.running_mean.detach().cpu().clone(), m.running_var.detach().cpu().clone()))
    return sd


def load_state_dict(net, sd):
    """Write a saved state dict back into a net (host modules or device-resident flat vectors alike)."""
    inner = net._inner()
    plist = inner.parameter_list()
    assert len(plist) == len(sd["params"]), "checkpoint does not match the network structure"
    with torch.no_grad():
        for (m, name), w in zip(plist, sd["params"]):
            getattr(m, name).copy_(w.reshape(getattr(m, name).shape))
        for m, (rm, rv) in zip(_bn_modules(inner), sd["bn"]):
            m.running_mean.copy_(rm)
            m.running_var.copy_(rv)
    if inner.device_net is not None:
        inner.device_net.params_changed()
    return net


def load_checkpoint(filename, image_dims=None):
    """train.lua:114-129 `--network`: -> {D, G, opt, epoch} (optimizer state is NOT restored, like the reference).
    Reads both this package's own file (state dicts) and a Torch7-serialised checkpoint written by the reference
    (t7_checkpoint.py; detected by its leading table tag) -- the latter returns nets, not state dicts."""
    with open(filename, "rb") as f:
        head = f.read(4)
    if head == b"\x03\x00\x00\x00":
        from . import t7_checkpoint
        return t7_checkpoint.load_checkpoint(filename, image_dims)
    return torch.load(filename, weights_only=False)


def save_checkpoint(filename=None, fmt="state_dict"):
    """adversarial.lua:319-329: rotate adversarial.net -> .old, save {D, G, opt, epoch}.
    fmt="torch7" writes Torch7's own serialisation (loadable by the reference's torch.load).
    The new file is written next to the target first and renamed into place only after a successful save, so a failed
    save never costs the previous checkpoint.  Data parallelism: replicas are identical, only rank 0 writes."""
    filename = filename or os.path.join(S.OPT.get("save", "logs"), "adversarial.net")
    if S._trainer is not None:
        S._trainer.finish_pending()            # a deferred D update (N > 1) belongs to the saved weights
    if S.dist is not None and S.dist.get_rank() != 0:
        return
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    print("<trainer> saving network to %s" % filename)
    prepareNetworkForSave(S.MODEL_D)
    prepareNetworkForSave(S.MODEL_G)
    tmp = filename + ".tmp"
    if fmt == "torch7":
        from . import t7_checkpoint
        t7_checkpoint.save_checkpoint(tmp, S.MODEL_D, S.MODEL_G, dict(S.OPT), S.EPOCH)
    else:
        torch.save({"D": state_dict(S.MODEL_D), "G": state_dict(S.MODEL_G), "opt": dict(S.OPT), "epoch": S.EPOCH}, tmp)
    if os.path.isfile(filename):
        os.replace(filename, filename + ".old")
    os.replace(tmp, filename)
