"""CPU-only: the C-ABI library loads, exports every symbol include/facegen_hip.h declares, fails loudly without a
GPU; the host-side mirror (models / nn / nn_utils) reproduces the reference's structure."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import torch7_nn as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from face_generator_amd import build, _lib
    build.build(verbose=False)
    return _lib.load_library()


def test_header_symbols_all_exported(lib):
    from face_generator_amd import _lib
    decls = _lib.parse_header()
    names = set(re.findall(r"\b(fg_[A-Za-z0-9_]+)\s*\(", open(_lib.HEADER).read()))
    names -= {"fg_layer_spec", "fg_layer_type"}
    assert names == set(decls), names ^ set(decls)       # the ctypes binding covers the whole header
    assert len(decls) >= 60
    for n in decls:
        assert hasattr(lib, n), n
    assert lib.fg_version().startswith(b"facegen_hip")


def test_default_library_has_no_trace_or_wrong_result_variants(lib):
    """VERDICT r5 item 3 / ADVICE r5: the s_memtime trace kernels, their DBG variants (results are WRONG by design) and the
    FG_DEBUG_NOSTORE switch are compiled only into libfacegen_hip_measure.so (-DFG_MEASURE, build.py --measure).  No environment
    variable can make the default library return garbage: the names are not even in its string table."""
    from face_generator_amd import _lib
    blob = open(_lib.lib_path(), "rb").read()
    hits = sorted(set(m.decode() for m in re.findall(rb"FG_[A-Z0-9_]*(?:DBG|TRACE|NOSTORE)[A-Z0-9_]*", blob)))
    assert hits == [], hits
    for name in (b"wino_trace", b"wino_wgrad_trace", b"igemm_ws_trace", b"trace_calib", b"FG_THIN_TPW", b"FG_THIN_NT"):
        assert name not in blob, name
    # the planning thresholds of the Winograd weight gradient are no longer read from the environment per call
    src = open(os.path.join(ROOT, "face_generator_amd", "csrc", "conv_ops.hip")).read()
    body = src[src.index("static bool choose_wino_wgrad"):]
    body = body[:body.index("\n}\n")]
    assert "getenv" not in body


def test_no_gpu_fails_loudly_not_silently(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = lib.fg_ctx_create(0, ctypes.byref(h))
    assert rc == -2 and b"no HIP device" in lib.fg_last_error(None)
    from face_generator_amd import models, nn_utils, FgError
    from face_generator_amd.state import S
    G = models.create_G((3, 32, 32), 100)
    with pytest.raises(FgError):
        G.forward(torch.zeros(2, 100))                     # no CPU fallback anywhere in the product path
    with pytest.raises(FgError):
        nn_utils.activateCuda(G)
    with pytest.raises(FgError):
        G.getParameters()


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "face_generator_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dp, f)


def test_models_mirror_reference_structure_and_parameter_order():
    from face_generator_amd import models, nn_utils
    for C in (3, 1):
        G = models.create_G((C, 32, 32), 100)
        D = models.create_D((C, 32, 32))
        oG = O.create_G32((C, 32, 32), 100, weight_init_=False)
        oD = O.create_D32b((C, 32, 32))
        for net, onet in ((G, oG), (D, oD)):
            assert [type(m).__name__ for m in net.modules] == [type(m).__name__ for m in onet.modules]
            sizes = [tuple(getattr(m, n).shape) for m, n in net.parameter_list()]
            osizes = [tuple(getattr(m, p).shape) for (m, p, g) in onet.parameters()]
            assert sizes == osizes                        # getParameters() order: module order, weight then bias
    assert sum(getattr(m, n).numel() for m, n in G.parameter_list()) == 2468100
    G3 = models.create_G((3, 32, 32), 100)
    assert sum(getattr(m, n).numel() for m, n in G3.parameter_list()) == 2470406
    assert nn_utils.getNumberOfParameters(D) == sum(
        m.weight.numel() for m in D.modules if getattr(m, "weight", None) is not None)  # biases excluded (nn_utils.lua:281)
    G16 = models.create_G((3, 16, 16), 100)                 # models.lua:27-51, same chain from a 4x4 map
    assert [type(m).__name__ for m in G16.modules] == [type(m).__name__ for m in O.create_G16((3, 16, 16), 100, weight_init_=False).modules]
    assert tuple(G16.modules[0].weight.shape) == (128 * 4 * 4, 100)
    D16 = models.create_D((3, 16, 16))                      # models.lua:279-316: ConcatTable{fine, dense} -> JoinTable -> Linear
    oD16 = O.create_D16_d((3, 16, 16))
    assert [type(m).__name__ for m in D16.modules] == ["ConcatTable", "JoinTable", "Linear", "Sigmoid"]
    assert [tuple(getattr(m, n).shape) for m, n in D16.parameter_list()] == \
           [tuple(getattr(m, p).shape) for (m, p, g) in oD16.parameters()]


def test_initialize_weights_semantics():
    """nn_utils.lua:17-29: every top-level .weight ~ N(0, 0.005^2) (incl. BN gamma and PReLU slope), .bias ~ N(0, 0.001^2)."""
    from face_generator_amd import models, nn_utils
    G = models.create_G((3, 32, 32), 100)
    nn_utils.initializeWeights(G, gen=torch.Generator().manual_seed(3))
    lin = G.modules[0]
    assert abs(lin.weight.std().item() - 0.005) < 2e-4 and abs(lin.bias.std().item() - 0.001) < 1e-4
    bn = G.modules[5]
    assert abs(bn.weight.std().item() - 0.005) < 1.5e-3 and abs(bn.weight.mean().item()) < 2e-3   # gamma no longer U(0,1)
    assert abs(G.modules[2].weight.item()) < 0.05                                                 # PReLU slope re-drawn
    assert isinstance(repr(G), str) and "cudnn.SpatialConvolution(128 -> 256, 5x5" in repr(G)       # print(MODEL_G), train.lua:155


def test_layer_specs_cover_reference_constructors():
    from face_generator_amd import models
    D = models.create_D((3, 32, 32))
    specs = D.layer_specs()
    assert specs[0] == ("CONV", 3, 64, 3, 1, 1.0) and specs[2][0] == "SPATIAL_DROPOUT" and abs(specs[2][5] - 0.2) < 1e-9
    assert specs[16] == ("VIEW", 2048, 0, 0) and specs[17] == ("LINEAR", 2048, 512)
    assert specs[19][0] == "DROPOUT" and specs[19][5] == 0.5
    G = models.create_G((3, 32, 32), 100)
    assert G.layer_specs()[1] == ("VIEW", 128, 8, 8) and G.layer_specs()[5][0] == "BATCHNORM"


def test_bench_flop_accounting_matches_survey():
    import bench
    assert abs(bench.alg_flops_per_iter(128) / 1e9 - 1019.19) < 0.01      # cfg2
    assert abs(bench.alg_flops_per_iter(128) / 128 / 1e9 - 7.962) < 0.001


def test_checkpoint_roundtrip_and_c2f_dataset(tmp_path):
    """adversarial.lua:319-329 / train.lua:114-129: {D, G, opt, epoch} saved with .old rotation and restored."""
    from face_generator_amd import models, nn_utils, dataset_c2f
    from face_generator_amd.state import S
    S.reset()
    S.MODEL_G = models.create_G((3, 32, 32), 100)
    S.MODEL_D = models.create_D((3, 32, 32))
    nn_utils.initializeWeights(S.MODEL_G, gen=torch.Generator().manual_seed(5))
    S.MODEL_G.modules[5].running_mean += 0.25
    S.EPOCH = 7
    fn = str(tmp_path / "logs" / "adversarial.net")
    nn_utils.save_checkpoint(fn)
    nn_utils.save_checkpoint(fn)
    assert os.path.isfile(fn) and os.path.isfile(fn + ".old")          # mv adversarial.net adversarial.net.old
    ck = nn_utils.load_checkpoint(fn)
    assert ck["epoch"] == 7 and ck["opt"]["batchSize"] == S.OPT["batchSize"]
    G2 = models.create_G((3, 32, 32), 100)
    nn_utils.load_state_dict(G2, ck["G"])
    for (m, n), (m2, n2) in zip(S.MODEL_G.parameter_list(), G2.parameter_list()):
        assert torch.equal(getattr(m, n), getattr(m2, n2))
    assert torch.equal(G2.modules[5].running_mean, S.MODEL_G.modules[5].running_mean)
    # dataset_c2f._toResult runs image.scale on the device (fg_c2f_coarse_diff): no CPU fallback (tests/test_gpu_image_scale.py)
    if not torch.cuda.is_available():
        from face_generator_amd import FgError
        with pytest.raises(FgError):
            dataset_c2f.toResult(torch.rand(5, 3, 64, 64), 32, 64)
    r = dataset_c2f.Result(torch.zeros(5, 3, 8, 8), torch.ones(5, 3, 8, 8), torch.zeros(5, 3, 8, 8))
    assert r.size() == 5 and len(r) == 5 and r[2].coarse.shape == (3, 8, 8) and r.getDiff(0, 2).shape[0] == 2
    S.reset()


def test_checkpoint_roundtrip_16px_nets_and_failed_save_keeps_the_old_file(tmp_path, monkeypatch):
    """`--scale 16` (create_D16_d: ConcatTable{conv branch, dense branch} -> JoinTable -> Linear) must survive the
    `EPOCH % saveFreq == 0` hook -- per-part layer specs, every nested parameter -- and a save that fails must not cost the
    previous checkpoint (the new file is renamed into place only after it was written)."""
    from face_generator_amd import models, nn_utils
    from face_generator_amd.state import S
    S.reset()
    S.MODEL_G = models.create_G((3, 16, 16), 100)
    S.MODEL_D = models.create_D((3, 16, 16))
    sd = nn_utils.state_dict(S.MODEL_D)
    assert set(sd["layers"]) == {"branches", "tail"} and len(sd["layers"]["branches"]) == 2
    assert sum(p.numel() for p in sd["params"]) == sum(getattr(m, n).numel() for m, n in S.MODEL_D.parameter_list())
    fn = str(tmp_path / "logs" / "adversarial.net")
    nn_utils.save_checkpoint(fn)
    ck = nn_utils.load_checkpoint(fn)
    D2 = models.create_D((3, 16, 16))
    nn_utils.load_state_dict(D2, ck["D"])
    for (m, n), (m2, n2) in zip(S.MODEL_D.parameter_list(), D2.parameter_list()):
        assert torch.equal(getattr(m, n), getattr(m2, n2))
    before = open(fn, "rb").read()

    def boom(*a, **k):
        raise IOError("disk full")
    monkeypatch.setattr(torch, "save", boom)
    with pytest.raises(IOError):
        nn_utils.save_checkpoint(fn)
    assert open(fn, "rb").read() == before and not os.path.exists(fn + ".old")
    S.reset()


def test_trainer_optstate_comes_from_opt():
    """train.lua:180-191: OPTSTATE.sgd.{D,G} = {learningRate = OPT.*_SGD_lr (default 0.02), momentum = OPT.*_SGD_momentum};
    OPTSTATE.adam.*.learningRate only when --*_adam_lr ~= -1 (the trainer used lr 1e-3 / momentum 0 for SGD)."""
    import inspect
    from face_generator_amd import adversarial
    src = inspect.getsource(adversarial.Trainer.__init__)
    assert '"D_SGD_lr", 0.02' in src and '"G_SGD_momentum", 0' in src and '_adam_lr", -1' in src


def test_weight_init_heuristic_in_create_G():
    """models.lua:48, 78 -> weight-init.lua:41-76: top-level nn.Linear is re-drawn with reset(sqrt(1 / (3 fan_in))) (range
    1 / sqrt(fan_in)), cudnn.SpatialConvolution is NOT reset (its type name is not in the list), every top-level bias is zero
    (convolutions, Linear, BatchNorm beta); create_D32b has no such call; on the c2f nets the call is a no-op (no recursion)."""
    from face_generator_amd import models, models_c2f, nn, weight_init
    resets = []
    orig_lin, orig_conv = nn.Linear.reset, nn.SpatialConvolution.reset
    nn.Linear.reset = lambda self, stdv=None, gen=None: (resets.append((type(self).__name__, stdv)), orig_lin(self, stdv, gen))[1]
    nn.SpatialConvolution.reset = lambda self, stdv=None, gen=None: (resets.append((type(self).__name__, stdv)), orig_conv(self, stdv, gen))[1]
    try:
        for dims, fan in (((3, 32, 32), 100), ((3, 16, 16), 100)):
            resets.clear()
            G = models.create_G(dims, fan, gen=torch.Generator().manual_seed(4))
            assert [r[0] for r in resets] == ["Linear"] and abs(resets[0][1] - (1.0 / (3 * fan)) ** 0.5) < 1e-12
            for m in G.modules:
                if getattr(m, "bias", None) is not None:
                    assert float(m.bias.abs().max()) == 0.0, type(m).__name__
            lin = G.modules[0]
            assert float(lin.weight.abs().max()) <= fan ** -0.5 and abs(lin.weight.std().item() - (1.0 / (3 * fan)) ** 0.5) < 2e-3
            convs = [m for m in G.modules if isinstance(m, nn.SpatialConvolution)]
            assert len(convs) == 3 and all(m._typename == "cudnn.SpatialConvolution" for m in convs)
            assert all(float(m.weight.abs().max()) > 0 for m in convs)
        resets.clear()
        D = models.create_D((3, 32, 32))
        assert not resets and any(float(m.bias.abs().max()) > 0 for m in D.modules if getattr(m, "bias", None) is not None)
        Gc = models_c2f.create_G((3, 16, 16)); Dc = models_c2f.create_D((3, 16, 16))
        assert not resets                                            # top level = {JoinTable | CAddTable, Sequential}: nothing matches
        assert all(float(m.bias.abs().max()) > 0 for m in Gc.inner.modules + Dc.inner.modules if getattr(m, "bias", None) is not None)
        # an nn.SpatialConvolution (not cudnn) at top level IS reset
        net = nn.Sequential().add(nn.SpatialConvolution(8, 8, 3, 3, 1, 1, 1)).add(nn.SpatialBatchNormalization(8))
        weight_init.w_init(net, "kaiming")
        assert [r[0] for r in resets] == ["SpatialConvolution"] and abs(resets[0][1] - (4.0 / (72 + 72)) ** 0.5) < 1e-12
        with pytest.raises(AssertionError):
            weight_init.w_init(net, "nope")
    finally:
        nn.Linear.reset, nn.SpatialConvolution.reset = orig_lin, orig_conv
    # the oracle's restatement has the same two properties and leaves the convolution weights as drawn
    for mk in (O.create_G32, O.create_G16):
        S0 = 32 if mk is O.create_G32 else 16
        a = mk((3, S0, S0), 100, np.random.default_rng(9))
        b = mk((3, S0, S0), 100, np.random.default_rng(9), weight_init_=False)
        for ma, mb in zip(a.modules, b.modules):
            if getattr(ma, "bias", None) is not None:
                assert np.abs(ma.bias).max() == 0 and (type(ma).__name__ == "SpatialBatchNormalization" or np.abs(mb.bias).max() > 0)
            if type(ma).__name__ == "SpatialConvolution":
                assert np.array_equal(ma.weight, mb.weight)
            if type(ma).__name__ == "Linear":
                assert not np.array_equal(ma.weight, mb.weight) and np.abs(ma.weight).max() <= 0.1 + 1e-7
