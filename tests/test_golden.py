"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle with fixed seeds):
CPU: the oracle still reproduces them; GPU: the HIP path matches them through the C ABI."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import torch7_nn as O

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "gan32_*.npz")))


def rebuild(g):
    """Re-create the seeded nets / state exactly as make_golden.py did."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(fn):
    g = np.load(fn)
    mg = rebuild(g)
    d = mg.make_cfg(int(g["C"]), int(g["B"]), int(g["seed"]))
    assert abs(d["pG0"].astype(np.float64).sum() - float(g["pG0_sum"])) < 1e-6
    np.testing.assert_allclose(d["D_out"], g["D_out"], atol=1e-6)
    np.testing.assert_allclose(d["G_samples"], g["G_samples"], atol=1e-6)
    np.testing.assert_allclose(d["D_grad_val"], g["D_grad_val"], atol=1e-6 * np.abs(g["D_grad_val"]).max() + 1e-9)
    np.testing.assert_allclose(d["G_grad_val"], g["G_grad_val"], atol=1e-5 * np.abs(g["G_grad_val"]).max() + 1e-9)
    assert (d["D_conf"] == g["D_conf"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_matches_golden(fn):
    from face_generator_amd import models, adversarial
    from face_generator_amd.runtime import get_context
    g = np.load(fn)
    C, B, seed = int(g["C"]), int(g["B"]), int(g["seed"])
    mg = rebuild(g)
    # initial parameters: regenerate from the seed (the fixture stores their checksum, not 21 MB of weights)
    rng = np.random.default_rng(seed)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False); D = O.create_D32b((C, 32, 32), rng)
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, O.SpatialBatchNormalization):
                m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
                m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
            if isinstance(m, O.PReLU):
                m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    pG0, _ = G.getParameters(); pD0, _ = D.getParameters()
    assert abs(pG0.astype(np.float64).sum() - float(g["pG0_sum"])) < 1e-6
    ctx = get_context(0)
    dev = ctx.device
    Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B)
    Dd = models.create_D((C, 32, 32)).cuda(ctx, max_batch=B)
    Gd.getParameters()[0].copy_(torch.tensor(pG0)); Dd.getParameters()[0].copy_(torch.tensor(pD0))
    Gd.device_net.params_changed(); Dd.device_net.params_changed()
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    real = t(g["real"]).permute(0, 2, 3, 1).contiguous()
    mD = [t(g["maskD%d" % i].reshape(-1)) for i in range(6)]
    mG = [t(g["maskG%d" % i].reshape(-1)) for i in range(6)]
    rd = tr.step_D(real, t(g["nzD"]), mD, keep_grad=True)
    assert np.abs(rd["outputs"].cpu().numpy().reshape(-1) - g["D_out"].reshape(-1)).max() < 1e-5
    assert abs(rd["loss"].item() - float(g["D_f_bce"])) <= 1e-5 * abs(float(g["D_f_bce"]))
    gd = rd["grad"].cpu().numpy()
    assert np.abs(gd[g["D_grad_idx"]] - g["D_grad_val"]).max() <= 1e-4 * np.abs(g["D_grad_val"]).max() + 1e-7
    assert abs(np.sqrt((gd.astype(np.float64) ** 2).sum()) - float(g["D_grad_l2"])) <= 1e-4 * float(g["D_grad_l2"])
    assert (rd["confusion"].cpu().numpy().reshape(2, 2) == g["D_conf"]).all()
    assert np.abs(Dd.getParameters()[0].cpu().numpy()[g["D_grad_idx"]] - g["pD1_val"]).max() < 2e-6
    rg = tr.step_G(t(g["nzG"]), mG, keep_grad=True)
    img = rg["samples"].permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(img - g["G_samples"]).max() < 1e-5                       # bar: 1e-4 (north_star)
    assert np.abs(rg["outputs"].cpu().numpy().reshape(-1) - g["G_out"].reshape(-1)).max() < 1e-5
    gg = rg["grad"].cpu().numpy()
    assert np.abs(gg[g["G_grad_idx"]] - g["G_grad_val"]).max() <= 1e-4 * np.abs(g["G_grad_val"]).max() + 1e-7
    assert abs(np.sqrt((gg.astype(np.float64) ** 2).sum()) - float(g["G_grad_l2"])) <= 1e-4 * float(g["G_grad_l2"])
    assert np.abs(Gd.getParameters()[0].cpu().numpy()[g["G_grad_idx"]] - g["pG1_val"]).max() < 2e-6


C2F_FILES = sorted(glob.glob(os.path.join(HERE, "golden", "c2f_*.npz")))


@pytest.mark.parametrize("fn", C2F_FILES, ids=[os.path.basename(f) for f in C2F_FILES])
def test_oracle_reproduces_golden_c2f(fn):
    """configs 4-5 (round 4): the coarse-to-fine closures of adversarial_c2f.lua at 16x16, B = 4."""
    g = np.load(fn)
    mg = rebuild(g)
    d = mg.make_c2f(int(g["S"]), int(g["B"]), int(g["seed"]))
    assert abs(d["pG0"].astype(np.float64).sum() - float(g["pG0_sum"])) < 1e-6
    np.testing.assert_allclose(d["D_out"], g["D_out"], atol=1e-6)
    np.testing.assert_allclose(d["G_samples"], g["G_samples"], atol=1e-6)
    np.testing.assert_allclose(d["D_grad_val"], g["D_grad_val"], atol=1e-6 * np.abs(g["D_grad_val"]).max() + 1e-9)
    np.testing.assert_allclose(d["G_grad_val"], g["G_grad_val"], atol=1e-5 * np.abs(g["G_grad_val"]).max() + 1e-9)
    assert (d["D_conf"] == g["D_conf"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fn", C2F_FILES, ids=[os.path.basename(f) for f in C2F_FILES])
def test_hip_path_matches_golden_c2f(fn):
    from face_generator_amd import models_c2f, adversarial_c2f
    from face_generator_amd.runtime import get_context
    g = np.load(fn)
    S, B, seed = int(g["S"]), int(g["B"]), int(g["seed"])
    rng = np.random.default_rng(seed)                    # the initial parameters: regenerated from the seed (checksum in the fixture)
    G = O.create_G_d((3, S, S), rng); D = O.create_D_c((3, S, S), rng)
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, O.PReLU):
                m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    st = O.GanState(G, D, O.C2F_OPT)
    assert abs(st.pG.astype(np.float64).sum() - float(g["pG0_sum"])) < 1e-6 and abs(st.pD.astype(np.float64).sum() - float(g["pD0_sum"])) < 1e-6
    ctx = get_context(0)
    dev = ctx.device
    Gd = models_c2f.create_G((3, S, S), cuda=True, max_batch=B)
    Dd = models_c2f.create_D((3, S, S), cuda=True, max_batch=B)
    Gd.getParameters()[0].copy_(torch.tensor(st.pG)); Dd.getParameters()[0].copy_(torch.tensor(st.pD))
    Gd.inner.device_net.params_changed(); Dd.inner.device_net.params_changed()
    tr = adversarial_c2f.TrainerC2F(ctx, Gd, Dd, dict(batchSize=B))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    nhwc = lambda a: t(a).permute(0, 2, 3, 1).contiguous()
    dm = lambda k: [nhwc(g["mask%s0" % k]).reshape(-1), t(g["mask%s1" % k].reshape(-1))]      # the 4-D Dropout mask in device order
    rd = tr.step_D(nhwc(g["diff_r"]), nhwc(g["cond_r"]), nhwc(g["nzD"]), nhwc(g["cond_f"]), dm("D"), keep_grad=True)
    assert np.abs(rd["outputs"].cpu().numpy().reshape(-1) - g["D_out"].reshape(-1)).max() < 1e-5
    assert abs(rd["loss"].item() - float(g["D_f_bce"])) <= 1e-5 * abs(float(g["D_f_bce"]))
    gd = rd["grad"].cpu().numpy()
    assert np.abs(gd[g["D_grad_idx"]] - g["D_grad_val"]).max() <= 1e-4 * np.abs(g["D_grad_val"]).max() + 1e-7
    assert abs(np.sqrt((gd.astype(np.float64) ** 2).sum()) - float(g["D_grad_l2"])) <= 1e-4 * float(g["D_grad_l2"])
    assert (rd["confusion"].cpu().numpy().reshape(2, 2) == g["D_conf"]).all()
    assert np.abs(Dd.getParameters()[0].cpu().numpy()[g["D_grad_idx"]] - g["pD1_val"]).max() < 2e-6
    rg = tr.step_G(nhwc(g["nzG"]), nhwc(g["cond_g"]), dm("G"), keep_grad=True)
    img = rg["samples"].permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(img - g["G_samples"]).max() < 1e-4 * max(1.0, np.abs(g["G_samples"]).max())        # bar: 1e-4 (north_star)
    assert np.abs(rg["outputs"].cpu().numpy().reshape(-1) - g["G_out"].reshape(-1)).max() < 1e-5
    gg = rg["grad"].cpu().numpy()
    assert np.abs(gg[g["G_grad_idx"]] - g["G_grad_val"]).max() <= 1e-4 * np.abs(g["G_grad_val"]).max() + 1e-7
    assert abs(np.sqrt((gg.astype(np.float64) ** 2).sum()) - float(g["G_grad_l2"])) <= 1e-4 * float(g["G_grad_l2"])
    assert np.abs(Gd.getParameters()[0].cpu().numpy()[g["G_grad_idx"]] - g["pG1_val"]).max() < 2e-6
