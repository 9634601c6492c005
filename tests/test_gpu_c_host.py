"""A torch-free, Python-free C host drives the C ABI the way LuaJIT would: tests/c_host/step_host.c is
compiled against include/facegen_hip.h and run as its own process -- device memory from fg_malloc, transfers through
fg_h2d / fg_d2h, the context on the default stream, the library the ONLY HIP user of the process (it resolves
/opt/rocm's libamdhip64, not the copy a torch wheel bundles).  Reference call pattern: train.lua:71-80, 134-152;
nn_utils.lua:355-362; adversarial.lua:240-288.  Its results are compared with the oracle at the bars of smoke()."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import torch7_nn as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "step_host.c")
SRC_C2F = os.path.join(ROOT, "tests", "c_host", "step_host_c2f.c")
LIBDIR = os.path.join(ROOT, "face_generator_amd")


def compile_host(out_dir, src=SRC):
    exe = os.path.join(out_dir, os.path.splitext(os.path.basename(src))[0])
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    cmd = [cc, "-std=c99", "-Wall", "-Werror", "-O1", src, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lfacegen_hip",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_host_compiles_against_the_public_header(tmp_path):
    """CPU: plain C99 + include/facegen_hip.h + -lfacegen_hip is all a host needs (no hipcc, no torch, no Python)."""
    for src in (SRC, SRC_C2F):
        exe = compile_host(str(tmp_path), src)
        deps = subprocess.check_output(["ldd", exe]).decode()
        assert "libfacegen_hip.so" in deps and "libtorch" not in deps and "libpython" not in deps


@pytest.mark.gpu
@pytest.mark.parametrize("B", [4, 128])
def test_c_host_runs_one_D_and_one_G_closure_without_torch(tmp_path, B):
    """(B = 128: the headline batch -- the wave-specialised kernels, the production split counts and the deferred finals, driven by
    a process that has no torch in it.)
    The flow of __graft_entry__.smoke() twice on identical inputs: once through the Python host inside this (torch) process
    -- the oracle adopts that run's PReLU branch decisions (oracle/device_branches.py) -- and once through the C host in its
    own torch-free process.  The C host's results must meet the smoke bars against the oracle AND equal the in-process run
    bit for bit (the kernels are deterministic: the same library on the same inputs, whoever allocated the memory)."""
    import torch
    from oracle.device_branches import adopt_device_branches
    from face_generator_amd import models, adversarial
    from face_generator_amd.runtime import get_context
    ctx = get_context(0)
    C = 3
    rng = np.random.default_rng(9000)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    st = O.GanState(G, D)
    pG0, pD0 = st.pG.copy(), st.pD.copy()
    real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
    nz_d = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    nz_g = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    mk = lambda: [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
                 [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]
    masks_d, masks_g = mk(), mk()
    # ---- in-process run (Python host), oracle on the device's branches ----
    Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B)
    Dd = models.create_D((C, 32, 32)).cuda(ctx, max_batch=B)
    Gd.getParameters()[0].copy_(torch.tensor(pG0)); Dd.getParameters()[0].copy_(torch.tensor(pD0))
    dnG, dnD = Gd.device_net, Dd.device_net
    dnG.params_changed(); dnD.params_changed()
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100))
    assert tr.gan is not None
    dv = lambda a: torch.tensor(np.ascontiguousarray(a), device=ctx.device)
    inD = tr.step_D(dv(real).permute(0, 2, 3, 1).contiguous(), dv(nz_d), [dv(m.reshape(-1)) for m in masks_d], keep_grad=True)
    inD = dict(prob=inD["outputs"].cpu().numpy().reshape(-1), grad=inD["grad"].cpu().numpy(), loss=inD["loss"].item(),
               params=Dd.getParameters()[0].cpu().numpy())
    adopt_device_branches(ctx, dnD, st.D)
    refD = O.step_D(st, real, nz_d, masks_d)
    adopt_device_branches(ctx, dnD, st.D, clear=True)
    pD_sync = st.pD.copy()
    Dd.getParameters()[0].copy_(torch.tensor(pD_sync)); dnD.params_changed()
    pG_before = dnG.params.clone()
    inG = tr.step_G(dv(nz_g), [dv(m.reshape(-1)) for m in masks_g], keep_grad=True)
    inG = dict(samples=inG["samples"].permute(0, 3, 1, 2).contiguous().cpu().numpy(), prob=inG["outputs"].cpu().numpy().reshape(-1),
               grad=inG["grad"].cpu().numpy(), loss=inG["loss"].item(), params=Gd.getParameters()[0].cpu().numpy())
    adopt_device_branches(ctx, dnD, st.D)
    adopt_device_branches(ctx, dnG, st.G, params=pG_before)
    refG = O.step_G(st, nz_g, masks_g)
    # ---- the C host, its own process ----
    d = str(tmp_path)
    for name, a in [("pG", pG0), ("pD", pD0), ("real", real), ("noise_d", nz_d), ("noise_g", nz_g), ("pD_sync", pD_sync)] + \
                   [("masks_d_%d" % i, m) for i, m in enumerate(masks_d)] + [("masks_g_%d" % i, m) for i, m in enumerate(masks_g)]:
        np.ascontiguousarray(a, np.float32).tofile(os.path.join(d, name + ".bin"))
    exe = compile_host(d)
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "PYTHONPATH")}
    r = subprocess.run([exe, d, str(B)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    print(r.stdout.decode())
    assert r.returncode == 0 and "step_host: OK" in r.stdout.decode(), r.stdout.decode()[-2000:]
    ld = lambda n: np.load(os.path.join(d, n))

    def close(got, ref, atol, what):
        err = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()
        assert err <= atol, "%s: max error %.3e > %.3e" % (what, err, atol)

    def adam_close(p_dev, p_ref, g_dev, g_ref, what, lr=1e-3):
        e = 1e-8 / np.sqrt(1 - 0.999)
        f = lambda g: g / (np.abs(g) + e)
        tol = 2e-6 + 1.01 * lr * np.abs(f(g_dev.astype(np.float64)) - f(g_ref.astype(np.float64)))
        assert (np.abs(p_dev.astype(np.float64) - p_ref) <= tol).all(), what

    # D closure: the library hands back the raw gradient (FG_STEP_NO_UPDATE); penalty + clamp of adversarial.lua:103-123 here
    close(ld("out_D_prob.npy"), refD["out"].reshape(-1), 1e-4, "D-step D outputs")
    assert abs(ld("out_D_loss.npy")[0] - refD["f_bce"]) <= 1e-5 * abs(refD["f_bce"])
    assert (ld("out_D_confusion.npy")[:4].reshape(2, 2) == refD["conf"]).all()
    gD = np.clip(ld("out_D_grad_raw.npy") + np.float32(1e-4) * pD0, -1.0, 1.0)
    close(gD, refD["grad"], 1e-4 * np.abs(refD["grad"]).max() + 1e-7, "D flat gradient")
    adam_close(ld("out_D_params.npy"), pD_sync, gD, refD["grad"], "D parameters after Adam differ")
    # G closure
    close(ld("out_G_samples.npy"), refG["samples"], 1e-4, "G images")
    close(ld("out_G_prob.npy"), refG["out"].reshape(-1), 1e-4, "G-step D outputs")
    assert abs(ld("out_G_loss.npy")[1] - refG["f_bce"]) <= 1e-5 * abs(refG["f_bce"])
    gG = np.clip(ld("out_G_grad_raw.npy"), -5.0, 5.0)
    close(gG, refG["grad"], 1e-4 * np.abs(refG["grad"]).max() + 1e-7, "G flat gradient")
    adam_close(ld("out_G_params.npy"), st.pG, gG, refG["grad"], "G parameters after Adam differ")
    # the two hosts drove the same computation: bit-identical results
    assert np.array_equal(ld("out_D_prob.npy"), inD["prob"]) and ld("out_D_loss.npy")[0] == np.float32(inD["loss"])
    assert np.array_equal(ld("out_D_params.npy"), inD["params"]), "D parameters: C host != Python host"
    assert np.array_equal(ld("out_G_samples.npy"), inG["samples"]) and np.array_equal(ld("out_G_prob.npy"), inG["prob"])
    assert np.array_equal(ld("out_G_params.npy"), inG["params"]), "G parameters: C host != Python host"


@pytest.mark.gpu
def test_c_host_runs_the_coarse_to_fine_closures_without_torch(tmp_path):
    """BASELINE configs 4-5 from a torch-free process (round 4): tests/c_host/step_host_c2f.c makes the calls lua/adversarial_c2f_hip.lua
    makes for one batch of adversarial_c2f.lua:40-187 -- create_G_d / create_D_c as layer specs (models_c2f.lua:113-145, 237-278), the
    step object in TABLE mode, one D closure and one G closure with Adam and train_c2f.lua's penalties.  As above: the oracle's bars,
    and bit for bit what the Python host (TrainerC2F) gets on the same inputs."""
    import torch
    from oracle.device_branches import adopt_device_branches
    from face_generator_amd import adversarial_c2f
    from face_generator_amd.runtime import get_context
    import test_gpu_c2f as C2F
    ctx = get_context(0)
    B, S, C = 4, 16, 3
    h = B // 2
    st, Gd, Dd, rng = C2F.build(ctx, S, B, seed=9100)
    pG0, pD0 = st.pG.copy(), st.pD.copy()
    u = lambda lo, hi, shape: rng.uniform(lo, hi, shape).astype(np.float32)
    diff_r, cond_r, cond_f, nz_d = u(-1, 1, (h, C, S, S)), u(0, 1, (h, C, S, S)), u(0, 1, (h, C, S, S)), u(-1, 1, (h, 1, S, S))
    cond_g, nz_g = u(0, 1, (B, C, S, S)), u(-1, 1, (B, 1, S, S))
    masks_d, masks_g = C2F.masks_for(rng, B, S), C2F.masks_for(rng, B, S)
    d = ctx.device
    nhwc = lambda a: torch.tensor(np.ascontiguousarray(a), device=d).permute(0, 2, 3, 1).contiguous()
    dnG, dnD = Gd.inner.device_net, Dd.inner.device_net
    tr = adversarial_c2f.TrainerC2F(ctx, Gd, Dd, dict(batchSize=B))
    assert tr.gan is not None
    # ---- in-process run (Python host), the oracle on the device's branches ----
    r = tr.step_D(nhwc(diff_r), nhwc(cond_r), nhwc(nz_d), nhwc(cond_f), C2F.dev_masks(masks_d, d), keep_grad=True)
    inD = dict(prob=r["outputs"].cpu().numpy().reshape(-1), loss=r["loss"].item(), grad=r["grad"].cpu().numpy(),
               params=Dd.getParameters()[0].cpu().numpy())
    adopt_device_branches(ctx, dnD, st.D)
    refD = O.step_D_c2f(st, diff_r, cond_r, nz_d, cond_f, masks_d)
    adopt_device_branches(ctx, dnD, st.D, clear=True)
    pD_sync = st.pD.copy()
    Dd.getParameters()[0].copy_(torch.tensor(pD_sync)); dnD.params_changed()
    r = tr.step_G(nhwc(nz_g), nhwc(cond_g), C2F.dev_masks(masks_g, d), keep_grad=True)
    inG = dict(samples=r["samples"].permute(0, 3, 1, 2).contiguous().cpu().numpy(), prob=r["outputs"].cpu().numpy().reshape(-1),
               loss=r["loss"].item(), grad=r["grad"].cpu().numpy(), params=Gd.getParameters()[0].cpu().numpy())
    adopt_device_branches(ctx, dnD, st.D)
    adopt_device_branches(ctx, dnG, st.G)
    refG = O.step_G_c2f(st, nz_g, cond_g, masks_g)
    # ---- the C host, its own process ----
    dd = str(tmp_path)
    dm = lambda ms: [m.cpu().numpy() for m in C2F.dev_masks(ms, torch.device("cpu"))]            # the device's element order
    files = [("pG", pG0), ("pD", pD0), ("diff_real", diff_r), ("cond_real", cond_r), ("cond_fake", cond_f), ("noise_d", nz_d),
             ("cond_g", cond_g), ("noise_g", nz_g), ("pD_sync", pD_sync)] + \
            [("masks_d_%d" % i, m) for i, m in enumerate(dm(masks_d))] + [("masks_g_%d" % i, m) for i, m in enumerate(dm(masks_g))]
    for name, a in files:
        np.ascontiguousarray(a, np.float32).tofile(os.path.join(dd, name + ".bin"))
    exe = compile_host(dd, SRC_C2F)
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "PYTHONPATH")}
    r = subprocess.run([exe, dd, str(B), str(S)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    print(r.stdout.decode())
    assert r.returncode == 0 and "step_host_c2f: OK" in r.stdout.decode(), r.stdout.decode()[-2000:]
    ld = lambda n: np.load(os.path.join(dd, n))

    def close(got, ref, atol, what):
        err = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()
        assert err <= atol, "%s: max error %.3e > %.3e" % (what, err, atol)

    close(ld("out_D_prob.npy"), refD["out"].reshape(-1), 1e-4, "c2f D-step D outputs")
    assert abs(ld("out_D_loss.npy")[0] - refD["f_bce"]) <= 1e-5 * abs(refD["f_bce"])
    assert (ld("out_D_confusion.npy")[:4].reshape(2, 2) == refD["conf"]).all()
    gD = np.clip(ld("out_D_grad_raw.npy") + np.float32(1e-7) * np.sign(pD0), -1.0, 1.0)        # adversarial_c2f.lua:62-67, 82-84
    close(gD, refD["grad"], 1e-4 * np.abs(refD["grad"]).max() + 1e-7, "c2f D flat gradient")
    close(ld("out_G_samples.npy"), refG["samples"], 1e-4 * max(1, np.abs(refG["samples"]).max()), "c2f G diff images")
    close(ld("out_G_prob.npy"), refG["out"].reshape(-1), 1e-4, "c2f G-step D outputs")
    gG = np.clip(ld("out_G_grad_raw.npy"), -5.0, 5.0)
    close(gG, refG["grad"], 1e-4 * np.abs(refG["grad"]).max() + 1e-7, "c2f G flat gradient")
    # the two hosts drove the same computation: bit-identical results
    assert np.array_equal(ld("out_D_prob.npy"), inD["prob"]) and ld("out_D_loss.npy")[0] == np.float32(inD["loss"])
    assert np.array_equal(gD, inD["grad"]), "c2f D gradient: C host != Python host"
    assert np.array_equal(ld("out_D_params.npy"), inD["params"]), "c2f D parameters: C host != Python host"
    assert np.array_equal(ld("out_G_samples.npy"), inG["samples"]) and np.array_equal(ld("out_G_prob.npy"), inG["prob"])
    assert ld("out_G_loss.npy")[1] == np.float32(inG["loss"])
    assert np.array_equal(ld("out_G_params.npy"), inG["params"]), "c2f G parameters: C host != Python host"
