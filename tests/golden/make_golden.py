"""Generate tests/golden/*.npz from the oracle (oracle/torch7_nn.py) with fixed seeds.

The reference (Lua/Torch7) cannot be executed in this environment and ships no vectors of its own, so these
fixtures freeze the ORACLE's answers (itself pinned against PyTorch-CPU autograd in tests/test_oracle.py): they
guard the oracle against drift and give the GPU parity tests a second, file-based target.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import torch7_nn as O  # noqa: E402


def sample_idx(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False))


def make_cfg(C, B, seed):
    rng = np.random.default_rng(seed)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, O.SpatialBatchNormalization):
                m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
                m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
            if isinstance(m, O.PReLU):
                m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    st = O.GanState(G, D)
    pG0, pD0 = st.pG.copy(), st.pD.copy()
    real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
    nzD = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    nzG = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    masksD = [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
             [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]
    masksG = [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
             [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]
    rd = O.step_D(st, real, nzD, masksD)
    rg = O.step_G(st, nzG, masksG)
    iG, iD = sample_idx(st.pG.size, 4096, seed + 1), sample_idx(st.pD.size, 4096, seed + 2)
    out = dict(C=C, B=B, seed=seed, pG0=pG0, pD0=pD0, real=real, nzD=nzD, nzG=nzG,
               D_out=rd["out"], D_f_bce=np.float64(rd["f_bce"]), D_conf=rd["conf"], D_grad_idx=iD, D_grad_val=rd["grad"][iD],
               D_grad_l2=np.float64(np.sqrt((rd["grad"].astype(np.float64) ** 2).sum())),
               pD1_val=st.pD[iD], G_samples=rg["samples"], G_out=rg["out"], G_f_bce=np.float64(rg["f_bce"]),
               G_grad_idx=iG, G_grad_val=rg["grad"][iG],
               G_grad_l2=np.float64(np.sqrt((rg["grad"].astype(np.float64) ** 2).sum())), pG1_val=st.pG[iG])
    for i, m in enumerate(masksD):
        out["maskD%d" % i] = m
    for i, m in enumerate(masksG):
        out["maskG%d" % i] = m
    return out


def make_c2f(S, B, seed):
    """BASELINE configs 4-5 at a size the oracle finishes in seconds: create_G_d / create_D_c (models_c2f.lua:113-145, 237-278) at
    S x S, one D closure and one G closure of adversarial_c2f.lua with train_c2f.lua's penalties (C2F_OPT)."""
    rng = np.random.default_rng(seed)
    G = O.create_G_d((3, S, S), rng)
    D = O.create_D_c((3, S, S), rng)
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, O.PReLU):
                m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    st = O.GanState(G, D, O.C2F_OPT)
    pG0, pD0 = st.pG.copy(), st.pD.copy()
    h = B // 2
    u = lambda lo, hi, shape: rng.uniform(lo, hi, shape).astype(np.float32)
    diff_r, cond_r, cond_f, nzD = u(-1, 1, (h, 3, S, S)), u(0, 1, (h, 3, S, S)), u(0, 1, (h, 3, S, S)), u(-1, 1, (h, 1, S, S))
    cond_g, nzG = u(0, 1, (B, 3, S, S)), u(-1, 1, (B, 1, S, S))
    mk = lambda: [(rng.random((B, 256, S // 4, S // 4)) < 0.5).astype(np.float32), (rng.random((B, 512)) < 0.5).astype(np.float32)]
    masksD, masksG = mk(), mk()
    rd = O.step_D_c2f(st, diff_r, cond_r, nzD, cond_f, masksD)
    rg = O.step_G_c2f(st, nzG, cond_g, masksG)
    iG, iD = sample_idx(st.pG.size, 4096, seed + 1), sample_idx(st.pD.size, 4096, seed + 2)
    out = dict(S=S, B=B, seed=seed, pG0=pG0, pD0=pD0, diff_r=diff_r, cond_r=cond_r, cond_f=cond_f, nzD=nzD, cond_g=cond_g, nzG=nzG,
               D_out=rd["out"], D_f_bce=np.float64(rd["f_bce"]), D_conf=rd["conf"], D_grad_idx=iD, D_grad_val=rd["grad"][iD],
               D_grad_l2=np.float64(np.sqrt((rd["grad"].astype(np.float64) ** 2).sum())), pD1_val=st.pD[iD],
               G_samples=rg["samples"], G_out=rg["out"], G_f_bce=np.float64(rg["f_bce"]), G_grad_idx=iG, G_grad_val=rg["grad"][iG],
               G_grad_l2=np.float64(np.sqrt((rg["grad"].astype(np.float64) ** 2).sum())), pG1_val=st.pG[iG])
    for i, m in enumerate(masksD):
        out["maskD%d" % i] = m
    for i, m in enumerate(masksG):
        out["maskG%d" % i] = m
    return out


if __name__ == "__main__":
    for (S, B, seed) in [(16, 4, 9105)]:
        d = make_c2f(S, B, seed)
        chk = dict(pG0_sum=np.float64(d["pG0"].astype(np.float64).sum()), pD0_sum=np.float64(d["pD0"].astype(np.float64).sum()))
        del d["pG0"], d["pD0"]
        d.update(chk)
        fn = os.path.join(HERE, "c2f_s%d_b%d.npz" % (S, B))
        np.savez_compressed(fn, **d)
        print(fn, os.path.getsize(fn) // 1024, "KiB")
    if "--c2f-only" in sys.argv:
        sys.exit(0)
    for (C, B, seed) in [(3, 4, 9001), (1, 4, 9002)]:
        d = make_cfg(C, B, seed)
        # parameters are regenerated from the seed by the tests; do not store the 21 MB vectors
        chk = dict(pG0_sum=np.float64(d["pG0"].astype(np.float64).sum()), pD0_sum=np.float64(d["pD0"].astype(np.float64).sum()))
        del d["pG0"], d["pD0"]
        d.update(chk)
        fn = os.path.join(HERE, "gan32_c%d_b%d.npz" % (C, B))
        np.savez_compressed(fn, **d)
        print(fn, os.path.getsize(fn) // 1024, "KiB")
