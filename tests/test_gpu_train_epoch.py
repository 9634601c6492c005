"""`adversarial.train` / `adversarial_c2f.train` / `approxParzen` / `createImages` / `sortImagesByPrediction` executed end to
end on the device and compared with the oracle's restatement of the same loops (adversarial.lua:30-335,
adversarial_c2f.lua:10-223, 305-344; nn_utils.lua:35-118).

How two chaotic trajectories are compared.  The device loop runs once; every `step_D` / `step_G` call is recorded (state
going in, inputs, dropout masks, results).  The oracle then runs ITS OWN loop -- its own stride, tail sizes, < 4 skip,
accuracy gate, D/G iteration counts -- drawing real-image picks from a replay of the same `math.random` stream and the
noise / masks from the recording; right before each of its steps the recorded device state is loaded into the oracle, so
each step is compared from identical state (parity is per step; trajectories diverge chaotically).  If the
two loops disagree in structure (number of steps, batch sizes, pick order, a gate decision) the replay runs out of sync
and the comparison fails.
"""
import os
import random

import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


class ListDataset:
    """The indexable `dataset` argument of adversarial.train: dataset[i], dataset:size()."""

    def __init__(self, items):
        self.items = items

    def size(self):
        return len(self.items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class Recorder:
    """Wraps a Trainer's step_D / step_G: snapshots the state going in and the results coming out."""

    def __init__(self, tr, dnG, dnD, table=False):
        self.tr, self.dnG, self.dnD, self.table = tr, dnG, dnD, table
        self.steps = []
        self._mask_log = []
        self._sd, self._sg = tr.step_D, tr.step_G
        tr.step_D, tr.step_G = self.step_D, self.step_G
        self._draw = dnD.draw_masks
        dnD.draw_masks = self.draw_masks

    def draw_masks(self, batch):
        m = self._draw(batch)
        self._mask_log.append([t.clone() for t in m])
        return m

    def _state(self):
        tr = self.tr
        tr.finish_pending()
        st = dict(pG=self.dnG.params.cpu().numpy().copy(), pD=self.dnD.params.cpu().numpy().copy(),
                  bufG=self.dnG.buffers.cpu().numpy().copy(), adam={})
        for w in ("D", "G"):
            a = tr.optstate["adam"][w]
            st["adam"][w] = dict(t=a.get("t", 0), m=a["m"].cpu().numpy().copy() if "m" in a else None,
                                 v=a["v"].cpu().numpy().copy() if "v" in a else None)
        return st

    @staticmethod
    def _np(a):
        return a.cpu().numpy().copy() if torch.is_tensor(a) else a

    def _finish(self, rec, r, n0):
        """noise / masks of the step: explicit arguments, or what the library drew inside the fused closure."""
        if r.get("masks") is not None:
            rec["masks"] = [m.cpu().numpy().copy() for m in r["masks"]]
        else:
            rec["masks"] = [m.cpu().numpy() for m in self._mask_log[n0]] if len(self._mask_log) > n0 else None
        rec["noise"] = r["noise"].cpu().numpy().copy() if r.get("noise") is not None else None
        rec.update(out=r["outputs"].cpu().numpy().reshape(-1).copy(), loss=float(r["loss"].item()))
        self.steps.append(rec)

    def step_D(self, *args, **kw):
        rec = dict(kind="D", state=self._state(), args=[self._np(a) for a in args])
        n0 = len(self._mask_log)
        r = self._sd(*args, **kw)
        rec.update(conf=r["confusion"].cpu().numpy().reshape(2, 2).copy(), trained=r["trained"])
        self._finish(rec, r, n0)
        return r

    def step_G(self, *args, **kw):
        rec = dict(kind="G", state=self._state(), args=[self._np(a) for a in args])
        n0 = len(self._mask_log)
        r = self._sg(*args, **kw)
        rec["samples"] = nchw(r["samples"]).copy()
        self._finish(rec, r, n0)
        return r


def load_state(st, s, G_bns):
    """Recorded device state -> oracle state (parameters, Adam moments and step counts, BN running statistics)."""
    st.pG[...] = s["pG"]; st.pD[...] = s["pD"]
    off = 0
    for m in G_bns:
        nf = m.running_mean.size
        m.running_mean[...] = s["bufG"][off:off + nf]; m.running_var[...] = s["bufG"][off + nf:off + 2 * nf]
        off += 2 * nf
    for w, ad in (("D", st.adamD), ("G", st.adamG)):
        a = s["adam"][w]
        ad.clear()
        if a["m"] is not None:
            ad.update(t=a["t"], m=a["m"].copy(), v=a["v"].copy(), denom=np.zeros_like(a["m"]))


def test_adversarial_train_epoch_config1_gray_batch16(ctx, tmp_path):
    """BASELINE configs[0]: 32x32 grayscale, noiseDim 100, batch 16 through `adversarial.train` (adversarial.lua:30-335):
    stride B/2, the shrinking tail (16,16,16,16,16,10 then the < 4 skip at N_epoch = 51), the maxAccuracyD gate with a
    short accsInterval, the deferred confusion read, `saveFreq` + the .old rotation, EPOCH += 1 -- two epochs."""
    from face_generator_amd import models, nn_utils, adversarial
    from face_generator_amd.state import S
    B, C, N = 16, 1, 40
    rng = np.random.default_rng(2100)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    st = O.GanState(G, D)
    S.reset()
    adversarial.accs.clear()
    S.OPT.update(batchSize=B, noiseDim=100, N_epoch=51, saveFreq=2, save=str(tmp_path), seed=7, grayscale=True)
    S.IMG_DIMENSIONS = (C, 32, 32)
    S.rng = random.Random(7)
    Gd = models.create_G((C, 32, 32), 100)
    Dd = models.create_D((C, 32, 32))
    S.MODEL_G = nn_utils.activateCuda(Gd)
    S.MODEL_D = nn_utils.activateCuda(Dd)
    pG, _ = S.MODEL_G.getParameters(); pD, _ = S.MODEL_D.getParameters()
    pG.copy_(torch.tensor(st.pG)); pD.copy_(torch.tensor(st.pD))
    dnG, dnD = Gd.device_net, Dd.device_net
    dnG.params_changed(); dnD.params_changed()
    data = ListDataset([rng.uniform(0, 1, (C, 32, 32)).astype(np.float32) for _ in range(N)])
    tr = S.trainer()
    rec = Recorder(tr, dnG, dnD)
    max_acc, interval = 0.6, 3
    replay = random.Random(7)
    G_bns = [m for m in st.G.modules if isinstance(m, O.SpatialBatchNormalization)]
    oracle_accs = []
    for epoch in (1, 2):
        first = len(rec.steps)
        accs_before = list(adversarial.accs)
        tV = adversarial.train(data, max_acc, interval)
        steps = rec.steps[first:]
        assert S.EPOCH == epoch + 1
        cursor = dict(k=0)

        def before_step(kind, k):
            s = steps[k]
            assert s["kind"] == kind, "step %d: device ran a %s-step, the reference loop a %s-step" % (k, s["kind"], kind)
            load_state(st, s["state"], G_bns)
            cursor["k"] = k

        # noise / masks in consumption order
        noise_q = [(s["noise"] if s["noise"] is not None else (s["args"][1] if s["kind"] == "D" else s["args"][0])).reshape(-1, 100)
                   for s in steps]
        mask_q = [s["masks"] for s in steps]
        real_q = [s["args"][0] for s in steps if s["kind"] == "D"]
        qi = dict(n=0, m=0)

        def draw_noise(n):
            z = noise_q[qi["n"]]; qi["n"] += 1
            assert z.shape[0] == n, "noise batch %d drawn by the device loop, %d by the reference loop" % (z.shape[0], n)
            return z

        def draw_masks(b):
            m = mask_q[qi["m"]]; qi["m"] += 1
            out = [mm.reshape(b, -1) for mm in m]
            return out

        log = O.train_epoch(st, data, dict(S.OPT), max_acc, interval, oracle_accs, lambda n: replay.randrange(n),
                            draw_noise, draw_masks, before_step)
        n_steps = sum(len(it["D"]) + len(it["G"]) for it in log["iters"])
        assert n_steps == len(steps), "the reference loop takes %d steps, the device loop took %d" % (n_steps, len(steps))
        assert [it["batch"] for it in log["iters"]] == [16, 16, 16, 16, 16, 10] and log["skipped_at"] == 49
        k = 0
        dsteps = iter(real_q)
        for it in log["iters"]:
            for r in it["D"]:
                s = steps[k]; k += 1
                real = next(dsteps)                                               # device NHWC [B/2, 32, 32, C]
                close(np.transpose(real, (0, 3, 1, 2)), r["inputs"][:real.shape[0]], atol=0, what="real half (pick order)")
                close(s["out"], r["out"].reshape(-1), atol=1e-5, what="epoch %d D-step outputs" % epoch)
                assert abs(s["loss"] - r["f_bce"]) <= 1e-5 * abs(r["f_bce"])
                assert (s["conf"] == r["conf"]).all()
                assert s["trained"] == r["trained"], "gate decision differs at step %d" % k
            for r in it["G"]:
                s = steps[k]; k += 1
                close(s["samples"], r["samples"], atol=1e-5, what="epoch %d G-step samples" % epoch)
                close(s["out"], r["out"].reshape(-1), atol=1e-5, what="epoch %d G-step D outputs" % epoch)
                assert abs(s["loss"] - r["f_bce"]) <= 1e-5 * abs(r["f_bce"])
        assert abs(tV - log["totalValid"]) < 1e-12
        assert S.CONFUSION == [int(log["conf"][0, 0]), int(log["conf"][0, 1]), int(log["conf"][1, 0]), int(log["conf"][1, 1])]
        assert adversarial.accs == oracle_accs and len(oracle_accs) <= interval
        assert 0 < log["not_trained"] + log["trained"] == 6
        # the device's parameters after the epoch == one more oracle step from the last recorded state (already checked
        # per step); the checkpoint hook: EPOCH % saveFreq == 0 -> adversarial.net (+ .old on the second write)
        ck = os.path.join(str(tmp_path), "adversarial.net")
        assert os.path.isfile(ck) == (epoch >= 2)
    ck = nn_utils.load_checkpoint(os.path.join(str(tmp_path), "adversarial.net"))
    assert ck["epoch"] == 2 and ck["opt"]["batchSize"] == B
    flat = np.concatenate([p.numpy().reshape(-1) for p in ck["D"]["params"]])
    close(flat, dnD.params.cpu().numpy(), atol=0, what="checkpointed D parameters")
    # a third epoch with saveFreq = 1 rotates the previous file to .old
    S.OPT["saveFreq"] = 1
    adversarial.train(data, 1.01, 20)
    assert os.path.isfile(os.path.join(str(tmp_path), "adversarial.net.old"))
    assert len(adversarial.accs) <= 20


def _cfg2_epoch_setup(ctx, tmp_path, B, C, seed):
    """A fresh pair of G32 / D32b nets + trainer from one seed (initial parameters, noise stream, dropout-mask keys all derived
    from it: S.set_dist re-keys the nets), so that two set-ups in one process walk the same trajectory."""
    from face_generator_amd import models, nn_utils, adversarial
    from face_generator_amd.state import S
    rng = np.random.default_rng(seed)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    st = O.GanState(G, D)
    S.reset()
    adversarial.accs.clear()
    S.OPT.update(batchSize=B, noiseDim=100, N_epoch=3 * B // 2, saveFreq=100, save=str(tmp_path), seed=7, D_iterations=2, G_iterations=2)
    S.IMG_DIMENSIONS = (C, 32, 32)
    Gd = models.create_G((C, 32, 32), 100)
    Dd = models.create_D((C, 32, 32))
    S.MODEL_G = nn_utils.activateCuda(Gd)
    S.MODEL_D = nn_utils.activateCuda(Dd)
    S.set_dist(None)                                 # rng / noise seed / mask keys <- OPT.seed
    pG, _ = S.MODEL_G.getParameters(); pD, _ = S.MODEL_D.getParameters()
    pG.copy_(torch.tensor(st.pG)); pD.copy_(torch.tensor(st.pD))
    dnG, dnD = Gd.device_net, Dd.device_net
    dnG.params_changed(); dnD.params_changed()
    tr = S.trainer()
    return st, dnG, dnD, tr, Recorder(tr, dnG, dnD)


def test_adversarial_train_epoch_cfg2_batch128_two_iterations_and_a_holding_gate(ctx, tmp_path):
    """BASELINE configs[1] at its own batch through the LOOP: 32x32x3, B = 128, D_iterations = 2,
    G_iterations = 2 (adversarial.lua:240-288 run the closures k times per batch), N_epoch = 192 -> batches of 128, 128 and the
    64-image tail, and the maxAccuracyD gate (adversarial.lua:156-178) HOLDING at least once and training at least once.
    The threshold is found, not guessed: a first pass with the gate open records D's accuracy per D-step; the threshold is put
    between the running maximum and the first accuracy above it, so the same trajectory (same seed -> same parameters, noise
    and masks) trains up to that step and holds there.  Then the real pass is compared step by step with the oracle's loop."""
    from face_generator_amd import adversarial
    from face_generator_amd.state import S
    B, C, N = 128, 3, 200
    data = ListDataset([np.random.default_rng(2300 + i).uniform(0, 1, (C, 32, 32)).astype(np.float32) for i in range(N)])
    st, dnG, dnD, tr, rec = _cfg2_epoch_setup(ctx, tmp_path, B, C, 2200)
    adversarial.train(data, 1.01, 1)
    tv = [float(s["conf"][0, 0] + s["conf"][1, 1]) / float(s["conf"].sum()) for s in rec.steps if s["kind"] == "D"]
    assert len(tv) == 6 and all(s["trained"] for s in rec.steps if s["kind"] == "D")
    hold_at = next((k for k in range(1, len(tv)) if tv[k] > max(tv[:k])), None)
    assert hold_at is not None, "D's accuracy never rose above its first values in the open-gate pass: %s" % tv
    max_acc = 0.5 * (max(tv[:hold_at]) + tv[hold_at])
    first_pass_D = [s["out"].copy() for s in rec.steps if s["kind"] == "D"]

    st, dnG, dnD, tr, rec = _cfg2_epoch_setup(ctx, tmp_path, B, C, 2200)
    interval = 1
    replay = random.Random(7)
    oracle_accs = []
    tV = adversarial.train(data, max_acc, interval)
    steps = rec.steps
    dsteps_dev = [s for s in steps if s["kind"] == "D"]
    for k in range(hold_at + 1):       # the same trajectory up to and including the step that holds
        assert np.array_equal(dsteps_dev[k]["out"], first_pass_D[k]), "the second set-up left the first one's trajectory at D-step %d" % k
    assert [s["trained"] for s in dsteps_dev[:hold_at + 1]] == [True] * hold_at + [False]
    G_bns = [m for m in st.G.modules if isinstance(m, O.SpatialBatchNormalization)]

    def before_step(kind, k):
        s = steps[k]
        assert s["kind"] == kind, "step %d: device ran a %s-step, the reference loop a %s-step" % (k, s["kind"], kind)
        load_state(st, s["state"], G_bns)

    noise_q = [(s["noise"] if s["noise"] is not None else (s["args"][1] if s["kind"] == "D" else s["args"][0])).reshape(-1, 100)
               for s in steps]
    mask_q = [s["masks"] for s in steps]
    real_q = [s["args"][0] for s in steps if s["kind"] == "D"]
    qi = dict(n=0, m=0)

    def draw_noise(n):
        z = noise_q[qi["n"]]; qi["n"] += 1
        assert z.shape[0] == n, "noise batch %d drawn by the device loop, %d by the reference loop" % (z.shape[0], n)
        return z

    def draw_masks(b):
        m = mask_q[qi["m"]]; qi["m"] += 1
        return [mm.reshape(b, -1) for mm in m]

    log = O.train_epoch(st, data, dict(S.OPT), max_acc, interval, oracle_accs, lambda n: replay.randrange(n),
                        draw_noise, draw_masks, before_step)
    assert [it["batch"] for it in log["iters"]] == [128, 128, 64] and all(len(it["D"]) == 2 and len(it["G"]) == 2 for it in log["iters"])
    assert sum(len(it["D"]) + len(it["G"]) for it in log["iters"]) == len(steps) == 12
    assert log["not_trained"] >= 1 and log["trained"] >= 1 and log["trained"] + log["not_trained"] == 6
    k = 0
    dsteps = iter(real_q)
    for it in log["iters"]:
        for r in it["D"]:
            s = steps[k]; k += 1
            real = next(dsteps)
            close(np.transpose(real, (0, 3, 1, 2)), r["inputs"][:real.shape[0]], atol=0, what="real half (pick order)")
            close(s["out"], r["out"].reshape(-1), atol=1e-5, what="D-step %d outputs (B = %d)" % (k, it["batch"]))
            assert abs(s["loss"] - r["f_bce"]) <= 1e-5 * abs(r["f_bce"])
            assert (s["conf"] == r["conf"]).all()
            assert s["trained"] == r["trained"], "gate decision differs at step %d" % k
        for r in it["G"]:
            s = steps[k]; k += 1
            close(s["samples"], r["samples"], atol=1e-5, what="G-step %d samples (B = %d)" % (k, it["batch"]))
            close(s["out"], r["out"].reshape(-1), atol=1e-5, what="G-step %d D outputs" % k)
            assert abs(s["loss"] - r["f_bce"]) <= 1e-5 * abs(r["f_bce"])
    assert abs(tV - log["totalValid"]) < 1e-12
    assert adversarial.accs == oracle_accs and len(oracle_accs) <= interval
    # a held D-step moved nothing: the state recorded before the next step equals the state before the held one
    held = next(i for i, s in enumerate(steps) if s["kind"] == "D" and not s["trained"])
    assert np.array_equal(steps[held]["state"]["pD"], steps[held + 1]["state"]["pD"])
    assert steps[held]["state"]["adam"]["D"]["t"] == steps[held + 1]["state"]["adam"]["D"]["t"]


def test_gate_blocks_training_when_accuracy_is_high(ctx):
    """adversarial.lua:167-178 + interruptable_optimizers.lua:60-66: when the mean accuracy is >= maxAccuracyD, fevalD
    returns false,false and D's parameters, Adam moments and step count do not move."""
    from face_generator_amd import models, adversarial
    B, C = 8, 3
    Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B)
    Dd = models.create_D((C, 32, 32)).cuda(ctx, max_batch=B)
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100))
    real = ctx.uniform((B // 2, 32, 32, C), 0.0, 1.0, seed=5)
    nz = ctx.uniform((B // 2, 100), -1.0, 1.0, seed=6)
    p0 = Dd.device_net.params.clone()
    r = tr.step_D(real, nz, gate=lambda acc: False)
    assert r["trained"] is False and torch.equal(Dd.device_net.params, p0) and "t" not in tr.optstate["adam"]["D"]
    r = tr.step_D(real, nz, gate=lambda acc: True)
    assert r["trained"] is True and not torch.equal(Dd.device_net.params, p0) and tr.optstate["adam"]["D"]["t"] == 1


def c2f_dataset(rng, n, S):
    from face_generator_amd import dataset_c2f
    fine = torch.tensor(rng.uniform(0, 1, (n, 3, S, S)).astype(np.float32))
    return dataset_c2f.toResult(fine, S // 2, S)


class DictView:
    """dataset_c2f Result -> the oracle's dict examples (same tensors)."""

    def __init__(self, res):
        self.res = res

    def __len__(self):
        return self.res.size()

    def __getitem__(self, i):
        e = self.res[i]
        return dict(diff=e.diff.numpy(), coarse=e.coarse.numpy(), fine=e.fine.numpy())


def test_adversarial_c2f_train_epoch_and_parzen(ctx, tmp_path):
    """adversarial_c2f.lua:10-223 (train) and :305-344 (approxParzen) at fineSize 16, batch 8, D_iterations = 2
    (configs[4] trains with D_iterations = 2): per-step parity from the recorded state, the tail batch, saveFreq."""
    from face_generator_amd import models_c2f, adversarial_c2f
    from face_generator_amd.state import S
    Sz, B = 16, 8
    rng = np.random.default_rng(2200)
    G = O.create_G_d((3, Sz, Sz), rng)
    D = O.create_D_c((3, Sz, Sz), rng)
    st = O.GanState(G, D, O.C2F_OPT)
    S.reset()
    S.OPT.update(batchSize=B, N_epoch=14, D_iterations=2, G_iterations=1, saveFreq=1, save=str(tmp_path), seed=3,
                 coarseSize=Sz // 2, fineSize=Sz)
    S.OPT.update(O.C2F_OPT)
    S.IMG_DIMENSIONS = (3, Sz, Sz)
    S.rng = random.Random(3)
    S.MODEL_G = models_c2f.create_G((3, Sz, Sz), cuda=True, max_batch=B)
    S.MODEL_D = models_c2f.create_D((3, Sz, Sz), cuda=True, max_batch=B)
    S.MODEL_G.getParameters()[0].copy_(torch.tensor(st.pG)); S.MODEL_D.getParameters()[0].copy_(torch.tensor(st.pD))
    dnG, dnD = S.MODEL_G.inner.device_net, S.MODEL_D.inner.device_net
    dnG.params_changed(); dnD.params_changed()
    res = c2f_dataset(rng, 12, Sz)
    tr = S._trainer = adversarial_c2f.TrainerC2F(ctx, S.MODEL_G, S.MODEL_D, S.OPT)
    rec = Recorder(tr, dnG, dnD, table=True)
    tV = adversarial_c2f.train(res)
    steps = rec.steps
    assert [s["kind"] for s in steps] == ["D", "D", "G", "D", "D", "G", "D", "D", "G"]      # t = 1, 5, 9 (8, 8, 6); t = 13 -> 2 < 4
    replay = random.Random(3)
    qi = dict(n=0, m=0)
    # recorded args -- D: (diff_real, cond_real, noise_half, cond_fake); G: (noise, cond), all device NHWC
    noise_q = [s["args"][2] if s["kind"] == "D" else s["args"][0] for s in steps]

    def draw_noise(n):
        z = noise_q[qi["n"]]; qi["n"] += 1
        assert z.shape[0] == n
        return np.transpose(z, (0, 3, 1, 2))

    def draw_masks(b):
        m = steps[qi["m"]]["masks"]; qi["m"] += 1
        m4 = m[0].reshape(b, Sz // 4, Sz // 4, 256).transpose(0, 3, 1, 2)     # device NHWC mask order -> NCHW
        return [m4, m[1].reshape(b, 512)]

    def before_step(kind, k):
        assert steps[k]["kind"] == kind
        load_state(st, steps[k]["state"], [])
    log = O.train_epoch_c2f(st, DictView(res), dict(S.OPT), lambda n: replay.randrange(n), draw_noise, draw_masks, before_step)
    assert [it["batch"] for it in log["iters"]] == [8, 8, 6] and log["skipped_at"] == 13
    k = 0
    for it in log["iters"]:
        for r in it["D"]:
            s = steps[k]; k += 1
            close(np.transpose(s["args"][0], (0, 3, 1, 2)), r["inputs"][:s["args"][0].shape[0]], atol=0, what="c2f real diffs (pick order)")
            close(np.transpose(s["args"][3], (0, 3, 1, 2)), r["cond"][s["args"][0].shape[0]:], atol=0, what="c2f fake conds (pick order)")
            close(s["out"], r["out"].reshape(-1), atol=1e-5, what="c2f epoch D-step outputs")
            assert abs(s["loss"] - r["f_bce"]) <= 1e-5 * abs(r["f_bce"]) and (s["conf"] == r["conf"]).all()
        for r in it["G"]:
            s = steps[k]; k += 1
            close(s["samples"], r["samples"], atol=2e-5, what="c2f epoch G-step samples")
            close(s["out"], r["out"].reshape(-1), atol=1e-5, what="c2f epoch G-step D outputs")
    assert abs(tV - log["totalValid"]) < 1e-12 and S.EPOCH == 2
    assert os.path.isfile(os.path.join(str(tmp_path), "adversarial_c2f_%d_to_%d.net" % (Sz // 2, Sz)))

    # approxParzen (adversarial_c2f.lua:305-344): same picks, recorded noise, the device's current G parameters
    st.pG[...] = dnG.params.cpu().numpy()
    pick_state = S.rng.getstate()
    noises = []
    nn_ = S.next_noise

    def rec_noise(c, n, dim):
        z = nn_(c, n, dim)
        noises.append(z.cpu().numpy().reshape(n, 1, Sz, Sz).copy())
        return z
    S.next_noise = rec_noise
    adversarial_c2f.best_dist = None
    dist = adversarial_c2f.approxParzen(res, 3, 5)
    S.next_noise = nn_
    replay = random.Random(); replay.setstate(pick_state)
    it = iter(noises)
    ref = O.approx_parzen(st.G, DictView(res), 3, 5, lambda n: replay.randrange(n), lambda n: next(it))
    close(dist.numpy(), ref, atol=0, rtol=1e-5, what="approxParzen distances")
    assert os.path.isfile(os.path.join(str(tmp_path), "adversarial_c2f_%d_to_%d.bestnet" % (Sz // 2, Sz)))
    assert abs(adversarial_c2f.best_dist - float(dist.mean())) < 1e-9


def test_create_images_and_sort_by_prediction(ctx):
    """nn_utils.lua:35-118 as sample.lua:69-90 uses them: createNoiseInputs, chunked G forward (N not a multiple of
    batchSize), D ranking -- against the oracle on the same noise, in evaluate mode (running statistics) and train mode."""
    from face_generator_amd import models, nn_utils
    from face_generator_amd.state import S
    C, N, bs = 3, 22, 8
    rng = np.random.default_rng(2300)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    st = O.GanState(G, D)
    S.reset()
    S.OPT.update(batchSize=bs, noiseDim=100)
    S.MODEL_G = nn_utils.activateCuda(models.create_G((C, 32, 32), 100), max_batch=bs)
    S.MODEL_D = nn_utils.activateCuda(models.create_D((C, 32, 32)), max_batch=bs)
    S.MODEL_G.getParameters()[0].copy_(torch.tensor(st.pG)); S.MODEL_D.getParameters()[0].copy_(torch.tensor(st.pD))
    S.MODEL_G._inner().device_net.params_changed(); S.MODEL_D._inner().device_net.params_changed()
    noise = nn_utils.createNoiseInputs(N)
    assert tuple(noise.shape) == (N, 100) and noise.device.type == "cpu" and float(noise.min()) >= -1 and float(noise.max()) < 1
    assert abs(float(noise.mean())) < 0.05
    nn_utils.switchToEvaluationMode(); st.G.evaluate(); st.D.evaluate()
    imgs = nn_utils.createImagesFromNoise(noise, False, True)          # 3rd argument accepted and ignored (quirk C2)
    ref = np.concatenate([st.G.forward(noise.numpy()[i:i + bs]) for i in range(0, N, bs)])
    close(imgs.numpy(), ref, atol=2e-5, what="createImagesFromNoise (evaluate mode, chunks of batchSize)")
    as_list = nn_utils.createImagesFromNoise(noise, True)
    assert isinstance(as_list, list) and len(as_list) == N and torch.equal(as_list[5], imgs[5])
    p_ref = np.concatenate([st.D.forward(ref[i:i + bs]) for i in range(0, N, bs)]).reshape(-1)
    srt, preds = nn_utils.sortImagesByPrediction(as_list, ascending=False, nbMaxOut=10)
    order = np.argsort(-p_ref, kind="stable")[:10]
    close(np.array(preds), p_ref[order], atol=2e-5, what="sortImagesByPrediction predictions (descending)")
    gaps = np.abs(np.diff(np.sort(p_ref)))
    if gaps.min() > 1e-4:                                              # ranking is only defined away from ties
        for k, i in enumerate(order):
            assert torch.equal(srt[k], as_list[i])
    srt_a, preds_a = nn_utils.sortImagesByPrediction(imgs, ascending=True)
    assert len(srt_a) == N and all(preds_a[i] <= preds_a[i + 1] for i in range(N - 1))
    # createImages draws its own noise: shape / range / chunking only
    out = nn_utils.createImages(11)
    assert tuple(out.shape) == (11, C, 32, 32) and float(out.min()) > 0 and float(out.max()) < 1
