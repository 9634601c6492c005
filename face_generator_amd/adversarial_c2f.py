"""ADVERSARIAL (coarse-to-fine): adversarial_c2f.lua re-hosted.  Same structure as adversarial.py with conditional
table inputs: G{noise[B,1,S,S], cond[B,C,S,S]} -> diff image, D{diff, cond}; optim.adam (same formula as
interruptableAdam) without the accuracy gate; G's gradient is MODEL_D.gradInput[1] (adversarial_c2f.lua:99)."""
import time

import torch

from .adversarial import Trainer
from .state import S


class TrainerC2F(Trainer):
    table_inputs = 1

    def __init__(self, ctx, model_G, model_D, opt, dist=None):
        o = dict(D_L1=1e-7, D_L2=0.0, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0)   # train_c2f.lua:27-34
        o.update(opt)
        super().__init__(ctx, model_G, model_D, o, dist)

    def step_D(self, diff_real, cond_real, noise_half, cond_fake, masks=None, keep_grad=False):
        """adversarial_c2f.lua:123-160 + fevalD (:40-80).  All inputs device NHWC; *_real/*_fake have B/2 rows."""
        if self.gan is not None:
            return self._step_D_fused(diff_real, noise_half, masks, keep_grad, None, cond_real=cond_real, cond_fake=cond_fake)
        self.finish_pending()
        half = diff_real.shape[0]
        B = 2 * half
        ctx = self.ctx
        fake = self.dnG.forward(self.G.combine_device(ctx, noise_half, cond_fake), train=True)
        inputs = self._inputs_for(B)
        inputs[:half].copy_(diff_real)
        inputs[half:].copy_(fake)
        cond = torch.cat([cond_real, cond_fake], 0)      # device-memory plumbing (batch assembly)
        targets = self._targets_for(B, "D")
        out = self.dnD.forward(self.D.combine_device(ctx, inputs, cond), masks=masks, train=True)
        loss, dprob, conf = self.crit.forward_backward_device(ctx, out.reshape(-1), targets)
        self.dnD.backward(dprob.view(B, 1), param_grads=True, input_grad=False)
        res = dict(loss=loss, outputs=out, confusion=conf, trained=True)
        pD, gD = self.dnD.params, self.dnD.grads
        if keep_grad:
            g = gD.clone()
            o = self.opt
            if o["D_L1"] != 0 or o["D_L2"] != 0:
                g += torch.sign(pD) * o["D_L1"] + pD * o["D_L2"]
            if o["D_clamp"] != 0:
                g.clamp_(-o["D_clamp"], o["D_clamp"])
            res["grad"] = g
            res["f"] = loss.item() + self.penalty_f("D", pD)
        self._allreduce(gD)
        self._update("D", pD, gD, loss)
        self.dnD.params_changed()
        return res

    def step_G(self, noise, cond, masks=None, keep_grad=False):
        """adversarial_c2f.lua:166-187 + fevalG_on_D (:83-119)."""
        if self.gan is not None:
            return self._step_G_fused(noise, masks, keep_grad, cond=cond, B=cond.shape[0])
        B = noise.shape[0]
        ctx = self.ctx
        samples = self.dnG.forward(self.G.combine_device(ctx, noise, cond), train=True)
        targets = self._targets_for(B, "G")
        out = self.dnD.forward(self.D.combine_device(ctx, samples, cond), masks=masks, train=True)
        loss, dprob, _ = self.crit.forward_backward_device(ctx, out.reshape(-1), targets, want_confusion=False)
        df_do = self.dnD.backward(dprob.view(B, 1), param_grads=False, input_grad=True)   # MODEL_D.gradInput[1]
        self.dnG.backward(df_do, param_grads=True, input_grad=False)
        res = dict(loss=loss, outputs=out, samples=samples)
        pG, gG = self.dnG.params, self.dnG.grads
        if keep_grad:
            g = gG.clone()
            o = self.opt
            if o["G_L1"] != 0 or o["G_L2"] != 0:
                g += torch.sign(pG) * o["G_L2"] + pG * o["G_L2"]
            if o["G_clamp"] != 0:
                g.clamp_(-o["G_clamp"], o["G_clamp"])
            res["grad"] = g
            res["f"] = loss.item() + self.penalty_f("G", pG)
        self._allreduce(gG)
        self._update("G", pG, gG, loss)
        self.dnG.params_changed()
        return res


def train(trainData):
    """adversarial.train(trainData) -- adversarial_c2f.lua:10-223.  trainData[i] has .diff and .coarse (CHW floats)."""
    OPT = S.OPT
    S.EPOCH = S.EPOCH or 1
    N_epoch = OPT["N_epoch"] if OPT["N_epoch"] > 0 else trainData.size()
    dataBatchSize = OPT["batchSize"] // 2
    t0 = time.time()
    if not isinstance(S._trainer, TrainerC2F):
        from .runtime import get_context
        S._trainer = TrainerC2F(get_context(), S.MODEL_G, S.MODEL_D, OPT, dist=S.dist)
    tr = S._trainer
    ctx = tr.ctx
    if tr.world > 1:
        N_epoch = tr.coll.min_int(N_epoch)        # the same number of iterations on every rank (adversarial.train)
    c, h, w = S.IMG_DIMENSIONS
    pending = []
    print("<trainer> Epoch #%d [batchSize = %d]" % (S.EPOCH, OPT["batchSize"]))

    def pick(n, field):
        idx = [S.rng.randrange(trainData.size()) for _ in range(n)]
        return idx, ctx.to_device_nhwc(torch.stack([torch.as_tensor(getattr(trainData[i], field), dtype=torch.float32)
                                                    for i in idx]))
    for t in range(1, N_epoch + 1, dataBatchSize):
        thisBatchSize = min(OPT["batchSize"], N_epoch - t + 1)
        if thisBatchSize < 4:
            print("[INFO] skipping batch at t=%d, because its size is less than 4" % t)
            break
        thisBatchSize -= thisBatchSize % 2
        half = thisBatchSize // 2
        for _ in range(OPT.get("D_iterations", 1)):
            idx = [S.rng.randrange(trainData.size()) for _ in range(half)]
            diff = ctx.to_device_nhwc(torch.stack([torch.as_tensor(trainData[i].diff, dtype=torch.float32) for i in idx]))
            cond_r = ctx.to_device_nhwc(torch.stack([torch.as_tensor(trainData[i].coarse, dtype=torch.float32) for i in idx]))
            _, cond_f = pick(half, "coarse")                                  # new random conds for the fake half (C13)
            nz = S.next_noise(ctx, half, h * w).view(half, h, w, 1)
            pending.append(tr.step_D(diff, cond_r, nz, cond_f)["confusion"].clone())
        for _ in range(OPT.get("G_iterations", 1)):
            _, cond = pick(thisBatchSize, "coarse")
            nz = S.next_noise(ctx, thisBatchSize, h * w).view(thisBatchSize, h, w, 1)
            tr.step_G(nz, cond)
    conf_total = torch.zeros(4, dtype=torch.int64)
    for cf in pending:
        conf_total += cf.cpu().to(torch.int64)
    dt = time.time() - t0
    print("<trainer> time required for this epoch = %d s" % dt)
    print("<trainer> time to learn 1 sample = %f ms" % (1000 * dt / N_epoch))
    cl = conf_total.tolist()
    tV = (cl[0] + cl[3]) / max(1, sum(cl))
    print("Confusion of D: [pred][target] = %s  totalValid = %.4f" % (cl, tV))
    S.CONFUSION = cl
    if S.EPOCH % OPT.get("saveFreq", 30) == 0:          # adversarial_c2f.lua:206-217
        from . import nn_utils
        import os
        nn_utils.save_checkpoint(os.path.join(OPT.get("save", "logs"), "adversarial_c2f_%d_to_%d.net"
                                              % (OPT.get("coarseSize", h // 2), OPT.get("fineSize", h))))
    S.EPOCH += 1
    return tV


best_dist = None


def approxParzen(ds, nsamples, nneighbors):
    """adversarial.approxParzen (adversarial_c2f.lua:305-344): nearest-neighbour distance of the ground truth fine
    image to `nneighbors` generations G(noise, coarse) + coarse; saves the .bestnet checkpoint on improvement."""
    global best_dist
    best_dist = 1e10 if best_dist is None else best_dist
    print("<trainer> evaluating approximate parzen ")
    from .runtime import get_context
    ctx = get_context()
    c, h, w = S.IMG_DIMENSIONS
    dist_dev, min_dev = ctx.empty(nneighbors), ctx.empty(nsamples)
    G = S.MODEL_G
    for n in range(nsamples):
        example = ds[S.rng.randrange(ds.size())]
        cond = ctx.to_device_nhwc(torch.as_tensor(example.coarse, dtype=torch.float32).unsqueeze(0).repeat(nneighbors, 1, 1, 1))
        noise = S.next_noise(ctx, nneighbors, h * w).view(nneighbors, h, w, 1)
        neighbors = G.inner.device_net.forward(G.combine_device(ctx, noise, cond), train=G.inner.train)
        fine = ctx.to_device_nhwc(torch.as_tensor(example.fine, dtype=torch.float32).unsqueeze(0))
        # neighbors:add(condInputs); min_i torch.dist(neighbors[i], fine)  (:322-329) -- one reduction kernel, NHWC throughout
        ctx.check(ctx.lib.fg_parzen_min_dist(ctx.h, neighbors.data_ptr(), cond.data_ptr(), fine.data_ptr(), nneighbors,
                                             c * h * w, dist_dev.data_ptr(), min_dev[n:].data_ptr()))
    distances = min_dev.cpu()
    mean = float(distances.mean())
    print("average || x_%s - G(x_%s) || = %f" % (S.OPT.get("fineSize", h), S.OPT.get("coarseSize", h // 2), mean))
    if mean < best_dist:
        best_dist = mean
        from . import nn_utils
        import os
        fn = os.path.join(S.OPT.get("save", "logs"), "adversarial_c2f_%d_to_%d.bestnet" % (S.OPT.get("coarseSize", h // 2),
                                                                                        S.OPT.get("fineSize", h)))
        nn_utils.save_checkpoint(fn)
    return distances
