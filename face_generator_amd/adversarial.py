"""ADVERSARIAL: the GAN training loop of adversarial.lua re-hosted on the device plan.

`train(dataset, maxAccuracyD, accsInterval)` keeps adversarial.lua:30's signature and loop structure (stride B/2,
thisBatchSize, skip < 4, D_iterations x D-step then G_iterations x G-step, confusion print, EPOCH += 1); the closure
bodies fevalD / fevalG_on_D (adversarial.lua:83-231) run as `Trainer.step_D` / `Trainer.step_G`: G->D hand-off,
BCE, backward, penalty + clamp + Adam all stay in HBM (the reference round-trips through host memory at every
nn.Copy).  Deliberate deviation C6: the G-step does not compute D's weight gradients (the reference
computes then discards them; GRAD_PARAMETERS_D is zeroed at :92 before its next use).

Multi-GPU: one process per GPU; the flat gradient vector of the net being updated is all-reduced (sum) with
torch.distributed (backend "nccl" == RCCL over xGMI) and scaled by 1/world inside the fused Adam pass; the clamp
is applied after the reduce (adversarial.lua:121-123 is non-linear).
"""
import time

import torch

from . import interruptable_optimizers as IO
from .nn import BCECriterion
from .state import S


class Trainer:
    def __init__(self, ctx, model_G, model_D, opt, dist=None):
        self.ctx = ctx
        self.G, self.D = model_G, model_D
        self.dnG, self.dnD = model_G._inner().device_net, model_D._inner().device_net
        self.opt = dict(D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0, D_optmethod="adam",
                        G_optmethod="adam")
        self.opt.update(opt)
        self.crit = BCECriterion()
        # OPTSTATE of train.lua:180-191: SGD takes its learning rate / momentum from --{D,G}_SGD_lr / _SGD_momentum
        # (defaults 0.02 / 0, train.lua:21-24), Adam an explicit learning rate only when --{D,G}_adam_lr ~= -1
        o = self.opt
        self.optstate = dict(adagrad=dict(D={}, G={}), adam=dict(D={}, G={}), rmsprop=dict(D={}, G={}),
                             sgd=dict(D=dict(learningRate=o.get("D_SGD_lr", 0.02), momentum=o.get("D_SGD_momentum", 0)),
                                      G=dict(learningRate=o.get("G_SGD_lr", 0.02), momentum=o.get("G_SGD_momentum", 0))))
        for w in ("D", "G"):
            if o.get(w + "_adam_lr", -1) != -1:
                self.optstate["adam"][w]["learningRate"] = o[w + "_adam_lr"]
        # the exchange carrier: None, a distributed.Collective (fg_comm_* of the library, or torch.distributed), or the
        # initialised torch.distributed module itself (wrapped)
        from .distributed import as_collective
        self.dist = self.coll = as_collective(dist)
        self.world = self.coll.get_world_size() if self.coll is not None else 1
        self.gscale = 1.0 / self.world   # BCE averages over the local batch: global mean = all-reduced sum / world
        self._targets = {}
        self._inputs = {}
        self._pending_D = None    # loss of a D update deferred behind the next G forward (its all-reduce is in flight)
        self.overlap = True       # N > 1: hide D's gradient all-reduce under the G-step's generator forward
        self.gan = None           # runtime.FusedGan: one C call per closure (fg_step_D / fg_step_G)
        self._make_fused()
        if self.gan is None and self.coll is not None and self.world > 1 and self.opt.get("sync_bn", False):
            # host-driven closures: exact B_global BatchNorm statistics, fp64 per-channel sums all-reduced at every
            # BatchNorm through a buffer owned by the DeviceNet.  (With the step object the library owns that buffer:
            # FusedGan.set_comm points both DeviceNets at it.)
            self.dnG.enable_sync_bn(self.coll.allreduce_sum_)

    table_inputs = 0              # 1: the {noise, cond} / {x, cond} table nets of adversarial_c2f.lua

    def _make_fused(self):
        """The step-level C entries (fg_step_D / fg_step_G) carry the closure whenever both nets are single compiled plans
        and the exchange carrier is the library's own communicator (or there is none).  The module-by-module host path
        below stays for ConcatTable nets (create_D16_d) and for torch.distributed carriers (gloo in the CPU / one-GPU tests)."""
        from .runtime import DeviceNet, FusedGan
        from .distributed import FgCollective
        if not (isinstance(self.dnG, DeviceNet) and isinstance(self.dnD, DeviceNet)):
            return
        if self.coll is not None and not isinstance(self.coll, FgCollective):
            return
        if self.dnG.n_masks:
            return
        B = max(int(self.opt.get("batchSize", 0) or 0), self.dnG.max_batch, self.dnD.max_batch)
        self.gan = FusedGan(self.ctx, self.dnG, self.dnD, self.table_inputs, B)
        o = self.opt
        self.gan.set_penalty(0, o["D_L1"], o["D_L2"], o["D_clamp"])
        self.gan.set_penalty(1, o["G_L1"], o["G_L2"], o["G_clamp"])
        from .state import S
        self.gan.set_seeds(S.noise_seed, S.noise_offset, self.dnD.mask_seed, self.dnD.mask_offset)
        if self.coll is not None:
            self.gan.set_comm(self.coll, sync_bn=bool(o.get("sync_bn", False)), overlap=1)

    def _sync_fused_config(self, which):
        """OPT / OPTSTATE are plain Lua tables the caller may edit between steps (train.lua:180-191): push them down."""
        w = 0 if which == "D" else 1
        o = self.opt
        self.gan.set_penalty(w, o[which + "_L1"], o[which + "_L2"], o[which + "_clamp"])
        method = o[which + "_optmethod"]
        self.gan.set_optimizer(w, method, self.optstate[method][which])

    def _publish_optstate(self, which):
        """Mirror the optimizer state held by the step object into OPTSTATE.<method>.<net> (t / m / v ...)."""
        w = 0 if which == "D" else 1
        method = self.opt[which + "_optmethod"]
        st = self.optstate[method][which]
        n = (self.dnD if w == 0 else self.dnG).n_params
        buf = self.gan.view("OPT_STATE_D" if w == 0 else "OPT_STATE_G")
        steps = self.gan.steps(w)
        if method == "adam":
            if steps > 0:
                st.update(t=steps, m=buf[:n], v=buf[n:2 * n])
        elif method == "sgd":
            st["evalCounter"] = steps
            if st.get("momentum", 0) != 0 and steps > 0:
                st["dfdx"] = buf[:n]
        else:
            st["evalCounter"] = steps
            if steps > 0:
                st["paramVariance"] = buf[:n]

    # -- helpers -----------------------------------------------------------------------------
    def _targets_for(self, B, kind):
        key = (B, kind)
        if key not in self._targets:
            t = self.ctx.zeros(B)
            if kind == "D":
                t[:B // 2] = 1.0      # Y_NOT_GENERATOR = 1 for the real half (adversarial.lua:247), fakes 0 (:255)
            else:
                t[:] = 1.0            # targets:fill(Y_NOT_GENERATOR) (adversarial.lua:277)
            self._targets[key] = t
        return self._targets[key]

    def _inputs_for(self, B):
        if B not in self._inputs:
            dn = self.dnD
            self._inputs[B] = self.ctx.empty(B, dn.in_h, dn.in_w, dn.in_c)
        return self._inputs[B]

    def _allreduce(self, grads):
        if self.world > 1:
            self.coll.allreduce_sum_(grads)

    def _update(self, which, params, grads, f):
        o = self.opt
        l1, l2, clamp = o[which + "_L1"], o[which + "_L2"], o[which + "_clamp"]
        if l1 == 0 and l2 == 0:
            l1_mul = 0.0
        else:
            # adversarial.lua:109 (D: sign*D_L1) vs :223 (G: sign*G_L2 -- quirk C4, preserved)
            l1_mul = l1 if which == "D" else l2
        fused = dict(gscale=self.gscale, l1_mul=l1_mul, l2=l2 if (l1 != 0 or l2 != 0) else 0.0, clamp=clamp)
        method = o[which + "_optmethod"]
        fn = dict(adam=IO.interruptableAdam, sgd=IO.interruptableSgd, adagrad=IO.interruptableAdagrad)[method]
        fn(lambda x: (f, grads), params, self.optstate[method][which], fused=fused)

    def penalty_f(self, which, params):
        """f += L1*|p|_1 + L2*|p|_2^2/2 (adversarial.lua:105-106): only evaluated on request (host sync)."""
        o = self.opt
        l1, l2 = o[which + "_L1"], o[which + "_L2"]
        if l1 == 0 and l2 == 0:
            return 0.0
        out, scr = self.ctx.empty(2), self.ctx.empty(1024)
        self.ctx.check(self.ctx.lib.fg_norms(self.ctx.h, params.data_ptr(), params.numel(), out.data_ptr(),
                                             scr.data_ptr()))
        n1, n2 = out.tolist()
        return l1 * n1 + l2 * n2 / 2

    # -- the two closures ----------------------------------------------------------------------
    def step_D(self, real_nhwc, noise_half, masks=None, keep_grad=False, gate=None):
        """adversarial.lua:240-268 + fevalD (:83-179).  real_nhwc: device [B/2,H,W,C]; noise_half: [B/2,noiseDim].
        gate(accuracy)->bool reproduces the maxAccuracyD interrupt (host sync only when a gate is given).
        noise_half = None: the library draws the noise (and the dropout masks, when masks is None) in one launch."""
        if self.gan is not None:
            return self._step_D_fused(real_nhwc, noise_half, masks, keep_grad, gate)
        self.finish_pending()
        half = real_nhwc.shape[0]
        B = 2 * half
        fake = self.dnG.forward(noise_half, train=True)          # C5: G in TRAIN mode, BN batch stats over B/2
        inputs = self._inputs_for(B)
        inputs[:half].copy_(real_nhwc)
        inputs[half:].copy_(fake)
        targets = self._targets_for(B, "D")
        out = self.dnD.forward(inputs, masks=masks, train=True)
        loss, dprob, conf = self.crit.forward_backward_device(self.ctx, out.reshape(-1), targets)
        self.dnD.backward(dprob.view(B, 1), param_grads=True, input_grad=False)
        res = dict(loss=loss, outputs=out, confusion=conf)
        pD, gD = self.dnD.params, self.dnD.grads
        if keep_grad:   # parity/debug: the penalised + clamped gradient fevalD returns, and f with the penalty
            g = gD.clone() / 1.0
            o = self.opt
            if o["D_L1"] != 0 or o["D_L2"] != 0:
                g += torch.sign(pD) * o["D_L1"] + pD * o["D_L2"]
            if o["D_clamp"] != 0:
                g.clamp_(-o["D_clamp"], o["D_clamp"])
            res["grad"] = g
            res["f"] = loss.item() + self.penalty_f("D", pD)
        do_train = True
        if gate is not None:
            if self.world > 1:
                # every rank must take the same branch (a skipped update skips its all-reduce too): the gate is decided
                # on the GLOBAL batch's confusion counts, so all replicas keep one identical `accs` history
                conf = self.coll.allreduce_sum_(conf.clone())
            c = conf.tolist()
            acc = (c[0] + c[3]) / max(1, sum(c))     # [pred0,t0] + [pred1,t1]
            do_train = gate(acc)
            if self.world > 1:
                do_train = self.coll.agree(do_train)      # one vote: a rank that disagrees cannot leave the others in an all-reduce
        if do_train:
            if self.world > 1 and self.overlap:
                # xGMI all-reduce of the 11.45 MB D gradient runs on RCCL's stream while the G-step's generator
                # forward (no dependency on D's parameters) computes; the update lands before D is next used.
                self.coll.allreduce_sum_async_(gD)
                self._pending_D = (loss,)
            else:
                self._allreduce(gD)
                self._update("D", pD, gD, loss)
                self.dnD.params_changed()
        res["trained"] = do_train
        return res

    def _grad_view(self, which, p, g):
        """feval's return value: the penalised + clamped gradient (adversarial.lua:103-123 / 218-228) -- parity / debug."""
        o = self.opt
        g = g.clone()
        if o[which + "_L1"] != 0 or o[which + "_L2"] != 0:
            g += torch.sign(p) * (o["D_L1"] if which == "D" else o["G_L2"]) + p * o[which + "_L2"]     # quirk C4 for G
        if o[which + "_clamp"] != 0:
            g.clamp_(-o[which + "_clamp"], o[which + "_clamp"])
        return g

    def _step_D_fused(self, real_nhwc, noise_half, masks, keep_grad, gate, cond_real=None, cond_fake=None):
        gan = self.gan
        half = real_nhwc.shape[0]
        B = 2 * half
        self._sync_fused_config("D")
        hold = keep_grad or gate is not None
        gan.step_D(B, real_nhwc, cond_real, cond_fake, noise_half, masks, gan.NO_UPDATE if hold else 0)
        conf = gan.view("CONFUSION").view(torch.int32)
        res = dict(loss=gan.view("LOSS")[0:1], outputs=gan.view("D_OUTPUT", B).view(B, 1), confusion=conf[:4])
        if noise_half is None:
            res["noise"] = gan.view("NOISE", half * (self.dnG.in_c if not self.table_inputs else self.dnG.in_h * self.dnG.in_w))
        if masks is None and gan.n_masks:
            res["masks"] = [gan.mask_view(i, B) for i in range(gan.n_masks)]
        pD, gD = self.dnD.params, self.dnD.grads
        if keep_grad:
            res["grad"] = self._grad_view("D", pD, gD)
            res["f"] = res["loss"].item() + self.penalty_f("D", pD)
        do_train = True
        if gate is not None:
            c = conf[4:8].tolist()                    # counts of the GLOBAL batch: every rank takes the same branch
            do_train = gate((c[0] + c[3]) / max(1, sum(c)))
            if self.world > 1:
                do_train = self.coll.agree(do_train)      # (see step_D)
        if hold and do_train:
            gan.update(0)
        res["trained"] = do_train
        self._publish_optstate("D")
        return res

    def _step_G_fused(self, noise, masks, keep_grad, cond=None, B=None):
        gan = self.gan
        B = noise.shape[0] if B is None else B
        self._sync_fused_config("G")
        gan.step_G(B, cond, noise, masks, gan.NO_UPDATE if keep_grad else 0)
        dn = self.dnD
        res = dict(loss=gan.view("LOSS")[1:2], outputs=gan.view("D_OUTPUT", B).view(B, 1),
                   samples=gan.view("D_INPUT", B * dn.in_c * dn.in_h * dn.in_w).view(B, dn.in_h, dn.in_w, dn.in_c))
        if noise is None:
            res["noise"] = gan.view("NOISE", B * (self.dnG.in_c if not self.table_inputs else self.dnG.in_h * self.dnG.in_w))
        if masks is None and gan.n_masks:
            res["masks"] = [gan.mask_view(i, B) for i in range(gan.n_masks)]
        if keep_grad:
            pG, gG = self.dnG.params, self.dnG.grads
            res["grad"] = self._grad_view("G", pG, gG)
            res["f"] = res["loss"].item() + self.penalty_f("G", pG)
            gan.update(1)
        self._publish_optstate("G")
        self._publish_optstate("D")        # a deferred D update lands inside the G-step
        return res

    def finish_pending(self):
        """Complete a deferred D update (wait for its all-reduce, fused Adam, re-pack)."""
        if self.gan is not None:
            self.gan.finish_pending()
            return
        if self._pending_D is not None:
            (loss,) = self._pending_D
            self._pending_D = None
            self.coll.wait()
            self._update("D", self.dnD.params, self.dnD.grads, loss)
            self.dnD.params_changed()

    def step_G(self, noise, masks=None, keep_grad=False):
        """adversarial.lua:275-288 + fevalG_on_D (:187-231).  `noise` may be an int B: the library draws the noise."""
        if self.gan is not None:
            if isinstance(noise, int):
                return self._step_G_fused(None, masks, keep_grad, B=noise)
            return self._step_G_fused(noise, masks, keep_grad)
        B = noise.shape[0]
        samples = self.dnG.forward(noise, train=True)
        self.finish_pending()                                     # D's update must land before D is evaluated
        targets = self._targets_for(B, "G")
        out = self.dnD.forward(samples, masks=masks, train=True)
        loss, dprob, _ = self.crit.forward_backward_device(self.ctx, out.reshape(-1), targets, want_confusion=False)
        df_do = self.dnD.backward(dprob.view(B, 1), param_grads=False, input_grad=True)   # MODEL_D.modules[1].gradInput
        pG, gG = self.dnG.params, self.dnG.grads
        bucketed = False
        if self.world > 1 and self.overlap and not keep_grad:
            # bucketed all-reduce overlapped with backward: G's gradients are produced output -> input; each finished
            # ~1 M-parameter range of the flat vector goes onto RCCL's stream while the earlier layers still compute
            for (s_from, s_to, lo, hi) in self._buckets_G():
                self.dnG.backward_range(df_do if s_from == self._last_stage_G else None, s_from, s_to)
                if hi > lo:
                    self.coll.allreduce_sum_async_(gG[lo:hi])
            bucketed = True
        else:
            self.dnG.backward(df_do, param_grads=True, input_grad=False)
        res = dict(loss=loss, outputs=out, samples=samples)
        if keep_grad:
            g = gG.clone()
            o = self.opt
            if o["G_L1"] != 0 or o["G_L2"] != 0:
                g += torch.sign(pG) * o["G_L2"] + pG * o["G_L2"]
            if o["G_clamp"] != 0:
                g.clamp_(-o["G_clamp"], o["G_clamp"])
            res["grad"] = g
            res["f"] = loss.item() + self.penalty_f("G", pG)
        if bucketed:
            self.coll.wait()
        else:
            self._allreduce(gG)
        self._update("G", pG, gG, loss)
        self.dnG.params_changed()
        return res

    def _buckets_G(self):
        if getattr(self, "_bk_G", None) is None:
            self._bk_G = self.dnG.grad_buckets()
            self._last_stage_G = self._bk_G[0][0]
        return self._bk_G

    def iteration(self, real_nhwc, seed_noise=None, D_iterations=1, G_iterations=1):
        """One loop body of adversarial.lua:54-288 on device-resident inputs (used by bench.py)."""
        B = 2 * real_nhwc.shape[0]
        nd = self.dnG.in_c
        for _ in range(D_iterations):
            self.step_D(real_nhwc, None if self.gan is not None else S.next_noise(self.ctx, B // 2, nd))
        for _ in range(G_iterations):
            self.step_G(B if self.gan is not None else S.next_noise(self.ctx, B, nd))


def mean(t):
    """adversarial.mean (adversarial.lua:15-27)."""
    v = [x for x in t if isinstance(x, (int, float))]
    return sum(v) / len(v)


accs = []


def train(dataset, maxAccuracyD=1.01, accsInterval=20):
    """adversarial.train(dataset, maxAccuracyD, accsInterval) -- adversarial.lua:30-335.  Reads the same globals
    (OPT, MODEL_G, MODEL_D, IMG_DIMENSIONS, EPOCH, CONFUSION ...) from face_generator_amd.state.S."""
    global accs
    OPT = S.OPT
    S.EPOCH = S.EPOCH or 1
    N_epoch = OPT["N_epoch"]
    if N_epoch <= 0:
        N_epoch = dataset.size()
    dataBatchSize = OPT["batchSize"] // 2
    t0 = time.time()
    tr = S.trainer()
    ctx = tr.ctx
    if tr.world > 1:
        # every rank runs the SAME number of iterations (each one holds collectives): a shard that is a batch longer on one rank is
        # cut to the shortest one instead of leaving that rank alone in an all-reduce
        N_epoch = tr.coll.min_int(N_epoch)
    countTrainedD = countNotTrainedD = 0
    conf_total = torch.zeros(4, dtype=torch.int64)
    pending = []
    print("<trainer> Epoch #%d [batchSize = %d]" % (S.EPOCH, OPT["batchSize"]))

    def gate(acc):
        accs.append(acc)
        if len(accs) > accsInterval:
            accs.pop(0)
        return mean(accs) < maxAccuracyD
    use_gate = maxAccuracyD <= 1.0

    for t in range(1, N_epoch + 1, dataBatchSize):
        thisBatchSize = min(OPT["batchSize"], N_epoch - t + 1)
        if thisBatchSize < 4:
            print("[INFO] skipping batch at t=%d, because its size is less than 4" % t)
            break
        # odd thisBatchSize (odd N_epoch): Lua's `for i = 1, thisBatchSize / 2` fills floor(thisBatchSize / 2) real + as many
        # fake rows and leaves the last row of inputs / targets uninitialised; that garbage row is dropped here (C9)
        thisBatchSize -= thisBatchSize % 2
        half = thisBatchSize // 2
        for _ in range(OPT.get("D_iterations", 1)):
            idx = [S.rng.randrange(dataset.size()) for _ in range(half)]            # math.random picks (:245)
            real = torch.stack([torch.as_tensor(dataset[i], dtype=torch.float32) for i in idx])
            real_d = ctx.to_device_nhwc(real)
            nz = None if tr.gan is not None else S.next_noise(ctx, half, OPT["noiseDim"])   # fused: drawn inside the step
            r = tr.step_D(real_d, nz, gate=gate if use_gate else None)
            pending.append(r["confusion"].clone())
            if r["trained"]:
                countTrainedD += 1
            else:
                countNotTrainedD += 1
        for _ in range(OPT.get("G_iterations", 1)):
            tr.step_G(thisBatchSize if tr.gan is not None else S.next_noise(ctx, thisBatchSize, OPT["noiseDim"]))
    for c in pending:                      # deferred: one host read at the end of the epoch
        ch = c.cpu().to(torch.int64)
        conf_total += ch
        if not use_gate:                   # adversarial.accs (:156-159) is kept even when the gate can never fire
            cl = ch.tolist()
            accs.append((cl[0] + cl[3]) / max(1, sum(cl)))
            if len(accs) > accsInterval:
                accs.pop(0)
    tr.finish_pending()
    dt = time.time() - t0
    print("<trainer> time required for this epoch = %d s" % dt)
    print("<trainer> time to learn 1 sample = %f ms" % (1000 * dt / N_epoch))
    print("<trainer> trained D %d of %d times." % (countTrainedD, countTrainedD + countNotTrainedD))
    c = conf_total.tolist()
    tV = (c[0] + c[3]) / max(1, sum(c))
    print("Confusion of normal D: [pred][target] = %s  totalValid = %.4f" % (c, tV))
    S.CONFUSION = c
    if S.EPOCH % OPT.get("saveFreq", 30) == 0:
        from . import nn_utils
        nn_utils.save_checkpoint()
    S.EPOCH += 1
    return tV
