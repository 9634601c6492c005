"""The Lua globals of train.lua:89-94 (OPT, MODEL_G, MODEL_D, CRITERION, PARAMETERS_*, OPTSTATE, IMG_DIMENSIONS,
EPOCH ...) that nn_utils.lua and adversarial.lua read directly -- they are part of the API surface."""
import random


class _State:
    def __init__(self):
        self.reset()

    def reset(self):
        self.OPT = dict(batchSize=32, noiseDim=100, N_epoch=1000, D_iterations=1, G_iterations=1, D_L1=0.0, D_L2=1e-4,
                        G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0, D_optmethod="adam", G_optmethod="adam",
                        saveFreq=30, save="logs", seed=1, scale=32, grayscale=False)
        self.MODEL_G = None
        self.MODEL_D = None
        self.CRITERION = None
        self.IMG_DIMENSIONS = (3, 32, 32)
        self.EPOCH = 1
        self.CONFUSION = None
        self.Y_GENERATOR, self.Y_NOT_GENERATOR = 0, 1
        self.rng = random.Random(1)          # math.randomseed(seed): real-image picks (adversarial.lua:245)
        self.noise_seed, self.noise_offset = 1, 0   # torch.manualSeed(seed): noise stream (nn_utils.lua:37)
        self._trainer = None
        self.dist = None
        self.OPTSTATE = None                 # train.lua:180-191, created with the trainer

    def next_noise(self, ctx, n, dim):
        """NN_UTILS.createNoiseInputs on the device: U(-1,1), Philox stream keyed by (seed, running offset)."""
        z = ctx.uniform((n, dim), -1.0, 1.0, self.noise_seed, self.noise_offset)
        self.noise_offset += (n * dim + 3) // 4
        return z

    def trainer(self):
        if self._trainer is None:
            from .adversarial import Trainer
            from .runtime import get_context
            self._trainer = Trainer(get_context(), self.MODEL_G, self.MODEL_D, self.OPT, dist=self.dist)
            self.OPTSTATE = self._trainer.optstate
        return self._trainer

    def set_dist(self, dist):
        """Data parallelism (one process per GPU): every rank draws ITS shard of the global batch -- the three RNG
        streams of train.lua:64-65, 80 (real-image picks, noise, dropout masks) are offset by the rank -- and only rank
        0 writes checkpoints (nn_utils.save_checkpoint)."""
        self.dist = dist
        rank = dist.get_rank() if dist is not None else 0
        seed = self.OPT.get("seed", 1)
        self.rng = random.Random(seed + rank)
        self.noise_seed = seed + rank
        k = 0
        for net in (self.MODEL_G, self.MODEL_D):
            dn = getattr(net._inner(), "device_net", None) if net is not None else None
            if dn is not None:
                for d in ([dn] if not hasattr(dn, "_nets") else dn._nets()):
                    k += 1
                    # one key per net and rank, in a namespace (upper 32 bits) no noise seed and no default net key reaches
                    d.mask_seed = (0x4D41534C << 32) + 1000 * (seed + rank) + k


S = _State()
