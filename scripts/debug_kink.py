import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_net as T
from oracle import torch7_nn as O
C, B = 1, 6
rng = np.random.default_rng(100 + C + B)
G = O.create_G32((C, 32, 32), 100, rng); D = O.create_D32b((C, 32, 32), rng)
for net in (G, D):
    for m in net.modules:
        if isinstance(m, O.SpatialBatchNormalization):
            m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
            m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
noise = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
G.forward(noise)
for i, (mod, xin) in enumerate(zip(G.modules, G._inputs)):
    if isinstance(mod, O.PReLU):
        a = np.abs(xin); j = np.unravel_index(a.argmin(), a.shape)
        print("PReLU", i + 1, "min|x|", a.min(), "at", j, "scale", a.max())
