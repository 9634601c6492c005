"""GPU: the independently assembled era-format checkpoint (tests/golden/adversarial_small.net, written by
tests/golden/make_t7_fixture.py with struct.pack only) loads through t7_checkpoint.load_checkpoint into device nets whose
evaluate-mode forward -- sample.lua:251-258 + :69-90, the consumer of a reference checkpoint -- matches the oracle's outputs stored
beside it (G images from 6 noise vectors, D's probabilities on them)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_loaded_checkpoint_forward_matches_the_oracle():
    from face_generator_amd import t7_checkpoint as C
    from face_generator_amd.runtime import get_context
    ctx = get_context(0)
    exp = np.load(os.path.join(GOLDEN, "adversarial_small_expect.npz"))
    ck = C.load_checkpoint(os.path.join(GOLDEN, "adversarial_small.net"))
    G, D = ck["G"], ck["D"]
    B = exp["noise"].shape[0]
    G.cuda(ctx, max_batch=B); D.cuda(ctx, max_batch=B)
    G.evaluate(); D.evaluate()                        # BN running stats from the file, SpatialDropout x (1 - p), Dropout identity
    y = G.device_net.forward(torch.tensor(exp["noise"], device=ctx.device))
    img = y.permute(0, 3, 1, 2).cpu().numpy()
    assert img.shape == exp["G_images"].shape == (B, 3, 16, 16)
    assert np.abs(img - exp["G_images"]).max() <= 2e-5
    p = D.device_net.forward(y.clone()).cpu().numpy().reshape(-1)
    assert np.abs(p - exp["D_out"].reshape(-1)).max() <= 2e-5
    assert exp["G_images"].std() > 1e-2 and exp["D_out"].std() > 1e-4        # not a degenerate (all 0.5) vector
