"""GPU: fg_scale_bilinear / fg_c2f_coarse_diff (csrc/pointwise.hip scale_bilinear_kernel) against the restatement of Torch7
`image.scale` (oracle/image_scale.py) -- BIT-FOR-BIT (the kernel performs the same IEEE float operations in the same order), in both
layouts, at the reference's sizes (dataset_c2f.lua:53-61: 64 -> 32 -> 64) and at ragged ones, and against the committed known-answer
vectors; dataset_c2f.toResult / toResultDevice (the host mirror of dataset._toResult) ride on it."""
import os

import numpy as np
import pytest
import torch

from oracle import image_scale as IS

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_scale.npz"))


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


CASES = [(64, 64, 32, 32), (32, 32, 64, 64), (5, 7, 3, 9), (9, 4, 13, 6), (6, 6, 6, 6), (1, 1, 4, 5), (7, 5, 1, 1), (10, 3, 4, 11),
         (48, 48, 20, 20), (20, 20, 48, 48), (33, 17, 64, 8)]


@pytest.mark.parametrize("hs,ws,hd,wd", CASES)
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_scale_bilinear_is_bit_exact(ctx, hs, ws, hd, wd, layout):
    from face_generator_amd import ops
    rng = np.random.default_rng(hs * 1000 + ws * 10 + hd)
    N, C = 3, 3
    x = rng.standard_normal((N, C, hs, ws)).astype(np.float32)
    want = np.stack([IS.scale(img, wd, hd) for img in x])
    xd = torch.tensor(x, device=ctx.device)
    if layout == "nhwc":
        got = ops.scale_bilinear(xd.permute(0, 2, 3, 1).contiguous(), wd, hd, layout="nhwc", ctx=ctx).permute(0, 3, 1, 2)
    else:
        got = ops.scale_bilinear(xd, wd, hd, layout="nchw", ctx=ctx)
    assert np.array_equal(got.cpu().numpy(), want)


def test_known_answer_vectors(ctx):
    from face_generator_amd import ops
    n = 0
    while "case%d_in" % n in GOLD:
        x, y = GOLD["case%d_in" % n], GOLD["case%d_out" % n]
        got = ops.scale_bilinear(torch.tensor(x[None], device=ctx.device), y.shape[2], y.shape[1], layout="nchw", ctx=ctx)
        assert np.array_equal(got.cpu().numpy()[0], y), n
        n += 1
    fine = torch.tensor(GOLD["c2f_fine"], device=ctx.device)
    coarse, diff = ops.c2f_coarse_diff(fine, 32, layout="nchw", ctx=ctx)
    assert np.array_equal(coarse.cpu().numpy(), GOLD["c2f_coarse"]) and np.array_equal(diff.cpu().numpy(), GOLD["c2f_diff"])


@pytest.mark.parametrize("B,C,S", [(128, 3, 64), (16, 1, 32), (5, 3, 16)])
def test_c2f_data_step_at_the_baseline_size(ctx, B, C, S):
    """dataset._toResult at configs[3]'s batch (128 x 3 x 64 x 64, coarse 32): coarse and diff bit-exact, both layouts equal;
    integer-valued images (dyadic 2 x 2 means) give the exact quarter sums."""
    from face_generator_amd import dataset_c2f
    rng = np.random.default_rng(B + S)
    fine = rng.uniform(0, 1, (B, C, S, S)).astype(np.float32)
    want_c, want_d = IS.to_result(fine, S // 2, S)
    res = dataset_c2f.toResult(torch.tensor(fine), S // 2, S, ctx=ctx)
    assert res.size() == B and res[1].coarse.shape == (C, S, S)
    assert np.array_equal(res.coarse.numpy(), want_c) and np.array_equal(res.diff.numpy(), want_d)
    assert torch.equal(res.fine, torch.tensor(fine))
    c2, d2 = dataset_c2f.toResultDevice(torch.tensor(fine, device=ctx.device).permute(0, 2, 3, 1).contiguous(), S // 2, ctx=ctx)
    assert np.array_equal(c2.permute(0, 3, 1, 2).cpu().numpy(), want_c) and np.array_equal(d2.permute(0, 3, 1, 2).cpu().numpy(), want_d)
    ints = rng.integers(0, 256, (2, C, S, S)).astype(np.float32)
    got = dataset_c2f.toResult(torch.tensor(ints), S // 2, S, ctx=ctx)
    quarter = ints.reshape(2, C, S // 2, 2, S // 2, 2).astype(np.float64).sum(axis=(3, 5)) / 4.0
    # the coarse image at even corners of the up-scale grid: row / column 0 and S - 1 are copies of the down-scaled end pixels
    assert np.array_equal(got.coarse.numpy()[:, :, 0, 0].astype(np.float64), quarter[:, :, 0, 0])
    assert np.array_equal(got.coarse.numpy()[:, :, -1, -1].astype(np.float64), quarter[:, :, -1, -1])


def test_bad_arguments_are_refused(ctx):
    from face_generator_amd import FgError
    x = ctx.zeros(1, 4, 4, 3)
    with pytest.raises(FgError):
        ctx.check(ctx.lib.fg_scale_bilinear(ctx.h, x.data_ptr(), x.data_ptr(), 1, 3, 4, 4, 0, 4, 0))
    with pytest.raises(FgError):
        ctx.check(ctx.lib.fg_scale_bilinear(ctx.h, x.data_ptr(), x.data_ptr(), 1, 3, 4, 4, 4, 4, 2))
