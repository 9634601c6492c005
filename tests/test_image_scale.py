"""CPU: the restatement of Torch7 `image.scale` (oracle/image_scale.py; dataset_c2f.lua:53-61) against
  * the committed known-answer vectors (tests/golden/image_scale.npz, written by an independent scalar walk of the C loop),
  * closed forms: an integer-factor down-scale is the box mean, an up-scale is corner-aligned linear interpolation
    (PyTorch's F.interpolate(align_corners=True) / avg_pool2d agree to rounding), equal sizes copy,
  * exact cases: images whose values and weights are dyadic give exact results."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import image_scale as IS

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_scale.npz"))


def test_restatement_reproduces_the_known_answer_vectors_bit_for_bit():
    n = 0
    while "case%d_in" % n in GOLD:
        x, y = GOLD["case%d_in" % n], GOLD["case%d_out" % n]
        got = IS.scale(x, y.shape[2], y.shape[1])
        assert got.dtype == np.float32 and np.array_equal(got, y), n
        n += 1
    assert n == 8
    coarse, diff = IS.to_result(GOLD["c2f_fine"], 32, 64)
    assert np.array_equal(coarse, GOLD["c2f_coarse"]) and np.array_equal(diff, GOLD["c2f_diff"])


def test_down_by_two_is_the_box_mean_and_up_is_corner_aligned():
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, (3, 64, 64)).astype(np.float32)
    down = IS.scale(x, 32, 32)
    ref = F.avg_pool2d(torch.tensor(x)[None], 2)[0].numpy()
    assert np.abs(down - ref).max() <= 2e-7                      # (a + b) / 2 per axis vs the 4-term mean: one rounding apart
    up = IS.scale(down, 64, 64)
    ref = F.interpolate(torch.tensor(down)[None], size=(64, 64), mode="bilinear", align_corners=True)[0].numpy()
    assert np.abs(up - ref).max() <= 1e-6
    # NOT the half-pixel convention (the round-5 stand-in): the two differ visibly at this size
    half = F.interpolate(torch.tensor(down)[None], size=(64, 64), mode="bilinear", align_corners=False)[0].numpy()
    assert np.abs(up - half).max() > 1e-2
    # corners map onto corners, equal sizes copy
    assert np.array_equal(up[:, 0, 0], down[:, 0, 0]) and np.array_equal(up[:, -1, -1], down[:, -1, -1])
    assert np.array_equal(IS.scale(x, 64, 64), x)


def test_fractional_box_mean_matches_the_closed_form():
    # 5 -> 4: scale 1.25; output d covers [1.25 d, 1.25 (d + 1)) with fractional end weights, divided by the covered length 1.25
    x = np.array([[[8.0, 16.0, 24.0, 32.0, 40.0]]], np.float32)
    y = IS.scale(x, 4, 1)[0, 0]
    want = [(8 + 0.25 * 16) / 1.25, (0.75 * 16 + 0.5 * 24) / 1.25, (0.5 * 24 + 0.75 * 32) / 1.25, (0.25 * 32 + 40) / 1.25]
    assert np.allclose(y, want, rtol=0, atol=2e-6)
    # 2 -> 4 up: scale 1/3 -> weights 0, 1/3, 2/3 and the copied end point
    y = IS.scale(np.array([[[3.0, 9.0]]], np.float32), 4, 1)[0, 0]
    assert np.allclose(y, [3.0, 5.0, 7.0, 9.0], rtol=0, atol=1e-6) and y[3] == 9.0 and y[0] == 3.0


def test_dyadic_images_are_exact():
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (3, 64, 64)).astype(np.float32)
    down = IS.scale(x, 32, 32)
    want = x.reshape(3, 32, 2, 32, 2).astype(np.float64).sum(axis=(2, 4)) / 4.0       # quarters of small integers: exact in fp32
    assert np.array_equal(down.astype(np.float64), want)
    coarse, diff = IS.to_result(x[None], 32, 64)
    assert np.array_equal((x[None] - coarse).astype(np.float32), diff)


@pytest.mark.parametrize("src,dst", [(1, 5), (5, 1), (7, 7), (64, 32), (32, 64), (48, 20), (20, 48), (3, 2), (2, 3)])
def test_axis_plan_covers_the_source_once(src, dst):
    """Weights of a down-scale row sum to its divisor and the rows tile [0, src) exactly once; an up-scale row's weights sum to 1."""
    cover = np.zeros(src)
    for terms, n in IS._axis_plan(src, dst):
        tot = sum(1.0 if w is None else float(w) for _, w in terms)
        if dst < src:
            assert abs(tot - float(n)) < 1e-5
            for si, w in terms:
                cover[si] += 1.0 if w is None else float(w)
        else:
            assert abs(tot - 1.0) < 1e-6
        assert all(0 <= si < src for si, _ in terms)
    if dst < src:
        assert np.allclose(cover, 1.0, atol=1e-4)
