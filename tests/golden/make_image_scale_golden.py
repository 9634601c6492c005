"""Generate tests/golden/image_scale.npz: known-answer vectors for Torch7 `image.scale` (bilinear) at the reference's call sites
(dataset_c2f.lua:53-61) and at ragged sizes.

The reference (Lua / Torch7 + the luarocks `image` package) cannot run in this environment and ships no vectors, so -- like every
other fixture here -- these are NOT outputs of the reference.  They are computed by the scalar loop below, a line-by-line float32
walk of `image_(Main_scaleLinear_rowcol)` written independently of oracle/image_scale.py's vectorised plan; tests/test_image_scale.py
checks the two against each other, against closed forms (2 x 2 box mean, corner-aligned interpolation) and against this file.
    python tests/golden/make_image_scale_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def rowcol(src, dst_len):
    """One row / column, scalar float32 arithmetic in the order of the C loop."""
    src_len = len(src)
    dst = np.zeros(dst_len, F)
    if dst_len > src_len:
        if src_len == 1:
            for di in range(dst_len - 1):
                dst[di] = src[0]
        else:
            scale = F(F(src_len - 1) / F(dst_len - 1))
            for di in range(dst_len - 1):
                si_f = F(F(di) * scale)
                si_i = int(si_f)
                si_f = F(si_f - F(si_i))
                a = F(F(F(1) - si_f) * src[si_i])
                b = F(si_f * src[si_i + 1])
                dst[di] = F(a + b)
        dst[dst_len - 1] = src[src_len - 1]
    elif dst_len < src_len:
        si0_i, si0_f = 0, F(0)
        scale = F(F(src_len) / F(dst_len))
        for di in range(dst_len):
            si1_f = F(F(di + 1) * scale)
            si1_i = int(si1_f)
            si1_f = F(si1_f - F(si1_i))
            acc = F(F(F(1) - si0_f) * src[si0_i])
            n = F(F(1) - si0_f)
            for si in range(si0_i + 1, si1_i):
                acc = F(acc + src[si])
                n = F(n + F(1))
            if si1_i < src_len:
                acc = F(acc + F(si1_f * src[si1_i]))
                n = F(n + si1_f)
            dst[di] = F(acc / n)
            si0_i, si0_f = si1_i, si1_f
    else:
        dst[:] = src
    return dst


def scale(img, width, height):
    """img [C][H][W] -> [C][height][width]: rows first, then the columns of the intermediate."""
    C, H, W = img.shape
    tmp = np.zeros((C, H, width), F)
    out = np.zeros((C, height, width), F)
    for k in range(C):
        for j in range(H):
            tmp[k, j] = rowcol(img[k, j], width)
        for i in range(width):
            out[k, :, i] = rowcol(tmp[k, :, i], height)
    return out


def main():
    rng = np.random.default_rng(20161)
    out = {}
    fine = rng.uniform(0, 1, (2, 3, 64, 64)).astype(F)            # dataset_c2f.lua:53-61 at fineScale 64 / coarseScale 32
    coarse = np.stack([scale(scale(f, 32, 32), 64, 64) for f in fine])
    out["c2f_fine"], out["c2f_coarse"], out["c2f_diff"] = fine, coarse, (fine - coarse).astype(F)
    cases = [(5, 7, 3, 9), (9, 4, 13, 6), (6, 6, 6, 6), (1, 1, 4, 5), (7, 5, 1, 1), (10, 3, 4, 11), (16, 16, 8, 8), (8, 8, 16, 16)]
    for n, (h, w, hd, wd) in enumerate(cases):                    # ragged: up / down / mixed / copy / 1-pixel axes
        x = rng.standard_normal((2, h, w)).astype(F)
        out["case%d_in" % n] = x
        out["case%d_out" % n] = scale(x, wd, hd)
    np.savez_compressed(os.path.join(HERE, "image_scale.npz"), **out)
    print("wrote image_scale.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
