"""CPU restatement of Torch7 `image.scale(src, width, height)` (default mode 'bilinear') -- TEST INFRASTRUCTURE ONLY
(tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does).

Where the reference calls it: dataset_c2f.lua:53-56 (`_toResult`):
    tmp    = image.scale(fineImages[i], coarseScale, coarseScale)
    coarse = image.scale(tmp, fineScale, fineScale)
    diff   = fine - coarse                                     (dataset_c2f.lua:59-61)

The arithmetic lives in the un-vendored luarocks package `image` (torch/image, era Dec 2015 - Feb 2016; not under /root/reference, not
version-pinned -- parity UNPINNED, like the rest of the oracle).  (!) Restated from upstream knowledge of `generic/image.c`:
`image_(Main_scaleBilinear)` runs `image_(Main_scaleLinear_rowcol)` over every row (width pass, into a temporary of
[channels][src_height][dst_width]) and then over every column of that temporary (height pass).  One axis, src_len -> dst_len:

  * dst_len > src_len  (up):   scale = (float)(src_len - 1) / (dst_len - 1);  for di < dst_len - 1:
                                   si_f = di * scale; si_i = (long)si_f; si_f -= si_i;
                                   dst[di] = (1 - si_f) * src[si_i] + si_f * src[si_i + 1]
                               dst[dst_len - 1] = src[src_len - 1]            (src_len == 1: every dst = src[0])
                               -- i.e. the "align corners" convention: end points map onto end points.
  * dst_len < src_len  (down): scale = (float)src_len / dst_len; a box filter with fractional coverage of the end pixels:
                                   acc = (1 - si0_f) * src[si0_i]; n = 1 - si0_f
                                   for si0_i < si < si1_i: acc += src[si]; n += 1
                                   if si1_i < src_len: acc += si1_f * src[si1_i]; n += si1_f
                                   dst[di] = acc / n            with (si0, si1) = (di, di + 1) * scale split into integer + fraction
  * dst_len == src_len: copy.

All arithmetic in C `float` (real = float for the FloatTensors of dataset_c2f.lua), products and sums rounded one by one (no fused
multiply-add: the 2016 builds targeted SSE2).  At the reference's sizes (64 -> 32 -> 64) the down pass is the exact 2 x 2 box mean
((a + b) / 2 per axis) and the up pass a 31/63-spaced linear interpolation.

Why this convention and not torch.nn.functional.interpolate(align_corners=False) (the round-5 stand-in): `image.scale` predates the
half-pixel convention; its up-scaling maps corner to corner and its down-scaling is an area mean, not a point-sampled bilinear tap.
The two differ by up to ~0.25 of a pixel step at the borders -- visible in `diff`, which is what G is trained on.

Every function here works on float32 numpy arrays and keeps the reference's order of operations, so a device kernel that does
the same IEEE operations agrees bit for bit."""
import numpy as np

F32 = np.float32


def _axis_plan(src_len, dst_len):
    """-> per destination index the list of (source index, weight) terms in the reference's order, the divisor n (None = no division)
    -- all float32, computed exactly as the C loop does."""
    plan = []
    if dst_len > src_len:
        if src_len == 1:
            return [([(0, None)], None) for _ in range(dst_len)]
        scale = F32(src_len - 1) / F32(dst_len - 1)
        for di in range(dst_len - 1):
            si_f = F32(di) * scale
            si_i = int(si_f)
            si_f = F32(si_f - F32(si_i))
            plan.append(([(si_i, F32(1) - si_f), (si_i + 1, si_f)], None))
        plan.append(([(src_len - 1, None)], None))
    elif dst_len < src_len:
        scale = F32(src_len) / F32(dst_len)
        si0_i, si0_f = 0, F32(0)
        for di in range(dst_len):
            si1_f = F32(di + 1) * scale
            si1_i = int(si1_f)
            si1_f = F32(si1_f - F32(si1_i))
            terms = [(si0_i, F32(1) - si0_f)]
            n = F32(1) - si0_f
            for si in range(si0_i + 1, si1_i):
                terms.append((si, None))
                n = F32(n + F32(1))
            if si1_i < src_len:
                terms.append((si1_i, si1_f))
                n = F32(n + si1_f)
            plan.append((terms, n))
            si0_i, si0_f = si1_i, si1_f
    else:
        plan = [([(i, None)], None) for i in range(dst_len)]
    return plan


def scale_axis(src, dst_len, axis):
    """image_(Main_scaleLinear_rowcol) along `axis` of a float32 array (all other axes are independent rows / columns)."""
    src = np.moveaxis(np.asarray(src, F32), axis, -1)
    out = np.empty(src.shape[:-1] + (dst_len,), F32)
    for di, (terms, n) in enumerate(_axis_plan(src.shape[-1], dst_len)):
        acc = None
        for (si, w) in terms:
            v = src[..., si] if w is None else (w * src[..., si]).astype(F32)     # one rounded product
            acc = v.astype(F32) if acc is None else (acc + v).astype(F32)          # one rounded sum
        out[..., di] = acc if n is None else (acc / n).astype(F32)
    return np.moveaxis(out, -1, axis)


def scale(src, width, height):
    """image.scale(src, width, height) for src [C][H][W] (or [H][W]): the width pass over every row first, then the height pass."""
    src = np.asarray(src, F32)
    tmp = scale_axis(src, width, src.ndim - 1)
    return scale_axis(tmp, height, src.ndim - 2)


def to_result(fine, coarse_scale, fine_scale):
    """dataset._toResult (dataset_c2f.lua:49-63) on a batch [N][C][S][S]: -> (coarse, diff)."""
    fine = np.asarray(fine, F32)
    coarse = np.stack([scale(scale(f, coarse_scale, coarse_scale), fine_scale, fine_scale) for f in fine])
    return coarse, (fine - coarse).astype(F32)
