"""DATASET (coarse-to-fine): the `_toResult` step of dataset_c2f.lua:49-109 that sits directly in front of the c2f
train step: coarse = fine scaled down to coarseScale and back up to fineScale, diff = fine - coarse, wrapped as an
indexable result with .fine/.coarse/.diff.  `image.scale` runs on the device through the library's
fg_c2f_coarse_diff (csrc/pointwise.hip scale_bilinear_kernel: the `image` package's bilinear algorithm bit for bit -- corner-aligned
linear interpolation up, fractional box mean down; the provenance note sits with the CPU restatement the parity tests use).  There is no CPU
fallback: without the library / a GPU this raises like every other compute entry.
JPEG loading (dataset_c2f.lua:111-215) is out of scope: the metric uses synthetic batches."""
import torch


class Example:
    __slots__ = ("coarse", "fine", "diff")

    def __init__(self, coarse, fine, diff):
        self.coarse, self.fine, self.diff = coarse, fine, diff


class Result:
    """dataset_c2f.lua:65-106: result.fine / .coarse / .diff tensors, result[i] -> {coarse, fine, diff}, :size()."""

    def __init__(self, fine, coarse, diff):
        self.fine, self.coarse, self.diff = fine, coarse, diff

    def size(self):
        return self.fine.shape[0]

    def __len__(self):
        return self.size()

    def getCoarse(self, index, endIndex=None):
        return self.coarse[index:endIndex] if endIndex is not None else self.coarse[index]

    def getFine(self, index, endIndex=None):
        return self.fine[index:endIndex] if endIndex is not None else self.fine[index]

    def getDiff(self, index, endIndex=None):
        return self.diff[index:endIndex] if endIndex is not None else self.diff[index]

    def __getitem__(self, i):
        return Example(self.coarse[i], self.fine[i], self.diff[i])


def toResult(fineImages, coarseScale, fineScale, ctx=None):
    """dataset._toResult (dataset_c2f.lua:49-63).  fineImages: FloatTensor [N, C, fineScale, fineScale] in [0, 1] (host, the
    reference's layout) -> Result of host tensors, computed on the device."""
    from . import ops
    from .runtime import get_context
    ctx = ctx or get_context()
    fine = torch.as_tensor(fineImages, dtype=torch.float32)
    if fine.shape[-1] != fineScale or fine.shape[-2] != fineScale:
        raise ValueError("toResult: fine images must be %dx%d" % (fineScale, fineScale))
    coarse, diff = ops.c2f_coarse_diff(fine.to(ctx.device), coarseScale, layout="nchw", ctx=ctx)
    return Result(fine.cpu(), coarse.cpu(), diff.cpu())


def toResultDevice(fine_nhwc, coarseScale, ctx=None):
    """The same step for a batch that already lives in HBM in the library's activation layout [N][S][S][C]: -> (coarse, diff) device
    tensors, the inputs of TrainerC2F.step_D / step_G."""
    from . import ops
    return ops.c2f_coarse_diff(fine_nhwc, coarseScale, layout="nhwc", ctx=ctx)
