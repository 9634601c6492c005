"""DATASET (coarse-to-fine): the `_toResult` step of dataset_c2f.lua:49-109 that sits directly in front of the c2f
train step: coarse = fine scaled down to coarseScale and back up to fineScale, diff = fine - coarse, wrapped as an
indexable result with .fine/.coarse/.diff.  Host-side data preparation: `image.scale` of the
Lua `image` package is bilinear; torch's bilinear resize stands in for it (not bit-identical, not on the timed path).
JPEG loading (dataset_c2f.lua:111-215) is out of scope: the metric uses synthetic batches."""
import torch
import torch.nn.functional as F


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


def toResult(fineImages, coarseScale, fineScale):
    """dataset._toResult (dataset_c2f.lua:49-63).  fineImages: FloatTensor [N, C, fineScale, fineScale] in [0, 1]."""
    fine = torch.as_tensor(fineImages, dtype=torch.float32)
    tmp = F.interpolate(fine, size=(coarseScale, coarseScale), mode="bilinear", align_corners=False, antialias=False)
    coarse = F.interpolate(tmp, size=(fineScale, fineScale), mode="bilinear", align_corners=False)
    return Result(fine, coarse, fine - coarse)
