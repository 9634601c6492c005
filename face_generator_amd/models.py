"""MODELS: the reference's model zoo for the hot path (models.lua), as layer lists for fg_net_create.

Same public names as models.lua: `create_G(dimensions, noiseDim)` (models.lua:87-93 -> create_G_decoder_upsampling32,
:57-81) and `create_D(dimensions)` (models.lua:98-104 -> create_D32b, :382-416).  Each returns an `nn.Sequential`
(face_generator_amd.nn) whose modules mirror the Torch7 modules one for one.
"""
from . import nn


def create_G_decoder_upsampling32(dimensions, noiseDim):
    """models.lua:57-81."""
    model = nn.Sequential()
    model.add(nn.Linear(noiseDim, 128 * 8 * 8))
    model.add(nn.View(128, 8, 8))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(nn.SpatialConvolution(128, 256, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(nn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU())

    model.add(nn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())
    model.input_dims = (noiseDim, 1, 1)
    return model


def create_G_decoder_upsampling16(dimensions, noiseDim):
    """models.lua:27-51: the same decoder started from a 4x4 map (no new kernel class)."""
    model = nn.Sequential()
    model.add(nn.Linear(noiseDim, 128 * 4 * 4))
    model.add(nn.View(128, 4, 4))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(nn.SpatialConvolution(128, 256, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(nn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU())

    model.add(nn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())
    model.input_dims = (noiseDim, 1, 1)
    return model


def create_G(dimensions, noiseDim):
    """models.lua:87-93."""
    if dimensions[1] == 16:
        return create_G_decoder_upsampling16(dimensions, noiseDim)
    return create_G_decoder_upsampling32(dimensions, noiseDim)


def create_D32b(dimensions):
    """models.lua:382-416."""
    conv = nn.Sequential()
    c, h, w = dimensions
    for (i, o) in ((c, 64), (64, 128), (128, 256), (256, 512)):
        conv.add(nn.SpatialConvolution(i, o, 3, 3, 1, 1, (3 - 1) // 2))
        conv.add(nn.PReLU())
        conv.add(nn.SpatialDropout(0.2))
        conv.add(nn.SpatialAveragePooling(2, 2, 2, 2))
    nfeat = int(512 * 0.25 * 0.25 * 0.25 * 0.25 * h * w)
    conv.add(nn.View(nfeat))
    conv.add(nn.Linear(nfeat, 512))
    conv.add(nn.PReLU())
    conv.add(nn.Dropout())
    conv.add(nn.Linear(512, 512))
    conv.add(nn.PReLU())
    conv.add(nn.Dropout())
    conv.add(nn.Linear(512, 1))
    conv.add(nn.Sigmoid())
    conv.input_dims = (c, h, w)
    return conv


def create_D(dimensions):
    """models.lua:98-104."""
    if dimensions[1] == 16:
        raise NotImplementedError("16x16 discriminator create_D16_d (models.lua:279-316: ConcatTable branches, stride-2 "
                                  "convolutions) is not built -- SURVEY 8(f) rank 4")
    return create_D32b(dimensions)
