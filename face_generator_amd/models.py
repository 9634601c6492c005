"""MODELS: the reference's model zoo for the hot path (models.lua), as layer lists for fg_net_create.

Same public names as models.lua: `create_G(dimensions, noiseDim)` (models.lua:87-93 -> create_G_decoder_upsampling32,
:57-81) and `create_D(dimensions)` (models.lua:98-104 -> create_D32b, :382-416).  Each returns an `nn.Sequential`
(face_generator_amd.nn) whose modules mirror the Torch7 modules one for one.
"""
from . import nn
from .nn import cudnn
from .weight_init import w_init


def create_G_decoder_upsampling32(dimensions, noiseDim, gen=None):
    """models.lua:57-81."""
    model = nn.Sequential()
    model.add(nn.Linear(noiseDim, 128 * 8 * 8))
    model.add(nn.View(128, 8, 8))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(128, 256, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU())

    model.add(cudnn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())
    model.input_dims = (noiseDim, 1, 1)
    model = w_init(model, 'heuristic', gen=gen)      # models.lua:48 / :78
    return model


def create_G_decoder_upsampling16(dimensions, noiseDim, gen=None):
    """models.lua:27-51: the same decoder started from a 4x4 map (no new kernel class)."""
    model = nn.Sequential()
    model.add(nn.Linear(noiseDim, 128 * 4 * 4))
    model.add(nn.View(128, 4, 4))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(128, 256, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU())

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU())

    model.add(cudnn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())
    model.input_dims = (noiseDim, 1, 1)
    model = w_init(model, 'heuristic', gen=gen)      # models.lua:48 / :78
    return model


def create_G(dimensions, noiseDim, gen=None):
    """models.lua:87-93."""
    if dimensions[1] == 16:
        return create_G_decoder_upsampling16(dimensions, noiseDim, gen=gen)
    return create_G_decoder_upsampling32(dimensions, noiseDim, gen=gen)


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


def create_D16_d(dimensions):
    """models.lua:279-316: ConcatTable{conv branch with two stride-2 convs, dense branch} -> JoinTable(2) -> Linear -> Sigmoid
    (SURVEY 8(f) rank 4)."""
    c, h, w = dimensions
    inputSz = c * h * w
    fine = nn.Sequential()
    fine.add(nn.SpatialConvolution(c, 128, 3, 3, 1, 1, (3 - 1) // 2))
    fine.add(nn.PReLU())
    fine.add(nn.SpatialConvolution(128, 128, 3, 3, 1, 1, (3 - 1) // 2))
    fine.add(nn.PReLU())
    fine.add(nn.SpatialAveragePooling(2, 2, 2, 2))
    fine.add(nn.SpatialConvolution(128, 512, 3, 3, 2, 2, (3 - 1) // 2))
    fine.add(nn.PReLU())
    fine.add(nn.SpatialConvolution(512, 1024, 3, 3, 2, 2, (3 - 1) // 2))
    fine.add(nn.PReLU())
    fine.add(nn.SpatialDropout())
    fine_size = int(1024 * 0.25 * 0.25 * 0.25 * h * w)
    fine.add(nn.View(fine_size))
    fine.add(nn.Linear(fine_size, 1024))
    fine.add(nn.PReLU())

    dense = nn.Sequential()
    dense.add(nn.View(inputSz))
    dense.add(nn.Linear(inputSz, 128))
    dense.add(nn.PReLU())
    dense.add(nn.Dropout())
    dense.add(nn.Linear(128, 128))
    dense.add(nn.PReLU())

    tail = nn.Sequential()
    tail.add(nn.Linear(1024 + 128, 1))
    tail.add(nn.Sigmoid())
    model = nn.ConcatSequential([fine, dense], tail)
    model.input_dims = (c, h, w)
    return model


def create_D(dimensions):
    """models.lua:98-104."""
    if dimensions[1] == 16:
        return create_D16_d(dimensions)
    return create_D32b(dimensions)
