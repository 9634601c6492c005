"""MODELS (coarse-to-fine): models_c2f.lua's live variants, `create_G(dimensions, cuda)` -> create_G_d (:113-145) and
`create_D(dimensions, cuda)` -> create_D_c (:237-278).  weight-init 'heuristic' (:138, :271) is a no-op on these nets
(it does not recurse into the inner Sequential), so Torch's default reset() initialisation applies."""
from . import nn
from .weight_init import w_init


def create_G_d(dimensions, cuda=False, max_batch=32, gen=None):
    c, h, w = dimensions
    inner = nn.Sequential()
    inner.add(nn.SpatialConvolutionUpsample(c + 1, 64, 3, 3, 1, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolutionUpsample(64, 64, 3, 3, 1, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolutionUpsample(64, 128, 5, 5, 1, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolutionUpsample(128, 256, 5, 5, 1, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolutionUpsample(256, c, 7, 7, 1, gen=gen))
    inner.add(nn.View(c, h, w))
    inner.input_dims = (c + 1, h, w)
    model_G = nn.TableSequential(nn.JoinTable(2, 2), inner)
    model_G = w_init(model_G, 'heuristic')        # models_c2f.lua:138 -- top level only: no module matches, a no-op
    if cuda:
        model_G.cuda(max_batch=max_batch)
    return model_G


def create_D_c(dimensions, cuda=False, max_batch=32, gen=None):
    c, h, w = dimensions
    inner = nn.Sequential()
    inner.add(nn.SpatialConvolution(c, 64, 3, 3, 1, 1, (3 - 1) // 2, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolution(64, 64, 3, 3, 1, 1, (3 - 1) // 2, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialMaxPooling(2, 2))
    inner.add(nn.SpatialConvolution(64, 128, 3, 3, 1, 1, (3 - 1) // 2, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialConvolution(128, 256, 3, 3, 1, 1, (3 - 1) // 2, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.SpatialMaxPooling(2, 2))
    inner.add(nn.Dropout())
    nfeat = int(256 * 0.25 * 0.25 * h * w)
    inner.add(nn.View(nfeat))
    inner.add(nn.Linear(nfeat, 512, gen=gen))
    inner.add(nn.PReLU())
    inner.add(nn.Dropout())
    inner.add(nn.Linear(512, 1, gen=gen))
    inner.add(nn.Sigmoid())
    inner.input_dims = (c, h, w)
    model_D = nn.TableSequential(nn.CAddTable(), inner)
    model_D = w_init(model_D, 'heuristic')        # models_c2f.lua:271 -- a no-op for the same reason
    if cuda:
        model_D.cuda(max_batch=max_batch)
    return model_D


def create_G(dimensions, cuda=False, **kw):
    """models_c2f.lua:12."""
    return create_G_d(dimensions, cuda, **kw)


def create_D(dimensions, cuda=False, **kw):
    """models_c2f.lua:152."""
    return create_D_c(dimensions, cuda, **kw)
