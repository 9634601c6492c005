"""cudnn.SpatialConvolutionUpsample with factor > 1 (layers/cudnnSpatialConvolutionUpsample.lua:4-58; named by north_star): a
'same' convolution to nOut * f^2 planes whose contiguous NCHW output is RE-VIEWED as [N][nOut][h f][w f] -- a flat
reinterpretation, not a pixel shuffle -- with gradOutput viewed back in both backward calls.  Module level
(fg_conv_upsample_view_* behind the nn protocol) and inside a compiled plan (one extra stage), against the oracle
(oracle/torch7_nn.py SpatialConvolutionUpsample, itself pinned against the flat-view formula in tests/test_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_net import check_flat_grads, draw_kink_safe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def test_view_entries_are_the_flat_nchw_view_and_its_inverse(ctx):
    """fg_conv_upsample_view_forward == Tensor:view on contiguous NCHW memory; _backward is its inverse; bad shapes are refused."""
    rng = np.random.default_rng(5)
    for (B, h, w, nout, f) in [(3, 4, 6, 2, 2), (2, 5, 3, 4, 3), (2, 8, 8, 16, 2)]:
        C = nout * f * f
        v = rng.standard_normal((B, C, h, w)).astype(np.float32)
        u_ref = v.reshape(B, nout, h * f, w * f)                                  # the reference's self.output:view(...)
        vd = nhwc(v, ctx.device)
        ud = ctx.empty(B, h * f, w * f, nout)
        ctx.check(ctx.lib.fg_conv_upsample_view_forward(ctx.h, vd.data_ptr(), ud.data_ptr(), B, h, w, C, f))
        assert np.array_equal(nchw(ud), u_ref)
        back = ctx.empty(B, h, w, C)
        ctx.check(ctx.lib.fg_conv_upsample_view_backward(ctx.h, ud.data_ptr(), back.data_ptr(), B, h, w, C, f))
        assert np.array_equal(nchw(back), v)
    assert ctx.lib.fg_conv_upsample_view_forward(ctx.h, vd.data_ptr(), ud.data_ptr(), 2, 8, 8, 30, 2) < 0     # 30 % 4 != 0


@pytest.mark.parametrize("nin,nout,k,f,S", [(8, 4, 3, 2, 8), (16, 4, 5, 3, 6), (64, 1, 3, 2, 8)])
def test_module_protocol_factor_gt_1(ctx, nin, nout, k, f, S):
    """updateOutput / updateGradInput / accGradParameters of the module itself (the last case is a thin-output convolution)."""
    from face_generator_amd import nn
    B = 3
    rng = np.random.default_rng(40 + nin)
    om = O.SpatialConvolutionUpsample(nin, nout, k, k, f, rng)
    m = nn.SpatialConvolutionUpsample(nin, nout, k, k, f)
    assert tuple(m.weight.shape) == om.weight.shape == (nout * f * f, nin, k, k)
    m.weight = dev(om.weight, ctx.device); m.bias = dev(om.bias, ctx.device)
    m.gradWeight = torch.zeros_like(m.weight); m.gradBias = torch.zeros_like(m.bias)
    x = rng.standard_normal((B, nin, S, S)).astype(np.float32)
    gy = rng.standard_normal((B, nout, S * f, S * f)).astype(np.float32)
    y_ref = om.forward(x)
    gx_ref = om.backward(x, gy)
    xd = nhwc(x, ctx.device)
    y = m.updateOutput(xd)
    assert tuple(y.shape) == (B, S * f, S * f, nout)
    close(nchw(y), y_ref, atol=2e-5 * max(1, np.abs(y_ref).max()), what="SpatialConvolutionUpsample forward")
    gx = m.updateGradInput(xd, nhwc(gy, ctx.device))
    close(nchw(gx), gx_ref, atol=1e-4 * np.abs(gx_ref).max(), what="SpatialConvolutionUpsample gradInput")
    m.accGradParameters(xd, nhwc(gy, ctx.device))
    close(m.gradWeight.cpu().numpy(), om.gradWeight, atol=1e-4 * np.abs(om.gradWeight).max(), what="gradWeight")
    close(m.gradBias.cpu().numpy(), om.gradBias, atol=1e-4 * np.abs(om.gradBias).max(), what="gradBias")


def test_compiled_plan_with_upsampling_convolutions(ctx):
    """A decoder in the style of models.lua's commented-out G variants: two SpatialConvolutionUpsample(f = 2) layers with a
    BatchNorm + PReLU between them, compiled by fg_net_create (fg_layer_spec.q = factor): forward, input gradient and the flat
    parameter gradient (getParameters() order) against the oracle."""
    from face_generator_amd import nn
    B, S = 4, 8
    rng = np.random.default_rng(77)
    onet = O.Sequential(O.SpatialConvolutionUpsample(8, 16, 3, 3, 2, rng), O.SpatialBatchNormalization(16, rng=rng), O.PReLU(),
                        O.SpatialConvolutionUpsample(16, 16, 5, 5, 2, rng), O.PReLU(),
                        O.SpatialConvolutionUpsample(16, 8, 3, 3, 1, rng), O.Sigmoid())
    for m in onet.modules:
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
        if isinstance(m, O.SpatialBatchNormalization):
            m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
            m.bias[...] = (0.2 * rng.standard_normal(m.bias.shape)).astype(np.float32)
    p_ref, g_ref = onet.getParameters()
    net = nn.Sequential()
    net.add(nn.SpatialConvolutionUpsample(8, 16, 3, 3, 2)).add(nn.SpatialBatchNormalization(16)).add(nn.PReLU())
    net.add(nn.SpatialConvolutionUpsample(16, 16, 5, 5, 2)).add(nn.PReLU())
    net.add(nn.SpatialConvolutionUpsample(16, 8, 3, 3, 1)).add(nn.Sigmoid())
    net.input_dims = (8, S, S)
    net.cuda(ctx, max_batch=B)
    dn = net.device_net
    assert dn.n_params == p_ref.size
    dn.params.copy_(torch.tensor(p_ref)); dn.params_changed()
    x, y_ref = draw_kink_safe(rng, lambda: rng.standard_normal((B, 8, S, S)).astype(np.float32), onet.forward, [onet])
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    g_ref[...] = 0
    gx_ref = onet.backward(x, gy)
    y = dn.forward(nhwc(x, ctx.device), train=True)
    assert tuple(y.shape) == (B, 4 * S, 4 * S, 8)
    close(nchw(dn.layer_output(0)), onet.modules[0].output, atol=2e-5 * np.abs(onet.modules[0].output).max(), what="layer 1 (viewed)")
    close(nchw(dn.layer_output(3)), onet.modules[3].output, atol=5e-5 * np.abs(onet.modules[3].output).max(), what="layer 4 (viewed)")
    close(nchw(y), y_ref, atol=1e-5, what="plan output")
    gx = dn.backward(nhwc(gy, ctx.device), param_grads=True, input_grad=True)
    close(nchw(gx), gx_ref, atol=1e-4 * np.abs(gx_ref).max() + 1e-8, what="plan gradInput")
    check_flat_grads(dn.grads.cpu().numpy(), onet, "upsampling decoder")


def test_nearest_upsample_in_front_of_a_factor_2_convolution_is_not_folded(ctx):
    """nn.SpatialUpSamplingNearest(2) followed by cudnn.SpatialConvolutionUpsample(.., factor = 2) -- the nearest-x2 tap
    folding of the plan compiler must leave this pair alone (the folded stage has no view behind it): dims (nOut, 4 h, 4 w), values
    and gradients against the oracle's un-folded modules."""
    from face_generator_amd import nn
    B, S = 3, 4
    rng = np.random.default_rng(91)
    onet = O.Sequential(O.SpatialUpSamplingNearest(2), O.SpatialConvolutionUpsample(8, 8, 3, 3, 2, rng), O.PReLU(),
                        O.SpatialUpSamplingNearest(2), O.SpatialConvolutionUpsample(8, 8, 5, 5, 1, rng))
    onet.modules[2].weight[0] = np.float32(0.3)
    p_ref, g_ref = onet.getParameters()
    net = nn.Sequential()
    net.add(nn.SpatialUpSamplingNearest(2)).add(nn.SpatialConvolutionUpsample(8, 8, 3, 3, 2)).add(nn.PReLU())
    net.add(nn.SpatialUpSamplingNearest(2)).add(nn.SpatialConvolutionUpsample(8, 8, 5, 5, 1))
    net.input_dims = (8, S, S)
    net.cuda(ctx, max_batch=B)
    dn = net.device_net
    assert dn.n_params == p_ref.size
    dn.params.copy_(torch.tensor(p_ref)); dn.params_changed()
    x, y_ref = draw_kink_safe(rng, lambda: rng.standard_normal((B, 8, S, S)).astype(np.float32), onet.forward, [onet])
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    g_ref[...] = 0
    gx_ref = onet.backward(x, gy)
    y = dn.forward(nhwc(x, ctx.device), train=True)
    assert tuple(y.shape) == (B, 8 * S, 8 * S, 8)
    close(nchw(dn.layer_output(1)), onet.modules[1].output, atol=2e-5 * np.abs(onet.modules[1].output).max(), what="viewed conv output")
    close(nchw(y), y_ref, atol=2e-5 * np.abs(y_ref).max(), what="plan output")
    gx = dn.backward(nhwc(gy, ctx.device), param_grads=True, input_grad=True)
    close(nchw(gx), gx_ref, atol=1e-4 * np.abs(gx_ref).max() + 1e-8, what="plan gradInput")
    check_flat_grads(dn.grads.cpu().numpy(), onet, "upsample + factor-2 convolution")
