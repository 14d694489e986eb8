"""Stand-alone forwards of the layers the hot path only runs fused (MaxPool3d, AvgPool1d, Upsample, depthwise Conv1d, the Conv3d stem shape, GlobalAvgPool2d,
ConvNeuralNetwork.forward) against plain PyTorch fp32 on the CPU: values and gradients.  fp32 compute; the GEMM-shaped ones run on the MFMA xf32 path, hence
the 2e-3 bound there; pooling / up-sampling are exact."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _run(mod_out_fn, x, wseed=3):
    """forward + backward of a scalar functional of the output; returns (y, dx) on the CPU"""
    x = x.clone().requires_grad_(True)
    y = mod_out_fn(x)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(wseed)).to(y.device)
    (y * w).sum().backward()
    if x.is_cuda:
        torch.cuda.synchronize()
    return y.detach().float().cpu(), x.grad.detach().float().cpu()


@pytest.mark.parametrize("k,s,pad,shape", [((1, 3, 3), (1, 2, 2), "same", (2, 8, 3, 11, 13)), ((1, 3, 3), (1, 2, 2), "same", (1, 64, 2, 44, 44)),
                                           ((1, 2, 2), (1, 2, 2), "valid", (2, 4, 2, 8, 10)), ((1, 5, 3), (1, 1, 2), "same", (1, 12, 2, 9, 7))])
def test_maxpool3d_standalone(k, s, pad, shape):
    import nnet
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1)) - 0.5            # mostly negative borders: the zero padding wins there, as in the reference
    mp = nnet.layers.MaxPool3d(k, s, padding=pad)
    y, dx = _run(mp, x.to(dev()))
    p = (k[2] // 2, (k[2] - 1) // 2, k[1] // 2, (k[1] - 1) // 2, 0, 0) if pad == "same" else (0,) * 6
    yr, dxr = _run(lambda t: F.max_pool3d(F.pad(t, p), k, s), x)
    assert y.shape == yr.shape
    assert torch.equal(y, yr) and rel_err(dx, dxr) < 1e-6               # an element that wins several overlapping windows sums their gradients in another order


def test_maxpool3d_unsupported_window_is_loud():
    import nnet
    with pytest.raises(RuntimeError, match="MaxPool3d on the HIP path"):
        nnet.layers.MaxPool3d((3, 3, 3), (1, 2, 2))(torch.zeros(1, 4, 3, 8, 8, device=dev()))


@pytest.mark.parametrize("cl", [False, True])
def test_avgpool1d_and_upsample_standalone(cl):
    import nnet
    x = torch.randn(3, 23, 16, generator=torch.Generator().manual_seed(2))              # (B, T, D)
    xin = x if cl else x.transpose(1, 2).contiguous()
    to_ref = (lambda t: t.transpose(1, 2)) if cl else (lambda t: t)
    ap = nnet.layers.AvgPool1d(4, channels_last=cl)
    y, dx = _run(ap, xin.to(dev()))
    yr, dxr = _run(lambda t: to_ref(F.avg_pool1d(to_ref(t), 4)), xin)
    assert y.shape == yr.shape and rel_err(y, yr) < 1e-6 and rel_err(dx, dxr) < 1e-6
    up = nnet.layers.Upsample(scale_factor=3, channels_last=cl)
    y, dx = _run(up, xin.to(dev()))
    yr, dxr = _run(lambda t: to_ref(F.interpolate(to_ref(t), scale_factor=3, mode="nearest")), xin)
    assert y.shape == yr.shape and torch.equal(y, yr) and rel_err(dx, dxr) < 1e-6


def test_global_avgpool2d_standalone():
    import nnet
    x = torch.randn(3, 8, 5, 7, generator=torch.Generator().manual_seed(4))
    for keep in (False, True):
        y, dx = _run(nnet.layers.GlobalAvgPool2d(keepdim=keep), x.to(dev()))
        yr, dxr = _run(lambda t: t.mean(dim=(2, 3), keepdim=keep), x)
        assert y.shape == yr.shape and rel_err(y, yr) < 1e-6 and rel_err(dx, dxr) < 1e-6


@pytest.mark.parametrize("K,stride,pad", [(15, 1, "same"), (15, 2, "same"), (7, 1, "causal")])
def test_depthwise_conv1d_standalone(K, stride, pad):
    import nnet
    torch.manual_seed(7)
    C = 16
    conv = nnet.layers.Conv1d(C, C, K, stride=stride, groups=C, padding=pad, channels_last=True).to(dev())
    x = torch.randn(2, 37, C, generator=torch.Generator().manual_seed(5))
    w, b = conv.weight.detach().cpu().contiguous().clone().requires_grad_(True), conv.bias.detach().cpu().clone().requires_grad_(True)
    pl = K - 1 if pad == "causal" else K // 2
    y, dx = _run(conv, x.to(dev()))
    yr, dxr = _run(lambda t: F.conv1d(F.pad(t.transpose(1, 2), (pl, K - 1 - pl)), w, b, stride=stride, groups=C).transpose(1, 2), x)
    assert y.shape == yr.shape and rel_err(y, yr) < 1e-5 and rel_err(dx, dxr) < 1e-5
    assert rel_err(conv.weight.grad.cpu(), w.grad) < 1e-5 and rel_err(conv.bias.grad.cpu(), b.grad) < 1e-5


def test_conv3d_stem_shape_standalone():
    import nnet
    torch.manual_seed(8)
    conv = nnet.layers.Conv3d(1, 64, (5, 7, 7), stride=(1, 2, 2), padding="same").to(dev())
    x = torch.randn(2, 1, 6, 24, 20, generator=torch.Generator().manual_seed(6))
    w, b = conv.weight.detach().cpu().contiguous().clone().requires_grad_(True), conv.bias.detach().cpu().clone().requires_grad_(True)
    y = conv(x.to(dev()))
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
    (y * g.to(dev())).sum().backward()
    torch.cuda.synchronize()
    yr = F.conv3d(F.pad(x, (3, 3, 3, 3, 2, 2)), w, b, stride=(1, 2, 2))
    (yr * g).sum().backward()
    assert y.shape == yr.shape and rel_err(y.cpu(), yr.detach()) < 2e-3
    assert rel_err(conv.weight.grad.cpu(), w.grad) < 2e-3 and rel_err(conv.bias.grad.cpu(), b.grad) < 2e-3
    with pytest.raises(RuntimeError, match="Conv3d on the HIP path"):
        nnet.layers.Conv3d(1, 8, (3, 3, 3)).to(dev())(x.to(dev()))


def test_conv_neural_network_forward_layer_by_layer():
    """ConvNeuralNetwork.forward on its own (nnet/modules.py:115-130): two Conv2d + BatchNorm2d + ReLU layers, lengths halved per layer, against the same
    stack of torch layers with the same weights."""
    import nnet
    torch.manual_seed(9)
    net = nnet.modules.ConvNeuralNetwork(8, [16, 16], 3, strides=[2, 1], norm="BatchNorm2d", act_fun="ReLU", dim=2).to(dev()).train()
    ref = torch.nn.Sequential()
    for layer in net.layers:
        conv, bn = layer[0], layer[1]
        c = torch.nn.Conv2d(conv.in_channels, conv.out_channels, 3, stride=conv.stride, padding=1)
        c.weight.data.copy_(conv.weight.detach().cpu()); c.bias.data.copy_(conv.bias.detach().cpu())
        n = torch.nn.BatchNorm2d(conv.out_channels)
        n.weight.data.copy_(bn.weight.detach().cpu()); n.bias.data.copy_(bn.bias.detach().cpu())
        ref.append(torch.nn.Sequential(c, n, torch.nn.ReLU()))
    ref.train()
    x = torch.randn(3, 8, 12, 10, generator=torch.Generator().manual_seed(10))
    xl = torch.tensor([12, 7, 3])
    xg = x.to(dev()).requires_grad_(True)
    y, yl = net(xg, xl.to(dev()))
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
    (y * g.to(dev())).sum().backward()
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * g).sum().backward()
    assert y.shape == yr.shape and rel_err(y.detach().cpu(), yr.detach()) < 2e-3
    assert yl.cpu().tolist() == [3, 2, 1]                                                # ((n - 1) // 2 + 1) twice
    assert rel_err(xg.grad.cpu(), xr.grad) < 5e-3
    for layer, rl in zip(net.layers, ref):
        assert rel_err(layer[0].weight.grad.cpu(), rl[0].weight.grad) < 5e-3
        assert rel_err(layer[1].weight.grad.cpu(), rl[1].weight.grad) < 5e-3
        assert rel_err(layer[1].running_var.cpu(), rl[1].running_var) < 1e-3
