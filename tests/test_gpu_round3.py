"""Round-3 GPU tests: BatchNorm-backward reductions folded into the backward-data epilogues of the ResNet (ops.BnbFuse), guarded optimizer step."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_resnet_bn_backward_fusion_matches_unfused():
    """bf16 ResNet-18 trunk (no stem): gradients with the BatchNorm-backward mask + (sum d, sum d*y) reductions computed inside the backward-data epilogues
    (stages 2-4 and the stage boundaries) against the stand-alone reduction kernels, same weights and input."""
    import avec_amd
    import nnet
    from avec_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(24, 22, 22, 64, generator=g).to(dev())
    res = {}
    try:
        avec_amd.set_compute_dtype("bf16")
        for fuse in (True, False):
            torch.manual_seed(11)
            net = nnet.ResNet(dim_input=64, dim_output=256, model="ResNet18", include_stem=False, include_head=True).to(dev()).train()
            ops.BNB_FUSE = fuse
            xin = x.to(torch.bfloat16).requires_grad_(True)
            y = net.forward_nhwc(xin)
            w = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev())
            (y.float() * w).sum().backward()
            torch.cuda.synchronize()
            res[fuse] = (y.detach().float().cpu(), xin.grad.detach().float().cpu(),
                         {n: p.grad.detach().float().cpu().clone() for n, p in net.named_parameters() if p.grad is not None})
    finally:
        ops.BNB_FUSE = False
        avec_amd.set_compute_dtype("f32")
    assert torch.equal(res[True][0], res[False][0])
    assert rel_err(res[True][1], res[False][1]) < 2e-2, "input gradient"
    assert len(res[True][2]) == len(res[False][2]) and len(res[True][2]) > 40
    for n in res[True][2]:
        assert rel_err(res[True][2][n], res[False][2][n]) < 3e-2, n
