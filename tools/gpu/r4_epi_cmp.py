"""f32-mode ResNet-18 trunk forward + backward (with and without the ReLU bit mask): save results for a fixed input (run once per library, compare the files)"""
import sys
import torch
import avec_amd
import nnet
from avec_amd import ops

out = sys.argv[1]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(16)
x = torch.randn(10, 22, 22, 64, generator=g).to(dev)
res = {}
avec_amd.set_compute_dtype("f32")
for bm in (True, False):
    torch.manual_seed(23)
    net = nnet.ResNet(dim_input=64, dim_output=256, model="ResNet18", include_stem=False, include_head=True).to(dev).train()
    ops.RELU_BITMASK = bm
    xin = x.clone().requires_grad_(True)
    y = net.forward_nhwc(xin)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    (y.float() * w).sum().backward()
    torch.cuda.synchronize()
    res["y_%d" % bm], res["dx_%d" % bm] = y.detach().float().cpu(), xin.grad.detach().float().cpu()
    for n, p in net.named_parameters():
        if p.grad is not None:
            res["g_%d_%s" % (bm, n)] = p.grad.detach().float().cpu().clone()
torch.save(res, out)
print("saved", len(res))
