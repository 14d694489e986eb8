"""nnet.Module: device tracking, added losses / infos (mirror of the reference's nnet/module.py:22-87 API)."""
from collections import OrderedDict

import torch
import torch.nn as nn


class Module(nn.Module):
    def __init__(self):
        super().__init__()
        self.modules_buffer = OrderedDict()
        self.device = torch.device("cpu")
        self.added_losses = OrderedDict()
        self.infos = OrderedDict()

    # -- losses / infos collected by Model.forward_model ------------------------------------
    def add_loss(self, name, loss, weight=1.0):
        self.added_losses[name] = {"loss": loss, "weight": weight}

    def add_info(self, name, info):
        self.infos[name] = info

    def reset_losses(self):
        self.added_losses = OrderedDict()

    def reset_infos(self):
        self.infos = OrderedDict()

    # -- frozen helper sub-networks (kept out of parameters()/state_dict) --------------------
    def register_module_buffer(self, name, module):
        self.set_require_grad(module, False)
        module.eval()
        self.modules_buffer[name] = module
        object.__setattr__(self, name, module)

    def set_require_grad(self, networks, require_grad=True):
        for net in (networks if isinstance(networks, list) else [networks]):
            if net is not None:
                net.requires_grad_(require_grad)

    def to(self, device):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        for key, value in self.modules_buffer.items():
            self.modules_buffer[key] = value.to(self.device)
        return super().to(device)

    def transfer_to_device(self, struct, device=None):
        dev = self.device if device is None else device
        if isinstance(struct, dict):
            return {k: self.transfer_to_device(v, dev) for k, v in struct.items()}
        if isinstance(struct, list):
            return [self.transfer_to_device(v, dev) for v in struct]
        if isinstance(struct, tuple):
            return tuple(self.transfer_to_device(v, dev) for v in struct)
        if isinstance(struct, (torch.Tensor, nn.Module)):
            return struct.to(dev)
        if struct is None:
            return None
        raise TypeError("cannot move a %s to a device (expected None, a tensor, a module, or a dict / list / tuple of those)" % type(struct).__name__)
