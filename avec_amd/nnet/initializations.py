"""Weight initialisers by name (registry contract of nnet/initializations.py:72-91)."""
import math

import torch.nn.init as init


def _ku(gain_kw):
    return lambda t, mode="fan_in": init.kaiming_uniform_(t, mode=mode, **gain_kw)


def _kn(gain_kw):
    return lambda t, mode="fan_in": init.kaiming_normal_(t, mode=mode, **gain_kw)


init_dict = {
    "uniform": init.uniform_,
    "normal": init.normal_,
    "ones": init.ones_,
    "zeros": init.zeros_,
    "scaled_uniform": _ku({"a": math.sqrt(5)}),           # U(+-sqrt(1/fan_in))
    "scaled_normal": _ku({"nonlinearity": "linear"}),      # (sic) the reference maps this name to a uniform draw
    "lecun_uniform": _ku({"nonlinearity": "linear"}),
    "lecun_normal": _kn({"nonlinearity": "linear"}),
    "he_uniform": _ku({}),
    "he_normal": _kn({}),
    "xavier_uniform": init.xavier_uniform_,
    "xavier_normal": init.xavier_normal_,
    "normal_02": lambda t: init.normal_(t, mean=0.0, std=0.02),
}


def apply_init(tensor, spec):
    """spec: "default" (keep torch's), a registry name, or {"class": name, "params": {...}}."""
    if spec == "default" or tensor is None:
        return
    if isinstance(spec, dict):
        init_dict[spec["class"]](tensor, **spec["params"])
    else:
        init_dict[spec](tensor)
