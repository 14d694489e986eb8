"""Stand-ins for optional third-party packages the reference's config files import but the hot path does not need (SURVEY section 7 "Environment").

`ensure_torchvision()` registers a minimal `torchvision` (only `transforms.RandomCrop / RandomHorizontalFlip / CenterCrop / Compose / ConvertImageDtype /
Grayscale`, the names configs/LRS23/AV/EffConfInterCTC.py:8,83-89 and nnet/datasets.py:187-196 use) when the real package is not installed; with the real
package present it does nothing."""
import importlib
import sys


def ensure_torchvision():
    try:
        importlib.import_module("torchvision")
        return False
    except Exception:
        from . import torchvision_fallback as tv
        sys.modules["torchvision"] = tv.build()
        sys.modules["torchvision.transforms"] = sys.modules["torchvision"].transforms
        return True
