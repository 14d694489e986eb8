"""`nnet`: host-side mirror of the reference's nnet API surface for the AV Efficient Conformer hot path, executing on MI355X through
libavec_hip.so.  Flat namespace like the reference's nnet/__init__.py:19-49."""
from . import (activations, attentions, blocks, collate_fn, datasets, decoders, embeddings, initializations, layers, losses, metrics, model, models_zoo,
               module, modules, networks, normalizations, optimizers, preprocessing, schedulers, transforms)
from .activations import *      # noqa: F401,F403
from .attentions import *       # noqa: F401,F403
from .blocks import *           # noqa: F401,F403
from .collate_fn import *       # noqa: F401,F403
from .decoders import *         # noqa: F401,F403
from .embeddings import *       # noqa: F401,F403
from .initializations import *  # noqa: F401,F403
from .layers import *           # noqa: F401,F403
from .losses import *           # noqa: F401,F403
from .metrics import *          # noqa: F401,F403
from .model import Model
from .models_zoo import *       # noqa: F401,F403
from .module import Module
from .modules import *          # noqa: F401,F403
from .networks import *         # noqa: F401,F403
from .normalizations import *   # noqa: F401,F403
from .optimizers import *       # noqa: F401,F403
from .preprocessing import *    # noqa: F401,F403
from .schedulers import *       # noqa: F401,F403
from .transforms import *       # noqa: F401,F403
