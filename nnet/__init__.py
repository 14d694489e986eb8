"""`import nnet` resolves to the MI355X-native implementation (avec_amd.nnet) so that reference configs
(`configs/LRS23/AV/EffConfInterCTC.py`: `import nnet`) run unchanged."""
import sys

import avec_amd.nnet as _impl

sys.modules[__name__] = _impl
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("avec_amd.nnet."):
        sys.modules["nnet." + _name[len("avec_amd.nnet."):]] = _mod
