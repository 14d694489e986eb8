"""avec_amd: MI355X-native (gfx950) hot path for the Audio-Visual Efficient Conformer.
   csrc/   HIP kernels + C ABI (include/avec_hip.h)        lib.py     ctypes binding (no fallback)
   ops.py  launchers + fused autograd Functions            runtime.py dtype / RNG / weight shadows / flat parameter arena
   nnet/   host-side mirror of the reference's nnet API"""
import os as _os

_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL, the SyncBatchNorm peer exchange buffers); only effective before the HIP runtime starts

from .runtime import compute_dtype, manual_seed, set_compute_dtype  # noqa: F401
