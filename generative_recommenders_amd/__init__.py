"""MI355X-native HSTU hot path: drop-in for the op layer of
facebookresearch/generative-recommenders (``generative_recommenders.ops.*`` and
``generative_recommenders.modules.stu``), implemented as hand-written HIP kernels for
gfx950 behind a C ABI (include/hstu_hip.h).  See DESIGN.md / INTEGRATION.md.
"""

import os as _os

# Hosts whose driver only supports dmabuf IPC: without this RCCL's buffer exchange (and CUDA-tensor sharing between processes)
# fails in hipIpcGetMemHandle.  The HSA runtime reads it when the process first touches the GPU, so it is set when the package
# is IMPORTED -- before any torch.cuda call of the caller -- and only if the environment does not say otherwise.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from generative_recommenders_amd.common import HammerKernel  # noqa: F401,E402

__all__ = ["HammerKernel"]
