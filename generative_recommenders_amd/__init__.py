"""MI355X-native HSTU hot path: drop-in for the op layer of
facebookresearch/generative-recommenders (``generative_recommenders.ops.*`` and
``generative_recommenders.modules.stu``), implemented as hand-written HIP kernels for
gfx950 behind a C ABI (include/hstu_hip.h).  See DESIGN.md / INTEGRATION.md.
"""

# (Importing the package does not touch the process environment.  Multi-process jobs on hosts whose driver only supports
# dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 before the first GPU call: data_parallel.init_from_env and bench.py set it,
# INTEGRATION.md section 6 says when an embedding application has to.)
from generative_recommenders_amd.common import HammerKernel  # noqa: F401

__all__ = ["HammerKernel"]
