"""MI355X-native HSTU hot path: drop-in for the op layer of
facebookresearch/generative-recommenders (``generative_recommenders.ops.*`` and
``generative_recommenders.modules.stu``), implemented as hand-written HIP kernels for
gfx950 behind a C ABI (include/hstu_hip.h).  See DESIGN.md / INTEGRATION.md.
"""

from generative_recommenders_amd.common import HammerKernel  # noqa: F401

__all__ = ["HammerKernel"]
