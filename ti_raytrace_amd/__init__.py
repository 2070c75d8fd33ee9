"""ti_raytrace_amd -- MI355X-native path-tracing core behind the ti-raytrace Python API.

Modules keep the reference's names: ``SceneData``, ``Scene``, ``Camera``, ``LBvh``,
``PT_RGB``, ``UtilsFunc``, ``Texture``, ``Example`` (+ ``scenes`` with the example set-ups).
The compute path is ``csrc/libtirt.so`` (hand-written HIP for gfx950) behind the C-ABI of
``include/tirt.h``; see DESIGN.md / INTEGRATION.md.
"""
import os as _os

# The render lanes (4 concurrent wavefront batches) want one hardware queue each;
# the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues.  Must be set
# before the HIP runtime initialises, hence here; an explicit user setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import SceneData, UtilsFunc, Texture, Camera, LBvh, Scene, PT_RGB, BDPT_RGB, Example  # noqa: F401

__all__ = ["SceneData", "UtilsFunc", "Texture", "Camera", "LBvh", "Scene", "PT_RGB", "BDPT_RGB", "Example"]
