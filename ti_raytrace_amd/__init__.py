"""ti_raytrace_amd -- MI355X-native path-tracing core behind the ti-raytrace Python API.

Modules keep the reference's names: ``SceneData``, ``Scene``, ``Camera``, ``LBvh``,
``PT_RGB``, ``UtilsFunc``, ``Texture``, ``Example`` (+ ``scenes`` with the example set-ups).
The compute path is ``csrc/libtirt.so`` (hand-written HIP for gfx950) behind the C-ABI of
``include/tirt.h``; see DESIGN.md / INTEGRATION.md.
"""
import os as _os

# Two things this package does to the PROCESS when it is imported, both before the HIP runtime initialises and both with an opt-out
# (TIRT_NO_ENV_TUNING=1 switches both off; include/tirt.h, "Embedding", says what an embedder who opts out should do instead):
#  * GPU_MAX_HW_QUEUES=8 unless the variable is set: the render lanes (up to 4 concurrent wavefront batches) want one hardware
#    queue each and the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues;
#  * _native.lib() loads PyTorch's bundled libamdhip64.so ahead of libtirt.so when torch is installed (see _native._prefer_torch_hip_runtime).
if _os.environ.get("TIRT_NO_ENV_TUNING", "0") in ("", "0"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import SceneData, UtilsFunc, Texture, Camera, LBvh, Scene, PT_RGB, BDPT_RGB, Example  # noqa: F401

__all__ = ["SceneData", "UtilsFunc", "Texture", "Camera", "LBvh", "Scene", "PT_RGB", "BDPT_RGB", "Example"]
