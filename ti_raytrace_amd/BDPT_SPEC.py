"""Spectral bidirectional path tracer (mirror of the reference's ``integrator/BDPT_SPEC.py``).

``BDPT.render()`` = one ``@ti.kernel render`` of the reference (:660-691): BDPT_RGB's eye and light sub-paths and (e, l) connections
carrying ONE wavelength per pixel sample -- reflectances and emitters through the RGB -> spectrum table (:134-155), dispersive glass
(``Glass.sample_lambda``), every connection splatted through the CIE observer (``AddSplat``, :178-181).  The host tables (observer,
D65 normalised to Y = 1, Rgb2Spec) are PT_Spec's; device side: ``csrc/tirt_bdpt.hip`` (``template <bool SPEC>``) through
``tirt_bdpt_spec_render``.  Used by ``example/prism_rainbow.py`` (``scenes.prism_rainbow``).
"""
from . import PT_Spec

MAX_DEPTH = 5
EYE_MAX_DEPTH = MAX_DEPTH + 2
LIGHT_MAX_DEPTH = MAX_DEPTH + 1


class BDPT(PT_Spec.PathTrace):
    def setup_data_gpu(self):
        self.scene.ctx.set_option("bdpt_stack_size", max(16, int(self.stack_size)))
        PT_Spec.PathTrace.setup_data_gpu(self)

    def render(self):
        self.scene.ctx.bdpt_spec_render(self.cam.frame, 1, self.seed)

    def render_frames(self, count):
        self.scene.ctx.bdpt_spec_render(self.cam.frame, count, self.seed)
