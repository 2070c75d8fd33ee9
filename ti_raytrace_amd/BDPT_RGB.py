"""Bidirectional path tracer (mirror of the reference's ``integrator/BDPT_RGB.py``).

``BDPT.render()`` = one ``@ti.kernel render`` of the reference (:595-642): per pixel an eye
sub-path (<= 7 vertices), a light sub-path (<= 6), every (e, l) connection of total depth
<= MAX_DEPTH with MIS, light-tracing contributions splatted onto other pixels, running-mean
film.  Device side: ``csrc/tirt_bdpt.hip`` through ``tirt_bdpt_rgb_render``.
"""
from .Scene import DeviceField
from .PT_RGB import default_tile_size

STOP_DEPTH = 10000
MAX_DEPTH = 5
EYE_MAX_DEPTH = MAX_DEPTH + 2
LIGHT_MAX_DEPTH = MAX_DEPTH + 1
VERTEX_NONE, VERTEX_LIGHT, VERTEX_LENS, VERTEX_SURFACE = 0, 1, 2, 3


class BDPT:
    def __init__(self, imgSizeX, imgSizeY, cam, scene, stack_size, seed=1, tile_rank=0, tile_count=1, tile_size=None):
        self.imgSizeX = imgSizeX
        self.imgSizeY = imgSizeY
        self.cam = cam
        self.scene = scene
        self.stack_size = stack_size
        self.seed = seed
        self.tile_rank, self.tile_count, self.tile_size = tile_rank, tile_count, tile_size or default_tile_size(imgSizeY)
        self.hdr = DeviceField("hdr", scene, lambda: self._download(True))
        self.rgb_film = DeviceField("rgb_film", scene, lambda: self._download(False))

    def _download(self, hdr):
        h, r = self.scene.ctx.film_download(self.imgSizeX, self.imgSizeY, want_hdr=hdr, want_rgb=not hdr)
        return h if hdr else r

    def setup_data_cpu(self):
        pass

    def setup_data_gpu(self):
        self.scene.ctx.set_option("bdpt_stack_size", max(16, int(self.stack_size)))
        self.scene.ctx.film_create(self.imgSizeX, self.imgSizeY, self.tile_rank, self.tile_count, self.tile_size)
        self.cam.attach(self.scene.ctx)

    def render(self):
        self.scene.ctx.bdpt_rgb_render(self.cam.frame, 1, self.seed)

    def render_frames(self, count):
        self.scene.ctx.bdpt_rgb_render(self.cam.frame, count, self.seed)
