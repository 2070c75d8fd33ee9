"""Unidirectional path tracer (mirror of the reference's ``integrator/PT_RGB.py``).

``PathTrace.render()`` = one ``@ti.kernel render`` launch of the reference (one sample per
pixel at frame ``cam.frame``, running-mean film, :44-136).  On the device it is not a
per-pixel megakernel but a wavefront pipeline (generate -> trace -> shade -> shadow-trace ->
film) over struct-of-arrays path queues in HBM -- see DESIGN.md.
"""
import numpy as np

from .Scene import DeviceField

MAX_DEPTH = 15           # integrator/PT_RGB.py:21


def default_tile_size(H):
    """Pixels per tile of the multi-GPU film split (tiles go round-robin over the ranks).  8 whole columns when that is a handy size:
    the device library then walks a tile in 8 x 8 pixel blocks (tirt_internal.h, local_to_pixel), which makes the 64 camera rays
    of a wave a compact bundle; otherwise 4096 linear pixels."""
    return 8 * H if (H % 8 == 0 and 4096 <= 8 * H <= 16384) else 4096


class PathTrace:
    def __init__(self, imgSizeX, imgSizeY, cam, scene, stack_size,
                 seed=1, tile_rank=0, tile_count=1, tile_size=None, flags=0):
        self.imgSizeX = imgSizeX
        self.imgSizeY = imgSizeY
        self.cam = cam
        self.scene = scene
        self.stack_size = stack_size
        # extensions: counter-based RNG seed (the reference's ti.random() is unseeded) and
        # the pixel-tile shard this context renders (multi-GPU)
        self.seed = seed
        self.tile_rank, self.tile_count, self.tile_size = tile_rank, tile_count, tile_size or default_tile_size(imgSizeY)
        self.flags = flags
        self.hdr = DeviceField("hdr", scene, lambda: self._download(True))
        self.rgb_film = DeviceField("rgb_film", scene, lambda: self._download(False))

    def _download(self, hdr):
        h, r = self.scene.ctx.film_download(self.imgSizeX, self.imgSizeY, want_hdr=hdr, want_rgb=not hdr)
        return h if hdr else r

    def setup_data_cpu(self):
        pass                                  # field placement has no host-side equivalent

    def setup_data_gpu(self):
        self.scene.ctx.film_create(self.imgSizeX, self.imgSizeY, self.tile_rank, self.tile_count, self.tile_size)
        self.cam.attach(self.scene.ctx)

    def render(self):
        """One frame at ``cam.frame`` (the caller advances it with ``cam.update_frame()``)."""
        self.scene.ctx.pt_rgb_render(self.cam.frame, 1, self.seed, MAX_DEPTH, self.stack_size, self.flags)

    def render_frames(self, count):
        """Extension: ``count`` consecutive frames starting at ``cam.frame`` in one call
        (identical film to calling render()/update_frame() ``count`` times)."""
        self.scene.ctx.pt_rgb_render(self.cam.frame, count, self.seed, MAX_DEPTH, self.stack_size, self.flags)
