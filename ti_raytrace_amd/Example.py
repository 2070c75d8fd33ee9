"""Headless mirror of the reference's ``example/Example.py`` harness (:11-59).

Same call sequence (``build_scene`` -> repeated ``render()`` until it returns 0), no
``ti.GUI`` window and no per-frame film read-back; ``out.png`` is written when
``frame == sample_count`` exactly like ``example/Example.py:48-53``.
"""
import numpy as np

from . import Camera, Scene
from . import SceneData as SCD
from . import UtilsFunc as UF


def write_png(rgb_film_np, path):
    """``ti.imwrite(field, path)`` convention: field[i, j] -> PNG (row H-1-j, col i)."""
    from PIL import Image
    img = np.clip(rgb_film_np, 0.0, 1.0)
    img = (np.transpose(img, (1, 0, 2))[::-1] * 255.0).astype(np.uint8)
    Image.fromarray(img, "RGB").save(path)


class example:
    def __init__(self, imgSizeX, imgSizeY, sample_count, device_id=None):
        self.imgSizeX = imgSizeX
        self.imgSizeY = imgSizeY
        self.sample_count = sample_count
        self.cam = Camera.Camera(imgSizeX, imgSizeY, sample_count)
        self.scene = Scene.Scene(device_id)
        self.integrator = None
        self.out_path = "out.png"
        self.exposure = 0.5                  # example/Example.py:43
        # progressive preview (example/Example.py:41-46 tone-maps and blits to a ti.GUI window every frame): headless here --
        # every `preview_every` frames the film so far is tone-mapped and written to `preview_path` (0 = no preview)
        self.preview_every = 0
        self.preview_path = "preview.png"

    def build_scene(self):
        self.scene.setup_data_cpu()
        self.integrator.setup_data_cpu()
        self.integrator.setup_data_gpu()
        self.scene.setup_data_gpu()
        # hint for the device library: this job renders sample_count frames (bounds its batch buffers)
        self.scene.ctx.set_option("job_frames", max(int(self.sample_count), 1))

    def add_sphere_light(self, pos=(0.0, 20.0, 0.0), radius=5.0, emission=50.0):
        """example/Example.py:27-36 (position/size/emission overridable for scaled scenes)."""
        shape = SCD.Shape()
        shape.type = SCD.SHPAE_SPHERE
        shape.pos = [float(pos[0]), float(pos[1]), float(pos[2])]
        shape.setRadius(radius)
        mat = SCD.Material()
        mat.type = SCD.MAT_LIGHT
        mat.setColor([emission, emission, emission])
        self.scene.add_shape(shape, mat)

    def frame_camera(self, scale_factor=0.8):
        """example/cornell_box.py:26-30: look at the AABB centre from 0.8 x |diagonal|."""
        centre = self.scene.maxboundarynp + self.scene.minboundarynp
        size = self.scene.maxboundarynp - self.scene.minboundarynp
        import math
        self.cam.scale = math.sqrt(size[0, 0] * size[0, 0] + size[0, 1] * size[0, 1] + size[0, 2] * size[0, 2]) * scale_factor
        self.cam.set_target(centre[0, 0] * 0.5, centre[0, 1] * 0.5, centre[0, 2] * 0.5)
        self.cam.update()

    def render(self):
        if self.cam.frame_cpu[0] < self.sample_count:
            self.integrator.render()
            self.cam.update_frame()
            if self.preview_every > 0 and self.cam.frame_cpu[0] % self.preview_every == 0 and self.cam.frame_cpu[0] < self.sample_count:
                UF.tone_map(self.exposure, self.integrator.hdr, self.integrator.rgb_film)
                write_png(self.integrator.rgb_film.to_numpy(), self.preview_path)
            return 1
        if self.cam.frame_cpu[0] == self.sample_count:
            UF.tone_map(self.exposure, self.integrator.hdr, self.integrator.rgb_film)
            write_png(self.integrator.rgb_film.to_numpy(), self.out_path)
            self.cam.update_frame()
        return 0

    def render_all(self, batch=4):
        """Extension: the whole ``sample_count`` loop with ``batch`` frames per device call."""
        while self.cam.frame < self.sample_count:
            k = min(batch, self.sample_count - self.cam.frame)
            self.integrator.render_frames(k)
            self.cam.update_frame(k)
