"""Hero-wavelength spectral path tracer (mirror of the reference's ``integrator/PT_Spec.py``).

Same class surface as the reference (``PathTrace(imgSizeX, imgSizeY, cam, scene, stack_size)``, ``setup_data_cpu``,
``setup_data_gpu``, ``render``, fields ``hdr`` / ``rgb_film``).  On the device it is the PT_RGB wavefront with a spectral shading
kernel and film update (csrc/tirt_render.hip: ``k_shade_spec``, ``k_film_spec``); four wavelengths 100 nm apart ride on every
path.  The tables the reference reads in ``setup_data_cpu`` (:56-91) come from ``assets/spectrum`` (the reference's own data
files); the RGB -> spectrum table -- ``spectrum/spec_table``, which the reference repository lacks -- is built on the device by
``Rgb2Spec.build_table`` unless ``spec_table_path`` names a file in the reference's format."""
import os

import numpy as np

from . import HeroSample as Hero
from . import Rgb2Spec as RGB2SPEC
from . import Sky
from . import Spectrum as Spec
from .PT_RGB import default_tile_size
from .Scene import DeviceField

MAX_DEPTH = 10           # integrator/PT_Spec.py:26
_SPECTRUM_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "spectrum")
_TABLE_CACHE = {}        # (res, who built it) -> (scale, data): the optimiser's output does not depend on the scene, but a table the
                         # tests made with the oracle's generator must not stand in for the device-built one (or the reverse) in one process


class PathTrace:
    def __init__(self, imgSizeX, imgSizeY, cam, scene, stack_size, seed=1, tile_rank=0, tile_count=1, tile_size=None, flags=0,
                 spec_table_path=None):
        self.imgSizeX = imgSizeX
        self.imgSizeY = imgSizeY
        self.lambda_min = 10000
        self.lambda_max = 0
        self.lambda_range = 0
        self.size = 0
        self.d65 = Spec.Spectrum()
        self.white = Spec.Spectrum()
        self.red = Spec.Spectrum()
        self.green = Spec.Spectrum()
        self.rgb2spec = RGB2SPEC.Rgb2Spec()
        self.sky = Sky.Sky(3.0, 0.5, 0.17)          # :52
        self.cam = cam
        self.scene = scene
        self.stack_size = stack_size
        self.seed = seed
        self.tile_rank, self.tile_count, self.tile_size = tile_rank, tile_count, tile_size or default_tile_size(imgSizeY)
        self.flags = flags
        self.spec_table_path = spec_table_path
        self.hdr = DeviceField("hdr", scene, lambda: self._download(True))
        self.rgb_film = DeviceField("rgb_film", scene, lambda: self._download(False))

    def _download(self, hdr):
        h, r = self.scene.ctx.film_download(self.imgSizeX, self.imgSizeY, want_hdr=hdr, want_rgb=not hdr)
        return h if hdr else r

    # -- integrator/PT_Spec.py:56-91 -----------------------------------------------------------------------------
    def setup_data_cpu(self):
        data = []
        for line in open(os.path.join(_SPECTRUM_DIR, "ciexyz31_1.csv"), "r"):
            values = line.split(',', 4)
            data.append((float(values[1]), float(values[2]), float(values[3])))
            Lambda = float(values[0])
            if self.size == 0:
                self.lambda_min = Lambda
            self.lambda_max = Lambda
            self.size += 1
        self.lambda_range = (self.lambda_max - self.lambda_min) / (self.size - 1)
        self.data_np = np.asarray(data, dtype=np.float32)
        self.d65.load_table(os.path.join(_SPECTRUM_DIR, "Illuminantd65.csv"))
        self.red.load_table(os.path.join(_SPECTRUM_DIR, "red-spec.csv"))
        self.green.load_table(os.path.join(_SPECTRUM_DIR, "green-spec.csv"))
        self.white.load_table(os.path.join(_SPECTRUM_DIR, "white-spec.csv"))
        if self.spec_table_path:
            self.rgb2spec.load_table(self.spec_table_path)

    # -- :160-176 cal_white_point, :93-100 normalize_spec.  The reference accumulates with float32 atomics in thread order; here
    #    the 471 terms are added in index order, in float32 -----------------------------------------------------------------
    def cal_white_point(self, spec):
        f = np.float32
        wp = np.zeros(3, np.float32)
        for i in range(self.size):
            Lambda = f(f(self.lambda_min) + f(f(i) * f(self.lambda_range)))
            h = float(self.lambda_max - self.lambda_min) / float(self.size - 1)
            weight = 3.0 / 8.0 * h
            if (i == 0) | (i == self.size - 1):
                weight = weight
            elif (i - 1) % 3 == 2:
                weight = weight * 2.0
            else:
                weight = weight * 3.0
            wp = (wp + (self.data_np[i] * spec.sample_np(Lambda)) * f(weight)).astype(np.float32)
        spec.white_point_np[0] = wp

    def normalize_spec(self, spec):
        self.cal_white_point(spec)
        coff = 1.0 / float(spec.white_point_np[0, 1])
        spec.scale(coff)

    def d65_from_360(self):
        """D65 at 360..830 nm in 1 nm steps, as spectrum/JakobSpecTable.py:395-402 reads it (BEFORE normalize_spec scales it)."""
        lam = self.d65.lambda_min + np.arange(self.d65.size) * self.d65.lambda_range
        return np.ascontiguousarray(self._d65_raw[lam >= 360.0][:471], dtype=np.float32)

    def setup_tables(self, build_table):
        """The host half of setup_data_gpu (sky configuration, the RGB -> spectrum table, D65 normalised to Y = 1).
        build_table(res, cie_xyz[471,3], d65[471]) -> (scale, data): the device's tirt_spec_table_build in the product, the
        oracle's generator in the CPU tests."""
        self.sky.setup_data_gpu()
        first = not hasattr(self, "_d65_raw")
        if first:
            self._d65_raw = self.d65.data_np.copy()          # D65 as read, before normalize_spec scales it (a second call must not take the scaled one)
        if self.rgb2spec.table_data_np is None:
            who = getattr(build_table, "__self__", None)
            # the device's generator (a bound method of a Context: any context builds the same bits) is one key; every other callable is its own
            # (two lambdas defined in one function share a __qualname__ -- ADVICE r4: the key holds the function object itself, which also keeps it alive)
            key = (64, type(who).__name__ if who is not None else getattr(build_table, "__func__", build_table))
            if key not in _TABLE_CACHE:
                _TABLE_CACHE[key] = build_table(64, self.data_np, self.d65_from_360())
            self.rgb2spec.table_res, self.rgb2spec.table_size = 64, 64 * 64 * 64 * 9
            self.rgb2spec.table_scale_np, self.rgb2spec.table_data_np = _TABLE_CACHE[key]
        if first:
            self.normalize_spec(self.d65)

    def setup_data_gpu(self):
        ctx = self.scene.ctx
        ctx.film_create(self.imgSizeX, self.imgSizeY, self.tile_rank, self.tile_count, self.tile_size)
        self.cam.attach(ctx)
        self.setup_tables(ctx.spec_table_build)
        ctx.spectral_upload(self.tables())

    def tables(self):
        """Everything the device (and, in tests, the CPU oracle) needs, as plain float32 arrays and numbers."""
        spds = [self.d65, self.white, self.red, self.green]
        return {
            "sensor": np.ascontiguousarray(self.data_np.reshape(-1)), "n_sensor": self.size,
            "s_min": float(self.lambda_min), "s_max": float(self.lambda_max), "s_range": float(self.lambda_range),
            "spd": np.ascontiguousarray(np.concatenate([s.data_np for s in spds]).astype(np.float32)),
            "spd_n": [s.size for s in spds], "spd_min": [float(s.lambda_min) for s in spds],
            "spd_max": [float(s.lambda_max) for s in spds], "spd_range": [float(s.lambda_range) for s in spds],
            "tbl_scale": np.ascontiguousarray(self.rgb2spec.table_scale_np), "tbl_data": np.ascontiguousarray(self.rgb2spec.table_data_np),
            "tbl_res": self.rgb2spec.table_res,
            "sky_cfg": np.ascontiguousarray(self.sky.configs_np.reshape(-1)), "sky_rad": np.ascontiguousarray(self.sky.radiances_np),
            "sun_dir": [float(x) for x in self.sky.sun_dir_np[0]],
        }

    def render(self):
        self.scene.ctx.pt_spec_render(self.cam.frame, 1, self.seed, MAX_DEPTH, self.stack_size, self.flags)

    def render_frames(self, count):
        self.scene.ctx.pt_spec_render(self.cam.frame, count, self.seed, MAX_DEPTH, self.stack_size, self.flags)
