"""Tabulated spectrum (mirror of the reference's ``spectrum/Spectrum.py``): ``load_table`` reads ``wavelength,value`` lines
(:18-36); ``sample`` (:44-52) runs on the device.  ``sample_np`` restates it in float32 numpy for the one host-side use the
reference has -- the white point behind ``PathTrace.normalize_spec`` (integrator/PT_Spec.py:93-100, 160-176)."""
import numpy as np


class Spectrum:
    def __init__(self):
        self.lambda_min = 10000
        self.lambda_max = 0
        self.lambda_range = 0
        self.size = 0
        self.data_np = None
        self.white_point_np = np.zeros((1, 3), np.float32)

    def load_table(self, table_path):
        data = []
        for line in open(table_path, "r"):
            values = line.split(',', 2)
            v1 = float(values[1]); v0 = float(values[0])
            data.append(v1)
            if self.size == 0:
                self.lambda_min = v0
            self.lambda_max = v0
            self.size += 1
        self.data_np = np.asarray(data, dtype=np.float32)
        self.lambda_range = (self.lambda_max - self.lambda_min) / (self.size - 1)

    def setup_data_gpu(self):
        pass                                   # the tables travel together in PT_Spec.PathTrace.setup_data_gpu

    def sample_np(self, Lambda):
        """Spectrum.sample for one float32 wavelength (weight = fract(offset), as the reference has it)."""
        f = np.float32
        Lambda = f(Lambda)
        if Lambda >= f(self.lambda_min) and Lambda <= f(self.lambda_max):
            offset = f(Lambda - f(self.lambda_min))
            idx = int(f(offset / f(self.lambda_range)))
            w = f(offset - np.floor(offset))
            i1 = min(idx + 1, self.size - 1)
            return f(f(self.data_np[idx] * f(f(1.0) - w)) + f(self.data_np[i1] * w))
        return f(0.0)

    def scale(self, coff):
        self.data_np = (self.data_np * np.float32(coff)).astype(np.float32)
