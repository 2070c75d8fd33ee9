"""Analytic sky dome (mirror of the reference's ``sky/Sky.py``, the Hosek-Wilkie full-spectral model): the host half -- loading
the coefficient tables (:55-83) and ``update`` (:100-172), which folds turbidity, ground albedo and solar elevation into nine
configuration values and one radiance per 40 nm band, in Python floats as the reference does.  ``get_solar_radiance`` (:232-264)
runs on the device (csrc/tirt_spectral.h); the tables for the sun's disc (``data_solar``, ``data_dark``) are loaded for
completeness -- the reference's ``get_solar_radiance`` has the direct-sun term commented out (:260)."""
import math
import os

import numpy as np

MATH_PI = 3.141592653589793
LAMDDA_DIV = 11
ALBEDO_NUM = 2
TURB_NUM = 10
THETA_NUM = 9
GAMMA_NUM = 6
PIECES = 45
ORDER = 4
DATA_NUM = TURB_NUM * ALBEDO_NUM * THETA_NUM * GAMMA_NUM
RAD_NUM = TURB_NUM * ALBEDO_NUM * 6
SOLAR_NUM = TURB_NUM * PIECES * ORDER
DARK_NUM = 6
MIN_LAMBDA = 320.0
MAX_LAMBDA = 720.0

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "sky")


def _load(name, cols):
    out = np.zeros((LAMDDA_DIV, cols), np.float32)
    i = 0
    for line in open(os.path.join(_DIR, name), "r"):
        values = line.split(',', cols)
        for j in range(cols):
            out[i, j] = values[j]
        i += 1
    return out


class Sky:
    def __init__(self, turbidity=3.0, albedo=0.5, elevation=10.0 * MATH_PI / 180.0):
        self.configs_np = np.zeros((LAMDDA_DIV, THETA_NUM), np.float32)
        self.radiances_np = np.zeros((LAMDDA_DIV), np.float32)
        self.sun_dir_np = np.zeros((1, 3), np.float32)
        self.turbidity = turbidity
        self.solar_radius = 0.51 * MATH_PI / 180.0 / 2.0
        self.albedo = albedo
        self.elevation = elevation
        self.data_np = _load("data.csv", DATA_NUM)
        self.data_rad_np = _load("data_rad.csv", RAD_NUM)
        self.data_solar_np = _load("data_solar.csv", SOLAR_NUM)
        self.data_dark_np = _load("data_dark.csv", DARK_NUM)

    def setup_data_gpu(self):
        self.update()
        self.sun_dir_np[0, 0] = 0.0
        self.sun_dir_np[0, 1] = math.sin(self.elevation)
        self.sun_dir_np[0, 2] = math.cos(self.elevation)

    @staticmethod
    def formula(t, A0, A1, A2, A3, A4, A5):
        return pow(1.0 - t, 5.0) * A0 + 5.0 * pow(1.0 - t, 4.0) * t * A1 + \
            10.0 * pow(1.0 - t, 3.0) * pow(t, 2.0) * A2 + 10.0 * pow(1.0 - t, 2.0) * pow(t, 3.0) * A3 + \
            5.0 * (1.0 - t) * pow(t, 4.0) * A4 + pow(t, 5.0) * A5

    def update(self):
        albedo = self.albedo
        int_turbidity = int(self.turbidity)
        turbidity_rem = self.turbidity - float(int_turbidity)
        solar_elevation = pow(self.elevation / (MATH_PI / 2.0), (1.0 / 3.0))
        d, r, F = self.data_np, self.data_rad_np, self.formula

        def cfg(j, i, index):
            return F(solar_elevation, d[j, index + i], d[j, index + i + 9], d[j, index + i + 18], d[j, index + i + 27], d[j, index + i + 36], d[j, index + i + 45])

        def rad(i, index):
            return F(solar_elevation, r[i, index + 0], r[i, index + 1], r[i, index + 2], r[i, index + 3], r[i, index + 4], r[i, index + 5])

        # configs: the four corners (albedo 0/1) x (turbidity floor/ceil); every += rounds to float32 like the reference's numpy cells
        index = 9 * 6 * (int_turbidity - 1)
        for j in range(LAMDDA_DIV):
            for i in range(THETA_NUM):
                self.configs_np[j, i] = (1.0 - albedo) * (1.0 - turbidity_rem) * cfg(j, i, index)
        index = 9 * 6 * 10 + 9 * 6 * (int_turbidity - 1)
        for j in range(LAMDDA_DIV):
            for i in range(THETA_NUM):
                self.configs_np[j, i] += (albedo) * (1.0 - turbidity_rem) * cfg(j, i, index)
        if int_turbidity < 10:
            index = 9 * 6 * int_turbidity
            for j in range(LAMDDA_DIV):
                for i in range(THETA_NUM):
                    self.configs_np[j, i] += (1.0 - albedo) * (turbidity_rem) * cfg(j, i, index)
            index = 9 * 6 * 10 + 9 * 6 * (int_turbidity)
            for j in range(LAMDDA_DIV):
                for i in range(THETA_NUM):
                    self.configs_np[j, i] += (albedo) * (turbidity_rem) * cfg(j, i, index)
        # radiances
        index = 6 * (int_turbidity - 1)
        for i in range(LAMDDA_DIV):
            self.radiances_np[i] = (1.0 - albedo) * (1.0 - turbidity_rem) * rad(i, index)
        index = 6 * 10 + 6 * (int_turbidity - 1)
        for i in range(LAMDDA_DIV):
            self.radiances_np[i] += (albedo) * (1.0 - turbidity_rem) * rad(i, index)
        if int_turbidity < 10:
            index = 6 * int_turbidity
            for i in range(LAMDDA_DIV):
                self.radiances_np[i] += (1.0 - albedo) * (turbidity_rem) * rad(i, index)
            index = 6 * 10 + 6 * (int_turbidity)
            for i in range(LAMDDA_DIV):
                self.radiances_np[i] += (albedo) * (turbidity_rem) * rad(i, index)
