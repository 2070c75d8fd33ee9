"""Pinhole orbit camera (mirror of the reference's ``Camera.py``).

Host side only: the reference computes eye / view / view_inv with numpy on the host
(``Camera.py:70-93``) and uploads them; ray generation (``get_ray_direction`` :130-142)
runs inside the device kernels from the same numbers.  The same matrices feed the HIP path
(``tirt_camera_set``) and, in tests, the CPU oracle.
"""
import math

import numpy as np

FULL_HGT = 2.4          # Camera.py:9


class Camera:
    def __init__(self, sizex, sizey, sample_count):
        self.wid = sizex
        self.hgt = sizey
        self.focal = 2.0
        self.ratio = sizex / sizey
        # Camera.py:31-34: fx is derived from the width only (quirk B5)
        self.fx = self.focal * sizex / FULL_HGT
        self.fy = self.fx
        self.cx = sizex * 0.5
        self.cy = sizey * 0.5

        self.eye_np = np.ones(shape=(1, 3), dtype=np.float32)
        self.target = np.array([0.0, 0.0, 0.0])
        self.up = np.array([0.0, 1.0, 0.0])
        self.view_np = np.zeros((1, 4, 4), dtype=np.float32)
        self.view_inv_np = np.zeros((1, 4, 4), dtype=np.float32)

        self.yaw = 0.0
        self.pitch = 0.0
        self.roll = 0.0
        self.scale = 1000.0

        # Camera.py:46-47 divides by int(sqrt(spp)) - 1 and so raises for spp < 4 (quirk B6);
        # sample_dis is never read on the PT path, so the crash is not reproduced.
        self.sample_count = int(math.sqrt(sample_count))
        self.sample_dis = 1.0 / float(self.sample_count - 1) if self.sample_count > 1 else 1.0
        self.frame_cpu = np.zeros(shape=(1), dtype=np.int32)
        self.frame = 0
        self.fps = 30.0
        self._sinks = []          # device contexts that receive updates

    # -- device plumbing -----------------------------------------------------------------
    def attach(self, ctx):
        if ctx not in self._sinks:
            self._sinks.append(ctx)
        self._push(ctx)

    def _push(self, ctx):
        ctx.camera_set(self.view_np[0], self.view_inv_np[0], self.eye_np[0], self.fx, self.fy, self.cx, self.cy)

    # -- reference API ---------------------------------------------------------------------
    def yaw_cam(self, targetx, targety, targetz):
        self.target[:] = (targetx, targety, targetz)
        if self.yaw < 3.14:
            self.set_view_point(self.yaw + 0.003, 0.0, 0.0, 3.0)

    def pitch_cam(self, targetx, targety, targetz):
        self.target[:] = (targetx, targety, targetz)
        if self.pitch < 0.5:
            self.set_view_point(0.0, self.pitch + 0.003, 0.0, 3.0)

    def update(self):
        self.pitch = max(min(self.pitch, 1.57), -1.57)
        cp, sp = math.cos(self.pitch), math.sin(self.pitch)
        cy, sy = math.cos(self.yaw), math.sin(self.yaw)
        self.eye_np[0, 0] = self.target[0] + self.scale * cp * sy
        self.eye_np[0, 1] = self.target[1] + self.scale * sp
        self.eye_np[0, 2] = self.target[2] + self.scale * cp * cy
        self.up[:] = (-sp * sy, cp, -sp * cy)

        eye = self.eye_np[0, :]
        zaxis = eye - self.target
        zaxis = zaxis / np.linalg.norm(zaxis)
        xaxis = np.cross(self.up, zaxis)
        xaxis = xaxis / np.linalg.norm(xaxis)
        yaxis = np.cross(zaxis, xaxis)
        # rows = axes, last column = -axis . eye; stored as f32, inverted in f32 (Camera.py:84-92)
        self.view_np[0] = np.array([
            [xaxis[0], xaxis[1], xaxis[2], -np.dot(xaxis, eye)],
            [yaxis[0], yaxis[1], yaxis[2], -np.dot(yaxis, eye)],
            [zaxis[0], zaxis[1], zaxis[2], -np.dot(zaxis, eye)],
            [0.0, 0.0, 0.0, 1.0]])
        self.view_inv_np = np.linalg.inv(self.view_np).astype(np.float32)
        for ctx in self._sinks:
            self._push(ctx)

    def set_view_point(self, yaw, pitch, roll, scale):
        self.pitch, self.yaw, self.roll, self.scale = pitch, yaw, roll, scale
        self.update()

    def set_target(self, targetx, targety, targetz):
        self.target[:] = (targetx, targety, targetz)
        self.update()

    def update_frame(self, count=1):
        self.frame += count
        self.frame_cpu[0] = self.frame
