"""LensCamera of the reference (camera.h:101-209), host side: the fields SetLens/set_Camera derive.

Same names and argument meaning as the reference class so that callers read the same; the values go
to the device through rt_set_camera.
"""
import math

import numpy as np

NBINS = 30


def _unit(v):
    return v / math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def _cross(a, b):  # vec3.h:126-133
    return np.array([a[1] * b[2] - a[2] * b[1], -(a[0] * b[2] - a[2] * b[0]), a[0] * b[1] - a[1] * b[0]])


class LensCamera:
    def __init__(self):
        self.film_sensitivity = 1.0  # camera.h:189-193
        self.aperture = 0.5
        self.film_width = 0.024
        self.film_height = 0.024
        self.n = np.full(NBINS, 2.0)
        self.d = 0.0
        self.lens_maker_coefficient = 0.0
        self.origin = np.zeros(3)
        self.film_lower_left_corner = np.zeros(3)
        self.horizontal = np.zeros(3)
        self.vertical = np.zeros(3)
        self.w = np.zeros(3)
        self.u = np.zeros(3)
        self.v = np.zeros(3)

    def SetLens(self, vfov, focal_length):  # camera.h:108-125
        theta = vfov * math.pi / 180
        self.d = (0.5 * self.film_height) / math.tan(theta / 2.0)
        self.lens_maker_coefficient = 1.0 / (focal_length * (self.n[NBINS // 2] - 1))

    def set_Camera(self, lookfrom, lookat, vup):  # camera.h:128-145
        lookfrom, lookat, vup = (np.asarray(x, dtype=np.float64) for x in (lookfrom, lookat, vup))
        self.origin = lookfrom.copy()
        self.w = _unit(lookfrom - lookat)
        self.u = _unit(_cross(vup, self.w))
        self.v = _cross(self.w, self.u)
        self.film_lower_left_corner = (self.origin - 0.5 * self.film_width * self.u - 0.5 * self.film_height * self.v
                                       + self.d * self.w)
        self.horizontal = self.film_width * self.u
        self.vertical = self.film_height * self.v

    def LensMakersFormula(self, wavelength):  # camera.h:197-200
        i = int(math.floor((wavelength - 400) / 10))
        return 1.0 / ((self.n[i] - 1) * self.lens_maker_coefficient)
