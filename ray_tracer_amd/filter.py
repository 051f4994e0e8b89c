"""Host-side mirror of the reference's Filter / BiliteralFilter (filter.h:6-27): same member names, FilterImage runs
on the MI355X through the C ABI."""
import numpy as np


class BiliteralFilter:  # spelling as in the reference
    def __init__(self, ctx, img, width, height):
        """ctx: capi.Context; img: width*height*4 doubles (RGBA, row major) as the reference's orig_img"""
        self.ctx = ctx
        self.orig_img = np.ascontiguousarray(img, dtype=np.float64).reshape(height, width, 4)
        self.img_width, self.img_height = width, height
        self.sigma_d, self.sigma_r, self.window = 1.0, 1.0, 0
        self.result = None

    def FilterImage(self):  # filter.cpp:13-51
        self.result = self.ctx.filter_bilateral(self.orig_img, self.window, self.sigma_d, self.sigma_r).reshape(-1)
