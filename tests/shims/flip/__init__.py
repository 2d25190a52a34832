"""Test shim for the reference's scripts/flip package (NVIDIA FLIP error metric), imported unconditionally by scripts/common.py but
only used by compute_error("FLIP", ...), which the PSNR path of run.py never calls."""
import numpy as np
from . import utils  # noqa: F401


def color_space_transform(img, kind):
    if kind == "linrgb2srgb":
        return np.where(img <= 0.0031308, 12.92 * img, 1.055 * np.power(np.maximum(img, 1e-12), 1 / 2.4) - 0.055)
    raise NotImplementedError(f"flip shim: {kind}")


def compute_flip(ref, test, pixels_per_degree):
    raise NotImplementedError("the FLIP metric is not part of this test shim")
