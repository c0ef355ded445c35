"""Synthetic image pairs for parity tests and ``bench.py``.

The reference's dataset is not distributed, so every measured configuration uses a
seeded synthetic pair: a max-entropy uint8 "other" image, a fixed global homography
with image-size-scaled perspective terms, uniform float32 keypoints and a smooth
parallax field plus noise so that the local homographies differ from cell to cell.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .geometry import final_size, get_mesh, get_vertice

#: name -> (width, height, keypoints, mesh cells per side, seed).  The sizes and seeds
#: are the ones BASELINE.json / SURVEY.md 8(d) fix.
CONFIGS = {
    "C1": (768, 768, 150, 20, 31),
    "C2": (1920, 1080, 500, 100, 1080),
    "C3": (3840, 2160, 2000, 200, 2160),
    "C4": (7680, 4320, 5000, 400, 4320),
    "C5": (3840, 2160, 2000, 100, 6400),
}


@dataclass
class Pair:
    img: np.ndarray | None      # (H, W, 3) uint8 "other" image (None when with_image=False)
    shape: tuple                # (H, W, 3)
    src: np.ndarray             # (N, 2) float32 keypoints in the other image
    dst: np.ndarray             # (N, 2) float32 keypoints in the centre image
    Hg: np.ndarray              # (3, 3) float64 global homography other -> centre
    final_w: int
    final_h: int
    off_x: int
    off_y: int
    mesh: np.ndarray            # (2, m + 1) float64 cell edges
    vertices: np.ndarray        # (m, m, 2) float64 cell sample points
    gamma: float = 0.5
    sigma: float = 100.0


class _Shape:
    def __init__(self, shape):
        self.shape = shape


def global_h(width, height):
    return np.array([[1.02, 0.01, 0.016 * width],
                     [-0.015, 0.99, 0.011 * height],
                     [1e-5 * 1920 / width, -2e-5 * 1920 / width, 1.0]])


def synth_pair(width, height, n, mesh_cells, seed, with_image=True, gamma=0.5, sigma=100.0):
    rng = np.random.default_rng(seed)
    shape = (height, width, 3)
    # the image is always drawn so that keypoints do not depend on with_image
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if not with_image:
        img = None
    Hg = global_h(width, height)
    src = (rng.random((n, 2)) * [width, height]).astype(np.float32)
    s = src.astype(np.float64)
    q = np.concatenate([s, np.ones((n, 1))], axis=1) @ Hg.T
    proj = q[:, :2] / q[:, 2:3]
    dst = proj + (3.0 * width / 3840.0) * np.sin(s * 9.6 / width) + rng.normal(0.0, 0.5, (n, 2))
    dst = dst.astype(np.float32)
    fw, fh, ox, oy = (int(v) for v in final_size(_Shape(shape), _Shape(shape), Hg))
    mesh = get_mesh((fw, fh), mesh_cells + 1)
    vertices = get_vertice((fw, fh), mesh_cells, (ox, oy))
    return Pair(img, shape, src, dst, Hg, fw, fh, ox, oy, mesh, vertices, gamma, sigma)


def config_pair(name, with_image=True, seed_offset=0):
    w, h, n, m, seed = CONFIGS[name]
    return synth_pair(w, h, n, m, seed + seed_offset, with_image=with_image)
