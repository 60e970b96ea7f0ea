"""SCENE-GEN v1 — deterministic synthetic scenes for configs C3/C4/C5. This file IS the definition.

PRNG: PCG32 XSH-RR (multiplier 6364136223846793005, stream 54), seed 42.  u() = (next_u32() >> 8) * 2^-24 as f32.
All derived values are computed in float32 in the order written below.
"""
from __future__ import annotations

import numpy as np

from . import capi

_MASK64 = (1 << 64) - 1
_MULT = 6364136223846793005

F = np.float32


class PCG32:
    def __init__(self, seed: int = 42, stream: int = 54):
        self.state = 0
        self.inc = ((stream << 1) | 1) & _MASK64
        self.next_u32()
        self.state = (self.state + seed) & _MASK64
        self.next_u32()

    def next_u32(self) -> int:
        old = self.state
        self.state = (old * _MULT + self.inc) & _MASK64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def u(self) -> np.float32:
        return F(self.next_u32() >> 8) * F(2.0 ** -24)


def generate(width: int, height: int, n_spheres: int, n_lights: int, seed: int = 42):
    """Returns (FlatObjects, FlatLights). Point lights only, intensity 20/L, white."""
    rng = PCG32(seed)
    ar = F(width) / F(height)
    scale = np.sqrt(F(1024.0) / F(n_spheres)).astype(F)
    cx = np.zeros(n_spheres, F); cy = np.zeros(n_spheres, F); cz = np.zeros(n_spheres, F); rad = np.zeros(n_spheres, F)
    rr = np.zeros(n_spheres, F); rg = np.zeros(n_spheres, F); rb = np.zeros(n_spheres, F)
    for k in range(n_spheres):
        z = -(F(4.0) + F(26.0) * rng.u())
        az = np.abs(z)
        x = ((F(2.0) * rng.u() - F(1.0)) * F(0.55)) * az * ar
        y = ((F(2.0) * rng.u() - F(1.0)) * F(0.55)) * az
        r = ((F(0.02) * az) * (F(0.5) + rng.u())) * scale
        cx[k], cy[k], cz[k], rad[k] = x, y, z, r
        rr[k] = F(0.2) + F(0.8) * rng.u()
        rg[k] = F(0.2) + F(0.8) * rng.u()
        rb[k] = F(0.2) + F(0.8) * rng.u()
    px = np.zeros(n_lights, F); py = np.zeros(n_lights, F); pz = np.zeros(n_lights, F)
    for m in range(n_lights):
        px[m] = F(12.0) * rng.u() - F(6.0)
        py[m] = F(12.0) * rng.u() - F(6.0)
        pz[m] = -(F(12.0) * rng.u())
    one = np.ones(n_lights, F)
    inten = np.full(n_lights, F(20.0) / F(max(n_lights, 1)), F)
    objs = capi.FlatObjects(cx, cy, cz, rad, rr, rg, rb)
    lights = capi.FlatLights(px, py, pz, one, one.copy(), one.copy(), inten, kind=np.zeros(n_lights, np.uint8))
    return objs, lights


# BASELINE.json configs: name -> (W, H, N spheres, L lights)
CONFIGS = {
    "C3": (3840, 2160, 1024, 4),
    "C4": (7680, 4320, 8192, 8),
    "C5": (3840, 2160, 4096, 32),
}


def config(name: str):
    w, h, n, l = CONFIGS[name]
    objs, lights = generate(w, h, n, l)
    return w, h, objs, lights
