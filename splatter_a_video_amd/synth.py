"""Seeded synthetic Gaussians x frames: the workload bench.py and the parity
tests share.  numpy only; no oracle, no torch."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class SynthScene:
    N: int
    W: int
    H: int
    F: int                     # frames in the clip
    xyz: np.ndarray            # [N,3] base positions (camera space for the ortho identity camera)
    phase: np.ndarray          # [N] per-Gaussian motion phase
    scale: np.ndarray          # [N,3] activated scales
    rotate: np.ndarray         # [N,4] unit quaternions (r,x,y,z)
    opacity: np.ndarray        # [N,1] in (0,1)
    shs: np.ndarray            # [N,16,3]
    feature: Optional[np.ndarray]  # [N,C] or None
    intr: np.ndarray           # [4] fx,fy,cx,cy
    extr: np.ndarray           # [4,4] row-major w2c
    ortho: bool
    bg: float = 0.0

    def positions(self, f: int) -> np.ndarray:
        """xy += 0.05*sin(2*pi*f/F + phase_i) (dynamic Gaussians, frame f)."""
        p = self.xyz.copy()
        d = (0.05 * np.sin(2.0 * math.pi * (f / float(self.F)) + self.phase)).astype(np.float32)
        p[:, 0] += d
        p[:, 1] += d
        return p


def make_scene(N: int, W: int, H: int, F: int = 50, C: int = 0, seed: int = 1234,
               sigma_px: float = 2.0, ortho: bool = True, clustered: float = 0.0, cluster_area: float = 0.1,
               blobs: int = 6) -> SynthScene:
    """``clustered`` > 0: that fraction of the Gaussians sits inside ``blobs`` discs that together cover ``cluster_area`` of
    the image (foreground objects of a DAVIS clip), the rest is spread uniformly -- long tile lists next to short ones."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-1.0, 1.0, size=(N, 2))
    if clustered > 0.0:
        crng = np.random.default_rng(seed + 7919)          # (separate stream: the uniform scene's draws stay what they were)
        r = math.sqrt(cluster_area * 4.0 / (blobs * math.pi))   # NDC radius of one disc: blobs * pi r^2 = area * 4
        centres = crng.uniform(-1.0 + r, 1.0 - r, size=(blobs, 2))
        n_in = int(round(clustered * N))
        which = crng.integers(0, blobs, size=n_in)
        rad = r * np.sqrt(crng.uniform(0.0, 1.0, size=n_in))
        ang = crng.uniform(0.0, 2.0 * math.pi, size=n_in)
        members = crng.permutation(N)[:n_in]
        xy[members] = centres[which] + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
    z = rng.uniform(0.1, 1.0, size=(N, 1))
    scale = np.exp(rng.normal(math.log(2.0 * sigma_px / W), 0.5, size=(N, 3)))
    q = rng.normal(0.0, 1.0, size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opacity = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, size=(N, 1))))
    shs = rng.normal(0.0, 0.3, size=(N, 16, 3))
    shs[:, 0, :] += 1.0
    phase = rng.uniform(0.0, 2.0 * math.pi, size=(N,))
    feature = rng.uniform(0.0, 1.0, size=(N, C)).astype(np.float32) if C > 0 else None
    extr = np.eye(4, dtype=np.float32)
    if ortho:
        xyz = np.concatenate([xy, z], axis=1)
        intr = np.array([W / 2.0, H / 2.0, W / 2.0, H / 2.0], np.float32)
    else:
        # pinhole camera looking down +z: fx = fy = W/2 (fovX = 90 deg); points at depth 2..6 spread so
        # that they cover the image; world scale grows with depth to keep ~sigma_px pixels.
        zc = rng.uniform(2.0, 6.0, size=(N, 1))
        xyz = np.concatenate([xy * zc * np.array([[1.0, H / float(W)]]), zc], axis=1)
        scale = scale * zc
        intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0], np.float32)
        extr[:3, :3] = _rot_xyz(0.05, -0.03, 0.02)
        extr[:3, 3] = np.array([0.1, -0.05, 0.2], np.float32)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return SynthScene(N=N, W=W, H=H, F=F, xyz=f32(xyz), phase=f32(phase), scale=f32(scale), rotate=f32(q),
                      opacity=f32(opacity), shs=f32(shs), feature=feature, intr=intr, extr=extr, ortho=ortho)


def _rot_xyz(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)
