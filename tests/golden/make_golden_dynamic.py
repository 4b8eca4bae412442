"""Golden vectors for row a15 (per-frame dynamic Gaussian evaluation) from the reference's own methods
DynamicGaussianWithBasePointCloud.get_position / get_rotation / get_opacity / get_scaling
(reference: src/dynamic_gaussian_with_base_point_cloud.py:171-198,236-250), called unbound on a
SimpleNamespace that carries exactly the attributes those methods read.  Build container only
(the reference tree is not present on the GPU box); the .npz it writes is what travels.

    python tests/golden/make_golden_dynamic.py
"""
from __future__ import annotations

import importlib.util
import math
import os
import sys
import types
from dataclasses import dataclass

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference_class():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls

    class PointCloud(torch.nn.Module):
        @dataclass
        class Config:
            pass

    dummy = lambda *a, **k: None
    mod("pytorch_msssim", ms_ssim=dummy)
    mod("imageio")
    mod("pointrix")
    mod("pointrix.point_cloud", PointCloud=PointCloud, POINTSCLOUD_REGISTRY=_Reg())
    mod("pointrix.point_cloud.utils", get_random_feauture=dummy, get_random_points=dummy)
    mod("pointrix.utils")
    mod("pointrix.utils.gaussian_points")
    mod("pointrix.utils.gaussian_points.gaussian_utils", build_covariance_from_scaling_rotation=dummy,
        inverse_sigmoid=dummy, gaussian_point_init=dummy)
    mod("pointrix.utils.dataset")
    mod("pointrix.utils.dataset.dataset_utils", fov2focal=dummy, focal2fov=dummy)
    mod("pointrix.dataset")
    mod("pointrix.dataset.base_data", SimplePointCloud=object)
    spec = importlib.util.spec_from_file_location(
        "ref_dyn", os.path.join(REF, "dynamic_gaussian_with_base_point_cloud.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.DynamicGaussianWithBasePointCloud


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    cls = _load_reference_class()
    rng = np.random.default_rng(2024)
    N, T = 400, 50                                  # Gaussians, frames of the clip
    I = math.ceil(T / 5)                            # one spline segment every 5 frames
    intervals_idx = torch.linspace(0, T - 1, I + 1).long()
    intervals = intervals_idx / (T - 1)             # float32 knots in [0,1]
    f32 = lambda *s, scale=1.0: torch.tensor(rng.normal(0, scale, size=s).astype(np.float32))
    P = dict(position=f32(N, 3), pos_cubic_node=f32(N, 4 * I * 3, scale=0.1), rotation=f32(N, 4),
             rot_poly_feat=f32(N, 4, 4, scale=0.05), rot_fourier_feat=f32(N, 8, 4, scale=0.05),
             opacity=f32(N, 1, scale=1.5), scaling=f32(N, 3, scale=0.5) - 4.0)
    out = {k: v.numpy() for k, v in P.items()}
    out.update(T=np.int32(T), I=np.int32(I), intervals=intervals.numpy(), start_frame_id=np.int32(0),
               time_len=np.int32(T - 1))
    times = [0, 1, 5, 24, 25, 44, 45, 49]           # spline knots, interior frames, first and last frame
    out["times"] = np.array(times, np.int32)
    for t in times:
        leaf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        ns = types.SimpleNamespace(
            delta_position=[None] * T, interval_num=I, intervals=intervals, start_frame_id=0, time_len=T - 1,
            poly_feature_dim=4, fourier_feature_dim=4 * 2, rotation_activation=torch.nn.functional.normalize,
            opacity_activation=torch.sigmoid, scaling_activation=torch.exp, **leaf)
        pos = cls.get_position(ns, t)
        rot = cls.get_rotation(ns, t)
        opa = cls.get_opacity.fget(ns)
        scl = cls.get_scaling.fget(ns)
        g = dict(pos=f32(N, 3), rot=f32(N, 4), opa=f32(N, 1), scl=f32(N, 3))
        ((pos * g["pos"]).sum() + (rot * g["rot"]).sum() + (opa * g["opa"]).sum()
         + (scl * g["scl"]).sum()).backward()
        pre = f"t{t}_"
        out.update({pre + "pos": pos.detach().numpy(), pre + "rot": rot.detach().numpy(),
                    pre + "opa": opa.detach().numpy(), pre + "scl": scl.detach().numpy()})
        out.update({pre + "g_" + k: v.numpy() for k, v in g.items()})
        out.update({pre + "d_position": leaf["position"].grad.numpy(),
                    pre + "d_cubic": leaf["pos_cubic_node"].grad.numpy(),
                    pre + "d_rotation": leaf["rotation"].grad.numpy(),
                    pre + "d_opacity": leaf["opacity"].grad.numpy(),
                    pre + "d_scaling": leaf["scaling"].grad.numpy()})
        # the reference detaches the polynomial / Fourier sums: those tables receive no gradient
        assert leaf["rot_poly_feat"].grad is None and leaf["rot_fourier_feat"].grad is None
    np.savez_compressed(os.path.join(HERE, "dynamic_400x50.npz"), **out)
    print("dynamic_400x50.npz", {k: v.shape for k, v in out.items() if k.startswith("t24_")})


if __name__ == "__main__":
    main()
