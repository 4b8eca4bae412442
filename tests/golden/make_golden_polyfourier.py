"""Golden vectors for the polynomial / Fourier position model (second point-cloud class) from the
reference's own ``DynamicGaussianPointCloud.get_position`` / ``get_rotation`` (src/dynamic_gaussian_points.py:138-186),
called unbound on a namespace that carries the attributes they read.  Build container only; the .npz is what travels.

    python tests/golden/make_golden_polyfourier.py    ->  tests/golden/polyfourier_300x40.npz
"""
from __future__ import annotations

import importlib.util
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
    spec = importlib.util.spec_from_file_location("ref_dyn_pf", os.path.join(REF, "dynamic_gaussian_points.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.DynamicGaussianPointCloud


def main():
    torch.manual_seed(0)
    cls = _load_reference_class()
    rng = np.random.default_rng(31)
    N, T = 300, 40
    f32 = lambda *s, scale=1.0: torch.tensor(rng.normal(0, scale, size=s).astype(np.float32))
    P = dict(position=f32(N, 3), pos_poly_feat=f32(N, 4, 3, scale=0.1), pos_fourier_feat=f32(N, 8, 3, scale=0.05),
             rotation=f32(N, 4), rot_poly_feat=f32(N, 4, 4, scale=0.05), rot_fourier_feat=f32(N, 8, 4, scale=0.05))
    out = {k: v.numpy() for k, v in P.items()}
    out.update(T=np.int32(T), start_frame_id=np.int32(2), time_len=np.int32(T - 1))
    times = [2, 3, 11, 20, 39, 41]
    out["times"] = np.array(times, np.int32)
    for t in times:
        for detach in (False, True):
            leaf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            ns = types.SimpleNamespace(start_frame_id=2, time_len=T - 1, poly_feature_dim=4, fourier_feature_dim=8,
                                       rotation_activation=torch.nn.functional.normalize, **leaf)
            pos = cls.get_position(ns, t, detach_pos=detach)
            rot = cls.get_rotation(ns, t)
            g_pos, g_rot = f32(N, 3), f32(N, 4)
            ((pos * g_pos).sum() + (rot * g_rot).sum()).backward()
            pre = f"t{t}_{'det_' if detach else ''}"
            out.update({pre + "pos": pos.detach().numpy(), pre + "rot": rot.detach().numpy(), pre + "g_pos": g_pos.numpy(),
                        pre + "g_rot": g_rot.numpy(),
                        pre + "d_position": (leaf["position"].grad.numpy() if leaf["position"].grad is not None
                                             else np.zeros((N, 3), np.float32)),
                        pre + "d_pos_poly": leaf["pos_poly_feat"].grad.numpy(),
                        pre + "d_pos_fourier": leaf["pos_fourier_feat"].grad.numpy(),
                        pre + "d_rotation": leaf["rotation"].grad.numpy()})
            assert leaf["rot_poly_feat"].grad is None
    np.savez_compressed(os.path.join(HERE, "polyfourier_300x40.npz"), **out)
    print("polyfourier_300x40.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
