"""Golden vectors for the densification statistics from the reference's own optimizer methods
(reference: src/pointrix/optimizer/atlas_gs_optimizer.py:93-121 update_structure, :199-251 generate_clone_mask /
generate_split_mask, :351-379 prune, :414-433 accumulate_viewspace_grad; batch reduction of
src/pointrix/renderer/dptr_ortho_enhanced.py:425-431), called unbound on namespaces that carry exactly the attributes
those methods read.  Build container only; the .npz is what travels.

    python tests/golden/make_golden_densify.py
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


def _load_optimizer_class():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls

    class BaseOptimizer:
        @dataclass
        class Config:
            pass

    mod("pointrix")
    mod("pointrix.utils")
    mod("pointrix.utils.config", C=lambda v, *a: v)
    mod("pointrix.model")
    mod("pointrix.model.base_model", BaseModel=object)
    mod("pointrix.utils.gaussian_points")
    mod("pointrix.utils.gaussian_points.gaussian_utils", inverse_sigmoid=lambda x: torch.log(x / (1 - x)),
        build_rotation=lambda q: None)
    pkg = mod("pointrix.optimizer")
    pkg.__path__ = []
    mod("pointrix.optimizer.optimizer", BaseOptimizer=BaseOptimizer, OPTIMIZER_REGISTRY=_Reg())
    spec = importlib.util.spec_from_file_location("pointrix.optimizer.atlas_gs_optimizer",
                                                  os.path.join(REF, "pointrix/optimizer/atlas_gs_optimizer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.AtlasGaussianSplattingOptimizer


def main():
    torch.manual_seed(0)
    cls = _load_optimizer_class()
    rng = np.random.default_rng(77)
    N, F, STEPS = 5000, 4, 3                       # Gaussians, frames per batch, optimiser steps
    out = dict(N=np.int32(N), F=np.int32(F), STEPS=np.int32(STEPS))
    cfg = types.SimpleNamespace(densify_stop_iter=15000, densify_start_iter=500)
    ns = types.SimpleNamespace(step=1, cfg=cfg, max_radii2D=torch.zeros(N), pos_gradient_accum=torch.zeros(N, 1),
                               denom=torch.zeros(N, 1), opacity_deferred=False, opacity_reset_interval=3000)
    for s in range(STEPS):
        radius = rng.integers(0, 24, size=(F, N)).astype(np.int32)
        radius[rng.random((F, N)) < 0.45] = 0                                 # culled / degenerate in that frame
        taps = (rng.normal(size=(F, N, 2)) * 1e-4).astype(np.float32)         # ndc.grad of every frame
        # batch reduction of the renderer (dptr_ortho_enhanced.py:425-431)
        vis = torch.cat([torch.tensor(radius[f] > 0).unsqueeze(0) for f in range(F)]).any(dim=0)
        radii = torch.cat([torch.tensor(radius[f]).unsqueeze(0) for f in range(F)], 0).max(dim=0).values
        vgrad = cls.accumulate_viewspace_grad(ns, [torch.zeros(N, 2)] * F, [torch.tensor(taps[f]) for f in range(F)])
        cls.update_structure(ns, vis, vgrad, radii.float())
        ns.step += 1
        out.update({f"s{s}_radius": radius, f"s{s}_taps": taps, f"s{s}_visibility": vis.numpy(), f"s{s}_radii": radii.numpy(),
                    f"s{s}_viewspace_grad": vgrad.numpy(), f"s{s}_max_radii2D": ns.max_radii2D.numpy().copy(),
                    f"s{s}_pos_gradient_accum": ns.pos_gradient_accum.numpy().copy(), f"s{s}_denom": ns.denom.numpy().copy()})
    # ---- clone / split / prune masks
    scaling_raw = rng.normal(-4.0, 1.2, size=(N, 3)).astype(np.float32)
    opacity_raw = rng.normal(-2.0, 3.0, size=(N, 1)).astype(np.float32)
    extent, thr, min_opacity = 3.7, 2.0e-4, 0.005
    captured = {}

    class PC:                                       # what the mask / prune code reads from the point cloud
        get_scaling = torch.exp(torch.tensor(scaling_raw))
        get_opacity = torch.sigmoid(torch.tensor(opacity_raw))

        def __len__(self):
            return N

        def remove_points(self, mask, opt):
            captured["valid"] = mask.clone()

    pc = PC()
    ms = types.SimpleNamespace(point_cloud=pc, cameras_extent=extent, densify_grad_threshold=thr, percent_dense=0.01,
                               device="cpu", min_opacity=min_opacity, max_radii2D=ns.max_radii2D, optimizer=None,
                               prune_postprocess=lambda m: None)
    grads = ns.pos_gradient_accum / ns.denom
    grads[grads.isnan()] = 0.0
    clone = cls.generate_clone_mask(ms, grads)
    split = cls.generate_split_mask(ms, grads)
    cls.prune(ms, 1000)
    out.update(scaling_raw=scaling_raw, opacity_raw=opacity_raw, cameras_extent=np.float32(extent),
               densify_grad_threshold=np.float32(thr), min_opacity=np.float32(min_opacity), percent_dense=np.float32(0.01),
               grads=grads.numpy(), clone_mask=clone.numpy(), split_mask=split.numpy(), prune_valid_mask=captured["valid"].numpy())
    np.savez_compressed(os.path.join(HERE, "densify_5000.npz"), **out)
    print("densify_5000.npz", int(clone.sum()), int(split.sum()), int((~captured["valid"]).sum()),
          float(ns.denom.max()), float(ns.max_radii2D.max()))


if __name__ == "__main__":
    main()
