"""Golden vectors for the structure surgery of densification (clone / split with the optimiser state)
from the reference's OWN methods, run on CPU in the build container:

  * AtlasGaussianSplattingOptimizer.densify_clone / densify_split / new_pos_scale / generate_*_mask / prune_postprocess /
    reset_densification_state (src/pointrix/optimizer/atlas_gs_optimizer.py:199-401), called unbound on a namespace;
  * PointCloud.select_atributes / extand_points / remove_points / extend_optimizer / prune_optimizer
    (src/pointrix/point_cloud/points.py:177-330), called unbound on a namespace that carries the attributes, driving a
    real torch.optim.Adam whose moments are non-zero;
  * build_rotation (src/pointrix/utils/gaussian_points/gaussian_utils.py:10-33).
The two files hard-code device="cuda" in torch.zeros(...); the loader hands them a torch proxy that maps that to the CPU
(nothing else is altered).  torch.normal's draws are recovered by re-seeding (same generator state, same call) and stored
as unit normals, so that the device kernel can be fed the very same random numbers.

    python tests/golden/make_golden_structure.py      ->  tests/golden/structure_3000.npz  (data only)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


class _TorchOnCpu:
    """forwards to torch; zeros(..., device='cuda') -> CPU"""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def zeros(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return torch.zeros(*a, **k)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    class _Reg:
        def __init__(self, *a, **k):
            pass

        def register(self, *a, **k):
            return lambda cls: cls

    class _Sub:
        def __getitem__(self, item):
            return self

    class BaseOptimizer:
        @dataclass
        class Config:
            pass

    _mod("jaxtyping", Float=_Sub(), Int=_Sub(), Bool=_Sub())
    _mod("plyfile", PlyData=object, PlyElement=object)
    _mod("simple_knn")
    _mod("simple_knn._C", distCUDA2=None)
    _mod("pointrix").__path__ = []
    _mod("pointrix.utils").__path__ = []
    _mod("pointrix.utils.config", C=lambda v, *a: v)
    _mod("pointrix.utils.system", mkdir_p=lambda p: None)
    _mod("pointrix.utils.base", BaseModule=object, BaseObject=object)
    _mod("pointrix.utils.registry", Registry=_Reg)
    _mod("pointrix.dataset").__path__ = []
    _mod("pointrix.dataset.base_data", SimplePointCloud=object)
    _mod("pointrix.logger").__path__ = []
    _mod("pointrix.logger.writer", Logger=types.SimpleNamespace(print=print, info=print, warn=print))
    _mod("pointrix.model").__path__ = []
    _mod("pointrix.model.base_model", BaseModel=object)
    _mod("pointrix.utils.gaussian_points").__path__ = []
    gu = _load(os.path.join(REF, "pointrix/utils/gaussian_points/gaussian_utils.py"), "pointrix.utils.gaussian_points.gaussian_utils")
    gu.torch = _TorchOnCpu()
    pcp = _mod("pointrix.point_cloud")
    pcp.__path__ = [os.path.join(REF, "pointrix/point_cloud")]
    pts = _load(os.path.join(REF, "pointrix/point_cloud/points.py"), "pointrix.point_cloud.points")
    op = _mod("pointrix.optimizer")
    op.__path__ = []
    _mod("pointrix.optimizer.optimizer", BaseOptimizer=BaseOptimizer, OPTIMIZER_REGISTRY=_Reg())
    atl = _load(os.path.join(REF, "pointrix/optimizer/atlas_gs_optimizer.py"), "pointrix.optimizer.atlas_gs_optimizer")
    atl.torch = _TorchOnCpu()
    return atl.AtlasGaussianSplattingOptimizer, pts.PointCloud


NAMES = ["position", "scaling", "rotation", "opacity", "features"]
WIDTH = {"position": (3,), "scaling": (3,), "rotation": (4,), "opacity": (1,), "features": (4, 3)}


def main():
    OPT, PC = load_reference()
    rng = np.random.default_rng(5)
    N, SPLIT = 3000, 2
    init = {"position": rng.uniform(-1, 1, size=(N, 3)), "scaling": rng.normal(-4.0, 1.0, size=(N, 3)),
            "rotation": rng.normal(size=(N, 4)), "opacity": rng.normal(-1.0, 2.0, size=(N, 1)),
            "features": rng.normal(size=(N, 4, 3))}
    init = {k: v.astype(np.float32) for k, v in init.items()}

    class _Cloud:           # a bare attribute holder the reference's PointCloud methods are bound to
        def __len__(self):
            return len(self.position)

    pc = _Cloud()
    pc.atributes = [{"name": n, "trainable": True} for n in NAMES]
    pc.prefix_name = "point_cloud."
    pc.scaling_inverse_activation = torch.log
    for n in NAMES:
        setattr(pc, n, nn.Parameter(torch.tensor(init[n])))
    pc.unwarp = lambda name: name.replace("point_cloud.", "")
    for meth in ("select_atributes", "extand_points", "remove_points", "extend_optimizer", "prune_optimizer"):
        setattr(pc, meth, types.MethodType(getattr(PC, meth), pc))

    class _PCView:          # get_scaling / position / rotation as the optimiser reads them (properties of the live tensors)
        pass

    adam = torch.optim.Adam([{"params": [getattr(pc, n)], "name": "point_cloud." + n, "lr": 1e-3} for n in NAMES], lr=0.0, eps=1e-15)
    for n in NAMES:         # two steps with random gradients: non-trivial exp_avg / exp_avg_sq
        getattr(pc, n).grad = torch.tensor(rng.normal(size=init[n].shape).astype(np.float32))
    adam.step()
    for n in NAMES:
        getattr(pc, n).grad = torch.tensor(rng.normal(size=init[n].shape).astype(np.float32))
    adam.step()

    def snapshot(tag, out):
        for g in adam.param_groups:
            n = g["name"].replace("point_cloud.", "")
            p = g["params"][0]
            st = adam.state[p]
            out[f"{tag}_{n}"] = p.detach().numpy().copy()
            out[f"{tag}_{n}_exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"{tag}_{n}_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()

    out = dict(N=np.int32(N), split_num=np.int32(SPLIT))
    snapshot("s0", out)
    grads = (np.abs(rng.normal(size=(N, 1))) * 2.5e-4).astype(np.float32)
    out["grads"] = grads
    extent, thr, pdense = 3.7, 2.0e-4, 0.01

    class Live:             # the attributes AtlasGaussianSplattingOptimizer reads from self.point_cloud
        def __len__(self):
            return len(pc.position)

        get_scaling = property(lambda self: torch.exp(pc.scaling))
        position = property(lambda self: pc.position)
        rotation = property(lambda self: pc.rotation)
        scaling_inverse_activation = staticmethod(torch.log)
        select_atributes = staticmethod(pc.select_atributes)
        extand_points = staticmethod(pc.extand_points)
        remove_points = staticmethod(pc.remove_points)

    ms = types.SimpleNamespace(point_cloud=Live(), optimizer=adam, cameras_extent=extent, densify_grad_threshold=thr,
                               percent_dense=pdense, device="cpu", split_num=SPLIT,
                               pos_gradient_accum=torch.zeros(N, 1), denom=torch.zeros(N, 1), max_radii2D=torch.zeros(N))
    for meth in ("generate_clone_mask", "generate_split_mask", "new_pos_scale", "reset_densification_state", "prune_postprocess"):
        setattr(ms, meth, types.MethodType(getattr(OPT, meth), ms))

    clone_mask = ms.generate_clone_mask(torch.tensor(grads))
    OPT.densify_clone(ms, torch.tensor(grads))
    out["clone_mask"] = clone_mask.numpy()
    snapshot("s1", out)                      # after the clone
    split_mask = ms.generate_split_mask(torch.tensor(grads))
    out["split_mask"] = split_mask.numpy()
    # the normal draws of new_pos_scale, recovered with the same generator state and the same call
    SEED = 2024
    torch.manual_seed(SEED)
    stds = torch.exp(pc.scaling)[split_mask].repeat(SPLIT, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3)), std=stds)
    out["unit_normals"] = (samples / stds).detach().numpy()
    torch.manual_seed(SEED)
    new_pos, new_scaling = ms.new_pos_scale(split_mask)
    out["new_pos"], out["new_scaling"] = new_pos.detach().numpy(), new_scaling.detach().numpy()
    torch.manual_seed(SEED)
    OPT.densify_split(ms, torch.tensor(grads))
    snapshot("s2", out)                      # after the split (children appended, parents removed)
    np.savez_compressed(os.path.join(HERE, "structure_3000.npz"), **out)
    print("structure_3000.npz: N", N, "clone", int(clone_mask.sum()), "-> N", out["s1_position"].shape[0], "split",
          int(split_mask.sum()), "-> N", out["s2_position"].shape[0])


if __name__ == "__main__":
    main()
