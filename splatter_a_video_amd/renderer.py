"""Native counterpart of the reference's configured renderer.

``OrthoEnhancedRenderer.render_iter`` / ``render_batch`` return what ``DPTROrthoEnhancedRender.render_iter`` /
``render_batch`` return (reference: src/pointrix/renderer/dptr_ortho_enhanced.py:205-383, :385-433; feature packing of
src/pointrix/utils/renderer/renderer_utils.py:31-72) with the same sequence of blends -- rgb through
``alpha_blending_enhanced`` with the ``ndc`` / ``abs_ndc`` taps, depth with ``bg = 1`` and ``ndc.detach()``, the extra
attributes with ``opacity.detach()`` -- but the per-frame geometry comes from ONE fused launch
(``gs.preprocess_ortho``) where the reference runs an eager-torch orthographic projection and EWA (~75 kernels) around
``gs.compute_cov3d``.  The reference's own class keeps working unchanged through the ``dptr`` shim; this one is for
callers that want the fused path.  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import gs
from .densify import DensifyState


class OrthoEnhancedRenderer:
    def __init__(self, white_bg: bool = False, densify_abs_grad_enable: bool = False):
        self.bg_color = 1.0 if white_bg else 0.0
        self.densify_abs_grad_enable = bool(densify_abs_grad_enable)

    def render_iter(self, height: int, width: int, extrinsic_matrix: Tensor, position: Tensor, opacity: Tensor,
                    scaling: Tensor, rotation: Tensor, shs: Tensor, bg_color: Optional[float] = None, num_idx: int = 10,
                    render_attributes: Optional[Dict[str, Tensor]] = None, rgb: Optional[Tensor] = None,
                    attribute_row: Optional[Tensor] = None, **_unused) -> dict:
        """One frame.  ``render_attributes`` maps names to per-Gaussian tensors [N, c] (the reference passes them as
        keyword arguments listed in ``render_attributes_list``).  ``rgb``: colours already evaluated from ``shs``
        (``render_batch`` evaluates them once per batch: the view direction is the same constant for every frame);
        ``attribute_row``: the attributes already concatenated [N, sum c] in the order of ``render_attributes`` (they do
        not depend on the frame either: one concatenation and one split of its gradient per batch instead of per frame)."""
        W, H = int(width), int(height)
        if rgb is None:
            rgb = self.colors(shs)
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(position, scaling, rotation, extrinsic_matrix, W, H,
                                                              nearest=0.01)     # :282-321 in one launch
        idx_sorted, tile_range, _ = gs.sort_gaussian_capped(uv, depth, W, H, radius, None, conic.detach(), opacity.detach())
        ndc = torch.zeros_like(uv, requires_grad=True)
        abs_ndc = torch.zeros_like(uv, requires_grad=True)
        bg = self.bg_color if bg_color is None else bg_color
        # the three blends of the reference (rgb enhanced with the taps; depth, bg = 1, ndc.detach(); attributes,
        # bg = 0, opacity.detach(), ndc.detach()) share one geometry: ONE forward pass, one native backward per set
        sets, bgs, detach, taps = [rgb, depth], [bg, 1.0], [False, False], [True, False]
        names = list(render_attributes) if render_attributes else []
        if names:
            sets.append(attribute_row if attribute_row is not None else torch.cat([render_attributes[k] for k in names], dim=-1))
            bgs.append(0.0); detach.append(True); taps.append(False)
        res = gs.alpha_blending_shared(uv, conic, opacity, sets, idx_sorted, tile_range, bgs, W, H, ndc, abs_ndc, K=num_idx,
                                       detach_opacity=detach, taps=taps)
        gs_idx = res[-1]
        out = {"rgb": res[0], "depth": res[1]}
        start = 0
        for k in names:
            c = render_attributes[k].shape[-1]
            out[k] = res[2][start:start + c]
            start += c
        return {"rendered_features_split": out,
                "viewspace_points": abs_ndc if self.densify_abs_grad_enable else ndc,
                "visibility_filter": radius > 0,
                "radii": radius,
                "gs_idx": gs_idx}

    @staticmethod
    def colors(shs: Tensor) -> Tensor:
        """SH -> RGB for the constant view direction (0, 0, 1) of the orthographic renderer (:270-272)"""
        direction = torch.zeros(shs.shape[0], 3, dtype=torch.float32, device=shs.device)
        direction[:, 2] = 1.0
        return gs.compute_sh(shs, 3, direction)

    def render_batch(self, render_dict: dict, batch: Sequence[dict]) -> dict:
        """``render_iter`` over the frames of a batch; features stacked, visibility OR-ed, radii max-ed (:385-433).
        When the SH coefficients are shared by the whole batch (``shs`` in ``render_dict``) the colours are evaluated
        once: they do not depend on the frame, and autograd sums the frames' colour gradients before the one SH
        backward -- the same images and gradients with 1/F of the SH work."""
        feats: Dict[str, List[Tensor]] = {}
        viewspace_points, vis, radii, gs_idx = [], [], [], []
        shared_rgb = self.colors(render_dict["shs"]) if "shs" in render_dict else None
        attrs = render_dict.get("render_attributes")
        shared_row = torch.cat([attrs[k] for k in attrs], dim=-1) if attrs else None
        for b_i in batch:
            args = dict(b_i)
            args.update(render_dict)
            if shared_rgb is not None:
                args["rgb"] = shared_rgb
            if shared_row is not None and "render_attributes" not in b_i:
                args["attribute_row"] = shared_row
            r = self.render_iter(**args)
            for k, v in r["rendered_features_split"].items():
                feats.setdefault(k, []).append(v)
            viewspace_points.append(r["viewspace_points"])
            vis.append(r["visibility_filter"].unsqueeze(0))
            radii.append(r["radii"].unsqueeze(0))
            gs_idx.append(r["gs_idx"].unsqueeze(0))
        return {**{k: torch.stack(v, dim=0) for k, v in feats.items()},
                "viewspace_points": viewspace_points,
                "visibility": torch.cat(vis).any(dim=0),
                "radii": torch.cat(radii, 0).max(dim=0).values,
                "gs_idx": torch.cat(gs_idx, 0)}

    @staticmethod
    def accumulate_densify(state: DensifyState, batch_result: dict) -> None:
        """after ``loss.backward()``: feed the batch's gradient taps / radii to the device-side statistics
        (what prepare_optimizer_dict + update_structure do in the reference, src/frag_model.py:326-343)"""
        state.begin_batch()
        n = len(batch_result["viewspace_points"])
        for f, vp in enumerate(batch_result["viewspace_points"]):
            # the batch's radii / visibility are already reduced: feed them with the first frame, zeros afterwards
            r = batch_result["radii"] if f == 0 else torch.zeros_like(batch_result["radii"])
            state.accumulate_frame(r.to(torch.int32), vp.grad)
        if n == 0:
            return
        state.update()


class PerspRenderer:
    """Native counterpart of the reference's pinhole renderer ``DPTRRender`` (src/pointrix/renderer/dptr.py:42-169, :171-224):
    same call signature and return value of ``render_iter`` / ``render_batch``, the same blend (rgb + depth (+ pixel_flow) in one
    ``alpha_blending`` with the ``ndc`` tap), but the three per-Gaussian operators (project_point, compute_cov3d, ewa_project)
    run as ONE fused launch per direction (``gs.preprocess_persp``).  Cameras are not differentiated here (the reference
    renderer does not optimise them either); its own class keeps working unchanged through the ``dptr`` shim."""

    def __init__(self, white_bg: bool = False, max_sh_degree: int = 3, update_sh_iter: int = 1000):
        self.bg_color = 1.0 if white_bg else 0.0
        # the trainer's hooks on a renderer (default_trainer: renderer.update_sh_degree(iteration) every step,
        # renderer.state_dict() / load_state_dict() in checkpoints; dptr.py:42-56, :218-228)
        self.max_sh_degree, self.update_sh_iter = int(max_sh_degree), int(update_sh_iter)
        self.active_sh_degree = 0

    def update_sh_degree(self, step: int) -> None:
        """one more SH band every ``update_sh_iter`` steps up to ``max_sh_degree`` (dptr.py:218-221)"""
        if step % self.update_sh_iter == 0 and self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def state_dict(self) -> dict:
        return {"active_sh_degree": self.active_sh_degree}

    def load_state_dict(self, state_dict: dict) -> None:
        self.active_sh_degree = int(state_dict["active_sh_degree"])

    def render_iter(self, FovX, FovY, height, width, extrinsic_matrix: Tensor, intrinsic_matrix: Tensor, camera_center: Tensor,
                    position: Tensor, opacity: Tensor, scaling: Tensor, rotation: Tensor, shs: Tensor,
                    scaling_modifier: float = 1.0, render_xyz: bool = False, **kwargs) -> dict:
        W, H = int(width), int(height)
        direction = position - camera_center.reshape(1, 3).to(position.device)
        direction = direction / direction.norm(dim=1, keepdim=True)
        rgb = gs.compute_sh(shs, 3, direction)      # dptr.py:107 evaluates degree 3 whatever active_sh_degree says; kept
        uv, depth, conic, radius, tiles = gs.preprocess_persp(position, scaling, rotation, intrinsic_matrix, extrinsic_matrix, W, H,
                                                              nearest=0.01)     # dptr.py:107-147 in one launch
        idx_sorted, tile_range, _ = gs.sort_gaussian_capped(uv, depth, W, H, radius, None, conic.detach(), opacity.detach())
        names, parts = ["rgb", "depth"], [rgb, depth]
        if "pixel_flow" in kwargs:
            names.append("pixel_flow"); parts.append(kwargs["pixel_flow"])
        widths = [p.shape[-1] for p in parts]
        ndc = torch.zeros_like(uv, requires_grad=True)
        rendered = gs.alpha_blending(uv, conic, opacity, torch.cat(parts, dim=-1), idx_sorted, tile_range, self.bg_color, W, H, ndc)
        split, c0 = {}, 0
        for k, c in zip(names, widths):
            split[k] = rendered[c0:c0 + c]
            c0 += c
        return {"rendered_features_split": split, "viewspace_points": ndc, "visibility_filter": radius > 0, "radii": radius}

    def render_batch(self, render_dict: dict, batch: Sequence[dict]) -> dict:
        """``render_iter`` per batch element with its own camera; features stacked, visibility OR-ed, radii max-ed (:171-224)"""
        feats: Dict[str, List[Tensor]] = {}
        viewspace_points, vis, radii = [], [], []
        for b_i in batch:
            args = dict(b_i)
            args.update(render_dict)
            r = self.render_iter(**args)
            for k, v in r["rendered_features_split"].items():
                feats.setdefault(k, []).append(v)
            viewspace_points.append(r["viewspace_points"])
            vis.append(r["visibility_filter"].unsqueeze(0))
            radii.append(r["radii"].unsqueeze(0))
        return {**{k: torch.stack(v, dim=0) for k, v in feats.items()},
                "viewspace_points": viewspace_points,
                "visibility": torch.cat(vis).any(dim=0),
                "radii": torch.cat(radii, 0).max(dim=0).values}
