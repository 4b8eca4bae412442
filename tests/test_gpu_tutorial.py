"""The DPTR tutorial "Fitting a 2D image with Gaussian Splatting" (reference: src/submodules/dptr/README.md:143-276) on the
MI355X-native operators: random colourful 3D Gaussians, activations, Adam(lr=0.01), ``gs.rasterization`` with the
tutorial's pinhole camera, SmoothL1 loss -- the loss must fall the way an end-to-end correct forward + backward makes
it fall.  The target is a procedural image (flat colour shapes on white, like the logo the tutorial
reads from disk)."""
import math

import numpy as np
import pytest
import torch

import dptr.gs as gs

pytestmark = pytest.mark.gpu


def _target(W, H):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.ones((H, W, 3), np.float32)
    img[((xx - 0.3 * W) ** 2 + (yy - 0.4 * H) ** 2) < (0.18 * W) ** 2] = (0.9, 0.2, 0.1)
    img[(np.abs(xx - 0.68 * W) < 0.14 * W) & (np.abs(yy - 0.55 * H) < 0.22 * H)] = (0.1, 0.3, 0.85)
    img[(yy > 0.8 * H) & (xx > 0.1 * W) & (xx < 0.9 * W)] = (0.15, 0.7, 0.3)
    return torch.from_numpy(img).cuda().permute(2, 0, 1).contiguous()


class SimpleGaussian:
    def __init__(self, num_points, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        N = int(num_points)
        raw = {"xyz": torch.rand((N, 3), generator=g) * 2 - 1, "scale": torch.rand((N, 3), generator=g) * 0.1,
               "rotate": torch.rand((N, 4), generator=g), "opacity": torch.rand((N, 1), generator=g),
               "rgb": torch.rand((N, 3), generator=g)}
        self._attributes = {k: torch.nn.Parameter(v.cuda()) for k, v in raw.items()}
        self._activations = {"scale": lambda x: torch.abs(x) + 1e-8, "rotate": torch.nn.functional.normalize,
                             "opacity": torch.sigmoid, "rgb": torch.sigmoid}
        self.optimizer = torch.optim.Adam(list(self._attributes.values()), lr=0.01)

    def step(self):
        self.optimizer.step()
        self.optimizer.zero_grad()

    def get_attribute(self, name):
        act = self._activations.get(name)
        return act(self._attributes[name]) if act is not None else self._attributes[name]


def test_tutorial_2d_fit_converges():
    W = H = 128
    gt = _target(W, H)
    bg = 1
    fov = math.pi / 2.0
    fx = 0.5 * float(W) / math.tan(0.5 * fov)
    fy = 0.5 * float(H) / math.tan(0.5 * fov)
    intr = torch.tensor([fx, fy, float(W) / 2, float(H) / 2], device="cuda")
    extr = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 2.5]], device="cuda")
    gaussians = SimpleGaussian(num_points=4000, seed=0)
    cal_loss = torch.nn.SmoothL1Loss()
    losses = []
    for iteration in range(400):
        rendered = gs.rasterization(gaussians.get_attribute("xyz"), gaussians.get_attribute("scale"), gaussians.get_attribute("rotate"),
                                    gaussians.get_attribute("opacity"), gaussians.get_attribute("rgb"), intr, extr, W, H, bg)
        loss = cal_loss(rendered, gt)
        loss.backward()
        gaussians.step()
        losses.append(float(loss))
    assert all(math.isfinite(x) for x in losses)
    first, last = np.mean(losses[:5]), np.mean(losses[-5:])
    assert last < 0.2 * first, (first, last)           # the fit converges: the loss falls by more than 5x
    assert last < 4e-3, last                            # and the rendering is close to the target
    assert np.mean(losses[195:205]) < np.mean(losses[45:55]) < first   # steadily
