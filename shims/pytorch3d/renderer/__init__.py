"""``from pytorch3d.renderer import look_at_rotation`` (reference: src/trainer_fragGS.py:31; calls at :1131,1164,1171 --
``look_at_rotation(camera_position, at=((0, 0, z),), device=...)`` for the novel-view visualisation cameras).  Plain torch;
the published convention: rows of R^T are the camera's x (left), y (up), z (forward) axes in world coordinates, i.e. the
returned matrix has them as COLUMNS, and a degenerate x axis (``up`` parallel to the view direction) is rebuilt from y × z."""
import torch
import torch.nn.functional as F


def _rows(v, device):
    t = torch.as_tensor(v, dtype=torch.float32, device=device)
    return t[None] if t.dim() == 1 else t


def look_at_rotation(camera_position=((0, 0, 0),), at=((0, 0, 0),), up=((0, 1, 0),), device="cpu"):
    pos, at_, up_ = (_rows(v, device) for v in (camera_position, at, up))
    n = max(pos.shape[0], at_.shape[0], up_.shape[0])
    pos, at_, up_ = (t.expand(n, 3) for t in (pos, at_, up_))
    z = F.normalize(at_ - pos, eps=1e-5)
    x = F.normalize(torch.cross(up_, z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    lost = torch.isclose(x, torch.zeros_like(x), atol=5e-3).all(dim=1, keepdim=True)
    x = torch.where(lost, F.normalize(torch.cross(y, z, dim=1), eps=1e-5), x)
    return torch.stack((x, y, z), dim=1).transpose(1, 2)


__all__ = ["look_at_rotation"]
