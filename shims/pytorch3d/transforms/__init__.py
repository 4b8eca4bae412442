"""``pytorch3d.transforms.quaternion_to_matrix`` / ``matrix_to_quaternion`` (reference: src/trainer_fragGS.py:1366-1370, the
editing demo: rotation -> matrix -> rotation).  Plain torch; quaternions are (w, x, y, z) with the real part first, any norm
accepted on the way in (the matrix is that of q / |q|), ``matrix_to_quaternion`` returns a unit quaternion with w >= 0
chosen through the numerically largest of the four candidates."""
import torch


def quaternion_to_matrix(quaternions):
    w, x, y, z = torch.unbind(quaternions, -1)
    s = 2.0 / (quaternions * quaternions).sum(-1)
    m = torch.stack((1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                     s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                     s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)), -1)
    return m.reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix):
    if matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {tuple(matrix.shape)}.")
    m = matrix.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    # 4 q_a q_b for every pair (a, b): row c = the quaternion scaled by 2 |q_c|
    cand = torch.stack((
        torch.stack((1 + m00 + m11 + m22, m21 - m12, m02 - m20, m10 - m01), -1),
        torch.stack((m21 - m12, 1 + m00 - m11 - m22, m10 + m01, m02 + m20), -1),
        torch.stack((m02 - m20, m10 + m01, 1 - m00 + m11 - m22, m12 + m21), -1),
        torch.stack((m10 - m01, m20 + m02, m21 + m12, 1 - m00 - m11 + m22), -1)), -2)      # [B, 4, 4]
    diag = torch.diagonal(cand, dim1=-2, dim2=-1)
    best = diag.argmax(-1)
    q = cand[torch.arange(cand.shape[0], device=cand.device), best]
    q = q / q.norm(dim=-1, keepdim=True)
    q = torch.where(q[:, :1] < 0, -q, q)
    return q.reshape(matrix.shape[:-2] + (4,))


__all__ = ["quaternion_to_matrix", "matrix_to_quaternion"]
