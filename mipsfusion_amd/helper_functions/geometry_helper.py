"""Pose helpers on the hot path (reference: helper_functions/geometry_helper.py:11-17 + the pytorch3d quaternion
functions it imports).  Tiny differentiable torch ops: they stay torch so that autograd reaches the quaternion /
translation Parameters of the caller's pose optimiser."""
import torch


def quaternion_to_matrix(q):
    """(w, x, y, z), not necessarily unit -> rotation matrices [..., 3, 3]."""
    w, x, y, z = torch.unbind(q, -1)
    k = 2.0 / (q * q).sum(-1)
    rows = (1 - k * (y * y + z * z), k * (x * y - z * w), k * (x * z + y * w),
            k * (x * y + z * w), 1 - k * (x * x + z * z), k * (y * z - x * w),
            k * (x * z - y * w), k * (y * z + x * w), 1 - k * (x * x + y * y))
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(R):
    """Rotation matrices [..., 3, 3] -> (w, x, y, z) with w >= 0, numerically stable branch selection."""
    lead = R.shape[:-2]
    m = R.reshape(lead + (9,))
    a, b, c, d, e, f, g, h, i = torch.unbind(m, -1)
    mag = torch.stack((1 + a + e + i, 1 + a - e - i, 1 - a + e - i, 1 - a - e + i), -1)
    mag = torch.where(mag > 0, mag, torch.zeros_like(mag)).sqrt()
    table = torch.stack((torch.stack((mag[..., 0] ** 2, h - f, c - g, d - b), -1),
                         torch.stack((h - f, mag[..., 1] ** 2, d + b, c + g), -1),
                         torch.stack((c - g, d + b, mag[..., 2] ** 2, f + h), -1),
                         torch.stack((d - b, g + c, h + f, mag[..., 3] ** 2), -1)), -2)
    table = table / (2.0 * mag[..., None].clamp(min=0.1))
    best = mag.argmax(-1)
    q = torch.gather(table, -2, best[..., None, None].expand(lead + (1, 4))).squeeze(-2)
    return torch.where(q[..., :1] < 0, -q, q)


def qt_to_transform_matrix(rot, trans):
    """rot [n,4] quaternion, trans [n,3] -> [n,4,4], differentiable (geometry_helper.py:11-17)."""
    n = rot.shape[0]
    T = torch.eye(4).to(rot)[None].repeat(n, 1, 1)
    T[:, :3, :3] = quaternion_to_matrix(rot)
    T[:, :3, 3] = trans
    return T
