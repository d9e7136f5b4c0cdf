"""Restatement of the three ``pytorch3d.transforms`` helpers the reference imports
(helper_functions/geometry_helper.py:3-4,14,24,33; RandomOptimizer.py:4,70,142;
Logger.py:4,134).  pytorch3d is absent from /root/reference and from this image and
its version is not pinned in environment.yaml => *** parity unpinned ***; these are
the standard real-first quaternion formulas.

TEST INFRASTRUCTURE: used by tests and by oracle/ref_import.py as the stand-in
module when the reference's geometry_helper is imported in the build container.
"""
import torch


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """[..., 4] (w, x, y, z), not necessarily unit -> [..., 3, 3]."""
    w, x, y, z = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    m = torch.stack((
        1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
        two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
        two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y),
    ), -1)
    return m.reshape(quaternions.shape[:-1] + (3, 3))


def standardize_quaternion(quaternions: torch.Tensor) -> torch.Tensor:
    """Flip sign so that the real part is non-negative."""
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """[..., 3, 3] rotation -> [..., 4] (w, x, y, z) with w >= 0 (largest-component branch)."""
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_positive_part(torch.stack((
        1.0 + m00 + m11 + m22,
        1.0 + m00 - m11 - m22,
        1.0 - m00 + m11 - m22,
        1.0 - m00 - m11 + m22), -1))
    cand = torch.stack((
        torch.stack((q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), -1),
        torch.stack((m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20), -1),
        torch.stack((m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21), -1),
        torch.stack((m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2), -1)), -2)
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    pick = torch.nn.functional.one_hot(q_abs.argmax(-1), num_classes=4) > 0.5
    out = cand[pick, :].reshape(batch + (4,))
    return standardize_quaternion(out)
