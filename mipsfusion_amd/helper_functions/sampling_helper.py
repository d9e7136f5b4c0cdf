"""Pixel samplers (reference: helper_functions/sampling_helper.py:7-68, mipsfusion.py:135-138).

These stay on the host ON PURPOSE: the selected pixel indices must be bit-identical to the reference's, and
they are defined by the CPU RNG streams (python ``random`` and torch's default CPU generator).  Calling the same
generators, with the same shapes, in the same order reproduces them by construction."""
import random

import torch


def pixel_indices_to_rc(indices, H, W):
    return torch.div(indices, W, rounding_mode="floor"), torch.remainder(indices, W)


def pixel_rc_to_indices(rows, cols, H, W):
    return rows * W + cols


def sample_pixels_random(img_h, img_w, num):
    return torch.tensor(random.sample(range(img_h * img_w), num))


def select_samples(H, W, samples):
    """MIPSFusion.select_samples (mipsfusion.py:135-138)."""
    return torch.tensor(random.sample(range(H * W), int(samples)))


def draw_pixel_scores(depth_image, out=None):
    """The RNG half of the valid-pixel samplers: one N(0,1) draw per pixel from torch's default CPU generator, exactly
    the ``torch.randn_like(mask)`` of sampling_helper.py:30 / :62.  Split out so that a producer thread can run the
    generator stream ahead of the (independent) scoring + top-k half.  out: a preallocated flat buffer of the image's
    size and dtype (same ``normal_`` fill, same stream; a fresh 1.1 MB tensor per call is an mmap + 280 page faults
    whenever glibc's allocator decides so -- the draw then takes 1.2 ms instead of 0.66, from one run to the next)."""
    if out is not None:
        return out.normal_()
    return torch.randn_like(depth_image.flatten(), dtype=depth_image.dtype)


def _valid_scores(depth_image, blocked=None, draw=None):
    valid = (depth_image > 0.).to(depth_image.dtype)
    if blocked is not None:
        valid[blocked[0], blocked[1]] = 0
    valid = valid.flatten()
    if draw is None:
        draw = torch.randn_like(valid)
    return valid * torch.abs(draw)                          # invalid pixels score 0, valid ones |N(0,1)|


def sample_valid_pixels_random(depth_image, num, draw=None):
    return torch.topk(_valid_scores(depth_image, draw=draw), num)[1]


def _lattice_axis(size, count):
    gap, rest = (size - count) // (count + 1), (size - count) % (count + 1)
    return torch.arange(0, count, dtype=torch.int64) * (gap + 1) + gap + rest // 2


def sample_pixels_uniformly(img_h, img_w, num_h, num_w):
    r, c = _lattice_axis(img_h, num_h), _lattice_axis(img_w, num_w)
    return r[:, None].repeat(1, num_w).reshape(-1), c[None, :].repeat(num_h, 1).reshape(-1)


def sample_pixels_mix(img_h, img_w, num_h, num_w, depth_image, num, draw=None):
    """``draw`` (extension): the per-pixel N(0,1) draw made earlier by ``draw_pixel_scores`` for this call."""
    rows, cols = sample_pixels_uniformly(img_h, img_w, num_h, num_w)
    extra = torch.topk(_valid_scores(depth_image, (rows, cols), draw), num - num_h * num_w)[1]
    r2, c2 = pixel_indices_to_rc(extra, img_h, img_w)
    return torch.cat([rows, r2], 0), torch.cat([cols, c2], 0)
