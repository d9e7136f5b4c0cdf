"""Loss utilities with the reference's names (helper_functions/utils.py:5-111).  In the product the losses are
computed inside the render kernel (csrc/render.hip); these thin functions only keep the call surface that other
reference code imports (``mse2psnr``, ``batchify``, ``get_loss_from_ret``)."""
import math

import torch


def mse2psnr(x):
    return -10. * torch.log(x) / math.log(10.)


def batchify(fn, chunk=1024 * 64):
    if chunk is None:
        return fn
    return lambda inputs: torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)


def get_loss_from_ret(ret, training_cfg, rgb=True, sdf=True, depth=True, fs=True):
    """MIPSFusion.get_loss_from_ret (mipsfusion.py:142-152)."""
    loss = 0
    if rgb:
        loss = loss + training_cfg["rgb_weight"] * ret["rgb_loss"]
    if depth:
        loss = loss + training_cfg["depth_weight"] * ret["depth_loss"]
    if sdf:
        loss = loss + training_cfg["sdf_weight"] * ret["sdf_loss"]
    if fs:
        loss = loss + training_cfg["fs_weight"] * ret["fs_loss"]
    return loss
