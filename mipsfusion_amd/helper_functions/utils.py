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


class _WeightedLossFn(torch.autograd.Function):
    """sum_k w_k * losses[k] as one dot product forward and one scaling backward."""

    @staticmethod
    def forward(ctx, losses, w):
        ctx.save_for_backward(w)
        return torch.dot(losses, w)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return g * w, None


_WEIGHT_CACHE = {}


def _loss_weights(training_cfg, device, n, flags):
    key = (str(device), n, flags, training_cfg["rgb_weight"], training_cfg["depth_weight"], training_cfg["sdf_weight"],
           training_cfg["fs_weight"])
    w = _WEIGHT_CACHE.get(key)
    if w is None:
        vals = [training_cfg["rgb_weight"] * flags[0], training_cfg["depth_weight"] * flags[1],
                training_cfg["sdf_weight"] * flags[2], training_cfg["fs_weight"] * flags[3]] + [0.0] * (n - 4)
        w = torch.tensor(vals, dtype=torch.float32, device=device)
        _WEIGHT_CACHE[key] = w
    return w


def get_loss_from_ret(ret, training_cfg, rgb=True, sdf=True, depth=True, fs=True):
    """MIPSFusion.get_loss_from_ret (mipsfusion.py:142-152).  When ``ret`` comes from this package's
    JointEncoding.forward the weighted sum is one fused op over the kernel's loss vector (same value: the products
    and the left-to-right sum of the reference, to fp32 rounding of a 4-term dot product)."""
    if isinstance(ret, dict) and "_loss_total" in ret and rgb and sdf and depth and fs and ret["_loss_total_weights"] == (
            float(training_cfg["rgb_weight"]), float(training_cfg["depth_weight"]), float(training_cfg["sdf_weight"]),
            float(training_cfg["fs_weight"])):
        return ret["_loss_total"]           # formed inside the loss kernel with exactly these weights
    vec = ret.get("_loss_vec") if isinstance(ret, dict) else None
    if vec is not None and vec.is_cuda:
        w = _loss_weights(training_cfg, vec.device, vec.shape[0], (float(rgb), float(depth), float(sdf), float(fs)))
        return _WeightedLossFn.apply(vec, w)
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


_ONES = {}


def backward_from_one(loss, **kw):
    """``loss.backward()`` with the root gradient taken from a cached tensor of ones: autograd otherwise materialises
    ``ones_like(loss)`` with a fill kernel on every call (a 5 us launch in a 140 us tracking iteration; inside a captured
    graph it is replayed every time).  Same computation, same result."""
    if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
        from .. import ops
        one = ops.unit_grad(loss.device)     # (the render backward recognises THIS tensor: its gradient is already written)
    else:
        key = (loss.device, loss.dtype, tuple(loss.shape))
        one = _ONES.get(key)
        if one is None:
            one = _ONES[key] = torch.ones_like(loss, requires_grad=False)
    loss.backward(gradient=one, **kw)
