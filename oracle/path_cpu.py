"""Torch-CPU restatement of the reference's render-and-optimise path.

TEST INFRASTRUCTURE (checker + cpu_baseline only; never imported by the product).

Pinned: every function here is compared, on seeded inputs, with the reference's own
Python imported in the build container (oracle/ref_import.py) by
tests/test_oracle_golden.py (test_oracle_equals_live_reference_on_fresh_seed), and the resulting vectors are committed under
tests/golden/ (generator: tests/golden/make_golden.py).  The hash-grid / frequency
arithmetic underneath (oracle/tcnn_cpu.py) is third-party and *parity unpinned*.

Reference map (all under /root/reference):
  place_samples      model/scene_rep.py:156-176   depth-guided placement, sort, jitter
  ray_points         model/scene_rep.py:179
  normalise_points   model/scene_rep.py:134-142   (float64 by construction, mipsfusion.py:94-96)
  decoder_forward    model/decoder.py:53-75
  query_normalised   model/scene_rep.py:118-128
  sdf_to_weights     model/scene_rep.py:58-78
  composite          model/scene_rep.py:81-103
  sdf_losses         helper_functions/utils.py:21-49,71-111
  train_forward      model/scene_rep.py:190-238
  total_loss         mipsfusion.py:142-152
  adam_reference     mipsfusion.py:580-584 (torch.optim.Adam semantics)
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import tcnn_cpu

DECODER_KEYS = ("pts_linear.0", "pts_linear.2", "rgb_linear.0", "sdf_linear.0", "sdf_linear.2")


# ------------------------------------------------------------------ sample placement
def place_samples(n_rays: int, target_d: Optional[torch.Tensor], tr: dict, cam: dict,
                  noise: Optional[torch.Tensor]) -> torch.Tensor:
    """-> z_vals [N,S] fp32.  ``noise`` is the U[0,1) tensor the reference draws with
    torch.rand at scene_rep.py:176 (an INPUT here so that placement is reproducible)."""
    near, far = cam["near"], cam["far"]
    if target_d is not None:
        around = torch.linspace(-tr["range_d"], tr["range_d"], steps=tr["n_range_d"]).to(target_d)
        z_near = around[None, :].repeat(n_rays, 1) + target_d
        no_depth = target_d.squeeze(-1) <= 0
        z_near[no_depth] = torch.linspace(near, far, steps=tr["n_range_d"]).to(target_d)
        if tr["n_samples_d"] > 0:
            uniform = torch.linspace(near, far, tr["n_samples_d"])[None, :].repeat(n_rays, 1).to(target_d)
            z_vals, _ = torch.sort(torch.cat([uniform, z_near], -1), -1)
        else:
            z_vals = z_near
    else:
        z_vals = torch.linspace(near, far, tr["n_samples"])[None, :].repeat(n_rays, 1)
    if tr["perturb"] > 0.0:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * noise
    return z_vals


def ray_points(rays_o, rays_d, z_vals):
    return rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]


def normalise_points(pts, grid_cfg: dict, bound64, half_len64):
    flat = pts.reshape(-1, pts.shape[-1])
    if grid_cfg["tcnn_encoding"]:
        if grid_cfg["use_bound_normalize"]:
            flat = (flat - bound64[:, 0]) / (bound64[:, 1] - bound64[:, 0])
        else:
            flat = (flat + half_len64) / (2 * half_len64)
    return flat


# -------------------------------------------------------------------------- decoder
def decoder_forward(w: Dict[str, torch.Tensor], embed, embed_pos, x32,
                    n_hidden_sdf=64, n_class=5):
    """w: state-dict style {'pts_linear.0.weight': ..., ...} -> [M,10]."""
    e = torch.cat([x32, embed_pos], -1)
    h = F.linear(e, w["pts_linear.0.weight"], w["pts_linear.0.bias"]).relu()
    h = F.linear(h, w["pts_linear.2.weight"], w["pts_linear.2.bias"])
    sdf_emb, rgb_emb = h[:, :n_hidden_sdf], h[:, n_hidden_sdf:]
    rgb = F.linear(torch.cat([rgb_emb, e], -1), w["rgb_linear.0.weight"], w["rgb_linear.0.bias"])
    g = F.linear(torch.cat([sdf_emb, embed], -1), w["sdf_linear.0.weight"], w["sdf_linear.0.bias"]).relu()
    prob = torch.softmax(F.linear(g, w["sdf_linear.2.weight"], w["sdf_linear.2.bias"]), -1)
    entropy = -(prob * torch.log2(prob + 1e-5)).sum(-1, keepdim=True)
    ids = torch.arange(0.0, n_class, 1.0)
    sdf = (prob * ids[None]).sum(-1, keepdim=True)
    sdf = (sdf / (n_class - 1) - 0.5) * 2
    return torch.cat([rgb, sdf, entropy, prob], -1)


# ---------------------------------------------------------------------- compositing
def sdf_to_weights(sdf, z_vals, trunc: float, sc_factor: float):
    w = torch.sigmoid(sdf / trunc) * torch.sigmoid(-sdf / trunc)
    crossing = (sdf[:, 1:] * sdf[:, :-1]) < 0.0
    first = torch.argmax(crossing.to(sdf.dtype), dim=1, keepdim=True)
    z_min = torch.gather(z_vals, 1, first)
    keep = (z_vals < z_min + sc_factor * trunc).to(sdf.dtype)
    w = w * keep
    return w / (w.sum(-1, keepdim=True) + 1e-8)


def composite(raw, z_vals, trunc: float, sc_factor: float):
    rgb = torch.sigmoid(raw[..., :3])
    w = sdf_to_weights(raw[..., 3], z_vals, trunc, sc_factor)
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth = (w * z_vals).sum(-1)
    var = (w * (z_vals - depth[:, None]) ** 2).sum(-1)
    acc = w.sum(-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    return dict(rgb=rgb_map, depth=depth, disp_map=disp, acc_map=acc, depth_var=var, weights=w)


# --------------------------------------------------------------------------- losses
def sdf_losses(z_vals, target_d, sdf, prob, truncation: float, n_class=5, emd_w=0.01):
    """target_d is [N,1].  Means run over ALL N*S elements (utils.py:52-67 -> F.mse_loss)."""
    front = (z_vals < target_d - truncation).to(z_vals.dtype)
    back = (z_vals > target_d + truncation).to(z_vals.dtype)
    has_depth = (target_d > 0.0).to(z_vals.dtype)
    band = (1.0 - front) * (1.0 - back) * has_depth
    n_front = torch.count_nonzero(front)
    n_band = torch.count_nonzero(band)
    total = n_front + n_band
    fs_weight = 1.0 - n_front / total
    sdf_weight = 1.0 - n_band / total
    fs = F.mse_loss(sdf * front, front) * fs_weight
    sd = F.mse_loss((z_vals + sdf * truncation) * band, target_d * band) * sdf_weight
    if emd_w > 0:
        top = n_class - 1
        ids = torch.arange(0, n_class).to(prob)
        fs_emd = (prob * (top - ids) * front[..., None]).sum(-1).mean() / 250
        gt_class = ((target_d - z_vals) + truncation) / (2.0 * truncation) * top
        sd_emd = ((gt_class[..., None] - ids).abs() * band[..., None] * prob).sum(-1).mean() / 5000
        fs = fs + fs_emd * emd_w
        sd = sd + sd_emd * emd_w
    return fs, sd


def total_loss(ret, tr: dict):
    return (tr["rgb_weight"] * ret["rgb_loss"] + tr["depth_weight"] * ret["depth_loss"]
            + tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"])


# ------------------------------------------------------------------- the scene model
class CpuScene(torch.nn.Module):
    """CPU counterpart of JointEncoding (scene_rep.py:11-238): same parameters, same
    state-dict keys, functional internals."""

    def __init__(self, cfg: dict, bound, half_len):
        super().__init__()
        self.cfg = cfg
        self.bound64 = torch.as_tensor(bound, dtype=torch.float64)
        self.half_len64 = torch.as_tensor(half_len, dtype=torch.float64)
        per_level_scale = float(2.0 ** (math.log2(256 / 16) / 15))
        self.embed_fn = tcnn_cpu.Encoding(3, {
            "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
            "log2_hashmap_size": cfg["grid"]["hash_size"], "base_resolution": 16,
            "per_level_scale": per_level_scale})
        self.embedpos_fn = tcnn_cpu.Encoding(3, {"otype": "Frequency", "n_frequencies": cfg["pos"]["n_bins"]})
        in_pos = self.embedpos_fn.n_output_dims + 3
        in_grid = self.embed_fn.n_output_dims
        dec = torch.nn.Module()
        dec.pts_linear = torch.nn.Sequential(torch.nn.Linear(in_pos, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128))
        dec.rgb_linear = torch.nn.Sequential(torch.nn.Linear(64 + in_pos, 3))
        dec.sdf_linear = torch.nn.Sequential(torch.nn.Linear(64 + in_grid, 128), torch.nn.ReLU(), torch.nn.Linear(128, 5))
        self.decoder = dec

    def decoder_weights(self):
        return {k: v for k, v in self.decoder.named_parameters()}

    def query_normalised(self, pts_norm):
        flat = pts_norm.reshape(-1, pts_norm.shape[-1]) / self.cfg["training"]["norm_factor"]
        embed = self.embed_fn(flat)
        embed_pos = self.embedpos_fn(flat)
        return decoder_forward(self.decoder_weights(), embed, embed_pos, flat.to(torch.float32))

    def run_network(self, pts):
        flat = normalise_points(pts, self.cfg["grid"], self.bound64, self.half_len64)
        out = self.query_normalised(flat)
        return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])

    def render_rays(self, rays_o, rays_d, target_d=None, noise=None):
        tr, cam = self.cfg["training"], self.cfg["cam"]
        z_vals = place_samples(rays_o.shape[0], target_d, tr, cam, noise)
        z_vals = z_vals.to(rays_o)
        raw = self.run_network(ray_points(rays_o, rays_d, z_vals))
        out = composite(raw, z_vals, tr["trunc"], self.cfg["data"]["sc_factor"])
        out.pop("weights")
        out["z_vals"] = z_vals
        out["raw"] = raw
        return out

    def train_forward(self, rays_o, rays_d, target_rgb, target_d, noise, emd_w=0.01):
        tr, cam = self.cfg["training"], self.cfg["cam"]
        rend = self.render_rays(rays_o, rays_d, target_d, noise)
        d = target_d.squeeze(-1)
        valid = (d > 0.0) & (d < cam["depth_trunc"])
        rgb_w = valid.clone().unsqueeze(-1)
        rgb_w[rgb_w == 0] = tr["rgb_missing"]
        rgb_loss = F.mse_loss(rend["rgb"] * rgb_w, target_rgb * rgb_w)
        psnr = -10.0 * torch.log(rgb_loss) / math.log(10.0)
        depth_loss = F.mse_loss(rend["depth"][valid], d[valid])
        truncation = tr["trunc"] * self.cfg["data"]["sc_factor"]
        fs_loss, sdf_loss = sdf_losses(rend["z_vals"], target_d, rend["raw"][..., 3], rend["raw"][..., 5:],
                                       truncation, 5, emd_w)
        return dict(rgb=rend["rgb"], depth=rend["depth"], rgb_loss=rgb_loss, depth_loss=depth_loss,
                    sdf_loss=sdf_loss, fs_loss=fs_loss, psnr=psnr, z_vals=rend["z_vals"], raw=rend["raw"])


# ----------------------------------------------------------------------------- Adam
def adam_reference(p, g, m, v, step: int, lr, beta1, beta2, eps, weight_decay):
    """One torch.optim.Adam step (amsgrad=False, maximize=False), returns new (p, m, v).
    ``step`` is the 1-based step count AFTER increment."""
    if weight_decay != 0:
        g = g + weight_decay * p
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
