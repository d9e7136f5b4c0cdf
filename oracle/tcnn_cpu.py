"""CPU restatement of the tiny-cuda-nn 1.7 encodings the reference reaches through
``tcnn.Encoding`` (reference call sites: model/encodings.py:14-25 HashGrid,
:31-38 Frequency, :43-49 Identity; pinned version environment.yaml:74).

*** TEST INFRASTRUCTURE -- parity unpinned ***
tinycudann is a CUDA-only third-party dependency that is absent from
/root/reference and cannot be built in this image, and the reference holds no
test or golden vector at this boundary.  What is restated here is tcnn's
*published* algorithm (include/tiny-cuda-nn/encodings/grid.h: kernel_grid,
kernel_grid_backward, kernel_grid_backward_input, offset-table construction;
common_device.h: grid_scale, grid_resolution, pos_fract, grid_index,
coherent_prime_hash; encodings/frequency.h; bindings/torch/tinycudann/modules.py).
Assumptions that would have to be re-verified against a real tcnn build:

 A1 params per level = min(next_multiple(res^3, 8), 2^log2_hashmap_size); levels
    are concatenated in level order; a level's entry ``e`` holds its F features at
    params[(offset+e)*F : (offset+e)*F+F].
 A2 scale_l = exp2f(l * log2f(per_level_scale)) * base_resolution - 1 in fp32,
    res_l = ceil(scale_l) + 1.
 A3 pos = fmaf(scale, x, 0.5); cell = (uint32)(int)floorf(pos); frac = pos - floorf(pos).
 A4 index = dense stride walk while stride <= level_size, replaced by the
    coherent prime hash {1, 2654435761, 805459861} when level_size < stride,
    then ``% level_size`` (all uint32 wrap-around arithmetic).
 A5 corner c in 0..7: bit d of c set => +1 on dim d and weight factor frac_d,
    else factor (1 - frac_d); weight starts at 1 and is multiplied in dim order.
 A6 result_f = fma(weight, value_f, result_f) over corners in order (nvcc's
    default contraction of ``result += weight * data``).
 A7 Frequency: out[d*2F + 2k + s] = sin(fma(ldexp(x_d, k), pi_f32, s * pi/2_f32)).
    tcnn evaluates this with the fast ``__sinf``; the oracle (and the HIP path)
    use an accurate sine of the same fp32 argument.
 A8 torch binding: input cast to fp32, fp32 params/outputs (dtype=torch.float),
    Frequency/Identity register an EMPTY ``params`` Parameter.

Everything here is plain torch-CPU / numpy; integer index math is done in int64
and masked to 32 bits so that it is bit-exact uint32 arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

_U32 = 0xFFFFFFFF
_PRIMES = (1, 2654435761, 805459861)
PI_F32 = float(np.float32(math.pi))            # tcnn's PI constant rounded to fp32
HALF_PI_F32 = float(np.float32(math.pi) / np.float32(2.0))


# --------------------------------------------------------------------------- meta
@dataclass
class GridMeta:
    """Level table of a tcnn HashGrid (grid.h constructor; A1, A2)."""
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float                      # as fp32
    log2_per_level_scale: float                 # log2f(per_level_scale) as fp32
    scales: List[float] = field(default_factory=list)        # fp32 values
    resolutions: List[int] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)         # in entries, len L+1

    @property
    def n_params(self) -> int:
        return self.offsets[-1] * self.n_features

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features


def exp2f_cr(x: np.float32) -> np.float32:
    """Correctly rounded exp2f of an fp32 argument."""
    return np.float32(2.0 ** np.float64(x))


def make_grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=2.0) -> GridMeta:
    pls = np.float32(per_level_scale)
    l2 = np.float32(np.log2(np.float64(pls)))
    meta = GridMeta(n_levels, n_features, log2_hashmap_size, base_resolution, float(pls), float(l2))
    off = 0
    for level in range(n_levels):
        arg = np.float32(level) * l2
        scale = exp2f_cr(arg) * np.float32(base_resolution) - np.float32(1.0)
        res = int(math.ceil(float(scale))) + 1
        max_params = _U32 // 2
        n = max_params if float(res) ** 3 > float(max_params) else res ** 3
        n = ((n + 7) // 8) * 8
        n = min(n, 1 << log2_hashmap_size)
        meta.scales.append(float(scale))
        meta.resolutions.append(res)
        meta.offsets.append(off)
        off += n
    meta.offsets.append(off)
    return meta


# ----------------------------------------------------------------- fp32 helpers
def _fma32(a: torch.Tensor, b, c) -> torch.Tensor:
    """fp32 fused multiply-add emulated through fp64 (product exact, one extra
    rounding of the sum that differs from a true fma with probability ~2^-29)."""
    bb = b.double() if torch.is_tensor(b) else float(b)
    cc = c.double() if torch.is_tensor(c) else float(c)
    return (a.double() * bb + cc).float()


def _cell_and_frac(x32: torch.Tensor, scale: float):
    pos = _fma32(x32, scale, 0.5)                       # A3
    fl = torch.floor(pos)
    cell = fl.to(torch.int32).to(torch.int64) & _U32    # (uint32)(int)tmp
    frac = pos - fl
    return cell, frac


def _grid_index(cell: torch.Tensor, size: int, res: int) -> torch.Tensor:
    """cell: [..., 3] int64 holding uint32 values -> entry index within the level (A4)."""
    stride = 1
    index = torch.zeros_like(cell[..., 0])
    for d in range(3):
        if stride > size:
            break
        index = (index + cell[..., d] * stride) & _U32
        stride = (stride * res) & _U32
    if size < stride:
        index = torch.zeros_like(cell[..., 0])
        for d in range(3):
            index = index ^ ((cell[..., d] * _PRIMES[d]) & _U32)
    return index % size


def _corner_tables(cell, frac):
    """-> corners [M,8,3] (uint32 in int64), weights [M,8] fp32 (A5)."""
    M = cell.shape[0]
    corners = torch.empty(M, 8, 3, dtype=torch.int64)
    weights = torch.empty(M, 8, dtype=torch.float32)
    one_minus = 1.0 - frac
    for c in range(8):
        w = torch.ones(M, dtype=torch.float32)
        for d in range(3):
            if (c >> d) & 1:
                w = w * frac[:, d]
                corners[:, c, d] = (cell[:, d] + 1) & _U32
            else:
                w = w * one_minus[:, d]
                corners[:, c, d] = cell[:, d]
        weights[:, c] = w
    return corners, weights


def hashgrid_indices(x32: torch.Tensor, meta: GridMeta) -> torch.Tensor:
    """[M,3] fp32 -> [M, L, 8] int64: entry index (within the level) of each corner."""
    out = torch.empty(x32.shape[0], meta.n_levels, 8, dtype=torch.int64)
    for level in range(meta.n_levels):
        size = meta.offsets[level + 1] - meta.offsets[level]
        cell, frac = _cell_and_frac(x32, meta.scales[level])
        corners, _ = _corner_tables(cell, frac)
        out[:, level] = _grid_index(corners, size, meta.resolutions[level])
    return out


def hashgrid_forward(x32: torch.Tensor, params: torch.Tensor, meta: GridMeta) -> torch.Tensor:
    """kernel_grid (grid.h), linear interpolation.  x32 [M,3] fp32 -> [M, L*F] fp32."""
    M, F = x32.shape[0], meta.n_features
    table = params.view(-1, F)
    out = torch.empty(M, meta.n_levels * F, dtype=torch.float32)
    for level in range(meta.n_levels):
        off = meta.offsets[level]
        size = meta.offsets[level + 1] - off
        cell, frac = _cell_and_frac(x32, meta.scales[level])
        corners, weights = _corner_tables(cell, frac)
        idx = _grid_index(corners, size, meta.resolutions[level]) + off       # [M,8]
        acc = torch.zeros(M, F, dtype=torch.float32)
        for c in range(8):
            acc = _fma32(table[idx[:, c]], weights[:, c:c + 1], acc)             # A6
        out[:, level * F:(level + 1) * F] = acc
    return out


def hashgrid_backward(x32, params, dL_dy, meta: GridMeta, need_dx=True):
    """kernel_grid_backward (scatter of weight*dL/dy) and kernel_grid_backward_input
    (dL/dx = sum_levels,features dL/dy * dy/dx with tcnn's dy_dx of kernel_grid)."""
    M, F = x32.shape[0], meta.n_features
    table = params.view(-1, F)
    dparams = torch.zeros_like(table)
    dx = torch.zeros(M, 3, dtype=torch.float32) if need_dx else None
    for level in range(meta.n_levels):
        off = meta.offsets[level]
        size = meta.offsets[level + 1] - off
        scale = meta.scales[level]
        res = meta.resolutions[level]
        g = dL_dy[:, level * F:(level + 1) * F]
        cell, frac = _cell_and_frac(x32, scale)
        corners, weights = _corner_tables(cell, frac)
        idx = _grid_index(corners, size, res) + off
        for c in range(8):
            dparams.index_add_(0, idx[:, c], weights[:, c:c + 1] * g)
        if need_dx:
            one_minus = 1.0 - frac
            for gd in range(3):
                others = [d for d in range(3) if d != gd]
                acc = torch.zeros(M, F, dtype=torch.float32)
                for sub in range(4):
                    w = torch.full((M,), scale, dtype=torch.float32)
                    pl = torch.empty(M, 3, dtype=torch.int64)
                    for j, d in enumerate(others):
                        if (sub >> j) & 1:
                            w = w * frac[:, d]
                            pl[:, d] = (cell[:, d] + 1) & _U32
                        else:
                            w = w * one_minus[:, d]
                            pl[:, d] = cell[:, d]
                    pl[:, gd] = cell[:, gd]
                    left = table[_grid_index(pl, size, res) + off]
                    pl[:, gd] = (cell[:, gd] + 1) & _U32
                    right = table[_grid_index(pl, size, res) + off]
                    acc = acc + w[:, None] * (right - left)
                dx[:, gd] += (acc * g).sum(-1)
    return dparams.view(-1), dx


# ---------------------------------------------------------------------- frequency
def frequency_forward(x32: torch.Tensor, n_frequencies: int) -> torch.Tensor:
    """frequency.h: [M,D] fp32 -> [M, D*2*n_frequencies] fp32 (A7)."""
    M, D = x32.shape
    out = torch.empty(M, D * 2 * n_frequencies, dtype=torch.float32)
    for d in range(D):
        for k in range(n_frequencies):
            t = torch.ldexp(x32[:, d], torch.tensor(k))
            for s in range(2):
                arg = _fma32(t, PI_F32, s * HALF_PI_F32)
                out[:, d * 2 * n_frequencies + 2 * k + s] = torch.sin(arg.double()).float()
    return out


def frequency_backward(x32, dL_dy, n_frequencies: int) -> torch.Tensor:
    """dL/dx_d = sum_{k,s} dL/dy * 2^k * pi * cos(arg)."""
    M, D = x32.shape
    dx = torch.zeros(M, D, dtype=torch.float32)
    for d in range(D):
        for k in range(n_frequencies):
            t = torch.ldexp(x32[:, d], torch.tensor(k))
            for s in range(2):
                arg = _fma32(t, PI_F32, s * HALF_PI_F32)
                dy_dx = (float(2 ** k) * PI_F32) * torch.cos(arg.double()).float()
                dx[:, d] += dL_dy[:, d * 2 * n_frequencies + 2 * k + s] * dy_dx
    return dx


# ------------------------------------------------------------- autograd + Module
class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x32, params, meta):
        ctx.meta = meta
        ctx.save_for_backward(x32, params)
        return hashgrid_forward(x32, params.detach(), meta)

    @staticmethod
    def backward(ctx, dL_dy):
        x32, params = ctx.saved_tensors
        dparams, dx = hashgrid_backward(x32, params.detach(), dL_dy.contiguous(), ctx.meta,
                                        need_dx=ctx.needs_input_grad[0])
        return dx, dparams, None


class _FrequencyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x32, n_frequencies):
        ctx.n = n_frequencies
        ctx.save_for_backward(x32)
        return frequency_forward(x32, n_frequencies)

    @staticmethod
    def backward(ctx, dL_dy):
        (x32,) = ctx.saved_tensors
        return frequency_backward(x32, dL_dy.contiguous(), ctx.n), None


class Encoding(torch.nn.Module):
    """Shape of ``tinycudann.Encoding`` (bindings/torch/tinycudann/modules.py; A8)."""

    def __init__(self, n_input_dims, encoding_config, dtype=torch.float, seed=1337):
        super().__init__()
        if dtype not in (torch.float, torch.float32):
            raise ValueError("oracle restates the fp32 configuration the reference uses")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.seed = seed
        self.dtype = dtype
        self.loss_scale = 1.0
        otype = encoding_config["otype"].lower()
        self.otype = otype
        if otype == "hashgrid":
            if n_input_dims != 3:
                raise ValueError("oracle restates the 3-D grid only")
            self.meta = make_grid_meta(
                n_levels=int(encoding_config.get("n_levels", 16)),
                n_features=int(encoding_config.get("n_features_per_level", 2)),
                log2_hashmap_size=int(encoding_config.get("log2_hashmap_size", 19)),
                base_resolution=int(encoding_config.get("base_resolution", 16)),
                per_level_scale=float(encoding_config.get("per_level_scale", 2.0)))
            self.n_output_dims = self.meta.n_output_dims
            g = torch.Generator().manual_seed(seed)
            init = (torch.rand(self.meta.n_params, generator=g) * 2.0 - 1.0) * 1e-4   # U(-1e-4, 1e-4)
        elif otype == "frequency":
            self.n_frequencies = int(encoding_config.get("n_frequencies", 12))
            self.n_output_dims = n_input_dims * 2 * self.n_frequencies
            init = torch.zeros(0)
        elif otype == "identity":
            self.n_output_dims = n_input_dims
            init = torch.zeros(0)
        else:
            raise ValueError(f"unsupported otype {otype}")
        self.params = torch.nn.Parameter(init.to(torch.float32), requires_grad=True)

    def forward(self, x):
        x32 = x.to(torch.float).contiguous()
        if self.otype == "hashgrid":
            return _HashGridFn.apply(x32, self.params, self.meta)
        if self.otype == "frequency":
            return _FrequencyFn.apply(x32, self.n_frequencies)
        return x32 * 1.0
