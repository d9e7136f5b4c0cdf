"""``get_encoder`` with the reference's signature (model/encodings.py:6-52), returning modules shaped like
``tinycudann.Encoding`` (flat fp32 ``params`` Parameter, ``n_output_dims``, ``forward(x[M,D]) -> [M,out]`` that
casts its input to fp32 and is differentiable wrt ``params`` and ``x``) but backed by the gfx950 HIP kernels of
libmipsf_hip.so.  No tinycudann, no CPU fallback."""
import numpy as np
import torch

from .. import _lib, ops


class Encoding(torch.nn.Module):
    """Stand-in for ``tcnn.Encoding(n_input_dims, encoding_config, dtype=torch.float)``."""

    def __init__(self, n_input_dims, encoding_config, dtype=torch.float, seed=1337):
        super().__init__()
        if dtype not in (torch.float, torch.float32):
            raise ValueError("only dtype=torch.float is built (the reference passes torch.float)")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.seed = seed
        self.dtype = dtype
        self.loss_scale = 1.0
        self.otype = encoding_config["otype"].lower()
        if self.otype == "hashgrid":
            if n_input_dims != 3:
                raise ValueError("HashGrid is built for 3 input dims")
            self.meta = _lib.make_grid_meta(
                n_levels=int(encoding_config.get("n_levels", 16)),
                n_features=int(encoding_config.get("n_features_per_level", 2)),
                log2_hashmap_size=int(encoding_config.get("log2_hashmap_size", 19)),
                base_resolution=int(encoding_config.get("base_resolution", 16)),
                per_level_scale=float(encoding_config.get("per_level_scale", 2.0)))
            self.n_output_dims = self.meta.n_levels * self.meta.n_features
            g = torch.Generator().manual_seed(seed)
            init = (torch.rand(self.meta.n_params, generator=g) * 2.0 - 1.0) * 1e-4     # tcnn: U(-1e-4, 1e-4)
        elif self.otype == "frequency":
            self.n_frequencies = int(encoding_config.get("n_frequencies", 12))
            self.n_output_dims = n_input_dims * 2 * self.n_frequencies
            init = torch.zeros(0)
        elif self.otype == "identity":
            self.n_output_dims = n_input_dims
            init = torch.zeros(0)
        else:
            raise ValueError(f"unsupported encoding otype {encoding_config['otype']}")
        # tcnn registers a flat fp32 `params` Parameter for every encoding (empty when it has none)
        self.params = torch.nn.Parameter(init.to(torch.float32), requires_grad=True)

    def __deepcopy__(self, memo):
        # ctypes level table is rebuilt instead of copied (copy.deepcopy(model), InactiveMap.py:67,81,107)
        new = Encoding(self.n_input_dims, self.encoding_config, self.dtype, self.seed)
        new.params = torch.nn.Parameter(self.params.detach().clone(), requires_grad=self.params.requires_grad)
        new.train(self.training)
        memo[id(self)] = new
        return new

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mipsfusion_amd encodings run on the GPU only (no CPU fallback)")
        x = x.to(torch.float)
        if self.otype == "hashgrid":
            return ops.HashGridFn.apply(x, self.params, self.meta)
        if self.otype == "frequency":
            return ops.FrequencyFn.apply(x, self.n_frequencies)
        return x * 1.0


def get_encoder(encoding, input_dim=3, n_bins=16, n_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=512):
    """Same arguments and return value as model/encodings.py:6-52 -> (embed module, out_dim)."""
    name = encoding.lower()
    if "hash" in name or "tiled" in name:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        embed = Encoding(input_dim, {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": level_dim,
                                     "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
                                     "per_level_scale": per_level_scale}, dtype=torch.float)
    elif "freq" in name:
        embed = Encoding(input_dim, {"otype": "Frequency", "n_frequencies": n_bins}, dtype=torch.float)
    elif "identity" in name:
        embed = Encoding(input_dim, {"otype": "Identity"}, dtype=torch.float)
    else:
        raise ValueError(f"unknown encoding {encoding}")
    return embed, embed.n_output_dims
