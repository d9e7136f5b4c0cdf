"""``MLP_reg`` with the reference's constructor, sub-module names and state-dict keys
(model/decoder.py:6-75): pts_linear.{0,2}, rgb_linear.0, sdf_linear.{0,2}.  ``forward`` runs the fused
MFMA decoder kernels (csrc/decoder.hip) instead of five nn.Linear calls; the nn.Linear modules only own the
Parameters (and their default initialisation)."""
import torch
import torch.nn as nn

from .. import ops


class MLP_reg(nn.Module):
    def __init__(self, cfg, input_ch=3, input_ch_pos=12, n_hidden=128, n_hidden_rgb=64, n_hidden_sdf=64,
                 n_hidden_branch=128, n_class=5, beta=80.):
        super().__init__()
        self.cfg = cfg
        self.input_ch = input_ch
        self.input_ch_pos = input_ch_pos + 3
        self.n_hidden = n_hidden
        self.n_hidden_rgb = n_hidden_rgb
        self.n_hidden_sdf = n_hidden_sdf
        self.n_hidden_branch = n_hidden_branch
        self.n_class = n_class
        self.max_class_Id = n_class - 1
        self.beta = beta
        # STATED LIMIT: the kernels are generated for ONE architecture -- the only one the reference's callers
        # ever build (model/scene_rep.py:45 passes input_ch / input_ch_pos of its two encodings and leaves every width at its
        # default; model/decoder.py:7-16 would accept others).  Operand images, LDS budgets and register tiles are sized by
        # these numbers at compile time (csrc/decoder_layout.h); another width is refused here, by name, not mis-computed.
        want = {"input_ch": 32, "input_ch_pos (+3)": 51, "n_hidden": 128, "n_hidden_rgb": 64, "n_hidden_sdf": 64,
                "n_hidden_branch": 128, "n_class": 5}
        got = {"input_ch": self.input_ch, "input_ch_pos (+3)": self.input_ch_pos, "n_hidden": n_hidden, "n_hidden_rgb": n_hidden_rgb,
               "n_hidden_sdf": n_hidden_sdf, "n_hidden_branch": n_hidden_branch, "n_class": n_class}
        bad = [f"{k} = {got[k]} (kernels: {want[k]})" for k in want if got[k] != want[k]]
        if bad:
            raise ValueError("mipsfusion_amd.model.MLP_reg: the HIP decoder kernels are built for the reference's architecture "
                             "(HashGrid 16 x 2 = 32 features, Frequency 8 bins x 2 x 3 + 3 = 51, hidden 128, rgb / sdf embeddings "
                             "64, branch 128, 5 classes; model/scene_rep.py:37-45) -- unsupported: " + "; ".join(bad))
        self.pts_linear = nn.Sequential(nn.Linear(self.input_ch_pos, n_hidden), nn.ReLU(),
                                        nn.Linear(n_hidden, n_hidden_sdf + n_hidden_rgb))
        self.rgb_linear = nn.Sequential(nn.Linear(n_hidden_rgb + self.input_ch_pos, 3))
        self.sdf_linear = nn.Sequential(nn.Linear(n_hidden_sdf + self.input_ch, n_hidden_branch), nn.ReLU(),
                                        nn.Linear(n_hidden_branch, n_class), nn.Softmax(dim=-1))

    def ordered_parameters(self):
        named = dict(self.named_parameters())
        return [named[k] for k in ops.DECODER_PARAM_ORDER]

    def forward(self, embed, embed_pos, query_pts):
        """embed [M,32], embed_pos [M,48], query_pts [M,3] -> [M,10] = rgb(3) sdf entropy prob(5)."""
        if not query_pts.is_cuda:
            raise RuntimeError("mipsfusion_amd decoder runs on the GPU only (no CPU fallback)")
        return ops.DecoderFn.apply(embed, embed_pos, query_pts, *self.ordered_parameters())
