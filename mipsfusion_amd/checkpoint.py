"""Checkpoint / hand-off format of the reference.

The reference stores a sub-map as ``torch.save(model.state_dict(), "model_<id>.pth")`` (Logger.py:33-34, 267-277,
284-293) and moves parameters between its two processes with ``load_state_dict`` of such dictionaries
(mipsfusion.py:616,632,642,683; InactiveMap.py:70,84,88,110).  ``JointEncoding`` here has the same keys, shapes and
flat tcnn parameter layout, so the files are interchangeable; the helpers below are the two directions plus the
same-process hand-off as flat device-to-device copies (no host dictionary in between).
"""
import torch

EXPECTED_KEYS = ("embedpos_fn.params", "embed_fn.params",
                 "decoder.pts_linear.0.weight", "decoder.pts_linear.0.bias",
                 "decoder.pts_linear.2.weight", "decoder.pts_linear.2.bias",
                 "decoder.rgb_linear.0.weight", "decoder.rgb_linear.0.bias",
                 "decoder.sdf_linear.0.weight", "decoder.sdf_linear.0.bias",
                 "decoder.sdf_linear.2.weight", "decoder.sdf_linear.2.bias")


def save_state_dict(model, save_path) -> None:
    """Logger.save_state_dict (Logger.py:33-34): a file the reference can ``load_state_dict``."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, save_path)


def load_state_dict(model, load_path) -> None:
    """Logger.load_state_dict (Logger.py:37-38) for a checkpoint written by the reference or by this package."""
    sd = torch.load(load_path, map_location="cpu")
    missing = [k for k in EXPECTED_KEYS if k not in sd]
    if missing:
        raise RuntimeError(f"{load_path}: not a MIPS-Fusion sub-map checkpoint (missing {missing})")
    model.load_state_dict(sd)


@torch.no_grad()
def copy_parameters_(dst_model, src_model) -> None:
    """Same-process hand-off (active <-> inactive sub-map): flat device-to-device copies of the parameter buffers
    instead of state_dict() -> load_state_dict()."""
    src = dict(src_model.named_parameters())
    for name, p in dst_model.named_parameters():
        q = src[name]
        if p.shape != q.shape:
            raise RuntimeError(f"{name}: {tuple(p.shape)} vs {tuple(q.shape)}")
        if p.numel():
            p.copy_(q, non_blocking=True)
