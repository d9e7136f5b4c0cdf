"""BUILD-CONTAINER-ONLY shim: import the reference's own Python hot-path modules from
/root/reference so that (i) oracle/path_cpu.py can be checked function by function
against them and (ii) golden fixtures can be generated (tests/golden/make_golden.py).

Nothing here travels: on the GPU box /root/reference does not exist and this module
raises.  It is never imported by the product, by ``-m gpu`` tests, smoke() or bench.py.

What is substituted and why:
  * ``tinycudann``            -> oracle.tcnn_cpu   (CUDA-only third party, absent)
  * ``pytorch3d.transforms``  -> oracle.p3d_cpu    (absent)
  * ``Tensor.cuda()/Module.cuda()`` -> identity    (model/decoder.py:29 hard-codes .cuda())
The reference's arithmetic in scene_rep.py / decoder.py / helper_functions/utils.py /
sampling_helper.py runs unmodified.
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


_loaded = {}


def load():
    """-> namespace with scene_rep, decoder, encodings, utils, sampling_helper, geometry_helper."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError("reference tree not present (this shim only works in the build container)")
    from . import p3d_cpu, tcnn_cpu

    sys.modules["tinycudann"] = tcnn_cpu
    p3d = types.ModuleType("pytorch3d")
    p3d.transforms = p3d_cpu
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.transforms"] = p3d_cpu

    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("model", "helper_functions", "utils", "datasets"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[name]
    _loaded["scene_rep"] = importlib.import_module("model.scene_rep")
    _loaded["decoder"] = importlib.import_module("model.decoder")
    _loaded["encodings"] = importlib.import_module("model.encodings")
    _loaded["utils"] = importlib.import_module("helper_functions.utils")
    _loaded["sampling_helper"] = importlib.import_module("helper_functions.sampling_helper")
    _loaded["geometry_helper"] = importlib.import_module("helper_functions.geometry_helper")
    return types.SimpleNamespace(**_loaded)
