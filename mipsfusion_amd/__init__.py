"""MI355X-native implementation of the MIPS-Fusion render-and-optimise hot path.

Host side (this package) mirrors the reference's ``model.scene_rep`` / ``model.decoder`` /
``model.encodings`` interfaces; the arithmetic runs in hand-written HIP kernels for gfx950
behind the C ABI declared in ``include/mipsf.h`` (``mipsfusion_amd/csrc``).  There is no CPU
fallback: using an operator without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
