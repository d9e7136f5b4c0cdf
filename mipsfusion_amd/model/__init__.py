"""Drop-in counterparts of the reference's ``model`` package for the hot path
(model/scene_rep.py, model/decoder.py, model/encodings.py)."""
from .encodings import get_encoder  # noqa: F401
from .decoder import MLP_reg  # noqa: F401
from .scene_rep import JointEncoding  # noqa: F401
