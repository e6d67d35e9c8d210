"""fastegnn_amd -- MI355X-native (gfx950) forward/backward of the FastEGNN model.

Only the hot path of GLAD-RUC/FastEGNN lives here: the drop-in ``FastEGNN`` module
(``model.py``), its ctypes binding (``_lib.py``) and the HIP sources (``csrc/``).
"""
from .model import FastEGNN, SortedGraph  # noqa: F401
from .egnn import EGNN  # noqa: F401
from .fastrf import FastRF  # noqa: F401

__all__ = ["FastEGNN", "FastRF", "EGNN", "SortedGraph"]
