"""recmodel_amd: MI355X-native engine for the WMF/ALS hot path of titoeb/RecModel.

``from recmodel_amd import WMF`` is the drop-in for ``from RecModel import WMF``
(RecModel/__init__.py:8).  Only the WMF path is provided; see DESIGN.md for scope.
"""
from .base_model import RecModel  # noqa: F401
from .wmf_model import WMF  # noqa: F401

__all__ = ["WMF", "RecModel"]
