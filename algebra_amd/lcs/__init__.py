"""Drop-in for the reference's `algebra.lcs` (algebra/lcs/__init__.py:4-11), backed by libmalcs."""

from .edit_distance import edit_distance, edit_distance_batch
from .lcs_graph import LCSgraph, trim

__all__ = ["LCSgraph", "edit_distance", "edit_distance_batch", "trim"]
