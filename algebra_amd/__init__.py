"""algebra_amd -- MI355X-native backend for the LCS hot path of mutalyzer/algebra.

`algebra_amd.lcs` mirrors `algebra.lcs` (edit_distance, LCSgraph, trim); everything runs in
libmalcs.so (HIP, gfx950) through the C-ABI in include/malcs.h.
"""

from .lcs import LCSgraph, edit_distance, edit_distance_batch, trim
from .relations import Relation, compare
from .variants import Variant, patch

__all__ = ["LCSgraph", "Relation", "Variant", "compare", "edit_distance", "edit_distance_batch", "patch", "trim"]
__version__ = "0.1.0"
