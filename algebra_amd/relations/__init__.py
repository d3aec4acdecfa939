"""Relations between variants (mirror of the reference's algebra/relations/__init__.py:13-26:
the package-level names are the variant-based flavour)."""

from .relation import Relation
from .variant_based import are_disjoint, are_equivalent, compare, compare_batch, contains, have_overlap, is_contained
from .matrix import compare_matrix

__all__ = ["Relation", "are_disjoint", "are_equivalent", "compare", "compare_batch", "compare_matrix", "contains", "have_overlap",
           "is_contained"]
