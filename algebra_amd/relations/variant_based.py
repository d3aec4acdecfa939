"""Relations between variant alleles (mirror of the reference's
algebra/relations/variant_based.py); this `compare` is the package-level `compare`
(reference algebra/__init__.py:5-6)."""

from ..lcs import LCSgraph
from . import graph_based
from .relation import Relation


def compare(reference, lhs, rhs):
    """Compare two variant alleles (reference variant_based.py:34-52)."""
    return graph_based.compare(reference, LCSgraph.from_variant(reference, lhs), LCSgraph.from_variant(reference, rhs))


def are_equivalent(reference, lhs, rhs):
    """Check if two variants are equivalent."""
    return compare(reference, lhs, rhs) == Relation.EQUIVALENT


def contains(reference, lhs, rhs):
    """Check if `lhs` contains `rhs`."""
    return compare(reference, lhs, rhs) == Relation.CONTAINS


def is_contained(reference, lhs, rhs):
    """Check if `lhs` is contained in `rhs`."""
    return compare(reference, lhs, rhs) == Relation.IS_CONTAINED


def are_disjoint(reference, lhs, rhs):
    """Check if two variants are disjoint."""
    return compare(reference, lhs, rhs) == Relation.DISJOINT


def have_overlap(reference, lhs, rhs):
    """Check if two variants overlap."""
    return compare(reference, lhs, rhs) == Relation.OVERLAP
