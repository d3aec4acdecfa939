"""Relations between variant alleles (mirror of the reference's
algebra/relations/variant_based.py); this `compare` is the package-level `compare`
(reference algebra/__init__.py:5-6)."""

from ..lcs import LCSgraph
from . import graph_based
from .relation import Relation


def compare(reference, lhs, rhs):
    """Compare two variant alleles (reference variant_based.py:34-52)."""
    return graph_based.compare(reference, LCSgraph.from_variant(reference, lhs), LCSgraph.from_variant(reference, rhs))


def compare_batch(reference, lhs_alleles, rhs_alleles, offset=10):
    """[compare(reference, l, r) for l, r in zip(lhs_alleles, rhs_alleles)], batched: all supremal graphs are
    built in shared rounds of window widening (`LCSgraph.from_variant_batch`), all union-window distances in one
    GPU batch.  This is the shape of BASELINE config 1 (many allele pairs against one reference)."""
    n = len(lhs_alleles)
    graphs = LCSgraph.from_variant_batch(reference, list(lhs_alleles) + list(rhs_alleles), offset)
    return graph_based.compare_batch(reference, graphs[:n], graphs[n:])


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
