"""Relations between LCS graphs (mirror of the reference's algebra/relations/graph_based.py).

The only DP on this path is one edit distance over the union window of the two supremal
variants (reference graph_based.py:57-65); it runs on the GPU (kernel family 1).
"""

from ..lcs import edit_distance
from .relation import Relation


def union_windows(reference, lhs_supremal, rhs_supremal):
    """The two observed sequences over the union of both supremal windows (reference :57-64)."""
    start = min(lhs_supremal.start, rhs_supremal.start)
    end = max(lhs_supremal.end, rhs_supremal.end)
    lhs = reference[start:lhs_supremal.start] + lhs_supremal.sequence + reference[lhs_supremal.end:end]
    rhs = reference[start:rhs_supremal.start] + rhs_supremal.sequence + reference[rhs_supremal.end:end]
    return lhs, rhs


def from_distances(lhs_distance, rhs_distance, distance):
    """Relation decided by the three distances alone, or None (reference :67-74)."""
    if lhs_distance + rhs_distance == distance:
        return Relation.DISJOINT
    if lhs_distance - rhs_distance == distance:
        return Relation.CONTAINS
    if rhs_distance - lhs_distance == distance:
        return Relation.IS_CONTAINED
    return None


def compare(reference, lhs, rhs):
    """Compare two LCS graphs (reference graph_based.py:33-79)."""
    if lhs.supremal == rhs.supremal:
        return Relation.EQUIVALENT
    if lhs.supremal.is_disjoint(rhs.supremal):
        return Relation.DISJOINT
    distance = edit_distance(*union_windows(reference, lhs.supremal, rhs.supremal))
    decided = from_distances(lhs.distance, rhs.distance, distance)
    if decided is not None:
        return decided
    if lhs.is_disjoint(rhs):
        return Relation.DISJOINT
    return Relation.OVERLAP


def are_equivalent(reference, lhs, rhs):
    """Check if two LCS graphs are equivalent."""
    return compare(reference, lhs, rhs) == Relation.EQUIVALENT


def contains(reference, lhs, rhs):
    """Check if `lhs` contains `rhs`."""
    return compare(reference, lhs, rhs) == Relation.CONTAINS


def is_contained(reference, lhs, rhs):
    """Check if `lhs` is contained in `rhs`."""
    return compare(reference, lhs, rhs) == Relation.IS_CONTAINED


def are_disjoint(reference, lhs, rhs):
    """Check if two LCS graphs are disjoint."""
    return compare(reference, lhs, rhs) == Relation.DISJOINT


def have_overlap(reference, lhs, rhs):
    """Check if two LCS graphs overlap."""
    return compare(reference, lhs, rhs) == Relation.OVERLAP
