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


def compare_batch(reference, lhs_graphs, rhs_graphs):
    """[compare(reference, l, r) for l, r in zip(lhs_graphs, rhs_graphs)] with ONE batch of edit distances:
    the pre-filters (:51-54) are evaluated on the host, every pair they leave undecided contributes its union
    windows to a single GPU batch, and only pairs the distances cannot settle (:76-79) look at atoms."""
    from ..lcs import edit_distance_batch
    relations = [None] * len(lhs_graphs)
    windows, owners = [], []
    for k, (lhs, rhs) in enumerate(zip(lhs_graphs, rhs_graphs)):
        if lhs.supremal == rhs.supremal:
            relations[k] = Relation.EQUIVALENT
        elif lhs.supremal.is_disjoint(rhs.supremal):
            relations[k] = Relation.DISJOINT
        else:
            windows.append(union_windows(reference, lhs.supremal, rhs.supremal))
            owners.append(k)
    if windows:
        for k, distance in zip(owners, edit_distance_batch(windows).tolist()):
            lhs, rhs = lhs_graphs[k], rhs_graphs[k]
            decided = from_distances(lhs.distance, rhs.distance, distance)
            if decided is None:
                decided = Relation.DISJOINT if lhs.is_disjoint(rhs) else Relation.OVERLAP
            relations[k] = decided
    return relations


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
