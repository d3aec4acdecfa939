"""Relations between observed sequences (mirror of the reference's
algebra/relations/sequence_based.py).  The three distances of a comparison go to the GPU as one
batch (reference :87-89 computes them one after the other)."""

from ..lcs import LCSgraph, edit_distance_batch
from .graph_based import from_distances
from .relation import Relation


def _distances(reference, lhs, rhs):
    d = edit_distance_batch([(reference, lhs), (reference, rhs), (lhs, rhs)])
    return int(d[0]), int(d[1]), int(d[2])


def _graphs_disjoint(reference, lhs, rhs):
    return LCSgraph.from_sequence(reference, lhs).is_disjoint(LCSgraph.from_sequence(reference, rhs))


def compare(reference, lhs, rhs):
    """Compare two variants given as observed sequences (reference sequence_based.py:66-103)."""
    if lhs == rhs:
        return Relation.EQUIVALENT
    decided = from_distances(*_distances(reference, lhs, rhs))
    if decided is not None:
        return decided
    return Relation.DISJOINT if _graphs_disjoint(reference, lhs, rhs) else Relation.OVERLAP


def are_equivalent(_reference, lhs, rhs):
    """Check if two variants are equivalent (reference :8-10)."""
    return lhs == rhs


def contains(reference, lhs, rhs):
    """Check if `lhs` contains `rhs` (reference :13-25)."""
    if lhs == rhs:
        return False
    lhs_distance, rhs_distance, distance = _distances(reference, lhs, rhs)
    if lhs_distance + rhs_distance == distance:
        return False
    return lhs_distance - rhs_distance == distance


def is_contained(reference, lhs, rhs):
    """Check if `lhs` is contained in `rhs` (reference :28-30)."""
    return contains(reference, rhs, lhs)


def are_disjoint(reference, lhs, rhs):
    """Check if two variants are disjoint (reference :33-47)."""
    if lhs == rhs:
        return False
    lhs_distance, rhs_distance, distance = _distances(reference, lhs, rhs)
    if lhs_distance + rhs_distance == distance:
        return True
    if distance == abs(lhs_distance - rhs_distance):
        return False
    return _graphs_disjoint(reference, lhs, rhs)


def have_overlap(reference, lhs, rhs):
    """Check if two variants overlap (reference :50-63)."""
    if lhs == rhs:
        return False
    lhs_distance, rhs_distance, distance = _distances(reference, lhs, rhs)
    if distance in (lhs_distance + rhs_distance, abs(lhs_distance - rhs_distance)):
        return False
    return not _graphs_disjoint(reference, lhs, rhs)
