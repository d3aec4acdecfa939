"""Relations between supremal variants (mirror of the reference's
algebra/relations/supremal_based.py)."""

from ..lcs import LCSgraph
from . import graph_based, sequence_based
from .graph_based import union_windows
from .relation import Relation


def _trivially_apart(lhs, rhs):
    return not lhs or not rhs or lhs.is_disjoint(rhs)


def _graphs(reference, lhs, rhs):
    built = LCSgraph.batch([(reference[lhs.start:lhs.end], lhs.sequence), (reference[rhs.start:rhs.end], rhs.sequence)],
                           [lhs.start, rhs.start])
    return built[0], built[1]


def compare(reference, lhs, rhs):
    """Compare two supremal variants (reference supremal_based.py:56-80)."""
    if lhs == rhs:
        return Relation.EQUIVALENT
    if _trivially_apart(lhs, rhs):
        return Relation.DISJOINT
    return graph_based.compare(reference, *_graphs(reference, lhs, rhs))


def are_equivalent(_reference, lhs, rhs):
    """Check if two variants are equivalent (reference :12-14)."""
    return lhs == rhs


def contains(reference, lhs, rhs):
    """Check if `lhs` contains `rhs` (reference :17-30)."""
    if lhs == rhs or _trivially_apart(lhs, rhs):
        return False
    start = min(lhs.start, rhs.start)
    end = max(lhs.end, rhs.end)
    lhs_observed, rhs_observed = union_windows(reference, lhs, rhs)
    return sequence_based.contains(reference[start:end], lhs_observed, rhs_observed)


def is_contained(reference, lhs, rhs):
    """Check if `lhs` is contained in `rhs` (reference :33-35)."""
    return contains(reference, rhs, lhs)


def are_disjoint(reference, lhs, rhs):
    """Check if two variants are disjoint (reference :38-45)."""
    if lhs == rhs:
        return False
    if _trivially_apart(lhs, rhs):
        return True
    return graph_based.are_disjoint(reference, *_graphs(reference, lhs, rhs))


def have_overlap(reference, lhs, rhs):
    """Check if two variants overlap (reference :48-53)."""
    if lhs == rhs or _trivially_apart(lhs, rhs):
        return False
    return graph_based.have_overlap(reference, *_graphs(reference, lhs, rhs))
