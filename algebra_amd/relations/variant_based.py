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


def _predicate(name, relation, doc):
    def test(reference, lhs, rhs):
        return compare(reference, lhs, rhs) is relation
    test.__name__ = test.__qualname__ = name
    test.__doc__ = doc
    return test


# the reference's five predicates (variant_based.py:55-77), each `compare(...) == <one relation>`
are_equivalent = _predicate("are_equivalent", Relation.EQUIVALENT, "Are the alleles `lhs` and `rhs` equivalent?")
contains = _predicate("contains", Relation.CONTAINS, "Does allele `lhs` contain allele `rhs`?")
is_contained = _predicate("is_contained", Relation.IS_CONTAINED, "Is allele `lhs` contained in allele `rhs`?")
are_disjoint = _predicate("are_disjoint", Relation.DISJOINT, "Are the alleles `lhs` and `rhs` disjoint?")
have_overlap = _predicate("have_overlap", Relation.OVERLAP, "Do the alleles `lhs` and `rhs` overlap?")
