"""The five relations between variants, as an enum with the reference's member names and string values
(reference algebra/relations/relation.py:7-13), plus the uint8 codes the label matrices of the device use."""

from enum import Enum

# order = uint8 code in a label matrix (kernel family 3 writes these numbers)
_VALUES = ("equivalent", "contains", "is_contained", "overlap", "disjoint")

Relation = Enum("Relation", [(value.upper(), value) for value in _VALUES], module=__name__, qualname="Relation")
Relation.__doc__ = "Relation between two variants (members: " + ", ".join(value.upper() for value in _VALUES) + ")."

CODES = {relation: code for code, relation in enumerate(Relation)}
FROM_CODE = list(Relation)
