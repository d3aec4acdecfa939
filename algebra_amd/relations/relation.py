"""Relation enum (mirror of the reference's algebra/relations/relation.py:7-13)."""

from enum import Enum


class Relation(Enum):
    """Relation enum."""
    EQUIVALENT = "equivalent"
    CONTAINS = "contains"
    IS_CONTAINED = "is_contained"
    OVERLAP = "overlap"
    DISJOINT = "disjoint"


# uint8 codes used in label matrices (order of the enum)
CODES = {relation: code for code, relation in enumerate(Relation)}
FROM_CODE = list(Relation)
