"""Variant value type, HGVS / SPDI parsers and helpers (host-side mirror of the reference's algebra.variants)."""

from .parser import parse_hgvs, parse_spdi
from .variant import DNA_NUCLEOTIDES, Variant, patch, reverse_complement, to_hgvs

__all__ = ["DNA_NUCLEOTIDES", "Variant", "parse_hgvs", "parse_spdi", "patch", "reverse_complement", "to_hgvs"]
