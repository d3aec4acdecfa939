"""Variant value type and helpers (host-side mirror of the reference's algebra.variants)."""

from .variant import DNA_NUCLEOTIDES, Variant, patch, reverse_complement, to_hgvs

__all__ = ["DNA_NUCLEOTIDES", "Variant", "patch", "reverse_complement", "to_hgvs"]
