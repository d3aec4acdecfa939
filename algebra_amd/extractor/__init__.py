"""Canonical variant extraction from LCS graphs (mirror of the reference's algebra.extractor,
algebra/extractor/__init__.py:4-14, without the experimental helpers)."""

from .extractor import canonical, extract, extract_sequence, extract_sequence_batch, extract_supremal, to_hgvs

__all__ = ["canonical", "extract", "extract_sequence", "extract_sequence_batch", "extract_supremal", "to_hgvs"]
