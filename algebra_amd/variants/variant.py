"""Deletion/insertion variants: the value type the LCS facade hands out on its edges.

Host-side mirror of the reference's `algebra/variants/variant.py` (same names, argument meaning,
ordering rules and error messages); only what the LCS path and its callers need.
Reference lines are cited per member.
"""

from itertools import combinations

DNA_NUCLEOTIDES = "ACGT"
_COMPLEMENT = str.maketrans(DNA_NUCLEOTIDES, DNA_NUCLEOTIDES[::-1])


def reverse_complement(sequence):
    """Reverse complement of a DNA string (reference variant.py:21-23)."""
    return sequence.translate(_COMPLEMENT)[::-1]


class Variant:
    """`reference[start:end]` replaced by `sequence` (reference variant.py:26-193)."""

    __slots__ = ("start", "end", "sequence")

    def __init__(self, start, end, sequence):
        self.start = start
        self.end = end
        self.sequence = sequence

    @classmethod
    def create_safe(cls, start, end, sequence):
        """Validating constructor (reference variant.py:34-68)."""
        for name, value, kind in (("start", start, int), ("end", end, int)):
            if not isinstance(value, kind):
                raise TypeError(f"{name} must be an integer")
        if not isinstance(sequence, str):
            raise TypeError("sequence must be a string")
        if start < 0:
            raise ValueError("start must be greater or equal to 0")
        if start > end:
            raise ValueError("start must not be after end")
        return cls(start, end, sequence)

    def _key(self):
        return self.start, self.end, self.sequence

    def __eq__(self, other):
        return self._key() == (other.start, other.end, other.sequence)

    def __hash__(self):
        return hash(self._key())

    def __len__(self):
        # deleted symbols plus inserted symbols (reference variant.py:77-78)
        return self.end - self.start + len(self.sequence)

    def __lt__(self, other):
        # overlapping or co-located variants have no order (reference variant.py:80-84)
        overlapping = other.start < self.end and self.start < other.end
        same_place = other.start == self.start and other.end == self.end
        if overlapping or same_place:
            raise ValueError("unorderable variants")
        return self.start < other.start or self.end < other.end

    def __repr__(self):
        return f"{self.start}:{self.end}/{self.sequence}"

    def atomics(self):
        """All ways to write this delins as single deletions and grouped insertions, in the
        reference's order (reference variant.py:89-124: lexicographic choice of the slots the
        inserted symbols take among the len(self) slots)."""
        n_ins = len(self.sequence)
        for slots in combinations(range(len(self)), n_ins):
            taken = set(slots)
            out = []
            pos = self.start
            pending = None
            nxt = 0
            for slot in range(len(self)):
                if slot in taken:
                    if pending is None:
                        pending = Variant(pos, pos, self.sequence[nxt])
                    else:
                        pending.sequence += self.sequence[nxt]
                    nxt += 1
                else:
                    if pending is not None:
                        out.append(pending)
                        pending = None
                    out.append(Variant(pos, pos + 1, ""))
                    pos += 1
            if pending is not None:
                out.append(pending)
            yield out

    def is_disjoint(self, other):
        """No common deleted position and no common inserted symbol at a shared boundary
        (reference variant.py:126-133)."""
        if other.start < self.end and self.start < other.end:
            return False
        apart = other.start > self.end or self.start > other.end
        return apart or set(self.sequence).isdisjoint(other.sequence)

    def reverse_complement(self, pivot):
        """Mirror around `pivot` (reference variant.py:135-138)."""
        return Variant(pivot - self.end - 1, pivot - self.start - 1, reverse_complement(self.sequence))

    def to_hgvs(self, reference=None, only_substitutions=True):
        """HGVS text of one variant (reference variant.py:140-181)."""
        width = self.end - self.start
        if width == 0:
            return f"{self.start}_{self.start + 1}ins{self.sequence}" if self.sequence else "="
        ref_part = reference[self.start:self.end] if reference is not None else ""
        deleted = ref_part if (reference is not None and not only_substitutions) else ""
        if width == 1:
            if not self.sequence:
                return f"{self.start + 1}del{deleted}"
            if len(self.sequence) == 1:
                return f"{self.start + 1}{ref_part}>{self.sequence}"
            return f"{self.start + 1}del{deleted}ins{self.sequence}"
        tail = f"ins{self.sequence}" if self.sequence else ""
        return f"{self.start + 1}_{self.end}del{deleted}{tail}"

    def to_spdi(self, reference_id=""):
        """SPDI text (reference variant.py:183-193)."""
        return f"{reference_id}:{self.start}:{self.end - self.start}:{self.sequence}"


def patch(reference, variants, sort=True):
    """Apply non-overlapping variants to `reference` (reference variant.py:196-233).
    Sorting raises ValueError("unorderable variants") for overlapping input."""
    pieces = []
    cursor = 0
    for variant in (sorted(variants) if sort else variants):
        pieces.append(reference[cursor:variant.start])
        pieces.append(variant.sequence)
        cursor = variant.end
    pieces.append(reference[cursor:])
    return "".join(pieces)


def to_hgvs(variants, reference=None, only_substitutions=True, sequence_prefix=False, sort=True):
    """HGVS text of an allele (reference variant.py:236-276)."""
    prefix = f"{reference}:g." if (reference is not None and sequence_prefix) else ""
    if not variants:
        return f"{prefix}="
    if len(variants) == 1:
        return f"{prefix}{variants[0].to_hgvs(reference, only_substitutions)}"
    ordered = sorted(variants) if sort else variants
    return f"{prefix}[{';'.join(v.to_hgvs(reference, only_substitutions) for v in ordered)}]"
