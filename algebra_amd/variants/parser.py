"""HGVS / SPDI text -> alleles (lists of `Variant`), the input side of BASELINE config 1.

Behavioural mirror of the reference's algebra/variants/parser.py (`parse_hgvs` :21-228, `parse_spdi`
:231-257): same accepted grammar (genomic coordinates; substitution, deletion, insertion, deletion/insertion,
duplication, inversion, HGVS and NCBI style repeats, `=`; alleles in `[...;...]`), same results, same
exception types and messages (they are part of the reference's tests, tests/variants/test_parser.py).
Written as a small cursor class rather than nested closures; nothing here touches the GPU.
"""

from .variant import DNA_NUCLEOTIDES, Variant, reverse_complement


class _Cursor:
    """Position in an expression plus the primitive matchers of the grammar."""

    def __init__(self, text, pos):
        self.text = text
        self.pos = pos

    def at_end(self):
        return self.pos == len(self.text)

    def expect(self, word):
        if self.pos > len(self.text) - len(word):
            raise ValueError("unexpected end of expression")
        if not self.text.startswith(word, self.pos):
            raise ValueError(f"expected '{word}' at {self.pos + 1}")
        self.pos += len(word)

    def accept(self, word):
        if self.pos <= len(self.text) - len(word) and self.text.startswith(word, self.pos):
            self.pos += len(word)
            return True
        return False

    def run(self, predicate, label):
        """One or more characters satisfying `predicate`."""
        if self.pos >= len(self.text):
            raise ValueError("unexpected end of expression")
        if not predicate(self.text[self.pos]):
            raise ValueError(f"expected {label} at {self.pos + 1}")
        begin = self.pos
        while self.pos < len(self.text) and predicate(self.text[self.pos]):
            self.pos += 1
        return self.text[begin:self.pos]

    def number(self):
        return int(self.run(str.isdigit, "digit"))

    def sequence(self):
        return self.run(DNA_NUCLEOTIDES.__contains__, "nucleotide")

    def optional_sequence(self):
        """A nucleotide run or None; the position only moves on success."""
        if self.pos < len(self.text) and self.text[self.pos] in DNA_NUCLEOTIDES:
            return self.sequence()
        return None

    def location(self):
        first = self.number()
        last = self.number() if self.accept("_") else first
        return first - 1, last

    def inserted(self):
        """`ACGT`, `ACGT[3]` or a compound `[ACGT[3];T;...]`."""
        def part():
            unit = self.sequence()
            count = 1
            if self.accept("["):
                count = self.number()
                self.expect("]")
            return unit * count

        compound = self.accept("[")
        result = part()
        if compound:
            while self.accept(";"):
                result += part()
            self.expect("]")
        return result


def _checked(given, start, end, reference, length_message):
    """A sequence spelled out in the expression must fit the range and the reference."""
    if len(given) != end - start:
        raise ValueError(length_message)
    if reference is not None and given != reference[start:end]:
        raise ValueError(f"'{given}' not found in reference at {start}")


def _variant(cur, reference):
    start, end = cur.location()
    anchor = cur.pos

    if cur.accept("dup"):
        given = cur.optional_sequence()
        if given is None:
            if reference is None:
                raise NotImplementedError(f"duplication without reference context at {anchor + 1}")
            if end > len(reference):
                raise ValueError("invalid range in reference")
            given = reference[start:end]
        else:
            _checked(given, start, end, reference, f"inconsistent duplicated length at {cur.pos}")
        return Variant.create_safe(start, end, given + given)

    if cur.accept("inv"):
        given = cur.optional_sequence()
        if given is None:
            if reference is None:
                raise NotImplementedError(f"inversion without reference context at {anchor + 1}")
            if end > len(reference):
                raise ValueError("invalid range in reference")
            given = reverse_complement(reference[start:end])
        else:
            if len(given) != end - start:
                raise ValueError(f"inconsistent inversion length at {anchor + 1}")
            if reference is not None and given != reverse_complement(reference[start:end]):
                raise ValueError(f"'{given}' not found in reference at {start}")
        return Variant.create_safe(start, end, given)

    if cur.accept("del"):
        if start == end:
            raise ValueError(f"invalid range at {anchor}")
        given = cur.optional_sequence()
        if given is not None:
            _checked(given, start, end, reference, f"inconsistent deleted length at {cur.pos}")
        return Variant.create_safe(start, end, cur.inserted() if cur.accept("ins") else "")

    if cur.accept("ins"):
        if end - start != 2:
            raise ValueError(f"invalid inserted range at {cur.pos}")
        return Variant.create_safe(start + 1, start + 1, cur.inserted())

    given = cur.optional_sequence() or ""

    if cur.accept(">"):
        if given:
            _checked(given, start, end, reference, f"inconstistent deletion length at {anchor + 1}")
        return Variant.create_safe(start, end, cur.sequence())

    if cur.accept("="):
        return Variant.create_safe(0, 0, "")

    if cur.accept("["):
        count = cur.number()
        cur.expect("]")
        if end - start != 1:                      # HGVS style: the range is the whole repeat tract
            return Variant.create_safe(start, end, count * given)
        if reference is None:                     # NCBI style: a position and a unit, the tract is looked up
            raise NotImplementedError(f"NCBI style repeat without reference context at {anchor + 1}")
        unit = len(given)
        copies = 0
        while reference[start + copies * unit:start + (copies + 1) * unit] == given:
            copies += 1
        if not copies:
            raise ValueError(f"'{given}' not found in reference at {start}")
        return Variant.create_safe(start, start + copies * unit, count * given)

    raise NotImplementedError(f"unsupported variant at {anchor + 1}")


def parse_hgvs(expression, reference=None):
    """The allele (sorted list of `Variant`) an HGVS expression describes (reference parser.py:21-228).

    `reference` is needed for duplications / inversions without explicit sequence and for NCBI style
    repeats, and is used to check sequences that the expression spells out.
    Raises TypeError (not a string), ValueError (syntax / simple semantics), NotImplementedError.
    """
    if not isinstance(expression, str):
        raise TypeError("expression must be a string")

    cur = _Cursor(expression, expression.find(":") + 1)
    cur.accept("g.")

    def finish(result):
        if not cur.at_end():
            raise ValueError(f"expected end of expression at {cur.pos + 1}")
        return result

    if cur.accept("="):
        return finish([])

    if cur.accept("["):
        allele = [_variant(cur, reference)]
        while cur.accept(";"):
            allele.append(_variant(cur, reference))
        cur.expect("]")
        finish(None)
        return sorted(v for v in allele if v)

    single = _variant(cur, reference)
    return finish([single] if single else [])


def parse_spdi(expression):
    """`sequence:position:deletion:insertion` -> one `Variant` (reference parser.py:231-257); the deletion
    is either a length or the deleted sequence."""
    if not isinstance(expression, str):
        raise TypeError("expression must be a string")
    _, position, deletion, insertion = expression.split(":")
    start = int(position)
    try:
        length = int(deletion)
    except ValueError:
        length = len(deletion)
    return [Variant.create_safe(start, start + length, insertion)]
