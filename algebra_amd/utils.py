"""Helpers around sequences, variants and LCS graphs: the serialisations and generators the reference keeps
in algebra/utils.py (FASTA / VCF one-liners :8-17, Graphviz export `to_dot` :20-62, `slice_sequence` :65-80,
seeded generators `random_sequence` :83-104 and `random_variants` :107-158).

BASELINE config 1 draws its alleles with `random_variants`; the generators therefore consume the `random`
module's stream in exactly the reference's order, so that a seed gives the same alleles in both packages.
"""

import random

from .variants import DNA_NUCLEOTIDES, Variant, reverse_complement, to_hgvs


def fasta_sequence(lines):
    """The concatenated sequence of (FASTA) lines; header lines are skipped."""
    return "".join(line.strip() for line in lines if not line.startswith(">"))


def vcf_variant(line):
    """The variant of a simple VCF record (CHROM POS ID REF ALT ...)."""
    fields = line.split()
    start = int(fields[1]) - 1
    return Variant(start, start + len(fields[3]), fields[4])


def to_dot(reference, graph, labels=True, hgvs=True, atomics=False):
    """Lines of the LCS graph in Graphviz DOT format (reference utils.py:20-62): nodes are numbered in order
    of first appearance in `graph.bfs_traversal(atomics)`, sinks get a double periphery."""
    supremal = graph.supremal
    yield "digraph{"
    yield f'comment="reference:{reference[supremal.start:supremal.end]}"'
    yield f'comment="observed:{supremal.sequence}"'
    yield "rankdir=LR"
    yield "edge[fontname=monospace]"
    yield f'node[fixedsize=true,fontname=serif,shape=circle,width={".8" if labels else "1"}]'
    yield 'si[label="",shape=none,width=0]'
    yield "si->s0"

    number = {}

    def declare(node, is_sink_side):
        index = number[node] = len(number)
        text = f"s{index}" if labels else str(node)
        final = ",peripheries=2" if is_sink_side and not node.edges else ""
        return f's{index}[label="{text}"{final}]'

    for tail, head, variant in graph.bfs_traversal(atomics):
        if tail not in number:
            yield declare(tail, False)
        if head not in number:
            yield declare(head, True)
        if hgvs:
            text = to_hgvs(variant, reference)
        elif atomics and len(variant) > 1:
            text = variant
        else:
            text = variant[0]
        yield f's{number[tail]}->s{number[head]}[label="{text}"]'

    if not number:
        node = graph._source
        number[node] = 0
        yield f's0[label="{"s0" if labels else str(node)}",peripheries=2]'

    yield "}"


def slice_sequence(sequence, positions, inversion=False):
    """Concatenation of sequence[p0:p1], sequence[p2:p3], ... (an odd tail runs to the end), optionally
    reverse complemented."""
    positions = list(positions)
    if len(positions) % 2:
        positions.append(len(sequence))
    observed = "".join(sequence[positions[i]:positions[i + 1]] for i in range(0, len(positions), 2))
    return reverse_complement(observed) if inversion else observed


def random_sequence(max_length, min_length=0, alphabet=DNA_NUCLEOTIDES, weights=None):
    """A random sequence of min_length..max_length symbols (length drawn first, then the symbols)."""
    length = random.randint(min_length, max_length)
    return "".join(random.choices(alphabet, weights=weights, k=length))


def random_variants(reference, p=None, mu_deletion=1, mu_insertion=1):
    """Generator of a random sorted allele of non-overlapping variants (reference utils.py:107-158):
    every position starts a variant with probability p; deletion and insertion lengths are
    int(expovariate(1/mu)); the first len(deleted) inserted symbols differ from the deleted ones."""
    if p is None:
        p = 1 / len(reference)
    pos = 0
    while pos < len(reference):
        deleted = 0
        if random.random() <= p:
            deleted = min(int(random.expovariate(1 / mu_deletion)), len(reference) - pos)
            inserted = int(random.expovariate(1 / mu_insertion))
            if deleted == inserted == 0:
                deleted = inserted = 1
            sequence = ""
            if inserted:
                sequence = "".join(random.choice(DNA_NUCLEOTIDES.replace(ch, "")) for ch in reference[pos:pos + deleted])
                if inserted > len(sequence):
                    sequence += "".join(random.choices(DNA_NUCLEOTIDES, k=inserted - len(sequence)))
            yield Variant(pos, pos + deleted, sequence)
        pos += deleted + 1
