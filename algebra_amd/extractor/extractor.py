"""Canonical alleles and their HGVS text from GPU-built LCS graphs.

Host-side restatement of the reference's `algebra/extractor/extractor.py`: `canonical()`
(:10-89, a BFS over the graph that merges equally short alternatives into one delins up to their
lowest common ancestor) and the repeat-aware `to_hgvs()` (:138-228).  The graphs come from
`algebra_amd.lcs.LCSgraph` (kernel family 2).  For those graphs `canonical()` runs inside the
library on the node / edge arrays (malcs_build.cpp, same procedure) so that a 10 kb tandem-repeat
graph with 400 k edges never has to become Python objects; the Python body below serves graphs
that were put together by hand and documents the procedure.
"""

from collections import deque

from ..lcs.lcs_graph import LCSgraph, trim
from ..variants import Variant, reverse_complement


def canonical(graph):
    """The canonical allele of an LCS graph (reference extractor.py:10-89).

    `best[node]` is (parent, edge into node, depth) of the first/merged shortest way to reach
    `node`; nodes are keyed by value, as in the reference."""
    arrays = graph._canonical_arrays() if hasattr(graph, "_canonical_arrays") else None
    if arrays is not None:
        # the library ran the same procedure on its node / edge arrays (malcs_graphs_canonical)
        window = graph._observed
        return [Variant(start, end, window[lo:hi]) for start, end, lo, hi in arrays.tolist()]

    observed = graph.supremal.sequence
    source = graph._source
    shift = source.row
    best = {}

    def end_of(node):
        return node.row + node.length, node.col + node.length

    def common_ancestor(left, left_edge, right, right_edge):
        # reference :33-40 -- the right-hand edge is *not* reset between outer rounds
        while left:
            probe = right
            while probe:
                if probe == left:
                    return left, left_edge, right_edge
                probe, right_edge, _ = best[probe]
            left, left_edge, _ = best[left]

    def spelled(anchor, start, target, end):
        return observed[anchor.col + start - anchor.row - shift:target.col + end - target.row - shift]

    sink = None
    todo = deque([(source, None, None, 0)])
    while todo:
        node, parent, edge, depth = todo.popleft()
        if not node.edges:
            sink = node
        if node not in best:
            best[node] = parent, edge, depth
            todo.extend((child, node, child_edge, depth + 1) for child, child_edge in node.edges)
            continue
        old_parent, old_edge, old_depth = best[node]
        if depth > old_depth:
            continue
        ancestor, left_edge, right_edge = common_ancestor(old_parent, old_edge, parent, edge)
        start = min(left_edge.start, right_edge.start)
        if old_parent != parent and end_of(old_parent) == end_of(parent):       # :63-69
            end = max(best[parent][1].end, best[old_parent][1].end)
            best[old_parent] = ancestor, Variant(start, end, spelled(ancestor, start, parent, end)), depth - 1
        else:                                                                   # :70-74
            end = max(old_edge.end, edge.end)
            best[node] = ancestor, Variant(start, end, spelled(ancestor, start, node, end)), depth

    variants = []
    node = sink
    while True:                                                                 # :82-89
        node, variant, _ = best[node]
        if not node:
            return variants
        variants.insert(0, variant)


def extract_sequence(reference, observed):
    """Canonical allele and graph for a reference and an observed sequence (reference :117-121)."""
    graph = LCSgraph.from_sequence(reference, observed)
    return canonical(graph), graph


def extract_sequence_batch(pairs, ctx=None):
    """`[extract_sequence(reference, observed) for reference, observed in pairs]` with all LCS graphs
    built in shared device batches (`LCSgraph.from_sequence_batch`)."""
    graphs = LCSgraph.from_sequence_batch(pairs, ctx=ctx)
    return [(canonical(graph), graph) for graph in graphs]


def extract_supremal(reference, supremal):
    """Canonical allele and graph for a supremal variant (reference :124-128)."""
    graph = LCSgraph.from_supremal(reference, supremal)
    return canonical(graph), graph


def extract(reference, variants):
    """Canonical allele and graph for an allele (reference :131-135)."""
    graph = LCSgraph.from_variant(reference, variants)
    return canonical(graph), graph


def _repeat_structure(word):
    """(unit, full copies, remainder length) of the shortest period of `word`
    (reference :141-159, KMP failure function)."""
    size = len(word)
    border = [0] * size
    k = 0
    for idx in range(1, size):
        while k and word[idx] != word[k]:
            k = border[k - 1]
        if word[idx] == word[k]:
            k += 1
        border[idx] = k
    period = size - (border[-1] if size else 0)
    if period == 0:
        return "", 0, 0
    return word[:period], size // period, size % period


def _position(start, end):
    """HGVS position text (reference :161-166)."""
    if end - start == 1:
        return f"{start + 1}"
    if start == end:
        return f"{start}_{start + 1}"
    return f"{start + 1}_{end}"


def _variant_hgvs(variant, reference):
    """One canonical variant as HGVS with dup / repeat / inv / delins rules (reference :168-220)."""
    ins_unit, ins_count, ins_rest = _repeat_structure(variant.sequence)
    deleted = reference[variant.start:variant.end]
    del_unit, del_count, del_rest = _repeat_structure(deleted)

    # prefer a longer unit when reference and observed agree on it (:173-183)
    excess = len(ins_unit) - len(del_unit)
    if excess < 0 and del_unit == variant.sequence[:len(ins_unit) - excess]:
        ins_unit, ins_count, ins_rest = del_unit, 1, del_rest
    elif excess > 0 and ins_unit == deleted[:len(del_unit) + excess]:
        del_unit, del_count, del_rest = ins_unit, 1, ins_rest

    if del_unit == ins_unit:                                                    # :186-198
        if del_count == ins_count:
            raise ValueError("empty variant")
        if del_count == 1 and ins_count == 2:
            at = variant.start + ins_rest
            return f"{_position(at, at + len(ins_unit))}dup"
        assert del_rest == ins_rest
        unit = variant.sequence[ins_rest:ins_rest + len(ins_unit)]             # shifted 3'
        return f"{_position(variant.start + del_rest, variant.end)}{unit}[{ins_count}]"

    lead, tail = trim(deleted, variant.sequence)                               # :201-202
    core = Variant(variant.start + lead, variant.end - tail, variant.sequence[lead:len(variant.sequence) - tail])

    if len(core.sequence) > 1 and core.sequence == reverse_complement(reference[core.start:core.end]):   # :205-206
        return f"{_position(core.start, core.end)}inv"

    unit, count, rest = _repeat_structure(core.sequence)                       # :209-217
    if count > 1:
        text = f"{unit}[{count}]"
        if rest:
            text = f"[{text};{unit[:rest]}]"
        kind = "ins" if core.start == core.end else "delins"
        return f"{_position(core.start, core.end)}{kind}{text}"

    return core.to_hgvs(reference)                                             # :220


def to_hgvs(variants, reference):
    """HGVS text of a canonical allele with tandem repeats and complex variants
    (reference extractor.py:138-228)."""
    if not variants:
        return "="
    if len(variants) == 1:
        return _variant_hgvs(variants[0], reference)
    return f"[{';'.join(_variant_hgvs(v, reference) for v in variants)}]"
