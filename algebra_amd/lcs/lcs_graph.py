"""LCS graphs (all minimal alignments) built on the GPU.

Drop-in for the reference's `algebra/lcs/lcs_graph.py`: same class, attributes, constructors,
traversals and exceptions.  `LCSgraph.__init__` (= `_lcs_nodes` + `_build_graph`, reference
:59-61) runs in libmalcs: kernel family 2 finds the optimal match runs, the library's host half
compresses them; this module only materialises `Node`/`Variant` objects from the arrays.

`LCSgraph.batch` builds many graphs in one device batch; the single-pair constructor is a batch
of one.
"""

from collections import deque
from collections.abc import MutableSequence

import numpy as np

from .. import _native
from ..variants import Variant, patch
from .encoding import TooManySymbols, deal, encode_pairs


class LCSgraph:
    """LCS graph class for storing all minimal alignments.

    Attributes
    ----------
    distance : int
        The simple edit distance.
    supremal : `Variant`
        The supremal variant representation.
    """

    class Node:
        """A stretch of matching symbols (reference lcs_graph.py:34-57)."""

        __slots__ = ("row", "col", "length", "edges")

        def __init__(self, row, col, length):
            self.row = row
            self.col = col
            self.length = length
            self.edges = []

        def __eq__(self, other):
            return (self.row, self.col, self.length) == (other.row, other.col, other.length)

        def __hash__(self):
            return hash((self.row, self.col, self.length))

        def __lt__(self, other):
            return (self.row, self.col) < (other.row, other.col)

        def __repr__(self):
            return f"{self.row, self.col, self.length}"

    def __init__(self, reference, observed, shift=0, max_distance=None):
        built = LCSgraph.batch([(reference, observed)], [shift], [max_distance])[0]
        self.__dict__.update(built.__dict__)

    # `_source` (reference lcs_graph.py:61) is materialised on first use: the library keeps nodes and
    # edges as arrays, and callers such as the extractor or the relation matrix often never walk them.
    @property
    def _source(self):
        source = self.__dict__.get("_built_source")
        if source is None:
            source = self._materialise()
        return source

    @_source.setter
    def _source(self, node):
        self.__dict__["_built_source"] = node

    def _materialise(self):
        observed = self._observed
        if self._result is not None:
            result, index = self._result
            nodes, edges = result.nodes(index), result.edges(index)
        else:
            nodes, edges = self._arrays
        made = [LCSgraph.Node(int(r), int(c), int(length)) for r, c, length in nodes.tolist()]
        for src, dst, v_start, v_end, v_lo, v_hi in edges.tolist():
            made[src].edges.append((made[dst], Variant(v_start, v_end, observed[v_lo:v_hi])))
        self.__dict__["_built_source"] = made[0]
        return made[0]

    @classmethod
    def _from_arrays(cls, observed, distance, supremal, nodes=None, edges=None, result=None):
        """Graph over the arrays of one pair: either explicit `nodes`/`edges`, or `result` =
        (Graphs handle, pair index) to fetch them from on demand."""
        self = cls.__new__(cls)
        self.distance = int(distance)
        start, end, lo, hi = supremal if type(supremal) is list else (int(x) for x in supremal)
        self.supremal = Variant(start, end, observed[lo:hi])
        self._observed = observed
        self._result = result
        self._arrays = (nodes, edges)
        return self

    def _canonical_arrays(self):
        """Canonical allele computed by the library on its arrays, or None for hand-built graphs."""
        if self.__dict__.get("_result") is None:
            return None
        result, index = self._result
        return result.canonical(index)

    @classmethod
    def batch(cls, pairs, shifts=None, max_distances=None, ctx=None, devices=None, raise_on_limit=True):
        """`[LCSgraph(reference, observed, shift, max_distance), ...]` in one device batch, or, with `devices`
        (a list of device ids; default MALCS_DEVICES when set), dealt over one context per entry.

        Raises ValueError("maximum distance exceeded") if any pair exceeds its limit
        (reference lcs_graph.py:345-346) -- one pair's limit then costs the caller the whole batch, which the
        reference's one-graph-per-call constructor never does.  `raise_on_limit=False` instead: the list holds None
        where the constructor would have raised, and its `.status` (numpy uint8, one per pair: _native.PAIR_OK /
        _native.PAIR_MAX_DISTANCE_EXCEEDED) says which."""
        pairs = list(pairs)
        n = len(pairs)
        if n == 0:
            out = GraphBatch()
            out.status = np.zeros(0, dtype=np.uint8)
            return out
        shift = np.zeros(n, dtype=np.int64) if shifts is None else np.asarray(list(shifts), dtype=np.int64)
        limit = None
        if max_distances is not None:
            limit = np.array([int(x) if x else 0 for x in max_distances], dtype=np.int64)
        contexts = _native.contexts_for(devices) if ctx is None else None
        if contexts and len(contexts) > 1 and n > 1:
            out = GraphBatch([None] * n)
            out.status = np.zeros(n, dtype=np.uint8)

            def part(job):
                context, index = job
                index = index.tolist()
                if not index:
                    return
                built = cls.batch([pairs[i] for i in index], shift[index], None if limit is None else limit[index], ctx=context,
                                  raise_on_limit=raise_on_limit)
                out.status[index] = built.status
                for i, graph in zip(index, built):
                    out[i] = graph
            _native.run_on_devices(part, list(zip(contexts, deal(pairs, len(contexts)))))
            return out
        ctx = ctx or (contexts[0] if contexts else _native.default_context())
        try:
            syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs(pairs)
        except TooManySymbols:
            if n < 2:
                raise
            half = n // 2
            lo = cls.batch(pairs[:half], shift[:half], None if limit is None else limit[:half], ctx=ctx, raise_on_limit=raise_on_limit)
            hi = cls.batch(pairs[half:], shift[half:], None if limit is None else limit[half:], ctx=ctx, raise_on_limit=raise_on_limit)
            out = GraphBatch(list(lo) + list(hi))
            out.status = np.concatenate([lo.status, hi.status])
            return out
        batch = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, n_symbols)
        try:
            res = batch.graphs(shift, limit)
        finally:
            batch.close()
        status = np.asarray(res.status, dtype=np.uint8).copy()
        beyond = status == _native.PAIR_MAX_DISTANCE_EXCEEDED
        if raise_on_limit and np.any(beyond):
            res.close()
            raise ValueError("maximum distance exceeded")
        # the graphs share the result handle; it is released when the last of them goes away
        distance, supremal = res.distance.tolist(), res.supremal.tolist()      # Python ints once, not per graph
        skip = beyond.tolist()
        out = GraphBatch(None if skip[i] else cls._from_arrays(pairs[i][1], distance[i], supremal[i], result=(res, i)) for i in range(n))
        out.status = status
        return out

    @classmethod
    def from_sequence(cls, reference, observed):
        """The supremal LCS graph for two sequences (reference lcs_graph.py:63-70)."""
        if reference == observed:
            return cls("", "")
        prefix_len, suffix_len = trim(reference, observed)
        delins = Variant(prefix_len, len(reference) - suffix_len, observed[prefix_len:len(observed) - suffix_len])
        return cls.from_variant(reference, [delins])

    @classmethod
    def from_supremal(cls, reference, supremal):
        """The supremal LCS graph for a supremal variant (reference lcs_graph.py:72-75)."""
        return cls(reference[supremal.start:supremal.end], supremal.sequence, shift=supremal.start)

    @classmethod
    def from_variant(cls, reference, variants, offset=10):
        """Iteratively find the supremal LCS graph for an allele by repeatedly widening a
        range of influence (reference lcs_graph.py:77-123).

        Parameters
        ----------
        reference : str
            The reference sequence.
        variants : list of `Variant`s
            The allele of interest.
        offset : int, optional
            The minimum offset around the variants.
        """
        if not variants:
            return cls("", "")
        start = min(v.start for v in variants)
        end = max(v.end for v in variants)
        observed = patch(reference[start:end], [Variant(v.start - start, v.end - start, v.sequence) for v in variants])
        if reference[start:end] == observed:
            return cls("", "")
        allele = Variant(start, end, observed)
        offset = max(offset, len(allele) // 2, 1)
        while True:
            lo = max(0, allele.start - offset)
            hi = min(len(reference), allele.end + offset)
            window = reference[lo:allele.start] + allele.sequence + reference[allele.end:hi]
            graph = cls(reference[lo:hi], window, shift=lo)
            inside_left = graph.supremal.start > lo or graph.supremal.start == 0
            inside_right = graph.supremal.end < hi or graph.supremal.end == len(reference)
            if inside_left and inside_right:
                return graph
            offset *= 2

    @classmethod
    def from_variant_batch(cls, reference, alleles, offset=10, ctx=None):
        """`[from_variant(reference, allele, offset) for allele in alleles]` with the windows of every
        widening round built in one device batch (reference lcs_graph.py:113-123 runs the rounds per
        allele; alleles are independent, so the rounds can be shared)."""
        alleles = list(alleles)
        made = VariantGraphBatch.build(cls, reference, alleles, offset, ctx)
        if made is not None:
            return made
        return cls._from_alleles_batch([(reference, variants) for variants in alleles], offset, ctx)

    @classmethod
    def from_sequence_batch(cls, pairs, ctx=None):
        """`[from_sequence(reference, observed) for reference, observed in pairs]` (reference
        lcs_graph.py:63-70) in shared device batches: every pair is trimmed to one delins variant on its
        own reference, then treated as `from_variant` treats an allele."""
        items = []
        for reference, observed in pairs:
            if reference == observed:
                items.append((reference, []))
                continue
            prefix_len, suffix_len = trim(reference, observed)
            items.append((reference, [Variant(prefix_len, len(reference) - suffix_len,
                                              observed[prefix_len:len(observed) - suffix_len])]))
        return cls._from_alleles_batch(items, 10, ctx)

    @classmethod
    def _from_alleles_batch(cls, items, offset, ctx):
        """items: [(reference, variants), ...]; the window-widening loop of `from_variant`, round by round
        for all items at once."""
        results = [None] * len(items)
        pending = []                                   # (index, reference, delins Variant, offset)
        trivial = []
        for idx, (reference, variants) in enumerate(items):
            if not variants:
                trivial.append(idx)
                continue
            if len(variants) == 1:                     # one variant: the patched stretch is its sequence
                start, end, observed = variants[0].start, variants[0].end, variants[0].sequence
            else:
                start = min(v.start for v in variants)
                end = max(v.end for v in variants)
                observed = patch(reference[start:end], [Variant(v.start - start, v.end - start, v.sequence) for v in variants])
            if reference[start:end] == observed:
                trivial.append(idx)
                continue
            allele = Variant(start, end, observed)
            pending.append((idx, reference, allele, max(offset, len(allele) // 2, 1)))
        if trivial:
            empty = cls.batch([("", "")], ctx=ctx)[0]
            for idx in trivial:
                results[idx] = empty
        while pending:
            windows, shifts, bounds = [], [], []
            for _, reference, allele, off in pending:
                lo = max(0, allele.start - off)
                hi = min(len(reference), allele.end + off)
                windows.append((reference[lo:hi], reference[lo:allele.start] + allele.sequence + reference[allele.end:hi]))
                shifts.append(lo)
                bounds.append((lo, hi))
            graphs = cls.batch(windows, shifts, ctx=ctx)
            again = []
            for (idx, reference, allele, off), graph, (lo, hi) in zip(pending, graphs, bounds):
                inside_left = graph.supremal.start > lo or graph.supremal.start == 0
                inside_right = graph.supremal.end < hi or graph.supremal.end == len(reference)
                if inside_left and inside_right:
                    results[idx] = graph
                else:
                    again.append((idx, reference, allele, off * 2))
            pending = again
        return GraphBatch(results)

    def bfs_traversal(self, atomics=False):
        """Generate all (source node, sink node, [variants]) edges in BFS order
        (reference lcs_graph.py:125-160; nodes compare by value, as there)."""
        seen = set()
        todo = deque([self._source])
        while todo:
            node = todo.popleft()
            if node in seen:
                continue
            for child, variant in node.edges:
                if atomics:
                    for atomic in variant.atomics():
                        yield node, child, atomic
                else:
                    yield node, child, [variant]
                todo.append(child)
            seen.add(node)

    def edges(self):
        """Set of all edges in the LCS graph (reference lcs_graph.py:162-164)."""
        return {variants[0] for _, _, variants in self.bfs_traversal()}

    def nodes(self):
        """Generate all nodes in the LCS graph (reference lcs_graph.py:166-177)."""
        seen = {self._source}
        stack = [self._source]
        while stack:
            node = stack.pop()
            yield node
            for child, _ in node.edges:
                if child not in seen:
                    seen.add(child)
                    stack.append(child)

    def paths(self, atomics=False):
        """Traverse all paths (alignments) in the LCS graph (reference lcs_graph.py:179-205)."""
        def walk(node, path):
            if not node.edges:
                yield path
            for child, variant in node.edges:
                if atomics:
                    for atomic in variant.atomics():
                        yield from walk(child, path + atomic)
                else:
                    yield from walk(child, path + [variant])

        return walk(self._source, [])

    def uniq_atomics(self):
        """The set of unique atomics for the whole LCS graph (reference lcs_graph.py:207-217)."""
        found = set()
        for _, _, variants in self.bfs_traversal():
            variant = variants[0]
            symbols = set(variant.sequence)
            for idx in range(variant.start, variant.end):
                found.add((idx, ""))
                found.update((idx, ch) for ch in symbols)
            found.update((variant.end, ch) for ch in symbols)
        return found

    def overlap(self, other):
        """Common and all unique atomics of two LCS graphs (reference lcs_graph.py:219-224)."""
        lhs = self.uniq_atomics()
        rhs = other.uniq_atomics()
        return lhs & rhs, lhs | rhs

    def is_disjoint(self, other):
        """Are two LCS graphs disjoint (reference lcs_graph.py:226-228).  For library-built DNA graphs the
        atoms are compared as per-position bit masks over the edge arrays, which keeps
        graphs with 1e5+ edges away from Python sets; the result is the same."""
        mine, theirs = self._atom_masks(), other._atom_masks() if isinstance(other, LCSgraph) else None
        if mine is None or theirs is None:
            return self.uniq_atomics().isdisjoint(other.uniq_atomics())
        (lo_a, mask_a), (lo_b, mask_b) = mine, theirs
        lo = max(lo_a, lo_b)
        hi = min(lo_a + len(mask_a), lo_b + len(mask_b))
        if hi <= lo:
            return True
        return not np.any(mask_a[lo - lo_a:hi - lo_a] & mask_b[lo - lo_b:hi - lo_b])

    _ATOM_BITS = {"A": 1, "C": 2, "G": 3, "T": 4}          # bit 0: the deletion atom (idx, "")

    def _atom_masks(self):
        """(first position, uint8 mask per position) with bit 0 = (pos, "") and bits 1-4 = (pos, A/C/G/T) in
        uniq_atomics(); None when the graph was put together by hand or uses other symbols."""
        cached = self.__dict__.get("_atoms")
        if cached is not None:
            return cached or None
        result = self.__dict__.get("_result")
        observed = self.__dict__.get("_observed", "")
        if result is None or not set(observed) <= set(self._ATOM_BITS):
            self.__dict__["_atoms"] = ()
            return None
        edges = result[0].edges(result[1])
        lo = self.supremal.start
        size = self.supremal.end - lo + 1
        masks = np.zeros(size, dtype=np.uint8)
        if len(edges):
            start, end = edges[:, 2] - lo, edges[:, 3] - lo
            codes = np.frombuffer(observed.encode("ascii"), dtype=np.uint8)
            span = np.zeros(size + 2, dtype=np.int32)
            np.add.at(span, start, 1)
            np.add.at(span, end, -1)                              # (idx, "") for idx in [start, end)
            masks |= (np.cumsum(span)[:size] > 0).astype(np.uint8)
            for ch, bit in self._ATOM_BITS.items():
                count = np.concatenate(([0], np.cumsum(codes == ord(ch))))
                present = count[edges[:, 5]] > count[edges[:, 4]]   # ch in observed[obs_from:obs_to]
                span[:] = 0
                np.add.at(span, start[present], 1)
                np.add.at(span, end[present] + 1, -1)             # (idx, ch) for idx in [start, end]
                masks |= (np.cumsum(span)[:size] > 0).astype(np.uint8) << bit
        self.__dict__["_atoms"] = (lo, masks)
        return self.__dict__["_atoms"]


class GraphBatch(list):
    """What the batch constructors return: a list of graphs that can also keep the per-variant columns the
    relation matrix reads (intervals, distances, supremal keys, symbol and atom masks: `relations.matrix.columns_of`),
    so that they are put together once per batch and not once per matrix.  No counterpart in the reference (its
    graphs are single objects); every list operation works, and the kept columns are dropped as soon as the list no
    longer holds the graphs they were made from."""

    __slots__ = ("_columns", "status")

    def columns(self, reference, make):
        kept = getattr(self, "_columns", None)
        # tuple comparison stops at identity for every element: 5 000 graphs in ~20 us
        if kept is None or kept[0] != tuple(self) or kept[1] != reference:
            kept = (tuple(self), reference, make(reference, self))
            self._columns = kept
        return kept[2]


_DNA_CODES = np.full(256, 255, dtype=np.uint8)
_DNA_CODES[[ord(ch) for ch in "ACGT"]] = np.arange(4, dtype=np.uint8)


class VariantGraphBatch(MutableSequence):
    """What `from_variant_batch` returns when the library ran the widening rounds itself (malcs_from_variant_batch):
    the graphs of all alleles as the library's arrays.  It reads like the list of `LCSgraph`s it stands for (length,
    indexing, slices, iteration; a graph object is made when somebody asks for it), and the relation matrix takes its
    per-variant columns straight from the library (`columns`), so 5 000 variants cost neither 5 000 window strings nor
    5 000 objects.  Changing it (assignment, del, insert, ...) turns it into the `GraphBatch` of its graphs first, which
    then does everything.  No counterpart in the reference (its graphs are single objects)."""

    @classmethod
    def build(cls, graph_cls, reference, alleles, offset, ctx):
        """None when the library's entry does not apply: symbols beyond Latin-1 or more than 256 of them."""
        try:
            ref_raw = np.frombuffer(reference.encode("latin-1"), dtype=np.uint8)
        except UnicodeEncodeError:
            return None
        starts, ends, seqs = [], [], []
        if all(len(variants) == 1 for variants in alleles):      # the usual case, without a Python statement per field and allele
            singles = [variants[0] for variants in alleles]
            starts = [v.start for v in singles]
            ends = [v.end for v in singles]
            seqs = [v.sequence for v in singles]
            alleles = ()
        for variants in alleles:
            if not variants:                                    # no variant: the allele changes nothing
                starts.append(0); ends.append(0); seqs.append("")
            elif len(variants) == 1:
                v = variants[0]
                starts.append(v.start); ends.append(v.end); seqs.append(v.sequence)
            else:                                               # several: the stretch they span, patched (reference lcs_graph.py:95-100)
                start = min(v.start for v in variants)
                end = max(v.end for v in variants)
                starts.append(start); ends.append(end)
                seqs.append(patch(reference[start:end], [Variant(v.start - start, v.end - start, v.sequence) for v in variants]))
        try:
            ins_raw = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8)
        except UnicodeEncodeError:
            return None
        ins_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(np.fromiter((len(x) for x in seqs), dtype=np.uint64, count=len(seqs)), out=ins_off[1:])
        if len(ins_raw) != int(ins_off[-1]):
            return None
        present = np.zeros(256, dtype=bool)
        present[ref_raw] = True
        present[ins_raw] = True
        symbols = np.nonzero(present)[0]
        dna = bool(np.all(_DNA_CODES[symbols] < 4))
        if dna:
            table, n_symbols = _DNA_CODES, 4
        else:
            table = np.zeros(256, dtype=np.uint8)
            table[symbols] = np.arange(len(symbols), dtype=np.uint8)
            n_symbols = max(len(symbols), 1)
        back = np.zeros(256, dtype=np.uint8)
        back[table[symbols]] = symbols
        ctx = ctx or _native.default_context()
        ref_codes = table[ref_raw]
        native = _native.VariantGraphs(ctx, ref_codes, starts, ends, table[ins_raw], ins_off, offset, n_symbols)
        self = cls()
        self._graph_cls, self._native, self._reference, self._ref_codes, self._back, self._dna = graph_cls, native, reference, ref_codes, back, dna
        self._made = [None] * native.n
        self._columns = None
        self._list = None
        return self

    def __len__(self):
        return self._native.n if self._list is None else len(self._list)

    def _graph(self, i):
        made = self._made[i]
        if made is None:
            nat = self._native
            observed = self._back[nat.observed[int(nat.obs_off[i]):int(nat.obs_off[i + 1])]].tobytes().decode("latin-1")
            made = self._made[i] = self._graph_cls._from_arrays(observed, nat.distance[i], nat.supremal[i].tolist(),
                                                                result=(nat.round(int(nat.where[i, 0])), int(nat.where[i, 1])))
        return made

    def __getitem__(self, i):
        if self._list is not None:
            return self._list[i]
        if isinstance(i, slice):
            return [self._graph(k) for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("graph index out of range")
        return self._graph(i)

    def _as_list(self):
        if self._list is None:
            self._list = GraphBatch(self._graph(k) for k in range(self._native.n))
            self._columns = None
        return self._list

    def __setitem__(self, i, graph):
        self._as_list()[i] = graph

    def __delitem__(self, i):
        del self._as_list()[i]

    def insert(self, i, graph):
        self._as_list().insert(i, graph)

    def __eq__(self, other):
        return list(self) == list(other)

    def columns(self, reference, make):
        """The per-variant columns of the relation matrix (`relations.matrix.columns_of`): from the library's arrays
        when they are asked for over the reference the batch was made on, else made from the graphs like any list's."""
        from ..relations.matrix import columns_of, columns_from_library
        if self._list is not None:
            return self._list.columns(reference, make)
        if make is columns_of and reference == self._reference:
            if self._columns is None:
                self._columns = (None, reference, columns_from_library(self))       # (the shape GraphBatch keeps)
            return self._columns[2]
        return make(reference, list(self))


def trim(lhs, rhs):
    """Lengths of the common prefix and of the common suffix of what remains
    (reference lcs_graph.py:231-235)."""
    limit = min(len(lhs), len(rhs))
    prefix = _common_prefix(lhs, rhs, limit)
    rest = limit - prefix
    suffix = _common_prefix(lhs[::-1], rhs[::-1], rest) if rest else 0
    return prefix, suffix


def _common_prefix(lhs, rhs, limit):
    """Number of leading positions (at most `limit`) where the two strings agree."""
    step = 1 << 16
    done = 0
    while done < limit:
        hi = min(limit, done + step)
        if lhs[done:hi] == rhs[done:hi]:
            done = hi
            continue
        lo_bad, hi_bad = done, hi          # first difference lies in [lo_bad, hi_bad)
        while hi_bad - lo_bad > 1:
            mid = (lo_bad + hi_bad) // 2
            if lhs[lo_bad:mid] == rhs[lo_bad:mid]:
                lo_bad = mid
            else:
                hi_bad = mid
        return lo_bad
    return done


def _lcs_nodes(reference, observed, shift=0, max_distance=None):
    """Distance and level-bucketed match runs (reference lcs_graph.py:238-348): the lists the reference's
    sweep fills, in its order, including the runs off every minimal alignment (kernel k_lcs_sweep).
    `LCSgraph.__init__` does not come through here: the library builds the graph from the runs that
    `_build_graph` keeps.
    """
    ctx = _native.default_context()
    syms, a_off, a_len, b_off, b_len, n_symbols = encode_pairs([(reference, observed)])
    batch = _native.Batch(ctx, syms, a_off, a_len, b_off, b_len, n_symbols)
    try:
        res = batch.lcs_nodes(np.array([shift], dtype=np.int64), np.array([int(max_distance or 0)], dtype=np.int64))
    finally:
        batch.close()
    try:
        if res.status[0] == _native.PAIR_MAX_DISTANCE_EXCEEDED:
            raise ValueError("maximum distance exceeded")
        levels = [[] for _ in range(int(res.counts[0, 1]))]
        for row, col, length, level in res.runs(0).tolist():
            levels[level].append(LCSgraph.Node(row, col, length))
        return int(res.distance[0]), levels
    finally:
        res.close()


def _build_graph(reference, observed, lcs_nodes, shift=0):
    """The compressed LCS graph from level lists of `LCSgraph.Node` (reference lcs_graph.py:351-453): returns
    `(source, supremal)` as the reference does.  Runs in the host half of the library (`malcs_graphs_from_runs`, the
    construction the batched entry points and the device kernel share); `lcs_nodes` is left as it was.
    """
    runs = [(node.row - shift, node.col - shift, node.length, level) for level, nodes in enumerate(lcs_nodes) for node in nodes]
    n_levels = len(lcs_nodes) if runs else 0
    res = _native.graphs_from_runs([len(reference)], [len(observed)], [shift], [n_levels],
                                   np.array(runs, dtype=np.int64).reshape(-1, 4), [0, len(runs)])
    try:
        graph = LCSgraph._from_arrays(observed, res.distance[0], res.supremal[0], res.nodes(0), res.edges(0))
    finally:
        res.close()
    return graph._source, graph.supremal
