"""ctypes binding of libmalcs.so (include/malcs.h).

The library is the product: there is no Python or CPU fallback.  Importing this module
without a built `algebra_amd/libmalcs.so` raises ImportError; creating a context without a
gfx950 device raises MalcsError.
"""

import atexit
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MALCS_LIB") or os.path.join(_HERE, "libmalcs.so")

MALCS_OK = 0
PAIR_OK = 0
PAIR_MAX_DISTANCE_EXCEEDED = 1


class MalcsError(RuntimeError):
    """A libmalcs call failed (message from malcs_last_error())."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C algebra_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); algebra_amd has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, u64, i64, i32, u32 = c.c_void_p, c.c_uint64, c.c_int64, c.c_int32, c.c_uint32
    P = c.POINTER
    sig = {
        "malcs_version": (c.c_int, []),
        "malcs_last_error": (c.c_char_p, []),
        "malcs_device_count": (c.c_int, [P(c.c_int)]),
        "malcs_ctx_create": (c.c_int, [c.c_int, P(vp)]),
        "malcs_ctx_destroy": (c.c_int, [vp]),
        "malcs_ctx_sync": (c.c_int, [vp]),
        "malcs_ctx_device_info": (c.c_int, [vp, c.c_char_p, c.c_int, P(c.c_int), P(c.c_int)]),
        "malcs_batch_create": (c.c_int, [vp, vp, u64, vp, vp, vp, vp, u64, c.c_int, P(vp)]),
"malcs_batch_create_packed": (c.c_int, [vp, vp, u64, c.c_int, vp, vp, vp, vp, u64, c.c_int, P(vp)]),
                "malcs_batch_destroy": (c.c_int, [vp]),
        "malcs_batch_size": (u64, [vp]),
        "malcs_batch_replan": (c.c_int, [vp]),
        "malcs_edit_distance_launch": (c.c_int, [vp]),
        "malcs_edit_distance_fetch": (c.c_int, [vp, vp]),
        "malcs_edit_distance_time": (c.c_int, [vp, c.c_int, P(c.c_float)]),
        "malcs_edit_distance_time_kernel": (c.c_int, [vp, c.c_int, P(c.c_float)]),
        "malcs_edit_distance_kernel_durations": (c.c_int, [vp, c.c_int, vp]),
        "malcs_edit_distance_step_durations": (c.c_int, [vp, c.c_int, vp]),
        "malcs_batch_plan_info": (c.c_int, [vp, P(c.c_uint64)]),
        "malcs_batch_plan_dump": (c.c_int, [vp, c.c_int, vp, c.c_uint64, P(c.c_uint64)]),
        "malcs_edit_distance_batch": (c.c_int, [vp, vp, u64, vp, vp, vp, vp, u64, c.c_int, vp]),
        "malcs_edit_distance_noalloc": (c.c_int, [vp, vp, vp, u64, vp, vp, vp, vp, u64, c.c_int, vp]),
        "malcs_lcs_graph_batch": (c.c_int, [vp, vp, vp, P(vp)]),
        "malcs_lcs_nodes_batch": (c.c_int, [vp, vp, vp, P(vp)]),
        "malcs_graphs_destroy": (c.c_int, [vp]),
        "malcs_graphs_from_runs": (c.c_int, [u64, vp, vp, vp, vp, vp, vp, P(vp)]),
        "malcs_graphs_counts": (c.c_int, [vp, vp]),
        "malcs_graphs_status": (c.c_int, [vp, vp, vp]),
        "malcs_graphs_supremal": (c.c_int, [vp, vp]),
        "malcs_graphs_columns": (c.c_int, [vp, vp, vp, vp, vp]),
        "malcs_lcs_graph_batch_columns": (c.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "malcs_graphs_runs": (c.c_int, [vp, u64, vp]),
        "malcs_graphs_nodes": (c.c_int, [vp, u64, vp]),
        "malcs_graphs_edges": (c.c_int, [vp, u64, vp]),
        "malcs_graphs_canonical_count": (c.c_int, [vp, u64, P(i64)]),
        "malcs_graphs_canonical": (c.c_int, [vp, u64, vp]),
        "malcs_graphs_timing": (c.c_int, [vp, vp]),
        "malcs_graphs_device_built": (c.c_int, [vp, P(u64), P(c.c_float)]),
        "malcs_graphs_tie_bytes": (c.c_int, [vp, P(u64), P(u64)]),
        "malcs_from_variant_batch": (c.c_int, [vp, vp, u64, u64, vp, vp, vp, vp, i64, c.c_int, P(vp)]),
        "malcs_vgraphs_destroy": (c.c_int, [vp]),
        "malcs_vgraphs_size": (u64, [vp]),
        "malcs_vgraphs_columns": (c.c_int, [vp, vp, vp, vp, vp]),
        "malcs_vgraphs_observed": (c.c_int, [vp, vp]),
        "malcs_vgraphs_rounds": (c.c_int, [vp, P(c.c_int)]),
        "malcs_vgraphs_round": (c.c_int, [vp, c.c_int, P(vp), P(u64)]),
        "malcs_vgraphs_matrix_columns": (c.c_int, [vp, vp, vp, P(c.c_int), vp, vp, vp, vp]),
        "malcs_relmatrix_create": (c.c_int, [vp, u64, vp, vp, vp, vp, u32, u32, u32, P(vp)]),
        "malcs_relmatrix_destroy": (c.c_int, [vp]),
        "malcs_relmatrix_pending_count": (c.c_int, [vp, P(u64)]),
        "malcs_relmatrix_pending": (c.c_int, [vp, vp]),
        "malcs_relmatrix_resolve": (c.c_int, [vp, vp]),
        "malcs_relmatrix_decide": (c.c_int, [vp, vp, u64, vp, vp, vp, vp, vp, c.c_int, P(u64)]),
        "malcs_graphs_atoms": (c.c_int, [vp, u64, vp, vp, vp, vp, vp]),
        "malcs_relmatrix_tiles": (c.c_int, [vp, vp]),
        "malcs_host_alloc": (c.c_int, [u64, P(vp)]),
        "malcs_host_free": (c.c_int, [vp]),
        "malcs_relmatrix_assemble": (c.c_int, [vp, vp, vp]),
        "malcs_relmatrix_tiles_device": (c.c_int, [vp, P(vp), P(u64)]),
        "malcs_relmatrix_assemble_device": (c.c_int, [vp, vp, vp]),
        "malcs_relmatrix_assemble_comm": (c.c_int, [vp, vp, vp]),
        "malcs_comm_unique_id": (c.c_int, [vp]),
        "malcs_comm_create": (c.c_int, [vp, vp, c.c_int, c.c_int, P(vp)]),
        "malcs_comm_destroy": (c.c_int, [vp]),
        "malcs_comm_info": (c.c_int, [vp, P(c.c_int), P(c.c_int)]),
        "malcs_comm_all_gather": (c.c_int, [vp, vp, vp, u64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


# Handles still open at interpreter exit are released here, in dependency order, while the HIP
# runtime is still alive (its own teardown runs after Python's).
_live = {"graphs": weakref.WeakSet(), "relmatrix": weakref.WeakSet(), "batch": weakref.WeakSet(), "comm": weakref.WeakSet(),
         "ctx": weakref.WeakSet()}


@atexit.register
def _close_all():
    for kind in ("graphs", "relmatrix", "batch", "comm", "ctx"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass


def check(rc):
    if rc != MALCS_OK:
        raise MalcsError(f"libmalcs error {rc}: {lib.malcs_last_error().decode()}")


def ptr(array):
    return array.ctypes.data_as(ctypes.c_void_p)


class Context:
    """One libmalcs context = one GPU (one process per GPU)."""

    def __init__(self, device=0):
        self.handle = ctypes.c_void_p()
        check(lib.malcs_ctx_create(int(device), ctypes.byref(self.handle)))
        self.device = int(device)
        _live["ctx"].add(self)

    def close(self):
        if self.handle:
            lib.malcs_ctx_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def sync(self):
        check(lib.malcs_ctx_sync(self.handle))

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cus, mhz = ctypes.c_int(), ctypes.c_int()
        check(lib.malcs_ctx_device_info(self.handle, name, 256, ctypes.byref(cus), ctypes.byref(mhz)))
        return name.value.decode(), cus.value, mhz.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Context on the device this process owns (MALCS_DEVICE, else LOCAL_RANK, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        device = int(os.environ.get("MALCS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(device)
    return _default_ctx


_device_contexts = {}


def contexts_for(devices=None):
    """One context per entry of `devices` (a device may be named twice: two contexts, two streams); None without a
    device list (argument or MALCS_DEVICES=0,1,...): the caller then uses the process's default context."""
    if devices is None:
        env = os.environ.get("MALCS_DEVICES", "")
        if not env.strip():
            return None
        devices = [int(x) for x in env.split(",") if x.strip()]
    out = []
    for slot, device in enumerate(devices):
        key = (slot, int(device))
        ctx = _device_contexts.get(key)
        if ctx is None or not ctx.handle:
            ctx = _device_contexts[key] = Context(int(device))
        out.append(ctx)
    return out


def run_on_devices(fn, jobs):
    """fn(job) for every job, one host thread each (the library releases the GIL inside its calls and binds the
    thread to the job's device); the first exception is re-raised after all have finished."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        futures = [pool.submit(fn, job) for job in jobs]
        errors = [f.exception() for f in futures]
    for e in errors:
        if e is not None:
            raise e


class Batch:
    """Sequence pairs resident in HBM (malcs_batch)."""

    def __init__(self, ctx, syms, a_off, a_len, b_off, b_len, n_symbols):
        self.ctx = ctx
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        a_off = np.ascontiguousarray(a_off, dtype=np.uint64)
        b_off = np.ascontiguousarray(b_off, dtype=np.uint64)
        a_len = np.ascontiguousarray(a_len, dtype=np.uint32)
        b_len = np.ascontiguousarray(b_len, dtype=np.uint32)
        n = len(a_off)
        if not (len(a_len) == len(b_off) == len(b_len) == n):
            raise ValueError("offset/length arrays differ in length")
        self.n_pairs = n
        self.handle = ctypes.c_void_p()
        check(lib.malcs_batch_create(ctx.handle, ptr(syms), syms.size, ptr(a_off), ptr(a_len), ptr(b_off), ptr(b_len),
                                     n, int(n_symbols), ctypes.byref(self.handle)))
        _live["batch"].add(self)

    @classmethod
    def from_packed(cls, ctx, planes, a_off, a_len, b_off, b_len, n_planes, n_dwords=None):
        """A batch from sequences that are bit planes already (malcs_batch_create_packed; include/malcs.h has the
        arena's format).  `planes`: a uint32 array in host memory, or an int = a device address on the context's device
        (then `n_dwords` says how long the arena is).  Offsets in dwords, lengths in symbols."""
        self = cls.__new__(cls)
        self.ctx = ctx
        tables = [np.ascontiguousarray(t, dtype=np.uint32) for t in (a_off, a_len, b_off, b_len)]
        n = len(tables[0])
        if any(len(t) != n for t in tables):
            raise ValueError("offset/length arrays differ in length")
        on_device = isinstance(planes, int)
        if on_device:
            if n_dwords is None:
                raise ValueError("a device arena needs n_dwords")
            where = ctypes.c_void_p(planes)
        else:
            planes = np.ascontiguousarray(planes, dtype=np.uint32)
            n_dwords = planes.size
            where = ptr(planes)
        self.n_pairs = n
        self.handle = ctypes.c_void_p()
        check(lib.malcs_batch_create_packed(ctx.handle, where, int(n_dwords), int(on_device), *[ptr(t) for t in tables],
                                            n, int(n_planes), ctypes.byref(self.handle)))
        _live["batch"].add(self)
        return self

    def close(self):
        if self.handle:
            if self.ctx.handle:          # a batch outliving its context is leaked, not freed through it
                lib.malcs_batch_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def replan(self):
        check(lib.malcs_batch_replan(self.handle))

    def launch_distances(self):
        check(lib.malcs_edit_distance_launch(self.handle))

    def fetch_distances(self):
        out = np.empty(self.n_pairs, dtype=np.int32)
        check(lib.malcs_edit_distance_fetch(self.handle, ptr(out)))
        return out

    def distances(self):
        self.launch_distances()
        return self.fetch_distances()

    def time_distances(self, iters):
        ms = ctypes.c_float()
        check(lib.malcs_edit_distance_time(self.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def time_dominant_kernel(self, iters):
        ms = ctypes.c_float()
        check(lib.malcs_edit_distance_time_kernel(self.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def kernel_durations(self, iters):
        """ms of each of `iters` back-to-back launches of the window kernel (one HIP event pair per launch)."""
        out = np.zeros(int(iters), dtype=np.float32)
        check(lib.malcs_edit_distance_kernel_durations(self.handle, int(iters), ptr(out)))
        return out

    def step_durations(self, iters):
        """ms of the window kernel in each of `iters` whole steps (replan + launch) queued back to back."""
        out = np.zeros(int(iters), dtype=np.float32)
        check(lib.malcs_edit_distance_step_durations(self.handle, int(iters), ptr(out)))
        return out

    def plan_info(self):
        info = (ctypes.c_uint64 * 16)()
        check(lib.malcs_batch_plan_info(self.handle, info))
        keys = ("window_pairs", "window_limbs", "window_threshold", "window_cells", "full_cells", "full_classes",
                "window_tiles", "two_limb_tiles", "pieces", "cut_tiles", "workgroups", "window_cells_unpadded",
                "sample_mean_x16", "sample_sd_x16", "worst_simd", "balanced_cuts")
        return dict(zip(keys, (int(x) for x in info)))

    def plan_dump(self, what):
        """One of the arrays of the window class's plan (malcs_batch_plan_dump), as uint32."""
        n = ctypes.c_uint64()
        check(lib.malcs_batch_plan_dump(self.handle, int(what), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            check(lib.malcs_batch_plan_dump(self.handle, int(what), ptr(out), n.value, ctypes.byref(n)))
        return out

    def lcs_nodes(self, shift=None, max_distance=None):
        """_lcs_nodes(reference_i, observed_i, shift_i, max_distance_i) for every pair: the reference's
        complete level lists (Graphs handle with runs only)."""
        return self.graphs(shift, max_distance, _entry=lib.malcs_lcs_nodes_batch)

    def graphs(self, shift=None, max_distance=None, _entry=None):
        """LCSgraph(reference_i, observed_i, shift_i, max_distance_i) for every pair."""
        sh = md = None
        if shift is not None:
            sh = np.ascontiguousarray(shift, dtype=np.int64)
        if max_distance is not None:
            md = np.ascontiguousarray(max_distance, dtype=np.int64)
        handle = ctypes.c_void_p()
        if _entry is None and self.n_pairs >= 4096:
            # graphs and their columns in one call: the pass over the handles runs on the context's threads
            columns = Graphs.empty_columns(self.n_pairs)
            check(lib.malcs_lcs_graph_batch_columns(self.handle, ptr(sh) if sh is not None else None, ptr(md) if md is not None else None,
                                                    ctypes.byref(handle), *(ptr(a) for a in columns)))
            return Graphs(handle, self.n_pairs, columns=columns)
        check((_entry or lib.malcs_lcs_graph_batch)(self.handle, ptr(sh) if sh is not None else None,
                                                     ptr(md) if md is not None else None, ctypes.byref(handle)))
        return Graphs(handle, self.n_pairs)


class Graphs:
    """Result handle of malcs_lcs_graph_batch, read out into numpy arrays."""

    @staticmethod
    def empty_columns(n_pairs):
        """counts, status, distance, supremal of n_pairs pairs.  Large handles read their columns into recycled pinned
        blocks: pages that exist already (fresh numpy arrays of 100 000 pairs cost 0.5 ms of page faults)."""
        empty = pinned_empty if n_pairs >= 4096 else np.empty
        return (empty((n_pairs, 4), dtype=np.int64), empty(n_pairs, dtype=np.int32), empty(n_pairs, dtype=np.int64),
                empty((n_pairs, 4), dtype=np.int64))

    def __init__(self, handle, n_pairs, owner=None, columns=None):
        self.handle = handle
        self.n_pairs = n_pairs
        self.owner = owner                       # set: the handle belongs to that object (a round of VariantGraphs) and lives with it
        _live["graphs"].add(self)
        try:
            # columns: filled already by the call that made the handle; else in one pass over the pairs here
            self.counts, self.status, self.distance, self.supremal = columns or self.empty_columns(n_pairs)
            self.timing = np.zeros(3, dtype=np.float32)
            if n_pairs and columns is None:
                check(lib.malcs_graphs_columns(handle, ptr(self.counts), ptr(self.status), ptr(self.distance), ptr(self.supremal)))
            check(lib.malcs_graphs_timing(handle, ptr(self.timing)))
            built, ms = ctypes.c_uint64(), ctypes.c_float()
            check(lib.malcs_graphs_device_built(handle, ctypes.byref(built), ctypes.byref(ms)))
            self.device_built, self.device_build_ms = built.value, ms.value
            tb, tp = ctypes.c_uint64(), ctypes.c_uint64()
            check(lib.malcs_graphs_tie_bytes(handle, ctypes.byref(tb), ctypes.byref(tp)))
            self.tie_bytes, self.tie_pairs = tb.value, tp.value
        except Exception:
            self.close()
            raise

    def runs(self, i):
        out = np.zeros((int(self.counts[i, 0]), 4), dtype=np.int64)
        check(lib.malcs_graphs_runs(self.handle, i, ptr(out)))
        return out

    def nodes(self, i):
        out = np.zeros((int(self.counts[i, 2]), 3), dtype=np.int64)
        check(lib.malcs_graphs_nodes(self.handle, i, ptr(out)))
        return out

    def edges(self, i):
        out = np.zeros((int(self.counts[i, 3]), 6), dtype=np.int64)
        check(lib.malcs_graphs_edges(self.handle, i, ptr(out)))
        return out

    def atoms(self, index, obs, obs_off, out_range, out):
        """Atom masks of the graphs `index` of this handle into out[out_range[k, 0]:out_range[k, 1]] (malcs_graphs_atoms)."""
        index = np.ascontiguousarray(index, dtype=np.uint64)
        out_range = np.ascontiguousarray(out_range, dtype=np.uint64)
        check(lib.malcs_graphs_atoms(self.handle, len(index), ptr(index), ptr(obs), ptr(obs_off), ptr(out_range), ptr(out)))

    def canonical(self, i):
        n = ctypes.c_int64()
        check(lib.malcs_graphs_canonical_count(self.handle, i, ctypes.byref(n)))
        out = np.zeros((n.value, 4), dtype=np.int64)
        check(lib.malcs_graphs_canonical(self.handle, i, ptr(out)))
        return out

    def close(self):
        if self.handle:
            if self.owner is None:
                lib.malcs_graphs_destroy(self.handle)
            self.handle = ctypes.c_void_p()
            self.owner = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VariantGraphs:
    """Result handle of malcs_from_variant_batch: the supremal LCS graphs of n alleles of one reference, as arrays."""

    def __init__(self, ctx, ref, start, end, ins, ins_off, offset, n_symbols):
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        start = np.ascontiguousarray(start, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        ins = np.ascontiguousarray(ins, dtype=np.uint8)
        ins_off = np.ascontiguousarray(ins_off, dtype=np.uint64)
        n = len(start)
        if len(end) != n or len(ins_off) != n + 1:
            raise ValueError("start / end / ins_off differ in length")
        self.n = n
        self.handle = ctypes.c_void_p()
        self._rounds = {}
        check(lib.malcs_from_variant_batch(ctx.handle, ptr(ref), ref.size, n, ptr(start), ptr(end), ptr(ins), ptr(ins_off), int(offset),
                                           int(n_symbols), ctypes.byref(self.handle)))
        _live["graphs"].add(self)
        self.where = np.empty((n, 4), dtype=np.int64)
        self.distance = np.empty(n, dtype=np.int64)
        self.supremal = np.empty((n, 4), dtype=np.int64)
        self.obs_off = np.empty(n + 1, dtype=np.uint64)
        check(lib.malcs_vgraphs_columns(self.handle, ptr(self.where), ptr(self.distance), ptr(self.supremal), ptr(self.obs_off)))
        self.observed = np.empty(int(self.obs_off[-1]) + 1, dtype=np.uint8)
        check(lib.malcs_vgraphs_observed(self.handle, ptr(self.observed)))
        count = ctypes.c_int()
        check(lib.malcs_vgraphs_rounds(self.handle, ctypes.byref(count)))
        self.n_rounds = count.value

    def round(self, r):
        """The Graphs handle of widening round r (nodes / edges / canonical of the graphs that ended there)."""
        made = self._rounds.get(r)
        if made is None:
            handle, size = ctypes.c_void_p(), ctypes.c_uint64()
            check(lib.malcs_vgraphs_round(self.handle, int(r), ctypes.byref(handle), ctypes.byref(size)))
            made = self._rounds[r] = Graphs(handle, size.value, owner=self)
        return made

    def matrix_columns(self, with_atoms):
        """key, symmask, folded, sup_off, sup, atom_off, atoms (malcs_vgraphs_matrix_columns)."""
        n = self.n
        key = np.empty(n, dtype=np.uint32)
        symmask = np.empty(n, dtype=np.uint64)
        sup_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(self.supremal[:, 3] - self.supremal[:, 2], out=sup_off[1:])
        sup = np.empty(int(sup_off[-1]) + 1, dtype=np.uint8)
        atom_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(self.supremal[:, 1] - self.supremal[:, 0] + 1, out=atom_off[1:])
        atoms = np.zeros(int(atom_off[-1]), dtype=np.uint8) if with_atoms else None
        folded = ctypes.c_int()
        check(lib.malcs_vgraphs_matrix_columns(self.handle, ptr(key), ptr(symmask), ctypes.byref(folded), ptr(sup_off), ptr(sup),
                                               ptr(atom_off), ptr(atoms) if with_atoms else None))
        return key, symmask, bool(folded.value), sup_off, sup[:-1], atom_off, atoms

    def close(self):
        if self.handle:
            for made in self._rounds.values():
                made.close()
            self._rounds = {}
            lib.malcs_vgraphs_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def graphs_from_runs(m, n, shift, n_levels, runs4, run_first):
    """_build_graph on caller-provided level lists (host only): Graphs handle of len(m) pairs."""
    m = np.ascontiguousarray(m, dtype=np.int64)
    n = np.ascontiguousarray(n, dtype=np.int64)
    shift = np.ascontiguousarray(shift, dtype=np.int64)
    n_levels = np.ascontiguousarray(n_levels, dtype=np.int64)
    runs4 = np.ascontiguousarray(runs4, dtype=np.int64).reshape(-1, 4)
    run_first = np.ascontiguousarray(run_first, dtype=np.uint64)
    handle = ctypes.c_void_p()
    check(lib.malcs_graphs_from_runs(len(m), ptr(m), ptr(n), ptr(shift), ptr(n_levels), ptr(runs4), ptr(run_first), ctypes.byref(handle)))
    return Graphs(handle, len(m))


def comm_unique_id():
    """The 128 bytes rank 0 hands to the other ranks of a job (ncclGetUniqueId)."""
    buf = ctypes.create_string_buffer(128)
    check(lib.malcs_comm_unique_id(buf))
    return buf.raw


class Comm:
    """This rank's end of the job's RCCL communicator (malcs_comm): one per process, on its context."""

    def __init__(self, ctx, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("the communicator id has 128 bytes")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.handle = ctypes.c_void_p()
        check(lib.malcs_comm_create(ctx.handle, ctypes.c_char_p(unique_id), self.rank, self.world, ctypes.byref(self.handle)))
        _live["comm"].add(self)

    def close(self):
        if self.handle:
            if self.ctx.handle:
                lib.malcs_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """One block of the library's recycled pinned host memory (malcs_host_alloc); it goes back to the library when
    the last array over it is gone."""

    def __init__(self, nbytes):
        address = ctypes.c_void_p()
        check(lib.malcs_host_alloc(nbytes, ctypes.byref(address)))
        self.address = address.value
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.address, False), "version": 3}

    def __del__(self):
        try:
            lib.malcs_host_free(ctypes.c_void_p(self.address))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """An uninitialised numpy array in pinned host memory: what a large device result is copied into at the link's
    speed (a pageable array of 25 MB costs 2.5 ms, this one 0.5)."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    try:
        block = _PinnedBlock(max(count * dtype.itemsize, 1))
    except MalcsError:                      # pinned memory is a scarce kind: a caller who holds many results gets pageable ones
        return np.empty(shape, dtype=dtype)
    return np.asarray(block)[:count * dtype.itemsize].view(dtype).reshape(shape)


class RelationMatrix:
    """This rank's tiles of the all-vs-all relation matrix on the device (malcs_relmatrix)."""

    def __init__(self, ctx, start, end, key, symmask, tile, rank=0, world=1):
        self.ctx = ctx
        start = np.ascontiguousarray(start, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        symmask = np.ascontiguousarray(symmask, dtype=np.uint64)
        self.n = len(start)
        if not (len(end) == len(key) == len(symmask) == self.n):
            raise ValueError("per-variant arrays differ in length")
        self.tile, self.rank, self.world = int(tile), int(rank), int(world)
        blocks = (self.n + self.tile - 1) // self.tile
        self.n_tiles = blocks * (blocks + 1) // 2
        self.per_rank = (self.n_tiles + self.world - 1) // self.world
        self.handle = ctypes.c_void_p()
        check(lib.malcs_relmatrix_create(ctx.handle, self.n, ptr(start), ptr(end), ptr(key), ptr(symmask), self.tile,
                                         self.rank, self.world, ctypes.byref(self.handle)))
        _live["relmatrix"].add(self)

    def pending(self):
        """Undecided pairs as an (k, 2) int64 array of (i, j), i < j."""
        count = ctypes.c_uint64()
        check(lib.malcs_relmatrix_pending_count(self.handle, ctypes.byref(count)))
        pairs = np.zeros((count.value, 2), dtype=np.int64)
        check(lib.malcs_relmatrix_pending(self.handle, ptr(pairs)))
        return pairs

    def decide(self, ref, sup, sup_off, distance, atoms, atom_off, n_symbols):
        """The undecided pairs decided on the device (union windows, distances, distance arithmetic, atom masks);
        returns how many pairs reached the DP."""
        n_dp = ctypes.c_uint64()
        keep = [np.ascontiguousarray(ref, dtype=np.uint8), np.ascontiguousarray(sup, dtype=np.uint8),
                np.ascontiguousarray(sup_off, dtype=np.uint64), np.ascontiguousarray(distance, dtype=np.int64),
                np.ascontiguousarray(atoms, dtype=np.uint8), np.ascontiguousarray(atom_off, dtype=np.uint64)]
        check(lib.malcs_relmatrix_decide(self.handle, ptr(keep[0]), keep[0].size, ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]),
                                         ptr(keep[5]), int(n_symbols), ctypes.byref(n_dp)))
        return n_dp.value

    def resolve(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        check(lib.malcs_relmatrix_resolve(self.handle, ptr(codes)))

    def tiles(self):
        out = np.zeros((self.per_rank, self.tile, self.tile), dtype=np.uint8)
        check(lib.malcs_relmatrix_tiles(self.handle, ptr(out)))
        return out

    def tiles_device(self):
        """(device address, bytes) of this rank's label block: what a device-side all-gather sends."""
        address, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
        check(lib.malcs_relmatrix_tiles_device(self.handle, ctypes.byref(address), ctypes.byref(nbytes)))
        return address.value, nbytes.value

    def assemble_device(self, address, nbytes):
        """`assemble` from gathered blocks that already sit on this handle's device (a completed collective)."""
        if nbytes != self.world * self.per_rank * self.tile * self.tile:
            raise ValueError("gathered label block has the wrong size")
        full = pinned_empty((self.n, self.n))        # every cell is written by the assembly
        check(lib.malcs_relmatrix_assemble_device(self.handle, ctypes.c_void_p(address), ptr(full)))
        return full

    def assemble_comm(self, comm):
        """All-gather of every rank's label block over RCCL (the library's own collective) + assembly."""
        full = pinned_empty((self.n, self.n))        # every cell is written by the assembly
        check(lib.malcs_relmatrix_assemble_comm(self.handle, comm.handle, ptr(full)))
        return full

    def assemble(self, gathered=None):
        full = pinned_empty((self.n, self.n))        # every cell is written by the assembly
        if gathered is not None:
            gathered = np.ascontiguousarray(gathered, dtype=np.uint8)
            if gathered.size != self.world * self.per_rank * self.tile * self.tile:
                raise ValueError("gathered label block has the wrong size")
        check(lib.malcs_relmatrix_assemble(self.handle, ptr(gathered) if gathered is not None else None, ptr(full)))
        return full

    def close(self):
        if self.handle:
            if self.ctx.handle:
                lib.malcs_relmatrix_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
