"""ctypes binding of libautomerge_hip.so (include/automerge_hip.h) and a host-side mirror of the
reference's `Automerge` bulk API (rust/automerge/src/automerge.rs: load :552, get_heads :1300,
get_changes :1306, hydrate :1415) over batches of documents.

There is no CPU fallback: if the HIP library is missing, importing the engine raises."""
import ctypes as C
import os
import sys
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libautomerge_hip.so")

_u8p = C.POINTER(C.c_uint8)


class BytesOut(C.Structure):
    _fields_ = [("data", _u8p), ("len", C.c_size_t)]


class ChangeView(C.Structure):
    _fields_ = [("hash", _u8p), ("raw_bytes", _u8p), ("raw_len", C.c_size_t), ("actor", _u8p), ("actor_len", C.c_size_t),
                ("seq", C.c_uint64), ("start_op", C.c_uint64), ("num_ops", C.c_uint64), ("timestamp", C.c_int64),
                ("num_deps", C.c_uint32)]


class SaveOpts(C.Structure):
    _fields_ = [("deflate", C.c_int), ("retain_orphans", C.c_int)]


class LoadOpts(C.Structure):
    _fields_ = [("verify_heads", C.c_int), ("on_partial_load", C.c_int)]


class SyncHaveC(C.Structure):
    _fields_ = [("last_sync32", C.c_char_p), ("n_last_sync", C.c_size_t), ("bloom", C.c_char_p), ("bloom_len", C.c_size_t)]


class SyncStateC(C.Structure):
    _fields_ = [("doc", C.c_uint32), ("flags", C.c_uint32),
                ("shared_heads32", C.c_char_p), ("n_shared_heads", C.c_size_t),
                ("last_sent_heads32", C.c_char_p), ("n_last_sent_heads", C.c_size_t),
                ("their_heads32", C.c_char_p), ("n_their_heads", C.c_size_t),
                ("their_need32", C.c_char_p), ("n_their_need", C.c_size_t),
                ("their_have", C.POINTER(SyncHaveC)), ("n_their_have", C.c_size_t),
                ("sent_hashes32", C.c_char_p), ("n_sent_hashes", C.c_size_t)]


class BatchStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_docs", "n_ok", "ops_loaded", "op_rows", "succ_entries", "changes",
                                          "delete_ops", "encoded_bytes", "change_bytes")]


EXPORTS = [
    "amh_ctx_create", "amh_ctx_destroy", "amh_last_error", "amh_stage_batch", "amh_staged_free", "amh_load_staged",
    "amh_load_batch", "amh_batch_free", "amh_bytes_free", "amh_doc_status", "amh_batch_get_stats", "amh_heads",
    "amh_changes", "amh_hydrate_json", "amh_doc_max_op", "amh_merge_batch", "amh_apply_changes", "amh_save_batch",
    "amh_sync_summaries", "amh_ctx_set_profiling", "amh_ctx_num_kernels", "amh_ctx_kernel_name", "amh_ctx_kernel_time",
    "amh_ctx_reset_profile", "amh_gen_map_docs", "amh_gen_text_docs", "amh_u64_free", "amh_apply_batch", "amh_doc_queue_len", "amh_merge_many", "amh_save_after",
    "amh_sync_summaries_noalloc", "amh_ctx_pool_bytes", "amh_ctx_set_limits", "amh_batch_release_ops", "amh_comm_unique_id", "amh_comm_init", "amh_sync_summaries_all",
    "amh_partition_docs", "amh_text_batch", "amh_text_batch_device", "amh_generate_sync_messages", "amh_generate_sync_messages_device", "amh_save_after_batch",
]

AMH_ERR_DOC_FAILED = -7

DOC_ERRORS = {
    0: "Ok", 1: "Parse", 2: "BadChecksum", 3: "InvalidChangeColumns", 5: "MissingDeps", 6: "DuplicateSeqNumber",
    20: "InflateDocument(ReadOp)", 21: "InflateDocument(OpsOutOfOrder)", 22: "InflateDocument(MissingActor)",
    23: "InflateDocument(MismatchingHeads)", 24: "InflateDocument(MissingOps)", 25: "InflateDocument(SuccOutOfOrder)",
    26: "InflateDocument(InvalidOp)", 30: "InvalidChanges(ChangesOutOfOrder)", 31: "InvalidChanges(MissingChange)",
    32: "InvalidChanges(ReadChange)", 33: "InvalidChanges(IncorrectMaxOp)", 34: "InvalidChanges(MissingActor)",
    35: "InvalidChanges(MissingOps)", 90: "TooLarge", 91: "Unsupported",
}


def bind(lib):
    """Attach argtypes/restypes to a loaded library handle."""
    vp = C.c_void_p
    lib.amh_ctx_create.restype = vp
    lib.amh_ctx_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.amh_ctx_destroy.argtypes = [vp]
    lib.amh_last_error.restype = C.c_char_p
    lib.amh_last_error.argtypes = [vp]
    lib.amh_stage_batch.restype = vp
    lib.amh_stage_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]
    lib.amh_staged_free.argtypes = [vp]
    lib.amh_load_staged.restype = vp
    lib.amh_load_staged.argtypes = [vp, vp, C.POINTER(LoadOpts)]
    lib.amh_load_batch.restype = vp
    lib.amh_load_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(LoadOpts)]
    lib.amh_batch_free.argtypes = [vp]
    lib.amh_bytes_free.argtypes = [C.POINTER(BytesOut)]
    lib.amh_doc_status.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    lib.amh_batch_get_stats.argtypes = [vp, C.POINTER(BatchStats)]
    lib.amh_heads.argtypes = [vp, C.c_uint32, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    lib.amh_changes.argtypes = [vp, C.c_uint32, C.POINTER(ChangeView), C.POINTER(C.c_size_t)]
    lib.amh_hydrate_json.argtypes = [vp, C.c_uint32, C.POINTER(BytesOut)]
    lib.amh_doc_max_op.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.amh_ctx_set_profiling.argtypes = [vp, C.c_int]
    lib.amh_ctx_kernel_name.restype = C.c_char_p
    lib.amh_ctx_kernel_name.argtypes = [C.c_int]
    lib.amh_ctx_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.amh_ctx_reset_profile.argtypes = [vp]
    lib.amh_ctx_pool_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.amh_ctx_set_limits.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.amh_batch_release_ops.argtypes = [vp, vp]
    lib.amh_comm_unique_id.argtypes = [C.c_char_p]
    lib.amh_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.amh_sync_summaries_all.argtypes = [vp, vp, C.POINTER(BytesOut), C.POINTER(C.c_size_t)]
    lib.amh_partition_docs.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    lib.amh_gen_map_docs.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                     C.POINTER(BytesOut), C.POINTER(C.POINTER(C.c_uint64))]
    lib.amh_gen_text_docs.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(BytesOut), C.POINTER(C.POINTER(C.c_uint64))]
    lib.amh_u64_free.argtypes = [C.POINTER(C.c_uint64)]
    lib.amh_sync_summaries.argtypes = [vp, vp, C.POINTER(BytesOut)]
    lib.amh_text_batch.argtypes = [vp, vp, C.POINTER(BytesOut), C.POINTER(C.POINTER(C.c_uint64))]
    lib.amh_text_batch_device.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    lib.amh_sync_summaries_noalloc.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.amh_save_batch.argtypes = [vp, vp, C.POINTER(SaveOpts), C.POINTER(BytesOut)]
    lib.amh_merge_batch.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]
    lib.amh_apply_changes.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]
    lib.amh_apply_batch.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int,
                                    C.POINTER(C.c_int)]
    lib.amh_doc_queue_len.argtypes = [vp, C.c_uint32, C.POINTER(C.c_size_t)]
    lib.amh_save_after.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(BytesOut)]
    lib.amh_save_after_batch.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(BytesOut),
                                         C.POINTER(C.POINTER(C.c_uint64))]
    lib.amh_generate_sync_messages_device.argtypes = [vp, vp, C.POINTER(SyncStateC), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(vp),
                                                      C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    lib.amh_generate_sync_messages.argtypes = [vp, vp, C.POINTER(SyncStateC), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(BytesOut),
                                               C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(BytesOut), C.POINTER(C.POINTER(C.c_uint64))]
    lib.amh_merge_many.argtypes = [vp, vp, C.POINTER(vp), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]
    return lib


def load_library(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the engine.")
    _share_torch_hip_runtime()
    return bind(C.CDLL(path))


def _share_torch_hip_runtime(names=("libhsa-runtime64.so", "libamdhip64.so")):
    """One HIP runtime per process.  The PyTorch ROCm wheel carries its own libamdhip64.so.7 / libhsa-runtime64.so.1; if
    this library is loaded first it pulls in /opt/rocm's copies, and a later `import torch` in the same process then
    finds no GPU.  Loading the wheel's copies first (same sonames, so this library's DT_NEEDED entries bind to them) makes
    the order of imports irrelevant.  torch itself is not imported; without a torch wheel nothing happens."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in names:
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


class AutomergeError(Exception):
    """Per-document failure, carrying the reference's error kind (automerge/src/error.rs:7-62)."""

    def __init__(self, kind, msg):
        self.kind = kind
        self.kind_name = DOC_ERRORS.get(kind, str(kind))
        super().__init__(f"{self.kind_name}: {msg}")


class EngineError(Exception):
    pass


def parse_sync_summaries(raw):
    import struct
    out, pos = [], 0
    while pos < len(raw):
        doc, status, nh = struct.unpack_from("<III", raw, pos)
        pos += 12
        heads = [raw[pos + 32 * k:pos + 32 * k + 32] for k in range(nh)]
        pos += 32 * nh
        (bl,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        out.append(dict(doc=doc, status=status, heads=heads, bloom=raw[pos:pos + bl]))
        pos += bl
    return out


class Batch:
    """N loaded documents (amh_batch)."""

    def __init__(self, engine, handle, n):
        self.engine = engine
        self.h = handle
        self.n = n
        engine._children.add(self)

    def close(self):
        if self.h:
            self.engine.lib.amh_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n

    def sync_summaries_all(self):
        """every rank's sync summaries on every rank (all-gather on device buffers inside the library, RCCL):
        the list of per-rank record buffers"""
        world = getattr(self.engine, "world", 1)
        o = BytesOut()
        offs = (C.c_size_t * (world + 1))()
        rc = self.engine.lib.amh_sync_summaries_all(self.engine.ctx, self.h, C.byref(o), offs)
        if rc != 0:
            raise EngineError("amh_sync_summaries_all failed: " + self.engine._err())
        raw = C.string_at(o.data, o.len)
        self.engine.lib.amh_bytes_free(C.byref(o))
        return [raw[offs[r]:offs[r + 1]] for r in range(world)]

    def release_ops(self):
        """keep the history (changes, hashes, heads), give the per-op arrays back to the pool: enough for a merge source"""
        rc = self.engine.lib.amh_batch_release_ops(self.engine.ctx, self.h)
        if rc != 0:
            raise EngineError(f"amh_batch_release_ops failed: {rc}")

    def status(self, i):
        kind = C.c_int()
        msg = C.c_char_p()
        rc = self.engine.lib.amh_doc_status(self.h, i, C.byref(kind), C.byref(msg))
        if rc != 0:
            raise EngineError(f"amh_doc_status failed: {rc}")
        return kind.value, (msg.value or b"").decode()

    def check(self, i):
        kind, msg = self.status(i)
        if kind != 0:
            raise AutomergeError(kind, msg)

    def get_heads(self, i):
        self.check(i)
        p = _u8p()
        n = C.c_size_t()
        rc = self.engine.lib.amh_heads(self.h, i, C.byref(p), C.byref(n))
        if rc != 0:
            raise EngineError(f"amh_heads failed: {rc}")
        raw = C.string_at(p, 32 * n.value)
        return [raw[k:k + 32] for k in range(0, len(raw), 32)]

    def get_changes(self, i):
        self.check(i)
        n = C.c_size_t(0)
        rc = self.engine.lib.amh_changes(self.h, i, None, C.byref(n))
        if rc != 0:
            raise EngineError(f"amh_changes failed: {rc}")
        arr = (ChangeView * max(1, n.value))()
        cap = C.c_size_t(n.value)
        rc = self.engine.lib.amh_changes(self.h, i, arr, C.byref(cap))
        if rc != 0:
            raise EngineError(f"amh_changes failed: {rc}")
        out = []
        for k in range(cap.value):
            v = arr[k]
            out.append(dict(hash=C.string_at(v.hash, 32), raw=C.string_at(v.raw_bytes, v.raw_len),
                            actor=C.string_at(v.actor, v.actor_len), seq=v.seq, start_op=v.start_op, len=v.num_ops,
                            timestamp=v.timestamp, num_deps=v.num_deps))
        return out

    def hydrate_json(self, i):
        self.check(i)
        o = BytesOut()
        rc = self.engine.lib.amh_hydrate_json(self.h, i, C.byref(o))
        if rc != 0:
            raise EngineError(f"amh_hydrate_json failed: {rc}")
        s = C.string_at(o.data, o.len).decode("utf-8")
        self.engine.lib.amh_bytes_free(C.byref(o))
        return s

    def max_op(self, i):
        self.check(i)
        v = C.c_uint64()
        rc = self.engine.lib.amh_doc_max_op(self.h, i, C.byref(v))
        if rc != 0:
            raise EngineError(f"amh_doc_max_op failed: {rc}")
        return v.value

    def merge(self, other, pairs):
        """== self[a].merge(other[b]) for every (a, b) in pairs (automerge.rs:917).  Returns the call status: 0, or
        -7 when some pairs were left unmerged (failed documents, duplicate seq numbers)."""
        n = len(pairs)
        a = (C.c_uint32 * max(1, n))(*[p[0] for p in pairs])
        b = (C.c_uint32 * max(1, n))(*[p[1] for p in pairs])
        rc = self.engine.lib.amh_merge_batch(self.engine.ctx, self.h, other.h, a, b, n)
        if rc not in (0, -7):
            raise EngineError(f"amh_merge_batch failed: {rc} {self.engine._err()}")
        return rc

    def merge_many(self, others, dst_idx, src_idx):
        """== self[dst_idx[p]].merge(others[0][src_idx[0][p]]); .merge(others[1][...]); ... in one pass over the rows.
        src_idx[s][p] = document of others[s] (None = nothing from that source).  Returns 0 or -7 like merge."""
        ns, n = len(others), len(dst_idx)
        hs = (C.c_void_p * max(1, ns))(*[o.h for o in others])
        if hasattr(dst_idx, "ctypes") and hasattr(src_idx, "ctypes"):
            # numpy uint32 arrays (dst_idx[n], src_idx[ns, n], 0xFFFFFFFF = nothing from that source) go to the C ABI as they
            # are: no Python object per pair (10,000 documents x 7 sources are 70,000 of them)
            import numpy as np
            dst_idx, src_idx = np.ascontiguousarray(dst_idx, dtype=np.uint32), np.ascontiguousarray(src_idx, dtype=np.uint32)
            assert src_idx.shape == (ns, n)
            a, b = dst_idx.ctypes.data_as(C.POINTER(C.c_uint32)), src_idx.ctypes.data_as(C.POINTER(C.c_uint32))
        else:
            a = (C.c_uint32 * max(1, n))(*dst_idx)
            flat = [0xFFFFFFFF if src_idx[s][p] is None else src_idx[s][p] for s in range(ns) for p in range(n)]
            b = (C.c_uint32 * max(1, ns * n))(*flat)
        rc = self.engine.lib.amh_merge_many(self.engine.ctx, self.h, hs, ns, a, b, n)
        if rc not in (0, -7):
            raise EngineError(f"amh_merge_many failed: {rc} {self.engine._err()}")
        return rc

    def apply_changes(self, i, changes):
        """== self[i].apply_changes(changes) (automerge.rs:785): `changes` = change chunks (Change::raw_bytes).
        Raises AutomergeError with the reference's kind when the call fails for the document."""
        res = self.apply_batch([(i, b"".join(bytes(c) for c in changes))])
        if res and res[0]:
            raise AutomergeError(res[0], "apply_changes")

    def apply_batch(self, items, incremental=False):
        """items = [(document, bytes)]: every buffer holds one or more chunks and is one apply_changes call on that document.
        incremental=False is apply_changes per document; True is load_incremental (automerge.rs:730-769: chunks in
        front of the first unreadable one are applied).  Returns the per-item outcome (0 or the AutomergeError kind)."""
        n = len(items)
        if n == 0:
            return []
        idx = (C.c_uint32 * n)(*[it[0] for it in items])
        bufs = (C.c_char_p * n)(*[bytes(it[1]) for it in items])
        lens = (C.c_size_t * n)(*[len(it[1]) for it in items])
        res = (C.c_int * n)()
        rc = self.engine.lib.amh_apply_batch(self.engine.ctx, self.h, idx, bufs, lens, n, 1 if incremental else 0, res)
        if rc not in (0, -7):
            raise EngineError(f"amh_apply_batch failed: {rc} {self.engine._err()}")
        return list(res)

    def apply_arena(self, doc_idx, blob, offsets, incremental=False):
        """apply_batch with the buffers lying back to back in `blob` (bytes): buffer i = blob[offsets[i]:offsets[i + 1]]
        goes to document doc_idx[i].  The pointer table is built with numpy instead of one Python object per buffer
        (a receive arena, as a sync server holds the messages of a round)."""
        import numpy as np
        idx = np.ascontiguousarray(doc_idx, dtype=np.uint32)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = int(idx.shape[0])
        if n == 0:
            return []
        if offs.shape[0] != n + 1 or int(offs[-1]) > len(blob):
            raise ValueError("offsets must have len(doc_idx) + 1 entries inside blob")
        keep = C.c_char_p(blob)
        base = C.cast(keep, C.c_void_p).value or 0
        ptrs = (offs[:n] + np.uint64(base)).astype(np.uint64)
        lens = (offs[1:] - offs[:n]).astype(np.uint64)
        res = np.zeros(n, dtype=np.int32)
        rc = self.engine.lib.amh_apply_batch(self.engine.ctx, self.h, C.cast(idx.ctypes.data, C.POINTER(C.c_uint32)),
                                             C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p)), C.cast(lens.ctypes.data, C.POINTER(C.c_size_t)),
                                             n, 1 if incremental else 0, C.cast(res.ctypes.data, C.POINTER(C.c_int)))
        if rc not in (0, -7):
            raise EngineError(f"amh_apply_batch failed: {rc} {self.engine._err()}")
        return res.tolist()

    def save_after(self, i, heads):
        """== self[i].save_after(heads) (automerge.rs:993): the changes `heads` has not seen, as change chunks."""
        self.check(i)
        o = BytesOut()
        rc = self.engine.lib.amh_save_after(self.h, i, b"".join(heads), len(heads), C.byref(o))
        if rc != 0:
            raise EngineError(f"amh_save_after failed: {rc}")
        raw = C.string_at(o.data, o.len) if o.len else b""
        self.engine.lib.amh_bytes_free(C.byref(o))
        return raw

    def queue_len(self, i):
        n = C.c_size_t()
        rc = self.engine.lib.amh_doc_queue_len(self.h, i, C.byref(n))
        if rc != 0:
            raise EngineError(f"amh_doc_queue_len failed: {rc}")
        return n.value

    def save_all(self, deflate=True):
        """== Automerge::save_with_options per document (automerge.rs:943); None for documents without a saved form."""
        opts = SaveOpts(1 if deflate else 0, 1)
        outs = (BytesOut * max(1, self.n))()
        rc = self.engine.lib.amh_save_batch(self.engine.ctx, self.h, C.byref(opts), outs)
        if rc != 0:
            raise EngineError(f"amh_save_batch failed: {rc} {self.engine._err()}")
        res = []
        for i in range(self.n):
            res.append(C.string_at(outs[i].data, outs[i].len) if outs[i].data else None)
            self.engine.lib.amh_bytes_free(C.byref(outs[i]))
        return res

    def text_all(self, decode=True):
        """amh_text_batch: the text of every document (ReadDoc::text of its text objects, back to back).  decode=False
        returns (bytes, offsets) without building Python strings."""
        o = BytesOut()
        offs = C.POINTER(C.c_uint64)()
        rc = self.engine.lib.amh_text_batch(self.engine.ctx, self.h, C.byref(o), C.byref(offs))
        if rc != 0:
            raise EngineError(f"amh_text_batch failed: {rc} {self.engine._err()}")
        n = self.n
        raw = C.string_at(o.data, o.len) if o.len else b""
        ofs = [offs[i] for i in range(n + 1)] if decode else None
        total = offs[n]
        self.engine.lib.amh_bytes_free(C.byref(o))
        self.engine.lib.amh_u64_free(offs)
        if not decode:
            return raw, total
        return [raw[ofs[i]:ofs[i + 1]].decode("utf-8") for i in range(n)]

    def text_device(self):
        """amh_text_batch_device: the texts stay in device memory; returns (device address of the texts, device address of
        the u32 offsets, total bytes)"""
        d_text, d_off, total = C.c_void_p(), C.c_void_p(), C.c_uint64()
        rc = self.engine.lib.amh_text_batch_device(self.engine.ctx, self.h, C.byref(d_text), C.byref(d_off), C.byref(total))
        if rc != 0:
            raise EngineError(f"amh_text_batch_device failed: {rc} {self.engine._err()}")
        return d_text.value, d_off.value, total.value

    def save_after_many(self, items):
        """Automerge::save_after for every (document index, heads) of `items` in one device pass (amh_save_after_batch)"""
        n = len(items)
        docs = (C.c_uint32 * max(n, 1))(*[d for d, _ in items])
        raws = [b"".join(h) for _, h in items]
        ptrs = (C.c_char_p * max(n, 1))(*raws)
        cnts = (C.c_size_t * max(n, 1))(*[len(h) for _, h in items])
        o, off = BytesOut(), C.POINTER(C.c_uint64)()
        rc = self.engine.lib.amh_save_after_batch(self.engine.ctx, self.h, docs, ptrs, cnts, n, C.byref(o), C.byref(off))
        if rc != 0 and rc != AMH_ERR_DOC_FAILED:
            raise EngineError(f"amh_save_after_batch failed: {rc} {self.engine._err()}")
        raw = C.string_at(o.data, o.len) if o.data and o.len else b""
        out = [raw[off[i]:off[i + 1]] for i in range(n)]
        self.engine.lib.amh_bytes_free(C.byref(o))
        self.engine.lib.amh_u64_free(off)
        return out

    def generate_sync_messages(self, states):
        """SyncDoc::generate_sync_message (sync.rs:134-221) for every (document index, SyncState) of `states` in one
        pass on the device.  Returns the encoded messages (None where the reference returns None; the AutomergeError of
        the load for a record whose document failed to load) and updates the States the way sync.rs:207-219 does."""
        n = len(states)
        arr, _keep = self.sync_state_array(states)
        kinds, raw, moff, sent, soff = self.generate_sync_messages_raw(arr, n)
        out = []
        for k in range(n):
            if kinds[k] == 3:   # the document failed to load: there is no Automerge to ask (the error its load gave)
                out.append(AutomergeError(*self.status(states[k][0])))
                continue
            m = raw[moff[k]:moff[k + 1]] if kinds[k] in (1, 2) else None
            if kinds[k] == 1:
                st = states[k][1]
                st.last_sent_heads = SyncMessage.decode(m).heads
                st.sent_hashes.update(sent[i:i + 32] for i in range(soff[k], soff[k + 1], 32))
                st.in_flight = True
            out.append(m)
        return out

    @staticmethod
    def sync_state_array(states):
        """[(document index, SyncState)] as the amh_sync_state array of the C ABI (+ the objects that keep its pointers alive)"""
        n = len(states)
        arr = (SyncStateC * max(n, 1))()
        keep = []
        for k, (doc, st) in enumerate(states):
            c = arr[k]
            c.doc = doc
            c.flags = ((1 if st.their_heads is not None else 0) | (2 if st.their_need is not None else 0) |
                       (4 if st.their_have is not None else 0) | (8 if st.in_flight else 0))

            def lst(hs):
                raw = b"".join(hs or [])
                keep.append(raw)
                return raw, len(hs or [])
            c.shared_heads32, c.n_shared_heads = lst(st.shared_heads)
            c.last_sent_heads32, c.n_last_sent_heads = lst(st.last_sent_heads)
            c.their_heads32, c.n_their_heads = lst(st.their_heads)
            c.their_need32, c.n_their_need = lst(st.their_need)
            c.sent_hashes32, c.n_sent_hashes = lst(sorted(st.sent_hashes))
            haves = st.their_have or []
            ha = (SyncHaveC * max(len(haves), 1))()
            for j, h in enumerate(haves):
                ha[j].last_sync32, ha[j].n_last_sync = lst(h.last_sync)
                keep.append(h.bloom)
                ha[j].bloom, ha[j].bloom_len = h.bloom, len(h.bloom)
            keep.append(ha)
            c.their_have, c.n_their_have = ha, len(haves)
        return arr, keep

    def generate_sync_messages_device(self, state_array, n):
        """amh_generate_sync_messages_device: the messages stay in HBM.  (kinds, d_msgs, d_msg_off, total bytes, d_sent,
        d_sent_idx, changes sent), device addresses as integers, valid until the next call on this batch."""
        kinds = (C.c_uint32 * max(n, 1))()
        d_msgs, d_off, d_sent, d_idx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        total, n_sent = C.c_uint64(), C.c_uint64()
        rc = self.engine.lib.amh_generate_sync_messages_device(self.engine.ctx, self.h, state_array, n, kinds, C.byref(d_msgs), C.byref(d_off),
                                                               C.byref(total), C.byref(d_sent), C.byref(d_idx), C.byref(n_sent))
        if rc != 0 and rc != AMH_ERR_DOC_FAILED:
            raise EngineError(f"amh_generate_sync_messages_device failed: {rc} {self.engine._err()}")
        return kinds, d_msgs.value, d_off.value, total.value, d_sent.value, d_idx.value, n_sent.value

    def generate_sync_messages_raw(self, state_array, n, want_sent=True, offsets=True):
        """amh_generate_sync_messages over a prepared SyncStateC array: (kinds, message bytes, offsets, sent hashes, offsets)"""
        kinds = (C.c_uint32 * max(n, 1))()
        msgs, sent = BytesOut(), BytesOut()
        moff, soff = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        rc = self.engine.lib.amh_generate_sync_messages(self.engine.ctx, self.h, state_array, n, kinds, C.byref(msgs), C.byref(moff),
                                                        C.byref(sent) if want_sent else None, C.byref(soff) if want_sent else None)
        raw = C.string_at(msgs.data, msgs.len) if msgs.data and msgs.len else b""
        sh = C.string_at(sent.data, sent.len) if want_sent and sent.data and sent.len else b""
        mo = [moff[i] for i in range(n + 1)] if moff and offsets else None
        so = [soff[i] for i in range(n + 1)] if want_sent and soff and offsets else None
        self.engine.lib.amh_bytes_free(C.byref(msgs))
        self.engine.lib.amh_bytes_free(C.byref(sent))
        if moff:
            self.engine.lib.amh_u64_free(moff)
        if soff:
            self.engine.lib.amh_u64_free(soff)
        if rc != 0 and not (rc == AMH_ERR_DOC_FAILED and offsets):   # failed documents: their records say so (kind 3)
            raise EngineError(f"amh_generate_sync_messages failed: {rc} {self.engine._err()}")
        return (list(kinds[:n]) if offsets else kinds), raw, mo, sh, so

    def sync_summaries_raw(self):
        """All sync summary records of the batch as one bytes object (amh_sync_summaries)."""
        o = BytesOut()
        rc = self.engine.lib.amh_sync_summaries(self.engine.ctx, self.h, C.byref(o))
        if rc != 0:
            raise EngineError(f"amh_sync_summaries failed: {rc} {self.engine._err()}")
        raw = C.string_at(o.data, o.len) if o.len else b""
        self.engine.lib.amh_bytes_free(C.byref(o))
        return raw

    def sync_summaries_noalloc(self, d_out_ptr, cap, d_meta_ptr):
        """amh_sync_summaries_noalloc: records into the caller's device buffer (raw device addresses), nothing allocated."""
        rc = self.engine.lib.amh_sync_summaries_noalloc(self.engine.ctx, self.h, d_out_ptr, cap, d_meta_ptr)
        if rc != 0:
            raise EngineError(f"amh_sync_summaries_noalloc failed: {rc} {self.engine._err()}")

    def sync_summaries(self):
        """Per document: dict(doc, status, heads=[32-byte hashes, sorted], bloom=bytes) — get_heads + make_bloom_filter(last_sync=[])."""
        return parse_sync_summaries(self.sync_summaries_raw())

    def stats(self):
        s = BatchStats()
        rc = self.engine.lib.amh_batch_get_stats(self.h, C.byref(s))
        if rc != 0:
            raise EngineError(f"amh_batch_get_stats failed: {rc}")
        return {name: getattr(s, name) for name, _ in BatchStats._fields_}


class SyncHave:
    """sync::Have (sync/state.rs:59-65); bloom = the filter's wire bytes (BloomFilter::to_bytes)"""

    def __init__(self, last_sync=None, bloom=b""):
        self.last_sync, self.bloom = list(last_sync or []), bytes(bloom)

    def __eq__(self, o):
        return self.last_sync == o.last_sync and self.bloom == o.bloom


class SyncState:
    """sync::State (sync/state.rs:31-53): what one side knows about one peer for one document"""

    def __init__(self):
        self.shared_heads, self.last_sent_heads = [], []
        self.their_heads = self.their_need = self.their_have = None
        self.sent_hashes = set()
        self.in_flight = False

    def encode(self):   # state.rs:72-76: only the shared heads persist
        return bytes([0x43]) + _uleb(len(self.shared_heads)) + b"".join(self.shared_heads)

    @staticmethod
    def decode(data):   # state.rs:87-109
        if not data or data[0] != 0x43:
            raise ValueError("wrong type for an encoded sync state")
        hs, pos = _read_hashes(data, 1)
        st = SyncState()
        st.shared_heads, st.their_have = hs, []
        return st


def _uleb(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_uleb(data, pos):
    v = shift = 0
    while True:
        if pos >= len(data):
            raise ValueError("not enough input")
        b = data[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _read_hashes(data, pos):
    n, pos = _read_uleb(data, pos)
    if pos + 32 * n > len(data):
        raise ValueError("not enough input")
    return [data[pos + 32 * i: pos + 32 * i + 32] for i in range(n)], pos + 32 * n


class SyncMessage:
    """sync::Message (sync.rs:433-442) in its wire form (Message::parse :482-524, encode :526-543); changes stay raw chunks"""

    def __init__(self, heads=(), need=(), have=(), changes=()):
        self.heads, self.need, self.have, self.changes = list(heads), list(need), list(have), list(changes)

    @staticmethod
    def decode(data):
        if not data or data[0] != 0x42:
            raise ValueError("wrong type for a sync message")
        heads, pos = _read_hashes(data, 1)
        need, pos = _read_hashes(data, pos)
        nh, pos = _read_uleb(data, pos)
        have = []
        for _ in range(nh):
            ls, pos = _read_hashes(data, pos)
            bl, pos = _read_uleb(data, pos)
            have.append(SyncHave(ls, data[pos:pos + bl]))
            pos += bl
        nc, pos = _read_uleb(data, pos)
        changes = []
        for _ in range(nc):
            cl, pos = _read_uleb(data, pos)
            if pos + cl > len(data):
                raise ValueError("not enough input")
            changes.append(data[pos:pos + cl])
            pos += cl
        return SyncMessage(heads, need, have, changes)

    def encode(self):
        out = bytearray([0x42])
        for hs in (self.heads, self.need):
            out += _uleb(len(hs)) + b"".join(hs)
        out += _uleb(len(self.have))
        for h in self.have:
            out += _uleb(len(h.last_sync)) + b"".join(h.last_sync) + _uleb(len(h.bloom)) + h.bloom
        out += _uleb(len(self.changes))
        for c in self.changes:
            out += _uleb(len(c)) + c
        return bytes(out)


class Arena:
    """Concatenated documents + offsets[n + 1], owned by the library (generator output)."""

    def __init__(self, engine, o, offs, n):
        self.engine, self.o, self.offs, self.n = engine, o, offs, n

    @property
    def nbytes(self):
        return self.o.len

    def doc(self, i):
        return C.string_at(C.cast(self.o.data, C.c_void_p).value + self.offs[i], self.offs[i + 1] - self.offs[i])

    def close(self):
        if self.o.data:
            self.engine.lib.amh_bytes_free(C.byref(self.o))
            self.engine.lib.amh_u64_free(self.offs)
            self.offs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Staged:
    def __init__(self, engine, handle, n):
        self.engine, self.h, self.n = engine, handle, n
        engine._children.add(self)

    def close(self):
        if self.h:
            self.engine.lib.amh_staged_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_first():
    """The library opens librccl.so.1 by name on first use.  Where a PyTorch ROCm wheel is installed its RCCL (and what
    that pulls in) must be loaded by `import torch` itself, not piecemeal before it: RCCL loaded on its own and torch
    imported afterwards ends in a double free at process exit.  Multi-GPU callers hand the communicator id around with
    torch.distributed anyway, so importing torch here costs them nothing."""
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


_live_engines = weakref.WeakSet()


def _close_engines_at_exit():
    """Contexts are destroyed while the HIP runtime (and RCCL, if it was opened) are still up: the interpreter runs this
    before any library's own exit handlers, whereas a context freed by garbage collection during shutdown may come after
    them."""
    for e in list(_live_engines):
        try:
            e.close()
        except Exception:
            pass


import atexit  # noqa: E402
atexit.register(_close_engines_at_exit)


class Engine:
    """One context = one GPU (amh_ctx)."""

    def __init__(self, gpu=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        self._children = weakref.WeakSet()   # batches and staged inputs: they hold handles into the context
        _live_engines.add(self)
        ids = (C.c_int * 1)(gpu)
        self.ctx = self.lib.amh_ctx_create(ids, 1)
        if not self.ctx:
            raise EngineError("amh_ctx_create failed: " + (self.lib.amh_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "ctx", None):
            for child in list(self._children):   # their memory goes back to the context's pool: free them first
                child.close()
            self.lib.amh_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return (self.lib.amh_last_error(self.ctx) or b"").decode()

    @staticmethod
    def _pack(docs):
        n = len(docs)
        ptrs = (C.c_char_p * max(1, n))(*docs) if n else (C.c_char_p * 1)()
        lens = (C.c_size_t * max(1, n))(*[len(d) for d in docs]) if n else (C.c_size_t * 1)()
        return ptrs, lens, n

    def stage(self, docs):
        ptrs, lens, n = self._pack(docs)
        h = self.lib.amh_stage_batch(self.ctx, ptrs, lens, n)
        if not h:
            raise EngineError("amh_stage_batch failed: " + self._err())
        return Staged(self, h, n)

    def stage_arena(self, arena):
        """Stage the documents of an Arena (concatenated bytes + offsets) without copying them into Python objects."""
        import numpy as np
        n = arena.n
        offs = np.ctypeslib.as_array(arena.offs, shape=(n + 1,)).astype(np.uint64)
        base = C.cast(arena.o.data, C.c_void_p).value or 0
        ptrs = (offs[:n] + np.uint64(base)).astype(np.uint64)
        lens = (offs[1:] - offs[:n]).astype(np.uint64)
        h = self.lib.amh_stage_batch(self.ctx, C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p)),
                                     C.cast(lens.ctypes.data, C.POINTER(C.c_size_t)), n)
        if not h:
            raise EngineError("amh_stage_batch failed: " + self._err())
        return Staged(self, h, n)

    def load_staged(self, staged, verify_heads=True):
        opts = LoadOpts(1 if verify_heads else 0, 0)
        h = self.lib.amh_load_staged(self.ctx, staged.h, C.byref(opts))
        if not h:
            raise EngineError("amh_load_staged failed: " + self._err())
        return Batch(self, h, staged.n)

    def load_batch(self, docs, verify_heads=True, on_partial_load="error"):
        """== Automerge::load_with_options per document (automerge.rs:601).  An input may be a document chunk followed
        by change chunks, or change chunks alone; on_partial_load = "error" | "ignore" (OnPartialLoad)."""
        ptrs, lens, n = self._pack(docs)
        opts = LoadOpts(1 if verify_heads else 0, 1 if on_partial_load == "ignore" else 0)
        h = self.lib.amh_load_batch(self.ctx, ptrs, lens, n, C.byref(opts))
        if not h:
            raise EngineError("amh_load_batch failed: " + self._err())
        return Batch(self, h, n)

    def load(self, data, verify_heads=True):
        """Single document: the n = 1 case; raises AutomergeError like Automerge::load."""
        b = self.load_batch([data], verify_heads)
        b.check(0)
        return b

    # --- measurement -------------------------------------------------------------------------
    def set_profiling(self, on):
        self.lib.amh_ctx_set_profiling(self.ctx, 1 if on else 0)

    # --- multi-GPU ---------------------------------------------------------------------------
    def comm_unique_id(self):
        """rank 0: the 128 bytes every rank passes to comm_init (hand them over with torch.distributed, MPI, a file)"""
        _torch_first()
        buf = C.create_string_buffer(128)
        if self.lib.amh_comm_unique_id(buf) != 0:
            raise EngineError("amh_comm_unique_id failed: " + (self.lib.amh_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        _torch_first()
        if self.lib.amh_comm_init(self.ctx, unique_id, rank, world) != 0:
            raise EngineError("amh_comm_init failed: " + self._err())
        self.world = world

    def partition_docs(self, lens, world):
        """per-rank document ranges of about equal encoded bytes: first[k] .. first[k + 1]"""
        n = len(lens)
        arr = (C.c_size_t * max(1, n))(*lens)
        first = (C.c_size_t * (world + 1))()
        if self.lib.amh_partition_docs(arr, n, world, first) != 0:
            raise ValueError("amh_partition_docs: invalid argument")
        return list(first)

    def set_limits(self, max_doc_rows=0, max_batch_bytes=0):
        """per-document row limit and per-batch byte budget of the derived device arrays (0 = defaults): documents over
        them fail with AMH_DOC_TOO_LARGE, the rest of their batch loads"""
        if self.lib.amh_ctx_set_limits(self.ctx, max_doc_rows, max_batch_bytes) != 0:
            raise ValueError("amh_ctx_set_limits: invalid argument")

    def pool_bytes(self):
        """(live, pooled) device bytes of the context's caching allocator"""
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.amh_ctx_pool_bytes(self.ctx, C.byref(a), C.byref(b))
        return a.value, b.value

    def reset_profile(self):
        self.lib.amh_ctx_reset_profile(self.ctx)

    def kernel_times(self):
        out = {}
        for k in range(self.lib.amh_ctx_num_kernels()):
            ms = C.c_double()
            n = C.c_uint64()
            self.lib.amh_ctx_kernel_time(self.ctx, k, C.byref(ms), C.byref(n))
            out[self.lib.amh_ctx_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    # --- synthetic workloads -----------------------------------------------------------------
    def gen_map_docs(self, seed, n_docs, n_actors=4, changes_per_actor=5, ops_per_change=50, n_keys=128, threads=8):
        o = BytesOut()
        offs = C.POINTER(C.c_uint64)()
        rc = self.lib.amh_gen_map_docs(seed, n_docs, n_actors, changes_per_actor, ops_per_change, n_keys, threads,
                                       C.byref(o), C.byref(offs))
        if rc != 0:
            raise EngineError(f"amh_gen_map_docs failed: {rc}")
        return self._take(o, offs, n_docs)

    def gen_map_arena(self, seed, n_docs, n_actors=4, changes_per_actor=5, ops_per_change=50, n_keys=128, threads=8):
        """Like gen_map_docs, but the bytes stay in the C buffer (large batches)."""
        o = BytesOut()
        offs = C.POINTER(C.c_uint64)()
        rc = self.lib.amh_gen_map_docs(seed, n_docs, n_actors, changes_per_actor, ops_per_change, n_keys, threads,
                                       C.byref(o), C.byref(offs))
        if rc != 0:
            raise EngineError(f"amh_gen_map_docs failed: {rc}")
        return Arena(self, o, offs, n_docs)

    def _take(self, o, offs, n):
        blob = C.string_at(o.data, o.len)
        offsets = [offs[i] for i in range(n + 1)]
        self.lib.amh_bytes_free(C.byref(o))
        self.lib.amh_u64_free(offs)
        return blob, offsets

    def gen_text_docs(self, seed, n_docs, n_replicas=8, base_len=1000, ops_per_replica=5000, ops_per_change=500, threads=8):
        o = BytesOut()
        offs = C.POINTER(C.c_uint64)()
        rc = self.lib.amh_gen_text_docs(seed, n_docs, n_replicas, base_len, ops_per_replica, ops_per_change, threads,
                                        C.byref(o), C.byref(offs))
        if rc != 0:
            raise EngineError(f"amh_gen_text_docs failed: {rc}")
        return self._take(o, offs, n_docs * n_replicas)
