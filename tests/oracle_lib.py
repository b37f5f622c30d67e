"""ctypes binding to the CPU oracle (oracle/libamo_oracle.so) — test infrastructure only."""
import ctypes as C
import os
import struct
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libamo_oracle.so")


def _ensure_built():
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "libamo_oracle.so"])


_ensure_built()
lib = C.CDLL(_SO)
_u8p = C.POINTER(C.c_uint8)
lib.amo_last_error.restype = C.c_char_p
lib.amo_new.restype = C.c_void_p
lib.amo_clone.restype = C.c_void_p
lib.amo_clone.argtypes = [C.c_void_p]
lib.amo_free.argtypes = [C.c_void_p]
lib.amo_load.restype = C.c_void_p
lib.amo_load.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
lib.amo_load_opts.restype = C.c_void_p
lib.amo_load_opts.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
lib.amo_change_compress.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_save_after.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_load_incremental.restype = C.c_long
lib.amo_load_incremental.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.amo_apply_changes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.amo_merge.argtypes = [C.c_void_p, C.c_void_p]
for name in ("amo_heads", "amo_changes", "amo_hydrate", "amo_bloom"):
    getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_save.argtypes = [C.c_void_p, C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_bloom_from_hashes.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_generate_sync_message.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t), C.POINTER(_u8p),
                                          C.POINTER(C.c_size_t)]
lib.amo_receive_sync_message.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_sync_message_reencode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_bloom_contains.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
lib.amo_num_ops.restype = C.c_ulong
lib.amo_num_ops.argtypes = [C.c_void_p]
lib.amo_max_op.restype = C.c_ulong
lib.amo_max_op.argtypes = [C.c_void_p]
lib.amo_queue_len.restype = C.c_ulong
lib.amo_queue_len.argtypes = [C.c_void_p]
lib.amo_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
lib.amo_change_rebuild.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_change_hash.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
lib.amo_codec_encode.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_codec_decode.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
lib.amo_strict_leb.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
lib.amo_bench_load_batch.restype = C.c_double
lib.amo_bench_load_batch.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_int, C.c_int,
                                     C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]

KINDS = {0: "ok", 1: "parse", 2: "bad_checksum", 3: "invalid_change_columns", 4: "inflate_document",
         5: "missing_deps", 6: "duplicate_seq", 7: "leftover", 8: "wrong_chunk_type", 9: "missing_dep_graph",
         10: "invalid_arg"}


class OracleError(Exception):
    def __init__(self):
        self.kind = lib.amo_last_kind()
        self.sub = lib.amo_last_sub()
        self.msg = lib.amo_last_error().decode("utf-8", "replace")
        super().__init__(f"{KINDS.get(self.kind, self.kind)}/{self.sub}: {self.msg}")


def _out(fn, *args):
    p = _u8p()
    n = C.c_size_t()
    if fn(*args, C.byref(p), C.byref(n)) != 0:
        raise OracleError()
    return C.string_at(p, n.value)


class Doc:
    """Mirror of the reference's `Automerge` for the bulk path (automerge.rs:552-1334)."""

    def __init__(self, handle=None):
        self.h = handle if handle is not None else lib.amo_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib.amo_free(self.h)
            self.h = None

    @staticmethod
    def load(data: bytes, check_heads=True, on_partial_load="error") -> "Doc":
        h = lib.amo_load_opts(data, len(data), 1 if check_heads else 0, 1 if on_partial_load == "ignore" else 0)
        if not h:
            raise OracleError()
        return Doc(h)

    def clone(self):
        return Doc(lib.amo_clone(self.h))

    def load_incremental(self, data: bytes) -> int:
        r = lib.amo_load_incremental(self.h, data, len(data))
        if r < 0:
            raise OracleError()
        return r

    def apply_changes(self, chunks) -> None:
        data = b"".join(chunks)
        if lib.amo_apply_changes(self.h, data, len(data)) != 0:
            raise OracleError()

    def merge(self, other: "Doc") -> None:
        if lib.amo_merge(self.h, other.h) != 0:
            raise OracleError()

    def save(self, deflate=True) -> bytes:
        return _out(lib.amo_save, self.h, 1 if deflate else 0)

    def save_after(self, heads) -> bytes:
        return _out(lib.amo_save_after, self.h, b"".join(heads), len(heads))

    def save_nocompress(self) -> bytes:
        return self.save(False)

    def get_heads(self):
        b = _out(lib.amo_heads, self.h)
        return [b[i:i + 32] for i in range(0, len(b), 32)]

    def changes(self):
        b = _out(lib.amo_changes, self.h)
        (n,) = struct.unpack_from("<I", b, 0)
        off = 4
        out = []
        for _ in range(n):
            h = b[off:off + 32]; off += 32
            seq, start_op, nops = struct.unpack_from("<QQQ", b, off); off += 24
            (nd,) = struct.unpack_from("<I", b, off); off += 4
            deps = [b[off + 32 * i: off + 32 * i + 32] for i in range(nd)]; off += 32 * nd
            (al,) = struct.unpack_from("<I", b, off); off += 4
            actor = b[off:off + al]; off += al
            (rl,) = struct.unpack_from("<I", b, off); off += 4
            raw = b[off:off + rl]; off += rl
            out.append(dict(hash=h, seq=seq, start_op=start_op, len=nops, deps=deps, actor=actor, raw=raw))
        return out

    def hydrate_json(self) -> str:
        return _out(lib.amo_hydrate, self.h).decode("utf-8")

    def bloom(self) -> bytes:
        return _out(lib.amo_bloom, self.h)

    def generate_sync_message(self, state_blob: bytes):
        """generate_sync_message (sync.rs:134-221) over a flat State (oracle/amo_sync.hpp): (message or None, new State)"""
        m, ml, st, sl = _u8p(), C.c_size_t(), _u8p(), C.c_size_t()
        rc = lib.amo_generate_sync_message(self.h, state_blob, len(state_blob), C.byref(m), C.byref(ml), C.byref(st), C.byref(sl))
        if rc < 0:
            raise OracleError()
        return (C.string_at(m, ml.value) if rc == 1 else None), C.string_at(st, sl.value)

    def receive_sync_message(self, state_blob: bytes, msg: bytes) -> bytes:
        """receive_sync_message (sync.rs:323-383): applies the message, returns the new flat State"""
        return _out(lib.amo_receive_sync_message, self.h, state_blob, len(state_blob), msg, len(msg))

    def num_ops(self) -> int:
        return lib.amo_num_ops(self.h)

    def max_op(self) -> int:
        return lib.amo_max_op(self.h)

    def queue_len(self) -> int:
        return lib.amo_queue_len(self.h)


def sync_message_reencode(msg: bytes) -> bytes:
    return _out(lib.amo_sync_message_reencode, msg, len(msg))


def bloom_contains(bloom: bytes, h: bytes) -> bool:
    r = lib.amo_bloom_contains(bloom, len(bloom), h)
    if r < 0:
        raise OracleError()
    return r == 1


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib.amo_sha256(data, len(data), out)
    return out.raw


def change_rebuild(chunk: bytes) -> bytes:
    return _out(lib.amo_change_rebuild, chunk, len(chunk))


def change_compress(chunk: bytes) -> bytes:
    return _out(lib.amo_change_compress, chunk, len(chunk))


def change_hash(chunk: bytes) -> bytes:
    out = C.create_string_buffer(32)
    if lib.amo_change_hash(chunk, len(chunk), out) != 0:
        raise OracleError()
    return out.raw


def bloom_from_hashes(hashes) -> bytes:
    data = b"".join(hashes)
    return _out(lib.amo_bloom_from_hashes, data, len(hashes))


def _pack_vals(kind, vals):
    if kind in (0, 1):
        fmt = "<BQ" if kind == 0 else "<Bq"
        return b"".join(struct.pack(fmt, 0 if v is None else 1, 0 if v is None else v) for v in vals)
    if kind in (2, 4):
        return bytes(1 if v else 0 for v in vals)
    out = []
    for v in vals:
        s = b"" if v is None else v.encode("utf-8")
        out.append(struct.pack("<BI", 0 if v is None else 1, len(s)) + s)
    return b"".join(out)


def codec_encode(kind: int, vals) -> bytes:
    blob = _pack_vals(kind, vals)
    return _out(lib.amo_codec_encode, kind, blob, len(vals))


def codec_decode(kind: int, data: bytes, count=0):
    b = _out(lib.amo_codec_decode, kind, data, len(data), count)
    out = []
    off = 0
    if kind in (0, 1):
        fmt = "<BQ" if kind == 0 else "<Bq"
        while off < len(b):
            pr, v = struct.unpack_from(fmt, b, off); off += 9
            out.append(v if pr else None)
    elif kind == 2:
        out = [bool(x) for x in b]
    else:
        while off < len(b):
            pr, l = struct.unpack_from("<BI", b, off); off += 5
            out.append(b[off:off + l].decode("utf-8") if pr else None); off += l
    return out


def strict_leb(signed: bool, data: bytes):
    v = C.c_uint64()
    n = C.c_size_t()
    rc = lib.amo_strict_leb(1 if signed else 0, data, len(data), C.byref(v), C.byref(n))
    val = v.value
    if signed and val >= 1 << 63:
        val -= 1 << 64
    return rc, val, n.value
