"""Change chunks built from op lists, for test inputs the document generator does not produce (adversarial shapes).
Columns are encoded by the CPU oracle's codecs and the chunk is put into canonical form by the oracle's own
change re-encoder, so a chunk made here is exactly what the reference's encoder writes for these ops
(storage/change/change_op_columns.rs:28-50 column ids, storage/change.rs:330-420 chunk layout)."""
import struct

import oracle_lib as o

MAKE_MAP, SET, MAKE_LIST, DEL, MAKE_TEXT, INC, MAKE_TABLE, MARK = 0, 1, 2, 3, 4, 5, 6, 7


def uleb(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def sleb(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if (n == 0 and not b & 0x40) or (n == -1 and b & 0x40):
            out.append(b)
            return bytes(out)
        out.append(b | 0x80)


def v_uint(n):
    raw = uleb(n)
    return (len(raw) << 4) | 3, raw


def v_int(n):
    raw = sleb(n)
    return (len(raw) << 4) | 4, raw


def v_str(s):
    raw = s.encode("utf-8")
    return (len(raw) << 4) | 6, raw


def v_counter(n):
    raw = sleb(n)
    return (len(raw) << 4) | 8, raw


def v_null():
    return 0, b""


def v_bool(b):
    return (2 if b else 1), b""


def op(action, obj=None, key=None, elem=None, insert=False, value=None, pred=(), expand=False, mark_name=None):
    """obj: None (root) or (ctr, actor index); key: map key, or elem: (ctr, actor index) with (0, None) = HEAD;
    value: (meta, raw) from v_*; pred: [(ctr, actor index)].  Actor index 0 is the change's author.
    Marks (marks.rs, transaction/inner.rs:716-735): two insert ops of action MARK in a sequence — the begin carries
    mark_name, the value and expand = "expand before"; the end carries no name, no value and expand = "expand after"."""
    return dict(action=action, obj=obj, key=key, elem=elem, insert=insert, value=value, pred=list(pred), expand=expand, mark_name=mark_name)


def mark_ops(obj, start_elem, end_elem, name, value, expand_before=False, expand_after=False):
    """the op pair of Transaction::mark: begin inserted behind start_elem, end behind end_elem"""
    return [op(MARK, obj=obj, elem=start_elem, insert=True, value=value, expand=expand_before, mark_name=name),
            op(MARK, obj=obj, elem=end_elem, insert=True, expand=expand_after)]


def build_change(actor, seq, start_op, deps, ops, others=(), time=0, message=b""):
    cols = []

    def add(spec, data):
        if data:
            cols.append((spec, data))

    add(1, o.codec_encode(0, [None if p["obj"] is None else p["obj"][1] for p in ops]))
    add(2, o.codec_encode(0, [None if p["obj"] is None else p["obj"][0] for p in ops]))
    add(17, o.codec_encode(0, [None if p["elem"] is None else p["elem"][1] for p in ops]))
    add(19, o.codec_encode(1, [None if p["elem"] is None else p["elem"][0] for p in ops]))
    add(21, o.codec_encode(3, [p["key"] for p in ops]))
    add(52, o.codec_encode(2, [p["insert"] for p in ops]))
    add(66, o.codec_encode(0, [p["action"] for p in ops]))
    add(86, o.codec_encode(0, [0 if p["value"] is None else p["value"][0] for p in ops]))
    add(87, b"".join(b"" if p["value"] is None else p["value"][1] for p in ops))
    add(112, o.codec_encode(0, [len(p["pred"]) for p in ops]))
    add(113, o.codec_encode(0, [a for p in ops for (_, a) in p["pred"]]))
    add(115, o.codec_encode(1, [c for p in ops for (c, _) in p["pred"]]))
    add(148, o.codec_encode(4, [p.get("expand", False) for p in ops]))                      # 0x94 expand (MaybeBoolean)
    if any(p.get("mark_name") is not None for p in ops):
        add(165, o.codec_encode(3, [p.get("mark_name") for p in ops]))                       # 0xA5 mark name
    body = bytearray()
    body += uleb(len(deps))
    for d in sorted(deps):
        body += d
    body += uleb(len(actor)) + actor + uleb(seq) + uleb(start_op) + sleb(time) + uleb(len(message)) + message
    body += uleb(len(others))
    for a in others:
        body += uleb(len(a)) + a
    body += uleb(len(cols))
    for spec, data in cols:
        body += uleb(spec) + uleb(len(data))
    for _, data in cols:
        body += data
    tail = bytes([1]) + uleb(len(body)) + bytes(body)
    chunk = b"\x85\x6f\x4a\x83" + o.sha256(tail)[:4] + tail
    canon = o.change_rebuild(chunk)
    return canon, o.change_hash(canon)


def actor_id(i):
    return struct.pack(">QQ", 0xAD7E25A21A000000 + i, i * 0x9E3779B97F4A7C15 & (2**64 - 1))
