"""Pin the oracle's change codec and merge semantics against the reference:
 * change chunks encoded by the reference's legacy JS `encodeChange` (tests/golden/js_changes.json):
   our parser + builder must reproduce the bytes and the hash;
 * the byte-level KAT of rust/automerge/tests/test.rs:1266-1290;
 * merge / RGA expectations restated from rust/automerge/tests/test.rs (line refs in the fixture)."""
import itertools
import json
import os

import pytest

import oracle_lib as o
from am_expect import simplify

KAT71 = bytes([
    0x85, 0x6f, 0x4a, 0x83, 0xb2, 0x98, 0x9e, 0xa9, 1, 61, 0, 2, 0x12, 0x34, 1, 1, 252, 250, 220, 255, 5,
    14, 73, 110, 105, 116, 105, 97, 108, 105, 122, 97, 116, 105, 111, 110, 0, 6, 0x15, 3, 0x34, 1, 0x42, 2,
    0x56, 2, 0x57, 1, 0x70, 2, 0x7f, 1, 0x78, 1, 0x7f, 1, 0x7f, 19, 1, 0x7f, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,
])


@pytest.fixture(scope="module")
def vectors(golden_dir):
    return json.load(open(os.path.join(golden_dir, "js_changes.json")))


def test_kat71_round_trip():
    # tests/test.rs:1266-1290: parse, rebuild from decoded ops => identical bytes; checksum b2989ea9
    assert len(KAT71) == 71
    assert o.change_rebuild(KAT71) == KAT71
    assert o.change_hash(KAT71)[:4].hex() == "b2989ea9"


def test_zero_start_op_is_rejected():
    # automerge/tests.rs:1297-1308 load_change_with_zero_start_op: must not panic; start_op 0 is an error
    data = bytes([133, 111, 74, 131, 0, 46, 128, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    with pytest.raises(o.OracleError):
        o.Doc.load(data)


def _uleb(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _inflate_change(raw):
    import zlib
    i = 9
    while raw[i] & 0x80:
        i += 1
    body = zlib.decompress(raw[i + 1:], -15)
    return raw[:8] + b"\x01" + _uleb(len(body)) + body


def _all_changes(vectors):
    for order in vectors["orders"]:
        for s in order["scenarios"]:
            for c in s["changes"]:
                yield c
    for c in vectors["random"]:
        yield c


def test_js_encoded_changes_rebuild_bit_exact(vectors):
    n = 0
    for c in _all_changes(vectors):
        raw = bytes.fromhex(c["hex"])
        assert o.change_hash(raw).hex() == c["hash"]
        if raw[8] == 2:  # the legacy JS deflates changes >= 256 bytes (columnar.js:899): compare the inflated chunk
            raw = _inflate_change(raw)
        assert o.change_rebuild(raw) == raw
        n += 1
    assert n > 80


def test_merge_scenarios_match_reference_expectations(vectors):
    for order in vectors["orders"]:
        for s in order["scenarios"]:
            chunks = [bytes.fromhex(c["hex"]) for c in s["changes"]]
            results = set()
            saves = set()
            for perm in itertools.permutations(range(len(chunks))):
                d = o.Doc()
                d.apply_changes([chunks[i] for i in perm])
                assert d.queue_len() == 0
                results.add(d.hydrate_json())
                # load(save(x)) reproduces state, hashes and bytes (automerge/tests.rs:1311-1443)
                saved = d.save_nocompress()
                d2 = o.Doc.load(saved)
                assert d2.hydrate_json() == d.hydrate_json()
                assert d2.save_nocompress() == saved
                assert [c["hash"] for c in d2.changes()] == [c["hash"] for c in d.changes()]
                assert [c["raw"] for c in d2.changes()] == [c["raw"] for c in d.changes()]
                d3 = o.Doc.load(d.save(deflate=True))
                assert d3.save_nocompress() == saved
                saves.add(saved)
            assert len(results) == 1, s["name"]
            if s["expect"] is not None:
                assert simplify(json.loads(results.pop())) == s["expect"], s["name"]


def test_merge_of_documents_equals_apply(vectors):
    for order in vectors["orders"]:
        for s in order["scenarios"]:
            chunks = [bytes.fromhex(c["hex"]) for c in s["changes"]]
            if len(chunks) < 3:
                continue
            a = o.Doc(); a.apply_changes(chunks[:2])
            b = o.Doc(); b.apply_changes([chunks[0]] + chunks[2:])
            if b.queue_len():  # later changes depend on chunks[1]: not a causally closed replica
                continue
            full = o.Doc(); full.apply_changes(chunks)
            a2 = o.Doc.load(a.save()); b2 = o.Doc.load(b.save())
            a2.merge(b2)
            assert a2.hydrate_json() == full.hydrate_json()
            assert sorted(h.hex() for h in a2.get_heads()) == sorted(h.hex() for h in full.get_heads())


def test_duplicate_seq_and_missing_deps(vectors):
    s = vectors["orders"][0]["scenarios"][1]  # counter scenario: c1, c2 (lo seq 2), c3 (hi seq 1)
    chunks = [bytes.fromhex(c["hex"]) for c in s["changes"]]
    # a change whose deps are absent is queued by apply_changes, but load() of bare change chunks errors
    d = o.Doc(); d.apply_changes([chunks[1]])
    assert d.queue_len() == 1 and d.num_ops() == 0
    with pytest.raises(o.OracleError) as ei:
        o.Doc.load(chunks[1])
    assert ei.value.kind == 5  # MissingDeps (automerge.rs:683-688)
    # orphans survive save/load (tests/test_save_load_orphans.rs)
    d.apply_changes([])
    d2 = o.Doc.load(d.save())
    assert d2.queue_len() == 1
    d2.apply_changes([chunks[0]])
    assert d2.queue_len() == 0 and d2.num_ops() == 2


def test_load_incremental_with_corrupted_tail(vectors):
    # tests/test.rs:1313-1329
    s = vectors["orders"][0]["scenarios"][0]
    chunks = [bytes.fromhex(c["hex"]) for c in s["changes"]]
    d = o.Doc(); d.apply_changes(chunks)
    data = d.save() + bytes([1, 2, 3, 4])
    fresh = o.Doc()
    fresh.load_incremental(data)
    assert fresh.hydrate_json() == d.hydrate_json()
    with pytest.raises(o.OracleError):
        o.Doc.load(data)
