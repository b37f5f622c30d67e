"""BASELINE.json configs[0]: the automerge-perf editing trace as one change of 259,779 ops (tools/golden/
make_edit_trace.py; change hash pinned by the reference's legacy JS encodeChange / decodeDocument).  The text the
trace produces is pinned by its SHA-256, computed by a plain replay of the edits."""
import gzip
import hashlib
import json
import os


def load_fixture(golden_dir):
    data = gzip.open(os.path.join(golden_dir, "edit_trace.automerge.gz")).read()
    meta = json.load(open(os.path.join(golden_dir, "edit_trace.json")))
    assert hashlib.sha256(data).hexdigest() == meta["doc_sha256"]
    return data, meta


def check_hydrated(hydrated_json, meta):
    h = json.loads(hydrated_json)
    (entry,) = h["text"]
    text = entry["text"]
    assert len(text) == meta["text_len"]
    assert hashlib.sha256(text.encode("utf-8")).hexdigest() == meta["text_sha256"]


def check_engine(engine, golden_dir):
    import oracle_lib as o
    data, meta = load_fixture(golden_dir)
    b = engine.load_batch([data])
    b.check(0)
    (c,) = b.get_changes(0)
    assert c["hash"].hex() == meta["change_hash"] and c["len"] == meta["n_ops"] and c["start_op"] == 1
    assert b.get_heads(0) == [bytes.fromhex(meta["change_hash"])]
    assert b.max_op(0) == meta["n_ops"]
    hj = b.hydrate_json(0)
    check_hydrated(hj, meta)
    ref = o.Doc.load(data)
    assert hj == ref.hydrate_json()
    assert c["raw"] == ref.changes()[0]["raw"]
    assert b.save_all(deflate=False)[0] == data          # re-save identity (automerge/tests.rs:1311-1443 style)
    assert b.stats()["ops_loaded"] == meta["n_ops"]
    # the same change delivered as a change chunk: load of a change-only input and apply onto an empty document
    b2 = engine.load_batch([c["raw"], b""])
    b2.check(0)
    assert b2.save_all(deflate=False)[0] == data
    b2.apply_changes(1, [c["raw"]])
    assert b2.save_all(deflate=False)[1] == data
