"""The reference's own binary fixtures (rust/automerge/tests/fixtures, tests/fuzz-crashers) with the
expectations of rust/automerge/tests/test.rs:1405-1464."""
import glob
import json
import os

import pytest

import oracle_lib as o
from am_expect import simplify


def _read(golden_dir, *p):
    return open(os.path.join(golden_dir, *p), "rb").read()


def test_fuzz_crashers_all_fail(golden_dir):  # tests/test.rs:1405-1415
    files = sorted(glob.glob(os.path.join(golden_dir, "ref_fuzz_crashers", "*")))
    assert len(files) == 8
    for f in files:
        with pytest.raises(o.OracleError):
            o.Doc.load(open(f, "rb").read())


def test_overlong_leb(golden_dir):  # tests/test.rs:1422-1429
    with pytest.raises(o.OracleError):
        o.Doc.load(_read(golden_dir, "ref_fixtures", "counter_value_has_incorrect_meta.automerge"))
    with pytest.raises(o.OracleError):
        o.Doc.load(_read(golden_dir, "ref_fixtures", "counter_value_is_overlong.automerge"))
    o.Doc.load(_read(golden_dir, "ref_fixtures", "counter_value_is_ok.automerge"))


@pytest.mark.parametrize("name", ["two_change_chunks", "two_change_chunks_compressed", "two_change_chunks_out_of_order"])
def test_load_fixtures(golden_dir, name):  # tests/test.rs:1432-1442: {a: {a: "b"}}
    d = o.Doc.load(_read(golden_dir, "ref_fixtures", name + ".automerge"))
    assert simplify(json.loads(d.hydrate_json())) == {"a": [{"map": {"a": ["b"]}}]}
    hashes = [c["hash"].hex() for c in d.changes()]
    # the two hashes the legacy JS decodeChanges yields for these files
    assert sorted(h[:12] for h in hashes) == sorted(["9ac3d730f262", "a7e727d50d6f"])
    assert [h.hex()[:12] for h in d.get_heads()] == ["a7e727d50d6f"]


def test_obj_id_64bits(golden_dir):  # tests/test.rs:1451-1464: either fails to load or is handled
    for name in ["64bit_obj_id_change", "64bit_obj_id_doc"]:
        try:
            d = o.Doc.load(_read(golden_dir, "ref_fixtures", name + ".automerge"))
        except o.OracleError:
            continue
        root = json.loads(d.hydrate_json())
        assert "map" in root["a"][0]


def test_empty_input_is_empty_doc():  # automerge.rs:605-608, tests/test.rs:790
    d = o.Doc.load(b"")
    assert d.hydrate_json() == "{}" and d.get_heads() == []
    d2 = o.Doc.load(d.save())
    assert d2.hydrate_json() == "{}"


def test_saved_documents_decode_in_reference_js(golden_dir):
    """tests/golden/docs/*.automerge were saved by the oracle; js_docs_decoded.json holds what the
    reference's legacy JS decodeDocument (columnar.js:1264, which re-derives the changes and verifies
    the heads) made of them.  Loading them here must re-derive the same change hashes and bytes."""
    decoded = json.load(open(os.path.join(golden_dir, "js_docs_decoded.json")))
    assert len(decoded) == 60
    for name, js_changes in decoded.items():
        d = o.Doc.load(_read(golden_dir, "docs", name))
        ours = d.changes()
        assert [c["hash"].hex() for c in ours] == [c["hash"] for c in js_changes], name
        for mine, js in zip(ours, js_changes):
            js_raw = bytes.fromhex(js["hex"])
            if js_raw[8] == 1:
                assert mine["raw"] == js_raw
            assert mine["len"] == js["nops"] and mine["start_op"] == js["startOp"] and mine["seq"] == js["seq"]
        plain = name.replace(".deflate", "")
        assert d.save_nocompress() == _read(golden_dir, "docs", plain)


def test_edit_trace_fixture(golden_dir):
    """configs[0]: hash pinned by the legacy JS encoder, text pinned by a plain replay of the trace"""
    import edit_trace_case as et
    data, meta = et.load_fixture(golden_dir)
    d = o.Doc.load(data)
    (c,) = d.changes()
    assert c["hash"].hex() == meta["change_hash"] and c["len"] == meta["n_ops"]
    et.check_hydrated(d.hydrate_json(), meta)
    assert d.save(deflate=False) == data


def test_reference_written_documents_resave_identically(golden_dir):
    """Three documents the reference itself saved (Automerge.save output quoted in rust/automerge-c/test/map_tests.c:156-162,
    :196-202 and list_tests.c:440-446): the expected value comes out, and save_nocompress(load(x)) == x pins the document
    ENCODER (DocOpColumns / DocChangeColumns, storage/document.rs:214-263) against bytes the reference wrote."""
    import json
    import os
    import oracle_lib as o
    fx = json.load(open(os.path.join(golden_dir, "ref_saved_docs.json")))["docs"]
    assert len(fx) == 3
    for f in fx:
        raw = bytes(f["bytes"])
        d = o.Doc.load(raw)
        h = json.loads(d.hydrate_json())
        assert list(h.keys()) == [f["key"]] and h[f["key"]][0]["str"] == f["value"], f["name"]
        assert d.save(deflate=False) == raw, f["name"]
        assert d.save(deflate=True) == raw, f["name"]   # every column is under 256 bytes: nothing is deflated
