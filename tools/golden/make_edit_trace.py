"""BASELINE.json configs[0] as a fixture (runs in the build container only): replay the editing trace the
reference's rust/edit-trace driver replays (main.rs:20-54: one transaction, put_object(ROOT, "text", Text) and one
splice_text per edit), as ONE change of 259,779 ops: makeText, then per edit either an insert of one character after
the element in front of the position or a delete of the element at the position (pred = that element).  The change is
encoded by the reference's legacy JS encodeChange (hash pinned by it), turned into a document by the CPU oracle, and the
document is decoded back by the JS decodeDocument (re-derives and re-hashes the change, verifies the heads).
Outputs: tests/golden/edit_trace.automerge.gz (save_nocompress bytes) + edit_trace.json (hash, op counts, sha256 of the
final text, its length).  The trace itself (reference data) is read from /root/reference and is not copied."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o  # noqa: E402

js_dir = sys.argv[1]
edits = json.load(open("/root/reference/rust/edit-trace/edits.json"))
ACTOR = "0e" + "d1" * 15
text_obj = f"1@{ACTOR}"
ops = [dict(action="makeText", obj="_root", key="text", insert=False, pred=[])]
elems, chars = [], []          # visible elements in order: op counter, character
ctr = 1
for e in edits:
    pos, ndel = e[0], e[1]
    assert ndel in (0, 1)
    if ndel:
        ctr += 1
        target = f"{elems[pos]}@{ACTOR}"
        ops.append(dict(action="del", obj=text_obj, elemId=target, insert=False, pred=[target]))
        del elems[pos]
        del chars[pos]
    for s in e[2:]:
        assert len(s) == 1 or len(s.encode("utf-16-le")) == 4
        ctr += 1
        ref = "_head" if pos == 0 else f"{elems[pos - 1]}@{ACTOR}"
        ops.append(dict(action="set", obj=text_obj, elemId=ref, insert=True, value=s, pred=[]))
        elems.insert(pos, ctr)
        chars.insert(pos, s)
        pos += 1
assert ctr == len(ops)
change = dict(actor=ACTOR, seq=1, startOp=1, time=0, message="", deps=[], ops=ops)
json.dump(change, open(os.path.join(js_dir, "trace_change.json"), "w"))
subprocess.check_call(["node", "--experimental-modules", "--max-old-space-size=6000", "driver.mjs", "encode_change", "trace_change.json", "trace_change_out.json"], cwd=js_dir)
enc = json.load(open(os.path.join(js_dir, "trace_change_out.json")))
chunk = bytes.fromhex(enc["hex"])
assert o.change_hash(chunk).hex() == enc["hash"]
d = o.Doc()
d.apply_changes([chunk])
saved = d.save_nocompress()
open(os.path.join(js_dir, "trace_doc.automerge"), "wb").write(saved)
subprocess.check_call(["node", "--experimental-modules", "--max-old-space-size=6000", "driver.mjs", "decode_doc", "trace_doc.automerge", "trace_doc_out.json"], cwd=js_dir)
dec = json.load(open(os.path.join(js_dir, "trace_doc_out.json")))
assert len(dec) == 1 and dec[0]["hash"] == enc["hash"] and dec[0]["nops"] == len(ops), dec
text = "".join(chars)
G = os.path.join(ROOT, "tests", "golden")
with gzip.GzipFile(os.path.join(G, "edit_trace.automerge.gz"), "wb", mtime=0) as f:
    f.write(saved)
json.dump(dict(change_hash=enc["hash"], n_ops=len(ops), n_rows=1 + len([1 for x in ops if x["action"] == "set"]), n_edits=len(edits),
               text_len=len(text), text_sha256=hashlib.sha256(text.encode("utf-8")).hexdigest(), doc_bytes=len(saved),
               doc_sha256=hashlib.sha256(saved).hexdigest(), actor=ACTOR),
          open(os.path.join(G, "edit_trace.json"), "w"), indent=1)
print(len(ops), len(saved), len(text), enc["hash"])
