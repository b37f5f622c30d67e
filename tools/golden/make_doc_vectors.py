"""Step 2 of the golden pipeline (runs in the build container only): build documents from the
JS-encoded changes with the CPU oracle, save them, and have the reference's legacy JS
`decodeDocument` (javascript/test/legacy/columnar.js:1264) decode them back.  decodeDocument
re-derives every change, re-encodes it and verifies the heads; its hashes are the pins."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
vec = json.load(open(os.path.join(G, "js_changes.json")))
out_dir = os.path.join(G, "docs")
for oi, order in enumerate(vec["orders"]):
    for s in order["scenarios"]:
        d = o.Doc()
        d.apply_changes([bytes.fromhex(c["hex"]) for c in s["changes"]])
        open(os.path.join(out_dir, f"{oi}_{s['name']}.automerge"), "wb").write(d.save_nocompress())
        open(os.path.join(out_dir, f"{oi}_{s['name']}.deflate.automerge"), "wb").write(d.save(True))
js_dir = sys.argv[1]  # scratch dir holding the .mjs copies of the legacy JS + driver.mjs
subprocess.check_call(["node", "--experimental-modules", "driver.mjs", "decode_docs", out_dir,
                       os.path.join(G, "js_docs_decoded.json")], cwd=js_dir)
