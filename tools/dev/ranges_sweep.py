"""Many-wave encode paths at several range sizes against the one-wave forms: saved bytes equal, loads verify (GPU)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from automerge_amd import Engine
e = Engine(gpu=0)
blob, offs = e.gen_text_docs(0x5EED, 6, 3, 300, 2500, 700, 4)
docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(18)]
blob, offs = e.gen_map_docs(0x5EED, 12, 4, 5, 60, 64, 4)
docs += [bytes(blob[offs[i]:offs[i + 1]]) for i in range(12)]
for k in ("AMH_BIGSAVE_ROWS", "AMH_BIGENC_ROWS"): os.environ[k] = "4000000000"
b = e.load_batch(docs)
assert all(b.status(i)[0] == 0 for i in range(len(docs)))
want = b.save_all(deflate=False)
heads = [b.get_heads(i) for i in range(len(docs))]
b.close()
for rng in (64, 128, 192, 1024, 4096):
    os.environ["AMH_BIGSAVE_ROWS"] = "1"; os.environ["AMH_BIGENC_ROWS"] = "1"
    os.environ["AMH_BIGSAVE_RANGE"] = str(rng); os.environ["AMH_BIGENC_RANGE"] = str(rng)
    for form in ("seg", "lane"):
        os.environ["AMH_ENCODE_FORM"] = form; os.environ["AMH_SAVE_FORM"] = form
        b = e.load_batch(docs)
        bad = [i for i in range(len(docs)) if b.status(i)[0] != 0 or b.get_heads(i) != heads[i]]
        got = b.save_all(deflate=False)
        diff = [i for i in range(len(docs)) if got[i] != want[i]]
        print("range %5d %-4s load failures %s, save differences %s" % (rng, form, bad, diff), flush=True)
        assert not bad and not diff
        b.close()
print("ok")
