"""amh_text_batch (== ReadDoc::text per document) against the oracle's materialization, shared by the emulation and GPU tiers."""
import json

import oracle_lib as o


def oracle_texts(doc_bytes):
    """the texts of the document's text objects in document order (hydrate lists an object's value as {"id", "text"})"""
    out = []

    def walk(v):
        if isinstance(v, dict):
            if "text" in v and "id" in v and len(v) == 2:
                out.append((v["id"], v["text"]))
                return
            for k in sorted(v):
                walk(v[k])
        elif isinstance(v, list):
            for x in v:
                walk(x)

    walk(json.loads(o.Doc.load(doc_bytes).hydrate_json()))

    def key(t):   # document order of objects = ascending (counter, actor bytes)
        c, a = t[0].split("@")
        return (int(c), bytes.fromhex(a))

    return "".join(t for _, t in sorted(out, key=key))


def check_texts(engine, docs, expect_nonempty=True):
    b = engine.load_batch(docs)
    texts = b.text_all()
    assert len(texts) == len(docs)
    for i, d in enumerate(docs):
        kind, _ = b.status(i)
        want = oracle_texts(d) if kind == 0 else ""
        assert texts[i] == want, (i, texts[i][:60], want[:60])
    assert not expect_nonempty or any(texts)
    assert b.text_device()[2] == sum(len(t.encode("utf-8")) for t in texts)   # the device-resident form: same bytes in HBM
    b.close()
    return texts
