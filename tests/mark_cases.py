"""Mark ops on the bulk path (the mark columns — action 7, `expand` 0x94, `mark_name` 0xA5 —
pass through decode, change re-derivation, merge, apply and save; mark *semantics* are out of scope).  Histories with
marks are built op by op (tests/change_builder.py: the legacy JS codec predates marks, so these vectors are pinned by the
oracle alone: PARITY UNPINNED against the reference, stated in DESIGN.md 2) and run through the engine as a change-only
load, an apply onto the first change, a saved document and a merge of two replicas, each against the oracle."""
import oracle_lib as o
from change_builder import DEL, MAKE_TEXT, MARK, SET, actor_id, build_change, mark_ops, op, v_bool, v_int, v_str


def marked_text(n_chars=40, n_marks=6):
    """a text of n_chars characters; actor 1 marks ranges of it (bold / link / comment, every expand combination) and types
    inside one; actor 2, concurrently, marks overlapping ranges, unmarks one (a mark with a null value) and deletes marked text"""
    a0, a1, a2 = actor_id(0), actor_id(1), actor_id(2)
    ops = [op(MAKE_TEXT, key="t")]
    for i in range(n_chars):
        ops.append(op(SET, obj=(1, 0), elem=(0, None) if i == 0 else (1 + i, 0), insert=True, value=v_str(chr(97 + i % 26))))
    c0, h0 = build_change(a0, 1, 1, [], ops)
    start = n_chars + 2

    def el(i):   # element of character i (0-based) as seen from another actor's change: actor index 1 = a0
        return (2 + i, 1)

    ops1 = []
    names = ["bold", "link", "comment"]
    for k in range(n_marks):
        s, e = (3 * k) % (n_chars - 8), (3 * k) % (n_chars - 8) + 5 + k % 3
        val = v_bool(True) if k % 3 == 0 else (v_str("https://example.org/%d" % k) if k % 3 == 1 else v_int(k))
        ops1 += mark_ops((1, 1), el(s), el(e), names[k % 3], val, expand_before=bool(k & 1), expand_after=bool(k & 2))
    # typing inside the first mark: an element behind the mark begin op (id start + 0 @ a1)
    ops1.append(op(SET, obj=(1, 1), elem=(start, 0), insert=True, value=v_str("X")))
    ops1.append(op(SET, obj=(1, 1), elem=(start + len(ops1) - 1, 0), insert=True, value=v_str("Y")))
    c1, h1 = build_change(a1, 1, start, [h0], ops1, others=[a0])
    ops2 = []
    ops2 += mark_ops((1, 1), el(2), el(20), "bold", v_bool(True), expand_before=True, expand_after=True)
    ops2 += mark_ops((1, 1), el(4), el(9), "link", (0, b""), expand_before=False, expand_after=False)   # unmark: null value, still named
    for i in (5, 6, 7):
        ops2.append(op(DEL, obj=(1, 1), elem=el(i), pred=[el(i)]))
    c2, h2 = build_change(a2, 1, start, [h0], ops2, others=[a0])
    # a later change of actor 1 that has seen both: one more mark whose ends hang off actor 2's mark ops
    ops3 = mark_ops((1, 1), (start, 2), (start + 1, 2), "comment", v_str("nested"), expand_before=False, expand_after=True)
    others3 = sorted([a0, a2])
    idx = {a: 1 + others3.index(a) for a in others3}
    ops3 = [dict(p, obj=(1, idx[a0]), elem=(p["elem"][0], idx[a2])) for p in ops3]
    c3, _ = build_change(a1, 2, start + len(ops1) + len(ops2) + 4, [h1, h2], ops3, others=others3)
    return [c0, c1, c2, c3]


def check_marks(engine):
    """every route, against the oracle: hydrate / heads / change hashes (check_doc), re-saved bytes"""
    import apply_cases as ac
    chunks = marked_text()
    ref = o.Doc()
    ref.apply_changes(chunks)
    want = ref.save(deflate=False)
    # the oracle's own round trip first (the vectors are its own: it must at least be self-consistent)
    assert o.Doc.load(want).save(deflate=False) == want
    # 1. change-only load
    b = engine.load_batch([b"".join(chunks)])
    ac.check_doc(b, 0, ref)
    assert b.save_all(deflate=False)[0] == want
    b.close()
    # 2. the saved document
    b = engine.load_batch([want])
    ac.check_doc(b, 0, ref)
    assert b.save_all(deflate=False)[0] == want
    assert b.save_all(deflate=True)[0] is not None
    b.close()
    # 3. apply onto the base
    base = o.Doc(); base.apply_changes(chunks[:1])
    b = engine.load_batch([base.save(deflate=False)])
    res = b.apply_batch([(0, b"".join(chunks[1:]))])
    assert not any(res), res
    ac.check_doc(b, 0, ref)
    assert b.save_all(deflate=False)[0] == want
    b.close()
    # 4. merge of the two replicas (base + actor 1 | base + actor 2), then the last change applied
    r1 = o.Doc(); r1.apply_changes(chunks[:2])
    r2 = o.Doc(); r2.apply_changes([chunks[0], chunks[2]])
    bd, bs = engine.load_batch([r1.save(deflate=False)]), engine.load_batch([r2.save(deflate=False)])
    assert bd.merge(bs, [(0, 0)]) == 0
    m = o.Doc.load(r1.save(deflate=False)); m.merge(o.Doc.load(r2.save(deflate=False)))
    ac.check_doc(bd, 0, m)
    assert bd.save_all(deflate=False)[0] == m.save(deflate=False)
    res = bd.apply_batch([(0, chunks[3])])
    assert not any(res), res
    ac.check_doc(bd, 0, ref)
    assert bd.save_all(deflate=False)[0] == want
    bd.close(); bs.close()
