"""Shapes that turn a per-list or per-group serial step into the whole job: one key overwritten tens of thousands of
times, tens of thousands of new keys or sibling elements landing in one gap, many children behind one new element,
one long typed run.  Each case gives the change chunks of a small history; the checks feed them to the engine as a
change-only load, as an apply onto the loaded first change, and as a saved document, against the oracle."""
import random
import time

import oracle_lib as o
from change_builder import DEL, INC, MAKE_LIST, MAKE_TEXT, SET, actor_id, build_change, op, v_counter, v_int, v_str, v_uint


def overwritten_key(n=50000):
    """one map key set n times, each op overwriting the one before (one key group of n ops, n - 1 of them with a successor)"""
    ops = [op(SET, key="k", value=v_uint(i), pred=[(i, 0)] if i else []) for i in range(n)]
    c, _ = build_change(actor_id(1), 1, 1, [], ops)
    return [c]


def counter_increments(n=20000):
    """one counter incremented n times: every increment names the counter op as its pred (one op with n successors)"""
    c0, h0 = build_change(actor_id(1), 1, 1, [], [op(SET, key="c", value=v_counter(0))])
    ops = [op(INC, key="c", value=v_int(1), pred=[(1, 0)]) for _ in range(n)]
    c1, _ = build_change(actor_id(1), 2, 2, [h0], ops)
    return [c0, c1]


def wide_overwrite(n=3000, n_actors=4):
    """n concurrent values of one key (no preds, several actors), then one op that overwrites them all: one op with n preds"""
    a0 = actor_id(0)
    c0, h0 = build_change(a0, 1, 1, [], [op(SET, key="base", value=v_uint(0))])
    out, heads, per = [c0], [], n // n_actors
    for j in range(1, n_actors + 1):
        c, h = build_change(actor_id(j), 1, 2, [h0], [op(SET, key="k", value=v_uint(j * per + i)) for i in range(per)])
        out.append(c); heads.append(h)
    actors = sorted(actor_id(j) for j in range(1, n_actors + 1))
    others = [a for a in actors]
    preds = sorted(((2 + i, 1 + others.index(actor_id(j))) for j in range(1, n_actors + 1) for i in range(per)),
                   key=lambda x: (x[0], others[x[1] - 1]))
    c, _ = build_change(a0, 2, 2 + per, heads, [op(SET, key="k", value=v_str("winner"), pred=preds)], others=others)
    out.append(c)
    return out


def deep_nesting(depth=600):
    """maps nested `depth` levels deep, a value at the bottom (the reference's hydrate has no depth limit)"""
    from change_builder import MAKE_MAP
    ops = [op(MAKE_MAP, obj=None if i == 0 else (i, 0), key="k") for i in range(depth)]
    ops.append(op(SET, obj=(depth, 0), key="bottom", value=v_str("here")))
    c, _ = build_change(actor_id(1), 1, 1, [], ops)
    return [c]


def new_map_keys(n=20000):
    rng = random.Random(11)
    keys = ["%s-%d" % ("".join(rng.choice("abcdefgh") for _ in range(rng.randint(1, 12))), i) for i in range(n)]
    ops = [op(SET, key=k, value=v_uint(i)) for i, k in enumerate(keys)]
    c, _ = build_change(actor_id(1), 1, 1, [], ops)
    return [c]


def head_siblings(n_actors=8, per=1250):
    """a list, then n_actors concurrent changes of `per` inserts each, every one of them at HEAD"""
    a0 = actor_id(0)
    c0, h0 = build_change(a0, 1, 1, [], [op(MAKE_LIST, key="l")])
    out = [c0]
    for j in range(1, n_actors + 1):
        ops = [op(SET, obj=(1, 1), elem=(0, None), insert=True, value=v_uint(j * per + i)) for i in range(per)]
        c, _ = build_change(actor_id(j), 1, 2, [h0], ops, others=[a0])
        out.append(c)
    return out


def siblings_behind_one_element(n_actors=8, per=1250):
    """as head_siblings, but every insert goes behind the same element, which arrives in the same call"""
    a0 = actor_id(0)
    c0, h0 = build_change(a0, 1, 1, [], [op(MAKE_LIST, key="l"), op(SET, obj=(1, 0), elem=(0, None), insert=True, value=v_str("e"))])
    out = [c0]
    for j in range(1, n_actors + 1):
        ops = [op(SET, obj=(1, 1), elem=(2, 1), insert=True, value=v_uint(j * per + i)) for i in range(per)]
        c, _ = build_change(actor_id(j), 1, 3, [h0], ops, others=[a0])
        out.append(c)
    return out


def typed_run(n=5000, base=100):
    """a text of `base` characters, then one change typing n characters in a row in its middle and deleting a few old ones"""
    a0, a1 = actor_id(0), actor_id(1)
    ops = [op(MAKE_TEXT, key="t")]
    for i in range(base):
        ops.append(op(SET, obj=(1, 0), elem=(0, None) if i == 0 else (1 + i, 0), insert=True, value=v_str(chr(97 + i % 26))))
    c0, h0 = build_change(a0, 1, 1, [], ops)
    start = base + 2
    ops = []
    for i in range(n):
        ops.append(op(SET, obj=(1, 1), elem=(base // 2 + 1, 1) if i == 0 else (start + i - 1, 0), insert=True, value=v_str(chr(65 + i % 26))))
    for k in range(5):
        ops.append(op(DEL, obj=(1, 1), elem=(3 + k, 1), pred=[(3 + k, 1)]))
    c1, _ = build_change(a1, 1, start, [h0], ops, others=[a0])
    return [c0, c1]


def keystrokes(n=20000):
    """a text typed one character per change by one actor: as many changes as ops, one dependency chain n deep"""
    a0 = actor_id(0)
    c, h = build_change(a0, 1, 1, [], [op(MAKE_TEXT, key="t")])
    chunks = [c]
    for i in range(n):
        c, h = build_change(a0, 2 + i, 2 + i, [h], [op(SET, obj=(1, 0), elem=(0, None) if i == 0 else (1 + i, 0), insert=True, value=v_str(chr(97 + i % 26)))])
        chunks.append(c)
    return chunks


CASES = {"keystrokes": keystrokes, "deep_nesting": deep_nesting, "wide_overwrite": wide_overwrite, "overwritten_key": overwritten_key, "counter_increments": counter_increments, "new_map_keys": new_map_keys,
         "head_siblings": head_siblings, "siblings_behind_one_element": siblings_behind_one_element, "typed_run": typed_run}


def check_case(engine, chunks, budget_s=None, log=None):
    """change-only load, apply onto the first change, and the saved document, each against the oracle"""
    import apply_cases as ac
    ref = o.Doc()
    ref.apply_changes(chunks)
    want = ref.save(deflate=False)
    times = {}
    t = time.time()
    b = engine.load_batch([b"".join(chunks)])
    times["change_only_load"] = time.time() - t
    ac.check_doc(b, 0, ref)
    assert b.save_all(deflate=False)[0] == want
    b.close()
    if len(chunks) > 1:
        base = o.Doc()
        base.apply_changes(chunks[:1])
        b = engine.load_batch([base.save(deflate=False)])
        t = time.time()
        res = b.apply_batch([(0, b"".join(chunks[1:]))])
        times["apply"] = time.time() - t
        assert not any(res), res
        ac.check_doc(b, 0, ref)
        assert b.save_all(deflate=False)[0] == want
        b.close()
    t = time.time()
    b = engine.load_batch([want])
    times["document_load"] = time.time() - t
    ac.check_doc(b, 0, ref)
    t = time.time()
    assert b.save_all(deflate=False)[0] == want
    times["save"] = time.time() - t
    b.close()
    if log is not None:
        log(times)
    if budget_s is not None:
        assert max(times.values()) < budget_s, times
    return times


def hostile_document(rows):
    """a well-formed document chunk of about 60 bytes (valid checksum) whose op id and succ-count columns are one run
    of `rows` values each: the count pass accepts it, and everything sized from the counts is sized for `rows` ops"""
    import hashlib
    from change_builder import sleb, uleb
    id_ctr = sleb(rows) + sleb(1)          # delta column: one run of +1
    succ_num = sleb(rows) + uleb(0)        # no op has a successor
    actor = bytes(range(16))
    body = uleb(1) + uleb(len(actor)) + actor + uleb(0)            # actors, heads
    body += uleb(0)                                                # change columns: none
    body += uleb(2) + uleb(35) + uleb(len(id_ctr)) + uleb(128) + uleb(len(succ_num))
    body += id_ctr + succ_num
    tail = bytes([0]) + uleb(len(body)) + body
    return b"\x85\x6f\x4a\x83" + hashlib.sha256(tail).digest()[:4] + tail


def check_hostile_documents(engine, good_docs, rows_hostile, max_doc_rows):
    """hostile documents between good ones: with a per-document limit they fail alone (TooLarge) and the others load;
    with a byte budget the largest documents of an oversized batch fail and the rest loads"""
    import oracle_lib as o
    engine.set_limits(max_doc_rows=max_doc_rows)
    try:
        docs = []
        for g in good_docs:
            docs += [hostile_document(rows_hostile), g]
        b = engine.load_batch(docs)
        for i, d in enumerate(docs):
            kind, _ = b.status(i)
            if i % 2 == 0:
                assert kind == 90, (i, kind)      # AMH_DOC_TOO_LARGE
            else:
                assert kind == 0, (i, kind)
                assert b.hydrate_json(i) == o.Doc.load(d).hydrate_json()
        b.close()
        # byte budget: room for the small documents only
        sizes = sorted(len(g) for g in good_docs)
        engine.set_limits(max_doc_rows=max_doc_rows, max_batch_bytes=1)
        b = engine.load_batch(list(good_docs))
        assert all(b.status(i)[0] == 90 for i in range(len(good_docs)))
        b.close()
        big = max(range(len(good_docs)), key=lambda i: len(good_docs[i]))
        engine.set_limits(max_doc_rows=max_doc_rows)

        def estimate(docs):   # the engine's own sizing rule (amh_host.cpp, derived_bytes), from the loaded counts
            bb = engine.load_batch(docs)
            st = bb.stats()
            bb.close()
            return st["op_rows"] * 224 + st["succ_entries"] * 72 + st["changes"] * (420 + 4 * 8)
        budget = estimate([g for i, g in enumerate(good_docs) if i != big]) + estimate([good_docs[big]]) // 2
        engine.set_limits(max_doc_rows=max_doc_rows, max_batch_bytes=budget)
        b = engine.load_batch(list(good_docs))
        st = [b.status(i)[0] for i in range(len(good_docs))]
        assert st[big] == 90 and sum(1 for s in st if s == 0) >= 1, st
        for i, g in enumerate(good_docs):
            if st[i] == 0:
                assert b.hydrate_json(i) == o.Doc.load(g).hydrate_json()
        b.close()
    finally:
        engine.set_limits(0, 0)


def check_byte_budget(engine, docs):
    """amh_ctx_set_limits(max_batch_bytes): the smallest budget under which the whole batch loads is found by bisection;
    at that budget every document loads, one byte below it exactly the largest document fails (TooLarge = 90) and the
    others are what they were.  `docs` must have one document clearly larger than the others."""
    def load(budget):
        engine.set_limits(0, budget)
        try:
            b = engine.load_batch(docs)
            st = [b.status(i)[0] for i in range(len(docs))]
            heads = [b.get_heads(i) if st[i] == 0 else None for i in range(len(docs))]
            b.close()
            return st, heads
        finally:
            engine.set_limits(0, 0)
    ref_st, ref_heads = load(1 << 40)
    assert all(s == 0 for s in ref_st)
    lo, hi = 1, 1 << 40          # lo: some document fails; hi: all load
    while hi - lo > 1:
        mid = (lo + hi) // 2
        st, _ = load(mid)
        if all(s == 0 for s in st):
            hi = mid
        else:
            lo = mid
    st, heads = load(hi)
    assert st == ref_st and heads == ref_heads
    st, heads = load(hi - 1)
    largest = max(range(len(docs)), key=lambda i: len(docs[i]))
    assert st[largest] == 90, st
    for i in range(len(docs)):
        if i != largest:
            assert st[i] == 0 and heads[i] == ref_heads[i], (i, st)
