"""Merge scenarios shared by the CPU-tier (emulation) and GPU tests: divergent documents built with
the oracle from generated histories, merged by the engine and by the oracle, compared bit for bit."""
import oracle_lib as o


def split_map_doc(doc_bytes, n_actors=4, rounds=5, common_rounds=3):
    """Two causally closed, divergent documents out of one generated map document: both share the first
    `common_rounds` rounds; A continues with the later changes of the first half of the actors, B with the others."""
    chunks = [c["raw"] for c in o.Doc.load(doc_bytes).changes()]
    common = chunks[:common_rounds * n_actors]
    a, b = list(common), list(common)
    for r in range(common_rounds, rounds):
        for k in range(n_actors):
            c = chunks[r * n_actors + k]
            if r == common_rounds:
                (a if k < n_actors // 2 else b).append(c)
    da, db = o.Doc(), o.Doc()
    da.apply_changes(a)
    db.apply_changes(b)
    return da.save(deflate=False), db.save(deflate=False)


def check_merge(engine, dst_docs, src_docs, pairs):
    """engine: dst[a].merge(src[b]) for the pairs; oracle the same; everything observable must agree"""
    bd, bs = engine.load_batch(dst_docs), engine.load_batch(src_docs)
    rc = bd.merge(bs, pairs)
    assert rc == 0
    paired = dict(pairs)
    for a, data in enumerate(dst_docs):
        ref = o.Doc.load(data)
        if a in paired:
            ref.merge(o.Doc.load(src_docs[paired[a]]))
        bd.check(a)
        assert bd.hydrate_json(a) == ref.hydrate_json(), a
        assert bd.get_heads(a) == ref.get_heads(), a
        assert [c["hash"] for c in bd.get_changes(a)] == [c["hash"] for c in ref.changes()], a   # history order
        assert bd.max_op(a) == ref.max_op()
    saved = bd.save_all(deflate=False)
    for a, data in enumerate(dst_docs):
        ref = o.Doc.load(data)
        if a in paired:
            ref.merge(o.Doc.load(src_docs[paired[a]]))
        assert saved[a] == ref.save(deflate=False), a
    return bd


def check_merge_many(engine, reps):
    """reps[r][i] = replica r of document i.  One merge_many call == replica 0 merging replicas 1.. one after the other
    (oracle); state, heads, history order and saved bytes must agree.  Sources are given per document in a rotated
    order and one document skips a source, so that the per-source bookkeeping is exercised."""
    n_rep, n_docs = len(reps), len(reps[0])
    bd = engine.load_batch(reps[0])
    others = [engine.load_batch(reps[r]) for r in range(1, n_rep)]
    live0 = engine.pool_bytes()[0]
    for k, ob in enumerate(others):
        if k % 2 == 0:     # half of the sources give their op rows back first: the merge only reads their history
            want = (ob.get_heads(0), [c["hash"] for c in ob.get_changes(0)])
            ob.release_ops()
            assert (ob.get_heads(0), [c["hash"] for c in ob.get_changes(0)]) == want
            try:
                ob.hydrate_json(0)
                raise AssertionError("hydrate of a released batch must fail")
            except AssertionError:
                raise
            except Exception:
                pass
    assert engine.pool_bytes()[0] < live0
    src_idx = [[i for i in range(n_docs)] for _ in range(n_rep - 1)]
    if n_docs > 1 and n_rep > 2:
        src_idx[1][1] = None
    rc = bd.merge_many(others, list(range(n_docs)), src_idx)
    assert rc == 0
    saved = bd.save_all(deflate=False)
    for i in range(n_docs):
        ref = o.Doc.load(bytes(reps[0][i]))
        for s in range(n_rep - 1):
            if src_idx[s][i] is not None:
                ref.merge(o.Doc.load(bytes(reps[s + 1][src_idx[s][i]])))
        bd.check(i)
        assert bd.hydrate_json(i) == ref.hydrate_json(), i
        assert bd.get_heads(i) == ref.get_heads(), i
        assert [c["hash"] for c in bd.get_changes(i)] == [c["hash"] for c in ref.changes()], i
        assert saved[i] == ref.save(deflate=False), i
    return bd


def check_merge_into_queued(engine, docs_pairs):
    """docs_pairs: [(base, other)] replicas of one document each.  dst = base with all but the first of other's new
    changes delivered early (they wait in the queue); src = base + that first change.  The merge brings the missing
    dependency, so the queue drains inside the merge (automerge.rs:821-825, 938).  One pair is left without a queue so
    that one call covers both routes."""
    dst_docs, src_docs, refs, early = [], [], [], []
    for k, (base, other) in enumerate(docs_pairs):
        have = {c["hash"] for c in o.Doc.load(base).changes()}
        new = [c["raw"] for c in o.Doc.load(other).changes() if c["hash"] not in have]
        assert len(new) >= 2
        src = o.Doc.load(base)
        src.apply_changes(new[:1])
        dst_docs.append(base)
        src_docs.append(src.save(deflate=False))
        early.append(new[1:] if k != 1 else [])
    bd, bs = engine.load_batch(dst_docs), engine.load_batch(src_docs)
    for i, e in enumerate(early):
        ref = o.Doc.load(dst_docs[i])
        if e:
            bd.apply_changes(i, e)
            ref.apply_changes(e)
            assert bd.queue_len(i) == len(e) == ref.queue_len()
        refs.append(ref)
    assert bd.merge(bs, [(i, i) for i in range(len(dst_docs))]) == 0
    saved = bd.save_all(deflate=False)
    for i, ref in enumerate(refs):
        ref.merge(o.Doc.load(src_docs[i]))
        bd.check(i)
        assert bd.queue_len(i) == ref.queue_len() == 0
        assert bd.hydrate_json(i) == ref.hydrate_json(), i
        assert bd.get_heads(i) == ref.get_heads(), i
        assert [c["hash"] for c in bd.get_changes(i)] == [c["hash"] for c in ref.changes()], i
        assert saved[i] == ref.save(deflate=False), i


def check_merge_many_duplicate_seq(engine):
    """amh_merge_many where the SECOND of three sources holds a change that repeats an (actor, seq) of the document with
    other content.  The sequential reference merges: merge(s0) succeeds, merge(s1) returns DuplicateSeqNumber from the
    middle of apply_changes (automerge.rs:804-812) with the changes in front of the offender applied, and a caller that
    stops at the error never merges s2.  amh_merge_many hands every source over as one apply_changes call and stops the
    same way: the document equals the oracle's after merge(s0) + the failed merge(s1); a second document of the same call
    merges all three of its (clean) sources."""
    from change_builder import SET, actor_id, build_change, op, v_uint
    a0, a1, a2, a3 = (actor_id(i) for i in range(4))
    c0, h0 = build_change(a0, 1, 1, [], [op(SET, key="a", value=v_uint(1))])
    c1, h1 = build_change(a0, 2, 2, [h0], [op(SET, key="b", value=v_uint(2))])
    A, hA = build_change(a1, 1, 2, [h0], [op(SET, key="s0", value=v_uint(10))])                 # source 0: clean
    G, hG = build_change(a2, 1, 2, [h0], [op(SET, key="s1", value=v_uint(11))])                 # source 1: a good change ...
    D, _ = build_change(a0, 2, 2, [h0], [op(SET, key="z", value=v_uint(26))])                   # ... and (a0, seq 2) again, other content
    B, _ = build_change(a3, 1, 2, [h0], [op(SET, key="s2", value=v_uint(12))])                  # source 2: clean, never reached

    def doc(chs):
        d = o.Doc(); d.apply_changes(chs)
        return d.save(deflate=False)

    dst, s0, s1, s2 = doc([c0, c1]), doc([c0, A]), doc([c0, G, D]), doc([c0, B])
    s1_clean = doc([c0, G])
    ref = o.Doc.load(dst)
    ref.merge(o.Doc.load(s0))
    try:
        ref.merge(o.Doc.load(s1))
        raise AssertionError("the oracle accepted a duplicate seq number")
    except o.OracleError as e:
        assert o.KINDS.get(e.kind) == "duplicate_seq", e
    ref_ok = o.Doc.load(dst)
    for s in (s0, s1_clean, s2):
        ref_ok.merge(o.Doc.load(s))
    bd = engine.load_batch([dst, dst])
    srcs = [engine.load_batch([s0, s0]), engine.load_batch([s1, s1_clean]), engine.load_batch([s2, s2])]
    rc = bd.merge_many(srcs, [0, 1], [[0, 1]] * 3)
    assert rc != 0, "a duplicate seq number must be reported"
    import apply_cases as ac
    ac.check_doc(bd, 0, ref)
    ac.check_doc(bd, 1, ref_ok)
    saved = bd.save_all(deflate=False)
    assert saved[0] == ref.save(deflate=False) and saved[1] == ref_ok.save(deflate=False)
    for b in srcs:
        b.close()
    bd.close()


def check_merge_many_numpy(engine, reps):
    """the index arguments of merge_many as numpy arrays (handed to the C ABI as they are) give what the lists give"""
    import numpy as np
    n_rep, n_docs = len(reps), len(reps[0])
    out = []
    for form in ("lists", "numpy"):
        bd = engine.load_batch(reps[0])
        others = [engine.load_batch(reps[r]) for r in range(1, n_rep)]
        ids = list(range(n_docs))
        if form == "lists":
            rc = bd.merge_many(others, ids, [ids] * (n_rep - 1))
        else:
            a = np.arange(n_docs, dtype=np.uint32)
            rc = bd.merge_many(others, a, np.tile(a, (n_rep - 1, 1)))
        assert rc == 0
        out.append(bd.save_all(deflate=False))
        for b in others:
            b.close()
        bd.close()
    # all source replicas of all documents as ONE batch, named once per source with shifted indices
    bd = engine.load_batch(reps[0])
    wide = engine.load_batch([d for r in range(1, n_rep) for d in reps[r]])
    wide.release_ops()
    a = np.arange(n_docs, dtype=np.uint32)
    src = (np.tile(a, (n_rep - 1, 1)) + (np.arange(n_rep - 1, dtype=np.uint32) * np.uint32(n_docs))[:, None]).astype(np.uint32)
    assert bd.merge_many([wide] * (n_rep - 1), a, src) == 0
    out.append(bd.save_all(deflate=False))
    wide.close(); bd.close()
    assert out[0] == out[1] == out[2]
    ref = o.Doc.load(bytes(reps[0][0]))
    for r in range(1, n_rep):
        ref.merge(o.Doc.load(bytes(reps[r][0])))
    assert out[1][0] == ref.save(deflate=False)


def check_merge_of_a_merged_batch(engine, reps):
    """a source that is itself the result of a merge (it adopted its changes: no per-change counts of its own) merged into
    a third replica: the same as the oracle's ref0.merge(ref1.merge(ref2))"""
    n_docs = len(reps[0])
    ids = list(range(n_docs))
    mid = engine.load_batch(reps[1])
    last = engine.load_batch(reps[2])
    assert mid.merge_many([last], ids, [ids]) == 0
    dst = engine.load_batch(reps[0])
    assert dst.merge_many([mid], ids, [ids]) == 0
    saved = dst.save_all(deflate=False)
    for i in range(n_docs):
        r1 = o.Doc.load(bytes(reps[1][i])); r1.merge(o.Doc.load(bytes(reps[2][i])))
        r0 = o.Doc.load(bytes(reps[0][i])); r0.merge(r1)
        assert dst.get_heads(i) == r0.get_heads(), i
        assert [c["hash"] for c in dst.get_changes(i)] == [c["hash"] for c in r0.changes()], i
        assert saved[i] == r0.save(deflate=False), i
    for b in (mid, last, dst):
        b.close()
