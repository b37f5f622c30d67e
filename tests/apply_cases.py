"""apply_changes / load_incremental / trailing-chunk scenarios shared by the CPU-tier (emulation) and GPU tests:
the engine and the oracle get the same bytes; everything observable must agree bit for bit."""
import random

import oracle_lib as o


def check_doc(batch, i, ref):
    batch.check(i)
    assert batch.hydrate_json(i) == ref.hydrate_json(), i
    assert batch.get_heads(i) == ref.get_heads(), i
    assert [c["hash"] for c in batch.get_changes(i)] == [c["hash"] for c in ref.changes()], i   # history order
    assert batch.max_op(i) == ref.max_op()
    assert batch.queue_len(i) == ref.queue_len()


def changes_of(data):
    return [c["raw"] for c in o.Doc.load(bytes(data)).changes()]


def check_apply_tail(engine, docs, cut):
    """every document is rebuilt from its first cut(n) changes, then gets the rest through apply_changes"""
    bases, tails = [], []
    for data in docs:
        chunks = changes_of(data)
        k = cut(len(chunks))
        base = o.Doc()
        base.apply_changes(chunks[:k])
        bases.append(base.save(deflate=False))
        tails.append(chunks[k:])
    b = engine.load_batch(bases)
    # the buffers of the call back to back in one arena (Batch.apply_arena: same call, pointer table built with numpy)
    items = [(i, b"".join(t)) for i, t in enumerate(tails) if t]
    offs = [0]
    for _, buf in items:
        offs.append(offs[-1] + len(buf))
    res = b.apply_arena([i for i, _ in items], b"".join(buf for _, buf in items), offs)
    assert not any(res), res
    saved = b.save_all(deflate=False)
    for i in range(len(docs)):
        ref = o.Doc.load(bases[i])
        ref.apply_changes(tails[i])
        check_doc(b, i, ref)
        assert saved[i] == ref.save(deflate=False), i
    return b


def check_apply_replicas(engine, docs, n_replicas):
    """docs[d * n_replicas + r] = replica r of document d: replica 0 receives the other replicas' changes one replica per
    call — in order, shuffled (queueing inside a call) and with duplicates"""
    nd = len(docs) // n_replicas
    bases = [bytes(docs[d * n_replicas]) for d in range(nd)]
    b = engine.load_batch(bases)
    refs = [o.Doc.load(x) for x in bases]
    rng = random.Random(5)
    for r in range(1, n_replicas):
        items = []
        for d in range(nd):
            have = {c["hash"] for c in refs[d].changes()}
            new = [c["raw"] for c in o.Doc.load(bytes(docs[d * n_replicas + r])).changes() if c["hash"] not in have]
            if d % 3 == 1:
                rng.shuffle(new)
            if d % 3 == 2:
                new = new + new[:2]
            refs[d].apply_changes(new)
            if new:
                items.append((d, b"".join(new)))
        res = b.apply_batch(items)
        assert not any(res), res
        saved = b.save_all(deflate=False)
        for d in range(nd):
            check_doc(b, d, refs[d])
            assert saved[d] == refs[d].save(deflate=False)
    return b


def check_queue_across_calls(engine, base, other):
    """changes delivered before their dependencies wait in the queue (automerge.rs:797-858), are kept by save
    (retain_orphans, :959-963) and apply once the missing change arrives"""
    have = {c["hash"] for c in o.Doc.load(base).changes()}
    new = [c for c in changes_of(other) if o.change_hash(c) not in have]
    assert len(new) >= 2
    b = engine.load_batch([base])
    ref = o.Doc.load(base)
    b.apply_changes(0, new[1:])
    ref.apply_changes(new[1:])
    check_doc(b, 0, ref)
    assert b.queue_len(0) == len(new) - 1
    saved = b.save_all(deflate=False)[0]
    assert saved == ref.save(deflate=False)
    # the saved form (document chunk + orphans) loads back into the same state
    b2 = engine.load_batch([saved])
    check_doc(b2, 0, o.Doc.load(saved))
    b.apply_changes(0, new[:1])
    ref.apply_changes(new[:1])
    check_doc(b, 0, ref)
    assert b.queue_len(0) == 0
    assert b.save_all(deflate=False)[0] == ref.save(deflate=False)


def corrupt(chunk):
    c = bytearray(chunk)
    c[-1] ^= 0x55
    return bytes(c)


def check_trailing_and_policies(engine, data):
    """inputs of the shapes load accepts (automerge.rs:601-705): document + change chunks (save_incremental), change
    chunks alone, a change stream missing its first change (MissingDeps), a bad chunk in the tail under both policies"""
    chunks = changes_of(data)
    assert len(chunks) >= 4
    k = len(chunks) // 2
    base = o.Doc()
    base.apply_changes(chunks[:k])
    head = base.save(deflate=False)
    inputs = [
        head + b"".join(chunks[k:]),                                 # document + incremental changes
        b"".join(chunks),                                            # changes only
        b"".join(chunks[1:]),                                        # changes only, first one missing
        head + b"".join(chunks[k:-1]) + corrupt(chunks[-1]),         # bad checksum in the tail
        head + b"".join(chunks[k + 1:]),                             # tail whose first change is missing: queued
        head + chunks[k][:len(chunks[k]) // 2],                      # truncated tail
        b"",
    ]
    for policy in ("error", "ignore"):
        b = engine.load_batch(inputs, on_partial_load=policy)
        for i, inp in enumerate(inputs):
            try:
                ref = o.Doc.load(inp, on_partial_load=policy)
            except o.OracleError as ex:
                kind, _ = b.status(i)
                assert kind != 0 and (ex.kind == 4 or kind == ex.kind), (policy, i, kind, ex)
                continue
            check_doc(b, i, ref)
            assert b.save_all(deflate=False)[i] == ref.save(deflate=False), (policy, i)


def check_incremental_and_errors(engine, data):
    chunks = changes_of(data)
    k = len(chunks) // 2
    base = o.Doc()
    base.apply_changes(chunks[:k])
    head = base.save(deflate=False)
    # load_incremental: the chunks in front of the first unreadable one are applied (automerge.rs:758-764)
    stream = b"".join(chunks[k:k + 2]) + corrupt(chunks[k + 2]) + b"".join(chunks[k + 3:])
    b = engine.load_batch([head, head, head])
    res = b.apply_batch([(0, stream)], incremental=True)
    assert res == [0]
    ref = o.Doc.load(head)
    ref.load_incremental(stream)
    check_doc(b, 0, ref)
    # apply_changes proper takes parsed changes: an unreadable one fails the call for that document only
    # (Change::try_from does not compare the checksum, change.rs:161-180: a damaged change that still parses is a
    # different change with a different hash)
    bad = corrupt(chunks[k])
    res = b.apply_batch([(1, bad), (2, chunks[k])])
    ref1 = o.Doc.load(head)
    try:
        ref1.apply_changes([bad])
        # a damaged change the document format cannot carry (its re-derived chunk hashes differently) is refused
        assert res[0] in (0, 91), res
        if res[0] == 91:
            ref1 = o.Doc.load(head)
    except o.OracleError as ex:
        assert res[0] == ex.kind, (res, ex)
        ref1 = o.Doc.load(head)
    assert res[1] == 0, res
    check_doc(b, 1, ref1)
    trunc = chunks[k][:-3]
    assert b.apply_batch([(1, trunc)]) == [1]
    check_doc(b, 1, ref1)
    ref2 = o.Doc.load(head)
    ref2.apply_changes([chunks[k]])
    check_doc(b, 2, ref2)
    # idempotent
    assert b.apply_batch([(2, chunks[k])]) == [0]
    check_doc(b, 2, ref2)
    # compressed changes (chunk type 2) are the changes they inflate to (chunk.rs:77-97)
    big = [c for c in chunks[k:] if len(c) > 256]
    if big:
        comp = o.change_compress(big[0])
        assert comp[8] == 2
        b3 = engine.load_batch([head])
        ref3 = o.Doc.load(head)
        pre = chunks[k:chunks.index(big[0])]
        b3.apply_changes(0, pre + [comp])
        ref3.apply_changes(pre + [big[0]])
        check_doc(b3, 0, ref3)


def check_incremental_with_documents(engine, replicas):
    """load_incremental accepts whole saved documents too (automerge.rs:723-727; a document chunk in a change stream
    stands for its changes, storage/load.rs:84-100): replica 0 takes in the saved form of every other replica, and a
    stream that mixes a document chunk with change chunks"""
    base = bytes(replicas[0])
    b = engine.load_batch([base, base])
    ref = o.Doc.load(base)
    for r in replicas[1:]:
        assert b.apply_batch([(0, bytes(r))], incremental=True) == [0]
        ref.load_incremental(bytes(r))
        check_doc(b, 0, ref)
    assert b.save_all(deflate=False)[0] == ref.save(deflate=False)
    ref2 = o.Doc.load(base)
    extra = changes_of(replicas[2])[-1]
    stream = bytes(replicas[1]) + extra + bytes(replicas[2])
    assert b.apply_batch([(1, stream)], incremental=True) == [0]
    ref2.load_incremental(stream)
    check_doc(b, 1, ref2)
    # a damaged document chunk stops the stream there
    bad = bytearray(replicas[1]); bad[len(bad) // 2] ^= 0x41
    stream = extra + bytes(bad) + bytes(replicas[2])
    b3 = engine.load_batch([base])
    ref3 = o.Doc.load(base)
    assert b3.apply_batch([(0, stream)], incremental=True) == [0]
    ref3.load_incremental(stream)
    check_doc(b3, 0, ref3)


def check_save_after_roundtrip(engine, replicas):
    """save_after(heads) on one side, load_incremental on the other (automerge.rs:993-1000, 730-769): replica 1 sends
    what replica 0's heads have not seen; unknown heads are ignored; save_after([]) is the whole history"""
    b = engine.load_batch([bytes(replicas[0]), bytes(replicas[1])])
    r0, r1 = o.Doc.load(bytes(replicas[0])), o.Doc.load(bytes(replicas[1]))
    h0 = r0.get_heads()
    cases = (h0, [], h0 + [b"\x07" * 32], r1.get_heads())
    for heads in cases:
        assert b.save_after(1, heads) == r1.save_after(heads)
    # the batched form (one device pass over many (document, heads) pairs) gives the same bytes
    items = [(1, h) for h in cases] + [(0, h) for h in cases] + [(1, r1.get_heads()[:1]), (0, [c["hash"] for c in r0.changes()][:2])]
    want = [(r1 if d == 1 else r0).save_after(h) for d, h in items]
    assert b.save_after_many(items) == want
    assert b.save_after_many([]) == []
    delta = b.save_after(1, h0)
    assert b.apply_batch([(0, delta)], incremental=True) == [0]
    r0.load_incremental(r1.save_after(h0))
    check_doc(b, 0, r0)
    assert b.save_all(deflate=False)[0] == r0.save(deflate=False)


def check_error_precedence(engine, data):
    """chunk parse errors come first, then the checksum, then whatever reconstruction finds (storage/load.rs:56-114,
    automerge.rs:601-656): a damaged document reports BadChecksum even when its damaged bytes no longer decode"""
    n = len(data)
    flip = bytearray(data); flip[n // 2] ^= 0x01                        # one bit of column data
    smash = bytearray(data); smash[n // 3:n // 3 + 96] = b"\xff" * 96    # columns that cannot be decoded any more
    tail = bytearray(data); tail[-1] ^= 0x80
    magic = bytearray(data); magic[1] ^= 0x10                            # not a chunk at all
    cases = [bytes(data), bytes(flip), bytes(smash), bytes(tail), bytes(magic)]
    b = engine.load_batch(cases)
    for i, c in enumerate(cases):
        try:
            o.Doc.load(c)
            want = 0
        except o.OracleError as e:
            want = e.kind
        kind, _ = b.status(i)
        got = 4 if 20 <= kind < 40 else kind     # the oracle reports every reconstruction error as kind 4
        assert got == want, (i, kind, want)
    assert b.status(0)[0] == 0 and b.status(1)[0] == 2 and b.status(2)[0] == 2 and b.status(4)[0] == 1


def check_duplicate_seq_applies_prefix(engine):
    """apply_changes([A, D, B]) where D repeats an (actor, seq) of the document with other content: the reference
    returns DuplicateSeqNumber from the middle of its loop (automerge.rs:804-812), so A stays applied, B is never
    looked at, and a change queued in front of D stays queued."""
    from change_builder import SET, actor_id, build_change, op, v_uint
    a0, a1, a2, a3 = (actor_id(i) for i in range(4))
    c0, h0 = build_change(a0, 1, 1, [], [op(SET, key="a", value=v_uint(1))])
    A, hA = build_change(a1, 1, 2, [h0], [op(SET, key="b", value=v_uint(2))])
    Q, _ = build_change(a3, 1, 9, [b"\x11" * 32], [op(SET, key="q", value=v_uint(9))])      # dependency unknown: queued
    D, _ = build_change(a0, 1, 1, [], [op(SET, key="z", value=v_uint(26))])                  # (a0, seq 1) again
    B, _ = build_change(a2, 1, 2, [h0], [op(SET, key="c", value=v_uint(3))])
    base = o.Doc()
    base.apply_changes([c0])
    saved = base.save(deflate=False)
    ref = o.Doc.load(saved)
    try:
        ref.apply_changes([A, Q, D, B])
        raise AssertionError("the oracle accepted a duplicate seq number")
    except o.OracleError as e:
        kind = e.kind
    b = engine.load_batch([saved, saved])
    res = b.apply_batch([(0, A + Q + D + B), (1, A + B)])
    assert res == [kind, 0], res
    check_doc(b, 0, ref)                       # A applied, Q queued, no B
    assert b.queue_len(0) == 1
    assert b.save_all(deflate=False)[0] == ref.save(deflate=False)
    ref1 = o.Doc.load(saved)
    ref1.apply_changes([A, B])
    check_doc(b, 1, ref1)


def check_pred_naming_no_op(engine):
    """A change whose op names a pred that does not exist.  The reference applies it (the pred finds no op to succeed,
    op_set.rs:194-203) but cannot round-trip it: its own save drops the dangling pred, the re-derived change hashes
    differently, and its own load of that save fails the heads check.  The engine re-verifies every document it
    changes, so it refuses the change up front (AMH_DOC_UNSUPPORTED, 91) and leaves the document as it was —
    INTEGRATION.md lists this under "Inputs answered with 91"."""
    from change_builder import SET, actor_id, build_change, op, v_uint
    a0, a1 = actor_id(0), actor_id(1)
    c0, h0 = build_change(a0, 1, 1, [], [op(SET, key="a", value=v_uint(1))])
    bad, _ = build_change(a1, 1, 2, [h0], [op(SET, key="a", value=v_uint(2), pred=[(77, 1)])], others=[a0])
    base = o.Doc()
    base.apply_changes([c0])
    saved = base.save(deflate=False)
    ref = o.Doc.load(saved)
    ref.apply_changes([bad])                       # accepted
    try:
        o.Doc.load(ref.save(deflate=False))
        round_trips = True
    except o.OracleError:
        round_trips = False
    assert not round_trips
    b = engine.load_batch([saved])
    res = b.apply_batch([(0, bad)])
    assert res == [91], res
    check_doc(b, 0, o.Doc.load(saved))             # unchanged
