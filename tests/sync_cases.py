"""Sync protocol scenarios shared by the CPU-tier (emulation) and GPU tests: conversations between two peers, driven by
the oracle's generate / receive (oracle/amo_sync.hpp), with the engine generating every message beside the oracle from the
same State (amh_generate_sync_messages); messages and States must agree byte for byte.  The scenarios restate the
reference's own sync tests (rust/automerge/src/sync.rs:662-1016) with fixed actors and explicit change graphs."""
import random
import struct

import oracle_lib as o
from automerge_amd import SyncHave, SyncMessage, SyncState
from change_builder import SET, actor_id, build_change, op, v_str, v_uint


# ---- State <-> the oracle's flat form (oracle/amo_sync.hpp: state_to_blob) ----
def to_blob(st):
    def hs(l):
        l = l or []
        return struct.pack("<I", len(l)) + b"".join(l)
    out = bytes([(1 if st.their_heads is not None else 0) | (2 if st.their_need is not None else 0) |
                 (4 if st.their_have is not None else 0) | (8 if st.in_flight else 0)])
    out += hs(st.shared_heads) + hs(st.last_sent_heads) + hs(st.their_heads) + hs(st.their_need)
    haves = st.their_have or []
    out += struct.pack("<I", len(haves))
    for h in haves:
        out += hs(h.last_sync) + struct.pack("<I", len(h.bloom)) + h.bloom
    return out + hs(sorted(st.sent_hashes))


def from_blob(b):
    pos = [1]

    def u32():
        (v,) = struct.unpack_from("<I", b, pos[0])
        pos[0] += 4
        return v

    def hs():
        n = u32()
        out = [b[pos[0] + 32 * i: pos[0] + 32 * i + 32] for i in range(n)]
        pos[0] += 32 * n
        return out
    st = SyncState()
    flags = b[0]
    st.shared_heads, st.last_sent_heads = hs(), hs()
    th, tn = hs(), hs()
    st.their_heads = th if flags & 1 else None
    st.their_need = tn if flags & 2 else None
    haves = []
    for _ in range(u32()):
        ls = hs()
        bl = u32()
        haves.append(SyncHave(ls, b[pos[0]:pos[0] + bl]))
        pos[0] += bl
    st.their_have = haves if flags & 4 else None
    st.sent_hashes = set(hs())
    st.in_flight = bool(flags & 8)
    return st


def same_state(a, b):
    return to_blob(a) == to_blob(b)


# ---- documents with explicit change graphs ----
class Author:
    """one actor writing changes on top of given deps; keeps its seq and op counter"""

    def __init__(self, i):
        self.actor, self.seq, self.max_op = actor_id(i), 0, 0

    def change(self, deps, key="x", value=None, start_op=None):
        self.seq += 1
        start = start_op if start_op is not None else self.max_op + 1
        chunk, h = build_change(self.actor, self.seq, start, deps, [op(SET, key=key, value=value or v_uint(self.seq))])
        self.max_op = start
        return chunk, h


def doc_of(chunks):
    d = o.Doc()
    if chunks:
        d.apply_changes(chunks)
    return d


class Peer:
    """a document held twice: by the oracle and as document `idx` of an engine batch"""

    def __init__(self, doc):
        self.doc, self.state, self.idx = doc, SyncState(), None


class Net:
    """peers of several conversations in one engine batch: every generate step is ONE engine call over all of them"""

    def __init__(self, engine, peers):
        self.engine, self.peers = engine, peers
        for i, p in enumerate(peers):
            p.idx = i
        self.batch = engine.load_batch([p.doc.save(deflate=False) for p in peers])
        self.n_messages = 0

    def generate(self, peers=None):
        peers = self.peers if peers is None else peers
        before = [to_blob(p.state) for p in peers]
        eng_states = [from_blob(b) for b in before]
        msgs = self.batch.generate_sync_messages([(p.idx, s) for p, s in zip(peers, eng_states)])
        out = []
        for p, blob, es, m in zip(peers, before, eng_states, msgs):
            want, new_blob = p.doc.generate_sync_message(blob)
            assert m == want, (p.idx, None if m is None else m.hex(), None if want is None else want.hex())
            assert to_blob(es) == new_blob, p.idx
            if want is not None:
                assert o.sync_message_reencode(want) == want
                assert SyncMessage.decode(want).encode() == want
                self.n_messages += 1
            p.state = from_blob(new_blob)
            out.append(want)
        return out

    def receive(self, peer, msg):
        peer.state = from_blob(peer.doc.receive_sync_message(to_blob(peer.state), msg))
        chunks = SyncMessage.decode(msg).changes
        if chunks:
            self.batch.apply_changes(peer.idx, chunks)
        assert self.batch.get_heads(peer.idx) == peer.doc.get_heads()
        assert self.batch.queue_len(peer.idx) == peer.doc.queue_len()

    def sync(self, a, b, max_iter=10):
        """the loop of sync.rs:989-1016"""
        it = 0
        while True:
            ma, mb = self.generate([a, b])
            if ma is None and mb is None:
                return it
            assert it <= max_iter, "failed to sync"
            if ma is not None:
                self.receive(b, ma)
            if mb is not None:
                self.receive(a, mb)
            it += 1

    def check_docs(self):
        saved = self.batch.save_all(deflate=False)
        for p in self.peers:
            assert [c["hash"] for c in self.batch.get_changes(p.idx)] == [c["hash"] for c in p.doc.changes()], p.idx
            assert saved[p.idx] == p.doc.save(deflate=False), p.idx


def n_changes(msg):
    return len(SyncMessage.decode(msg).changes)


# ---- scenarios ----
def check_simultaneous_messages(engine):
    """sync.rs:688-779 should_allow_simultaneous_messages_during_synchronisation, with the counts it asserts"""
    a1, a2 = Author(1), Author(2)
    ca, cb, da, db = [], [], [], []
    for _ in range(5):
        c, h = a1.change(da); ca.append(c); da = [h]
        c, h = a2.change(db, key="y"); cb.append(c); db = [h]
    p1, p2 = Peer(doc_of(ca)), Peer(doc_of(cb))
    head1, head2 = da[0], db[0]
    net = Net(engine, [p1, p2])
    m12, m21 = net.generate()
    for m in (m12, m21):
        mm = SyncMessage.decode(m)
        assert len(mm.changes) == 0 and mm.have[0].last_sync == []
    net.receive(p1, m21); net.receive(p2, m12)
    m12, m21 = net.generate()
    assert n_changes(m12) == 5 and n_changes(m21) == 5
    net.receive(p1, m21); net.receive(p2, m12)
    assert p1.doc.queue_len() == 0 and p2.doc.queue_len() == 0
    m12, m21 = net.generate()
    assert n_changes(m12) == 0 and n_changes(m21) == 0
    net.receive(p1, m21); net.receive(p2, m12)
    assert p1.state.shared_heads == p2.state.shared_heads
    assert net.generate() == [None, None]
    # one more change on 1: its Have now names the heads both sides shared
    c, h = a1.change(sorted([head1, head2]))
    p1.doc.apply_changes([c]); net.batch.apply_changes(p1.idx, [c])
    (m12,) = net.generate([p1])
    assert sorted(SyncMessage.decode(m12).have[0].last_sync) == sorted([head1, head2])
    net.check_docs()
    net.batch.close()


def check_basic_rules(engine):
    """sync.rs:662-686: a second generate says nothing; no reply to a peer without data when we have none either"""
    a1 = Author(1)
    c, _ = a1.change([])
    p = Peer(doc_of([c]))
    e1, e2 = Peer(o.Doc()), Peer(o.Doc())
    net = Net(engine, [p, e1, e2])
    (m,) = net.generate([p])
    assert m is not None
    assert net.generate([p]) == [None]
    (m1,) = net.generate([e1])
    assert m1 is not None
    net.receive(e2, m1)
    assert net.generate([e2]) == [None]
    net.batch.close()


def _false_positive_change(author_idx, deps, start_op, bloom, tag):
    """a change on top of deps whose hash the filter claims to hold (the search loops of sync.rs:801-823, 860-887)"""
    for i in range(100000):
        a = Author(author_idx)
        c, h = a.change(deps, value=v_str(f"{i} {tag}"), start_op=start_op)
        if o.bloom_contains(bloom, h):
            return c, h
    raise AssertionError("no false positive found")


def check_false_positive_head(engine):
    """sync.rs:781-835: n2 is a false positive of the filter holding {n1}; both sides must still converge"""
    a1 = Author(1)
    base, deps = [], []
    for _ in range(10):
        c, h = a1.change(deps); base.append(c); deps = [h]
    p1, p2 = Peer(doc_of(base)), Peer(o.Doc())
    net = Net(engine, [p1, p2])
    net.sync(p1, p2)
    n1, h1 = Author(3).change(deps, value=v_str("n1"), start_op=11)
    n2, h2 = _false_positive_change(4, deps, 11, o.bloom_from_hashes([h1]), "@ n2")
    for p, c in ((p1, n1), (p2, n2)):
        p.doc.apply_changes([c]); net.batch.apply_changes(p.idx, [c])
        p.state = SyncState.decode(p.state.encode())
    net.sync(p1, p2)
    assert p1.doc.get_heads() == p2.doc.get_heads() == sorted([h1, h2])
    net.check_docs()
    net.batch.close()


def check_chains_of_false_positives(engine):
    """sync.rs:837-900: two false positives in a row on one branch, then a change on top of them"""
    a1 = Author(1)
    base, deps = [], []
    for _ in range(5):
        c, h = a1.change(deps); base.append(c); deps = [h]
    p1, p2 = Peer(doc_of(base)), Peer(o.Doc())
    net = Net(engine, [p1, p2])
    net.sync(p1, p2)
    c5, h5 = a1.change(deps)
    bloom = o.bloom_from_hashes([h5])
    f1, g1 = _false_positive_change(4, deps, 6, bloom, "at 4")
    # the second one builds on the first (seq 2 of the same actor)
    f2 = g2 = None
    for i in range(100000):
        a = Author(4); a.seq = 1
        c, h = a.change([g1], value=v_str(f"{i} again"), start_op=7)
        if o.bloom_contains(bloom, h):
            f2, g2 = c, h
            break
    assert f2 is not None
    a = Author(4); a.seq = 2
    f3, g3 = a.change([g2], value=v_str("final"), start_op=8)
    p1.doc.apply_changes([c5]); net.batch.apply_changes(p1.idx, [c5])
    p2.doc.apply_changes([f1, f2, f3]); net.batch.apply_changes(p2.idx, [f1, f2, f3])
    for p in (p1, p2):
        p.state = SyncState.decode(p.state.encode())
    net.sync(p1, p2)
    assert p1.doc.get_heads() == p2.doc.get_heads() == sorted([h5, g3])
    net.check_docs()
    net.batch.close()


def check_branching_and_merging(engine):
    """sync.rs:902-945: two actors exchanging every change (a ladder of merges), a third branch joining late"""
    a1, a2, a3 = Author(1), Author(2), Author(3)
    c0, h0 = a1.change([])
    d1, d2 = [c0], [c0]
    a2.max_op = a3.max_op = a1.max_op
    c3, h3 = a3.change([h0])
    heads1, heads2 = [h0], [h0]
    for _ in range(1, 20):
        x1, g1 = a1.change(sorted(heads1), key="n1", start_op=max(a1.max_op, a2.max_op) + 1)
        x2, g2 = a2.change(sorted(heads2), key="n2", start_op=a1.max_op)
        a2.max_op = a1.max_op
        d1 += [x1, x2]; d2 += [x2, x1]
        heads1 = heads2 = [g1, g2]
    p1, p2 = Peer(doc_of(d1)), Peer(doc_of(d2))
    net = Net(engine, [p1, p2])
    net.sync(p1, p2)
    p2.doc.apply_changes([c3]); net.batch.apply_changes(p2.idx, [c3])
    top = max(a1.max_op, a2.max_op) + 1
    f1, _ = a1.change(sorted(heads1), key="n1", value=v_str("final"), start_op=top)
    f2, _ = a2.change(sorted(heads2 + [h3]), key="n1", value=v_str("final"), start_op=top)
    p1.doc.apply_changes([f1]); net.batch.apply_changes(p1.idx, [f1])
    p2.doc.apply_changes([f2]); net.batch.apply_changes(p2.idx, [f2])
    net.sync(p1, p2)
    assert p1.doc.get_heads() == p2.doc.get_heads()
    net.check_docs()
    net.batch.close()


def check_in_flight(engine, rounds=12):
    """sync.rs:947-987: a change made between receiving a message and answering it must still get across"""
    peers = []
    for k in range(rounds):
        peers += [Peer(o.Doc()), Peer(o.Doc())]
    net = Net(engine, peers)
    for k in range(rounds):
        p1, p2 = peers[2 * k], peers[2 * k + 1]
        net.sync(p1, p2)
        c2, h2 = Author(100 + 2 * k).change([], value=v_uint(k))
        p2.doc.apply_changes([c2]); net.batch.apply_changes(p2.idx, [c2])
        (m,) = net.generate([p2])
        assert m is not None
        net.receive(p1, m)
        a = Author(101 + 2 * k); a.max_op = 1
        c1, _ = a.change([h2], value=v_uint(1000 + k))
        p1.doc.apply_changes([c1]); net.batch.apply_changes(p1.idx, [c1])
        net.sync(p1, p2)
        assert p1.doc.get_heads() == p2.doc.get_heads()
    net.check_docs()
    net.batch.close()


def check_reset_and_queue(engine):
    """a peer whose last_sync this document does not know gets the reset message (sync.rs:150-166); queued changes make
    their missing deps the `need` of the message and, not being among their heads, keep the Have out of it (:137-148)"""
    a1, a2 = Author(1), Author(2)
    c1, h1 = a1.change([])
    c2, h2 = a1.change([h1])
    c3, h3 = a1.change([h2])
    x1, g1 = a2.change([h1], start_op=2)
    p = Peer(doc_of([c1]))
    q = Peer(doc_of([c1]))
    net = Net(engine, [p, q])
    # reset: their first Have names a hash this document lacks
    p.state.their_heads, p.state.their_need = [h3], []
    p.state.their_have = [SyncHave([h3], o.bloom_from_hashes([h3])), SyncHave([], b"")]
    before = to_blob(p.state)
    (m,) = net.generate([p])
    mm = SyncMessage.decode(m)
    assert mm.heads == [h1] and mm.need == [] and mm.changes == [] and len(mm.have) == 1 and mm.have[0] == SyncHave([], b"")
    assert to_blob(p.state) == before
    # queue: c3 waits for c2; their heads name g1 (unknown here) as well
    q.doc.apply_changes([c3]); net.batch.apply_changes(q.idx, [c3])
    assert q.doc.queue_len() == 1
    q.state.their_heads, q.state.their_need, q.state.their_have = [g1], [], []
    (m,) = net.generate([q])
    mm = SyncMessage.decode(m)
    assert mm.need == sorted([h2, g1]) and mm.have == []
    # their heads name the queued change itself: it is not "needed"
    q.state = SyncState()
    q.state.their_heads, q.state.their_need, q.state.their_have = sorted([h2, h3]), [], []
    (m,) = net.generate([q])
    mm = SyncMessage.decode(m)
    assert mm.need == [h2] and len(mm.have) == 1
    net.batch.close()


def random_states_case(seed, n_docs=6):
    """documents with random change graphs and, per document, a few random States: hashes the document has and hashes
    it lacks in every list, Bloom filters over random subsets (several Haves), duplicates in `need`, every flag"""
    rng = random.Random(seed)
    docs, records = [], []
    for d in range(n_docs):
        n_auth = rng.randint(1, 4)
        authors = [Author(10 * d + k + 1) for k in range(n_auth)]
        chunks, hashes, heads, max_op = [], [], [], 0
        for _ in range(rng.randint(0, 24)):
            a = rng.choice(authors)
            deps = sorted(rng.sample(heads, rng.randint(1, len(heads)))) if heads else []
            own = [h for h, au in hashes if au is a]
            if own and own[-1] not in deps and rng.random() < 0.8:
                deps = sorted(set(deps) | {own[-1]})
            max_op += 1
            c, h = a.change(deps, key=rng.choice("abc"), start_op=max_op)
            chunks.append(c)
            hashes.append((h, a))
            heads = [x for x in heads if x not in deps] + [h]
        doc = doc_of(chunks)
        docs.append(doc)
        known = [h for h, _ in hashes]
        unknown = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(4)]

        def pick(k_max, p_unknown=0.2):
            out = []
            for _ in range(rng.randint(0, k_max)):
                out.append(rng.choice(unknown) if (not known or rng.random() < p_unknown) else rng.choice(known))
            return out
        for _ in range(5):
            st = SyncState()
            st.shared_heads = sorted(set(pick(3)))
            st.last_sent_heads = doc.get_heads() if rng.random() < 0.5 else sorted(set(pick(2)))
            if rng.random() < 0.8:
                st.their_heads = doc.get_heads() if rng.random() < 0.3 else sorted(set(pick(3)))
            if rng.random() < 0.8:
                st.their_need = pick(4, 0.3)
            if rng.random() < 0.85:
                st.their_have = []
                for _ in range(rng.choice([0, 1, 1, 1, 2, 3])):
                    sub = [h for h in known if rng.random() < 0.5]
                    st.their_have.append(SyncHave(sorted(set(pick(2, 0.05 if rng.random() < 0.9 else 0.9))), o.bloom_from_hashes(sub)))
            st.sent_hashes = set(pick(5, 0.3))
            st.in_flight = rng.random() < 0.3
            records.append((d, st))
    return docs, records


def check_random_states(engine, seed):
    docs, records = random_states_case(seed)
    batch = engine.load_batch([d.save(deflate=False) for d in docs])
    if seed % 2:
        batch.release_ops()   # the sender side reads the history only: a batch that gave its op rows back still serves it
    blobs = [to_blob(st) for _, st in records]
    eng_states = [from_blob(b) for b in blobs]
    msgs = batch.generate_sync_messages([(d, s) for (d, _), s in zip(records, eng_states)])
    # the same call with the results left in device memory: sizes must agree (the emulation's "device" is host memory,
    # so there the bytes are compared as well)
    arr, _keep = batch.sync_state_array([(d, from_blob(b)) for (d, _), b in zip(records, blobs)])
    dk, d_msgs, d_off, total, d_sent, d_idx, n_sent = batch.generate_sync_messages_device(arr, len(records))
    assert total == sum(len(m) for m in msgs if m is not None)
    assert n_sent == sum(len(SyncMessage.decode(m).changes) for m in msgs if m is not None)
    if getattr(engine, "is_emulation", False) and total:
        import ctypes
        assert ctypes.string_at(d_msgs, total) == b"".join(m for m in msgs if m is not None)
    kinds = {"none": 0, "message": 0, "with_changes": 0, "reset": 0}
    for (d, _), blob, es, m in zip(records, blobs, eng_states, msgs):
        want, new_blob = docs[d].generate_sync_message(blob)
        assert m == want, (seed, d, None if m is None else m.hex(), None if want is None else want.hex())
        assert to_blob(es) == new_blob, (seed, d)
        if want is None:
            kinds["none"] += 1
        elif new_blob == blob:
            kinds["reset"] += 1
        else:
            kinds["message"] += 1
            kinds["with_changes"] += 1 if SyncMessage.decode(want).changes else 0
    batch.close()
    return kinds


def check_long_history(engine, n=3000):
    """records over a long history (a text typed one character per change) beside records over short ones in the same
    call: the per-change kernels (Bloom bits, length prefixes, copy jobs) find their record across very different sizes"""
    import adversarial_cases as adv
    import oracle_lib as o
    long_doc = o.Doc(); long_doc.apply_changes(adv.keystrokes(n))
    hashes = [c["hash"] for c in long_doc.changes()]
    small_docs, _ = random_states_case(0x10F6)
    docs = [small_docs[0], long_doc] + list(small_docs[1:3])
    batch = engine.load_batch([d.save(deflate=False) for d in docs])
    records = []
    for d, dd in enumerate(docs):
        hs = hashes if dd is long_doc else [c["hash"] for c in dd.changes()]
        for ls in ([], [hs[len(hs) // 2]], [hs[-1]]):
            st = SyncState(); st.their_heads, st.their_need, st.their_have = list(ls), [], [SyncHave(list(ls), b"")]
            records.append((d, st))
        st = SyncState(); st.shared_heads = [hs[len(hs) // 3]]   # a Have with a Bloom filter over the later two thirds
        records.append((d, st))
    blobs = [to_blob(st) for _, st in records]
    msgs = batch.generate_sync_messages([(d, from_blob(b)) for (d, _), b in zip(records, blobs)])
    n_changes = 0
    for (d, _), blob, m in zip(records, blobs, msgs):
        want, _new = docs[d].generate_sync_message(blob)
        assert m == want, (d, None if m is None else len(m), None if want is None else len(want))
        n_changes += len(SyncMessage.decode(m).changes) if m is not None else 0
    batch.close()
    assert n_changes >= n + n // 2
    return n_changes


def check_golden_documents(engine, golden_dir, seed=0x601D):
    """every golden document (the reference-written ones included) and a failed one, each with random States: the records
    of good documents agree with the oracle, the record of the failed document reports its load error"""
    import glob
    import os
    from automerge_amd import AutomergeError
    rng = random.Random(seed)
    files = sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))
    datas = [open(f, "rb").read() for f in files]
    bad = bytearray(datas[0]); bad[len(bad) // 2] ^= 0x55
    datas.insert(1, bytes(bad))
    docs = [None if i == 1 else o.Doc.load(d) for i, d in enumerate(datas)]
    batch = engine.load_batch(datas)
    records = []
    for d, doc in enumerate(docs):
        if doc is None:
            records.append((d, SyncState()))
            continue
        known = [c["hash"] for c in doc.changes()]
        unknown = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(2)]
        for _ in range(3):
            st = SyncState()
            st.shared_heads = sorted(set(rng.sample(known, min(len(known), rng.randint(0, 2)))))
            st.their_heads = sorted(set(rng.sample(known, min(len(known), rng.randint(0, 2))) + rng.sample(unknown, rng.randint(0, 1))))
            st.their_need = rng.sample(known, min(len(known), rng.randint(0, 2)))
            st.their_have = [SyncHave(sorted(rng.sample(known, min(len(known), rng.randint(0, 1)))),
                                      o.bloom_from_hashes([h for h in known if rng.random() < 0.6]))]
            records.append((d, st))
    blobs = [to_blob(st) for _, st in records]
    eng_states = [from_blob(b) for b in blobs]
    msgs = batch.generate_sync_messages([(d, s) for (d, _), s in zip(records, eng_states)])
    n_with_changes = 0
    for (d, _), blob, es, m in zip(records, blobs, eng_states, msgs):
        if docs[d] is None:
            assert isinstance(m, AutomergeError) and to_blob(es) == blob
            continue
        want, new_blob = docs[d].generate_sync_message(blob)
        assert m == want, (files[d - 1 if d > 1 else d], None if m is None else m.hex(), None if want is None else want.hex())
        assert to_blob(es) == new_blob
        n_with_changes += 1 if want is not None and SyncMessage.decode(want).changes else 0
    assert n_with_changes > 0
    batch.close()
