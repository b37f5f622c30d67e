"""CPU tier: the kernel bodies of the device pipeline (automerge_amd/csrc/amh_kernels.inl) compiled for
the host (tests/emu) and driven through the same host orchestration + C ABI, checked against the
oracle: golden documents, the reference's fixtures / fuzz crashers, and generated workloads.  The
generators themselves are checked against the oracle by load + re-save identity."""
import glob
import os

import pytest

import oracle_lib as o
import apply_cases as ac
from emu_engine import emu_engine


@pytest.fixture(scope="module")
def emu():
    return emu_engine()


def compare(batch, i, data):
    ref = o.Doc.load(data)
    batch.check(i)
    assert batch.hydrate_json(i) == ref.hydrate_json()
    mine, theirs = batch.get_changes(i), ref.changes()
    assert [c["hash"] for c in mine] == [c["hash"] for c in theirs]
    assert [c["raw"] for c in mine] == [c["raw"] for c in theirs]
    assert [(c["seq"], c["start_op"], c["len"]) for c in mine] == [(c["seq"], c["start_op"], c["len"]) for c in theirs]
    assert batch.get_heads(i) == ref.get_heads()
    assert batch.max_op(i) == ref.max_op()


def test_golden_documents(emu, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))
    assert len(files) == 60
    docs = [open(f, "rb").read() for f in files]
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        compare(batch, i, d)


def test_reference_written_documents(emu, golden_dir):
    """documents saved by the reference itself (tests/golden/ref_saved_docs.json): re-saved bytes == input"""
    import json
    fx = json.load(open(os.path.join(golden_dir, "ref_saved_docs.json")))["docs"]
    docs = [bytes(f["bytes"]) for f in fx]
    batch = emu.load_batch(docs)
    saved = batch.save_all(deflate=False)
    for i, f in enumerate(fx):
        batch.check(i)
        assert json.loads(batch.hydrate_json(i))[f["key"]][0]["str"] == f["value"]
        assert saved[i] == docs[i]


def test_bad_documents_fail_like_the_oracle(emu, golden_dir):
    """fuzz crashers (tests/test.rs:1405-1415) and invalid fixtures: same accept / reject decision; a
    bad document never poisons its neighbours in the batch."""
    files = sorted(glob.glob(os.path.join(golden_dir, "ref_fuzz_crashers", "*"))) + sorted(glob.glob(os.path.join(golden_dir, "ref_fixtures", "*")))
    good = open(sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))[0], "rb").read()
    docs = []
    for f in files:
        docs += [open(f, "rb").read(), good]
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        try:
            ref = o.Doc.load(d)
        except o.OracleError:
            ref = None
        kind, _ = batch.status(i)
        assert kind != 91   # every fixture shape is handled, change-chunk inputs included (tests/test.rs:1422-1464)
        assert (kind == 0) == (ref is not None), (files[i // 2] if i % 2 == 0 else "good", kind)
        if ref is not None:
            compare(batch, i, d)


def test_generated_map_documents(emu):
    blob, offs = emu.gen_map_docs(0xA070, 24, 4, 5, 50, 128, 4)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(24)]
    for d in docs:
        assert o.Doc.load(d).save(deflate=False) == d   # generator output is canonical reference-format bytes
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        compare(batch, i, d)
    st = batch.stats()
    assert st["ops_loaded"] == 24 * 1000 and st["delete_ops"] > 0


def test_generated_text_documents(emu):
    blob, offs = emu.gen_text_docs(0xB070, 3, 4, 100, 400, 64, 4)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(12)]
    for d in docs:
        assert o.Doc.load(d).save(deflate=False) == d
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        compare(batch, i, d)


def test_empty_and_ragged_batch(emu):
    blob, offs = emu.gen_map_docs(7, 3, 2, 1, 3, 4, 1)
    docs = [b"", blob[offs[0]:offs[1]], b"", blob[offs[1]:offs[2]][:-1], blob[offs[2]:offs[3]]]
    batch = emu.load_batch(docs)
    assert batch.status(0)[0] == 0 and batch.get_changes(0) == [] and batch.get_heads(0) == []
    compare(batch, 1, docs[1])
    assert batch.status(3)[0] != 0
    compare(batch, 4, docs[4])
    assert len(emu.load_batch([])) == 0


def test_deep_dependency_chains(emu):
    """more dependency levels than the level-parallel hashing handles (AMH_LEVEL_CAP): sequential per-document hashing"""
    blob, offs = emu.gen_text_docs(0xD070, 2, 2, 20, 600, 10, 2)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(4)]
    blob2, offs2 = emu.gen_map_docs(11, 3, 3, 4, 20, 16, 2)   # shallow documents in the same batch
    docs += [blob2[offs2[i]:offs2[i + 1]] for i in range(3)]
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        compare(batch, i, d)
    assert len(batch.get_changes(0)) == 61


def test_sync_summaries(emu, golden_dir):
    """heads + Bloom filter per document == get_heads / make_bloom_filter(last_sync = []) of the oracle; a failed
    document gets a record with its status and no heads"""
    import oracle_lib as o
    docs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))[:20]]
    blob, offs = emu.gen_map_docs(21, 6, 4, 5, 10, 16, 2)
    docs += [blob[offs[i]:offs[i + 1]] for i in range(6)] + [b"", b"\x85\x6f\x4a\x83garbage"]
    batch = emu.load_batch(docs)
    recs = batch.sync_summaries()
    assert len(recs) == len(docs)
    for i, (d, rec) in enumerate(zip(docs, recs)):
        assert rec["doc"] == i
        if i == len(docs) - 1:
            assert rec["status"] != 0 and rec["heads"] == [] and rec["bloom"] == b""
            continue
        ref = o.Doc.load(d)
        assert rec["status"] == 0
        assert rec["heads"] == ref.get_heads() == batch.get_heads(i)
        assert rec["bloom"] == ref.bloom()


def test_save(emu, golden_dir):
    """save_nocompress(load(x)) is byte-identical to the oracle's; save() = same chunk with columns >= 256 bytes
    DEFLATEd (compressed bytes are zlib's: parity unpinned, so compare after loading them back)"""
    import oracle_lib as o
    docs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))]
    blob, offs = emu.gen_map_docs(31, 5, 4, 5, 50, 64, 2)
    docs += [blob[offs[i]:offs[i + 1]] for i in range(5)]
    blob, offs = emu.gen_text_docs(32, 2, 2, 100, 400, 80, 2)
    docs += [blob[offs[i]:offs[i + 1]] for i in range(4)] + [b"\x85\x6f\x4a\x83garbage"]
    batch = emu.load_batch(docs)
    plain, packed = batch.save_all(deflate=False), batch.save_all(deflate=True)
    assert plain[-1] is None and packed[-1] is None
    smaller = 0
    for i, d in enumerate(docs[:-1]):
        ref = o.Doc.load(d)
        assert plain[i] == ref.save(deflate=False), i
        again = o.Doc.load(packed[i])
        assert again.save(deflate=False) == plain[i]
        assert len(packed[i]) <= len(plain[i])
        smaller += len(packed[i]) < len(plain[i])
    assert smaller > 0
    # and the saved bytes load back through the engine to the same state
    b2 = emu.load_batch([p for p in packed[:-1]])
    for i in range(len(docs) - 1):
        assert b2.hydrate_json(i) == batch.hydrate_json(i) and b2.get_heads(i) == batch.get_heads(i)


def test_merge_map_documents(emu):
    """concurrent map edits on both sides: conflicts, overwrites of common ops (new succ entries), counters, deletes,
    new keys, actors only one side knows"""
    from merge_cases import check_merge, split_map_doc
    blob, offs = emu.gen_map_docs(41, 6, 4, 5, 30, 24, 2)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(6)]
    pairs_docs = [split_map_doc(d) for d in docs]
    dst = [a for a, _ in pairs_docs] + [docs[0]]          # the last document is not part of any pair
    src = [b for _, b in pairs_docs]
    check_merge(emu, dst, src, [(i, i) for i in range(6)])
    # merging the other way round gives the same document state
    check_merge(emu, src, [a for a, _ in pairs_docs], [(i, 5 - i) for i in range(6) if i == 5 - i] or [(0, 0)])


def test_merge_text_replicas(emu):
    """eight divergent replicas of text documents merged one after the other (RGA interleaving, deletes of
    common elements, runs typed at the same position)"""
    from merge_cases import check_merge
    n_docs, n_rep = 3, 5
    blob, offs = emu.gen_text_docs(0xE070, n_docs, n_rep, 60, 240, 40, 2)
    reps = [[blob[offs[i * n_rep + r]:offs[i * n_rep + r + 1]] for i in range(n_docs)] for r in range(n_rep)]
    cur = reps[0]
    for r in range(1, n_rep):
        bd = check_merge(emu, cur, reps[r], [(i, i) for i in range(n_docs)])
        cur = bd.save_all(deflate=False)


@pytest.mark.parametrize("history", ["adopted", "re-derived"])
def test_merge_many_replicas_in_one_pass(emu, monkeypatch, history):
    """history = adopted: the merged documents keep their sources' change chunks and hashes (the default when every
    source is a loaded batch); re-derived (AMH_VERIFY_MERGE=1): every change is re-encoded from the MERGED rows, hashed
    and compared with the hash it came with, so a row the merge put in the wrong place cannot go unnoticed.  Both ways
    the saved bytes are compared with the oracle's."""
    if history == "re-derived":
        monkeypatch.setenv("AMH_VERIFY_MERGE", "1")
    from merge_cases import check_merge_many, split_map_doc
    n_docs, n_rep = 4, 6
    blob, offs = emu.gen_text_docs(0xE171, n_docs, n_rep, 60, 240, 40, 2)
    check_merge_many(emu, [[bytes(blob[offs[i * n_rep + r]:offs[i * n_rep + r + 1]]) for i in range(n_docs)] for r in range(n_rep)])
    blob, offs = emu.gen_map_docs(47, 3, 4, 5, 30, 24, 2)
    halves = [split_map_doc(bytes(blob[offs[i]:offs[i + 1]])) for i in range(3)]
    check_merge_many(emu, [[a for a, _ in halves], [b for _, b in halves], [bytes(blob[offs[i]:offs[i + 1]]) for i in range(3)]])


def test_merge_many_stops_at_a_duplicate_seq_like_sequential_merges(emu):
    from merge_cases import check_merge_many_duplicate_seq
    check_merge_many_duplicate_seq(emu)


def test_merge_into_documents_with_queued_changes(emu):
    from merge_cases import check_merge_into_queued
    blob, offs = emu.gen_text_docs(0xA11CE, 3, 2, 60, 300, 50, 1)
    docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(6)]
    check_merge_into_queued(emu, [(docs[0], docs[1]), (docs[2], docs[3]), (docs[4], docs[5])])


def test_merge_nothing_new_and_failures(emu):
    from merge_cases import split_map_doc
    blob, offs = emu.gen_map_docs(43, 2, 4, 5, 10, 8, 1)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(2)]
    a, b = split_map_doc(docs[0])
    bd, bs = emu.load_batch([docs[0], a, a]), emu.load_batch([a, b"\x85\x6f\x4a\x83garbage", b])
    before = bd.save_all(deflate=False)
    assert bd.merge(bs, [(0, 0), (1, 1)]) == -7          # (0,0): src is an ancestor, nothing to do; (1,1): src failed to load
    assert bd.save_all(deflate=False) == before
    assert bd.merge(bs, [(2, 2)]) == 0
    import oracle_lib as o
    ref = o.Doc.load(a); ref.merge(o.Doc.load(b))
    assert bd.save_all(deflate=False)[2] == ref.save(deflate=False)


def test_apply_changes_map_documents(emu):
    blob, offs = emu.gen_map_docs(7, 6, 3, 3, 12, 16, 1)
    docs = [blob[offs[i]:offs[i + 1]] for i in range(6)]
    ac.check_apply_tail(emu, docs, lambda n: n // 2)
    ac.check_apply_tail(emu, docs, lambda n: 0)        # onto empty documents
    ac.check_apply_tail(emu, docs, lambda n: n - 1)


def test_apply_changes_text_replicas(emu):
    blob, offs = emu.gen_text_docs(0xC0DE, 3, 4, 80, 400, 50, 1)
    docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(12)]
    ac.check_apply_tail(emu, docs[:4], lambda n: 1)
    ac.check_apply_replicas(emu, docs, 4)
    ac.check_queue_across_calls(emu, docs[0], docs[1])
    ac.check_duplicate_seq_applies_prefix(emu)
    ac.check_pred_naming_no_op(emu)
    ac.check_incremental_with_documents(emu, docs[4:8])
    ac.check_save_after_roundtrip(emu, docs[8:12])


def test_load_with_change_chunks_and_incremental(emu):
    blob, offs = emu.gen_map_docs(21, 2, 3, 3, 40, 24, 1)
    for i in range(2):
        d = bytes(blob[offs[i]:offs[i + 1]])
        ac.check_trailing_and_policies(emu, d)
        ac.check_incremental_and_errors(emu, d)
    blob, offs = emu.gen_text_docs(0xC1, 1, 2, 50, 300, 60, 1)
    ac.check_trailing_and_policies(emu, bytes(blob[offs[1]:offs[2]]))
    ac.check_incremental_and_errors(emu, bytes(blob[offs[1]:offs[2]]))


def test_edit_trace(emu, golden_dir):
    import edit_trace_case as et
    et.check_engine(emu, golden_dir)


def test_many_documents_take_the_threaded_host_path(emu):
    """the host bookkeeping of apply / merge_many runs on several threads from 128 documents up"""
    from merge_cases import check_merge_many
    n = 150
    blob, offs = emu.gen_map_docs(77, n, 2, 2, 6, 8, 2)
    docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(n)]
    ac.check_apply_tail(emu, docs, lambda k: k // 2)
    blob, offs = emu.gen_text_docs(0x7E57, n, 3, 10, 30, 10, 2)
    check_merge_many(emu, [[bytes(blob[offs[i * 3 + r]:offs[i * 3 + r + 1]]) for i in range(n)] for r in range(3)])


def test_hash_id_tables(emu, golden_dir, monkeypatch):
    """documents whose op ids are sparse use an open-addressing id table instead of the direct one: force that form"""
    monkeypatch.setenv("AMH_HT_HASH", "1")
    files = sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))[:20]
    docs = [open(f, "rb").read() for f in files]
    blob, offs = emu.gen_text_docs(0xE070, 2, 3, 60, 240, 40, 2)
    docs += [bytes(blob[offs[i]:offs[i + 1]]) for i in range(6)]
    batch = emu.load_batch(docs)
    for i, d in enumerate(docs):
        compare(batch, i, d)
    from merge_cases import check_merge_many
    check_merge_many(emu, [[docs[20 + 3 * i + r] for i in range(2)] for r in range(3)])


def test_checksum_outranks_reconstruction_errors(emu):
    blob, offs = emu.gen_map_docs(0xE770, 2, 3, 3, 20, 16, 1)
    for i in range(2):
        ac.check_error_precedence(emu, blob[offs[i]:offs[i + 1]])


def test_repeated_save_and_summaries_do_not_grow_the_batch(emu):
    """amh_save_batch / amh_sync_summaries take their scratch from the pool per call and give it back (a long-lived
    batch that is saved or summarised in a loop must not grow)"""
    blob, offs = emu.gen_map_docs(5, 8, 3, 3, 10, 16, 1)
    b = emu.load_batch([blob[offs[i]:offs[i + 1]] for i in range(8)])
    b.save_all(deflate=False); b.sync_summaries_raw()
    live0 = emu.pool_bytes()[0]
    for _ in range(50):
        b.save_all(deflate=False)
        b.sync_summaries_raw()
    assert emu.pool_bytes()[0] == live0


@pytest.mark.parametrize("name,args", [("overwritten_key", (300,)), ("new_map_keys", (300,)), ("head_siblings", (3, 50)), ("typed_run", (300, 20)),
                                       ("siblings_behind_one_element", (3, 50)), ("counter_increments", (200,)), ("wide_overwrite", (120, 3)), ("deep_nesting", (600,)), ("keystrokes", (300,))])
def test_adversarial_shapes_small(emu, name, args):
    """the shapes of tests/test_gpu_adversarial.py at sizes the thread-by-thread emulation runs quickly"""
    import adversarial_cases as adv
    adv.check_case(emu, adv.CASES[name](*args))


def test_text_of_every_document(emu, golden_dir):
    """amh_text_batch == ReadDoc::text: text replicas with deletes and concurrent inserts, merged documents, documents with
    marks and non-string elements, map documents (no text), a failed document"""
    import mark_cases
    import text_cases
    blob, offs = emu.gen_text_docs(0x7E87, 3, 3, 80, 400, 50, 2)
    docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(9)]
    chunks = mark_cases.marked_text()
    import oracle_lib as orc
    ref = orc.Doc(); ref.apply_changes(chunks)
    docs.append(ref.save(deflate=False))
    blob, offs = emu.gen_map_docs(5, 2, 3, 3, 20, 16, 1)
    docs += [bytes(blob[offs[i]:offs[i + 1]]) for i in range(2)] + [b"\x85\x6f\x4a\x83junk"]
    docs += [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))[:12]]
    text_cases.check_texts(emu, docs)
    # merged replicas
    bs = [emu.load_batch([docs[r], docs[3 + r]]) for r in range(3)]
    assert bs[0].merge_many(bs[1:], [0, 1], [[0, 1]] * 2) == 0
    merged = bs[0].save_all(deflate=False)
    assert bs[0].text_all() == [text_cases.oracle_texts(m) for m in merged]


def test_mark_ops(emu):
    """mark ops through every route of the host build against the oracle (tests/mark_cases.py)"""
    import mark_cases
    mark_cases.check_marks(emu)


def test_hostile_documents_fail_alone(emu):
    import adversarial_cases as adv
    blob, offs = emu.gen_map_docs(77, 5, 3, 3, 20, 16, 1)
    good = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(5)]
    blob, offs = emu.gen_text_docs(78, 1, 1, 50, 2000, 100, 1)
    good.append(bytes(blob[offs[0]:offs[1]]))
    adv.check_hostile_documents(emu, good, 100000, 50000)


def test_partition_docs_balances_bytes(emu):
    """amh_partition_docs: contiguous ranges, every document in exactly one, byte load within one document of even"""
    import random
    rng = random.Random(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 1000):
            lens = [rng.randint(1, 5000) if rng.random() < 0.9 else rng.randint(50000, 200000) for _ in range(n)]
            first = emu.partition_docs(lens, world)
            assert first[0] == 0 and first[-1] == n and all(first[k] <= first[k + 1] for k in range(world))
            if n:
                loads = [sum(lens[first[k]:first[k + 1]]) for k in range(world)]
                assert max(loads) <= sum(lens) / world + max(lens)


def test_sync_conversations(emu):
    """amh_generate_sync_messages beside the oracle's generate_sync_message through whole conversations: the reference's
    own sync scenarios (sync.rs:662-987) restated over explicit change graphs"""
    import sync_cases as sc
    sc.check_basic_rules(emu)
    sc.check_simultaneous_messages(emu)
    sc.check_false_positive_head(emu)
    sc.check_chains_of_false_positives(emu)
    sc.check_branching_and_merging(emu)
    sc.check_in_flight(emu)
    sc.check_reset_and_queue(emu)


def test_sync_messages_over_a_long_history(emu):
    import sync_cases as sc
    sc.check_long_history(emu, 1500)


def test_sync_messages_for_golden_documents(emu, golden_dir):
    import sync_cases as sc
    sc.check_golden_documents(emu, golden_dir)


@pytest.mark.parametrize("seed", range(8))
def test_sync_messages_random_states(emu, seed):
    """random States over random change graphs: every branch of generate_sync_message against the oracle, byte for byte"""
    import sync_cases as sc
    kinds = sc.check_random_states(emu, 0x5C00 + seed)
    assert sum(kinds.values()) > 0


def test_byte_budget_sheds_the_largest_document(emu):
    import adversarial_cases as adv
    blob, offs = emu.gen_map_docs(91, 4, 3, 3, 20, 16, 1)
    docs = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(4)]
    blob, offs = emu.gen_text_docs(92, 1, 1, 50, 3000, 100, 1)
    docs.insert(2, bytes(blob[offs[0]:offs[1]]))
    adv.check_byte_budget(emu, docs)


def test_merge_many_with_numpy_indices(emu):
    from merge_cases import check_merge_many_numpy
    n_docs, n_rep = 3, 4
    blob, offs = emu.gen_text_docs(0xE172, n_docs, n_rep, 40, 160, 30, 2)
    reps = [[bytes(blob[offs[i * n_rep + r]:offs[i * n_rep + r + 1]]) for i in range(n_docs)] for r in range(n_rep)]
    check_merge_many_numpy(emu, reps)
    from merge_cases import check_merge_of_a_merged_batch
    check_merge_of_a_merged_batch(emu, reps)


def test_wide_merges_from_two_threads(emu):
    """Enough documents per call for the host passes to run on the library's worker pool (>= 128 pairs), and two
    contexts driven from two threads at once: the second caller finds the pool busy and runs its loops on threads of
    its own.  Every merged document is compared with the oracle's."""
    import threading
    from merge_cases import check_merge_many, split_map_doc
    blob, offs = emu.gen_map_docs(0x9001, 160, 4, 5, 24, 16, 2)
    halves = [split_map_doc(bytes(blob[offs[i]:offs[i + 1]])) for i in range(160)]
    reps = [[a for a, _ in halves], [b for _, b in halves], [bytes(blob[offs[i]:offs[i + 1]]) for i in range(160)]]
    errors = []

    def run(engine):
        try:
            for _ in range(2):
                check_merge_many(engine, reps)
        except BaseException as e:   # reported by the main thread
            errors.append(e)

    other = emu_engine()
    threads = [threading.Thread(target=run, args=(e,)) for e in (emu, other)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
