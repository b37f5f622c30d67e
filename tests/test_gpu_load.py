"""GPU parity tests: the HIP load path (through the C ABI of libautomerge_hip.so) against the CPU
oracle and the committed golden fixtures.  Bit-exact bar: change hashes, re-derived change bytes,
heads, max_op and the canonical materialized state must be identical."""
import glob
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from automerge_amd import Engine
    return Engine(gpu=0)


def _compare(batch, i, data):
    import oracle_lib as o
    ref = o.Doc.load(data)
    batch.check(i)
    assert batch.hydrate_json(i) == ref.hydrate_json()
    mine, theirs = batch.get_changes(i), ref.changes()
    assert [c["hash"] for c in mine] == [c["hash"] for c in theirs]
    assert [c["raw"] for c in mine] == [c["raw"] for c in theirs]
    assert [(c["seq"], c["start_op"], c["len"]) for c in mine] == [(c["seq"], c["start_op"], c["len"]) for c in theirs]
    assert batch.get_heads(i) == ref.get_heads()
    assert batch.max_op(i) == ref.max_op()


def test_golden_documents_bit_exact(engine, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "docs", "*.automerge")))
    assert len(files) == 60
    docs = [open(f, "rb").read() for f in files]
    batch = engine.load_batch(docs)
    for i, d in enumerate(docs):
        _compare(batch, i, d)
