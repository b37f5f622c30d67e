"""GPU: shapes that make a per-list step the whole job (tests/adversarial_cases.py) at the sizes where a quadratic or
one-lane path shows: results against the oracle (change-only load, apply, document load, save), and every phase
inside a time budget that an O(n^2) step on one lane would miss by an order of magnitude."""
import pytest

import adversarial_cases as adv

pytestmark = pytest.mark.gpu

BUDGET_S = 2.0


@pytest.fixture(scope="module")
def engine():
    from automerge_amd import Engine
    return Engine(gpu=0)


@pytest.mark.parametrize("name,args", [
    ("overwritten_key", (50000,)),              # one map key overwritten 50k times
    ("new_map_keys", (20000,)),                 # a change-only load of 20k new map keys (one gap)
    ("head_siblings", (8, 1250)),               # 10k siblings inserted at HEAD by different actors
    ("typed_run", (5000, 100)),                 # a 5k-character typed run applied as one change
    ("typed_run", (50000, 100)),                # ... 50k characters: one subtree 50k deep, placed by list ranking (k_ap_rank_jump)
    ("siblings_behind_one_element", (8, 1250)), # 10k children of one element that arrives in the same call
    ("counter_increments", (20000,)),           # one op gaining 20k successors
    ("wide_overwrite", (3000, 4)),              # one op with 3k preds
    ("keystrokes", (20000,)),                   # one change per character: 20k changes of one actor, a dependency chain 20k deep
    ("deep_nesting", (600,)),                   # maps nested 600 deep (hydrate walks with an explicit stack)
])
def test_adversarial_shape(engine, name, args):
    chunks = adv.CASES[name](*args)
    adv.check_case(engine, chunks, budget_s=BUDGET_S)


def test_hostile_documents_fail_alone(engine):
    """documents of ~60 bytes that claim 2^28 - 1 op rows (about 60 GB of derived arrays each) inside a good batch:
    under a per-document limit they fail alone; under the default limit a batch of six of them exceeds the device and
    the byte budget fails the largest documents instead of the batch"""
    blob, offs = engine.gen_map_docs(77, 5, 3, 3, 20, 16, 1)
    good = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(5)]
    blob, offs = engine.gen_text_docs(78, 1, 1, 50, 2000, 100, 1)
    good.append(bytes(blob[offs[0]:offs[1]]))
    adv.check_hostile_documents(engine, good, (1 << 28) - 1, 1 << 22)
    docs = []
    for g in good:
        docs += [adv.hostile_document((1 << 28) - 1), g]
    b = engine.load_batch(docs)       # default limits: 6 x 60 GB does not fit
    st = [b.status(i)[0] for i in range(len(docs))]
    assert all(st[i] == 0 for i in range(1, len(docs), 2)), st
    assert all(st[i] != 0 for i in range(0, len(docs), 2)), st
    assert st.count(90) >= 2, st
