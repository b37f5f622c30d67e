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


CASES = {"wide_overwrite": wide_overwrite, "overwritten_key": overwritten_key, "counter_increments": counter_increments, "new_map_keys": new_map_keys,
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
