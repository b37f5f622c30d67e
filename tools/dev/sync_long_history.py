"""One sync record over a long history: a text typed one character per change (n changes of one actor), a peer that holds
nothing / a peer that shares the first half.  Parity with the oracle and the time of the call.  usage: sync_long_history.py N"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import adversarial_cases as adv
import oracle_lib as o
import sync_cases as sc
from automerge_amd import Engine, SyncHave, SyncState
n = int(sys.argv[1])
chunks = adv.keystrokes(n); ref = o.Doc(); ref.apply_changes(chunks); doc = ref.save(deflate=False)
hashes = [c["hash"] for c in ref.changes()]
e = Engine(gpu=0)
b = e.load_batch([doc])
for label, ls in (("peer holds nothing", []), ("peer shares the first half", [hashes[n // 2]])):
    st = SyncState(); st.their_heads, st.their_need, st.their_have = list(ls), [], [SyncHave(list(ls), b"")]
    blob = sc.to_blob(st)
    b.generate_sync_messages([(0, sc.from_blob(blob))])
    e.set_profiling(True); e.reset_profile()
    t = time.time(); (m,) = b.generate_sync_messages([(0, sc.from_blob(blob))]); dt = time.time() - t
    e.set_profiling(False)
    want, _ = ref.generate_sync_message(blob)
    kt = e.kernel_times()
    print("%-28s %d changes: call %.1f ms, %d bytes, same as the oracle: %s; %s" % (label, n, dt * 1e3, len(m), m == want, {k: round(v[0], 2) for k, v in kt.items() if k.startswith("sm_")}), flush=True)
