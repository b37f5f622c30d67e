"""GPU: one source load of the headline workload (configs[2]: `docs` text documents, one replica each: a 1,000-character base +
5,000 edits), repeated; per-kernel HIP-event times.  Usage: python tools/dev/load_text_time.py [docs] [reps] [top]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from automerge_amd import Engine

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
eng = Engine(gpu=0)
blob, offs = eng.gen_text_docs(0xB070, docs, 1, 1000, 5000, 500, 16)
staged = eng.stage([blob[offs[i]:offs[i + 1]] for i in range(docs)])
eng.load_staged(staged).close()
t0 = time.perf_counter()
for _ in range(reps):
    eng.load_staged(staged).close()
wall = (time.perf_counter() - t0) / reps * 1e3
eng.set_profiling(True); eng.reset_profile()
b = eng.load_staged(staged)
eng.set_profiling(False)
st = b.stats()
assert st["n_ok"] == docs or os.environ.get("AMH_ENC_GMASK"), st
kt = eng.kernel_times()
print(f"docs {docs} ops {st['ops_loaded']} rows {st['op_rows']} wall {wall:.2f} ms/load  kernels {sum(v[0] for v in kt.values()):.2f} ms")
for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:top]:
    if v[1]:
        print(f"  {k:18s} {v[0]:8.3f} ms {v[1]:4d} launches")
