"""Randomised parity soak of amh_generate_sync_messages on the GPU against the oracle: random States over random change
graphs (tests/sync_cases.check_random_states).  usage: sync_soak.py [seeds]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import sync_cases as sc
from automerge_amd import Engine
e = Engine(gpu=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tot, t0 = {}, time.time()
for s in range(n):
    for k, v in sc.check_random_states(e, 0x77000 + s).items():
        tot[k] = tot.get(k, 0) + v
    if s % 20 == 19:
        print("seed %d ok  %s  [%.0f s]" % (s, tot, time.time() - t0), flush=True)
print("soak ok", tot)
