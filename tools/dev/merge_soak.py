"""Randomised merge_many / apply parity soak on the GPU against the oracle: text documents of varied replica counts, sizes
and change sizes (tests/merge_cases.check_merge_many).  usage: merge_soak.py [rounds]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import merge_cases as mc
from automerge_amd import Engine
e = Engine(gpu=0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(0x50A4)
t0 = time.time()
for k in range(rounds):
    nrep = rng.choice([2, 3, 4, 8]); ndoc = rng.choice([1, 2, 5])
    base = rng.choice([1, 10, 100, 400]); ops = rng.choice([30, 200, 900, 2500]); per = rng.choice([1, 3, 40, 500])
    if per == 1: ops = min(ops, 200)
    seed = rng.randrange(1 << 30)
    blob, offs = e.gen_text_docs(seed, ndoc, nrep, base, ops, per, 2)
    reps = [[bytes(blob[offs[i * nrep + r]:offs[i * nrep + r + 1]]) for i in range(ndoc)] for r in range(nrep)]
    mc.check_merge_many(e, reps)
    print("round %2d ok: %d replicas x %d documents, base %d, %d ops in changes of %d (seed %d)  [%.0f s]" % (k, nrep, ndoc, base, ops, per, seed, time.time() - t0), flush=True)
print("soak ok")
