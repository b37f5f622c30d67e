"""Times of the adversarial shapes (tests/adversarial_cases.py) at full size on the GPU: usage adv_times.py [case ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adversarial_cases as adv
from automerge_amd import Engine
eng = Engine(gpu=0)
for name in (sys.argv[1:] or list(adv.CASES)):
    t = time.time()
    chunks = adv.CASES[name]()
    tb = time.time() - t
    times = adv.check_case(eng, chunks)
    print(name, "build %.1fs" % tb, {k: round(v, 3) for k, v in times.items()}, flush=True)
