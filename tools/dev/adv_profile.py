"""Per-kernel time of one adversarial shape: usage adv_profile.py case"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adversarial_cases as adv
import oracle_lib as o
from automerge_amd import Engine
eng = Engine(gpu=0)
for name in sys.argv[1:]:
    chunks = adv.CASES[name]()
    ref = o.Doc(); ref.apply_changes(chunks); saved = ref.save(deflate=False)
    for what, data in (("change-only load", b"".join(chunks)), ("document load", saved)):
        eng.load_batch([data]).close()
        eng.set_profiling(True); eng.reset_profile()
        t = time.time(); b = eng.load_batch([data]); dt = time.time() - t
        eng.set_profiling(False)
        kt = eng.kernel_times()
        print(name, what, "%.1f ms; kernels %.1f ms" % (dt * 1e3, sum(v[0] for v in kt.values())))
        for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:6]:
            print("   %-16s %9.2f ms  x%d" % (k, ms, n))
        b.close()
