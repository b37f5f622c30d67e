"""A text typed one character per change (n changes of one actor): document load, kernel by kernel.  usage: keystroke_time.py N"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import adversarial_cases as adv
import oracle_lib as o
from automerge_amd import Engine
n = int(sys.argv[1])
t = time.time(); chunks = adv.keystrokes(n); ref = o.Doc(); ref.apply_changes(chunks); doc = ref.save(deflate=False)
print("built %d changes, %d bytes in %.1f s" % (n, len(doc), time.time() - t), flush=True)
e = Engine(gpu=0)
b = e.load_batch([doc]); b.close()
e.set_profiling(True); e.reset_profile()
t = time.time(); b = e.load_batch([doc]); dt = time.time() - t
print("document load %.1f ms, status %s" % (dt * 1e3, b.status(0)[0]))
for k, (ms, c) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:8]:
    print("  %-18s %9.2f ms  x%d" % (k, ms, c))
e.reset_profile()
t = time.time(); out = b.save_all(deflate=False); dt = time.time() - t
print("save %.1f ms, same bytes %s" % (dt * 1e3, out[0] == doc))
b.close()
e.reset_profile()
t = time.time(); b = e.load_batch([b"".join(chunks)]); dt = time.time() - t
print("change-only load %.1f ms, status %s" % (dt * 1e3, b.status(0)[0]))
for k, (ms, c) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:8]:
    print("  %-18s %9.2f ms  x%d" % (k, ms, c))
