"""Per-kernel time of loading one large document (the edit-trace fixture): where a single big document spends its time."""
import gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automerge_amd import Engine
data = gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "edit_trace.automerge.gz")).read()
e = Engine(gpu=0)
e.load_batch([data])
e.set_profiling(True)
e.reset_profile()
t = time.time()
b = e.load_batch([data])
dt = time.time() - t
kt = e.kernel_times()
print("load %.1f ms" % (dt * 1e3))
for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:14]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
e.reset_profile()
t = time.time()
b.save_all(deflate=False)
print("save %.1f ms" % ((time.time() - t) * 1e3))
for k, (ms, n) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:5]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
