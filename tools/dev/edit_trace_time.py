"""configs[0] (the edit trace: one change of ~260k ops): load + save time and the kernels of a load"""
import gzip, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from automerge_amd import Engine
data = gzip.open(os.path.join(R, "tests", "golden", "edit_trace.automerge.gz")).read()
e = Engine(gpu=0)
for rep in range(3):
    t = time.time(); b = e.load_batch([data]); dt = time.time() - t
    t = time.time(); out = b.save_all(deflate=False); ds = time.time() - t
    print("load %.1f ms, save %.1f ms, %d bytes, same %s" % (dt * 1e3, ds * 1e3, len(data), out[0] == data))
    b.close()
e.set_profiling(True); e.reset_profile()
b = e.load_batch([data])
kt = e.kernel_times()
print("kernel total %.1f ms" % sum(v[0] for v in kt.values()))
for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:12]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
