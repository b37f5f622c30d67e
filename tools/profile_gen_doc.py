"""Per-kernel time of loading ONE generated text document of N ops (configs[3] shape): usage profile_gen_doc.py N [ops_per_change] [replicas]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automerge_amd import Engine
N = int(sys.argv[1]); per = int(sys.argv[2]) if len(sys.argv) > 2 else 10000; R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
e = Engine(gpu=0)
t = time.time()
blob, offs = e.gen_text_docs(0xB16D0C, 1, R, 1000, N // R, per, 16)
docs = [blob[offs[i]:offs[i + 1]] for i in range(R)]
print("gen %.1f s, %d bytes" % (time.time() - t, len(blob)), flush=True)
st = e.stage(docs[:1])
e.load_staged(st)
e.set_profiling(True); e.reset_profile()
t = time.time()
b = e.load_staged(st)
dt = time.time() - t
print("load %.1f ms, status %s, ops %d" % (dt * 1e3, b.status(0), b.stats()["ops_loaded"]), flush=True)
for k, (ms, n) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:16]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
e.reset_profile()
t = time.time()
out = b.save_all(deflate=False)
print("save %.1f ms same=%s" % ((time.time() - t) * 1e3, out[0] == docs[0]), flush=True)
for k, (ms, n) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:6]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
