"""Per-kernel time of one configs[2] step (load 8 replicas + merge 7) at N documents: usage profile_merge.py N"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automerge_amd import Engine
N = int(sys.argv[1]); R = 8
e = Engine(gpu=0)
blob, offs = e.gen_text_docs(0xB070, N, R, 1000, 5000, 500, 16)
reps = [[blob[offs[i * R + r]:offs[i * R + r + 1]] for i in range(N)] for r in range(R)]
staged = [e.stage(reps[r]) for r in range(R)]
ids = list(range(N))
def step(verbose=False):
    t0 = time.time()
    bs = []
    t_rel = 0.0
    for r, s in enumerate(staged):
        bs.append(e.load_staged(s))
        if r:
            tr = time.time()
            bs[-1].release_ops()
            t_rel += time.time() - tr
    t1 = time.time()
    if verbose:
        print("of the loads: release_ops x%d %.1f ms" % (R - 1, t_rel * 1e3))
    assert bs[0].merge_many(bs[1:], ids, [ids] * (R - 1)) == 0
    t2 = time.time()
    if verbose:
        print("load x%d %.1f ms, merge_many %.1f ms" % (R, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    return bs
for b in step(): b.close()
t = time.time(); bs = step(True); print("unprofiled step %.1f ms" % ((time.time() - t) * 1e3))
t = time.time()
for b in bs: b.close()
print("close %.1f ms" % ((time.time() - t) * 1e3))
e.set_profiling(True); e.reset_profile()
t = time.time(); bs = step(); dt = time.time() - t
print("step %.1f ms" % (dt * 1e3))
kt = e.kernel_times()
tot = sum(v[0] for v in kt.values())
print("kernel total %.1f ms" % tot)
for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:25]:
    print("%-16s %9.2f ms  x%d" % (k, ms, n))
