"""Timeline of loading one large text document (AMH_TIMING=1 prints the host phases): usage big_load_time.py OPS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from automerge_amd import Engine
N = int(sys.argv[1]); R = 16
e = Engine(gpu=0)
blob, offs = e.gen_text_docs(0xB16D0C, 1, R, 1000, N // R, max(1, N // R // 60), 16)
docs = [blob[offs[i]:offs[i + 1]] for i in range(R)]
bs = [e.load_batch([d]) for d in docs]
assert bs[0].merge_many(bs[1:], [0], [[0]] * (R - 1)) == 0
doc = bs[0].save_all(deflate=False)[0]
for b in bs: b.close()
st = e.stage([doc])
for rep in range(3):
    print("---- load %d" % rep, flush=True)
    t = time.time(); b = e.load_staged(st); dt = time.time() - t
    print("load_staged %.1f ms" % (dt * 1e3), flush=True)
    b.close()
