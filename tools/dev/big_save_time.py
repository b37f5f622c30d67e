"""Time of saving one large text document: usage big_save_time.py OPS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from automerge_amd import Engine
N = int(sys.argv[1]); R = 16
e = Engine(gpu=0)
blob, offs = e.gen_text_docs(0xB16D0C, 1, R, 1000, N // R, max(1, N // R // 60), 16)
docs = [blob[offs[i]:offs[i + 1]] for i in range(R)]
bs = [e.load_batch([d]) for d in docs]
assert bs[0].merge_many(bs[1:], [0], [[0]] * (R - 1)) == 0
b = bs[0]
for rep in range(3):
    t = time.time(); out = b.save_all(deflate=False); dt = time.time() - t
    print("save %d bytes in %.1f ms" % (len(out[0]), dt * 1e3))
os.environ["AMH_BIGSAVE_ROWS"] = "4000000000"
t = time.time(); ref = b.save_all(deflate=False); dt = time.time() - t
print("one wave per column group: %.1f ms, same bytes: %s" % (dt * 1e3, ref[0] == out[0]))
b2 = e.load_batch([out[0]])
print("reload ok:", b2.status(0)[0] == 0, b2.get_heads(0) == b.get_heads(0))
