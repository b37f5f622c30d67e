import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
from automerge_amd import Engine
eng = Engine(gpu=0)
blob, offs = eng.gen_text_docs(0xB070, 10000, 1, 1000, 5000, 500, 16)
b = eng.load_batch([blob[offs[i]:offs[i + 1]] for i in range(10000)])
for k in range(3):
    t = time.perf_counter(); raw, tot = b.text_all(decode=False); print("text_all", (time.perf_counter() - t) * 1e3, "ms", tot)
