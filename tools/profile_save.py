"""Per-kernel time of amh_save_batch on the configs[1] batch (save_nocompress form; host deflate excluded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automerge_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
e = Engine(gpu=0)
arena = e.gen_map_arena(0xA070, n, 4, 5, 50, 128, 16)
b = e.load_staged(e.stage_arena(arena))
import ctypes as C
from automerge_amd import api
def save_device_only():
    # amh_save_batch copies every document to the host; kernel times come from the profiling events
    opts = api.SaveOpts(0, 1)
    outs = (api.BytesOut * n)()
    t = time.time()
    rc = e.lib.amh_save_batch(e.ctx, b.h, C.byref(opts), outs)
    dt = time.time() - t
    assert rc == 0
    tot = sum(outs[i].len for i in range(n))
    for i in range(n):
        e.lib.amh_bytes_free(C.byref(outs[i]))
    return dt, tot
save_device_only()
e.set_profiling(True); e.reset_profile()
dt, tot = save_device_only()
st = b.stats()
print("save %d docs, %d ops, %.1f MB: %.1f ms wall (incl. D2H + per-document host copies)" % (n, st["ops_loaded"], tot / 1e6, dt * 1e3))
ks = 0.0
for k, (ms, c) in sorted(e.kernel_times().items(), key=lambda x: -x[1][0])[:6]:
    print("%-14s %8.2f ms x%d" % (k, ms, c)); ks += ms
print("kernels %.1f ms -> %.3g rows/s" % (ks, st["op_rows"] / (ks * 1e-3)))
