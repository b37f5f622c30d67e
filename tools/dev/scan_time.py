"""Time of the exclusive scan (amh_scan_inplace) on its own: n items, both forms (AMH_SCAN_ONE_PASS_MIN chooses).
usage: scan_time.py N [N ...]"""
import ctypes, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(R, "automerge_amd", "csrc", "libautomerge_hip.so"))
scan = lib._Z16amh_scan_inplacePjjS_Pv; scan.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]; scan.restype = None
words = lib._Z18amh_scan_tmp_wordsj; words.argtypes = [ctypes.c_uint32]; words.restype = ctypes.c_uint32
for n in map(int, sys.argv[1:]):
    src = torch.randint(0, 5, (n + 4,), dtype=torch.int32, device="cuda")
    data = src.clone()
    tmp = torch.zeros(words(n) + 64, dtype=torch.int32, device="cuda")
    scan(data.data_ptr(), n, tmp.data_ptr(), None)
    torch.cuda.synchronize()
    want = torch.cumsum(src[:n].to(torch.int64), 0)
    ok = bool((data[1:n + 1].to(torch.int64) == want).all()) and int(data[0]) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps): scan(data.data_ptr(), n, tmp.data_ptr(), None)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("n = %d: %.3f ms, %.2f TB/s of 8 B / item; matches cumsum: %s (AMH_SCAN_ONE_PASS_MIN=%s)" % (n, ms, 8 * n / ms / 1e9, ok, os.environ.get("AMH_SCAN_ONE_PASS_MIN")), flush=True)
