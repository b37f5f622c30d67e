"""GPU: the change re-encode in its forms (AMH_ENCODE_FORM = lane | seg) on the same documents: the first (change, column)
whose length or bytes differ.  Usage: python tools/dev/enc_forms_diff.py [case]"""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from automerge_amd import Engine
import adversarial_cases as adv
import oracle_lib as o


def fetch(lib, h, name, dt):
    lib.amh_debug_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t()
    assert lib.amh_debug_array(h, name.encode(), None, 0, C.byref(n)) == 0
    buf = np.empty(max(1, n.value), dtype=np.uint8)
    assert lib.amh_debug_array(h, name.encode(), buf.ctypes.data, n.value, C.byref(n)) == 0
    return buf[:n.value].view(dt)


def main():
    eng = Engine(gpu=0)
    case = sys.argv[1] if len(sys.argv) > 1 else "overwritten_key"
    if case == "text":
        blob, offs = eng.gen_text_docs(0xF00D, 3, 3, 150, 900, 300, 2)
        docs = [blob[offs[i]:offs[i + 1]] for i in range(9)]
    elif case == "golden":
        docs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "docs", "*.automerge")))]
    else:
        chunks = adv.CASES[case]()
        ref = o.Doc(); ref.apply_changes(chunks)
        docs = [ref.save(deflate=False)]
    res = {}
    for form in ("lane", "seg"):
        os.environ["AMH_ENCODE_FORM"] = form
        b = eng.load_batch(docs)
        res[form] = {k: fetch(eng.lib, b.h, k, dt).copy() for k, dt in (("status", np.int32), ("c_col_len", np.uint32), ("c_wide", np.uint32), ("c_out_off", np.uint32), ("c_cols_off", np.uint32), ("out_arena", np.uint8), ("c_nops", np.uint32))}
        print(form, "status", res[form]["status"][:8], "wide", res[form]["c_wide"][:8])
        b.close()
    a, s = res["lane"], res["seg"]
    la, ls = a["c_col_len"].reshape(-1, 14), s["c_col_len"].reshape(-1, 14)
    bad = np.argwhere(la != ls)
    print("changes", len(la), "length mismatches", len(bad))
    for c, k in bad[:10]:
        print("  change", c, "col", k, "nops", a["c_nops"][c], "lane", la[c, k], "seg", ls[c, k])
        for nm, r_, l_ in (("lane", a, la), ("seg", s, ls)):
            off = int(r_["c_out_off"][c]) + int(r_["c_cols_off"][c]) + int(l_[c, :k].sum())
            print("   ", nm, r_["out_arena"][off:off + min(48, int(l_[c, k]))].tolist())
    if len(bad) == 0:
        for c in range(len(la)):
            off = int(a["c_out_off"][c]) + int(a["c_cols_off"][c])
            for k in range(14):
                n = int(la[c, k])
                x, y = a["out_arena"][off:off + n], s["out_arena"][off:off + n]
                if not np.array_equal(x, y):
                    i = int(np.argmax(x != y))
                    print("  bytes differ: change", c, "col", k, "len", n, "at", i, "lane", x[max(0, i - 4):i + 12].tolist(), "seg", y[max(0, i - 4):i + 12].tolist())
                    break
                off += n
    print("arena equal", np.array_equal(a["out_arena"], s["out_arena"]))


main()
