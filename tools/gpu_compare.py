"""Debug aid: run the same batch through the HIP build and the host emulation of the same kernel
bodies (tests/emu) and report the first device array that differs."""
import ctypes as C, glob, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from automerge_amd import api

NAMES = ["status", "fail_line", "cnt", "row_base", "succ_base", "chg_base", "dep_base", "actor_base", "ht_base", "dt_base", "map_base",
         "doc_hash", "doc_max_op", "actor_off", "actor_len", "actor_rank", "actor_by_rank", "a_first_chg", "a_last_chg",
         "c_actor", "c_seq", "c_max_op", "c_time", "c_msg_off", "c_msg_len", "c_ndeps", "c_dep_base", "c_extra_off", "c_extra_len",
         "c_doc", "dep_idx", "r_id_actor", "r_id_ctr", "r_obj_actor", "r_obj_ctr", "r_key_actor", "r_key_ctr", "r_key_off", "r_key_len",
         "r_insert", "r_action", "r_expand", "r_vmeta", "r_vraw_off", "r_mark_off", "r_mark_len", "r_succ_cnt", "r_succ_base", "r_doc",
         "s_actor", "s_ctr", "s_row", "r_inc_cnt", "r_inc_sum", "t_pred_cnt", "c_nops", "c_start_op", "c_op_base", "c_next_same_actor",
         "ops_sorted", "t_pred_base", "c_map_base", "c_n_other", "chg_actor_map", "c_col_len", "c_out_len", "c_body_len", "c_cols_off",
         "c_out_off", "out_arena", "c_hash", "heads_ok", "r_visible", "r_winner"]


def fetch(lib, h, name):
    n = C.c_size_t()
    lib.amh_debug_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = lib.amh_debug_array(h, name.encode(), None, 0, C.byref(n))
    assert rc == 0, (name, rc)
    buf = (C.c_uint8 * max(1, n.value))()
    rc = lib.amh_debug_array(h, name.encode(), buf, n.value, C.byref(n))
    assert rc == 0, (name, rc)
    return np.frombuffer(bytes(buf)[:n.value], dtype=np.uint8)


def main():
    files = sorted(glob.glob("tests/golden/docs/*.automerge"))
    if len(sys.argv) > 1:
        files = files[:int(sys.argv[1])]
    docs = [open(f, "rb").read() for f in files]
    g = api.Engine()
    e = api.Engine(lib=api.bind(C.CDLL("tests/emu/libamh_emu_for_tests.so")))
    bg, be = g.load_batch(docs), e.load_batch(docs)
    for name in NAMES:
        a, b = fetch(g.lib, bg.h, name), fetch(e.lib, be.h, name)
        if a.shape != b.shape:
            print(f"{name}: size differs gpu={a.size} emu={b.size}")
            continue
        if not np.array_equal(a, b):
            idx = np.nonzero(a != b)[0]
            print(f"{name}: {idx.size} bytes differ, first at byte {idx[0]} (gpu={a[idx[0]:idx[0]+16].tobytes().hex()} emu={b[idx[0]:idx[0]+16].tobytes().hex()})")
        else:
            print(f"{name}: same ({a.size} B)")


main()
