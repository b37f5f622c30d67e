"""GPU: the library's RCCL binding in a process that has NOT imported torch when the communicator is made, and imports it
afterwards (the order that ended in `double free or corruption` at exit in round 2).  The C ABI is driven through ctypes
directly so that automerge_amd.api's own torch-first precaution is out of the picture.  Exit code 0 = clean."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "automerge_amd", "csrc", "libautomerge_hip.so"))
assert "torch" not in sys.modules


class Bytes(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t)]


lib.amh_ctx_create.restype = C.c_void_p
lib.amh_load_batch.restype = C.c_void_p
gpu = (C.c_int * 1)(0)
ctx = C.c_void_p(lib.amh_ctx_create(gpu, 1))
assert ctx.value
uid = (C.c_uint8 * 128)()
assert lib.amh_comm_unique_id(uid) == 0
assert lib.amh_comm_init(ctx, uid, 0, 1) == 0
docs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "docs", "*.automerge")))[:8]]
ptrs = (C.c_char_p * len(docs))(*docs)
lens = (C.c_size_t * len(docs))(*[len(d) for d in docs])
b = C.c_void_p(lib.amh_load_batch(ctx, ptrs, lens, len(docs), None))
assert b.value
out, offs = Bytes(), (C.c_size_t * 2)()
assert lib.amh_sync_summaries_all(ctx, b, C.byref(out), offs) == 0 and out.len > 0
n_rec = out.len
lib.amh_bytes_free(C.byref(out))
import torch  # noqa: E402  (after the communicator exists)
x = torch.ones(4, device="cuda").sum().item()
lib.amh_batch_free(b)
lib.amh_ctx_destroy(ctx)
print(f"comm before torch: {n_rec} summary bytes, torch sum {x}: ok")
