"""The C-ABI library loads and exports every symbol include/automerge_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(ROOT, "include", "automerge_hip.h")).read()
    declared = set(re.findall(r"\b(amh_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(os.path.join(ROOT, "automerge_amd", "csrc", "libautomerge_hip.so"))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from automerge_amd import api
    assert set(api.EXPORTS) <= declared


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    from automerge_amd import Engine, EngineError
    try:
        Engine(gpu=0)
    except EngineError as e:
        assert "no HIP device" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("Engine() must fail without a GPU: there is no CPU fallback")
