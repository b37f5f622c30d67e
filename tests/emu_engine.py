"""CPU-only test tier: drives the SAME host orchestration + kernel bodies as libautomerge_hip.so, but
compiled for the host (tests/emu).  Test infrastructure only — the package never loads this."""
import ctypes as C
import os
import subprocess

from automerge_amd import api

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_SO = os.path.join(_DIR, "libamh_emu_for_tests.so")


def emu_engine():
    subprocess.check_call(["make", "-s", "-C", _DIR, "libamh_emu_for_tests.so"])
    e = api.Engine(lib=api.bind(C.CDLL(_SO)))
    e.is_emulation = True   # "device" addresses are host addresses: tests may read them directly
    return e
