"""The register writer of the device encoders (FastWriter: bytes collected in two registers, 16-byte stores, long
literal-run headers widened in place) must produce the bytes of the byte-wise writer (ColWriter), which the oracle pins
(tests/test_emu_pipeline.py).  Runs the same code compiled for the host (tests/emu)."""
import ctypes as C
import random

import pytest

from emu_engine import emu_engine


@pytest.fixture(scope="module")
def lib():
    return emu_engine().lib


def _check(lib, vals, kind):
    n = len(vals)
    arr = (C.c_longlong * max(1, n))(*vals)
    cap = 32 * n + 64
    a, b = (C.c_ubyte * cap)(), (C.c_ubyte * cap)()
    ln = C.c_uint()
    rc = lib.amh_emu_test_writers(arr, n, kind, a, b, cap, C.byref(ln))
    assert rc == 0, (rc, kind, n)
    return bytes(a[:ln.value])


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_writers_agree_on_random_sequences(lib, kind):
    rng = random.Random(1234 + kind)
    for trial in range(300):
        n = rng.choice([0, 1, 2, 3, 10, 63, 64, 65, 66, 127, 128, 129, 500, 3000])
        p_null, p_rep = rng.choice([(0, 0), (0.05, 0.1), (0.3, 0.5), (0, 0.02)])
        vals, last = [], 0
        for i in range(n):
            r = rng.random()
            if r < p_null:
                vals.append(-1)
            elif r < p_null + p_rep and vals:
                vals.append(last)
            else:
                last = rng.choice([rng.randrange(8), rng.randrange(1 << 14), rng.randrange(1 << 40)]) if kind != 2 else rng.randrange(1 << 20)
                vals.append(last)
        _check(lib, vals, kind)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_long_literal_runs_widen_the_header(lib, kind):
    """65 / 8193 distinct values in a row: the literal header needs 2 / 3 bytes (sLEB128 of -k)"""
    for n in (64, 65, 100, 8192, 8193, 9000):
        vals = [(i * 7919) % 1000003 for i in range(n)]
        out = _check(lib, vals, kind)
        if kind == 0 and n == 65:
            assert out[:2] == bytes([0xBF, 0x7F])   # sLEB128(-65)
        # a long run followed by a repeat run and another long literal stretch
        _check(lib, vals + [5] * 10 + vals[::-1], kind)
