"""Malformed input must never take the pipeline out of bounds ("no panics on malformed input", rust/automerge/tests/
test.rs:1405-1415, restated for the batch paths): tests/emu/fuzz_apply.cpp pushes mutated documents and change chunks
through load / apply (both modes) / save / hydrate / sync summaries in the host build of the kernel bodies under
AddressSanitizer + UndefinedBehaviorSanitizer.  (The device cannot run sanitizers; the kernel bodies are the same code.)"""
import os
import subprocess

import pytest

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzzed_load_and_apply_stay_in_bounds(seed):
    subprocess.check_call(["make", "-s", "-C", _DIR, "fuzz_apply"])
    r = subprocess.run([os.path.join(_DIR, "fuzz_apply"), "700", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_host_loops_are_race_free():
    """tests/emu/pool_tsan.cpp under ThreadSanitizer: wide merge / apply / sync-message calls whose host passes run on
    the library's worker pool, from one thread and from two threads with a context each.  Skipped where the sanitizer
    cannot start (some kernels refuse its memory layout)."""
    subprocess.check_call(["make", "-s", "-C", _DIR, "pool_tsan"])
    r = subprocess.run([os.path.join(_DIR, "pool_tsan")], capture_output=True, text=True, timeout=900)
    if "FATAL: ThreadSanitizer" in r.stderr:
        pytest.skip("ThreadSanitizer cannot start here: " + r.stderr.strip().splitlines()[0])
    assert r.returncode == 0 and "pool_tsan: 0 0 0" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-6000:])
