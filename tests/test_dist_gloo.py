"""N > 1 path on CPU: world_size-2 gloo run of the document-sharded load + sync-summary all-gather."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_gloo():
    from emu_engine import emu_engine
    emu_engine()   # build the emulation library once, before the ranks start
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    assert all("summaries verified" in out for out in outs)
