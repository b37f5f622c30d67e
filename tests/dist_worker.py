"""Worker of test_dist_gloo.py: rank r loads its shard of a generated batch with the host emulation of
the device pipeline, all-gathers the sync summaries over gloo, and every rank checks the gathered
records of ALL documents against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch.distributed as dist

import oracle_lib as o
from automerge_amd import api
from automerge_amd.dist import all_gather_summaries, shard_range
from emu_engine import emu_engine


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = emu_engine()
    n = 11
    blob, offs = eng.gen_map_docs(0x5EED, n, 3, 3, 12, 16, 1)   # every rank generates the same batch, keeps its shard
    docs = [blob[offs[i]:offs[i + 1]] for i in range(n)]
    lo, hi = shard_range(n, rank, world)
    batch = eng.load_batch(docs[lo:hi])
    raw = batch.sync_summaries_raw()
    parts = all_gather_summaries(raw)
    assert len(parts) == world
    seen = 0
    for r, part in enumerate(parts):
        rlo, rhi = shard_range(n, r, world)
        recs = api.parse_sync_summaries(part)
        assert len(recs) == rhi - rlo
        for k, rec in enumerate(recs):
            ref = o.Doc.load(docs[rlo + k])
            assert rec["doc"] == k and rec["status"] == 0
            assert rec["heads"] == ref.get_heads()
            assert rec["bloom"] == ref.bloom()
            seen += 1
    assert seen == n
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {seen} summaries verified")


if __name__ == "__main__":
    main()
