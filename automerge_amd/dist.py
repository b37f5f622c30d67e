"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  The load path shards by document and needs no collective; the only
exchange step is the all-gather of per-document sync summaries."""
import torch
import torch.distributed as dist


def shard_range(n_docs, rank, world):
    """Contiguous document range of a rank (documents are independent: no cross-rank traffic on load)."""
    per = (n_docs + world - 1) // world
    lo = min(n_docs, rank * per)
    return lo, min(n_docs, lo + per)


def init_engine_comm(engine):
    """Give the engine's context its RCCL communicator: rank 0 makes the id (amh_comm_unique_id), torch.distributed
    carries the 128 bytes to the other ranks, every rank calls amh_comm_init.  After this Batch.sync_summaries_all()
    gathers inside the library, on device buffers."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    engine.comm_init(box[0], rank, world)


def shard_by_bytes(engine, lens, rank, world):
    """Document range of a rank with the encoded bytes balanced across ranks (amh_partition_docs)."""
    first = engine.partition_docs(list(lens), world)
    return first[rank], first[rank + 1]


def all_gather_summaries(raw: bytes, device=None):
    """all-gather(v) of each rank's sync-summary buffer (Batch.sync_summaries_raw()) through torch.distributed: one
    all_gather of the sizes, then one all_gather of the buffers padded to the largest.  Returns the list of per-rank
    buffers.  This is the host-side route (the gloo tests, callers without a communicator); on GPUs use
    init_engine_comm + Batch.sync_summaries_all(), which gathers inside the library without the D2H / H2D bounce."""
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    size = torch.tensor([len(raw)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if raw:
        buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    gathered = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, buf)
    return [bytes(g[:n].cpu().numpy().tobytes()) for g, n in zip(gathered, sizes)]
