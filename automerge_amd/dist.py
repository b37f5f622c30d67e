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


def all_gather_summaries(raw: bytes, device=None):
    """all-gather(v) of each rank's sync-summary buffer (Batch.sync_summaries_raw()): one all_gather of the
    sizes, then one all_gather of the buffers padded to the largest.  Returns the list of per-rank buffers."""
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
