"""amh_sync_summaries_noalloc (the entry point that allocates nothing) against amh_sync_summaries, through the C ABI:
two calls in a row on one GPU, the second with a smaller batch (the static scratch array keeps the first call's
contents), plus the too-small-buffer and over-the-limit behaviour.  Bit-exact."""
import pytest

pytestmark = pytest.mark.gpu


def _docs(eng, seed, n):
    blob, offs = eng.gen_map_docs(seed, n, 3, 3, 20, 32, 2)
    return [blob[offs[i]:offs[i + 1]] for i in range(n)]


def test_noalloc_matches_sync_summaries_twice():
    import torch
    from automerge_amd import Engine
    eng = Engine(gpu=0)
    big = _docs(eng, 0xC0DE, 3000)
    big[7] = big[7][:40]              # a failed document keeps its (status-only) record
    small = _docs(eng, 0xBEEF, 37)
    out = torch.empty(8 << 20, dtype=torch.uint8, device="cuda:0")
    meta = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    for docs in (big, small):         # second call: smaller input over the same scratch
        b = eng.load_batch(docs)
        want = b.sync_summaries_raw()
        out.fill_(0xEE)
        torch.cuda.synchronize()
        b.sync_summaries_noalloc(out.data_ptr(), out.numel(), meta.data_ptr())
        m = meta.cpu().tolist()
        assert m == [len(want), 0]
        assert bytes(out[:len(want)].cpu().numpy().tobytes()) == want
        assert int(out[len(want)].item()) == 0xEE      # nothing written past the records
        b.close()
    # a buffer that is too small: the total and the flag are reported, the records that fit are written whole
    b = eng.load_batch(small)
    want = b.sync_summaries_raw()
    cap = len(want) // 2
    out.fill_(0xEE)
    torch.cuda.synchronize()
    b.sync_summaries_noalloc(out.data_ptr(), cap, meta.data_ptr())
    assert meta.cpu().tolist() == [len(want), 1]
    got = bytes(out[:cap + 1].cpu().numpy().tobytes())
    from automerge_amd.api import parse_sync_summaries
    recs = parse_sync_summaries(want)
    pos = 0
    for r in recs:
        ln = 12 + 32 * len(r["heads"]) + 4 + len(r["bloom"])
        if pos + ln > cap:
            break
        pos += ln
    assert got[:pos] == want[:pos] and all(x == 0xEE for x in got[pos:])
    b.close()
    eng.close()
