#!/usr/bin/env python3
"""bench.py — throughput of the batched document-load hot path on MI355X.

Workload (BASELINE.json configs[1]): `--docs` JSON-map documents of ~1k ops each (4 actors x 5
concurrent rounds x 50 ops, Zipf keys, conflicts, counters, deletes, one nested map), synthetic,
generated on the host in the reference's storage format.  One step = one pass of the device load
pipeline (parse, checksum, column decode, op linking, change re-derivation + SHA-256, heads check,
visibility) over the whole batch, with the encoded documents already resident in HBM.
Multi-GPU: one process per GPU, the batch is sharded by document (weak scaling: every rank loads
`--docs` documents of its own); no data-path collective.

Prints ONE JSON line (rank 0).  `value` = ops loaded per second over all ranks, ops = sum of
Change::len over the loaded documents."""
import argparse
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(st, cols):
    """Compulsory HBM bytes of each kernel for one pass (DESIGN.md 'Kernels'): what it must read and
    write once, independent of the implementation.  R op rows, S succ entries, T ops (rows + deletes),
    C changes, D documents, P dep entries; column bytes from the parsed descriptors."""
    R, S, T, C, D = st["op_rows"], st["succ_entries"], st["ops_loaded"], st["changes"], st["n_docs"]
    Ein, Eout = st["encoded_bytes"], st["change_bytes"]
    return {
        "parse": cols["meta"] + 216 * D,
        "sha_doc": Ein + 32 * D,
        "count": cols["count"] + 16 * D,
        "extract": 16 * D + 40 * D,
        "actors": cols["actors"] + 24 * st["actors"],
        "chg_init": 28 * C,
        "decode_chg": cols["chg"] + 72 * C,
        "decode_ops": cols["ops"] + 44 * R + 8 * S,          # E + 44 + 8s per op row
        "rows": 44 * R + 12 * R + 4 * R,                     # row in, id table entry out, r_doc
        "obj_order": 8 * R,
        "succ": 2 * 8 * S + 12 * R,                          # SURVEY 8d link/visibility
        "bucket": (8 + 4) * T,                               # SURVEY 8d change bucketing
        "chg_finish": 24 * C,
        "place": 8 * T,
        "pred_slot_cnt": 12 * T,
        "pred_slot_base": 12 * T,
        "pred_fill": 12 * S,
        "pred_sort": 16 * S,
        "actor_mark": 12 * T,
        "actor_rank": 8 * C + 4 * st["actor_map"],
        "change_view": 44 * T + 48 * T,                      # (only batches of small changes, and changes cut into ranges, still build it)
        # SURVEY 8d re-encode = 44 + E' per op, charged ONCE for the stage: the one-pass walk gathers every op's fields and
        # writes the column bytes into their slots ("encode_size"); "encode_store" is the copy of those bytes into the chunks
        # (E' read + E' written) plus the second walk of columns that outgrew their slots (none in these workloads).
        # Batches of small changes (configs[1]) keep the two-pass lane form over the change view: charged the same way.
        "encode_size": 44 * T + Eout + 56 * C,
        "chg_layout": 100 * C,
        "encode_store": 2 * Eout,
        "chg_header": 100 * C,
        "chg_hash": Eout + 32 * C + 4 * C * 5,
        "chg_hash_seq": Eout,
        "heads": 64 * C + 40 * D,
        "ht_size": 24 * D,
        "dt_size": 8 * D,
        "succ_link": 8 * S + 4 * S,
        "visible": 12 * R + 2 * R,
        "scan": 8 * (9 * D + 2 * C + T),
        "text_len": 14 * R, "text_fill": 8 * R + 2 * st.get("text_bytes", 0), "text_doc": 8 * D,
    }


def merge_bytes(NU, PU, NN, RD, SD, RM, SM, NCH, CM, TOPS):
    """Compulsory HBM bytes of the merge / apply kernels for one amh_merge_many call: NU ops of the added
    changes (rows of the staged batch U, deletes included), PU their pred entries, NN new rows (ops that are not
    deletes), RD / SD rows and succ entries of dst, RM / SM of the merged documents, NCH staged change chunks, CM changes
    of the merged documents, TOPS new rows that hang off a dst row.  Per unit: what the kernel must read of its inputs and
    write of its outputs once (4-byte fields; a row = 44 B as in SURVEY 8d), lookups in an id table = 8 B."""
    return {
        "ap_parse": 160 * NCH, "ap_count": 2 * NU + 32 * NCH, "ap_chg_meta": 140 * NCH,
        "ap_verify": 44 * NU + 8 * PU + 8 * NU + 4 * PU + 8 * NU,        # row + preds in, u_chg / u_fill + s_row + normalised object out
        "ap_map": 28 * NU + 8 * PU + (24 + 4 + 8 + 16) * NU + 4 * PU,     # fields in; six link fields, succ count, id, remapped fields out
        "ap_new_flag": 9 * NU + 4 * NU,
        "ap_link": 8 * NU + (28 + 8 + 16) * NN + 12 * NN,                 # own fields, id-table lookup, the dst row(s) compared with; gap / anchor / parent out
        "ap_children": 12 * NU + 8 * NU + 8 * (NN - TOPS), "ap_sort_children": 8 * NU + 16 * (NN - TOPS),
        "ap_succ_init": 21 * NU + 12 * NU, "ap_up_jump": 4 * NU + 8 * (NN - TOPS), "ap_succ_build": 24 * NU + 8 * NN,
        "ap_rank_jump": 8 * NU + 16 * (NN - TOPS), "ap_weight": 4 * NU + 12 * TOPS,
        "ap_top_place": 12 * NU + 28 * TOPS, "ap_ties": 12 * NU + 8 * TOPS, "ap_top_k": 12 * NU + 40 * TOPS,
        "ap_emit": 12 * NU + 12 * NN + 8 * NN,
        "ap_pred_cnt": 8 * NU + (8 + 8 + 4 + 4) * PU, "ap_succ_fill_u": 4 * PU, "ap_succ_sort_u": 8 * NU, "ap_succ_fill_d": (4 + 4 + 8 + 8) * PU,
        "mg_new_flag": 13 * NU, "mg_pair_counts": 32 * CM,
        "mg_subtree": (25 + 8) * RD + 12 * RD + 8 * RD,                   # parent lookup; minima; ends
        "mg_rows_dst": (44 + 44 + 4 + 4) * RD, "mg_rows_new": (8 + 44 + 44) * NN,
        "mg_succ_dst": 8 * RD + 16 * SD, "mg_succ_new": 8 * NN + 16 * (SM - SD), "mg_succ_sort": 8 * RD + 8 * (SM - SD), "mg_doc_succ": 16 * CM,
        "mg_changes": 120 * CM, "mg_copy_chunks": 0,
        "rows_m": 44 * RM + 12 * RM + 4 * RM,
    }


# SHA-256 on the vector ALU: per 64-byte block 64 rounds of (Sigma1 5 + Ch 3 + Sigma0 5 + Maj 4 + 7 adds) = 24 and 48
# message-schedule steps of (sigma0 5 + sigma1 5 + 3 adds) = 13 32-bit integer operations (a rotate = one v_alignbit).
SHA_INT_OPS_PER_BLOCK = 64 * 24 + 48 * 13
# 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz, one 32-bit integer operation per lane and cycle (MI355X_MICROARCH.md)
INT32_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def sha_roofline(bytes_hashed, ms):
    """SHA-256 against the integer-op roofline and as GB/s hashed"""
    if ms <= 0:
        return None
    tops = bytes_hashed / 64.0 * SHA_INT_OPS_PER_BLOCK / (ms * 1e-3) / 1e12
    return {"bound": "valu-int32", "GB_hashed_per_s": round(bytes_hashed / (ms * 1e-3) / 1e9, 2), "int_ops_per_64B_block": SHA_INT_OPS_PER_BLOCK,
            "achieved_Tops": round(tops, 3), "peak_Tops": round(INT32_PEAK_TOPS, 1), "frac": round(tops / INT32_PEAK_TOPS, 4)}


def column_bytes(eng, batch, n_docs):
    """Encoded bytes each decode kernel has to read, from the parsed per-document descriptors."""
    import ctypes as C
    import numpy as np
    lib = eng.lib
    lib.amh_debug_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t()
    lib.amh_debug_array(batch.h, b"desc", None, 0, C.byref(n))
    buf = np.empty(n.value, dtype=np.uint8)
    lib.amh_debug_array(batch.h, b"desc", buf.ctypes.data, n.value, C.byref(n))
    dt = np.dtype({"names": ["chunk_off", "hdr", "ccol", "ocol"], "formats": ["<u8", ("<u4", (9,)), ("<u4", (9, 2)), ("<u4", (16, 2))],
                   "offsets": [0, 8, 44, 116], "itemsize": 248})   # struct DocDesc (amh_types.h)
    d = buf.view(dt)
    ccol, ocol = d["ccol"][:, :, 1].astype(np.int64), d["ocol"][:, :, 1].astype(np.int64)
    hdr = d["hdr"].astype(np.int64)  # chunk_len, data_off, data_len, n_actors, actors_off, n_heads, heads_off, suffix_off, suffix_len
    data_start = np.minimum(d["ccol"][:, :, 0].astype(np.int64).max(axis=1), hdr[:, 1] + hdr[:, 2])
    return {
        "meta": int((hdr[:, 1] + hdr[:, 2] - ccol.sum(axis=1) - ocol.sum(axis=1)).sum()),
        "count": int(ocol[:, 6].sum() + ocol[:, 11].sum() + ccol[:, 0].sum() + ccol[:, 5].sum()),
        "actors": int((hdr[:, 6] - hdr[:, 4]).clip(min=0).sum()),
        "chg": int(ccol.sum()),
        "ops": int(ocol.sum()),
    }


# what holds a kernel below the HBM roofline, where it was measured (DESIGN.md section in brackets)
ROOFLINE_NOTES = {
    "encode_size": "scattered 4-byte gathers (op -> row -> ~12 fields, 64 cache lines per wave instruction): 2.63 of 3.72 ms per 6e7 ops with the encode logic removed [4.2b]",
    "decode_ops": "instruction issue of the token / header / expand steps: 7 waves per SIMD, ~0.9 us per 64-value step [5a]",
    "mg_rows_new": "scattered gathers of 15 fields per new row out of change order [5a]",
    "encode_store": "per-lane column state machines of the lane-form encoder (50-op changes) [5a]",
}


def pmc_traffic(workload, dom):
    """HBM bytes per launch of kernel `dom` from the committed rocprofv3 --pmc passes of this same command
    (profiles/traffic_<workload>.json, tools/pmc_traffic.py: 2 x FETCH_SIZE + WRITE_SIZE, separate passes); None when
    the file is missing.  A bench kernel name may cover several device kernels (forms chosen per batch): the launches of
    all of them are averaged together."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", f"traffic_{workload}.json")))
        if tj.get("workload") != workload:
            return None
        forms = {"decode_ops": ["g_decode_seg(BatchDev)", "void g_decode_seg<false>(BatchDev, unsigned int)"], "count": ["g_count_seg(BatchDev)"],
                 "encode_size": ["void g_encode_one<0>(BatchDev, unsigned int, unsigned int)"] + [f"void g_encode_grp<false, {g}>(BatchDev)" for g in range(1, 7)],
                 "encode_store": ["g_encode_place(BatchDev)"] + [f"void g_encode_grp<true, {g}>(BatchDev)" for g in range(1, 7)],
                 "visible": ["g_visible_seg(BatchDev, unsigned int)"],
                 "save_size": ["void g_save_seg<false>(BatchDev)", "void g_save_lds<false>(BatchDev)"],
                 "save_store": ["void g_save_seg<true>(BatchDev)", "void g_save_lds<true>(BatchDev)"]}
        names = forms.get(dom, [f"g_k_{dom}(BatchDev, unsigned int)", f"g_k_{dom}(MergeDev const*, unsigned int)"])
        hit = [tj["kernels"][nm] for nm in names if nm in tj["kernels"]]
        if not hit:
            return None
        # the forms of one bench kernel run side by side in a launch of the pipeline step (e.g. the six column groups
        # of the lane-form encoder): bytes per pipeline launch = sum over forms of their per-launch averages
        if dom in ("encode_size", "encode_store") and any("g_encode_grp" in nm for nm in names if nm in tj["kernels"]) and not any("g_encode_one" in nm or "g_encode_place" in nm for nm in names if nm in tj["kernels"]):
            return sum(h["traffic"] for h in hit)
        tot_l = sum(h["launches"] for h in hit)
        return sum(h["traffic"] * h["launches"] for h in hit) / tot_l
    except Exception:
        return None


def bench_merge(args, eng, rank, world, dist, torch, red_dev):
    """configs[2] (the headline line): `--docs` text documents per GPU (default 10,000), each a 1,000-character base +
    `--replicas` divergent replicas of 5,000 edits (70 % inserts in runs, 30 % deletes).  One step = every document's
    replicas loaded (one batch per replica index) and replicas 1..R-1 merged into replica 0.  All encoded replicas are
    resident in HBM before the timed region; a step streams them through the pipeline `--slice` documents at a time
    (the derived arrays of a slice are released before the next one starts), so the working set stays well inside one
    GPU's HBM while the step still covers all `--docs` documents."""
    import ctypes as C
    R = args.replicas
    if args.sequential_merges or R < 3:
        args.source_batches = 0   # pairwise merges take one batch per replica
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), 32))
    n_slices = max(1, (args.docs + args.slice - 1) // args.slice)
    bounds = [(k * args.docs // n_slices, (k + 1) * args.docs // n_slices) for k in range(n_slices)]
    t0 = time.time()
    staged, enc_bytes, sample, t_stage = [], 0, None, 0.0
    for k, (lo, hi) in enumerate(bounds):   # generated and staged slice by slice: the host never holds more than one slice of inputs
        blob, offs = eng.gen_text_docs(0xB070 + rank * args.docs + lo, hi - lo, R, 1000, 5000, 500, threads)
        reps = [[blob[offs[i * R + r]:offs[i * R + r + 1]] for i in range(hi - lo)] for r in range(R)]
        ts = time.perf_counter()
        if args.source_batches == 1:   # replica 0 of every document, then the R-1 source replicas of all of them as one batch (replica-major)
            staged.append([eng.stage(reps[0]), eng.stage([d for r in range(1, R) for d in reps[r]])])
        else:
            staged.append([eng.stage(reps[r]) for r in range(R)])
        t_stage += time.perf_counter() - ts
        enc_bytes += len(blob)
        if k == 0:
            sample = [[reps[r][i] for r in range(R)] for i in range(min(hi - lo, 256))]
        del blob, reps
    t_gen = time.time() - t0

    text_s, text_total, text_host_ms = [0.0], [0], [0.0]

    def step(keep_stats=False):
        ops = 0
        stats = []
        for k, (lo, hi) in enumerate(bounds):
            ids = list(range(hi - lo))
            ids_np = np.arange(hi - lo, dtype=np.uint32)
            src_np = np.tile(ids_np, (R - 1, 1))
            if args.source_batches == 1:   # source s of document p = document s * n + p of the one wide batch
                src_np = (src_np + (np.arange(R - 1, dtype=np.uint32) * np.uint32(hi - lo))[:, None]).astype(np.uint32)
            batches = []
            for r, s in enumerate(staged[k]):
                batches.append(eng.load_staged(s))   # every replica is loaded and verified in full
                if keep_stats:   # per-batch counts of the source loads for the byte model (before the merge replaces batch 0)
                    stats.append([]) if r == 0 else None
                    stats[-1].append((batches[-1].stats(), column_bytes(eng, batches[-1], batches[-1].n)))
                if r and not args.sequential_merges:
                    batches[-1].release_ops()       # a merge source is read for its history only: its op rows go back to the pool
            if args.sequential_merges:
                for r in range(1, R):
                    rc = batches[0].merge(batches[r], [(i, i) for i in ids])
                    assert rc == 0, rc
            else:   # the same result (history order included) with one pass over the rows
                rc = batches[0].merge_many(batches[1:] if args.source_batches != 1 else [batches[1]] * (R - 1), ids_np, src_np)
                assert rc == 0, rc
            # ... + materialize text: ReadDoc::text of every merged document, left in HBM like the inputs
            # (amh_text_batch_device); the copy to the host is timed once, outside the steps (config.text_to_host_ms)
            tt = time.perf_counter()
            _, _, text_bytes = batches[0].text_device()
            text_s[0] += time.perf_counter() - tt
            text_total[0] = text_bytes
            if keep_stats:
                tt = time.perf_counter()
                batches[0].text_all(decode=False)
                text_host_ms[0] = (time.perf_counter() - tt) * 1e3
            st = batches[0].stats()
            assert st["n_ok"] == hi - lo and text_bytes > 0
            ops += st["ops_loaded"]
            if keep_stats:
                stats[-1].append((st, column_bytes(eng, batches[0], hi - lo)))
            for b in batches:
                b.close()
        return ops, stats

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    text_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ops, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    text_ms = text_s[0] / args.steps * 1e3
    # one more, profiled step: HIP events around every kernel launch on the pipeline's own stream
    eng.set_profiling(True)
    eng.reset_profile()
    _, stats = step(keep_stats=True)
    eng.set_profiling(False)
    kt = eng.kernel_times()
    tot = torch.tensor([float(elapsed), float(ops)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max, ops_all = float(tmax[0]), float(tot[1])
    else:
        elapsed_max, ops_all = elapsed, float(ops)
    if rank == 0:
        # bytes every load-pipeline kernel must move, summed over the document batches it runs over in a step: the R source
        # loads and, for the kernels of the tail that still run over them, the merged documents (the kernels of the change
        # chunks handed to the apply path run under the same names and are in the times but not in the bytes: the
        # fraction is a lower bound)
        model = {}
        # kernels that do not run over the merged documents: their rows come from the merge (nothing is parsed or decoded),
        # and unless AMH_VERIFY_MERGE is set their changes are the sources' chunks (nothing is re-encoded or hashed)
        not_on_merged = {"parse", "sha_doc", "count", "extract", "actors", "chg_init", "decode_chg", "decode_ops", "rows"}
        if not os.environ.get("AMH_VERIFY_MERGE") and not args.sequential_merges:
            not_on_merged |= {"place", "pred_slot_cnt", "pred_slot_base", "pred_fill", "pred_sort", "actor_mark", "actor_rank", "change_view",
                              "encode_size", "chg_layout", "encode_store", "chg_header", "chg_hash"}
        for sl in stats:
            for bi, (st, cols) in enumerate(sl):
                st = dict(st)
                st["actors"] = 2 * st["n_docs"]
                st["actor_map"] = 2 * st["changes"]
                merged = bi == len(sl) - 1
                for kname, v in algorithmic_bytes(st, cols).items():
                    if merged and kname in not_on_merged:
                        continue
                    model[kname] = model.get(kname, 0) + v
        # the merge itself (amh_merge_many): the added changes are staged as a batch U of change chunks (decoded by the same
        # column decoder, hashed by the same SHA kernel), then the merge / apply kernels place their ops among dst's rows
        hashed_bytes = 0
        for sl in stats:
            d0, mg_st = sl[0][0], sl[-1][0]
            NU, NN = mg_st["ops_loaded"] - d0["ops_loaded"], mg_st["op_rows"] - d0["op_rows"]
            PU, CM, NCH = mg_st["succ_entries"] - d0["succ_entries"], mg_st["changes"], mg_st["changes"] - d0["changes"]
            EU = mg_st["change_bytes"] - d0["change_bytes"]
            TOPS = NN // 10   # new rows that hang off a dst row (run starts landing on old text): an estimate, a minor term everywhere
            model["decode_ops"] = model.get("decode_ops", 0) + EU + 44 * NU + 8 * PU
            model["sha_doc"] = model.get("sha_doc", 0) + EU + 32 * NCH
            for kname, v in merge_bytes(NU, PU, NN, d0["op_rows"], d0["succ_entries"], mg_st["op_rows"], mg_st["succ_entries"], NCH, CM, TOPS).items():
                model[kname] = model.get(kname, 0) + v
            model["text_len"] = model.get("text_len", 0) + 14 * mg_st["op_rows"]
            model["text_fill"] = model.get("text_fill", 0) + 8 * mg_st["op_rows"] + 2 * text_total[0] // max(1, len(stats))
            model["text_doc"] = model.get("text_doc", 0) + 8 * mg_st["n_docs"]
            hashed_bytes += sum(x[0]["encoded_bytes"] + x[0]["change_bytes"] for x in sl[:-1]) + EU
        for k in ("ap_up_jump", "ap_rank_jump"):   # one pass of the model per launch (the host repeats them until no pointer moved)
            if k in model and k in kt and kt[k][1] > len(stats):
                model[k] *= kt[k][1] // len(stats)
        # sha_doc runs on the side stream beside the count pass and visible beside the change hashing: their event times
        # include the time they share the device, so they are reported (visible's standalone time: DESIGN 5) but not ranked
        OVERLAPPED = ("sha_doc", "visible")
        ranked = sorted((k for k in kt if kt[k][1] > 0 and k not in OVERLAPPED and k in model), key=lambda k: -kt[k][0])
        dom = ranked[0]
        dom_ms, dom_calls = kt[dom]
        achieved = model[dom] / (dom_ms * 1e-3) / 1e9
        total_kernel_ms = sum(v[0] for k, v in kt.items() if k not in OVERLAPPED)
        kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1],
                       "GBps": round(model[k] / (v[0] * 1e-3) / 1e9, 2) if k in model and v[0] > 0 else None,
                       **({"overlapped": True} if k in OVERLAPPED else {})}
                   for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1] > 0}
        traffic = pmc_traffic("merge", dom)
        sha_ms = sum(kt.get(k, (0, 0))[0] for k in ("chg_hash", "chg_hash_seq", "sha_doc"))
        out = {"metric": "ops loaded+merged/sec (node)", "value": ops_all * args.steps / elapsed_max, "unit": "ops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "synthetic",
               "config": {"workload": f"configs[2]: {args.docs} text documents x {R} divergent replicas of 5k concurrent inserts/deletes per GPU, load {R} + merge {R - 1} + materialize text",
                          "text_ms_per_step": round(text_ms, 2), "text_bytes_per_step": text_total[0], "text_to_host_ms": round(text_host_ms[0], 1),
                          "docs_per_gpu": args.docs, "replicas": R, "ops_per_merged_doc": ops / args.docs, "encoded_bytes_per_gpu": enc_bytes,
                          "slices_per_step": n_slices, "docs_per_slice": args.slice, "sharding": f"documents x{world} (replicas of a document co-located)",
                          "merge_call": "sequential amh_merge_batch" if args.sequential_merges else "amh_merge_many",
                          "source_loads": (f"the {R - 1} source replicas of all documents as ONE batch of {(R - 1) * args.slice} documents (one amh_load_staged call), replica 0 as another"
                                           if args.source_batches == 1 else f"one batch per replica ({R} amh_load_staged calls)"),
                          "gen_s": round(t_gen, 2),
                          "stage_s": round(t_stage, 3),   # host bytes -> HBM through the pinned double buffer (amh_stage_batch), outside the timed region
                          "pcie_inclusive_ops_per_s": ops_all / (elapsed_max / args.steps + t_stage),
                          "note": "ops counted once per merged document; every replica is loaded in full (every change re-derived from the ops, SHA-256 hashed, heads checked); the merged document keeps those verified change chunks, as the reference's merge keeps stored changes (AMH_VERIFY_MERGE=1 re-derives them from the merged rows as well)"},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic, "algorithmic_bytes_per_launch": model[dom] / dom_calls, "launch_ms": dom_ms / dom_calls,
                            "launches_per_step": dom_calls, "kernel_ms_total": total_kernel_ms,
                            "limited_by": ROOFLINE_NOTES.get(dom),
                            "sha256": sha_roofline(hashed_bytes, sha_ms)},
               "kernels": kernels}
        if not args.no_cpu and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o   # CPU oracle: the reported baseline only (never the measured path)
            from concurrent.futures import ThreadPoolExecutor
            cores = max(1, min(threads, 16))

            def one(docs):   # ctypes releases the GIL inside the oracle calls: the threads run in parallel
                d = o.Doc.load(docs[0])
                for r in range(1, R):
                    d.merge(o.Doc.load(docs[r]))
                return d.num_ops()

            t0 = time.perf_counter()
            n_done, n_ops = 0, 0
            with ThreadPoolExecutor(cores) as ex:
                for lo in range(0, len(sample), cores):
                    n_ops += sum(ex.map(one, sample[lo:lo + cores]))
                    n_done += len(sample[lo:lo + cores])
                    if time.perf_counter() - t0 > 20.0:
                        break
            secs = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_ops / secs, "unit": "ops/s", "cores": cores, "kind": "port",
                                   "sample": f"first {n_done} documents of the same batch, load {R} + merge {R - 1}, one document per thread (CPU restatement in "
                                             f"oracle/ with a plain O(n) sequence insert per op: a correctness oracle, not the Rust reference: no toolchain)"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_bigdoc(args, eng, rank, world, dist, torch, red_dev):
    """configs[3]: ONE text document of `--ops` ops (default 10M) in 1,000 changes by `--replicas` actors (default 16):
    the actors' divergent histories are generated as replicas and merged into one document (untimed; the engine's own
    merge, re-verified).  One step = load of that document with its bytes resident in HBM.  Does not shard: with N > 1
    every rank loads a replica of the same document ("replicas only")."""
    R = args.replicas
    per_change = max(1, args.ops // 1000)
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), 32))
    t0 = time.time()
    blob, offs = eng.gen_text_docs(0xB16D0C, 1, R, 1000, args.ops // R, per_change, threads)
    reps = [blob[offs[r]:offs[r + 1]] for r in range(R)]
    if R > 1:
        bs = [eng.load_batch([x]) for x in reps]
        rc = bs[0].merge_many(bs[1:], [0], [[0]] * (R - 1))
        assert rc == 0, rc
        doc = bs[0].save_all(deflate=False)[0]
        for b in bs:
            b.close()
    else:
        doc = reps[0]
    t_gen = time.time() - t0
    staged = eng.stage([doc])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.load_staged(staged).close()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b = eng.load_staged(staged)
        b.close()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(True)
    eng.reset_profile()
    batch = eng.load_staged(staged)
    eng.set_profiling(False)
    kt = eng.kernel_times()
    batch.check(0)
    st = batch.stats()
    resaved = batch.save_all(deflate=False)[0]
    assert resaved == doc, "save(load(x)) != x"
    ops = st["ops_loaded"]
    tot = torch.tensor([float(elapsed), float(ops)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max, ops_all = float(tmax[0]), float(tot[1])
    else:
        elapsed_max, ops_all = elapsed, float(ops)
    if rank == 0:
        st["actors"] = R + 1
        st["actor_map"] = (R + 1) * st["changes"]
        model = algorithmic_bytes(st, column_bytes(eng, batch, 1))
        OVERLAPPED = ("sha_doc", "visible")
        dom = max((k for k in kt if kt[k][1] > 0 and k not in OVERLAPPED and k in model), key=lambda k: kt[k][0])
        dom_ms, dom_calls = kt[dom]
        achieved = model[dom] / (dom_ms * 1e-3) / 1e9
        out = {"metric": "ops loaded+merged/sec (node)", "value": ops_all * args.steps / elapsed_max, "unit": "ops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "synthetic",
               "config": {"workload": f"configs[3]: one text document of {ops} ops, {st['changes']} changes, {R} concurrent actors (+ the base author): load",
                          "op_rows": st["op_rows"], "encoded_bytes": len(doc), "gen_s": round(t_gen, 1), "sharding": "replicas only",
                          "checks": "every change re-derived and hashed, heads verified, save(load(x)) == x",
                          "note": "chunk and change SHA-256 of a document this size run on host threads beside the device pipeline (one message is a serial chain)"},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": None, "algorithmic_bytes_per_launch": model[dom] / dom_calls, "launch_ms": dom_ms / dom_calls,
                            "kernel_ms_total": sum(v[0] for k, v in kt.items() if k not in OVERLAPPED)},
               "kernels": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1] > 0}}
        if not args.no_cpu and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o   # CPU oracle: the reported baseline only (never the measured path)
            sample_ops = min(args.ops, 2000000)   # bounded: the oracle loads ~2e5 ops/s
            sb, so = eng.gen_text_docs(0xB16D0C, 1, 1, 1000, sample_ops, max(1, sample_ops // 1000), threads)
            t0 = time.perf_counter()
            d = o.Doc.load(sb[so[0]:so[1]])
            secs = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": d.num_ops() / secs, "unit": "ops/s", "cores": 1, "kind": "port",
                                   "sample": f"one text document of {d.num_ops()} ops by one actor, same generator (CPU restatement in oracle/, not the Rust reference: no toolchain); "
                                             f"a single document does not parallelise over host threads in the reference either"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_trace(args, eng, rank, world, dist, torch, red_dev):
    """configs[0]: the automerge-perf editing trace (259,778 edits) as the reference's edit-trace driver leaves it: ONE change
    of ~260k ops, saved, then Automerge::load (rust/edit-trace/src/main.rs:20-54).  The document is the committed fixture
    tests/golden/edit_trace.automerge.gz (made from the reference's edits.json by tools/golden/make_edit_trace.py).
    One step = one load (change re-derived over many waves and hashed, heads verified); the save is timed beside it."""
    import gzip
    doc = gzip.open(os.path.join(ROOT, "tests", "golden", "edit_trace.automerge.gz")).read()
    staged = eng.stage([doc])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.load_staged(staged).close()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b = eng.load_staged(staged)
        b.close()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(True)
    eng.reset_profile()
    batch = eng.load_staged(staged)
    eng.set_profiling(False)
    kt = eng.kernel_times()
    batch.check(0)
    st = batch.stats()
    batch.save_all(deflate=False)
    t1 = time.perf_counter()
    resaved = batch.save_all(deflate=False)[0]
    save_ms = (time.perf_counter() - t1) * 1e3
    assert resaved == doc, "save(load(x)) != x"
    ops = st["ops_loaded"]
    tot = torch.tensor([float(elapsed)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.MAX)
    elapsed_max = float(tot[0])
    if rank == 0:
        st["actors"] = 1
        st["actor_map"] = st["changes"]
        model = algorithmic_bytes(st, column_bytes(eng, batch, 1))
        OVERLAPPED = ("sha_doc", "visible")
        dom = max((k for k in kt if kt[k][1] > 0 and k not in OVERLAPPED and k in model), key=lambda k: kt[k][0])
        dom_ms, dom_calls = kt[dom]
        achieved = model[dom] / (dom_ms * 1e-3) / 1e9
        out = {"metric": "ops loaded+merged/sec (node)", "value": world * ops * args.steps / elapsed_max, "unit": "ops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "the reference's edit trace (tests/golden fixture)",
               "config": {"workload": f"configs[0]: the automerge-perf editing trace as one change of {ops} ops: load", "encoded_bytes": len(doc),
                          "save_ms": round(save_ms, 2), "sharding": "replicas only",
                          "checks": "the change re-derived and hashed (hash pinned by the fixture), heads verified, save(load(x)) == x"},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": None, "algorithmic_bytes_per_launch": model[dom] / dom_calls, "launch_ms": dom_ms / dom_calls,
                            "kernel_ms_total": sum(v[0] for k, v in kt.items() if k not in OVERLAPPED)},
               "kernels": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1] > 0}}
        if not args.no_cpu and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o   # CPU oracle: the reported baseline only (never the measured path)
            t0 = time.perf_counter()
            d = o.Doc.load(doc)
            secs = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": d.num_ops() / secs, "unit": "ops/s", "cores": 1, "kind": "port",
                                   "sample": "the same document, one load (CPU restatement in oracle/, not the Rust reference: no toolchain)"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_sync(args, eng, rank, world, dist, torch, red_dev):
    """configs[4]: sync-server sweep.  `--docs` documents per GPU (default 125k = 1M over 8 GPUs) of 8 changes x 20 map puts.
    One step = load the rank's documents (every change re-derived and SHA-256 hashed, heads), heads + Bloom summary per
    document (amh_sync_summaries), and the all-gather of the summary records over the process group (RCCL over xGMI)."""
    from automerge_amd import dist as amd_dist
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")) // max(1, world if world > 1 else 1) or 1, 32))
    t0 = time.time()
    arena = eng.gen_map_arena(0x5A5A + rank * args.docs, args.docs, 2, 4, 20, 64, max(4, threads))
    t_gen = time.time() - t0
    staged = eng.stage_arena(arena)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gathered_bytes = 0

    use_comm = dist is not None and dist.get_backend() == "nccl"   # RCCL inside the library; gloo rehearsals gather through torch
    if use_comm:
        amd_dist.init_engine_comm(eng)

    # the sender side of the sync protocol for every document (amh_generate_sync_messages, sync.rs:134-221): one peer per
    # document that holds nothing of it yet and has said so (their_heads = [], their_need = [], one Have with no
    # last_sync and a Bloom filter over 8 hashes the document does not contain): every change is probed against the
    # filter and goes into the message with the dependents of whatever the filter did not claim (sync.rs:276-303)
    import ctypes as C
    from automerge_amd.api import SyncHaveC, SyncStateC
    import hashlib
    bits = bytearray(10)
    for k in range(8):   # BloomFilter::add_hash (sync/bloom.rs:69-102) of 8 foreign hashes
        h = hashlib.sha256(b"peer %d" % k).digest()
        x, y, z = (int.from_bytes(h[i:i + 4], "little") % 80 for i in (0, 4, 8))
        for i in range(7):
            if i:
                x, y = (x + y) % 80, (y + z) % 80
            bits[x >> 3] |= 1 << (x & 7)
    bloom = bytes([8, 10, 7]) + bytes(bits)
    have = (SyncHaveC * 1)()
    have[0].last_sync32, have[0].n_last_sync, have[0].bloom, have[0].bloom_len = None, 0, bloom, len(bloom)
    states = (SyncStateC * args.docs)()
    for i in range(args.docs):
        states[i].doc, states[i].flags, states[i].their_have, states[i].n_their_have = i, 7, have, 1
    msg_bytes = msg_changes = 0

    def step():
        nonlocal gathered_bytes, msg_bytes, msg_changes
        b = eng.load_staged(staged)
        raw = None if use_comm else b.sync_summaries_raw()
        if dist is not None:
            parts = b.sync_summaries_all() if use_comm else amd_dist.all_gather_summaries(raw)
            gathered_bytes = sum(len(x) for x in parts)
        else:
            gathered_bytes = len(raw)
        _, _, _, msg_bytes, _, _, msg_changes = b.generate_sync_messages_device(states, args.docs)   # the messages stay in HBM
        return b

    for _ in range(args.warmup):
        step().close()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b = step()
        if _ + 1 < args.steps:
            b.close()
    barrier()
    elapsed = time.perf_counter() - t0
    st = b.stats()
    assert st["n_ok"] == args.docs
    eng.set_profiling(True)
    eng.reset_profile()
    b2 = eng.load_staged(staged)
    b2.sync_summaries_raw()
    t1 = time.perf_counter()
    b2.generate_sync_messages_device(states, args.docs)
    sm_ms = (time.perf_counter() - t1) * 1e3
    eng.set_profiling(False)
    t1 = time.perf_counter()
    b2.generate_sync_messages_raw(states, args.docs, want_sent=False, offsets=False)   # the same with the bytes copied to the host
    sm_host_ms = (time.perf_counter() - t1) * 1e3
    kt = eng.kernel_times()
    # a peer that holds nothing is sent every change but the filter's false positives without a sent dependency (1 %)
    assert 0.97 * st["changes"] <= msg_changes <= st["changes"], (msg_changes, st["changes"])
    tot = torch.tensor([float(elapsed), float(st["ops_loaded"]), float(st["changes"]), float(st["change_bytes"])], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max = float(tmax[0])
    else:
        elapsed_max = elapsed
    ops_all, chg_all, chg_bytes_all = float(tot[1]), float(tot[2]), float(tot[3])
    if rank == 0:
        sha_ms = kt.get("chg_hash", (0, 0))[0] + kt.get("chg_hash_seq", (0, 0))[0] + kt.get("sha_doc", (0, 0))[0]
        out = {"metric": "ops loaded+merged/sec (node)", "value": ops_all * args.steps / elapsed_max, "unit": "ops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "synthetic",
               "config": {"workload": "configs[4]: sync sweep — load (re-derive + SHA-256 every change, heads) + heads/Bloom summary per document + all-gather + one sync message per document to a peer that holds none of it",
                          "docs_per_gpu": args.docs, "docs_per_s": world * args.docs * args.steps / elapsed_max,
                          "changes_hashed_per_s": chg_all * args.steps / elapsed_max, "summary_bytes_gathered": gathered_bytes,
                          "sha256": {"kernel_ms": round(sha_ms, 3), "GB_hashed_per_s": round((chg_bytes_all / max(1, world) + st["encoded_bytes"]) / max(sha_ms, 1e-9) / 1e6, 2),
                                     "note": "VALU-integer bound (64 rounds of 32-bit ALU per 64-byte block), reported as bytes hashed, not against HBM"},
                          "sync_kernels_ms": {k: round(kt[k][0], 3) for k in ("sync_size", "sync_fill") if k in kt},
                          "sync_messages": {"records": args.docs, "message_bytes": msg_bytes, "changes_sent": msg_changes, "call_ms": round(sm_ms, 2), "call_with_host_copy_ms": round(sm_host_ms, 2),
                                            "kernels_ms": {k: round(kt[k][0], 3) for k in ("sm_lookup", "sm_plan", "sm_fill", "sm_bloom", "sm_jobs", "sm_copy") if k in kt},
                                            # the chunk copy moves every change chunk once (read + write); plan reads the change rows of a record a few times
                                            "sm_copy_GBps": round(2 * st["change_bytes"] / max(kt.get("sm_copy", (1e-9, 0))[0], 1e-9) / 1e6, 1),
                                            "sm_plan_GBps": round((st["changes"] * 120 + args.docs * 64) / max(kt.get("sm_plan", (1e-9, 0))[0], 1e-9) / 1e6, 1)},
                          "gen_s": round(t_gen, 2)}}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_apply(args, eng, rank, world, dist, torch, red_dev):
    """Sync-server round: `--docs` map documents per GPU (default 100k, configs[1]'s shape after 4 of its 5 rounds: 800 ops
    each) receive the 4 changes of the last round (200 ops) through one amh_apply_batch call.  The deltas are what
    save_after(base heads) of the full documents yields.  One step = load the base batch + apply; the apply share is
    reported separately.  ops = ops of the resulting documents, counted once."""
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), 32))
    seed = 0xA070 + rank * args.docs
    t0 = time.time()
    base_arena = eng.gen_map_arena(seed, args.docs, 4, 4, 50, 128, threads)
    full_arena = eng.gen_map_arena(seed, args.docs, 4, 5, 50, 128, threads)
    base_staged = eng.stage_arena(base_arena)
    bb = eng.load_staged(base_staged)
    fb = eng.load_staged(eng.stage_arena(full_arena))
    items = [(i, fb.save_after(i, bb.get_heads(i))) for i in range(args.docs)]
    delta_bytes = sum(len(x[1]) for x in items)
    # the received buffers of the round in one arena (as a server holds them): the call's pointer table is numpy's
    import numpy as np
    d_idx = np.arange(args.docs, dtype=np.uint32)
    d_offs = np.zeros(args.docs + 1, dtype=np.uint64)
    np.cumsum(np.fromiter((len(x[1]) for x in items), dtype=np.uint64, count=args.docs), out=d_offs[1:])
    d_blob = b"".join(x[1] for x in items)
    del items
    fb.close()
    bb.close()
    t_prep = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        b = eng.load_staged(base_staged)
        t1 = time.perf_counter()
        res = b.apply_arena(d_idx, d_blob, d_offs, incremental=True)
        return b, time.perf_counter() - t1, res

    for _ in range(args.warmup):
        step()[0].close()
    barrier()
    t0 = time.perf_counter()
    apply_s = 0.0
    for k in range(args.steps):
        b, dt, res = step()
        apply_s += dt
        if k + 1 < args.steps:
            b.close()
    barrier()
    elapsed = time.perf_counter() - t0
    assert not any(res)
    st = b.stats()
    assert st["n_ok"] == args.docs
    tot = torch.tensor([float(elapsed), float(st["ops_loaded"])], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max, ops_all = float(tmax[0]), float(tot[1])
    else:
        elapsed_max, ops_all = elapsed, float(st["ops_loaded"])
    if rank == 0:
        print(json.dumps({"metric": "ops loaded+merged/sec (node)", "value": ops_all * args.steps / elapsed_max, "unit": "ops/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "synthetic",
                          "config": {"workload": "sync-server round: load 800-op map documents, apply 4 received changes (200 ops) to each with one amh_apply_batch call",
                                     "docs_per_gpu": args.docs, "delta_bytes_per_gpu": delta_bytes, "apply_ms_per_step": apply_s / args.steps * 1e3,
                                     "docs_applied_per_s": world * args.docs / (apply_s / args.steps), "prep_s": round(t_prep, 1),
                                     "note": "the received buffers lie in one arena (pointer table built with numpy, inside apply_ms); every changed document is re-verified after the apply"}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=None, help="documents per GPU (default: 10k for merge = configs[2]; 100k for load = configs[1])")
    ap.add_argument("--cpu-sample", type=int, default=16000, help="documents of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", default="merge", choices=["merge", "load", "bigdoc", "sync", "apply", "trace"],
                    help="merge = configs[2] (the headline line: BASELINE.json quotes the metric on it): 10k text documents x 8 divergent replicas, "
                         "load 8 + merge 7; load = configs[1]: 100k JSON-map documents; bigdoc = configs[3]: one 10M-op text document")
    ap.add_argument("--slice", type=int, default=10000, help="merge workload: documents streamed through the pipeline at a time")
    ap.add_argument("--ops", type=int, default=10000000, help="bigdoc workload: ops of the document")
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--sequential-merges", action="store_true", help="merge workload: R-1 amh_merge_batch calls instead of one amh_merge_many")
    ap.add_argument("--source-batches", type=int, default=1, help="merge workload: 1 = the R-1 source replicas of all documents are loaded as ONE "
                    "batch of (R-1) x docs documents (default: one load call, one set of dependency levels to hash); 0 = one batch per replica")
    args = ap.parse_args()
    if args.docs is None:
        args.docs = {"load": 100000, "merge": 10000, "sync": 125000, "apply": 100000, "bigdoc": 1, "trace": 1}[args.workload]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    # one process per GPU; AMH_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("AMH_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("AMH_BENCH_FORCE_DIST"):   # the latter: rehearse the RCCL code path with one rank on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from automerge_amd import Engine
    eng = Engine(gpu=local_rank)
    if args.workload == "merge":
        return bench_merge(args, eng, rank, world, dist, torch, red_dev)
    if args.workload == "bigdoc":
        if args.replicas == 8:
            args.replicas = 16
        return bench_bigdoc(args, eng, rank, world, dist, torch, red_dev)
    if args.workload == "trace":
        return bench_trace(args, eng, rank, world, dist, torch, red_dev)
    if args.workload == "sync":
        return bench_sync(args, eng, rank, world, dist, torch, red_dev)
    if args.workload == "apply":
        return bench_apply(args, eng, rank, world, dist, torch, red_dev)
    omp = int(os.environ.get("OMP_NUM_THREADS", "16"))
    threads = max(1, min(omp if world == 1 else max(4, omp // world), 32))   # host threads of the (untimed) generator
    seed = 0xA070 + rank * args.docs
    t0 = time.time()
    arena = eng.gen_map_arena(seed, args.docs, 4, 5, 50, 128, threads)
    t_gen = time.time() - t0
    t0 = time.time()
    staged = eng.stage_arena(arena)   # host ingest + H2D: not part of the timed region (inputs resident in HBM)
    t_stage = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.load_staged(staged).close()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b = eng.load_staged(staged)   # returns after the stream has drained (per-document status is read back)
        b.close()
    barrier()
    elapsed = time.perf_counter() - t0
    # one more, profiled pass: HIP events around every kernel launch on the pipeline's own stream
    eng.set_profiling(True)
    eng.reset_profile()
    batch = eng.load_staged(staged)
    eng.set_profiling(False)
    kt = eng.kernel_times()
    st = batch.stats()
    n_bad = st["n_docs"] - st["n_ok"]
    if n_bad:
        raise SystemExit(f"{n_bad} documents failed to load: the benchmark input must load cleanly")
    ops = st["ops_loaded"]
    tot = torch.tensor([float(elapsed), float(ops)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max, ops_all = float(tmax[0]), float(tot[1])
    else:
        elapsed_max, ops_all = elapsed, float(ops)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    import ctypes as C
    st["actors"] = 4 * st["n_docs"]
    st["actor_map"] = 4 * st["changes"]
    cols = column_bytes(eng, batch, st["n_docs"])
    model = algorithmic_bytes(st, cols)
    # sha_doc and visible run on the engine's side stream beside the count pass / the change encode: their event
    # times include the time they share the device, so they are reported but neither ranked nor summed
    OVERLAPPED = ("sha_doc", "visible")
    dom = max((k for k in kt if kt[k][1] > 0 and k not in OVERLAPPED), key=lambda k: kt[k][0])
    dom_ms, dom_calls = kt[dom]
    per_launch_ms = dom_ms / dom_calls
    per_launch_bytes = model[dom] / dom_calls
    achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    total_kernel_ms = sum(v[0] for k, v in kt.items() if k not in OVERLAPPED)
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    # (profiles/traffic_latest.json, tools/pmc_traffic.py); null when the file is missing
    traffic = pmc_traffic("load", dom)
    kernels = {k: {"ms": round(v[0] / max(1, v[1]), 4), "launches": v[1],
                   "GBps": round(model.get(k, 0) / v[1] / (v[0] / v[1] * 1e-3) / 1e9, 2) if v[0] > 0 and v[1] > 0 else None,
                   **({"overlapped": True} if k in OVERLAPPED else {})}
               for k, v in kt.items() if v[1] > 0}
    out = {
        "metric": "ops loaded+merged/sec (node)", "value": ops_all * args.steps / elapsed_max, "unit": "ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64", "data": "synthetic",
        "config": {"workload": "configs[1]: JSON-map documents batch load + visibility on one MI355X per rank",
                   "docs_per_gpu": args.docs, "ops_per_doc": ops / max(1, st["n_docs"]), "encoded_bytes_per_gpu": st["encoded_bytes"],
                   "sharding": f"documents x{world}", "gen_s": round(t_gen, 2), "stage_s": round(t_stage, 2),
                   "pcie_inclusive_ops_per_s": ops / (elapsed / args.steps + t_stage)},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": per_launch_bytes, "limited_by": ROOFLINE_NOTES.get(dom),
                     "launch_ms": per_launch_ms, "kernel_ms_total": total_kernel_ms,
                     "sha256": sha_roofline(st["encoded_bytes"] + st["change_bytes"], sum(kt.get(k, (0, 0))[0] for k in ("chg_hash", "chg_hash_seq", "sha_doc")))},
        "kernels": kernels,
    }
    if not args.no_cpu and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as o   # CPU oracle: the reported baseline only (never the measured path)
        import numpy as np
        ns = min(args.cpu_sample, args.docs)
        cores = max(1, min(threads, 16))
        offs = np.ctypeslib.as_array(arena.offs, shape=(ns + 1,)).astype(np.uint64)
        ops_c, fails = C.c_ulong(), C.c_ulong()
        secs = o.lib.amo_bench_load_batch(C.cast(arena.o.data, C.c_char_p), C.cast(offs.ctypes.data, C.POINTER(C.c_uint64)), ns, cores, 1,
                                          C.byref(ops_c), C.byref(fails))
        out["cpu_baseline"] = {"value": ops_c.value / secs, "unit": "ops/s", "cores": cores, "kind": "port",
                               "sample": f"first {ns} documents of the same batch, load + materialize, one document per thread "
                                         f"(CPU restatement in oracle/, not the Rust reference: no toolchain)", "failures": fails.value}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
