/* automerge_hip.h — C ABI of libautomerge_hip.so: the MI355X-native batched load / merge / save
 * engine for Automerge documents.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference/rust, reference = aasthaagarwal2003/automerge, `automerge` 0.5.2).  A batch is N
 * independent documents; entry point X over a batch == the reference call per document, and a
 * failing document never aborts the batch (its per-document status carries the reference's error
 * kind).  The single-document calls of the reference are the n = 1 case.
 *
 * Ownership: inputs are caller-owned and only read during the call.  Outputs are library-owned until
 * the matching *_free (mirrors AMresultFree, automerge-c/src/result.rs:532).  A context is used from
 * one host thread at a time; one context drives one GPU (one process per GPU under torch.distributed).
 * There is no CPU fallback: without a HIP device every call fails with AMH_ERR_NO_DEVICE.
 */
#ifndef AUTOMERGE_HIP_H
#define AUTOMERGE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amh_ctx amh_ctx;
typedef struct amh_batch amh_batch;
typedef struct amh_staged amh_staged;

/* call-level status (per-document results are in amh_doc_status) */
enum {
  AMH_SUCCESS = 0,
  AMH_ERR_INVALID_ARG = -1,
  AMH_ERR_NO_DEVICE = -2,      /* HIP runtime / device missing */
  AMH_ERR_DEVICE = -3,         /* a HIP call failed */
  AMH_ERR_OOM = -4,
  AMH_ERR_BATCH_TOO_LARGE = -5,/* more than 2^32-1 rows / bytes of derived data in one batch */
  AMH_ERR_UNIMPLEMENTED = -6,  /* declared by the boundary, not built in this round */
  AMH_ERR_DOC_FAILED = -7      /* the addressed document is in an error state */
};

/* per-document error kinds: AutomergeError (automerge/src/error.rs:7-62) with the nested
 * storage::load::Error (automerge/src/storage/load.rs:15-30), reconstruct_document::Error
 * (storage/load/reconstruct_document.rs:14-32) and change_collector::Error
 * (storage/load/change_collector.rs:19-33) discriminants flattened. */
enum {
  AMH_DOC_OK = 0,
  AMH_DOC_PARSE = 1,
  AMH_DOC_BAD_CHECKSUM = 2,
  AMH_DOC_INVALID_CHANGE_COLUMNS = 3,
  AMH_DOC_MISSING_DEPS = 5,
  AMH_DOC_DUPLICATE_SEQ = 6,
  AMH_DOC_READ_OP = 20,
  AMH_DOC_OPS_OUT_OF_ORDER = 21,
  AMH_DOC_MISSING_ACTOR = 22,
  AMH_DOC_MISMATCHING_HEADS = 23,
  AMH_DOC_MISSING_OPS = 24,
  AMH_DOC_SUCC_OUT_OF_ORDER = 25,
  AMH_DOC_INVALID_OP = 26,
  AMH_DOC_CHANGES_OUT_OF_ORDER = 30,
  AMH_DOC_MISSING_CHANGE = 31,
  AMH_DOC_READ_CHANGE = 32,
  AMH_DOC_INCORRECT_MAX_OP = 33,
  AMH_DOC_COLLECT_MISSING_ACTOR = 34,
  AMH_DOC_COLLECT_MISSING_OPS = 35,
  AMH_DOC_TOO_LARGE = 90,      /* engine limit, not a reference error */
  AMH_DOC_UNSUPPORTED = 91     /* valid for the reference, not handled by this round's device path */
};

/* LoadOptions (automerge/src/automerge.rs:58-135).  verify_heads = VerificationMode::Check. */
typedef struct amh_load_opts {
  int verify_heads;        /* default 1 */
  int on_partial_load;     /* 0 = Error (default), 1 = Ignore */
} amh_load_opts;

/* SaveOptions (automerge/src/automerge.rs:1816-1830) */
typedef struct amh_save_opts {
  int deflate;             /* default 1 */
  int retain_orphans;      /* default 1 */
} amh_save_opts;

typedef struct amh_bytes_out {
  uint8_t* data;
  size_t len;
} amh_bytes_out;

/* Change accessors of automerge/src/change.rs:47-117 as one view; pointers are into the batch. */
typedef struct amh_change_view {
  const uint8_t* hash;       /* 32 bytes — Change::hash */
  const uint8_t* raw_bytes;  /* uncompressed change chunk — Change::raw_bytes */
  size_t raw_len;
  const uint8_t* actor;      /* Change::actor_id */
  size_t actor_len;
  uint64_t seq;              /* Change::seq */
  uint64_t start_op;         /* Change::start_op */
  uint64_t num_ops;          /* Change::len */
  int64_t timestamp;         /* Change::timestamp */
  uint32_t num_deps;         /* Change::deps().len(); the hashes are inside raw_bytes */
} amh_change_view;

typedef struct amh_batch_stats {
  uint64_t n_docs, n_ok;
  uint64_t ops_loaded;       /* sum of Change::len over documents that loaded (SURVEY.md 8d "ops") */
  uint64_t op_rows, succ_entries, changes, delete_ops;
  uint64_t encoded_bytes;    /* input bytes */
  uint64_t change_bytes;     /* re-derived change chunk bytes */
} amh_batch_stats;

/* ---- context ----------------------------------------------------------------------------- */
/* n_gpus must be 1: one context (and one process) per GPU; multi-GPU = one rank per GPU sharding
 * the batch by document.  A context owns two HIP streams (the main chain and a side stream
 * for kernels that are independent of it), a caching device allocator and a pinned staging area; calls on one
 * context must not overlap (one caller thread at a time), every call returns with both streams drained. */
amh_ctx* amh_ctx_create(const int* gpu_ids, int n_gpus);
void amh_ctx_destroy(amh_ctx*);
const char* amh_last_error(const amh_ctx*);   /* message of the last failed call (may be NULL ctx) */

/* ---- load ------------------------------------------------------------------------------------
 * == Automerge::load_with_options per document (automerge/src/automerge.rs:601-705; C precedent
 *    AMload, automerge-c/src/doc.rs:512).
 * amh_load_batch = amh_stage_batch (host ingest: inflate of DEFLATEd columns, document.rs /
 * compression.rs:51-74, + one H2D copy) followed by amh_load_staged (the device pipeline).
 * An input is what the reference's load accepts: a document chunk, optionally followed by change chunks
 * (save_incremental output), or change chunks alone; the change chunks go through the apply path with the
 * reference's policies (MissingDeps when the first chunk is not a document, on_partial_load). */
/* Multi-GPU: one process and one context per GPU.  Load, merge, apply and save shard by document and
 * need no collective: amh_partition_docs cuts a batch into per-rank ranges of equal encoded bytes.  The one exchange
 * step of the path, the all-gather of sync summaries (what the reference's sync server keeps per peer, sync.rs:243-250),
 * runs inside the library on device buffers through RCCL over xGMI: amh_comm_unique_id on rank 0, the 128 bytes handed
 * to every rank out of band (torch.distributed, MPI, a file), amh_comm_init on every rank, then
 * amh_sync_summaries_all.  rank_off[world + 1] receives where each rank's records start in out. */
#define AMH_COMM_ID_BYTES 128
int amh_comm_unique_id(uint8_t* id128);
int amh_comm_init(amh_ctx*, const uint8_t* id128, int rank, int world);
int amh_sync_summaries_all(amh_ctx*, const amh_batch*, amh_bytes_out* out, size_t* rank_off);
int amh_partition_docs(const size_t* lens, size_t n, int world, size_t* first);

/* Limits against hostile inputs (no counterpart in the reference, which loads one document at a time and loses only
 * that document): a document whose op rows, succ entries, changes or deps exceed max_doc_rows (0 = the built-in
 * 2^28), and the largest documents of a batch whose derived device arrays would exceed max_batch_bytes, fail with
 * AMH_DOC_TOO_LARGE; the other documents of the batch load.  The size of the derived arrays is an estimate from the
 * batch's counts (224 bytes per op row, 72 per succ entry, 420 per change, a little on the high side).
 * max_batch_bytes = 0: 90 % of the device's total memory, a figure that does not depend on what is allocated at the
 * moment — the same batch gets the same per-document answers on every run; a batch under it that does not fit what is
 * free right now fails as a whole (amh_load_batch returns NULL, "device out of memory"), it does not shed documents.
 * tests: test_byte_budget_sheds_the_largest_document (a batch just under the budget loads whole, one byte less fails
 * exactly its largest document). */
int amh_ctx_set_limits(amh_ctx*, uint64_t max_doc_rows, uint64_t max_batch_bytes);
amh_staged* amh_stage_batch(amh_ctx*, const uint8_t* const* docs, const size_t* lens, size_t n);
void amh_staged_free(amh_staged*);
amh_batch* amh_load_staged(amh_ctx*, const amh_staged*, const amh_load_opts*);
amh_batch* amh_load_batch(amh_ctx*, const uint8_t* const* docs, const size_t* lens, size_t n, const amh_load_opts*);
void amh_batch_free(amh_batch*);
/* Keep a loaded batch's history (changes, hashes, heads: what amh_heads, amh_changes, amh_save_after,
 * amh_sync_summaries and the source side of amh_merge_many read) and give the per-op arrays back to the context's
 * pool — about 95 % of a batch.  The reference has no counterpart (a merge source there is a whole Automerge); it is
 * what lets 10,000 documents x 8 replicas be merged in one pass on one GPU.  Calls that need the op rows (hydrate,
 * save, apply, merge as dst, amh_merge_batch) answer AMH_ERR_INVALID_ARG afterwards. */
int amh_batch_release_ops(amh_ctx*, amh_batch*);
void amh_bytes_free(amh_bytes_out*);

/* per-document Result: Ok / AutomergeError kind (automerge/src/error.rs:7-62) */
int amh_doc_status(const amh_batch*, uint32_t doc, int* err_kind, const char** msg);
int amh_batch_get_stats(const amh_batch*, amh_batch_stats* out);

/* == Automerge::get_heads (automerge.rs:1300-1304; AMgetHeads, automerge-c/src/doc.rs:402): sorted */
int amh_heads(const amh_batch*, uint32_t doc, const uint8_t** hashes32, size_t* n);
/* == Automerge::get_changes(&[]) (automerge.rs:1306; AMgetChanges): history order.
 * On input *n = capacity of out (0 to query the count), on output the number of changes. */
int amh_changes(const amh_batch*, uint32_t doc, amh_change_view* out, size_t* n);
/* canonical materialization for parity == hydrate / get_all (automerge/src/hydrate.rs:128-167,
 * automerge.rs:1528-1565, iter/top_ops.rs:43-102); format in DESIGN.md */
int amh_hydrate_json(const amh_batch*, uint32_t doc, amh_bytes_out* out);
/* == ReadDoc::text (read.rs; OpSet::text op_set.rs:307-311, OpType::to_str types.rs:278-286) for every document of the
 * batch at once: the winners of the text objects' elements in document order, string values as they are, any other
 * visible value as U+FFFC.  A document's text = the texts of its text objects in document order, back to back (one
 * text object per document, as in BASELINE.json configs[2] / [3]: exactly text(obj)).  out = all texts back to back
 * (UTF-8), offsets[n_docs + 1] = where each document's starts; freed with amh_bytes_free / amh_u64_free. */
int amh_text_batch(amh_ctx*, const amh_batch*, amh_bytes_out* out, uint64_t** offsets);
/* The same texts left in DEVICE memory for callers that consume them on the GPU (an indexer, a tokenizer): *d_text = all
 * texts back to back, *d_doc_off = u32[n_docs + 1] where each document's starts, *total = bytes.  The two buffers belong to
 * the batch and stay valid until the next amh_text_batch* call on it, a merge / apply into it, or amh_batch_free. */
int amh_text_batch_device(amh_ctx*, const amh_batch*, const uint8_t** d_text, const uint32_t** d_doc_off, uint64_t* total);
/* max op counter of the document (Automerge::max_op, automerge.rs:181) */
int amh_doc_max_op(const amh_batch*, uint32_t doc, uint64_t* max_op);

/* ---- merge / apply / save ---------------------------------------------------------------------
 * == dst[dst_idx[p]].merge(&mut src[src_idx[p]]) for every pair, in place (automerge.rs:917-940; AMmerge,
 * automerge-c/src/doc.rs:564).  dst and src are different batches; a document appears in at most one pair
 * per call.  Returns AMH_ERR_DOC_FAILED when some pairs were left unmerged (a document that failed to load,
 * DuplicateSeqNumber); their dst documents are unchanged.  Every merged document is re-verified (change
 * hashes, heads) before it replaces the old one. */
int amh_merge_batch(amh_ctx*, amh_batch* dst, const amh_batch* src, const uint32_t* dst_idx, const uint32_t* src_idx, size_t n_pairs);
/* == dst.merge(&mut s0); dst.merge(&mut s1); ... per pair with ONE pass over the rows (same result, history order
 * included, as the sequential amh_merge_batch calls: the changes each source adds, automerge.rs:1311-1334, go through
 * the apply path as one apply_changes call per source).  src_idx[s * n_pairs + p] = document of srcs[s] merged into
 * dst[dst_idx[p]], or 0xFFFFFFFF for none.  Built for configs[2] (8 replicas per document): rows are placed once instead
 * of 7 times.  A source whose added changes repeat an (actor, seq) of the document ends that document's merges where
 * the reference's sequential calls would (DuplicateSeqNumber from the middle of apply_changes, automerge.rs:804-812):
 * the sources in front of it and its own changes in front of the offender stay applied, later sources are not merged,
 * and the call returns AMH_ERR_DOC_FAILED (tests/merge_cases.py check_merge_many_duplicate_seq, against the oracle).
 * History: when every source is a loaded batch, the merged document ADOPTS the sources' change chunks and hashes (each
 * was re-derived from its ops and hashed when its source was loaded), as the reference's merge keeps the stored
 * changes; they are not derived again from the merged rows unless AMH_VERIFY_MERGE=1 is set (the test suites run both).
 * The same batch may be named as several sources (srcs[s] equal, src_idx shifted): all source replicas of all
 * documents can be loaded as one batch. */
int amh_merge_many(amh_ctx*, amh_batch* dst, const amh_batch* const* srcs, size_t n_srcs, const uint32_t* dst_idx,
                   const uint32_t* src_idx, size_t n_pairs);
/* == Automerge::apply_changes (automerge.rs:785-858; AMapplyChanges, automerge-c/src/doc.rs:84) on one document:
 * changes[i] = one change chunk, plain or compressed (what Change::try_from takes, change.rs:161-180: the stored
 * checksum is not compared).  Changes the document already has are skipped; changes whose dependencies are missing
 * wait in the document's queue (amh_doc_queue_len) and apply when a later call brings the missing ones; save keeps
 * them behind the document chunk (retain_orphans).  Returns AMH_ERR_DOC_FAILED when a chunk does not parse (the
 * document is unchanged) or a change repeats a sequence number (DuplicateSeqNumber: as in the reference,
 * automerge.rs:804-812, the changes in front of the offending one stay applied and queued ones stay queued).  This is
 * the n = 1 form of amh_apply_batch and, like it, reloads the whole batch: apply to many documents with one
 * amh_apply_batch call. */
int amh_apply_changes(amh_ctx*, amh_batch*, uint32_t doc, const uint8_t* const* changes, const size_t* lens, size_t n);
/* Batched form.  Buffer i holds one or more concatenated chunks for document doc_idx[i]; several buffers may name
 * the same document: each buffer is one apply_changes / load_incremental call on it, in the order given (the queue
 * of waiting changes is drained at the end of every buffer, automerge.rs:797-827).
 *   incremental = 0: apply_changes per document, as above.
 *   incremental = 1: == Automerge::load_incremental on a loaded document (automerge.rs:730-769; AMloadIncremental,
 *                    automerge-c/src/doc.rs:537): checksums are verified and the chunks in front of the first
 *                    unreadable one are applied (storage/load.rs:56-114).
 * results (may be NULL) receives per buffer the outcome for its document: AMH_DOC_OK or the AutomergeError kind
 * (AMH_DOC_PARSE, AMH_DOC_DUPLICATE_SEQ, AMH_DOC_UNSUPPORTED = the applied changes do not survive the trip through
 * the document format, e.g. a pred naming no op).  Every changed document is re-verified (change hashes, heads). */
int amh_apply_batch(amh_ctx*, amh_batch*, const uint32_t* doc_idx, const uint8_t* const* bufs, const size_t* lens, size_t n,
                    int incremental, int* results);
/* number of changes waiting for dependencies (Automerge::queue, automerge.rs:168) */
int amh_doc_queue_len(const amh_batch*, uint32_t doc, size_t* n);
/* == Automerge::save_with_options (automerge.rs:943-965; AMsave, automerge-c/src/doc.rs:751);
 * out = array of n_docs amh_bytes_out, each freed with amh_bytes_free.  A document that failed to load (or whose
 * stored actor table is not in bytewise order: not a shape writers produce) gets data = NULL.
 * deflate = 0 is save_nocompress (bit-exact); deflate = 1 compresses columns >= 256 bytes on the host with zlib
 * (same policy as the reference; compressed bytes are not flate2's: parity unpinned). */
int amh_save_batch(amh_ctx*, const amh_batch*, const amh_save_opts*, amh_bytes_out* out);

/* == Automerge::save_after(heads) (automerge.rs:993-1000): the changes that are not ancestors of `heads`
 * (get_changes, :1306 / :1022-1047; heads the document does not know are ignored), as raw change chunks in
 * history order — the counterpart of amh_apply_batch(incremental = 1) on the receiving side.  out is freed
 * with amh_bytes_free. */
int amh_save_after(const amh_batch*, uint32_t doc, const uint8_t* heads32, size_t n_heads, amh_bytes_out* out);
/* The same for n (document, heads) pairs in one pass on the device (a server sending every peer what it lacks): the clock
 * of the heads' ancestors, the changes above it, their chunks gathered in device memory and copied out once.
 * out = the n results back to back, result i = bytes [off[i], off[i + 1]) (freed with amh_bytes_free / amh_u64_free).
 * A document that failed to load gives an empty result and the call returns AMH_ERR_DOC_FAILED. */
int amh_save_after_batch(amh_ctx*, const amh_batch*, const uint32_t* doc_idx, const uint8_t* const* heads32, const size_t* n_heads,
                         size_t n, amh_bytes_out* out, uint64_t** off);

/* ---- sync summaries ---------------------------------------------------------------------------
 * per document: sorted heads (get_heads) + Bloom filter over all change hashes
 * (sync.rs:243-250 make_bloom_filter with last_sync = [], sync/bloom.rs:38-47,69-90,125-146).
 * Record layout per document (little-endian): u32 doc, u32 status (0 = loaded), u32 n_heads,
 * heads[32 x n], u32 bloom_len, bloom bytes (BloomFilter::to_bytes: uleb n, uleb 10, uleb 7, bits).
 * Across ranks: amh_sync_summaries_all below gathers the records of every rank inside the library (RCCL over xGMI). */
int amh_sync_summaries(amh_ctx*, const amh_batch*, amh_bytes_out* out);
/* No counterpart in the reference.  For callers that cannot allocate: tight serving loops, harnesses that forbid
 * hipMalloc, code between the captures of a HIP graph (the call itself runs on the context's own stream and waits for
 * it, so it cannot be captured).  The records of amh_sync_summaries, byte for byte, written into the
 * caller's DEVICE buffer d_out (cap bytes); d_meta (device, 2 x u32) receives {total bytes of all records, 1 if cap
 * was too small, or the records exceed 4 GiB - 1: nothing is written then}: a record that would end beyond cap is not
 * written, the ones in front of it are.  The call allocates
 * nothing on the host or the device (no pool, no event, no host copy): record offsets and scan temporaries live in a
 * static device array inside the library (4 MiB per GPU the library runs on, held until the process ends).
 * Limit: at most 1,048,576 documents in the batch (AMH_ERR_BATCH_TOO_LARGE beyond).  Calls on one GPU share that
 * array: calls that overlap on one GPU (two contexts, two threads) race.  Returns with the stream drained. */
int amh_sync_summaries_noalloc(amh_ctx*, const amh_batch*, uint8_t* d_out, size_t cap, uint32_t* d_meta);

/* ---- sync messages -----------------------------------------------------------------------------
 * == SyncDoc::generate_sync_message (sync.rs:134-221) for n (document, peer) records in one pass on the device: what
 * this side needs (get_missing_deps, automerge.rs:1776-1799), the Bloom filter of the changes since the shared heads
 * (make_bloom_filter, sync.rs:243-250), the changes the peer lacks (get_changes_to_send, sync.rs:252-321, probing the
 * peer's filters with BloomFilter::contains_hash, sync/bloom.rs:69-123), the reset message for a peer whose last_sync
 * is unknown here (sync.rs:150-166), the "nothing to say" cases (:178-205), and the bytes of Message::encode
 * (sync.rs:526-563) with the change chunks copied out of the batch's device memory.
 * amh_sync_state is sync::State (sync/state.rs:31-53) as the reference reads it; hashes are 32 bytes each, lists in the
 * order the State holds them (shared_heads goes into the message as it is).  flags: which of the Option fields are Some,
 * and in_flight.  A Bloom filter is its wire form (BloomFilter::to_bytes; empty = the default filter).
 * Results: kinds[i] = AMH_SYNC_NONE (generate_sync_message returned None: no bytes), AMH_SYNC_MESSAGE (the encoded
 * message; the caller then does what sync.rs:207-219 does to the State: last_sent_heads = the document's heads,
 * sent_hashes += the hashes in `sent`, in_flight = true), AMH_SYNC_RESET (the reset message; the State stays as it is)
 * or AMH_SYNC_FAILED (the document failed to load; the call returns AMH_ERR_DOC_FAILED, the other records are served).
 * msgs = the messages of all records back to back, message i = bytes [msg_off[i], msg_off[i + 1]); sent (may be NULL,
 * then sent_off is ignored) = the 32-byte hashes of the changes in the messages, those of record i at
 * [sent_off[i], sent_off[i + 1]).  msgs / sent are freed with amh_bytes_free, the offset arrays (n + 1 entries) with
 * amh_u64_free.
 * A filter with entries but no bits (the reference divides by zero on it) is AMH_ERR_INVALID_ARG.
 * The receiving side (receive_sync_message, sync.rs:323-383) is not part of the library yet: a caller applies a
 * message's changes with amh_apply_changes and keeps the State itself (tests/sync_cases.py shows the loop). */
typedef struct amh_sync_have {
  const uint8_t* last_sync32; size_t n_last_sync;
  const uint8_t* bloom; size_t bloom_len;
} amh_sync_have;
enum { AMH_SYNC_THEIR_HEADS = 1, AMH_SYNC_THEIR_NEED = 2, AMH_SYNC_THEIR_HAVE = 4, AMH_SYNC_IN_FLIGHT = 8 };
enum { AMH_SYNC_NONE = 0, AMH_SYNC_MESSAGE = 1, AMH_SYNC_RESET = 2, AMH_SYNC_FAILED = 3 };
typedef struct amh_sync_state {
  uint32_t doc;
  uint32_t flags;
  const uint8_t* shared_heads32; size_t n_shared_heads;
  const uint8_t* last_sent_heads32; size_t n_last_sent_heads;
  const uint8_t* their_heads32; size_t n_their_heads;
  const uint8_t* their_need32; size_t n_their_need;
  const amh_sync_have* their_have; size_t n_their_have;
  const uint8_t* sent_hashes32; size_t n_sent_hashes;
} amh_sync_state;
int amh_generate_sync_messages(amh_ctx*, const amh_batch*, const amh_sync_state* states, size_t n, uint32_t* kinds,
                               amh_bytes_out* msgs, uint64_t** msg_off, amh_bytes_out* sent, uint64_t** sent_off);
/* The same call with the results left in device memory (a sender that hands the bytes to a GPU-aware transport, or
 * batches further device work): d_msgs = the messages back to back, message i = bytes [d_msg_off[i], d_msg_off[i + 1])
 * (n + 1 u32 in device memory), total = their size; d_sent (may be NULL) = the 32-byte hashes of the changes in the
 * messages, those of record i being entries [d_sent_idx[i], d_sent_idx[i + 1]) of it, n_sent in all.  kinds comes to
 * the host as above.  The buffers belong to the batch and live until the next amh_generate_sync_messages* call on it or
 * amh_batch_free. */
int amh_generate_sync_messages_device(amh_ctx*, const amh_batch*, const amh_sync_state* states, size_t n, uint32_t* kinds,
                                      const uint8_t** d_msgs, const uint32_t** d_msg_off, uint64_t* total,
                                      const uint8_t** d_sent, const uint32_t** d_sent_idx, uint64_t* n_sent);

/* ---- measurement ------------------------------------------------------------------------------
 * HIP-event timing of every kernel launch of the context's stream (for bench.py's roofline leg). */
int amh_ctx_set_profiling(amh_ctx*, int enable);
int amh_ctx_num_kernels(void);
const char* amh_ctx_kernel_name(int k);
/* accumulated milliseconds and launch count since the last reset */
int amh_ctx_kernel_time(amh_ctx*, int k, double* ms, uint64_t* launches);
int amh_ctx_reset_profile(amh_ctx*);
/* device memory of the context's caching allocator: bytes held by live batches / staged inputs, and bytes parked for reuse */
int amh_ctx_pool_bytes(const amh_ctx*, uint64_t* live, uint64_t* pooled);

/* Diagnostics (no counterpart in the reference): copy one named device array of a batch to the host
 * (names = fields of BatchDev, automerge_amd/csrc/amh_types.h).  *len receives the byte size; with
 * cap == 0 only the size is returned.  Used by tests to diff the HIP build against the host
 * emulation of the same kernel bodies stage by stage. */
int amh_debug_array(const amh_batch*, const char* name, void* out, size_t cap, size_t* len);

/* ---- synthetic workloads --------------------------------------------------------
 * Deterministic generator of reference-format documents (save_nocompress form): `n_docs` map
 * documents of n_actors x changes_per_actor x ops_per_change ops on the root map and one nested map,
 * keys drawn Zipf from `n_keys` strings (BASELINE.json configs[1]).  Output: concatenated documents
 * + offsets[n_docs + 1]; free both with amh_bytes_free / amh_u64_free. */
int amh_gen_map_docs(uint64_t seed, uint32_t n_docs, uint32_t n_actors, uint32_t changes_per_actor,
                     uint32_t ops_per_change, uint32_t n_keys, int threads, amh_bytes_out* docs, uint64_t** offsets);
/* Text documents (BASELINE.json configs[2], [3]): per document one base change (makeText + base_len
 * characters) and n_replicas divergent replicas of ops_per_replica edits (70 % inserts in runs, 30 %
 * deletes) in changes of ops_per_change ops.  Output: n_docs * n_replicas saved documents, replica r
 * of document i at index i * n_replicas + r (base + that replica's changes). */
int amh_gen_text_docs(uint64_t seed, uint32_t n_docs, uint32_t n_replicas, uint32_t base_len, uint32_t ops_per_replica,
                      uint32_t ops_per_change, int threads, amh_bytes_out* docs, uint64_t** offsets);
void amh_u64_free(uint64_t*);

#ifdef __cplusplus
}
#endif
#endif
