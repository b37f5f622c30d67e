// Plain-data layouts shared by the host orchestration and the HIP kernels of the batched
// load path.  Everything is structure-of-arrays over the whole batch ("all docs, all rows"),
// segmented by per-document base offsets — see DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define AMH_NULL32 0xFFFFFFFFu
#define AMH_MAX_ROWS_PER_DOC (1u << 28)

// per-document status = the reference's per-document Result (error.rs:7-62, storage/load.rs:15-30,
// reconstruct_document.rs:14-32, change_collector.rs:19-33).  0 = Ok.
enum AmhStatus : i32 {
  AMH_OK = 0,
  AMH_E_PARSE = 1,                 // load::Error::Parse (chunk / column metadata level)
  AMH_E_BAD_CHECKSUM = 2,          // load::Error::BadChecksum
  AMH_E_INVALID_CHANGE_COLUMNS = 3,
  AMH_E_READ_OP = 20,              // InflateDocument(ReadOp)
  AMH_E_OPS_OUT_OF_ORDER = 21,     // InflateDocument(OpsOutOfOrder)
  AMH_E_MISSING_ACTOR = 22,        // InflateDocument(MissingActor)
  AMH_E_MISMATCHING_HEADS = 23,    // InflateDocument(MismatchingHeads)
  AMH_E_MISSING_OPS = 24,          // InflateDocument(MissingOps)
  AMH_E_SUCC_OUT_OF_ORDER = 25,    // InflateDocument(SuccOutOfOrder)
  AMH_E_INVALID_OP = 26,           // invalid action / non numeric increment
  AMH_E_CHANGES_OUT_OF_ORDER = 30, // InvalidChanges(ChangesOutOfOrder)
  AMH_E_MISSING_CHANGE = 31,       // InvalidChanges(MissingChange)
  AMH_E_READ_CHANGE = 32,          // InvalidChanges(ReadChange)
  AMH_E_INCORRECT_MAX_OP = 33,     // InvalidChanges(IncorrectMaxOp)
  AMH_E_COLLECT_MISSING_ACTOR = 34,
  AMH_E_COLLECT_MISSING_OPS = 35,
  AMH_E_MISSING_DEPS = 5,          // AutomergeError::MissingDeps
  AMH_E_DUPLICATE_SEQ = 6,         // AutomergeError::DuplicateSeqNumber
  // not reference errors: limits of this engine, reported instead of guessing
  AMH_E_TOO_LARGE = 90,            // document exceeds AMH_MAX_ROWS_PER_DOC / 4 GiB chunk
  AMH_E_UNSUPPORTED = 91,          // well-formed for the reference but outside what this round handles
};

// op-column slots of a document chunk (doc_op_columns.rs:24-32) in decode-stream order
enum OpCol {
  OC_OBJ_ACTOR = 0, OC_OBJ_CTR, OC_KEY_ACTOR, OC_KEY_CTR, OC_KEY_STR, OC_ID_ACTOR, OC_ID_CTR, OC_INSERT, OC_ACTION,
  OC_VAL_META, OC_VAL_RAW, OC_SUCC_NUM, OC_SUCC_ACTOR, OC_SUCC_CTR, OC_EXPAND, OC_MARK_NAME, OC_COUNT
};
// change-metadata column slots (doc_change_columns.rs:18-24)
enum ChgCol {
  CC_ACTOR = 0, CC_SEQ, CC_MAX_OP, CC_TIME, CC_MESSAGE, CC_DEPS_NUM, CC_DEPS_IDX, CC_EXTRA_META, CC_EXTRA_RAW, CC_COUNT
};

struct ColRef {
  u32 off;  // byte offset relative to the start of the document's chunk
  u32 len;
};

// Result of the prefix parse of one document chunk (document.rs:113-154)
struct DocDesc {
  u64 chunk_off;    // offset of the chunk in the batch arena
  u32 chunk_len;    // header + data
  u32 data_off;     // relative to chunk start
  u32 data_len;
  u32 n_actors;
  u32 actors_off;   // relative offset of the first length-prefixed actor
  u32 n_heads;
  u32 heads_off;    // relative offset of heads[32 x n]
  u32 suffix_off;   // head indices (may be absent)
  u32 suffix_len;
  ColRef ccol[CC_COUNT];
  ColRef ocol[OC_COUNT];
};

// per-document segment bases (exclusive scans of the counts)
struct DocCounts {
  u32 n_rows, n_succ, n_chg, n_deps;
};

// everything the kernels touch, as one POD so that launches take a single argument block
struct BatchDev {
  u32 n_docs;
  u32 tot_rows, tot_succ, tot_chg, tot_deps, tot_actors, tot_dt, tot_ops;
  u32 check_heads;         // VerificationMode::Check (reconstruct_document.rs:86-90)
  u32 no_checksum;         // the stored checksum is not compared (Change::try_from, change.rs:161-180, does not look at it)
  // deferred checksum verdict (batch load): k_sha_doc runs beside the count pass on the side stream, looks at the
  // statuses as k_parse left them (sha_gate) and only notes a mismatch (sha_bad); k_extract, behind the join, turns
  // the note into the document's status, ahead of whatever the count made of the damaged bytes — the order the
  // reference checks in — and before anything is sized or decoded.  Null = the kernel fails the document itself
  // (apply's change chunks).
  u32 *a_tab_base, *a_tab;            // [tot_actors + 1] / [tot_chg + 1] an actor's changes by sequence number (k_atab_*)
  u64* a_tab_bn;                     // [tot_actors + 1] {a_tab_base, number of changes} of an actor in one word (k_bucket's dependent-load chain)
  u32* a_tab_max;                    // [tot_chg + 1] max_op (saturated to 32 bits) beside a_tab: the key k_bucket searches
  u32 *cd_err, *cd_code, *cd_none;   // [n_docs * 8] change-metadata columns (k_decode_chg_cols): first change / entry that does not read, its error, "column exhausted behind the last change"
  const i32* sha_gate;
  u32 host_chain_min;      // != 0: the changes of a document with more dependency levels than the level-parallel hashing handles and at least this many changes are hashed by the host (a chain of hashes is serial: ~8 us per change on a lone lane, ~0.3 us on a core)
  u32 sha_host_min;        // != 0: chunks of at least this many bytes are hashed by the host (one message is a serial chain: a lone lane is ~50x slower than a core)
  u8* sha_bad;
  u32 cur_level;           // start, in lvl_order, of the dependency level hashed by the current k_chg_hash launch
  u32* lvl_order;          // [tot_chg] changes grouped by dependency level
  u32* lvl_start;          // [AMH_LEVEL_CAP + 2] start of each level in lvl_order
  u32* max_level;          // [1] deepest dependency level in the batch
  u32* doc_level;          // [n_docs] deepest level per document
  const u8* arena;
  const u64* doc_off;      // [2 * n_docs] (start, end) byte range of each input in the arena
  i32* status;             // [n_docs]
  u32* fail_line;          // [n_docs] source line of the check that failed the document (diagnostics)
  DocDesc* desc;           // [n_docs]
  DocCounts* cnt;          // [n_docs]
  // [n_docs + 1] exclusive scans of the per-document counts
  u32 *row_base, *succ_base, *chg_base, *dep_base, *actor_base, *ht_base, *dt_base, *map_base;
  u8* doc_hash;            // [n_docs * 32] SHA-256 of the document chunk
  u32* doc_max_op;         // [n_docs] max counter over op ids and succ ids (reconstruct_document.rs:136-143)
  u32* heads_ok;           // [n_docs]
  // actors [tot_actors]
  u32 *actor_off, *actor_len;    // bytes, relative to chunk start
  u32 *actor_rank, *actor_by_rank;  // bytewise rank inside the document (lamport tie-break, types.rs:517-521)
  u32 *a_first_chg, *a_last_chg; // per-actor change chain in file order
  // change metadata rows (file order) [tot_chg]
  u32* c_actor;
  u64 *c_seq, *c_max_op;
  i64* c_time;
  u32 *c_msg_off, *c_msg_len;    // len = AMH_NULL32 for "no message"
  u32 *c_ndeps, *c_dep_base;     // dep_base: index into dep_idx (global)
  u32 *c_extra_off, *c_extra_len;
  u32* c_doc;
  u32* dep_idx;                  // [tot_deps] document-local change index
  // derived per change
  u32* c_nops;
  u32* c_npred;                  // [tot_chg] pred entries of the change's ops (k_chg_layout; nullptr where the batch's changes were not re-encoded)
  u64* c_start_op;
  u32* c_op_base;                // [tot_chg + 1] slot base in ops_sorted
  u32* c_next_same_actor;
  u32* c_level;                  // dependency depth of the change inside its document
  u8* c_is_dep;                  // some later change lists it as a dep (not a head)
  u8* c_hash;                    // [tot_chg * 32]
  u32* c_out_off;                // [tot_chg + 1] offset of the re-derived change chunk in out_arena
  u32 *c_out_len, *c_body_len, *c_cols_off;
  u32* c_col_len;                // [tot_chg * 14] encoded column lengths
  u32* c_wide;                   // [tot_chg] bit k: column k holds a literal run of more than 64 values
  u32* c_map_base;               // base into chg_actor_map
  u32* c_n_other;
  u32* chg_actor_map;            // [sum over changes of n_actors(doc)]: index in the change's actor table
  // op rows (document order) [tot_rows]
  u32 *r_id_actor, *r_id_ctr, *r_obj_actor, *r_obj_ctr, *r_key_actor, *r_key_ctr;
  u32 *r_key_off, *r_key_len;    // key string (relative to chunk), len = AMH_NULL32 when not a map key
  u8 *r_insert, *r_action, *r_expand;
  u64* r_vmeta;
  u32* r_vraw_off;               // offset inside the value-raw column
  u32 *r_mark_off, *r_mark_len;  // AMH_NULL32 len = none
  u32 *r_succ_cnt, *r_succ_base; // base is global
  u32* r_doc;
  u32* r_inc_cnt;
  i64* r_inc_sum;
  u8 *r_visible, *r_winner;
  // succ entries [tot_succ]
  u32 *s_actor, *s_ctr;
  u32* s_row;                    // owning row (global)
  u32* s_target;                 // op index t of the op this entry names
  // id -> row tables, one region per document, sized after the decode (k_ht_size).  Documents whose ids are dense
  // (actors x highest counter within a few times the row count: what writers produce) get a direct table
  // dn_val[dn_base[d] + actor * dn_stride[d] + ctr]; the others an open-addressing hash region [tot_ht]
  u32 ht_mode;                   // 0 = choose per document, 1 = hash tables only (tests)
  u32 *dn_base, *dn_stride;      // [n_docs + 1] / [n_docs]: stride = highest counter + 1, 0 = the document uses the hash region
  u32* dn_val;                   // row, AMH_NULL32 = empty
  u64* ht_key;                   // (ctr << 32 | actor) + 1, 0 = empty
  u32* ht_val;
  // synthesized deletes live in the slots of a second table [tot_dt]
  u64* dt_key;
  u32 *d_doc, *d_first_pred_row;
  // unified op space t in [0, tot_rows + tot_dt)
  u32 *t_pred_cnt, *t_pred_base, *t_pred_fill, *t_chg;
  u64* pred_ids;                 // [tot_succ] (ctr << 32 | actor)
  u32* ops_sorted;               // [tot_ops] ops grouped by change in id order
  u32* j_pred_base;              // [tot_ops + 1] base of each slot's pred list in pred_ids (change order)
  // change view: per-op fields in change (slot) order, gathered once for the re-encode kernels.  The fields of a
  // column group are interleaved per op: cv[0] = {0, 1} x tot_ops, cv[2] = {2..5}, cv[6] = {6, 7}, cv[8] = {8..11}
  // (the other entries of cv are unused):
  // 0 obj actor (change-local, NULL32 = root)  1 obj ctr  2 key actor (NULL32 unless elem)  3 key ctr (NULL32 = prop)
  // 4 key str off  5 key str len (NULL32 unless prop)  6 insert | expand << 1 | action << 8  7 pred count
  // 8 value meta lo  9 value meta hi | kind << 8  10, 11 raw offset + length, or the i64 of a re-encoded increment
  u32* cv[12];
  u8* out_arena;                 // re-derived change chunks
  u8* enc_tmp;                   // one-pass change re-encode (amh_segenc1.inl): a slot per (change, column), written right-aligned
  i32* sv_status;                // [n_docs] save: 0 ok, else why the document has no saved form
  u32* sv_col_len;               // [n_docs * 25]
  u32* sv_wide;                  // [n_docs] bit k: column k holds a literal run of more than 64 values (left to k_save_store)
  u32* sv_body_len;              // [n_docs]
  u32* sv_off;                   // [n_docs + 1] chunk offsets in sv_out
  u8* sv_out;
  // save of documents with 64k rows and more: their op columns are cut into ranges of sv_range rows, a wave
  // each (g_sv_big).  A "big column" is one op column (9..24) of one such document; a job is one range of it.
  // The change re-encode of a load uses the same tables for changes of tens of thousands of ops (g_en_big): a big
  // column is then one of the 14 columns of such a change, svb_doc names the change and sv_big is per change.
  u32 sv_nj, sv_nbc, sv_range;
  const u32 *svj_bc, *svj_k;                            // [sv_nj] big column of the job, its range index
  const u32 *svb_doc, *svb_col, *svb_first, *svb_nr;    // [sv_nbc] document, column, first job, ranges
  u32 *svj_sum, *svj_cin;                               // [9 * sv_nj] what the range does to the carry (ScSummary); [3 * sv_nj] the carry it receives
  u32 *svj_total, *svj_any, *svj_tail;                  // [sv_nj] bytes of the range, a non-null (true) value seen, bytes of the ranges behind it
  const u8* sv_big;                                     // [n_docs] 1 = the op columns are encoded by g_sv_big
  u32* sync_off;                 // [n_docs + 1] record offsets of the sync summaries
  u8* sync_out;
  u32 sync_cap;                  // 0 = sync_out holds every record; else its capacity: records that end beyond it are not written
  u32* tx_off;                   // [tot_rows + 1] amh_text_batch: bytes every row contributes to its document's text, then their exclusive sums
  u32* tx_doc;                   // [n_docs + 1] where every document's text starts in tx_out
  u8* tx_out;
  // lists too long for the thread that owns them (more than AMH_BIG_LIST entries): the thread-per-item kernel records
  // them here and a block per list sorts them afterwards (amh_kernels.hip, g_big_*)
  const u8* expect_hash;         // [tot_chg * 32] merge / apply: the hashes the changes of a merged document came with (k_verify_hash)
  u32* first_len;                // [n_docs] bytes of the document chunk an input starts with (0: it starts with a change chunk; the input's
                                 // length: nothing readable behind it) — what the host needs to know to find trailing change chunks
  // A column of one document over many waves (device form; amh_segdec.inl): column streams of AMH_BIGCOL bytes and more
  // are cut into segments of AMH_SEG bytes.  bc_* = per big column, sg_* = per segment.  bc_n = nullptr: none in this batch.
  u32* bc_n;                     // [4] big columns found, segments, exit-table entries
  u32 bc_cap_cols, bc_cap_segs, bc_cap_bytes;
  u32* bc_index;                 // [n_docs * 13] entry of (document, decode slot) in bc_*, AMH_NULL32 = a small column
  u32 *bc_doc, *bc_slot, *bc_seg0, *bc_nseg, *bc_ebase, *bc_bad;
  u64 *bc_rows, *bc_acc;         // totals over the column: rows, running sum
  u32 *sg_col, *sg_first, *sg_second, *sg_ntok, *sg_entry, *sg_pos, *sg_end, *sg_btok;
  u64 *sg_rows, *sg_acc;         // per segment, then the exclusive prefix inside the column
  u32* sg_exit;                  // [bc_cap_bytes] per token of a segment: (tokens to skip in the following segments << 1) | the first of them is a run's value
  // runs of AMH_BIGRUN rows and more met by the column decoders are not expanded by the wave that decodes them: they are
  // recorded here and filled by the whole grid afterwards (g_fill_runs).  fr_n = nullptr: expand in place.
  u32* fr_n;                     // [1]
  u32* fr_items;                 // [fr_cap * 8] column, document, first row (absolute), rows, a, b, c0, step
  u32 fr_cap;
  u32 skip_doc_sha;              // chunks written by merge / apply for the engine's own reload: no checksum is written or checked
  u32 max_doc_rows;              // per-document limit on op rows / succ entries / changes / deps (amh_ctx_set_limits; default AMH_MAX_ROWS_PER_DOC)
  u32* big_n;                    // [1] recorded lists
  u32* big_items;                // [big_cap] op t of a long pred list
  u32 big_cap;
};

// ---------------------------------------------------------------------------------------------
// merge: dst[p_a[p]].merge(src[p_b[p]]) for P pairs (automerge.rs:917-940).  D, S = the two loaded
// batches, M = a scratch batch of P merged documents in "loaded" form (rows in document order, succ
// lists, change metadata) over a virtual chunk = [dst chunk | src chunk]; the save kernels turn M into
// document chunks, which are loaded again (so every merged document is re-verified end to end).
// The host decides which changes are new and in which order they apply (O(changes) replay of
// automerge.rs:797-858, 1311-1334) and unions the actor tables; everything per op is on the device.
// ---------------------------------------------------------------------------------------------
// change chunk header (storage/change.rs:99-156); offsets are relative to the start of the chunk
struct ChgHdr {
  u32 deps_off, n_deps;            // 32-byte hashes
  u32 actor_off, actor_len;
  u32 msg_off, msg_len;            // len 0 = no message
  u32 others_off, n_other;         // length-prefixed actor ids
  u32 extra_off, extra_len;
  u64 seq, start_op;
  i64 time;
  u32 mark_len;                    // length of the chunk's mark-name column (0 = none): all the host needs of its descriptor
  u32 head_len;                    // bytes in front of the op column metadata (all the host reads of a device-resident chunk)
};
// apply_changes (automerge.rs:785-914): the change chunks of all pairs are staged as one batch U whose
// "documents" are the chunks (same column decoders as the document path; the pred group sits in the succ
// slots, there are no id columns).  S is then a view of U's rows per pair, and the merge kernels run with
// new rows = the ops of the applied changes in document order (k_ap_order).
struct ApplyDev {
  ChgHdr* hdr;                     // [n_chunks]
  const u32* c_pair;               // [n_chunks] pair the chunk belongs to
  const u32* c_rel;                // [n_chunks] offset of the chunk inside its pair's region of the arena
  const u8* c_applied;             // [n_chunks] host decision: the change is applied by this call
  const u32* amap_base;            // [n_chunks + 1]
  const u32* amap;                 // change-local actor index -> index in the pair's actor table of S
  const u32* ix_base;              // [P + 1] applied changes of a pair sorted by (actor, start_op): id -> row of U
  const u32 *ix_actor, *ix_start, *ix_nops, *ix_row;
  u32* u_chg;                      // [U rows] chunk of the row
  u32 *gap, *anchor, *parent;      // [U rows] dst-local gap (AMH_NULL32 = inside a new subtree), dst row it hangs off, parent row in U
  u32* p_tgt;                      // [U preds] bit 31 set: dst row (global), else row of U; AMH_NULL32 = none
  u32 *o_next, *o_head_upd, *o_head_ins;   // [U rows] sibling / same-gap list links, heads of a row's update and child-element lists
  u32 *o_top, *o_tmp, *o_stack;            // [U rows] rows in front of a top row within a long same-gap list, weight of a top row, scratch of the long-list sorts / place of a top row among the new rows
  u32* o_up;                       // [U rows] elements: the row behind the element's subtree (AP_UP_*)
  u64* o_rank;                     // [U rows] {a row in front of this one in its subtree's preorder, distance to it}; ranked: {top row, place behind it}
  u32* g_head;                     // [gap counters] top rows sharing a gap
  u32* flag;                       // [1] a pointer moved in this k_ap_up_jump / k_ap_rank_jump launch
  u32* u_fill;                     // [U rows]
};
struct MergeDev {
  BatchDev D, S, M;
  BatchDev U;                      // apply only
  ApplyDev A;
  u32 n_pairs;
  const u32 *p_a, *p_b;            // [P] document index in D / S
  i32* p_status;                   // [P] 0 = ok (host pre-sets failures; kernels may add AMH_E_UNSUPPORTED)
  const u32 *d_pair, *s_pair;      // [D.n_docs] / [S.n_docs] pair of the document, or AMH_NULL32
  const u32 *mapa_base, *mapb_base;// [P + 1] bases into map_a / map_b (+ map_b2a)
  const u32 *map_a, *map_b;        // document-local actor index -> index in the merged (sorted) table
  const u32* map_b2a;              // src actor index -> dst actor index, or AMH_NULL32
  const u8* s_added;               // [S.tot_chg] the change is applied by this merge
  const u32* a_len_pad;            // [P] dst chunk length rounded up to 16 = offset of the src chunk in the virtual chunk
  const u32* mc_from;              // [M.tot_chg] source of each merged change: bit 31 = from S, low bits = global change index
  // new rows (rows of S whose change is added), compacted in S order
  u32* new_idx;                    // [S.tot_rows + 1] flag, then exclusive scan
  u32 tot_new;
  u32 *new_src, *new_gap;          // [tot_new] S row; dst-local row index the row is inserted before (AMH_NULL32 = inherit)
  u32* gap_base;                   // [P + 1] per pair region of rows(dst) + 1 counters
  u32* gap_cnt;                    // counters, then exclusive scan
  // dst sequence structure
  u32 *sub_end, *upd_end;          // [D.tot_rows] (document-local indices) end of an element's subtree / of its own update rows
  u32* stack;                      // [D.tot_rows] scratch of k_mg_subtree; device form: parent position + 1 of an element row (0 = HEAD or
                                   // the first row of an object, AMH_NULL32 = not an element row)
  u32 *sub_min1, *sub_min2;        // [D.tot_rows / 64 + 2], [D.tot_rows / 4096 + 2] minimum of `stack` per tile (device form)
  u32* x_cnt;                      // [D.tot_rows] succ entries a dst row gains
  u32* x_fill;                     // [D.tot_rows]
  u32* d_to_m;                     // [D.tot_rows] merged row of a dst row
  // lists of more than AMH_BIG_LIST entries (children of one new element, top rows sharing a gap, succ lists): recorded
  // by the thread-per-row kernels, then sorted by a block each (amh_kernels.hip, g_big_mg).  big_n[kind] = lists
  // recorded, big_n[4 + kind] = scratch taken from A.o_stack; items of kind k start at big_items + 2 * k * big_cap.
  u32* big_n;                      // [8]
  u32* big_items;                  // [4 * 2 * big_cap] pairs (a, b)
  u32 big_cap;
};
// generate_sync_message over a batch of (document, peer) records (sync.rs:134-221).  Every hash list of a record's State sits
// in one pool; list l of record r = pool entries [r_list[r * SM_LISTS + l], r_list[r * SM_LISTS + l + 1]).
enum { SM_SHARED = 0, SM_LAST_SENT, SM_THEIR_HEADS, SM_THEIR_NEED, SM_SENT, SM_QDEPS, SM_QHASH, SM_LAST_SYNC, SM_LISTS };
enum { SMF_THEIR_HEADS = 1, SMF_THEIR_NEED = 2, SMF_THEIR_HAVE = 4, SMF_IN_FLIGHT = 8 };
enum { SMK_NONE = 0, SMK_MESSAGE = 1, SMK_RESET = 2, SMK_FAILED = 3 };
struct SyncDev {
  BatchDev B;
  u32 n_rec, n_pool;
  u32 raw;                         // 1 = amh_save_after_batch: per record the changes since the heads in SM_LAST_SYNC, chunks back to back, no framing
  const u32 *r_doc, *r_flags;
  const u8* pool;                  // [n_pool * 32]
  const u32* pool_rec;             // [n_pool] record of the hash
  u32* pool_idx;                   // [n_pool] place of the hash in its document's history, AMH_NULL32 = not there (k_sm_lookup)
  const u32* r_list;               // [n_rec * SM_LISTS + 1]; SM_LAST_SYNC spans the last_sync hashes of all the record's Haves
  const u32* r_have;               // [n_rec + 1] Haves of a record
  const u32* h_ls;                 // [2 * n_have] last_sync of a Have: pool range {first, end}
  const u32 *h_entries, *h_probes; // [n_have] Bloom filter parameters (sync/bloom.rs:13-18)
  const u64* h_bits;               // [n_have] its bit array: offset in blooms
  const u32* h_nbytes;             // [n_have] bytes of it
  const u8* blooms;
  // scratch, one region per record
  const u64 *r_flag_base, *r_clock_base, *r_word_base;   // [n_rec]
  u8* c_flag;                      // [sum n_chg] SMC_* bits per (record, change)
  u64* clock;                      // [sum n_actors]
  u32* words;                      // per record: heads[n_chg] | need[n_qdeps + n_their_heads] | send[n_their_need + n_chg] | spos[same]
  // results
  u32 *r_kind, *r_nheads, *r_nneed, *r_nsend, *r_bloom_n;   // [n_rec]; r_bloom_n = AMH_NULL32: no Have in the message
  u32* r_bits_at;                  // [n_rec] where the Bloom filter's bits start in msg_out (0 = none)
  u32* msg_off;                    // [n_rec + 1] message sizes, then their exclusive scan
  u32* job_base;                   // [n_rec + 1] r_nsend, scanned: places in the copy job tables and in sent_out
  u8* msg_out;
  u8* sent_out;                    // [jobs * 32] hashes of the changes in the messages
  u64 *job_src, *job_dst;          // [jobs] copy jobs of the change chunks (amh_copy_bytes64)
  u32* job_len;
};
enum { SMC_ANC_SHARED = 1, SMC_ANC_SYNC = 2, SMC_SENT = 4, SMC_SEND = 8, SMC_SINCE = 16, SMC_BLOOM = 32 };
enum { BIGK_CHILDREN = 0, BIGK_TIES = 1, BIGK_SUCC_U = 2, BIGK_SUCC_M = 3 };
#define AMH_BIG_LIST 48u
