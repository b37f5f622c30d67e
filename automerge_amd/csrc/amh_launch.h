// Launch interface between the host orchestration and the kernels.  amh_kernels.hip implements it
// with HIP launches on a stream; tests/emu implements it with host loops (CPU-only test tier).
#pragma once
#include "amh_types.h"

enum AmhKernel {
  AK_PARSE = 0, AK_SHA_DOC, AK_COUNT, AK_EXTRACT, AK_ACTORS, AK_CHG_INIT, AK_DECODE_CHG, AK_DECODE_OPS, AK_ROWS,
  AK_OBJ_ORDER, AK_SUCC, AK_BUCKET, AK_CHG_FINISH, AK_PLACE, AK_PRED_SLOT_CNT, AK_PRED_SLOT_BASE, AK_PRED_FILL, AK_PRED_SORT, AK_ACTOR_MARK,
  AK_ACTOR_RANK, AK_CHANGE_VIEW, AK_ENCODE_SIZE, AK_CHG_LAYOUT, AK_ENCODE_STORE, AK_CHG_HEADER, AK_CHG_HASH, AK_CHG_HASH_SEQ, AK_HEADS, AK_VISIBLE, AK_SYNC_SIZE, AK_SYNC_FILL, AK_SAVE_CHECK, AK_SAVE_SIZE, AK_SAVE_LAYOUT, AK_SAVE_STORE, AK_SAVE_FINISH, AK_HT_SIZE, AK_DT_SIZE, AK_SUCC_LINK, AK_SCAN, AK_M_INIT, AK_ROWS_M, AK_VERIFY_HASH, AK_DECODE_CHG_COLS, AK_CHG_PATCH, AK_ATAB_BASE, AK_ATAB_FILL, AK_TEXT_LEN, AK_TEXT_FILL, AK_TEXT_DOC, AK_NUM
};
static const char* const AMH_KERNEL_NAMES[AK_NUM] = {
  "parse", "sha_doc", "count", "extract", "actors", "chg_init", "decode_chg", "decode_ops", "rows",
  "obj_order", "succ", "bucket", "chg_finish", "place", "pred_slot_cnt", "pred_slot_base", "pred_fill", "pred_sort", "actor_mark",
  "actor_rank", "change_view", "encode_size", "chg_layout", "encode_store", "chg_header", "chg_hash", "chg_hash_seq", "heads", "visible", "sync_size", "sync_fill", "save_check", "save_size", "save_layout", "save_store", "save_finish", "ht_size", "dt_size", "succ_link", "scan", "m_init", "rows_m", "verify_hash", "decode_chg_cols", "chg_patch", "atab_base", "atab_fill", "text_len", "text_fill", "text_doc"};

// run kernel `k` over n threads
void amh_launch(int k, const BatchDev& B, u32 n, void* stream);
// in-place exclusive scan of data[0..n), total written to data[n] (saturating at 0xFFFFFFFF).
// tmp must hold amh_scan_tmp_words(n) u32 words.
void amh_scan_inplace(u32* data, u32 n, u32* tmp, void* stream);
u32 amh_scan_tmp_words(u32 n);
// group the changes by dependency level (chg_level_bucket): B.lvl_order, B.lvl_start[0 .. cap + 1].
// tmp must hold amh_level_sort_tmp_words(B.tot_chg) u32 words.
void amh_level_sort(const BatchDev& B, u32* tmp, void* stream);
u32 amh_level_sort_tmp_words(u32 tot_chg);

// merge kernels (amh_merge.inl): G is the MergeDev in device memory (host memory in the emulation)
enum MgKernel {
  MK_NEW_FLAG = 0, MK_PAIR_COUNTS, MK_SUBTREE, MK_GAP, MK_INHERIT, MK_EXTRA_CNT, MK_ROWS_DST, MK_ROWS_NEW, MK_SUCC_DST,
  MK_SUCC_NEW, MK_SUCC_EXTRA, MK_SUCC_SORT, MK_DOC_SUCC, MK_CHANGES, MK_COPY_CHUNKS,
  MK_AP_PARSE, MK_AP_COUNT, MK_AP_VERIFY, MK_AP_MAP, MK_AP_CHG_META, MK_AP_NEW_FLAG, MK_AP_LINK, MK_AP_CHILDREN, MK_AP_SORT_CHILDREN, MK_AP_SUCC_INIT, MK_AP_UP_JUMP, MK_AP_SUCC_BUILD, MK_AP_RANK_JUMP, MK_AP_WEIGHT, MK_AP_TOP_K, MK_AP_TOP_PLACE, MK_AP_EMIT, MK_AP_PRED_CNT,
  MK_AP_SUCC_FILL_U, MK_AP_SUCC_SORT_U, MK_AP_SUCC_FILL_D, MK_AP_TIES, MK_NUM
};
void amh_launch_mg(int k, const MergeDev* G, u32 n, void* stream);
// sync message kernels (amh_syncmsg.inl): G is the SyncDev in device memory (host memory in the emulation)
enum SmKernel { SK_LOOKUP = 0, SK_PLAN, SK_FILL, SK_COPY, SK_BLOOM, SK_JOBS, SK_NUM };   // SK_COPY: the change chunks into the messages (amh_copy_bytes64), timed under this id
void amh_launch_sm(int k, const SyncDev* G, u32 n, void* stream);
// out[2 r] = changes, out[2 r + 1] = actors of the document of record r (0, 0 for a failed document); doc[] and out[] in device memory
void amh_sm_doc_sizes(const BatchDev& B, const u32* doc, u32* out, u32 n, void* stream);
// copy n byte segments inside device memory: segment i = len[i] bytes from address src[i] to dst + dst_off[i]
// (16-byte aligned on both sides).  The three tables live in device memory.
void amh_copy_segments(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void* stream);
// the same for segments of any alignment and length (change chunks into an out_arena)
void amh_copy_bytes(const u64* src, const u32* len, const u32* dst_off, u8* dst, u32 n, void* stream);
void amh_copy_bytes64(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void* stream);   // 64-bit places
// columns encoded over many waves (device build only; nothing when B.sv_nj == 0).  Save of very large documents:
// phase 0 after AK_SAVE_SIZE (column lengths of the big columns), phase 1 after AK_SAVE_STORE (their bytes).  The
// change re-encode of very large changes: phase 2 after AK_ENCODE_SIZE, phase 3 after AK_ENCODE_STORE.
void amh_launch_save_big(const BatchDev& B, int phase, void* stream);
// The change re-encode of a load (device build): which form the batch takes.  Batches of small changes keep the lane-per-change
// form over the change view (k_change_view); larger changes are encoded in one pass straight from the op rows into
// per-column slots of B.enc_tmp (amh_segenc1.inl), which must hold amh_encode_tmp_bytes(B) bytes; the view is then only
// needed by changes whose columns are cut into ranges (B.sv_nj != 0).
bool amh_encode_one_pass(const BatchDev& B);
u64 amh_encode_tmp_bytes(const BatchDev& B);
// heads of staged change chunks: len16[u] = bytes in front of chunk u's column metadata rounded up to 16 (0 for failed
// chunks); the copy packs them at off[u] (exclusive sums of len16) of dst
// actor tables of a batch's documents packed for the host: len[d] = bytes from the first actor id of document d to the
// end of its last one; the copy puts them at off[d] (exclusive sums of len) of dst
void amh_actor_pack_len(const BatchDev& B, u32* len, void* stream);
void amh_actor_pack_copy(const BatchDev& B, const u32* len, const u32* off, u8* dst, void* stream);
// op / pred counts of chunks whose source batches know them: chunk u = place[k] / 32 of U gets *nops[k] rows and *npred[k]
// pred entries (U.cnt, and U.row_base / U.succ_base as k_ap_count leaves them for the scans)
void amh_ap_counts_given(const BatchDev& U, const u64* nops, const u64* npred, const u64* place, u32 n, void* stream);
void amh_ap_head_len(const ChgHdr* hdr, const i32* status, u32* len16, u32 n, void* stream);
void amh_ap_head_copy(const u8* arena, const u64* doc_off, const u32* len16, const u32* off, u8* dst, u32 n, void* stream);

// amh_sync_summaries_noalloc: size / scan / fill with the offsets and scan temporaries in a static device array
// (amh_kernels.hip).  0 = launched; 1 = device error; 2 = over the static limit.  HIP build only.
#define AMH_NOALLOC_MAX_DOCS (1u << 20)
int amh_sync_noalloc_run(const BatchDev& B, int device, u8* d_out, u32 cap, u32* d_meta, void* stream, const char** err);
