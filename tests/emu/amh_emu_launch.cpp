// CPU-only test tier: runs the per-thread kernel bodies of automerge_amd/csrc/amh_kernels.inl in
// host loops so that the kernel logic can be checked against the oracle where no GPU exists.
// TEST INFRASTRUCTURE: built only by tests/emu/Makefile into tests/emu/libamh_emu_for_tests.so,
// never part of libautomerge_hip.so and never loaded by the automerge_amd package.
#define AMH_EMU
#include "../../automerge_amd/csrc/amh_kernels.inl"
#include "../../automerge_amd/csrc/amh_merge.inl"
#include "../../automerge_amd/csrc/amh_apply.inl"
#include "../../automerge_amd/csrc/amh_syncmsg.inl"
#include "../../automerge_amd/csrc/amh_launch.h"

void amh_launch(int k, const BatchDev& B, u32 n, void*) {
#define L(id, name) case id: for (u32 i = 0; i < n; i++) name(B, i); break;
  switch (k) {
    L(AK_PARSE, k_parse) L(AK_SHA_DOC, k_sha_doc) L(AK_COUNT, k_count) L(AK_EXTRACT, k_extract)
    L(AK_ACTORS, k_actors) L(AK_CHG_INIT, k_chg_init) L(AK_DECODE_CHG, k_decode_chg) L(AK_DECODE_CHG_COLS, k_decode_chg_cols) L(AK_CHG_PATCH, k_chg_patch) L(AK_ATAB_BASE, k_atab_base) L(AK_ATAB_FILL, k_atab_fill)
    L(AK_DECODE_OPS, k_decode_ops) L(AK_ROWS, k_rows) L(AK_OBJ_ORDER, k_obj_order) L(AK_SUCC, k_succ)
    L(AK_BUCKET, k_bucket) L(AK_CHG_FINISH, k_chg_finish) L(AK_PLACE, k_place) L(AK_PRED_SLOT_CNT, k_pred_slot_cnt) L(AK_PRED_SLOT_BASE, k_pred_slot_base) L(AK_PRED_FILL, k_pred_fill)
    L(AK_PRED_SORT, k_pred_sort) L(AK_ACTOR_MARK, k_actor_mark) L(AK_ACTOR_RANK, k_actor_rank) L(AK_CHANGE_VIEW, k_change_view)
    L(AK_ENCODE_SIZE, k_encode_size) L(AK_CHG_LAYOUT, k_chg_layout) L(AK_ENCODE_STORE, k_encode_store)
    L(AK_CHG_HEADER, k_chg_header) L(AK_CHG_HASH, k_chg_hash) L(AK_CHG_HASH_SEQ, k_chg_hash_seq) L(AK_HEADS, k_heads) L(AK_SYNC_SIZE, k_sync_size) L(AK_SYNC_FILL, k_sync_fill) L(AK_SAVE_CHECK, k_save_check) L(AK_SAVE_SIZE, k_save_size) L(AK_SAVE_LAYOUT, k_save_layout) L(AK_SAVE_STORE, k_save_store) L(AK_SAVE_FINISH, k_save_finish) L(AK_VISIBLE, k_visible) L(AK_HT_SIZE, k_ht_size) L(AK_DT_SIZE, k_dt_size) L(AK_SUCC_LINK, k_succ_link) L(AK_M_INIT, k_m_init) L(AK_ROWS_M, k_rows_m) L(AK_VERIFY_HASH, k_verify_hash)
    L(AK_TEXT_LEN, k_text_len) L(AK_TEXT_FILL, k_text_fill) L(AK_TEXT_DOC, k_text_doc)
    default: break;
  }
#undef L
}
u32 amh_scan_tmp_words(u32) { return 2; }
void amh_scan_inplace(u32* data, u32 n, u32*, void*) {
  u64 run = 0;
  for (u32 i = 0; i < n; i++) {
    u32 v = data[i];
    data[i] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)run;
    run += v;
  }
  data[n] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)run;
}
u32 amh_level_sort_tmp_words(u32) { return 2; }
void amh_level_sort(const BatchDev& B, u32*, void*) {
  u32 cnt[AMH_LEVEL_CAP + 2] = {0};
  for (u32 c = 0; c < B.tot_chg; c++) cnt[chg_level_bucket(B, c) + 1]++;
  for (u32 l = 0; l <= AMH_LEVEL_CAP; l++) cnt[l + 1] += cnt[l];
  for (u32 l = 0; l <= AMH_LEVEL_CAP + 1; l++) B.lvl_start[l] = cnt[l];
  for (u32 c = 0; c < B.tot_chg; c++) B.lvl_order[cnt[chg_level_bucket(B, c)]++] = c;
}
void amh_launch_mg(int k, const MergeDev* G, u32 n, void*) {
#define M(id, name) case id: for (u32 i = 0; i < n; i++) name(*G, i); break;
  switch (k) {
    M(MK_NEW_FLAG, k_mg_new_flag) M(MK_PAIR_COUNTS, k_mg_pair_counts) M(MK_SUBTREE, k_mg_subtree) M(MK_GAP, k_mg_gap)
    M(MK_INHERIT, k_mg_inherit) M(MK_EXTRA_CNT, k_mg_extra_cnt) M(MK_ROWS_DST, k_mg_rows_dst) M(MK_ROWS_NEW, k_mg_rows_new)
    M(MK_SUCC_DST, k_mg_succ_dst) M(MK_SUCC_NEW, k_mg_succ_new) M(MK_SUCC_EXTRA, k_mg_succ_extra) M(MK_SUCC_SORT, k_mg_succ_sort)
    M(MK_DOC_SUCC, k_mg_doc_succ) M(MK_CHANGES, k_mg_changes) M(MK_COPY_CHUNKS, k_mg_copy_chunks)
    M(MK_AP_PARSE, k_ap_parse) M(MK_AP_COUNT, k_ap_count) M(MK_AP_VERIFY, k_ap_verify) M(MK_AP_MAP, k_ap_map) M(MK_AP_CHG_META, k_ap_chg_meta) M(MK_AP_NEW_FLAG, k_ap_new_flag) M(MK_AP_LINK, k_ap_link) M(MK_AP_CHILDREN, k_ap_children) M(MK_AP_SORT_CHILDREN, k_ap_sort_children) M(MK_AP_SUCC_INIT, k_ap_succ_init) M(MK_AP_UP_JUMP, k_ap_up_jump) M(MK_AP_SUCC_BUILD, k_ap_succ_build) M(MK_AP_RANK_JUMP, k_ap_rank_jump) M(MK_AP_WEIGHT, k_ap_weight) M(MK_AP_TOP_K, k_ap_top_k) M(MK_AP_TOP_PLACE, k_ap_top_place) M(MK_AP_EMIT, k_ap_emit) M(MK_AP_PRED_CNT, k_ap_pred_cnt) M(MK_AP_SUCC_FILL_U, k_ap_succ_fill_u) M(MK_AP_SUCC_SORT_U, k_ap_succ_sort_u) M(MK_AP_SUCC_FILL_D, k_ap_succ_fill_d) M(MK_AP_TIES, k_ap_tie_find)
    default: break;
  }
#undef M
}
void amh_sm_doc_sizes(const BatchDev& B, const u32* doc, u32* out, u32 n, void*) {
  for (u32 r = 0; r < n; r++) {
    const u32 d = doc[r];
    const bool ok = B.status[d] == 0;
    out[2 * r] = ok ? B.cnt[d].n_chg : 0;
    out[2 * r + 1] = ok ? B.desc[d].n_actors : 0;
  }
}
void amh_ap_counts_given(const BatchDev& U, const u64* nops, const u64* npred, const u64* place, u32 n, void*) {
  for (u32 k = 0; k < n; k++) {
    const u32 c = (u32)(place[k] / 32), rows = *(const u32*)(uintptr_t)nops[k], preds = *(const u32*)(uintptr_t)npred[k];
    const bool ok = U.status[c] == 0;
    U.cnt[c].n_rows = ok ? rows : 0; U.cnt[c].n_succ = ok ? preds : 0;
    U.row_base[c] = ok ? rows : 0; U.succ_base[c] = ok ? preds : 0;
  }
}
void amh_launch_sm(int k, const SyncDev* G, u32 n, void*) {
  for (u32 i = 0; i < n; i++) {
    if (k == SK_LOOKUP) k_sm_lookup(*G, i);
    else if (k == SK_PLAN) k_sm_plan(*G, i);
    else if (k == SK_FILL) k_sm_fill(*G, i);
    else if (k == SK_BLOOM) k_sm_bloom(*G, i);
    else if (k == SK_JOBS) k_sm_jobs(*G, i);
  }
}
void amh_copy_bytes(const u64* src, const u32* len, const u32* dst_off, u8* dst, u32 n, void*) {
  for (u32 i = 0; i < n; i++) memcpy(dst + dst_off[i], (const u8*)(uintptr_t)src[i], len[i]);
}
void amh_copy_bytes64(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void*) {
  for (u32 i = 0; i < n; i++) memcpy(dst + dst_off[i], (const u8*)(uintptr_t)src[i], len[i]);
}
void amh_copy_segments(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void*) {
  for (u32 i = 0; i < n; i++) memcpy(dst + dst_off[i], (const u8*)(uintptr_t)src[i], len[i]);
}
void amh_actor_pack_len(const BatchDev& B, u32* len, void*) {
  for (u32 d = 0; d < B.n_docs; d++) {
    const u32 na = B.status[d] ? 0 : B.desc[d].n_actors, ab = B.actor_base[d];
    len[d] = na ? B.actor_off[ab + na - 1] + B.actor_len[ab + na - 1] - B.actor_off[ab] : 0;
  }
}
void amh_actor_pack_copy(const BatchDev& B, const u32* len, const u32* off, u8* dst, void*) {
  for (u32 d = 0; d < B.n_docs; d++)
    if (len[d]) memcpy(dst + off[d], B.arena + B.desc[d].chunk_off + B.actor_off[B.actor_base[d]], len[d]);
}
void amh_ap_head_len(const ChgHdr* hdr, const i32* status, u32* len16, u32 n, void*) {
  for (u32 u = 0; u < n; u++) len16[u] = status[u] ? 0u : (hdr[u].head_len + 15u) & ~15u;
}
void amh_ap_head_copy(const u8* arena, const u64* doc_off, const u32* len16, const u32* off, u8* dst, u32 n, void*) {
  for (u32 u = 0; u < n; u++) memcpy(dst + off[u], arena + doc_off[2 * (u64)u], len16[u]);
}

// Test hook: the register writer (FastWriter: 16-byte stores, in-place widening of long literal headers) against the
// byte-wise writer (ColWriter) over the same value sequence, for the three encoder kinds.  Returns 0 when the bytes
// agree.  vals[i] < 0 = null; `mod` folds the values so that runs and literal stretches of all lengths show up.
extern "C" int amh_emu_test_writers(const long long* vals, unsigned n, int kind, unsigned char* scratch_a, unsigned char* scratch_b, unsigned cap, unsigned* len_out) {
  ColWriter cw; cw.out = scratch_a; cw.pos = 0;
  FastWriter fw; fw_init(fw, scratch_b);
  SizeWriter sw; sw_init(sw);
  static const char* strs[8] = {"", "a", "bc", "k012", "hello", "some longer key", "\xc3\xa9", "zzzzzzzzzzzzzzzzzzzzzzzz"};
  auto run = [&](auto& w) {
    RleVal x; x.v = 0; x.s = 0; x.n = 0;
    if (kind == 0) { RleEnc<0> e; re_init(e); for (unsigned i = 0; i < n; i++) { if (vals[i] < 0) re_null(w, e); else { x.v = (u64)vals[i]; re_value(w, e, x); } } re_finish(w, e); }
    else if (kind == 1) { DeltaEnc e; re_init(e.r); e.abs = 0; for (unsigned i = 0; i < n; i++) { if (vals[i] < 0) re_null(w, e.r); else de_value(w, e, (i64)vals[i]); } re_finish(w, e.r); }
    else { RleEnc<2> e; re_init(e); for (unsigned i = 0; i < n; i++) { if (vals[i] < 0) re_null(w, e); else { RleVal sv; sv.v = 0; sv.s = (const u8*)strs[vals[i] & 7]; sv.n = (u32)strlen(strs[vals[i] & 7]); re_value(w, e, sv); } } re_finish(w, e); }
  };
  run(cw);
  run(sw);
  if (cw.pos > cap) return -1;
  run(fw);
  fw_finish(fw);
  *len_out = cw.pos;
  if (cw.pos != fw.pos || cw.pos != sw.pos) return 1;
  return memcmp(scratch_a, scratch_b, cw.pos) == 0 ? 0 : 2;
}
