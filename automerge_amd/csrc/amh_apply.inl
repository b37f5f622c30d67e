// Per-thread bodies of the apply_changes kernels (ApplyDev / MergeDev in amh_types.h).  Included after
// amh_merge.inl.  References: storage/change.rs:99-156 (Change::parse_following_header), :173-198
// (verify_ops), change/change_op_columns.rs:361-393 (ChangeOpsIter), automerge.rs:829-914
// (apply_change / import_ops), :1258-1280 (insert_op), query/opid.rs:136-184, op_tree.rs:212-239,
// op_set.rs:194-203 (add_succ).
//
// Index spaces: c = change chunk of the call (a "document" of the batch U), r = row of U = one op of a
// change in change order (deletes included), e = pred entry of U.
#pragma once
#include "amh_merge.inl"

// ------------------------------------------------------------------------------------------
// A1  change chunk header + column layout.  Thread per chunk.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_ap_parse(const MergeDev& G, u32 c) {
  const BatchDev& U = G.U;
  DocDesc& D = U.desc[c];
  DocCounts& C = U.cnt[c];
  ChgHdr& H = G.A.hdr[c];
  C.n_rows = C.n_succ = C.n_chg = C.n_deps = 0;
  D.n_actors = D.n_heads = 0;
  for (int i = 0; i < CC_COUNT; i++) { D.ccol[i].off = 0; D.ccol[i].len = 0; }
  for (int i = 0; i < OC_COUNT; i++) { D.ocol[i].off = 0; D.ocol[i].len = 0; }
  H.deps_off = H.n_deps = H.actor_off = H.actor_len = H.msg_off = H.msg_len = H.others_off = H.n_other = H.extra_off = H.extra_len = 0;
  H.seq = H.start_op = 0; H.time = 0; H.mark_len = 0; H.head_len = 0;
  u64 start = U.doc_off[2 * (u64)c], avail = U.doc_off[2 * (u64)c + 1] - start;
  D.chunk_off = start;
  D.chunk_len = 0;
  D.data_off = D.data_len = 0;
  D.actors_off = D.heads_off = D.suffix_off = D.suffix_len = 0;
  U.row_base[c] = 0;
  U.succ_base[c] = 0;
  if (avail >= 0xFFFFFFF0ull) { amh_fail(U, c, AMH_E_TOO_LARGE); return; }
  const u8* p = U.arena + start;
  // the host splits the input at chunk boundaries and inflates compressed change chunks (chunk.rs:77-97),
  // so every chunk here is a whole type-1 chunk
  if (avail < 10 || p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83 || p[8] != 1) { amh_fail(U, c, AMH_E_PARSE); return; }
  u64 pos = 9, end = avail, len;
  if (rd_uleb_strict(p, pos, end, len) || len != end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
  D.data_off = (u32)pos;
  D.data_len = (u32)len;
  D.chunk_len = (u32)avail;
  u64 n;
  if (rd_uleb_strict(p, pos, end, n) || n > (end - pos) / 32) { amh_fail(U, c, AMH_E_PARSE); return; }
  H.deps_off = (u32)pos; H.n_deps = (u32)n;
  pos += 32 * n;
  if (rd_uleb_strict(p, pos, end, n) || n > end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
  H.actor_off = (u32)pos; H.actor_len = (u32)n;
  pos += n;
  if (rd_uleb_strict(p, pos, end, H.seq)) { amh_fail(U, c, AMH_E_PARSE); return; }
  if (rd_uleb_strict(p, pos, end, H.start_op) || H.start_op == 0) { amh_fail(U, c, AMH_E_PARSE); return; }   // NonZeroU64 (change.rs:120)
  if (rd_sleb_strict(p, pos, end, H.time)) { amh_fail(U, c, AMH_E_PARSE); return; }
  if (rd_uleb_strict(p, pos, end, n) || n > end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
  if (!amh_valid_utf8(p + pos, n)) { amh_fail(U, c, AMH_E_PARSE); return; }
  H.msg_off = (u32)pos; H.msg_len = (u32)n;
  pos += n;
  if (rd_uleb_strict(p, pos, end, n) || n > end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
  H.others_off = (u32)pos; H.n_other = (u32)n;
  for (u64 i = 0; i < n; i++) {
    u64 al;
    if (rd_uleb_strict(p, pos, end, al) || al > end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
    pos += al;
  }
  u64 meta_pos = pos, otot = 0;
  H.head_len = (u32)meta_pos;
  int rc = parse_col_meta<2>(p, pos, end, len, otot, (ColMapper*)0);
  if (rc) { amh_fail(U, c, rc == 2 ? AMH_E_UNSUPPORTED : AMH_E_PARSE); return; }
  if (otot > end - pos) { amh_fail(U, c, AMH_E_PARSE); return; }
  u64 odata = pos;
  pos += otot;
  H.extra_off = (u32)pos; H.extra_len = (u32)(end - pos);
  ColMapper m;
  u64 t2;
  cm_init(m, odata, otot, D.ocol);
  if (parse_col_meta<2>(p, meta_pos, end, len, t2, &m) || cm_finish<2>(m)) { amh_fail(U, c, AMH_E_PARSE); return; }
  H.mark_len = D.ocol[OC_MARK_NAME].len;
}

// ------------------------------------------------------------------------------------------
// A2  ops and pred entries per chunk: the items of the action column (change_op_columns.rs:357-359)
//     and the sum of the pred counts.  Thread per chunk.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_ap_count(const MergeDev& G, u32 c) {
  const BatchDev& U = G.U;
  U.cnt[c].n_rows = 0; U.cnt[c].n_succ = 0;
  U.row_base[c] = 0; U.succ_base[c] = 0;
  if (U.status[c]) return;
  const DocDesc& D = U.desc[c];
  const u8* p = U.arena + D.chunk_off;
  u64 v; u32 aux; bool isnull;
  RleDec r;
  rle_init(r, p, D.ocol[OC_ACTION].off, D.ocol[OC_ACTION].len);
  u64 rows = 0;
  while (r.pos < r.end) {   // run headers only
    i64 n;
    if (!rd_sleb_lenient(r.p, r.pos, r.end, n)) { amh_fail(U, c, AMH_E_READ_OP); return; }
    if (n > 0) {
      if (!rle_read_value<0>(r, v, aux)) { amh_fail(U, c, AMH_E_READ_OP); return; }
      rows += (u64)n;
    } else if (n < 0) {
      u64 k = (n == INT64_MIN) ? (u64)INT64_MAX : (u64)(-n);
      if (k > r.end - r.pos) { amh_fail(U, c, AMH_E_READ_OP); return; }
      for (u64 i = 0; i < k; i++)
        if (!rle_read_value<0>(r, v, aux)) { amh_fail(U, c, AMH_E_READ_OP); return; }
      rows += k;
    } else {
      u64 nulls;
      if (!rd_uleb_lenient(r.p, r.pos, r.end, nulls)) { amh_fail(U, c, AMH_E_READ_OP); return; }
      if (nulls) { amh_fail(U, c, AMH_E_READ_OP); return; }   // "unexpected null action"
    }
    if (rows > U.max_doc_rows) { amh_fail(U, c, AMH_E_TOO_LARGE); return; }
  }
  // OpId::new takes a u32 counter (types.rs:502-504)
  if (G.A.hdr[c].start_op + rows > 0xFFFFFFFEull) { amh_fail(U, c, AMH_E_READ_OP); return; }
  rle_init(r, p, D.ocol[OC_SUCC_NUM].off, D.ocol[OC_SUCC_NUM].len);
  u64 preds = 0;
  for (u64 i = 0; i < rows; i++) {
    int rc = rle_next<0>(r, v, aux, isnull);
    if (rc != 0 || isnull) { amh_fail(U, c, AMH_E_READ_OP); return; }
    preds += v;
    if (v > U.max_doc_rows || preds > U.max_doc_rows) { amh_fail(U, c, AMH_E_TOO_LARGE); return; }
  }
  U.cnt[c].n_rows = (u32)rows;
  U.cnt[c].n_succ = (u32)preds;
  U.row_base[c] = (u32)rows;
  U.succ_base[c] = (u32)preds;
}

// ------------------------------------------------------------------------------------------
// A3  row validation + normalisation (obj_id.rs:123-141, key.rs:146-187, value.rs:209-308,
//     types.rs:236-250), owning row of every pred entry.  Thread per row of U.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_ap_verify(const MergeDev& G, u32 r) {
  const BatchDev& U = G.U;
  u32 c = U.r_doc[r];
  if (c == AMH_NULL32) { c = find_doc(U.row_base, U.n_docs, r); U.r_doc[r] = c; }
  G.A.u_chg[r] = c;
  G.A.u_fill[r] = 0;
  if (U.status[c]) return;
  {
    u32 sc = U.r_succ_cnt[r], sb = U.r_succ_base[r];
    for (u32 k = 0; k < sc; k++) U.s_row[sb + k] = r;
  }
  const DocDesc& D = U.desc[c];
  u32 na = 1 + G.A.hdr[c].n_other;
  u32 oa = U.r_obj_actor[r], oc = U.r_obj_ctr[r];
  if (D.ocol[OC_OBJ_ACTOR].len == 0 && D.ocol[OC_OBJ_CTR].len == 0) { oa = 0; oc = 0; }
  else if (oa == AMH_NULL32 && oc == AMH_NULL32) { oa = 0; oc = 0; }
  else if (oa != AMH_NULL32 && oc != AMH_NULL32) { if (oc == 0) oa = 0; }
  else if (oc == 0) { oa = 0; }
  else { amh_fail(U, c, AMH_E_READ_OP); return; }
  if (oc != 0 && oa >= na) { amh_fail(U, c, AMH_E_MISSING_ACTOR); return; }
  U.r_obj_actor[r] = oa;
  U.r_obj_ctr[r] = oc;
  u32 ka = U.r_key_actor[r], kc = U.r_key_ctr[r], kl = U.r_key_len[r];
  bool an = ka == AMH_NULL32, cn = kc == AMH_NULL32, sn = kl == AMH_NULL32;
  if (an && cn && !sn) { /* map key */ }
  else if (an && !cn && kc == 0 && sn) { U.r_key_actor[r] = 0; }   // HEAD
  else if (!an && !cn && sn) {
    if (ka >= na) { amh_fail(U, c, AMH_E_MISSING_ACTOR); return; }
    if (kc == 0 && ka != 0) { amh_fail(U, c, AMH_E_UNSUPPORTED); return; }
    if (kc == 0) U.r_key_actor[r] = 0;
  } else { amh_fail(U, c, AMH_E_READ_OP); return; }
  u64 m = U.r_vmeta[r];
  u32 vt = (u32)(m & 15);
  u64 vl = m >> 4;
  const u8* p = U.arena + D.chunk_off;
  if (vt == 3 || vt == 4 || vt == 8 || vt == 9) {
    i64 tmp;
    if (!val_int(U, r, tmp)) { amh_fail(U, c, AMH_E_READ_OP); return; }
  } else if (vt == 5) {
    if (vl != 8) { amh_fail(U, c, AMH_E_READ_OP); return; }
  } else if (vt == 6) {
    if (!amh_valid_utf8(p + D.ocol[OC_VAL_RAW].off + U.r_vraw_off[r], vl)) { amh_fail(U, c, AMH_E_READ_OP); return; }
  }
  if (U.r_action[r] == 5 && vt != 3 && vt != 4) { amh_fail(U, c, AMH_E_INVALID_OP); return; }
  u32 sc = U.r_succ_cnt[r], sb = U.r_succ_base[r];
  for (u32 k = 0; k < sc; k++)
    if (U.s_actor[sb + k] >= na) { amh_fail(U, c, AMH_E_MISSING_ACTOR); return; }
}

// ------------------------------------------------------------------------------------------
// A4  rows of the applied changes in the terms of their pair (import_ops, automerge.rs:860-914): op id
//     = start_op + i @ author, actor indices -> the pair's actor table of S, byte offsets -> relative to
//     the pair's region of the arena.  Thread per row of U.
// ------------------------------------------------------------------------------------------
AMH_DEV bool ap_row_live(const MergeDev& G, u32 r, u32& c, u32& p) {
  c = G.A.u_chg[r];
  if (!G.A.c_applied[c]) return false;
  p = G.A.c_pair[c];
  return mg_pair_ok(G, p);
}
AMH_DEV void k_ap_map(const MergeDev& G, u32 r) {
  const BatchDev& U = G.U;
  u32 c, p;
  G.A.gap[r] = AMH_NULL32; G.A.anchor[r] = AMH_NULL32; G.A.parent[r] = AMH_NULL32;
  G.A.o_next[r] = AMH_NULL32; G.A.o_head_upd[r] = AMH_NULL32; G.A.o_head_ins[r] = AMH_NULL32;
  G.S.r_succ_cnt[r] = 0;
  if (!ap_row_live(G, r, c, p)) { U.r_doc[r] = G.A.c_pair[G.A.u_chg[r]]; G.new_idx[r] = 0; return; }
  G.new_idx[r] = U.r_action[r] != 3 ? 1 : 0;   // k_ap_new_flag's verdict, while the row is at hand (one pass less over U)
  const u32* amap = G.A.amap + G.A.amap_base[c];
  u32 rel = G.A.c_rel[c];
  U.r_id_actor[r] = amap[0];
  U.r_id_ctr[r] = (u32)G.A.hdr[c].start_op + (r - U.row_base[c]);
  if (U.r_obj_ctr[r] != 0) U.r_obj_actor[r] = amap[U.r_obj_actor[r]];
  if (U.r_key_len[r] != AMH_NULL32) U.r_key_off[r] += rel;
  else if (U.r_key_ctr[r] != 0) U.r_key_actor[r] = amap[U.r_key_actor[r]];
  if (U.r_mark_len[r] != AMH_NULL32) U.r_mark_off[r] += rel;
  U.r_vraw_off[r] += rel + U.desc[c].ocol[OC_VAL_RAW].off;
  u32 sc = U.r_succ_cnt[r], sb = U.r_succ_base[r];
  for (u32 k = 0; k < sc; k++) U.s_actor[sb + k] = amap[U.s_actor[sb + k]];
  U.r_doc[r] = p;   // from here on the row belongs to S's document p
}
// change metadata of S (k_mg_changes reads it).  Thread per chunk.
AMH_DEV void k_ap_chg_meta(const MergeDev& G, u32 c) {
  const BatchDev &U = G.U, &S = G.S;
  const ChgHdr& H = G.A.hdr[c];
  u32 rel = G.A.c_rel[c];
  S.c_actor[c] = G.A.c_applied[c] ? G.A.amap[G.A.amap_base[c]] : 0;
  S.c_seq[c] = H.seq;
  S.c_max_op[c] = H.start_op + U.cnt[c].n_rows - 1;
  S.c_time[c] = H.time;
  S.c_msg_len[c] = H.msg_len ? H.msg_len : AMH_NULL32;   // an empty message is no message (change.rs:137-141)
  S.c_msg_off[c] = H.msg_len ? rel + H.msg_off : 0;
  S.c_extra_len[c] = H.extra_len;
  S.c_extra_off[c] = rel + H.extra_off;
  for (int i = 0; i < 32; i++) S.c_hash[(u64)c * 32 + i] = U.doc_hash[(u64)c * 32 + i];
}
// new rows of the merged documents: the ops of the applied changes except deletes, which only add succ
// entries (automerge.rs:1273-1278).  Thread per row of U.
AMH_DEV void k_ap_new_flag(const MergeDev& G, u32 r) {
  u32 c, p;
  G.new_idx[r] = (ap_row_live(G, r, c, p) && G.U.r_action[r] != 3) ? 1 : 0;
}

// row of U holding op (ctr, actor in S terms) of pair p, or AMH_EMPTY32
AMH_DEV u32 ap_find_in_u(const MergeDev& G, u32 p, u32 ctr, u32 s_actor) {
  const ApplyDev& A = G.A;
  u32 b0 = A.ix_base[p], lo = b0, hi = A.ix_base[p + 1];
  while (lo < hi) {   // first entry after (s_actor, ctr)
    u32 mid = (lo + hi) / 2;
    if (A.ix_actor[mid] < s_actor || (A.ix_actor[mid] == s_actor && A.ix_start[mid] <= ctr)) lo = mid + 1; else hi = mid;
  }
  if (lo == b0) return AMH_EMPTY32;
  u32 e = lo - 1;
  if (A.ix_actor[e] != s_actor || ctr - A.ix_start[e] >= A.ix_nops[e]) return AMH_EMPTY32;
  return A.ix_row[e] + (ctr - A.ix_start[e]);
}

// ------------------------------------------------------------------------------------------
// A5  where each new op goes relative to the dst rows (same rules as k_mg_gap), or which new op it
//     hangs off.  Thread per row of U.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_ap_link(const MergeDev& G, u32 r) {
  const BatchDev &S = G.S, &D = G.D;
  u32 c, p;
  if (!ap_row_live(G, r, c, p) || S.r_action[r] == 3) return;
  u32 a = G.p_a[p], rbA = D.row_base[a], nA = D.cnt[a].n_rows;
  u32 oc, orank;
  mg_s_obj(G, p, r, oc, orank);
  u32 ic = S.r_id_ctr[r], irank = G.map_b[G.mapb_base[p] + S.r_id_actor[r]];
  bool obj_known = oc == 0 || mg_find_in_d(G, p, oc, S.r_obj_actor[r]) != AMH_EMPTY32;
  if (!obj_known) {
    u32 mk = ap_find_in_u(G, p, oc, S.r_obj_actor[r]);
    if (mk == AMH_EMPTY32 || !is_make(S.r_action[mk])) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }   // an op on an object nobody made
  }
  const u8* region = S.arena + S.desc[p].chunk_off;
  u32 gap = AMH_NULL32, anchor = AMH_NULL32, parent = AMH_NULL32;
  if (row_is_prop(S, r)) {
    gap = mg_lower_bound(G, p, oc, orank, true, region + S.r_key_off[r], S.r_key_len[r], ic, irank);
  } else if (!obj_known) {
    u32 kc = S.r_key_ctr[r];
    if (kc == 0) {
      if (!S.r_insert[r]) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
      gap = mg_lower_bound(G, p, oc, orank, false, (const u8*)0, 0, 0, 0);
    } else {
      parent = ap_find_in_u(G, p, kc, S.r_key_actor[r]);
      if (parent == AMH_EMPTY32) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
    }
  } else if (S.r_insert[r]) {
    u32 kc = S.r_key_ctr[r];
    u32 first = 0, end = 0;
    bool known = true;
    if (kc == 0) {
      first = mg_lower_bound(G, p, oc, orank, false, (const u8*)0, 0, 0, 0);
      end = first;
      while (end < nA) { u32 cc, kk; mg_d_obj(G, p, rbA + end, cc, kk); if (cc != oc || kk != orank) break; end = G.sub_end[rbA + end] > end ? G.sub_end[rbA + end] : end + 1; }
    } else {
      u32 pr = mg_find_in_d(G, p, kc, S.r_key_actor[r]);
      if (pr == AMH_EMPTY32) known = false;
      else { anchor = pr - rbA; first = G.upd_end[pr]; end = G.sub_end[pr]; }
    }
    if (!known) {
      parent = ap_find_in_u(G, p, kc, S.r_key_actor[r]);
      if (parent == AMH_EMPTY32) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }   // the element it goes behind does not exist
    } else {
      u32 q = first;
      while (q < end) {
        u32 cr = rbA + q;
        if (mg_lam(D.r_id_ctr[cr], G.map_a[G.mapa_base[p] + D.r_id_actor[cr]], ic, irank) < 0) break;
        q = G.sub_end[cr];
      }
      gap = q;
    }
  } else {
    u32 kc = S.r_key_ctr[r];
    if (kc == 0) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
    u32 er = mg_find_in_d(G, p, kc, S.r_key_actor[r]);
    if (er == AMH_EMPTY32) {
      parent = ap_find_in_u(G, p, kc, S.r_key_actor[r]);
      if (parent == AMH_EMPTY32) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
    } else {
      anchor = er - rbA;
      u32 q = er - rbA + 1, end = G.upd_end[er];
      while (q < end && mg_lam(D.r_id_ctr[rbA + q], G.map_a[G.mapa_base[p] + D.r_id_actor[rbA + q]], ic, irank) < 0) q++;
      gap = q;
    }
  }
  if (parent != AMH_NULL32 && (S.r_action[parent] == 3 || !S.r_insert[parent])) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
  G.A.gap[r] = gap; G.A.anchor[r] = anchor; G.A.parent[r] = parent;
}

// ------------------------------------------------------------------------------------------
// A6  document order of the new rows of a pair.  Rows that hang off another new row form subtrees
//     (an element: its updates in ascending id, then its child elements in descending id, each with its
//     subtree — query/opid.rs:170-183, op_tree.rs:212-239); the remaining "top" rows are sorted by where
//     they go among the dst rows.  All per row: child lists by lock-free push + per-parent ordering, subtree
//     weights by pointer jumping, a counting sort of the top rows on their gap (ties ranked pairwise), and one
//     thread per top row that writes its subtree in preorder.
// ------------------------------------------------------------------------------------------
AMH_DEV int ap_id_cmp(const MergeDev& G, u32 p, u32 x, u32 y) {
  const BatchDev& S = G.S;
  const u32* mb = G.map_b + G.mapb_base[p];
  return mg_lam(S.r_id_ctr[x], mb[S.r_id_actor[x]], S.r_id_ctr[y], mb[S.r_id_actor[y]]);
}
AMH_DEV bool ap_top_less(const MergeDev& G, u32 p, u32 x, u32 y) {
  const BatchDev& S = G.S;
  u32 xc, xr, yc, yr;
  mg_s_obj(G, p, x, xc, xr);
  mg_s_obj(G, p, y, yc, yr);
  int c = mg_lam(xc, xr, yc, yr);
  if (c) return c < 0;
  bool xp = row_is_prop(S, x), yp = row_is_prop(S, y);
  if (xp != yp) return xp;
  if (xp) {
    const u8* region = S.arena + S.desc[p].chunk_off;
    c = amh_memcmp(region + S.r_key_off[x], S.r_key_len[x], region + S.r_key_off[y], S.r_key_len[y]);
    if (c) return c < 0;
    return ap_id_cmp(G, p, x, y) < 0;
  }
  u32 gx = G.A.gap[x], gy = G.A.gap[y];
  if (gx != gy) return gx < gy;
  // same gap: the row hanging off the deeper dst row comes first (its anchor is the later row); rows that
  // hang off the object itself (HEAD, a new object) have no anchor and come last
  u32 ax = G.A.anchor[x] + 1, ay = G.A.anchor[y] + 1;   // AMH_NULL32 + 1 = 0
  if (ax != ay) return ax > ay;
  bool xi = S.r_insert[x] != 0, yi = S.r_insert[y] != 0;
  if (xi != yi) return !xi;   // updates of the element before its new children
  c = ap_id_cmp(G, p, x, y);
  return xi ? c > 0 : c < 0;
}
// A6a  child lists: every new row that hangs off a new row is pushed onto its parent's list (lock-free; the lists
//      are put in order by A6b).  Also the start of the pointer jumping of A6c.  Thread per row of U.
AMH_DEV bool ap_row_new(const MergeDev& G, u32 r) { return G.new_idx[r + 1] != G.new_idx[r]; }
AMH_DEV void k_ap_children(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  A.o_top[r] = AMH_NULL32;   // top rows of a long same-gap list: rows in front of it within the gap (k_ap_tie_find)
  A.o_tmp[r] = 0;            // weight
  if (!ap_row_new(G, r)) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 par = A.parent[r];
  if (par == AMH_NULL32) return;
  if (!ap_row_new(G, par)) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }   // hangs off a row that is not applied: cannot happen for causally ready changes
  u32* head = G.S.r_insert[r] ? &A.o_head_ins[par] : &A.o_head_upd[par];
  A.o_next[r] = amh_atomic_exch(head, r);
}
// A6b  order the two child lists of a row: updates ascending, child elements descending by lamport id (lists are
//      short: one or two entries except when many elements are typed behind the same new element).  Thread per row.
AMH_DEV u32 ap_sort_list(const MergeDev& G, u32 p, u32 head, bool desc) {
  const ApplyDev& A = G.A;
  u32 sorted = AMH_NULL32, cur = head;
  while (cur != AMH_NULL32) {
    u32 nx = A.o_next[cur];
    u32 prev = AMH_NULL32, q = sorted;
    while (q != AMH_NULL32) {
      int c = ap_id_cmp(G, p, q, cur);
      if (desc ? c < 0 : c > 0) break;
      prev = q; q = A.o_next[q];
    }
    A.o_next[cur] = q;
    if (prev == AMH_NULL32) sorted = cur; else A.o_next[prev] = cur;
    cur = nx;
  }
  return sorted;
}
AMH_DEV bool ap_list_longer(const MergeDev& G, u32 head, u32 limit) {
  u32 n = 0;
  for (u32 q = head; q != AMH_NULL32 && n <= limit; q = G.A.o_next[q]) n++;
  return n > limit;
}
AMH_DEV void k_ap_sort_children(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  u32 hu = A.o_head_upd[r], hi = A.o_head_ins[r];
  if (hu == AMH_NULL32 && hi == AMH_NULL32) return;
  u32 p = G.S.r_doc[r];
  if (p >= G.n_pairs || !mg_pair_ok(G, p)) return;
#ifndef AMH_EMU
  // a list of more than AMH_BIG_LIST entries is left to a block (g_big_mg<BIGK_CHILDREN>)
  if (ap_list_longer(G, hu, AMH_BIG_LIST)) { big_push(G.big_n + BIGK_CHILDREN, G.big_items + 2 * (size_t)BIGK_CHILDREN * G.big_cap, G.big_cap, r, 0); hu = AMH_NULL32; }
  if (ap_list_longer(G, hi, AMH_BIG_LIST)) { big_push(G.big_n + BIGK_CHILDREN, G.big_items + 2 * (size_t)BIGK_CHILDREN * G.big_cap, G.big_cap, r, 1); hi = AMH_NULL32; }
#endif
  if (hu != AMH_NULL32 && A.o_next[hu] != AMH_NULL32) A.o_head_upd[r] = ap_sort_list(G, p, hu, false);
  if (hi != AMH_NULL32 && A.o_next[hi] != AMH_NULL32) A.o_head_ins[r] = ap_sort_list(G, p, hi, true);
}
// A6c  The rows of a subtree in preorder form a list (an element, its updates, then its child elements each with its
//      subtree); every row finds the row behind it locally, except where a subtree ends: there the walk goes on at
//      the next sibling of the nearest ancestor that has one.  That "row behind the subtree" (o_up, elements only) is
//      resolved by pointer jumping over the chain of last children (k_ap_up_jump).  Every row then tells the row
//      behind it who is in front (k_ap_succ_build), and the lists are ranked towards their top rows by pointer jumping
//      (k_ap_rank_jump over o_rank = {a row in front, distance to it}: the distances double per launch, threads read
//      and write whole pairs, so any interleaving keeps "distance to the row named" true).  What is left in o_rank is
//      every row's top row and its place behind it; the last row of a list gives the top row its weight (k_ap_weight).
//      Thread per row of U; the host repeats the jump launches until no pointer moved.
#define AP_UP_END 0x7FFFFFFFu          // nothing behind the subtree (the subtree of a top row ends the list)
#define AP_UP_TAIL 0x7FFFFFFEu         // left by k_ap_succ_build: the row is the last of its list
#define AP_UP_OPEN 0x80000000u         // | parent: same as behind the parent's subtree, not yet looked up
#define AP_RANK_TOP 0x80000000u        // | row: the row named is the list's top row
AMH_DEV bool ap_row_elem(const BatchDev& S, u32 r) { return S.r_insert[r] && !row_is_prop(S, r); }
AMH_DEV void k_ap_succ_init(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  A.o_up[r] = AP_UP_END;
  A.o_rank[r] = (u64)(AP_RANK_TOP | r);
  if (!ap_row_new(G, r)) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p) || !ap_row_elem(G.S, r)) return;
  u32 par = A.parent[r];
  if (par == AMH_NULL32) return;
  u32 sib = A.o_next[r];
  A.o_up[r] = sib != AMH_NULL32 ? sib : (AP_UP_OPEN | par);
}
AMH_DEV void k_ap_up_jump(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  u32 u = A.o_up[r];
  if (!(u & AP_UP_OPEN)) return;
  for (int step = 0; step < 3 && (u & AP_UP_OPEN); step++) u = A.o_up[u & ~AP_UP_OPEN];   // a few jumps per launch: fewer passes over the rows
  A.o_up[r] = u;
  *A.flag = 1;
}
AMH_DEV void k_ap_succ_build(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  if (!ap_row_new(G, r)) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 s = AMH_NULL32, par = A.parent[r];
  if (ap_row_elem(G.S, r)) {
    s = A.o_head_upd[r];
    if (s == AMH_NULL32) s = A.o_head_ins[r];
    if (s == AMH_NULL32 && A.o_up[r] != AP_UP_END) s = A.o_up[r];
  } else if (par != AMH_NULL32) {   // an update of a new element: the next one, then the element's children, then what follows its subtree
    s = A.o_next[r];
    if (s == AMH_NULL32) s = A.o_head_ins[par];
    if (s == AMH_NULL32 && A.o_up[par] != AP_UP_END) s = A.o_up[par];
  }
  if (s != AMH_NULL32) A.o_rank[s] = (u64)r | (1ull << 32);
  else A.o_up[r] = AP_UP_TAIL;   // nobody else reads it: a row that ends a list has no children
}
AMH_DEV void k_ap_rank_jump(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  u64 me = amh_load64(&A.o_rank[r]);
  if ((u32)me & AP_RANK_TOP) return;
  for (int step = 0; step < 3 && !((u32)me & AP_RANK_TOP); step++) {   // a few jumps per launch: fewer passes over the rows
    const u64 nx = amh_load64(&A.o_rank[(u32)me]);
    me = (u64)(u32)nx | (((me >> 32) + (nx >> 32)) << 32);
  }
  amh_store64(&A.o_rank[r], me);
  if (!((u32)me & AP_RANK_TOP)) *A.flag = 1;
}
// A6d  weight of a top row = rows of its subtree, itself included.  Thread per row.
AMH_DEV void k_ap_weight(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  if (A.o_up[r] != AP_UP_TAIL) return;
  u64 me = A.o_rank[r];
  A.o_tmp[(u32)me & ~AP_RANK_TOP] = (u32)(me >> 32) + 1;
}
// A6e  top rows per gap: rows in front of a dst row (gap_cnt, scanned next) and the list of top rows sharing the gap.
// Where two top rows of one gap and one object go relative to each other, as one number (ap_top_less from the anchor
// on: deeper anchor first, updates before child elements, updates by ascending and elements by descending counter);
// equal numbers (same counter, different actors) and map rows are left to ap_top_less.  Kept in o_rank of the top row,
// which the ranking is done with by then.
#define AP_KEY_NONE (1ull << 63)
AMH_DEV u64 ap_top_key(const MergeDev& G, u32 r) {
  const BatchDev& S = G.S;
  if (row_is_prop(S, r)) return AP_KEY_NONE;
  const u32 ax = G.A.anchor[r] + 1;   // AMH_NULL32 + 1 = 0
  if (ax > 0x0FFFFFFFu) return AP_KEY_NONE;
  const bool xi = S.r_insert[r] != 0;
  const u32 ctr = S.r_id_ctr[r];
  return ((u64)(0x0FFFFFFFu - ax) << 33) | ((u64)(xi ? 1 : 0) << 32) | (u64)(xi ? ~ctr : ctr);
}
AMH_DEV void k_ap_top_place(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  if (!ap_row_new(G, r) || A.parent[r] != AMH_NULL32) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 g = A.gap[r];
  if (g == AMH_NULL32) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
  u32 slot = G.gap_base[p] + g;
  amh_atomic_add(&G.gap_cnt[slot], A.o_tmp[r]);
  A.o_next[r] = amh_atomic_exch(&A.g_head[slot], r);
  A.o_rank[r] = ap_top_key(G, r);
}
// A6e'  gaps shared by more than AMH_BIG_LIST top rows: the list's head row records the gap; a block then sorts its
//       top rows and leaves, in o_top of each (unused for a top row otherwise), the rows that go in front of it within
//       the gap (g_big_mg<BIGK_TIES>).  Thread per row of U.
AMH_DEV void k_ap_tie_find(const MergeDev& G, u32 r) {
#ifndef AMH_EMU
  const ApplyDev& A = G.A;
  if (!ap_row_new(G, r) || A.parent[r] != AMH_NULL32) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 g = A.gap[r];
  if (g == AMH_NULL32) return;
  u32 slot = G.gap_base[p] + g;
  if (A.g_head[slot] != r || !ap_list_longer(G, r, AMH_BIG_LIST)) return;
  big_push(G.big_n + BIGK_TIES, G.big_items + 2 * (size_t)BIGK_TIES * G.big_cap, G.big_cap, slot, p);
#else
  (void)G; (void)r;
#endif
}
// A6f  where a top row goes: behind the new rows of smaller gaps (scanned gap_cnt) and behind the top rows of its own
//      gap that sort in front of it (with their subtrees).  Thread per top row.
AMH_DEV void k_ap_top_k(const MergeDev& G, u32 r) {
  const BatchDev& S = G.S;
  const ApplyDev& A = G.A;
  if (!ap_row_new(G, r) || A.parent[r] != AMH_NULL32) return;
  u32 p = S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 g = A.gap[r];
  const u32* pre = G.gap_cnt + G.gap_base[p];
  u32 k = G.new_idx[S.row_base[p]] + (pre[g] - pre[0]);
  if (A.o_top[r] != AMH_NULL32) k += A.o_top[r];   // a long list: ranked by g_big_mg<BIGK_TIES>
  else {
    u32 rc, rr;
    mg_s_obj(G, p, r, rc, rr);
    const u64 kr = A.o_rank[r];
    for (u32 m = A.g_head[G.gap_base[p] + g]; m != AMH_NULL32; m = A.o_next[m]) {
      if (m == r) continue;
      const u64 km = A.o_rank[m];
      bool less;
      if (((km | kr) >> 63) == 0) {
        u32 mc, mr;
        mg_s_obj(G, p, m, mc, mr);
        const int c = mg_lam(mc, mr, rc, rr);
        less = c ? c < 0 : (km != kr ? km < kr : ap_top_less(G, p, m, r));
      } else less = ap_top_less(G, p, m, r);
      if (less) k += A.o_tmp[m];
    }
  }
  A.o_stack[r] = k;
}
// A6g  emit: every new row goes behind its top row by its place in the subtree's preorder list.  Thread per row of U.
AMH_DEV void k_ap_emit(const MergeDev& G, u32 r) {
  const ApplyDev& A = G.A;
  if (!ap_row_new(G, r)) return;
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  const bool top = A.parent[r] == AMH_NULL32;   // (o_rank of a top row holds its ordering key by now)
  const u64 me = top ? 0 : A.o_rank[r];
  const u32 t = top ? r : ((u32)me & ~AP_RANK_TOP);
  const u32 k = A.o_stack[t] + (u32)(me >> 32);
  G.new_src[k] = r; G.new_gap[k] = A.gap[t];
}

// ------------------------------------------------------------------------------------------
// A7  succ entries: every op listed in a new op's pred gains the new op's id (op_set.rs:194-203;
//     the reference only looks inside the new op's key group, op_tree.rs:212-239 — for well-formed
//     changes that is where the preds are).  Thread per row of U (count), per pred entry (fill).
// ------------------------------------------------------------------------------------------
AMH_DEV void k_ap_pred_cnt(const MergeDev& G, u32 r) {
  const BatchDev& U = G.U;
  u32 c, p;
  u32 sc = U.r_succ_cnt[r], sb = U.r_succ_base[r];
  if (U.status[G.A.u_chg[r]]) return;   // the chunk did not decode: its counts are not to be trusted
  if (!ap_row_live(G, r, c, p)) { for (u32 k = 0; k < sc; k++) G.A.p_tgt[sb + k] = AMH_NULL32; return; }
  for (u32 k = 0; k < sc; k++) {
    u32 pa = U.s_actor[sb + k], pc = U.s_ctr[sb + k];
    u32 tgt = AMH_NULL32;
    u32 t = ap_find_in_u(G, p, pc, pa);
    if (t != AMH_EMPTY32) {
      if (G.new_idx[t + 1] != G.new_idx[t]) { amh_atomic_add(&G.S.r_succ_cnt[t], 1); tgt = t; }
    } else {
      u32 dr = mg_find_in_d(G, p, pc, pa);
      if (dr != AMH_EMPTY32) { amh_atomic_add(&G.x_cnt[dr], 1); tgt = 0x80000000u | dr; }
    }
    G.A.p_tgt[sb + k] = tgt;
  }
}
AMH_DEV void k_ap_succ_fill_u(const MergeDev& G, u32 e) {
  u32 tgt = G.A.p_tgt[e];
  if (tgt == AMH_NULL32 || (tgt & 0x80000000u)) return;
  u32 r = G.U.s_row[e];
  if (!mg_pair_ok(G, G.S.r_doc[r])) return;
  u32 o = G.S.r_succ_base[tgt] + amh_atomic_add(&G.A.u_fill[tgt], 1);
  G.S.s_actor[o] = G.S.r_id_actor[r];
  G.S.s_ctr[o] = G.S.r_id_ctr[r];
}
AMH_DEV void k_ap_succ_sort_u(const MergeDev& G, u32 r) {   // ascending lamport order (types/opids.rs:51-74)
  const BatchDev& S = G.S;
  u32 n = S.r_succ_cnt[r];
  if (n < 2) return;
  u32 p = S.r_doc[r];
  if (p >= G.n_pairs || !mg_pair_ok(G, p)) return;
  const u32* mb = G.map_b + G.mapb_base[p];
  u32 o = S.r_succ_base[r];
#ifndef AMH_EMU
  if (n > AMH_BIG_LIST) { big_push(G.big_n + BIGK_SUCC_U, G.big_items + 2 * (size_t)BIGK_SUCC_U * G.big_cap, G.big_cap, r, p); return; }
#endif
  for (u32 i = 1; i < n; i++) {
    u32 xa = S.s_actor[o + i], xc = S.s_ctr[o + i];
    u32 j = i;
    while (j > 0 && mg_lam(S.s_ctr[o + j - 1], mb[S.s_actor[o + j - 1]], xc, mb[xa]) > 0) {
      S.s_actor[o + j] = S.s_actor[o + j - 1]; S.s_ctr[o + j] = S.s_ctr[o + j - 1]; j--;
    }
    S.s_actor[o + j] = xa; S.s_ctr[o + j] = xc;
  }
}
AMH_DEV void k_ap_succ_fill_d(const MergeDev& G, u32 e) {
  u32 tgt = G.A.p_tgt[e];
  if (tgt == AMH_NULL32 || !(tgt & 0x80000000u)) return;
  u32 drow = tgt & 0x7FFFFFFFu;
  u32 r = G.U.s_row[e];
  u32 p = G.S.r_doc[r];
  if (!mg_pair_ok(G, p)) return;
  u32 m = G.d_to_m[drow];
  u32 o = G.M.r_succ_base[m] + G.D.r_succ_cnt[drow] + amh_atomic_add(&G.x_fill[drow], 1);
  G.M.s_actor[o] = G.map_b[G.mapb_base[p] + G.S.r_id_actor[r]];
  G.M.s_ctr[o] = G.S.r_id_ctr[r];
}
