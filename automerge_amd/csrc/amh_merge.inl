// Per-thread bodies of the merge kernels (see MergeDev in amh_types.h).  Included after
// amh_kernels.inl.  References: automerge.rs:917-940 (merge), :1258-1280 (insert_op),
// query/opid.rs:136-184 (SimpleOpIdSearch: RGA position), op_tree.rs:212-239, 393-408,
// op_set.rs:194-203 (add_succ), types.rs:517-521 (lamport_cmp).
//
// The merged op set is the union of both documents' ops.  Both row lists are already in document
// order, and the relative order of two ops never depends on other ops, so the merge is a merge by
// rank: every new row (a row of S whose change D lacks) finds the dst row it goes in front of, and
// rows shift by the number of new rows in front of them.  Map rows rank by binary search on
// (object, key bytes, lamport id).  Sequence rows use the insert tree: a new element whose parent D
// knows goes after the parent's children with a greater id (query/opid.rs:170-183); a new element
// whose parent is itself new stays behind its parent (its whole subtree is a contiguous run of S).
#pragma once
#include "amh_kernels.inl"

AMH_DEV int mg_lam(u32 c1, u32 r1, u32 c2, u32 r2) {   // ids with actors as merged-table indices (= bytewise order)
  if (c1 != c2) return c1 < c2 ? -1 : 1;
  return r1 < r2 ? -1 : (r1 > r2 ? 1 : 0);
}
AMH_DEV bool mg_pair_ok(const MergeDev& G, u32 p) { return G.p_status[p] == 0; }
AMH_DEV bool row_is_prop(const BatchDev& B, u32 r) { return B.r_key_len[r] != AMH_NULL32; }

// 1. flag the new rows of S
AMH_DEV void k_mg_new_flag(const MergeDev& G, u32 r) {
  const BatchDev& S = G.S;
  u32 b = S.r_doc[r];
  u32 f = 0;
  if (b < S.n_docs) {
    u32 p = G.s_pair[b];
    if (p != AMH_NULL32 && mg_pair_ok(G, p)) { u32 c = S.t_chg[r]; f = (c != AMH_EMPTY32 && G.s_added[c]) ? 1 : 0; }
  }
  G.new_idx[r] = f;
}
// 2. per pair sizes (rows of the merged document, counter region)
AMH_DEV void k_mg_pair_counts(const MergeDev& G, u32 p) {
  u32 a = G.p_a[p], b = G.p_b[p];
  bool ok = mg_pair_ok(G, p);
  u32 na = ok ? G.D.cnt[a].n_rows : 0, nb = ok ? G.S.cnt[b].n_rows : 0;
  u32 rb = G.S.row_base[b];
  u32 nn = ok ? G.new_idx[rb + nb] - G.new_idx[rb] : 0;
  G.M.row_base[p] = na + nn;
  G.gap_base[p] = ok ? na + 1 : 0;
}
// 3. subtree ends of the elements of the dst document's sequences (thread per pair; rows are in
//    preorder, so an element's subtree ends where the first row of a non-descendant begins)
AMH_DEV void k_mg_subtree(const MergeDev& G, u32 p) {
  if (!mg_pair_ok(G, p)) return;
  const BatchDev& D = G.D;
  u32 a = G.p_a[p], rb = D.row_base[a], n = D.cnt[a].n_rows;
  u32* st = G.stack + rb;
  u32 sp = 0, last_ins = AMH_NULL32;
  for (u32 i = 0; i <= n; i++) {
    bool obj_change = i == n || i == 0 || D.r_obj_ctr[rb + i] != D.r_obj_ctr[rb + i - 1] || D.r_obj_actor[rb + i] != D.r_obj_actor[rb + i - 1];
    if (obj_change) {
      while (sp) G.sub_end[rb + st[--sp]] = i;
      if (last_ins != AMH_NULL32) G.upd_end[rb + last_ins] = i;
      last_ins = AMH_NULL32;
      if (i == n) break;
    }
    u32 r = rb + i;
    G.sub_end[r] = i + 1;
    G.upd_end[r] = i + 1;
    if (row_is_prop(D, r) || !D.r_insert[r]) continue;
    if (last_ins != AMH_NULL32) G.upd_end[rb + last_ins] = i;
    last_ins = i;
    // parent = the element named by the key; every stacked element that is not the parent has ended
    u32 kc = D.r_key_ctr[r], ka = D.r_key_actor[r];
    if (kc == 0) { while (sp) G.sub_end[rb + st[--sp]] = i; }
    else {
      while (sp && !(D.r_id_ctr[rb + st[sp - 1]] == kc && D.r_id_actor[rb + st[sp - 1]] == ka)) G.sub_end[rb + st[--sp]] = i;
      if (!sp) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }   // an element whose parent is not an ancestor: not preorder
    }
    st[sp++] = i;
  }
}
// object id of a row of S in merged terms
AMH_DEV void mg_s_obj(const MergeDev& G, u32 p, u32 r, u32& ctr, u32& rank) {
  ctr = G.S.r_obj_ctr[r];
  rank = ctr ? G.map_b[G.mapb_base[p] + G.S.r_obj_actor[r]] : 0;
}
AMH_DEV void mg_d_obj(const MergeDev& G, u32 p, u32 r, u32& ctr, u32& rank) {
  ctr = G.D.r_obj_ctr[r];
  rank = ctr ? G.map_a[G.mapa_base[p] + G.D.r_obj_actor[r]] : 0;
}
// first dst-local index in [lo, hi) whose row does not sort before (object oc/orank, [key], id): lower bound
AMH_DEV u32 mg_lower_bound(const MergeDev& G, u32 p, u32 oc, u32 orank, bool with_key, const u8* key, u32 klen, u32 ic, u32 irank) {
  const BatchDev& D = G.D;
  u32 a = G.p_a[p], rb = D.row_base[a], lo = 0, hi = D.cnt[a].n_rows;
  const u8* chunk = D.arena + D.desc[a].chunk_off;
  while (lo < hi) {
    u32 mid = (lo + hi) / 2, r = rb + mid;
    u32 c, k;
    mg_d_obj(G, p, r, c, k);
    int cmp = mg_lam(c, k, oc, orank);
    if (cmp == 0 && with_key) {
      if (!row_is_prop(D, r)) cmp = 1;   // cannot happen inside one object; keep the search total
      else {
        cmp = amh_memcmp(chunk + D.r_key_off[r], D.r_key_len[r], key, klen);
        if (cmp == 0) cmp = mg_lam(D.r_id_ctr[r], G.map_a[G.mapa_base[p] + D.r_id_actor[r]], ic, irank);
      }
    } else if (cmp == 0) {
      cmp = 1;   // object-only search: the first row of the object
    }
    if (cmp < 0) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// dst row (global) of an op id given in S terms, or AMH_EMPTY32
AMH_DEV u32 mg_find_in_d(const MergeDev& G, u32 p, u32 ctr, u32 s_actor) {
  u32 da = G.map_b2a[G.mapb_base[p] + s_actor];
  if (da == AMH_NULL32) return AMH_EMPTY32;
  return ht_find(G.D, G.p_a[p], pack_id(ctr, da));
}
// 4. where each new row goes (thread per row of S)
AMH_DEV void k_mg_gap(const MergeDev& G, u32 r) {
  const BatchDev &S = G.S, &D = G.D;
  if (r + 1 > S.tot_rows || G.new_idx[r + 1] == G.new_idx[r]) return;   // not a new row
  u32 k = G.new_idx[r];
  u32 b = S.r_doc[r], p = G.s_pair[b], a = G.p_a[p];
  G.new_src[k] = r;
  u32 rbA = D.row_base[a], nA = D.cnt[a].n_rows;
  u32 oc, orank;
  mg_s_obj(G, p, r, oc, orank);
  u32 ic = S.r_id_ctr[r], irank = G.map_b[G.mapb_base[p] + S.r_id_actor[r]];
  const u8* chunkB = S.arena + S.desc[b].chunk_off;
  u32 gap;
  if (row_is_prop(S, r)) {
    gap = mg_lower_bound(G, p, oc, orank, true, chunkB + S.r_key_off[r], S.r_key_len[r], ic, irank);
  } else {
    // sequence row.  Is the object one D has?
    bool obj_known = oc == 0 || mg_find_in_d(G, p, oc, S.r_obj_actor[r]) != AMH_EMPTY32;
    if (!obj_known) {
      gap = mg_lower_bound(G, p, oc, orank, false, (const u8*)0, 0, 0, 0);
      // rows of a new object arrive in S order behind each other: same gap for all of them
    } else if (S.r_insert[r]) {
      u32 kc = S.r_key_ctr[r];
      u32 first, end;   // dst-local range of the parent's children
      bool known = true;
      if (kc == 0) {    // HEAD: the object's top-level elements
        first = mg_lower_bound(G, p, oc, orank, false, (const u8*)0, 0, 0, 0);
        end = first;
        while (end < nA) { u32 c, kk; mg_d_obj(G, p, rbA + end, c, kk); if (c != oc || kk != orank) break; end = G.sub_end[rbA + end] > end ? G.sub_end[rbA + end] : end + 1; }
      } else {
        u32 pr = mg_find_in_d(G, p, kc, S.r_key_actor[r]);
        if (pr == AMH_EMPTY32) known = false;
        else { first = G.upd_end[pr]; end = G.sub_end[pr]; }
      }
      if (!known) gap = AMH_NULL32;   // parent is new as well: stays behind it
      else {
        u32 c = first;
        while (c < end) {   // skip the children with a greater id (they, and their subtrees, come first)
          u32 cr = rbA + c;
          if (mg_lam(D.r_id_ctr[cr], G.map_a[G.mapa_base[p] + D.r_id_actor[cr]], ic, irank) < 0) break;
          c = G.sub_end[cr];
        }
        gap = c;
      }
    } else {
      // update of an element: among that element's update rows, ascending id (op_tree.rs:212-239)
      u32 er = mg_find_in_d(G, p, S.r_key_ctr[r], S.r_key_actor[r]);
      if (er == AMH_EMPTY32) gap = AMH_NULL32;
      else {
        u32 c = er - rbA + 1, end = G.upd_end[er];
        while (c < end && mg_lam(D.r_id_ctr[rbA + c], G.map_a[G.mapa_base[p] + D.r_id_actor[rbA + c]], ic, irank) < 0) c++;
        gap = c;
      }
    }
  }
  G.new_gap[k] = gap;
}
// 5. inherit gaps inside runs of new rows, check monotonicity, count rows per gap (thread per pair)
AMH_DEV void k_mg_inherit(const MergeDev& G, u32 p) {
  if (!mg_pair_ok(G, p)) return;
  u32 b = G.p_b[p], rb = G.S.row_base[b], nb = G.S.cnt[b].n_rows;
  u32 k0 = G.new_idx[rb], k1 = G.new_idx[rb + nb];
  u32* cnt = G.gap_cnt + G.gap_base[p];
  u32 prev = 0;
  bool have = false;
  for (u32 k = k0; k < k1; k++) {
    u32 g = G.new_gap[k];
    if (g == AMH_NULL32) {
      if (!have) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }
      g = prev;
      G.new_gap[k] = g;
    }
    if (have && g < prev) { G.p_status[p] = AMH_E_UNSUPPORTED; return; }   // would contradict S's own order
    prev = g; have = true;
    cnt[g]++;
  }
}
// 7. succ entries a dst row gains: entries of the same op in S made by a new op (op_set.rs:194-203)
AMH_DEV bool mg_extra_succ(const MergeDev& G, u32 e, u32& p, u32& drow) {
  const BatchDev& S = G.S;
  u32 r = S.s_row[e];
  if (r == AMH_EMPTY32) return false;
  u32 b = S.r_doc[r];
  if (b >= S.n_docs) return false;
  p = G.s_pair[b];
  if (p == AMH_NULL32 || !mg_pair_ok(G, p)) return false;
  if (G.new_idx[r + 1] != G.new_idx[r]) return false;   // a new row carries all its succ entries itself
  u32 t = S.s_target[e];
  if (t == AMH_EMPTY32) return false;
  u32 c = S.t_chg[t];
  if (c == AMH_EMPTY32 || !G.s_added[c]) return false;
  drow = mg_find_in_d(G, p, S.r_id_ctr[r], S.r_id_actor[r]);
  return drow != AMH_EMPTY32;
}
AMH_DEV void k_mg_extra_cnt(const MergeDev& G, u32 e) {
  u32 p, drow;
  if (mg_extra_succ(G, e, p, drow)) amh_atomic_add(&G.x_cnt[drow], 1);
}
// 8. rows of the merged documents
AMH_DEV void mg_copy_row(const MergeDev& G, u32 p, const BatchDev& X, u32 r, const u32* amap, u32 str_shift, u32 raw_col_off, u32 m, u32 succ_cnt) {
  const BatchDev& M = G.M;
  M.r_doc[m] = p;
  M.r_id_actor[m] = amap[X.r_id_actor[r]];
  M.r_id_ctr[m] = X.r_id_ctr[r];
  u32 oc = X.r_obj_ctr[r];
  M.r_obj_ctr[m] = oc;
  M.r_obj_actor[m] = oc ? amap[X.r_obj_actor[r]] : 0;
  bool prop = row_is_prop(X, r);
  M.r_key_len[m] = X.r_key_len[r];
  M.r_key_off[m] = prop ? X.r_key_off[r] + str_shift : 0;
  u32 kc = X.r_key_ctr[r];
  M.r_key_ctr[m] = kc;
  M.r_key_actor[m] = prop ? AMH_NULL32 : (kc ? amap[X.r_key_actor[r]] : 0);
  M.r_insert[m] = X.r_insert[r];
  M.r_action[m] = X.r_action[r];
  M.r_expand[m] = X.r_expand[r];
  M.r_vmeta[m] = X.r_vmeta[r];
  M.r_vraw_off[m] = raw_col_off + X.r_vraw_off[r] + str_shift;   // absolute inside the virtual chunk (M's raw column starts at 0)
  M.r_mark_len[m] = X.r_mark_len[r];
  M.r_mark_off[m] = X.r_mark_len[r] != AMH_NULL32 ? X.r_mark_off[r] + str_shift : 0;
  M.r_succ_cnt[m] = succ_cnt;
}
AMH_DEV void k_mg_rows_dst(const MergeDev& G, u32 r) {
  const BatchDev& D = G.D;
  u32 a = D.r_doc[r];
  if (a >= D.n_docs) return;
  u32 p = G.d_pair[a];
  if (p == AMH_NULL32 || !mg_pair_ok(G, p)) return;
  u32 i = r - D.row_base[a];
  const u32* pre = G.gap_cnt + G.gap_base[p];
  u32 m = G.M.row_base[p] + i + (pre[i + 1] - pre[0]);
  G.d_to_m[r] = m;
  mg_copy_row(G, p, D, r, G.map_a + G.mapa_base[p], 0, D.desc[a].ocol[OC_VAL_RAW].off, m, D.r_succ_cnt[r] + G.x_cnt[r]);
}
AMH_DEV void k_mg_rows_new(const MergeDev& G, u32 k) {
  const BatchDev& S = G.S;
  u32 r = G.new_src[k];
  if (r == AMH_NULL32) return;   // the pair was given up before its new rows were placed
  u32 b = S.r_doc[r], p = G.s_pair[b];
  if (!mg_pair_ok(G, p)) return;
  u32 k0 = G.new_idx[S.row_base[b]];
  u32 m = G.M.row_base[p] + G.new_gap[k] + (k - k0);
  G.new_src[k] = r;
  G.new_gap[k] = m;   // from here on: the merged row
  mg_copy_row(G, p, S, r, G.map_b + G.mapb_base[p], G.a_len_pad[p], S.desc[b].ocol[OC_VAL_RAW].off, m, S.r_succ_cnt[r]);
}
// 10. succ lists (M.r_succ_base is the scan of M.r_succ_cnt)
AMH_DEV void k_mg_succ_dst(const MergeDev& G, u32 r) {
  const BatchDev& D = G.D;
  u32 a = D.r_doc[r];
  if (a >= D.n_docs) return;
  u32 p = G.d_pair[a];
  if (p == AMH_NULL32 || !mg_pair_ok(G, p)) return;
  u32 m = G.d_to_m[r], o = G.M.r_succ_base[m], sb = D.r_succ_base[r], n = D.r_succ_cnt[r];
  const u32* amap = G.map_a + G.mapa_base[p];
  for (u32 k = 0; k < n; k++) { G.M.s_actor[o + k] = amap[D.s_actor[sb + k]]; G.M.s_ctr[o + k] = D.s_ctr[sb + k]; }
}
AMH_DEV void k_mg_succ_new(const MergeDev& G, u32 k) {
  const BatchDev& S = G.S;
  u32 r = G.new_src[k];
  if (r == AMH_NULL32) return;
  u32 p = G.s_pair[S.r_doc[r]];
  if (!mg_pair_ok(G, p)) return;
  u32 m = G.new_gap[k], o = G.M.r_succ_base[m], sb = S.r_succ_base[r], n = S.r_succ_cnt[r];
  const u32* amap = G.map_b + G.mapb_base[p];
  for (u32 q = 0; q < n; q++) { G.M.s_actor[o + q] = amap[S.s_actor[sb + q]]; G.M.s_ctr[o + q] = S.s_ctr[sb + q]; }
}
AMH_DEV void k_mg_succ_extra(const MergeDev& G, u32 e) {
  u32 p, drow;
  if (!mg_extra_succ(G, e, p, drow)) return;
  u32 m = G.d_to_m[drow];
  u32 o = G.M.r_succ_base[m] + G.D.r_succ_cnt[drow] + amh_atomic_add(&G.x_fill[drow], 1);
  G.M.s_actor[o] = G.map_b[G.mapb_base[p] + G.S.s_actor[e]];
  G.M.s_ctr[o] = G.S.s_ctr[e];
}
AMH_DEV void k_mg_succ_sort(const MergeDev& G, u32 r) {   // rows that gained entries: restore ascending lamport order (types/opids.rs:51-74)
  if (G.x_cnt[r] == 0) return;
  const BatchDev& D = G.D;
  u32 a = D.r_doc[r];
  if (a >= D.n_docs) return;
  u32 p = G.d_pair[a];
  if (p == AMH_NULL32 || !mg_pair_ok(G, p)) return;
  u32 m = G.d_to_m[r], o = G.M.r_succ_base[m], n = G.M.r_succ_cnt[m];
#ifndef AMH_EMU
  if (n > AMH_BIG_LIST) { big_push(G.big_n + BIGK_SUCC_M, G.big_items + 2 * (size_t)BIGK_SUCC_M * G.big_cap, G.big_cap, m, p); return; }
#endif
  for (u32 i = 1; i < n; i++) {
    u32 xa = G.M.s_actor[o + i], xc = G.M.s_ctr[o + i];
    u32 j = i;
    while (j > 0 && mg_lam(G.M.s_ctr[o + j - 1], G.M.s_actor[o + j - 1], xc, xa) > 0) {
      G.M.s_actor[o + j] = G.M.s_actor[o + j - 1]; G.M.s_ctr[o + j] = G.M.s_ctr[o + j - 1]; j--;
    }
    G.M.s_actor[o + j] = xa; G.M.s_ctr[o + j] = xc;
  }
}
// per merged document: succ segment
AMH_DEV void k_mg_doc_succ(const MergeDev& G, u32 p) {
  const BatchDev& M = G.M;
  u32 rb = M.row_base[p], re = M.row_base[p + 1];
  u32 lo = M.r_succ_base[rb], hi = M.r_succ_base[re];
  M.succ_base[p] = lo;
  M.cnt[p].n_succ = hi - lo;
  M.cnt[p].n_rows = re - rb;
}
// 11. change metadata of the merged documents (deps / is_dep come from the host)
AMH_DEV void k_mg_changes(const MergeDev& G, u32 c) {
  const BatchDev& M = G.M;
  u32 from = G.mc_from[c];
  bool from_s = (from >> 31) != 0;
  u32 x = from & 0x7FFFFFFFu;
  const BatchDev& X = from_s ? G.S : G.D;
  u32 p = M.c_doc[c];
  const u32* amap = from_s ? G.map_b + G.mapb_base[p] : G.map_a + G.mapa_base[p];
  u32 shift = from_s ? G.a_len_pad[p] : 0;
  M.c_actor[c] = amap[X.c_actor[x]];
  M.c_seq[c] = X.c_seq[x];
  M.c_max_op[c] = X.c_max_op[x];
  M.c_time[c] = X.c_time[x];
  M.c_msg_len[c] = X.c_msg_len[x];
  M.c_msg_off[c] = X.c_msg_len[x] != AMH_NULL32 ? X.c_msg_off[x] + shift : 0;
  M.c_extra_len[c] = X.c_extra_len[x];
  M.c_extra_off[c] = X.c_extra_off[x] + shift;
  for (int i = 0; i < 32; i++) M.c_hash[(u64)c * 32 + i] = X.c_hash[(u64)x * 32 + i];
}
// 12. the virtual chunk [dst chunk | src chunk] (thread per 16 bytes of a pair's region would need a search; thread per pair)
AMH_DEV void k_mg_copy_chunks(const MergeDev& G, u32 p) {
  if (!mg_pair_ok(G, p)) return;
  u32 a = G.p_a[p], b = G.p_b[p];
  u8* dst = const_cast<u8*>(G.M.arena) + G.M.desc[p].chunk_off;
  const u8* sa = G.D.arena + G.D.desc[a].chunk_off;
  const u8* sb = G.S.arena + G.S.desc[b].chunk_off;
  u32 la = G.D.desc[a].chunk_len, lb = G.S.desc[b].chunk_len;
  for (u32 i = 0; i < la; i++) dst[i] = sa[i];
  for (u32 i = 0; i < lb; i++) dst[G.a_len_pad[p] + i] = sb[i];
}
