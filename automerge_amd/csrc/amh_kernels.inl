// Per-thread bodies of the batched document-load kernels.  One function = one kernel; the thread's
// global index is passed in.  Included by amh_kernels.hip (HIP launch wrappers) and by the host
// emulation used in the CPU-only test tier.  Reference citations are relative to
// /root/reference/rust/automerge/src.
//
// Index spaces:  d = document, c = change (global), r = op row (global, document order),
// e = succ entry (global), t = op (t < tot_rows: row r = t; otherwise synthesized delete in slot
// t - tot_rows of the delete table), a = actor slot (global).
#pragma once
#include "amh_dev.h"

#define AMH_EMPTY32 0xFFFFFFFFu
#define AMH_NCOLS 14  // change op columns in write order (change_op_columns.rs:253-329)

AMH_DEV u64 amh_mix64(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
AMH_DEV u64 pack_id(u32 ctr, u32 actor) { return ((u64)ctr << 32) | actor; }

// id -> row lookup in the document's table region (direct or open addressing). AMH_EMPTY32 = absent.
AMH_DEV u32 ht_find(const BatchDev& B, u32 d, u64 id) {
  u32 stride = B.dn_stride[d];
  if (stride) {
    u32 ctr = (u32)(id >> 32), actor = (u32)id;
    if (ctr >= stride || actor >= B.desc[d].n_actors) return AMH_EMPTY32;
    return B.dn_val[B.dn_base[d] + actor * stride + ctr];
  }
  u64 base = B.ht_base[d], size = B.ht_base[d + 1] - base;
  if (size == 0) return AMH_EMPTY32;
  u64 key = id + 1, mask = size - 1, h = amh_mix64(key) & mask;
  for (u64 i = 0; i < size; i++) {
    u64 k = B.ht_key[base + h];
    if (k == key) return B.ht_val[base + h];
    if (k == 0) return AMH_EMPTY32;
    h = (h + 1) & mask;
  }
  return AMH_EMPTY32;
}

// ------------------------------------------------------------------------------------------
// K1a  chunk header + document prefix parse — chunk.rs:223-266, document.rs:113-154,
//      raw_column.rs:205-272, columns.rs:134-355, doc_op_columns.rs:398-504,
//      doc_change_columns.rs:272-338.  One thread per document.
// ------------------------------------------------------------------------------------------
// The column layout state machine of columns.rs:186-311, fed one raw column at a time (no per-thread
// arrays: the metadata block is walked twice instead).  Out-of-line on purpose: it keeps k_parse's
// control flow small (see AMH_DEV_NOINLINE).
struct ColMapper {
  int st;            // 0 ready, 1 in value, 2 in group
  u32 cur_id;
  u64 cur_s, cur_e, cur_rs, cur_re;
  u32 nkids, kid0_type, kid1_type;
  u64 kid0_s, kid0_e, kid1_s, kid1_e;
  bool g_in_value, bad;
  u64 off, data_rel_off, data_len;
  ColRef* slots;
};
AMH_DEV void cm_set(ColMapper& m, int slot, u64 s, u64 e) {
  m.slots[slot].off = (u32)(m.data_rel_off + s);
  m.slots[slot].len = (u32)(e - s);
}
template <int KIND>
AMH_DEV void cm_emit_simple(ColMapper& m, u32 id, u32 type, u64 s, u64 e) {
  // KIND: 0 = change metadata columns of a document, 1 = op columns of a document, 2 = op columns of a change
  // chunk (change_op_columns.rs:422-523: no id columns, the group is pred = 7)
  constexpr bool IS_OPS = KIND != 0;
  int slot = -1;
  if (IS_OPS) {
    if (id == 0 && type == 1) slot = OC_OBJ_ACTOR;
    else if (id == 0 && type == 2) slot = OC_OBJ_CTR;
    else if (id == 1 && type == 1) slot = OC_KEY_ACTOR;
    else if (id == 1 && type == 3) slot = OC_KEY_CTR;
    else if (id == 1 && type == 5) slot = OC_KEY_STR;
    else if (KIND == 1 && id == 2 && type == 1) slot = OC_ID_ACTOR;
    else if (KIND == 1 && id == 2 && type == 3) slot = OC_ID_CTR;
    else if (id == 3 && type == 4) slot = OC_INSERT;
    else if (id == 4 && type == 2) slot = OC_ACTION;
    else if (id == 9 && type == 4) slot = OC_EXPAND;
    else if (id == 10 && type == 5) slot = OC_MARK_NAME;
  } else {
    if (id == 0 && type == 1) slot = CC_ACTOR;
    else if (id == 0 && type == 3) slot = CC_SEQ;
    else if (id == 1 && type == 3) slot = CC_MAX_OP;
    else if (id == 2 && type == 3) slot = CC_TIME;
    else if (id == 3 && type == 5) slot = CC_MESSAGE;
  }
  if (slot >= 0) cm_set(m, slot, s, e);
}
template <int KIND>
AMH_DEV void cm_emit_value(ColMapper& m) {
  constexpr bool IS_OPS = KIND != 0;
  if (m.cur_id == 5) {
    cm_set(m, IS_OPS ? (int)OC_VAL_META : (int)CC_EXTRA_META, m.cur_s, m.cur_e);
    cm_set(m, IS_OPS ? (int)OC_VAL_RAW : (int)CC_EXTRA_RAW, m.cur_rs, m.cur_re);
  }
}
template <int KIND>
AMH_DEV void cm_emit_group(ColMapper& m) {
  constexpr bool IS_OPS = KIND != 0;
  if (IS_OPS && m.cur_id == (KIND == 2 ? 7u : 8u)) {
    cm_set(m, OC_SUCC_NUM, m.cur_s, m.cur_e);
    if (m.nkids == 0) { cm_set(m, OC_SUCC_ACTOR, 0, 0); cm_set(m, OC_SUCC_CTR, 0, 0); }
    else if (m.nkids == 2 && (m.kid0_type == 1 || m.kid0_type == 2) && m.kid1_type == 3) {
      cm_set(m, OC_SUCC_ACTOR, m.kid0_s, m.kid0_e);
      cm_set(m, OC_SUCC_CTR, m.kid1_s, m.kid1_e);
    } else m.bad = true;  // MismatchingColumn
  } else if (!IS_OPS && m.cur_id == 4) {
    cm_set(m, CC_DEPS_NUM, m.cur_s, m.cur_e);
    if (m.nkids == 0) cm_set(m, CC_DEPS_IDX, 0, 0);
    else if (m.nkids == 1 && m.kid0_type == 3) cm_set(m, CC_DEPS_IDX, m.kid0_s, m.kid0_e);
    else m.bad = true;
  }
}
AMH_DEV void cm_add_kid(ColMapper& m, u32 type, u64 s, u64 e) {
  if (m.nkids == 0) { m.kid0_type = type; m.kid0_s = s; m.kid0_e = e; }
  else if (m.nkids == 1) { m.kid1_type = type; m.kid1_s = s; m.kid1_e = e; }
  m.nkids++;
}
AMH_DEV void cm_init(ColMapper& m, u64 data_rel_off, u64 data_len, ColRef* slots) {
  m.st = 0; m.cur_id = 0; m.cur_s = m.cur_e = m.cur_rs = m.cur_re = 0;
  m.nkids = 0; m.kid0_type = m.kid1_type = 0; m.kid0_s = m.kid0_e = m.kid1_s = m.kid1_e = 0;
  m.g_in_value = false; m.bad = false; m.off = 0; m.data_rel_off = data_rel_off; m.data_len = data_len; m.slots = slots;
}
// one raw column; 1 = layout error
template <int KIND>
AMH_DEV_NOINLINE int cm_feed(ColMapper& m, u32 spec, u64 len) {
  u32 id = spec >> 4, type = spec & 7;
  u64 s = m.off, e = m.off + len;
  if (e > m.data_len || e < s) return 1;  // DataOutOfRange
  m.off = e;
  for (int pass = 0; pass < 2; pass++) {   // a column that closes a value / group is looked at again in the ready state
    if (m.st == 0) {
      if (type == 0) { m.st = 2; m.cur_id = id; m.cur_s = s; m.cur_e = e; m.nkids = 0; m.g_in_value = false; }
      else if (type == 6) { m.st = 1; m.cur_id = id; m.cur_s = s; m.cur_e = e; m.cur_rs = m.cur_re = 0; }
      else if (type == 7) return 1;  // LoneRawValueColumn
      else cm_emit_simple<KIND>(m, id, type, s, e);
      return 0;
    }
    if (m.st == 1) {
      if (type == 7) {
        if (m.cur_id != id) return 1;  // MismatchingValueMetadataId
        m.cur_rs = s; m.cur_re = e;
        cm_emit_value<KIND>(m);
        m.st = 0;
        return 0;
      }
      cm_emit_value<KIND>(m);
      m.st = 0;
      continue;
    }
    // in group
    if (m.cur_id != id) { cm_emit_group<KIND>(m); m.st = 0; continue; }
    if (!m.g_in_value) {
      if (type == 0 || type == 7) return 1;  // NestedGroup / LoneRawValueColumn
      if (type == 6) m.g_in_value = true;    // a value column inside a group: not a shape this path maps
      else cm_add_kid(m, type, s, e);
      return 0;
    }
    m.g_in_value = false;
    cm_add_kid(m, 6, 0, 0);
    if (type == 7) return 0;
    // the raw half is missing: the column is a fresh child of the same group
    if (type == 0) return 1;
    if (type == 6) m.g_in_value = true;
    else cm_add_kid(m, type, s, e);
    return 0;
  }
  return 0;
}
template <int KIND>
AMH_DEV_NOINLINE int cm_finish(ColMapper& m) {
  if (m.st == 1) cm_emit_value<KIND>(m);
  else if (m.st == 2) { if (m.g_in_value) cm_add_kid(m, 6, 0, 0); cm_emit_group<KIND>(m); }
  return m.bad ? 1 : 0;
}

// one column-metadata block (raw_column.rs:205-272): count, then (spec, len) pairs in normal order.
// mapper == null: validate and total the lengths; otherwise feed the layout machine.
// 0 ok, 1 parse error, 2 too many columns for this engine
template <int KIND>
AMH_DEV_NOINLINE int parse_col_meta(const u8* p, u64& pos, u64 end, u64 chunk_data_len, u64& total, ColMapper* mapper) {
  u64 nc;
  if (rd_uleb_strict(p, pos, end, nc)) return 1;
  if (nc > chunk_data_len) return 1;
  if (nc > 4096) return 2;
  u32 prev_norm = 0;
  total = 0;
  for (u64 i = 0; i < nc; i++) {
    u64 spec, clen;
    if (rd_uleb_strict(p, pos, end, spec)) return 1;
    if (spec > 0xFFFFFFFFull) return 1;
    if (rd_uleb_strict(p, pos, end, clen)) return 1;
    u32 norm = (u32)spec & ~8u;
    if (i > 0 && norm < prev_norm) return 1;  // NotInNormalOrder
    prev_norm = norm;
    // ingest inflates deflated columns on the host (compression.rs:51-74) before upload
    if (spec & 8) return 1;  // a column that ingest could not inflate (ParseError::Deflate)
    if (clen > chunk_data_len) return 1;
    total += clen;
    if (mapper && cm_feed<KIND>(*mapper, (u32)spec, clen)) return 1;
  }
  return 0;
}

AMH_DEV void k_parse(const BatchDev& B, u32 d) {
  DocDesc& D = B.desc[d];
  DocCounts& C = B.cnt[d];
  C.n_rows = C.n_succ = C.n_chg = C.n_deps = 0;
  D.n_actors = D.n_heads = 0;
  for (int i = 0; i < CC_COUNT; i++) { D.ccol[i].off = 0; D.ccol[i].len = 0; }
  for (int i = 0; i < OC_COUNT; i++) { D.ocol[i].off = 0; D.ocol[i].len = 0; }
  u64 start = B.doc_off[2 * (u64)d], avail = B.doc_off[2 * (u64)d + 1] - start;  // [start, end) pairs: chunk starts are 16-byte aligned
  D.chunk_off = start;
  D.chunk_len = 0;
  D.data_off = D.data_len = 0;
  D.actors_off = D.heads_off = D.suffix_off = D.suffix_len = 0;
  if (B.first_len) {   // where the first chunk ends, leniently (the host's rule for "is there anything behind it")
    u64 fl = avail;
    const u8* q = B.arena + start;
    if (avail >= 10 && avail < 0xFFFFFFF0ull && q[0] == 0x85 && q[1] == 0x6f && q[2] == 0x4a && q[3] == 0x83) {
      if (q[8] == 1 || q[8] == 2) fl = 0;
      else if (q[8] == 0) {
        u64 l = 0, ps = 9; u32 shift = 0; bool good = false;
        while (ps < avail && shift <= 63) { u8 bb = q[ps++]; l |= (u64)(bb & 0x7f) << shift; if (!(bb & 0x80)) { good = true; break; } shift += 7; }
        if (good && l <= avail - ps) fl = ps + l;
      }
    }
    B.first_len[d] = (u32)(fl > 0xFFFFFFFFull ? 0xFFFFFFFFull : fl);
  }
  if (avail == 0) return;  // empty input = empty document (automerge.rs:605-608)
  if (avail >= 0xFFFFFFF0ull) { amh_fail(B, d, AMH_E_TOO_LARGE); return; }
  const u8* p = B.arena + start;
  u64 pos = 0, end = avail;
  if (avail < 9 || p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83) { amh_fail(B, d, AMH_E_PARSE); return; }
  u8 type = p[8];
  if (type > 2) { amh_fail(B, d, AMH_E_PARSE); return; }
  pos = 9;
  u64 len;
  if (rd_uleb_strict(p, pos, end, len)) { amh_fail(B, d, AMH_E_PARSE); return; }
  if (len > end - pos) { amh_fail(B, d, AMH_E_PARSE); return; }
  D.data_off = (u32)pos;
  D.data_len = (u32)len;
  D.chunk_len = (u32)(pos + len);
  // a change chunk first = an empty document that the changes are then applied to; chunks after the
  // first one are applied as well (automerge.rs:657-695): the host hands both to the apply path
  if (type != 0) { D.chunk_len = 0; D.data_off = D.data_len = 0; return; }
  end = pos + len;
  u64 n;
  if (rd_uleb_strict(p, pos, end, n)) { amh_fail(B, d, AMH_E_PARSE); return; }
  if (n > len) { amh_fail(B, d, AMH_E_PARSE); return; }
  D.actors_off = (u32)pos;
  for (u64 i = 0; i < n; i++) {
    u64 al;
    if (rd_uleb_strict(p, pos, end, al) || al > end - pos) { amh_fail(B, d, AMH_E_PARSE); return; }
    pos += al;
  }
  D.n_actors = (u32)n;
  if (rd_uleb_strict(p, pos, end, n)) { amh_fail(B, d, AMH_E_PARSE); return; }
  if (n > (end - pos) / 32) { amh_fail(B, d, AMH_E_PARSE); return; }
  D.heads_off = (u32)pos;
  D.n_heads = (u32)n;
  pos += 32 * n;
  // walk 1: validate both metadata blocks and total their data
  u64 meta_pos = pos, ctot = 0, otot = 0;
  int rc = parse_col_meta<false>(p, pos, end, len, ctot, (ColMapper*)0);
  if (rc == 0) rc = parse_col_meta<true>(p, pos, end, len, otot, (ColMapper*)0);
  if (rc) { amh_fail(B, d, rc == 2 ? AMH_E_UNSUPPORTED : AMH_E_PARSE); return; }
  if (ctot > end - pos) { amh_fail(B, d, AMH_E_PARSE); return; }
  u64 cdata = pos;
  pos += ctot;
  if (otot > end - pos) { amh_fail(B, d, AMH_E_PARSE); return; }
  u64 odata = pos;
  pos += otot;
  D.suffix_off = (u32)pos;
  D.suffix_len = (u32)(end - pos);
  if (pos < end) {  // head indices: must be exactly n_heads strict ulebs (document.rs:140-154; chunk.rs:72-74)
    for (u32 i = 0; i < D.n_heads; i++) {
      u64 hi;
      if (rd_uleb_strict(p, pos, end, hi)) { amh_fail(B, d, AMH_E_PARSE); return; }
    }
    if (pos != end) { amh_fail(B, d, AMH_E_PARSE); return; }
  }
  // walk 2: the column layout
  ColMapper m;
  u64 t2;
  cm_init(m, cdata, ctot, D.ccol);
  if (parse_col_meta<false>(p, meta_pos, end, len, t2, &m) || cm_finish<false>(m)) { amh_fail(B, d, AMH_E_PARSE); return; }
  cm_init(m, odata, otot, D.ocol);
  if (parse_col_meta<true>(p, meta_pos, end, len, t2, &m) || cm_finish<true>(m)) { amh_fail(B, d, AMH_E_PARSE); return; }
}

// ------------------------------------------------------------------------------------------
// K1b  SHA-256 of the document chunk and checksum check — chunk.rs:277-293.  Thread per document.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_sha_doc(const BatchDev& B, u32 d) {
  if (B.sha_gate ? B.sha_gate[d] : B.status[d]) return;
  const DocDesc& D = B.desc[d];
  if (D.chunk_len == 0 || B.skip_doc_sha) return;
  if (B.sha_host_min && D.chunk_len >= B.sha_host_min) return;   // the host hashes this one (amh_host.cpp, BigSha)
  const u8* p = B.arena + D.chunk_off;
  // type byte + uleb(len) exactly as stored (strict parse => canonical) and the data: one contiguous range
  u8* h = B.doc_hash + (u64)d * 32;
  sha256_contig(p + 8, (u64)(D.data_off - 8) + D.data_len, h);
  if (!B.no_checksum && (h[0] != p[4] || h[1] != p[5] || h[2] != p[6] || h[3] != p[7])) {
    if (B.sha_bad) B.sha_bad[d] = 1; else amh_fail(B, d, AMH_E_BAD_CHECKSUM);
  }
}

// ------------------------------------------------------------------------------------------
// K2  count pass: rows (items of the id-counter column, doc_op_columns.rs:327-331), succ entries,
//     change rows and dep entries.  Thread per document.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_count(const BatchDev& B, u32 d) {
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  DocCounts& C = B.cnt[d];
  if (D.chunk_len == 0) return;
  const u8* p = B.arena + D.chunk_off;
  u64 v; u32 aux; bool isnull;
  {
    RleDec r;
    rle_init(r, p, D.ocol[OC_ID_CTR].off, D.ocol[OC_ID_CTR].len);
    u64 rows = 0;
    // count run lengths without expanding: headers only
    while (r.pos < r.end) {
      i64 n;
      if (!rd_sleb_lenient(r.p, r.pos, r.end, n)) { amh_fail(B, d, AMH_E_READ_OP); return; }
      if (n > 0) {
        if (!rle_read_value<1>(r, v, aux)) { amh_fail(B, d, AMH_E_READ_OP); return; }
        rows += (u64)n;
      } else if (n < 0) {
        u64 k = (n == INT64_MIN) ? (u64)INT64_MAX : (u64)(-n);
        if (k > r.end - r.pos) { amh_fail(B, d, AMH_E_READ_OP); return; }  // each literal takes >= 1 byte
        for (u64 i = 0; i < k; i++)
          if (!rle_read_value<1>(r, v, aux)) { amh_fail(B, d, AMH_E_READ_OP); return; }
        rows += k;
      } else {
        u64 nulls;
        if (!rd_uleb_lenient(r.p, r.pos, r.end, nulls)) { amh_fail(B, d, AMH_E_READ_OP); return; }
        if (nulls) { amh_fail(B, d, AMH_E_READ_OP); return; }  // a null id is "unexpected null" (opid.rs:114-117)
      }
      if (rows > B.max_doc_rows) { amh_fail(B, d, AMH_E_TOO_LARGE); return; }
    }
    C.n_rows = (u32)rows;
    rle_init(r, p, D.ocol[OC_SUCC_NUM].off, D.ocol[OC_SUCC_NUM].len);
    u64 succ = 0;
    for (u64 i = 0; i < rows; i++) {
      int rc = rle_next<0>(r, v, aux, isnull);
      if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_OP); return; }
      succ += v;
      if (v > B.max_doc_rows || succ > B.max_doc_rows) { amh_fail(B, d, AMH_E_TOO_LARGE); return; }
    }
    C.n_succ = (u32)succ;
  }
  {
    RleDec r, dn;
    rle_init(r, p, D.ccol[CC_ACTOR].off, D.ccol[CC_ACTOR].len);
    rle_init(dn, p, D.ccol[CC_DEPS_NUM].off, D.ccol[CC_DEPS_NUM].len);
    u64 chg = 0, deps = 0;
    for (;;) {
      int rc = rle_next<0>(r, v, aux, isnull);
      if (rc < 0) { amh_fail(B, d, AMH_E_READ_CHANGE); return; }
      if (rc == 1 || isnull) break;
      chg++;
      rc = rle_next<0>(dn, v, aux, isnull);
      if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_CHANGE); return; }
      deps += v;
      if (chg > B.max_doc_rows || v > B.max_doc_rows || deps > B.max_doc_rows) { amh_fail(B, d, AMH_E_TOO_LARGE); return; }
    }
    C.n_chg = (u32)chg;
    C.n_deps = (u32)deps;
  }
}

// ------------------------------------------------------------------------------------------
// K3  actor table + bytewise rank (document.rs:121; lamport tie-break types.rs:517-521).
//     Thread per document.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_actors(const BatchDev& B, u32 d) {
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  const u8* p = B.arena + D.chunk_off;
  u32 base = B.actor_base[d];
  u64 pos = D.actors_off, end = D.data_off + (u64)D.data_len;
  for (u32 i = 0; i < D.n_actors; i++) {
    u64 al = 0;
    rd_uleb_strict(p, pos, end, al);
    B.actor_off[base + i] = (u32)pos;
    B.actor_len[base + i] = (u32)al;
    pos += al;
    B.a_first_chg[base + i] = AMH_EMPTY32;
    B.a_last_chg[base + i] = AMH_EMPTY32;
  }
  bool dup = false;
  for (u32 i = 0; i < D.n_actors; i++) {
    u32 rank = 0;
    for (u32 j = 0; j < D.n_actors; j++) {
      if (j == i) continue;
      int c = amh_memcmp(p + B.actor_off[base + j], B.actor_len[base + j], p + B.actor_off[base + i], B.actor_len[base + i]);
      if (c < 0) rank++;
      if (c == 0) dup = true;
    }
    B.actor_rank[base + i] = rank;
  }
  if (dup) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }  // duplicate actor ids in one table: not a shape writers produce
  for (u32 i = 0; i < D.n_actors; i++) B.actor_by_rank[base + B.actor_rank[base + i]] = i;
}

// ------------------------------------------------------------------------------------------
// K4  change metadata rows in lock-step — doc_change_columns.rs:183-270, deps.rs:45-102.
//     Thread per document.
// ------------------------------------------------------------------------------------------
// The eight change-metadata columns are decoded by a thread each (index = column * n_docs + document: the lanes of a
// wave run the same decoder over 64 documents), every thread walking its own run-length stream from the start; what a
// column cannot say on its own (a change's place among the dependency indexes, levels, ordering) is left to
// k_decode_chg, one thread per document, which reads the decoded arrays.  The reference reads the columns change by
// change and stops at the first error (doc_change_columns.rs:196-270): every column records the first change (for the
// dependency indexes: the first entry) it cannot deliver, and k_decode_chg fails the document where the reference's
// loop would have met the earliest of them.
#define CD_COLS 8u   // 0 actor, 1 seq, 2 max op, 3 time, 4 message, 5 deps num, 6 deps index, 7 extra
AMH_DEV void k_decode_chg_cols(const BatchDev& B, u32 idx) {
  const u32 col = idx / B.n_docs, d = idx % B.n_docs;
  u32* err = B.cd_err + (u64)d * CD_COLS + col;
  u32* code = B.cd_code + (u64)d * CD_COLS + col;
  u32* none = B.cd_none + (u64)d * CD_COLS + col;
  *err = AMH_NULL32; *code = AMH_E_READ_CHANGE; *none = 0;
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  const DocCounts& C = B.cnt[d];
  if (D.chunk_len == 0) return;
  const u8* p = B.arena + D.chunk_off;
  const u32 cb = B.chg_base[d], nc = C.n_chg;
  u64 v; u32 aux; bool isnull; i64 sv;
  if (col == 0) {
    RleDec r; rle_init(r, p, D.ccol[CC_ACTOR].off, D.ccol[CC_ACTOR].len);
    for (u32 i = 0; i < nc; i++) {
      if (rle_next<0>(r, v, aux, isnull) != 0 || isnull) { *err = i; return; }
      if (v >= D.n_actors) { *err = i; *code = AMH_E_COLLECT_MISSING_ACTOR; return; }  // change_collector.rs:185-192
      B.c_actor[cb + i] = (u32)v;
    }
  } else if (col == 1 || col == 2 || col == 3) {
    const int cc = col == 1 ? CC_SEQ : (col == 2 ? CC_MAX_OP : CC_TIME);
    DeltaDec r; delta_init(r, p, D.ccol[cc].off, D.ccol[cc].len);
    for (u32 i = 0; i < nc; i++) {
      if (delta_next(r, sv, isnull) != 0 || isnull || (col != 3 && sv < 0)) { *err = i; return; }
      if (col == 1) B.c_seq[cb + i] = (u64)sv; else if (col == 2) B.c_max_op[cb + i] = (u64)sv; else B.c_time[cb + i] = sv;
    }
    *none = delta_next(r, sv, isnull) == 1;
  } else if (col == 4) {
    const bool has_msg = D.ccol[CC_MESSAGE].len != 0;
    RleDec r; rle_init(r, p, D.ccol[CC_MESSAGE].off, D.ccol[CC_MESSAGE].len);
    for (u32 i = 0; i < nc; i++) {
      B.c_msg_off[cb + i] = 0;
      B.c_msg_len[cb + i] = AMH_NULL32;
      if (has_msg) {
        int rc = rle_next<2>(r, v, aux, isnull);
        if (rc < 0) { *err = i; return; }
        if (rc == 0 && !isnull) { B.c_msg_off[cb + i] = (u32)v; B.c_msg_len[cb + i] = aux; }
      }
    }
  } else if (col == 5) {
    RleDec r; rle_init(r, p, D.ccol[CC_DEPS_NUM].off, D.ccol[CC_DEPS_NUM].len);
    for (u32 i = 0; i < nc; i++) {
      if (rle_next<0>(r, v, aux, isnull) != 0 || isnull) { *err = i; return; }
      B.c_ndeps[cb + i] = v > 0xFFFFFFFFull ? AMH_NULL32 : (u32)v;   // (more than the column can hold either way)
    }
    *none = rle_next<0>(r, v, aux, isnull) == 1;
  } else if (col == 6) {
    DeltaDec r; delta_init(r, p, D.ccol[CC_DEPS_IDX].off, D.ccol[CC_DEPS_IDX].len);
    const u32 db = B.dep_base[d];
    for (u32 e = 0; e < C.n_deps; e++) {   // err = the first ENTRY that does not read
      if (delta_next(r, sv, isnull) != 0 || isnull || sv < 0) { *err = e; return; }
      B.dep_idx[db + e] = sv > 0xFFFFFFFFll ? AMH_NULL32 : (u32)sv;
    }
  } else {
    // extra bytes: must be a Bytes value when present (doc_change_columns.rs:256-270)
    RleDec r; rle_init(r, p, D.ccol[CC_EXTRA_META].off, D.ccol[CC_EXTRA_META].len);
    u64 extra_off = 0;
    for (u32 i = 0; i < nc; i++) {
      B.c_extra_off[cb + i] = 0;
      B.c_extra_len[cb + i] = 0;
      int rc = rle_next<0>(r, v, aux, isnull);
      if (rc < 0) { *err = i; return; }
      if (rc == 0) {
        if (isnull || (v & 15) != 7) { *err = i; return; }
        u64 el = v >> 4;
        if (el > D.ccol[CC_EXTRA_RAW].len || extra_off > D.ccol[CC_EXTRA_RAW].len - el) { *err = i; return; }
        B.c_extra_off[cb + i] = (u32)(D.ccol[CC_EXTRA_RAW].off + extra_off);
        B.c_extra_len[cb + i] = (u32)el;
        extra_off += el;
      }
    }
  }
}
AMH_DEV void k_decode_chg(const BatchDev& B, u32 d) {
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  const DocCounts& C = B.cnt[d];
  if (D.chunk_len == 0) return;
  u32 cb = B.chg_base[d], db = B.dep_base[d];
  u32 ndep = 0, max_lvl = 0;
  u32 err[CD_COLS];
  for (u32 k = 0; k < CD_COLS; k++) err[k] = B.cd_err[(u64)d * CD_COLS + k];
  const i32* code = (const i32*)(B.cd_code + (u64)d * CD_COLS);
  for (u32 i = 0; i < C.n_chg; i++) {
    u32 c = cb + i;
    for (u32 k = 0; k < 6; k++) if (err[k] == i) { amh_fail(B, d, code[k]); return; }   // the columns in the order the reference reads them
    const u32 v = B.c_ndeps[c];
    B.c_dep_base[c] = db + ndep;
    u32 lvl = 0;
    for (u32 k = 0; k < v; k++) {
      if (ndep >= C.n_deps || ndep == err[6]) { amh_fail(B, d, AMH_E_READ_CHANGE); return; }
      const u32 sv = B.dep_idx[db + ndep];
      if (sv >= i) { amh_fail(B, d, AMH_E_MISSING_CHANGE); return; }  // change_collector.rs:163-177
      B.c_is_dep[cb + sv] = 1;
      u32 dl = B.c_level[cb + sv] + 1;
      if (dl > lvl) lvl = dl;
      ndep++;
    }
    B.c_level[c] = lvl;
    if (lvl > max_lvl) max_lvl = lvl;
    if (err[7] == i) { amh_fail(B, d, AMH_E_READ_CHANGE); return; }
  }
  // termination quirk (doc_change_columns.rs:196-207, 243-253): the actor column ran out or produced
  // a null; the row set is complete only if at least one other column is exhausted as well
  {
    const u32* none = B.cd_none + (u64)d * CD_COLS;
    if (!(none[1] || none[2] || none[3] || none[5])) { amh_fail(B, d, AMH_E_READ_CHANGE); return; }
  }
  B.doc_level[d] = max_lvl;
  amh_atomic_max(B.max_level, max_lvl);
  // per-actor chains in file order + ordering checks (change_collector.rs:56-62, 105-118)
  u32 ab = B.actor_base[d];
  for (u32 i = 0; i < C.n_chg; i++) {
    u32 c = cb + i, a = ab + B.c_actor[c];
    u32 last = B.a_last_chg[a];
    if (last == AMH_EMPTY32) {
      B.a_first_chg[a] = c;
      if (B.c_seq[c] != 1) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
    } else {
      if (B.c_max_op[last] > B.c_max_op[c]) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
      if (B.c_seq[last] != B.c_seq[c] - 1) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
      B.c_next_same_actor[last] = c;
    }
    B.a_last_chg[a] = c;
  }
}

// K4m  merge / apply hand their merged rows and change metadata straight to the pipeline (no save + decode in between):
//      what k_extract, k_chg_init and the dependency / chain part of k_decode_chg leave behind, from change metadata
//      that is already in arrays.  Thread per document.
AMH_DEV void k_m_init(const BatchDev& B, u32 d) {
  B.heads_ok[d] = 0;
  B.doc_level[d] = 0;
  B.ht_base[d] = 0;
  B.dt_base[d] = 0;
  if (B.status[d]) return;
  u32 cb = B.chg_base[d], nc = B.cnt[d].n_chg, ab = B.actor_base[d], na = B.desc[d].n_actors;
  for (u32 a = 0; a < na; a++) { B.a_first_chg[ab + a] = AMH_EMPTY32; B.a_last_chg[ab + a] = AMH_EMPTY32; }
  u32 max_lvl = 0;
  for (u32 i = 0; i < nc; i++) {
    u32 c = cb + i;
    B.c_nops[c] = 0;
    B.c_next_same_actor[c] = AMH_EMPTY32;
    B.c_map_base[c] = B.map_base[d] + i * na;
    if (B.c_actor[c] >= na) { amh_fail(B, d, AMH_E_COLLECT_MISSING_ACTOR); return; }
    u32 lvl = 0;
    for (u32 k = 0; k < B.c_ndeps[c]; k++) {
      u32 di = B.dep_idx[B.c_dep_base[c] + k];
      if (di >= i) { amh_fail(B, d, AMH_E_MISSING_CHANGE); return; }
      B.c_is_dep[cb + di] = 1;
      u32 dl = B.c_level[cb + di] + 1;
      if (dl > lvl) lvl = dl;
    }
    B.c_level[c] = lvl;
    if (lvl > max_lvl) max_lvl = lvl;
  }
  B.doc_level[d] = max_lvl;
  amh_atomic_max(B.max_level, max_lvl);
  for (u32 i = 0; i < nc; i++) {   // per-actor chains in history order + ordering checks, as k_decode_chg
    u32 c = cb + i, a = ab + B.c_actor[c];
    u32 last = B.a_last_chg[a];
    if (last == AMH_EMPTY32) {
      B.a_first_chg[a] = c;
      if (B.c_seq[c] != 1) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
    } else {
      if (B.c_max_op[last] > B.c_max_op[c]) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
      if (B.c_seq[last] != B.c_seq[c] - 1) { amh_fail(B, d, AMH_E_CHANGES_OUT_OF_ORDER); return; }
      B.c_next_same_actor[last] = c;
    }
    B.a_last_chg[a] = c;
  }
}
// the re-derived hash of every change of a merged document against the hash the change came with.  Thread per change.
AMH_DEV void k_verify_hash(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  if (d >= B.n_docs || B.status[d]) return;
  const u8* x = B.c_hash + (u64)c * 32;
  const u8* y = B.expect_hash + (u64)c * 32;
  bool same = true;
  for (int i = 0; i < 32; i++) same = same && x[i] == y[i];
  if (!same) amh_fail(B, d, AMH_E_MISMATCHING_HEADS);
}

// ------------------------------------------------------------------------------------------
// K5  op column decode — rle.rs:175-210, delta.rs:79-95, boolean.rs:84-101, value.rs:209-272
//     (meta only).  Thread per (document, column stream); each stream expands its runs into the
//     column array of the batch-wide SoA.  Null / exhausted = AMH_NULL32.
// ------------------------------------------------------------------------------------------
// byte range of the op columns inside the chunk, and the staging class of the document:
// 0 = fits the 16 KB LDS image, 1 = fits the 64 KB image, 2 = decoded straight from HBM
#define AMH_LDS_CLASS0 16384u
#define AMH_LDS_CLASS1 65536u
AMH_DEV u32 op_col_region(const DocDesc& D, u32& lo, u32& hi) {
  lo = 0xFFFFFFFFu; hi = 0;
  for (int i = 0; i < OC_COUNT; i++) {
    if (D.ocol[i].len == 0) continue;
    if (D.ocol[i].off < lo) lo = D.ocol[i].off;
    if (D.ocol[i].off + D.ocol[i].len > hi) hi = D.ocol[i].off + D.ocol[i].len;
  }
  if (hi == 0) { lo = 0; return 0; }
  u32 span = hi - (lo & ~15u);
  return span <= AMH_LDS_CLASS0 ? 0 : (span <= AMH_LDS_CLASS1 ? 1 : 2);
}

// one column stream of one document.  `p` addresses the chunk bytes shifted by `bias`: p[x - bias] is
// chunk byte x (bias = 0: the arena itself; otherwise an LDS image of the op-column region).
AMH_DEV void decode_ops_stream(const BatchDev& B, u32 d, u32 col, const u8* p, u32 bias) {
  const DocDesc& D = B.desc[d];
  const DocCounts& C = B.cnt[d];
  if (C.n_rows == 0) return;
  u32 rb = B.row_base[d], n = C.n_rows;
  u32 len = D.ocol[col].len, off = len ? D.ocol[col].off - bias : 0;
  u64 v; u32 aux; bool isnull; i64 sv;
  switch (col) {
    case OC_OBJ_ACTOR: case OC_OBJ_CTR: case OC_KEY_ACTOR: case OC_ID_ACTOR: case OC_ACTION: {
      RleDec r;
      rle_init(r, p, off, len);
      u32* out32 = col == OC_OBJ_ACTOR ? B.r_obj_actor : col == OC_OBJ_CTR ? B.r_obj_ctr : col == OC_KEY_ACTOR ? B.r_key_actor : B.r_id_actor;
      for (u32 i = 0; i < n; i++) {
        int rc = rle_next<0>(r, v, aux, isnull);
        if (rc < 0) { amh_fail(B, d, AMH_E_READ_OP); return; }
        bool none = rc == 1 || isnull;
        if (col == OC_ACTION) {
          if (none) { amh_fail(B, d, AMH_E_READ_OP); return; }
          if (v > 7) { amh_fail(B, d, AMH_E_INVALID_OP); return; }
          B.r_action[rb + i] = (u8)v;
        } else {
          if (!none && v >= 0xFFFFFFFFull) { amh_fail(B, d, col == OC_OBJ_CTR ? AMH_E_READ_OP : AMH_E_MISSING_ACTOR); return; }
          out32[rb + i] = none ? AMH_NULL32 : (u32)v;
        }
      }
      break;
    }
    case OC_KEY_CTR: case OC_ID_CTR: {
      DeltaDec r;
      delta_init(r, p, off, len);
      u32* out32 = col == OC_KEY_CTR ? B.r_key_ctr : B.r_id_ctr;
      u32 mx = 0;
      for (u32 i = 0; i < n; i++) {
        int rc = delta_next(r, sv, isnull);
        if (rc < 0) { amh_fail(B, d, AMH_E_READ_OP); return; }
        bool none = rc == 1 || isnull;
        if (!none && (sv < 0 || sv >= 0xFFFFFFFFll)) { amh_fail(B, d, AMH_E_READ_OP); return; }  // key.rs:172-179, opid.rs:107-113
        out32[rb + i] = none ? AMH_NULL32 : (u32)sv;
        if (col == OC_ID_CTR) { B.r_doc[rb + i] = d; if (!none && (u32)sv > mx) mx = (u32)sv; }
      }
      if (col == OC_ID_CTR) amh_atomic_max(&B.doc_max_op[d], mx);  // max_op over op ids (reconstruct_document.rs:136-140)
      break;
    }
    case OC_KEY_STR: case OC_MARK_NAME: {
      RleDec r;
      rle_init(r, p, off, len);
      u32* o_off = col == OC_KEY_STR ? B.r_key_off : B.r_mark_off;
      u32* o_len = col == OC_KEY_STR ? B.r_key_len : B.r_mark_len;
      for (u32 i = 0; i < n; i++) {
        int rc = rle_next<2>(r, v, aux, isnull);
        if (rc < 0) { amh_fail(B, d, AMH_E_READ_OP); return; }
        bool none = rc == 1 || isnull;
        o_off[rb + i] = none ? 0 : (u32)v + bias;
        o_len[rb + i] = none ? AMH_NULL32 : aux;
      }
      break;
    }
    case OC_INSERT: {
      BoolDec r;
      bool_init(r, p, off, len);
      for (u32 i = 0; i < n; i++) {
        bool b;
        int rc = bool_next(r, b);
        if (rc != 0) { amh_fail(B, d, AMH_E_READ_OP); return; }  // exhausted = unexpected null (column_decoder.rs:44-50)
        B.r_insert[rb + i] = b;
      }
      break;
    }
    case OC_EXPAND: {
      BoolDec r;
      bool_init(r, p, off, len);
      for (u32 i = 0; i < n; i++) {
        bool b = false;
        if (len) {
          int rc = bool_next(r, b);
          if (rc < 0) { amh_fail(B, d, AMH_E_READ_OP); return; }
          if (rc == 1) b = false;
        }
        B.r_expand[rb + i] = b;
      }
      break;
    }
    case OC_VAL_META: {
      RleDec r;
      rle_init(r, p, off, len);
      u64 raw_off = 0, raw_len = D.ocol[OC_VAL_RAW].len;
      for (u32 i = 0; i < n; i++) {
        int rc = rle_next<0>(r, v, aux, isnull);
        if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_OP); return; }
        B.r_vmeta[rb + i] = v;
        B.r_vraw_off[rb + i] = (u32)raw_off;
        if ((v & 15) >= 3) {  // null / bool never touch the raw column (value.rs:221-224)
          u64 l = v >> 4;
          if (l > raw_len || raw_off > raw_len - l) { amh_fail(B, d, AMH_E_READ_OP); return; }
          raw_off += l;
        }
      }
      break;
    }
    case OC_SUCC_NUM: {
      RleDec r;
      rle_init(r, p, off, len);
      u32 sb = B.succ_base[d], run = 0;
      for (u32 i = 0; i < n; i++) {
        int rc = rle_next<0>(r, v, aux, isnull);
        if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_OP); return; }
        B.r_succ_cnt[rb + i] = (u32)v;
        B.r_succ_base[rb + i] = sb + run;   // the owning row of each entry (s_row) is filled in by k_rows
        run += (u32)v;
      }
      break;
    }
    case OC_SUCC_ACTOR: {
      RleDec r;
      rle_init(r, p, off, len);
      u32 sb = B.succ_base[d];
      for (u32 i = 0; i < C.n_succ; i++) {
        int rc = rle_next<0>(r, v, aux, isnull);
        if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_OP); return; }
        if (v >= 0xFFFFFFFFull) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }
        B.s_actor[sb + i] = (u32)v;
      }
      break;
    }
    case OC_SUCC_CTR: {
      DeltaDec r;
      delta_init(r, p, off, len);
      u32 sb = B.succ_base[d], mx = 0;
      for (u32 i = 0; i < C.n_succ; i++) {
        int rc = delta_next(r, sv, isnull);
        if (rc != 0 || isnull) { amh_fail(B, d, AMH_E_READ_OP); return; }
        if (sv < 0 || sv >= 0xFFFFFFFFll) { amh_fail(B, d, AMH_E_READ_OP); return; }
        B.s_ctr[sb + i] = (u32)sv;
        if ((u32)sv > mx) mx = (u32)sv;
      }
      amh_atomic_max(&B.doc_max_op[d], mx);  // ... and over succ ids (reconstruct_document.rs:141-143)
      break;
    }
    default: break;  // OC_VAL_RAW has no stream of its own
  }
}
AMH_DEV void k_decode_ops(const BatchDev& B, u32 gid) {
  // column-major: the 64 lanes of a wave decode the same column of 64 documents (no switch divergence)
  u32 col = gid / B.n_docs, d = gid % B.n_docs;
  if (col >= OC_COUNT || B.status[d]) return;
  decode_ops_stream(B, d, col, B.arena + B.desc[d].chunk_off, 0);
}

// parse the payload of an int-like value (types 3,4,8,9) strictly; false = malformed
AMH_DEV bool val_int(const BatchDev& B, u32 r, i64& out) {
  u32 d = B.r_doc[r];
  const DocDesc& D = B.desc[d];
  const u8* p = B.arena + D.chunk_off;
  u64 m = B.r_vmeta[r], l = m >> 4;
  u64 pos = (u64)D.ocol[OC_VAL_RAW].off + B.r_vraw_off[r], end = pos + l;
  if ((m & 15) == 3) {
    u64 u;
    if (rd_uleb_strict(p, pos, end, u) || pos != end) return false;
    out = (i64)u;
    return true;
  }
  if (rd_sleb_strict(p, pos, end, out) || pos != end) return false;
  return true;
}

AMH_DEV u32 find_doc(const u32* base, u32 n_docs, u32 idx) {  // largest d with base[d] <= idx
  u32 lo = 0, hi = n_docs;
  while (hi - lo > 1) {
    u32 mid = (lo + hi) / 2;
    if (base[mid] <= idx) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// K6  row validation + normalisation + id table insert — composite iterators obj_id.rs:123-141,
//     key.rs:146-187, opid.rs:95-120, value.rs:209-308; import_op reconstruct_document.rs:342-375.
//     Thread per row.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_rows(const BatchDev& B, u32 r) {
  u32 d = B.r_doc[r];   // written by the id-counter stream of k_decode_ops; AMH_NULL32 when that document failed earlier
  B.r_inc_cnt[r] = 0;
  B.r_inc_sum[r] = 0;
  if (d == AMH_NULL32) { B.r_doc[r] = find_doc(B.row_base, B.n_docs, r); return; }
  if (B.status[d]) return;
  {  // owning row of each succ entry
    u32 sc = B.r_succ_cnt[r], sb = B.r_succ_base[r];
    for (u32 k = 0; k < sc; k++) B.s_row[sb + k] = r;
  }
  const DocDesc& D = B.desc[d];
  u32 na = D.n_actors;
  u32 ia = B.r_id_actor[r], ic = B.r_id_ctr[r];
  if (ia == AMH_NULL32 || ic == AMH_NULL32) { amh_fail(B, d, AMH_E_READ_OP); return; }
  if (ia >= na) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }
  // object id
  u32 oa = B.r_obj_actor[r], oc = B.r_obj_ctr[r];
  if (D.ocol[OC_OBJ_ACTOR].len == 0 && D.ocol[OC_OBJ_CTR].len == 0) { oa = 0; oc = 0; }
  else if (oa == AMH_NULL32 && oc == AMH_NULL32) { oa = 0; oc = 0; }
  else if (oa != AMH_NULL32 && oc != AMH_NULL32) { if (oc == 0) oa = 0; }
  else if (oc == 0) { oa = 0; }
  else { amh_fail(B, d, AMH_E_READ_OP); return; }
  if (oc != 0 && oa >= na) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }
  if (oc == 0 && na == 0) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }  // check_opid(root) needs actor slot 0
  B.r_obj_actor[r] = oa;
  B.r_obj_ctr[r] = oc;
  // key
  u32 ka = B.r_key_actor[r], kc = B.r_key_ctr[r], kl = B.r_key_len[r];
  bool an = ka == AMH_NULL32, cn = kc == AMH_NULL32, sn = kl == AMH_NULL32;
  if (an && cn && !sn) { /* map key */ }
  else if (an && !cn && kc == 0 && sn) { B.r_key_actor[r] = 0; }  // HEAD
  else if (!an && !cn && sn) {
    if (ka >= na) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }
    if (kc == 0 && ka != 0) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }  // ElemId(0@a), a != 0: not HEAD, never written
  } else { amh_fail(B, d, AMH_E_READ_OP); return; }
  // value payload
  u64 m = B.r_vmeta[r];
  u32 vt = (u32)(m & 15);
  u64 vl = m >> 4;
  const u8* p = B.arena + D.chunk_off;
  if (vt == 3 || vt == 4 || vt == 8 || vt == 9) {
    i64 tmp;
    if (!val_int(B, r, tmp)) { amh_fail(B, d, AMH_E_READ_OP); return; }
  } else if (vt == 5) {
    if (vl != 8) { amh_fail(B, d, AMH_E_READ_OP); return; }
  } else if (vt == 6) {
    if (!amh_valid_utf8(p + D.ocol[OC_VAL_RAW].off + B.r_vraw_off[r], vl)) { amh_fail(B, d, AMH_E_READ_OP); return; }
  }
  u8 action = B.r_action[r];
  if (action == 5 && vt != 3 && vt != 4) { amh_fail(B, d, AMH_E_INVALID_OP); return; }
  // id -> row
  u32 stride = B.dn_stride[d];
  if (stride) {
    if (ic >= stride) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }   // cannot happen: stride covers doc_max_op
    if (amh_atomic_cas(&B.dn_val[B.dn_base[d] + ia * stride + ic], AMH_NULL32, r) != AMH_NULL32) amh_fail(B, d, AMH_E_UNSUPPORTED);   // duplicate op id in one document
    return;
  }
  u64 base = B.ht_base[d], size = B.ht_base[d + 1] - base, mask = size - 1;
  u64 key = pack_id(ic, ia) + 1, h = amh_mix64(key) & mask;
  for (u64 i = 0; i < size; i++) {
    u64 old = amh_atomic_cas64(&B.ht_key[base + h], 0, key);
    if (old == 0) { B.ht_val[base + h] = r; break; }
    if (old == key) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }  // duplicate op id in one document
    h = (h + 1) & mask;
  }
}

// K6m  the rows of a merged document come normalised and validated (they were rows of loaded documents and of decoded
//      changes): what is left of k_rows is the owning row of each succ entry and the id -> row table.  Thread per row.
AMH_DEV void k_rows_m(const BatchDev& B, u32 r) {
  u32 d = B.r_doc[r];
  B.r_inc_cnt[r] = 0;
  B.r_inc_sum[r] = 0;
  if (d >= B.n_docs || B.status[d]) return;
  {
    u32 sc = B.r_succ_cnt[r], sb = B.r_succ_base[r];
    for (u32 k = 0; k < sc; k++) B.s_row[sb + k] = r;
  }
  u32 ia = B.r_id_actor[r], ic = B.r_id_ctr[r];
  u32 stride = B.dn_stride[d];
  if (stride) {
    if (ic >= stride || ia >= B.desc[d].n_actors) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }
    if (amh_atomic_cas(&B.dn_val[B.dn_base[d] + ia * stride + ic], AMH_NULL32, r) != AMH_NULL32) amh_fail(B, d, AMH_E_UNSUPPORTED);
    return;
  }
  u64 base = B.ht_base[d], size = B.ht_base[d + 1] - base, mask = size - 1;
  u64 key = pack_id(ic, ia) + 1, h = amh_mix64(key) & mask;
  for (u64 i = 0; i < size; i++) {
    u64 old = amh_atomic_cas64(&B.ht_key[base + h], 0, key);
    if (old == 0) { B.ht_val[base + h] = r; break; }
    if (old == key) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }
    h = (h + 1) & mask;
  }
}

AMH_DEV bool is_make(u8 action) { return action == 0 || action == 2 || action == 4 || action == 6; }

// ------------------------------------------------------------------------------------------
// K7  object order: every object is one contiguous run, runs ascend by (ctr, actor index), and the
//     object's create op precedes its first use — reconstruct_document.rs:159-183.  Thread per row.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_obj_order(const BatchDev& B, u32 r) {
  u32 d = B.r_doc[r];
  if (B.status[d]) return;
  u32 oc = B.r_obj_ctr[r], oa = B.r_obj_actor[r];
  u32 pc = 0, pa = 0;
  if (r > B.row_base[d]) { pc = B.r_obj_ctr[r - 1]; pa = B.r_obj_actor[r - 1]; }
  if (oc == pc && oa == pa) return;
  u32 mk = ht_find(B, d, pack_id(oc, oa));
  if (mk == AMH_EMPTY32 || mk >= r || !is_make(B.r_action[mk])) { amh_fail(B, d, AMH_E_OPS_OUT_OF_ORDER); return; }
  if (pack_id(oc, oa) < pack_id(pc, pa)) { amh_fail(B, d, AMH_E_OPS_OUT_OF_ORDER); return; }
}

// delete-table insert-or-find; returns global slot, AMH_EMPTY32 if the table is full (cannot happen)
AMH_DEV u32 dt_claim(const BatchDev& B, u32 d, u64 id) {
  u64 base = B.dt_base[d], size = B.dt_base[d + 1] - base, mask = size - 1;
  u64 key = id + 1, h = amh_mix64(key) & mask;
  for (u64 i = 0; i < size; i++) {
    u64 old = amh_atomic_cas64(&B.dt_key[base + h], 0, key);
    if (old == 0 || old == key) return (u32)(base + h);
    h = (h + 1) & mask;
  }
  return AMH_EMPTY32;
}

AMH_DEV int lamport_cmp(const BatchDev& B, u32 d, u32 c1, u32 a1, u32 c2, u32 a2) {
  if (c1 != c2) return c1 < c2 ? -1 : 1;
  u32 ab = B.actor_base[d];
  u32 r1 = B.actor_rank[ab + a1], r2 = B.actor_rank[ab + a2];
  return r1 < r2 ? -1 : (r1 > r2 ? 1 : 0);
}

// ------------------------------------------------------------------------------------------
// K8  succ entries: ordering check, pred = inverse of succ inside the object, counter folding,
//     delete discovery — reconstruct_document.rs:141-143, 285-287, 299-310, 347-358.
//     Thread per succ entry, two passes.  k_succ finds the target row of every entry; an entry without one is
//     the pred of a delete, and is only counted (per document, in dt_base).  The delete table is then sized
//     for the entries that need it (k_dt_size) instead of for every succ entry of the batch, and with it the
//     op space [0, tot_rows + tot_dt) that the per-op kernels and arrays run over.  k_succ_link counts the preds
//     of the rows and claims the delete slots.
// ------------------------------------------------------------------------------------------
#define AMH_PENDING32 0xFFFFFFFEu
AMH_DEV u32 amh_pow2_at_least(u32 v) { u32 p = 1; while (p < v) p <<= 1; return p; }
AMH_DEV void k_succ(const BatchDev& B, u32 e) {
  B.s_target[e] = AMH_EMPTY32;
  u32 r = B.s_row[e];
  if (r == AMH_EMPTY32) return;  // owning document failed while decoding
  u32 d = B.r_doc[r];
  if (B.status[d]) return;
  u32 sa = B.s_actor[e], sc = B.s_ctr[e];
  if (sa >= B.desc[d].n_actors) { amh_fail(B, d, AMH_E_MISSING_ACTOR); return; }
  if (e > B.r_succ_base[r]) {
    u32 pa = B.s_actor[e - 1], pc = B.s_ctr[e - 1];
    if (pa >= B.desc[d].n_actors) return;  // flagged by that entry's own thread
    if (lamport_cmp(B, d, pc, pa, sc, sa) >= 0) { amh_fail(B, d, AMH_E_SUCC_OUT_OF_ORDER); return; }
  }
  u32 tr = ht_find(B, d, pack_id(sc, sa));
  if (tr != AMH_EMPTY32 && B.r_obj_ctr[tr] == B.r_obj_ctr[r] && B.r_obj_actor[tr] == B.r_obj_actor[r]) {
    B.s_target[e] = tr;
    if (B.r_action[r] == 1 && (B.r_vmeta[r] & 15) == 8 && B.r_action[tr] == 5) {
      i64 inc = 0;
      if (val_int(B, tr, inc)) {
        amh_atomic_add_i64(&B.r_inc_sum[r], inc);
        amh_atomic_add(&B.r_inc_cnt[r], 1);
      }
    }
  } else {
    B.s_target[e] = AMH_PENDING32;
    amh_atomic_inc_grouped(B.dt_base, d);   // zero after the scan of the (still empty) table sizes: used as the counter
  }
}
AMH_DEV void k_dt_size(const BatchDev& B, u32 d) {
  u32 n = B.status[d] ? 0 : B.dt_base[d];
  B.dt_base[d] = n ? amh_pow2_at_least(2 * n) : 0;
}
AMH_DEV void k_succ_link(const BatchDev& B, u32 e) {
  u32 t = B.s_target[e];
  if (t == AMH_EMPTY32) return;
  u32 r = B.s_row[e];
  u32 d = B.r_doc[r];
  if (B.status[d]) { B.s_target[e] = AMH_EMPTY32; return; }
  if (t != AMH_PENDING32) { amh_atomic_add(&B.t_pred_cnt[t], 1); return; }
  u32 slot = dt_claim(B, d, pack_id(B.s_ctr[e], B.s_actor[e]));
  if (slot == AMH_EMPTY32) { B.s_target[e] = AMH_EMPTY32; amh_fail(B, d, AMH_E_UNSUPPORTED); return; }
  t = B.tot_rows + slot;
  B.d_doc[slot] = d;
  amh_atomic_min(&B.d_first_pred_row[slot], r);
  amh_atomic_add(&B.t_pred_cnt[t], 1);
  B.s_target[e] = t;
}

// accessors over the unified op index space -------------------------------------------------
AMH_DEV bool op_is_del(const BatchDev& B, u32 t) { return t >= B.tot_rows; }
AMH_DEV u32 op_doc(const BatchDev& B, u32 t) { return t < B.tot_rows ? B.r_doc[t] : B.d_doc[t - B.tot_rows]; }
AMH_DEV void op_id(const BatchDev& B, u32 t, u32& ctr, u32& actor) {
  if (t < B.tot_rows) { ctr = B.r_id_ctr[t]; actor = B.r_id_actor[t]; }
  else { u64 k = B.dt_key[t - B.tot_rows] - 1; ctr = (u32)(k >> 32); actor = (u32)k; }
}
// the row whose object / key a synthesized delete inherits (reconstruct_document.rs:316-320, 270-279)
AMH_DEV u32 op_key_row(const BatchDev& B, u32 t) { return t < B.tot_rows ? t : B.d_first_pred_row[t - B.tot_rows]; }

// ------------------------------------------------------------------------------------------
// K9  op -> change — change_collector.rs:81-96 (first change of the op's actor with max_op >= ctr).
//     Also checks the delete's first pred is a Put/Make (reconstruct_document.rs:270, 317-320).
//     Thread per op t.
// ------------------------------------------------------------------------------------------
// The changes of every actor as an array (a_tab, in the document's stretch of the change index space: an actor's
// changes at a_tab_base[actor], the change with sequence number s at + s - 1; the chains were checked to run 1, 2, ...
// with non-decreasing max_op), so that k_bucket finds an op's change by binary search instead of walking the actor's
// chain: a document typed one keystroke per change has as many changes as ops.  Thread per document / per change.
AMH_DEV void k_atab_base(const BatchDev& B, u32 d) {
  if (B.status[d]) return;
  const u32 ab = B.actor_base[d], na = B.desc[d].n_actors;
  u32 run = B.chg_base[d];
  for (u32 a = 0; a < na; a++) {
    B.a_tab_base[ab + a] = run;
    const u32 last = B.a_last_chg[ab + a];
    const u32 cnt = last != AMH_EMPTY32 ? (u32)B.c_seq[last] : 0;
    B.a_tab_bn[ab + a] = (u64)run | ((u64)cnt << 32);
    run += cnt;
  }
}
AMH_DEV void k_atab_fill(const BatchDev& B, u32 c) {
  const u32 d = B.c_doc[c];
  if (d >= B.n_docs || B.status[d]) return;
  const u32 at = B.a_tab_base[B.actor_base[d] + B.c_actor[c]] + (u32)B.c_seq[c] - 1;
  B.a_tab[at] = c;
  B.a_tab_max[at] = B.c_max_op[c] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)B.c_max_op[c];
}
AMH_DEV void k_bucket(const BatchDev& B, u32 t) {
  B.t_chg[t] = AMH_EMPTY32;
  if (op_is_del(B, t) && B.dt_key[t - B.tot_rows] == 0) return;
  u32 d = op_doc(B, t);
  if (B.status[d]) return;
  if (op_is_del(B, t)) {
    u32 fp = B.d_first_pred_row[t - B.tot_rows];
    u8 a = B.r_action[fp];
    if (!(a == 1 || is_make(a))) { amh_fail(B, d, AMH_E_MISSING_OPS); return; }
  }
  u32 ctr, actor;
  op_id(B, t, ctr, actor);
  const u64 bn = B.a_tab_bn[B.actor_base[d] + actor];
  const u32 n = (u32)(bn >> 32);
  if (n == 0) { amh_fail(B, d, AMH_E_COLLECT_MISSING_ACTOR); return; }
  const u32* tmax = B.a_tab_max + (u32)bn;
  u32 lo = 0, hi = n;   // first change of the actor with max_op >= ctr
  while (lo < hi) {
    const u32 mid = lo + (hi - lo) / 2;
    if (tmax[mid] < ctr) lo = mid + 1; else hi = mid;
  }
  if (lo == n) { amh_fail(B, d, AMH_E_MISSING_CHANGE); return; }
  const u32 c = B.a_tab[(u32)bn + lo];
  B.t_chg[t] = c;
  amh_atomic_inc_grouped(B.c_nops, c);
}

// K10  start_op per change — change_collector.rs:179, 194-202.  Thread per change.
AMH_DEV void k_chg_finish(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  B.c_start_op[c] = 1;
  if (B.status[d]) { B.c_nops[c] = 0; return; }
  u64 n = B.c_nops[c], mo = B.c_max_op[c];
  if (n > mo) { amh_fail(B, d, AMH_E_INCORRECT_MAX_OP); B.c_nops[c] = 0; return; }
  u64 start = mo - n + 1;
  if (start == 0) { amh_fail(B, d, AMH_E_COLLECT_MISSING_OPS); B.c_nops[c] = 0; return; }
  B.c_start_op[c] = start;
}

// ------------------------------------------------------------------------------------------
// K11  order the ops of each change by id (change_collector.rs:180).  The ids of a well-formed
//      change are exactly start_op..max_op of one actor, so the sorted position is ctr - start_op:
//      a scatter, not a sort.  A gap or collision means the document is not of that shape.
//      Thread per op t.
// ------------------------------------------------------------------------------------------
AMH_DEV void k_place(const BatchDev& B, u32 t) {
  u32 c = B.t_chg[t];
  if (c == AMH_EMPTY32) return;
  u32 d = B.c_doc[c];
  if (B.status[d]) return;
  u32 ctr, actor;
  op_id(B, t, ctr, actor);
  u64 start = B.c_start_op[c];
  if (ctr < start || ctr - start >= B.c_nops[c]) { amh_fail(B, d, AMH_E_UNSUPPORTED); return; }
  u32 slot = B.c_op_base[c] + (u32)(ctr - start);
  if (amh_atomic_cas(&B.ops_sorted[slot], AMH_EMPTY32, t) != AMH_EMPTY32) amh_fail(B, d, AMH_E_UNSUPPORTED);
}

// K11b  pred lists are laid out in change order (slot order of ops_sorted), so that the flattened pred
//       list of a change (opid_list.rs:52-78) is one contiguous run of pred_ids.  Thread per slot.
AMH_DEV void k_pred_slot_cnt(const BatchDev& B, u32 j) {
  u32 t = B.ops_sorted[j];
  B.j_pred_base[j] = t == AMH_EMPTY32 ? 0 : B.t_pred_cnt[t];
}
AMH_DEV void k_pred_slot_base(const BatchDev& B, u32 j) {
  u32 t = B.ops_sorted[j];
  if (t != AMH_EMPTY32) B.t_pred_base[t] = B.j_pred_base[j];
}

// K12  pred lists: scatter each succ edge to its target, then order each list by lamport id
//      (reconstruct_document.rs:299-301, 325).  Thread per succ entry / per op.
AMH_DEV void k_pred_fill(const BatchDev& B, u32 e) {
  u32 t = B.s_target[e];
  if (t == AMH_EMPTY32) return;
  u32 r = B.s_row[e];
  if (r == AMH_EMPTY32 || B.status[B.r_doc[r]]) return;
  u32 pos = B.t_pred_base[t] + amh_atomic_add(&B.t_pred_fill[t], 1);
  B.pred_ids[pos] = pack_id(B.r_id_ctr[r], B.r_id_actor[r]);
}
AMH_DEV void k_actor_mark(const BatchDev& B, u32 t);
AMH_DEV void k_pred_sort(const BatchDev& B, u32 t) {
  u32 n = B.t_pred_cnt[t];
  if (n >= 2) {
    u32 d = op_doc(B, t);
    if (B.status[d]) return;
    u64* v = B.pred_ids + B.t_pred_base[t];
#ifndef AMH_EMU
    if (n > AMH_BIG_LIST) big_push(B.big_n, B.big_items, B.big_cap, t, d);   // sorted by a block afterwards (g_big_pred)
    else
#endif
    for (u32 i = 1; i < n; i++) {
      u64 x = v[i];
      u32 j = i;
      while (j > 0 && lamport_cmp(B, d, (u32)(v[j - 1] >> 32), (u32)v[j - 1], (u32)(x >> 32), (u32)x) > 0) { v[j] = v[j - 1]; j--; }
      v[j] = x;
    }
  }
  k_actor_mark(B, t);   // same thread, same pred list: the actors this op contributes to its change's table
}

// op views used by the actor table and the encoder -------------------------------------------
struct OpKey { u8 kind; u32 actor, ctr, str_off, str_len; };  // kind 0 prop, 1 head, 2 elem
AMH_DEV OpKey op_key(const BatchDev& B, u32 t) {
  OpKey k;
  u32 r = op_key_row(B, t);
  k.actor = k.ctr = k.str_off = k.str_len = 0;
  if (B.r_key_len[r] != AMH_NULL32) { k.kind = 0; k.str_off = B.r_key_off[r]; k.str_len = B.r_key_len[r]; return k; }
  if (op_is_del(B, t) && B.r_insert[r]) { k.kind = 2; k.actor = B.r_id_actor[r]; k.ctr = B.r_id_ctr[r]; return k; }
  if (B.r_key_ctr[r] == 0) { k.kind = 1; return k; }
  k.kind = 2; k.actor = B.r_key_actor[r]; k.ctr = B.r_key_ctr[r];
  return k;
}

// K13  actors referenced by each change (change_actors.rs:74-97).  Thread per op t.
AMH_DEV void k_actor_mark(const BatchDev& B, u32 t) {
  u32 c = B.t_chg[t];
  if (c == AMH_EMPTY32) return;
  u32 d = B.c_doc[c];
  if (B.status[d]) return;
  u32* map = B.chg_actor_map + B.c_map_base[c];
  OpKey k = op_key(B, t);
  if (k.kind == 2) map[k.actor] = 1;
  u32 kr = op_key_row(B, t);
  if (B.r_obj_ctr[kr] != 0) map[B.r_obj_actor[kr]] = 1;
  u32 n = B.t_pred_cnt[t];
  const u64* pv = B.pred_ids + B.t_pred_base[t];
  for (u32 i = 0; i < n; i++) map[(u32)pv[i]] = 1;
}
// K14  change actor table: author first, the others in bytewise order (change_actors.rs:98-106).
//      Thread per change.
AMH_DEV void k_actor_rank(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  B.c_n_other[c] = 0;
  if (B.status[d]) return;
  u32 na = B.desc[d].n_actors, ab = B.actor_base[d];
  u32* map = B.chg_actor_map + B.c_map_base[c];
  u32 author = B.c_actor[c], next = 1;
  for (u32 k = 0; k < na; k++) {
    u32 a = B.actor_by_rank[ab + k];
    if (a == author) { map[a] = 0; continue; }
    if (map[a] == 1) map[a] = next++;
    else map[a] = AMH_EMPTY32;
  }
  B.c_n_other[c] = next - 1;
}

// ------------------------------------------------------------------------------------------
// K15  per-change column encode — change_op_columns.rs:146-251 driving rle.rs:8-141,
//      delta.rs:10-43, boolean.rs:11-51,105-138, key.rs:236-257, obj_id.rs:158-172,
//      opid_list.rs:244-254, value.rs:320-369,412-432; op views per op_as_changeop.rs:86-145.
//      Thread per (change, column).  Runs twice: size pass (out == null) and store pass.
// ------------------------------------------------------------------------------------------
// Writers.  ColWriter: bytes straight to memory (or size only when out == null), may widen a literal
// header in place.  SizeWriter: counts, and notes columns with a literal run of more than 64 values
// (their header is wider than the one byte reserved).  FastWriter: collects bytes in two registers
// and stores 16 at a time; it cannot widen a header, so columns SizeWriter flagged go through
// ColWriter instead (k_encode_store, restricted to those columns).
struct ColWriter {
  u8* out;   // null = size only
  u32 pos;
};
struct SizeWriter {
  u32 pos;
  u32 wide;
};
struct FastWriter {
  u8* out;   // null = this column is not written by this writer
  u32 pos, na;
  u64 a0, a1;
};
AMH_DEV void fw_init(FastWriter& w, u8* out) { w.out = out; w.pos = 0; w.na = 0; w.a0 = 0; w.a1 = 0; }
AMH_DEV void sw_init(SizeWriter& w) { w.pos = 0; w.wide = 0; }
AMH_DEV void amh_store16(u8* p, u64 lo, u64 hi) {
  struct __attribute__((packed)) P { u64 lo, hi; };
  P v; v.lo = lo; v.hi = hi;
  *(P*)p = v;
}
// x holds n (1..8) bytes, low byte first, upper bytes zero
AMH_DEV void fw_put(FastWriter& w, u64 x, u32 n) {
  if (!w.out) { w.pos += n; return; }
  u32 na = w.na;
  u64 carry = 0;
  if (na < 8) {
    w.a0 |= x << (8 * na);
    if (na + n > 8) w.a1 |= x >> (8 * (8 - na));
  } else {
    u32 k = na - 8;
    w.a1 |= x << (8 * k);
    if (k + n > 8) carry = x >> (8 * (8 - k));
  }
  na += n;
  w.pos += n;
  if (na >= 16) {
    amh_store16(w.out + (w.pos - na), w.a0, w.a1);
    w.a0 = carry; w.a1 = 0; na -= 16;
  }
  w.na = na;
}
AMH_DEV void fw_finish(FastWriter& w) {   // the 0..15 bytes still in the registers: at most four stores (8, 4, 2, 1)
  if (!w.out) return;
  u8* o = w.out + (w.pos - w.na);
  u32 na = w.na;
  u64 a = w.a0;
  struct __attribute__((packed)) P8 { u64 v; };
  struct __attribute__((packed)) P4 { u32 v; };
  struct __attribute__((packed)) P2 { unsigned short v; };
  if (na & 8) { ((P8*)o)->v = a; o += 8; a = w.a1; }
  if (na & 4) { ((P4*)o)->v = (u32)a; o += 4; a >>= 32; }
  if (na & 2) { ((P2*)o)->v = (unsigned short)a; o += 2; a >>= 16; }
  if (na & 1) *o = (u8)a;
  w.na = 0; w.a0 = 0; w.a1 = 0;
}
// spread the low 56 bits of v into eight 7-bit groups, one per byte
AMH_DEV u64 amh_leb_spread(u64 x) {
  x = (x & 0x000000000fffffffull) | ((x & 0x00fffffff0000000ull) << 4);
  x = (x & 0x00003fff00003fffull) | ((x & 0x0fffc0000fffc000ull) << 2);
  x = (x & 0x007f007f007f007full) | ((x & 0x3f803f803f803f80ull) << 1);
  return x;
}
AMH_DEV u32 amh_clz64(u64 v) { return v ? (u32)__builtin_clzll(v) : 64; }
AMH_DEV u32 uleb_size_fast(u64 v) { u32 bits = 64 - amh_clz64(v | 1); return (bits + 6) / 7; }
AMH_DEV u32 sleb_size_fast(i64 v) { u64 m = v < 0 ? ~(u64)v : (u64)v; u32 bits = 65 - amh_clz64(m); return (bits + 6) / 7; }
AMH_DEV u64 amh_cont_bits(u32 nb) { return nb >= 8 ? 0x0080808080808080ull : (0x8080808080808080ull & (((u64)1 << (8 * (nb - 1))) - 1)); }
AMH_DEV void cw_uleb(FastWriter& w, u64 v) {
  u32 nb = uleb_size_fast(v);
  if (nb <= 8) { fw_put(w, amh_leb_spread(v) | amh_cont_bits(nb), nb); return; }
  fw_put(w, amh_leb_spread(v) | 0x8080808080808080ull, 8);
  u64 rest = v >> 56;
  u32 rb = nb - 8;
  fw_put(w, amh_leb_spread(rest) | amh_cont_bits(rb), rb);
}
#define AMH_CV_INLINE 0x80000000u
AMH_DEV u64 amh_low_bytes(u32 n) { return n >= 8 ? ~(u64)0 : (((u64)1 << (8 * n)) - 1); }   // mask of the low n bytes
AMH_DEV u64 sleb_bytes8(i64 v, u32 nb) {   // the nb <= 8 bytes of sleb128(v), low byte first
  u64 m = nb == 8 ? (u64)v & 0x00ffffffffffffffull : (u64)v & (((u64)1 << (7 * nb)) - 1);
  return amh_leb_spread(m) | amh_cont_bits(nb);
}
AMH_DEV void cw_sleb(FastWriter& w, i64 v) {
  u32 nb = sleb_size_fast(v);
  if (nb <= 8) {
    fw_put(w, sleb_bytes8(v, nb), nb);
    return;
  }
  fw_put(w, amh_leb_spread((u64)v & 0x00ffffffffffffffull) | 0x8080808080808080ull, 8);
  u64 g8 = (u64)(v >> 56) & 0x7f, g9 = (u64)(v >> 63) & 0x7f;
  if (nb == 9) fw_put(w, g8, 1); else fw_put(w, (g8 | 0x80) | (g9 << 8), 2);
}
AMH_DEV void cw_bytes(FastWriter& w, const u8* p, u32 n) {
  if (!w.out) { w.pos += n; return; }
  u32 i = 0;
  for (; i + 8 <= n; i += 8) fw_put(w, amh_peek8_any(p + i), 8);
  if (i < n) { u32 r = n - i; fw_put(w, amh_peek8_any(p + i) & (((u64)1 << (8 * r)) - 1), r); }
}
AMH_DEV void cw_inline(FastWriter& w, u64 bytes, u32 n) { if (n) fw_put(w, bytes, n); }   // n <= 8 bytes held in a register
AMH_DEV void cw_inline(SizeWriter& w, u64, u32 n) { w.pos += n; }
AMH_DEV void cw_uleb(SizeWriter& w, u64 v) { w.pos += uleb_size_fast(v); }
AMH_DEV void cw_sleb(SizeWriter& w, i64 v) { w.pos += sleb_size_fast(v); }
AMH_DEV void cw_bytes(SizeWriter& w, const u8*, u32 n) { w.pos += n; }
AMH_DEV void cw_uleb(ColWriter& w, u64 v) { w.pos += wr_uleb(w.out ? w.out + w.pos : (u8*)0, v); }
AMH_DEV void cw_sleb(ColWriter& w, i64 v) { w.pos += wr_sleb(w.out ? w.out + w.pos : (u8*)0, v); }
AMH_DEV void cw_bytes(ColWriter& w, const u8* p, u32 n) {
  if (w.out) for (u32 i = 0; i < n; i++) w.out[w.pos + i] = p[i];
  w.pos += n;
}
AMH_DEV void cw_inline(ColWriter& w, u64 bytes, u32 n) {
  if (w.out) for (u32 i = 0; i < n; i++) w.out[w.pos + i] = (u8)(bytes >> (8 * i));
  w.pos += n;
}
// literal-run header: one byte reserved at open, written at close (rle.rs:112-119, 128-133)
AMH_DEV u32 cw_lit_open(ColWriter& w) { u32 at = w.pos; w.pos += 1; return at; }
AMH_DEV u32 cw_lit_open(SizeWriter& w) { u32 at = w.pos; w.pos += 1; return at; }
AMH_DEV u32 cw_lit_open(FastWriter& w) { u32 at = w.pos; fw_put(w, 0, 1); return at; }
AMH_DEV void cw_lit_close(ColWriter& w, u32 at, u64 cnt) {
  i64 hdr = -(i64)cnt;
  u32 h = sleb_size(hdr);
  if (h > 1) {
    if (w.out) for (u32 i = w.pos; i > at + 1; i--) w.out[i - 1 + (h - 1)] = w.out[i - 1];
    w.pos += h - 1;
  }
  if (w.out) wr_sleb(w.out + at, hdr);
}
AMH_DEV void cw_lit_close(SizeWriter& w, u32, u64 cnt) {
  u32 h = sleb_size_fast(-(i64)cnt);
  if (h > 1) { w.pos += h - 1; w.wide = 1; }
}
AMH_DEV void cw_lit_close(FastWriter& w, u32 at, u64 cnt) {
  u32 h = sleb_size_fast(-(i64)cnt);
  if (!w.out) { if (h > 1) w.pos += h - 1; return; }
  if (h == 1) {
    u64 hdr = (u64)(-(i64)cnt) & 0x7f;
    u32 first = w.pos - w.na;   // oldest byte still in the registers
    if (at >= first) {
      u32 k = at - first;
      if (k < 8) w.a0 |= hdr << (8 * k); else w.a1 |= hdr << (8 * (k - 8));
    } else {
      w.out[at] = (u8)hdr;
    }
    return;
  }
  // more than 64 values: the header is wider than the byte reserved for it.  The run's bytes are already in place
  // behind that byte: drain the registers, move the run up by h - 1 (highest bytes first, 8 at a time) and write
  // the header in front of it — once per long run, instead of sending the whole column through the byte-wise writer.
  fw_finish(w);
  u32 delta = h - 1, lo = at + 1, i = w.pos;
  struct __attribute__((packed)) P8 { u64 v; };
  while (i - lo >= 8) { u64 v = ((const P8*)(w.out + i - 8))->v; ((P8*)(w.out + i - 8 + delta))->v = v; i -= 8; }
  while (i > lo) { w.out[i - 1 + delta] = w.out[i - 1]; i--; }
  wr_sleb(w.out + at, -(i64)cnt);
  w.pos += delta;
}

// RLE state machine restated with O(1) state: a literal run's values are streamed behind a
// one-byte header slot which is widened in place if the run turns out longer than 64 values.
// KIND 0 = u64, 1 = i64, 2 = string (value = pointer + length)
struct RleVal { u64 v; const u8* s; u32 n; };
// KIND 2: s == null means the string's n <= 8 bytes are in v (change view, short keys)
template <int KIND> AMH_DEV bool rv_eq(const RleVal& a, const RleVal& b) {
  if (KIND != 2) return a.v == b.v;
  if (a.n != b.n) return false;
  if (!a.s || !b.s) {
    u64 x = a.s ? amh_peek8_any(a.s) & amh_low_bytes(a.n) : a.v, y = b.s ? amh_peek8_any(b.s) & amh_low_bytes(b.n) : b.v;
    return x == y;
  }
  if (a.s == b.s) return true;
  for (u32 i = 0; i < a.n; i += 8) {
    u64 d = amh_peek8_any(a.s + i) ^ amh_peek8_any(b.s + i);
    u32 r = a.n - i;
    if (r < 8) d &= ((u64)1 << (8 * r)) - 1;
    if (d) return false;
  }
  return true;
}
template <int KIND, class W> AMH_DEV void rv_write(W& w, const RleVal& x) {
  if (KIND == 0) cw_uleb(w, x.v);
  else if (KIND == 1) cw_sleb(w, (i64)x.v);
  else { cw_uleb(w, (u64)x.n); if (x.s) cw_bytes(w, x.s, x.n); else cw_inline(w, x.v, x.n); }
}
template <int KIND> struct RleEnc {
  int st;        // 0 Empty 1 InitialNullRun 2 NullRun 3 LiteralRun 4 LoneVal 5 Run
  u64 n;         // run / null-run length
  RleVal val;    // LoneVal / Run value / LiteralRun "last"
  u32 lit_pos;   // header slot of the open literal run
  u64 lit_cnt;   // values already streamed into it
};
template <int KIND> AMH_DEV void re_init(RleEnc<KIND>& e) { e.st = 0; e.n = 0; e.lit_pos = 0; e.lit_cnt = 0; e.val.v = 0; e.val.s = 0; e.val.n = 0; }
template <int KIND, class W> AMH_DEV void re_lit_open(W& w, RleEnc<KIND>& e) { e.lit_pos = cw_lit_open(w); e.lit_cnt = 0; }
template <int KIND, class W> AMH_DEV void re_lit_close(W& w, RleEnc<KIND>& e) { cw_lit_close(w, e.lit_pos, e.lit_cnt); }
// One step of the encoder (rle.rs:31-46 finish, 71-90 append_null, 92-126 append_value).  The state machine only
// decides WHAT is written (a set of RE_* actions on the state before the step); the bytes are produced by one copy
// of the writer code below it.  On the device that matters twice: lanes in different states run the writer together
// instead of one state after the other, and the code is a fraction of the size it has with the writer inlined into
// every transition.
enum { RE_OPEN = 1, RE_PUSH = 2, RE_CLOSE = 4, RE_RUN = 8, RE_NULLS = 16 };
enum { RE_VALUE = 0, RE_NULL = 1, RE_FINISH = 2 };
template <int KIND, class W> AMH_DEV void re_step(W& w, RleEnc<KIND>& e, const RleVal& x, int mode) {
  const int st = e.st;
  u32 f = 0;
  int nst = st;
  u64 nn = e.n;
  bool take = false;
  if (mode == RE_VALUE) {
    const bool eq = (st == 3 || st == 4 || st == 5) && rv_eq<KIND>(e.val, x);
    switch (st) {
      case 0: nst = 4; take = true; break;
      case 4: if (eq) { nst = 5; nn = 2; } else { f = RE_OPEN | RE_PUSH; nst = 3; take = true; } break;
      case 5: if (eq) nn++; else { f = RE_RUN; nst = 4; take = true; } break;
      case 3: if (eq) { f = RE_CLOSE; nst = 5; nn = 2; } else { f = RE_PUSH; take = true; } break;
      default: f = RE_NULLS; nst = 4; take = true; break;   // 1, 2
    }
  } else if (mode == RE_NULL) {
    switch (st) {
      case 0: nst = 1; nn = 1; break;
      case 1: case 2: nn++; break;
      case 4: f = RE_OPEN | RE_PUSH | RE_CLOSE; nst = 2; nn = 1; break;
      case 5: f = RE_RUN; nst = 2; nn = 1; break;
      default: f = RE_PUSH | RE_CLOSE; nst = 2; nn = 1; break;   // 3
    }
  } else {
    switch (st) {
      case 2: f = RE_NULLS; break;
      case 4: f = RE_OPEN | RE_PUSH | RE_CLOSE; break;
      case 5: f = RE_RUN; break;
      case 3: f = RE_PUSH | RE_CLOSE; break;
      default: break;   // Empty, InitialNullRun: nothing (rle.rs:33)
    }
  }
  if (f) {
    if (f & RE_OPEN) re_lit_open(w, e);
    if (f & RE_RUN) cw_sleb(w, (i64)e.n);
    if (f & RE_NULLS) { cw_sleb(w, (i64)0); cw_uleb(w, e.n); }
    if (f & (RE_PUSH | RE_RUN)) { rv_write<KIND>(w, e.val); if (f & RE_PUSH) e.lit_cnt++; }
    if (f & RE_CLOSE) re_lit_close(w, e);
  }
  e.st = nst;
  e.n = nn;
  if (take) e.val = x;
}
template <int KIND, class W> AMH_DEV void re_null(W& w, RleEnc<KIND>& e) { RleVal x; x.v = 0; x.s = 0; x.n = 0; re_step(w, e, x, RE_NULL); }
template <int KIND, class W> AMH_DEV void re_value(W& w, RleEnc<KIND>& e, const RleVal& x) { re_step(w, e, x, RE_VALUE); }
template <int KIND, class W> AMH_DEV void re_finish(W& w, RleEnc<KIND>& e) { RleVal x; x.v = 0; x.s = 0; x.n = 0; re_step(w, e, x, RE_FINISH); }
struct DeltaEnc { RleEnc<1> r; i64 abs; };
template <class W> AMH_DEV void de_step(W& w, DeltaEnc& e, i64 v, bool isnull) {  // delta.rs:23-27, 30-32
  i64 dlt = 0;
  if (!isnull) {
    if (__builtin_sub_overflow(v, e.abs, &dlt)) dlt = v > e.abs ? INT64_MAX : INT64_MIN;
    e.abs = v;
  }
  RleVal x; x.v = (u64)dlt; x.s = 0; x.n = 0;
  re_step<1>(w, e.r, x, isnull ? RE_NULL : RE_VALUE);
}
template <class W> AMH_DEV void de_value(W& w, DeltaEnc& e, i64 v) { de_step(w, e, v, false); }

// the value an op carries into a change (op_as_changeop.rs:94-103): kind 0 = null, 1 = stored
// bytes as they are, 2 = integer re-encoded as Int (increments)
struct OpVal { int kind; u64 meta; const u8* raw; u32 raw_len; i64 i; };
AMH_DEV OpVal op_val(const BatchDev& B, u32 t) {
  OpVal v; v.kind = 0; v.meta = 0; v.raw = 0; v.raw_len = 0; v.i = 0;
  if (op_is_del(B, t)) return v;
  u8 a = B.r_action[t];
  if (is_make(a) || a == 3) return v;
  if (a == 7 && B.r_mark_len[t] == AMH_NULL32) return v;
  if (a == 5) {
    val_int(B, t, v.i);
    v.kind = 2;
    v.meta = ((u64)sleb_size(v.i) << 4) | 4;
    return v;
  }
  u64 m = B.r_vmeta[t];
  u32 vt = (u32)(m & 15);
  v.kind = 1;
  if (vt <= 2) { v.meta = vt; return v; }
  const DocDesc& D = B.desc[B.r_doc[t]];
  v.meta = m;
  v.raw = B.arena + D.chunk_off + D.ocol[OC_VAL_RAW].off + B.r_vraw_off[t];
  v.raw_len = (u32)(m >> 4);
  return v;
}

// encode one column of one change; returns its length
AMH_DEV u32 encode_change_col(const BatchDev& B, u32 c, u32 col, u8* out) {
  u32 d = B.c_doc[c];
  const u8* chunk = B.arena + B.desc[d].chunk_off;
  const u32* ops = B.ops_sorted + B.c_op_base[c];
  const u32* amap = B.chg_actor_map + B.c_map_base[c];
  u32 n = B.c_nops[c];
  ColWriter w; w.out = out; w.pos = 0;
  RleVal x; x.v = 0; x.s = 0; x.n = 0;
  switch (col) {
    case 0: case 1: {  // objActor / objCtr (obj_id.rs:158-172); both vanish when every op is on the root
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 kr = op_key_row(B, ops[i]);
        if (B.r_obj_ctr[kr] == 0) re_null(w, e);
        else { x.v = col == 0 ? amap[B.r_obj_actor[kr]] : B.r_obj_ctr[kr]; re_value(w, e, x); }
      }
      re_finish(w, e);
      break;
    }
    case 2: {  // keyActor
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        OpKey k = op_key(B, ops[i]);
        if (k.kind == 2) { x.v = amap[k.actor]; re_value(w, e, x); } else re_null(w, e);
      }
      re_finish(w, e);
      break;
    }
    case 3: {  // keyCtr (delta; HEAD = 0)
      DeltaEnc e; re_init(e.r); e.abs = 0;
      for (u32 i = 0; i < n; i++) {
        OpKey k = op_key(B, ops[i]);
        if (k.kind == 0) re_null(w, e.r);
        else de_value(w, e, k.kind == 1 ? 0 : (i64)k.ctr);
      }
      re_finish(w, e.r);
      break;
    }
    case 4: {  // keyStr
      RleEnc<2> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        OpKey k = op_key(B, ops[i]);
        if (k.kind == 0) { x.s = chunk + k.str_off; x.n = k.str_len; re_value(w, e, x); } else re_null(w, e);
      }
      re_finish(w, e);
      break;
    }
    case 5: {  // insert (boolean.rs:35-50)
      bool last = false; u64 cnt = 0;
      for (u32 i = 0; i < n; i++) {
        u32 t = ops[i];
        bool v = op_is_del(B, t) ? false : B.r_insert[t] != 0;
        if (v == last) cnt++; else { cw_uleb(w, cnt); last = v; cnt = 1; }
      }
      if (cnt > 0) cw_uleb(w, cnt);
      break;
    }
    case 6: {  // action
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) { u32 t = ops[i]; x.v = op_is_del(B, t) ? 3 : B.r_action[t]; re_value(w, e, x); }
      re_finish(w, e);
      break;
    }
    case 7: {  // valMeta
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) { OpVal v = op_val(B, ops[i]); x.v = v.meta; re_value(w, e, x); }
      re_finish(w, e);
      break;
    }
    case 8: {  // valRaw
      for (u32 i = 0; i < n; i++) {
        OpVal v = op_val(B, ops[i]);
        if (v.kind == 1) cw_bytes(w, v.raw, v.raw_len);
        else if (v.kind == 2) cw_sleb(w, v.i);
      }
      break;
    }
    case 9: {  // predNum
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) { x.v = B.t_pred_cnt[ops[i]]; re_value(w, e, x); }
      re_finish(w, e);
      break;
    }
    case 10: {  // predActor over the flattened lists
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 t = ops[i], pn = B.t_pred_cnt[t];
        const u64* pv = B.pred_ids + B.t_pred_base[t];
        for (u32 k = 0; k < pn; k++) { x.v = amap[(u32)pv[k]]; re_value(w, e, x); }
      }
      re_finish(w, e);
      break;
    }
    case 11: {  // predCtr: one delta chain across the flattened lists (opid_list.rs:69-72)
      DeltaEnc e; re_init(e.r); e.abs = 0;
      for (u32 i = 0; i < n; i++) {
        u32 t = ops[i], pn = B.t_pred_cnt[t];
        const u64* pv = B.pred_ids + B.t_pred_base[t];
        for (u32 k = 0; k < pn; k++) de_value(w, e, (i64)(pv[k] >> 32));
      }
      re_finish(w, e.r);
      break;
    }
    case 12: {  // expand (MaybeBoolean: nothing unless some op is a mark with expand, boolean.rs:131-137)
      bool any = false;
      for (u32 i = 0; i < n && !any; i++) { u32 t = ops[i]; any = !op_is_del(B, t) && B.r_action[t] == 7 && B.r_expand[t]; }
      if (!any) break;
      bool last = false; u64 cnt = 0;
      for (u32 i = 0; i < n; i++) {
        u32 t = ops[i];
        bool v = !op_is_del(B, t) && B.r_action[t] == 7 && B.r_expand[t];
        if (v == last) cnt++; else { cw_uleb(w, cnt); last = v; cnt = 1; }
      }
      if (cnt > 0) cw_uleb(w, cnt);
      break;
    }
    case 13: {  // markName
      RleEnc<2> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 t = ops[i];
        if (!op_is_del(B, t) && B.r_action[t] == 7 && B.r_mark_len[t] != AMH_NULL32) { x.s = chunk + B.r_mark_off[t]; x.n = B.r_mark_len[t]; re_value(w, e, x); }
        else re_null(w, e);
      }
      re_finish(w, e);
      break;
    }
  }
  return w.pos;
}

// K14b  change view: the fields the column encoders need, gathered once per op into slot order so
//       that the encoders stream them (op_as_changeop.rs:15-146).  Thread per slot j.
AMH_DEV void k_change_view(const BatchDev& B, u32 j) {
  u32 t = B.ops_sorted[j];
  if (t == AMH_EMPTY32) return;
  u32 c = B.t_chg[t];
  if (c == AMH_EMPTY32 || B.status[B.c_doc[c]]) return;
  const u32* amap = B.chg_actor_map + B.c_map_base[c];
  u32 kr = op_key_row(B, t);
  u32 oc = B.r_obj_ctr[kr];
  // the fields of a column group are interleaved per op, so that the encoder's tile loads are contiguous runs
  // (8 ops x 8 / 16 bytes) instead of 32-byte pieces of 12 separate arrays
  u32* g_obj = B.cv[0] + 2 * (u64)j;
  u32* g_key = B.cv[2] + 4 * (u64)j;
  u32* g_misc = B.cv[6] + 2 * (u64)j;
  u32* g_val = B.cv[8] + 4 * (u64)j;
  g_obj[0] = oc ? amap[B.r_obj_actor[kr]] : AMH_NULL32;
  g_obj[1] = oc;
  const u8* chunk = B.arena + B.desc[B.c_doc[c]].chunk_off;
  OpKey ky = op_key(B, t);
  if (ky.kind == 0 && ky.str_len <= 8) {
    // a short map key travels in the view itself (bytes, AMH_CV_INLINE | length): the encoders, one lane per
    // change, compare and write it out of registers instead of going back to the chunk for every op
    u64 kb = ky.str_len ? amh_peek8_any(chunk + ky.str_off) & amh_low_bytes(ky.str_len) : 0;
    g_key[0] = (u32)kb; g_key[1] = (u32)(kb >> 32);
    g_key[2] = ky.str_off;
    g_key[3] = AMH_CV_INLINE | ky.str_len;
  } else {
    g_key[0] = ky.kind == 2 ? amap[ky.actor] : AMH_NULL32;
    g_key[1] = ky.kind == 0 ? AMH_NULL32 : (ky.kind == 1 ? 0 : ky.ctr);
    g_key[2] = ky.str_off;
    g_key[3] = ky.kind == 0 ? ky.str_len : AMH_NULL32;
  }
  bool del = op_is_del(B, t);
  u32 action = del ? 3 : B.r_action[t];
  u32 ins = del ? 0 : (B.r_insert[t] != 0);
  u32 ex = (!del && action == 7 && B.r_expand[t]) ? 1 : 0;
  g_misc[0] = ins | (ex << 1) | (action << 8);
  g_misc[1] = B.t_pred_cnt[t];
  OpVal v = op_val(B, t);
  g_val[0] = (u32)v.meta;
  g_val[1] = (u32)(v.meta >> 32) | ((u32)v.kind << 8);
  // value bytes of up to eight bytes (every integer below 2^55 in its LEB128 form, short strings) are inlined as
  // kind 3 with the length in bits 16..19; longer ones stay references into the chunk (kind 1) / integers (kind 2)
  if (v.kind == 2) {
    u32 nb = sleb_size_fast(v.i);
    if (nb <= 8) {
      u64 eb = sleb_bytes8(v.i, nb);
      g_val[1] = (u32)(v.meta >> 32) | (3u << 8) | (nb << 16);
      g_val[2] = (u32)eb; g_val[3] = (u32)(eb >> 32);
    } else { g_val[2] = (u32)(u64)v.i; g_val[3] = (u32)((u64)v.i >> 32); }
  } else if (v.kind == 1 && v.raw_len && v.raw_len <= 8) {
    u64 eb = amh_peek8_any(v.raw) & amh_low_bytes(v.raw_len);
    g_val[1] = (u32)(v.meta >> 32) | (3u << 8) | (v.raw_len << 16);
    g_val[2] = (u32)eb; g_val[3] = (u32)(eb >> 32);
  } else {
    g_val[2] = (v.kind == 1 && v.raw_len) ? B.desc[B.r_doc[t]].ocol[OC_VAL_RAW].off + B.r_vraw_off[t] : 0;
    g_val[3] = v.raw_len;
  }
}

AMH_DEV u32 chg_spec(u32 col) {
  const u32 specs[AMH_NCOLS] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x34, 0x42, 0x56, 0x57, 0x70, 0x71, 0x73, 0x94, 0xA5};
  return specs[col];
}

AMH_DEV void k_encode_size(const BatchDev& B, u32 gid) {
  u32 col = gid / B.tot_chg, c = gid % B.tot_chg;   // column-major: a wave encodes one column of 64 changes
  if (col >= AMH_NCOLS) return;
  B.c_col_len[(u64)c * AMH_NCOLS + col] = 0;
  if (B.status[B.c_doc[c]]) return;
  B.c_col_len[(u64)c * AMH_NCOLS + col] = encode_change_col(B, c, col, (u8*)0);
}

// K16  chunk layout of each change — storage/change.rs:460-488, chunk.rs:181-221.  Thread per change.
AMH_DEV void k_chg_layout(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  B.c_out_len[c] = 0;
  B.c_body_len[c] = 0;
  B.c_cols_off[c] = 0;
  if (B.c_npred) B.c_npred[c] = 0;
  if (B.status[d]) return;
  if (B.c_npred && B.c_nops[c]) B.c_npred[c] = B.j_pred_base[B.c_op_base[c] + B.c_nops[c]] - B.j_pred_base[B.c_op_base[c]];
  u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
  if (cl[0] == 0) cl[1] = 0;     // ObjIdRange::encode -> None (obj_id.rs:79-81)
  if (cl[10] == 0) cl[11] = 0;   // pred actor/ctr only when there are entries (change_op_columns.rs:304-315)
  u32 ab = B.actor_base[d], na = B.desc[d].n_actors;
  const u32* amap = B.chg_actor_map + B.c_map_base[c];
  u64 len = 0;
  len += uleb_size(B.c_ndeps[c]) + 32ull * B.c_ndeps[c];
  u32 al = B.actor_len[ab + B.c_actor[c]];
  len += uleb_size(al) + al;
  len += uleb_size(B.c_seq[c]) + uleb_size(B.c_start_op[c]) + sleb_size(B.c_time[c]);
  u32 ml = B.c_msg_len[c] == AMH_NULL32 ? 0 : B.c_msg_len[c];
  len += uleb_size(ml) + ml;
  len += uleb_size(B.c_n_other[c]);
  for (u32 a = 0; a < na; a++)
    if (amap[a] != AMH_EMPTY32 && amap[a] != 0) len += uleb_size(B.actor_len[ab + a]) + B.actor_len[ab + a];
  u32 ncols = 0;
  u64 meta = 0, data = 0;
  for (u32 k = 0; k < AMH_NCOLS; k++)
    if (cl[k]) { ncols++; meta += uleb_size(chg_spec(k)) + uleb_size(cl[k]); data += cl[k]; }
  len += uleb_size(ncols) + meta;
  u64 cols_off = len;
  len += data + B.c_extra_len[c];
  if (len > 0x7FFFFFF0ull) { amh_fail(B, d, AMH_E_TOO_LARGE); return; }
  u32 hdr = 4 + 4 + 1 + uleb_size(len);
  B.c_body_len[c] = (u32)len;
  B.c_out_len[c] = hdr + (u32)len;
  B.c_cols_off[c] = hdr + (u32)cols_off;
}

AMH_DEV void k_encode_store(const BatchDev& B, u32 gid) {
  u32 col = gid / B.tot_chg, c = gid % B.tot_chg;
  if (col >= AMH_NCOLS) return;
  if (B.status[B.c_doc[c]]) return;
  const u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
  if (cl[col] == 0) return;
#ifndef AMH_EMU
  if (!((B.c_wide[c] >> col) & 1)) return;   // written by g_encode_lds
#endif
  u64 off = B.c_out_off[c] + B.c_cols_off[c];
  for (u32 k = 0; k < col; k++) off += cl[k];
  encode_change_col(B, c, col, B.out_arena + off);
}

// ------------------------------------------------------------------------------------------
// K17  change chunks: header, hash, heads — storage/change.rs:460-495, chunk.rs:181-193, 286-293;
//      change_collector.rs:121-133, 163-178.
//   k_chg_header  (thread per change)  everything of the chunk but the dep hashes and the checksum
//   k_chg_hash    (thread per change of dependency level B.cur_level)  dep hashes (sorted), SHA-256
//   k_chg_hash_seq(thread per document whose dependency chains are deeper than AMH_LEVEL_CAP) the same, in file order
//   k_heads       (thread per document)  derived heads against the stored ones
// A change's level is 1 + the deepest level among its deps (k_decode_chg), so all changes of one
// level can be hashed at once: their deps were hashed by earlier launches.
// ------------------------------------------------------------------------------------------
#define AMH_LEVEL_CAP 48u
AMH_DEV void k_chg_header(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  const u8* chunk = B.arena + D.chunk_off;
  u32 ab = B.actor_base[d], na = D.n_actors;
  u8* out = B.out_arena + B.c_out_off[c];
  u32 body_len = B.c_body_len[c];
  u32 nd = B.c_ndeps[c];
  // both runs of header bytes go through the 16-byte register writer: magic, checksum slot (k_chg_hash), type,
  // length, dep count | (dep hashes: k_chg_hash) | actor .. column metadata
  FastWriter w;
  fw_init(w, out);
  fw_put(w, 0x00000000834a6f85ull, 8);
  fw_put(w, 1, 1);
  cw_uleb(w, body_len);
  cw_uleb(w, nd);
  fw_finish(w);
  u32 pos = w.pos + 32 * nd;
  fw_init(w, out + pos);
  u32 a = ab + B.c_actor[c];
  cw_uleb(w, B.actor_len[a]);
  cw_bytes(w, chunk + B.actor_off[a], B.actor_len[a]);
  cw_uleb(w, B.c_seq[c]);
  cw_uleb(w, B.c_start_op[c]);
  cw_sleb(w, B.c_time[c]);
  u32 ml = B.c_msg_len[c] == AMH_NULL32 ? 0 : B.c_msg_len[c];
  cw_uleb(w, ml);
  cw_bytes(w, chunk + B.c_msg_off[c], ml);
  cw_uleb(w, B.c_n_other[c]);
  const u32* amap = B.chg_actor_map + B.c_map_base[c];
  for (u32 k = 0; k < na; k++) {  // other actors in bytewise order
    u32 oa = B.actor_by_rank[ab + k];
    if (amap[oa] == AMH_EMPTY32 || amap[oa] == 0) continue;
    cw_uleb(w, B.actor_len[ab + oa]);
    cw_bytes(w, chunk + B.actor_off[ab + oa], B.actor_len[ab + oa]);
  }
  const u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
  u32 ncols = 0, data = 0;
  for (u32 k = 0; k < AMH_NCOLS; k++) if (cl[k]) ncols++;
  cw_uleb(w, ncols);
  for (u32 k = 0; k < AMH_NCOLS; k++)
    if (cl[k]) { cw_uleb(w, chg_spec(k)); cw_uleb(w, cl[k]); data += cl[k]; }
  fw_finish(w);
  pos += w.pos + data;  // column data was written by the store pass at c_cols_off
  u32 el = B.c_extra_len[c];
  if (el) {
    fw_init(w, out + pos);
    cw_bytes(w, chunk + B.c_extra_off[c], el);
    fw_finish(w);
  }
}
// a 32-byte hash as four big-endian words: bytewise order = lexicographic order of the words
struct Hash4 { u64 w[4]; };
AMH_DEV Hash4 hash4_load(const u8* p) {   // any alignment
  struct __attribute__((packed)) P8 { u64 v; };
  Hash4 h;
  for (int i = 0; i < 4; i++) h.w[i] = ((const P8*)p)[i].v;
  return h;
}
AMH_DEV void hash4_store(u8* p, const Hash4& h) {
  struct __attribute__((packed)) P8 { u64 v; };
  for (int i = 0; i < 4; i++) ((P8*)p)[i].v = h.w[i];
}
AMH_DEV bool hash4_greater(const Hash4& a, const Hash4& b) {
  for (int i = 0; i < 4; i++) {
    u64 x = __builtin_bswap64(a.w[i]), y = __builtin_bswap64(b.w[i]);
    if (x != y) return x > y;
  }
  return false;
}
AMH_DEV void chg_hash_one(const BatchDev& B, u32 c, u32 cb) {
  u8* out = B.out_arena + B.c_out_off[c];
  u32 body_len = B.c_body_len[c];
  u32 hdr = B.c_out_len[c] - body_len;
  u32 nd = B.c_ndeps[c];
  u8* deps = out + hdr + uleb_size(nd);
  // deps: hashes of earlier changes, sorted (change_collector.rs:178; change.rs:335-341)
  for (u32 k = 0; k < nd; k++) {
    Hash4 h = hash4_load(B.c_hash + (u64)(cb + B.dep_idx[B.c_dep_base[c] + k]) * 32);
    u32 j = k;
    while (j > 0) {
      Hash4 prev = hash4_load(deps + 32 * (j - 1));
      if (!hash4_greater(prev, h)) break;
      hash4_store(deps + 32 * j, prev);
      j--;
    }
    hash4_store(deps + 32 * j, h);
  }
  u8* h = B.c_hash + (u64)c * 32;
  sha256_contig(out + 8, (u64)(hdr - 8) + body_len, h);
  out[4] = h[0]; out[5] = h[1]; out[6] = h[2]; out[7] = h[3];
}
// the changes of document d are hashed by the host (amh_host.cpp, big_hash_changes): a large chunk, or a long chain
AMH_DEV bool chg_host_hashed(const BatchDev& B, u32 d) {
  if (B.sha_host_min && B.desc[d].chunk_len >= B.sha_host_min) return true;
  return B.host_chain_min && B.doc_level[d] >= AMH_LEVEL_CAP && B.cnt[d].n_chg >= B.host_chain_min;
}
// bucket of a change in the level sort: its level, or AMH_LEVEL_CAP = "not hashed by k_chg_hash"
AMH_DEV u32 chg_level_bucket(const BatchDev& B, u32 c) {
  u32 d = B.c_doc[c];
  if (B.status[d] || B.doc_level[d] >= AMH_LEVEL_CAP) return AMH_LEVEL_CAP;
  if (chg_host_hashed(B, d)) return AMH_LEVEL_CAP;
  return B.c_level[c];
}
// changes hashed by the host (big_hash_changes: documents of AMH_HOST_SHA_MIN bytes and more): the host sends the hashes
// back, the dependency hashes and the checksum go into the chunks here.  Thread per change.
AMH_DEV void k_chg_patch(const BatchDev& B, u32 c) {
  const u32 d = B.c_doc[c];
  if (B.status[d] || !chg_host_hashed(B, d)) return;
  const u32 cb = B.chg_base[d];
  u8* out = B.out_arena + B.c_out_off[c];
  const u32 hdr = B.c_out_len[c] - B.c_body_len[c], nd = B.c_ndeps[c];
  u8* deps = out + hdr + uleb_size(nd);
  for (u32 k = 0; k < nd; k++) {
    Hash4 h = hash4_load(B.c_hash + (u64)(cb + B.dep_idx[B.c_dep_base[c] + k]) * 32);
    u32 j = k;
    while (j > 0) {
      Hash4 prev = hash4_load(deps + 32 * (j - 1));
      if (!hash4_greater(prev, h)) break;
      hash4_store(deps + 32 * j, prev);
      j--;
    }
    hash4_store(deps + 32 * j, h);
  }
  const u8* h = B.c_hash + (u64)c * 32;
  out[4] = h[0]; out[5] = h[1]; out[6] = h[2]; out[7] = h[3];
}
AMH_DEV void k_chg_hash(const BatchDev& B, u32 i) {   // i-th change of the current level (lvl_order is grouped by level)
  u32 c = B.lvl_order[B.cur_level + i];
  chg_hash_one(B, c, B.chg_base[B.c_doc[c]]);
}
AMH_DEV void k_chg_hash_seq(const BatchDev& B, u32 d) {
  if (B.status[d] || B.doc_level[d] < AMH_LEVEL_CAP) return;
  if (chg_host_hashed(B, d)) return;
  u32 cb = B.chg_base[d], nchg = B.cnt[d].n_chg;
  for (u32 i = 0; i < nchg; i++) chg_hash_one(B, cb + i, cb);
}
AMH_DEV void k_heads(const BatchDev& B, u32 d) {
  if (B.status[d]) return;
  const DocDesc& D = B.desc[d];
  const u8* chunk = B.arena + D.chunk_off;
  u32 cb = B.chg_base[d], nchg = B.cnt[d].n_chg;
  // heads: hashes nobody depends on, compared as a set with the stored heads
  // (change_collector.rs:121-133; reconstruct_document.rs:211-221)
  u32 derived = 0;
  bool ok = true;
  for (u32 i = 0; i < nchg && ok; i++) {
    if (B.c_is_dep[cb + i]) continue;
    const u8* h = B.c_hash + (u64)(cb + i) * 32;
    derived++;
    bool found = false;
    for (u32 k = 0; k < D.n_heads && !found; k++) found = amh_memcmp(chunk + D.heads_off + 32 * k, 32, h, 32) == 0;
    if (!found) ok = false;
  }
  // stored heads must all be derived heads as well
  u32 distinct_stored = 0;
  for (u32 k = 0; k < D.n_heads && ok; k++) {
    const u8* h = chunk + D.heads_off + 32 * k;
    bool dupe = false;
    for (u32 j = 0; j < k && !dupe; j++) dupe = amh_memcmp(chunk + D.heads_off + 32 * j, 32, h, 32) == 0;
    if (!dupe) distinct_stored++;
  }
  if (ok && distinct_stored != derived) ok = false;
  B.heads_ok[d] = ok;
  if (!ok && B.check_heads) amh_fail(B, d, AMH_E_MISMATCHING_HEADS);
}

// ------------------------------------------------------------------------------------------
// K18  visibility and per-key winner — types.rs:712-720, 792-798; iter/top_ops.rs:58-65.
//      Thread per row.
// ------------------------------------------------------------------------------------------
AMH_DEV bool row_visible(const BatchDev& B, u32 r) {
  u8 a = B.r_action[r];
  if (a == 5 || a == 7) return false;
  if (a == 1 && (B.r_vmeta[r] & 15) == 8) return B.r_succ_cnt[r] <= B.r_inc_cnt[r];
  return B.r_succ_cnt[r] == 0;
}
AMH_DEV bool same_group(const BatchDev& B, u32 a, u32 b) {  // same object and same elemid_or_key
  if (B.r_doc[a] != B.r_doc[b]) return false;
  if (B.r_obj_ctr[a] != B.r_obj_ctr[b] || B.r_obj_actor[a] != B.r_obj_actor[b]) return false;
  bool ap = B.r_key_len[a] != AMH_NULL32 && !B.r_insert[a], bp = B.r_key_len[b] != AMH_NULL32 && !B.r_insert[b];
  if (ap != bp) return false;
  if (ap) {
    if (B.r_key_len[a] != B.r_key_len[b]) return false;
    const u8* chunk = B.arena + B.desc[B.r_doc[a]].chunk_off;
    RleVal x, y;   // eight bytes per step (rv_eq), not one
    x.v = 0; x.s = chunk + B.r_key_off[a]; x.n = B.r_key_len[a];
    y.v = 0; y.s = chunk + B.r_key_off[b]; y.n = x.n;
    return rv_eq<2>(x, y);
  }
  u32 aa = B.r_insert[a] ? B.r_id_actor[a] : B.r_key_actor[a], ac = B.r_insert[a] ? B.r_id_ctr[a] : B.r_key_ctr[a];
  u32 ba = B.r_insert[b] ? B.r_id_actor[b] : B.r_key_actor[b], bc = B.r_insert[b] ? B.r_id_ctr[b] : B.r_key_ctr[b];
  return aa == ba && ac == bc;
}
AMH_DEV void k_visible(const BatchDev& B, u32 r) {
  u32 d = B.r_doc[r];
  B.r_visible[r] = 0;
  B.r_winner[r] = 0;
  if (B.status[d]) return;
  bool vis = row_visible(B, r);
  B.r_visible[r] = vis;
  if (!vis) return;
  u32 end = B.row_base[d] + B.cnt[d].n_rows;
  bool win = true;
  for (u32 j = r + 1; j < end && same_group(B, r, j); j++)
    if (row_visible(B, j)) { win = false; break; }
  B.r_winner[r] = win;
}

// ------------------------------------------------------------------------------------------
// K18b  text materialization — ReadDoc::text = OpSet::text (op_set.rs:307-311): the winners of a text object's elements
//       in document order, each as OpType::to_str (types.rs:278-286): a string value as it is, any other visible value
//       (numbers, nested objects, ...) as U+FFFC; marks are never visible.  Thread per row: bytes the row contributes
//       (k_text_len; scanned), then the bytes (k_text_fill); k_text_doc (thread per document) notes where every document's
//       text starts.  A document's text = the texts of its text objects in document order, back to back.
// ------------------------------------------------------------------------------------------
AMH_DEV u32 text_row_len(const BatchDev& B, u32 r, bool& is_str) {
  is_str = false;
  if (!B.r_winner[r]) return 0;
  const u32 d = B.r_doc[r];
  const u32 oc = B.r_obj_ctr[r];
  if (oc == 0) return 0;   // the root is a map
  const u32 mk = ht_find(B, d, pack_id(oc, B.r_obj_actor[r]));
  if (mk == AMH_EMPTY32 || B.r_action[mk] != 4) return 0;   // not a text object
  if (B.r_action[r] == 1 && (B.r_vmeta[r] & 15) == 6) { is_str = true; return (u32)(B.r_vmeta[r] >> 4); }
  return 3;   // U+FFFC
}
AMH_DEV void k_text_len(const BatchDev& B, u32 r) {
  bool is_str;
  B.tx_off[r] = B.status[B.r_doc[r]] ? 0 : text_row_len(B, r, is_str);
}
AMH_DEV void k_text_fill(const BatchDev& B, u32 r) {
  const u32 n = B.tx_off[r + 1] - B.tx_off[r];
  if (n == 0) return;
  // a row with bytes is a winner inside a text object (k_text_len looked the object up): only its kind is left to tell
  const bool is_str = B.r_action[r] == 1 && (B.r_vmeta[r] & 15) == 6;
  u8* o = B.tx_out + B.tx_off[r];
  if (is_str) {
    const DocDesc& D = B.desc[B.r_doc[r]];
    const u8* p = B.arena + D.chunk_off + D.ocol[OC_VAL_RAW].off + B.r_vraw_off[r];
    for (u32 i = 0; i < n; i++) o[i] = p[i];
  } else { o[0] = 0xEF; o[1] = 0xBF; o[2] = 0xBC; }
}
AMH_DEV void k_text_doc(const BatchDev& B, u32 d) {   // d in [0, n_docs]
  B.tx_doc[d] = d < B.n_docs ? B.tx_off[B.row_base[d]] : B.tx_off[B.tot_rows];
}

// ------------------------------------------------------------------------------------------
// K19  sync summaries: sorted heads (automerge.rs:1300-1304) and the Bloom filter over every change
//      hash of the document (sync.rs:243-250 with last_sync = [], sync/bloom.rs:9-10, 38-47, 69-102,
//      125-146).  Record: u32 doc, u32 status, u32 n_heads, heads[32 x n], u32 bloom_len, bloom bytes.
//      Thread per document: k_sync_size -> record sizes (scanned into sync_off), k_sync_fill.
// ------------------------------------------------------------------------------------------
AMH_DEV u32 bloom_wire_len(u32 n) {  // uleb(n) uleb(10) uleb(7) + ceil(10 n / 8) bytes; nothing for an empty filter
  return n ? uleb_size(n) + 2 + (u32)(((u64)n * 10 + 7) / 8) : 0;
}
AMH_DEV void k_sync_size(const BatchDev& B, u32 d) {
  u32 nh = 0, nc = 0;
  if (B.status[d] == 0) {
    nc = B.cnt[d].n_chg;
    u32 cb = B.chg_base[d];
    for (u32 i = 0; i < nc; i++) if (!B.c_is_dep[cb + i]) nh++;
  }
  B.sync_off[d] = 12 + 32 * nh + 4 + bloom_wire_len(nc);
}
AMH_DEV void sync_put32(u8* o, u32 v) { o[0] = (u8)v; o[1] = (u8)(v >> 8); o[2] = (u8)(v >> 16); o[3] = (u8)(v >> 24); }
AMH_DEV void k_sync_fill(const BatchDev& B, u32 d) {
  // caller-provided buffer too small, or the record offsets saturated (amh_sync_summaries_noalloc): nothing is written
  if (B.sync_cap && (B.sync_off[d + 1] > B.sync_cap || B.sync_off[d + 1] == 0xFFFFFFFFu)) return;
  u8* o = B.sync_out + B.sync_off[d];
  u32 rec_len = B.sync_off[d + 1] - B.sync_off[d];
  sync_put32(o, d);
  sync_put32(o + 4, (u32)B.status[d]);
  u32 nh = 0, nc = 0, cb = 0;
  if (B.status[d] == 0) { nc = B.cnt[d].n_chg; cb = B.chg_base[d]; }
  u8* hs = o + 12;
  for (u32 i = 0; i < nc; i++) {   // heads in ascending hash order (BTreeSet<ChangeHash>)
    if (B.c_is_dep[cb + i]) continue;
    const u8* h = B.c_hash + (u64)(cb + i) * 32;
    u32 j = nh;
    while (j > 0 && amh_memcmp(hs + 32 * (j - 1), 32, h, 32) > 0) {
      for (int b = 0; b < 32; b++) hs[32 * j + b] = hs[32 * (j - 1) + b];
      j--;
    }
    for (int b = 0; b < 32; b++) hs[32 * j + b] = h[b];
    nh++;
  }
  sync_put32(o + 8, nh);
  u8* bl = hs + 32 * nh;
  u32 blen = bloom_wire_len(nc);
  sync_put32(bl, blen);
  bl += 4;
  if (blen && 12 + 32 * nh + 4 + blen == rec_len) {
    u32 pos = wr_uleb(bl, nc);
    bl[pos++] = 10;
    bl[pos++] = 7;
    u8* bits = bl + pos;
    u32 nbytes = blen - pos, modulo = 8 * nbytes;
    for (u32 i = 0; i < nbytes; i++) bits[i] = 0;
    for (u32 i = 0; i < nc; i++) {
      const u8* h = B.c_hash + (u64)(cb + i) * 32;
      u32 x = ((u32)h[0] | ((u32)h[1] << 8) | ((u32)h[2] << 16) | ((u32)h[3] << 24)) % modulo;
      u32 y = ((u32)h[4] | ((u32)h[5] << 8) | ((u32)h[6] << 16) | ((u32)h[7] << 24)) % modulo;
      u32 z = ((u32)h[8] | ((u32)h[9] << 8) | ((u32)h[10] << 16) | ((u32)h[11] << 24)) % modulo;
      bits[x >> 3] |= (u8)(1u << (x & 7));
      for (u32 k = 1; k < 7; k++) {
        x = (u32)(((u64)x + y) % modulo);
        y = (u32)(((u64)y + z) % modulo);
        bits[x >> 3] |= (u8)(1u << (x & 7));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K20  save: the document chunk of every loaded document — automerge.rs:943-965,
//      storage/save/document.rs:21-148, storage/document.rs:214-263, doc_op_columns.rs:104-311,
//      doc_change_columns.rs:87-162, convert/op_as_docop.rs:23-162, indexed_cache.rs:72-107.
//   k_save_size   (thread per (document, column))  column sizes -> sv_col_len
//   k_save_layout (thread per document)            chunk size -> sv_off (scanned)
//   k_save_store  (thread per (document, column))  column bytes
//   k_save_finish (thread per document)            actors, heads, column metadata, head indices, header + SHA-256
//   Columns 0..8 = change metadata, 9..24 = ops.  Actor numbers are ranks in the bytewise-sorted table.
// ------------------------------------------------------------------------------------------
#define AMH_SV_CHG_COLS 9
#define AMH_SV_NCOLS 25
AMH_DEV u32 sv_spec(u32 col) {
  const u32 specs[AMH_SV_NCOLS] = {0x01, 0x03, 0x13, 0x23, 0x35, 0x40, 0x43, 0x56, 0x57,
                                   0x01, 0x02, 0x11, 0x13, 0x15, 0x21, 0x23, 0x34, 0x42, 0x56, 0x57, 0x80, 0x81, 0x83, 0x94, 0xA5};
  return specs[col];
}
// k-th dep of change c in the order of the deps' hashes (save/document.rs:141-146: deps() is sorted by hash)
AMH_DEV u32 sv_dep_sorted(const BatchDev& B, u32 c, u32 cb, u32 k) {
  u32 nd = B.c_ndeps[c];
  const u32* di = B.dep_idx + B.c_dep_base[c];
  for (u32 a = 0; a < nd; a++) {   // the dep with exactly k deps ordered before it
    const u8* ha = B.c_hash + (u64)(cb + di[a]) * 32;
    u32 before = 0;
    for (u32 b = 0; b < nd; b++) {
      if (b == a) continue;
      int cmp = amh_memcmp(B.c_hash + (u64)(cb + di[b]) * 32, 32, ha, 32);
      if (cmp < 0 || (cmp == 0 && b < a)) before++;
    }
    if (before == k) return di[a];
  }
  return 0;
}
// change-metadata columns 0..8 of a document (doc_change_columns.rs:87-162), any writer
template <class W>
AMH_DEV void save_encode_chg_col(const BatchDev& B, u32 d, u32 col, W& w) {
  const DocDesc& D = B.desc[d];
  const u8* chunk = B.arena + D.chunk_off;
  const u32* rank = B.actor_rank + B.actor_base[d];
  RleVal x; x.v = 0; x.s = 0; x.n = 0;
  u32 cb = B.chg_base[d], n = B.cnt[d].n_chg;
  switch (col) {
    case 0: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = rank[B.c_actor[cb + i]]; re_value(w, e, x); } re_finish(w, e); break; }
    case 1: { DeltaEnc e; re_init(e.r); e.abs = 0; for (u32 i = 0; i < n; i++) de_value(w, e, (i64)B.c_seq[cb + i]); re_finish(w, e.r); break; }
    case 2: { DeltaEnc e; re_init(e.r); e.abs = 0; for (u32 i = 0; i < n; i++) de_value(w, e, (i64)B.c_max_op[cb + i]); re_finish(w, e.r); break; }
    case 3: { DeltaEnc e; re_init(e.r); e.abs = 0; for (u32 i = 0; i < n; i++) de_value(w, e, B.c_time[cb + i]); re_finish(w, e.r); break; }
    case 4: {
      RleEnc<2> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 ml = B.c_msg_len[cb + i];
        if (ml == AMH_NULL32 || ml == 0) re_null(w, e);
        else { RleVal sv; sv.v = 0; sv.s = chunk + B.c_msg_off[cb + i]; sv.n = ml; re_value(w, e, sv); }
      }
      re_finish(w, e);
      break;
    }
    case 5: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = B.c_ndeps[cb + i]; re_value(w, e, x); } re_finish(w, e); break; }
    case 6: {
      DeltaEnc e; re_init(e.r); e.abs = 0;
      for (u32 i = 0; i < n; i++) { u32 nd = B.c_ndeps[cb + i]; for (u32 k = 0; k < nd; k++) de_value(w, e, (i64)sv_dep_sorted(B, cb + i, cb, k)); }
      re_finish(w, e.r);
      break;
    }
    case 7: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = ((u64)B.c_extra_len[cb + i] << 4) | 7; re_value(w, e, x); } re_finish(w, e); break; }
    case 8: { for (u32 i = 0; i < n; i++) cw_bytes(w, chunk + B.c_extra_off[cb + i], B.c_extra_len[cb + i]); break; }
  }
}
AMH_DEV u32 save_encode_col(const BatchDev& B, u32 d, u32 col, u8* out) {
  const DocDesc& D = B.desc[d];
  const u8* chunk = B.arena + D.chunk_off;
  const u32* rank = B.actor_rank + B.actor_base[d];
  ColWriter w; w.out = out; w.pos = 0;
  RleVal x; x.v = 0; x.s = 0; x.n = 0;
  if (col < AMH_SV_CHG_COLS) {
    save_encode_chg_col(B, d, col, w);
    return w.pos;
  }
  u32 rb = B.row_base[d], n = B.cnt[d].n_rows;
  switch (col) {
    case 9: case 10: {
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 r = rb + i;
        if (B.r_obj_ctr[r] == 0) re_null(w, e);
        else { x.v = col == 9 ? rank[B.r_obj_actor[r]] : B.r_obj_ctr[r]; re_value(w, e, x); }
      }
      re_finish(w, e);
      break;
    }
    case 11: {
      RleEnc<0> e; re_init(e);
      for (u32 i = 0; i < n; i++) { OpKey k = op_key(B, rb + i); if (k.kind == 2) { x.v = rank[k.actor]; re_value(w, e, x); } else re_null(w, e); }
      re_finish(w, e);
      break;
    }
    case 12: {
      DeltaEnc e; re_init(e.r); e.abs = 0;
      for (u32 i = 0; i < n; i++) { OpKey k = op_key(B, rb + i); if (k.kind == 0) re_null(w, e.r); else de_value(w, e, k.kind == 1 ? 0 : (i64)k.ctr); }
      re_finish(w, e.r);
      break;
    }
    case 13: {
      RleEnc<2> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        OpKey k = op_key(B, rb + i);
        if (k.kind == 0) { RleVal sv; sv.v = 0; sv.s = chunk + k.str_off; sv.n = k.str_len; re_value(w, e, sv); } else re_null(w, e);
      }
      re_finish(w, e);
      break;
    }
    case 14: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = rank[B.r_id_actor[rb + i]]; re_value(w, e, x); } re_finish(w, e); break; }
    case 15: { DeltaEnc e; re_init(e.r); e.abs = 0; for (u32 i = 0; i < n; i++) de_value(w, e, (i64)B.r_id_ctr[rb + i]); re_finish(w, e.r); break; }
    case 16: {
      bool last = false; u64 cnt = 0;
      for (u32 i = 0; i < n; i++) { bool v = B.r_insert[rb + i] != 0; if (v == last) cnt++; else { cw_uleb(w, cnt); last = v; cnt = 1; } }
      if (cnt > 0) cw_uleb(w, cnt);
      break;
    }
    case 17: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = B.r_action[rb + i]; re_value(w, e, x); } re_finish(w, e); break; }
    case 18: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { OpVal v = op_val(B, rb + i); x.v = v.meta; re_value(w, e, x); } re_finish(w, e); break; }
    case 19: {
      for (u32 i = 0; i < n; i++) {
        OpVal v = op_val(B, rb + i);
        if (v.kind == 1) cw_bytes(w, v.raw, v.raw_len); else if (v.kind == 2) cw_sleb(w, v.i);
      }
      break;
    }
    case 20: { RleEnc<0> e; re_init(e); for (u32 i = 0; i < n; i++) { x.v = B.r_succ_cnt[rb + i]; re_value(w, e, x); } re_finish(w, e); break; }
    case 21: {
      RleEnc<0> e; re_init(e);
      u32 sb = B.succ_base[d], ns = B.cnt[d].n_succ;
      for (u32 i = 0; i < ns; i++) { x.v = rank[B.s_actor[sb + i]]; re_value(w, e, x); }
      re_finish(w, e);
      break;
    }
    case 22: {
      DeltaEnc e; re_init(e.r); e.abs = 0;
      u32 sb = B.succ_base[d], ns = B.cnt[d].n_succ;
      for (u32 i = 0; i < ns; i++) de_value(w, e, (i64)B.s_ctr[sb + i]);
      re_finish(w, e.r);
      break;
    }
    case 23: {   // MaybeBoolean (boolean.rs:131-137)
      bool any = false;
      for (u32 i = 0; i < n && !any; i++) any = B.r_action[rb + i] == 7 && B.r_expand[rb + i];
      if (!any) break;
      bool last = false; u64 cnt = 0;
      for (u32 i = 0; i < n; i++) { bool v = B.r_action[rb + i] == 7 && B.r_expand[rb + i]; if (v == last) cnt++; else { cw_uleb(w, cnt); last = v; cnt = 1; } }
      if (cnt > 0) cw_uleb(w, cnt);
      break;
    }
    case 24: {
      RleEnc<2> e; re_init(e);
      for (u32 i = 0; i < n; i++) {
        u32 r = rb + i;
        if (B.r_action[r] == 7 && B.r_mark_len[r] != AMH_NULL32) { RleVal sv; sv.v = 0; sv.s = chunk + B.r_mark_off[r]; sv.n = B.r_mark_len[r]; re_value(w, e, sv); }
        else re_null(w, e);
      }
      re_finish(w, e);
      break;
    }
  }
  return w.pos;
}
AMH_DEV bool sv_doc_ok(const BatchDev& B, u32 d) { return B.status[d] == 0 && B.sv_status[d] == 0; }
AMH_DEV void k_save_check(const BatchDev& B, u32 d) {
  B.sv_status[d] = 0;
  if (B.status[d]) { B.sv_status[d] = B.status[d]; return; }
  u32 ab = B.actor_base[d], na = B.desc[d].n_actors;
  // object runs are kept in file order, which is lamport order only if the stored actor table is sorted
  for (u32 a = 0; a < na; a++) if (B.actor_rank[ab + a] != a) { B.sv_status[d] = AMH_E_UNSUPPORTED; return; }
}
AMH_DEV void k_save_size(const BatchDev& B, u32 gid) {
  u32 col = gid / B.n_docs, d = gid % B.n_docs;
  if (col >= AMH_SV_NCOLS) return;
  B.sv_col_len[(u64)d * AMH_SV_NCOLS + col] = sv_doc_ok(B, d) ? save_encode_col(B, d, col, (u8*)0) : 0;
}
AMH_DEV u32 sv_heads(const BatchDev& B, u32 d) {
  u32 nh = 0, cb = B.chg_base[d], nc = B.cnt[d].n_chg;
  for (u32 i = 0; i < nc; i++) if (!B.c_is_dep[cb + i]) nh++;
  return nh;
}
AMH_DEV void k_save_layout(const BatchDev& B, u32 d) {
  B.sv_off[d] = 0;
  B.sv_body_len[d] = 0;
  if (!sv_doc_ok(B, d)) return;
  u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
  if (cl[9] == 0) cl[10] = 0;     // ObjIdRange::encode -> None (obj_id.rs:79-81)
  if (cl[21] == 0) cl[22] = 0;    // succ actor / ctr only when there are entries (doc_op_columns.rs:286-297)
  const DocDesc& D = B.desc[d];
  u32 ab = B.actor_base[d], na = D.n_actors;
  u64 len = uleb_size(na);
  for (u32 a = 0; a < na; a++) len += uleb_size(B.actor_len[ab + a]) + B.actor_len[ab + a];
  u32 nh = sv_heads(B, d), cb = B.chg_base[d], nc = B.cnt[d].n_chg;
  len += uleb_size(nh) + 32ull * nh;
  for (u32 i = 0; i < nc; i++) if (!B.c_is_dep[cb + i]) len += uleb_size(i);
  u32 ncc = 0, noc = 0;
  for (u32 k = 0; k < AMH_SV_NCOLS; k++) {
    if (!cl[k]) continue;
    if (k < AMH_SV_CHG_COLS) ncc++; else noc++;
    len += uleb_size(sv_spec(k)) + uleb_size(cl[k]) + cl[k];
  }
  len += uleb_size(ncc) + uleb_size(noc);
  if (len > 0x7FFFFFF0ull) { B.sv_status[d] = AMH_E_TOO_LARGE; return; }
  B.sv_body_len[d] = (u32)len;
  B.sv_off[d] = (4 + 4 + 1 + uleb_size(len) + (u32)len + 15u) & ~15u;   // chunks start 16-byte aligned in sv_out
}
AMH_DEV u32 sv_chunk_len(const BatchDev& B, u32 d) { return 9 + uleb_size(B.sv_body_len[d]) + B.sv_body_len[d]; }
// where the column data starts inside the chunk (after header, actors, heads and both metadata blocks)
AMH_DEV u32 sv_data_start(const BatchDev& B, u32 d) {
  const u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
  u32 tot = 0;
  for (u32 k = 0; k < AMH_SV_NCOLS; k++) tot += cl[k];
  u32 nh = sv_heads(B, d), cb = B.chg_base[d], nc = B.cnt[d].n_chg, suffix = 0;
  for (u32 i = 0; i < nc; i++) if (!B.c_is_dep[cb + i]) suffix += uleb_size(i);
  (void)nh;
  return sv_chunk_len(B, d) - suffix - tot;
}
AMH_DEV void k_save_store(const BatchDev& B, u32 gid) {
  u32 col = gid / B.n_docs, d = gid % B.n_docs;
  if (col >= AMH_SV_NCOLS || !sv_doc_ok(B, d)) return;
  const u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
  if (cl[col] == 0) return;
#ifndef AMH_EMU
  if (!((B.sv_wide[d] >> col) & 1)) return;   // written by g_save_lds
#endif
  u64 off = (u64)B.sv_off[d] + sv_data_start(B, d);
  for (u32 k = 0; k < col; k++) off += cl[k];
  save_encode_col(B, d, col, B.sv_out + off);
}
AMH_DEV void k_save_finish(const BatchDev& B, u32 d) {
  if (!sv_doc_ok(B, d)) return;
  const DocDesc& D = B.desc[d];
  const u8* chunk = B.arena + D.chunk_off;
  const u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
  u8* out = B.sv_out + B.sv_off[d];
  u32 body_len = B.sv_body_len[d];
  u32 hdr = 9 + uleb_size(body_len), pos = hdr;
  u32 ab = B.actor_base[d], na = D.n_actors, cb = B.chg_base[d], nc = B.cnt[d].n_chg;
  pos += wr_uleb(out + pos, na);
  for (u32 k = 0; k < na; k++) {   // bytewise order (document.rs:233)
    u32 a = ab + B.actor_by_rank[ab + k];
    pos += wr_uleb(out + pos, B.actor_len[a]);
    for (u32 b = 0; b < B.actor_len[a]; b++) out[pos++] = chunk[B.actor_off[a] + b];
  }
  u32 nh = sv_heads(B, d);
  pos += wr_uleb(out + pos, nh);
  u8* heads = out + pos;
  // heads sorted by hash; their change indices follow the column data in the same order
  u32 placed = 0;
  for (u32 i = 0; i < nc; i++) {
    if (B.c_is_dep[cb + i]) continue;
    const u8* h = B.c_hash + (u64)(cb + i) * 32;
    u32 j = placed;
    while (j > 0 && amh_memcmp(heads + 32 * (j - 1), 32, h, 32) > 0) {
      for (int b = 0; b < 32; b++) heads[32 * j + b] = heads[32 * (j - 1) + b];
      j--;
    }
    for (int b = 0; b < 32; b++) heads[32 * j + b] = h[b];
    placed++;
  }
  pos += 32 * nh;
  for (int blk = 0; blk < 2; blk++) {
    u32 lo = blk ? AMH_SV_CHG_COLS : 0, hi = blk ? AMH_SV_NCOLS : AMH_SV_CHG_COLS, cnt = 0;
    for (u32 k = lo; k < hi; k++) if (cl[k]) cnt++;
    pos += wr_uleb(out + pos, cnt);
    for (u32 k = lo; k < hi; k++) if (cl[k]) { pos += wr_uleb(out + pos, sv_spec(k)); pos += wr_uleb(out + pos, cl[k]); }
  }
  for (u32 k = 0; k < AMH_SV_NCOLS; k++) pos += cl[k];   // column data: written by k_save_store
  // suffix: index of each head's change, in head order
  for (u32 k = 0; k < nh; k++) {
    for (u32 i = 0; i < nc; i++) {
      if (B.c_is_dep[cb + i]) continue;
      if (amh_memcmp(B.c_hash + (u64)(cb + i) * 32, 32, heads + 32 * k, 32) == 0) { pos += wr_uleb(out + pos, i); break; }
    }
  }
  out[0] = 0x85; out[1] = 0x6f; out[2] = 0x4a; out[3] = 0x83;
  out[8] = 0;
  wr_uleb(out + 9, body_len);
  if (B.skip_doc_sha) { out[4] = out[5] = out[6] = out[7] = 0; return; }   // read back by the engine only (merge / apply reload)
  if (B.sha_host_min && hdr + body_len >= B.sha_host_min) return;   // large chunk: the host, which receives the bytes anyway, hashes it (amh_save_batch)
  u8 h[32];
  sha256_contig(out + 8, (u64)(hdr - 8) + body_len, h);
  out[4] = h[0]; out[5] = h[1]; out[6] = h[2]; out[7] = h[3];
}

// ------------------------------------------------------------------------------------------
// small glue kernels
// ------------------------------------------------------------------------------------------
// id table of a document: direct when actors x (highest counter + 1) stays within 8 slots per row, else a hash
// region of at least twice the rows.  doc_max_op is final here (op ids and succ ids, k_decode_ops).  Thread per document.
AMH_DEV void k_ht_size(const BatchDev& B, u32 d) {
  B.ht_base[d] = 0; B.dn_base[d] = 0; B.dn_stride[d] = 0;
  if (B.status[d]) return;
  u32 rows = B.cnt[d].n_rows, na = B.desc[d].n_actors;
  if (rows == 0) return;
  u64 stride = (u64)B.doc_max_op[d] + 1;
  u64 dense = stride * na;
  if (B.ht_mode == 0 && dense <= 8ull * rows + 64 && stride < 0xFFFFFFFFull) { B.dn_base[d] = (u32)dense; B.dn_stride[d] = (u32)stride; }
  else B.ht_base[d] = amh_pow2_at_least(2 * rows);
}

// per-document sizes, written into the *_base arrays and scanned in place afterwards
AMH_DEV void k_extract(const BatchDev& B, u32 d) {
  if (B.sha_bad && B.sha_bad[d]) {   // the checksum is checked before the document is reconstructed: it outranks what the count made of the bytes
    B.status[d] = AMH_E_BAD_CHECKSUM;
    B.fail_line[d] = __LINE__;
  }
  bool ok = B.status[d] == 0;
  const DocCounts& C = B.cnt[d];
  u32 rows = ok ? C.n_rows : 0, succ = ok ? C.n_succ : 0, chg = ok ? C.n_chg : 0, deps = ok ? C.n_deps : 0;
  u32 na = ok ? B.desc[d].n_actors : 0;
  B.row_base[d] = rows;
  B.succ_base[d] = succ;
  B.chg_base[d] = chg;
  B.dep_base[d] = deps;
  B.actor_base[d] = na;
  B.ht_base[d] = 0;   // the id tables are sized after the decode, when the highest counter is known (k_ht_size)
  B.dt_base[d] = 0;   // the delete table is sized after k_succ, for the succ entries that point at deletes (k_dt_size)
  u64 m = (u64)chg * na;
  B.map_base[d] = m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)m;
  B.doc_max_op[d] = 0;
  B.heads_ok[d] = 0;
  B.doc_level[d] = 0;
}
AMH_DEV void k_chg_init(const BatchDev& B, u32 c) {
  u32 d = find_doc(B.chg_base, B.n_docs, c);
  B.c_doc[c] = d;
  B.c_nops[c] = 0;
  B.c_next_same_actor[c] = AMH_EMPTY32;
  B.c_map_base[c] = B.map_base[d] + (c - B.chg_base[d]) * B.desc[d].n_actors;
  B.c_ndeps[c] = 0;
  B.c_extra_len[c] = 0;
  B.c_msg_len[c] = AMH_NULL32;
  B.c_level[c] = 0;
  B.c_is_dep[c] = 0;
}
