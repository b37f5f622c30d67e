// Per-thread bodies of the sync message kernels (see SyncDev in amh_types.h).  Included after amh_kernels.inl.
// Sender side of the sync protocol for a batch of (document, peer) records: Automerge::generate_sync_message
// (sync.rs:134-221) with get_missing_deps (automerge.rs:1776-1799), make_bloom_filter (sync.rs:243-250; get_changes =
// get_changes_clock, automerge.rs:1022-1047, over ChangeGraph::clock_for_heads, change_graph.rs:128-142),
// get_changes_to_send (sync.rs:252-321), BloomFilter::contains_hash (sync/bloom.rs:69-123) and Message::encode
// (sync.rs:526-563).
//
// A document's history is a topological order (a dep sits in front of the change naming it: k_decode_chg checks it on
// load, apply and merge append in causal order), so every graph walk of the reference is one linear pass here:
//   ancestors of a set of heads      backward pass, a marked change marks its deps
//   clock of those heads             per actor the highest seq among the marked changes
//   "dependents of a change to send" forward pass, a change is sent when it is Bloom-negative or one of its deps is sent
//                                    (sync.rs:276-303: the dependents map only holds changes since last_sync)
//   k_sm_lookup  thread per pool hash    where the hash sits in its document's history
//   k_sm_plan    thread per record       need, have, changes to send, None / reset decision, message size
//   k_sm_fill    thread per record       type, heads, need, the Have's last_sync and Bloom header, change count
//   k_sm_bloom   thread per (record, change)  the Bloom filter's bits (atomic OR into the zeroed message buffer)
//   k_sm_jobs    thread per change sent  its length prefix, its copy job (the chunk stays in HBM), its hash for sent_hashes
#pragma once
#include "amh_kernels.inl"

AMH_DEV bool sm_hash_eq(const u8* a, const u8* b) {
  for (int i = 0; i < 32; i++) if (a[i] != b[i]) return false;
  return true;
}
AMH_DEV void k_sm_lookup(const SyncDev& G, u32 i) {
  const BatchDev& B = G.B;
  const u32 r = G.pool_rec[i], d = G.r_doc[r];
  u32 found = AMH_NULL32;
  if (B.status[d] == 0) {
    const u8* h = G.pool + (u64)i * 32;
    const u32 cb = B.chg_base[d], nc = B.cnt[d].n_chg;
    for (u32 c = 0; c < nc; c++) {
      const u8* q = B.c_hash + (u64)(cb + c) * 32;
      if (q[0] == h[0] && q[1] == h[1] && sm_hash_eq(q, h)) { found = c; break; }
    }
  }
  G.pool_idx[i] = found;
}

struct SmRec {
  u32 r, d, cb, nc, na, flags;
  const u32* L;      // the record's row of r_list
  u8* cf;            // [nc] SMC_* bits
  u64* clock;        // [na]
  u32 *heads, *need, *send, *spos;   // spos[k]: where the k-th change sent starts (its length prefix) inside the message
};
AMH_DEV SmRec sm_rec(const SyncDev& G, u32 r) {
  const BatchDev& B = G.B;
  SmRec R;
  R.r = r; R.d = G.r_doc[r]; R.flags = G.r_flags[r];
  R.cb = B.chg_base[R.d];
  const bool ok = B.status[R.d] == 0;
  R.nc = ok ? B.cnt[R.d].n_chg : 0;
  R.na = ok ? B.desc[R.d].n_actors : 0;
  R.L = G.r_list + (u64)r * SM_LISTS;
  R.cf = G.c_flag + G.r_flag_base[r];
  R.clock = G.clock + G.r_clock_base[r];
  R.heads = G.words + G.r_word_base[r];
  R.need = R.heads + R.nc;
  R.send = R.need + (R.L[SM_QDEPS + 1] - R.L[SM_QDEPS]) + (R.L[SM_THEIR_HEADS + 1] - R.L[SM_THEIR_HEADS]);
  R.spos = R.send + (R.L[SM_THEIR_NEED + 1] - R.L[SM_THEIR_NEED]) + R.nc;
  return R;
}
AMH_DEV const u8* sm_chg_hash(const SyncDev& G, const SmRec& R, u32 c) { return G.B.c_hash + (u64)(R.cb + c) * 32; }
AMH_DEV const u8* sm_pool(const SyncDev& G, u32 i) { return G.pool + (u64)i * 32; }

// clock_for_heads over the known hashes of pool range [lo, hi): marks the ancestors with `bit`, leaves the clock in R.clock
AMH_DEV void sm_clock(const SyncDev& G, const SmRec& R, u32 lo, u32 hi, u8 bit) {
  const BatchDev& B = G.B;
  for (u32 a = 0; a < R.na; a++) R.clock[a] = 0;
  for (u32 i = lo; i < hi; i++) { u32 c = G.pool_idx[i]; if (c != AMH_NULL32) R.cf[c] |= bit; }
  for (u32 c = R.nc; c-- > 0;) {
    if (!(R.cf[c] & bit)) continue;
    const u32 gc = R.cb + c, a = B.c_actor[gc];
    if (B.c_seq[gc] > R.clock[a]) R.clock[a] = B.c_seq[gc];
    for (u32 k = 0; k < B.c_ndeps[gc]; k++) R.cf[B.dep_idx[B.c_dep_base[gc] + k]] |= bit;
  }
}
AMH_DEV bool sm_since(const SyncDev& G, const SmRec& R, u32 c) {   // get_changes_clock: seq above the actor's clock entry
  return G.B.c_seq[R.cb + c] > R.clock[G.B.c_actor[R.cb + c]];
}
AMH_DEV bool sm_bloom_has(const SyncDev& G, u32 hv, const u8* h) {   // bloom.rs:69-90, 110-123
  if (G.h_entries[hv] == 0) return false;
  const u8* bits = G.blooms + G.h_bits[hv];
  const u32 modulo = 8 * G.h_nbytes[hv];   // the host rejects filters with entries and no bits (the reference divides by zero)
  u32 x = ((u32)h[0] | ((u32)h[1] << 8) | ((u32)h[2] << 16) | ((u32)h[3] << 24)) % modulo;
  u32 y = ((u32)h[4] | ((u32)h[5] << 8) | ((u32)h[6] << 16) | ((u32)h[7] << 24)) % modulo;
  const u32 z = ((u32)h[8] | ((u32)h[9] << 8) | ((u32)h[10] << 16) | ((u32)h[11] << 24)) % modulo;
  const u32 probes = G.h_probes[hv] ? G.h_probes[hv] : 1;   // the first probe is pushed in front of the 1..num_probes loop
  for (u32 k = 0; k < probes; k++) {
    if (k) { x = (u32)(((u64)x + y) % modulo); y = (u32)(((u64)y + z) % modulo); }
    if (!(bits[x >> 3] & (1u << (x & 7)))) return false;
  }
  return true;
}
AMH_DEV u32 sm_hashes_len(u32 n) { return uleb_size(n) + 32 * n; }

AMH_DEV void k_sm_plan(const SyncDev& G, u32 r) {
  const BatchDev& B = G.B;
  const SmRec R = sm_rec(G, r);
  G.r_nheads[r] = 0; G.r_nneed[r] = 0; G.r_nsend[r] = 0; G.r_bloom_n[r] = AMH_NULL32;
  G.msg_off[r] = 0; G.job_base[r] = 0;
  if (B.status[R.d] != 0) { G.r_kind[r] = SMK_FAILED; return; }
  for (u32 c = 0; c < R.nc; c++) R.cf[c] = 0;
  if (G.raw) {   // save_after (automerge.rs:993-1000) = get_changes(heads): the changes above the heads' clock, history order
    sm_clock(G, R, R.L[SM_LAST_SYNC], R.L[SM_LAST_SYNC + 1], SMC_ANC_SYNC);
    u32 ns = 0;
    u64 len = 0;
    for (u32 c = 0; c < R.nc; c++) if (sm_since(G, R, c)) { R.spos[ns] = (u32)len; R.send[ns++] = c; len += B.c_out_len[R.cb + c]; }
    G.r_kind[r] = SMK_MESSAGE;
    G.r_nsend[r] = ns;
    G.job_base[r] = ns;
    G.msg_off[r] = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len;
    return;
  }
  // our_heads: ascending by hash (get_heads, automerge.rs:1300-1304)
  u32 nh = 0;
  for (u32 c = 0; c < R.nc; c++) {
    if (B.c_is_dep[R.cb + c]) continue;
    u32 j = nh;
    while (j > 0 && amh_memcmp(sm_chg_hash(G, R, R.heads[j - 1]), 32, sm_chg_hash(G, R, c), 32) > 0) { R.heads[j] = R.heads[j - 1]; j--; }
    R.heads[j] = c;
    nh++;
  }
  G.r_nheads[r] = nh;
  // our_need = get_missing_deps(their_heads): deps of queued changes (the host left out those that are queued themselves)
  // and their heads, as far as the history lacks them and they are not queued; sorted, distinct
  u32 nn = 0;
  for (int pass = 0; pass < 2; pass++) {
    const u32 l = pass ? SM_THEIR_HEADS : SM_QDEPS;
    for (u32 i = R.L[l]; i < R.L[l + 1]; i++) {
      if (G.pool_idx[i] != AMH_NULL32) continue;
      const u8* h = sm_pool(G, i);
      bool queued = false;
      if (pass) for (u32 q = R.L[SM_QHASH]; q < R.L[SM_QHASH + 1] && !queued; q++) queued = sm_hash_eq(sm_pool(G, q), h);
      if (queued) continue;
      u32 j = nn;
      bool dupe = false;
      while (j > 0) {
        int cmp = amh_memcmp(sm_pool(G, R.need[j - 1]), 32, h, 32);
        if (cmp == 0) dupe = true;
        if (cmp <= 0) break;
        j--;
      }
      if (dupe) continue;
      for (u32 k = nn; k > j; k--) R.need[k] = R.need[k - 1];
      R.need[j] = i;
      nn++;
    }
  }
  G.r_nneed[r] = nn;
  // our_have: a Bloom filter of the changes since shared_heads, unless something is needed that is not one of their heads
  bool all_in = true;
  for (u32 k = 0; k < nn && all_in; k++) {
    bool in = false;
    for (u32 i = R.L[SM_THEIR_HEADS]; i < R.L[SM_THEIR_HEADS + 1] && !in; i++) in = sm_hash_eq(sm_pool(G, i), sm_pool(G, R.need[k]));
    all_in = in;
  }
  u32 have_len = 1;   // the Have count
  if (all_in) {
    sm_clock(G, R, R.L[SM_SHARED], R.L[SM_SHARED + 1], SMC_ANC_SHARED);
    u32 nb = 0;
    for (u32 c = 0; c < R.nc; c++) if (sm_since(G, R, c)) { R.cf[c] |= SMC_BLOOM; nb++; }
    G.r_bloom_n[r] = nb;
    const u32 bl = bloom_wire_len(nb);
    have_len += sm_hashes_len(R.L[SM_SHARED + 1] - R.L[SM_SHARED]) + uleb_size(bl) + bl;
  }
  // a peer whose last_sync names changes this document lacks gets a reset message (sync.rs:150-166)
  const u32 hv0 = G.r_have[r], hv1 = G.r_have[r + 1];
  if ((R.flags & SMF_THEIR_HAVE) && hv1 > hv0) {
    bool known = true;
    for (u32 i = G.h_ls[2 * hv0]; i < G.h_ls[2 * hv0 + 1] && known; i++) known = G.pool_idx[i] != AMH_NULL32;
    if (!known) {
      G.r_kind[r] = SMK_RESET;
      G.r_nneed[r] = 0;
      G.r_bloom_n[r] = AMH_NULL32;
      G.msg_off[r] = 1 + sm_hashes_len(nh) + 1 + 3 + 1;   // type, heads, no need, one empty Have, no changes
      return;
    }
  }
  // changes_to_send (sync.rs:168-176, 252-321)
  u32 ns = 0;
  if ((R.flags & SMF_THEIR_HAVE) && (R.flags & SMF_THEIR_NEED)) {
    if (hv1 == hv0) {
      for (u32 i = R.L[SM_THEIR_NEED]; i < R.L[SM_THEIR_NEED + 1]; i++) if (G.pool_idx[i] != AMH_NULL32) R.send[ns++] = G.pool_idx[i];
    } else {
      sm_clock(G, R, R.L[SM_LAST_SYNC], R.L[SM_LAST_SYNC + 1], SMC_ANC_SYNC);
      for (u32 c = 0; c < R.nc; c++) {
        if (!sm_since(G, R, c)) continue;
        const u32 gc = R.cb + c;
        bool send = true;   // in none of their filters
        for (u32 hv = hv0; hv < hv1 && send; hv++) send = !sm_bloom_has(G, hv, sm_chg_hash(G, R, c));
        for (u32 k = 0; k < B.c_ndeps[gc] && !send; k++) send = (R.cf[B.dep_idx[B.c_dep_base[gc] + k]] & SMC_SEND) != 0;
        R.cf[c] |= SMC_SINCE | (send ? SMC_SEND : 0);
      }
      for (u32 i = R.L[SM_THEIR_NEED]; i < R.L[SM_THEIR_NEED + 1]; i++) {
        const u32 c = G.pool_idx[i];
        if (c != AMH_NULL32 && !(R.cf[c] & SMC_SEND)) R.send[ns++] = c;
      }
      for (u32 c = 0; c < R.nc; c++) if (R.cf[c] & SMC_SEND) R.send[ns++] = c;
    }
  }
  // without those already sent in this session (sync.rs:187-196)
  for (u32 i = R.L[SM_SENT]; i < R.L[SM_SENT + 1]; i++) { u32 c = G.pool_idx[i]; if (c != AMH_NULL32) R.cf[c] |= SMC_SENT; }
  u32 kept = 0;
  for (u32 k = 0; k < ns; k++) if (!(R.cf[R.send[k]] & SMC_SENT)) R.send[kept++] = R.send[k];
  ns = kept;
  // nothing new to say (sync.rs:178-205)
  auto heads_are = [&](u32 l) {
    if (R.L[l + 1] - R.L[l] != nh) return false;
    for (u32 k = 0; k < nh; k++) if (!sm_hash_eq(sm_pool(G, R.L[l] + k), sm_chg_hash(G, R, R.heads[k]))) return false;
    return true;
  };
  const bool heads_unchanged = heads_are(SM_LAST_SENT);
  const bool heads_equal = (R.flags & SMF_THEIR_HEADS) && heads_are(SM_THEIR_HEADS);
  if (heads_unchanged && ((heads_equal && ns == 0) || (R.flags & SMF_IN_FLIGHT))) { G.r_kind[r] = SMK_NONE; return; }
  G.r_kind[r] = SMK_MESSAGE;
  G.r_nsend[r] = ns;
  G.job_base[r] = ns;
  u64 len = 1 + sm_hashes_len(nh) + sm_hashes_len(nn) + have_len + uleb_size(ns);
  for (u32 k = 0; k < ns; k++) { const u32 cl = B.c_out_len[R.cb + R.send[k]]; R.spos[k] = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len; len += uleb_size(cl) + cl; }
  G.msg_off[r] = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len;
}

AMH_DEV void k_sm_fill(const SyncDev& G, u32 r) {
  const u32 kind = G.r_kind[r];
  G.r_bits_at[r] = 0;
  if (G.raw || (kind != SMK_MESSAGE && kind != SMK_RESET)) return;
  const SmRec R = sm_rec(G, r);
  u8* o = G.msg_out + G.msg_off[r];   // the buffer was zeroed: the Bloom filter's bits are ORed into it by k_sm_bloom
  u32 pos = 0;
  o[pos++] = 0x42;
  const u32 nh = G.r_nheads[r];
  pos += wr_uleb(o + pos, nh);
  for (u32 k = 0; k < nh; k++) { const u8* h = sm_chg_hash(G, R, R.heads[k]); for (int b = 0; b < 32; b++) o[pos++] = h[b]; }
  if (kind == SMK_RESET) { o[pos++] = 0; o[pos++] = 1; o[pos++] = 0; o[pos++] = 0; o[pos++] = 0; return; }
  const u32 nn = G.r_nneed[r];
  pos += wr_uleb(o + pos, nn);
  for (u32 k = 0; k < nn; k++) { const u8* h = sm_pool(G, R.need[k]); for (int b = 0; b < 32; b++) o[pos++] = h[b]; }
  const u32 nb = G.r_bloom_n[r];
  if (nb == AMH_NULL32) {
    o[pos++] = 0;
  } else {
    o[pos++] = 1;
    const u32 nsh = R.L[SM_SHARED + 1] - R.L[SM_SHARED];
    pos += wr_uleb(o + pos, nsh);
    for (u32 k = 0; k < nsh; k++) { const u8* h = sm_pool(G, R.L[SM_SHARED] + k); for (int b = 0; b < 32; b++) o[pos++] = h[b]; }
    const u32 bl = bloom_wire_len(nb);
    pos += wr_uleb(o + pos, bl);
    if (bl) {
      const u32 head = wr_uleb(o + pos, nb);
      o[pos + head] = 10;
      o[pos + head + 1] = 7;
      G.r_bits_at[r] = G.msg_off[r] + pos + head + 2;
      pos += bl;
    }
  }
  pos += wr_uleb(o + pos, G.r_nsend[r]);
}
// the record an index of a per-record region belongs to: the last r with base[r] <= i (bases ascend; empty regions repeat)
template <class T>
AMH_DEV u32 sm_find_rec(const T* base, u32 n, u64 i) {
  u32 lo = 0, hi = n;
  while (hi - lo > 1) { const u32 mid = lo + (hi - lo) / 2; if ((u64)base[mid] <= i) lo = mid; else hi = mid; }
  return lo;
}
// one change of one record: if the message's Have holds it (SMC_BLOOM), its 7 probes (bloom.rs:69-102) into the bits
AMH_DEV void k_sm_bloom(const SyncDev& G, u32 i) {
  const u32 r = sm_find_rec(G.r_flag_base, G.n_rec, i);
  const u32 at = G.r_bits_at[r];
  if (at == 0) return;
  const u32 c = (u32)(i - G.r_flag_base[r]);
  if (!(G.c_flag[i] & SMC_BLOOM)) return;
  const u32 nb = G.r_bloom_n[r];
  const u32 nbytes = (u32)(((u64)nb * 10 + 7) / 8), modulo = 8 * nbytes;
  const u8* h = G.B.c_hash + (u64)(G.B.chg_base[G.r_doc[r]] + c) * 32;
  u32 x = ((u32)h[0] | ((u32)h[1] << 8) | ((u32)h[2] << 16) | ((u32)h[3] << 24)) % modulo;
  u32 y = ((u32)h[4] | ((u32)h[5] << 8) | ((u32)h[6] << 16) | ((u32)h[7] << 24)) % modulo;
  const u32 z = ((u32)h[8] | ((u32)h[9] << 8) | ((u32)h[10] << 16) | ((u32)h[11] << 24)) % modulo;
  for (u32 k = 0; k < 7; k++) {
    if (k) { x = (u32)(((u64)x + y) % modulo); y = (u32)(((u64)y + z) % modulo); }
    const u64 byte = (u64)at + (x >> 3);   // a byte of msg_out: set through the aligned word that holds it
    amh_atomic_or((u32*)(G.msg_out + (byte & ~(u64)3)), (1u << (x & 7)) << (8 * (u32)(byte & 3)));
  }
}
// one change sent by one record: length prefix, copy job, hash
AMH_DEV void k_sm_jobs(const SyncDev& G, u32 j) {
  const BatchDev& B = G.B;
  const u32 r = sm_find_rec(G.job_base, G.n_rec, j);
  const SmRec R = sm_rec(G, r);
  const u32 k = j - G.job_base[r];
  const u32 gc = R.cb + R.send[k], cl = B.c_out_len[gc];
  u64 at = (u64)G.msg_off[r] + R.spos[k];
  if (!G.raw) at += wr_uleb(G.msg_out + at, cl);
  G.job_src[j] = (u64)(uintptr_t)(B.out_arena + B.c_out_off[gc]);
  G.job_len[j] = cl;
  G.job_dst[j] = at;
  const u8* h = B.c_hash + (u64)gc * 32;
  for (int b = 0; b < 32; b++) G.sent_out[(u64)j * 32 + b] = h[b];
}
