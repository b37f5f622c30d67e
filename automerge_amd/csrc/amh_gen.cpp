// Synthetic workload generator: authors Automerge histories on the host and writes
// them in the reference's storage format (save_nocompress form: one document chunk per document,
// storage/document.rs:214-263; change hashes per storage/change.rs:444-512).  Independent of the
// device path and of oracle/: the tests check its output against the oracle (load + re-save identity).
//
// Workloads:
//   amh_gen_map_docs   BASELINE.json configs[1]: JSON-map documents, n_actors concurrent writers per
//                      round, Zipf keys, overwrites / conflicts / counters / deletes, one nested map.
//   amh_gen_text_docs  BASELINE.json configs[2]/[3]: text documents, a base change plus divergent
//                      replicas doing runs of inserts and deletes; each replica saved on its own.
#define AMH_HOST_HELPERS
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/automerge_hip.h"
#include "amh_dev.h"
#include "amh_ingest.h"

namespace {

struct Rng {
  u64 s;
  explicit Rng(u64 seed) : s(seed) {}
  u64 next() {  // splitmix64
    u64 z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  u32 below(u32 n) { return (u32)(next() % n); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Buf {
  std::vector<u8> v;
  void uleb(u64 x) { u8 t[10]; u32 n = wr_uleb(t, x); v.insert(v.end(), t, t + n); }
  void sleb(i64 x) { u8 t[10]; u32 n = wr_sleb(t, x); v.insert(v.end(), t, t + n); }
  void bytes(const void* p, size_t n) { const u8* b = (const u8*)p; v.insert(v.end(), b, b + n); }
  void append(const Buf& o) { v.insert(v.end(), o.v.begin(), o.v.end()); }
};

// ---- column encoders (columnar/encoding/rle.rs:31-133, delta.rs:10-43, boolean.rs:11-51) ------
// Values arrive as (is_null, value); runs are grouped first, then runs of length one are gathered
// into literal runs — the closed form of the reference's state machine.
template <class T, class W>
static void rle_encode(const std::vector<u8>& is_null, const std::vector<T>& vals, Buf& out, W write) {
  size_t n = vals.size();
  bool any = false;
  for (size_t i = 0; i < n && !any; i++) any = !is_null[i];
  if (!any) return;  // all null (or empty): nothing (rle.rs:33)
  size_t i = 0;
  while (i < n) {
    if (is_null[i]) {
      size_t j = i;
      while (j < n && is_null[j]) j++;
      out.sleb(0);
      out.uleb(j - i);
      i = j;
      continue;
    }
    size_t j = i + 1;
    while (j < n && !is_null[j] && vals[j] == vals[i]) j++;
    if (j - i >= 2) {
      out.sleb((i64)(j - i));
      write(out, vals[i]);
      i = j;
      continue;
    }
    // literal run: consecutive non-null values none of which starts a run of two
    size_t k = i;
    while (k < n && !is_null[k] && !(k + 1 < n && !is_null[k + 1] && vals[k + 1] == vals[k])) k++;
    out.sleb(-(i64)(k - i));
    for (size_t q = i; q < k; q++) write(out, vals[q]);
    i = k;
  }
}
static void rle_u64(const std::vector<u8>& nul, const std::vector<u64>& v, Buf& out) {
  rle_encode(nul, v, out, [](Buf& o, u64 x) { o.uleb(x); });
}
static void rle_str(const std::vector<u8>& nul, const std::vector<std::string>& v, Buf& out) {
  rle_encode(nul, v, out, [](Buf& o, const std::string& s) { o.uleb(s.size()); o.bytes(s.data(), s.size()); });
}
static void delta_i64(const std::vector<u8>& nul, const std::vector<i64>& v, Buf& out) {
  std::vector<i64> d(v.size());
  i64 abs = 0;
  for (size_t i = 0; i < v.size(); i++)
    if (!nul[i]) { d[i] = v[i] - abs; abs = v[i]; } else d[i] = 0;
  rle_encode(nul, d, out, [](Buf& o, i64 x) { o.sleb(x); });
}
static void bool_encode(const std::vector<u8>& v, Buf& out) {
  bool last = false;
  u64 cnt = 0;
  for (u8 b : v) {
    if ((b != 0) == last) cnt++;
    else { out.uleb(cnt); last = b != 0; cnt = 1; }
  }
  if (cnt) out.uleb(cnt);
}

// ---- authoring model ------------------------------------------------------------------------
struct Id {
  u32 ctr = 0, actor = 0;  // actor = index into the document's sorted actor table
  bool operator==(const Id& o) const { return ctr == o.ctr && actor == o.actor; }
  bool operator<(const Id& o) const { return ctr != o.ctr ? ctr < o.ctr : actor < o.actor; }  // lamport (types.rs:517-521): the table is sorted
};
enum { KEY_PROP = 0, KEY_HEAD = 1, KEY_ELEM = 2 };
enum { V_NULL = 0, V_INT = 4, V_STR = 6, V_COUNTER = 8 };
struct GOp {
  Id id, obj, elem;  // obj.ctr == 0: root
  u8 key_kind = KEY_PROP;
  std::string key;
  bool insert = false;
  u8 action = 1;  // types.rs:223-234
  u8 vtag = V_NULL;
  i64 ival = 0;
  std::string sval;
  std::vector<Id> pred, succ;
  std::vector<u32> succ_idx;  // op indices of succ (generator bookkeeping)
  bool dead = false;          // overwritten in an earlier round: invisible to every later op
  u32 epoch = 0;   // visibility bookkeeping of the generator (round / replica)
  u32 n_inc = 0;
};
struct GChange {
  u32 actor = 0;
  u64 seq = 0, start_op = 0;
  std::vector<u32> deps;   // indices of earlier changes
  std::vector<u32> ops;    // indices into the op store, in id order
  u8 hash[32];
  std::vector<u8> chunk;
};
struct GDoc {
  std::vector<std::string> actors;  // sorted bytewise
  std::vector<GOp> ops;
  std::vector<GChange> changes;
};

static void write_value(const GOp& o, u64& meta, Buf& raw) {
  size_t before = raw.v.size();
  switch (o.vtag) {
    case V_INT: case V_COUNTER: raw.sleb(o.ival); break;
    case V_STR: raw.bytes(o.sval.data(), o.sval.size()); break;
    default: break;
  }
  meta = ((u64)(raw.v.size() - before) << 4) | o.vtag;
}

static void put_col(Buf& meta, Buf& data, u32& ncols, u32 spec, const Buf& col) {
  if (col.v.empty()) return;  // raw_column.rs:168-171
  meta.uleb(spec);
  meta.uleb(col.v.size());
  data.append(col);
  ncols++;
}

static void chunk_wrap(u8 type, const Buf& body, std::vector<u8>& out, u8* hash32) {
  Buf pre;
  pre.v.push_back(type);
  pre.uleb(body.v.size());
  amh_host_sha256(pre.v.data(), pre.v.size(), body.v.data(), body.v.size(), hash32);
  out.clear();
  const u8 magic[4] = {0x85, 0x6f, 0x4a, 0x83};
  out.insert(out.end(), magic, magic + 4);
  out.insert(out.end(), hash32, hash32 + 4);
  out.insert(out.end(), pre.v.begin(), pre.v.end());
  out.insert(out.end(), body.v.begin(), body.v.end());
}

// change chunk (storage/change.rs:444-512; columns change_op_columns.rs:253-329)
static void build_change(const GDoc& D, GChange& c) {
  // actor table: author first, the others that ops mention in bytewise order (change_actors.rs:74-106)
  std::vector<u8> used(D.actors.size(), 0);
  for (u32 oi : c.ops) {
    const GOp& o = D.ops[oi];
    if (o.obj.ctr) used[o.obj.actor] = 1;
    if (o.key_kind == KEY_ELEM) used[o.elem.actor] = 1;
    for (const Id& p : o.pred) used[p.actor] = 1;
  }
  std::vector<u32> amap(D.actors.size(), 0xFFFFFFFFu), others;
  amap[c.actor] = 0;
  for (u32 a = 0; a < D.actors.size(); a++)
    if (a != c.actor && used[a]) { amap[a] = (u32)others.size() + 1; others.push_back(a); }
  size_t n = c.ops.size();
  std::vector<u8> obj_nul(n), ka_nul(n), kc_nul(n), ks_nul(n), ins(n), none(n, 0);
  std::vector<u64> obj_a(n), obj_c(n), key_a(n), action(n), vmeta(n), pnum(n), pa;
  std::vector<i64> key_c(n), pc;
  std::vector<std::string> key_s(n);
  Buf vraw;
  for (size_t i = 0; i < n; i++) {
    const GOp& o = D.ops[c.ops[i]];
    obj_nul[i] = o.obj.ctr == 0;
    obj_a[i] = o.obj.ctr ? amap[o.obj.actor] : 0;
    obj_c[i] = o.obj.ctr;
    ka_nul[i] = o.key_kind != KEY_ELEM;
    key_a[i] = o.key_kind == KEY_ELEM ? amap[o.elem.actor] : 0;
    kc_nul[i] = o.key_kind == KEY_PROP;
    key_c[i] = o.key_kind == KEY_ELEM ? o.elem.ctr : 0;
    ks_nul[i] = o.key_kind != KEY_PROP;
    if (o.key_kind == KEY_PROP) key_s[i] = o.key;
    ins[i] = o.insert;
    action[i] = o.action;
    write_value(o, vmeta[i], vraw);
    pnum[i] = o.pred.size();
    for (const Id& p : o.pred) { pa.push_back(amap[p.actor]); pc.push_back(p.ctr); }
  }
  std::vector<u8> pnone(pa.size(), 0);
  Buf c_obj_a, c_obj_c, c_key_a, c_key_c, c_key_s, c_ins, c_act, c_vm, c_pn, c_pa, c_pc;
  rle_u64(obj_nul, obj_a, c_obj_a); rle_u64(obj_nul, obj_c, c_obj_c);
  rle_u64(ka_nul, key_a, c_key_a); delta_i64(kc_nul, key_c, c_key_c); rle_str(ks_nul, key_s, c_key_s);
  bool_encode(ins, c_ins);
  rle_u64(none, action, c_act); rle_u64(none, vmeta, c_vm); rle_u64(none, pnum, c_pn);
  rle_u64(pnone, pa, c_pa); delta_i64(pnone, pc, c_pc);
  Buf meta, data;
  u32 ncols = 0;
  put_col(meta, data, ncols, 0x01, c_obj_a); put_col(meta, data, ncols, 0x02, c_obj_c);
  put_col(meta, data, ncols, 0x11, c_key_a); put_col(meta, data, ncols, 0x13, c_key_c); put_col(meta, data, ncols, 0x15, c_key_s);
  put_col(meta, data, ncols, 0x34, c_ins); put_col(meta, data, ncols, 0x42, c_act);
  put_col(meta, data, ncols, 0x56, c_vm); put_col(meta, data, ncols, 0x57, vraw);
  put_col(meta, data, ncols, 0x70, c_pn); put_col(meta, data, ncols, 0x71, c_pa); put_col(meta, data, ncols, 0x73, c_pc);
  Buf body;
  std::vector<const u8*> deps;
  for (u32 d : c.deps) deps.push_back(D.changes[d].hash);
  std::sort(deps.begin(), deps.end(), [](const u8* a, const u8* b) { return memcmp(a, b, 32) < 0; });
  body.uleb(deps.size());
  for (const u8* h : deps) body.bytes(h, 32);
  body.uleb(D.actors[c.actor].size());
  body.bytes(D.actors[c.actor].data(), D.actors[c.actor].size());
  body.uleb(c.seq);
  body.uleb(c.start_op);
  body.sleb(0);  // time
  body.uleb(0);  // no message
  body.uleb(others.size());
  for (u32 a : others) { body.uleb(D.actors[a].size()); body.bytes(D.actors[a].data(), D.actors[a].size()); }
  body.uleb(ncols);
  body.append(meta);
  body.append(data);
  chunk_wrap(1, body, c.chunk, c.hash);
}

// the rows of one object in document order; `seq_order` = element order of a sequence object
struct DocOrder {
  std::vector<u32> rows;  // op indices
};

// document chunk (storage/document.rs:214-263, doc_op_columns.rs:227-311, doc_change_columns.rs:118-162)
static void build_document(GDoc& D, const std::vector<u32>& rows, std::vector<u8>& out) {
  for (GChange& c : D.changes) build_change(D, c);
  size_t n = rows.size();
  std::vector<u8> obj_nul(n), ka_nul(n), kc_nul(n), ks_nul(n), ins(n), none(n, 0);
  std::vector<u64> obj_a(n), obj_c(n), key_a(n), id_a(n), action(n), vmeta(n), snum(n), sa;
  std::vector<i64> key_c(n), id_c(n), sc;
  std::vector<std::string> key_s(n);
  Buf vraw;
  for (size_t i = 0; i < n; i++) {
    GOp& o = D.ops[rows[i]];
    obj_nul[i] = o.obj.ctr == 0;
    obj_a[i] = o.obj.actor; obj_c[i] = o.obj.ctr;
    ka_nul[i] = o.key_kind != KEY_ELEM;
    key_a[i] = o.elem.actor;
    kc_nul[i] = o.key_kind == KEY_PROP;
    key_c[i] = o.key_kind == KEY_ELEM ? o.elem.ctr : 0;
    ks_nul[i] = o.key_kind != KEY_PROP;
    if (o.key_kind == KEY_PROP) key_s[i] = o.key;
    id_a[i] = o.id.actor; id_c[i] = o.id.ctr;
    ins[i] = o.insert;
    action[i] = o.action;
    write_value(o, vmeta[i], vraw);
    std::sort(o.succ.begin(), o.succ.end());
    snum[i] = o.succ.size();
    for (const Id& s : o.succ) { sa.push_back(s.actor); sc.push_back(s.ctr); }
  }
  std::vector<u8> snone(sa.size(), 0);
  Buf c_obj_a, c_obj_c, c_key_a, c_key_c, c_key_s, c_id_a, c_id_c, c_ins, c_act, c_vm, c_sn, c_sa, c_sc;
  rle_u64(obj_nul, obj_a, c_obj_a); rle_u64(obj_nul, obj_c, c_obj_c);
  rle_u64(ka_nul, key_a, c_key_a); delta_i64(kc_nul, key_c, c_key_c); rle_str(ks_nul, key_s, c_key_s);
  rle_u64(none, id_a, c_id_a); delta_i64(none, id_c, c_id_c);
  bool_encode(ins, c_ins);
  rle_u64(none, action, c_act); rle_u64(none, vmeta, c_vm); rle_u64(none, snum, c_sn);
  rle_u64(snone, sa, c_sa); delta_i64(snone, sc, c_sc);
  Buf ometa, odata;
  u32 nocols = 0;
  put_col(ometa, odata, nocols, 0x01, c_obj_a); put_col(ometa, odata, nocols, 0x02, c_obj_c);
  put_col(ometa, odata, nocols, 0x11, c_key_a); put_col(ometa, odata, nocols, 0x13, c_key_c); put_col(ometa, odata, nocols, 0x15, c_key_s);
  put_col(ometa, odata, nocols, 0x21, c_id_a); put_col(ometa, odata, nocols, 0x23, c_id_c);
  put_col(ometa, odata, nocols, 0x34, c_ins); put_col(ometa, odata, nocols, 0x42, c_act);
  put_col(ometa, odata, nocols, 0x56, c_vm); put_col(ometa, odata, nocols, 0x57, vraw);
  put_col(ometa, odata, nocols, 0x80, c_sn); put_col(ometa, odata, nocols, 0x81, c_sa); put_col(ometa, odata, nocols, 0x83, c_sc);
  // change metadata
  size_t nc = D.changes.size();
  std::vector<u8> cnone(nc, 0), allnull(nc, 1);
  std::vector<u64> c_actor(nc), c_dn(nc), c_em(nc, 7);  // extra bytes: empty Bytes value (doc_change_columns.rs:102-105)
  std::vector<i64> c_seq(nc), c_max(nc), c_time(nc, 0), c_di;
  std::vector<u8> is_dep(nc, 0);
  for (size_t i = 0; i < nc; i++) {
    const GChange& c = D.changes[i];
    c_actor[i] = c.actor; c_seq[i] = (i64)c.seq; c_max[i] = (i64)(c.start_op + c.ops.size() - 1);
    std::vector<u32> deps = c.deps;  // in the order of the change's sorted dep hashes (save/document.rs:141-146)
    std::sort(deps.begin(), deps.end(), [&](u32 a, u32 b) { return memcmp(D.changes[a].hash, D.changes[b].hash, 32) < 0; });
    c_dn[i] = deps.size();
    for (u32 d : deps) { c_di.push_back(d); is_dep[d] = 1; }
  }
  std::vector<u8> dnone(c_di.size(), 0);
  Buf b_actor, b_seq, b_max, b_time, b_dn, b_di, b_em;
  rle_u64(cnone, c_actor, b_actor); delta_i64(cnone, c_seq, b_seq); delta_i64(cnone, c_max, b_max); delta_i64(cnone, c_time, b_time);
  rle_u64(cnone, c_dn, b_dn); delta_i64(dnone, c_di, b_di); rle_u64(cnone, c_em, b_em);
  Buf cmeta, cdata;
  u32 nccols = 0;
  put_col(cmeta, cdata, nccols, 0x01, b_actor); put_col(cmeta, cdata, nccols, 0x03, b_seq); put_col(cmeta, cdata, nccols, 0x13, b_max);
  put_col(cmeta, cdata, nccols, 0x23, b_time); put_col(cmeta, cdata, nccols, 0x40, b_dn); put_col(cmeta, cdata, nccols, 0x43, b_di);
  put_col(cmeta, cdata, nccols, 0x56, b_em);
  // heads: changes nobody depends on, sorted by hash, with their indices
  std::vector<u32> heads;
  for (size_t i = 0; i < nc; i++) if (!is_dep[i]) heads.push_back((u32)i);
  std::sort(heads.begin(), heads.end(), [&](u32 a, u32 b) { return memcmp(D.changes[a].hash, D.changes[b].hash, 32) < 0; });
  Buf body;
  body.uleb(D.actors.size());
  for (auto& a : D.actors) { body.uleb(a.size()); body.bytes(a.data(), a.size()); }
  body.uleb(heads.size());
  for (u32 h : heads) body.bytes(D.changes[h].hash, 32);
  body.uleb(nccols); body.append(cmeta);
  body.uleb(nocols); body.append(ometa);
  body.append(cdata);
  body.append(odata);
  for (u32 h : heads) body.uleb(h);
  u8 hash[32];
  chunk_wrap(0, body, out, hash);
}

static std::vector<std::string> make_actors(Rng& rng, u32 n) {
  std::vector<std::string> a;
  for (u32 i = 0; i < n; i++) {
    std::string s(16, 0);
    for (int k = 0; k < 16; k += 8) { u64 x = rng.next(); memcpy(&s[k], &x, 8); }
    a.push_back(s);
  }
  std::sort(a.begin(), a.end());
  a.erase(std::unique(a.begin(), a.end()), a.end());
  return a;
}

struct Zipf {
  std::vector<double> cdf;
  explicit Zipf(u32 n) {
    double tot = 0;
    for (u32 i = 1; i <= n; i++) tot += 1.0 / i;
    double run = 0;
    for (u32 i = 1; i <= n; i++) { run += 1.0 / i / tot; cdf.push_back(run); }
  }
  u32 draw(Rng& r) const {
    double u = r.unit();
    return (u32)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) % (u32)cdf.size();
  }
};

// ---- map documents ----------------------------------------------------------------------------
// Rounds of concurrent changes: in round r every actor commits one change of ops_per_change ops on
// top of all changes of round r-1, so ops of one round share counters and differ by actor (lamport
// tie-breaks), overwrite each other's earlier values and leave conflicts behind.
static void gen_map_doc(u64 seed, u32 n_actors, u32 rounds, u32 ops_per_change, const std::vector<std::string>& keys, const Zipf& zipf,
                        std::vector<u8>& out) {
  Rng rng(seed);
  GDoc D;
  D.actors = make_actors(rng, n_actors);
  n_actors = (u32)D.actors.size();
  // per (object, key): op indices in creation order
  struct Slot { std::vector<u32> ops; };
  std::vector<Slot> slots(2 * keys.size() + 1);
  const u32 NESTED_SLOT = (u32)(2 * keys.size());
  D.ops.reserve((size_t)rounds * n_actors * ops_per_change);
  Id nested_obj{};        // set by the first op of the document
  u32 nested_epoch = 0xFFFFFFFFu, nested_actor = 0;
  for (u32 r = 0; r < rounds; r++) {
    u64 start_op = (u64)r * ops_per_change + 1;
    for (u32 a = 0; a < n_actors; a++) {
      GChange c;
      c.actor = a;
      c.seq = r + 1;
      c.start_op = start_op;
      if (r > 0) for (u32 k = 0; k < n_actors; k++) c.deps.push_back((r - 1) * n_actors + k);
      auto knows = [&](const GOp& o) { return o.epoch < r || (o.epoch == r && o.id.actor == a); };
      for (u32 i = 0; i < ops_per_change; i++) {
        GOp o;
        o.id = Id{(u32)(start_op + i), a};
        o.epoch = r;
        bool first = r == 0 && a == 0 && i == 0;
        bool nested_known = nested_epoch != 0xFFFFFFFFu && (nested_epoch < r || nested_actor == a);
        u32 slot;
        if (first) {
          slot = NESTED_SLOT;
          o.key = "nested";
          o.action = 0;  // makeMap
        } else {
          bool in_nested = nested_known && rng.below(4) == 0;
          u32 k = zipf.draw(rng);
          slot = (in_nested ? (u32)keys.size() : 0) + k;
          o.key = keys[k];
          if (in_nested) o.obj = nested_obj;
        }
        // what this actor sees at the key: ops it knows that no known op has overwritten
        std::vector<u32> vis;
        for (u32 oi : slots[slot].ops) {
          const GOp& p = D.ops[oi];
          if (p.dead || !knows(p) || p.action == 5 || p.action == 3) continue;
          u32 hidden = 0, incs = 0, old_hidden = 0, old_incs = 0;
          for (u32 qi : p.succ_idx) {
            const GOp& q = D.ops[qi];
            if (!knows(q)) continue;
            hidden++;
            if (q.action == 5) incs++;
            if (q.epoch < r) { old_hidden++; if (q.action == 5) old_incs++; }
          }
          bool counter = p.vtag == V_COUNTER && p.action == 1;
          if (counter ? old_hidden > old_incs : old_hidden > 0) D.ops[oi].dead = true;  // every later op knows those succs too
          bool visible = counter ? hidden <= incs : hidden == 0;
          if (visible) vis.push_back(oi);
        }
        {  // forget ops nobody can see any more
          auto& so = slots[slot].ops;
          so.erase(std::remove_if(so.begin(), so.end(), [&](u32 oi) { return D.ops[oi].dead; }), so.end());
        }
        if (!first) {
          u32 dice = rng.below(100);
          bool all_counters = !vis.empty();
          for (u32 oi : vis) all_counters = all_counters && D.ops[oi].action == 1 && D.ops[oi].vtag == V_COUNTER;
          if (dice < 5 && !vis.empty()) {
            o.action = 3;  // delete
          } else if (dice < 10) {
            if (all_counters) { o.action = 5; o.vtag = V_INT; o.ival = (i64)rng.below(20) - 5; }
            else { o.action = 1; o.vtag = V_COUNTER; o.ival = (i64)rng.below(1000); }
          } else if (dice < 40) {
            o.action = 1; o.vtag = V_STR;
            u32 len = 4 + rng.below(21);
            for (u32 q = 0; q < len; q++) o.sval.push_back((char)('a' + rng.below(26)));
          } else {
            o.action = 1; o.vtag = V_INT; o.ival = (i64)rng.below(2000) - 1000;
          }
        }
        for (u32 oi : vis) o.pred.push_back(D.ops[oi].id);
        std::sort(o.pred.begin(), o.pred.end());
        u32 idx = (u32)D.ops.size();
        for (u32 oi : vis) { D.ops[oi].succ.push_back(o.id); D.ops[oi].succ_idx.push_back(idx); }
        if (first) { nested_obj = o.id; nested_epoch = r; nested_actor = a; }
        D.ops.push_back(std::move(o));
        slots[slot].ops.push_back(idx);
        c.ops.push_back(idx);
      }
      D.changes.push_back(std::move(c));
    }
  }
  // document order: objects by id (root first), keys bytewise, ids lamport; deletes have no row
  std::vector<u32> rows;
  for (u32 i = 0; i < D.ops.size(); i++) if (D.ops[i].action != 3) rows.push_back(i);
  std::sort(rows.begin(), rows.end(), [&](u32 x, u32 y) {
    const GOp &a = D.ops[x], &b = D.ops[y];
    if (!(a.obj == b.obj)) return a.obj < b.obj;
    int c = a.key.compare(b.key);
    if (c) return c < 0;
    return a.id < b.id;
  });
  build_document(D, rows, out);
}

// ---- text documents ---------------------------------------------------------------------------
// One text object at root key "text".  The base change (actor 0) types base_len characters; every
// replica then forks from the base and applies ops_per_replica edits in changes of ops_per_change:
// runs of inserts at a random visible position (geometric run length, p = 0.2) and deletes.
// Output per document: n_replicas saved documents (base + that replica's changes).
struct Elem { u32 op; bool deleted; };
// The element order of the text object under random edits: blocks of at most 2 * SEQ_BLOCK elements with their visible
// counts, so that "the k-th visible element" and "insert here" cost a scan of the block table plus one block (the flat
// vector this replaces made a 10M-op document quadratic).  Same positions, same documents as before.
struct ElemSeq {
  enum { SEQ_BLOCK = 4096 };
  struct Block { std::vector<Elem> e; u32 vis = 0; };
  std::vector<Block> blocks;
  struct Pos { u32 b, i; };   // block, index inside it (i == size: behind the block's last element)
  ElemSeq() { blocks.emplace_back(); }
  void push_back(Elem x) {
    if (blocks.back().e.size() >= SEQ_BLOCK) blocks.emplace_back();
    blocks.back().e.push_back(x);
    blocks.back().vis += x.deleted ? 0 : 1;
  }
  // the k-th (0-based) visible element
  Pos kth_visible(u32 k) const {
    u32 b = 0;
    while (k >= blocks[b].vis) { k -= blocks[b].vis; b++; }
    const std::vector<Elem>& e = blocks[b].e;
    u32 i = 0;
    for (u32 seen = 0;; i++) if (!e[i].deleted) { if (seen == k) break; seen++; }
    return Pos{b, i};
  }
  Elem& at(Pos p) { return blocks[p.b].e[p.i]; }
  void mark_deleted(Pos p) { at(p).deleted = true; blocks[p.b].vis--; }
  // insert in front of p; returns the position behind the new element
  Pos insert(Pos p, Elem x) {
    Block& bl = blocks[p.b];
    bl.e.insert(bl.e.begin() + p.i, x);
    bl.vis += x.deleted ? 0 : 1;
    p.i++;
    if (bl.e.size() >= 2 * SEQ_BLOCK) {   // split; the cursor follows its element
      Block nb;
      nb.e.assign(bl.e.begin() + SEQ_BLOCK, bl.e.end());
      bl.e.resize(SEQ_BLOCK);
      for (const Elem& q : nb.e) nb.vis += q.deleted ? 0 : 1;
      bl.vis -= nb.vis;
      blocks.insert(blocks.begin() + p.b + 1, std::move(nb));
      if (p.i > SEQ_BLOCK) { p.b++; p.i -= SEQ_BLOCK; }
    }
    return p;
  }
};
static void gen_text_doc(u64 seed, u32 n_replicas, u32 base_len, u32 ops_per_replica, u32 ops_per_change, std::vector<std::vector<u8>>& outs) {
  Rng rng0(seed);
  std::vector<std::string> actors = make_actors(rng0, n_replicas + 1);
  u32 n_act = (u32)actors.size();
  u32 base_actor_g = (u32)(rng0.next() % n_act);
  // the base change, authored against a two-actor table {base author, replica author}
  auto make_base = [&](GDoc& base, u32 base_actor) {
    Rng rb(seed ^ 0x7e57ba5eull);
    GChange c;
    c.actor = base_actor; c.seq = 1; c.start_op = 1;
    GOp mk;
    mk.id = Id{1, base_actor}; mk.key = "text"; mk.action = 4;  // makeText
    base.ops.push_back(mk);
    c.ops.push_back(0);
    Id prev{};
    for (u32 i = 0; i < base_len; i++) {
      GOp o;
      o.id = Id{2 + i, base_actor}; o.obj = mk.id; o.insert = true; o.action = 1;
      o.key_kind = i == 0 ? KEY_HEAD : KEY_ELEM; o.elem = prev;
      o.vtag = V_STR; o.sval = std::string(1, (char)('a' + rb.below(26)));
      prev = o.id;
      c.ops.push_back((u32)base.ops.size());
      base.ops.push_back(o);
    }
    base.changes.push_back(c);
  };
  u32 rep = 0;
  for (u32 ag = 0; ag < n_act && rep < n_replicas; ag++) {
    if (ag == base_actor_g) continue;
    Rng rng(seed ^ (0x51ed270b9f3ull * (rep + 1)));
    GDoc D;
    u32 a = ag < base_actor_g ? 0 : 1, base_actor = 1 - a;
    D.actors.resize(2);
    D.actors[a] = actors[ag];
    D.actors[base_actor] = actors[base_actor_g];
    make_base(D, base_actor);
    ElemSeq seq;  // element order of the text object
    for (u32 i = 1; i < D.ops.size(); i++) seq.push_back(Elem{i, false});
    u32 n_vis = (u32)D.ops.size() - 1;
    u32 ctr = (u32)D.ops.size() + 1;  // next op counter
    u32 done = 0, seqno = 1, last_change = 0;
    while (done < ops_per_replica) {
      GChange c;
      c.actor = a; c.seq = seqno++; c.start_op = ctr;
      c.deps.push_back(last_change);
      u32 in_change = std::min(ops_per_change, ops_per_replica - done);
      u32 made = 0;
      while (made < in_change) {
        bool del = n_vis > 0 && rng.below(100) < 30;
        if (del) {
          // the k-th visible element
          u32 k = rng.below(n_vis);
          ElemSeq::Pos pos = seq.kth_visible(k);
          const u32 target = seq.at(pos).op;
          GOp o;
          o.id = Id{ctr++, a}; o.obj = D.ops[0].id; o.action = 3;
          o.key_kind = KEY_ELEM; o.elem = D.ops[target].id;
          o.pred.push_back(D.ops[target].id);
          D.ops[target].succ.push_back(o.id);
          seq.mark_deleted(pos);
          n_vis--;
          c.ops.push_back((u32)D.ops.size());
          D.ops.push_back(o);
          made++;
          continue;
        }
        // a run of inserts after the k-th visible element (k = 0: at the head)
        u32 k = rng.below(n_vis + 1);  // insert behind the k-th visible element; 0 = at the head
        ElemSeq::Pos pos{0, 0};
        Id prev{};
        bool head = k == 0;
        if (k > 0) { pos = seq.kth_visible(k - 1); prev = D.ops[seq.at(pos).op].id; pos.i++; }
        // RGA: a new element goes right after its reference element, before that element's older
        // children; a reference that is deleted keeps its place, so insert at `pos`
        u32 run = 1;
        while (rng.below(5) != 0) run++;
        run = std::min(run, in_change - made);
        for (u32 q = 0; q < run; q++) {
          GOp o;
          o.id = Id{ctr++, a}; o.obj = D.ops[0].id; o.insert = true; o.action = 1;
          o.key_kind = head ? KEY_HEAD : KEY_ELEM; o.elem = prev;
          o.vtag = V_STR; o.sval = std::string(1, (char)('a' + rng.below(26)));
          u32 idx = (u32)D.ops.size();
          D.ops.push_back(o);
          c.ops.push_back(idx);
          pos = seq.insert(pos, Elem{idx, false});
          n_vis++;
          prev = D.ops[idx].id; head = false;
          made++;
        }
      }
      done += made;
      last_change = (u32)D.changes.size();
      D.changes.push_back(std::move(c));
    }
    std::vector<u32> rows;
    rows.push_back(0);
    for (const ElemSeq::Block& bl : seq.blocks) for (const Elem& e : bl.e) rows.push_back(e.op);
    outs.emplace_back();
    build_document(D, rows, outs.back());
    rep++;
  }
}

template <class F>
static int run_parallel(u32 n, int threads, F f) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; t++)
    ts.emplace_back([&, t]() { for (u32 i = (u32)t; i < n; i += (u32)threads) f(i); });
  for (auto& th : ts) th.join();
  return 0;
}

static int pack(const std::vector<std::vector<u8>>& docs, amh_bytes_out* out, uint64_t** offsets) {
  size_t tot = 0;
  for (auto& d : docs) tot += d.size();
  out->data = (uint8_t*)malloc(tot ? tot : 1);
  *offsets = (uint64_t*)malloc((docs.size() + 1) * sizeof(uint64_t));
  if (!out->data || !*offsets) { free(out->data); free(*offsets); out->data = nullptr; *offsets = nullptr; return AMH_ERR_OOM; }
  size_t off = 0;
  for (size_t i = 0; i < docs.size(); i++) {
    (*offsets)[i] = off;
    if (!docs[i].empty()) memcpy(out->data + off, docs[i].data(), docs[i].size());
    off += docs[i].size();
  }
  (*offsets)[docs.size()] = off;
  out->len = tot;
  return AMH_SUCCESS;
}

}  // namespace

extern "C" int amh_gen_map_docs(uint64_t seed, uint32_t n_docs, uint32_t n_actors, uint32_t changes_per_actor, uint32_t ops_per_change,
                                uint32_t n_keys, int threads, amh_bytes_out* out, uint64_t** offsets) {
  if (!out || !offsets || n_actors == 0 || n_actors > 256 || changes_per_actor == 0 || ops_per_change == 0 || n_keys == 0 || n_keys > 100000)
    return AMH_ERR_INVALID_ARG;
  std::vector<std::string> keys;
  for (u32 i = 0; i < n_keys; i++) { char b[16]; snprintf(b, sizeof b, "k%03u", i); keys.push_back(b); }
  Zipf zipf(n_keys);
  std::vector<std::vector<u8>> docs(n_docs);
  run_parallel(n_docs, threads, [&](u32 i) { gen_map_doc(seed + i, n_actors, changes_per_actor, ops_per_change, keys, zipf, docs[i]); });
  return pack(docs, out, offsets);
}

extern "C" int amh_gen_text_docs(uint64_t seed, uint32_t n_docs, uint32_t n_replicas, uint32_t base_len, uint32_t ops_per_replica,
                                 uint32_t ops_per_change, int threads, amh_bytes_out* out, uint64_t** offsets) {
  if (!out || !offsets || n_replicas == 0 || n_replicas > 255 || ops_per_change == 0) return AMH_ERR_INVALID_ARG;
  std::vector<std::vector<std::vector<u8>>> per(n_docs);
  run_parallel(n_docs, threads, [&](u32 i) { gen_text_doc(seed + i, n_replicas, base_len, ops_per_replica, ops_per_change, per[i]); });
  std::vector<std::vector<u8>> docs;
  for (auto& p : per) for (auto& d : p) docs.push_back(std::move(d));
  return pack(docs, out, offsets);
}
