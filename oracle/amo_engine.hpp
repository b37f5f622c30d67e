// ORACLE — TEST INFRASTRUCTURE ONLY (see amo_codec.hpp header).
//
// CPU restatement of the reference's document engine for the bulk path: load (document
// reconstruction with change re-derivation), apply_changes / merge (sequential op insertion exactly
// as the reference's op-tree search places ops), save, heads, Bloom summary and the canonical
// materialization used for parity.  Paths cited are relative to /root/reference/rust/automerge/src.
#pragma once
#include <cmath>
#include <functional>
#include <unordered_map>

#include "amo_storage.hpp"

namespace amo {

// in-memory op — types.rs:625-633 (+ Counter state of value.rs:369-393)
struct Op {
  OpId id;
  bool key_is_prop = false;
  std::string prop;
  OpId elem;
  bool insert = false;
  uint64_t action = 0;  // action index, types.rs:223-234
  Scalar val;
  bool expand = false;
  std::optional<std::string> mark_name;
  std::vector<OpId> succ, pred;
  int64_t counter_current = 0;
  std::vector<std::pair<OpId, int64_t>> incs;

  bool is_counter() const { return action == 1 && val.tag == 8; }
  bool is_inc() const { return action == 5; }
  bool is_mark() const { return action == 7; }
  bool is_make() const { return action == 0 || action == 2 || action == 4 || action == 6; }
  bool is_delete() const { return action == 3; }
  bool visible() const {  // types.rs:712-720
    if (is_inc() || is_mark()) return false;
    if (is_counter()) return succ.size() <= incs.size();
    return succ.empty();
  }
};

// OpType::from_action_and_value (types.rs:252-276) followed by the value/expand/mark_name a
// re-encoder reads back (op_as_changeop.rs:94-145, op_as_docop.rs:89-128): what survives per action.
inline void normalize_op(Op& op) {
  switch (op.action) {
    case 0: case 2: case 4: case 6: case 3:
      op.val = Scalar(); op.expand = false; op.mark_name.reset(); break;
    case 1:
      op.expand = false; op.mark_name.reset();
      if (op.val.tag == 8) op.counter_current = op.val.i;
      break;
    case 5: {
      int64_t i = op.val.tag == 3 ? (int64_t)op.val.u : op.val.i;
      if (op.val.tag != 3 && op.val.tag != 4) throw DecodeError("non numeric increment");
      op.val = sc_int(i); op.expand = false; op.mark_name.reset(); break;
    }
    case 7:
      if (!op.mark_name) op.val = Scalar();
      break;
    default: throw DecodeError("unknown action");
  }
}

struct ObjTree {  // op_tree.rs OpTree: {internal, objtype, parent}
  uint64_t objtype = 0;  // make-action code: 0 map, 2 list, 4 text, 6 table
  std::optional<OpId> parent;
  std::vector<Op> ops;
  bool is_seq() const { return objtype == 2 || objtype == 4; }
};

struct Doc {  // automerge.rs:166-185
  std::vector<Bytes> actors;                     // ops.m.actors (IndexedCache, insertion order)
  std::map<Bytes, uint32_t> actor_lookup;
  std::vector<StoredChange> history;
  std::map<Hash, size_t> history_index;
  std::map<uint32_t, std::vector<size_t>> states;
  std::set<Hash> deps;
  uint64_t max_op = 0;
  std::vector<StoredChange> queue;
  std::map<OpId, ObjTree> trees;                  // keyed by (ctr, actor index)

  Doc() { trees[OpId{0, 0}] = ObjTree(); }
  bool is_empty() const { return history.empty() && queue.empty(); }
  uint32_t cache_actor(const Bytes& a) {          // indexed_cache.rs:32-41
    auto it = actor_lookup.find(a);
    if (it != actor_lookup.end()) return it->second;
    uint32_t n = (uint32_t)actors.size();
    actors.push_back(a);
    actor_lookup[a] = n;
    return n;
  }
  int lamport_cmp(const OpId& a, const OpId& b) const {  // types.rs:517-521
    if (a.ctr != b.ctr) return a.ctr < b.ctr ? -1 : 1;
    const Bytes& x = actors[a.actor];
    const Bytes& y = actors[b.actor];
    if (x < y) return -1;
    if (y < x) return 1;
    return 0;
  }
  std::vector<Hash> get_heads() const { return std::vector<Hash>(deps.begin(), deps.end()); }  // automerge.rs:1300 (BTreeSet order = sorted)
};

inline Error inflate_err(ErrSub s, const std::string& m) { return Error(E_INFLATE_DOCUMENT, m, s); }

// ChangeActors::new + translate (storage/change/change_actors.rs:70-123) and op_as_actor_id
// (convert/op_as_changeop.rs): doc-actor-indexed ops of one change -> change-actor-indexed rows.
inline StoredChange build_change_from_ops(const Doc& doc, uint32_t author, std::vector<Hash> deps, uint64_t seq,
                                          uint64_t start_op, int64_t ts, const std::string& message,
                                          const Bytes& extra, const std::vector<std::pair<OpId, const Op*>>& ops) {
  const Bytes& author_bytes = doc.actors[author];
  std::set<Bytes> others;
  auto note = [&](const OpId& o) {
    const Bytes& a = doc.actors[o.actor];
    if (a != author_bytes) others.insert(a);
  };
  for (auto& po : ops) {
    const Op& op = *po.second;
    if (!op.key_is_prop && op.elem.ctr != 0) note(op.elem);
    for (auto& p : op.pred) note(p);
    if (po.first.ctr != 0) note(po.first);
  }
  std::vector<Bytes> other_actors(others.begin(), others.end());
  std::map<Bytes, uint32_t> index;
  index[author_bytes] = 0;
  for (size_t i = 0; i < other_actors.size(); i++) index[other_actors[i]] = (uint32_t)(i + 1);
  auto tr = [&](const OpId& o) { return OpId{o.ctr, index.at(doc.actors[o.actor])}; };
  std::vector<StoredOp> rows;
  rows.reserve(ops.size());
  for (auto& po : ops) {
    const Op& op = *po.second;
    StoredOp r;
    r.obj = po.first.ctr == 0 ? OpId{0, 0} : tr(po.first);
    r.key_is_prop = op.key_is_prop;
    r.prop = op.prop;
    r.elem = (op.key_is_prop || op.elem.ctr == 0) ? OpId{0, 0} : tr(op.elem);
    r.insert = op.insert;
    r.action = op.action;
    r.val = op.val;
    for (auto& p : op.pred) r.ids.push_back(tr(p));
    r.expand = op.is_mark() ? op.expand : false;
    r.mark_name = op.mark_name;
    rows.push_back(std::move(r));
  }
  return build_change(std::move(deps), author_bytes, other_actors, seq, start_op, ts, message, extra, rows);
}

// ---------------------------------------------------------------------------------------------
// reconstruct_document — storage/load/reconstruct_document.rs:93-230 with LoadingObject :236-340,
// import_op :342-375, ChangeCollector (storage/load/change_collector.rs, whole file)
// ---------------------------------------------------------------------------------------------
struct Reconstructed {
  uint64_t max_op = 0;
  std::vector<StoredChange> changes;
  std::set<Hash> heads;
};

inline Reconstructed reconstruct_document(const ParsedDocument& pd, bool check_heads, Doc* out_doc) {
  Doc tmp;
  Doc& doc = out_doc ? *out_doc : tmp;
  doc.actors = pd.actors;  // OpSetMetadata::from_actors (op_set.rs:437-442): file order, last duplicate wins in lookup
  doc.actor_lookup.clear();
  for (size_t i = 0; i < pd.actors.size(); i++) doc.actor_lookup[pd.actors[i]] = (uint32_t)i;
  size_t n_actors = doc.actors.size();

  // ---- ChangeCollector::new (:45-78)
  struct Partial {
    size_t index;
    ChangeMeta meta;
    std::vector<std::pair<OpId, Op>> ops;
  };
  std::map<uint64_t, std::vector<Partial>> by_actor;
  std::vector<ChangeMeta> metas;
  try {
    metas = read_change_meta(pd);
  } catch (DecodeError& e) {
    throw inflate_err(S_READ_CHANGE, e.what());
  } catch (ReadChangeError& e) {
    throw inflate_err(S_READ_CHANGE, e.what());
  }
  for (size_t i = 0; i < metas.size(); i++) {
    auto& v = by_actor[metas[i].actor];
    if (!v.empty() && v.back().meta.max_op > metas[i].max_op) throw inflate_err(S_CHANGES_OUT_OF_ORDER, "changes out of order");
    v.push_back(Partial{i, metas[i], {}});
  }
  auto collect = [&](const OpId& obj, const Op& op) {  // :81-96
    auto it = by_actor.find(op.id.actor);
    if (it == by_actor.end()) throw inflate_err(S_COLLECT_MISSING_ACTOR, "missing actor for op");
    auto& v = it->second;
    size_t idx = 0;
    while (idx < v.size() && v[idx].meta.max_op < op.id.ctr) idx++;  // partition_point
    if (idx >= v.size()) throw inflate_err(S_MISSING_CHANGE, "missing change for op");
    v[idx].ops.push_back({obj, op});
  };

  // ---- main loop (:120-207)
  struct Loading {
    OpId id;
    std::optional<OpId> parent;
    uint64_t objtype = 0;
    std::vector<Op> ops;
    std::map<OpId, std::vector<OpId>> preds;
    struct KeyRef { bool is_prop; std::string prop; OpId elem; };
    std::map<OpId, KeyRef> set_ops;
    std::map<OpId, int64_t> inc_ops;
  };
  auto sort_lamport = [&](std::vector<OpId>& v) {
    std::stable_sort(v.begin(), v.end(), [&](const OpId& a, const OpId& b) { return doc.lamport_cmp(a, b) < 0; });
  };
  auto finish_obj = [&](Loading& lo) {  // :292-339
    ObjTree tree;
    tree.objtype = lo.objtype;
    tree.parent = lo.parent;
    for (auto& op0 : lo.ops) {
      Op op = op0;
      auto pit = lo.preds.find(op.id);
      if (pit != lo.preds.end()) {
        op.pred = pit->second;
        sort_lamport(op.pred);
        lo.preds.erase(pit);
      }
      if (op.is_counter()) {
        for (auto& s : op.succ) {
          auto iit = lo.inc_ops.find(s);
          if (iit != lo.inc_ops.end()) {
            op.counter_current += iit->second;
            op.incs.push_back({s, iit->second});
          }
        }
      }
      collect(lo.id, op);
      tree.ops.push_back(std::move(op));
    }
    for (auto& kv : lo.preds) {
      auto sit = lo.set_ops.find(kv.second[0]);
      if (sit == lo.set_ops.end()) throw inflate_err(S_MISSING_OPS, "no delete operation found");
      Op del;
      del.id = kv.first;
      del.pred = kv.second;
      sort_lamport(del.pred);
      del.insert = false;
      del.key_is_prop = sit->second.is_prop;
      del.prop = sit->second.prop;
      del.elem = sit->second.elem;
      del.action = 3;
      collect(lo.id, del);
    }
    doc.trees[lo.id] = std::move(tree);
  };
  auto append_op = [&](Loading& lo, const Op& op) {  // :267-290
    if (op.action == 1 || op.is_make()) {
      Loading::KeyRef k;
      k.is_prop = op.key_is_prop;
      k.prop = op.prop;
      k.elem = op.key_is_prop ? OpId{0, 0} : (op.insert ? op.id : op.elem);
      lo.set_ops[op.id] = k;
    }
    if (op.is_inc()) lo.inc_ops[op.id] = op.val.i;
    for (auto& s : op.succ) lo.preds[s].push_back(op.id);
    lo.ops.push_back(op);
  };

  doc.trees.clear();
  Reconstructed rec;
  Loading cur;
  cur.id = OpId{0, 0};
  cur.objtype = 0;
  struct CreateOp { uint64_t objtype; OpId parent; };
  std::map<OpId, CreateOp> create_ops;
  std::set<OpId> objs_loaded;
  uint64_t max_op = 0;

  OpRowIter it(pd.op_data.data(), pd.op_cols, true);
  StoredOp row;
  for (;;) {
    bool more;
    try {
      more = it.next(row);
    } catch (DecodeError& e) {
      throw inflate_err(S_READ_OP, e.what());
    }
    if (!more) break;
    max_op = std::max<uint64_t>(max_op, row.id.ctr);
    for (auto& s : row.ids) max_op = std::max<uint64_t>(max_op, s.ctr);
    OpId obj = row.obj;
    auto check_actor = [&](const OpId& o) {  // check_opid :367-375
      if (o.actor >= n_actors) throw inflate_err(S_MISSING_ACTOR, "missing actor");
    };
    check_actor(obj);
    // import_op :342-362
    Op op;
    op.key_is_prop = row.key_is_prop;
    op.prop = row.prop;
    if (!row.key_is_prop) { check_actor(row.elem); op.elem = row.elem; }
    for (auto& s : row.ids) check_actor(s);
    op.action = row.action;
    op.val = row.val;
    op.expand = row.expand;
    op.mark_name = row.mark_name;
    try {
      normalize_op(op);  // the reference would hit unreachable!() on an invalid action here; the oracle errors
    } catch (DecodeError& e) {
      throw inflate_err(S_INVALID_OP, e.what());
    }
    check_actor(row.id);
    op.id = row.id;
    for (size_t i = 1; i < row.ids.size(); i++)
      if (doc.lamport_cmp(row.ids[i - 1], row.ids[i]) >= 0) throw inflate_err(S_SUCC_OUT_OF_ORDER, "succ out of order");
    op.succ = row.ids;
    op.insert = row.insert;

    if (op.is_make()) create_ops[op.id] = CreateOp{op.action, obj};
    if (obj == cur.id) {
      append_op(cur, op);
    } else {
      auto cit = create_ops.find(obj);
      if (cit == create_ops.end()) throw inflate_err(S_OPS_OUT_OF_ORDER, "op for object without create op");
      if (obj < cur.id) throw inflate_err(S_OPS_OUT_OF_ORDER, "object ids out of order");
      finish_obj(cur);
      objs_loaded.insert(cur.id);
      cur = Loading();
      cur.id = obj;
      cur.parent = cit->second.parent;
      cur.objtype = cit->second.objtype;
      append_op(cur, op);
    }
  }
  finish_obj(cur);
  objs_loaded.insert(cur.id);
  for (auto& kv : create_ops) {
    if (!objs_loaded.count(kv.first)) {
      ObjTree t;
      t.objtype = kv.second.objtype;
      t.parent = kv.second.parent;
      doc.trees[kv.first] = std::move(t);
    }
  }

  // ---- ChangeCollector::finish (:99-136) + PartialChange::finish (:158-221)
  std::vector<Partial*> in_order;
  for (auto& kv : by_actor) {
    bool first = true;
    uint64_t prev = 0;
    for (auto& pc : kv.second) {
      if (first) {
        if (pc.meta.seq != 1) throw inflate_err(S_CHANGES_OUT_OF_ORDER, "first seq not 1");
      } else if (prev != pc.meta.seq - 1) {
        throw inflate_err(S_CHANGES_OUT_OF_ORDER, "seq gap");
      }
      first = false;
      prev = pc.meta.seq;
      in_order.push_back(&pc);
    }
  }
  std::sort(in_order.begin(), in_order.end(), [](Partial* a, Partial* b) { return a->index < b->index; });
  std::vector<Hash> hashes;
  for (size_t index = 0; index < in_order.size(); index++) {
    Partial& pc = *in_order[index];
    std::vector<Hash> deps;
    for (uint64_t d : pc.meta.deps) {
      if (d >= hashes.size()) throw inflate_err(S_MISSING_CHANGE, "could not find dependency");
      deps.push_back(hashes[d]);
    }
    uint64_t num_ops = pc.ops.size();
    std::stable_sort(pc.ops.begin(), pc.ops.end(),
                     [](const std::pair<OpId, Op>& a, const std::pair<OpId, Op>& b) { return a.second.id < b.second.id; });
    if (pc.meta.actor >= n_actors) throw inflate_err(S_COLLECT_MISSING_ACTOR, "actor out of bounds");
    if (num_ops > pc.meta.max_op) throw inflate_err(S_INCORRECT_MAX_OP, "incorrect max op");
    uint64_t start_op = pc.meta.max_op - num_ops + 1;
    if (start_op == 0) throw inflate_err(S_COLLECT_MISSING_OPS, "missing ops");
    std::vector<std::pair<OpId, const Op*>> refs;
    for (auto& po : pc.ops) refs.push_back({po.first, &po.second});
    StoredChange ch = build_change_from_ops(doc, (uint32_t)pc.meta.actor, deps, pc.meta.seq, start_op, pc.meta.timestamp,
                                            pc.meta.message.value_or(std::string()), pc.meta.extra, refs);
    hashes.push_back(ch.hash);
    for (auto& d : ch.deps) rec.heads.erase(d);
    rec.heads.insert(ch.hash);
    rec.changes.push_back(std::move(ch));
  }
  if (check_heads) {
    std::set<Hash> expected(pd.heads.begin(), pd.heads.end());
    if (expected != rec.heads) throw inflate_err(S_MISMATCHING_HEADS, "mismatching heads");
  }
  rec.max_op = max_op;
  return rec;
}

// ---------------------------------------------------------------------------------------------
// chunk stream parsing — storage/chunk.rs:50-100, storage/load.rs:56-114
// ---------------------------------------------------------------------------------------------
struct ParsedChunk {
  uint8_t type;
  size_t total_len;
  bool checksum_valid;
  ParsedDocument doc;        // type 0
  StoredChange change;       // type 1 / 2 (unverified: len not yet counted)
};
inline ParsedChunk parse_chunk(const uint8_t* p, size_t n) {
  ParsedChunk pc;
  try {
    ChunkHeader h = parse_chunk_header(p, n);
    pc.type = h.type;
    pc.total_len = h.header_size + h.data_len;
    if (h.type == CHUNK_CHANGE) {
      pc.change = parse_change_body(p, h);
      pc.checksum_valid = h.checksum_valid();
    } else if (h.type == CHUNK_DOC) {
      pc.doc = parse_document_body(p, h);
      pc.checksum_valid = h.checksum_valid();
    } else {
      Bytes inflated;
      if (!inflate_raw(p + h.header_size, h.data_len, inflated)) throw ParseFail("unable to decompress compressed chunk");
      Bytes inner;
      write_chunk(inner, CHUNK_CHANGE, h.checksum, inflated);
      ChunkHeader ih = parse_chunk_header(inner.data(), inner.size());
      pc.change = parse_change_body(inner.data(), ih);
      pc.change.compressed.assign(p, p + pc.total_len);
      pc.checksum_valid = ih.checksum_valid();  // chunk.rs:106-108
    }
  } catch (ParseFail& e) {
    throw Error(E_PARSE, e.what());
  }
  return pc;
}

// single change from bytes — change.rs:161-180
inline StoredChange change_from_bytes(const uint8_t* p, size_t n) {
  ParsedChunk pc = parse_chunk(p, n);
  if (pc.total_len != n) throw Error(E_LEFTOVER, "leftover data after parsing");
  if (pc.type == CHUNK_DOC) throw Error(E_WRONG_CHUNK_TYPE, "wrong chunk type");
  try {
    verify_change_ops(pc.change);
  } catch (DecodeError& e) {
    throw Error(E_PARSE, e.what());
  }
  return pc.change;
}

// load_changes: returns changes and, if a chunk failed, the error that stopped it (Partial)
inline std::vector<StoredChange> load_changes(const uint8_t* p, size_t n, std::optional<Error>& partial) {
  std::vector<StoredChange> out;
  size_t off = 0;
  while (off < n) {
    try {
      ParsedChunk pc = parse_chunk(p + off, n - off);
      if (!pc.checksum_valid) throw Error(E_BAD_CHECKSUM, "bad checksum");
      if (pc.type == CHUNK_DOC) {
        Reconstructed r = reconstruct_document(pc.doc, false, nullptr);
        for (auto& c : r.changes) out.push_back(std::move(c));
      } else {
        try {
          verify_change_ops(pc.change);
        } catch (DecodeError& e) {
          throw Error(E_INVALID_CHANGE_COLUMNS, e.what());
        }
        out.push_back(std::move(pc.change));
      }
      off += pc.total_len;
    } catch (Error& e) {
      partial = e;
      return out;
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// apply_changes — automerge.rs:771-914, 1112-1140, 1258-1280; op placement: op_tree.rs:212-239,
// 349-362, 393-408; query/opid.rs:136-184; op_set.rs:194-253; types.rs:659-684; opids.rs:51-74
// ---------------------------------------------------------------------------------------------
inline bool same_key(const Op& a, const Op& b) {  // elemid_or_key equality (types.rs:792-798)
  bool ap = a.key_is_prop && !a.insert, bp = b.key_is_prop && !b.insert;
  if (ap != bp) return false;
  if (ap) return a.prop == b.prop;
  OpId ae = a.insert ? a.id : a.elem, be = b.insert ? b.id : b.elem;
  return ae == be;
}

inline void insert_op(Doc& doc, const OpId& obj, Op op) {
  auto tit = doc.trees.find(obj);
  size_t pos = 0;
  std::vector<size_t> succ_pos;
  if (tit != doc.trees.end()) {
    std::vector<Op>& ops = tit->second.ops;
    if (!op.key_is_prop) {
      // SimpleOpIdSearch (query/opid.rs:144-184)
      bool found = op.elem.ctr == 0;
      size_t i = 0;
      for (; i < ops.size(); i++) {
        const Op& e = ops[i];
        if (!found) {
          if (e.id == op.elem) {
            found = true;
            if (!op.insert) break;
          }
        } else if (e.insert && doc.lamport_cmp(e.id, op.id) < 0) {
          break;
        }
      }
      pos = i;
    } else {
      // binary_search_by key_cmp (op_tree.rs:393-408; op_set.rs:444-449: String::cmp = bytewise)
      size_t left = 0, right = ops.size();
      while (left < right) {
        size_t mid = (left + right) / 2;
        const Op& e = ops[mid];
        if (!e.key_is_prop) throw Error(E_INVALID_ARG, "map op applied to a sequence object (reference panics)");
        if (e.prop < op.prop) left = mid + 1;
        else right = mid;
      }
      pos = left;
    }
    // found_op_without_patch_log (op_tree.rs:212-239)
    while (pos < ops.size()) {
      const Op& e = ops[pos];
      if (!same_key(e, op)) break;
      if (doc.lamport_cmp(e.id, op.id) > 0) break;
      bool overwrites = false;
      for (auto& p : op.pred) if (p == e.id) { overwrites = true; break; }
      if (overwrites) succ_pos.push_back(pos);
      pos++;
    }
    // add_succ (op_set.rs:194-203; types.rs:659-664; opids.rs:51-74)
    for (size_t sp : succ_pos) {
      Op& old = ops[sp];
      size_t k = 0;
      bool dup = false;
      for (; k < old.succ.size(); k++) {
        int c = doc.lamport_cmp(old.succ[k], op.id);
        if (c == 0) { dup = true; break; }
        if (c > 0) break;
      }
      if (!dup) old.succ.insert(old.succ.begin() + k, op.id);
      if (op.is_inc() && old.is_counter()) {
        old.counter_current += op.val.i;
        old.incs.push_back({op.id, op.val.i});
      }
    }
  }
  if (op.is_delete()) return;
  // OpSet::insert (op_set.rs:233-253)
  if (op.is_make()) {
    ObjTree t;
    t.objtype = op.action;
    t.parent = obj;
    doc.trees[op.id] = std::move(t);
  }
  tit = doc.trees.find(obj);
  if (tit != doc.trees.end()) tit->second.ops.insert(tit->second.ops.begin() + pos, std::move(op));
}

inline void apply_change(Doc& doc, const StoredChange& ch) {  // automerge.rs:829-914, 1112-1133
  uint32_t actor = doc.cache_actor(ch.actor);
  std::vector<uint32_t> actors{actor};
  for (auto& a : ch.other_actors) actors.push_back(doc.cache_actor(a));
  std::vector<StoredOp> rows = ch.ops();
  std::vector<std::pair<OpId, Op>> ops;
  auto tr = [&](const OpId& o) {
    if (o.actor >= actors.size()) throw Error(E_INVALID_ARG, "op references an actor outside the change's table (reference panics)");
    return OpId{o.ctr, actors[o.actor]};
  };
  for (size_t i = 0; i < rows.size(); i++) {
    StoredOp& r = rows[i];
    Op op;
    uint64_t c = ch.start_op + i;
    if (c > 0xffffffffull) throw Error(E_INVALID_ARG, "op counter exceeds u32 (reference panics)");
    op.id = OpId{(uint32_t)c, actor};
    op.key_is_prop = r.key_is_prop;
    op.prop = r.prop;
    if (!r.key_is_prop) op.elem = r.elem.ctr == 0 ? OpId{0, 0} : tr(r.elem);
    OpId obj = r.obj.ctr == 0 ? OpId{0, 0} : tr(r.obj);
    for (auto& p : r.ids) op.pred.push_back(tr(p));
    std::stable_sort(op.pred.begin(), op.pred.end(), [&](const OpId& a, const OpId& b) { return doc.lamport_cmp(a, b) < 0; });
    op.action = r.action;
    op.val = r.val;
    op.expand = r.expand;
    op.mark_name = r.mark_name;
    op.insert = r.insert;
    normalize_op(op);
    ops.push_back({obj, std::move(op)});
  }
  // update_history
  doc.max_op = std::max<uint64_t>(doc.max_op, ch.start_op + ops.size() - 1);
  for (auto& d : ch.deps) doc.deps.erase(d);
  doc.deps.insert(ch.hash);
  size_t hidx = doc.history.size();
  doc.states[actor].push_back(hidx);
  doc.history_index[ch.hash] = hidx;
  doc.history.push_back(ch);
  for (auto& po : ops) insert_op(doc, po.first, std::move(po.second));
}

inline bool causally_ready(const Doc& doc, const StoredChange& ch) {
  for (auto& d : ch.deps) if (!doc.history_index.count(d)) return false;
  return true;
}

inline void apply_changes(Doc& doc, const std::vector<StoredChange>& changes) {  // automerge.rs:797-858
  for (auto& c : changes) {
    if (doc.history_index.count(c.hash)) continue;
    auto ait = doc.actor_lookup.find(c.actor);
    if (ait != doc.actor_lookup.end()) {  // duplicate_seq :771-779
      auto sit = doc.states.find(ait->second);
      if (sit != doc.states.end() && sit->second.size() >= c.seq) throw Error(E_DUPLICATE_SEQ, "duplicate seq number");
    }
    if (causally_ready(doc, c)) apply_change(doc, c);
    else doc.queue.push_back(c);
  }
  for (;;) {  // pop_next_causally_ready_change: linear scan + swap_remove
    size_t idx = 0;
    bool got = false;
    StoredChange c;
    while (idx < doc.queue.size()) {
      if (causally_ready(doc, doc.queue[idx])) {
        c = doc.queue[idx];
        doc.queue[idx] = doc.queue.back();
        doc.queue.pop_back();
        got = true;
        break;
      }
      idx++;
    }
    if (!got) break;
    if (!doc.history_index.count(c.hash)) apply_change(doc, c);
  }
}

// Automerge::load_with_options — automerge.rs:601-705 (default options: OnPartialLoad::Error,
// VerificationMode::Check unless told otherwise)
inline Doc load(const uint8_t* p, size_t n, bool check_heads = true, bool error_on_partial = true) {
  Doc doc;
  if (n == 0) return doc;
  ParsedChunk first = parse_chunk(p, n);
  if (!first.checksum_valid) throw Error(E_BAD_CHECKSUM, "bad checksum");
  std::vector<StoredChange> pending;
  bool first_was_doc = false;
  if (first.type == CHUNK_DOC) {
    first_was_doc = true;
    Reconstructed r = reconstruct_document(first.doc, check_heads, &doc);
    for (size_t i = 0; i < r.changes.size(); i++) {
      auto lit = doc.actor_lookup.find(r.changes[i].actor);
      uint32_t ai = lit->second;
      doc.states[ai].push_back(i);
      doc.history_index[r.changes[i].hash] = i;
    }
    doc.history = std::move(r.changes);
    doc.deps = r.heads;
    doc.max_op = r.max_op;
  } else {
    try {
      verify_change_ops(first.change);
    } catch (DecodeError& e) {
      throw Error(E_INVALID_CHANGE_COLUMNS, e.what());
    }
    pending.push_back(std::move(first.change));
  }
  std::optional<Error> partial;
  std::vector<StoredChange> rest = load_changes(p + first.total_len, n - first.total_len, partial);
  if (!partial) {
    for (auto& c : rest) pending.push_back(std::move(c));
    apply_changes(doc, pending);
    if (!doc.queue.empty() && !first_was_doc && error_on_partial) throw Error(E_MISSING_DEPS, "missing dependencies");
  } else if (error_on_partial) {
    throw *partial;
  }
  // (with OnPartialLoad::Ignore and a partial tail nothing after the first chunk is applied: :690-694)
  return doc;
}

// load_incremental — automerge.rs:730-769.  Returns number of ops applied (op-store delta).
inline size_t opset_len(const Doc& d) {
  size_t n = 0;
  for (auto& t : d.trees) n += t.second.ops.size();
  return n;
}
inline size_t load_incremental(Doc& doc, const uint8_t* p, size_t n) {
  if (doc.is_empty()) {
    doc = load(p, n, true, false);
    return opset_len(doc);
  }
  std::optional<Error> partial;
  std::vector<StoredChange> cs = load_changes(p, n, partial);
  size_t before = opset_len(doc);
  apply_changes(doc, cs);
  return opset_len(doc) - before;
}

// get_changes(have_deps) — automerge.rs:1306, 1022-1047 (get_changes_clock), change_graph.rs:128-182:
// the clock of the given heads = per actor the highest seq among their ancestors (heads the document does not have
// are skipped); result = every change whose seq is above its actor's clock entry, in history order.
inline std::vector<size_t> get_changes_since(const Doc& doc, const std::vector<Hash>& have_deps) {
  std::map<Bytes, uint64_t> clock;
  std::vector<size_t> to_visit;
  for (auto& h : have_deps) {
    auto it = doc.history_index.find(h);
    if (it != doc.history_index.end()) to_visit.push_back(it->second);
  }
  std::set<size_t> visited;
  while (!to_visit.empty()) {
    size_t i = to_visit.back();
    to_visit.pop_back();
    if (!visited.insert(i).second) continue;
    const StoredChange& c = doc.history[i];
    uint64_t& s = clock[c.actor];
    if (c.seq > s) s = c.seq;
    for (auto& d : c.deps) {
      auto it = doc.history_index.find(d);
      if (it != doc.history_index.end()) to_visit.push_back(it->second);
    }
  }
  std::vector<size_t> out;
  for (size_t i = 0; i < doc.history.size(); i++) {
    auto it = clock.find(doc.history[i].actor);
    if (it == clock.end() || doc.history[i].seq > it->second) out.push_back(i);
  }
  return out;
}
// save_after — automerge.rs:993-1000: the raw bytes of those changes, one after the other
inline Bytes save_after(const Doc& doc, const std::vector<Hash>& heads) {
  Bytes out;
  for (size_t i : get_changes_since(doc, heads)) out.insert(out.end(), doc.history[i].bytes.begin(), doc.history[i].bytes.end());
  return out;
}

// merge — automerge.rs:917-940, get_changes_added :1311-1334
inline std::vector<Hash> merge(Doc& dst, const Doc& src) {
  std::vector<Hash> stack = src.get_heads();
  std::set<Hash> seen;
  std::vector<Hash> added;
  while (!stack.empty()) {
    Hash h = stack.back();
    stack.pop_back();
    if (!seen.count(h) && !dst.history_index.count(h)) {
      seen.insert(h);
      added.push_back(h);
      auto it = src.history_index.find(h);
      if (it != src.history_index.end())
        for (auto& d : src.history[it->second].deps) stack.push_back(d);
    }
  }
  std::reverse(added.begin(), added.end());
  std::vector<StoredChange> cs;
  for (auto& h : added) {
    auto it = src.history_index.find(h);
    if (it != src.history_index.end()) cs.push_back(src.history[it->second]);
  }
  apply_changes(dst, cs);
  return dst.get_heads();
}

// ---------------------------------------------------------------------------------------------
// save — automerge.rs:943-965; storage/save/document.rs:21-148; op_set.rs:67-75;
// convert/op_as_docop.rs; indexed_cache.rs:72-107
// ---------------------------------------------------------------------------------------------
inline Bytes save(const Doc& doc, bool deflate, bool retain_orphans = true) {
  // encode_index: application actor index -> index in the bytewise-sorted table
  std::vector<Bytes> sorted = doc.actors;
  std::sort(sorted.begin(), sorted.end());
  std::vector<uint32_t> lookup(doc.actors.size());
  for (size_t i = 0; i < doc.actors.size(); i++)
    lookup[i] = (uint32_t)(std::find(sorted.begin(), sorted.end(), doc.actors[i]) - sorted.begin());
  auto tr = [&](const OpId& o) { return OpId{o.ctr, lookup[o.actor]}; };
  // objects in lamport order (root first: ctr 0)
  std::vector<OpId> objs;
  for (auto& t : doc.trees) objs.push_back(t.first);
  std::stable_sort(objs.begin(), objs.end(), [&](const OpId& a, const OpId& b) {
    if (a.ctr == 0 || b.ctr == 0) return a.ctr < b.ctr;  // root's actor slot 0 may not exist
    return doc.lamport_cmp(a, b) < 0;
  });
  std::vector<StoredOp> rows;
  for (auto& obj : objs) {
    for (auto& op : doc.trees.at(obj).ops) {
      StoredOp r;
      r.id = tr(op.id);
      r.obj = obj.ctr == 0 ? OpId{0, 0} : tr(obj);
      r.key_is_prop = op.key_is_prop;
      r.prop = op.prop;
      r.elem = (op.key_is_prop || op.elem.ctr == 0) ? OpId{0, 0} : tr(op.elem);
      r.insert = op.insert;
      r.action = op.action;
      r.val = op.val;
      for (auto& s : op.succ) r.ids.push_back(tr(s));
      r.expand = op.is_mark() ? op.expand : false;
      r.mark_name = op.mark_name;
      rows.push_back(std::move(r));
    }
  }
  std::map<Hash, uint64_t> index_by_hash;
  for (size_t i = 0; i < doc.history.size(); i++) index_by_hash[doc.history[i].hash] = i;
  std::vector<ChangeMeta> metas;
  for (auto& c : doc.history) {
    ChangeMeta m;
    m.actor = lookup[doc.actor_lookup.at(c.actor)];
    m.seq = c.seq;
    m.max_op = c.max_op();
    m.timestamp = c.timestamp;
    if (!c.message.empty()) m.message = c.message;
    for (auto& d : c.deps) m.deps.push_back(index_by_hash.at(d));
    m.extra = c.extra();
    metas.push_back(std::move(m));
  }
  std::vector<std::pair<Hash, uint64_t>> heads;
  for (auto& h : doc.get_heads()) heads.push_back({h, index_by_hash.at(h)});
  Bytes out = build_document(doc.actors, heads, rows, metas, deflate);
  if (retain_orphans)
    for (auto& q : doc.queue) out.insert(out.end(), q.bytes.begin(), q.bytes.end());
  return out;
}

// ---------------------------------------------------------------------------------------------
// Bloom summary — sync/bloom.rs:9-10, 38-47, 69-102, 125-146; sync.rs:243-250
// ---------------------------------------------------------------------------------------------
inline Bytes bloom_from_hashes(const std::vector<Hash>& hashes) {
  uint32_t n = (uint32_t)hashes.size();
  Bytes out;
  if (n == 0) return out;
  size_t nbytes = (size_t)std::ceil(((double)n * 10.0) / 8.0);
  Bytes bits(nbytes, 0);
  uint32_t modulo = 8 * (uint32_t)nbytes;
  for (auto& h : hashes) {
    auto le = [&](int o) { return (uint32_t)h.b[o] | ((uint32_t)h.b[o + 1] << 8) | ((uint32_t)h.b[o + 2] << 16) | ((uint32_t)h.b[o + 3] << 24); };
    uint32_t x = le(0) % modulo, y = le(4) % modulo, z = le(8) % modulo;
    for (int k = 0; k < 7; k++) {
      if (k) { x = (x + y) % modulo; y = (y + z) % modulo; }
      bits[x >> 3] |= (uint8_t)(1u << (x & 7));
    }
  }
  put_uleb(out, n);
  put_uleb(out, 10);
  put_uleb(out, 7);
  out.insert(out.end(), bits.begin(), bits.end());
  return out;
}

// ---------------------------------------------------------------------------------------------
// canonical materialization (SURVEY §8a "materialized document state"): per object, per key /
// element, the visible ops in ascending lamport order (= get_all, automerge.rs:1528-1565; the last
// entry is the winner = get / hydrate, iter/top_ops.rs:43-102, hydrate.rs:128-167); maps in
// bytewise key order, sequences in document (RGA) order; text = winners' to_str (op_set.rs:307-311,
// types.rs:278-286).
// ---------------------------------------------------------------------------------------------
inline std::string hex(const uint8_t* p, size_t n) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
  return s;
}
inline void json_str(std::string& out, const uint8_t* p, size_t n) {
  out.push_back('"');
  for (size_t i = 0; i < n; i++) {
    uint8_t c = p[i];
    if (c == '"') out += "\\\"";
    else if (c == '\\') out += "\\\\";
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
    else out.push_back((char)c);
  }
  out.push_back('"');
}
inline void hydrate_obj(const Doc& doc, const OpId& obj, uint64_t objtype, std::string& out, int depth);
inline void hydrate_value(const Doc& doc, const Op& op, std::string& out, int depth) {
  out += "{\"id\":\"" + std::to_string(op.id.ctr) + "@" + hex(doc.actors[op.id.actor].data(), doc.actors[op.id.actor].size()) + "\",";
  if (op.is_make()) {
    const char* names[] = {"map", "", "list", "", "text", "", "table"};
    out += std::string("\"") + names[op.action] + "\":";
    hydrate_obj(doc, op.id, op.action, out, depth + 1);
  } else {
    const Scalar& v = op.val;
    switch (v.tag) {
      case 0: out += "\"null\":null"; break;
      case 1: out += "\"bool\":false"; break;
      case 2: out += "\"bool\":true"; break;
      case 3: out += "\"uint\":" + std::to_string(v.u); break;
      case 4: out += "\"int\":" + std::to_string(v.i); break;
      case 5: out += "\"f64\":\"" + hex(v.bytes.data(), v.bytes.size()) + "\""; break;
      case 6: out += "\"str\":"; json_str(out, v.bytes.data(), v.bytes.size()); break;
      case 7: out += "\"bytes\":\"" + hex(v.bytes.data(), v.bytes.size()) + "\""; break;
      case 8: out += "\"counter\":" + std::to_string(op.counter_current); break;
      case 9: out += "\"ts\":" + std::to_string(v.i); break;
      default: out += "\"unknown\":[" + std::to_string(v.tag) + ",\"" + hex(v.bytes.data(), v.bytes.size()) + "\"]"; break;
    }
  }
  out += "}";
}
inline void hydrate_obj(const Doc& doc, const OpId& obj, uint64_t objtype, std::string& out, int depth) {
  if (depth > 10000) throw Error(E_INVALID_ARG, "object nesting too deep");   // this recursive restatement guards its own stack; hydrate.rs:128-167 has no limit
  auto tit = doc.trees.find(obj);
  static const std::vector<Op> none;
  const std::vector<Op>& ops = tit == doc.trees.end() ? none : tit->second.ops;
  bool is_map = objtype == 0 || objtype == 6;
  bool is_text = objtype == 4;
  if (is_text) {
    std::string text;
    size_t i = 0;
    while (i < ops.size()) {
      size_t j = i;
      const Op* last = nullptr;
      while (j < ops.size() && same_key(ops[j], ops[i])) { if (ops[j].visible()) last = &ops[j]; j++; }
      if (last) {
        if (last->action == 1 && last->val.tag == 6) text.append((const char*)last->val.bytes.data(), last->val.bytes.size());
        else text += "\xef\xbf\xbc";
      }
      i = j;
    }
    json_str(out, (const uint8_t*)text.data(), text.size());
    return;
  }
  out += is_map ? "{" : "[";
  bool first = true;
  size_t i = 0;
  while (i < ops.size()) {
    size_t j = i;
    std::vector<const Op*> vis;
    while (j < ops.size() && same_key(ops[j], ops[i])) { if (ops[j].visible()) vis.push_back(&ops[j]); j++; }
    if (!vis.empty()) {
      if (!first) out += ",";
      first = false;
      if (is_map) { json_str(out, (const uint8_t*)ops[i].prop.data(), ops[i].prop.size()); out += ":"; }
      out += "[";
      for (size_t k = 0; k < vis.size(); k++) { if (k) out += ","; hydrate_value(doc, *vis[k], out, depth); }
      out += "]";
    }
    i = j;
  }
  out += is_map ? "}" : "]";
}
inline std::string hydrate_json(const Doc& doc) {
  std::string out;
  hydrate_obj(doc, OpId{0, 0}, 0, out, 0);
  return out;
}

}  // namespace amo
