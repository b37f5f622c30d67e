// ORACLE — TEST INFRASTRUCTURE ONLY (see amo_codec.hpp header).
// The sync protocol's sender side, restated: Automerge::generate_sync_message (sync.rs:134-221),
// get_missing_deps (automerge.rs:1776-1799), make_bloom_filter (sync.rs:243-250), get_changes_to_send
// (sync.rs:252-321), BloomFilter::contains_hash (sync/bloom.rs:69-123), Message::encode / parse
// (sync.rs:482-543), sync::State (sync/state.rs:30-53).  The receiver side (receive_sync_message_inner,
// sync.rs:323-383; advance_heads :565-589; filter_changes, automerge.rs:1005-1019) is restated as well, because
// the tests need it to drive a conversation between two peers; the product has no counterpart of it yet.
#pragma once
#include <algorithm>
#include <optional>
#include <set>
#include <vector>

#include "amo_engine.hpp"

namespace amo {

struct Bloom {  // sync/bloom.rs:13-29
  uint32_t num_entries = 0, bits_per_entry = 10, num_probes = 7;
  Bytes bits;
  Bytes to_bytes() const {  // bloom.rs:38-47
    Bytes out;
    if (num_entries != 0) {
      put_uleb(out, num_entries);
      put_uleb(out, bits_per_entry);
      put_uleb(out, num_probes);
      out.insert(out.end(), bits.begin(), bits.end());
    }
    return out;
  }
  static Bloom parse(const uint8_t* p, size_t n) {  // bloom.rs:49-67
    Bloom b;
    if (n == 0) return b;
    Cursor c(p, n);
    auto u32of = [&]() {
      uint64_t v = read_uleb_strict(c);
      if (v > 0xFFFFFFFFull) throw Error(E_PARSE, "leb128 too large for u32");
      return (uint32_t)v;
    };
    b.num_entries = u32of();
    b.bits_per_entry = u32of();
    b.num_probes = u32of();
    size_t cap = (size_t)std::ceil(((double)b.num_entries * (double)b.bits_per_entry) / 8.0);
    const uint8_t* q = c.take_n(cap);
    b.bits.assign(q, q + cap);
    return b;
  }
  bool contains(const Hash& h) const {  // bloom.rs:69-90, 110-123
    if (num_entries == 0) return false;
    uint32_t modulo = 8 * (uint32_t)bits.size();
    if (modulo == 0) throw Error(E_INVALID_ARG, "bloom filter without bits (the reference divides by zero)");
    auto le = [&](int o) { return (uint32_t)h.b[o] | ((uint32_t)h.b[o + 1] << 8) | ((uint32_t)h.b[o + 2] << 16) | ((uint32_t)h.b[o + 3] << 24); };
    uint32_t x = le(0) % modulo, y = le(4) % modulo, z = le(8) % modulo;
    for (uint32_t k = 0; k < (num_probes ? num_probes : 1); k++) {   // probes.push(x) happens before the 1..num_probes loop
      if (k) { x = (uint32_t)(((uint64_t)x + y) % modulo); y = (uint32_t)(((uint64_t)y + z) % modulo); }
      if (!(bits[x >> 3] & (1u << (x & 7)))) return false;
    }
    return true;
  }
};

struct Have {  // sync/state.rs:59-65
  std::vector<Hash> last_sync;
  Bloom bloom;
};

struct SyncState {  // sync/state.rs:31-53
  std::vector<Hash> shared_heads, last_sent_heads;
  std::optional<std::vector<Hash>> their_heads, their_need;
  std::optional<std::vector<Have>> their_have;
  std::set<Hash> sent_hashes;
  bool in_flight = false;
};

struct SyncMessage {  // sync.rs:433-442
  std::vector<Hash> heads, need;
  std::vector<Have> have;
  std::vector<Bytes> changes;  // raw chunk bytes
};

inline void encode_hashes(Bytes& out, const std::vector<Hash>& hs) {  // sync.rs:557-563
  put_uleb(out, hs.size());
  for (auto& h : hs) out.insert(out.end(), h.b.begin(), h.b.end());
}
inline Bytes encode_message(const SyncMessage& m) {  // sync.rs:526-543
  Bytes out{0x42};
  encode_hashes(out, m.heads);
  encode_hashes(out, m.need);
  put_uleb(out, m.have.size());
  for (auto& h : m.have) {
    encode_hashes(out, h.last_sync);
    Bytes b = h.bloom.to_bytes();
    put_uleb(out, b.size());
    out.insert(out.end(), b.begin(), b.end());
  }
  put_uleb(out, m.changes.size());
  for (auto& c : m.changes) {
    put_uleb(out, c.size());
    out.insert(out.end(), c.begin(), c.end());
  }
  return out;
}
inline std::vector<Hash> parse_hashes(Cursor& c) {
  uint64_t n = read_uleb_strict(c);
  std::vector<Hash> out;
  for (uint64_t i = 0; i < n; i++) {
    Hash h;
    memcpy(h.b.data(), c.take_n(32), 32);
    out.push_back(h);
  }
  return out;
}
inline SyncMessage parse_message(const uint8_t* p, size_t n) {  // sync.rs:482-524
  try {
    Cursor c(p, n);
    uint8_t t = c.take1();
    if (t != 0x42) throw Error(E_PARSE, "wrong message type");
    SyncMessage m;
    m.heads = parse_hashes(c);
    m.need = parse_hashes(c);
    uint64_t nh = read_uleb_strict(c);
    for (uint64_t i = 0; i < nh; i++) {
      Have h;
      h.last_sync = parse_hashes(c);
      uint64_t bl = read_uleb_strict(c);
      const uint8_t* q = c.take_n(bl);
      h.bloom = Bloom::parse(q, bl);
      m.have.push_back(std::move(h));
    }
    uint64_t nc = read_uleb_strict(c);
    for (uint64_t i = 0; i < nc; i++) {
      uint64_t l = read_uleb_strict(c);
      const uint8_t* q = c.take_n(l);
      m.changes.emplace_back(q, q + l);
    }
    return m;
  } catch (DecodeError& e) {
    throw Error(E_PARSE, e.what());
  } catch (StrictLebError& e) {
    throw Error(E_PARSE, e.what());
  }
}

inline std::vector<Hash> get_missing_deps(const Doc& doc, const std::vector<Hash>& heads) {  // automerge.rs:1776-1799
  std::set<Hash> in_queue, missing;
  for (auto& q : doc.queue) in_queue.insert(q.hash);
  for (auto& q : doc.queue)
    for (auto& d : q.deps)
      if (!doc.history_index.count(d)) missing.insert(d);
  for (auto& h : heads)
    if (!doc.history_index.count(h)) missing.insert(h);
  std::vector<Hash> out;
  for (auto& h : missing)
    if (!in_queue.count(h)) out.push_back(h);
  return out;  // std::set iterates in ascending order = missing.sort()
}

inline Have make_bloom_filter(const Doc& doc, const std::vector<Hash>& last_sync) {  // sync.rs:243-250
  Have h;
  h.last_sync = last_sync;
  std::vector<Hash> hs;
  for (size_t i : get_changes_since(doc, last_sync)) hs.push_back(doc.history[i].hash);
  Bytes wire = bloom_from_hashes(hs);
  h.bloom = Bloom::parse(wire.data(), wire.size());
  return h;
}

inline std::vector<size_t> get_changes_to_send(const Doc& doc, const std::vector<Have>& have, const std::vector<Hash>& need) {  // sync.rs:252-321
  std::vector<size_t> out;
  if (have.empty()) {
    for (auto& h : need) {
      auto it = doc.history_index.find(h);
      if (it != doc.history_index.end()) out.push_back(it->second);
    }
    return out;
  }
  std::set<Hash> ls;
  for (auto& h : have) ls.insert(h.last_sync.begin(), h.last_sync.end());
  std::vector<size_t> changes = get_changes_since(doc, std::vector<Hash>(ls.begin(), ls.end()));
  std::map<Hash, std::vector<Hash>> dependents;
  std::set<Hash> to_send;
  for (size_t i : changes) {
    const StoredChange& c = doc.history[i];
    for (auto& d : c.deps) dependents[d].push_back(c.hash);
    bool in_any = false;
    for (auto& h : have) in_any = in_any || h.bloom.contains(c.hash);
    if (!in_any) to_send.insert(c.hash);
  }
  std::vector<Hash> stack(to_send.begin(), to_send.end());
  while (!stack.empty()) {
    Hash h = stack.back();
    stack.pop_back();
    auto it = dependents.find(h);
    if (it == dependents.end()) continue;
    for (auto& d : it->second)
      if (to_send.insert(d).second) stack.push_back(d);
  }
  for (auto& h : need)
    if (!to_send.count(h)) {
      auto it = doc.history_index.find(h);
      if (it != doc.history_index.end()) out.push_back(it->second);
    }
  for (size_t i : changes)
    if (to_send.count(doc.history[i].hash)) out.push_back(i);
  return out;
}

// generate_sync_message — sync.rs:134-221.  Returns false for None.
inline bool generate_sync_message(const Doc& doc, SyncState& st, SyncMessage& msg) {
  std::vector<Hash> our_heads = doc.get_heads();
  std::vector<Hash> our_need = get_missing_deps(doc, st.their_heads ? *st.their_heads : std::vector<Hash>());
  std::set<Hash> their_heads_set;
  if (st.their_heads) their_heads_set.insert(st.their_heads->begin(), st.their_heads->end());
  std::vector<Have> our_have;
  bool all_in = true;
  for (auto& h : our_need) all_in = all_in && their_heads_set.count(h);
  if (all_in) our_have.push_back(make_bloom_filter(doc, st.shared_heads));
  if (st.their_have && !st.their_have->empty()) {
    bool known = true;
    for (auto& h : st.their_have->front().last_sync) known = known && doc.history_index.count(h);
    if (!known) {
      msg = SyncMessage();
      msg.heads = our_heads;
      msg.have.push_back(Have());
      return true;
    }
  }
  std::vector<size_t> to_send;
  if (st.their_have && st.their_need) to_send = get_changes_to_send(doc, *st.their_have, *st.their_need);
  bool heads_unchanged = st.last_sent_heads == our_heads;
  bool heads_equal = st.their_heads && *st.their_heads == our_heads;
  std::vector<size_t> fresh;
  for (size_t i : to_send)
    if (!st.sent_hashes.count(doc.history[i].hash)) fresh.push_back(i);
  if (heads_unchanged) {
    if (heads_equal && fresh.empty()) return false;
    if (st.in_flight) return false;
  }
  st.last_sent_heads = our_heads;
  for (size_t i : fresh) st.sent_hashes.insert(doc.history[i].hash);
  msg = SyncMessage();
  msg.heads = our_heads;
  msg.have = our_have;
  msg.need = our_need;
  for (size_t i : fresh) msg.changes.push_back(doc.history[i].bytes);
  st.in_flight = true;
  return true;
}

// receive_sync_message_inner — sync.rs:323-383 (test driver only)
inline void receive_sync_message(Doc& doc, SyncState& st, const SyncMessage& m) {
  st.in_flight = false;
  std::vector<Hash> before = doc.get_heads();
  if (!m.changes.empty()) {
    std::vector<StoredChange> cs;
    for (auto& c : m.changes) cs.push_back(change_from_bytes(c.data(), c.size()));
    apply_changes(doc, cs);
    // advance_heads — sync.rs:565-589
    std::vector<Hash> now = doc.get_heads();
    std::set<Hash> old_set(before.begin(), before.end()), new_set(now.begin(), now.end()), adv;
    for (auto& h : now) if (!old_set.count(h)) adv.insert(h);
    for (auto& h : st.shared_heads) if (new_set.count(h)) adv.insert(h);
    st.shared_heads.assign(adv.begin(), adv.end());
  }
  {  // filter_changes — automerge.rs:1005-1019, change_graph.rs:144-182
    std::vector<size_t> to_visit;
    for (auto& h : m.heads) {
      auto it = doc.history_index.find(h);
      if (it != doc.history_index.end()) to_visit.push_back(it->second);
    }
    std::set<size_t> visited;
    while (!to_visit.empty()) {
      size_t i = to_visit.back();
      to_visit.pop_back();
      if (!visited.insert(i).second) continue;
      st.sent_hashes.erase(doc.history[i].hash);
      for (auto& d : doc.history[i].deps) to_visit.push_back(doc.history_index.at(d));
    }
  }
  if (m.changes.empty() && m.heads == before) st.last_sent_heads = m.heads;
  std::vector<Hash> known;
  for (auto& h : m.heads) if (doc.history_index.count(h)) known.push_back(h);
  if (known.size() == m.heads.size()) {
    st.shared_heads = m.heads;
    if (m.heads.empty()) {
      st.last_sent_heads.clear();
      st.sent_hashes.clear();
    }
  } else {
    std::set<Hash> u(st.shared_heads.begin(), st.shared_heads.end());
    u.insert(known.begin(), known.end());
    st.shared_heads.assign(u.begin(), u.end());
  }
  st.their_have = m.have;
  st.their_heads = m.heads;
  st.their_need = m.need;
}

// ---- flat form of a State for the C entry points (tests/sync_cases.py writes and reads the same layout) ----
// u8 flags (1 their_heads, 2 their_need, 4 their_have, 8 in_flight); u32 n + hashes for shared_heads, last_sent_heads,
// their_heads, their_need; u32 n_have x { u32 n + hashes (last_sync); u32 len + bloom wire bytes }; u32 n + hashes (sent)
inline void blob_u32(Bytes& b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
inline void blob_hashes(Bytes& b, const std::vector<Hash>& hs) {
  blob_u32(b, (uint32_t)hs.size());
  for (auto& h : hs) b.insert(b.end(), h.b.begin(), h.b.end());
}
inline Bytes state_to_blob(const SyncState& st) {
  Bytes b;
  b.push_back((uint8_t)((st.their_heads ? 1 : 0) | (st.their_need ? 2 : 0) | (st.their_have ? 4 : 0) | (st.in_flight ? 8 : 0)));
  blob_hashes(b, st.shared_heads);
  blob_hashes(b, st.last_sent_heads);
  blob_hashes(b, st.their_heads ? *st.their_heads : std::vector<Hash>());
  blob_hashes(b, st.their_need ? *st.their_need : std::vector<Hash>());
  blob_u32(b, st.their_have ? (uint32_t)st.their_have->size() : 0);
  if (st.their_have)
    for (auto& h : *st.their_have) {
      blob_hashes(b, h.last_sync);
      Bytes w = h.bloom.to_bytes();
      blob_u32(b, (uint32_t)w.size());
      b.insert(b.end(), w.begin(), w.end());
    }
  blob_hashes(b, std::vector<Hash>(st.sent_hashes.begin(), st.sent_hashes.end()));
  return b;
}
inline SyncState state_from_blob(const uint8_t* p, size_t n) {
  Cursor c(p, n);
  auto u32of = [&]() { const uint8_t* q = c.take_n(4); return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
  auto hashes = [&]() {
    uint32_t k = u32of();
    std::vector<Hash> out(k);
    for (uint32_t i = 0; i < k; i++) memcpy(out[i].b.data(), c.take_n(32), 32);
    return out;
  };
  SyncState st;
  uint8_t flags = c.take1();
  st.shared_heads = hashes();
  st.last_sent_heads = hashes();
  auto th = hashes();
  auto tn = hashes();
  if (flags & 1) st.their_heads = th;
  if (flags & 2) st.their_need = tn;
  uint32_t nh = u32of();
  std::vector<Have> hv;
  for (uint32_t i = 0; i < nh; i++) {
    Have h;
    h.last_sync = hashes();
    uint32_t bl = u32of();
    const uint8_t* q = c.take_n(bl);
    h.bloom = Bloom::parse(q, bl);
    hv.push_back(std::move(h));
  }
  if (flags & 4) st.their_have = hv;
  for (auto& h : hashes()) st.sent_hashes.insert(h);
  st.in_flight = (flags & 8) != 0;
  return st;
}

}  // namespace amo
