// ORACLE — TEST INFRASTRUCTURE ONLY (see amo_codec.hpp header).
// Plain C entry points over the CPU restatement so that tests/ (ctypes) and bench.py's cpu_baseline
// leg can drive it.  Nothing in the product links or loads this library.
#include <chrono>
#include <thread>

#include "amo_engine.hpp"
#include "amo_sync.hpp"

using namespace amo;

struct amo_doc {
  Doc doc;
};

static thread_local std::string g_err;
static thread_local int g_kind = 0, g_sub = 0;
static thread_local Bytes g_out;

static void set_err(const Error& e) {
  g_err = e.what();
  g_kind = e.kind;
  g_sub = e.sub;
}
static void set_err_generic(const std::exception& e) {
  g_err = e.what();
  g_kind = E_INVALID_ARG;
  g_sub = 0;
}
#define GUARD_BEGIN try {
#define GUARD_END(failret)                 \
  }                                        \
  catch (Error & e) {                      \
    set_err(e);                            \
    return failret;                        \
  }                                        \
  catch (std::exception & e) {             \
    set_err_generic(e);                    \
    return failret;                        \
  }

static int give(const Bytes& b, const uint8_t** out, size_t* len) {
  g_out = b;
  *out = g_out.data();
  *len = g_out.size();
  return 0;
}
static void put_u32(Bytes& b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
static void put_u64(Bytes& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }

extern "C" {

const char* amo_last_error() { return g_err.c_str(); }
int amo_last_kind() { return g_kind; }
int amo_last_sub() { return g_sub; }

amo_doc* amo_new() { return new amo_doc(); }
void amo_free(amo_doc* d) { delete d; }
amo_doc* amo_clone(const amo_doc* d) { return new amo_doc(*d); }

// Automerge::load_with_options(data, {verification_mode}) — automerge.rs:601
amo_doc* amo_load(const uint8_t* data, size_t len, int check_heads) {
  GUARD_BEGIN
  auto* d = new amo_doc();
  try {
    d->doc = load(data, len, check_heads != 0, true);
  } catch (...) {
    delete d;
    throw;
  }
  return d;
  GUARD_END(nullptr)
}

// load_with_options with OnPartialLoad::{Error, Ignore} — automerge.rs:58-135, 678-695
amo_doc* amo_load_opts(const uint8_t* data, size_t len, int check_heads, int ignore_partial) {
  GUARD_BEGIN
  auto* d = new amo_doc();
  try {
    d->doc = load(data, len, check_heads != 0, ignore_partial == 0);
  } catch (...) {
    delete d;
    throw;
  }
  return d;
  GUARD_END(nullptr)
}
// Change::compress (storage/change.rs:287-293) of one change chunk; bytes are zlib's: parity unpinned, used as test input only
int amo_change_compress(const uint8_t* data, size_t len, const uint8_t** out, size_t* olen) {
  GUARD_BEGIN
  StoredChange ch = change_from_bytes(data, len);
  return give(compress_change(ch), out, olen);
  GUARD_END(-1)
}

// Automerge::save_after(heads) — automerge.rs:993-1000; heads = n x 32 bytes
int amo_save_after(const amo_doc* d, const uint8_t* heads, size_t n, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  std::vector<Hash> hs(n);
  for (size_t i = 0; i < n; i++) memcpy(hs[i].b.data(), heads + 32 * i, 32);
  return give(save_after(d->doc, hs), out, len);
  GUARD_END(-1)
}

// Automerge::load_incremental — automerge.rs:730
long amo_load_incremental(amo_doc* d, const uint8_t* data, size_t len) {
  GUARD_BEGIN
  return (long)load_incremental(d->doc, data, len);
  GUARD_END(-1)
}

// Automerge::apply_changes over concatenated change chunks (each parsed with Change::try_from)
int amo_apply_changes(amo_doc* d, const uint8_t* data, size_t len) {
  GUARD_BEGIN
  std::vector<StoredChange> cs;
  size_t off = 0;
  while (off < len) {
    ChunkHeader h;
    try {
      h = parse_chunk_header(data + off, len - off);
    } catch (ParseFail& e) {
      throw Error(E_PARSE, e.what());
    }
    size_t tl = h.header_size + h.data_len;
    cs.push_back(change_from_bytes(data + off, tl));
    off += tl;
  }
  apply_changes(d->doc, cs);
  return 0;
  GUARD_END(-1)
}

// Automerge::merge — automerge.rs:917
int amo_merge(amo_doc* dst, const amo_doc* src) {
  GUARD_BEGIN
  merge(dst->doc, src->doc);
  return 0;
  GUARD_END(-1)
}

// Automerge::save / save_nocompress — automerge.rs:943-985
int amo_save(const amo_doc* d, int deflate, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  return give(save(d->doc, deflate != 0), out, len);
  GUARD_END(-1)
}

// Automerge::get_heads — automerge.rs:1300 ; out = n x 32 bytes, sorted
int amo_heads(const amo_doc* d, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  Bytes b;
  for (auto& h : d->doc.get_heads()) b.insert(b.end(), h.b.begin(), h.b.end());
  return give(b, out, len);
  GUARD_END(-1)
}

// history in order: [u32 n] then per change: hash[32] seq u64 start_op u64 nops u64 ndeps u32 deps[32]..
// actor_len u32 actor bytes, raw_len u32 raw chunk bytes
int amo_changes(const amo_doc* d, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  Bytes b;
  put_u32(b, (uint32_t)d->doc.history.size());
  for (auto& c : d->doc.history) {
    b.insert(b.end(), c.hash.b.begin(), c.hash.b.end());
    put_u64(b, c.seq);
    put_u64(b, c.start_op);
    put_u64(b, c.len);
    put_u32(b, (uint32_t)c.deps.size());
    for (auto& dd : c.deps) b.insert(b.end(), dd.b.begin(), dd.b.end());
    put_u32(b, (uint32_t)c.actor.size());
    b.insert(b.end(), c.actor.begin(), c.actor.end());
    put_u32(b, (uint32_t)c.bytes.size());
    b.insert(b.end(), c.bytes.begin(), c.bytes.end());
  }
  return give(b, out, len);
  GUARD_END(-1)
}

int amo_hydrate(const amo_doc* d, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  std::string s = hydrate_json(d->doc);
  return give(Bytes(s.begin(), s.end()), out, len);
  GUARD_END(-1)
}

// make_bloom_filter(last_sync = []) over all changes — sync.rs:243-250, bloom.rs:125-140,38-47
int amo_bloom(const amo_doc* d, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  std::vector<Hash> hs;
  for (auto& c : d->doc.history) hs.push_back(c.hash);
  return give(bloom_from_hashes(hs), out, len);
  GUARD_END(-1)
}
int amo_bloom_from_hashes(const uint8_t* hashes, size_t n, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  std::vector<Hash> hs(n);
  for (size_t i = 0; i < n; i++) memcpy(hs[i].b.data(), hashes + 32 * i, 32);
  return give(bloom_from_hashes(hs), out, len);
  GUARD_END(-1)
}

// generate_sync_message (sync.rs:134-221) over a flat State (layout: amo_sync.hpp).  Returns 1 and the encoded message
// (Message::encode, sync.rs:526-543) or 0 for None; the updated State comes back in the same flat form either way.
static thread_local Bytes g_out2;
int amo_generate_sync_message(const amo_doc* d, const uint8_t* state, size_t state_len, const uint8_t** msg, size_t* msg_len,
                              const uint8_t** new_state, size_t* new_state_len) {
  GUARD_BEGIN
  SyncState st = state_from_blob(state, state_len);
  SyncMessage m;
  bool some = generate_sync_message(d->doc, st, m);
  g_out = some ? encode_message(m) : Bytes();
  g_out2 = state_to_blob(st);
  *msg = g_out.data();
  *msg_len = g_out.size();
  *new_state = g_out2.data();
  *new_state_len = g_out2.size();
  return some ? 1 : 0;
  GUARD_END(-1)
}
// receive_sync_message (sync.rs:223-231, 323-383): applies the message to the document and the flat State
int amo_receive_sync_message(amo_doc* d, const uint8_t* state, size_t state_len, const uint8_t* msg, size_t msg_len,
                             const uint8_t** new_state, size_t* new_state_len) {
  GUARD_BEGIN
  SyncState st = state_from_blob(state, state_len);
  SyncMessage m = parse_message(msg, msg_len);
  receive_sync_message(d->doc, st, m);
  return give(state_to_blob(st), new_state, new_state_len);
  GUARD_END(-1)
}
// Message::decode then Message::encode: must reproduce the bytes (sync.rs:652-659 round-trip property)
int amo_sync_message_reencode(const uint8_t* msg, size_t msg_len, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  return give(encode_message(parse_message(msg, msg_len)), out, len);
  GUARD_END(-1)
}
// BloomFilter::contains_hash over wire bytes (bloom.rs:49-67, 110-123)
int amo_bloom_contains(const uint8_t* bloom, size_t len, const uint8_t* hash32) {
  GUARD_BEGIN
  Hash h;
  memcpy(h.b.data(), hash32, 32);
  return Bloom::parse(bloom, len).contains(h) ? 1 : 0;
  GUARD_END(-1)
}

unsigned long amo_num_ops(const amo_doc* d) {  // ops = sum of change.len()
  unsigned long n = 0;
  for (auto& c : d->doc.history) n += c.len;
  return n;
}
unsigned long amo_max_op(const amo_doc* d) { return d->doc.max_op; }
unsigned long amo_queue_len(const amo_doc* d) { return d->doc.queue.size(); }

// SHA-256 of a buffer (chunk.rs:286-293 hashes type||uleb(len)||data; this is the bare digest)
int amo_sha256(const uint8_t* data, size_t n, uint8_t* out32) {
  Hash h = sha256(data, n);
  memcpy(out32, h.b.data(), 32);
  return 0;
}

// parse one change chunk, decode its ops, rebuild it with the builder: must reproduce the bytes
int amo_change_rebuild(const uint8_t* data, size_t len, const uint8_t** out, size_t* olen) {
  GUARD_BEGIN
  StoredChange c = change_from_bytes(data, len);
  StoredChange r = build_change(c.deps, c.actor, c.other_actors, c.seq, c.start_op, c.timestamp, c.message, c.extra(), c.ops());
  return give(r.bytes, out, olen);
  GUARD_END(-1)
}
// hash of a single change chunk (type 1 or 2)
int amo_change_hash(const uint8_t* data, size_t len, uint8_t* out32) {
  GUARD_BEGIN
  StoredChange c = change_from_bytes(data, len);
  memcpy(out32, c.hash.b.data(), 32);
  return 0;
  GUARD_END(-1)
}

// ---- primitive codecs, for the golden vectors ---------------------------------------------
// values blob: n x { u8 present, 8 bytes LE value } for kinds u64/i64(delta); n x u8 for bool;
// n x { u8 present, u32 len, bytes } for strings.  kind: 0 rle-u64, 1 delta, 2 boolean, 3 rle-string,
// 4 maybe-boolean
int amo_codec_encode(int kind, const uint8_t* vals, size_t n, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  Bytes b;
  const uint8_t* p = vals;
  auto rd64 = [&]() { uint64_t v; memcpy(&v, p, 8); p += 8; return v; };
  if (kind == 0) {
    RleEncoder<uint64_t> e;
    for (size_t i = 0; i < n; i++) { bool pr = *p++; uint64_t v = rd64(); if (pr) e.append_value(v); else e.append_null(); }
    b = e.finish();
  } else if (kind == 1) {
    DeltaEncoder e;
    for (size_t i = 0; i < n; i++) { bool pr = *p++; int64_t v = (int64_t)rd64(); if (pr) e.append_value(v); else e.append_null(); }
    b = e.finish();
  } else if (kind == 2 || kind == 4) {
    if (kind == 2) { BooleanEncoder e; for (size_t i = 0; i < n; i++) e.append(p[i] != 0); b = e.finish(); }
    else { MaybeBooleanEncoder e; for (size_t i = 0; i < n; i++) e.append(p[i] != 0); b = e.finish(); }
  } else if (kind == 3) {
    RleEncoder<std::string> e;
    for (size_t i = 0; i < n; i++) {
      bool pr = *p++;
      uint32_t l; memcpy(&l, p, 4); p += 4;
      std::string s((const char*)p, l); p += l;
      if (pr) e.append_value(s); else e.append_null();
    }
    b = e.finish();
  } else {
    throw Error(E_INVALID_ARG, "bad codec kind");
  }
  return give(b, out, len);
  GUARD_END(-1)
}
// decode `count` rows (boolean) or until exhausted (others) into the same blob format
int amo_codec_decode(int kind, const uint8_t* data, size_t dlen, size_t count, const uint8_t** out, size_t* len) {
  GUARD_BEGIN
  Bytes b;
  try {
    if (kind == 0) {
      RleDecoder<uint64_t> d(data, dlen);
      std::optional<uint64_t> v;
      while (d.next(v)) { b.push_back(v ? 1 : 0); put_u64(b, v ? *v : 0); }
    } else if (kind == 1) {
      DeltaDecoder d(data, dlen);
      std::optional<int64_t> v;
      while (d.next(v)) { b.push_back(v ? 1 : 0); put_u64(b, v ? (uint64_t)*v : 0); }
    } else if (kind == 2) {
      BooleanDecoder d(data, dlen);
      bool v;
      for (size_t i = 0; i < count && d.next(v); i++) b.push_back(v ? 1 : 0);
    } else if (kind == 3) {
      RleDecoder<std::string> d(data, dlen);
      std::optional<std::string> v;
      while (d.next(v)) {
        b.push_back(v ? 1 : 0);
        put_u32(b, v ? (uint32_t)v->size() : 0);
        if (v) b.insert(b.end(), v->begin(), v->end());
      }
    } else {
      throw Error(E_INVALID_ARG, "bad codec kind");
    }
  } catch (DecodeError& e) {
    throw Error(E_PARSE, e.what());
  }
  return give(b, out, len);
  GUARD_END(-1)
}

// strict LEB parser KATs (storage/parse/leb128.rs:106-301): returns 0 ok, 1 too large, 2 overlong,
// 3 incomplete; *consumed = bytes read
int amo_strict_leb(int is_signed, const uint8_t* data, size_t len, uint64_t* value, size_t* consumed) {
  Cursor c(data, len);
  try {
    if (is_signed) *value = (uint64_t)read_sleb_strict(c);
    else *value = read_uleb_strict(c);
    *consumed = c.off;
    return 0;
  } catch (StrictLebError& e) {
    return strstr(e.what(), "overlong") ? 2 : 1;
  } catch (DecodeError&) {
    return 3;
  }
}

// ---- CPU baseline: load (+ materialize) a batch of documents on `threads` host threads ----
// docs = concatenated, offs[n+1] offsets.  Returns elapsed seconds; *ops_out = sum of ops loaded.
double amo_bench_load_batch(const uint8_t* docs, const uint64_t* offs, size_t n, int threads, int materialize,
                            unsigned long* ops_out, unsigned long* failures) {
  std::vector<unsigned long> ops(threads, 0), fails(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; t++) {
    ts.emplace_back([&, t]() {
      for (size_t i = t; i < n; i += threads) {
        try {
          Doc d = load(docs + offs[i], offs[i + 1] - offs[i], true, true);
          for (auto& c : d.history) ops[t] += c.len;
          if (materialize) {
            std::string s = hydrate_json(d);
            if (s.empty()) fails[t]++;
          }
        } catch (std::exception&) {
          fails[t]++;
        }
      }
    });
  }
  for (auto& th : ts) th.join();
  auto t1 = std::chrono::steady_clock::now();
  unsigned long o = 0, f = 0;
  for (int t = 0; t < threads; t++) { o += ops[t]; f += fails[t]; }
  *ops_out = o;
  *failures = f;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
