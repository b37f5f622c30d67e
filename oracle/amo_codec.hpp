// ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not measured, never linked by the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the reference's columnar primitive codecs (sequential, one value at a time,
// exactly as the reference's state machines run).  Each block cites the reference file:line it
// follows (paths relative to /root/reference/rust/automerge/src).
//
// Pinning: see oracle/README.md — KATs from the reference's own tests, the reference's binary
// fixtures, and vectors produced by the reference's legacy JS codec (tests/golden/).
#pragma once
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace amo {

using Bytes = std::vector<uint8_t>;

// ---------------------------------------------------------------------------------------------
// errors: one exception type carrying (kind, sub-kind) mirroring AutomergeError / load::Error
// (error.rs:7-62, storage/load.rs:15-30, reconstruct_document.rs:14-32, change_collector.rs:19-33)
// ---------------------------------------------------------------------------------------------
enum ErrKind : int {
  E_OK = 0,
  E_PARSE = 1,                   // load::Error::Parse
  E_BAD_CHECKSUM = 2,            // load::Error::BadChecksum
  E_INVALID_CHANGE_COLUMNS = 3,  // load::Error::InvalidChangeColumns
  E_INFLATE_DOCUMENT = 4,        // load::Error::InflateDocument(sub)
  E_MISSING_DEPS = 5,            // AutomergeError::MissingDeps
  E_DUPLICATE_SEQ = 6,           // AutomergeError::DuplicateSeqNumber
  E_LEFTOVER = 7,                // change::LoadError::LeftoverData
  E_WRONG_CHUNK_TYPE = 8,        // change::LoadError::WrongChunkType
  E_MISSING_DEP_GRAPH = 9,       // ChangeGraph MissingDep
  E_INVALID_ARG = 10,
};
enum ErrSub : int {
  S_NONE = 0,
  S_OPS_OUT_OF_ORDER = 1,
  S_READ_OP = 2,
  S_MISSING_ACTOR = 3,
  S_MISMATCHING_HEADS = 4,
  S_MISSING_OPS = 5,
  S_SUCC_OUT_OF_ORDER = 6,
  S_INVALID_OP = 7,
  S_CHANGES_OUT_OF_ORDER = 8,   // InvalidChanges(ChangesOutOfOrder)
  S_MISSING_CHANGE = 9,         // InvalidChanges(MissingChange)
  S_READ_CHANGE = 10,           // InvalidChanges(ReadChange)
  S_INCORRECT_MAX_OP = 11,      // InvalidChanges(IncorrectMaxOp)
  S_COLLECT_MISSING_ACTOR = 12, // InvalidChanges(MissingActor)
  S_COLLECT_MISSING_OPS = 13,   // InvalidChanges(MissingOps)
};
struct Error : std::runtime_error {
  ErrKind kind;
  ErrSub sub;
  Error(ErrKind k, const std::string& m, ErrSub s = S_NONE) : std::runtime_error(m), kind(k), sub(s) {}
};
// decode errors below the chunk level; callers wrap them into the Error their context demands
struct DecodeError : std::runtime_error {
  explicit DecodeError(const std::string& m) : std::runtime_error(m) {}
};

// ---------------------------------------------------------------------------------------------
// LEB128 writers — encodable_impls.rs:137-185 (via the `leb128` crate's canonical writer);
// sizes — columnar/encoding/leb128.rs:2-20
// ---------------------------------------------------------------------------------------------
inline size_t put_uleb(Bytes& out, uint64_t v) {
  size_t n = 0;
  do {
    uint8_t b = v & 0x7f;
    v >>= 7;
    if (v) b |= 0x80;
    out.push_back(b);
    n++;
  } while (v);
  return n;
}
inline size_t put_sleb(Bytes& out, int64_t v) {
  size_t n = 0;
  for (;;) {
    uint8_t b = v & 0x7f;
    v >>= 7;  // arithmetic
    bool done = (v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40));
    if (!done) b |= 0x80;
    out.push_back(b);
    n++;
    if (done) break;
  }
  return n;
}
inline uint64_t ulebsize(uint64_t v) {
  uint64_t n = 1;
  while (v >>= 7) n++;
  return n;
}
inline uint64_t lebsize(int64_t v) {
  Bytes tmp;
  return put_sleb(tmp, v);
}

// ---------------------------------------------------------------------------------------------
// byte cursor + the two LEB readers.
//  * lenient = the third-party `leb128` crate 0.2.x readers used by column decoders
//    (decodable_impls.rs:58-96): accepts over-long forms, rejects only >64-bit overflow.
//    (crate source is not in the reference tree: behaviour restated from the published crate;
//     non-canonical accept-set parity is UNPINNED, canonical input is what every writer produces)
//  * strict  = storage/parse/leb128.rs:15-96 used by chunk headers and value payloads.
// ---------------------------------------------------------------------------------------------
struct Cursor {
  const uint8_t* p = nullptr;
  size_t len = 0;
  size_t off = 0;
  Cursor() = default;
  Cursor(const uint8_t* d, size_t n) : p(d), len(n) {}
  bool done() const { return off >= len; }
  bool empty() const { return len == 0; }
  size_t remaining() const { return len - off; }
  uint8_t take1() {
    if (off >= len) throw DecodeError("unexpected end of input");
    return p[off++];
  }
  const uint8_t* take_n(size_t n) {
    if (n > len - off) throw DecodeError("unexpected end of input");
    const uint8_t* r = p + off;
    off += n;
    return r;
  }
};

inline uint64_t read_uleb_lenient(Cursor& c) {
  uint64_t result = 0;
  unsigned shift = 0;
  for (;;) {
    uint8_t b = c.take1();
    if (shift == 63 && b != 0x00 && b != 0x01) {
      while (b & 0x80) b = c.take1();
      throw DecodeError("leb128 overflow");
    }
    result |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return result;
    shift += 7;
    if (shift > 63) {
      // the crate would shift past 64 bits only after rejecting at shift==63 above; a continuation
      // byte of 0x80/0x81 at shift 63 reaches here: keep consuming like the crate (bits fall off)
      for (;;) {
        uint8_t b2 = c.take1();
        if (!(b2 & 0x80)) return result;
      }
    }
  }
}
inline int64_t read_sleb_lenient(Cursor& c) {
  int64_t result = 0;
  unsigned shift = 0;
  uint8_t b;
  for (;;) {
    b = c.take1();
    if (shift == 63 && b != 0x00 && b != 0x7f) {
      while (b & 0x80) b = c.take1();
      throw DecodeError("leb128 overflow");
    }
    result |= (int64_t)((uint64_t)(b & 0x7f) << shift);
    shift += 7;
    if (!(b & 0x80)) break;
    if (shift > 63) {
      for (;;) {
        uint8_t b2 = c.take1();
        if (!(b2 & 0x80)) return result;
      }
    }
  }
  if (shift < 64 && (b & 0x40)) result |= (int64_t)(~(uint64_t)0 << shift);
  return result;
}

struct StrictLebError : std::runtime_error {
  explicit StrictLebError(const char* m) : std::runtime_error(m) {}
};
inline uint64_t read_uleb_strict(Cursor& c) {  // parse/leb128.rs:15-40
  uint64_t res = 0;
  unsigned shift = 0;
  for (;;) {
    uint8_t b = c.take1();
    if (shift < 64) res |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) {
      if (shift > 64 && b > 1) throw StrictLebError("leb128 too large");
      if (shift > 7 && b == 0) throw StrictLebError("leb128 overlong");
      return res;
    } else if (shift > 64) {
      throw StrictLebError("leb128 too large");
    }
  }
}
inline int64_t read_sleb_strict(Cursor& c) {  // parse/leb128.rs:42-76
  uint64_t res = 0;
  unsigned shift = 0;
  uint8_t prev = 0;
  for (;;) {
    uint8_t b = c.take1();
    if (shift < 64) res |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) {
      if (shift > 64 && b != 0 && b != 0x7f) throw StrictLebError("leb128 too large");
      if (shift > 7 && ((b == 0 && !(prev & 0x40)) || (b == 0x7f && (prev & 0x40))))
        throw StrictLebError("leb128 overlong");
      if (shift < 64 && (b & 0x40)) res |= ~(uint64_t)0 << shift;
      return (int64_t)res;
    } else if (shift > 64) {
      throw StrictLebError("leb128 too large");
    }
    prev = b;
  }
}
inline uint32_t read_u32_strict(Cursor& c) {  // parse/leb128.rs:78-85
  uint64_t v = read_uleb_strict(c);
  if (v > 0xffffffffull) throw StrictLebError("leb128 too large");
  return (uint32_t)v;
}

// Rust's str::from_utf8 acceptance (no surrogates, no over-long, <= U+10FFFF)
inline bool valid_utf8(const uint8_t* s, size_t n) {
  size_t i = 0;
  while (i < n) {
    uint8_t b = s[i];
    if (b < 0x80) { i++; continue; }
    if (b >= 0xc2 && b <= 0xdf) {
      if (i + 1 >= n || (s[i + 1] & 0xc0) != 0x80) return false;
      i += 2;
    } else if (b >= 0xe0 && b <= 0xef) {
      if (i + 2 >= n) return false;
      uint8_t b1 = s[i + 1], b2 = s[i + 2];
      if ((b1 & 0xc0) != 0x80 || (b2 & 0xc0) != 0x80) return false;
      if (b == 0xe0 && b1 < 0xa0) return false;
      if (b == 0xed && b1 >= 0xa0) return false;
      i += 3;
    } else if (b >= 0xf0 && b <= 0xf4) {
      if (i + 3 >= n) return false;
      uint8_t b1 = s[i + 1], b2 = s[i + 2], b3 = s[i + 3];
      if ((b1 & 0xc0) != 0x80 || (b2 & 0xc0) != 0x80 || (b3 & 0xc0) != 0x80) return false;
      if (b == 0xf0 && b1 < 0x90) return false;
      if (b == 0xf4 && b1 >= 0x90) return false;
      i += 4;
    } else {
      return false;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// value traits for RLE<T>: T in {uint64_t, int64_t, std::string}
// decode: decodable_impls.rs:58-129 ; encode: encodable_impls.rs
// ---------------------------------------------------------------------------------------------
constexpr size_t MAX_ALLOCATION = 1000000000;  // decodable_impls.rs:9

template <class T> struct Codec;
template <> struct Codec<uint64_t> {
  static uint64_t read(Cursor& c) { return read_uleb_lenient(c); }
  static void write(Bytes& o, uint64_t v) { put_uleb(o, v); }
};
template <> struct Codec<int64_t> {
  static int64_t read(Cursor& c) { return read_sleb_lenient(c); }
  static void write(Bytes& o, int64_t v) { put_sleb(o, v); }
};
template <> struct Codec<std::string> {
  static std::string read(Cursor& c) {
    uint64_t len = read_uleb_lenient(c);
    if (len == 0) return std::string();
    if (len > MAX_ALLOCATION) throw DecodeError("overlarge allocation");
    const uint8_t* p = c.take_n((size_t)len);
    if (!valid_utf8(p, (size_t)len)) throw DecodeError("bad string");
    return std::string((const char*)p, (size_t)len);
  }
  static void write(Bytes& o, const std::string& s) {
    put_uleb(o, s.size());
    o.insert(o.end(), s.begin(), s.end());
  }
};

// ---------------------------------------------------------------------------------------------
// RLE decoder — columnar/encoding/rle.rs:163-239
// next(): returns false when exhausted; otherwise sets `out` (nullopt = null row)
// ---------------------------------------------------------------------------------------------
template <class T> struct RleDecoder {
  Cursor c;
  std::optional<T> last;
  int64_t count = 0;
  bool literal = false;
  RleDecoder() = default;
  RleDecoder(const uint8_t* d, size_t n) : c(d, n) {}
  bool done() const { return c.done() && count == 0; }  // rle.rs:171-173
  bool next(std::optional<T>& out) {                     // rle.rs:175-210
    while (count == 0) {
      if (c.done()) return false;
      int64_t n = read_sleb_lenient(c);
      if (n > 0) {
        count = n;
        last = Codec<T>::read(c);
        literal = false;
      } else if (n < 0) {
        count = (n == INT64_MIN) ? INT64_MAX : -n;
        literal = true;
      } else {
        uint64_t nulls = read_uleb_lenient(c);
        count = (int64_t)nulls;
        if (count < 0) throw DecodeError("null run too long");
        last.reset();
        literal = false;
      }
    }
    count--;
    if (literal) out = Codec<T>::read(c);
    else out = last;
    return true;
  }
};

// Delta decoder — columnar/encoding/delta.rs:53-95 (saturating add; nulls do not advance)
struct DeltaDecoder {
  RleDecoder<int64_t> rle;
  int64_t abs = 0;
  DeltaDecoder() = default;
  DeltaDecoder(const uint8_t* d, size_t n) : rle(d, n) {}
  bool done() const { return rle.done(); }
  bool next(std::optional<int64_t>& out) {
    std::optional<int64_t> d;
    if (!rle.next(d)) return false;
    if (d) {
      int64_t r;
      if (__builtin_add_overflow(abs, *d, &r)) r = (*d > 0) ? INT64_MAX : INT64_MIN;
      abs = r;
      out = abs;
    } else {
      out.reset();
    }
    return true;
  }
};

// Boolean decoder — columnar/encoding/boolean.rs:60-101 ; MaybeBoolean :142-167
struct BooleanDecoder {
  Cursor c;
  bool last = true;
  uint64_t count = 0;
  BooleanDecoder() = default;
  BooleanDecoder(const uint8_t* d, size_t n) : c(d, n) {}
  bool next(bool& out) {
    while (count == 0) {
      if (c.done()) return false;
      count = read_uleb_lenient(c);
      last = !last;
    }
    count--;
    out = last;
    return true;
  }
};
struct MaybeBooleanDecoder {
  BooleanDecoder b;
  MaybeBooleanDecoder() = default;
  MaybeBooleanDecoder(const uint8_t* d, size_t n) : b(d, n) {}
  // returns nullopt for "None" (empty column, or exhausted)
  std::optional<bool> next() {
    if (b.c.empty()) return std::nullopt;
    bool v;
    if (!b.next(v)) return std::nullopt;
    return v;
  }
};

// ---------------------------------------------------------------------------------------------
// RLE encoder — columnar/encoding/rle.rs:8-153 (state machine restated literally)
// ---------------------------------------------------------------------------------------------
template <class T> struct RleEncoder {
  enum St { Empty, InitialNullRun, NullRun, LiteralRun, LoneVal, Run } st = Empty;
  Bytes buf;
  size_t n = 0;         // run / null-run length
  T val{};              // LoneVal / Run value / LiteralRun "last"
  std::vector<T> lit;   // LiteralRun prefix

  void flush_run(const T& v, size_t len) { put_sleb(buf, (int64_t)len); Codec<T>::write(buf, v); }
  void flush_null_run(size_t len) { put_sleb(buf, 0); put_uleb(buf, len); }
  void flush_lit_run(const std::vector<T>& run) {
    put_sleb(buf, -(int64_t)run.size());
    for (auto& v : run) Codec<T>::write(buf, v);
  }
  void append_null() {  // rle.rs:71-90
    switch (st) {
      case Empty: st = InitialNullRun; n = 1; break;
      case InitialNullRun: n++; break;
      case NullRun: n++; break;
      case LoneVal: flush_lit_run({val}); st = NullRun; n = 1; break;
      case Run: flush_run(val, n); st = NullRun; n = 1; break;
      case LiteralRun: lit.push_back(val); flush_lit_run(lit); lit.clear(); st = NullRun; n = 1; break;
    }
  }
  void append_value(const T& v) {  // rle.rs:92-126
    switch (st) {
      case Empty: st = LoneVal; val = v; break;
      case LoneVal:
        if (val == v) { st = Run; n = 2; }
        else { lit.clear(); lit.push_back(val); val = v; st = LiteralRun; }
        break;
      case Run:
        if (val == v) n++;
        else { flush_run(val, n); st = LoneVal; val = v; }
        break;
      case LiteralRun:
        if (val == v) { flush_lit_run(lit); lit.clear(); st = Run; n = 2; }
        else { lit.push_back(val); val = v; }
        break;
      case NullRun:
      case InitialNullRun: flush_null_run(n); st = LoneVal; val = v; break;
    }
  }
  void append(const std::optional<T>& v) { if (v) append_value(*v); else append_null(); }
  Bytes finish() {  // rle.rs:31-46
    switch (st) {
      case InitialNullRun: break;
      case NullRun: flush_null_run(n); break;
      case LoneVal: flush_lit_run({val}); break;
      case Run: flush_run(val, n); break;
      case LiteralRun: lit.push_back(val); flush_lit_run(lit); break;
      case Empty: break;
    }
    st = Empty;
    return std::move(buf);
  }
};

// Delta encoder — columnar/encoding/delta.rs:10-43 (saturating_sub)
struct DeltaEncoder {
  RleEncoder<int64_t> rle;
  int64_t abs = 0;
  void append_value(int64_t v) {
    int64_t d;
    if (__builtin_sub_overflow(v, abs, &d)) d = (v > abs) ? INT64_MAX : INT64_MIN;
    rle.append_value(d);
    abs = v;
  }
  void append_null() { rle.append_null(); }
  void append(const std::optional<int64_t>& v) { if (v) append_value(*v); else append_null(); }
  Bytes finish() { return rle.finish(); }
};

// Boolean encoders — columnar/encoding/boolean.rs:11-51, 105-138
struct BooleanEncoder {
  Bytes buf;
  bool last = false;
  uint64_t count = 0;
  void append(bool v) {
    if (v == last) count++;
    else { put_uleb(buf, count); last = v; count = 1; }
  }
  Bytes finish() { if (count > 0) put_uleb(buf, count); return std::move(buf); }
};
struct MaybeBooleanEncoder {
  BooleanEncoder e;
  bool all_false = true;
  void append(bool v) { if (v) all_false = false; e.append(v); }
  Bytes finish() { if (all_false) return Bytes(); return e.finish(); }
};

}  // namespace amo
