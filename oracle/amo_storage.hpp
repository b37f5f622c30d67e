// ORACLE — TEST INFRASTRUCTURE ONLY (see amo_codec.hpp header).
//
// CPU restatement of the reference's storage layer: chunk container, column metadata, value
// columns, change chunks (parse / verify / build) and document chunks (parse / build).
// Paths cited are relative to /root/reference/rust/automerge/src.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <map>
#include <set>

#include "amo_codec.hpp"
#include "amo_sha256.hpp"

namespace amo {

// ---------------------------------------------------------------------------------------------
// raw DEFLATE via zlib.  The reference uses flate2 ^1.0.22 (raw_column.rs:54,82; compressed.rs:23;
// chunk.rs:79).  Inflate is deterministic for any conforming inflater (pinned by the fixture
// two_change_chunks_compressed.automerge).  DEFLATE OUTPUT BYTES: PARITY UNPINNED — flate2's
// bitstream is not reproducible here and no reference test compares compressed bytes.
// ---------------------------------------------------------------------------------------------
inline bool inflate_raw(const uint8_t* src, size_t n, Bytes& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  uint8_t buf[16384];
  int rc;
  do {
    zs.next_out = buf;
    zs.avail_out = sizeof(buf);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
      inflateEnd(&zs);
      return false;
    }
    out.insert(out.end(), buf, buf + (sizeof(buf) - zs.avail_out));
    if (rc == Z_BUF_ERROR && zs.avail_in == 0) break;  // truncated input
  } while (rc != Z_STREAM_END);
  inflateEnd(&zs);
  // flate2's read_to_end on a truncated stream yields an "unexpected EOF" io error
  return rc == Z_STREAM_END;
}
inline Bytes deflate_raw(const uint8_t* src, size_t n) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  Bytes out(deflateBound(&zs, n));
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  zs.next_out = out.data();
  zs.avail_out = (uInt)out.size();
  deflate(&zs, Z_FINISH);
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

// ---------------------------------------------------------------------------------------------
// scalar values — value.rs:457-472 (ScalarValue), columnar/column_range/value.rs:386-432 (meta)
// tag = wire type code: 0 null 1 false 2 true 3 uint 4 int 5 f64 6 str 7 bytes 8 counter 9 timestamp
// 10..15 unknown.  f64 is kept as its 8 raw bytes (bit pass-through).
// ---------------------------------------------------------------------------------------------
struct Scalar {
  uint8_t tag = 0;
  uint64_t u = 0;   // uint
  int64_t i = 0;    // int / counter start / timestamp
  Bytes bytes;      // str / bytes / unknown / f64 (8 bytes LE)
  bool operator==(const Scalar& o) const { return tag == o.tag && u == o.u && i == o.i && bytes == o.bytes; }
};
inline Scalar sc_null() { return Scalar(); }
inline Scalar sc_uint(uint64_t v) { Scalar s; s.tag = 3; s.u = v; return s; }
inline Scalar sc_int(int64_t v) { Scalar s; s.tag = 4; s.i = v; return s; }
inline Scalar sc_str(const std::string& v) { Scalar s; s.tag = 6; s.bytes.assign(v.begin(), v.end()); return s; }
inline Scalar sc_bytes(const Bytes& v) { Scalar s; s.tag = 7; s.bytes = v; return s; }
inline Scalar sc_counter(int64_t v) { Scalar s; s.tag = 8; s.i = v; return s; }

inline uint64_t value_meta(const Scalar& s) {  // value.rs:412-432
  switch (s.tag) {
    case 0: return 0;
    case 1: return 1;
    case 2: return 2;
    case 3: return (ulebsize(s.u) << 4) | 3;
    case 4: return (lebsize(s.i) << 4) | 4;
    case 5: return (8 << 4) | 5;
    case 8: return (lebsize(s.i) << 4) | 8;
    case 9: return (lebsize(s.i) << 4) | 9;
    default: return ((uint64_t)s.bytes.size() << 4) | s.tag;  // 6, 7, unknown
  }
}
inline void encode_val(Bytes& out, const Scalar& s) {  // value.rs:356-369
  switch (s.tag) {
    case 0: case 1: case 2: break;
    case 3: put_uleb(out, s.u); break;
    case 4: case 8: case 9: put_sleb(out, s.i); break;
    default: out.insert(out.end(), s.bytes.begin(), s.bytes.end()); break;  // 5,6,7,unknown
  }
}

// ValueIter — columnar/column_range/value.rs:203-313
struct ValueDecoder {
  RleDecoder<uint64_t> meta;
  Cursor raw;
  ValueDecoder() = default;
  ValueDecoder(const uint8_t* m, size_t mn, const uint8_t* r, size_t rn) : meta(m, mn), raw(r, rn) {}
  bool done() const { return meta.done(); }
  // false = exhausted; throws DecodeError on malformed rows
  bool next(Scalar& out) {
    std::optional<uint64_t> m;
    if (!meta.next(m)) return false;
    if (!m) throw DecodeError("unexpected null in value meta");
    uint64_t len = *m >> 4;
    uint8_t code = (uint8_t)(*m & 0x0f);
    out = Scalar();
    out.tag = code;
    if (code <= 2) return true;
    if (len > raw.remaining()) throw DecodeError("value: trying to read past end");
    const uint8_t* p = raw.take_n((size_t)len);
    switch (code) {
      case 3: case 4: case 8: case 9: {
        Cursor c(p, (size_t)len);
        try {
          if (code == 3) out.u = read_uleb_strict(c);
          else out.i = read_sleb_strict(c);
        } catch (StrictLebError& e) {
          throw DecodeError(std::string("value: ") + e.what());
        }
        if (!c.done()) throw DecodeError("value: extra bytes");
        break;
      }
      case 5:
        if (len != 8) throw DecodeError("float should have length 8");
        out.bytes.assign(p, p + 8);
        break;
      case 6:
        if (!valid_utf8(p, (size_t)len)) throw DecodeError("value: invalid utf8");
        out.bytes.assign(p, p + len);
        break;
      default:  // 7 bytes, 10..15 unknown
        out.bytes.assign(p, p + len);
        break;
    }
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// column metadata — storage/columns/raw_column.rs:205-272, column_specification.rs:5-42,
// columns.rs:134-355 (layout parser; note `last_spec` is never assigned there, so neither the
// duplicate nor the out-of-order check of the layout parser can fire — only are_normal_sorted does)
// ---------------------------------------------------------------------------------------------
enum ColType { T_GROUP = 0, T_ACTOR = 1, T_INT = 2, T_DELTA = 3, T_BOOL = 4, T_STR = 5, T_VMETA = 6, T_VRAW = 7 };
struct RawCol {
  uint32_t spec = 0;
  size_t start = 0, end = 0;  // offsets within the column data block
  uint32_t id() const { return spec >> 4; }
  int type() const { return spec & 7; }
  bool deflate() const { return spec & 8; }
  uint32_t norm() const { return spec & ~8u; }
  size_t len() const { return end - start; }
};
struct ParseFail : std::runtime_error {
  explicit ParseFail(const std::string& m) : std::runtime_error(m) {}
};

inline std::vector<RawCol> parse_raw_columns(Cursor& c) {
  uint64_t n = read_uleb_strict(c);
  std::vector<RawCol> cols;
  size_t off = 0;
  for (uint64_t i = 0; i < n; i++) {
    RawCol rc;
    rc.spec = read_u32_strict(c);
    uint64_t len = read_uleb_strict(c);
    rc.start = off;
    size_t end;
    if (__builtin_add_overflow(off, (size_t)len, &end)) end = SIZE_MAX;  // saturating (raw_column.rs:225)
    rc.end = end;
    off = end;
    cols.push_back(rc);
  }
  for (size_t i = 1; i < cols.size(); i++)
    if (cols[i].norm() < cols[i - 1].norm()) throw ParseFail("columns were not in normalized order");
  return cols;
}
inline size_t total_column_len(const std::vector<RawCol>& cols) {
  size_t t = 0;
  for (auto& c : cols) t += c.len();
  return t;
}
inline void write_raw_columns(Bytes& out, const std::vector<std::pair<uint32_t, size_t>>& cols) {
  // RawColumns::from_iter drops empty columns (raw_column.rs:168-171); write :245-252
  size_t n = 0;
  for (auto& c : cols) if (c.second) n++;
  put_uleb(out, n);
  for (auto& c : cols) if (c.second) { put_uleb(out, c.first); put_uleb(out, c.second); }
}

// logical columns produced by the layout parser
struct LCol {
  uint32_t id = 0;
  int type = 0;               // type of the first raw column
  size_t s = 0, e = 0;        // simple range / value-meta range / group-num range
  size_t rs = 0, re = 0;      // value raw range
  std::vector<LCol> kids;     // group children (simple or value)
};
inline std::vector<LCol> parse_layout(const std::vector<RawCol>& raw, size_t data_size) {
  std::vector<LCol> out;
  enum { Ready, InValue, InGroup } st = Ready;
  bool group_in_value = false;
  LCol cur;       // pending value column or group
  LCol gval;      // pending value inside group
  auto finish_pending = [&]() {
    if (st == InValue) { out.push_back(cur); }
    else if (st == InGroup) {
      if (group_in_value) { cur.kids.push_back(gval); group_in_value = false; }
      out.push_back(cur);
    }
    st = Ready;
  };
  for (size_t idx = 0; idx < raw.size(); idx++) {
    const RawCol& rc = raw[idx];
    if (rc.end > data_size) throw ParseFail("bad column layout: data out of range");
    bool again = true;
    while (again) {
      again = false;
      if (st == Ready) {
        LCol c; c.id = rc.id(); c.type = rc.type(); c.s = rc.start; c.e = rc.end;
        switch (rc.type()) {
          case T_GROUP: cur = c; st = InGroup; group_in_value = false; break;
          case T_VMETA: cur = c; cur.rs = cur.re = 0; st = InValue; break;
          case T_VRAW: throw ParseFail("bad column layout: raw value column without metadata column");
          default: out.push_back(c); break;
        }
      } else if (st == InValue) {
        if (rc.type() == T_VRAW) {
          if (cur.id != rc.id()) throw ParseFail("bad column layout: mismatching value metadata id");
          cur.rs = rc.start; cur.re = rc.end;
          out.push_back(cur);
          st = Ready;
        } else {
          out.push_back(cur);
          st = Ready;
          again = true;
        }
      } else {  // InGroup
        if (cur.id != rc.id()) {
          finish_pending();
          again = true;
        } else if (!group_in_value) {
          LCol c; c.id = rc.id(); c.type = rc.type(); c.s = rc.start; c.e = rc.end;
          switch (rc.type()) {
            case T_GROUP: throw ParseFail("bad column layout: nested group");
            case T_VRAW: throw ParseFail("bad column layout: raw value column without metadata column");
            case T_VMETA: gval = c; gval.rs = gval.re = 0; group_in_value = true; break;
            default: cur.kids.push_back(c); break;
          }
        } else {
          if (rc.type() == T_VRAW) {
            gval.rs = rc.start; gval.re = rc.end;
            cur.kids.push_back(gval);
            group_in_value = false;
          } else {
            cur.kids.push_back(gval);
            group_in_value = false;
            again = true;
          }
        }
      }
    }
  }
  finish_pending();
  return out;
}

// ---------------------------------------------------------------------------------------------
// chunk container — storage/chunk.rs:181-293, storage.rs:24
// ---------------------------------------------------------------------------------------------
static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};
enum ChunkType { CHUNK_DOC = 0, CHUNK_CHANGE = 1, CHUNK_COMPRESSED = 2 };

inline Hash chunk_hash(uint8_t type, const uint8_t* data, size_t n) {  // chunk.rs:286-293
  Bytes pre;
  pre.push_back(type);
  put_uleb(pre, n);
  Sha256 s;
  s.update(pre.data(), pre.size());
  s.update(data, n);
  return s.finish();
}
struct ChunkHeader {
  uint8_t checksum[4];
  uint8_t type;
  size_t header_size;
  size_t data_len;
  Hash hash;
  bool checksum_valid() const { return memcmp(checksum, hash.b.data(), 4) == 0; }
};
// Header::parse — chunk.rs:223-266.  `in` is positioned at the start of a chunk.
inline ChunkHeader parse_chunk_header(const uint8_t* p, size_t n) {
  Cursor c(p, n);
  ChunkHeader h;
  try {
    const uint8_t* m = c.take_n(4);
    if (memcmp(m, MAGIC, 4) != 0) throw ParseFail("invalid magic bytes");
    memcpy(h.checksum, c.take_n(4), 4);
    h.type = c.take1();
    if (h.type > 2) throw ParseFail("unknown chunk type");
    uint64_t len = read_uleb_strict(c);
    h.header_size = c.off;
    if (len > c.remaining()) throw ParseFail("incomplete chunk");
    h.data_len = (size_t)len;
  } catch (DecodeError& e) {
    throw ParseFail(e.what());
  } catch (StrictLebError& e) {
    throw ParseFail(e.what());
  }
  h.hash = chunk_hash(h.type, p + h.header_size, h.data_len);
  return h;
}
inline void write_chunk(Bytes& out, uint8_t type, const uint8_t checksum[4], const Bytes& data) {
  out.insert(out.end(), MAGIC, MAGIC + 4);
  out.insert(out.end(), checksum, checksum + 4);
  out.push_back(type);
  put_uleb(out, data.size());
  out.insert(out.end(), data.begin(), data.end());
}

// ---------------------------------------------------------------------------------------------
// op ids in storage form: actor = index into the chunk's own actor table
// ---------------------------------------------------------------------------------------------
struct OpId {
  uint32_t ctr = 0;
  uint32_t actor = 0;
  bool operator==(const OpId& o) const { return ctr == o.ctr && actor == o.actor; }
  bool operator!=(const OpId& o) const { return !(*this == o); }
  bool operator<(const OpId& o) const { return ctr != o.ctr ? ctr < o.ctr : actor < o.actor; }  // derived Ord, types.rs:498
};

// a row of a change chunk (storage/change/change_op_columns.rs:38-48) or of a document chunk
// (storage/document/doc_op_columns.rs:35-46): same shape, `ids` = pred (change) or succ (doc)
struct StoredOp {
  OpId id;             // doc rows only
  OpId obj;            // ctr 0 = root
  bool key_is_prop = false;
  std::string prop;
  OpId elem;           // ctr 0 = HEAD
  bool insert = false;
  uint64_t action = 0;
  Scalar val;
  std::vector<OpId> ids;  // pred (change) / succ (doc)
  bool expand = false;
  std::optional<std::string> mark_name;
};

inline void check_action_value(uint64_t action, const Scalar& v) {  // types.rs:236-250
  if (action <= 4 || action == 6 || action == 7) return;
  if (action == 5) {
    if (v.tag == 3 || v.tag == 4) return;
    throw DecodeError("non numeric increment");
  }
  throw DecodeError("unknown action");
}

struct ColRange {
  size_t s = 0, e = 0;
  bool empty() const { return s == e; }
};
struct OpColumns {
  ColRange obj_actor, obj_ctr, key_actor, key_ctr, key_str, id_actor, id_ctr, insert, action, val_meta, val_raw,
      grp_num, grp_actor, grp_ctr, expand, mark_name;
};
struct MismatchingColumn : std::runtime_error {
  MismatchingColumn() : std::runtime_error("mismatching column") {}
};
// DocOpColumns::try_from (doc_op_columns.rs:398-504) / ChangeOpsColumns::try_from
// (change_op_columns.rs:422-523).  grp_id = 8 (succ) for documents, 7 (pred) for changes.
inline OpColumns map_op_columns(const std::vector<LCol>& cols, bool is_doc) {
  OpColumns o;
  uint32_t grp_id = is_doc ? 8 : 7;
  for (auto& c : cols) {
    auto r = ColRange{c.s, c.e};
    if (c.id == 0 && c.type == T_ACTOR) o.obj_actor = r;
    else if (c.id == 0 && c.type == T_INT) o.obj_ctr = r;
    else if (c.id == 1 && c.type == T_ACTOR) o.key_actor = r;
    else if (c.id == 1 && c.type == T_DELTA) o.key_ctr = r;
    else if (c.id == 1 && c.type == T_STR) o.key_str = r;
    else if (is_doc && c.id == 2 && c.type == T_ACTOR) o.id_actor = r;
    else if (is_doc && c.id == 2 && c.type == T_DELTA) o.id_ctr = r;
    else if (c.id == 3 && c.type == T_BOOL) o.insert = r;
    else if (c.id == 4 && c.type == T_INT) o.action = r;
    else if (c.id == 5 && c.type == T_VMETA) { o.val_meta = r; o.val_raw = ColRange{c.rs, c.re}; }
    else if (c.id == grp_id && c.type == T_GROUP) {
      o.grp_num = r;
      if (c.kids.empty()) {
        o.grp_actor = ColRange{};
        o.grp_ctr = ColRange{};
      } else {
        if (c.kids.size() != 2) throw MismatchingColumn();
        const LCol& a = c.kids[0];
        const LCol& b = c.kids[1];
        bool a_ok = (a.type == T_ACTOR || a.type == T_INT);
        bool b_ok = (b.type == T_DELTA);
        if (!a_ok || !b_ok) throw MismatchingColumn();
        o.grp_actor = ColRange{a.s, a.e};
        o.grp_ctr = ColRange{b.s, b.e};
      }
    } else if (c.id == 9 && c.type == T_BOOL) o.expand = r;
    else if (c.id == 10 && c.type == T_STR) o.mark_name = r;
    // anything else: unknown column, kept aside by the reference and dropped on re-save
  }
  return o;
}

// lock-step row iterator — DocOpColumnIter (doc_op_columns.rs:314-384) / ChangeOpsIter
// (change_op_columns.rs:343-408) with the composite decoders: ObjIdIter (obj_id.rs:116-142),
// KeyIter (key.rs:138-188), OpIdIter (opid.rs:82-121), OpIdListIter (opid_list.rs:173-227)
struct OpRowIter {
  bool is_doc;
  bool has_obj;
  RleDecoder<uint64_t> obj_actor, obj_ctr, key_actor, id_actor, action, grp_num, grp_actor;
  DeltaDecoder key_ctr, id_ctr, grp_ctr;
  RleDecoder<std::string> key_str, mark_name;
  BooleanDecoder insert;
  MaybeBooleanDecoder expand;
  ValueDecoder val;

  OpRowIter(const uint8_t* d, const OpColumns& c, bool is_doc_) : is_doc(is_doc_) {
    auto P = [&](const ColRange& r) { return d + r.s; };
    auto N = [&](const ColRange& r) { return r.e - r.s; };
    // ObjIdRange::new returns None when both ranges are empty (obj_id.rs:19-25)
    has_obj = !(c.obj_actor.empty() && c.obj_ctr.empty());
    obj_actor = RleDecoder<uint64_t>(P(c.obj_actor), N(c.obj_actor));
    obj_ctr = RleDecoder<uint64_t>(P(c.obj_ctr), N(c.obj_ctr));
    key_actor = RleDecoder<uint64_t>(P(c.key_actor), N(c.key_actor));
    key_ctr = DeltaDecoder(P(c.key_ctr), N(c.key_ctr));
    key_str = RleDecoder<std::string>(P(c.key_str), N(c.key_str));
    id_actor = RleDecoder<uint64_t>(P(c.id_actor), N(c.id_actor));
    id_ctr = DeltaDecoder(P(c.id_ctr), N(c.id_ctr));
    insert = BooleanDecoder(P(c.insert), N(c.insert));
    action = RleDecoder<uint64_t>(P(c.action), N(c.action));
    val = ValueDecoder(P(c.val_meta), N(c.val_meta), P(c.val_raw), N(c.val_raw));
    grp_num = RleDecoder<uint64_t>(P(c.grp_num), N(c.grp_num));
    grp_actor = RleDecoder<uint64_t>(P(c.grp_actor), N(c.grp_actor));
    grp_ctr = DeltaDecoder(P(c.grp_ctr), N(c.grp_ctr));
    expand = MaybeBooleanDecoder(P(c.expand), N(c.expand));
    mark_name = RleDecoder<std::string>(P(c.mark_name), N(c.mark_name));
  }
  bool done() const { return is_doc ? id_ctr.done() : action.done(); }

  static uint32_t ctr32(uint64_t v) {
    // OpId::new unwraps a u32 conversion (types.rs:502-504): the reference panics or errors on
    // larger counters (tests/test.rs:1451-1464 accepts "fails to load"); the oracle errors.
    if (v > 0xffffffffull) throw DecodeError("counter too large");
    return (uint32_t)v;
  }
  static uint32_t actor32(uint64_t v) {
    if (v > 0xffffffffull) throw DecodeError("actor index too large");
    return (uint32_t)v;
  }

  void read_obj(StoredOp& op) {
    if (!has_obj) { op.obj = OpId{0, 0}; return; }
    std::optional<uint64_t> a, c;
    bool ha = obj_actor.next(a), hc = obj_ctr.next(c);
    bool an = !ha || !a, cn = !hc || !c;
    if (an && cn) { op.obj = OpId{0, 0}; return; }
    if (!an && !cn) { op.obj = OpId{ctr32(*c), actor32(*a)}; return; }
    if (!cn && *c == 0) { op.obj = OpId{0, 0}; return; }
    // obj_id.rs:134-140 — when both streams are exhausted the iterator still yields root above
    if (an) throw DecodeError("unexpected null obj actor");
    throw DecodeError("unexpected null obj counter");
  }
  void read_key(StoredOp& op) {
    std::optional<uint64_t> a;
    std::optional<int64_t> c;
    std::optional<std::string> s;
    bool ha = key_actor.next(a), hc = key_ctr.next(c), hs = key_str.next(s);
    bool an = !ha || !a, cn = !hc || !c, sn = !hs || !s;
    if (!an && !cn && !sn) throw DecodeError("key: too many values");
    if (an && cn && !sn) { op.key_is_prop = true; op.prop = *s; return; }
    if (an && !cn && *c == 0 && sn) { op.key_is_prop = false; op.elem = OpId{0, 0}; return; }
    if (!an && !cn && sn) {
      if (*c < 0) throw DecodeError("negative value for counter");
      op.key_is_prop = false;
      op.elem = OpId{ctr32((uint64_t)*c), actor32(*a)};
      return;
    }
    if (an && cn && sn) throw DecodeError("unexpected null key");  // iterator yields None -> next_in_col errors
    if (an) throw DecodeError("unexpected null key actor");
    throw DecodeError("unexpected null key counter");
  }
  void read_group(StoredOp& op) {
    std::optional<uint64_t> n;
    if (!grp_num.next(n)) throw DecodeError("unexpected null group");
    if (!n) throw DecodeError("unexpected null group num");
    op.ids.clear();
    for (uint64_t i = 0; i < *n; i++) {
      std::optional<uint64_t> a;
      std::optional<int64_t> c;
      bool ha = grp_actor.next(a), hc = grp_ctr.next(c);
      if (!ha || !a) throw DecodeError("unexpected null group actor");
      if (!hc || !c) throw DecodeError("unexpected null group counter");
      if (*c < 0) throw DecodeError("negative value for counter");
      op.ids.push_back(OpId{ctr32((uint64_t)*c), actor32(*a)});
    }
  }
  // returns false when exhausted
  bool next(StoredOp& op) {
    if (done()) return false;
    op = StoredOp();
    if (is_doc) {
      std::optional<uint64_t> a;
      std::optional<int64_t> c;
      bool ha = id_actor.next(a), hc = id_ctr.next(c);
      if (!ha && !hc) return false;
      if (!ha || !a || !hc || !c) throw DecodeError("unexpected null id");
      if (*c < 0 || *c > 0xffffffffll) throw DecodeError("negative or large value for id counter");
      op.id = OpId{(uint32_t)*c, actor32(*a)};
    }
    auto read_action = [&]() {
      std::optional<uint64_t> act;
      if (!action.next(act) || !act) throw DecodeError("unexpected null action");
      op.action = *act;
    };
    auto read_insert = [&]() {
      bool ins;
      if (!insert.next(ins)) throw DecodeError("unexpected null insert");
      op.insert = ins;
    };
    auto read_val = [&]() {
      if (!val.next(op.val)) throw DecodeError("unexpected null value");
    };
    if (is_doc) {
      read_action(); read_obj(op); read_key(op); read_val(); read_group(op); read_insert();
    } else {
      read_obj(op); read_key(op); read_insert(); read_action(); read_val(); read_group(op);
    }
    op.expand = expand.next().value_or(false);
    std::optional<std::string> mn;
    if (mark_name.next(mn)) op.mark_name = mn;
    if (!is_doc) check_action_value(op.action, op.val);  // change_op_columns.rs:378-380
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// op column encoders — ChangeOpsColumns::encode (change_op_columns.rs:146-329) and
// DocOpColumns::encode (doc_op_columns.rs:104-311).  The reference switches between a column-wise
// and a row-wise driver by op count; both drive the same encoders with the same value sequences
// (asserted by value.rs:531-537), so one restatement serves.
// ---------------------------------------------------------------------------------------------
struct EncodedOpCols {
  std::vector<std::pair<uint32_t, size_t>> meta;  // (spec, len) in write order, empties included
  Bytes data;
};
inline EncodedOpCols encode_op_columns(const std::vector<StoredOp>& ops, bool is_doc) {
  RleEncoder<uint64_t> obj_actor, obj_ctr, key_actor, id_actor, action, vmeta, gnum, gactor;
  DeltaEncoder key_ctr, id_ctr, gctr;
  RleEncoder<std::string> key_str, mark_name;
  BooleanEncoder insert;
  MaybeBooleanEncoder expand;
  Bytes vraw;
  for (auto& op : ops) {
    if (op.obj.ctr == 0) { obj_actor.append_null(); obj_ctr.append_null(); }           // obj_id.rs:158-172
    else { obj_actor.append_value(op.obj.actor); obj_ctr.append_value(op.obj.ctr); }
    if (op.key_is_prop) { key_str.append_value(op.prop); key_actor.append_null(); key_ctr.append_null(); }  // key.rs:236-257
    else if (op.elem.ctr == 0) { key_str.append_null(); key_actor.append_null(); key_ctr.append_value(0); }
    else { key_str.append_null(); key_actor.append_value(op.elem.actor); key_ctr.append_value(op.elem.ctr); }
    if (is_doc) { id_actor.append_value(op.id.actor); id_ctr.append_value(op.id.ctr); }  // opid.rs:137-140
    insert.append(op.insert);
    action.append_value(op.action);
    vmeta.append_value(value_meta(op.val));  // value.rs:326-331
    encode_val(vraw, op.val);
    gnum.append_value(op.ids.size());        // opid_list.rs:244-254
    for (auto& i : op.ids) { gactor.append_value(i.actor); gctr.append_value(i.ctr); }
    expand.append(op.expand);
    if (op.mark_name) mark_name.append_value(*op.mark_name); else mark_name.append_null();
  }
  EncodedOpCols out;
  auto add = [&](uint32_t id, int type, Bytes b) {
    out.meta.push_back({(id << 4) | (uint32_t)type, b.size()});
    out.data.insert(out.data.end(), b.begin(), b.end());
  };
  Bytes oa = obj_actor.finish(), oc = obj_ctr.finish();
  // ObjIdRange::encode returns None (both columns dropped) when the actor column is empty
  // (obj_id.rs:79-81); the row-wise encoder drops only when both are empty (:193-201) — the same
  // thing, since the two columns are null on exactly the same rows.
  if (oa.empty()) oc.clear();
  add(0, T_ACTOR, oa);
  add(0, T_INT, oc);
  add(1, T_ACTOR, key_actor.finish());
  add(1, T_DELTA, key_ctr.finish());
  add(1, T_STR, key_str.finish());
  if (is_doc) { add(2, T_ACTOR, id_actor.finish()); add(2, T_DELTA, id_ctr.finish()); }
  add(3, T_BOOL, insert.finish());
  add(4, T_INT, action.finish());
  add(5, T_VMETA, vmeta.finish());
  add(5, T_VRAW, vraw);
  uint32_t gid = is_doc ? 8 : 7;
  add(gid, T_GROUP, gnum.finish());
  Bytes ga = gactor.finish(), gc = gctr.finish();
  if (ga.empty()) gc.clear();  // change_op_columns.rs:304-315 / doc_op_columns.rs:286-297
  add(gid, T_ACTOR, ga);
  add(gid, T_DELTA, gc);
  add(9, T_BOOL, expand.finish());
  add(10, T_STR, mark_name.finish());
  return out;
}

// ---------------------------------------------------------------------------------------------
// change chunk — storage/change.rs:43-60, :99-156 (parse), :173-198 (verify), :444-512 (build)
// ---------------------------------------------------------------------------------------------
struct StoredChange {
  Bytes bytes;     // whole (uncompressed) chunk including header
  Hash hash;
  uint8_t checksum[4];
  size_t header_size = 0;
  std::vector<Hash> deps;
  Bytes actor;
  std::vector<Bytes> other_actors;
  uint64_t seq = 0;
  uint64_t start_op = 0;
  int64_t timestamp = 0;
  std::string message;  // empty = None
  OpColumns cols;
  size_t ops_start = 0, ops_end = 0;      // column data, offsets into bytes
  size_t extra_start = 0, extra_end = 0;
  size_t len = 0;                          // number of ops (change.rs:21,33)
  Bytes compressed;                        // original type-2 chunk bytes if it arrived compressed

  uint64_t max_op() const { return start_op + len - 1; }  // change.rs:63-65
  Bytes extra() const { return Bytes(bytes.begin() + extra_start, bytes.begin() + extra_end); }
  std::vector<StoredOp> ops() const {
    std::vector<StoredOp> out;
    OpRowIter it(bytes.data() + ops_start, cols, false);
    StoredOp op;
    while (it.next(op)) out.push_back(op);
    return out;
  }
};

// Change::parse_following_header.  `chunk` = whole chunk bytes incl. header; h = its parsed header.
inline StoredChange parse_change_body(const uint8_t* chunk, const ChunkHeader& h) {
  StoredChange ch;
  ch.bytes.assign(chunk, chunk + h.header_size + h.data_len);
  ch.hash = h.hash;
  memcpy(ch.checksum, h.checksum, 4);
  ch.header_size = h.header_size;
  Cursor c(ch.bytes.data() + h.header_size, h.data_len);
  size_t base = h.header_size;
  try {
    uint64_t ndeps = read_uleb_strict(c);
    for (uint64_t i = 0; i < ndeps; i++) {
      Hash d;
      memcpy(d.b.data(), c.take_n(32), 32);
      ch.deps.push_back(d);
    }
    uint64_t alen = read_uleb_strict(c);
    if (alen > c.remaining()) throw ParseFail("incomplete");
    const uint8_t* ap = c.take_n((size_t)alen);
    ch.actor.assign(ap, ap + alen);
    ch.seq = read_uleb_strict(c);
    ch.start_op = read_uleb_strict(c);
    if (ch.start_op == 0) throw ParseFail("unexpected zero start_op");
    ch.timestamp = read_sleb_strict(c);
    uint64_t mlen = read_uleb_strict(c);
    if (mlen > c.remaining()) throw ParseFail("incomplete");
    const uint8_t* mp = c.take_n((size_t)mlen);
    if (!valid_utf8(mp, (size_t)mlen)) throw ParseFail("invalid utf8 in message");
    ch.message.assign((const char*)mp, (size_t)mlen);
    uint64_t nother = read_uleb_strict(c);
    for (uint64_t i = 0; i < nother; i++) {
      uint64_t l = read_uleb_strict(c);
      if (l > c.remaining()) throw ParseFail("incomplete");
      const uint8_t* p = c.take_n((size_t)l);
      ch.other_actors.emplace_back(p, p + l);
    }
    std::vector<RawCol> raw = parse_raw_columns(c);
    size_t total = total_column_len(raw);
    if (total > c.remaining()) throw ParseFail("incomplete column data");
    ch.ops_start = base + c.off;
    c.take_n(total);
    ch.ops_end = base + c.off;
    ch.extra_start = base + c.off;
    ch.extra_end = base + c.len;
    for (auto& rc : raw)
      if (rc.deflate()) throw ParseFail("change contained compressed columns");
    std::vector<LCol> layout = parse_layout(raw, total);
    ch.cols = map_op_columns(layout, false);
  } catch (DecodeError& e) {
    throw ParseFail(e.what());
  } catch (StrictLebError& e) {
    throw ParseFail(e.what());
  } catch (MismatchingColumn& e) {
    throw ParseFail(e.what());
  }
  return ch;
}

// Change::new_from_unverified → verify_ops (change.rs:29-45, storage/change.rs:173-198)
inline void verify_change_ops(StoredChange& ch) {
  OpRowIter it(ch.bytes.data() + ch.ops_start, ch.cols, false);
  StoredOp op;
  size_t n = 0;
  while (it.next(op)) n++;  // throws DecodeError on malformed rows
  if (ch.start_op > 0xffffffffull) throw DecodeError("counter too large");
  ch.len = n;
}

// ChangeBuilder::build — storage/change.rs:444-512.  `ops` carry actor indices into
// [actor] ++ other_actors already (ChangeActors translation done by the caller).
inline StoredChange build_change(std::vector<Hash> deps, const Bytes& actor, const std::vector<Bytes>& other_actors,
                                 uint64_t seq, uint64_t start_op, int64_t timestamp, const std::string& message,
                                 const Bytes& extra, const std::vector<StoredOp>& ops) {
  std::sort(deps.begin(), deps.end());  // with_dependencies sorts (change.rs:335-341)
  EncodedOpCols cols = encode_op_columns(ops, false);
  Bytes data;
  put_uleb(data, deps.size());
  for (auto& d : deps) data.insert(data.end(), d.b.begin(), d.b.end());
  put_uleb(data, actor.size());
  data.insert(data.end(), actor.begin(), actor.end());
  put_uleb(data, seq);
  put_uleb(data, start_op);
  put_sleb(data, timestamp);
  put_uleb(data, message.size());
  data.insert(data.end(), message.begin(), message.end());
  put_uleb(data, other_actors.size());
  for (auto& a : other_actors) {
    put_uleb(data, a.size());
    data.insert(data.end(), a.begin(), a.end());
  }
  write_raw_columns(data, cols.meta);
  data.insert(data.end(), cols.data.begin(), cols.data.end());
  data.insert(data.end(), extra.begin(), extra.end());
  Hash h = chunk_hash(CHUNK_CHANGE, data.data(), data.size());
  Bytes bytes;
  write_chunk(bytes, CHUNK_CHANGE, h.b.data(), data);
  ChunkHeader hdr = parse_chunk_header(bytes.data(), bytes.size());
  StoredChange ch = parse_change_body(bytes.data(), hdr);
  ch.len = ops.size();
  return ch;
}

// Change::compress — storage/change.rs:287-293 + change/compressed.rs:18-35 (len > 256 only).
// Output bytes are zlib's, not flate2's: PARITY UNPINNED for the compressed form.
inline Bytes compress_change(const StoredChange& ch) {
  if (ch.bytes.size() <= 256) return ch.bytes;
  Bytes body = deflate_raw(ch.bytes.data() + ch.header_size, ch.bytes.size() - ch.header_size);
  Bytes out;
  write_chunk(out, CHUNK_COMPRESSED, ch.checksum, body);
  return out;
}

// ---------------------------------------------------------------------------------------------
// document chunk — storage/document.rs:68-212 (parse), document/doc_change_columns.rs
// ---------------------------------------------------------------------------------------------
struct ChangeMeta {  // doc_change_columns.rs:26-35
  uint64_t actor = 0;
  uint64_t seq = 0;
  uint64_t max_op = 0;
  int64_t timestamp = 0;
  std::optional<std::string> message;
  std::vector<uint64_t> deps;
  Bytes extra;
};
struct ChangeColumns {
  ColRange actor, seq, max_op, time, message, deps_num, deps_idx, extra_meta, extra_raw;
};
struct ReadChangeError : std::runtime_error {
  explicit ReadChangeError(const std::string& m) : std::runtime_error(m) {}
};
inline ChangeColumns map_change_columns(const std::vector<LCol>& cols) {  // doc_change_columns.rs:272-338
  ChangeColumns o;
  for (auto& c : cols) {
    auto r = ColRange{c.s, c.e};
    if (c.id == 0 && c.type == T_ACTOR) o.actor = r;
    else if (c.id == 0 && c.type == T_DELTA) o.seq = r;
    else if (c.id == 1 && c.type == T_DELTA) o.max_op = r;
    else if (c.id == 2 && c.type == T_DELTA) o.time = r;
    else if (c.id == 3 && c.type == T_STR) o.message = r;
    else if (c.id == 4 && c.type == T_GROUP) {
      o.deps_num = r;
      if (c.kids.empty()) o.deps_idx = ColRange{};
      else {
        if (c.kids[0].type != T_DELTA) throw MismatchingColumn();
        if (c.kids.size() > 1) throw MismatchingColumn();
        o.deps_idx = ColRange{c.kids[0].s, c.kids[0].e};
      }
    } else if (c.id == 5 && c.type == T_VMETA) { o.extra_meta = r; o.extra_raw = ColRange{c.rs, c.re}; }
  }
  return o;
}

struct ParsedDocument {
  ChunkHeader header;
  std::vector<Bytes> actors;
  std::vector<Hash> heads;
  std::vector<uint64_t> head_indices;
  Bytes change_data, op_data;  // uncompressed column data blocks
  ChangeColumns change_cols;
  OpColumns op_cols;
};

// decompress a column block: compression::decompress (document/compression.rs:51-74) +
// RawColumns::uncompress (raw_column.rs:143-165)
inline void uncompress_block(const uint8_t* data, std::vector<RawCol>& cols, Bytes& out) {
  size_t start = 0;
  for (auto& rc : cols) {
    size_t before = out.size();
    if (rc.deflate()) {
      if (!inflate_raw(data + rc.start, rc.len(), out)) throw ParseFail("deflate error in column");
    } else {
      out.insert(out.end(), data + rc.start, data + rc.end);
    }
    size_t len = out.size() - before;
    rc.spec &= ~8u;
    rc.start = start;
    rc.end = start + len;
    start += len;
  }
}

inline ParsedDocument parse_document_body(const uint8_t* chunk, const ChunkHeader& h) {
  ParsedDocument d;
  d.header = h;
  Cursor c(chunk + h.header_size, h.data_len);
  try {
    uint64_t nactors = read_uleb_strict(c);
    for (uint64_t i = 0; i < nactors; i++) {
      uint64_t l = read_uleb_strict(c);
      if (l > c.remaining()) throw ParseFail("incomplete");
      const uint8_t* p = c.take_n((size_t)l);
      d.actors.emplace_back(p, p + l);
    }
    uint64_t nheads = read_uleb_strict(c);
    for (uint64_t i = 0; i < nheads; i++) {
      Hash hh;
      memcpy(hh.b.data(), c.take_n(32), 32);
      d.heads.push_back(hh);
    }
    std::vector<RawCol> change_meta = parse_raw_columns(c);
    std::vector<RawCol> ops_meta = parse_raw_columns(c);
    size_t clen = total_column_len(change_meta);
    if (clen > c.remaining()) throw ParseFail("incomplete change columns");
    const uint8_t* cdata = c.take_n(clen);
    size_t olen = total_column_len(ops_meta);
    if (olen > c.remaining()) throw ParseFail("incomplete op columns");
    const uint8_t* odata = c.take_n(olen);
    if (!c.done()) {  // optional head-index suffix (document.rs:140-154)
      for (uint64_t i = 0; i < nheads; i++) d.head_indices.push_back(read_uleb_strict(c));
    }
    if (!c.done()) throw Error(E_PARSE, "leftover data in document chunk");  // chunk.rs:72-74
    uncompress_block(cdata, change_meta, d.change_data);
    uncompress_block(odata, ops_meta, d.op_data);
    d.op_cols = map_op_columns(parse_layout(ops_meta, d.op_data.size()), true);
    d.change_cols = map_change_columns(parse_layout(change_meta, d.change_data.size()));
  } catch (DecodeError& e) {
    throw ParseFail(e.what());
  } catch (StrictLebError& e) {
    throw ParseFail(e.what());
  } catch (MismatchingColumn& e) {
    throw ParseFail(e.what());
  }
  return d;
}

// DocChangeColumnIter — doc_change_columns.rs:183-270 (+ DepsIter deps.rs:45-102)
inline std::vector<ChangeMeta> read_change_meta(const ParsedDocument& d) {
  const uint8_t* base = d.change_data.data();
  const ChangeColumns& cc = d.change_cols;
  auto P = [&](const ColRange& r) { return base + r.s; };
  auto N = [&](const ColRange& r) { return r.e - r.s; };
  RleDecoder<uint64_t> actors(P(cc.actor), N(cc.actor)), deps_num(P(cc.deps_num), N(cc.deps_num));
  DeltaDecoder seq(P(cc.seq), N(cc.seq)), max_op(P(cc.max_op), N(cc.max_op)), time(P(cc.time), N(cc.time)),
      deps_idx(P(cc.deps_idx), N(cc.deps_idx));
  bool has_message = !cc.message.empty();
  RleDecoder<std::string> message(P(cc.message), N(cc.message));
  ValueDecoder extra(P(cc.extra_meta), N(cc.extra_meta), P(cc.extra_raw), N(cc.extra_raw));
  std::vector<ChangeMeta> out;
  auto read_deps = [&](std::vector<uint64_t>& v) -> bool {  // false = iterator exhausted
    std::optional<uint64_t> n;
    if (!deps_num.next(n)) return false;
    if (!n) throw DecodeError("unexpected null deps group");
    for (uint64_t i = 0; i < *n; i++) {
      std::optional<int64_t> e;
      if (!deps_idx.next(e) || !e) throw DecodeError("unexpected null deps");
      if (*e < 0) throw DecodeError("negative dep index");
      v.push_back((uint64_t)*e);
    }
    return true;
  };
  for (;;) {
    ChangeMeta m;
    std::optional<uint64_t> a;
    bool ha = actors.next(a);
    if (!ha || !a) {
      // check_done (:243-253): "done" if ANY of the other iterators is exhausted
      std::optional<int64_t> t;
      std::vector<uint64_t> dv;
      bool s_none = !seq.next(t);
      bool m_none = !max_op.next(t);
      bool t_none = !time.next(t);
      bool d_none = !read_deps(dv);
      if (s_none || m_none || t_none || d_none) return out;
      throw ReadChangeError("unexpected null actor");
    }
    m.actor = *a;
    std::optional<int64_t> v;
    if (!seq.next(v) || !v) throw DecodeError("unexpected null seq");
    if (*v < 0) throw DecodeError("negative seq");
    m.seq = (uint64_t)*v;
    if (!max_op.next(v) || !v) throw DecodeError("unexpected null max_op");
    if (*v < 0) throw DecodeError("negative max_op");
    m.max_op = (uint64_t)*v;
    if (!time.next(v) || !v) throw DecodeError("unexpected null time");
    m.timestamp = *v;
    if (has_message) {
      std::optional<std::string> s;
      if (message.next(s)) m.message = s;
    }
    if (!read_deps(m.deps)) throw DecodeError("unexpected null deps");
    Scalar ex;
    if (extra.next(ex)) {
      if (ex.tag != 7) throw ReadChangeError("invalid extra bytes");
      m.extra = ex.bytes;
    }
    out.push_back(std::move(m));
  }
}

// DocChangeColumns::encode — doc_change_columns.rs:87-162
inline void encode_change_columns(const std::vector<ChangeMeta>& changes,
                                  std::vector<std::pair<uint32_t, size_t>>& meta, Bytes& data) {
  RleEncoder<uint64_t> actor, deps_num, extra_meta;
  DeltaEncoder seq, max_op, time, deps_idx;
  RleEncoder<std::string> message;
  Bytes extra_raw;
  for (auto& c : changes) {
    actor.append_value(c.actor);
    seq.append_value((int64_t)c.seq);
    max_op.append_value((int64_t)c.max_op);
    time.append_value(c.timestamp);
    message.append(c.message);
    deps_num.append_value(c.deps.size());
    for (auto d : c.deps) deps_idx.append_value((int64_t)d);
    Scalar ex = sc_bytes(c.extra);
    extra_meta.append_value(value_meta(ex));
    encode_val(extra_raw, ex);
  }
  auto add = [&](uint32_t id, int type, Bytes b) {
    meta.push_back({(id << 4) | (uint32_t)type, b.size()});
    data.insert(data.end(), b.begin(), b.end());
  };
  add(0, T_ACTOR, actor.finish());
  add(0, T_DELTA, seq.finish());
  add(1, T_DELTA, max_op.finish());
  add(2, T_DELTA, time.finish());
  add(3, T_STR, message.finish());
  add(4, T_GROUP, deps_num.finish());
  add(4, T_DELTA, deps_idx.finish());
  add(5, T_VMETA, extra_meta.finish());
  add(5, T_VRAW, extra_raw);
}

// Document::new — storage/document.rs:214-309 (uncompressed form) and the DEFLATE pass of
// document/compression.rs:29-49,346-355 + raw_column.rs:49-62 (columns >= threshold).
// Compressed bytes: PARITY UNPINNED (zlib vs flate2).
inline Bytes build_document(std::vector<Bytes> actors, const std::vector<std::pair<Hash, uint64_t>>& heads_with_indices,
                            const std::vector<StoredOp>& ops, const std::vector<ChangeMeta>& changes, bool deflate) {
  EncodedOpCols oc = encode_op_columns(ops, true);
  std::vector<std::pair<uint32_t, size_t>> cmeta;
  Bytes cdata;
  encode_change_columns(changes, cmeta, cdata);
  std::sort(actors.begin(), actors.end());
  if (deflate) {
    auto squeeze = [](std::vector<std::pair<uint32_t, size_t>>& meta, Bytes& data) {
      Bytes out;
      size_t off = 0;
      for (auto& m : meta) {
        size_t len = m.second;
        if (len >= 256) {
          Bytes z = deflate_raw(data.data() + off, len);
          out.insert(out.end(), z.begin(), z.end());
          m.first |= 8;
          m.second = z.size();
        } else {
          out.insert(out.end(), data.begin() + off, data.begin() + off + len);
        }
        off += len;
      }
      data.swap(out);
    };
    squeeze(cmeta, cdata);
    squeeze(oc.meta, oc.data);
  }
  Bytes data;
  put_uleb(data, actors.size());
  for (auto& a : actors) {
    put_uleb(data, a.size());
    data.insert(data.end(), a.begin(), a.end());
  }
  put_uleb(data, heads_with_indices.size());
  for (auto& h : heads_with_indices) data.insert(data.end(), h.first.b.begin(), h.first.b.end());
  write_raw_columns(data, cmeta);
  write_raw_columns(data, oc.meta);
  data.insert(data.end(), cdata.begin(), cdata.end());
  data.insert(data.end(), oc.data.begin(), oc.data.end());
  for (auto& h : heads_with_indices) put_uleb(data, h.second);
  Hash hash = chunk_hash(CHUNK_DOC, data.data(), data.size());
  Bytes bytes;
  write_chunk(bytes, CHUNK_DOC, hash.b.data(), data);
  return bytes;
}

}  // namespace amo
