// Device-side building blocks of the batched load path: LEB128 readers/writers, the RLE / delta /
// boolean run decoders, UTF-8 validation and streaming SHA-256.  All functions work on the batch
// arena in HBM through plain pointers and carry their own bounds.
//
// AMH_EMU: the same per-thread bodies can be compiled for the host so that the CPU-only test tier
// (`-m "not gpu"`) can exercise the kernel logic where no GPU exists.  That build is test
// infrastructure (tests/emu), is never part of libautomerge_hip.so and is never loaded by the package.
// AMH_HOST_HELPERS: the LEB / SHA-256 helpers alone, for host-side ingest (inflate + re-checksum).
#pragma once
#include "amh_types.h"

#if defined(AMH_EMU) || defined(AMH_HOST_HELPERS)
#include <string.h>
#define AMH_DEV static inline
#define AMH_DEV_NOINLINE static
static inline u32 amh_atomic_add(u32* p, u32 v) { u32 o = *p; *p = o + v; return o; }
static inline void amh_atomic_add_i64(i64* p, i64 v) { *p += v; }
static inline u32 amh_atomic_min(u32* p, u32 v) { u32 o = *p; if (v < o) *p = v; return o; }
static inline u32 amh_atomic_max(u32* p, u32 v) { u32 o = *p; if (v > o) *p = v; return o; }
static inline u32 amh_atomic_cas(u32* p, u32 cmp, u32 val) { u32 o = *p; if (o == cmp) *p = val; return o; }
static inline u32 amh_atomic_exch(u32* p, u32 val) { u32 o = *p; *p = val; return o; }
static inline void amh_atomic_or(u32* p, u32 v) { *p |= v; }
static inline void amh_atomic_inc_grouped(u32* base, u32 idx) { base[idx]++; }
static inline u64 amh_atomic_cas64(u64* p, u64 cmp, u64 val) { u64 o = *p; if (o == cmp) *p = val; return o; }
static inline i32 amh_atomic_cas_i32(i32* p, i32 cmp, i32 val) { i32 o = *p; if (o == cmp) *p = val; return o; }
static inline u64 amh_load64(const u64* p) { return *p; }
static inline void amh_store64(u64* p, u64 v) { *p = v; }
#else
#include <hip/hip_runtime.h>
#define AMH_DEV __device__ __forceinline__
// out-of-line helpers: keep the callers' control flow small (a large fully inlined decoder made the
// gfx950 backend drop a value on one path of the string decoder)
#define AMH_DEV_NOINLINE __device__ __noinline__ static
AMH_DEV u32 amh_atomic_add(u32* p, u32 v) { return atomicAdd(p, v); }
AMH_DEV void amh_atomic_add_i64(i64* p, i64 v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
AMH_DEV u32 amh_atomic_min(u32* p, u32 v) { return atomicMin(p, v); }
AMH_DEV u32 amh_atomic_max(u32* p, u32 v) { return atomicMax(p, v); }
AMH_DEV u32 amh_atomic_cas(u32* p, u32 cmp, u32 val) { return atomicCAS(p, cmp, val); }
AMH_DEV u32 amh_atomic_exch(u32* p, u32 val) { return atomicExch(p, val); }
AMH_DEV void amh_atomic_or(u32* p, u32 v) { atomicOr(p, v); }
// base[idx] += 1 where many lanes of a wave name the same few words (ops of one document counting into its changes):
// the lanes that share an index send one atomic with their number instead of one each
AMH_DEV void amh_atomic_inc_grouped(u32* base, u32 idx) {
  const u32 lane = __lane_id();
  for (;;) {
    u32 first = (u32)__builtin_amdgcn_readfirstlane((int)idx);
    unsigned long long same = __ballot(idx == first);
    if (idx == first) {
      if (lane == (u32)__builtin_ctzll(same)) atomicAdd(base + idx, (u32)__builtin_popcountll(same));
      break;
    }
  }
}
AMH_DEV u64 amh_atomic_cas64(u64* p, u64 cmp, u64 val) {
  return atomicCAS((unsigned long long*)p, (unsigned long long)cmp, (unsigned long long)val);
}
AMH_DEV i32 amh_atomic_cas_i32(i32* p, i32 cmp, i32 val) { return atomicCAS(p, cmp, val); }
// whole 8-byte words that other threads of the same launch rewrite (pointer jumping over packed pairs): never torn
AMH_DEV u64 amh_load64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
AMH_DEV void amh_store64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
// a list too long for its thread: recorded for the block-per-list kernels (device build; the emulation sorts in place)
AMH_DEV void big_push(u32* n, u32* items, u32 cap, u32 a, u32 b) {
  u32 i = amh_atomic_add(n, 1);
  if (i < cap) { items[2 * (size_t)i] = a; items[2 * (size_t)i + 1] = b; }
}


// first error wins; any non-zero status makes every later stage skip the document
AMH_DEV void amh_fail_at(const BatchDev& B, u32 doc, i32 code, u32 line) {
  if (amh_atomic_cas_i32(&B.status[doc], 0, code) == 0) B.fail_line[doc] = line;  // where in amh_kernels.inl: diagnostics
}
#define amh_fail(B, doc, code) amh_fail_at(B, doc, code, __LINE__)

// ---------------------------------------------------------------------------------------------
// LEB128.  lenient = what the reference's column decoders accept (the `leb128` crate:
// decodable_impls.rs:58-96); strict = storage/parse/leb128.rs:15-96 (chunk headers, value payloads).
// All readers return false on malformed / truncated input and never read at or past `end`.
// ---------------------------------------------------------------------------------------------
// Fast path: the next 8 stream bytes are fetched with three aligned 32-bit loads (one memory round
// trip instead of one per byte) and the number is decoded in registers.  `p + (pos & ~3)` must be
// 4-byte aligned and 12 bytes readable — true for the batch arena (16-byte aligned chunks, padded
// tail) and for the LDS windows over it.  Numbers longer than 8 bytes take the byte loop.
AMH_DEV u64 amh_peek8(const u8* p, u64 pos) {
  const u32* w = (const u32*)(p + (pos & ~(u64)3));
  u32 w0 = w[0], w1 = w[1], w2 = w[2];
  u32 sh = (u32)(pos & 3) * 8;
  u64 lo = (u64)w0 | ((u64)w1 << 32);
  return sh ? (lo >> sh) | ((u64)w2 << (64 - sh)) : lo;
}
// number of bytes of the LEB128 number at the start of x (1..8), 0 if none of the first `avail` (<= 8) bytes ends it
AMH_DEV u32 amh_leb_len8(u64 x, u64 avail) {
  u64 t = ~x & 0x8080808080808080ull;
  if (avail < 8) t &= ((u64)1 << (8 * avail)) - 1;
  if (t == 0) return 0;
  return (u32)(__builtin_ctzll(t) >> 3) + 1;
}
AMH_DEV u64 amh_leb_groups(u64 x, u32 nbytes) {  // the 7-bit groups of the first nbytes bytes, packed
  if (nbytes < 8) x &= ((u64)1 << (8 * nbytes)) - 1;
  x = (x & 0x007f007f007f007full) | ((x & 0x7f007f007f007f00ull) >> 1);
  x = (x & 0x00003fff00003fffull) | ((x & 0x3fff00003fff0000ull) >> 2);
  x = (x & 0x000000000fffffffull) | ((x & 0x0fffffff00000000ull) >> 4);
  return x;
}
AMH_DEV bool rd_uleb_lenient_slow(const u8* p, u64& pos, u64 end, u64& out) {
  u64 result = 0;
  u32 shift = 0;
  for (;;) {
    if (pos >= end) return false;
    u8 b = p[pos++];
    if (shift == 63 && b != 0x00 && b != 0x01) return false;  // overflow
    result |= (u64)(b & 0x7f) << shift;
    if (!(b & 0x80)) { out = result; return true; }
    shift += 7;
  }
}
AMH_DEV bool rd_sleb_lenient_slow(const u8* p, u64& pos, u64 end, i64& out) {
  u64 result = 0;
  u32 shift = 0;
  u8 b;
  for (;;) {
    if (pos >= end) return false;
    b = p[pos++];
    if (shift == 63 && b != 0x00 && b != 0x7f) return false;
    result |= (u64)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) break;
  }
  if (shift < 64 && (b & 0x40)) result |= ~(u64)0 << shift;
  out = (i64)result;
  return true;
}
AMH_DEV bool rd_uleb_lenient(const u8* p, u64& pos, u64 end, u64& out) {
  if (pos >= end) return false;
  u64 x = amh_peek8(p, pos);
  u32 nb = amh_leb_len8(x, end - pos);
  if (nb == 0) { if (end - pos < 8) return false; return rd_uleb_lenient_slow(p, pos, end, out); }
  out = amh_leb_groups(x, nb);
  pos += nb;
  return true;
}
AMH_DEV bool rd_sleb_lenient(const u8* p, u64& pos, u64 end, i64& out) {
  if (pos >= end) return false;
  u64 x = amh_peek8(p, pos);
  u32 nb = amh_leb_len8(x, end - pos);
  if (nb == 0) { if (end - pos < 8) return false; return rd_sleb_lenient_slow(p, pos, end, out); }
  u64 r = amh_leb_groups(x, nb);
  u32 bits = 7 * nb;
  if (r & ((u64)1 << (bits - 1))) r |= ~(u64)0 << bits;   // sign bit = bit 6 of the last byte (bits <= 56)
  out = (i64)r;
  pos += nb;
  return true;
}
// 0 ok, 1 error
AMH_DEV int rd_uleb_strict(const u8* p, u64& pos, u64 end, u64& out) {
  u64 res = 0;
  u32 shift = 0;
  for (;;) {
    if (pos >= end) return 1;
    u8 b = p[pos++];
    if (shift < 64) res |= (u64)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) {
      if (shift > 64 && b > 1) return 1;
      if (shift > 7 && b == 0) return 1;
      out = res;
      return 0;
    } else if (shift > 64) {
      return 1;
    }
  }
}
AMH_DEV int rd_sleb_strict(const u8* p, u64& pos, u64 end, i64& out) {
  u64 res = 0;
  u32 shift = 0;
  u8 prev = 0;
  for (;;) {
    if (pos >= end) return 1;
    u8 b = p[pos++];
    if (shift < 64) res |= (u64)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) {
      if (shift > 64 && b != 0 && b != 0x7f) return 1;
      if (shift > 7 && ((b == 0 && !(prev & 0x40)) || (b == 0x7f && (prev & 0x40)))) return 1;
      if (shift < 64 && (b & 0x40)) res |= ~(u64)0 << shift;
      out = (i64)res;
      return 0;
    } else if (shift > 64) {
      return 1;
    }
    prev = b;
  }
}
AMH_DEV u32 uleb_size(u64 v) {
  u32 n = 1;
  while (v >>= 7) n++;
  return n;
}
AMH_DEV u32 sleb_size(i64 v) {
  u32 n = 0;
  for (;;) {
    u8 b = v & 0x7f;
    v >>= 7;
    n++;
    if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) return n;
  }
}
// writers: `out` may be null (size-only pass)
AMH_DEV u32 wr_uleb(u8* out, u64 v) {
  u32 n = 0;
  do {
    u8 b = v & 0x7f;
    v >>= 7;
    if (v) b |= 0x80;
    if (out) out[n] = b;
    n++;
  } while (v);
  return n;
}
AMH_DEV u32 wr_sleb(u8* out, i64 v) {
  u32 n = 0;
  for (;;) {
    u8 b = v & 0x7f;
    v >>= 7;
    bool done = (v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40));
    if (!done) b |= 0x80;
    if (out) out[n] = b;
    n++;
    if (done) return n;
  }
}

AMH_DEV u64 amh_peek8_any(const u8* p) {   // 8 bytes at any address; reads at most 11 bytes past p (the arena is padded)
  u64 a = (u64)p;
  const u32* w = (const u32*)(a & ~(u64)3);
  u32 w0 = w[0], w1 = w[1], w2 = w[2];
  u32 sh = (u32)(a & 3) * 8;
  u64 lo = (u64)w0 | ((u64)w1 << 32);
  return sh ? (lo >> sh) | ((u64)w2 << (64 - sh)) : lo;
}

// Rust str::from_utf8 acceptance
AMH_DEV_NOINLINE bool amh_valid_utf8(const u8* s, u64 n) {
  u64 i = 0;
  // ASCII, eight bytes per step (nearly every key and most values); the first byte with its high bit set starts the
  // general loop at a character boundary
  while (i < n) {
    u64 w = amh_peek8_any(s + i);
    u64 r = n - i;
    if (r < 8) w &= ((u64)1 << (8 * r)) - 1;
    if (w & 0x8080808080808080ull) break;
    i += r < 8 ? r : 8;
  }
  while (i < n) {
    u8 b = s[i];
    if (b < 0x80) { i++; continue; }
    if (b >= 0xc2 && b <= 0xdf) {
      if (i + 1 >= n || (s[i + 1] & 0xc0) != 0x80) return false;
      i += 2;
    } else if (b >= 0xe0 && b <= 0xef) {
      if (i + 2 >= n) return false;
      u8 b1 = s[i + 1], b2 = s[i + 2];
      if ((b1 & 0xc0) != 0x80 || (b2 & 0xc0) != 0x80) return false;
      if (b == 0xe0 && b1 < 0xa0) return false;
      if (b == 0xed && b1 >= 0xa0) return false;
      i += 3;
    } else if (b >= 0xf0 && b <= 0xf4) {
      if (i + 3 >= n) return false;
      u8 b1 = s[i + 1], b2 = s[i + 2], b3 = s[i + 3];
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
AMH_DEV int amh_memcmp(const u8* a, u32 an, const u8* b, u32 bn) {
  u32 n = an < bn ? an : bn;
  for (u32 i = 0; i < n; i++)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return an < bn ? -1 : (an > bn ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// run decoders (columnar/encoding/rle.rs:163-239, delta.rs:53-95, boolean.rs:60-101).
// kind: 0 = u64 values, 1 = i64 values, 2 = strings (value = offset of the bytes, aux = length)
// next(): 0 = value produced, 1 = exhausted, -1 = malformed
// ---------------------------------------------------------------------------------------------
struct RleDec {
  const u8* p;    // where the LEB headers / values are read: the arena, or an LDS window over it
  const u8* gp;   // the arena itself (string payloads are validated there)
  u64 pos, end;
  i64 count;
  u64 last;
  u32 last_aux;
  bool last_null;
  bool literal;
};
AMH_DEV void rle_init(RleDec& d, const u8* p, u64 off, u64 len) {
  d.p = p; d.gp = p; d.pos = off; d.end = off + len; d.count = 0; d.last = 0; d.last_aux = 0; d.last_null = true; d.literal = false;
}
AMH_DEV bool rle_done(const RleDec& d) { return d.pos >= d.end && d.count == 0; }
template <int KIND>
AMH_DEV bool rle_read_value(RleDec& d, u64& v, u32& aux) {
  if (KIND == 0) return rd_uleb_lenient(d.p, d.pos, d.end, v);
  if (KIND == 1) { i64 s; if (!rd_sleb_lenient(d.p, d.pos, d.end, s)) return false; v = (u64)s; return true; }
  u64 len;
  if (!rd_uleb_lenient(d.p, d.pos, d.end, len)) return false;
  if (len > d.end - d.pos) return false;
  if (!amh_valid_utf8(d.gp + d.pos, len)) return false;
  v = d.pos;
  aux = (u32)len;
  d.pos += len;
  return true;
}
template <int KIND>
AMH_DEV int rle_next(RleDec& d, u64& v, u32& aux, bool& is_null) {
  while (d.count == 0) {
    if (d.pos >= d.end) return 1;
    i64 n;
    if (!rd_sleb_lenient(d.p, d.pos, d.end, n)) return -1;
    if (n > 0) {
      d.count = n;
      u64 lv = 0; u32 la = 0;
      if (!rle_read_value<KIND>(d, lv, la)) return -1;
      d.last = lv; d.last_aux = la;
      d.last_null = false;
      d.literal = false;
    } else if (n < 0) {
      d.count = (n == INT64_MIN) ? INT64_MAX : -n;
      d.literal = true;
    } else {
      u64 nulls;
      if (!rd_uleb_lenient(d.p, d.pos, d.end, nulls)) return -1;
      if (nulls > (u64)INT64_MAX) return -1;
      d.count = (i64)nulls;
      d.last_null = true;
      d.literal = false;
    }
  }
  d.count--;
  if (d.literal) {
    u64 lv = 0; u32 la = 0;
    if (!rle_read_value<KIND>(d, lv, la)) return -1;
    v = lv; aux = la;
    is_null = false;
  } else {
    v = d.last;
    aux = d.last_aux;
    is_null = d.last_null;
  }
  return 0;
}
struct DeltaDec {
  RleDec r;
  i64 abs;
};
AMH_DEV void delta_init(DeltaDec& d, const u8* p, u64 off, u64 len) { rle_init(d.r, p, off, len); d.abs = 0; }
AMH_DEV int delta_next(DeltaDec& d, i64& v, bool& is_null) {
  u64 raw;
  u32 aux;
  int rc = rle_next<1>(d.r, raw, aux, is_null);
  if (rc) return rc;
  if (!is_null) {
    i64 delta = (i64)raw, r;
    if (__builtin_add_overflow(d.abs, delta, &r)) r = delta > 0 ? INT64_MAX : INT64_MIN;  // saturating_add
    d.abs = r;
    v = r;
  }
  return 0;
}
struct BoolDec {
  const u8* p;
  u64 pos, end;
  u64 count;
  bool last;
};
AMH_DEV void bool_init(BoolDec& d, const u8* p, u64 off, u64 len) { d.p = p; d.pos = off; d.end = off + len; d.count = 0; d.last = true; }
AMH_DEV int bool_next(BoolDec& d, bool& v) {
  while (d.count == 0) {
    if (d.pos >= d.end) return 1;
    if (!rd_uleb_lenient(d.p, d.pos, d.end, d.count)) return -1;
    d.last = !d.last;
  }
  d.count--;
  v = d.last;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// streaming SHA-256 (FIPS 180-4; the reference hashes type || uleb(len) || data, chunk.rs:286-293)
// ---------------------------------------------------------------------------------------------
struct Sha256Dev {
  u32 h[8];
  u32 w[16];   // pending block as big-endian words
  u32 fill;    // bytes in w
  u64 total;
};
AMH_DEV u32 sha_rotr(u32 x, u32 n) { return (x >> n) | (x << (32 - n)); }
#if defined(AMH_EMU) || defined(AMH_HOST_HELPERS)
static const u32 AMH_SHA_K[64] = {
#else
__device__ __constant__ static const u32 AMH_SHA_K[64] = {
#endif
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

AMH_DEV void sha_init(Sha256Dev& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  s.fill = 0;
  s.total = 0;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
}
AMH_DEV void sha_block(Sha256Dev& s) {
  u32 w[16];
  for (int i = 0; i < 16; i++) w[i] = s.w[i];
  u32 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
  for (int i = 0; i < 64; i++) {
    u32 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
      u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + AMH_SHA_K[i] + wi;
    u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
}
AMH_DEV void sha_byte(Sha256Dev& s, u8 b) {
  s.w[s.fill >> 2] |= (u32)b << (24 - 8 * (s.fill & 3));
  s.fill++;
  if (s.fill == 64) sha_block(s);
}
AMH_DEV void sha_update(Sha256Dev& s, const u8* p, u64 n) {
  s.total += n;
  u64 i = 0;
  while (i < n && (s.fill & 3)) sha_byte(s, p[i++]);
  while (n - i >= 4) {
    u32 word = ((u32)p[i] << 24) | ((u32)p[i + 1] << 16) | ((u32)p[i + 2] << 8) | (u32)p[i + 3];
    s.w[s.fill >> 2] = word;
    s.fill += 4;
    i += 4;
    if (s.fill == 64) sha_block(s);
  }
  while (i < n) sha_byte(s, p[i++]);
}
AMH_DEV void sha_finish(Sha256Dev& s, u8* out32) {
  u64 bits = s.total * 8;
  sha_byte(s, 0x80);
  while (s.fill != 56) sha_byte(s, 0);
  for (int i = 0; i < 8; i++) sha_byte(s, (u8)(bits >> (56 - 8 * i)));
  for (int i = 0; i < 8; i++) {
    out32[4 * i] = (u8)(s.h[i] >> 24);
    out32[4 * i + 1] = (u8)(s.h[i] >> 16);
    out32[4 * i + 2] = (u8)(s.h[i] >> 8);
    out32[4 * i + 3] = (u8)s.h[i];
  }
}

// ---------------------------------------------------------------------------------------------
// SHA-256 of one contiguous byte range, block at a time out of registers: sixteen message words per
// block come from aligned 32-bit loads funnel-shifted to the range's byte phase (no byte loads, no
// indexed scratch buffer).  Reads at most 67 bytes past the range (the arenas are padded).
// ---------------------------------------------------------------------------------------------
AMH_DEV u32 amh_bswap32(u32 x) { return __builtin_bswap32(x); }
struct ShaState { u32 h[8]; };
AMH_DEV void sha_compress16(ShaState& s, u32 (&w)[16]) {
  u32 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    u32 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
      u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + AMH_SHA_K[i] + wi;
    u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}
AMH_DEV void sha256_contig(const u8* p, u64 n, u8* out32) {
  ShaState s;
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  u64 addr = (u64)p;
  const u32* base = (const u32*)(addr & ~(u64)3);
  u32 sh = (u32)(addr & 3) * 8;
  u64 full = n / 64;
  u32 w[16];
  for (u64 k = 0; k < full; k++) {
    const u32* m = base + 16 * k;
    u32 prev = m[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      u32 next = m[i + 1];
      u32 le = sh ? (prev >> sh) | (next << (32 - sh)) : prev;
      w[i] = amh_bswap32(le);
      prev = next;
    }
    sha_compress16(s, w);
  }
  // tail: the remaining r < 64 bytes, 0x80, zeros, 64-bit big-endian bit count (one or two blocks)
  u32 r = (u32)(n & 63);
  const u32* m = base + 16 * full;
  u64 bits = n * 8;
  int nblk = r >= 56 ? 2 : 1;
  for (int blk = 0; blk < nblk; blk++) {
    u32 prev = m[16 * blk];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      u32 next = m[16 * blk + i + 1];
      u32 le = sh ? (prev >> sh) | (next << (32 - sh)) : prev;
      prev = next;
      u32 o = (u32)(64 * blk + 4 * i);   // byte offset of this word inside the tail region
      u32 be = amh_bswap32(le);
      u32 word;
      if (o + 4 <= r) word = be;
      else if (o >= r + 1) word = 0;
      else {  // the word holding the 0x80 terminator (and possibly the last message bytes)
        u32 keep = r - o;   // message bytes in this word: 0..3
        u32 mask = keep ? 0xFFFFFFFFu << (32 - 8 * keep) : 0;
        word = (be & mask) | (0x80u << (24 - 8 * keep));
      }
      if (blk == nblk - 1 && i == 14) word = (u32)(bits >> 32);
      if (blk == nblk - 1 && i == 15) word = (u32)bits;
      w[i] = word;
    }
    sha_compress16(s, w);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    out32[4 * i] = (u8)(s.h[i] >> 24);
    out32[4 * i + 1] = (u8)(s.h[i] >> 16);
    out32[4 * i + 2] = (u8)(s.h[i] >> 8);
    out32[4 * i + 3] = (u8)s.h[i];
  }
}
