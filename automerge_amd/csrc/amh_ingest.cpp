// Host ingest of one input buffer: documents whose columns carry the DEFLATE bit are inflated and
// re-serialised in uncompressed form before upload, exactly what the reference does on parse
// (storage/document/compression.rs:51-74, storage/columns/raw_column.rs:76-89).  Any conforming
// inflater yields the same bytes, so this is not a device stage.
//
// The stored checksum covers the bytes as stored (compressed); it is verified here on the host and
// the rewritten chunk gets the checksum of its new body, or a deliberately wrong one when the
// original did not verify, so that the device's checksum stage reports BadChecksum for it.
#define AMH_HOST_HELPERS
#include "amh_ingest.h"

#include <zlib.h>

#include "amh_dev.h"

static bool inflate_raw(const u8* src, size_t n, std::vector<u8>& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  u8 buf[16384];
  int rc;
  do {
    zs.next_out = buf;
    zs.avail_out = sizeof(buf);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { inflateEnd(&zs); return false; }
    out.insert(out.end(), buf, buf + (sizeof(buf) - zs.avail_out));
    if (rc == Z_BUF_ERROR && zs.avail_in == 0) break;
  } while (rc != Z_STREAM_END);
  inflateEnd(&zs);
  return rc == Z_STREAM_END;
}

// SHA-256 of large chunks runs on the host (one message is a serial chain: a GPU lane hashes 16-20 MB/s).  Where the
// CPU has the SHA extensions (every EPYC / recent Xeon) the compression function is the sha256rnds2 / sha256msg1 /
// sha256msg2 sequence: ~1.5-2 GB/s per thread instead of ~0.4 with the portable rounds below.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_ni_blocks(u32 state[8], const u8* data, size_t nblocks) {
  static const u32 K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i tmp = _mm_loadu_si128((const __m128i*)&state[0]);      // a b c d
  __m128i st1 = _mm_loadu_si128((const __m128i*)&state[4]);      // e f g h
  tmp = _mm_shuffle_epi32(tmp, 0xB1);                             // c d a b
  st1 = _mm_shuffle_epi32(st1, 0x1B);                             // h g f e
  __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                     // a b e f
  st1 = _mm_blend_epi16(st1, tmp, 0xF0);                          // c d g h
  while (nblocks--) {
    const __m128i save0 = st0, save1 = st1;
    __m128i m[4];
    for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * i)), bswap);
    for (int r = 0; r < 16; r++) {   // four rounds per step
      __m128i w = m[r & 3];
      __m128i msg = _mm_add_epi32(w, _mm_loadu_si128((const __m128i*)&K[4 * r]));
      st1 = _mm_sha256rnds2_epu32(st1, st0, msg);
      msg = _mm_shuffle_epi32(msg, 0x0E);
      st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
      if (r < 12) {   // schedule the words of step r + 4 into the slot this step just used
        __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
        x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
        m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
      }
    }
    st0 = _mm_add_epi32(st0, save0);
    st1 = _mm_add_epi32(st1, save1);
    data += 64;
  }
  tmp = _mm_shuffle_epi32(st0, 0x1B);                             // f e b a
  st1 = _mm_shuffle_epi32(st1, 0xB1);                             // d c h g
  st0 = _mm_blend_epi16(tmp, st1, 0xF0);                          // d c b a
  st1 = _mm_alignr_epi8(st1, tmp, 8);                             // h g f e
  _mm_storeu_si128((__m128i*)&state[0], st0);
  _mm_storeu_si128((__m128i*)&state[4], st1);
}
static bool sha_ni_ok() {
  static const bool ok = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
  return ok;
}
static void sha256_ni(const u8* a, size_t an, const u8* b, size_t bn, u8* out32) {
  u32 st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  u8 buf[128];
  size_t fill = 0;
  const u8* parts[2] = {a, b};
  size_t lens[2] = {an, bn};
  for (int k = 0; k < 2; k++) {
    const u8* p = parts[k];
    size_t n = lens[k];
    if (fill) {
      size_t take = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { sha256_ni_blocks(st, buf, 1); fill = 0; }
    }
    if (n >= 64) { sha256_ni_blocks(st, p, n / 64); p += n & ~(size_t)63; n &= 63; }
    if (n) { memcpy(buf + fill, p, n); fill += n; }
  }
  const u64 bits = ((u64)an + (u64)bn) * 8;
  buf[fill++] = 0x80;
  size_t padto = fill <= 56 ? 64 : 128;
  memset(buf + fill, 0, padto - fill);
  for (int i = 0; i < 8; i++) buf[padto - 1 - i] = (u8)(bits >> (8 * i));
  sha256_ni_blocks(st, buf, padto / 64);
  for (int i = 0; i < 8; i++) { out32[4 * i] = (u8)(st[i] >> 24); out32[4 * i + 1] = (u8)(st[i] >> 16); out32[4 * i + 2] = (u8)(st[i] >> 8); out32[4 * i + 3] = (u8)st[i]; }
}
#endif

void amh_host_sha256(const u8* a, size_t an, const u8* b, size_t bn, u8* out32) {
#if defined(__x86_64__)
  if (an + bn >= 4096 && sha_ni_ok()) { sha256_ni(a, an, b, bn, out32); return; }
#endif
  Sha256Dev s;
  sha_init(s);
  sha_update(s, a, an);
  sha_update(s, b, bn);
  sha_finish(s, out32);
}

static void push_uleb(std::vector<u8>& o, u64 v) {
  u8 tmp[10];
  u32 n = wr_uleb(tmp, v);
  o.insert(o.end(), tmp, tmp + n);
}

// Returns true and fills `out` when the buffer is a single document chunk with deflated columns that
// inflated cleanly; false = leave the input as it is (the device path will judge it).
bool amh_ingest_normalise(const u8* p, size_t len, std::vector<u8>& out) {
  if (len < 10 || p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83 || p[8] != 0) return false;
  u64 pos = 9, end = len, dlen;
  if (rd_uleb_strict(p, pos, end, dlen) || dlen > end - pos) return false;
  end = pos + dlen;   // chunks behind the document chunk (trailing changes) are carried over as they are
  u64 data_off = pos;
  u64 n;
  if (rd_uleb_strict(p, pos, end, n)) return false;
  for (u64 i = 0; i < n; i++) {
    u64 al;
    if (rd_uleb_strict(p, pos, end, al) || al > end - pos) return false;
    pos += al;
  }
  if (rd_uleb_strict(p, pos, end, n) || n > (end - pos) / 32) return false;
  pos += 32 * n;
  u64 prefix_end = pos;
  struct Col { u32 spec; u64 len; };
  std::vector<Col> cols[2];
  bool any_deflate = false;
  for (int w = 0; w < 2; w++) {
    u64 nc;
    if (rd_uleb_strict(p, pos, end, nc) || nc > end - pos) return false;
    for (u64 i = 0; i < nc; i++) {
      u64 spec, cl;
      if (rd_uleb_strict(p, pos, end, spec) || spec > 0xFFFFFFFFull || rd_uleb_strict(p, pos, end, cl)) return false;
      if (cl > end) return false;
      if (spec & 8) any_deflate = true;
      cols[w].push_back(Col{(u32)spec, cl});
    }
  }
  if (!any_deflate) return false;
  std::vector<u8> data[2];
  std::vector<Col> ncols[2];
  for (int w = 0; w < 2; w++) {
    for (auto& c : cols[w]) {
      if (c.len > end - pos) return false;
      size_t before = data[w].size();
      if (c.spec & 8) {
        if (!inflate_raw(p + pos, c.len, data[w])) return false;  // reference: ParseError::Deflate; the device sees the deflate bit and rejects
      } else {
        data[w].insert(data[w].end(), p + pos, p + pos + c.len);
      }
      ncols[w].push_back(Col{c.spec & ~8u, data[w].size() - before});
      pos += c.len;
    }
  }
  std::vector<u8> body(p + data_off, p + prefix_end);
  for (int w = 0; w < 2; w++) {
    push_uleb(body, ncols[w].size());
    for (auto& c : ncols[w]) { push_uleb(body, c.spec); push_uleb(body, c.len); }
  }
  body.insert(body.end(), data[0].begin(), data[0].end());
  body.insert(body.end(), data[1].begin(), data[1].end());
  body.insert(body.end(), p + pos, p + end);  // head-index suffix
  // verify the stored checksum over the stored (compressed) bytes
  u8 h[32];
  amh_host_sha256(p + 8, data_off - 8, p + data_off, dlen, h);
  bool ok = memcmp(h, p + 4, 4) == 0;
  std::vector<u8> pre;
  pre.push_back(0);
  push_uleb(pre, body.size());
  amh_host_sha256(pre.data(), pre.size(), body.data(), body.size(), h);
  if (!ok) h[0] ^= 0xFF;
  out.clear();
  out.insert(out.end(), p, p + 4);
  out.insert(out.end(), h, h + 4);
  out.insert(out.end(), pre.begin(), pre.end());
  out.insert(out.end(), body.begin(), body.end());
  out.insert(out.end(), p + end, p + len);
  return true;
}

// A compressed change chunk (type 2) as the type-1 chunk it stands for: same checksum, inflated body
// (chunk.rs:77-97, 106-108).  false = not a type-2 chunk or the body does not inflate.
bool amh_ingest_inflate_change(const u8* p, size_t len, std::vector<u8>& out) {
  if (len < 10 || p[8] != 2) return false;
  u64 pos = 9, end = len, dlen;
  if (rd_uleb_strict(p, pos, end, dlen) || dlen != end - pos) return false;
  std::vector<u8> body;
  if (!inflate_raw(p + pos, dlen, body)) return false;
  out.clear();
  out.insert(out.end(), p, p + 8);
  out.push_back(1);
  push_uleb(out, body.size());
  out.insert(out.end(), body.begin(), body.end());
  return true;
}

static bool deflate_raw(const u8* src, size_t n, std::vector<u8>& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  u8 buf[16384];
  int rc;
  do {
    zs.next_out = buf;
    zs.avail_out = sizeof(buf);
    rc = deflate(&zs, Z_FINISH);
    out.insert(out.end(), buf, buf + (sizeof(buf) - zs.avail_out));
  } while (rc == Z_OK || rc == Z_BUF_ERROR);
  deflateEnd(&zs);
  return rc == Z_STREAM_END;
}

// SaveOptions::deflate (automerge.rs:1816-1830): every column of at least 256 bytes is raw-DEFLATEd and
// gets the deflate bit, then the chunk is re-hashed over the compressed body
// (storage/document/compression.rs:29-49, 346-355; raw_column.rs:49-62).  Host side by design
// (SURVEY.md 8a row a22); the compressed bytes are zlib's, not flate2's: parity unpinned.
bool amh_host_deflate_document(const u8* p, size_t len, std::vector<u8>& out) {
  if (len < 10 || p[8] != 0) return false;
  u64 pos = 9, end = len, dlen;
  if (rd_uleb_strict(p, pos, end, dlen) || dlen != end - pos) return false;
  u64 data_off = pos, n;
  if (rd_uleb_strict(p, pos, end, n)) return false;
  for (u64 i = 0; i < n; i++) { u64 al; if (rd_uleb_strict(p, pos, end, al) || al > end - pos) return false; pos += al; }
  if (rd_uleb_strict(p, pos, end, n) || n > (end - pos) / 32) return false;
  pos += 32 * n;
  u64 prefix_end = pos;
  struct Col { u32 spec; u64 len; };
  std::vector<Col> cols[2];
  for (int w = 0; w < 2; w++) {
    u64 nc;
    if (rd_uleb_strict(p, pos, end, nc) || nc > end - pos) return false;
    for (u64 i = 0; i < nc; i++) {
      u64 spec, cl;
      if (rd_uleb_strict(p, pos, end, spec) || rd_uleb_strict(p, pos, end, cl) || cl > end) return false;
      cols[w].push_back(Col{(u32)spec, cl});
    }
  }
  std::vector<u8> data[2];
  bool any = false;
  for (int w = 0; w < 2; w++)
    for (auto& c : cols[w]) {
      if (c.len > end - pos) return false;
      if (c.len >= 256 && !(c.spec & 8)) {
        size_t before = data[w].size();
        if (!deflate_raw(p + pos, c.len, data[w])) return false;
        pos += c.len;
        c.spec |= 8;
        c.len = data[w].size() - before;
        any = true;
      } else {
        data[w].insert(data[w].end(), p + pos, p + pos + c.len);
        pos += c.len;
      }
    }
  if (!any) return false;
  std::vector<u8> body(p + data_off, p + prefix_end);
  for (int w = 0; w < 2; w++) {
    push_uleb(body, cols[w].size());
    for (auto& c : cols[w]) { push_uleb(body, c.spec); push_uleb(body, c.len); }
  }
  body.insert(body.end(), data[0].begin(), data[0].end());
  body.insert(body.end(), data[1].begin(), data[1].end());
  body.insert(body.end(), p + pos, p + end);
  std::vector<u8> pre;
  pre.push_back(0);
  push_uleb(pre, body.size());
  u8 h[32];
  amh_host_sha256(pre.data(), pre.size(), body.data(), body.size(), h);
  out.clear();
  out.insert(out.end(), p, p + 4);
  out.insert(out.end(), h, h + 4);
  out.insert(out.end(), pre.begin(), pre.end());
  out.insert(out.end(), body.begin(), body.end());
  return true;
}
