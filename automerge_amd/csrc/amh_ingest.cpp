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

void amh_host_sha256(const u8* a, size_t an, const u8* b, size_t bn, u8* out32) {
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
