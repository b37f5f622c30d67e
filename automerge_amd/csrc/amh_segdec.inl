// Segment-parallel column decode (device only; included by amh_kernels.hip).
//
// One wave cooperates on ONE column stream of one document (rle.rs:175-210, delta.rs:79-95, boolean.rs:84-101) instead
// of one lane walking it: the 64 lanes of a wave then do the same work on 64 tokens / 64 rows at a time, every load and
// store is a coalesced wave access, and a large document uses a whole wave per column instead of one lane.
//
//   tile    256 column bytes -> LDS (one coalesced dword load per lane)
//   tokens  LEB128 boundaries = bytes without the continuation bit: 4 ballots give the rank of every token end, the end
//           offsets go to LDS in token order (a token that does not end inside the tile is left for the next tile, which
//           starts at its first byte: nothing but a few scalars is carried between tiles)
//   values  lane t decodes token t out of LDS (one unaligned peek, the 7-bit groups packed in registers).  Groups whose
//           tokens all fit 4 bytes (values below 2^28: nearly everything) run in 32-bit arithmetic, the others in 64-bit
//   headers which tokens are run headers is a chain (a header says where the next one is: 2 tokens on for a run or a
//           null run, 1 + n for a literal run of n).  No literal header in sight: the headers are every second token
//           (bit arithmetic on one ballot).  Otherwise pointer doubling over the 64 tokens: 5 rounds of "reached tokens
//           flag their successor in LDS, every token squares its successor function" — vector work of fixed cost,
//           because the scalar unit is shared by the four SIMDs of a CU and a scalar loop per header starved it
//   items   run header -> (count, value of the next token); null header -> (count = next token, null); literal value ->
//           (1, own value).  An exclusive wave scan of the counts gives every item its first row; a second scan carries
//           the running sum of a delta column (or the raw-value offsets, or the succ bases) across the items
//   expand  64 rows at a time.  A group of literal values only: every item is one row, the values are compacted through
//           LDS.  Otherwise item starts are scattered into an LDS row of flags, an inclusive max-scan turns them into
//           "item of this row", the item's fields come over by ds_bpermute.  One coalesced store per output array.
//
// Error semantics are the serial decoder's: a malformed token only fails the document if the decoder would have read
// it before producing the last row it was asked for (rows in front of the token < rows wanted); the first such token in
// stream order decides the error kind.  A column that ends early yields nulls where the reference's iterator does
// (`None`), and fails where it is required.
#pragma once

#define SD_TILE 256u
#define SD_WAVES 4u            // waves (= column streams) per workgroup

struct SdWave {
  u32 tile[SD_TILE / 4 + 4];   // + 16 bytes: the 8-byte peek of the last token may read past the tile
  unsigned short tokend[SD_TILE];
  u32 xs[64], ys[64];
  u8 flag[192];              // header flags [0, 64], then one scratch slot per lane
};

#include "amh_segdec_prims.inl"

// what the wave carries from group to group (all uniform)
struct SdStream {
  const u8* src;      // arena address of the document's chunk
  u32 pos, end;       // byte cursor / end of the column, chunk relative
  u32 want;           // rows the caller wants
  u32 row;            // rows produced
  u64 lit;            // values of an open literal run still to come
  u32 btok;           // boolean columns: tokens consumed so far (parity = value)
  u64 acc;            // delta: running value; value meta: raw offset; succ count: entries so far
};

template <bool WIDE> struct SdT { typedef u32 U; typedef i32 S; };
template <> struct SdT<true> { typedef u64 U; typedef i64 S; };

// One group of <= 64 tokens, one per lane, classified: what every lane's token contributes.
template <bool WIDE> struct SdItems {
  u32 cnt;            // rows of this lane's item (0: the token is the header of a literal run, a run's value, ...), clamped to the rows still wanted
  u32 rs;             // rows of the group in front of the item (exclusive scan of cnt)
  u32 total;          // rows of the whole group (uniform; may exceed what is wanted)
  bool isnull;
  typename SdT<WIDE>::U u;   // value as unsigned LEB128
  typename SdT<WIDE>::S s;   // value as signed LEB128
};

enum { SD_RLE_U = 0, SD_RLE_S = 1, SD_BOOL = 2 };

// load the tile that holds S.pos, find its tokens; returns their number (uniform).  tb = tile start (chunk relative).
__device__ __forceinline__ u32 sd_load_tile(SdWave& W, const SdStream& S, u32 lane, u32& tb, u32& skip, u32& avail) {
  tb = S.pos & ~3u;
  skip = S.pos - tb;
  avail = S.end - tb < SD_TILE ? S.end - tb : SD_TILE;
  u32 w = 4 * lane < avail ? *(const u32*)(S.src + tb + 4 * lane) : 0u;   // chunks start 16-byte aligned; the arena is padded
  W.tile[lane] = w;
  if (lane < 4) W.tile[64 + lane] = 0;
  sd_lds_sync();
  u32 ntok = 0;
  const u8* tb8 = (const u8*)W.tile;
#pragma unroll
  for (u32 j = 0; j < 4; j++) {
    u32 bi = 64 * j + lane;
    u8 b = tb8[bi];
    bool isend = bi >= skip && bi < avail && !(b & 0x80);
    u64 m = __ballot(isend);
    if (isend) W.tokend[ntok + sd_mbcnt(m)] = (unsigned short)bi;
    ntok += (u32)__builtin_popcountll(m);
  }
  sd_lds_sync();
  return ntok;
}

// token starting at tile byte b0, nb bytes: value both ways; ok_u / ok_s = a valid u64 / i64 for the lenient readers
// (decodable_impls.rs:58-96)
__device__ __forceinline__ void sd_token64(const SdWave& W, u32 b0, u32 nb, u64& u, i64& s, bool& ok_u, bool& ok_s) {
  const u32* w = W.tile + (b0 >> 2);
  u32 w0 = w[0], w1 = w[1], w2 = w[2];
  u32 sh = (b0 & 3) * 8;
  u64 lo = (u64)w0 | ((u64)w1 << 32);
  u64 x = sh ? (lo >> sh) | ((u64)w2 << (64 - sh)) : lo;
  ok_u = ok_s = true;
  if (nb <= 8) {
    u64 g = amh_leb_groups(x, nb);
    u32 bits = 7 * nb;
    u = g;
    s = (i64)((g & ((u64)1 << (bits - 1))) ? g | (~(u64)0 << bits) : g);
  } else {   // 9 or 10 bytes (values of 2^56 and more), or an over-long run of continuation bytes
    const u8* p = (const u8*)W.tile + b0;
    u64 r = 0;
    u32 shift = 0;
    u8 b = 0;
    u32 lim = nb < 10 ? nb : 10;
    for (u32 i = 0; i < lim; i++) { b = p[i]; r |= (u64)(b & 0x7f) << shift; shift += 7; }
    u = r;
    if (shift < 64 && (b & 0x40)) r |= ~(u64)0 << shift;
    s = (i64)r;
    if (nb > 10) ok_u = ok_s = false;
    else if (nb == 10) { ok_u = b <= 1; ok_s = b == 0 || b == 0x7f; }
  }
}
// the same for a token of at most 4 bytes
__device__ __forceinline__ void sd_token32(const SdWave& W, u32 b0, u32 nb, u32& u, i32& s) {
  const u32* w = W.tile + (b0 >> 2);
  u32 w0 = w[0], w1 = w[1];
  u32 x = __builtin_amdgcn_alignbyte(w1, w0, b0 & 3);
  x &= 0xFFFFFFFFu >> (32 - 8 * nb);
  x = (x & 0x007f007fu) | ((x & 0x7f007f00u) >> 1);
  x = (x & 0x00003fffu) | ((x & 0x3fff0000u) >> 2);
  u = x;
  u32 sh = 32 - 7 * nb;
  s = (i32)(x << sh) >> sh;
}

// The group of tokens t0 .. t0 + m of the tile (lane = token; b0 / nb = where the lane's token lies).  Returns the number
// of tokens consumed (uniform; 0 = the group ends in a header whose value lies in the next tile).  terr = this lane's
// token (or the value token behind its header) is malformed for the role it plays.  lit / btok: the stream's state
// behind the consumed tokens (the caller commits them).
template <int KIND, bool WIDE>
__device__ __forceinline__ u32 sd_group(SdWave& W, const SdStream& S, u32 lane, u32 m, u32 b0, u32 nb, SdItems<WIDE>& I, bool& terr, u64& lit_out, u32& btok_out) {
  typedef typename SdT<WIDE>::U UV;
  typedef typename SdT<WIDE>::S SV;
  UV u = 0; SV s = 0; bool ok_u = true, ok_s = true;
  if (WIDE) { u64 uu = 0; i64 ss = 0; if (lane < m) sd_token64(W, b0, nb, uu, ss, ok_u, ok_s); u = (UV)uu; s = (SV)ss; }
  else { u32 uu; i32 ss; sd_token32(W, b0, nb, uu, ss); if (lane >= m) { uu = 0; ss = 0; } u = (UV)uu; s = (SV)ss; }
  const u32 rem = S.want - S.row;
  u32 c;
  UV cnt = 0;
  bool isnull = false;
  terr = false;
  I.u = 0; I.s = 0;
  lit_out = S.lit; btok_out = S.btok;
  if (KIND == SD_BOOL) {   // every token is a count; the value alternates, starting with false (boolean.rs:84-101)
    c = m;
    if (lane < m) { cnt = u; terr = !ok_u; }
    I.u = (S.btok + lane) & 1;   // the value the count stands for
    btok_out = S.btok + c;
  } else {
    // Which tokens are headers.  In front: p0 values of a literal run that opened in an earlier group.
    const u64 NEG = __ballot(lane < m && s < 0);
    const u32 p0 = S.lit < m ? (u32)S.lit : m;
    u64 lit = S.lit - p0;
    bool reach = false;
    c = m;
    if (p0 < m) {
      if (!(NEG >> p0)) {
        // no literal header in sight: run / null headers hop by 2, each needs its value token inside the group
        const u64 hm = (0x5555555555555555ull << (p0 & 1)) & (~(u64)0 << p0) & sd_low_bits(m - 1);
        reach = (hm >> lane) & 1;
        c = p0 + 2 * (u32)__builtin_popcountll(hm);
      } else {
        // the chain of headers by pointer doubling: a header at t names the next one at J(t) = t + 2 (run, null) or
        // t + 1 + n (literal run of n); a chain makes at most 32 hops inside 64 tokens = 5 rounds.  Reached tokens are
        // flagged in LDS (lanes that are not reached yet write to a slot of their own behind the flags).
        u32 hop = 2;
        if (s < 0) { UV n = (UV)0 - (UV)s; hop = 1 + (n < 64 ? (u32)n : 64u); }
        const u32 jraw = lane + hop;
        u32 J = jraw < 64 ? jraw : 64;
        reach = lane == p0;
        sd_lds_sync();
        W.flag[lane] = reach ? 1 : 0;
        sd_lds_sync();
#pragma unroll
        for (int r = 0; r < 5; r++) {
          W.flag[reach ? J : 65 + lane] = 1;
          sd_lds_sync();
          reach = W.flag[lane] != 0;
          u32 jn = (u32)__shfl((int)J, (int)(J & 63), 64);
          J = J >= 64 ? 64 : jn;
          sd_lds_sync();
        }
        reach = reach && lane < m;
        // the last header decides where the group ends and what it leaves open
        const u32 h = 63 - (u32)__builtin_clzll(__ballot(reach));
        if ((NEG >> h) & 1) {
          u64 n = WIDE ? (u64)0 - sd_bcast64((u64)s, h) : (u64)((u32)0 - sd_bcast((u32)s, h));
          lit = h + 1 + n - m;            // the run ends at or behind the end of the group (else a later header were reached)
        } else if (h + 2 > m) c = h;      // its value token lies in the next group / tile
      }
    }
    lit_out = lit;
    const bool prev_rh = sd_prev((u32)(reach && s >= 0)) != 0;   // the token in front is a run / null header: this is its value
    const bool is_hdr = reach && s >= 0 && lane < c, is_lith = reach && s < 0, is_litv = lane < c && !reach && !prev_rh;
    // the token behind a run / null header
    UV nu; SV ns; bool n_ok_u = true, n_ok_s = true;
    if (WIDE) {
      nu = (UV)((u64)sd_next((u32)u) | ((u64)sd_next((u32)((u64)u >> 32)) << 32));
      ns = (SV)((u64)sd_next((u32)s) | ((u64)sd_next((u32)((u64)s >> 32)) << 32));
      n_ok_u = sd_next((u32)ok_u); n_ok_s = sd_next((u32)ok_s);
    } else { nu = (UV)sd_next((u32)u); ns = (SV)sd_next((u32)s); }
    if (is_hdr) {
      if (s > 0) { cnt = (UV)s; I.u = nu; I.s = ns; terr = !ok_s || !(KIND == SD_RLE_S ? n_ok_s : n_ok_u); }
      else { cnt = nu; isnull = true; terr = !ok_s || !n_ok_u || (WIDE && (u64)nu > (u64)INT64_MAX); }
    } else if (is_litv) {
      cnt = 1; I.u = u; I.s = s; terr = !(KIND == SD_RLE_S ? ok_s : ok_u);
    } else if (is_lith) {
      terr = !ok_s;
    }
    if (lane >= c) { cnt = 0; terr = false; }
  }
  I.isnull = isnull;
  I.cnt = cnt < rem ? (u32)cnt : rem;
  u32 inc = sd_scan_add(I.cnt);
  I.rs = inc - I.cnt;
  I.total = sd_bcast(inc, 63);
  return c;
}

// rows [r0, r0 + 64) of a group: the lane (= item) each row belongs to.  Needs total > r0.
__device__ __forceinline__ u32 sd_row_item(SdWave& W, u32 cnt, u32 rs, u32 lane, u32 r0) {
  sd_lds_sync();
  W.xs[lane] = 0;
  sd_lds_sync();
  const bool has = cnt != 0;
  if (has && rs >= r0 && rs < r0 + 64) W.xs[rs - r0] = lane + 1;
  sd_lds_sync();
  u64 cover = __ballot(has && rs <= r0 && r0 - rs < cnt);   // the item that holds row r0: exactly one
  u32 f = W.xs[lane];
  if (lane == 0) f = (u32)__builtin_ctzll(cover) + 1;
  return sd_scan_max(f) - 1;
}

// A column stream of AMH_BIGCOL bytes and more is not walked by one wave: see "a column over many waves" at the end.
#define AMH_BIGRUN 8192u
#define AMH_BIGCOL (32u << 10)
#define AMH_SEG 4096u
__device__ __forceinline__ bool bc_is_big(const BatchDev& B, u32 len) { return B.bc_n != nullptr && len >= AMH_BIGCOL; }

template <u32 COL> struct SdCol {
  static constexpr bool SUCC_ENTRY = COL == OC_SUCC_ACTOR || COL == OC_SUCC_CTR;
  static constexpr bool IS_DELTA = COL == OC_KEY_CTR || COL == OC_ID_CTR || COL == OC_SUCC_CTR;
  static constexpr bool IS_BOOL = COL == OC_INSERT || COL == OC_EXPAND;
  static constexpr int KIND = IS_BOOL ? SD_BOOL : (IS_DELTA ? SD_RLE_S : SD_RLE_U);
  // a column that ends early (or is absent): the reference's iterator yields None, which these accept as null
  static constexpr bool NULLABLE = COL == OC_OBJ_ACTOR || COL == OC_OBJ_CTR || COL == OC_KEY_ACTOR || COL == OC_KEY_CTR ||
                                   COL == OC_ID_ACTOR || COL == OC_ID_CTR || COL == OC_EXPAND;
  static constexpr bool HAS_ACC = IS_DELTA || COL == OC_VAL_META || COL == OC_SUCC_NUM;
  // slot of the column in the decode launch (blockIdx.y - 2) and in bc_index
  static constexpr u32 SLOT = COL == OC_OBJ_ACTOR ? 0 : COL == OC_OBJ_CTR ? 1 : COL == OC_KEY_ACTOR ? 2 : COL == OC_KEY_CTR ? 3 : COL == OC_ID_ACTOR ? 4 :
                              COL == OC_ID_CTR ? 5 : COL == OC_INSERT ? 6 : COL == OC_ACTION ? 7 : COL == OC_VAL_META ? 8 : COL == OC_SUCC_NUM ? 9 :
                              COL == OC_SUCC_ACTOR ? 10 : COL == OC_SUCC_CTR ? 11 : 12;
};

template <u32 COL>
__device__ __forceinline__ void sd_store_row(const BatchDev& B, u32 d, u32 o, u32 a, u32 b, u32 c) {
  if (COL == OC_KEY_CTR) B.r_key_ctr[o] = c;
  else if (COL == OC_ID_CTR) { B.r_id_ctr[o] = c; B.r_doc[o] = d; }
  else if (COL == OC_SUCC_CTR) B.s_ctr[o] = c;
  else if (COL == OC_VAL_META) { B.r_vmeta[o] = (u64)a | ((u64)b << 32); B.r_vraw_off[o] = c; }
  else if (COL == OC_SUCC_NUM) { B.r_succ_cnt[o] = a; B.r_succ_base[o] = c; }
  else if (COL == OC_OBJ_ACTOR) B.r_obj_actor[o] = a;
  else if (COL == OC_OBJ_CTR) B.r_obj_ctr[o] = a;
  else if (COL == OC_KEY_ACTOR) B.r_key_actor[o] = a;
  else if (COL == OC_ID_ACTOR) B.r_id_actor[o] = a;
  else if (COL == OC_SUCC_ACTOR) B.s_actor[o] = a;
  else if (COL == OC_ACTION) B.r_action[o] = (u8)a;
  else if (COL == OC_INSERT) B.r_insert[o] = (u8)a;
  else if (COL == OC_EXPAND) B.r_expand[o] = (u8)a;
}

// One group of one column stream: classify, compute the items, check, write the rows, advance the stream.
// Returns the tokens consumed (0 = none could be), or SD_REDO when 32-bit arithmetic does not hold the group (the
// caller runs it again with WIDE); fail receives the error that ends the document.
#define SD_REDO 0xFFFFFFFFu
// DRY: measure only (big columns, §"a column over many waves" below): the rows and the running sum a range of the column
// stands for, from a relative state (rows from 0, sum from 0, everything wanted).  Malformed tokens still end the walk;
// what a value may be (id ranges, raw-column bounds, nulls) is left to the real pass, which sees absolute values.
template <u32 COL, bool WIDE, bool DRY = false>
__device__ __forceinline__ u32 sd_do_group(const BatchDev& B, u32 d, SdWave& W, SdStream& S, u32 lane, u32 m, u32 b0, u32 nb,
                                           u32 obase, u32 sbase, u32 raw_len, u32& mx, i32& fail) {
  typedef SdCol<COL> C;
  SdItems<WIDE> I;
  bool terr;
  u64 lit_new; u32 btok_new;
  const u32 rem = S.want - S.row;
  const u32 c = sd_group<C::KIND, WIDE>(W, S, lane, m, b0, nb, I, terr, lit_new, btok_new);
  if (c == 0) return 0;
  // ---- what the item's rows hold: a (per item), b (per item), cc + step * k for row k of the item -----------------
  const u32 cnt = I.cnt;
  const u32 reach = I.rs < rem ? (rem - I.rs < cnt ? rem - I.rs : cnt) : 0;   // rows of the item the decoder delivers
  i32 ecode = terr ? AMH_E_READ_OP : 0;
  u32 e0 = 0, e1 = 0, e2 = 0, step = 0;
  u64 acc_new = S.acc;
  u32 mx_new = mx;
  if (C::IS_DELTA) {
    if (WIDE) {
      u64 sv = (u64)(i64)I.s;
      u64 tot = I.isnull ? 0 : sv * cnt;
      u64 inc = sd_scan_add64(tot);
      u64 first = S.acc + (inc - tot) + sv;           // value of the item's first row
      if (I.isnull) { e2 = AMH_NULL32; if (!DRY && C::SUCC_ENTRY && reach && !ecode) ecode = AMH_E_READ_OP; }
      else {
        e2 = (u32)first; step = (u32)sv;
        if (!DRY && reach && !ecode) {
          i64 lo = (i64)first, hi = (i64)(first + sv * (reach - 1));
          if (lo > hi) { i64 t = lo; lo = hi; hi = t; }
          if (lo < 0 || hi >= 0xFFFFFFFFll) ecode = AMH_E_READ_OP;   // key.rs:172-179, opid.rs:107-113
          else if (COL != OC_KEY_CTR && (u32)hi > mx_new) mx_new = (u32)hi;
        }
      }
      acc_new = S.acc + sd_bcast64(inc, 63);
    } else {
      // 32-bit: the running value is below 2^30 (caller) and no item moves it by 2^24 or more, so nothing wraps
      i32 sv = (i32)I.s;
      u32 mag = sv < 0 ? (u32)-sv : (u32)sv;
      bool big = !I.isnull && (__umulhi(mag, cnt) != 0 || mag * cnt >= (1u << 24));
      if (__ballot(big)) return SD_REDO;
      i32 tot = I.isnull ? 0 : sv * (i32)cnt;
      i32 inc = (i32)sd_scan_add((u32)tot);
      i32 first = (i32)(u32)S.acc + (inc - tot) + sv;
      if (I.isnull) { e2 = AMH_NULL32; if (!DRY && C::SUCC_ENTRY && reach && !ecode) ecode = AMH_E_READ_OP; }
      else {
        e2 = (u32)first; step = (u32)sv;
        if (!DRY && reach && !ecode) {
          i32 lo = first, hi = first + sv * (i32)(reach - 1);
          if (lo > hi) { i32 t = lo; lo = hi; hi = t; }
          if (lo < 0) ecode = AMH_E_READ_OP;
          else if (COL != OC_KEY_CTR && (u32)hi > mx_new) mx_new = (u32)hi;
        }
      }
      acc_new = (u64)((i64)S.acc + (i64)(i32)sd_bcast((u32)inc, 63));
    }
  } else if (C::IS_BOOL) {
    e0 = (u32)I.u;
  } else if (COL == OC_VAL_META) {
    u64 uv = (u64)I.u;
    u64 l = ((uv & 15) >= 3) ? (uv >> 4) : 0;
    e0 = (u32)uv; e1 = (u32)(uv >> 32);
    if (WIDE) {
      u64 tot = l * cnt;
      u64 inc = sd_scan_add64(tot);
      u64 first = S.acc + (inc - tot);
      e2 = (u32)first; step = (u32)l;
      if (!DRY && reach && !ecode) {
        if (I.isnull) ecode = AMH_E_READ_OP;
        else if ((uv & 15) >= 3) {   // null / bool never touch the raw column (value.rs:221-224)
          u64 last = first + l * (reach - 1);
          if (l > raw_len || last > raw_len - l) ecode = AMH_E_READ_OP;
        }
      }
      acc_new = S.acc + sd_bcast64(inc, 63);
    } else {
      u32 l32 = (u32)l;
      bool big = __umulhi(l32, cnt) != 0 || l32 * cnt >= (1u << 24);
      if (__ballot(big)) return SD_REDO;
      u32 tot = l32 * cnt;
      u32 inc = sd_scan_add(tot);
      u32 first = (u32)S.acc + (inc - tot);
      e2 = first; step = l32;
      if (!DRY && reach && !ecode) {
        if (I.isnull) ecode = AMH_E_READ_OP;
        else if ((uv & 15) >= 3) {
          u32 last = first + l32 * (reach - 1);
          if (l32 > raw_len || last > raw_len - l32) ecode = AMH_E_READ_OP;
        }
      }
      acc_new = S.acc + sd_bcast(inc, 63);
    }
  } else if (COL == OC_SUCC_NUM && DRY) {
    // the measured sum sizes the succ arrays: exact in 64 bits, absurd counts end the measure (the serial count pass,
    // with the reader's own error rules, then decides)
    if (__ballot(cnt != 0 && !I.isnull && (u64)I.u > 0x7FFFFFFFull)) { fail = AMH_E_TOO_LARGE; return c; }
    u64 tot = I.isnull ? 0 : (u64)I.u * cnt;
    u64 inc = sd_scan_add64(tot);
    acc_new = S.acc + sd_bcast64(inc, 63);
  } else if (COL == OC_SUCC_NUM) {
    u32 v = (u32)I.u;
    u32 tot = v * cnt;
    u32 inc = sd_scan_add(tot);
    e0 = v; e2 = sbase + (u32)S.acc + (inc - tot); step = v;
    if (!DRY && reach && !ecode && I.isnull) ecode = AMH_E_READ_OP;
    acc_new = S.acc + sd_bcast(inc, 63);
  } else {   // plain RLE of u64: obj actor / ctr, key actor, id actor, action, succ actor
    e0 = I.isnull ? AMH_NULL32 : (u32)I.u;
    if (!DRY && reach && !ecode) {
      const bool over = WIDE ? (u64)I.u >= 0xFFFFFFFFull : false;
      if (COL == OC_ACTION) { if (I.isnull) ecode = AMH_E_READ_OP; else if (I.u > 7) ecode = AMH_E_INVALID_OP; }
      else if (COL == OC_SUCC_ACTOR) { if (I.isnull) ecode = AMH_E_READ_OP; else if (over) ecode = AMH_E_MISSING_ACTOR; }
      else if (!I.isnull && over) ecode = COL == OC_OBJ_CTR ? AMH_E_READ_OP : AMH_E_MISSING_ACTOR;
    }
  }
  // a malformed token counts only if the decoder gets to it: rows in front of it < rows wanted
  const u64 bad = __ballot(ecode != 0 && I.rs < rem);
  if (bad) { fail = (i32)sd_bcast((u32)ecode, (u32)__builtin_ctzll(bad)); return c; }
  // ---- expand ------------------------------------------------------------------------------------------------
  const u32 R = I.total < rem ? I.total : rem;
  if (DRY) { S.row += R; S.lit = lit_new; S.btok = btok_new; S.acc = acc_new; return c; }
  const u32 orow = obase + S.row;
  if (__ballot(cnt > 1) == 0) {
    // literal values only (and empty runs): every item is one row, R <= 64; compact through LDS
    constexpr bool TWO = COL == OC_VAL_META || COL == OC_SUCC_NUM;
    u32 v1 = (C::IS_DELTA) ? e2 : e0;
    sd_lds_sync();
    if (cnt) { W.xs[I.rs] = v1; if (TWO) W.ys[I.rs] = e2; }
    sd_lds_sync();
    u32 a = W.xs[lane], cc = TWO ? W.ys[lane] : a, b = 0;
    if (WIDE && COL == OC_VAL_META) {   // the upper half of a value meta is zero unless WIDE
      sd_lds_sync();
      if (cnt) W.xs[I.rs] = e1;
      sd_lds_sync();
      b = W.xs[lane];
    }
    if (lane < R) sd_store_row<COL>(B, d, orow + lane, a, b, cc);
  } else {
    // a run of AMH_BIGRUN rows and more is left to the whole grid (g_fill_runs): one wave storing 64 rows at a time is
    // the serial part of a 10M-row document whose object / action / value-meta columns are a handful of runs
    const bool defer = B.fr_n != nullptr && reach >= AMH_BIGRUN;
    if (defer) {   // in pieces of 64 Ki rows: a block of g_fill_runs each
      const u32 st = COL == OC_SUCC_NUM ? e0 : step;
      const u32 c0 = (C::IS_DELTA || COL == OC_VAL_META || COL == OC_SUCC_NUM) ? e2 : 0u;
      for (u32 q = 0; q < reach; q += 65536u) {
        u32 i = atomicAdd(B.fr_n, 1u);
        if (i >= B.fr_cap) { amh_fail(B, d, AMH_E_TOO_LARGE); break; }   // cannot happen (fr_cap counts every possible piece): never silently
        u32* it = B.fr_items + 8 * (size_t)i;
        it[0] = COL; it[1] = d; it[2] = orow + I.rs + q; it[3] = reach - q < 65536u ? reach - q : 65536u;
        it[4] = (C::IS_DELTA) ? 0u : e0; it[5] = e1; it[6] = c0 + st * q; it[7] = st;
      }
    }
    const u64 defer_m = __ballot(defer);
    for (u32 r0 = 0; r0 < R; r0 += 64) {
      if (defer_m) {   // whole blocks of 64 rows inside a deferred run: skipped (its first and last partial blocks are written here too)
        const u64 cover = __ballot(cnt != 0 && I.rs <= r0 && r0 - I.rs < cnt) & defer_m;
        if (cover) {
          const u32 tl = (u32)__builtin_ctzll(cover);
          const u32 endr = sd_bcast(I.rs + reach, tl);
          if (endr >= r0 + 64) { r0 += 64 * ((endr - r0) / 64 - 1); continue; }
        }
      }
      const u32 t = sd_row_item(W, cnt, I.rs, lane, r0);
      const u32 my = r0 + lane;
      const u32 k = my - (u32)__shfl((int)I.rs, (int)t, 64);
      u32 a = 0, b = 0, cc = 0;
      if (C::IS_DELTA) cc = (u32)__shfl((int)e2, (int)t, 64) + (u32)__shfl((int)step, (int)t, 64) * k;
      else if (COL == OC_VAL_META) {
        a = (u32)__shfl((int)e0, (int)t, 64);
        if (WIDE) b = (u32)__shfl((int)e1, (int)t, 64);
        cc = (u32)__shfl((int)e2, (int)t, 64) + (u32)__shfl((int)step, (int)t, 64) * k;
      } else if (COL == OC_SUCC_NUM) {
        a = (u32)__shfl((int)e0, (int)t, 64);
        cc = (u32)__shfl((int)e2, (int)t, 64) + a * k;
      } else a = (u32)__shfl((int)e0, (int)t, 64);
      if (my < R) sd_store_row<COL>(B, d, orow + my, a, b, cc);
    }
  }
  S.row += R;
  S.lit = lit_new;
  S.btok = btok_new;
  S.acc = acc_new;
  mx = mx_new;
  return c;
}

// The walk of one column stream: tiles, groups of tokens, the carry between them.  F supplies what a group means:
//   template <bool WIDE> u32 group(SdWave&, SdStream&, lane, m, b0, nb, i32& fail)   tokens consumed / 0 / SD_REDO
//   bool wide(const SdStream&)                                                        force 64-bit arithmetic
// Returns 0 or the error that ends the document; `stuck` is the error for tokens that cannot be read.  Stops when
// S.row reaches S.want or the column ends (the caller decides what an early end means; S.lit tells whether it fell
// inside a literal run).
template <class F>
__device__ __forceinline__ i32 sd_walk(SdWave& W, SdStream& S, u32 lane, F& f, i32 stuck) {
  i32 fail = 0;
  while (S.row < S.want && S.pos < S.end) {
    u32 tb, skip, avail;
    sd_lds_sync();   // the previous tile has been read
    const u32 ntok = sd_load_tile(W, S, lane, tb, skip, avail);
    u32 t0 = 0;
    while (t0 < ntok && S.row < S.want) {
      const u32 m = ntok - t0 < 64 ? ntok - t0 : 64;
      const u32 t = t0 + (lane < m ? lane : 0);
      const u32 b0 = t ? (u32)W.tokend[t - 1] + 1 : skip;
      const u32 nb = lane < m ? (u32)W.tokend[t] - b0 + 1 : 1;
      // 32-bit arithmetic where every token has at most 4 bytes (and the functor does not object)
      const bool wide = __ballot(nb > 4) != 0 || f.wide(S);
      u32 c = SD_REDO;
      if (!wide) c = f.template group<false>(W, S, lane, m, b0, nb, fail);
      if (c == SD_REDO) c = f.template group<true>(W, S, lane, m, b0, nb, fail);
      if (c == 0 || fail) break;
      t0 += c;
    }
    if (fail) return fail;
    // next tile: from the first token that was not consumed
    const u32 npos = sd_uniform(tb + (t0 ? (u32)W.tokend[t0 - 1] + 1 : skip));
    if (npos == S.pos) {
      // no token could be consumed: a token longer than the tile, one cut off by the end of the column, or a run /
      // null header whose value the column no longer holds — the serial reader fails on all of them
      return S.row < S.want ? stuck : 0;
    }
    S.pos = npos;
  }
  return 0;
}

template <u32 COL> struct SdDecodeF {
  const BatchDev& B;
  u32 d, obase, sbase, raw_len, mx;
  __device__ __forceinline__ bool wide(const SdStream& S) const { return SdCol<COL>::HAS_ACC && S.acc >= (1u << 30); }
  template <bool WIDE>
  __device__ __forceinline__ u32 group(SdWave& W, SdStream& S, u32 lane, u32 m, u32 b0, u32 nb, i32& fail) {
    return sd_do_group<COL, WIDE>(B, d, W, S, lane, m, b0, nb, obase, sbase, raw_len, mx, fail);
  }
};

// One column stream.  COL selects what an item means and where its rows go.
// range = a stretch of the column with the reader's state at its start (the whole column: everything zero); `last` =
// the stretch ends where the column ends (only then does running out of bytes mean the column is exhausted)
template <u32 COL>
__device__ __forceinline__ void sd_decode_range(const BatchDev& B, u32 d, SdWave& W, u32 pos, u32 end, u32 row0, u64 lit, u32 btok, u64 acc, bool last) {
  typedef SdCol<COL> C;
  const u32 lane = threadIdx.x & 63;
  const DocDesc& D = B.desc[d];
  const u32 want = C::SUCC_ENTRY ? B.cnt[d].n_succ : B.cnt[d].n_rows;
  if (want == 0 || (row0 >= want && !last)) return;
  const u32 obase = C::SUCC_ENTRY ? B.succ_base[d] : B.row_base[d];
  SdStream S;
  S.src = B.arena + D.chunk_off;
  S.pos = pos;
  S.end = end;
  S.want = want; S.row = row0 < want ? row0 : want; S.lit = lit; S.btok = btok; S.acc = acc;
  SdDecodeF<COL> f{B, d, obase, COL == OC_SUCC_NUM ? B.succ_base[d] : 0u, D.ocol[OC_VAL_RAW].len, 0u};
  const i32 fail = sd_walk(W, S, lane, f, AMH_E_READ_OP);
  if (fail) { if (lane == 0) amh_fail(B, d, fail); return; }
  if (COL == OC_ID_CTR || COL == OC_SUCC_CTR) {
    u32 m = sd_scan_max(f.mx);
    if (lane == 63 && m) amh_atomic_max(&B.doc_max_op[d], m);
  }
  if (!last) return;
  if (S.row < want) {
    // the column is exhausted.  Inside a literal run the reader fails (the announced value is missing); otherwise
    // the iterator yields None from here on
    bool ok = C::NULLABLE && S.lit == 0;
    if (!ok) { if (lane == 0) amh_fail(B, d, AMH_E_READ_OP); return; }
    for (u32 r = S.row + lane; r < want; r += 64) sd_store_row<COL>(B, d, obase + r, COL == OC_EXPAND ? 0u : AMH_NULL32, 0, AMH_NULL32);
  }
}
template <u32 COL>
__device__ __forceinline__ void sd_decode_column(const BatchDev& B, u32 d, SdWave& W) {
  const DocDesc& D = B.desc[d];
  if (bc_is_big(B, D.ocol[COL].len) && B.bc_index[d * 13u + SdCol<COL>::SLOT] != AMH_NULL32) return;   // cut into ranges, a wave each: g_bc_decode
  sd_decode_range<COL>(B, d, W, D.ocol[COL].off, D.ocol[COL].off + D.ocol[COL].len, 0, 0, 0, 0, true);
}

// All op columns of a batch in one launch.  blockIdx.y = column (uniform per workgroup).  y = 0, 1: the two string columns,
// one wave per 64 documents (decode_str_body; first in the grid so that these few long-running waves start at once and
// run beside the rest); y >= 2: one wave per (document, column).
#define SD_NCOLS 13
union SdBlockShared {
  SdWave w[SD_WAVES];
  DsShared s;
};
__global__ void __launch_bounds__(64 * SD_WAVES) g_decode_seg(BatchDev B) {
  __shared__ SdBlockShared L;
  if (blockIdx.y < 2) {
    if (threadIdx.x >= 64 || blockIdx.x * 64 >= B.n_docs) return;
    if (blockIdx.y == 0) decode_str_body<OC_KEY_STR>(B, L.s);
    else decode_str_body<OC_MARK_NAME>(B, L.s);
    return;
  }
  const u32 wave = sd_uniform(threadIdx.x >> 6);   // uniform on purpose: descriptors, cursors and loop conditions stay scalar
  const u32 d = blockIdx.x * SD_WAVES + wave;
  if (d >= B.n_docs || B.status[d] != 0) return;
  SdWave& w = L.w[wave];
  // (measured and dropped, round 3: one wave taking all 13 columns of a small document one after the other — 90 VGPRs,
  // 5 waves / SIMD; the 700,000 change chunks of the headline merge 11.7 -> 11.0 ms, configs[1] 5.0 -> 5.7 ms)
  switch (blockIdx.y - 2) {
#define DC(i, col) case i: sd_decode_column<col>(B, d, w); break;
    DC(0, OC_OBJ_ACTOR) DC(1, OC_OBJ_CTR) DC(2, OC_KEY_ACTOR) DC(3, OC_KEY_CTR) DC(4, OC_ID_ACTOR) DC(5, OC_ID_CTR) DC(6, OC_INSERT)
    DC(7, OC_ACTION) DC(8, OC_VAL_META) DC(9, OC_SUCC_NUM) DC(10, OC_SUCC_ACTOR) DC(11, OC_SUCC_CTR) DC(12, OC_EXPAND)
#undef DC
    default: break;
  }
}

// the deferred runs: a block per piece of at most 64 Ki rows
__global__ void __launch_bounds__(AMH_BLOCK) g_fill_runs(BatchDev B) {
  u32 n = *B.fr_n;
  if (n > B.fr_cap) n = B.fr_cap;
  const u32 tid = threadIdx.x, nth = AMH_BLOCK;
  for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
    const u32* it = B.fr_items + 8 * (size_t)i;
    const u32 col = it[0], d = it[1], o = it[2], cnt = it[3], a = it[4], b = it[5], c0 = it[6], st = it[7];
    switch (col) {
#define X(i_, c_) case c_: for (u32 k = tid; k < cnt; k += nth) sd_store_row<c_>(B, d, o + k, a, b, c0 + st * k); break;
      X(0, OC_OBJ_ACTOR) X(1, OC_OBJ_CTR) X(2, OC_KEY_ACTOR) X(3, OC_KEY_CTR) X(4, OC_ID_ACTOR) X(5, OC_ID_CTR) X(6, OC_INSERT)
      X(7, OC_ACTION) X(8, OC_VAL_META) X(9, OC_SUCC_NUM) X(10, OC_SUCC_ACTOR) X(11, OC_SUCC_CTR) X(12, OC_EXPAND)
#undef X
      default: break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// A column of one document over many waves.  One wave per column leaves a 10M-op document (columns of megabytes) to 15
// waves walking 256 bytes at a time.  A run-length stream cannot be entered anywhere: which tokens are headers is a chain
// from the start.  But the chain is cheap to resolve for EVERY possible entry of a 4 KiB segment at once:
//   chain    (wave per segment) token starts by ballots; every token, taken as a header, names the next header
//            (+2 run / null run, +1+n literal run); pointer doubling over the segment's tokens in LDS (12 rounds) turns
//            that into the exit of the chain entered at each token: how many tokens of the following segments are
//            spoken for (a run's value, or the rest of a literal run).  Exits go to sg_exit.
//   resolve  (thread per column) the true entry of every segment: one dependent lookup per segment from the column start;
//            then where every range starts (behind a pending run value) and ends.
//   measure  (wave per range) the decode's own group code in dry mode: rows and running sum of the range.
//   prefix   (thread per column) exclusive sums: every range's absolute row, running sum, boolean token parity.
//   decode   (wave per range) sd_decode_range with the full entry state.  The count pass takes the totals.
// Boolean columns have no chain (every token is a count): they only use measure / prefix / decode.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 bc_slot_col(u32 slot) {
  const u32 cols[13] = {OC_OBJ_ACTOR, OC_OBJ_CTR, OC_KEY_ACTOR, OC_KEY_CTR, OC_ID_ACTOR, OC_ID_CTR, OC_INSERT, OC_ACTION, OC_VAL_META, OC_SUCC_NUM, OC_SUCC_ACTOR, OC_SUCC_CTR, OC_EXPAND};
  return cols[slot];
}
#define BC_SLOT_ID_CTR 5u
#define BC_SLOT_SUCC_NUM 9u
// thread per (document, decode slot)
__global__ void __launch_bounds__(AMH_BLOCK) g_bc_find(BatchDev B) {
  u32 i = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (i >= B.n_docs * 13u) return;
  u32 d = i / 13u, slot = i % 13u;
  B.bc_index[i] = AMH_NULL32;
  if (B.status[d] != 0 || B.desc[d].chunk_len == 0) return;
  const ColRef cr = B.desc[d].ocol[bc_slot_col(slot)];
  if (cr.len < AMH_BIGCOL) return;
  u32 nseg = (cr.len + AMH_SEG - 1) / AMH_SEG;
  u32 e = atomicAdd(&B.bc_n[0], 1u);
  u32 s0 = atomicAdd(&B.bc_n[1], nseg);
  u32 eb = atomicAdd(&B.bc_n[2], cr.len);
  if (e >= B.bc_cap_cols || s0 + nseg > B.bc_cap_segs || eb + cr.len > B.bc_cap_bytes) return;   // cannot happen: sized from the input lengths
  B.bc_doc[e] = d; B.bc_slot[e] = slot; B.bc_seg0[e] = s0; B.bc_nseg[e] = nseg; B.bc_ebase[e] = eb; B.bc_bad[e] = 0;
  for (u32 k = 0; k < nseg; k++) B.sg_col[s0 + k] = e;
  B.bc_index[i] = e;
}
struct BcWave {
  u32 tile[SD_TILE / 4 + 4];
  unsigned short tokpos[AMH_SEG];
  u32 nxt[AMH_SEG];
};
#define BC_WAVES 2u
__global__ void __launch_bounds__(64 * BC_WAVES) g_bc_chain(BatchDev B) {
  __shared__ BcWave L[BC_WAVES];
  const u32 wave = sd_uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 seg = blockIdx.x * BC_WAVES + wave;
  u32 nsegs = B.bc_n[1];
  if (nsegs > B.bc_cap_segs) nsegs = B.bc_cap_segs;
  if (seg >= nsegs) return;
  BcWave& W = L[wave];
  const u32 e = B.sg_col[seg], d = B.bc_doc[e], slot = B.bc_slot[e];
  const ColRef cr = B.desc[d].ocol[bc_slot_col(slot)];
  const u8* src = B.arena + B.desc[d].chunk_off;
  const u32 k = seg - B.bc_seg0[e];
  const u32 cb = cr.off, ce = cr.off + cr.len;
  const u32 b = cb + k * AMH_SEG, en = b + AMH_SEG < ce ? b + AMH_SEG : ce;
  // ---- token starts: a byte starts a token if it is the column's first or the byte in front has no continuation bit
  u32 ntok = 0;
  u32 prevb = b > cb ? src[b - 1] : 0u;   // byte in front of the current tile
  const u32 a0 = b & ~3u;
  for (u32 tb = a0; tb < en; tb += SD_TILE) {
    sd_lds_sync();
    W.tile[lane] = *(const u32*)(src + tb + 4 * lane);   // the arena is padded: reading past the column is harmless
    sd_lds_sync();
    const u8* t8 = (const u8*)W.tile;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      const u32 bi = 64 * j + lane, pos = tb + bi;
      const u32 pb = bi ? t8[bi - 1] : prevb;
      const bool st = pos >= b && pos < en && (pos == cb || !(pb & 0x80));
      const u64 mk = __ballot(st);
      if (st) W.tokpos[ntok + sd_mbcnt(mk)] = (unsigned short)(pos - b);
      ntok += (u32)__builtin_popcountll(mk);
    }
    prevb = sd_uniform((u32)t8[SD_TILE - 1]);
  }
  sd_lds_sync();
  // ---- every token as a header: where the next header is
  const bool is_bool = slot == 6 || slot == 12;
  if (!is_bool) {
    for (u32 j = lane; j < ntok; j += 64) {
      const u8* p = src + b + W.tokpos[j];
      const u8* lim = src + ce;
      u64 r = 0; u32 shift = 0; u8 by = 0;
      for (u32 i = 0; i < 10 && p + i < lim; i++) { by = p[i]; if (shift < 64) r |= (u64)(by & 0x7f) << shift; shift += 7; if (!(by & 0x80)) break; }
      if (shift < 64 && (by & 0x40)) r |= ~(u64)0 << shift;
      const i64 sv = (i64)r;
      u64 hop = 2;
      if (sv < 0) { u64 n = (u64)0 - (u64)sv; hop = 1 + (n < 0x3FFFFFF0ull ? n : 0x3FFFFFF0ull); }
      const u64 x = (u64)j + hop;
      u32 v;
      if (x < ntok) v = (u32)x;
      else { u64 skip = x - ntok; if (skip > 0x3FFFFFFFull) skip = 0x3FFFFFFFull; v = 0x80000000u | ((u32)skip << 1) | ((sv >= 0 && skip == 1) ? 1u : 0u); }
      W.nxt[j] = v;
    }
    sd_lds_sync();
    for (u32 round = 0; round < 12; round++) {   // pointer doubling: 2^12 = AMH_SEG tokens at most
      for (u32 j = lane; j < ntok; j += 64) {
        u32 v = W.nxt[j];
        if (!(v & 0x80000000u)) W.nxt[j] = W.nxt[v];   // in place: whatever a racing lane left there is further along the same chain
      }
      sd_lds_sync();
    }
    u32* ex = B.sg_exit + B.bc_ebase[e] + k * AMH_SEG;
    for (u32 j = lane; j < ntok; j += 64) ex[j] = W.nxt[j] & 0x7FFFFFFFu;
  }
  if (lane == 0) {
    B.sg_ntok[seg] = ntok;
    B.sg_first[seg] = ntok ? b + W.tokpos[0] : AMH_NULL32;
    B.sg_second[seg] = ntok > 1 ? b + W.tokpos[1] : AMH_NULL32;
  }
}
// thread per big column
__global__ void __launch_bounds__(64) g_bc_resolve(BatchDev B) {
  u32 e = blockIdx.x * 64 + threadIdx.x;
  u32 ncols = B.bc_n[0];
  if (ncols > B.bc_cap_cols) ncols = B.bc_cap_cols;
  if (e >= ncols) return;
  const u32 slot = B.bc_slot[e], s0 = B.bc_seg0[e], ns = B.bc_nseg[e];
  const bool is_bool = slot == 6 || slot == 12;
  const ColRef cr = B.desc[B.bc_doc[e]].ocol[bc_slot_col(slot)];
  u32 skip = 0, rv = 0;
  for (u32 k = 0; k < ns; k++) {
    const u32 seg = s0 + k;
    B.sg_entry[seg] = (skip << 1) | rv;
    if (is_bool) continue;
    const u32 nt = B.sg_ntok[seg];
    if (skip < nt) { u32 x = B.sg_exit[B.bc_ebase[e] + k * AMH_SEG + skip]; skip = x >> 1; rv = x & 1; }
    else { skip -= nt; if (skip == 0) rv = 0; }
  }
  // ranges: a segment's range starts at its first token, or behind it when that token is the value of a run whose header
  // lies in an earlier segment; it ends where the next range starts
  u32 next = cr.off + cr.len;
  for (u32 k = ns; k-- > 0;) {
    const u32 seg = s0 + k;
    const u32 nt = B.sg_ntok[seg], en = B.sg_entry[seg];
    u32 start;
    if (nt == 0) start = next;
    else if (en & 1) start = nt > 1 ? B.sg_second[seg] : next;
    else start = B.sg_first[seg];
    B.sg_pos[seg] = start;
    B.sg_end[seg] = next;
    next = start;
  }
}
template <u32 COL> struct SdDryF {
  const BatchDev& B;
  u32 d, mx;
  __device__ __forceinline__ bool wide(const SdStream& S) const { return SdCol<COL>::HAS_ACC && S.acc >= (1u << 30); }
  template <bool WIDE>
  __device__ __forceinline__ u32 group(SdWave& W, SdStream& S, u32 lane, u32 m, u32 b0, u32 nb, i32& fail) {
    return sd_do_group<COL, WIDE, true>(B, d, W, S, lane, m, b0, nb, 0u, 0u, 0u, mx, fail);
  }
};
// the reader's state at the start of a range, from the resolved entry of its segment
__device__ __forceinline__ u64 bc_entry_lit(u32 entry) { return (entry & 1) ? 0ull : (u64)(entry >> 1); }
template <u32 COL>
__device__ __forceinline__ void bc_measure_range(const BatchDev& B, u32 seg, u32 e, u32 d, SdWave& W) {
  const u32 lane = threadIdx.x & 63;
  SdStream S;
  S.src = B.arena + B.desc[d].chunk_off;
  S.pos = B.sg_pos[seg]; S.end = B.sg_end[seg];
  S.want = 0xFFFFFFFFu; S.row = 0; S.lit = bc_entry_lit(B.sg_entry[seg]); S.btok = 0; S.acc = 0;
  SdDryF<COL> f{B, d, 0u};
  i32 fail = 0;
  if (S.pos < S.end) fail = sd_walk(W, S, lane, f, AMH_E_READ_OP);
  const bool last = seg + 1 == B.bc_seg0[e] + B.bc_nseg[e];
  if (lane == 0) {
    B.sg_rows[seg] = S.row; B.sg_acc[seg] = S.acc; B.sg_btok[seg] = S.btok;
    if (fail || S.row == 0xFFFFFFFFu || (last && S.lit != 0)) B.bc_bad[e] = 1;   // malformed, absurdly long, or the column ends inside a literal run
  }
}
#define BC_COLS(X) X(0, OC_OBJ_ACTOR) X(1, OC_OBJ_CTR) X(2, OC_KEY_ACTOR) X(3, OC_KEY_CTR) X(4, OC_ID_ACTOR) X(5, OC_ID_CTR) X(6, OC_INSERT) \
  X(7, OC_ACTION) X(8, OC_VAL_META) X(9, OC_SUCC_NUM) X(10, OC_SUCC_ACTOR) X(11, OC_SUCC_CTR) X(12, OC_EXPAND)
__global__ void __launch_bounds__(64 * SD_WAVES) g_bc_measure(BatchDev B) {
  __shared__ SdWave L[SD_WAVES];
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 seg = blockIdx.x * SD_WAVES + wave;
  u32 nsegs = B.bc_n[1];
  if (nsegs > B.bc_cap_segs) nsegs = B.bc_cap_segs;
  if (seg >= nsegs) return;
  const u32 e = B.sg_col[seg], d = B.bc_doc[e];
  switch (B.bc_slot[e]) {
#define X(i, col) case i: bc_measure_range<col>(B, seg, e, d, L[wave]); break;
    BC_COLS(X)
#undef X
    default: break;
  }
}
// thread per big column: exclusive sums over its ranges, totals
__global__ void __launch_bounds__(64) g_bc_prefix(BatchDev B) {
  u32 e = blockIdx.x * 64 + threadIdx.x;
  u32 ncols = B.bc_n[0];
  if (ncols > B.bc_cap_cols) ncols = B.bc_cap_cols;
  if (e >= ncols) return;
  const u32 s0 = B.bc_seg0[e], ns = B.bc_nseg[e];
  u64 rows = 0, acc = 0; u32 btok = 0;
  for (u32 k = 0; k < ns; k++) {
    const u32 seg = s0 + k;
    const u64 r = B.sg_rows[seg], a = B.sg_acc[seg]; const u32 t = B.sg_btok[seg];
    B.sg_rows[seg] = rows; B.sg_acc[seg] = acc; B.sg_btok[seg] = btok;
    rows += r; acc += a; btok += t;
  }
  B.bc_rows[e] = rows; B.bc_acc[e] = acc;
}
__global__ void __launch_bounds__(64 * SD_WAVES) g_bc_decode(BatchDev B) {
  __shared__ SdWave L[SD_WAVES];
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 seg = blockIdx.x * SD_WAVES + wave;
  u32 nsegs = B.bc_n[1];
  if (nsegs > B.bc_cap_segs) nsegs = B.bc_cap_segs;
  if (seg >= nsegs) return;
  const u32 e = B.sg_col[seg], d = B.bc_doc[e];
  if (B.status[d] != 0) return;
  const u32 pos = B.sg_pos[seg], end = B.sg_end[seg];
  const bool last = seg + 1 == B.bc_seg0[e] + B.bc_nseg[e];
  if (pos >= end && !last) return;
  const u64 r64 = B.sg_rows[seg];
  const u32 row0 = r64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)r64;
  const u64 lit = bc_entry_lit(B.sg_entry[seg]), acc = B.sg_acc[seg];
  const u32 btok = B.sg_btok[seg];
  switch (B.bc_slot[e]) {
#define X(i, col) case i: sd_decode_range<col>(B, d, L[wave], pos, end, row0, lit, btok, acc, last); break;
    BC_COLS(X)
#undef X
    default: break;
  }
}

// ------------------------------------------------------------------------------------------
// Count pass (doc_op_columns.rs:327-331; device form of k_count): rows = items of the id-counter column, succ entries
// = sum of the first `rows` succ counts, changes = values of the change actor column up to its first null, deps = sum
// of the first `changes` dep counts.  The same walk without the expand: one wave per (document, half) — the op
// columns and the change columns are independent.
// ------------------------------------------------------------------------------------------
enum { SD_CNT_ITEMS = 0, SD_CNT_SUM = 1, SD_CNT_UNTIL_NULL = 2 };
template <int MODE> struct SdCountF {
  i32 err;            // what a malformed token is called here (ReadOp / ReadChange)
  u64 sum;            // SD_CNT_SUM: per lane
  bool stopped;       // SD_CNT_UNTIL_NULL: the first null ended the column
  u64 limit;          // per-document limit on a count (BatchDev::max_doc_rows)
  __device__ __forceinline__ bool wide(const SdStream&) const { return false; }
  template <bool WIDE>
  __device__ __forceinline__ u32 group(SdWave& W, SdStream& S, u32 lane, u32 m, u32 b0, u32 nb, i32& fail) {
    SdItems<WIDE> I;
    bool terr;
    u64 lit_new; u32 btok_new;
    const u32 rem = S.want - S.row;
    const u32 c = sd_group<MODE == SD_CNT_ITEMS ? SD_RLE_S : SD_RLE_U, WIDE>(W, S, lane, m, b0, nb, I, terr, lit_new, btok_new);
    if (c == 0) return 0;
    const u32 reach = I.rs < rem ? (rem - I.rs < I.cnt ? rem - I.rs : I.cnt) : 0;
    i32 ecode = terr ? err : 0;
    u32 R = I.total < rem ? I.total : rem;
    u64 live = ~(u64)0;      // lanes in front of the point where the reader stops
    if (MODE == SD_CNT_ITEMS) {
      if (!ecode && I.isnull && I.cnt) ecode = err;   // a null id is "unexpected null" (opid.rs:114-117)
    } else if (MODE == SD_CNT_SUM) {
      if (!ecode && reach) {
        if (I.isnull) ecode = err;
        else if ((u64)I.u > limit) ecode = AMH_E_TOO_LARGE;
        else sum += (u64)I.u * reach;
      }
    } else {
      const u64 nul = __ballot(I.isnull && I.cnt != 0);
      if (nul) {   // the first null ends the column: nothing behind it is read
        const u32 L = (u32)__builtin_ctzll(nul);
        live = sd_low_bits(L + 1);
        const u32 before = sd_bcast(I.rs, L);
        if (before < rem) { R = before; stopped = true; }
      }
    }
    const u64 bad = __ballot(ecode != 0 && I.rs < rem) & live;
    if (bad) { fail = (i32)sd_bcast((u32)ecode, (u32)__builtin_ctzll(bad)); return c; }
    S.row += R;
    if (stopped) S.want = S.row;
    S.lit = lit_new;
    return c;
  }
};

template <int MODE>
__device__ __forceinline__ i32 sd_count_column(const BatchDev& B, u32 d, SdWave& W, ColRef cr, u32 want, i32 err, u32& out) {
  const u32 lane = threadIdx.x & 63;
  SdStream S;
  S.src = B.arena + B.desc[d].chunk_off;
  S.pos = cr.off; S.end = cr.off + cr.len;
  S.want = want; S.row = 0; S.lit = 0; S.btok = 0; S.acc = 0;
  SdCountF<MODE> f{err, 0, false, (u64)B.max_doc_rows};
  out = 0;
  if (want == 0) return 0;
  i32 fail = sd_walk(W, S, lane, f, err);
  if (fail) return fail;
  if (MODE == SD_CNT_SUM) {
    if (S.row < want) return err;   // fewer values than rows
    u64 tot = sd_bcast64(sd_scan_add64(f.sum), 63);
    if (tot > B.max_doc_rows) return AMH_E_TOO_LARGE;
    out = (u32)tot;
  } else {
    if (S.row > B.max_doc_rows) return AMH_E_TOO_LARGE;   // want = the limit + 1
    if (!f.stopped && S.lit != 0) return err;                    // the column ends inside a literal run
    out = S.row;
  }
  return 0;
}

__global__ void __launch_bounds__(64 * SD_WAVES) g_count_seg(BatchDev B) {
  __shared__ SdWave W[SD_WAVES];
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 d = blockIdx.x * SD_WAVES + wave;
  if (d >= B.n_docs || B.status[d] != 0 || B.desc[d].chunk_len == 0) return;
  const u32 lane = threadIdx.x & 63;
  const DocDesc& D = B.desc[d];
  SdWave& w = W[wave];
  u32 a = 0, b = 0;
  i32 fail;
  if (blockIdx.y == 0) {
    // columns that were cut into ranges: the totals are known (a column with a malformed token in it, or a succ-count
    // column that does not hold exactly one value per row, is walked as before: the serial reader's error rules)
    const u32 ei = bc_is_big(B, D.ocol[OC_ID_CTR].len) ? B.bc_index[d * 13u + BC_SLOT_ID_CTR] : AMH_NULL32;
    const u32 es = bc_is_big(B, D.ocol[OC_SUCC_NUM].len) ? B.bc_index[d * 13u + BC_SLOT_SUCC_NUM] : AMH_NULL32;
    bool have_rows = false;
    fail = 0;
    if (ei != AMH_NULL32 && !B.bc_bad[ei]) {
      const u64 r = B.bc_rows[ei];
      if (r > B.max_doc_rows) fail = AMH_E_TOO_LARGE; else { a = (u32)r; have_rows = true; }
    }
    if (!fail && !have_rows) fail = sd_count_column<SD_CNT_ITEMS>(B, d, w, D.ocol[OC_ID_CTR], B.max_doc_rows + 1, AMH_E_READ_OP, a);
    if (!fail) {
      if (es != AMH_NULL32 && !B.bc_bad[es] && B.bc_rows[es] == (u64)a && B.bc_acc[es] <= (u64)B.max_doc_rows) b = (u32)B.bc_acc[es];
      else fail = sd_count_column<SD_CNT_SUM>(B, d, w, D.ocol[OC_SUCC_NUM], a, AMH_E_READ_OP, b);
    }
  } else {
    fail = sd_count_column<SD_CNT_UNTIL_NULL>(B, d, w, D.ccol[CC_ACTOR], B.max_doc_rows + 1, AMH_E_READ_CHANGE, a);
    if (!fail) fail = sd_count_column<SD_CNT_SUM>(B, d, w, D.ccol[CC_DEPS_NUM], a, AMH_E_READ_CHANGE, b);
  }
  if (lane == 0) {
    if (fail) amh_fail(B, d, fail);
    else if (blockIdx.y == 0) { B.cnt[d].n_rows = a; B.cnt[d].n_succ = b; }
    else { B.cnt[d].n_chg = a; B.cnt[d].n_deps = b; }
  }
}
