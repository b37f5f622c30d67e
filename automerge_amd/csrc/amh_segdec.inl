// Segment-parallel column decode (device only; included by amh_kernels.hip).
//
// One wave cooperates on ONE column stream of one document (rle.rs:175-210, delta.rs:79-95, boolean.rs:84-101) instead
// of one lane walking it: the 64 lanes of a wave then do the same work on 64 tokens / 64 rows at a time, every load and
// store is a coalesced wave access, and a large document uses a whole wave per column instead of one lane.
//
//   tile    256 column bytes -> LDS (one coalesced dword load per lane)
//   tokens  LEB128 boundaries = bytes without the continuation bit: 4 ballots give the rank of every token end, the end
//           offsets go to LDS in token order (a token that does not end inside the tile is left for the next tile, which
//           starts at its first byte: nothing but four scalars is carried between tiles)
//   values  lane t decodes token t out of LDS (one unaligned 8-byte peek, the 7-bit groups packed in registers)
//   headers which tokens are run headers is a chain (a header says where the next one is).  Run and null headers hop by
//           exactly 2 tokens, so from a header at p the chain visits p, p + 2, ... until it meets a NEGATIVE token at that
//           parity: one ballot of the signs + scalar bit arithmetic resolves every run / null header up to the next
//           literal header at once; the scalar loop iterates per LITERAL header only
//   items   run header -> (count, value of the next token); null header -> (count = next token, null); literal value ->
//           (1, own value).  An exclusive wave scan of the counts gives every item its first row; a second scan carries
//           the running sum of a delta column (or the raw-value offsets, or the succ bases) across the items
//   expand  64 rows at a time: item starts are scattered into an LDS row of flags, an inclusive max-scan turns them into
//           "item of this row", the item's fields come over by ds_bpermute, one coalesced store per output array
//
// Error semantics are the serial decoder's: a malformed token only fails the document if the decoder would have read
// it before producing the last row it was asked for (row index of the token < rows wanted); the first such token in
// stream order decides the error kind.  A column that ends early yields nulls where the reference's iterator does
// (`None`), and fails where it is required.
#pragma once

#define SD_TILE 256u
#define SD_WAVES 4u            // waves (= column streams) per workgroup

struct SdWave {
  u32 tile[SD_TILE / 4 + 4];   // + 16 bytes: the 8-byte peek of the last token may read past the tile
  unsigned short tokend[SD_TILE];
  u32 xs[64];
};

#include "amh_segdec_prims.inl"

// what the wave carries from tile to tile
struct SdStream {
  const u8* src;      // arena address of the document's chunk
  u32 pos, end;       // byte cursor / end of the column, chunk relative
  u32 want;           // rows the caller wants
  u32 row;            // rows produced
  u64 lit;            // values of an open literal run still to come
  u32 btok;           // boolean columns: tokens consumed so far (parity = value)
  i32 err;            // first error (0 = none)
};

// One group of <= 64 tokens, one per lane, classified: what every lane's token contributes.
struct SdItems {
  u32 cnt;            // rows of this lane's item (0: the token is a header of a literal run, a run's value, ...), clamped to the rows still wanted
  u32 rs;             // rows of the group in front of the item (exclusive scan of cnt)
  u32 total;          // rows of the whole group (uniform; may exceed what is wanted)
  bool isnull;
  u64 u;              // value as unsigned LEB128
  i64 s;              // value as signed LEB128
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

// token t of the tile: value both ways; ok_u / ok_s = a valid u64 / i64 for the lenient readers (decodable_impls.rs:58-96)
__device__ __forceinline__ void sd_token(const SdWave& W, u32 t, u32 skip, u64& u, i64& s, bool& ok_u, bool& ok_s) {
  u32 e = W.tokend[t];
  u32 b0 = t ? (u32)W.tokend[t - 1] + 1 : skip;
  u32 nb = e - b0 + 1;
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

// The next group of tokens of the tile, from token t0.  Returns the number of tokens consumed (uniform; 0 = the group
// ends in a header whose value lies in the next tile, or nothing is left).  terr = this lane's token (or the value token
// behind its header) is malformed for the role it plays.
template <int KIND>
__device__ __forceinline__ u32 sd_group(const SdWave& W, SdStream& S, u32 lane, u32 t0, u32 ntok, u32 skip, SdItems& I, bool& terr) {
  const u32 m = ntok - t0 < 64 ? ntok - t0 : 64;
  u64 u = 0; i64 s = 0; bool ok_u = true, ok_s = true;
  if (lane < m) sd_token(W, t0 + lane, skip, u, s, ok_u, ok_s);
  const u32 rem = S.want - S.row;
  u32 c;
  u64 cnt64 = 0;
  bool isnull = false;
  terr = false;
  if (KIND == SD_BOOL) {   // every token is a count; the value alternates, starting with false (boolean.rs:84-101)
    c = m;
    if (lane < m) { cnt64 = u; terr = !ok_u; }
    I.u = (S.btok + lane) & 1;   // the value the count stands for
    I.s = 0;
    S.btok += c;
  } else {
    const u64 NEG = __ballot(lane < m && s < 0);
    u64 HDR = 0, LITH = 0, LITV = 0;
    u64 lit = S.lit;
    u32 p = 0;
    if (lit) { u32 k = lit < m ? (u32)lit : m; LITV = sd_low_bits(k); lit -= k; p = k; }
    while (p < m) {
      u64 par = 0x5555555555555555ull << (p & 1), above = ~(u64)0 << p;
      u64 cand = NEG & par & above;
      if (!cand) {   // run / null headers to the end of the group; each needs its value token inside the group
        u64 hm = par & above & sd_low_bits(m - 1);
        HDR |= hm;
        p += 2 * (u32)__builtin_popcountll(hm);
        break;
      }
      u32 q = (u32)__builtin_ctzll(cand);
      HDR |= par & above & sd_low_bits(q);
      LITH |= (u64)1 << q;
      u64 nlit = (u64)0 - (u64)sd_bcast64((u64)s, q);
      u32 room = m - (q + 1);
      u32 k = nlit < room ? (u32)nlit : room;
      if (k) LITV |= sd_low_bits(k) << (q + 1);
      lit = nlit - k;
      p = q + 1 + k;
    }
    c = p < m ? p : m;
    S.lit = lit;
    const bool is_hdr = (HDR >> lane) & 1, is_lith = (LITH >> lane) & 1, is_litv = (LITV >> lane) & 1;
    // the token behind a run / null header
    u64 nu = (u64)__shfl_down((u32)u, 1, 64) | ((u64)__shfl_down((u32)(u >> 32), 1, 64) << 32);
    i64 ns = (i64)((u64)__shfl_down((u32)s, 1, 64) | ((u64)__shfl_down((u32)((u64)s >> 32), 1, 64) << 32));
    bool n_ok_u = __shfl_down((int)ok_u, 1, 64), n_ok_s = __shfl_down((int)ok_s, 1, 64);
    if (is_hdr) {
      if (s > 0) { cnt64 = (u64)s; I.u = nu; I.s = ns; terr = !ok_s || !(KIND == SD_RLE_S ? n_ok_s : n_ok_u); }
      else { cnt64 = nu; isnull = true; I.u = 0; I.s = 0; terr = !ok_s || !n_ok_u || nu > (u64)INT64_MAX; }
    } else if (is_litv) {
      cnt64 = 1; I.u = u; I.s = s; terr = !(KIND == SD_RLE_S ? ok_s : ok_u);
    } else if (is_lith) {
      terr = !ok_s;
    }
    if (lane >= c) { cnt64 = 0; terr = false; }
  }
  I.isnull = isnull;
  I.cnt = cnt64 < rem ? (u32)cnt64 : rem;
  u32 inc = sd_scan_add(I.cnt);
  I.rs = inc - I.cnt;
  I.total = sd_bcast(inc, 63);
  return c;
}

// rows [r0, r0 + 64) of a group: the lane (= item) each row belongs to.  Needs I.total > r0.
__device__ __forceinline__ u32 sd_row_item(SdWave& W, const SdItems& I, u32 lane, u32 r0) {
  sd_lds_sync();
  W.xs[lane] = 0;
  sd_lds_sync();
  const bool has = I.cnt != 0;
  if (has && I.rs >= r0 && I.rs < r0 + 64) W.xs[I.rs - r0] = lane + 1;
  sd_lds_sync();
  u64 cover = __ballot(has && I.rs <= r0 && r0 - I.rs < I.cnt);   // the item that holds row r0: exactly one
  u32 f = W.xs[lane];
  if (lane == 0) f = (u32)__builtin_ctzll(cover) + 1;
  return sd_scan_max(f) - 1;
}

// One column stream.  COL selects what an item means and where its rows go.
template <u32 COL>
__device__ __forceinline__ void sd_decode_column(const BatchDev& B, u32 d, SdWave& W) {
  constexpr bool SUCC_ENTRY = COL == OC_SUCC_ACTOR || COL == OC_SUCC_CTR;
  constexpr bool IS_DELTA = COL == OC_KEY_CTR || COL == OC_ID_CTR || COL == OC_SUCC_CTR;
  constexpr bool IS_BOOL = COL == OC_INSERT || COL == OC_EXPAND;
  constexpr int KIND = IS_BOOL ? SD_BOOL : (IS_DELTA ? SD_RLE_S : SD_RLE_U);
  // a column that ends early (or is absent): the reference's iterator yields None, which these accept as null
  constexpr bool NULLABLE = COL == OC_OBJ_ACTOR || COL == OC_OBJ_CTR || COL == OC_KEY_ACTOR || COL == OC_KEY_CTR ||
                            COL == OC_ID_ACTOR || COL == OC_ID_CTR || COL == OC_EXPAND;
  const u32 lane = threadIdx.x & 63;
  const DocDesc& D = B.desc[d];
  const u32 want = SUCC_ENTRY ? B.cnt[d].n_succ : B.cnt[d].n_rows;
  if (want == 0) return;
  const u32 obase = SUCC_ENTRY ? B.succ_base[d] : B.row_base[d];
  SdStream S;
  S.src = B.arena + D.chunk_off;
  S.pos = D.ocol[COL].off;
  S.end = S.pos + D.ocol[COL].len;
  S.want = want; S.row = 0; S.lit = 0; S.btok = 0; S.err = 0;
  u64 acc = 0;                 // delta: running value; value meta: raw offset; succ count: entries so far
  u32 mx = 0;
  const u64 raw_len = D.ocol[OC_VAL_RAW].len;
  const u32 sbase = COL == OC_SUCC_NUM ? B.succ_base[d] : 0;
  i32 fail = 0;
  while (S.row < want && S.pos < S.end && !fail) {
    u32 tb, skip, avail;
    sd_lds_sync();   // the previous tile has been read
    const u32 ntok = sd_load_tile(W, S, lane, tb, skip, avail);
    u32 t0 = 0;
    while (t0 < ntok && S.row < want) {
      SdItems I;
      bool terr;
      const u32 rem = want - S.row;
      const u32 c = sd_group<KIND>(W, S, lane, t0, ntok, skip, I, terr);
      if (c == 0) break;
      // ---- what the item's rows hold -----------------------------------------------------------
      const u32 cnt = I.cnt;
      const u32 reach = I.rs < rem ? (rem - I.rs < cnt ? rem - I.rs : cnt) : 0;   // rows of the item the decoder delivers
      i32 ecode = terr ? AMH_E_READ_OP : 0;
      u32 e0 = 0, e1 = 0, e2 = 0, step = 0;
      if (IS_DELTA) {
        u64 tot = I.isnull ? 0 : (u64)I.s * cnt;
        u64 inc = sd_scan_add64(tot);
        u64 first = acc + (inc - tot) + (u64)I.s;           // value of the item's first row
        if (I.isnull) { e2 = AMH_NULL32; step = 0; if (SUCC_ENTRY && reach && !ecode) ecode = AMH_E_READ_OP; }
        else {
          e2 = (u32)first; step = (u32)(u64)I.s;
          if (reach && !ecode) {
            i64 lo = (i64)first, hi = (i64)(first + (u64)I.s * (reach - 1));
            if (lo > hi) { i64 t = lo; lo = hi; hi = t; }
            if (lo < 0 || hi >= 0xFFFFFFFFll) ecode = AMH_E_READ_OP;   // key.rs:172-179, opid.rs:107-113
            else if (COL != OC_KEY_CTR && (u32)hi > mx) mx = (u32)hi;
          }
        }
        acc += sd_bcast64(inc, 63);
      } else if (IS_BOOL) {
        e0 = (u32)I.u;
      } else if (COL == OC_VAL_META) {
        u64 l = ((I.u & 15) >= 3) ? (I.u >> 4) : 0;
        u64 tot = l * cnt;
        u64 inc = sd_scan_add64(tot);
        u64 first = acc + (inc - tot);
        e0 = (u32)I.u; e1 = (u32)(I.u >> 32); e2 = (u32)first; step = (u32)l;
        if (reach && !ecode) {
          if (I.isnull) ecode = AMH_E_READ_OP;
          else if ((I.u & 15) >= 3) {   // null / bool never touch the raw column (value.rs:221-224)
            u64 last = first + l * (reach - 1);
            if (l > raw_len || last > raw_len - l) ecode = AMH_E_READ_OP;
          }
        }
        acc += sd_bcast64(inc, 63);
      } else if (COL == OC_SUCC_NUM) {
        u32 v = (u32)I.u;
        u32 tot = v * cnt;
        u32 inc = sd_scan_add(tot);
        e0 = v; e2 = sbase + (u32)acc + (inc - tot); step = v;
        if (reach && !ecode && I.isnull) ecode = AMH_E_READ_OP;
        acc += sd_bcast(inc, 63);
      } else {   // plain RLE of u64: obj actor / ctr, key actor, id actor, action, succ actor
        e0 = I.isnull ? AMH_NULL32 : (u32)I.u;
        if (reach && !ecode) {
          if (COL == OC_ACTION) { if (I.isnull) ecode = AMH_E_READ_OP; else if (I.u > 7) ecode = AMH_E_INVALID_OP; }
          else if (COL == OC_SUCC_ACTOR) { if (I.isnull) ecode = AMH_E_READ_OP; else if (I.u >= 0xFFFFFFFFull) ecode = AMH_E_MISSING_ACTOR; }
          else if (!I.isnull && I.u >= 0xFFFFFFFFull) ecode = COL == OC_OBJ_CTR ? AMH_E_READ_OP : AMH_E_MISSING_ACTOR;
        }
      }
      // a malformed token counts only if the decoder gets to it: rows in front of it < rows wanted
      const u64 bad = __ballot(ecode != 0 && I.rs < rem);
      if (bad) { fail = (i32)sd_bcast((u32)ecode, (u32)__builtin_ctzll(bad)); break; }
      // ---- expand ------------------------------------------------------------------------------
      const u32 R = I.total < rem ? I.total : rem;
      for (u32 r0 = 0; r0 < R; r0 += 64) {
        const u32 t = sd_row_item(W, I, lane, r0);
        const u32 my = r0 + lane;
        const u32 k = my - (u32)__shfl((int)I.rs, (int)t, 64);
        const u32 o = obase + S.row + my;
        const bool live = my < R;
        if (IS_DELTA) {
          u32 v = (u32)__shfl((int)e2, (int)t, 64) + (u32)__shfl((int)step, (int)t, 64) * k;
          if (live) {
            if (COL == OC_KEY_CTR) B.r_key_ctr[o] = v;
            else if (COL == OC_ID_CTR) { B.r_id_ctr[o] = v; B.r_doc[o] = d; }
            else B.s_ctr[o] = v;
          }
        } else if (COL == OC_VAL_META) {
          u32 a = (u32)__shfl((int)e0, (int)t, 64), b = (u32)__shfl((int)e1, (int)t, 64);
          u32 ro = (u32)__shfl((int)e2, (int)t, 64) + (u32)__shfl((int)step, (int)t, 64) * k;
          if (live) { B.r_vmeta[o] = (u64)a | ((u64)b << 32); B.r_vraw_off[o] = ro; }
        } else if (COL == OC_SUCC_NUM) {
          u32 a = (u32)__shfl((int)e0, (int)t, 64);
          u32 sb = (u32)__shfl((int)e2, (int)t, 64) + a * k;
          if (live) { B.r_succ_cnt[o] = a; B.r_succ_base[o] = sb; }
        } else {
          u32 a = (u32)__shfl((int)e0, (int)t, 64);
          if (live) {
            if (COL == OC_OBJ_ACTOR) B.r_obj_actor[o] = a;
            else if (COL == OC_OBJ_CTR) B.r_obj_ctr[o] = a;
            else if (COL == OC_KEY_ACTOR) B.r_key_actor[o] = a;
            else if (COL == OC_ID_ACTOR) B.r_id_actor[o] = a;
            else if (COL == OC_SUCC_ACTOR) B.s_actor[o] = a;
            else if (COL == OC_ACTION) B.r_action[o] = (u8)a;
            else if (COL == OC_INSERT) B.r_insert[o] = (u8)a;
            else if (COL == OC_EXPAND) B.r_expand[o] = (u8)a;
          }
        }
      }
      S.row += R;
      t0 += c;
    }
    if (fail) break;
    // next tile: from the first token that was not consumed
    u32 npos = tb + (t0 ? (u32)W.tokend[t0 - 1] + 1 : skip);
    npos = sd_uniform(npos);
    if (npos == S.pos) {
      // no token could be consumed: a token longer than the tile, one cut off by the end of the column, or a run /
      // null header whose value the column no longer holds — the serial reader fails on all of them
      if (S.row < want) fail = AMH_E_READ_OP;
      break;
    }
    S.pos = npos;
  }
  if (fail) { if (lane == 0) amh_fail(B, d, fail); return; }
  if (S.row < want) {
    // the column is exhausted.  Inside a literal run the reader fails (the announced value is missing); otherwise
    // the iterator yields None from here on
    bool ok = NULLABLE && S.lit == 0;
    if (!ok) { if (lane == 0) amh_fail(B, d, AMH_E_READ_OP); return; }
    for (u32 r = S.row + lane; r < want; r += 64) {
      u32 o = obase + r;
      if (COL == OC_OBJ_ACTOR) B.r_obj_actor[o] = AMH_NULL32;
      else if (COL == OC_OBJ_CTR) B.r_obj_ctr[o] = AMH_NULL32;
      else if (COL == OC_KEY_ACTOR) B.r_key_actor[o] = AMH_NULL32;
      else if (COL == OC_KEY_CTR) B.r_key_ctr[o] = AMH_NULL32;
      else if (COL == OC_ID_ACTOR) B.r_id_actor[o] = AMH_NULL32;
      else if (COL == OC_ID_CTR) { B.r_id_ctr[o] = AMH_NULL32; B.r_doc[o] = d; }
      else if (COL == OC_EXPAND) B.r_expand[o] = 0;
    }
  }
  if (COL == OC_ID_CTR || COL == OC_SUCC_CTR) {
    u32 m = sd_scan_max(mx);
    if (lane == 63 && m) amh_atomic_max(&B.doc_max_op[d], m);
  }
}

// blockIdx.y = column (uniform per workgroup), one wave per document
#define SD_NCOLS 13
__global__ void __launch_bounds__(64 * SD_WAVES) g_decode_seg(BatchDev B) {
  __shared__ SdWave W[SD_WAVES];
  const u32 wave = threadIdx.x >> 6;
  const u32 d = blockIdx.x * SD_WAVES + wave;
  if (d >= B.n_docs || B.status[d] != 0) return;
  SdWave& w = W[wave];
  switch (blockIdx.y) {
#define DC(i, col) case i: sd_decode_column<col>(B, d, w); break;
    DC(0, OC_OBJ_ACTOR) DC(1, OC_OBJ_CTR) DC(2, OC_KEY_ACTOR) DC(3, OC_KEY_CTR) DC(4, OC_ID_ACTOR) DC(5, OC_ID_CTR) DC(6, OC_INSERT)
    DC(7, OC_ACTION) DC(8, OC_VAL_META) DC(9, OC_SUCC_NUM) DC(10, OC_SUCC_ACTOR) DC(11, OC_SUCC_CTR) DC(12, OC_EXPAND)
#undef DC
    default: break;
  }
}
