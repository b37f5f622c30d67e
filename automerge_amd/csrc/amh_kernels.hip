// HIP launch wrappers for the batched document-load kernels (gfx950).  The per-thread bodies live
// in amh_kernels.inl; this file gives each of them a grid.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "amh_kernels.inl"
#include "amh_merge.inl"
#include "amh_apply.inl"
#include "amh_syncmsg.inl"
#include "amh_launch.h"

#define AMH_BLOCK 256

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, MI355X_MICROARCH.md), each with an L2
// of its own.  The thread-per-row / per-op kernels scatter 4-byte stores into small per-document regions (id tables,
// ops_sorted, pred lists): with the plain mapping the four blocks that hold one document's rows sit on four XCDs and
// every line of such a region is written back in pieces by four L2s.  amh_xcd_block() gives XCD x the x-th eighth of
// the index space instead, so the rows of a document meet in one L2.  Grids are rounded up to a multiple of 8.
__device__ __forceinline__ u32 amh_xcd_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
static inline u32 amh_xcd_grid(u32 n) { return (((n + AMH_BLOCK - 1) / AMH_BLOCK) + 7u) & ~7u; }

#define AMH_GLOBAL(name)                                        \
  __global__ void __launch_bounds__(AMH_BLOCK) g_##name(BatchDev B, u32 n) { \
    u32 i = blockIdx.x * AMH_BLOCK + threadIdx.x;               \
    if (i < n) name(B, i);                                      \
  }
// XCD-aware form, for the kernels it was measured to help (place 3.28 -> 2.91 ms, pred_fill 2.16 -> 1.79, succ_link
// 1.87 -> 1.73, change_view 4.08 -> 3.93, pred_slot_base 0.50 -> 0.43).  Kernels whose threads meet in per-document
// atomic counters got slower with it (succ 2.2 -> 4.7 ms, bucket 2.2 -> 3.8: a document's blocks then hit its
// counters at the same time in one L2) and keep the plain mapping.
#define AMH_GLOBAL_X(name)                                      \
  __global__ void __launch_bounds__(AMH_BLOCK) g_##name(BatchDev B, u32 n) { \
    u32 i = amh_xcd_block() * AMH_BLOCK + threadIdx.x;          \
    if (i < n) name(B, i);                                      \
  }

AMH_GLOBAL(k_parse)
AMH_GLOBAL(k_sha_doc)
AMH_GLOBAL(k_count)
AMH_GLOBAL(k_extract)
AMH_GLOBAL(k_actors)
AMH_GLOBAL(k_chg_init)
AMH_GLOBAL(k_decode_chg)
AMH_GLOBAL(k_decode_chg_cols)
AMH_GLOBAL(k_chg_patch)
AMH_GLOBAL(k_atab_base)
AMH_GLOBAL(k_atab_fill)
AMH_GLOBAL(k_decode_ops)
AMH_GLOBAL(k_rows)
AMH_GLOBAL(k_obj_order)
AMH_GLOBAL(k_succ)
AMH_GLOBAL(k_bucket)
AMH_GLOBAL(k_chg_finish)
AMH_GLOBAL_X(k_place)
AMH_GLOBAL(k_pred_slot_cnt)
AMH_GLOBAL_X(k_pred_slot_base)
AMH_GLOBAL_X(k_pred_fill)
AMH_GLOBAL(k_pred_sort)
AMH_GLOBAL(k_actor_mark)
AMH_GLOBAL(k_actor_rank)
AMH_GLOBAL_X(k_change_view)
AMH_GLOBAL(k_encode_size)
AMH_GLOBAL(k_chg_layout)
AMH_GLOBAL(k_encode_store)
AMH_GLOBAL(k_chg_header)
AMH_GLOBAL(k_chg_hash)
AMH_GLOBAL(k_chg_hash_seq)
AMH_GLOBAL(k_heads)
AMH_GLOBAL(k_sync_size)
AMH_GLOBAL(k_sync_fill)
AMH_GLOBAL(k_save_check)
AMH_GLOBAL(k_save_size)
AMH_GLOBAL(k_save_layout)
AMH_GLOBAL(k_save_store)
AMH_GLOBAL(k_save_finish)
AMH_GLOBAL(k_visible)
AMH_GLOBAL(k_text_len) AMH_GLOBAL(k_text_fill) AMH_GLOBAL(k_text_doc)
AMH_GLOBAL(k_m_init) AMH_GLOBAL(k_rows_m) AMH_GLOBAL(k_verify_hash)
AMH_GLOBAL(k_ht_size)
AMH_GLOBAL(k_dt_size)
AMH_GLOBAL_X(k_succ_link)

// ------------------------------------------------------------------------------------------
// Op-column decode.  Every numeric and boolean column (and the count pass) goes through the segment-parallel decoder,
// one wave per column stream (amh_segdec.inl).  The two string columns (map keys, mark names) are decoded here in two
// phases per wave of 64 documents: their streams cannot be tokenised by continuation bits (string bytes follow the
// lengths), but they are run-dominated (a map's rows are sorted by key), so the serial part is short:
//   walk    lane = document: each lane reads the next elements of its column through a 64-byte LDS window (refilled
//           cooperatively, 4-byte coalesced loads) and notes one ITEM per run / literal value / null run — (first row,
//           offset, length) — in LDS, DS_K items per lane and pass.  Strings are validated once per item, not per row.
//   expand  the wave takes the 64 documents in turn and writes the rows of the noted items, 64 rows per store.
// ------------------------------------------------------------------------------------------
#define DEC_W 64u          // window bytes per lane
#define DEC_NEED 24u       // a walk step reads at most two LEB128 numbers past pos
#define DS_K 8u            // items noted per lane between expands

struct DsShared {
  u32 win[64][DEC_W / 4 + 1];
  u64 src[64];               // arena byte address of each lane's window
  u32 it_rs[64][DS_K + 1];   // first row (document relative) of each noted item; [n] = the row behind the last one
  u32 it_off[64][DS_K], it_len[64][DS_K];
  u32 it_n[64], rb[64];
};

__device__ __forceinline__ u32 wave_max_u32(u32 v) {
  for (int off = 32; off > 0; off >>= 1) { u32 o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}

#include "amh_segdec_prims.inl"
template <u32 COL>
__device__ __forceinline__ void decode_str_body(const BatchDev& B, DsShared& S) {
  const u32 lane = threadIdx.x;
  const u32 d = blockIdx.x * 64 + lane;
  const bool live = d < B.n_docs && B.status[d] == 0;
  const u32 dd = d < B.n_docs ? d : 0;
  const DocDesc& D = B.desc[dd];
  const u32 want = live ? B.cnt[dd].n_rows : 0;
  const u32 off = D.ocol[COL].off, len = D.ocol[COL].len;
  // a column no document of this wave has: the host pre-filled the arrays with the "null" pattern (amh_load_staged)
  if (!__ballot(want != 0 && len != 0)) return;
  S.rb[lane] = B.row_base[dd];
  u32* o_off = COL == OC_KEY_STR ? B.r_key_off : B.r_mark_off;
  u32* o_len = COL == OC_KEY_STR ? B.r_key_len : B.r_mark_len;
  const u8* gp = B.arena + D.chunk_off;
  const u8* wp = (const u8*)S.win[lane];
  u32 wbase = off & ~3u;
  const u8* p = wp - wbase;
  u64 pos = off;
  const u64 end = (u64)off + len;
  u64 lit = 0;
  u32 row = 0;
  bool failed = false, first = true;
  while (__ballot(row < want && !failed)) {
    u32 n_it = 0;
    for (u32 step = 0; step < DS_K; step++) {
      const bool active = row < want && !failed;
      // ---- window refill ------------------------------------------------------------------
      bool need = first || (active && pos < end && pos + DEC_NEED > (u64)wbase + DEC_W && (u64)wbase + DEC_W < end);
      if (__ballot(need)) {
        first = false;
        u32 nb = (pos > end ? (u32)end : (u32)pos) & ~3u;
        wbase = nb;
        S.src[lane] = D.chunk_off + nb;
        sd_lds_sync();
#pragma unroll 4
        for (u32 l0 = 0; l0 < 64; l0 += 64 / (DEC_W / 4)) {
          u32 l = l0 + lane / (DEC_W / 4), k = lane % (DEC_W / 4);
          S.win[l][k] = *(const u32*)(B.arena + S.src[l] + 4 * k);   // the arena is padded: reading past a column is harmless
        }
        sd_lds_sync();
        p = wp - wbase;
      }
      // ---- one element per lane ----------------------------------------------------------------
      if (active) {
        const u32 rem = want - row;
        u32 cnt = 0, v_off = 0, v_len = AMH_NULL32;
        bool want_str = false;
        if (lit > 0) { lit--; cnt = 1; want_str = true; }
        else if (pos >= end) cnt = rem;                       // exhausted: the iterator yields None from here on
        else {
          i64 n;
          if (!rd_sleb_lenient(p, pos, end, n)) failed = true;
          else if (n > 0) { cnt = (u64)n < rem ? (u32)n : rem; want_str = true; }
          else if (n < 0) { lit = (n == INT64_MIN ? (u64)INT64_MAX : (u64)(-n)) - 1; cnt = 1; want_str = true; }
          else {
            u64 nulls;
            if (!rd_uleb_lenient(p, pos, end, nulls) || nulls > (u64)INT64_MAX) failed = true;
            else cnt = nulls < rem ? (u32)nulls : rem;
          }
        }
        if (want_str && !failed) {   // rle_read_value<2>: length, bytes (validated where they lie, in the arena)
          u64 sl;
          if (!rd_uleb_lenient(p, pos, end, sl) || sl > end - pos || !amh_valid_utf8(gp + pos, sl)) failed = true;
          else { v_off = (u32)pos; v_len = (u32)sl; pos += sl; }
        }
        if (failed) amh_fail(B, dd, AMH_E_READ_OP);
        else if (cnt) {
          S.it_rs[lane][n_it] = row; S.it_off[lane][n_it] = v_off; S.it_len[lane][n_it] = v_len;
          n_it++;
          row += cnt;
        }
      }
    }
    S.it_n[lane] = n_it;
    S.it_rs[lane][n_it] = row;
    sd_lds_sync();
    // ---- expand: the documents in turn, 64 rows per store ---------------------------------------
    for (u32 l = 0; l < 64; l++) {
      const u32 n = S.it_n[l];
      if (n == 0) continue;
      const u32 r_begin = S.it_rs[l][0], r_end = S.it_rs[l][n], rb = S.rb[l];
      for (u32 r0 = r_begin; r0 < r_end; r0 += 64) {
        // whole blocks of 64 rows inside one null item (a text document's key column is one null run of millions of
        // rows): the arrays already hold the null pattern (the host's fill), nothing to store
        u32 k0 = 0;
        for (u32 j = 1; j < n; j++) k0 = S.it_rs[l][j] <= r0 ? j : k0;
        if (S.it_len[l][k0] == AMH_NULL32 && S.it_rs[l][k0 + 1] >= r0 + 64) {
          r0 += 64 * ((S.it_rs[l][k0 + 1] - r0) / 64 - 1);
          continue;
        }
        const u32 r = r0 + lane;
        u32 k = 0;
        for (u32 j = 1; j < n; j++) k = S.it_rs[l][j] <= r ? j : k;
        if (r < r_end) { o_off[rb + r] = S.it_off[l][k]; o_len[rb + r] = S.it_len[l][k]; }
      }
    }
    sd_lds_sync();
  }
}
#include "amh_segdec.inl"

// ------------------------------------------------------------------------------------------
// Change re-encode, LDS-staged (replaces the thread-per-(change, column) kernels k_encode_size /
// k_encode_store on the device; those remain the host-emulation form of the same encoders).
//
// One wave = 64 consecutive changes, lane = change.  The per-op fields a group of columns needs are
// gathered cooperatively — consecutive threads take consecutive ops of ops_sorted, so the index reads
// coalesce and the row gathers are independent loads instead of a dependent chain per lane — into
// LDS tiles of ENC_K ops per change; each lane then runs the reference's run-length state machines
// (rle.rs:31-133, delta.rs:10-43, boolean.rs:11-51) over its own change out of LDS.  Tiles are
// padded to an odd stride so that the 64 lanes walk 64 different banks.
// STORE = false: column sizes -> c_col_len;  STORE = true: bytes -> out_arena.
// ------------------------------------------------------------------------------------------
#ifndef ENC_K
#define ENC_K 8
#endif
#define ENC_STRIDE (ENC_K + 1)
#define ENC_TILE (64 * ENC_STRIDE)

template <int NT> struct EncShared {
  u32 t[NT][ENC_TILE];
  u32 n[64], base[64], mapb[64], pbase[64], pn[64];
};

template <class F>
__device__ __forceinline__ void enc_load_tile(const u32* cnt, u32 t0, F f) {
#pragma unroll 8
  for (u32 l0 = 0; l0 < 64; l0 += 64 / ENC_K) {
    u32 l = l0 + threadIdx.x / ENC_K, k = threadIdx.x % ENC_K;
    if (t0 + k < cnt[l]) f(l, k, t0 + k);
  }
}
template <class W>
__device__ __forceinline__ void bool_feed(W& w, bool& last, u64& cnt, bool v) {  // boolean.rs:35-50
  if (v == last) cnt++; else { cw_uleb(w, cnt); last = v; cnt = 1; }
}

template <bool STORE> struct EncW { typedef SizeWriter T; };
template <> struct EncW<true> { typedef FastWriter T; };
__device__ __forceinline__ void w_init(SizeWriter& w, u8*) { sw_init(w); }
__device__ __forceinline__ void w_init(FastWriter& w, u8* o) { fw_init(w, o); }
__device__ __forceinline__ void w_done(SizeWriter&) {}
__device__ __forceinline__ void w_done(FastWriter& w) { fw_finish(w); }

// One kernel per column group G (1 object id, 2 key, 3 insert/action/predNum/expand, 4 value, 5 pred lists, 6 mark
// names): a group's state machines and tiles alone need about half the registers and LDS of all six together, so
// twice as many waves are resident to hide the latency of these serial per-lane loops.
template <int G> struct EncTiles { enum { N = (G == 2 || G == 4) ? 4 : 2 }; };
template <bool STORE, int G>
__global__ void __launch_bounds__(64) g_encode_grp(BatchDev B) {
  typedef typename EncW<STORE>::T W;
  __shared__ EncShared<EncTiles<G>::N> S;
  const u32 lane = threadIdx.x;
  const u32 c = blockIdx.x * 64 + lane;
  // (a change whose columns are cut into ranges, g_en_big, counts as not live here: lengths 0, set by g_sv_big_total)
  const bool live = c < B.tot_chg && B.status[B.c_doc[c]] == 0 && !(B.sv_big && B.sv_big[c]);
  const u32 d = live ? B.c_doc[c] : 0;
  if (G == 6) {   // only when a document of this wave has a mark column
    bool has_marks = live && B.desc[d].ocol[OC_MARK_NAME].len != 0;
    if (!__ballot(has_marks)) {
      if (!STORE && c < B.tot_chg) B.c_col_len[(u64)c * AMH_NCOLS + 13] = 0;
      return;
    }
  }
  const u32 n = live ? B.c_nops[c] : 0;
  const u32 base = live ? B.c_op_base[c] : 0;
  const u8* chunk = B.arena + B.desc[d].chunk_off;
  u32* cl = B.c_col_len + (u64)(c < B.tot_chg ? c : 0) * AMH_NCOLS;
  S.n[lane] = n;
  S.base[lane] = base;
  u32 pn = 0;
  if (G == 5) {
    S.mapb[lane] = live ? B.c_map_base[c] : 0;
    u32 pb = 0;
    if (live && n) { pb = B.j_pred_base[base]; pn = B.j_pred_base[base + n] - pb; }
    S.pbase[lane] = pb;
    S.pn[lane] = pn;
  }
  __syncthreads();
  const u32 n_max = wave_max_u32(n);
  // output cursor of column k in the store pass: columns back to back at c_cols_off (storage/change.rs:486-488);
  // long literal runs are widened in place by the register writer (cw_lit_close)
  auto col_out = [&](int k) -> u8* {
    if (!STORE || !live || cl[k] == 0) return (u8*)0;
    u64 off = (u64)B.c_out_off[c] + B.c_cols_off[c];
    for (int j = 0; j < k; j++) off += cl[j];
    return B.out_arena + off;
  };
  auto put_len = [&](int k, u32 v) { if (!STORE && c < B.tot_chg) cl[k] = live ? v : 0; };
  const u32 row = lane * ENC_STRIDE;
  RleVal x; x.v = 0; x.s = 0; x.n = 0;

  if constexpr (G == 1) {   // object id (columns 0, 1; obj_id.rs:158-172)
    RleEnc<0> ea, ec; re_init(ea); re_init(ec);
    W wa, wc; w_init(wa, col_out(0)); w_init(wc, col_out(1));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 j = S.base[l] + i, o = l * ENC_STRIDE + k;
        uint2 g = ((const uint2*)B.cv[0])[j];
        S.t[0][o] = g.x; S.t[1][o] = g.y;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 oc = S.t[1][row + i];
        const int m = oc == 0 ? RE_NULL : RE_VALUE;
        x.v = S.t[0][row + i]; re_step(wa, ea, x, m);
        x.v = oc; re_step(wc, ec, x, m);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec);
    w_done(wa); w_done(wc);
    put_len(0, wa.pos); put_len(1, wc.pos);
  }
  if constexpr (G == 2) {   // key (columns 2, 3, 4; key.rs:236-257)
    RleEnc<0> ea; re_init(ea);
    DeltaEnc ec; re_init(ec.r); ec.abs = 0;
    RleEnc<2> es; re_init(es);
    W wa, wc, ws; w_init(wa, col_out(2)); w_init(wc, col_out(3)); w_init(ws, col_out(4));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 j = S.base[l] + i, o = l * ENC_STRIDE + k;
        uint4 g = ((const uint4*)B.cv[2])[j];
        S.t[0][o] = g.x; S.t[1][o] = g.y; S.t[2][o] = g.z; S.t[3][o] = g.w;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 ka = S.t[0][row + i], kc = S.t[1][row + i], kl = S.t[3][row + i];
        const bool inl = kl != AMH_NULL32 && (kl & AMH_CV_INLINE);   // short map key: its bytes are in fields 0, 1
        RleVal sv;
        sv.v = inl ? (u64)ka | ((u64)kc << 32) : 0;
        sv.s = inl ? (const u8*)0 : chunk + S.t[2][row + i];
        sv.n = inl ? (kl & 0xFF) : kl;
        if (inl) { ka = AMH_NULL32; kc = AMH_NULL32; }
        x.v = ka; re_step(wa, ea, x, ka == AMH_NULL32 ? RE_NULL : RE_VALUE);
        de_step(wc, ec, (i64)kc, kc == AMH_NULL32);
        re_step(ws, es, sv, kl == AMH_NULL32 ? RE_NULL : RE_VALUE);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec.r); re_finish(ws, es);
    w_done(wa); w_done(wc); w_done(ws);
    put_len(2, wa.pos); put_len(3, wc.pos); put_len(4, ws.pos);
  }
  if constexpr (G == 3) {   // insert, action, predNum, expand (columns 5, 6, 9, 12)
    bool ins_last = false, exp_last = false, exp_any = false;
    u64 ins_cnt = 0, exp_cnt = 0;
    RleEnc<0> eact, epn; re_init(eact); re_init(epn);
    W wi, wact, wpn, wex; w_init(wi, col_out(5)); w_init(wact, col_out(6)); w_init(wpn, col_out(9)); w_init(wex, col_out(12));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 j = S.base[l] + i, o = l * ENC_STRIDE + k;
        uint2 g = ((const uint2*)B.cv[6])[j];
        S.t[0][o] = g.x; S.t[1][o] = g.y;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 f = S.t[0][row + i];
        bool_feed(wi, ins_last, ins_cnt, (f & 1) != 0);
        x.v = f >> 8; re_value(wact, eact, x);
        x.v = S.t[1][row + i]; re_value(wpn, epn, x);
        bool ex = (f & 2) != 0;
        exp_any = exp_any || ex;
        bool_feed(wex, exp_last, exp_cnt, ex);
      }
      __syncthreads();
    }
    if (ins_cnt > 0) cw_uleb(wi, ins_cnt);
    re_finish(wact, eact); re_finish(wpn, epn);
    if (exp_any && exp_cnt > 0) cw_uleb(wex, exp_cnt);   // MaybeBoolean: nothing unless a true was appended (boolean.rs:131-137)
    w_done(wi); w_done(wact); w_done(wpn); w_done(wex);
    put_len(5, wi.pos); put_len(6, wact.pos); put_len(9, wpn.pos); put_len(12, exp_any ? wex.pos : 0);
  }
  if constexpr (G == 4) {   // value (columns 7, 8; value.rs:320-369, op_as_changeop.rs:94-103)
    RleEnc<0> em; re_init(em);
    W wm, wr; w_init(wm, col_out(7)); w_init(wr, col_out(8));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 j = S.base[l] + i, o = l * ENC_STRIDE + k;
        uint4 g = ((const uint4*)B.cv[8])[j];
        S.t[0][o] = g.x; S.t[1][o] = g.y; S.t[2][o] = g.z; S.t[3][o] = g.w;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 hi = S.t[1][row + i], kind = (hi >> 8) & 0xFF;
        x.v = (u64)S.t[0][row + i] | ((u64)(hi & 0xFF) << 32);
        re_value(wm, em, x);
        if (kind == 3) cw_inline(wr, (u64)S.t[2][row + i] | ((u64)S.t[3][row + i] << 32), (hi >> 16) & 0xF);
        else if (kind == 1) cw_bytes(wr, chunk + S.t[2][row + i], S.t[3][row + i]);
        else if (kind == 2) cw_sleb(wr, (i64)((u64)S.t[2][row + i] | ((u64)S.t[3][row + i] << 32)));
      }
      __syncthreads();
    }
    re_finish(wm, em);
    w_done(wm); w_done(wr);
    put_len(7, wm.pos); put_len(8, wr.pos);
  }
  if constexpr (G == 5) {   // flattened pred lists (columns 10, 11; opid_list.rs:52-78)
    RleEnc<0> ea; re_init(ea);
    DeltaEnc ec; re_init(ec.r); ec.abs = 0;
    W wa, wc; w_init(wa, col_out(10)); w_init(wc, col_out(11));
    const u32 p_max = wave_max_u32(pn);
    for (u32 t0 = 0; t0 < p_max; t0 += ENC_K) {
      enc_load_tile(S.pn, t0, [&](u32 l, u32 k, u32 i) {
        u64 id = B.pred_ids[S.pbase[l] + i];
        S.t[0][l * ENC_STRIDE + k] = B.chg_actor_map[S.mapb[l] + (u32)id];
        S.t[1][l * ENC_STRIDE + k] = (u32)(id >> 32);
      });
      __syncthreads();
      u32 cnt = pn > t0 ? (pn - t0 < ENC_K ? pn - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        x.v = S.t[0][row + i]; re_value(wa, ea, x);
        de_value(wc, ec, (i64)S.t[1][row + i]);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec.r);
    w_done(wa); w_done(wc);
    put_len(10, wa.pos); put_len(11, wc.pos);
  }
  if constexpr (G == 6) {   // mark names (column 13)
    RleEnc<2> es; re_init(es);
    W ws; w_init(ws, col_out(13));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 t = B.ops_sorted[S.base[l] + i];
        bool m = !op_is_del(B, t) && B.r_action[t] == 7 && B.r_mark_len[t] != AMH_NULL32;
        S.t[0][l * ENC_STRIDE + k] = m ? B.r_mark_off[t] : 0;
        S.t[1][l * ENC_STRIDE + k] = m ? B.r_mark_len[t] : AMH_NULL32;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 ml = S.t[1][row + i];
        RleVal sv; sv.v = 0; sv.s = chunk + S.t[0][row + i]; sv.n = ml;
        re_step(ws, es, sv, ml == AMH_NULL32 ? RE_NULL : RE_VALUE);
      }
      __syncthreads();
    }
    re_finish(ws, es);
    w_done(ws);
    put_len(13, ws.pos);
  }
  (void)chunk; (void)pn;
}
#include "amh_segenc.inl"
#include "amh_segenc1.inl"
// Two forms of the change re-encode.  Batches of small changes (the headline: 50 ops per change): lane = change, 64
// changes per wave in step (g_encode_grp) — a wave per change would spend itself on per-column set-up, and the
// scalar unit of a CU, shared by its four SIMDs, would be the limit.  Larger changes: one wave per (change, column
// group), 64 values per step (g_encode_one, amh_segenc1.inl), whose cost does not depend on how the ops are spread over the changes.
static bool encode_lane_form(const BatchDev& B) {
  // AMH_ENCODE_FORM=seg | lane forces one form (tests run every document shape through both)
  const char* form = getenv("AMH_ENCODE_FORM");
  return form ? form[0] == 'l' : (u64)B.tot_ops < (u64)B.tot_chg * 192;
}
bool amh_encode_one_pass(const BatchDev& B) { return !encode_lane_form(B); }
u64 amh_encode_tmp_bytes(const BatchDev& B) {
  return (u64)B.tot_ops * SC1_OP_SUM + (u64)B.tot_succ * SC1_PRED_SUM + (u64)B.tot_chg * (AMH_NCOLS * SC1_PAD) + 256;
}
template <bool STORE>
static void launch_encode(const BatchDev& B, hipStream_t s) {
  if (encode_lane_form(B)) {
    const dim3 g((B.tot_chg + 63) / 64), b(64);
    hipLaunchKernelGGL((g_encode_grp<STORE, 1>), g, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_grp<STORE, 2>), g, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_grp<STORE, 3>), g, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_grp<STORE, 4>), g, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_grp<STORE, 5>), g, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_grp<STORE, 6>), g, b, 0, s, B);
    return;
  }
  // one pass (amh_segenc1.inl): "size" = the walk that writes every column into its slot and leaves the lengths;
  // "store" (behind the layout) = the copy into the chunks + a second walk for the columns that outgrew their slots
  const u32 parts = (u64)B.tot_ops >= (u64)B.tot_chg * 1024 ? SC_GROUPS : 1;
  const dim3 g(((((B.tot_chg + SD_WAVES - 1) / SD_WAVES) * parts) + 7u) & ~7u), g1((B.tot_chg + SD_WAVES - 1) / SD_WAVES), b(64 * SD_WAVES);
  const char* gm = getenv("AMH_ENC_GMASK");
  const u32 gmask = gm ? (u32)atoi(gm) : 0x1Fu;
  if (!STORE) {
    hipLaunchKernelGGL((g_encode_one<0>), g, b, 0, s, B, gmask, parts);
  } else {
    hipLaunchKernelGGL(g_encode_place, g1, b, 0, s, B);
    hipLaunchKernelGGL((g_encode_one<1>), g, b, 0, s, B, gmask, parts);
  }
}


// ------------------------------------------------------------------------------------------
// Document save, LDS-staged (device form of k_save_size / k_save_store; those remain the host-emulation form and
// the writer of "wide literal" columns).  Same scheme as g_encode_grp with lane = document: the rows of 64 documents
// are gathered ENC_K at a time into LDS tiles with coalesced loads (the actor rank lookups and the value conversion
// are done by the gathering threads, one row each), then every lane runs the reference's column encoders over its
// own document out of LDS and writes through the 16-byte register writer.  Columns 0..8 (change metadata: a handful
// of rows per document) are encoded by the lane straight from the change arrays.
// ------------------------------------------------------------------------------------------
struct SaveShared {
  u32 t0[ENC_TILE], t1[ENC_TILE], t2[ENC_TILE], t3[ENC_TILE];
  u32 n[64], base[64], mapb[64], pbase[64], pn[64];
};
template <bool STORE>
__global__ void __launch_bounds__(64) g_save_lds(BatchDev B) {
  typedef typename EncW<STORE>::T W;
  __shared__ SaveShared S;
  const u32 lane = threadIdx.x;
  const u32 d = blockIdx.x * 64 + lane;
  const bool live = d < B.n_docs && B.status[d] == 0 && B.sv_status[d] == 0;
  const u32 dd = d < B.n_docs ? d : 0;
  const bool big = B.sv_big && d < B.n_docs && (B.sv_big[dd] & 1);    // op columns: g_sv_big (their lengths are 0 here, set by g_sv_big_total)
  const bool bigc = B.sv_big && d < B.n_docs && (B.sv_big[dd] & 2);   // change-metadata columns: g_sv_big
  const u32 n = (live && !big) ? B.cnt[dd].n_rows : 0, ns = (live && !big) ? B.cnt[dd].n_succ : 0;
  const u8* chunk = B.arena + B.desc[dd].chunk_off;
  u32* cl = B.sv_col_len + (u64)dd * AMH_SV_NCOLS;
  S.n[lane] = n;
  S.base[lane] = B.row_base[dd];
  S.mapb[lane] = B.actor_base[dd];
  S.pbase[lane] = B.succ_base[dd];
  S.pn[lane] = ns;
  __syncthreads();
  const u32 n_max = wave_max_u32(n);
  const u32 wide_in = 0;   // the register writer widens long literal headers itself (cw_lit_close)
  u32 wide = 0;
  const u64 data0 = (STORE && live) ? (u64)B.sv_off[dd] + sv_data_start(B, dd) : 0;
  auto out_ptr = [&](u32 k) -> u8* {   // where column k goes (store pass), null = not written here
    if (!STORE || !live || cl[k] == 0 || ((wide_in >> k) & 1) || (big && k >= AMH_SV_CHG_COLS) || (bigc && k < AMH_SV_CHG_COLS)) return (u8*)0;
    u64 off = data0;
    for (u32 q = 0; q < k; q++) off += cl[q];
    return B.sv_out + off;
  };
  auto done = [&](W& w, u32 k) { w_done(w); if (!STORE && d < B.n_docs) cl[k] = live ? w.pos : 0; };
  const u32 row = lane * ENC_STRIDE;
  RleVal x; x.v = 0; x.s = 0; x.n = 0;

  // blockIdx.y = column group: the groups of a document are independent (each finds its columns' places from the
  // column lengths), so a batch of few, large documents still fills the device
  const u32 grp = blockIdx.y;
  // ---- change metadata, columns 0..8 ------------------------------------------------------------
  if (grp == 0) for (u32 k = 0; k < AMH_SV_CHG_COLS; k++) {
    W w; w_init(w, out_ptr(k));
    if (live && !bigc) save_encode_chg_col(B, dd, k, w);
    done(w, k);
  }
  // ---- object id (columns 9, 10) ---------------------------------------------------------------------
  if (grp == 1) {
    RleEnc<0> ea, ec; re_init(ea); re_init(ec);
    W wa, wc; w_init(wa, out_ptr(9)); w_init(wc, out_ptr(10));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
        u32 oc = B.r_obj_ctr[r];
        S.t1[o] = oc;
        S.t0[o] = oc ? B.actor_rank[S.mapb[l] + B.r_obj_actor[r]] : 0;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 oc = S.t1[row + i];
        const int m = oc == 0 ? RE_NULL : RE_VALUE;
        x.v = S.t0[row + i]; re_step(wa, ea, x, m);
        x.v = oc; re_step(wc, ec, x, m);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec);
    done(wa, 9); done(wc, 10);
  }
  // ---- key (columns 11, 12, 13; key.rs:236-257) --------------------------------------------------------
  if (grp == 2) {
    RleEnc<0> ea; re_init(ea);
    DeltaEnc ec; re_init(ec.r); ec.abs = 0;
    RleEnc<2> es; re_init(es);
    W wa, wc, ws; w_init(wa, out_ptr(11)); w_init(wc, out_ptr(12)); w_init(ws, out_ptr(13));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
        u32 kl = B.r_key_len[r], kc = B.r_key_ctr[r];
        bool prop = kl != AMH_NULL32;
        S.t3[o] = kl;
        S.t2[o] = prop ? B.r_key_off[r] : 0;
        S.t1[o] = prop ? AMH_NULL32 : kc;
        S.t0[o] = (!prop && kc != 0) ? B.actor_rank[S.mapb[l] + B.r_key_actor[r]] : AMH_NULL32;
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 ka = S.t0[row + i], kc = S.t1[row + i], kl = S.t3[row + i];
        x.v = ka; re_step(wa, ea, x, ka == AMH_NULL32 ? RE_NULL : RE_VALUE);
        de_step(wc, ec, (i64)kc, kc == AMH_NULL32);
        RleVal sv; sv.v = 0; sv.s = chunk + S.t2[row + i]; sv.n = kl;
        re_step(ws, es, sv, kl == AMH_NULL32 ? RE_NULL : RE_VALUE);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec.r); re_finish(ws, es);
    done(wa, 11); done(wc, 12); done(ws, 13);
  }
  // ---- op id (columns 14, 15) --------------------------------------------------------------------------
  if (grp == 3) {
    RleEnc<0> ea; re_init(ea);
    DeltaEnc ec; re_init(ec.r); ec.abs = 0;
    W wa, wc; w_init(wa, out_ptr(14)); w_init(wc, out_ptr(15));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
        S.t0[o] = B.actor_rank[S.mapb[l] + B.r_id_actor[r]];
        S.t1[o] = B.r_id_ctr[r];
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        x.v = S.t0[row + i]; re_value(wa, ea, x);
        de_value(wc, ec, (i64)S.t1[row + i]);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec.r);
    done(wa, 14); done(wc, 15);
  }
  // ---- insert, action, succ count, expand (columns 16, 17, 20, 23) -------------------------------------
  if (grp == 4) {
    bool ins_last = false, exp_last = false, exp_any = false;
    u64 ins_cnt = 0, exp_cnt = 0;
    RleEnc<0> eact, esn; re_init(eact); re_init(esn);
    W wi, wact, wsn, wex; w_init(wi, out_ptr(16)); w_init(wact, out_ptr(17)); w_init(wsn, out_ptr(20)); w_init(wex, out_ptr(23));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
        u32 action = B.r_action[r];
        S.t0[o] = (B.r_insert[r] ? 1u : 0u) | ((action == 7 && B.r_expand[r]) ? 2u : 0u) | (action << 8);
        S.t1[o] = B.r_succ_cnt[r];
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 f = S.t0[row + i];
        bool_feed(wi, ins_last, ins_cnt, (f & 1) != 0);
        x.v = f >> 8; re_value(wact, eact, x);
        x.v = S.t1[row + i]; re_value(wsn, esn, x);
        bool ex = (f & 2) != 0;
        exp_any = exp_any || ex;
        bool_feed(wex, exp_last, exp_cnt, ex);
      }
      __syncthreads();
    }
    if (ins_cnt > 0) cw_uleb(wi, ins_cnt);
    re_finish(wact, eact); re_finish(wsn, esn);
    if (exp_any && exp_cnt > 0) cw_uleb(wex, exp_cnt);   // MaybeBoolean (boolean.rs:131-137)
    if (!exp_any) wex.pos = 0;
    done(wi, 16); done(wact, 17); done(wsn, 20); done(wex, 23);
  }
  // ---- value (columns 18, 19; op_as_docop.rs, value.rs:320-369) ------------------------------------------
  if (grp == 5) {
    RleEnc<0> em; re_init(em);
    W wm, wr; w_init(wm, out_ptr(18)); w_init(wr, out_ptr(19));
    for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
      enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
        u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
        OpVal v = op_val(B, r);
        S.t0[o] = (u32)v.meta;
        S.t1[o] = (u32)(v.meta >> 32) | ((u32)v.kind << 8);
        if (v.kind == 2) { S.t2[o] = (u32)(u64)v.i; S.t3[o] = (u32)((u64)v.i >> 32); }
        else {
          S.t2[o] = (v.kind == 1 && v.raw_len) ? B.desc[B.r_doc[r]].ocol[OC_VAL_RAW].off + B.r_vraw_off[r] : 0;
          S.t3[o] = v.raw_len;
        }
      });
      __syncthreads();
      u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        u32 hi = S.t1[row + i], kind = hi >> 8;
        x.v = (u64)S.t0[row + i] | ((u64)(hi & 0xFF) << 32);
        re_value(wm, em, x);
        if (kind == 1) cw_bytes(wr, chunk + S.t2[row + i], S.t3[row + i]);
        else if (kind == 2) cw_sleb(wr, (i64)((u64)S.t2[row + i] | ((u64)S.t3[row + i] << 32)));
      }
      __syncthreads();
    }
    re_finish(wm, em);
    done(wm, 18); done(wr, 19);
  }
  // ---- flattened succ lists (columns 21, 22; opid_list.rs:52-78) -----------------------------------------
  if (grp == 6) {
    RleEnc<0> ea; re_init(ea);
    DeltaEnc ec; re_init(ec.r); ec.abs = 0;
    W wa, wc; w_init(wa, out_ptr(21)); w_init(wc, out_ptr(22));
    const u32 s_max = wave_max_u32(ns);
    for (u32 t0 = 0; t0 < s_max; t0 += ENC_K) {
      enc_load_tile(S.pn, t0, [&](u32 l, u32 k, u32 i) {
        u32 e = S.pbase[l] + i, o = l * ENC_STRIDE + k;
        S.t0[o] = B.actor_rank[S.mapb[l] + B.s_actor[e]];
        S.t1[o] = B.s_ctr[e];
      });
      __syncthreads();
      u32 cnt = ns > t0 ? (ns - t0 < ENC_K ? ns - t0 : ENC_K) : 0;
      for (u32 i = 0; i < cnt; i++) {
        x.v = S.t0[row + i]; re_value(wa, ea, x);
        de_value(wc, ec, (i64)S.t1[row + i]);
      }
      __syncthreads();
    }
    re_finish(wa, ea); re_finish(wc, ec.r);
    done(wa, 21); done(wc, 22);
  }
  // ---- mark names (column 24) — only when a document of this wave has marks ----------------------------------
  if (grp == 7) {
    bool has_marks = live && B.desc[dd].ocol[OC_MARK_NAME].len != 0;
    if (__ballot(has_marks)) {
      RleEnc<2> es; re_init(es);
      W ws; w_init(ws, out_ptr(24));
      for (u32 t0 = 0; t0 < n_max; t0 += ENC_K) {
        enc_load_tile(S.n, t0, [&](u32 l, u32 k, u32 i) {
          u32 r = S.base[l] + i, o = l * ENC_STRIDE + k;
          bool m = B.r_action[r] == 7 && B.r_mark_len[r] != AMH_NULL32;
          S.t0[o] = m ? B.r_mark_off[r] : 0;
          S.t1[o] = m ? B.r_mark_len[r] : AMH_NULL32;
        });
        __syncthreads();
        u32 cnt = n > t0 ? (n - t0 < ENC_K ? n - t0 : ENC_K) : 0;
        for (u32 i = 0; i < cnt; i++) {
          u32 ml = S.t1[row + i];
          RleVal sv; sv.v = 0; sv.s = chunk + S.t0[row + i]; sv.n = ml;
          re_step(ws, es, sv, ml == AMH_NULL32 ? RE_NULL : RE_VALUE);
        }
        __syncthreads();
      }
      re_finish(ws, es);
      done(ws, 24);
    } else if (!STORE && d < B.n_docs) cl[24] = 0;
  }
  if (!STORE && d < B.n_docs && grp == 0) B.sv_wide[d] = 0;   // nothing is left to the byte-wise writer
  (void)wide;
}

// ------------------------------------------------------------------------------------------
// Lists of more than AMH_BIG_LIST entries (the children of one new element, the top rows of one gap, the succ list
// of one op, the preds of one op).  The thread-per-item kernels sort what fits one thread and record the rest; here a
// block takes one recorded list at a time and sorts it with a bitonic network over global memory.  All
// compare-exchanges put the smaller entry at the lower index, so positions at and beyond n stand for +infinity and
// are never touched (any n, not only powers of two).  O(n log^2 n / 256) steps per list instead of O(n^2) on one lane.
// ------------------------------------------------------------------------------------------
#define BIG_GRID 64u
template <class Less, class Swap> __device__ void blk_bitonic(u32 n, Less less, Swap swp) {
  u32 np = 1;
  while (np < n) np <<= 1;
  for (u32 k = 2; k <= np; k <<= 1) {
    for (u32 t = threadIdx.x; t < np / 2; t += AMH_BLOCK) {   // halves of a k-block, mirrored
      u32 blk = t / (k / 2), off = t % (k / 2);
      u32 i = blk * k + off, j = blk * k + k - 1 - off;
      if (j < n && less(j, i)) swp(i, j);
    }
    __syncthreads();
    for (u32 jj = k >> 2; jj > 0; jj >>= 1) {
      for (u32 t = threadIdx.x; t < np / 2; t += AMH_BLOCK) {
        u32 i = (t / jj) * 2 * jj + (t % jj), j = i + jj;
        if (j < n && less(j, i)) swp(i, j);
      }
      __syncthreads();
    }
  }
}
__global__ void __launch_bounds__(AMH_BLOCK) g_big_pred(const BatchDev B) {
  u32 cnt = *B.big_n;
  if (cnt > B.big_cap) cnt = B.big_cap;
  for (u32 w = blockIdx.x; w < cnt; w += gridDim.x) {
    u32 t = B.big_items[2 * (size_t)w], d = B.big_items[2 * (size_t)w + 1];
    u64* v = B.pred_ids + B.t_pred_base[t];
    blk_bitonic(B.t_pred_cnt[t],
                [&](u32 i, u32 j) { return lamport_cmp(B, d, (u32)(v[i] >> 32), (u32)v[i], (u32)(v[j] >> 32), (u32)v[j]) < 0; },
                [&](u32 i, u32 j) { u64 x = v[i]; v[i] = v[j]; v[j] = x; });
  }
}
// ------------------------------------------------------------------------------------------
// Visibility and winners (types.rs:712-720, iter/top_ops.rs:58-65), device form of k_visible: the winner of a key /
// element group is its LAST visible row, i.e. a wave-level segmented reduction over the rows in document order.
// One wave per tile of 64 rows: every lane loads its own row (coalesced), the group boundaries come from comparing
// each row with the one in front (the neighbour's fields by DPP, lane 0 reads row r - 1), and two ballots — VIS
// (visible rows) and GS (rows that start a group) — decide every lane at once: a visible row wins iff a group starts
// before the next visible row behind it.  Lanes whose group runs past the tile look at the following tiles together
// (ballots again); that loop ends at the first tile with a visible row or a group start, so the work stays linear in
// the rows (one key overwritten 50k times: only its last row is visible).
// ------------------------------------------------------------------------------------------
struct VisRow { u32 doc, oc, oa, kl, ko, ea, ec; bool prop, vis; };
__device__ __forceinline__ VisRow vis_load(const BatchDev& B, u32 r) {
  VisRow v;
  v.doc = B.r_doc[r]; v.oc = B.r_obj_ctr[r]; v.oa = B.r_obj_actor[r];
  v.kl = B.r_key_len[r];
  const bool ins = B.r_insert[r] != 0;
  v.prop = v.kl != AMH_NULL32 && !ins;
  v.ko = v.prop ? B.r_key_off[r] : 0u;
  v.ea = ins ? B.r_id_actor[r] : B.r_key_actor[r];
  v.ec = ins ? B.r_id_ctr[r] : B.r_key_ctr[r];
  v.vis = v.doc < B.n_docs && B.status[v.doc] == 0 && row_visible(B, r);
  return v;
}
// rows a (in front) and b in the same key / element group (same_group of amh_kernels.inl)
__device__ __forceinline__ bool vis_same(const BatchDev& B, const VisRow& a, const VisRow& b) {
  if (a.doc != b.doc || a.oc != b.oc || a.oa != b.oa || a.prop != b.prop) return false;
  if (!a.prop) return a.ea == b.ea && a.ec == b.ec;
  if (a.kl != b.kl) return false;
  if (a.ko == b.ko) return true;   // rows of one run of the key column share the string
  const u8* chunk = B.arena + B.desc[a.doc].chunk_off;
  RleVal x, y;
  x.v = 0; x.s = chunk + a.ko; x.n = a.kl;
  y.v = 0; y.s = chunk + b.ko; y.n = b.kl;
  return rv_eq<2>(x, y);
}
__global__ void __launch_bounds__(AMH_BLOCK) g_visible_seg(BatchDev B, u32 n) {
  const u32 r = blockIdx.x * AMH_BLOCK + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  const u32 tb = r - lane;
  if (tb >= n) return;
  const bool valid = r < n;
  VisRow me;
  me.doc = AMH_NULL32; me.oc = me.oa = me.kl = me.ko = me.ea = me.ec = 0; me.prop = false; me.vis = false;
  if (valid) me = vis_load(B, r);
  // the row in front: lane - 1's by DPP, lane 0 reads it
  VisRow pv;
  pv.doc = sd_prev(me.doc); pv.oc = sd_prev(me.oc); pv.oa = sd_prev(me.oa); pv.kl = sd_prev(me.kl); pv.ko = sd_prev(me.ko);
  pv.ea = sd_prev(me.ea); pv.ec = sd_prev(me.ec); pv.prop = sd_prev((u32)me.prop) != 0; pv.vis = false;
  bool gs = true;
  if (lane == 0) { if (valid && r > 0) { pv = vis_load(B, r - 1); gs = !vis_same(B, pv, me); } }
  else if (valid) gs = !vis_same(B, pv, me);
  const u64 VALID = __ballot(valid);
  const u64 VIS = __ballot(me.vis), GS = __ballot(gs && valid) | ~VALID;   // rows behind the last one end every group
  bool win = false, open = false;
  if (me.vis) {
    const u64 behind = lane == 63 ? 0 : (~(u64)0 << (lane + 1));
    const u64 later = VIS & behind, gsl = GS & behind;
    if (later) { const u32 j = (u32)__builtin_ctzll(later); win = (gsl & sd_low_bits(j + 1)) != 0; }
    else if (gsl) win = true;
    else open = true;
  }
  // groups that run past the tile: the following tiles, the open lanes together
  for (u32 nb = tb + 64; __ballot(open) != 0; nb += 64) {
    if (nb >= n) { if (open) { win = true; open = false; } break; }
    const u32 q = nb + lane;
    const bool v2 = q < n;
    VisRow x;
    x.doc = AMH_NULL32; x.vis = false;
    bool g2 = true;
    if (v2) { x = vis_load(B, q); const VisRow px = vis_load(B, q - 1); g2 = !vis_same(B, px, x); }
    const u64 VALID2 = __ballot(v2);
    const u64 VIS2 = __ballot(x.vis), GS2 = __ballot(g2 && v2) | ~VALID2;
    if (open) {
      if (VIS2) { const u32 j = (u32)__builtin_ctzll(VIS2); win = (GS2 & sd_low_bits(j + 1)) != 0; open = false; }
      else if (GS2) { win = true; open = false; }
    }
  }
  if (valid) { B.r_visible[r] = me.vis ? 1 : 0; B.r_winner[r] = win ? 1 : 0; }
}

void amh_launch(int k, const BatchDev& B, u32 n, void* stream) {
  if (n == 0) return;
  hipStream_t s = (hipStream_t)stream;
  if (k == AK_COUNT) {
    if (B.bc_n) {   // documents with columns of AMH_BIGCOL bytes and more: cut them into ranges first (amh_segdec.inl)
      hipLaunchKernelGGL(g_bc_find, dim3((B.n_docs * 13u + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, s, B);
      hipLaunchKernelGGL(g_bc_chain, dim3((B.bc_cap_segs + BC_WAVES - 1) / BC_WAVES), dim3(64 * BC_WAVES), 0, s, B);
      hipLaunchKernelGGL(g_bc_resolve, dim3((B.bc_cap_cols + 63) / 64), dim3(64), 0, s, B);
      hipLaunchKernelGGL(g_bc_measure, dim3((B.bc_cap_segs + SD_WAVES - 1) / SD_WAVES), dim3(64 * SD_WAVES), 0, s, B);
      hipLaunchKernelGGL(g_bc_prefix, dim3((B.bc_cap_cols + 63) / 64), dim3(64), 0, s, B);
    }
    hipLaunchKernelGGL(g_count_seg, dim3((B.n_docs + SD_WAVES - 1) / SD_WAVES, 2), dim3(64 * SD_WAVES), 0, s, B);
    return;
  }
  if (k == AK_DECODE_OPS) {
    hipLaunchKernelGGL(g_decode_seg, dim3((B.n_docs + SD_WAVES - 1) / SD_WAVES, SD_NCOLS + 2), dim3(64 * SD_WAVES), 0, s, B);
    if (B.bc_n) hipLaunchKernelGGL(g_bc_decode, dim3((B.bc_cap_segs + SD_WAVES - 1) / SD_WAVES), dim3(64 * SD_WAVES), 0, s, B);
    if (B.fr_n) hipLaunchKernelGGL(g_fill_runs, dim3(2048), dim3(AMH_BLOCK), 0, s, B);
    return;
  }
  if (k == AK_SAVE_SIZE || k == AK_SAVE_STORE) {
    // documents of a few hundred rows and more: a wave per (document, column group) encodes the op columns, 64 rows per
    // step (g_save_seg); the change-metadata columns, and batches of tiny documents, stay with lane = document (g_save_lds).
    // AMH_SAVE_FORM=seg | lane forces one form (tests).
    const char* form = getenv("AMH_SAVE_FORM");
    const bool seg = form ? form[0] == 's' : (u64)B.tot_rows >= (u64)B.n_docs * 256;
    const dim3 gl((B.n_docs + 63) / 64, seg ? 1 : 8), gs((B.n_docs + SD_WAVES - 1) / SD_WAVES, SV_GROUPS);
    if (k == AK_SAVE_SIZE) {
      hipLaunchKernelGGL((g_save_lds<false>), gl, dim3(64), 0, s, B);
      if (seg) hipLaunchKernelGGL((g_save_seg<false>), gs, dim3(64 * SD_WAVES), 0, s, B);
    } else {
      hipLaunchKernelGGL((g_save_lds<true>), gl, dim3(64), 0, s, B);
      if (seg) hipLaunchKernelGGL((g_save_seg<true>), gs, dim3(64 * SD_WAVES), 0, s, B);
    }
    return;
  }
  if (k == AK_VISIBLE) { hipLaunchKernelGGL(g_visible_seg, dim3((n + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, s, B, n); return; }
  if (k == AK_ENCODE_SIZE) { launch_encode<false>(B, s); return; }
  if (k == AK_ENCODE_STORE) {
    launch_encode<true>(B, s);
    return;
  }
  dim3 grid((n + AMH_BLOCK - 1) / AMH_BLOCK), gridx(amh_xcd_grid(n)), block(AMH_BLOCK);
  if (k == AK_PRED_SORT) {   // the thread-per-op kernel, then a block per pred list it found too long for one thread
    hipLaunchKernelGGL(g_k_pred_sort, grid, block, 0, s, B, n);
    hipLaunchKernelGGL(g_big_pred, dim3(BIG_GRID), block, 0, s, B);
    return;
  }
  switch (k) {
#define LX(id, name) case id: hipLaunchKernelGGL(g_##name, gridx, block, 0, s, B, n); return;
    LX(AK_PLACE, k_place) LX(AK_PRED_FILL, k_pred_fill) LX(AK_SUCC_LINK, k_succ_link) LX(AK_CHANGE_VIEW, k_change_view) LX(AK_PRED_SLOT_BASE, k_pred_slot_base)
#undef LX
    default: break;
  }
#define L(id, name) case id: hipLaunchKernelGGL(g_##name, grid, block, 0, s, B, n); break;
  switch (k) {
    L(AK_PARSE, k_parse) L(AK_SHA_DOC, k_sha_doc) L(AK_COUNT, k_count) L(AK_EXTRACT, k_extract)
    L(AK_ACTORS, k_actors) L(AK_CHG_INIT, k_chg_init) L(AK_DECODE_CHG, k_decode_chg) L(AK_DECODE_CHG_COLS, k_decode_chg_cols) L(AK_CHG_PATCH, k_chg_patch) L(AK_ATAB_BASE, k_atab_base) L(AK_ATAB_FILL, k_atab_fill)
    L(AK_DECODE_OPS, k_decode_ops) L(AK_ROWS, k_rows) L(AK_OBJ_ORDER, k_obj_order) L(AK_SUCC, k_succ)
    L(AK_BUCKET, k_bucket) L(AK_CHG_FINISH, k_chg_finish) L(AK_PLACE, k_place) L(AK_PRED_SLOT_CNT, k_pred_slot_cnt) L(AK_PRED_SLOT_BASE, k_pred_slot_base) L(AK_PRED_FILL, k_pred_fill)
    L(AK_PRED_SORT, k_pred_sort) L(AK_ACTOR_MARK, k_actor_mark) L(AK_ACTOR_RANK, k_actor_rank) L(AK_CHANGE_VIEW, k_change_view)
    L(AK_ENCODE_SIZE, k_encode_size) L(AK_CHG_LAYOUT, k_chg_layout) L(AK_ENCODE_STORE, k_encode_store)
    L(AK_CHG_HEADER, k_chg_header) L(AK_CHG_HASH, k_chg_hash) L(AK_CHG_HASH_SEQ, k_chg_hash_seq) L(AK_HEADS, k_heads) L(AK_SYNC_SIZE, k_sync_size) L(AK_SYNC_FILL, k_sync_fill) L(AK_SAVE_CHECK, k_save_check) L(AK_SAVE_SIZE, k_save_size) L(AK_SAVE_LAYOUT, k_save_layout) L(AK_SAVE_STORE, k_save_store) L(AK_SAVE_FINISH, k_save_finish) L(AK_VISIBLE, k_visible) L(AK_HT_SIZE, k_ht_size) L(AK_DT_SIZE, k_dt_size) L(AK_SUCC_LINK, k_succ_link) L(AK_M_INIT, k_m_init) L(AK_ROWS_M, k_rows_m) L(AK_VERIFY_HASH, k_verify_hash)
    L(AK_TEXT_LEN, k_text_len) L(AK_TEXT_FILL, k_text_fill) L(AK_TEXT_DOC, k_text_doc)
    default: break;
  }
#undef L
}

// ------------------------------------------------------------------------------------------
// exclusive scan (u32, saturating): tiles of 2048 = 256 threads x 8 items.
//   pass 1: tile totals -> tmp        (g_scan_reduce)
//   pass 2: scan of the tile totals   (recursive)
//   pass 3: tile-local scan + offset  (g_scan_apply)
// ------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (AMH_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ u32 sat_add(u32 a, u32 b) { u32 s = a + b; return s < a ? 0xFFFFFFFFu : s; }

__device__ u32 block_exclusive_scan(u32 v, u32* total) {
  __shared__ u32 wave_sums[AMH_BLOCK / 64];
  u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 x = v;
  for (int off = 1; off < 64; off <<= 1) {
    u32 y = __shfl_up(x, off, 64);
    if (lane >= (u32)off) x = sat_add(x, y);
  }
  if (lane == 63) wave_sums[wave] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
  for (u32 w = 0; w < AMH_BLOCK / 64; w++) {
    if (w < wave) base = sat_add(base, wave_sums[w]);
    tot = sat_add(tot, wave_sums[w]);
  }
  __syncthreads();
  *total = tot;
  return sat_add(base, x - v);  // a saturated x makes the grand total saturate, which the host rejects
}

// a thread's 8 consecutive items: two 16-byte loads when the array is 16-byte aligned and the items are all there
// (8 separate dword loads at a lane stride of 32 bytes made every wave touch each of its cache lines 8 times: 2.2 TB/s)
__device__ __forceinline__ void scan_load8(const u32* data, u32 n, u32 start, bool vec, u32 (&v)[SCAN_ITEMS]) {
  if (vec && start + SCAN_ITEMS <= n) {
    const uint4 a = *(const uint4*)(data + start), b = *(const uint4*)(data + start + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (start + k < n) ? data[start + k] : 0;
  }
}
__global__ void __launch_bounds__(AMH_BLOCK) g_scan_reduce(const u32* data, u32 n, u32* tile_sums) {
  const u32 start = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const bool vec = ((uintptr_t)data & 15) == 0;
  u32 v[SCAN_ITEMS];
  scan_load8(data, n, start, vec, v);
  u32 sum = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) sum = sat_add(sum, v[k]);
  u32 tot;
  block_exclusive_scan(sum, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(AMH_BLOCK) g_scan_apply(u32* data, u32 n, const u32* tile_offsets, u32* total_out) {
  const u32 start = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const bool vec = ((uintptr_t)data & 15) == 0;
  u32 v[SCAN_ITEMS];
  scan_load8(data, n, start, vec, v);
  u32 sum = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) sum = sat_add(sum, v[k]);
  u32 tot;
  u32 excl = block_exclusive_scan(sum, &tot);
  u32 run = sat_add(tile_offsets ? tile_offsets[blockIdx.x] : 0, excl);
  u32 o[SCAN_ITEMS];
  for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = run; run = sat_add(run, v[k]); }
  if (vec && start + SCAN_ITEMS <= n) {
    *(uint4*)(data + start) = make_uint4(o[0], o[1], o[2], o[3]);
    *(uint4*)(data + start + 4) = make_uint4(o[4], o[5], o[6], o[7]);
  } else {
    for (int k = 0; k < SCAN_ITEMS; k++) if (start + k < n) data[start + k] = o[k];
  }
  // the last tile writes the grand total
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == AMH_BLOCK - 1) *total_out = run;
}

// (A one-pass decoupled look-back form — ticket per tile, {state, value} status words, 64 tiles per look — was built and
// measured on the 3.8e8-row scans of the merge step: 2.18 ms against 0.79 ms for the three passes below with 16-byte
// accesses; 185,000 tiles of 2048 items each pay a ticket atomic and cache-bypassing status reads.  Removed.)
u32 amh_scan_tmp_words(u32 n) {
  u32 words = 0;
  while (n > SCAN_TILE) {
    u32 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    words += tiles + 1;
    n = tiles;
  }
  return words + 2;
}

void amh_scan_inplace(u32* data, u32 n, u32* tmp, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(data, 0, sizeof(u32), s);
    return;
  }
  u32 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (tiles == 1) {
    hipLaunchKernelGGL(g_scan_apply, dim3(1), dim3(AMH_BLOCK), 0, s, data, n, (const u32*)nullptr, data + n);
    return;
  }
  hipLaunchKernelGGL(g_scan_reduce, dim3(tiles), dim3(AMH_BLOCK), 0, s, (const u32*)data, n, tmp);
  amh_scan_inplace(tmp, tiles, tmp + tiles + 1, s);  // tmp[tiles] receives the total of the tile sums
  hipLaunchKernelGGL(g_scan_apply, dim3(tiles), dim3(AMH_BLOCK), 0, s, data, n, (const u32*)tmp, data + n);
}

// ------------------------------------------------------------------------------------------
// changes grouped by dependency level: one radix split on the level (49 buckets).
//   g_level_hist     per block of 256 changes, a histogram in LDS -> hist[bucket * nblocks + block]
//   (exclusive scan of hist)
//   g_level_scatter  each change to base[bucket][block] + its rank inside the block
// ------------------------------------------------------------------------------------------
#define LVL_BUCKETS (AMH_LEVEL_CAP + 1)
__global__ void __launch_bounds__(AMH_BLOCK) g_level_hist(BatchDev B, u32* hist, u32 nblocks) {
  __shared__ u32 h[LVL_BUCKETS];
  for (u32 i = threadIdx.x; i < LVL_BUCKETS; i += AMH_BLOCK) h[i] = 0;
  __syncthreads();
  u32 c = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (c < B.tot_chg) atomicAdd(&h[chg_level_bucket(B, c)], 1u);
  __syncthreads();
  for (u32 i = threadIdx.x; i < LVL_BUCKETS; i += AMH_BLOCK) hist[(u64)i * nblocks + blockIdx.x] = h[i];
}
__global__ void __launch_bounds__(AMH_BLOCK) g_level_scatter(BatchDev B, const u32* hist, u32 nblocks) {
  __shared__ u32 h[LVL_BUCKETS];
  for (u32 i = threadIdx.x; i < LVL_BUCKETS; i += AMH_BLOCK) h[i] = hist[(u64)i * nblocks + blockIdx.x];
  __syncthreads();
  u32 c = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (c < B.tot_chg) B.lvl_order[atomicAdd(&h[chg_level_bucket(B, c)], 1u)] = c;
  if (blockIdx.x == 0)
    for (u32 i = threadIdx.x; i <= LVL_BUCKETS; i += AMH_BLOCK) B.lvl_start[i] = i < LVL_BUCKETS ? hist[(u64)i * nblocks] : B.tot_chg;
}
u32 amh_level_sort_tmp_words(u32 tot_chg) {
  u32 nblocks = (tot_chg + AMH_BLOCK - 1) / AMH_BLOCK;
  u32 n = LVL_BUCKETS * nblocks;
  return n + 2 + amh_scan_tmp_words(n);
}
void amh_level_sort(const BatchDev& B, u32* tmp, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  u32 nblocks = (B.tot_chg + AMH_BLOCK - 1) / AMH_BLOCK;
  if (nblocks == 0) { (void)hipMemsetAsync(B.lvl_start, 0, (LVL_BUCKETS + 1) * sizeof(u32), s); return; }
  u32 n = LVL_BUCKETS * nblocks;
  hipLaunchKernelGGL(g_level_hist, dim3(nblocks), dim3(AMH_BLOCK), 0, s, B, tmp, nblocks);
  amh_scan_inplace(tmp, n, tmp + n + 2, s);
  hipLaunchKernelGGL(g_level_scatter, dim3(nblocks), dim3(AMH_BLOCK), 0, s, B, (const u32*)tmp, nblocks);
}

// ------------------------------------------------------------------------------------------
// merge kernels: one thread per unit, the MergeDev block lives in device memory
// ------------------------------------------------------------------------------------------
#define AMH_GLOBAL_MG(name)                                                     \
  __global__ void __launch_bounds__(AMH_BLOCK) g_##name(const MergeDev* G, u32 n) { \
    u32 i = blockIdx.x * AMH_BLOCK + threadIdx.x;                              \
    if (i < n) name(*G, i);                                                    \
  }
AMH_GLOBAL_MG(k_mg_new_flag) AMH_GLOBAL_MG(k_mg_pair_counts) AMH_GLOBAL_MG(k_mg_subtree) AMH_GLOBAL_MG(k_mg_gap)
AMH_GLOBAL_MG(k_mg_inherit) AMH_GLOBAL_MG(k_mg_extra_cnt) AMH_GLOBAL_MG(k_mg_rows_dst) AMH_GLOBAL_MG(k_mg_rows_new)
AMH_GLOBAL_MG(k_mg_succ_dst) AMH_GLOBAL_MG(k_mg_succ_new) AMH_GLOBAL_MG(k_mg_succ_extra) AMH_GLOBAL_MG(k_mg_succ_sort)
AMH_GLOBAL_MG(k_mg_doc_succ) AMH_GLOBAL_MG(k_mg_changes) AMH_GLOBAL_MG(k_mg_copy_chunks)
AMH_GLOBAL_MG(k_ap_parse)
AMH_GLOBAL_MG(k_ap_count)
AMH_GLOBAL_MG(k_ap_verify)
AMH_GLOBAL_MG(k_ap_map)
AMH_GLOBAL_MG(k_ap_chg_meta)
AMH_GLOBAL_MG(k_ap_new_flag)
AMH_GLOBAL_MG(k_ap_link)
AMH_GLOBAL_MG(k_ap_children)
AMH_GLOBAL_MG(k_ap_sort_children)
AMH_GLOBAL_MG(k_ap_succ_init)
AMH_GLOBAL_MG(k_ap_up_jump)
AMH_GLOBAL_MG(k_ap_succ_build)
AMH_GLOBAL_MG(k_ap_rank_jump)
AMH_GLOBAL_MG(k_ap_weight)
AMH_GLOBAL_MG(k_ap_top_k)
AMH_GLOBAL_MG(k_ap_top_place)
AMH_GLOBAL_MG(k_ap_emit)
AMH_GLOBAL_MG(k_ap_pred_cnt)
AMH_GLOBAL_MG(k_ap_succ_fill_u)
AMH_GLOBAL_MG(k_ap_succ_sort_u)
AMH_GLOBAL_MG(k_ap_succ_fill_d)
AMH_GLOBAL_MG(k_ap_tie_find)

template <int KIND> __global__ void __launch_bounds__(AMH_BLOCK) g_big_mg(const MergeDev* Gp) {
  const MergeDev& G = *Gp;
  const ApplyDev& A = G.A;
  __shared__ u32 sh_n, sh_off, sh_part[AMH_BLOCK];
  u32 cnt = G.big_n[KIND];
  if (cnt > G.big_cap) cnt = G.big_cap;
  const u32* items = G.big_items + 2 * (size_t)KIND * G.big_cap;
  for (u32 w = blockIdx.x; w < cnt; w += gridDim.x) {
    u32 a = items[2 * (size_t)w], b = items[2 * (size_t)w + 1];
    if (KIND == BIGK_SUCC_U) {   // a = row of U, b = pair
      const BatchDev& S = G.S;
      const u32* mb = G.map_b + G.mapb_base[b];
      u32 o = S.r_succ_base[a];
      blk_bitonic(S.r_succ_cnt[a],
                  [&](u32 i, u32 j) { return mg_lam(S.s_ctr[o + i], mb[S.s_actor[o + i]], S.s_ctr[o + j], mb[S.s_actor[o + j]]) < 0; },
                  [&](u32 i, u32 j) { u32 x = S.s_ctr[o + i]; S.s_ctr[o + i] = S.s_ctr[o + j]; S.s_ctr[o + j] = x;
                                      x = S.s_actor[o + i]; S.s_actor[o + i] = S.s_actor[o + j]; S.s_actor[o + j] = x; });
    } else if (KIND == BIGK_SUCC_M) {   // a = merged row
      const BatchDev& M = G.M;
      u32 o = M.r_succ_base[a];
      blk_bitonic(M.r_succ_cnt[a],
                  [&](u32 i, u32 j) { return mg_lam(M.s_ctr[o + i], M.s_actor[o + i], M.s_ctr[o + j], M.s_actor[o + j]) < 0; },
                  [&](u32 i, u32 j) { u32 x = M.s_ctr[o + i]; M.s_ctr[o + i] = M.s_ctr[o + j]; M.s_ctr[o + j] = x;
                                      x = M.s_actor[o + i]; M.s_actor[o + i] = M.s_actor[o + j]; M.s_actor[o + j] = x; });
    } else {
      // linked lists: gathered into a stretch of A.o_stack (unused between k_ap_children and k_ap_emit), sorted there.
      // BIGK_CHILDREN: a = parent row, b = 0 its updates (ascending id) / 1 its child elements (descending id)
      // BIGK_TIES: a = gap counter slot, b = pair
      u32 head = KIND == BIGK_CHILDREN ? (b ? A.o_head_ins[a] : A.o_head_upd[a]) : A.g_head[a];
      u32 p = KIND == BIGK_CHILDREN ? G.S.r_doc[a] : b;
      if (threadIdx.x == 0) {
        u32 n = 0;
        for (u32 q = head; q != AMH_NULL32; q = A.o_next[q]) n++;
        u32 off = atomicAdd(&G.big_n[4 + KIND], n);
        n = 0;
        for (u32 q = head; q != AMH_NULL32; q = A.o_next[q]) A.o_stack[off + n++] = q;
        sh_n = n; sh_off = off;
      }
      __syncthreads();
      u32 n = sh_n;
      u32* v = A.o_stack + sh_off;
      if (KIND == BIGK_CHILDREN) {
        blk_bitonic(n, [&](u32 i, u32 j) { int c = ap_id_cmp(G, p, v[i], v[j]); return b ? c > 0 : c < 0; },
                    [&](u32 i, u32 j) { u32 x = v[i]; v[i] = v[j]; v[j] = x; });
        for (u32 i = threadIdx.x; i < n; i += AMH_BLOCK) A.o_next[v[i]] = i + 1 < n ? v[i + 1] : AMH_NULL32;
        if (threadIdx.x == 0) { if (b) A.o_head_ins[a] = v[0]; else A.o_head_upd[a] = v[0]; }
      } else {
        blk_bitonic(n, [&](u32 i, u32 j) { return ap_top_less(G, p, v[i], v[j]); },
                    [&](u32 i, u32 j) { u32 x = v[i]; v[i] = v[j]; v[j] = x; });
        // rows in front of each top row inside the gap = exclusive prefix sum of the subtree weights in sorted order
        u32 per = (n + AMH_BLOCK - 1) / AMH_BLOCK, i0 = threadIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
        u32 sum = 0;
        for (u32 i = i0; i < i1; i++) sum += A.o_tmp[v[i]];
        sh_part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) { u32 run = 0; for (u32 t = 0; t < AMH_BLOCK; t++) { u32 x = sh_part[t]; sh_part[t] = run; run += x; } }
        __syncthreads();
        sum = sh_part[threadIdx.x];
        for (u32 i = i0; i < i1; i++) { u32 row = v[i]; A.o_top[row] = sum; sum += A.o_tmp[row]; }
      }
    }
    __syncthreads();
  }
}
// Subtree ends of the dst documents' sequence elements, thread per row (the emulation tier keeps the per-document
// stack walk, k_mg_subtree).  Rows are in preorder, so the subtree of element e ends at the first later row that is the
// first row of an object, or an element whose parent sits in front of e: with q[j] = parent position + 1 (0 for HEAD and
// for an object's first row, "none" for rows that are not elements) that is the first j > e with q[j] <= e.  Pass 1
// looks the parents up (id -> row table) and leaves q and its minimum per 64 and per 4096 rows; pass 2 finds the ends,
// skipping tiles whose minimum is too large; pass 3 checks that every parent is an ancestor (its subtree is still open).
__global__ void __launch_bounds__(AMH_BLOCK) g_mg_sub_parent(const MergeDev* Gp, u32 n) {
  const MergeDev& G = *Gp;
  const BatchDev& D = G.D;
  u32 r = blockIdx.x * AMH_BLOCK + threadIdx.x;
  u32 q = AMH_NULL32;
  if (r < n) {
    u32 a = D.r_doc[r];
    u32 p = a < D.n_docs ? G.d_pair[a] : AMH_NULL32;
    if (p != AMH_NULL32 && G.p_status[p] == 0) {
      u32 rb = D.row_base[a], i = r - rb;
      bool first = i == 0 || D.r_obj_ctr[r] != D.r_obj_ctr[r - 1] || D.r_obj_actor[r] != D.r_obj_actor[r - 1];
      if (first) q = 0;
      if (D.r_key_len[r] == AMH_NULL32 && D.r_insert[r]) {
        u32 kc = D.r_key_ctr[r];
        if (kc == 0) q = 0;
        else {
          u32 pr = ht_find(D, a, pack_id(kc, D.r_key_actor[r]));
          // the parent is an element of the same object in front of this row (anything else is not preorder)
          bool ok = !first && pr != AMH_EMPTY32 && pr >= rb && pr < r && D.r_key_len[pr] == AMH_NULL32 && D.r_insert[pr] &&
                    D.r_obj_ctr[pr] == D.r_obj_ctr[r] && D.r_obj_actor[pr] == D.r_obj_actor[r];
          if (!ok) G.p_status[p] = AMH_E_UNSUPPORTED;
          else q = pr - rb + 1;
        }
      }
    } else {
      q = 0;   // rows of documents that are not merged end every search
    }
    G.stack[r] = q;
  }
  u32 m = q;
  for (int off = 32; off; off >>= 1) { u32 o = __shfl_xor(m, off); m = o < m ? o : m; }
  if ((threadIdx.x & 63) == 0 && r < n) {
    G.sub_min1[r >> 6] = m;
    atomicMin(&G.sub_min2[r >> 12], m);
  }
}
__global__ void __launch_bounds__(AMH_BLOCK) g_mg_sub_end(const MergeDev* Gp, u32 n) {
  const MergeDev& G = *Gp;
  const BatchDev& D = G.D;
  u32 r = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (r >= n) return;
  u32 a = D.r_doc[r];
  u32 p = a < D.n_docs ? G.d_pair[a] : AMH_NULL32;
  if (p == AMH_NULL32 || G.p_status[p] != 0) return;
  u32 rb = D.row_base[a], i = r - rb, end = rb + D.cnt[a].n_rows;
  u32 se = i + 1, ue = i + 1;
  if (D.r_key_len[r] == AMH_NULL32 && D.r_insert[r]) {
    // own update rows: up to the next element row or object start
    u32 j = r + 1;
    while (j < end && G.stack[j] == AMH_NULL32) j++;
    ue = j - rb;
    // subtree: first j > r with q[j] <= i
    j = r + 1;
    for (;;) {
      if (j >= end) { j = end; break; }
      if ((j & 63) == 0) {   // at a tile start: skip tiles (and groups of 64 tiles) that hold nothing small enough
        if ((j & 4095) == 0 && G.sub_min2[j >> 12] > i) { j += 4096; continue; }
        if (G.sub_min1[j >> 6] > i) { j += 64; continue; }
      }
      if (G.stack[j] <= i) break;
      j++;
    }
    se = j - rb;
  }
  G.sub_end[r] = se;
  G.upd_end[r] = ue;
}
__global__ void __launch_bounds__(AMH_BLOCK) g_mg_sub_check(const MergeDev* Gp, u32 n) {
  const MergeDev& G = *Gp;
  const BatchDev& D = G.D;
  u32 r = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (r >= n) return;
  u32 q = G.stack[r];
  if (q == AMH_NULL32 || q == 0) return;
  u32 a = D.r_doc[r];
  u32 p = a < D.n_docs ? G.d_pair[a] : AMH_NULL32;
  if (p == AMH_NULL32 || G.p_status[p] != 0) return;
  u32 rb = D.row_base[a];
  if (G.sub_end[rb + q - 1] <= r - rb) G.p_status[p] = AMH_E_UNSUPPORTED;   // the parent's subtree ended before this row: not preorder
}
// the virtual chunk [dst chunk | src chunk]: one block per pair, 16-byte copies (chunks are 16-byte aligned and padded)
__global__ void __launch_bounds__(AMH_BLOCK) g_mg_copy_chunks_blk(const MergeDev* Gp) {
  const MergeDev& G = *Gp;
  u32 p = blockIdx.x;
  if (p >= G.n_pairs || G.p_status[p] != 0) return;
  u32 a = G.p_a[p], b = G.p_b[p];
  uint4* dst = (uint4*)(const_cast<u8*>(G.M.arena) + G.M.desc[p].chunk_off);
  const uint4* sa = (const uint4*)(G.D.arena + G.D.desc[a].chunk_off);
  const uint4* sb = (const uint4*)(G.S.arena + G.S.desc[b].chunk_off);
  u32 na = (G.D.desc[a].chunk_len + 15) / 16, nb = (G.S.desc[b].chunk_len + 15) / 16, shift = G.a_len_pad[p] / 16;
  for (u32 i = threadIdx.x; i < na; i += AMH_BLOCK) dst[i] = sa[i];
  for (u32 i = threadIdx.x; i < nb; i += AMH_BLOCK) dst[shift + i] = sb[i];
}

void amh_launch_mg(int k, const MergeDev* G, u32 n, void* stream) {
  if (n == 0) return;
  hipStream_t s = (hipStream_t)stream;
  if (k == MK_SUBTREE) {   // n = rows of the dst batch
    dim3 g((n + AMH_BLOCK - 1) / AMH_BLOCK), b(AMH_BLOCK);
    hipLaunchKernelGGL(g_mg_sub_parent, g, b, 0, s, G, n);
    hipLaunchKernelGGL(g_mg_sub_end, g, b, 0, s, G, n);
    hipLaunchKernelGGL(g_mg_sub_check, g, b, 0, s, G, n);
    return;
  }
  if (k == MK_COPY_CHUNKS) { hipLaunchKernelGGL(g_mg_copy_chunks_blk, dim3(n), dim3(AMH_BLOCK), 0, s, G); return; }
  dim3 grid((n + AMH_BLOCK - 1) / AMH_BLOCK), block(AMH_BLOCK);
#define M(id, name) case id: hipLaunchKernelGGL(g_##name, grid, block, 0, s, G, n); break;
  // the thread-per-item kernel, then a block per list it found too long for one thread
  if (k == MK_AP_SORT_CHILDREN) { hipLaunchKernelGGL(g_k_ap_sort_children, grid, block, 0, s, G, n); hipLaunchKernelGGL(g_big_mg<BIGK_CHILDREN>, dim3(BIG_GRID), block, 0, s, G); return; }
  if (k == MK_AP_TIES) { hipLaunchKernelGGL(g_k_ap_tie_find, grid, block, 0, s, G, n); hipLaunchKernelGGL(g_big_mg<BIGK_TIES>, dim3(BIG_GRID), block, 0, s, G); return; }
  if (k == MK_AP_SUCC_SORT_U) { hipLaunchKernelGGL(g_k_ap_succ_sort_u, grid, block, 0, s, G, n); hipLaunchKernelGGL(g_big_mg<BIGK_SUCC_U>, dim3(BIG_GRID), block, 0, s, G); return; }
  if (k == MK_SUCC_SORT) { hipLaunchKernelGGL(g_k_mg_succ_sort, grid, block, 0, s, G, n); hipLaunchKernelGGL(g_big_mg<BIGK_SUCC_M>, dim3(BIG_GRID), block, 0, s, G); return; }
  switch (k) {
    M(MK_NEW_FLAG, k_mg_new_flag) M(MK_PAIR_COUNTS, k_mg_pair_counts) M(MK_SUBTREE, k_mg_subtree) M(MK_GAP, k_mg_gap)
    M(MK_INHERIT, k_mg_inherit) M(MK_EXTRA_CNT, k_mg_extra_cnt) M(MK_ROWS_DST, k_mg_rows_dst) M(MK_ROWS_NEW, k_mg_rows_new)
    M(MK_SUCC_DST, k_mg_succ_dst) M(MK_SUCC_NEW, k_mg_succ_new) M(MK_SUCC_EXTRA, k_mg_succ_extra) M(MK_SUCC_SORT, k_mg_succ_sort)
    M(MK_DOC_SUCC, k_mg_doc_succ) M(MK_CHANGES, k_mg_changes) M(MK_COPY_CHUNKS, k_mg_copy_chunks)
    M(MK_AP_PARSE, k_ap_parse) M(MK_AP_COUNT, k_ap_count) M(MK_AP_VERIFY, k_ap_verify) M(MK_AP_MAP, k_ap_map) M(MK_AP_CHG_META, k_ap_chg_meta) M(MK_AP_NEW_FLAG, k_ap_new_flag) M(MK_AP_LINK, k_ap_link) M(MK_AP_CHILDREN, k_ap_children) M(MK_AP_SORT_CHILDREN, k_ap_sort_children) M(MK_AP_SUCC_INIT, k_ap_succ_init) M(MK_AP_UP_JUMP, k_ap_up_jump) M(MK_AP_SUCC_BUILD, k_ap_succ_build) M(MK_AP_RANK_JUMP, k_ap_rank_jump) M(MK_AP_WEIGHT, k_ap_weight) M(MK_AP_TOP_K, k_ap_top_k) M(MK_AP_TOP_PLACE, k_ap_top_place) M(MK_AP_EMIT, k_ap_emit) M(MK_AP_PRED_CNT, k_ap_pred_cnt) M(MK_AP_SUCC_FILL_U, k_ap_succ_fill_u) M(MK_AP_SUCC_SORT_U, k_ap_succ_sort_u) M(MK_AP_SUCC_FILL_D, k_ap_succ_fill_d)
    default: break;
  }
#undef M
}

// sync message kernels: one thread per unit, the SyncDev block lives in device memory
#define AMH_GLOBAL_SM(name)                                                     \
  __global__ void __launch_bounds__(AMH_BLOCK) g_##name(const SyncDev* G, u32 n) { \
    u32 i = blockIdx.x * AMH_BLOCK + threadIdx.x;                              \
    if (i < n) name(*G, i);                                                    \
  }
AMH_GLOBAL_SM(k_sm_lookup) AMH_GLOBAL_SM(k_sm_plan) AMH_GLOBAL_SM(k_sm_fill) AMH_GLOBAL_SM(k_sm_bloom) AMH_GLOBAL_SM(k_sm_jobs)
__global__ void __launch_bounds__(AMH_BLOCK) g_ap_counts_given(BatchDev U, const u64* nops, const u64* npred, const u64* place, u32 n) {
  const u32 k = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (k >= n) return;
  const u32 c = (u32)(place[k] / 32), rows = *(const u32*)(uintptr_t)nops[k], preds = *(const u32*)(uintptr_t)npred[k];
  const bool ok = U.status[c] == 0;
  U.cnt[c].n_rows = ok ? rows : 0; U.cnt[c].n_succ = ok ? preds : 0;
  U.row_base[c] = ok ? rows : 0; U.succ_base[c] = ok ? preds : 0;
}
void amh_ap_counts_given(const BatchDev& U, const u64* nops, const u64* npred, const u64* place, u32 n, void* stream) {
  if (n) hipLaunchKernelGGL(g_ap_counts_given, dim3((n + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, U, nops, npred, place, n);
}
__global__ void __launch_bounds__(AMH_BLOCK) g_sm_doc_sizes(BatchDev B, const u32* doc, u32* out, u32 n) {
  const u32 r = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (r >= n) return;
  const u32 d = doc[r];
  const bool ok = B.status[d] == 0;
  out[2 * r] = ok ? B.cnt[d].n_chg : 0;
  out[2 * r + 1] = ok ? B.desc[d].n_actors : 0;
}
void amh_sm_doc_sizes(const BatchDev& B, const u32* doc, u32* out, u32 n, void* stream) {
  if (n) hipLaunchKernelGGL(g_sm_doc_sizes, dim3((n + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, B, doc, out, n);
}
void amh_launch_sm(int k, const SyncDev* G, u32 n, void* stream) {
  if (n == 0) return;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((n + AMH_BLOCK - 1) / AMH_BLOCK), block(AMH_BLOCK);
  if (k == SK_LOOKUP) hipLaunchKernelGGL(g_k_sm_lookup, grid, block, 0, s, G, n);
  else if (k == SK_PLAN) hipLaunchKernelGGL(g_k_sm_plan, grid, block, 0, s, G, n);
  else if (k == SK_FILL) hipLaunchKernelGGL(g_k_sm_fill, grid, block, 0, s, G, n);
  else if (k == SK_BLOOM) hipLaunchKernelGGL(g_k_sm_bloom, grid, block, 0, s, G, n);
  else if (k == SK_JOBS) hipLaunchKernelGGL(g_k_sm_jobs, grid, block, 0, s, G, n);
}

// one block per segment, 16-byte copies (segments are 16-byte aligned and padded on both sides)
__global__ void __launch_bounds__(AMH_BLOCK) g_copy_segments(const u64* src, const u32* len, const u64* dst_off, u8* dst) {
  u32 i = blockIdx.x;
  const uint4* s = (const uint4*)(uintptr_t)src[i];
  uint4* d = (uint4*)(dst + dst_off[i]);
  u32 n = (len[i] + 15) / 16;
  for (u32 k = threadIdx.x; k < n; k += AMH_BLOCK) d[k] = s[k];
}
// one block per segment, any alignment: dwords where source and destination share their phase, bytes otherwise
template <class OFF>
__global__ void __launch_bounds__(AMH_BLOCK) g_copy_bytes(const u64* src, const u32* len, const OFF* dst_off, u8* dst) {
  const u32 i = blockIdx.x;
  const u8* s = (const u8*)(uintptr_t)src[i];
  u8* d = dst + dst_off[i];
  const u32 n = len[i];
  if ((((uintptr_t)s ^ (uintptr_t)d) & 3) == 0) {
    u32 head = (4 - ((uintptr_t)d & 3)) & 3;
    if (head > n) head = n;
    if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
    const u32 words = (n - head) / 4;
    const u32* sw = (const u32*)(s + head);
    u32* dw = (u32*)(d + head);
    for (u32 k = threadIdx.x; k < words; k += AMH_BLOCK) dw[k] = sw[k];
    const u32 done = head + 4 * words;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
  } else {
    for (u32 k = threadIdx.x; k < n; k += AMH_BLOCK) d[k] = s[k];
  }
}
void amh_copy_bytes(const u64* src, const u32* len, const u32* dst_off, u8* dst, u32 n, void* stream) {
  if (n == 0) return;
  hipLaunchKernelGGL((g_copy_bytes<u32>), dim3(n), dim3(AMH_BLOCK), 0, (hipStream_t)stream, src, len, dst_off, dst);
}
void amh_copy_bytes64(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void* stream) {
  if (n == 0) return;
  hipLaunchKernelGGL((g_copy_bytes<u64>), dim3(n), dim3(AMH_BLOCK), 0, (hipStream_t)stream, src, len, dst_off, dst);
}
void amh_copy_segments(const u64* src, const u32* len, const u64* dst_off, u8* dst, u32 n, void* stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(g_copy_segments, dim3(n), dim3(AMH_BLOCK), 0, (hipStream_t)stream, src, len, dst_off, dst);
}
// The heads of the staged change chunks (everything in front of the op column metadata: dependency hashes, author,
// message, other actors), packed back to back for the copy to the host: the decisions of an apply read nothing else of
// a chunk, and the heads are a fraction of the bytes.  Lengths and places in units of 16 bytes (chunks start 16-byte
// aligned and the gap behind each is zero).
__global__ void __launch_bounds__(AMH_BLOCK) g_ap_head_len(const ChgHdr* hdr, const i32* status, u32* len16, u32 n) {
  const u32 u = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (u < n) len16[u] = status[u] ? 0u : (hdr[u].head_len + 15u) & ~15u;
}
__global__ void __launch_bounds__(AMH_BLOCK) g_ap_head_copy(const u8* arena, const u64* doc_off, const u32* len16, const u32* off, u8* dst, u32 n) {
  const u32 u = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (u >= n) return;
  const uint4* s = (const uint4*)(arena + doc_off[2 * (u64)u]);
  uint4* d = (uint4*)(dst + off[u]);
  const u32 k = len16[u] / 16;
  for (u32 i = 0; i < k; i++) d[i] = s[i];
}
// The actor tables of the documents of a batch (the bytes from the first actor id to the end of the last one, length
// prefixes included), packed back to back for the host: all an apply reads of the documents it changes.
__global__ void __launch_bounds__(AMH_BLOCK) g_actor_pack_len(BatchDev B, u32* len) {
  const u32 d = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (d >= B.n_docs) return;
  const u32 na = B.status[d] ? 0 : B.desc[d].n_actors, ab = B.actor_base[d];
  len[d] = na ? B.actor_off[ab + na - 1] + B.actor_len[ab + na - 1] - B.actor_off[ab] : 0;
}
__global__ void __launch_bounds__(AMH_BLOCK) g_actor_pack_copy(BatchDev B, const u32* len, const u32* off, u8* dst) {
  const u32 d = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (d >= B.n_docs || len[d] == 0) return;
  const u8* s = B.arena + B.desc[d].chunk_off + B.actor_off[B.actor_base[d]];
  u8* o = dst + off[d];
  for (u32 i = 0; i < len[d]; i++) o[i] = s[i];
}
void amh_actor_pack_len(const BatchDev& B, u32* len, void* stream) {
  if (B.n_docs == 0) return;
  hipLaunchKernelGGL(g_actor_pack_len, dim3((B.n_docs + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, B, len);
}
void amh_actor_pack_copy(const BatchDev& B, const u32* len, const u32* off, u8* dst, void* stream) {
  if (B.n_docs == 0) return;
  hipLaunchKernelGGL(g_actor_pack_copy, dim3((B.n_docs + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, B, len, off, dst);
}
void amh_ap_head_len(const ChgHdr* hdr, const i32* status, u32* len16, u32 n, void* stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(g_ap_head_len, dim3((n + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, hdr, status, len16, n);
}
void amh_ap_head_copy(const u8* arena, const u64* doc_off, const u32* len16, const u32* off, u8* dst, u32 n, void* stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(g_ap_head_copy, dim3((n + AMH_BLOCK - 1) / AMH_BLOCK), dim3(AMH_BLOCK), 0, (hipStream_t)stream, arena, doc_off, len16, off, dst, n);
}


// ------------------------------------------------------------------------------------------
// amh_sync_summaries_noalloc (include/automerge_hip.h): the sync summaries of a batch written into the caller's device
// buffer with no allocation of any kind.  What amh_sync_summaries takes from the pool per call — the record offsets
// and the scan's tile sums — lives in this file-scope device array instead.  Each GPU the library runs on holds its own
// copy of the array (at its own address) until the process ends; every call on a GPU shares it, so calls that overlap
// on one GPU race.  Nothing in it has to start at zero: the size kernel writes sync_off[0 .. n), the scan writes
// sync_off[n] and every tile sum before it reads it.
// ------------------------------------------------------------------------------------------
#define AMH_NOALLOC_SCAN_WORDS 1024u   // amh_scan_tmp_words(AMH_NOALLOC_MAX_DOCS) = 512 + 1 + 2, rounded up
#define AMH_NOALLOC_WORDS (AMH_NOALLOC_MAX_DOCS + 64u + AMH_NOALLOC_SCAN_WORDS)
static __device__ __align__(256) u32 scratch[AMH_NOALLOC_WORDS];   // 4 MiB + 4 KiB
#define AMH_NOALLOC_MAX_DEVICES 64
static u32* g_scratch_ptr[AMH_NOALLOC_MAX_DEVICES];                // device address of `scratch` on each GPU (never the array's name)

__global__ void g_sync_meta(const u32* total, u32 cap, u32* meta) {
  u32 t = *total;
  meta[0] = t;
  meta[1] = (t > cap || t == 0xFFFFFFFFu) ? 1u : 0u;   // does not fit, or the record offsets saturated (more than 4 GiB - 1 of records)
}

int amh_sync_noalloc_run(const BatchDev& B0, int device, u8* d_out, u32 cap, u32* d_meta, void* stream, const char** err) {
  hipStream_t s = (hipStream_t)stream;
  if (device < 0 || device >= AMH_NOALLOC_MAX_DEVICES) { *err = "device index out of range"; return 1; }
  if (B0.n_docs > AMH_NOALLOC_MAX_DOCS) { *err = "too many documents for the static scratch area (AMH_NOALLOC_MAX_DOCS)"; return 2; }
  if (amh_scan_tmp_words(B0.n_docs) > AMH_NOALLOC_SCAN_WORDS) { *err = "scan scratch too small"; return 2; }
  if (!g_scratch_ptr[device]) {   // once per device, with that device current
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return 1; }
    u32* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) { *err = "hipGetSymbolAddress(scratch) failed"; return 1; }
    g_scratch_ptr[device] = p;
  }
  u32* ptr = g_scratch_ptr[device];
  BatchDev B = B0;
  const u32 n = B.n_docs;
  B.sync_off = ptr;
  B.sync_out = d_out;
  B.sync_cap = cap ? cap : 1u;   // 0 would mean "no limit": a zero-capacity buffer receives nothing
  if (n == 0) {
    if (hipMemsetAsync(d_meta, 0, 2 * sizeof(u32), s) != hipSuccess) { *err = "hipMemsetAsync failed"; return 1; }
    return 0;
  }
  amh_launch(AK_SYNC_SIZE, B, n, stream);
  amh_scan_inplace(ptr, n, ptr + AMH_NOALLOC_MAX_DOCS + 64u, stream);
  amh_launch(AK_SYNC_FILL, B, n, stream);
  hipLaunchKernelGGL(g_sync_meta, dim3(1), dim3(1), 0, s, (const u32*)(ptr + n), cap, d_meta);
  if (hipGetLastError() != hipSuccess) { *err = "kernel launch failed"; return 1; }
  return 0;
}
