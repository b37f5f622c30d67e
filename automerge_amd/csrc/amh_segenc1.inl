// One-pass change re-encode (device only; included by amh_kernels.hip behind amh_segenc.inl).
//
// Round 2 spent three passes over the ops of every change: k_change_view gathered twelve fields per op into change
// order (92 B/op of traffic), g_encode_seg<false> walked them for the column lengths and g_encode_seg<true> walked them
// again for the bytes.  Here a wave per (change, column group) walks the change's ops ONCE, reading each op's fields
// through the op -> row indirection (ops_sorted) where they lie, and writes the column bytes as it goes:
//   * the walk is the backward walk of amh_segenc.inl (run structure from ballots, four carried scalars), so bytes are
//     placed counting down from the END of the column — which needs no column length, only a place to put the end:
//     every (change, column) owns a slot in a scratch arena, sized from a per-op budget, and is written right-aligned
//     into it;
//   * the column length falls out of the same walk (c_col_len); k_chg_layout then fixes the chunk layout and
//     g_encode_place copies every column from the tail of its slot to its place in the chunk (E' bytes, 1-6 B/op);
//   * a column that outgrows its slot keeps counting (its length stays exact) but stops storing, and is flagged in
//     c_wide; g_encode_one<1> walks the flagged columns again, straight into the chunk.  Budgets are sized so that
//     text and ordinary map changes never take that path (hostile inputs may: they cost a second walk, not an error);
//   * the tile in front of the one being encoded is loaded before the encode of the current one starts, so its gathers
//     (ops_sorted -> row -> fields, a chain of three dependent loads) are in flight during the ballots and scans.
// Compulsory traffic of the stage: 44 B/op gathered + E' written + E' copied (44 + E').
#pragma once

// slot budget of column k in bytes per item (op, or pred entry for columns 10 / 11) + SC1_PAD per column
#define SC1_PAD 16u
__host__ __device__ constexpr u32 sc1_per(u32 k) {
  return k == 0 ? 2u : k == 1 ? 2u : k == 2 ? 2u : k == 3 ? 4u : k == 4 ? 4u : k == 5 ? 1u : k == 6 ? 2u : k == 7 ? 2u : k == 8 ? 4u
       : k == 9 ? 2u : k == 10 ? 2u : k == 11 ? 4u : k == 12 ? 1u : 2u;
}
__host__ __device__ constexpr bool sc1_is_pred(u32 k) { return k == 10 || k == 11; }
__host__ __device__ constexpr u32 sc1_prefix(u32 k, bool pred) {   // budgets of the columns in front of k, per op / per pred
  u32 s = 0;
  for (u32 q = 0; q < k; q++) if (sc1_is_pred(q) == pred) s += sc1_per(q);
  return s;
}
#define SC1_OP_SUM sc1_prefix(AMH_NCOLS, false)
#define SC1_PRED_SUM sc1_prefix(AMH_NCOLS, true)
__device__ __forceinline__ u8* sc1_slot(const BatchDev& B, u32 c, u32 k, u32 base, u32 n, u32 pb, u32 pn, u32& cap) {
  const u64 region = (u64)base * SC1_OP_SUM + (u64)pb * SC1_PRED_SUM + (u64)c * (AMH_NCOLS * SC1_PAD);
  cap = (sc1_is_pred(k) ? pn : n) * sc1_per(k) + SC1_PAD;
  return B.enc_tmp + region + (u64)n * sc1_prefix(k, false) + (u64)pn * sc1_prefix(k, true) + (u64)k * SC1_PAD;
}

struct Sc1Col {
  u8* end;       // one past the last byte of the slot (or of the column's place in the chunk)
  u32 cap;       // bytes that may be written in front of `end`
  u32 total;     // bytes of the values walked so far: exact even when they no longer fit
  ScCarry carry;
  bool any;      // a non-null (true) value seen
};
__device__ __forceinline__ void sc1_init(Sc1Col& o, u8* slot, u32 cap) {
  o.end = slot ? slot + cap : (u8*)0; o.cap = slot ? cap : 0; o.total = 0; o.carry = sc_carry_init(); o.any = false;
}
__device__ __forceinline__ u32 sc1_prev(u32 v) { return sd_prev(v); }
__device__ __forceinline__ u64 sc1_prev(u64 v) { return sd_prev64(v); }
__device__ __forceinline__ u32 sc1_last(u32 v) { return sd_bcast(v, 63); }
__device__ __forceinline__ u64 sc1_last(u64 v) { return sd_bcast64(v, 63); }
// a tile whose values all continue the run of the value in front of it writes nothing: only the carry moves
__device__ __forceinline__ void sc1_carry_run(ScCarry& c, u32 cnt) {
  c.runlen0 = c.start0 ? cnt : cnt + c.runlen0;
  c.start0 = false; c.lone0 = false; c.litlen0 = 0;
}

// ---- one tile of a numeric column.  T = u32 | u64; SIGNED: sleb128 of (i64)v.  pv0 / pn0: the value in front of the
// tile (lane 63 of the tile loaded ahead; ignored at the start of the column) -----------------------------------------
template <bool SIGNED, class T>
__device__ __forceinline__ void sc1_num(Sc1Col& o, T v, bool nul, T pv0, bool pn0, u32 base, u32 cnt, u32 lane) {
  const bool valid = lane < cnt;
  if (!valid) { v = 0; nul = true; }
  T pv = sc1_prev(v); bool pn = sd_prev((u32)nul) != 0;
  if (lane == 0) { pv = pv0; pn = pn0; }
  const bool eqp = valid && base + lane > 0 && pn == nul && (nul || pv == v);
  const u64 VALID = sd_low_bits(cnt), EQP = __ballot(eqp), NUL = __ballot(nul && valid);
  o.any = o.any || (~NUL & VALID) != 0;
  if (EQP == VALID && !o.carry.lone0) { sc1_carry_run(o.carry, cnt); return; }
  const ScRoles r = sc_roles(EQP, NUL, cnt, base, lane, o.carry);
  const bool hasP = r.emit_p, hasH = r.is_start && !r.is_lone, hasV = r.is_start, hasL = r.is_lh;
  const i64 nP = -(i64)r.k_p, nH = nul ? 0 : (i64)r.rl, nL = -(i64)r.k_lh;
  const i64 sv = sizeof(T) == 8 ? (i64)v : (i64)(u64)v;
  const u32 sP = hasP ? sleb_size_fast(nP) : 0, sH = hasH ? sleb_size_fast(nH) : 0, sL = hasL ? sleb_size_fast(nL) : 0;
  const u32 sV = !hasV ? 0 : (nul ? uleb_size_fast(r.rl) : (SIGNED ? sleb_size_fast(sv) : uleb_size_fast((u64)v)));
  const u32 len = sP + sH + sV + sL;
  const u32 inc = sd_scan_add(len);
  o.total += sd_bcast(inc, 63);
  if (len && o.total <= o.cap) {
    u8* p = o.end - o.total + (inc - len);
    if (hasP) p += wr_sleb(p, nP);
    if (hasH) p += wr_sleb(p, nH);
    if (hasV) p += nul ? wr_uleb(p, r.rl) : (SIGNED ? wr_sleb(p, sv) : wr_uleb(p, (u64)v));
    if (hasL) wr_sleb(p, nL);
  }
}
// ---- one tile of a boolean column (boolean.rs:35-50); o.any = a true value seen --------------------------------------
__device__ __forceinline__ void sc1_bool(Sc1Col& o, bool v, bool pv0, u32 base, u32 cnt, u32 lane) {
  const bool valid = lane < cnt;
  v = v && valid;
  bool pv = sd_prev((u32)v) != 0;
  if (lane == 0) pv = pv0;
  const bool eqp = valid && base + lane > 0 && pv == v;
  const u64 VALID = sd_low_bits(cnt), EQP = __ballot(eqp);
  o.any = o.any || __ballot(v) != 0;
  if (EQP == VALID) { sc1_carry_run(o.carry, cnt); return; }
  const ScRoles r = sc_roles(EQP, 0, cnt, base, lane, o.carry);
  const bool hasZ = base == 0 && lane == 0 && v;   // a column that starts with `true` starts with an empty run of `false`
  const u32 len = (hasZ ? 1u : 0u) + (r.is_start ? uleb_size_fast(r.rl) : 0u);
  const u32 inc = sd_scan_add(len);
  o.total += sd_bcast(inc, 63);
  if (len && o.total <= o.cap) {
    u8* p = o.end - o.total + (inc - len);
    if (hasZ) *p++ = 0;
    if (r.is_start) wr_uleb(p, r.rl);
  }
}
// ---- one tile of a string column; pv = the string in front of this lane's ------------------------------------------
__device__ __forceinline__ ScStr sc1_str_null() { ScStr s; s.bytes = 0; s.p = 0; s.len = 0; s.inl = true; s.isnull = true; return s; }
__device__ __forceinline__ ScStr sc1_str_prev(const ScStr& v, const ScStr& front, u32 lane) {
  // (the broadcasts of the front tile's last string are taken by the whole wave, outside the lane-0 branch)
  const u64 fb = sd_bcast64(front.bytes, 63), fp = sd_bcast64((u64)front.p, 63);
  const u32 fl = sd_bcast(front.len, 63), ff = sd_bcast((front.inl ? 1u : 0u) | (front.isnull ? 2u : 0u), 63);
  u64 b = sd_prev64(v.bytes), q = sd_prev64((u64)v.p);
  u32 l = sd_prev(v.len), f = sd_prev((v.inl ? 1u : 0u) | (v.isnull ? 2u : 0u));
  if (lane == 0) { b = fb; q = fp; l = fl; f = ff; }
  ScStr p;
  p.bytes = b; p.p = (const u8*)q; p.len = l; p.inl = (f & 1) != 0; p.isnull = (f & 2) != 0;
  return p;
}
__device__ __forceinline__ void sc1_str(Sc1Col& o, ScStr v, const ScStr& front, u32 base, u32 cnt, u32 lane) {
  const bool valid = lane < cnt;
  if (!valid) v = sc1_str_null();
  // (every lane takes part in the shuffles; lane 0 of the column's first tile compares with nothing)
  const ScStr pv = sc1_str_prev(v, front, lane);
  const u64 VALID = sd_low_bits(cnt), NUL = __ballot(v.isnull && valid);
  o.any = o.any || (~NUL & VALID) != 0;
  bool eqp = valid && base + lane > 0;
  if (NUL == VALID) eqp = eqp && pv.isnull;   // a tile of nulls (text changes: every tile): no string compares
  else eqp = eqp && sc_str_eq(v, pv);
  const u64 EQP = __ballot(eqp);
  if (EQP == VALID && !o.carry.lone0) { sc1_carry_run(o.carry, cnt); return; }
  const ScRoles r = sc_roles(EQP, NUL, cnt, base, lane, o.carry);
  const bool nul = v.isnull;
  const bool hasP = r.emit_p, hasH = r.is_start && !r.is_lone, hasV = r.is_start, hasL = r.is_lh;
  const i64 nP = -(i64)r.k_p, nH = nul ? 0 : (i64)r.rl, nL = -(i64)r.k_lh;
  const u32 sP = hasP ? sleb_size_fast(nP) : 0, sH = hasH ? sleb_size_fast(nH) : 0, sL = hasL ? sleb_size_fast(nL) : 0;
  const u32 sV = !hasV ? 0 : (nul ? uleb_size_fast(r.rl) : uleb_size_fast(v.len) + v.len);
  const u32 len = sP + sH + sV + sL;
  const u32 inc = sd_scan_add(len);
  o.total += sd_bcast(inc, 63);
  if (len && o.total <= o.cap) {
    u8* p = o.end - o.total + (inc - len);
    if (hasP) p += wr_sleb(p, nP);
    if (hasH) p += wr_sleb(p, nH);
    if (hasV) {
      if (nul) p += wr_uleb(p, r.rl);
      else {
        p += wr_uleb(p, v.len);
        if (v.inl) for (u32 i = 0; i < v.len; i++) p[i] = (u8)(v.bytes >> (8 * i));
        else for (u32 i = 0; i < v.len; i++) p[i] = v.p[i];
        p += v.len;
      }
    }
    if (hasL) wr_sleb(p, nL);
  }
}
// ---- one tile of the raw value bytes (value.rs:320-369) ------------------------------------------------------------
__device__ __forceinline__ void sc1_raw(Sc1Col& o, const ScRaw& v, u32 cnt, u32 lane) {
  const u32 len = lane >= cnt || v.kind == 0 ? 0 : (v.kind == 2 ? sleb_size_fast((i64)v.x) : v.len);
  const u32 inc = sd_scan_add(len);
  o.total += sd_bcast(inc, 63);
  if (len && o.total <= o.cap) {
    u8* p = o.end - o.total + (inc - len);
    if (v.kind == 2) wr_sleb(p, (i64)v.x);
    else if (v.kind == 3) for (u32 i = 0; i < len; i++) p[i] = (u8)(v.x >> (8 * i));
    else for (u32 i = 0; i < len; i++) p[i] = v.p[i];
  }
}

// the walk: tiles of 64 values from the last to the first.  The tile in front is loaded before the current one is
// encoded: the step needs its last value anyway, and its gathers (ops_sorted -> row -> fields -> actor map, three or four
// dependent loads) are in flight during the ballots and scans.  Two tiles ahead was measured: 18 more registers, one
// wave per SIMD fewer, 4.77 -> 5.26 ms.
template <class V, class Load, class Step>
__device__ __forceinline__ void sc1_walk(u32 n, u32 lane, Load load, Step step) {
  if (n == 0) return;
  u32 base = (n - 1) & ~63u;
  V cur = load(base + lane, base + lane < n), nxt = cur;
  for (;;) {
    const u32 cnt = n - base < 64 ? n - base : 64;
    if (base) nxt = load(base - 64 + lane, true);
    step(cur, nxt, base, cnt);
    if (base == 0) break;
    cur = nxt; base -= 64;
  }
}

struct Sc1Obj { u32 oa, oc; };
struct Sc1Key { u32 ka; u32 kind; u64 delta; ScStr s; };    // kind 0 prop, 1 head, 2 elem, 3 = not loaded
struct Sc1Misc { u32 f, np; };                              // insert | expand << 1 | action << 8, pred count
struct Sc1Val { u64 meta; ScRaw raw; };
struct Sc1Pred { u32 pa; u64 delta; };

// MODE 0: every column into its slot, lengths -> c_col_len, columns that outgrew their slot -> c_wide.
// MODE 1 (behind k_chg_layout): the flagged columns again, straight into the chunk.
// The column groups of a change are dealt over `parts` workgroups (1 = one wave encodes all the columns of a change:
// batches of smaller changes; SC_GROUPS = a wave per group).  The group is the FAST index of the grid: workgroups that
// are resident together then work on different groups, so the waves that mostly wait for their gathers (object ids: every
// tile a continued run) share a SIMD with the ones that mostly compute (keys, values) — with the group as the slow index
// the launch ran group after group, 8.1 ms of group times against 4.8 ms; mixed: see DESIGN 5.
template <int MODE>
__global__ void __launch_bounds__(64 * SD_WAVES) g_encode_one(BatchDev B, u32 gmask, u32 parts) {
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  // XCD x takes the x-th eighth of the changes (the grid is a multiple of 8): the changes of one document gather the same
  // rows — each touches most sectors of every field it reads — so they sit behind ONE L2 instead of three
  const u32 blk = sd_uniform(amh_xcd_block());
  const u32 c = (blk / parts) * SD_WAVES + wave;
  if (c >= B.tot_chg) return;
  const u32 lane = threadIdx.x & 63;
  u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
  const u32 d = B.c_doc[c];
  const bool big = B.sv_big && B.sv_big[c];   // its columns are cut into ranges (g_en_big): lengths set by g_sv_big_total
  const bool live = B.status[d] == 0 && !big;
  const u32 flags = MODE == 1 ? B.c_wide[c] : 0;
  if (MODE == 1 && (!live || flags == 0)) return;
  const u32 n = live ? B.c_nops[c] : 0;
  const u32 base = B.c_op_base[c];
  const u8* chunk = B.arena + B.desc[d].chunk_off;
  const u32* ops = B.ops_sorted + base;
  const u32* amap = B.chg_actor_map + B.c_map_base[c];
  u32 pb = 0, pn = 0;
  if (n) { pb = B.j_pred_base[base]; pn = B.j_pred_base[base + n] - pb; }
  u32 coff[AMH_NCOLS], clen[AMH_NCOLS];
  u8* cols = (u8*)0;
  if (MODE == 1) {
    cols = B.out_arena + (u64)B.c_out_off[c] + B.c_cols_off[c];
    u32 o = 0;
#pragma unroll
    for (int k = 0; k < AMH_NCOLS; k++) { clen[k] = cl[k]; coff[k] = o; o += clen[k]; }
  }
  auto open = [&](Sc1Col& o, u32 k) {
    if (MODE == 0) { u32 cap; u8* s = sc1_slot(B, c, k, base, n, pb, pn, cap); if (gmask & 0x100u) { s += cap; cap = 0; } sc1_init(o, s, cap); }   // (development, bit 8 of AMH_ENC_GMASK: no column fits its slot, every one is written by the second walk — what the first walk costs without its stores)
    else if (((flags >> k) & 1) && clen[k]) sc1_init(o, cols + coff[k], clen[k]);
    else sc1_init(o, (u8*)0, 0);
  };
  auto close = [&](const Sc1Col& o, u32 k, bool always) {
    if (MODE != 0 || lane != 0) return;
    const u32 len = (always || o.any) ? o.total : 0;   // a column of nulls is not written at all (rle.rs:33, boolean.rs:131-137)
    cl[k] = len;
    if (len > o.cap) atomicOr(&B.c_wide[c], 1u << k);
  };
  for (u32 G = blk % parts; G < SC_GROUPS; G += parts) {
    const u32 gm = G == 0 ? 0x2003u : G == 1 ? 0x001Cu : G == 2 ? 0x1260u : G == 3 ? 0x0180u : 0x0C00u;   // the group's columns
    if (!((gmask >> G) & 1)) {   // (development: AMH_ENC_GMASK times the column groups one by one; a group left out has empty columns)
      if (MODE == 0 && lane == 0) for (u32 k = 0; k < AMH_NCOLS; k++) if ((gm >> k) & 1) cl[k] = 0;
      continue;
    }
    if (MODE == 1 && !(flags & gm)) continue;   // groups without a flagged column
    if (G == 0) {   // object id (columns 0, 1) and mark names (13)
      Sc1Col ca, cc;
      open(ca, 0); open(cc, 1);
      sc1_walk<Sc1Obj>(n, lane,
        [&](u32 e, bool valid) {
          Sc1Obj v; v.oa = 0; v.oc = 0;
          if (valid) { const u32 kr = op_key_row(B, ops[e]); v.oc = B.r_obj_ctr[kr]; if (v.oc) v.oa = amap[B.r_obj_actor[kr]]; }
          return v;
        },
        [&](const Sc1Obj& v, const Sc1Obj& f, u32 tb, u32 cnt) {
          if (gmask & 0x200u) { ca.total += (u32)__popcll(__ballot(((v.oa ^ v.oc) & 1) != 0)); return; }   // (development, bit 9: gathers only)
          const bool pn0 = sc1_last(f.oc) == 0;
          sc1_num<false, u32>(ca, v.oa, v.oc == 0, sc1_last(f.oa), pn0, tb, cnt, lane);
          sc1_num<false, u32>(cc, v.oc, v.oc == 0, sc1_last(f.oc), pn0, tb, cnt, lane);
        });
      close(ca, 0, false); close(cc, 1, false);
      Sc1Col cm;
      open(cm, 13);
      if (live && B.desc[d].ocol[OC_MARK_NAME].len != 0) {   // only documents with a mark column
        ScMarkSrc sm{B, ops, chunk};
        sc1_walk<ScStr>(n, lane,
          [&](u32 e, bool valid) { return valid ? sm.load(e) : sc1_str_null(); },
          [&](const ScStr& v, const ScStr& f, u32 tb, u32 cnt) { sc1_str(cm, v, f, tb, cnt, lane); });
      }
      close(cm, 13, false);
    } else if (G == 1) {   // key (2, 3, 4)
      Sc1Col ca, ck, cs;
      open(ca, 2); open(ck, 3); open(cs, 4);
      auto key_ctr = [&](u32 e, u32& ctr) -> bool {   // the op's key counter (HEAD = 0); false: a map key (null)
        const OpKey k = op_key(B, ops[e]);
        ctr = k.kind == 2 ? k.ctr : 0;
        return k.kind != 0;
      };
      sc1_walk<Sc1Key>(n, lane,
        [&](u32 e, bool valid) {
          Sc1Key v; v.ka = 0; v.kind = 3; v.delta = 0; v.s = sc1_str_null();
          u32 ctr = 0;
          if (valid) {
            const OpKey k = op_key(B, ops[e]);
            v.kind = k.kind;
            if (k.kind == 2) { v.ka = amap[k.actor]; ctr = k.ctr; }
            if (k.kind == 0) {
              v.s.isnull = false; v.s.len = k.str_len; v.s.p = chunk + k.str_off; v.s.inl = k.str_len <= 8;
              if (v.s.inl) v.s.bytes = k.str_len ? amh_peek8_any(v.s.p) & amh_low_bytes(k.str_len) : 0;
            }
          }
          // delta column: the difference to the last non-null counter in front (delta.rs:23-27): inside the tile by a
          // shuffle, else (the first lanes of a tile, mostly lane 0 alone) by walking back over the ops in front
          const bool nn = valid && v.kind != 0;
          const u64 NN = __ballot(nn), below = NN & sd_low_bits(lane);
          u32 pc = (u32)__shfl((int)ctr, below ? 63 - (int)__builtin_clzll(below) : 0, 64);
          if (nn && !below) {
            pc = 0;
            for (u32 j = e; j > 0; j--) { u32 t; if (key_ctr(j - 1, t)) { pc = t; break; } }
          }
          v.delta = (u64)((i64)ctr - (i64)pc);
          return v;
        },
        [&](const Sc1Key& v, const Sc1Key& f, u32 tb, u32 cnt) {
          if (gmask & 0x200u) { ca.total += (u32)__popcll(__ballot(((v.ka ^ v.kind ^ (u32)v.delta ^ v.s.len) & 1) != 0)); return; }
          const u32 fk = sc1_last(f.kind);
          sc1_num<false, u32>(ca, v.ka, v.kind != 2, sc1_last(f.ka), fk != 2, tb, cnt, lane);
          sc1_num<true, u64>(ck, v.delta, v.kind == 0 || v.kind == 3, sc1_last(f.delta), fk == 0, tb, cnt, lane);
          sc1_str(cs, v.s, f.s, tb, cnt, lane);
        });
      close(ca, 2, false); close(ck, 3, false); close(cs, 4, false);
    } else if (G == 2) {   // insert, action, predNum, expand (5, 6, 9, 12)
      Sc1Col ci, ca, cp, ce;
      open(ci, 5); open(ca, 6); open(cp, 9); open(ce, 12);
      sc1_walk<Sc1Misc>(n, lane,
        [&](u32 e, bool valid) {
          Sc1Misc v; v.f = 0; v.np = 0;
          if (valid) {
            const u32 t = ops[e];
            const bool del = op_is_del(B, t);
            const u32 action = del ? 3 : B.r_action[t];
            const u32 ins = del ? 0 : (B.r_insert[t] != 0);
            const u32 ex = (!del && action == 7 && B.r_expand[t]) ? 1 : 0;
            v.f = ins | (ex << 1) | (action << 8);
            v.np = B.t_pred_cnt[t];
          }
          return v;
        },
        [&](const Sc1Misc& v, const Sc1Misc& f, u32 tb, u32 cnt) {
          if (gmask & 0x200u) { ci.total += (u32)__popcll(__ballot(((v.f ^ v.np) & 1) != 0)); return; }
          const u32 ff = sc1_last(f.f);
          sc1_bool(ci, (v.f & 1) != 0, (ff & 1) != 0, tb, cnt, lane);
          sc1_num<false, u32>(ca, v.f >> 8, false, ff >> 8, false, tb, cnt, lane);
          sc1_num<false, u32>(cp, v.np, false, sc1_last(f.np), false, tb, cnt, lane);
          sc1_bool(ce, (v.f & 2) != 0, (ff & 2) != 0, tb, cnt, lane);
        });
      close(ci, 5, true); close(ca, 6, false); close(cp, 9, false); close(ce, 12, false);
    } else if (G == 3) {   // value (7, 8)
      Sc1Col cm, cr;
      open(cm, 7); open(cr, 8);
      sc1_walk<Sc1Val>(n, lane,
        [&](u32 e, bool valid) {
          Sc1Val v; v.meta = 0; v.raw.kind = 0; v.raw.len = 0; v.raw.x = 0; v.raw.p = 0;
          if (valid) {
            const OpVal x = op_val(B, ops[e]);
            v.meta = x.meta;
            if (x.kind == 2) { v.raw.kind = 2; v.raw.x = (u64)x.i; }
            else if (x.kind == 1 && x.raw_len) {
              v.raw.len = x.raw_len; v.raw.p = x.raw;
              if (x.raw_len <= 8) { v.raw.kind = 3; v.raw.x = amh_peek8_any(x.raw) & amh_low_bytes(x.raw_len); }
              else v.raw.kind = 1;
            }
          }
          return v;
        },
        [&](const Sc1Val& v, const Sc1Val& f, u32 tb, u32 cnt) {
          if (gmask & 0x200u) { cm.total += (u32)__popcll(__ballot((((u32)v.meta ^ (u32)v.raw.x ^ v.raw.len) & 1) != 0)); return; }
          sc1_num<false, u64>(cm, v.meta, false, sc1_last(f.meta), false, tb, cnt, lane);
          sc1_raw(cr, v.raw, cnt, lane);
        });
      close(cm, 7, false); close(cr, 8, true);
    } else {   // pred lists (10, 11): one delta chain across the flattened lists (opid_list.rs:69-72)
      Sc1Col ca, cc;
      open(ca, 10); open(cc, 11);
      const u64* ids = B.pred_ids + pb;
      sc1_walk<Sc1Pred>(pn, lane,
        [&](u32 e, bool valid) {
          Sc1Pred v; v.pa = 0; v.delta = 0;
          if (valid) {
            const u64 id = ids[e];
            v.pa = amap[(u32)id];
            v.delta = (u64)((i64)(id >> 32) - (e ? (i64)(ids[e - 1] >> 32) : 0));
          }
          return v;
        },
        [&](const Sc1Pred& v, const Sc1Pred& f, u32 tb, u32 cnt) {
          if (gmask & 0x200u) { ca.total += (u32)__popcll(__ballot(((v.pa ^ (u32)v.delta) & 1) != 0)); return; }
          sc1_num<false, u32>(ca, v.pa, false, sc1_last(f.pa), false, tb, cnt, lane);
          sc1_num<true, u64>(cc, v.delta, false, sc1_last(f.delta), false, tb, cnt, lane);
        });
      close(ca, 10, false); close(cc, 11, false);
    }
  }
}

// The columns from the tails of their slots to their places in the chunk (behind k_chg_layout; columns flagged in c_wide
// are written by g_encode_one<1> instead).  One wave per change, four bytes per lane and step.
__global__ void __launch_bounds__(64 * SD_WAVES) g_encode_place(BatchDev B) {
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 c = blockIdx.x * SD_WAVES + wave;
  if (c >= B.tot_chg) return;
  const u32 lane = threadIdx.x & 63;
  const u32 d = B.c_doc[c];
  if (B.status[d] != 0 || (B.sv_big && B.sv_big[c])) return;
  const u32 n = B.c_nops[c], base = B.c_op_base[c];
  if (n == 0) return;
  const u32 pb = B.j_pred_base[base], pn = B.j_pred_base[base + n] - pb;
  const u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
  const u32 flags = B.c_wide[c];
  u8* dst = B.out_arena + (u64)B.c_out_off[c] + B.c_cols_off[c];
  struct __attribute__((packed)) P4 { u32 v; };
#pragma unroll
  for (u32 k = 0; k < AMH_NCOLS; k++) {
    const u32 len = cl[k];
    if (len && !((flags >> k) & 1)) {
      u32 cap;
      const u8* src = sc1_slot(B, c, k, base, n, pb, pn, cap) + cap - len;
      const u32 words = len / 4;
      for (u32 i = lane; i < words; i += 64) ((P4*)dst)[i].v = ((const P4*)src)[i].v;
      const u32 done = words * 4;
      if (lane < len - done) dst[done + lane] = src[done + lane];
    }
    dst += len;
  }
}
