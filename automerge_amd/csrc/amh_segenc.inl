// Segment-parallel column ENCODE (device only; included by amh_kernels.hip): the reference's run-length encoders
// (rle.rs:31-133, delta.rs:10-43, boolean.rs:11-51; value.rs:320-369 for the raw column) as wave-wide bit arithmetic
// instead of one lane stepping a state machine.
//
// One wave encodes one column of one change (or document), 64 values per step, and walks the column BACKWARDS.
// The canonical encoding is a function of run structure alone ("parallel form"):
//   * value i starts a run iff it differs from value i - 1 (nulls compare equal to nulls), ends one iff it differs from
//     value i + 1;
//   * a non-null run of length 1 is a "lone" value; maximal stretches of lone values are literal runs: sleb(-k), then
//     the k values;
//   * a longer run is sleb(length), value; a null run is 0, uleb(length); a column of nothing but nulls is empty.
// What a run start must write depends on where the run ENDS (its length) — which lies ahead of it.  Walking backwards,
// everything ahead of a tile is already known and is carried in four scalars (does the next value start a run / is it
// lone, how long are the run and the literal stretch that begin there), so no header is ever reserved and patched,
// and a literal run of any length costs nothing extra.  Inside a tile the run structure is three ballots (equal to
// the previous value, null, valid) and shifts of them; every lane finds the end of its run with a count of trailing
// zeros.  Each lane then knows the up to four numbers it has to write: [header of a literal run that starts the
// column] [its own run header] [its value / null count] [header of the literal run that starts behind it]; an
// exclusive scan of their sizes places them.  The size pass runs the same walk without the stores and gives the
// column length from which the store pass counts down.
#pragma once

struct ScCarry {       // about the first value behind the tile (all uniform)
  bool start0;         // it starts a run (true when there is none)
  bool lone0;          // it is a lone value
  u32 runlen0;         // length of its run counted from it
  u32 litlen0;         // lone values from it on
};
struct ScRoles {
  bool is_start, is_lone, is_lh, emit_p;
  u32 rl;              // length of the run that starts here
  u32 k_lh;            // is_lh: length of the literal run that starts behind this value
  u32 k_p;             // emit_p: length of the literal run that starts the column
};
__device__ __forceinline__ u32 sc_ctz64(u64 x) { return x ? (u32)__builtin_ctzll(x) : 64u; }

// EQP bit i: value base + i equals value base + i - 1 (bit 0 looks at the value in front of the tile; clear at the
// start of the column).  NUL bit i: value base + i is null.  cnt = values in the tile.  Updates the carry.
__device__ __forceinline__ ScRoles sc_roles(u64 EQP, u64 NUL, u32 cnt, u32 base, u32 lane, ScCarry& c) {
  const u64 VALID = sd_low_bits(cnt), top = (u64)1 << (cnt - 1);
  const u64 START = ~EQP & VALID;
  const u64 END = ((START >> 1) | (c.start0 ? top : 0)) & VALID;
  const u64 LONE = START & END & ~NUL;
  const u64 LONEN = ((LONE >> 1) | (c.lone0 ? top : 0)) & VALID;
  const u64 LH = ~LONE & LONEN & VALID;
  ScRoles r;
  r.is_start = (START >> lane) & 1; r.is_lone = (LONE >> lane) & 1; r.is_lh = (LH >> lane) & 1;
  // the run that starts here ends at the next END bit, or behind the tile
  { const u64 e = END >> lane; r.rl = e ? (u32)__builtin_ctzll(e) + 1 : (cnt - lane) + c.runlen0; }
  // lone values from lane + 1 on
  {
    const u32 room = cnt - (lane < cnt ? lane + 1 : cnt);
    const u32 z = lane + 1 < 64 ? sc_ctz64(~LONE >> (lane + 1)) : 0;
    r.k_lh = z >= room ? room + (c.lone0 ? c.litlen0 : 0) : z;
  }
  // the carry for the tile in front of this one
  ScCarry n;
  n.start0 = START & 1;
  n.lone0 = LONE & 1;
  n.runlen0 = END ? (u32)__builtin_ctzll(END) + 1 : cnt + c.runlen0;
  { const u32 z = sc_ctz64(~LONE); n.litlen0 = (LONE & 1) ? (z >= cnt ? cnt + (c.lone0 ? c.litlen0 : 0) : z) : 0; }
  c = n;
  r.emit_p = base == 0 && lane == 0 && n.lone0;
  r.k_p = n.litlen0;
  return r;
}
// the carry update of sc_roles alone
__device__ __forceinline__ void sc_carry_step(u64 EQP, u64 NUL, u32 cnt, ScCarry& c) {
  const u64 VALID = sd_low_bits(cnt), top = (u64)1 << (cnt - 1);
  const u64 START = ~EQP & VALID;
  const u64 END = ((START >> 1) | (c.start0 ? top : 0)) & VALID;
  const u64 LONE = START & END & ~NUL;
  ScCarry n;
  n.start0 = START & 1;
  n.lone0 = LONE & 1;
  n.runlen0 = END ? (u32)__builtin_ctzll(END) + 1 : cnt + c.runlen0;
  { const u32 z = sc_ctz64(~LONE); n.litlen0 = (LONE & 1) ? (z >= cnt ? cnt + (c.lone0 ? c.litlen0 : 0) : z) : 0; }
  c = n;
}
// A column over many waves (documents and changes of tens of thousands of rows: g_sv_big, g_en_big).  The column is cut into ranges of
// rows, a wave each.  All a range needs of the rows behind it is the carry, and what a range does to the carry depends
// on the carry it receives only in three ways: which of the three possible (start0, lone0) pairs it is, and the two
// lengths, which are added where a run or a literal stretch reaches the range's end.  So every range first walks its
// rows once for all three cases with a marker in the lengths (ScSummary), one thread per column then chains the
// ranges from the last to the first, and the size and store walks run with the true carries.
#define SC_MARK 0x80000000u   // lengths stay below it (rows and succ entries of a document < 2^31)
struct ScSummary { ScCarry h[3]; };   // received (start0, lone0) = (1, 1), (1, 0), (0, 0)
__device__ __forceinline__ void sc_summary_init(ScSummary& s) {
  s.h[0].start0 = true;  s.h[0].lone0 = true;  s.h[0].runlen0 = 0;       s.h[0].litlen0 = SC_MARK;
  s.h[1].start0 = true;  s.h[1].lone0 = false; s.h[1].runlen0 = 0;       s.h[1].litlen0 = 0;
  s.h[2].start0 = false; s.h[2].lone0 = false; s.h[2].runlen0 = SC_MARK; s.h[2].litlen0 = 0;
}
__device__ __forceinline__ ScCarry sc_summary_apply(const ScSummary& s, const ScCarry& in) {
  ScCarry o = s.h[in.start0 ? (in.lone0 ? 0 : 1) : 2];
  if (o.runlen0 >= SC_MARK) o.runlen0 = o.runlen0 - SC_MARK + in.runlen0;
  if (o.litlen0 >= SC_MARK) o.litlen0 = o.litlen0 - SC_MARK + in.litlen0;
  return o;
}
struct ScRange { u32 lo, hi; ScCarry carry; u32 total0; };   // rows [lo, hi) (lo a multiple of 64; hi too, or the column's end), the carry from behind, bytes behind the range
__device__ __forceinline__ ScCarry sc_carry_init() { ScCarry c; c.start0 = true; c.lone0 = false; c.runlen0 = 0; c.litlen0 = 0; return c; }

// lane i receives lane i - 1's 64-bit value
__device__ __forceinline__ u64 sd_prev64(u64 v) { return (u64)sd_prev((u32)v) | ((u64)sd_prev((u32)(v >> 32)) << 32); }

// ---- numeric columns ---------------------------------------------------------------------------------------------
// Src: void load(u32 e, u64& v, bool& isnull) const — value e of the column (any lane may ask for any e < n).
// SIGNED: values are written as sleb128 (deltas), else uleb128.  STORE: out = first byte of the column, col_len = its
// length (from the size pass).  Returns the column length.
template <class Src>
__device__ __forceinline__ void sc_num_tile(const Src& src, u32 base, u32 cnt, u32 lane, u64& v, bool& nul, u64& EQP, u64& NUL) {
  const u32 e = base + lane;
  const bool valid = lane < cnt;
  v = 0; nul = true;
  if (valid) src.load(e, v, nul);
  u64 pv = sd_prev64(v); bool pn = sd_prev((u32)nul) != 0;
  if (lane == 0 && base > 0) src.load(base - 1, pv, pn);
  const bool eqp = valid && e > 0 && pn == nul && (nul || pv == v);
  EQP = __ballot(eqp); NUL = __ballot(nul && valid);
}
template <class Src>
__device__ __forceinline__ void sc_rle_summary(const Src& src, u32 lo, u32 hi, u32 lane, ScSummary& S, bool& any) {
  for (u32 base = (hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = hi - base < 64 ? hi - base : 64;
    u64 v, EQP, NUL; bool nul;
    sc_num_tile(src, base, cnt, lane, v, nul, EQP, NUL);
    any = any || (~NUL & sd_low_bits(cnt)) != 0;
    sc_carry_step(EQP, NUL, cnt, S.h[0]); sc_carry_step(EQP, NUL, cnt, S.h[1]); sc_carry_step(EQP, NUL, cnt, S.h[2]);
    if (base == lo) break;
  }
}
template <bool SIGNED, bool STORE, class Src>
__device__ __forceinline__ u32 sc_rle_range(const Src& src, const ScRange& rg, u8* out, u32 col_len, u32 lane, bool& any) {
  ScCarry carry = rg.carry;
  u32 total = rg.total0;
  for (u32 base = (rg.hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = rg.hi - base < 64 ? rg.hi - base : 64;
    u64 v, EQP, NUL; bool nul;
    sc_num_tile(src, base, cnt, lane, v, nul, EQP, NUL);
    any = any || (~NUL & sd_low_bits(cnt)) != 0;
    const ScRoles r = sc_roles(EQP, NUL, cnt, base, lane, carry);
    // the numbers this lane writes, in order: P (literal header at the start of the column), H (run header), V (value
    // or null count), L (header of the literal run that starts behind this value)
    const bool hasP = r.emit_p, hasH = r.is_start && !r.is_lone, hasV = r.is_start, hasL = r.is_lh;
    const i64 nP = -(i64)r.k_p, nH = nul ? 0 : (i64)r.rl, nL = -(i64)r.k_lh;
    const u32 sP = hasP ? sleb_size_fast(nP) : 0, sH = hasH ? sleb_size_fast(nH) : 0, sL = hasL ? sleb_size_fast(nL) : 0;
    const u32 sV = !hasV ? 0 : (nul ? uleb_size_fast(r.rl) : (SIGNED ? sleb_size_fast((i64)v) : uleb_size_fast(v)));
    const u32 len = sP + sH + sV + sL;
    const u32 inc = sd_scan_add(len);
    const u32 T = sd_bcast(inc, 63);
    total += T;
    if (STORE && len) {
      u8* o = out + (col_len - total) + (inc - len);
      if (hasP) o += wr_sleb(o, nP);
      if (hasH) o += wr_sleb(o, nH);
      if (hasV) o += nul ? wr_uleb(o, r.rl) : (SIGNED ? wr_sleb(o, (i64)v) : wr_uleb(o, v));
      if (hasL) wr_sleb(o, nL);
    }
    if (base == rg.lo) break;
  }
  return total;
}
template <bool SIGNED, bool STORE, class Src>
__device__ __forceinline__ u32 sc_rle_column(const Src& src, u32 n, u8* out, u32 col_len, u32 lane) {
  if (n == 0 || (STORE && col_len == 0)) return 0;
  ScRange rg; rg.lo = 0; rg.hi = n; rg.carry = sc_carry_init(); rg.total0 = 0;
  bool any = false;
  const u32 total = sc_rle_range<SIGNED, STORE>(src, rg, out, col_len, lane, any);
  return any ? total : 0;   // a column of nulls is not written at all (rle.rs:33)
}

// ---- boolean columns (boolean.rs:35-50): alternating run lengths, the first one counts `false` ----------------------
// Src: bool load(u32 e) const.  MAYBE: nothing unless some value is true (MaybeBooleanEncoder, boolean.rs:131-137).
template <class Src>
__device__ __forceinline__ void sc_bool_tile(const Src& src, u32 base, u32 cnt, u32 lane, bool& v, u64& EQP, bool& any_true) {
  const u32 e = base + lane;
  const bool valid = lane < cnt;
  v = valid ? src.load(e) : false;
  bool pv = sd_prev((u32)v) != 0;
  if (lane == 0 && base > 0) pv = src.load(base - 1);
  const bool eqp = valid && e > 0 && pv == v;
  EQP = __ballot(eqp);
  any_true = any_true || __ballot(v && valid) != 0;
}
template <class Src>
__device__ __forceinline__ void sc_bool_summary(const Src& src, u32 lo, u32 hi, u32 lane, ScSummary& S, bool& any_true) {
  for (u32 base = (hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = hi - base < 64 ? hi - base : 64;
    bool v; u64 EQP;
    sc_bool_tile(src, base, cnt, lane, v, EQP, any_true);
    sc_carry_step(EQP, 0, cnt, S.h[0]); sc_carry_step(EQP, 0, cnt, S.h[1]); sc_carry_step(EQP, 0, cnt, S.h[2]);
    if (base == lo) break;
  }
}
template <bool STORE, class Src>
__device__ __forceinline__ u32 sc_bool_range(const Src& src, const ScRange& rg, u8* out, u32 col_len, u32 lane, bool& any_true) {
  ScCarry carry = rg.carry;
  u32 total = rg.total0;
  for (u32 base = (rg.hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = rg.hi - base < 64 ? rg.hi - base : 64;
    bool v; u64 EQP;
    sc_bool_tile(src, base, cnt, lane, v, EQP, any_true);
    const ScRoles r = sc_roles(EQP, 0, cnt, base, lane, carry);
    // a column that starts with `true` starts with an empty run of `false`
    const bool hasZ = base == 0 && lane == 0 && v;
    const u32 sV = r.is_start ? uleb_size_fast(r.rl) : 0;
    const u32 len = (hasZ ? 1u : 0u) + sV;
    const u32 inc = sd_scan_add(len);
    total += sd_bcast(inc, 63);
    if (STORE && len) {
      u8* o = out + (col_len - total) + (inc - len);
      if (hasZ) *o++ = 0;
      if (r.is_start) wr_uleb(o, r.rl);
    }
    if (base == rg.lo) break;
  }
  return total;
}
template <bool MAYBE, bool STORE, class Src>
__device__ __forceinline__ u32 sc_bool_column(const Src& src, u32 n, u8* out, u32 col_len, u32 lane) {
  if (n == 0 || (STORE && col_len == 0)) return 0;
  ScRange rg; rg.lo = 0; rg.hi = n; rg.carry = sc_carry_init(); rg.total0 = 0;
  bool any_true = false;
  const u32 total = sc_bool_range<STORE>(src, rg, out, col_len, lane, any_true);
  return (MAYBE && !any_true) ? 0 : total;
}

// ---- string columns ------------------------------------------------------------------------------------------------
// A string value is (length, bytes): up to 8 bytes travel in a register (`inl`), longer ones are a pointer.
struct ScStr { u64 bytes; const u8* p; u32 len; bool inl, isnull; };
__device__ __forceinline__ bool sc_str_eq(const ScStr& a, const ScStr& b) {
  if (a.isnull != b.isnull) return false;
  if (a.isnull) return true;
  if (a.len != b.len) return false;
  if (a.inl && b.inl) return a.bytes == b.bytes;
  if (a.inl || b.inl) {   // one side in a register (<= 8 bytes), the other a pointer to as many bytes
    const u64 x = a.inl ? a.bytes : amh_peek8_any(a.p) & amh_low_bytes(a.len), y = b.inl ? b.bytes : amh_peek8_any(b.p) & amh_low_bytes(b.len);
    return x == y;
  }
  if (a.p == b.p) return true;
  for (u32 i = 0; i < a.len; i += 8) {
    u64 d = amh_peek8_any(a.p + i) ^ amh_peek8_any(b.p + i);
    const u32 rest = a.len - i;
    if (rest < 8) d &= amh_low_bytes(rest);
    if (d) return false;
  }
  return true;
}
// Src: ScStr load(u32 e) const
template <class Src>
__device__ __forceinline__ void sc_str_tile(const Src& src, u32 base, u32 cnt, u32 lane, ScStr& v, u64& EQP, u64& NUL) {
  const u32 e = base + lane;
  const bool valid = lane < cnt;
  v.bytes = 0; v.p = 0; v.len = 0; v.inl = true; v.isnull = true;
  if (valid) v = src.load(e);
  ScStr pv; pv.isnull = true; pv.bytes = 0; pv.p = 0; pv.len = 0; pv.inl = true;
  if (valid && e > 0) pv = src.load(e - 1);   // (a second load of the neighbour: the values are in L1 / L2)
  const bool eqp = valid && e > 0 && sc_str_eq(v, pv);
  EQP = __ballot(eqp); NUL = __ballot(v.isnull && valid);
}
template <class Src>
__device__ __forceinline__ void sc_str_summary(const Src& src, u32 lo, u32 hi, u32 lane, ScSummary& S, bool& any) {
  for (u32 base = (hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = hi - base < 64 ? hi - base : 64;
    ScStr v; u64 EQP, NUL;
    sc_str_tile(src, base, cnt, lane, v, EQP, NUL);
    any = any || (~NUL & sd_low_bits(cnt)) != 0;
    sc_carry_step(EQP, NUL, cnt, S.h[0]); sc_carry_step(EQP, NUL, cnt, S.h[1]); sc_carry_step(EQP, NUL, cnt, S.h[2]);
    if (base == lo) break;
  }
}
template <bool STORE, class Src>
__device__ __forceinline__ u32 sc_str_range(const Src& src, const ScRange& rg, u8* out, u32 col_len, u32 lane, bool& any) {
  ScCarry carry = rg.carry;
  u32 total = rg.total0;
  for (u32 base = (rg.hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = rg.hi - base < 64 ? rg.hi - base : 64;
    ScStr v; u64 EQP, NUL;
    sc_str_tile(src, base, cnt, lane, v, EQP, NUL);
    any = any || (~NUL & sd_low_bits(cnt)) != 0;
    const ScRoles r = sc_roles(EQP, NUL, cnt, base, lane, carry);
    const bool nul = v.isnull;
    const bool hasP = r.emit_p, hasH = r.is_start && !r.is_lone, hasV = r.is_start, hasL = r.is_lh;
    const i64 nP = -(i64)r.k_p, nH = nul ? 0 : (i64)r.rl, nL = -(i64)r.k_lh;
    const u32 sP = hasP ? sleb_size_fast(nP) : 0, sH = hasH ? sleb_size_fast(nH) : 0, sL = hasL ? sleb_size_fast(nL) : 0;
    const u32 sV = !hasV ? 0 : (nul ? uleb_size_fast(r.rl) : uleb_size_fast(v.len) + v.len);
    const u32 len = sP + sH + sV + sL;
    const u32 inc = sd_scan_add(len);
    total += sd_bcast(inc, 63);
    if (STORE && len) {
      u8* o = out + (col_len - total) + (inc - len);
      if (hasP) o += wr_sleb(o, nP);
      if (hasH) o += wr_sleb(o, nH);
      if (hasV) {
        if (nul) o += wr_uleb(o, r.rl);
        else {
          o += wr_uleb(o, v.len);
          if (v.inl) for (u32 i = 0; i < v.len; i++) o[i] = (u8)(v.bytes >> (8 * i));
          else for (u32 i = 0; i < v.len; i++) o[i] = v.p[i];
          o += v.len;
        }
      }
      if (hasL) wr_sleb(o, nL);
    }
    if (base == rg.lo) break;
  }
  return total;
}
template <bool STORE, class Src>
__device__ __forceinline__ u32 sc_str_column(const Src& src, u32 n, u8* out, u32 col_len, u32 lane) {
  if (n == 0 || (STORE && col_len == 0)) return 0;
  ScRange rg; rg.lo = 0; rg.hi = n; rg.carry = sc_carry_init(); rg.total0 = 0;
  bool any = false;
  const u32 total = sc_str_range<STORE>(src, rg, out, col_len, lane, any);
  return any ? total : 0;
}

// ---- raw value bytes (value.rs:320-369): the payloads back to back ---------------------------------------------------
// kind 0 nothing, 1 bytes at p, 2 an integer written as sleb128, 3 up to 8 bytes in a register
struct ScRaw { u32 kind, len; u64 x; const u8* p; };
template <bool STORE, class Src>
__device__ __forceinline__ u32 sc_raw_range(const Src& src, const ScRange& rg, u8* out, u32 col_len, u32 lane) {
  u32 total = rg.total0;
  for (u32 base = (rg.hi - 1) & ~63u;; base -= 64) {
    const u32 cnt = rg.hi - base < 64 ? rg.hi - base : 64;
    ScRaw v; v.kind = 0; v.len = 0; v.x = 0; v.p = 0;
    if (lane < cnt) v = src.load(base + lane);
    const u32 len = v.kind == 0 ? 0 : (v.kind == 2 ? sleb_size_fast((i64)v.x) : v.len);
    const u32 inc = sd_scan_add(len);
    total += sd_bcast(inc, 63);
    if (STORE && len) {
      u8* o = out + (col_len - total) + (inc - len);
      if (v.kind == 2) wr_sleb(o, (i64)v.x);
      else if (v.kind == 3) for (u32 i = 0; i < len; i++) o[i] = (u8)(v.x >> (8 * i));
      else for (u32 i = 0; i < len; i++) o[i] = v.p[i];
    }
    if (base == rg.lo) break;
  }
  return total;
}
template <bool STORE, class Src>
__device__ __forceinline__ u32 sc_raw_column(const Src& src, u32 n, u8* out, u32 col_len, u32 lane) {
  if (n == 0 || (STORE && col_len == 0)) return 0;
  ScRange rg; rg.lo = 0; rg.hi = n; rg.carry = sc_carry_init(); rg.total0 = 0;
  return sc_raw_range<STORE>(src, rg, out, col_len, lane);
}

// ------------------------------------------------------------------------------------------
// Change re-encode (change_op_columns.rs:146-329): one wave per (change, column group), the group's columns one after
// the other out of the change view (k_change_view).  STORE = false: column lengths -> c_col_len; STORE = true: bytes
// -> out_arena at the offsets k_chg_layout derived from those lengths.
// Groups: 0 object id (columns 0, 1)  1 key (2, 3, 4)  2 insert / action / predNum / expand (5, 6, 9, 12)
//         3 value (7, 8)  4 pred lists (10, 11)  5 mark names (13)
// ------------------------------------------------------------------------------------------
// The change view is read through a one-value cache per lane: the wave fetches the tile it starts with (the last one)
// for all of its columns at once, before the first column is encoded, so the columns of a small change do not each
// wait for memory in turn.
template <class T> struct ScCached {
  const T* g; u32 ce; T cx;
  __device__ __forceinline__ void init(const T* p, u32 e, bool valid) { g = p; ce = 0xFFFFFFFFu; if (valid) { cx = p[e]; ce = e; } }
  __device__ __forceinline__ T get(u32 e) const { T r = cx; if (e != ce) r = g[e]; return r; }
};
struct ScObjSrc {   // cv[0]: {object actor (change-local) or NULL32, object counter (0 = root)}
  ScCached<uint2> g; bool ctr;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { uint2 x = g.get(e); isnull = x.y == 0; v = ctr ? x.y : x.x; }
};
// cv[2]: {key actor, key counter, string offset, string length | flags}
__device__ __forceinline__ bool sc_key_inl(const uint4& x) { return x.w != AMH_NULL32 && (x.w & AMH_CV_INLINE); }
struct ScKeyActorSrc {
  ScCached<uint4> g;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { uint4 x = g.get(e); isnull = sc_key_inl(x) || x.x == AMH_NULL32; v = x.x; }
};
struct ScKeyCtrSrc {   // delta column: the value is the difference to the last non-null counter in front (delta.rs:23-27)
  ScCached<uint4> g;
  __device__ __forceinline__ bool ctr(u32 e, u32& c) const { uint4 x = g.get(e); c = x.y; return !(sc_key_inl(x) || x.y == AMH_NULL32); }
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    u32 c, pc = 0;
    isnull = !ctr(e, c);
    v = 0;
    if (isnull) return;
    for (u32 j = e; j > 0; j--) { u32 t; if (ctr(j - 1, t)) { pc = t; break; } }
    v = (u64)((i64)c - (i64)pc);
  }
};
struct ScKeyStrSrc {
  ScCached<uint4> g; const u8* chunk;
  __device__ __forceinline__ ScStr load(u32 e) const {
    uint4 x = g.get(e);
    ScStr s; s.isnull = x.w == AMH_NULL32; s.inl = sc_key_inl(x);
    s.bytes = s.inl ? ((u64)x.x | ((u64)x.y << 32)) : 0;
    s.len = s.isnull ? 0 : (s.inl ? (x.w & 0xFF) : x.w);
    s.p = chunk + x.z;
    return s;
  }
};
// cv[6]: {insert | expand << 1 | action << 8, pred count}
struct ScInsertSrc { ScCached<uint2> g; __device__ __forceinline__ bool load(u32 e) const { return (g.get(e).x & 1) != 0; } };
struct ScExpandSrc { ScCached<uint2> g; __device__ __forceinline__ bool load(u32 e) const { return (g.get(e).x & 2) != 0; } };
struct ScActionSrc { ScCached<uint2> g; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = g.get(e).x >> 8; isnull = false; } };
struct ScPredNumSrc { ScCached<uint2> g; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = g.get(e).y; isnull = false; } };
// cv[8]: {meta lo, meta hi | kind << 8 | bytes << 16, payload or reference}
struct ScValMetaSrc {
  ScCached<uint4> g;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { uint4 x = g.get(e); v = (u64)x.x | ((u64)(x.y & 0xFF) << 32); isnull = false; }
};
struct ScValRawSrc {
  ScCached<uint4> g; const u8* chunk;
  __device__ __forceinline__ ScRaw load(u32 e) const {
    uint4 x = g.get(e);
    ScRaw r; r.kind = (x.y >> 8) & 0xFF; r.x = (u64)x.z | ((u64)x.w << 32); r.p = chunk + x.z;
    r.len = r.kind == 3 ? ((x.y >> 16) & 0xF) : (r.kind == 1 ? x.w : 0);
    return r;
  }
};
struct ScPredSrc { const u64* ids; const u32* amap; };   // the change's flattened pred lists (opid_list.rs:52-78)
struct ScPredActorSrc : ScPredSrc { __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = amap[(u32)ids[e]]; isnull = false; } };
struct ScPredCtrSrc : ScPredSrc {
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    const i64 c = (i64)(ids[e] >> 32), pc = e ? (i64)(ids[e - 1] >> 32) : 0;
    v = (u64)(c - pc); isnull = false;
  }
};
struct ScMarkSrc {
  const BatchDev& B; const u32* ops; const u8* chunk;
  __device__ __forceinline__ ScStr load(u32 e) const {
    const u32 t = ops[e];
    ScStr s; s.bytes = 0; s.inl = false; s.p = chunk; s.len = 0;
    s.isnull = !(!op_is_del(B, t) && B.r_action[t] == 7 && B.r_mark_len[t] != AMH_NULL32);
    if (!s.isnull) { s.p = chunk + B.r_mark_off[t]; s.len = B.r_mark_len[t]; }
    return s;
  }
};

// The change re-encode over these sources (a wave per (change, column group) walking the change view twice, size then
// store) was round 2's g_encode_seg; round 3 replaced it by the one-pass walk over the op rows (amh_segenc1.inl).  The
// sources stay: g_en_big (changes cut into ranges) reads the change view through them.
#define SC_GROUPS 5

// ------------------------------------------------------------------------------------------
// Document save (doc_op_columns.rs:104-311; device form of k_save_size / k_save_store for the op columns 9..24): one
// wave per (document, column group), the values read straight from the row arrays in document order.  The nine
// change-metadata columns (a handful of rows per document) stay with g_save_lds group 0.
// Groups: 0 object id (9, 10)  1 key (11, 12, 13)  2 op id (14, 15)  3 insert / action / succ count / expand (16, 17,
//         20, 23)  4 value (18, 19)  5 succ lists (21, 22)  6 mark names (24)
// ------------------------------------------------------------------------------------------
struct SvCtx { const BatchDev& B; u32 rb, ab; const u8* chunk; };
struct SvObjSrc {
  const SvCtx& c; bool ctr;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    const u32 r = c.rb + e, oc = c.B.r_obj_ctr[r];
    isnull = oc == 0;
    v = ctr ? oc : (isnull ? 0 : c.B.actor_rank[c.ab + c.B.r_obj_actor[r]]);
  }
};
struct SvKeyActorSrc {
  const SvCtx& c;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    const u32 r = c.rb + e;
    const bool prop = c.B.r_key_len[r] != AMH_NULL32;
    isnull = prop || c.B.r_key_ctr[r] == 0;
    v = isnull ? 0 : c.B.actor_rank[c.ab + c.B.r_key_actor[r]];
  }
};
struct SvKeyCtrSrc {   // delta over the non-null counters (HEAD = 0)
  const SvCtx& c;
  __device__ __forceinline__ bool ctr(u32 e, u32& k) const { const u32 r = c.rb + e; k = c.B.r_key_ctr[r]; return c.B.r_key_len[r] == AMH_NULL32; }
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    u32 k, pk = 0;
    isnull = !ctr(e, k);
    v = 0;
    if (isnull) return;
    for (u32 j = e; j > 0; j--) { u32 t; if (ctr(j - 1, t)) { pk = t; break; } }
    v = (u64)((i64)k - (i64)pk);
  }
};
struct SvKeyStrSrc {
  const SvCtx& c;
  __device__ __forceinline__ ScStr load(u32 e) const {
    const u32 r = c.rb + e, kl = c.B.r_key_len[r];
    ScStr s; s.bytes = 0; s.inl = false; s.isnull = kl == AMH_NULL32; s.len = s.isnull ? 0 : kl;
    s.p = c.chunk + (s.isnull ? 0 : c.B.r_key_off[r]);
    return s;
  }
};
struct SvIdActorSrc { const SvCtx& c; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = c.B.actor_rank[c.ab + c.B.r_id_actor[c.rb + e]]; isnull = false; } };
struct SvIdCtrSrc {
  const SvCtx& c;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    const i64 k = c.B.r_id_ctr[c.rb + e], pk = e ? (i64)c.B.r_id_ctr[c.rb + e - 1] : 0;
    v = (u64)(k - pk); isnull = false;
  }
};
struct SvInsertSrc { const SvCtx& c; __device__ __forceinline__ bool load(u32 e) const { return c.B.r_insert[c.rb + e] != 0; } };
struct SvExpandSrc { const SvCtx& c; __device__ __forceinline__ bool load(u32 e) const { const u32 r = c.rb + e; return c.B.r_action[r] == 7 && c.B.r_expand[r]; } };
struct SvActionSrc { const SvCtx& c; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = c.B.r_action[c.rb + e]; isnull = false; } };
struct SvSuccNumSrc { const SvCtx& c; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = c.B.r_succ_cnt[c.rb + e]; isnull = false; } };
struct SvValMetaSrc { const SvCtx& c; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = op_val(c.B, c.rb + e).meta; isnull = false; } };
struct SvValRawSrc {
  const SvCtx& c;
  __device__ __forceinline__ ScRaw load(u32 e) const {
    const OpVal v = op_val(c.B, c.rb + e);
    ScRaw r; r.kind = (u32)v.kind; r.len = v.raw_len; r.x = (u64)v.i; r.p = v.raw;
    if (r.kind == 1 && r.len == 0) r.kind = 0;
    return r;
  }
};
struct SvSuccActorSrc { const SvCtx& c; u32 sb; __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { v = c.B.actor_rank[c.ab + c.B.s_actor[sb + e]]; isnull = false; } };
struct SvSuccCtrSrc {
  const SvCtx& c; u32 sb;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    const i64 k = c.B.s_ctr[sb + e], pk = e ? (i64)c.B.s_ctr[sb + e - 1] : 0;
    v = (u64)(k - pk); isnull = false;
  }
};
struct SvMarkSrc {
  const SvCtx& c;
  __device__ __forceinline__ ScStr load(u32 e) const {
    const u32 r = c.rb + e;
    ScStr s; s.bytes = 0; s.inl = false; s.p = c.chunk; s.len = 0;
    s.isnull = !(c.B.r_action[r] == 7 && c.B.r_mark_len[r] != AMH_NULL32);
    if (!s.isnull) { s.p = c.chunk + c.B.r_mark_off[r]; s.len = c.B.r_mark_len[r]; }
    return s;
  }
};

// change-metadata columns 0..8 (doc_change_columns.rs:87-162; the values save_encode_chg_col feeds its encoders), for
// documents with thousands of changes (g_sv_big)
struct SvChgCtx { const BatchDev& B; u32 cb, db, nc; const u32* rank; const u8* chunk; };
struct SvChgNumSrc {   // which: 0 author (actor rank), 5 number of deps, 7 extra-bytes meta
  const SvChgCtx& c; u32 which;
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const {
    isnull = false;
    v = which == 0 ? c.rank[c.B.c_actor[c.cb + e]] : (which == 5 ? c.B.c_ndeps[c.cb + e] : (((u64)c.B.c_extra_len[c.cb + e] << 4) | 7));
  }
};
struct SvChgDeltaSrc {   // which: 1 seq, 2 max op, 3 time
  const SvChgCtx& c; u32 which;
  __device__ __forceinline__ i64 at(u32 e) const { return which == 1 ? (i64)c.B.c_seq[c.cb + e] : (which == 2 ? (i64)c.B.c_max_op[c.cb + e] : c.B.c_time[c.cb + e]); }
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { isnull = false; v = (u64)(at(e) - (e ? at(e - 1) : 0)); }
};
struct SvChgMsgSrc {
  const SvChgCtx& c;
  __device__ __forceinline__ ScStr load(u32 e) const {
    const u32 ml = c.B.c_msg_len[c.cb + e];
    ScStr s; s.bytes = 0; s.inl = false; s.isnull = ml == AMH_NULL32 || ml == 0; s.len = s.isnull ? 0 : ml;
    s.p = c.chunk + (s.isnull ? 0 : c.B.c_msg_off[c.cb + e]);
    return s;
  }
};
struct SvChgDepSrc {   // the flattened dependency indexes, every change's in the order of the dependencies' hashes
  const SvChgCtx& c;
  __device__ __forceinline__ i64 at(u32 e) const {
    u32 lo = 0, hi = c.nc;   // the last change whose entries start at or in front of e (changes without deps share a start: the last one owns it)
    while (hi - lo > 1) { const u32 mid = lo + (hi - lo) / 2; if (c.B.c_dep_base[c.cb + mid] - c.db <= e) lo = mid; else hi = mid; }
    return (i64)sv_dep_sorted(c.B, c.cb + lo, c.cb, e - (c.B.c_dep_base[c.cb + lo] - c.db));
  }
  __device__ __forceinline__ void load(u32 e, u64& v, bool& isnull) const { isnull = false; v = (u64)(at(e) - (e ? at(e - 1) : 0)); }
};
struct SvChgExtraSrc {
  const SvChgCtx& c;
  __device__ __forceinline__ ScRaw load(u32 e) const {
    ScRaw r; r.len = c.B.c_extra_len[c.cb + e]; r.kind = r.len ? 1 : 0; r.x = 0; r.p = c.chunk + c.B.c_extra_off[c.cb + e];
    return r;
  }
};

#define SV_GROUPS 7
template <bool STORE>
__global__ void __launch_bounds__(64 * SD_WAVES) g_save_seg(BatchDev B) {
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 d = blockIdx.x * SD_WAVES + wave;
  if (d >= B.n_docs) return;
  const u32 lane = threadIdx.x & 63;
  const bool big = B.sv_big && (B.sv_big[d] & 1);   // its op columns are cut into ranges (g_sv_big); lengths 0 here, set by g_sv_big_total
  const bool live = B.status[d] == 0 && B.sv_status[d] == 0 && !big;
  u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
  if (STORE && !live) return;
  const u32 n = live ? B.cnt[d].n_rows : 0, ns = live ? B.cnt[d].n_succ : 0;
  SvCtx c{B, B.row_base[d], B.actor_base[d], B.arena + B.desc[d].chunk_off};
  u32 clen[AMH_SV_NCOLS], coff[AMH_SV_NCOLS];
  u8* cols = (u8*)0;
  if (STORE) {
    cols = B.sv_out + (u64)B.sv_off[d] + sv_data_start(B, d);
    u32 o = 0;
#pragma unroll
    for (int k = 0; k < AMH_SV_NCOLS; k++) { clen[k] = cl[k]; coff[k] = o; o += clen[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < AMH_SV_NCOLS; k++) { clen[k] = 0; coff[k] = 0; }
  }
  auto out = [&](int k) -> u8* { return cols + coff[k]; };
  auto put = [&](int k, u32 v) { if (!STORE && lane == 0) cl[k] = v; };
  const u32 G = blockIdx.y;
  if (G == 0) {
    SvObjSrc sa{c, false}, sc{c, true};
    u32 la = sc_rle_column<false, STORE>(sa, n, out(9), clen[9], lane);
    u32 lc = sc_rle_column<false, STORE>(sc, n, out(10), clen[10], lane);
    put(9, la); put(10, lc);
  } else if (G == 1) {
    SvKeyActorSrc sa{c}; SvKeyCtrSrc sk{c}; SvKeyStrSrc ss{c};
    u32 la = sc_rle_column<false, STORE>(sa, n, out(11), clen[11], lane);
    u32 lk = sc_rle_column<true, STORE>(sk, n, out(12), clen[12], lane);
    u32 ls = sc_str_column<STORE>(ss, n, out(13), clen[13], lane);
    put(11, la); put(12, lk); put(13, ls);
  } else if (G == 2) {
    SvIdActorSrc sa{c}; SvIdCtrSrc sk{c};
    u32 la = sc_rle_column<false, STORE>(sa, n, out(14), clen[14], lane);
    u32 lk = sc_rle_column<true, STORE>(sk, n, out(15), clen[15], lane);
    put(14, la); put(15, lk);
  } else if (G == 3) {
    SvInsertSrc si{c}; SvActionSrc sa{c}; SvSuccNumSrc sn{c}; SvExpandSrc se{c};
    u32 li = sc_bool_column<false, STORE>(si, n, out(16), clen[16], lane);
    u32 la = sc_rle_column<false, STORE>(sa, n, out(17), clen[17], lane);
    u32 ln = sc_rle_column<false, STORE>(sn, n, out(20), clen[20], lane);
    u32 le = sc_bool_column<true, STORE>(se, n, out(23), clen[23], lane);
    put(16, li); put(17, la); put(20, ln); put(23, le);
  } else if (G == 4) {
    SvValMetaSrc sm{c}; SvValRawSrc sr{c};
    u32 lm = sc_rle_column<false, STORE>(sm, n, out(18), clen[18], lane);
    u32 lr = sc_raw_column<STORE>(sr, n, out(19), clen[19], lane);
    put(18, lm); put(19, lr);
  } else if (G == 5) {
    SvSuccActorSrc sa{c, B.succ_base[d]}; SvSuccCtrSrc sk{c, B.succ_base[d]};
    u32 la = sc_rle_column<false, STORE>(sa, ns, out(21), clen[21], lane);
    u32 lk = sc_rle_column<true, STORE>(sk, ns, out(22), clen[22], lane);
    put(21, la); put(22, lk);
  } else {
    u32 lm = 0;
    if (live && B.desc[d].ocol[OC_MARK_NAME].len != 0) {
      SvMarkSrc sm{c};
      lm = sc_str_column<STORE>(sm, n, out(24), clen[24], lane);
    }
    put(24, lm);
  }
}

// ------------------------------------------------------------------------------------------
// Document save, a column over many waves (see ScSummary): one wave per job = (big column, range of rows).
// MODE 0: what the range does to the carry -> svj_sum.  g_sv_big_chain: the carry every range receives -> svj_cin.
// MODE 1: bytes of the range -> svj_total.  g_sv_big_total: column lengths -> sv_col_len, bytes behind every range
// -> svj_tail.  MODE 2 (after k_save_layout): the bytes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sv_carry_pack(const ScCarry& c, u32* w) { w[0] = (c.start0 ? 1u : 0u) | (c.lone0 ? 2u : 0u); w[1] = c.runlen0; w[2] = c.litlen0; }
__device__ __forceinline__ ScCarry sv_carry_unpack(const u32* w) { ScCarry c; c.start0 = (w[0] & 1) != 0; c.lone0 = (w[0] & 2) != 0; c.runlen0 = w[1]; c.litlen0 = w[2]; return c; }
template <int MODE>
__global__ void __launch_bounds__(64 * SD_WAVES) g_sv_big(BatchDev B) {
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 j = blockIdx.x * SD_WAVES + wave;
  if (j >= B.sv_nj) return;
  const u32 lane = threadIdx.x & 63;
  const u32 bc = B.svj_bc[j], k = B.svj_k[j];
  const u32 d = B.svb_doc[bc], col = B.svb_col[bc];
  const bool live = B.status[d] == 0 && B.sv_status[d] == 0;
  const u32 n = live ? B.cnt[d].n_rows : 0, ns = live ? B.cnt[d].n_succ : 0;
  const u32 nchg = live ? B.cnt[d].n_chg : 0, ndeps = live ? B.cnt[d].n_deps : 0;
  u32 ncol = col < AMH_SV_CHG_COLS ? (col == 6 ? ndeps : nchg) : ((col == 21 || col == 22) ? ns : n);
  if (col == 24 && B.desc[d].ocol[OC_MARK_NAME].len == 0) ncol = 0;   // only documents with a mark column
  ScSummary S;
  sc_summary_init(S);
  bool any = false;
  u32 total = 0;
  ScRange rg;
  rg.lo = k * B.sv_range;
  rg.hi = ncol - rg.lo < B.sv_range ? ncol : rg.lo + B.sv_range;
  const bool work = rg.lo < ncol;
  u8* out = (u8*)0;
  u32 col_len = 0;
  if (MODE != 0) { rg.carry = sv_carry_unpack(B.svj_cin + 3 * (u64)j); rg.total0 = 0; }
  if (MODE == 2) {
    if (!work) return;
    const u32* cl = B.sv_col_len + (u64)d * AMH_SV_NCOLS;
    col_len = cl[col];
    if (col_len == 0) return;
    u64 o = (u64)B.sv_off[d] + sv_data_start(B, d);
    for (u32 q = 0; q < col; q++) o += cl[q];
    out = B.sv_out + o;
    rg.total0 = B.svj_tail[j];
  }
  if (work) {
    SvCtx c{B, B.row_base[d], B.actor_base[d], B.arena + B.desc[d].chunk_off};
    SvChgCtx cc{B, B.chg_base[d], B.dep_base[d], nchg, B.actor_rank + B.actor_base[d], B.arena + B.desc[d].chunk_off};
#define SV_NUM(SIGNED, src) { if constexpr (MODE == 0) sc_rle_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_rle_range<SIGNED, false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_rle_range<SIGNED, true>(src, rg, out, col_len, lane, any); }
#define SV_BOOL(src) { if constexpr (MODE == 0) sc_bool_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_bool_range<false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_bool_range<true>(src, rg, out, col_len, lane, any); }
#define SV_STR(src) { if constexpr (MODE == 0) sc_str_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_str_range<false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_str_range<true>(src, rg, out, col_len, lane, any); }
    switch (col) {
      case 0: case 5: case 7: { SvChgNumSrc s{cc, col}; SV_NUM(false, s) } break;
      case 1: case 2: case 3: { SvChgDeltaSrc s{cc, col}; SV_NUM(true, s) } break;
      case 4:  { SvChgMsgSrc s{cc}; SV_STR(s) } break;
      case 6:  { SvChgDepSrc s{cc}; SV_NUM(true, s) } break;
      case 8:  { SvChgExtraSrc s{cc};
                 if constexpr (MODE == 1) total = sc_raw_range<false>(s, rg, (u8*)0, 0, lane);
                 else if constexpr (MODE == 2) sc_raw_range<true>(s, rg, out, col_len, lane); } break;
      case 9:  { SvObjSrc s{c, false}; SV_NUM(false, s) } break;
      case 10: { SvObjSrc s{c, true}; SV_NUM(false, s) } break;
      case 11: { SvKeyActorSrc s{c}; SV_NUM(false, s) } break;
      case 12: { SvKeyCtrSrc s{c}; SV_NUM(true, s) } break;
      case 13: { SvKeyStrSrc s{c}; SV_STR(s) } break;
      case 14: { SvIdActorSrc s{c}; SV_NUM(false, s) } break;
      case 15: { SvIdCtrSrc s{c}; SV_NUM(true, s) } break;
      case 16: { SvInsertSrc s{c}; SV_BOOL(s) } break;
      case 17: { SvActionSrc s{c}; SV_NUM(false, s) } break;
      case 18: { SvValMetaSrc s{c}; SV_NUM(false, s) } break;
      case 19: { SvValRawSrc s{c};
                 if constexpr (MODE == 1) total = sc_raw_range<false>(s, rg, (u8*)0, 0, lane);
                 else if constexpr (MODE == 2) sc_raw_range<true>(s, rg, out, col_len, lane); } break;
      case 20: { SvSuccNumSrc s{c}; SV_NUM(false, s) } break;
      case 21: { SvSuccActorSrc s{c, B.succ_base[d]}; SV_NUM(false, s) } break;
      case 22: { SvSuccCtrSrc s{c, B.succ_base[d]}; SV_NUM(true, s) } break;
      case 23: { SvExpandSrc s{c}; SV_BOOL(s) } break;
      default: { SvMarkSrc s{c}; SV_STR(s) } break;
    }
#undef SV_NUM
#undef SV_BOOL
#undef SV_STR
  }
  if (lane == 0) {
    if (MODE == 0) { u32* w = B.svj_sum + 9 * (u64)j; sv_carry_pack(S.h[0], w); sv_carry_pack(S.h[1], w + 3); sv_carry_pack(S.h[2], w + 6); }
    if (MODE == 1) { B.svj_total[j] = total; B.svj_any[j] = any ? 1u : 0u; }
  }
}
// thread per big column: the ranges from the last to the first
__global__ void __launch_bounds__(AMH_BLOCK) g_sv_big_chain(BatchDev B) {
  const u32 bc = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (bc >= B.sv_nbc) return;
  const u32 first = B.svb_first[bc], nr = B.svb_nr[bc];
  ScCarry cur = sc_carry_init();
  for (u32 k = nr; k-- > 0;) {
    const u64 j = (u64)first + k;
    sv_carry_pack(cur, B.svj_cin + 3 * j);
    ScSummary S;
    const u32* w = B.svj_sum + 9 * j;
    S.h[0] = sv_carry_unpack(w); S.h[1] = sv_carry_unpack(w + 3); S.h[2] = sv_carry_unpack(w + 6);
    cur = sc_summary_apply(S, cur);
  }
}
template <bool SAVE>
__global__ void __launch_bounds__(AMH_BLOCK) g_sv_big_total(BatchDev B) {
  const u32 bc = blockIdx.x * AMH_BLOCK + threadIdx.x;
  if (bc >= B.sv_nbc) return;
  const u32 first = B.svb_first[bc], nr = B.svb_nr[bc], d = B.svb_doc[bc], col = B.svb_col[bc];   // d: document (save) / change (re-encode)
  u32 run = 0, any = 0;
  for (u32 k = nr; k-- > 0;) {
    const u64 j = (u64)first + k;
    B.svj_tail[j] = run;
    run += B.svj_total[j];
    any |= B.svj_any[j];
  }
  // a column of nothing but nulls (rle.rs:33) / a MaybeBoolean column without a true (boolean.rs:131-137) is not written
  if (SAVE) {
    const bool always = col == 16 || col == 19 || col == 8;   // insert, raw values, extra bytes
    B.sv_col_len[(u64)d * AMH_SV_NCOLS + col] = (always || any) ? run : 0;
  } else {
    const bool always = col == 5 || col == 8;
    B.c_col_len[(u64)d * AMH_NCOLS + col] = (always || any) ? run : 0;
  }
}
// The change re-encode over many waves: g_sv_big with the columns of a change (the unit of a big column is a change;
// sources as in g_encode_seg).  Modes as above.
template <int MODE>
__global__ void __launch_bounds__(64 * SD_WAVES) g_en_big(BatchDev B) {
  const u32 wave = sd_uniform(threadIdx.x >> 6);
  const u32 j = blockIdx.x * SD_WAVES + wave;
  if (j >= B.sv_nj) return;
  const u32 lane = threadIdx.x & 63;
  const u32 bc = B.svj_bc[j], k = B.svj_k[j];
  const u32 c = B.svb_doc[bc], col = B.svb_col[bc];
  const u32 d = B.c_doc[c];
  const bool live = B.status[d] == 0;
  const u32 n = live ? B.c_nops[c] : 0;
  const u32 base = B.c_op_base[c];
  u32 pb = 0, pn = 0;
  if (n) { pb = B.j_pred_base[base]; pn = B.j_pred_base[base + n] - pb; }
  u32 ncol = (col == 10 || col == 11) ? pn : n;
  if (col == 13 && B.desc[d].ocol[OC_MARK_NAME].len == 0) ncol = 0;
  ScSummary S;
  sc_summary_init(S);
  bool any = false;
  u32 total = 0;
  ScRange rg;
  rg.lo = k * B.sv_range;
  rg.hi = ncol - rg.lo < B.sv_range ? ncol : rg.lo + B.sv_range;
  const bool work = rg.lo < ncol;
  u8* out = (u8*)0;
  u32 col_len = 0;
  if (MODE != 0) { rg.carry = sv_carry_unpack(B.svj_cin + 3 * (u64)j); rg.total0 = 0; }
  if (MODE == 2) {
    if (!work) return;
    const u32* cl = B.c_col_len + (u64)c * AMH_NCOLS;
    col_len = cl[col];
    if (col_len == 0) return;
    u64 o = (u64)B.c_out_off[c] + B.c_cols_off[c];
    for (u32 q = 0; q < col; q++) o += cl[q];
    out = B.out_arena + o;
    rg.total0 = B.svj_tail[j];
  }
  if (work) {
    const u8* chunk = B.arena + B.desc[d].chunk_off;
    ScCached<uint2> cobj, cmisc;
    ScCached<uint4> ckey, cval;
    cobj.init((const uint2*)B.cv[0] + base, 0, false); ckey.init((const uint4*)B.cv[2] + base, 0, false);
    cmisc.init((const uint2*)B.cv[6] + base, 0, false); cval.init((const uint4*)B.cv[8] + base, 0, false);
#define EN_NUM(SIGNED, src) { if constexpr (MODE == 0) sc_rle_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_rle_range<SIGNED, false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_rle_range<SIGNED, true>(src, rg, out, col_len, lane, any); }
#define EN_BOOL(src) { if constexpr (MODE == 0) sc_bool_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_bool_range<false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_bool_range<true>(src, rg, out, col_len, lane, any); }
#define EN_STR(src) { if constexpr (MODE == 0) sc_str_summary(src, rg.lo, rg.hi, lane, S, any); \
      else if constexpr (MODE == 1) total = sc_str_range<false>(src, rg, (u8*)0, 0, lane, any); \
      else sc_str_range<true>(src, rg, out, col_len, lane, any); }
    switch (col) {
      case 0:  { ScObjSrc s{cobj, false}; EN_NUM(false, s) } break;
      case 1:  { ScObjSrc s{cobj, true}; EN_NUM(false, s) } break;
      case 2:  { ScKeyActorSrc s{ckey}; EN_NUM(false, s) } break;
      case 3:  { ScKeyCtrSrc s{ckey}; EN_NUM(true, s) } break;
      case 4:  { ScKeyStrSrc s{ckey, chunk}; EN_STR(s) } break;
      case 5:  { ScInsertSrc s{cmisc}; EN_BOOL(s) } break;
      case 6:  { ScActionSrc s{cmisc}; EN_NUM(false, s) } break;
      case 7:  { ScValMetaSrc s{cval}; EN_NUM(false, s) } break;
      case 8:  { ScValRawSrc s{cval, chunk};
                 if constexpr (MODE == 1) total = sc_raw_range<false>(s, rg, (u8*)0, 0, lane);
                 else if constexpr (MODE == 2) sc_raw_range<true>(s, rg, out, col_len, lane); } break;
      case 9:  { ScPredNumSrc s{cmisc}; EN_NUM(false, s) } break;
      case 10: { ScPredActorSrc s; s.ids = B.pred_ids + pb; s.amap = B.chg_actor_map + B.c_map_base[c]; EN_NUM(false, s) } break;
      case 11: { ScPredCtrSrc s; s.ids = B.pred_ids + pb; s.amap = B.chg_actor_map + B.c_map_base[c]; EN_NUM(true, s) } break;
      case 12: { ScExpandSrc s{cmisc}; EN_BOOL(s) } break;
      default: { ScMarkSrc s{B, B.ops_sorted + base, chunk}; EN_STR(s) } break;
    }
#undef EN_NUM
#undef EN_BOOL
#undef EN_STR
  }
  if (lane == 0) {
    if (MODE == 0) { u32* w = B.svj_sum + 9 * (u64)j; sv_carry_pack(S.h[0], w); sv_carry_pack(S.h[1], w + 3); sv_carry_pack(S.h[2], w + 6); }
    if (MODE == 1) { B.svj_total[j] = total; B.svj_any[j] = any ? 1u : 0u; }
  }
}
void amh_launch_save_big(const BatchDev& B, int phase, void* stream) {
  if (B.sv_nj == 0) return;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gj((B.sv_nj + SD_WAVES - 1) / SD_WAVES), bj(64 * SD_WAVES), gc((B.sv_nbc + AMH_BLOCK - 1) / AMH_BLOCK), bcb(AMH_BLOCK);
  if (phase == 0) {          // save: column lengths
    hipLaunchKernelGGL((g_sv_big<0>), gj, bj, 0, s, B);
    hipLaunchKernelGGL(g_sv_big_chain, gc, bcb, 0, s, B);
    hipLaunchKernelGGL((g_sv_big<1>), gj, bj, 0, s, B);
    hipLaunchKernelGGL((g_sv_big_total<true>), gc, bcb, 0, s, B);
  } else if (phase == 1) {   // save: bytes
    hipLaunchKernelGGL((g_sv_big<2>), gj, bj, 0, s, B);
  } else if (phase == 2) {   // change re-encode: column lengths
    hipLaunchKernelGGL((g_en_big<0>), gj, bj, 0, s, B);
    hipLaunchKernelGGL(g_sv_big_chain, gc, bcb, 0, s, B);
    hipLaunchKernelGGL((g_en_big<1>), gj, bj, 0, s, B);
    hipLaunchKernelGGL((g_sv_big_total<false>), gc, bcb, 0, s, B);
  } else {                   // change re-encode: bytes
    hipLaunchKernelGGL((g_en_big<2>), gj, bj, 0, s, B);
  }
}
