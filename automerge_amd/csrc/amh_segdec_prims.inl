// Wave-level primitives of the segment-parallel column codecs (wave64, gfx950): DPP scans, broadcasts, lane ranks.
#pragma once
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ u32 sd_dpp(u32 old, u32 src) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
// inclusive wave scans (wave64): Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals
// by row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
__device__ __forceinline__ u32 sd_scan_add(u32 v) {
  v += sd_dpp<0x111, 0xF, 0xF>(0, v);
  v += sd_dpp<0x112, 0xF, 0xF>(0, v);
  v += sd_dpp<0x114, 0xF, 0xF>(0, v);
  v += sd_dpp<0x118, 0xF, 0xF>(0, v);
  v += sd_dpp<0x142, 0xA, 0xF>(0, v);
  v += sd_dpp<0x143, 0xC, 0xF>(0, v);
  return v;
}
__device__ __forceinline__ u32 sd_max(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 sd_scan_max(u32 v) {
  v = sd_max(v, sd_dpp<0x111, 0xF, 0xF>(0, v));
  v = sd_max(v, sd_dpp<0x112, 0xF, 0xF>(0, v));
  v = sd_max(v, sd_dpp<0x114, 0xF, 0xF>(0, v));
  v = sd_max(v, sd_dpp<0x118, 0xF, 0xF>(0, v));
  v = sd_max(v, sd_dpp<0x142, 0xA, 0xF>(0, v));
  v = sd_max(v, sd_dpp<0x143, 0xC, 0xF>(0, v));
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 sd_dpp64(u64 v) {
  u32 lo = sd_dpp<CTRL, ROW_MASK, 0xF>(0, (u32)v), hi = sd_dpp<CTRL, ROW_MASK, 0xF>(0, (u32)(v >> 32));
  return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ u64 sd_scan_add64(u64 v) {
  v += sd_dpp64<0x111, 0xF>(v);
  v += sd_dpp64<0x112, 0xF>(v);
  v += sd_dpp64<0x114, 0xF>(v);
  v += sd_dpp64<0x118, 0xF>(v);
  v += sd_dpp64<0x142, 0xA>(v);
  v += sd_dpp64<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ u32 sd_bcast(u32 v, u32 src_lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)src_lane); }
__device__ __forceinline__ u64 sd_bcast64(u64 v, u32 src_lane) {
  return (u64)sd_bcast((u32)v, src_lane) | ((u64)sd_bcast((u32)(v >> 32), src_lane) << 32);
}
// lane i receives lane i + 1's value (lane 63: 0): DPP wave_shl:1
__device__ __forceinline__ u32 sd_next(u32 v) { return sd_dpp<0x130, 0xF, 0xF>(0, v); }
// lane i receives lane i - 1's value (lane 0: 0): DPP wave_shr:1
__device__ __forceinline__ u32 sd_prev(u32 v) { return sd_dpp<0x138, 0xF, 0xF>(0, v); }
__device__ __forceinline__ u32 sd_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 sd_low_bits(u32 k) { return k >= 64 ? ~(u64)0 : (((u64)1 << k) - 1); }   // bits [0, k)
__device__ __forceinline__ u32 sd_mbcnt(u64 m) {   // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
}

// Lanes of a wave that hand data to each other through LDS: the hardware runs a wave's LDS instructions in order, but
// the compiler only sees one thread, and may forward a lane's own store to its later load of the same word (missing what
// another lane stored there in between) or move the accesses past each other.  This fence (no instruction) stops that.
__device__ __forceinline__ void sd_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
