// HIP launch wrappers for the batched document-load kernels (gfx950).  The per-thread bodies live
// in amh_kernels.inl; this file gives each of them a grid.
#include <hip/hip_runtime.h>

#include "amh_kernels.inl"
#include "amh_launch.h"

#define AMH_BLOCK 256

#define AMH_GLOBAL(name)                                        \
  __global__ void __launch_bounds__(AMH_BLOCK) g_##name(BatchDev B, u32 n) { \
    u32 i = blockIdx.x * AMH_BLOCK + threadIdx.x;               \
    if (i < n) name(B, i);                                      \
  }

AMH_GLOBAL(k_parse)
AMH_GLOBAL(k_sha_doc)
AMH_GLOBAL(k_count)
AMH_GLOBAL(k_extract)
AMH_GLOBAL(k_actors)
AMH_GLOBAL(k_chg_init)
AMH_GLOBAL(k_decode_chg)
AMH_GLOBAL(k_decode_ops)
AMH_GLOBAL(k_rows)
AMH_GLOBAL(k_obj_order)
AMH_GLOBAL(k_succ)
AMH_GLOBAL(k_bucket)
AMH_GLOBAL(k_chg_finish)
AMH_GLOBAL(k_place)
AMH_GLOBAL(k_pred_fill)
AMH_GLOBAL(k_pred_sort)
AMH_GLOBAL(k_actor_mark)
AMH_GLOBAL(k_actor_rank)
AMH_GLOBAL(k_encode_size)
AMH_GLOBAL(k_chg_layout)
AMH_GLOBAL(k_encode_store)
AMH_GLOBAL(k_assemble_hash)
AMH_GLOBAL(k_visible)

void amh_launch(int k, const BatchDev& B, u32 n, void* stream) {
  if (n == 0) return;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((n + AMH_BLOCK - 1) / AMH_BLOCK), block(AMH_BLOCK);
#define L(id, name) case id: hipLaunchKernelGGL(g_##name, grid, block, 0, s, B, n); break;
  switch (k) {
    L(AK_PARSE, k_parse) L(AK_SHA_DOC, k_sha_doc) L(AK_COUNT, k_count) L(AK_EXTRACT, k_extract)
    L(AK_ACTORS, k_actors) L(AK_CHG_INIT, k_chg_init) L(AK_DECODE_CHG, k_decode_chg)
    L(AK_DECODE_OPS, k_decode_ops) L(AK_ROWS, k_rows) L(AK_OBJ_ORDER, k_obj_order) L(AK_SUCC, k_succ)
    L(AK_BUCKET, k_bucket) L(AK_CHG_FINISH, k_chg_finish) L(AK_PLACE, k_place) L(AK_PRED_FILL, k_pred_fill)
    L(AK_PRED_SORT, k_pred_sort) L(AK_ACTOR_MARK, k_actor_mark) L(AK_ACTOR_RANK, k_actor_rank)
    L(AK_ENCODE_SIZE, k_encode_size) L(AK_CHG_LAYOUT, k_chg_layout) L(AK_ENCODE_STORE, k_encode_store)
    L(AK_ASSEMBLE_HASH, k_assemble_hash) L(AK_VISIBLE, k_visible)
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

__global__ void __launch_bounds__(AMH_BLOCK) g_scan_reduce(const u32* data, u32 n, u32* tile_sums) {
  u32 start = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  u32 sum = 0;
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (start + k < n) sum = sat_add(sum, data[start + k]);
  u32 tot;
  block_exclusive_scan(sum, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(AMH_BLOCK) g_scan_apply(u32* data, u32 n, const u32* tile_offsets, u32* total_out) {
  u32 start = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  u32 sum = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (start + k < n) ? data[start + k] : 0;
    sum = sat_add(sum, v[k]);
  }
  u32 tot;
  u32 excl = block_exclusive_scan(sum, &tot);
  u32 run = sat_add(tile_offsets ? tile_offsets[blockIdx.x] : 0, excl);
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (start + k < n) data[start + k] = run;
    run = sat_add(run, v[k]);
  }
  // the last tile writes the grand total
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == AMH_BLOCK - 1) *total_out = run;
}

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
