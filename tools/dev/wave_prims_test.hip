// Micro-test of the wave primitives of amh_segdec.inl against plain loops (development aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../automerge_amd/csrc/amh_dev.h"
struct BatchDev;
#include "../../automerge_amd/csrc/amh_segdec_prims.inl"
__global__ void k(const u32* in, u32* o_add, u32* o_max, u64* o_add64, u32* o_mb, u32* o_nx) {
  u32 l = threadIdx.x;
  u32 v = in[l];
  o_add[l] = sd_scan_add(v);
  o_max[l] = sd_scan_max(v);
  o_add64[l] = sd_scan_add64(((u64)v << 28) | v);
  u64 m = __ballot(v & 1);
  o_mb[l] = sd_mbcnt(m);
  o_nx[l] = sd_next(v);
}
int main() {
  u32 h[64], *d, *a, *m, *mb, *nx; u64* a64;
  hipMalloc(&d, 256); hipMalloc(&a, 256); hipMalloc(&m, 256); hipMalloc(&mb, 256); hipMalloc(&nx, 256); hipMalloc(&a64, 512);
  int bad = 0;
  for (int it = 0; it < 20; it++) {
    for (int i = 0; i < 64; i++) h[i] = (u32)rand() % (it < 10 ? 1000 : 0x7fffffff);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, m, a64, mb, nx);
    u32 ra[64], rm[64], rmb[64], rnx[64]; hipMemcpy(rnx, nx, 256, hipMemcpyDeviceToHost); u64 r64[64];
    hipMemcpy(ra, a, 256, hipMemcpyDeviceToHost); hipMemcpy(rm, m, 256, hipMemcpyDeviceToHost);
    hipMemcpy(rmb, mb, 256, hipMemcpyDeviceToHost); hipMemcpy(r64, a64, 512, hipMemcpyDeviceToHost);
    u32 s = 0, mx = 0, c = 0; u64 s64 = 0;
    for (int i = 0; i < 64; i++) {
      s += h[i]; if (h[i] > mx) mx = h[i]; s64 += ((u64)h[i] << 28) | h[i];
      if (ra[i] != s || rm[i] != mx || r64[i] != s64 || rmb[i] != c || rnx[i] != (i < 63 ? h[i + 1] : 0u)) { if (bad < 10) printf("it %d lane %d: add %u/%u max %u/%u add64 %llu/%llu mb %u/%u\n", it, i, ra[i], s, rm[i], mx, (unsigned long long)r64[i], (unsigned long long)s64, rmb[i], c); bad++; }
      c += h[i] & 1;
    }
  }
  printf(bad ? "FAILED %d\n" : "wave prims ok\n", bad);
  return bad != 0;
}
