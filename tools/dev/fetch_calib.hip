// FETCH_SIZE calibration for this library's access patterns (MI355X_MICROARCH.md, HBM: widths other than 16 B per lane
// are uncalibrated).  Streams a 1 GiB buffer once with 4-byte, 8-byte and 16-byte coalesced loads per lane, and with the
// 256-byte tile pattern of g_decode_seg (a wave reads 256 consecutive bytes, 4 per lane, then moves on by a document
// stride); run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with the known byte count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <class T> __global__ void stream(const T* p, size_t n, unsigned long long* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (; i < n; i += stride) { T v = p[i]; acc += ((const unsigned*)&v)[0]; }
  if (acc == 0x123456789ull) *sink = acc;
}
// wave w walks "its" 64 KiB region 256 bytes at a time (serial tiles), like a column walk
__global__ void tiles(const unsigned* p, size_t n_words, unsigned long long* sink) {
  size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
  size_t base = wave * (16384);   // words
  unsigned long long acc = 0;
  if (base + 16384 <= n_words)
    for (size_t t = 0; t < 16384; t += 64) acc += p[base + t + lane];
  if (acc == 0x123456789ull) *sink = acc;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void* p; unsigned long long* sink;
  hipMalloc(&p, bytes); hipMalloc(&sink, 8);
  hipMemset(p, 1, bytes);
  hipDeviceSynchronize();
  stream<unsigned><<<4096, 256>>>((const unsigned*)p, bytes / 4, sink);
  stream<uint2><<<4096, 256>>>((const uint2*)p, bytes / 8, sink);
  stream<uint4><<<4096, 256>>>((const uint4*)p, bytes / 16, sink);
  tiles<<<(unsigned)(bytes / 65536 / 4), 256>>>((const unsigned*)p, bytes / 4, sink);
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", bytes);
  return 0;
}
