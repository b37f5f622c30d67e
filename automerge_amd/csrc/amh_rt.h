// Thin device-runtime layer used by the host orchestration: HIP in the product build, plain host
// memory in the CPU-only emulation build used by tests (AMH_EMU; never part of the product library).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#ifdef AMH_EMU
#include <stdlib.h>
#include <string.h>

#include <chrono>
typedef void* rt_stream;
struct rt_event { std::chrono::steady_clock::time_point t; };
inline bool rt_init(int, rt_stream* s, std::string&) { *s = nullptr; return true; }
inline void rt_shutdown(rt_stream) {}
inline void* rt_malloc(size_t n) { return malloc(n ? n : 1); }
inline void rt_free(void* p) { free(p); }
inline void* rt_host_alloc(size_t n) { return malloc(n ? n : 1); }
inline void rt_host_free(void* p) { free(p); }
inline bool rt_h2d(void* d, const void* h, size_t n, rt_stream) { if (n) memcpy(d, h, n); return true; }
inline bool rt_d2h(void* h, const void* d, size_t n, rt_stream) { if (n) memcpy(h, d, n); return true; }
inline bool rt_d2h_async(void* h, const void* d, size_t n, rt_stream) { if (n) memcpy(h, d, n); return true; }
inline bool rt_d2d(void* d, const void* s, size_t n, rt_stream) { if (n) memcpy(d, s, n); return true; }
inline bool rt_memset(void* d, int v, size_t n, rt_stream) { if (n) memset(d, v, n); return true; }
inline bool rt_sync(rt_stream) { return true; }
inline bool rt_mem_info(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)64 << 30; return true; }
inline void rt_event_create(rt_event*) {}
inline void rt_event_sync(rt_event*) {}
inline void rt_event_destroy(rt_event*) {}
inline void rt_event_record(rt_event* e, rt_stream) { e->t = std::chrono::steady_clock::now(); }
inline float rt_event_ms(rt_event* a, rt_event* b) { return std::chrono::duration<float, std::milli>(b->t - a->t).count(); }
inline void rt_set_device(int) {}
inline bool rt_stream_create(rt_stream* s) { *s = nullptr; return true; }
inline void rt_stream_wait(rt_stream, rt_event*) {}
inline const char* rt_last_error() { return "emu"; }
#else
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>
typedef hipStream_t rt_stream;
struct rt_event { hipEvent_t e; };
// AMH_COPY_TRACE=1 (development): transfers and fills counted per calling line of amh_host.cpp, printed at exit
struct RtCopyTrace {
  std::mutex m;
  std::map<std::pair<char, int>, std::pair<size_t, size_t>> sites;   // (kind, line) -> (calls, bytes)
  bool on = getenv("AMH_COPY_TRACE") != nullptr;
  void note(char kind, int line, size_t n) { std::lock_guard<std::mutex> lk(m); auto& s = sites[{kind, line}]; s.first++; s.second += n; }
  ~RtCopyTrace() {
    if (!on) return;
    for (auto& kv : sites) fprintf(stderr, "[amh copies] %c line %5d: %7zu calls %14zu bytes\n", kv.first.first, kv.first.second, kv.second.first, kv.second.second);
  }
};
inline RtCopyTrace& rt_copy_trace() { static RtCopyTrace t; return t; }
#define RT_NOTE(kind) do { RtCopyTrace& t_ = rt_copy_trace(); if (t_.on) t_.note(kind, line, n); } while (0)
inline bool rt_init(int device, rt_stream* s, std::string& err) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { err = "no HIP device available (the HIP path is required; there is no CPU fallback)"; return false; }
  if (device >= n) { err = "gpu id out of range"; return false; }
  if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return false; }
  if (hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
  return true;
}
inline void rt_shutdown(rt_stream s) { (void)hipStreamDestroy(s); }
inline void* rt_malloc(size_t n, int line = __builtin_LINE()) { RT_NOTE('m'); void* p = nullptr; if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr; return p; }
inline void rt_free(void* p, size_t n = 0, int line = __builtin_LINE()) { RT_NOTE('r'); (void)hipFree(p); }
inline void* rt_host_alloc(size_t n) { void* p = nullptr; if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return nullptr; return p; }
inline void rt_host_free(void* p) { (void)hipHostFree(p); }
inline bool rt_h2d(void* d, const void* h, size_t n, rt_stream s, int line = __builtin_LINE()) { RT_NOTE('u'); return n == 0 || hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess; }
inline bool rt_d2h(void* h, const void* d, size_t n, rt_stream s, int line = __builtin_LINE()) {
  RT_NOTE('d');
  if (n == 0) return true;
  if (hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) != hipSuccess) return false;
  return hipStreamSynchronize(s) == hipSuccess;
}
inline bool rt_d2h_async(void* h, const void* d, size_t n, rt_stream s, int line = __builtin_LINE()) { RT_NOTE('a'); return n == 0 || hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess; }   // h pinned
inline bool rt_d2d(void* d, const void* src, size_t n, rt_stream s, int line = __builtin_LINE()) { RT_NOTE('c'); return n == 0 || hipMemcpyAsync(d, src, n, hipMemcpyDeviceToDevice, s) == hipSuccess; }
inline bool rt_memset(void* d, int v, size_t n, rt_stream s, int line = __builtin_LINE()) { RT_NOTE('f'); return n == 0 || hipMemsetAsync(d, v, n, s) == hipSuccess; }
inline bool rt_sync(rt_stream s) { return hipStreamSynchronize(s) == hipSuccess; }
inline bool rt_mem_info(size_t* free_b, size_t* total_b) { return hipMemGetInfo(free_b, total_b) == hipSuccess; }
inline void rt_event_create(rt_event* e) { (void)hipEventCreate(&e->e); }
inline void rt_event_sync(rt_event* e) { (void)hipEventSynchronize(e->e); }
inline void rt_event_destroy(rt_event* e) { (void)hipEventDestroy(e->e); }
inline void rt_event_record(rt_event* e, rt_stream s) { (void)hipEventRecord(e->e, s); }
inline float rt_event_ms(rt_event* a, rt_event* b) { float ms = 0; (void)hipEventSynchronize(b->e); (void)hipEventElapsedTime(&ms, a->e, b->e); return ms; }
inline void rt_set_device(int d) { (void)hipSetDevice(d); }   // per host thread
inline bool rt_stream_create(rt_stream* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess; }
inline void rt_stream_wait(rt_stream s, rt_event* e) { (void)hipStreamWaitEvent(s, e->e, 0); }   // s goes on once e has happened
inline const char* rt_last_error() { return hipGetErrorString(hipGetLastError()); }
#endif
