// Host orchestration of the batched load path and the C ABI of libautomerge_hip.so
// (include/automerge_hip.h).  The host stages bytes, sizes buffers and launches kernels; every
// per-op / per-change computation happens in the kernels of amh_kernels.inl.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <memory>
#include <set>
#include <cmath>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unistd.h>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/automerge_hip.h"
#include "amh_ingest.h"
#include "amh_launch.h"
#ifndef AMH_EMU
#include "amh_comm.h"
#endif
#include "amh_rt.h"
#include "amh_types.h"

#define AMH_EMPTY32 0xFFFFFFFFu
#define AMH_NCOLS 14
#define AMH_BIGCOL_HOST (32u << 10)   // = AMH_BIGCOL / AMH_SEG of amh_segdec.inl
#define AMH_SEG_HOST 4096u
#define AMH_LEVEL_CAP_HOST 48u   // = AMH_LEVEL_CAP of amh_kernels.inl
#define AMH_HOST_CHAIN_MIN 2048u

static thread_local std::string g_last_error;

// AMH_TIMING=1: wall-clock of the host phases of merge / apply on stderr (diagnostics)
#include <chrono>
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  const char* who;
  PhaseTimer(const char* w) : on(getenv("AMH_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), who(w) {}
  void mark(const char* phase, rt_stream* s = nullptr) {
    if (!on) return;
    if (s) rt_sync(*s);
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[amh timing] %s: %-28s %8.2f ms\n", who, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// ---------------------------------------------------------------------------------------------
// caching device allocator: steady-state batches allocate nothing
// ---------------------------------------------------------------------------------------------
struct DevPool {
  std::multimap<size_t, void*> free_blocks;
  size_t live_bytes = 0, pooled_bytes = 0;
  void* alloc(size_t n, size_t* got) {
    n = (n + 255) & ~(size_t)255;
    if (n == 0) n = 256;
    auto it = free_blocks.lower_bound(n);
    // a block up to a quarter larger than asked for; for large requests (a hipMalloc of a gigabyte takes milliseconds, and
    // a request that keeps missing makes the pool grow by one block per call) up to twice as large
    if (it != free_blocks.end() && (it->first <= n + n / 4 + 4096 || (n >= ((size_t)64 << 20) && it->first <= 2 * n))) {
      void* p = it->second;
      *got = it->first;
      pooled_bytes -= it->first;
      free_blocks.erase(it);
      live_bytes += *got;
      return p;
    }
    static const bool trace = getenv("AMH_POOL_TRACE") != nullptr;   // (development: which requests miss the pool)
    if (trace) fprintf(stderr, "[amh pool] hipMalloc %zu bytes (live %zu, pooled %zu in %zu blocks)\n", n, live_bytes, pooled_bytes, free_blocks.size());
    void* p = rt_malloc(n);
    if (!p) {  // release the pool and retry once
      trim();
      p = rt_malloc(n);
    }
    if (p) { *got = n; live_bytes += n; }
    return p;
  }
  void release(void* p, size_t n) {
    if (!p) return;
    free_blocks.insert({n, p});
    live_bytes -= n;
    pooled_bytes += n;
  }
  void trim() {
    for (auto& kv : free_blocks) rt_free(kv.second);
    free_blocks.clear();
    pooled_bytes = 0;
  }
};

struct amh_ctx {
  int device = 0;
  rt_stream stream{};
  rt_stream side{};          // second stream: kernels that are independent of the main chain run beside it (fork / join by events)
  rt_event ev_fork{}, ev_fork2{}, ev_join{};
  DevPool pool;
  size_t total_mem = 0;      // device memory of this GPU (batch_byte_budget)
  // pinned staging area of apply / merge_many calls, kept between calls (grow only): no page faults on fresh memory
  // every call, and the H2D copy of the change chunks runs at the pinned rate
  u8* stage = nullptr;
  size_t stage_cap = 0;
  u8* tabs = nullptr;             // pinned: the per-chunk tables of a wide merge (sources, places, hashes, counts) on their way to the device
  size_t tabs_cap = 0;
  u8* heads = nullptr;            // pinned: the packed heads of device-resident change chunks on their way to the host
  size_t heads_cap = 0;
  u8* res = nullptr;              // pinned: per-chunk results of an apply's decode pass (headers, counts, hashes)
  size_t res_cap = 0;
  u8* bigc = nullptr;             // pinned: the re-derived change chunks of a large document on their way to the host's SHA-256
  size_t bigc_cap = 0;
  u8* pin[2] = {nullptr, nullptr};   // pinned staging buffers of amh_stage_batch (double-buffered H2D), kept for the context's life
  std::mutex pin_mu;                 // ... also the way out for large results (d2h_large): one copy at a time
  rt_event ev_pin[2]{};
  bool profiling = false;
  // limits (amh_ctx_set_limits): a document over the per-document limit, or the largest documents of a batch whose
  // derived arrays would not fit the byte budget, fail with AMH_DOC_TOO_LARGE and the rest of the batch goes on
  struct AmhComm* comm = nullptr;   // RCCL communicator of this process's rank (amh_comm_init); none = single GPU
  u32 max_doc_rows = AMH_MAX_ROWS_PER_DOC;
  u64 max_batch_bytes = 0;   // 0 = what the device has free (plus what the pool holds) when the batch is sized
  double kernel_ms[AK_NUM + MK_NUM + SK_NUM] = {0};      // load / save kernels, the merge / apply kernels, the sync message kernels
  uint64_t kernel_calls[AK_NUM + MK_NUM + SK_NUM] = {0};
  struct Pending { int k; rt_event a, b; };
  std::vector<Pending> pending;
  std::string last_error;
};

static void set_error(amh_ctx* ctx, const std::string& m) {
  g_last_error = m;
  if (ctx) ctx->last_error = m;
}

struct DevAlloc {
  void* p = nullptr;
  size_t cap = 0;
};

struct amh_staged {
  amh_ctx* ctx = nullptr;
  uint32_t n_docs = 0;
  std::vector<uint8_t> h_arena;   // normalised input bytes (kept for string / actor access on the host)
  std::vector<u64> h_doc_off;
  DevAlloc d_arena, d_doc_off;
  size_t arena_bytes = 0;         // bytes of the arena (host and device copies alike)
  bool host_ok = true;            // h_arena is filled (arenas assembled on the device fetch it on first use)
  bool single_chunk = false;      // every input is exactly one document chunk (arenas assembled by merge / apply)
  bool internal = false;          // the chunks were written by the engine for its own reload: they carry no checksum
  int refs = 1;
};
// host copy of an arena that was assembled on the device (merge / apply results): fetched when an accessor needs bytes
static bool ensure_host_arena(const amh_staged* sc) {
  amh_staged* s = const_cast<amh_staged*>(sc);
  if (s->host_ok) return true;
  s->h_arena.resize(s->arena_bytes);
  if (!rt_d2h(s->h_arena.data(), s->d_arena.p, s->arena_bytes, s->ctx->stream) || !rt_sync(s->ctx->stream)) return false;
  s->host_ok = true;
  return true;
}
static void staged_unref(amh_staged* s) {
  if (!s) return;
  if (--s->refs > 0) return;
  if (s->d_arena.cap) s->ctx->pool.release(s->d_arena.p, s->d_arena.cap);   // cap 0: borrowed (rows_to_batch, until the merge hands it over)
  s->ctx->pool.release(s->d_doc_off.p, s->d_doc_off.cap);
  delete s;
}

// std::vector without the zero fill of resize(): the mirrors below are overwritten in full by the copy that follows
// (a 70,000-document batch's history is 60 MB: zeroing it first costs as much as fetching it)
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U> void construct(U* p) { ::new ((void*)p) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using HVec = std::vector<T, NoInitAlloc<T>>;
struct HostView {  // lazily fetched mirrors for the accessors
  bool base_ok = false, chg_ok = false, out_ok = false;
  HVec<DocCounts> cnt;
  HVec<DocDesc> desc;
  HVec<u32> row_base, succ_base, chg_base, dep_base, actor_base, doc_max_op;
  HVec<u32> c_actor, c_nops, c_ndeps, c_dep_base, c_out_off, c_out_len, dep_idx, actor_off, actor_len, c_doc;
  HVec<u64> c_seq, c_start_op;
  HVec<i64> c_time;
  HVec<u8> c_hash, out_arena;
  std::map<u32, std::vector<u8>> heads_cache;
  // the actor tables of the documents, packed (ensure_actor_pack): what an apply reads of the documents it changes when
  // their bytes are not on the host
  bool pack_ok = false;
  std::vector<u8> actor_pack;
  std::vector<u32> actor_pack_off;   // [n_docs + 1]
};

struct amh_batch {
  amh_ctx* ctx = nullptr;
  amh_staged* staged = nullptr;
  BatchDev B{};
  std::vector<DevAlloc> allocs;
  std::vector<i32> status;
  std::vector<u32> fail_line;
  mutable std::string msg_buf;
  u32 tot_ht = 0, tot_map = 0, tot_out = 0;
  bool history_only = false;   // amh_batch_release_ops: the per-op arrays are gone, the history (changes, hashes, heads) stays
  DevAlloc sm_msg, sm_sent, sm_off, sm_job;   // amh_generate_sync_messages_device: the messages of the last call (sm_to_device)
  u64 sm_total = 0, sm_jobs = 0; u32 sm_n = 0;
  DevAlloc text_out, text_doc; // amh_text_batch_device: the texts of the last call, kept until the next one or until the batch is freed
  u64 text_total = 0;
  mutable HostView hv;
  // changes waiting for their dependencies (Automerge::queue, automerge.rs:168): raw chunks per document
  std::unordered_map<u32, std::vector<std::vector<u8>>> queue;
};

static amh_batch* load_staged_impl(amh_ctx* ctx, const amh_staged* st_c, const amh_load_opts* opts, bool trailing);
// job tables of the columns that are encoded over many waves (save_big_prepare / enc_big_prepare)
struct SaveBigHost {
  std::vector<u32> j_bc, j_k, b_doc, b_col, b_first, b_nr, nops, opbase;
  std::vector<u8> big;
  std::vector<DocCounts> cnt;
};
static bool enc_big_prepare(amh_ctx* ctx, amh_batch* b, BatchDev& B, SaveBigHost& H);
static void save_big_launch(amh_ctx* ctx, const BatchDev& B, int phase);

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
static void launch_on(amh_ctx* ctx, rt_stream st, int k, const BatchDev& B, u32 n) {
  if (n == 0) return;
  if (ctx->profiling) {
    amh_ctx::Pending p;
    p.k = k;
    rt_event_create(&p.a);
    rt_event_create(&p.b);
    rt_event_record(&p.a, st);
    amh_launch(k, B, n, (void*)st);
    rt_event_record(&p.b, st);
    ctx->pending.push_back(p);
  } else {
    amh_launch(k, B, n, (void*)st);
  }
}
static void launch(amh_ctx* ctx, int k, const BatchDev& B, u32 n) { launch_on(ctx, ctx->stream, k, B, n); }
static const char* const MK_KERNEL_NAMES[MK_NUM] = {
  "mg_new_flag", "mg_pair_counts", "mg_subtree", "mg_gap", "mg_inherit", "mg_extra_cnt", "mg_rows_dst", "mg_rows_new", "mg_succ_dst",
  "mg_succ_new", "mg_succ_extra", "mg_succ_sort", "mg_doc_succ", "mg_changes", "mg_copy_chunks",
  "ap_parse", "ap_count", "ap_verify", "ap_map", "ap_chg_meta", "ap_new_flag", "ap_link", "ap_children", "ap_sort_children", "ap_succ_init", "ap_up_jump", "ap_succ_build", "ap_rank_jump", "ap_weight", "ap_top_k",
  "ap_top_place", "ap_emit", "ap_pred_cnt", "ap_succ_fill_u", "ap_succ_sort_u", "ap_succ_fill_d", "ap_ties"};
// a merge / apply kernel, timed like the others when the context is profiling
static void launch_mg_timed(amh_ctx* ctx, int k, const MergeDev* G, u32 n) {
  if (ctx->profiling) {
    amh_ctx::Pending p;
    p.k = AK_NUM + k;
    rt_event_create(&p.a);
    rt_event_create(&p.b);
    rt_event_record(&p.a, ctx->stream);
    amh_launch_mg(k, G, n, (void*)ctx->stream);
    rt_event_record(&p.b, ctx->stream);
    ctx->pending.push_back(p);
  } else {
    amh_launch_mg(k, G, n, (void*)ctx->stream);
  }
}
static void collect_profile(amh_ctx* ctx) {
  for (auto& p : ctx->pending) {
    ctx->kernel_ms[p.k] += rt_event_ms(&p.a, &p.b);
    ctx->kernel_calls[p.k]++;
    rt_event_destroy(&p.a);
    rt_event_destroy(&p.b);
  }
  ctx->pending.clear();
}

template <class T>
static T* balloc(amh_batch* b, size_t count, bool* ok) {
  DevAlloc a;
  a.p = b->ctx->pool.alloc(count * sizeof(T), &a.cap);
  if (!a.p) { *ok = false; return nullptr; }
  b->allocs.push_back(a);
  return (T*)a.p;
}

// Device memory a call takes for its own use (not part of the batch's state): handed back to the pool when the call
// returns, on every path.  The call's stream work must have drained by then (the calls read their results back first).
struct CallScratch {
  amh_batch* b;
  size_t mark;
  explicit CallScratch(amh_batch* bb) : b(bb), mark(bb->allocs.size()) {}
  ~CallScratch() {
    rt_sync(b->ctx->stream);
    while (b->allocs.size() > mark) { b->ctx->pool.release(b->allocs.back().p, b->allocs.back().cap); b->allocs.pop_back(); }
  }
};

static bool scan_inplace(amh_batch* b, u32* data, u32 n, u32* total) {
  amh_ctx* ctx = b->ctx;
  bool ok = true;
  u32* tmp = balloc<u32>(b, amh_scan_tmp_words(n), &ok);
  if (!ok) return false;
  if (ctx->profiling) {
    amh_ctx::Pending p;
    p.k = AK_SCAN;
    rt_event_create(&p.a);
    rt_event_create(&p.b);
    rt_event_record(&p.a, ctx->stream);
    amh_scan_inplace(data, n, tmp, (void*)ctx->stream);
    rt_event_record(&p.b, ctx->stream);
    ctx->pending.push_back(p);
  } else {
    amh_scan_inplace(data, n, tmp, (void*)ctx->stream);
  }
  if (total) return rt_d2h(total, data + n, sizeof(u32), ctx->stream);
  return true;
}

// A large result into caller-visible (pageable) host memory: through the context's two pinned buffers, the DMA of one
// piece running while the host copies the piece before it out (a pageable hipMemcpy runs at a fraction of the link).
static bool ensure_pins(amh_ctx* ctx) {
  const size_t PIN = (size_t)32 << 20;
  for (int k = 0; k < 2; k++)
    if (!ctx->pin[k]) {
      ctx->pin[k] = (u8*)rt_host_alloc(PIN);
      rt_event_create(&ctx->ev_pin[k]);
      if (!ctx->pin[k]) return false;
    }
  return true;
}
static bool d2h_large(amh_ctx* ctx, void* dst, const void* d_src, size_t n, size_t min_bytes = (size_t)4 << 20) {
  const size_t PIN = (size_t)32 << 20;
  const rt_stream stream = ctx->stream;
  if (n < min_bytes) return rt_d2h(dst, d_src, n, stream);
  std::lock_guard<std::mutex> pin_lock(ctx->pin_mu);   // the two pinned buffers serve one copy at a time
  if (!ensure_pins(ctx)) return rt_d2h(dst, d_src, n, stream);
  const size_t pieces = (n + PIN - 1) / PIN;
  bool ok = true;
  for (size_t i = 0; i <= pieces && ok; i++) {
    if (i < pieces) {
      const size_t o = i * PIN, l = std::min(PIN, n - o);
      ok = rt_d2h_async(ctx->pin[i & 1], (const u8*)d_src + o, l, stream);
      rt_event_record(&ctx->ev_pin[i & 1], stream);
    }
    if (i > 0) {
      const size_t o = (i - 1) * PIN, l = std::min(PIN, n - o);
      rt_event_sync(&ctx->ev_pin[(i - 1) & 1]);
      // fresh pageable memory faults its pages in on first touch: four threads take a quarter of the piece each
      const u8* src = ctx->pin[(i - 1) & 1];
      const size_t q = ((l / 4) + 4095) & ~(size_t)4095;
      std::thread th[3];
      for (int t = 0; t < 3; t++) th[t] = std::thread([=]() { const size_t a = std::min(l, (size_t)(t + 1) * q), e = std::min(l, (size_t)(t + 2) * q); if (e > a) memcpy((u8*)dst + o + a, src + a, e - a); });
      memcpy((u8*)dst + o, src, std::min(l, q));
      for (int t = 0; t < 3; t++) th[t].join();
    }
  }
  return ok && rt_sync(stream);
}

// ---------------------------------------------------------------------------------------------
// stage: host ingest + one H2D copy
// ---------------------------------------------------------------------------------------------
// inputs of this size and more have their SHA-256 taken by host threads (see BigSha below)
#define AMH_HOST_SHA_MIN (256u << 10)
extern "C" amh_staged* amh_stage_batch(amh_ctx* ctx, const uint8_t* const* docs, const size_t* lens, size_t n) {
  if (!ctx || (n && (!docs || !lens)) || n >= 0x7FFFFFFFull) { set_error(ctx, "invalid argument"); return nullptr; }
  auto* st = new amh_staged();
  st->ctx = ctx;
  st->n_docs = (uint32_t)n;
  // pass 1: normalise deflated documents (rare), in parallel
  std::vector<std::vector<uint8_t>> rewritten(n);
  std::vector<uint8_t> changed(n, 0);
  unsigned nthreads = std::min<size_t>(std::max<size_t>(1, std::thread::hardware_concurrency()), 16);
  if (n < 64) nthreads = 1;
  {
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < nthreads; t++)
      ts.emplace_back([&, t]() {
        for (size_t i = t; i < n; i += nthreads)
          if (lens[i] && amh_ingest_normalise(docs[i], lens[i], rewritten[i])) changed[i] = 1;
      });
    for (auto& th : ts) th.join();
  }
  st->h_doc_off.resize(n + 1);
  u64 off = 0;
  for (size_t i = 0; i < n; i++) {
    st->h_doc_off[i] = off;
    off += changed[i] ? rewritten[i].size() : lens[i];
    off = (off + 15) & ~(u64)15;  // 16-byte aligned chunk starts: wide loads in the streaming kernels
  }
  st->h_doc_off[n] = off;
  // document i occupies [doc_off[i], doc_off[i] + len_i); the kernels need the exact length, so the
  // offsets table carries 2 entries per document: start and end
  std::vector<u64> se(2 * n + 2);
  bool any_large = false;
  for (size_t i = 0; i < n; i++) {
    size_t l = changed[i] ? rewritten[i].size() : lens[i];
    se[2 * i] = st->h_doc_off[i];
    se[2 * i + 1] = st->h_doc_off[i] + l;
    any_large = any_large || l >= AMH_HOST_SHA_MIN;
  }
  st->arena_bytes = (size_t)off + 256;   // tail padding: the staged decoders read whole windows
  st->d_arena.p = ctx->pool.alloc(st->arena_bytes, &st->d_arena.cap);
  st->d_doc_off.p = ctx->pool.alloc(se.size() * sizeof(u64), &st->d_doc_off.cap);
  if (!st->d_arena.p || !st->d_doc_off.p) { set_error(ctx, "device out of memory while staging"); staged_unref(st); return nullptr; }
  // The bytes go to the device through two pinned buffers: while one is on its way over PCIe (async copy), the host
  // threads fill the other.  No pageable copy, and no host-side mirror of the arena: accessors that need bytes on the
  // host (actor ids, keys of a hydrate) fetch it on first use, except for batches with documents large enough for
  // host-side SHA-256 (those threads read the mirror while the device pipeline runs).
  const size_t PIN = (size_t)32 << 20;
  if (!ensure_pins(ctx)) { set_error(ctx, "out of pinned host memory while staging"); staged_unref(st); return nullptr; }
  if (any_large) st->h_arena.assign(st->arena_bytes, 0);
  st->host_ok = any_large;
  bool ok = rt_h2d(st->d_doc_off.p, se.data(), se.size() * sizeof(u64), ctx->stream);
  const u64 total = (u64)off + 256;
  size_t first_doc = 0;   // first document that reaches into the current piece
  int piece = 0;
  bool pending[2] = {false, false};
  for (u64 c0 = 0; c0 < total && ok; c0 += PIN, piece++) {
    const u64 c1 = std::min<u64>(total, c0 + PIN);
    const int k = piece & 1;
    u8* buf = ctx->pin[k];
    if (pending[k]) rt_event_sync(&ctx->ev_pin[k]);   // the copy that last read this buffer is done
    while (first_doc < n && (first_doc + 1 < n ? se[2 * (first_doc + 1)] : total) <= c0) first_doc++;   // its bytes and the gap behind them lie in front of the piece
    size_t last_doc = first_doc;
    while (last_doc < n && se[2 * last_doc] < c1) last_doc++;
    auto fill = [&](size_t i) {   // document i's bytes and the zero gap behind them, clipped to [c0, c1)
      const uint8_t* src = changed[i] ? rewritten[i].data() : docs[i];
      const u64 s0 = se[2 * i], s1 = se[2 * i + 1], g1 = i + 1 < n ? se[2 * (i + 1)] : total;
      u64 lo = std::max(s0, c0), hi = std::min(s1, c1);
      if (lo < hi) memcpy(buf + (lo - c0), src + (lo - s0), (size_t)(hi - lo));
      lo = std::max(s1, c0); hi = std::min(g1, c1);
      if (lo < hi) memset(buf + (lo - c0), 0, (size_t)(hi - lo));
    };
    const size_t nd = last_doc - first_doc;
    if (n == 0) memset(buf, 0, (size_t)(c1 - c0));
    else if (nd < 64 || nthreads == 1) { for (size_t i = first_doc; i < last_doc; i++) fill(i); }
    else {
      std::vector<std::thread> ts;
      for (unsigned t = 0; t < nthreads; t++)
        ts.emplace_back([&, t]() { for (size_t i = first_doc + t; i < last_doc; i += nthreads) fill(i); });
      for (auto& th : ts) th.join();
    }
    if (any_large) memcpy(st->h_arena.data() + c0, buf, (size_t)(c1 - c0));
    ok = rt_h2d((u8*)st->d_arena.p + c0, buf, (size_t)(c1 - c0), ctx->stream);
    rt_event_record(&ctx->ev_pin[k], ctx->stream);
    pending[k] = true;
  }
  ok = ok && rt_sync(ctx->stream);
  if (!ok) { set_error(ctx, std::string("H2D copy failed: ") + rt_last_error()); staged_unref(st); return nullptr; }
  // keep exact ends on the host too
  st->h_doc_off.resize(2 * n + 2);
  for (size_t i = 0; i < 2 * n + 2; i++) st->h_doc_off[i] = se[i];
  return st;
}
extern "C" void amh_staged_free(amh_staged* s) { staged_unref(s); }

// ---------------------------------------------------------------------------------------------
// the device pipeline
// ---------------------------------------------------------------------------------------------
extern "C" void amh_batch_free(amh_batch* b) {
  if (!b) return;
  for (auto& a : b->allocs) b->ctx->pool.release(a.p, a.cap);
  for (DevAlloc* a : {&b->sm_msg, &b->sm_sent, &b->sm_off, &b->sm_job}) if (a->p) b->ctx->pool.release(a->p, a->cap);
  if (b->text_out.p) b->ctx->pool.release(b->text_out.p, b->text_out.cap);
  if (b->text_doc.p) b->ctx->pool.release(b->text_doc.p, b->text_doc.cap);
  staged_unref(b->staged);
  delete b;
}

static bool ensure_chg(const amh_batch* b);
// Keep the history of a loaded batch, give everything per op back to the pool.  What stays serves amh_doc_status,
// amh_heads, amh_changes, amh_save_after, amh_sync_summaries and the source side of amh_merge_many; everything that needs
// the op rows (hydrate, save, apply, merge as dst, the pairwise amh_merge_batch) answers AMH_ERR_INVALID_ARG afterwards.
extern "C" int amh_batch_release_ops(amh_ctx* ctx, amh_batch* b) {
  if (!ctx || !b) return AMH_ERR_INVALID_ARG;
  if (b->history_only) return AMH_SUCCESS;
  PhaseTimer pt("amh_batch_release_ops");
  if (!ensure_chg(b) || !rt_sync(ctx->stream) || !rt_sync(ctx->side)) return AMH_ERR_DEVICE;   // the host's view of the history, while its sources exist
  pt.mark("host view of the history");
  const BatchDev& B = b->B;
  const void* keep[] = {B.status, B.fail_line, B.desc, B.cnt, B.row_base, B.succ_base, B.chg_base, B.dep_base, B.actor_base, B.doc_hash, B.doc_max_op,
                        B.heads_ok, B.actor_off, B.actor_len, B.actor_rank, B.actor_by_rank, B.c_actor, B.c_seq, B.c_max_op, B.c_time, B.c_msg_off,
                        B.c_msg_len, B.c_ndeps, B.c_dep_base, B.c_extra_off, B.c_extra_len, B.c_doc, B.dep_idx, B.c_nops, B.c_start_op, B.c_is_dep,
                        B.c_hash, B.c_out_off, B.c_out_len, B.out_arena, B.c_npred};
  std::vector<DevAlloc> kept;
  for (auto& a : b->allocs) {
    bool k = false;
    for (const void* q : keep) k = k || (q && q == a.p);
    if (k) kept.push_back(a); else ctx->pool.release(a.p, a.cap);
  }
  b->allocs.swap(kept);
  b->history_only = true;
  return AMH_SUCCESS;
}
#define AMH_NEEDS_OPS(b, who)                                                                                                        \
  if ((b)->history_only) { set_error((b)->ctx, std::string(who) + ": the batch's op rows were released (amh_batch_release_ops)"); return AMH_ERR_INVALID_ARG; }

static int load_trailing_chunks(amh_ctx* ctx, amh_batch* b, const amh_load_opts* opts);
extern "C" amh_batch* amh_load_staged(amh_ctx* ctx, const amh_staged* st_c, const amh_load_opts* opts) {
  return load_staged_impl(ctx, st_c, opts, true);
}
// SHA-256 of one message is a serial chain: the device can only put a single lane on it (~20 MB/s), a host core does
// it 10-50x faster.  The first chunk of inputs of AMH_HOST_SHA_MIN bytes and more is therefore hashed by host threads
// out of the staged host copy WHILE the device pipeline runs (k_sha_doc skips those chunks); the verdict is applied
// with the statuses.  Batches of small documents never come here: their hashes are data-parallel on the device.
struct BigSha {
  std::vector<u32> docs;
  std::vector<u8> bad, hash;   // per entry of docs
  std::vector<u64> chunk_len;  // per entry of docs: header + data of the first chunk (0 = header unreadable)
  std::vector<std::thread> threads;
  void start(const amh_staged* st) {
    const u32 n = st->n_docs;
    if (!st->host_ok) return;
    for (u32 d = 0; d < n; d++)
      if (st->h_doc_off[2 * (size_t)d + 1] - st->h_doc_off[2 * (size_t)d] >= AMH_HOST_SHA_MIN) docs.push_back(d);
    if (docs.empty()) return;
    bad.assign(docs.size(), 0);
    chunk_len.assign(docs.size(), 0);
    hash.assign(docs.size() * 32, 0);
    auto next = std::make_shared<std::atomic<size_t>>(0);
    unsigned nt = (unsigned)std::min<size_t>(docs.size(), 8);
    for (unsigned t = 0; t < nt; t++)
      threads.emplace_back([this, st, next]() {
        for (;;) {
          size_t k = next->fetch_add(1);
          if (k >= docs.size()) return;
          const u32 d = docs[k];
          const u8* p = st->h_arena.data() + st->h_doc_off[2 * (size_t)d];
          const size_t avail = (size_t)(st->h_doc_off[2 * (size_t)d + 1] - st->h_doc_off[2 * (size_t)d]);
          // chunk header (chunk.rs:223-266): magic, checksum, type, uleb length; a header the device parser rejects is its business
          if (avail < 10 || p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83) continue;
          u64 len = 0; u32 shift = 0; size_t pos = 9; bool okh = false;
          while (pos < avail && shift < 64) { u8 b = p[pos++]; len |= (u64)(b & 0x7f) << shift; shift += 7; if (!(b & 0x80)) { okh = true; break; } }
          if (!okh || len > avail - pos) continue;
          chunk_len[k] = pos + len;
          u8* h = hash.data() + 32 * k;
          amh_host_sha256(p + 8, pos - 8, p + pos, (size_t)len, h);
          bad[k] = (h[0] != p[4] || h[1] != p[5] || h[2] != p[6] || h[3] != p[7]) ? 1 : 0;
        }
      });
  }
  void join() { for (auto& t : threads) t.join(); threads.clear(); }
  ~BigSha() { join(); }
};

// The re-derived changes of such a document: every change's SHA-256 is again one serial message, and a change cannot
// be hashed before the changes it depends on (their hashes are part of its bytes).  The host takes the chunks as
// k_chg_header left them, fills in the dep hashes and hashes level by level (the changes of a level in parallel),
// and sends bytes and hashes back; k_chg_hash / k_chg_hash_seq skip these documents.  (change_collector.rs:163-178,
// chunk.rs:181-193)
template <class F> static void host_parallel_for(u32 n, F f, u32 min_items = 128, u32 grain = 32);
static bool big_hash_changes(amh_batch* b, u32 d) {
  amh_ctx* ctx = b->ctx;
  const BatchDev& B = b->B;
  u32 cb2[2];
  DocCounts cnt;
  if (!rt_d2h(cb2, B.chg_base + d, 8, ctx->stream) || !rt_d2h(&cnt, B.cnt + d, sizeof cnt, ctx->stream)) return false;
  const u32 cb = cb2[0], nc = cnt.n_chg;
  if (nc == 0) return true;
  std::vector<u32> out_off(nc + 1), out_len(nc), body_len(nc), ndeps(nc), dep_base(nc), level(nc);
  u32 db2[2];
  if (!rt_d2h(out_off.data(), B.c_out_off + cb, (nc + 1) * 4, ctx->stream) || !rt_d2h(out_len.data(), B.c_out_len + cb, nc * 4, ctx->stream) ||
      !rt_d2h(body_len.data(), B.c_body_len + cb, nc * 4, ctx->stream) || !rt_d2h(ndeps.data(), B.c_ndeps + cb, nc * 4, ctx->stream) ||
      !rt_d2h(dep_base.data(), B.c_dep_base + cb, nc * 4, ctx->stream) || !rt_d2h(level.data(), B.c_level + cb, nc * 4, ctx->stream) ||
      !rt_d2h(db2, B.dep_base + d, 8, ctx->stream))
    return false;
  std::vector<u32> dep_idx(cnt.n_deps + 1);
  if (cnt.n_deps && !rt_d2h(dep_idx.data(), B.dep_idx + db2[0], (size_t)cnt.n_deps * 4, ctx->stream)) return false;
  const size_t lo = out_off[0], hi = out_off[nc];
  if (hi - lo + 1 > ctx->bigc_cap) {
    if (ctx->bigc) rt_host_free(ctx->bigc);
    ctx->bigc_cap = (hi - lo + 1) + (hi - lo + 1) / 4;
    ctx->bigc = (u8*)rt_host_alloc(ctx->bigc_cap);
    if (!ctx->bigc) { ctx->bigc_cap = 0; return false; }
  }
  struct { u8* p; u8* data() const { return p; } } bytes{ctx->bigc};
  std::vector<u8> hashes((size_t)nc * 32);
  if (!rt_d2h(bytes.data(), B.out_arena + lo, hi - lo, ctx->stream)) return false;
  // changes grouped by level (counting sort; file order inside a level)
  u32 max_lvl = 0;
  for (u32 i = 0; i < nc; i++) max_lvl = std::max(max_lvl, level[i]);
  std::vector<u32> start(max_lvl + 2, 0), order(nc);
  for (u32 i = 0; i < nc; i++) start[level[i] + 1]++;
  for (u32 l = 0; l <= max_lvl; l++) start[l + 1] += start[l];
  { std::vector<u32> fill(start.begin(), start.end() - 1); for (u32 i = 0; i < nc; i++) order[fill[level[i]]++] = i; }
  auto one = [&](u32 i) {
    u8* out = bytes.data() + (out_off[i] - lo);
    const u32 hdr = out_len[i] - body_len[i], nd = ndeps[i];
    u32 nb = 1; for (u32 v = nd; v >= 0x80; v >>= 7) nb++;
    u8* deps = out + hdr + nb;
    std::vector<std::array<u8, 32>> hs(nd);
    for (u32 k = 0; k < nd; k++) memcpy(hs[k].data(), hashes.data() + (size_t)dep_idx[dep_base[i] - db2[0] + k] * 32, 32);
    std::sort(hs.begin(), hs.end());
    for (u32 k = 0; k < nd; k++) memcpy(deps + 32 * k, hs[k].data(), 32);
    u8* h = hashes.data() + (size_t)i * 32;
    amh_host_sha256(out + 8, hdr - 8, out + hdr, body_len[i], h);
    memcpy(out + 4, h, 4);
  };
  // One set of threads for all levels: a level's changes are dealt out by a counter, and a thread goes on to the next
  // level once every change of this one is hashed (the hashes of a level are the dependencies of later ones).
  {
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16);
    if (const char* e = getenv("OMP_NUM_THREADS")) { int v = atoi(e); if (v > 0) nt = std::min<unsigned>(nt, (unsigned)v); }
    if (nc < 4) nt = 1;
    nt = std::max(1u, std::min<unsigned>(nt, nc / (max_lvl + 1)));   // a chain: one thread (a level's hand-over costs more than its one hash)
    std::atomic<u32> next{0}, done{0};
    auto work = [&]() {
      for (u32 l = 0; l <= max_lvl; l++) {
        const u32 e = start[l + 1];
        for (;;) {
          u32 k = next.load(std::memory_order_relaxed);
          if (k >= e) break;
          if (!next.compare_exchange_weak(k, k + 1)) continue;
          one(order[k]);
          done.fetch_add(1, std::memory_order_release);
        }
        while (done.load(std::memory_order_acquire) < e) std::this_thread::yield();
      }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < nt; t++) ts.emplace_back(work);
    work();
    for (auto& th : ts) th.join();
  }
  // the hashes go back; the dependency hashes and checksums are written into the device's chunks by k_chg_patch
  return rt_h2d(B.c_hash + (size_t)cb * 32, hashes.data(), (size_t)nc * 32, ctx->stream) && rt_sync(ctx->stream);
}

// merge: the change chunks of a merged document are those of its sources (already verified when they were loaded /
// decoded): where they lie on the device, in history order
struct AdoptHistory { std::vector<u64> src; std::vector<u32> len; };
static amh_batch* load_tail(amh_ctx* ctx, amh_batch* b, struct BigSha& big, const amh_load_opts* opts, bool trailing, bool from_rows, const u8* expect_hash, const AdoptHistory* adopt = nullptr);
// bytes of derived arrays a batch with these totals takes on the device through load (rows: the op columns, the
// change view, the id tables; succ: entries, preds and the delete table; changes: metadata and column lengths), a
// little on the high side
static u64 derived_bytes(u64 rows, u64 succ, u64 chg, u64 deps, u64 map) { return rows * 224 + succ * 72 + chg * 420 + deps * 8 + map * 4; }
// What a batch may need before its largest documents are failed one by one (AMH_DOC_TOO_LARGE): the caller's limit
// (amh_ctx_set_limits), else 90 % of the device's TOTAL memory — a figure that does not move with what this or other
// processes hold at the moment, so the same batch gets the same answer on every run.  A batch under the budget that
// still does not fit what is free right now fails as a whole with AMH_ERR_OOM, as before.
static u64 batch_byte_budget(amh_ctx* ctx) {
  if (ctx->max_batch_bytes) return ctx->max_batch_bytes;
  if (!ctx->total_mem) {   // asked once: the total does not change, and the query is a driver call
    size_t free_b = 0, total_b = 0;
    if (!rt_mem_info(&free_b, &total_b)) return (u64)16 << 30;
    ctx->total_mem = total_b;
  }
  return (u64)ctx->total_mem / 10 * 9;
}
static amh_batch* load_staged_impl(amh_ctx* ctx, const amh_staged* st_c, const amh_load_opts* opts, bool trailing) {
  if (!ctx || !st_c) { set_error(ctx, "invalid argument"); return nullptr; }
  amh_staged* st = const_cast<amh_staged*>(st_c);
  auto* b = new amh_batch();
  b->ctx = ctx;
  b->staged = st;
  st->refs++;
  BatchDev& B = b->B;
  u32 n = st->n_docs;
  PhaseTimer pt("load");
  B.n_docs = n;
  B.check_heads = opts ? (opts->verify_heads != 0) : 1;
  B.max_doc_rows = ctx->max_doc_rows;
  B.skip_doc_sha = st->internal ? 1 : 0;
  B.arena = (const u8*)st->d_arena.p;
  B.doc_off = (const u64*)st->d_doc_off.p;
  bool ok = true;
  BigSha big;
#ifndef AMH_EMU
  big.start(st);
  if (!big.docs.empty()) B.sha_host_min = AMH_HOST_SHA_MIN;
#endif
  auto fail = [&](const std::string& m) -> amh_batch* {
    big.join();
    set_error(ctx, m);
    rt_sync(ctx->side);   // nothing of this batch may still be running when its memory goes back to the pool
    collect_profile(ctx);
    amh_batch_free(b);
    return nullptr;
  };
  B.status = balloc<i32>(b, n + 1, &ok);
  B.fail_line = balloc<u32>(b, n + 1, &ok);
  B.first_len = trailing ? balloc<u32>(b, n + 1, &ok) : nullptr;
  B.desc = balloc<DocDesc>(b, n + 1, &ok);
  B.cnt = balloc<DocCounts>(b, n + 1, &ok);
  u32** bases[8] = {&B.row_base, &B.succ_base, &B.chg_base, &B.dep_base, &B.actor_base, &B.ht_base, &B.dt_base, &B.map_base};
  for (auto pb : bases) *pb = balloc<u32>(b, n + 2, &ok);
  B.doc_hash = balloc<u8>(b, (size_t)n * 32 + 32, &ok);
  B.doc_max_op = balloc<u32>(b, n + 1, &ok);
  B.dn_base = balloc<u32>(b, n + 2, &ok); B.dn_stride = balloc<u32>(b, n + 2, &ok);
  B.ht_mode = getenv("AMH_HT_HASH") ? 1 : 0;
  B.heads_ok = balloc<u32>(b, n + 1, &ok);
  B.doc_level = balloc<u32>(b, n + 1, &ok);
  B.max_level = balloc<u32>(b, 4, &ok);
  if (!ok) return fail("device out of memory");
  rt_memset(B.status, 0, (n + 1) * sizeof(i32), ctx->stream);
  rt_memset(B.fail_line, 0, (n + 1) * sizeof(u32), ctx->stream);
  rt_memset(B.max_level, 0, 16, ctx->stream);
  launch(ctx, AK_PARSE, B, n);
  {   // checksums on the side stream, beside the count pass (both are one long serial walk per lane: few waves each)
    i32* gate = balloc<i32>(b, n + 1, &ok);
    B.sha_bad = balloc<u8>(b, n + 1, &ok);
    if (!ok) return fail("device out of memory");
    B.sha_gate = gate;
#ifdef AMH_EMU
    memcpy(gate, B.status, (size_t)n * sizeof(i32));
#else
    (void)hipMemcpyAsync(gate, B.status, (size_t)n * sizeof(i32), hipMemcpyDeviceToDevice, ctx->stream);
#endif
    rt_memset(B.sha_bad, 0, n + 1, ctx->stream);
    rt_event_record(&ctx->ev_fork, ctx->stream);
    rt_stream_wait(ctx->side, &ctx->ev_fork);
    launch_on(ctx, ctx->side, AK_SHA_DOC, B, n);
    rt_event_record(&ctx->ev_join, ctx->side);
  }
#ifndef AMH_EMU
  {   // documents large enough to hold a column of AMH_BIGCOL_HOST bytes: their big columns are cut into ranges, a wave each
    u64 cand = 0, bytes = 0, segs = 0;
    for (u32 d = 0; d < n; d++) {
      u64 l = st->h_doc_off[2 * (size_t)d + 1] - st->h_doc_off[2 * (size_t)d];
      if (l >= AMH_BIGCOL_HOST) { cand++; bytes += l; segs += l / AMH_SEG_HOST + 13; }
    }
    B.bc_n = nullptr;
    if (cand && !getenv("AMH_NO_BIGCOL") && bytes < 0xF0000000ull && segs < 0x7FFFFFFFull && cand * 13 < 0x7FFFFFFFull) {
      B.bc_cap_cols = (u32)(cand * 13); B.bc_cap_segs = (u32)segs; B.bc_cap_bytes = (u32)bytes;
      B.bc_n = balloc<u32>(b, 4, &ok);
      B.bc_index = balloc<u32>(b, (size_t)n * 13 + 1, &ok);
      u32** cols[] = {&B.bc_doc, &B.bc_slot, &B.bc_seg0, &B.bc_nseg, &B.bc_ebase, &B.bc_bad};
      for (auto pc : cols) *pc = balloc<u32>(b, (size_t)B.bc_cap_cols + 1, &ok);
      B.bc_rows = balloc<u64>(b, (size_t)B.bc_cap_cols + 1, &ok); B.bc_acc = balloc<u64>(b, (size_t)B.bc_cap_cols + 1, &ok);
      u32** sgs[] = {&B.sg_col, &B.sg_first, &B.sg_second, &B.sg_ntok, &B.sg_entry, &B.sg_pos, &B.sg_end, &B.sg_btok};
      for (auto pc : sgs) *pc = balloc<u32>(b, (size_t)B.bc_cap_segs + 1, &ok);
      B.sg_rows = balloc<u64>(b, (size_t)B.bc_cap_segs + 1, &ok); B.sg_acc = balloc<u64>(b, (size_t)B.bc_cap_segs + 1, &ok);
      B.sg_exit = balloc<u32>(b, (size_t)B.bc_cap_bytes + 1, &ok);
      if (!ok) return fail("device out of memory");
      rt_memset(B.bc_n, 0, 4 * sizeof(u32), ctx->stream);
    }
  }
#endif
  launch(ctx, AK_COUNT, B, n);
  rt_stream_wait(ctx->stream, &ctx->ev_join);   // k_extract applies the checksum verdicts
  launch(ctx, AK_EXTRACT, B, n);
  u32 totals[8] = {0};
  for (int attempt = 0;; attempt++) {
    for (int i = 0; i < 8; i++)
      if (!scan_inplace(b, *bases[i], n, &totals[i])) return fail(std::string("scan failed: ") + rt_last_error());
    bool overflow = false;
    for (int i = 0; i < 8; i++) overflow = overflow || totals[i] == 0xFFFFFFFFu;
    const u64 budget = batch_byte_budget(ctx);
    if (!overflow && derived_bytes(totals[0], totals[1], totals[2], totals[3], totals[7]) <= budget) break;
    if (attempt) return fail("batch too large: a derived count exceeds 2^32-1 or the device's memory");
    // The batch does not fit.  Its largest documents are failed (AMH_DOC_TOO_LARGE) until the rest does, and the counts
    // are taken again: a few hostile inputs (a 2^28-row run header is a handful of bytes) must not take the other
    // documents of the batch down with them.
    launch(ctx, AK_EXTRACT, B, n);   // the per-document counts again (the scans replaced them)
    std::vector<u32> per[5];
    u32* src[5] = {B.row_base, B.succ_base, B.chg_base, B.dep_base, B.map_base};
    std::vector<i32> st(n);
    std::vector<u32> fl(n);
    bool cok = rt_d2h(st.data(), B.status, (size_t)n * sizeof(i32), ctx->stream) && rt_d2h(fl.data(), B.fail_line, (size_t)n * sizeof(u32), ctx->stream);
    for (int i = 0; i < 5; i++) { per[i].resize(n); cok = cok && rt_d2h(per[i].data(), src[i], (size_t)n * sizeof(u32), ctx->stream); }
    if (!cok || !rt_sync(ctx->stream)) return fail("D2H copy failed");
    std::vector<std::pair<u64, u32>> weight(n);
    u64 total = 0;
    for (u32 d = 0; d < n; d++) { weight[d] = {derived_bytes(per[0][d], per[1][d], per[2][d], per[3][d], per[4][d]), d}; total += weight[d].first; }
    std::sort(weight.begin(), weight.end(), [](const std::pair<u64, u32>& x, const std::pair<u64, u32>& y) { return x.first > y.first; });
    for (u32 k = 0; k < n && total > budget; k++) {
      total -= weight[k].first;
      st[weight[k].second] = AMH_E_TOO_LARGE; fl[weight[k].second] = __LINE__;
    }
    if (!rt_h2d(B.status, st.data(), (size_t)n * sizeof(i32), ctx->stream) || !rt_h2d(B.fail_line, fl.data(), (size_t)n * sizeof(u32), ctx->stream) || !rt_sync(ctx->stream))
      return fail("H2D copy failed");
    launch(ctx, AK_EXTRACT, B, n);
  }
  pt.mark("load: parse, count, scans");
  B.tot_rows = totals[0]; B.tot_succ = totals[1]; B.tot_chg = totals[2]; B.tot_deps = totals[3];
  B.tot_actors = totals[4]; b->tot_ht = totals[5]; B.tot_dt = totals[6]; b->tot_map = totals[7];
  size_t NA = B.tot_actors + 1, NC = B.tot_chg + 2, NR = B.tot_rows + 1, NS = B.tot_succ + 1;
  B.actor_off = balloc<u32>(b, NA, &ok); B.actor_len = balloc<u32>(b, NA, &ok);
  B.actor_rank = balloc<u32>(b, NA, &ok); B.actor_by_rank = balloc<u32>(b, NA, &ok);
  B.a_first_chg = balloc<u32>(b, NA, &ok); B.a_last_chg = balloc<u32>(b, NA, &ok);
  B.c_actor = balloc<u32>(b, NC, &ok); B.c_seq = balloc<u64>(b, NC, &ok); B.c_max_op = balloc<u64>(b, NC, &ok);
  B.c_time = balloc<i64>(b, NC, &ok); B.c_msg_off = balloc<u32>(b, NC, &ok); B.c_msg_len = balloc<u32>(b, NC, &ok);
  B.c_ndeps = balloc<u32>(b, NC, &ok); B.c_dep_base = balloc<u32>(b, NC, &ok);
  B.c_extra_off = balloc<u32>(b, NC, &ok); B.c_extra_len = balloc<u32>(b, NC, &ok); B.c_doc = balloc<u32>(b, NC, &ok);
  B.dep_idx = balloc<u32>(b, B.tot_deps + 1, &ok);
  B.c_nops = balloc<u32>(b, NC, &ok); B.c_start_op = balloc<u64>(b, NC, &ok); B.c_op_base = balloc<u32>(b, NC, &ok);
  B.c_npred = balloc<u32>(b, NC, &ok);
  B.c_next_same_actor = balloc<u32>(b, NC, &ok); B.c_level = balloc<u32>(b, NC, &ok); B.c_is_dep = balloc<u8>(b, NC, &ok); B.c_hash = balloc<u8>(b, NC * 32, &ok);
  B.c_out_off = balloc<u32>(b, NC, &ok); B.c_out_len = balloc<u32>(b, NC, &ok); B.c_body_len = balloc<u32>(b, NC, &ok);
  B.c_cols_off = balloc<u32>(b, NC, &ok); B.c_col_len = balloc<u32>(b, NC * AMH_NCOLS, &ok); B.c_wide = balloc<u32>(b, NC, &ok);
  B.c_map_base = balloc<u32>(b, NC, &ok); B.c_n_other = balloc<u32>(b, NC, &ok);
  B.chg_actor_map = balloc<u32>(b, (size_t)b->tot_map + 1, &ok);
  u32** rcols[] = {&B.r_id_actor, &B.r_id_ctr, &B.r_obj_actor, &B.r_obj_ctr, &B.r_key_actor, &B.r_key_ctr, &B.r_key_off,
                   &B.r_key_len, &B.r_vraw_off, &B.r_mark_off, &B.r_mark_len, &B.r_succ_cnt, &B.r_succ_base, &B.r_doc, &B.r_inc_cnt};
  for (auto pc : rcols) *pc = balloc<u32>(b, NR, &ok);
  B.r_insert = balloc<u8>(b, NR, &ok); B.r_action = balloc<u8>(b, NR, &ok); B.r_expand = balloc<u8>(b, NR, &ok);
  B.r_visible = balloc<u8>(b, NR, &ok); B.r_winner = balloc<u8>(b, NR, &ok);
  B.r_vmeta = balloc<u64>(b, NR, &ok); B.r_inc_sum = balloc<i64>(b, NR, &ok);
  B.s_actor = balloc<u32>(b, NS, &ok); B.s_ctr = balloc<u32>(b, NS, &ok); B.s_row = balloc<u32>(b, NS, &ok);
  B.s_target = balloc<u32>(b, NS, &ok); B.pred_ids = balloc<u64>(b, NS, &ok);
  if (!ok) return fail("device out of memory");
  pt.mark("load: arrays allocated");
  rt_memset(B.s_row, 0xFF, NS * 4, ctx->stream);
  rt_memset(B.r_doc, 0xFF, NR * 4, ctx->stream);
  // "absent column" defaults of the two string columns (their staged decoder skips columns no document of a wave has);
  // every other column writes its own nulls (amh_segdec.inl)
  rt_memset(B.r_key_off, 0, NR * 4, ctx->stream); rt_memset(B.r_key_len, 0xFF, NR * 4, ctx->stream);
  rt_memset(B.r_mark_off, 0, NR * 4, ctx->stream); rt_memset(B.r_mark_len, 0xFF, NR * 4, ctx->stream);
#ifdef AMH_EMU
  rt_memset(B.r_obj_actor, 0xFF, NR * 4, ctx->stream); rt_memset(B.r_obj_ctr, 0xFF, NR * 4, ctx->stream);
  rt_memset(B.r_key_actor, 0xFF, NR * 4, ctx->stream); rt_memset(B.r_key_ctr, 0xFF, NR * 4, ctx->stream);
  rt_memset(B.r_expand, 0, NR, ctx->stream);
#endif
  rt_memset(B.chg_actor_map, 0, ((size_t)b->tot_map + 1) * 4, ctx->stream);
  launch(ctx, AK_ACTORS, B, n);
  launch(ctx, AK_CHG_INIT, B, B.tot_chg);
  B.cd_err = balloc<u32>(b, (size_t)n * 8 + 1, &ok); B.cd_code = balloc<u32>(b, (size_t)n * 8 + 1, &ok); B.cd_none = balloc<u32>(b, (size_t)n * 8 + 1, &ok);
  if (!ok) return fail("device out of memory");
  launch(ctx, AK_DECODE_CHG_COLS, B, n * 8);
  launch(ctx, AK_DECODE_CHG, B, n);
#ifndef AMH_EMU
  B.fr_n = nullptr;
  if ((u64)B.tot_rows + B.tot_succ >= 8192 && !getenv("AMH_NO_BIGCOL")) {   // runs of 8192 rows and more are filled by the whole grid
    B.fr_cap = (u32)(2 * 13 * (((u64)B.tot_rows + B.tot_succ) / 8192 + 1) + 64);   // runs of >= 8192 rows, in pieces of <= 65536
    B.fr_n = balloc<u32>(b, 4, &ok); B.fr_items = balloc<u32>(b, 8 * (size_t)B.fr_cap, &ok);
    if (!ok) return fail("device out of memory");
    rt_memset(B.fr_n, 0, 16, ctx->stream);
  }
#endif
  launch(ctx, AK_DECODE_OPS, B, n * OC_COUNT);
  pt.mark("load: parse .. decode launched");
  return load_tail(ctx, b, big, opts, trailing, false, nullptr);
}

// The load pipeline from the decoded op rows on: id tables, validation, succ / pred linking, op -> change, the change
// re-encode and hashing, heads, visibility.  Entered after the decode of a document batch (load_staged_impl) or with the
// rows a merge / apply just put in document order (mg_execute, from_rows: nothing is saved and decoded in between; the
// re-derived hashes are then checked against the hashes the changes came with, expect_hash).
static amh_batch* load_tail(amh_ctx* ctx, amh_batch* b, BigSha& big, const amh_load_opts* opts, bool trailing, bool from_rows, const u8* expect_hash, const AdoptHistory* adopt) {
  BatchDev& B = b->B;
  const u32 n = B.n_docs;
  const size_t NC = (size_t)B.tot_chg + 2;
  bool ok = true;
  PhaseTimer pt("load tail");
  auto fail = [&](const std::string& m) -> amh_batch* {
    big.join();
    set_error(ctx, m);
    rt_sync(ctx->side);   // nothing of this batch may still be running when its memory goes back to the pool
    collect_profile(ctx);
    amh_batch_free(b);
    return nullptr;
  };
  {   // id -> row tables, sized now that the highest counter of every document is known
    launch(ctx, AK_HT_SIZE, B, n);
    u32 tot_ht = 0, tot_dn = 0;
    if (!scan_inplace(b, B.ht_base, n, &tot_ht) || !scan_inplace(b, B.dn_base, n, &tot_dn)) return fail("scan failed");
    if (tot_ht == 0xFFFFFFFFu || tot_dn == 0xFFFFFFFFu) return fail("batch too large: id tables exceed 2^32-1 slots");
    b->tot_ht = tot_ht;
    B.ht_key = balloc<u64>(b, (size_t)tot_ht + 1, &ok); B.ht_val = balloc<u32>(b, (size_t)tot_ht + 1, &ok);
    B.dn_val = balloc<u32>(b, (size_t)tot_dn + 1, &ok);
    if (!ok) return fail("device out of memory");
    pt.mark("tail: id tables sized (waits for the decode)");
    rt_memset(B.ht_key, 0, ((size_t)tot_ht + 1) * 8, ctx->stream);
    rt_memset(B.dn_val, 0xFF, ((size_t)tot_dn + 1) * 4, ctx->stream);
    pt.mark("tail: id tables cleared (launched)");
  }
  launch(ctx, from_rows ? AK_ROWS_M : AK_ROWS, B, B.tot_rows);
  launch(ctx, AK_OBJ_ORDER, B, B.tot_rows);
  launch(ctx, AK_SUCC, B, B.tot_succ);
  // the delete table and the op space [0, tot_rows + tot_dt), sized for the succ entries that point at deletes
  launch(ctx, AK_DT_SIZE, B, n);
  if (!scan_inplace(b, B.dt_base, n, &B.tot_dt)) return fail("scan failed");
  if (B.tot_dt == 0xFFFFFFFFu || (u64)B.tot_rows + B.tot_dt >= 0x7FFFFFFFull) return fail("batch too large: more than 2^31 ops");
  const u32 T = B.tot_rows + B.tot_dt;
  B.dt_key = balloc<u64>(b, (size_t)B.tot_dt + 1, &ok); B.d_doc = balloc<u32>(b, (size_t)B.tot_dt + 1, &ok);
  B.d_first_pred_row = balloc<u32>(b, (size_t)B.tot_dt + 1, &ok);
  B.t_pred_cnt = balloc<u32>(b, (size_t)T + 2, &ok); B.t_pred_base = balloc<u32>(b, (size_t)T + 2, &ok);
  B.t_pred_fill = balloc<u32>(b, (size_t)T + 2, &ok); B.t_chg = balloc<u32>(b, (size_t)T + 2, &ok);
  if (!ok) return fail("device out of memory");
  pt.mark("tail: rows .. delete table sized");
  rt_memset(B.dt_key, 0, ((size_t)B.tot_dt + 1) * 8, ctx->stream);
  rt_memset(B.d_first_pred_row, 0xFF, ((size_t)B.tot_dt + 1) * 4, ctx->stream);
  rt_memset(B.t_pred_cnt, 0, ((size_t)T + 2) * 4, ctx->stream);
  rt_memset(B.t_pred_fill, 0, ((size_t)T + 2) * 4, ctx->stream);
  pt.mark("tail: delete table cleared (launched)");
  launch(ctx, AK_SUCC_LINK, B, B.tot_succ);
  B.a_tab_base = balloc<u32>(b, (size_t)B.tot_actors + 2, &ok); B.a_tab = balloc<u32>(b, (size_t)B.tot_chg + 2, &ok);
  B.a_tab_bn = balloc<u64>(b, (size_t)B.tot_actors + 2, &ok); B.a_tab_max = balloc<u32>(b, (size_t)B.tot_chg + 2, &ok);
  if (!ok) return fail("device out of memory");
  launch(ctx, AK_ATAB_BASE, B, n);
  launch(ctx, AK_ATAB_FILL, B, B.tot_chg);
  launch(ctx, AK_BUCKET, B, T);
  launch(ctx, AK_CHG_FINISH, B, B.tot_chg);
  // ops grouped by change
#ifdef AMH_EMU
  memcpy(B.c_op_base, B.c_nops, (size_t)B.tot_chg * 4);
#else
  (void)hipMemcpyAsync(B.c_op_base, B.c_nops, (size_t)B.tot_chg * 4, hipMemcpyDeviceToDevice, ctx->stream);
#endif
  u32 tot_ops = 0;
  if (!scan_inplace(b, B.c_op_base, B.tot_chg, &tot_ops)) return fail("scan failed");
  if (tot_ops == 0xFFFFFFFFu) return fail("batch too large");
  B.tot_ops = tot_ops;
  if (adopt) {
    // The history of a merged document = the verified change chunks of its sources, in the merged order: gathered into
    // the batch's out_arena (as the reference's merge keeps the stored changes); no op -> change-order view, re-encode or
    // hashing.  AMH_VERIFY_MERGE=1 takes the other branch (every change re-derived from the merged rows and compared).
    const u32 nc = B.tot_chg;
    std::vector<u32> off((size_t)nc + 1, 0);
    u64 run = 0;
    for (u32 c = 0; c < nc; c++) { off[c] = (u32)run; run += adopt->len[c]; if (run >= 0xFFFFFFF0ull) return fail("batch too large: change chunks exceed 4 GiB"); }
    off[nc] = (u32)run;
    b->tot_out = (u32)run;
    B.out_arena = balloc<u8>(b, (size_t)run + 256, &ok);
    u64* d_src = balloc<u64>(b, (size_t)nc + 1, &ok);
    if (!ok) return fail("device out of memory");
    if ((nc && (!rt_h2d(d_src, adopt->src.data(), (size_t)nc * sizeof(u64), ctx->stream) || !rt_h2d(B.c_out_len, adopt->len.data(), (size_t)nc * sizeof(u32), ctx->stream))) ||
        !rt_h2d(B.c_out_off, off.data(), ((size_t)nc + 1) * sizeof(u32), ctx->stream))
      return fail("H2D copy failed");
    launch(ctx, AK_VISIBLE, B, B.tot_rows);
    rt_event_record(&ctx->ev_join, ctx->stream);
    amh_copy_bytes(d_src, B.c_out_len, B.c_out_off, B.out_arena, nc, (void*)ctx->stream);
  } else {
  B.ops_sorted = balloc<u32>(b, (size_t)tot_ops + 1, &ok);
  if (!ok) return fail("device out of memory");
  rt_memset(B.ops_sorted, 0xFF, ((size_t)tot_ops + 1) * 4, ctx->stream);
  launch(ctx, AK_PLACE, B, T);
  B.j_pred_base = balloc<u32>(b, (size_t)tot_ops + 2, &ok);
  if (!ok) return fail("device out of memory");
  rt_memset(B.t_pred_base, 0, ((size_t)T + 2) * 4, ctx->stream);
  launch(ctx, AK_PRED_SLOT_CNT, B, tot_ops);
  if (!scan_inplace(b, B.j_pred_base, tot_ops, nullptr)) return fail("scan failed");
  launch(ctx, AK_PRED_SLOT_BASE, B, tot_ops);
  launch(ctx, AK_PRED_FILL, B, B.tot_succ);
  B.big_cap = B.tot_succ / AMH_BIG_LIST + 8;
  B.big_n = balloc<u32>(b, 4, &ok); B.big_items = balloc<u32>(b, 2 * (size_t)B.big_cap, &ok);
  if (!ok) return fail("device out of memory");
  rt_memset(B.big_n, 0, 4 * sizeof(u32), ctx->stream);
  launch(ctx, AK_PRED_SORT, B, T);   // + the actors each op refers to (k_actor_mark), in the same pass over the pred lists
  launch(ctx, AK_ACTOR_RANK, B, B.tot_chg);
  SaveBigHost ebh;   // changes of tens of thousands of ops: their columns are cut into ranges, a wave each (g_en_big)
  if (!enc_big_prepare(ctx, b, B, ebh)) return fail("device pipeline failed: job tables of the change re-encode");
  for (int k = 0; k < 12; k++) B.cv[k] = nullptr;   // one interleaved array per column group: obj (2 fields), key (4), misc (2), value (4)
  B.enc_tmp = nullptr;
  bool need_view = true;
#ifndef AMH_EMU
  if (amh_encode_one_pass(B)) {
    // changes of a few hundred ops and more: one walk per change straight from the op rows into per-column slots
    // (amh_segenc1.inl); the change view is only built for changes whose columns are cut into ranges
    need_view = B.sv_nj != 0;
    B.enc_tmp = balloc<u8>(b, (size_t)amh_encode_tmp_bytes(B), &ok);
    if (!ok) return fail("device out of memory");
  }
#endif
  rt_memset(B.c_wide, 0, NC * 4, ctx->stream);
  if (need_view) {
    B.cv[0] = balloc<u32>(b, 2 * ((size_t)tot_ops + 1), &ok); B.cv[2] = balloc<u32>(b, 4 * ((size_t)tot_ops + 1), &ok);
    B.cv[6] = balloc<u32>(b, 2 * ((size_t)tot_ops + 1), &ok); B.cv[8] = balloc<u32>(b, 4 * ((size_t)tot_ops + 1), &ok);
    if (!ok) return fail("device out of memory");
    launch(ctx, AK_CHANGE_VIEW, B, tot_ops);
  }
  auto launch_visible = [&]() {
    if (getenv("AMH_VISIBLE_MAIN")) {
      launch(ctx, AK_VISIBLE, B, B.tot_rows);
      rt_event_record(&ctx->ev_join, ctx->stream);
    } else {
      rt_event_record(&ctx->ev_fork2, ctx->stream);
      rt_stream_wait(ctx->side, &ctx->ev_fork2);
      launch_on(ctx, ctx->side, AK_VISIBLE, B, B.tot_rows);
      rt_event_record(&ctx->ev_join, ctx->side);
    }
  };
  // batches of small changes: the lane-per-change encoders are bound by instruction issue and leave the memory system
  // idle, so visibility runs beside them (as in round 2); the one-pass encoder is bound by the memory system itself:
  // there visibility runs beside the hashing instead (below)
  const bool visible_early = B.enc_tmp == nullptr;
  if (visible_early) launch_visible();
  launch(ctx, AK_ENCODE_SIZE, B, B.tot_chg * AMH_NCOLS);
  save_big_launch(ctx, B, 2);
  launch(ctx, AK_CHG_LAYOUT, B, B.tot_chg);
#ifdef AMH_EMU
  memcpy(B.c_out_off, B.c_out_len, (size_t)B.tot_chg * 4);
#else
  (void)hipMemcpyAsync(B.c_out_off, B.c_out_len, (size_t)B.tot_chg * 4, hipMemcpyDeviceToDevice, ctx->stream);
#endif
  u32 tot_out = 0;
  if (!scan_inplace(b, B.c_out_off, B.tot_chg, &tot_out)) return fail("scan failed");
  if (tot_out == 0xFFFFFFFFu) return fail("batch too large: re-derived changes exceed 4 GiB");
  b->tot_out = tot_out;
  B.out_arena = balloc<u8>(b, (size_t)tot_out + 256, &ok);
  if (!ok) return fail("device out of memory");
  launch(ctx, AK_ENCODE_STORE, B, B.tot_chg * AMH_NCOLS);
  save_big_launch(ctx, B, 3);
  B.sv_nj = 0; B.sv_nbc = 0; B.sv_big = nullptr;   // (the tables were this pass's)
  launch(ctx, AK_CHG_HEADER, B, B.tot_chg);
  // Visibility only needs the linked rows.  It runs on the side stream beside the hashing of the changes: one launch per
  // dependency level, a lane per change — a chain of launches that each occupy a fraction of the SIMDs and leave the memory
  // system idle (round 2 ran it beside the change encode; the one-pass encoder is bound by the memory system itself and
  // the pair then took as long as one after the other).  AMH_VISIBLE_MAIN=1: on the main stream, for its standalone time.
  if (!visible_early) launch_visible();
  // hash the changes level by level: every change of a level only needs hashes of earlier levels
  B.lvl_order = balloc<u32>(b, (size_t)B.tot_chg + 1, &ok);
  B.lvl_start = balloc<u32>(b, AMH_LEVEL_CAP_HOST + 4, &ok);
  u32* lvl_tmp = balloc<u32>(b, amh_level_sort_tmp_words(B.tot_chg), &ok);
  if (!ok) return fail("device out of memory");
#ifndef AMH_EMU
  B.host_chain_min = AMH_HOST_CHAIN_MIN;
#endif
  amh_level_sort(B, lvl_tmp, (void*)ctx->stream);
  u32 lvl_start[AMH_LEVEL_CAP_HOST + 2] = {0};
  if (!rt_d2h(lvl_start, B.lvl_start, sizeof(lvl_start), ctx->stream)) return fail("device pipeline failed");
  for (u32 lvl = 0; lvl < AMH_LEVEL_CAP_HOST; lvl++) {
    if (lvl_start[lvl + 1] == lvl_start[lvl]) continue;
    B.cur_level = lvl_start[lvl];
    launch(ctx, AK_CHG_HASH, B, lvl_start[lvl + 1] - lvl_start[lvl]);
  }
  if (lvl_start[AMH_LEVEL_CAP_HOST + 1] > lvl_start[AMH_LEVEL_CAP_HOST]) launch(ctx, AK_CHG_HASH_SEQ, B, n);
  pt.mark("tail: rows .. change hashes launched");
  if (B.sha_host_min) {   // large documents: their changes are hashed by the host
    big.join();
    pt.mark("tail: wait for the host's document checksums");
    for (size_t k = 0; k < big.docs.size(); k++) {
      if (big.chunk_len[k] < AMH_HOST_SHA_MIN) continue;
      i32 stt = 0;
      if (!rt_d2h(&stt, B.status + big.docs[k], sizeof stt, ctx->stream)) return fail("device pipeline failed");
      if (stt == 0 && !big_hash_changes(b, big.docs[k])) return fail(std::string("device pipeline failed: ") + rt_last_error());
    }
  }
  bool chain_docs = false;
  if (B.host_chain_min && lvl_start[AMH_LEVEL_CAP_HOST + 1] > lvl_start[AMH_LEVEL_CAP_HOST]) {
    // documents with a long chain of dependency levels (a text typed one change per keystroke): the host walks the chain
    std::vector<u32> lvl(n);
    std::vector<DocCounts> cn(n);
    std::vector<i32> stt(n);
    if (!rt_d2h(lvl.data(), B.doc_level, (size_t)n * sizeof(u32), ctx->stream) || !rt_d2h(cn.data(), B.cnt, (size_t)n * sizeof(DocCounts), ctx->stream) ||
        !rt_d2h(stt.data(), B.status, (size_t)n * sizeof(i32), ctx->stream))
      return fail("device pipeline failed");
    for (u32 d = 0; d < n; d++) {
      if (stt[d] || lvl[d] < AMH_LEVEL_CAP_HOST || cn[d].n_chg < B.host_chain_min) continue;
      bool done_above = false;   // a large chunk: hashed in the loop above already
      if (B.sha_host_min) for (size_t k = 0; k < big.docs.size(); k++) done_above = done_above || (big.docs[k] == d && big.chunk_len[k] >= AMH_HOST_SHA_MIN);
      if (done_above) continue;
      chain_docs = true;
      if (!big_hash_changes(b, d)) return fail(std::string("device pipeline failed: ") + rt_last_error());
    }
  }
  if ((B.sha_host_min && !big.docs.empty()) || chain_docs) launch(ctx, AK_CHG_PATCH, B, B.tot_chg);
  pt.mark("tail: host change hashes");
  if (expect_hash) { B.expect_hash = expect_hash; launch(ctx, AK_VERIFY_HASH, B, B.tot_chg); }
  }
  launch(ctx, AK_HEADS, B, n);
  rt_stream_wait(ctx->stream, &ctx->ev_join);   // visibility is in
  b->status.resize(n);
  b->fail_line.resize(n);
  if (!rt_d2h(b->status.data(), B.status, n * sizeof(i32), ctx->stream) || !rt_d2h(b->fail_line.data(), B.fail_line, n * sizeof(u32), ctx->stream) || !rt_sync(ctx->stream))
    return fail(std::string("device pipeline failed: ") + rt_last_error());
  pt.mark("tail: heads, statuses back");
  // the host-side checksums of the large chunks: a mismatch outranks whatever the pipeline made of the damaged bytes
  big.join();
  for (size_t k = 0; k < big.docs.size(); k++) {
    const u32 d = big.docs[k];
    if (b->status[d] == AMH_E_PARSE) continue;   // the chunk header itself did not parse: that comes first
    rt_h2d(B.doc_hash + (size_t)d * 32, big.hash.data() + 32 * k, 32, ctx->stream);
    if (big.bad[k]) {
      b->status[d] = AMH_E_BAD_CHECKSUM;
      b->fail_line[d] = __LINE__;
      rt_h2d(B.status + d, &b->status[d], sizeof(i32), ctx->stream);
    }
  }
  if (!big.docs.empty() && !rt_sync(ctx->stream)) return fail("device pipeline failed");
  B.host_chain_min = 0;
  B.sha_host_min = 0;   // a property of this load, not of the batch (merge / save run the kernels with settings of their own)
  B.skip_doc_sha = 0;
  collect_profile(ctx);
  // change chunks: a first chunk that is a change, and everything behind the first chunk (automerge.rs:657-695)
  if (trailing && load_trailing_chunks(ctx, b, opts) != AMH_SUCCESS) { amh_batch_free(b); return nullptr; }
  return b;
}

extern "C" amh_batch* amh_load_batch(amh_ctx* ctx, const uint8_t* const* docs, const size_t* lens, size_t n, const amh_load_opts* opts) {
  amh_staged* st = amh_stage_batch(ctx, docs, lens, n);
  if (!st) return nullptr;
  amh_batch* b = amh_load_staged(ctx, st, opts);
  staged_unref(st);
  return b;
}

// ---------------------------------------------------------------------------------------------
// host views for the accessors
// ---------------------------------------------------------------------------------------------
template <class V, class T>
static bool fetch(const amh_batch* b, V& v, const T* dev, size_t n) {
  v.resize(n);
  return d2h_large(b->ctx, v.data(), dev, n * sizeof(T), (size_t)1 << 20);   // the history of a wide batch is tens of MB
}
static bool ensure_base(const amh_batch* b) {
  HostView& hv = b->hv;
  if (hv.base_ok) return true;
  const BatchDev& B = b->B;
  u32 n = B.n_docs;
  bool ok = fetch(b, hv.cnt, B.cnt, n) && fetch(b, hv.desc, B.desc, n) && fetch(b, hv.row_base, B.row_base, n + 1) &&
            fetch(b, hv.succ_base, B.succ_base, n + 1) && fetch(b, hv.chg_base, B.chg_base, n + 1) &&
            fetch(b, hv.dep_base, B.dep_base, n + 1) && fetch(b, hv.actor_base, B.actor_base, n + 1) &&
            fetch(b, hv.doc_max_op, B.doc_max_op, n);
  hv.base_ok = ok;
  return ok;
}
static bool ensure_chg(const amh_batch* b) {
  if (!ensure_base(b)) return false;
  HostView& hv = b->hv;
  if (hv.chg_ok) return true;
  const BatchDev& B = b->B;
  size_t nc = B.tot_chg;
  bool ok = fetch(b, hv.c_actor, B.c_actor, nc) && fetch(b, hv.c_nops, B.c_nops, nc) && fetch(b, hv.c_ndeps, B.c_ndeps, nc) &&
            fetch(b, hv.c_dep_base, B.c_dep_base, nc) && fetch(b, hv.c_out_off, B.c_out_off, nc + 1) &&
            fetch(b, hv.c_out_len, B.c_out_len, nc) && fetch(b, hv.dep_idx, B.dep_idx, (size_t)B.tot_deps) &&
            fetch(b, hv.c_seq, B.c_seq, nc) && fetch(b, hv.c_start_op, B.c_start_op, nc) && fetch(b, hv.c_time, B.c_time, nc) &&
            fetch(b, hv.c_hash, B.c_hash, nc * 32) && fetch(b, hv.actor_off, B.actor_off, (size_t)B.tot_actors) &&
            fetch(b, hv.actor_len, B.actor_len, (size_t)B.tot_actors) && fetch(b, hv.c_doc, B.c_doc, nc);
  hv.chg_ok = ok;
  return ok;
}

// main thread only (allocates from the pool)
static bool ensure_actor_pack(amh_batch* b) {
  HostView& hv = b->hv;
  if (hv.pack_ok) return true;
  amh_ctx* ctx = b->ctx;
  const BatchDev& B = b->B;
  const u32 n = B.n_docs;
  bool ok = true;
  CallScratch scratch(b);
  u32* d_len = balloc<u32>(b, (size_t)n + 2, &ok); u32* d_off = balloc<u32>(b, (size_t)n + 2, &ok);
  if (!ok) return false;
  amh_actor_pack_len(B, d_len, (void*)ctx->stream);
  if (!rt_d2d(d_off, d_len, (size_t)n * sizeof(u32), ctx->stream)) return false;
  u32 total = 0;
  if (!scan_inplace(b, d_off, n, &total) || total == 0xFFFFFFFFu) return false;
  u8* d_pack = balloc<u8>(b, (size_t)total + 16, &ok);
  if (!ok) return false;
  amh_actor_pack_copy(B, d_len, d_off, d_pack, (void*)ctx->stream);
  hv.actor_pack.resize((size_t)total + 16);
  hv.actor_pack_off.resize((size_t)n + 1);
  if (!rt_d2h(hv.actor_pack.data(), d_pack, total, ctx->stream) || !rt_d2h(hv.actor_pack_off.data(), d_off, ((size_t)n + 1) * sizeof(u32), ctx->stream)) return false;
  hv.pack_ok = true;
  return true;
}

static const char* status_msg(int s) {
  switch (s) {
    case AMH_OK: return "ok";
    case AMH_E_PARSE: return "unable to parse chunk";
    case AMH_E_BAD_CHECKSUM: return "bad checksum";
    case AMH_E_INVALID_CHANGE_COLUMNS: return "invalid change columns";
    case AMH_E_READ_OP: return "error inflating document chunk ops: error reading operation";
    case AMH_E_OPS_OUT_OF_ORDER: return "error inflating document chunk ops: the document contained ops which were out of order";
    case AMH_E_MISSING_ACTOR: return "error inflating document chunk ops: an operation referenced a missing actor id";
    case AMH_E_MISMATCHING_HEADS: return "error inflating document chunk ops: mismatching heads";
    case AMH_E_MISSING_OPS: return "error inflating document chunk ops: missing operations";
    case AMH_E_SUCC_OUT_OF_ORDER: return "error inflating document chunk ops: succ out of order";
    case AMH_E_INVALID_OP: return "error inflating document chunk ops: invalid op type";
    case AMH_E_CHANGES_OUT_OF_ORDER: return "error inflating document chunk ops: invalid changes: changes out of order";
    case AMH_E_MISSING_CHANGE: return "error inflating document chunk ops: invalid changes: missing change";
    case AMH_E_READ_CHANGE: return "error inflating document chunk ops: invalid changes: unable to read change metadata";
    case AMH_E_INCORRECT_MAX_OP: return "error inflating document chunk ops: invalid changes: incorrect max op";
    case AMH_E_COLLECT_MISSING_ACTOR: return "error inflating document chunk ops: invalid changes: a change referenced an actor index we couldn't find";
    case AMH_E_COLLECT_MISSING_OPS: return "error inflating document chunk ops: invalid changes: missing ops";
    case AMH_E_MISSING_DEPS: return "missing dependencies";
    case AMH_E_DUPLICATE_SEQ: return "duplicate seq number";
    case AMH_E_TOO_LARGE: return "document exceeds the engine's size limits";
    case AMH_E_UNSUPPORTED: return "document shape not handled by this round's device path (change chunks / apply_changes)";
    default: return "unknown";
  }
}

extern "C" int amh_doc_status(const amh_batch* b, uint32_t doc, int* err_kind, const char** msg) {
  if (!b || doc >= b->B.n_docs) return AMH_ERR_INVALID_ARG;
  if (err_kind) *err_kind = b->status[doc];
  if (msg) {
    b->msg_buf = status_msg(b->status[doc]);
    if (b->status[doc]) b->msg_buf += " [check at amh_kernels.inl:" + std::to_string(b->fail_line[doc]) + "]";
    *msg = b->msg_buf.c_str();
  }
  return AMH_SUCCESS;
}

extern "C" int amh_batch_get_stats(const amh_batch* b, amh_batch_stats* out) {
  if (!b || !out) return AMH_ERR_INVALID_ARG;
  if (!ensure_chg(b)) return AMH_ERR_DEVICE;
  memset(out, 0, sizeof(*out));
  const HostView& hv = b->hv;
  out->n_docs = b->B.n_docs;
  for (u32 d = 0; d < b->B.n_docs; d++) {
    if (b->status[d]) continue;
    out->n_ok++;
    out->op_rows += hv.cnt[d].n_rows;
    out->succ_entries += hv.cnt[d].n_succ;
    out->changes += hv.cnt[d].n_chg;
    for (u32 c = hv.chg_base[d]; c < hv.chg_base[d] + hv.cnt[d].n_chg; c++) {
      out->ops_loaded += hv.c_nops[c];
      out->change_bytes += hv.c_out_len[c];
    }
  }
  out->delete_ops = out->ops_loaded - out->op_rows;
  for (u32 d = 0; d < b->B.n_docs; d++) out->encoded_bytes += b->staged->h_doc_off[2 * (size_t)d + 1] - b->staged->h_doc_off[2 * (size_t)d];
  return AMH_SUCCESS;
}

extern "C" int amh_doc_max_op(const amh_batch* b, uint32_t doc, uint64_t* max_op) {
  if (!b || doc >= b->B.n_docs || !max_op) return AMH_ERR_INVALID_ARG;
  if (b->status[doc]) return AMH_ERR_DOC_FAILED;
  if (!ensure_base(b)) return AMH_ERR_DEVICE;
  *max_op = b->hv.doc_max_op[doc];
  return AMH_SUCCESS;
}

extern "C" int amh_heads(const amh_batch* b, uint32_t doc, const uint8_t** hashes32, size_t* n) {
  if (!b || doc >= b->B.n_docs || !hashes32 || !n) return AMH_ERR_INVALID_ARG;
  if (b->status[doc]) return AMH_ERR_DOC_FAILED;
  if (!ensure_chg(b)) return AMH_ERR_DEVICE;
  HostView& hv = b->hv;
  auto it = hv.heads_cache.find(doc);
  if (it == hv.heads_cache.end()) {
    u32 cb = hv.chg_base[doc], nc = hv.cnt[doc].n_chg;
    std::vector<uint8_t> is_dep(nc, 0);
    for (u32 i = 0; i < nc; i++)
      for (u32 k = 0; k < hv.c_ndeps[cb + i]; k++) is_dep[hv.dep_idx[hv.c_dep_base[cb + i] + k]] = 1;
    std::vector<std::array<uint8_t, 32>> hs;
    for (u32 i = 0; i < nc; i++)
      if (!is_dep[i]) {
        std::array<uint8_t, 32> h;
        memcpy(h.data(), hv.c_hash.data() + (size_t)(cb + i) * 32, 32);
        hs.push_back(h);
      }
    std::sort(hs.begin(), hs.end());
    hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
    std::vector<u8> flat;
    for (auto& h : hs) flat.insert(flat.end(), h.begin(), h.end());
    it = hv.heads_cache.emplace(doc, std::move(flat)).first;
  }
  *hashes32 = it->second.data();
  *n = it->second.size() / 32;
  return AMH_SUCCESS;
}

extern "C" int amh_changes(const amh_batch* b, uint32_t doc, amh_change_view* out, size_t* n) {
  if (!b || doc >= b->B.n_docs || !n) return AMH_ERR_INVALID_ARG;
  if (b->status[doc]) return AMH_ERR_DOC_FAILED;
  if (!ensure_chg(b)) return AMH_ERR_DEVICE;
  HostView& hv = b->hv;
  if (!hv.out_ok) {
    if (!fetch(b, hv.out_arena, b->B.out_arena, (size_t)b->tot_out)) return AMH_ERR_DEVICE;
    hv.out_ok = true;
  }
  u32 cb = hv.chg_base[doc], nc = hv.cnt[doc].n_chg;
  size_t cap = *n;
  *n = nc;
  if (!out || cap < nc) return cap == 0 ? AMH_SUCCESS : AMH_ERR_INVALID_ARG;
  if (!ensure_host_arena(b->staged)) return AMH_ERR_DEVICE;
  const uint8_t* chunk = b->staged->h_arena.data() + b->staged->h_doc_off[2 * (size_t)doc];
  u32 ab = hv.actor_base[doc];
  for (u32 i = 0; i < nc; i++) {
    u32 c = cb + i;
    out[i].hash = hv.c_hash.data() + (size_t)c * 32;
    out[i].raw_bytes = hv.out_arena.data() + hv.c_out_off[c];
    out[i].raw_len = hv.c_out_len[c];
    out[i].actor = chunk + hv.actor_off[ab + hv.c_actor[c]];
    out[i].actor_len = hv.actor_len[ab + hv.c_actor[c]];
    out[i].seq = hv.c_seq[c];
    out[i].start_op = hv.c_start_op[c];
    out[i].num_ops = hv.c_nops[c];
    out[i].timestamp = hv.c_time[c];
    out[i].num_deps = hv.c_ndeps[c];
  }
  return AMH_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// canonical materialization: formatting only — order, visibility and counter values come from the
// device arrays (r_visible, r_inc_sum, document order)
// ---------------------------------------------------------------------------------------------
static std::string hexs(const uint8_t* p, size_t n) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
  return s;
}
static void json_str(std::string& out, const uint8_t* p, size_t n) {
  out.push_back('"');
  for (size_t i = 0; i < n; i++) {
    uint8_t c = p[i];
    if (c == '"') out += "\\\"";
    else if (c == '\\') out += "\\\\";
    else if (c < 0x20) { char bf[8]; snprintf(bf, sizeof bf, "\\u%04x", c); out += bf; }
    else out.push_back((char)c);
  }
  out.push_back('"');
}
struct DocRows {
  u32 n = 0;
  std::vector<u32> id_actor, id_ctr, obj_actor, obj_ctr, key_actor, key_ctr, key_off, key_len, vraw_off, succ_cnt;
  std::vector<u8> insert, action, visible;
  std::vector<u64> vmeta;
  std::vector<i64> inc_sum;
  const uint8_t* chunk = nullptr;
  const DocDesc* D = nullptr;
  std::vector<std::string> actor_hex;
  std::map<u64, std::pair<u32, u32>> obj_range;  // object id -> [start, end) rows
};
static bool same_key_host(const DocRows& R, u32 a, u32 b) {
  bool ap = R.key_len[a] != AMH_NULL32 && !R.insert[a], bp = R.key_len[b] != AMH_NULL32 && !R.insert[b];
  if (ap != bp) return false;
  if (ap) return R.key_len[a] == R.key_len[b] && memcmp(R.chunk + R.key_off[a], R.chunk + R.key_off[b], R.key_len[a]) == 0;
  u32 aa = R.insert[a] ? R.id_actor[a] : R.key_actor[a], ac = R.insert[a] ? R.id_ctr[a] : R.key_ctr[a];
  u32 ba = R.insert[b] ? R.id_actor[b] : R.key_actor[b], bc = R.insert[b] ? R.id_ctr[b] : R.key_ctr[b];
  return aa == ba && ac == bc;
}
static i64 host_sleb(const uint8_t* p, size_t n, bool is_unsigned) {
  u64 res = 0;
  unsigned shift = 0;
  uint8_t b = 0;
  for (size_t i = 0; i < n; i++) {
    b = p[i];
    if (shift < 64) res |= (u64)(b & 0x7f) << shift;
    shift += 7;
  }
  if (!is_unsigned && shift < 64 && (b & 0x40)) res |= ~(u64)0 << shift;
  return (i64)res;
}
// one scalar value, or the opening of a nested object (returns true: the caller descends into it)
static bool hydrate_value_open(const DocRows& R, u32 r, std::string& out) {
  out += "{\"id\":\"" + std::to_string(R.id_ctr[r]) + "@" + R.actor_hex[R.id_actor[r]] + "\",";
  u8 a = R.action[r];
  if (a == 0 || a == 2 || a == 4 || a == 6) {
    const char* names[] = {"map", "", "list", "", "text", "", "table"};
    out += std::string("\"") + names[a] + "\":";
    return true;
  }
  u64 m = R.vmeta[r];
  u32 t = (u32)(m & 15);
  size_t l = (size_t)(m >> 4);
  const uint8_t* raw = R.chunk + R.D->ocol[OC_VAL_RAW].off + R.vraw_off[r];
  switch (t) {
    case 0: out += "\"null\":null"; break;
    case 1: out += "\"bool\":false"; break;
    case 2: out += "\"bool\":true"; break;
    case 3: out += "\"uint\":" + std::to_string((u64)host_sleb(raw, l, true)); break;
    case 4: out += "\"int\":" + std::to_string(host_sleb(raw, l, false)); break;
    case 5: out += "\"f64\":\"" + hexs(raw, l) + "\""; break;
    case 6: out += "\"str\":"; json_str(out, raw, l); break;
    case 7: out += "\"bytes\":\"" + hexs(raw, l) + "\""; break;
    case 8: out += "\"counter\":" + std::to_string(host_sleb(raw, l, false) + R.inc_sum[r]); break;
    case 9: out += "\"ts\":" + std::to_string(host_sleb(raw, l, false)); break;
    default: out += "\"unknown\":[" + std::to_string(t) + ",\"" + hexs(raw, l) + "\"]"; break;
  }
  out += "}";
  return false;
}
// The materialized state of an object and everything below it (hydrate.rs:128-167).  An explicit stack of objects being
// written instead of recursion: nesting has no depth limit in the reference, and must not have one here.
static void hydrate_obj(const DocRows& R, u64 root_obj, u8 root_type, std::string& out) {
  struct Frame { u32 i, e; bool is_map, first, in_group, close_value; std::vector<u32> vis; size_t k; };
  std::vector<Frame> stack;
  auto open = [&](u64 obj, u8 objtype, bool close_value) {
    auto it = R.obj_range.find(obj);
    u32 s = 0, e = 0;
    if (it != R.obj_range.end()) { s = it->second.first; e = it->second.second; }
    if (objtype == 4) {   // text: the visible characters in order, no nesting
      std::string text;
      u32 i = s;
      while (i < e) {
        u32 j = i;
        int last = -1;
        while (j < e && same_key_host(R, j, i)) { if (R.visible[j]) last = (int)j; j++; }
        if (last >= 0) {
          if (R.action[last] == 1 && (R.vmeta[last] & 15) == 6)
            text.append((const char*)R.chunk + R.D->ocol[OC_VAL_RAW].off + R.vraw_off[last], (size_t)(R.vmeta[last] >> 4));
          else text += "\xef\xbf\xbc";
        }
        i = j;
      }
      json_str(out, (const uint8_t*)text.data(), text.size());
      if (close_value) out += "}";
      return;
    }
    Frame f;
    f.i = s; f.e = e; f.is_map = objtype == 0 || objtype == 6; f.first = true; f.in_group = false; f.close_value = close_value; f.k = 0;
    out += f.is_map ? "{" : "[";
    stack.push_back(std::move(f));
  };
  open(root_obj, root_type, false);
  while (!stack.empty()) {
    Frame& f = stack.back();
    if (f.in_group) {
      if (f.k < f.vis.size()) {
        u32 r = f.vis[f.k++];
        if (f.k > 1) out += ",";
        if (hydrate_value_open(R, r, out)) open(((u64)R.id_ctr[r] << 32) | R.id_actor[r], R.action[r], true);   // may reallocate the stack: f is not used below
      } else {
        out += "]";
        f.in_group = false;
      }
      continue;
    }
    if (f.i < f.e) {
      u32 i = f.i, j = i;
      f.vis.clear();
      while (j < f.e && same_key_host(R, j, i)) { if (R.visible[j]) f.vis.push_back(j); j++; }
      f.i = j;
      if (!f.vis.empty()) {
        if (!f.first) out += ",";
        f.first = false;
        if (f.is_map) { json_str(out, R.chunk + R.key_off[i], R.key_len[i]); out += ":"; }
        out += "[";
        f.in_group = true; f.k = 0;
      }
      continue;
    }
    out += f.is_map ? "}" : "]";
    if (f.close_value) out += "}";
    stack.pop_back();
  }
}

extern "C" int amh_hydrate_json(const amh_batch* b, uint32_t doc, amh_bytes_out* out) {
  if (!b || doc >= b->B.n_docs || !out) return AMH_ERR_INVALID_ARG;
  if (b->status[doc]) return AMH_ERR_DOC_FAILED;
  AMH_NEEDS_OPS(b, "amh_hydrate_json")
  if (!ensure_chg(b)) return AMH_ERR_DEVICE;
  const HostView& hv = b->hv;
  const BatchDev& B = b->B;
  DocRows R;
  R.n = hv.cnt[doc].n_rows;
  u32 rb = hv.row_base[doc];
  bool ok = fetch(b, R.id_actor, B.r_id_actor + rb, R.n) && fetch(b, R.id_ctr, B.r_id_ctr + rb, R.n) &&
            fetch(b, R.obj_actor, B.r_obj_actor + rb, R.n) && fetch(b, R.obj_ctr, B.r_obj_ctr + rb, R.n) &&
            fetch(b, R.key_actor, B.r_key_actor + rb, R.n) && fetch(b, R.key_ctr, B.r_key_ctr + rb, R.n) &&
            fetch(b, R.key_off, B.r_key_off + rb, R.n) && fetch(b, R.key_len, B.r_key_len + rb, R.n) &&
            fetch(b, R.vraw_off, B.r_vraw_off + rb, R.n) && fetch(b, R.succ_cnt, B.r_succ_cnt + rb, R.n) &&
            fetch(b, R.insert, B.r_insert + rb, R.n) && fetch(b, R.action, B.r_action + rb, R.n) &&
            fetch(b, R.visible, B.r_visible + rb, R.n) && fetch(b, R.vmeta, B.r_vmeta + rb, R.n) &&
            fetch(b, R.inc_sum, B.r_inc_sum + rb, R.n);
  if (!ok) return AMH_ERR_DEVICE;
  if (!ensure_host_arena(b->staged)) return AMH_ERR_DEVICE;
  R.chunk = b->staged->h_arena.data() + b->staged->h_doc_off[2 * (size_t)doc];
  R.D = &hv.desc[doc];
  u32 ab = hv.actor_base[doc];
  for (u32 a = 0; a < R.D->n_actors; a++) R.actor_hex.push_back(hexs(R.chunk + hv.actor_off[ab + a], hv.actor_len[ab + a]));
  u32 i = 0;
  while (i < R.n) {
    u32 j = i;
    while (j < R.n && R.obj_ctr[j] == R.obj_ctr[i] && R.obj_actor[j] == R.obj_actor[i]) j++;
    R.obj_range[((u64)R.obj_ctr[i] << 32) | R.obj_actor[i]] = {i, j};
    i = j;
  }
  std::string s;
  hydrate_obj(R, 0, 0, s);
  out->data = (uint8_t*)malloc(s.size() + 1);
  if (!out->data) return AMH_ERR_OOM;
  memcpy(out->data, s.data(), s.size());
  out->data[s.size()] = 0;
  out->len = s.size();
  return AMH_SUCCESS;
}

extern "C" void amh_bytes_free(amh_bytes_out* o) {
  if (!o) return;
  free(o->data);
  o->data = nullptr;
  o->len = 0;
}
extern "C" void amh_u64_free(uint64_t* p) { free(p); }

// ---------------------------------------------------------------------------------------------
// diagnostics: copy one named device array of a batch to the host (tests compare the arrays of the
// HIP build with those of the host emulation stage by stage)
// ---------------------------------------------------------------------------------------------
extern "C" int amh_debug_array(const amh_batch* b, const char* name, void* out, size_t cap, size_t* len) {
  if (!b || !name || !len) return AMH_ERR_INVALID_ARG;
  AMH_NEEDS_OPS(b, "amh_debug_array")
  const BatchDev& B = b->B;
  size_t n = B.n_docs, T = (size_t)B.tot_rows + B.tot_dt, NC = B.tot_chg, NR = B.tot_rows, NS = B.tot_succ;
  struct Ent { const char* name; const void* p; size_t bytes; };
  const Ent tab[] = {
    {"status", B.status, n * 4}, {"fail_line", B.fail_line, n * 4}, {"desc", B.desc, n * sizeof(DocDesc)}, {"cnt", B.cnt, n * sizeof(DocCounts)},
    {"row_base", B.row_base, (n + 1) * 4}, {"succ_base", B.succ_base, (n + 1) * 4}, {"chg_base", B.chg_base, (n + 1) * 4},
    {"dep_base", B.dep_base, (n + 1) * 4}, {"actor_base", B.actor_base, (n + 1) * 4}, {"ht_base", B.ht_base, (n + 1) * 4},
    {"dt_base", B.dt_base, (n + 1) * 4}, {"map_base", B.map_base, (n + 1) * 4}, {"doc_hash", B.doc_hash, n * 32},
    {"doc_max_op", B.doc_max_op, n * 4}, {"heads_ok", B.heads_ok, n * 4},
    {"actor_off", B.actor_off, (size_t)B.tot_actors * 4}, {"actor_len", B.actor_len, (size_t)B.tot_actors * 4},
    {"actor_rank", B.actor_rank, (size_t)B.tot_actors * 4}, {"actor_by_rank", B.actor_by_rank, (size_t)B.tot_actors * 4},
    {"a_first_chg", B.a_first_chg, (size_t)B.tot_actors * 4}, {"a_last_chg", B.a_last_chg, (size_t)B.tot_actors * 4},
    {"c_actor", B.c_actor, NC * 4}, {"c_seq", B.c_seq, NC * 8}, {"c_max_op", B.c_max_op, NC * 8}, {"c_time", B.c_time, NC * 8},
    {"c_msg_off", B.c_msg_off, NC * 4}, {"c_msg_len", B.c_msg_len, NC * 4}, {"c_ndeps", B.c_ndeps, NC * 4},
    {"c_dep_base", B.c_dep_base, NC * 4}, {"c_extra_off", B.c_extra_off, NC * 4}, {"c_extra_len", B.c_extra_len, NC * 4},
    {"c_doc", B.c_doc, NC * 4}, {"dep_idx", B.dep_idx, (size_t)B.tot_deps * 4}, {"c_nops", B.c_nops, NC * 4},
    {"c_start_op", B.c_start_op, NC * 8}, {"c_op_base", B.c_op_base, (NC + 1) * 4}, {"c_next_same_actor", B.c_next_same_actor, NC * 4}, {"c_level", B.c_level, NC * 4}, {"c_wide", B.c_wide, NC * 4},
    {"c_hash", B.c_hash, NC * 32}, {"c_out_off", B.c_out_off, (NC + 1) * 4}, {"c_out_len", B.c_out_len, NC * 4},
    {"c_body_len", B.c_body_len, NC * 4}, {"c_cols_off", B.c_cols_off, NC * 4}, {"c_col_len", B.c_col_len, NC * AMH_NCOLS * 4},
    {"c_map_base", B.c_map_base, NC * 4}, {"c_n_other", B.c_n_other, NC * 4}, {"chg_actor_map", B.chg_actor_map, (size_t)b->tot_map * 4},
    {"r_id_actor", B.r_id_actor, NR * 4}, {"r_id_ctr", B.r_id_ctr, NR * 4}, {"r_obj_actor", B.r_obj_actor, NR * 4},
    {"r_obj_ctr", B.r_obj_ctr, NR * 4}, {"r_key_actor", B.r_key_actor, NR * 4}, {"r_key_ctr", B.r_key_ctr, NR * 4},
    {"r_key_off", B.r_key_off, NR * 4}, {"r_key_len", B.r_key_len, NR * 4}, {"r_insert", B.r_insert, NR}, {"r_action", B.r_action, NR},
    {"r_expand", B.r_expand, NR}, {"r_vmeta", B.r_vmeta, NR * 8}, {"r_vraw_off", B.r_vraw_off, NR * 4}, {"r_mark_off", B.r_mark_off, NR * 4},
    {"r_mark_len", B.r_mark_len, NR * 4}, {"r_succ_cnt", B.r_succ_cnt, NR * 4}, {"r_succ_base", B.r_succ_base, NR * 4},
    {"r_doc", B.r_doc, NR * 4}, {"r_inc_cnt", B.r_inc_cnt, NR * 4}, {"r_inc_sum", B.r_inc_sum, NR * 8}, {"r_visible", B.r_visible, NR},
    {"r_winner", B.r_winner, NR}, {"s_actor", B.s_actor, NS * 4}, {"s_ctr", B.s_ctr, NS * 4}, {"s_row", B.s_row, NS * 4},
    {"s_target", B.s_target, NS * 4}, {"t_pred_cnt", B.t_pred_cnt, T * 4}, {"t_pred_base", B.t_pred_base, (T + 1) * 4},
    {"t_chg", B.t_chg, T * 4}, {"j_pred_base", B.j_pred_base, ((size_t)B.tot_ops + 1) * 4}, {"pred_ids", B.pred_ids, NS * 8}, {"ops_sorted", B.ops_sorted, (size_t)B.tot_ops * 4},
    {"out_arena", B.out_arena, (size_t)b->tot_out},
    {"cv_obj", B.cv[0], (size_t)B.tot_ops * 8}, {"cv_key", B.cv[2], (size_t)B.tot_ops * 16}, {"cv_misc", B.cv[6], (size_t)B.tot_ops * 8},
    {"cv_val", B.cv[8], (size_t)B.tot_ops * 16},
  };
  for (const Ent& e : tab) {
    if (strcmp(e.name, name) != 0) continue;
    *len = e.bytes;
    if (!out || cap < e.bytes) return cap == 0 ? AMH_SUCCESS : AMH_ERR_INVALID_ARG;
    if (!e.p) { *len = 0; return AMH_SUCCESS; }
    return rt_d2h(out, e.p, e.bytes, b->ctx->stream) ? AMH_SUCCESS : AMH_ERR_DEVICE;
  }
  return AMH_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------
// boundary entry points declared for the path but not built in this round: they fail loudly
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// merge — automerge.rs:917-940.  Host: which changes are new and in which order they apply
// (get_changes_added :1311-1334 + the causal queue of apply_changes :797-858), actor-table union.
// Device: everything per op (amh_merge.inl), then the save kernels over the merged rows; the merged
// documents are loaded again through the normal pipeline, which re-derives every change hash and
// checks the heads, so a wrong merge cannot go unnoticed.
// ---------------------------------------------------------------------------------------------
template <class T>
static T* upload(amh_batch* b, const std::vector<T>& v, bool* ok) {
  T* p = balloc<T>(b, v.size() + 1, ok);
  if (p && !v.empty() && !rt_h2d(p, v.data(), v.size() * sizeof(T), b->ctx->stream)) *ok = false;
  return p;
}
// Save of documents with 64k rows and more (g_sv_big, amh_segenc.inl): the job tables.  `B` is the caller's
// copy of the batch descriptor (the sv_* fields are per call); the vectors stay alive until the caller has synchronised.
// AMH_BIGSAVE_ROWS / AMH_BIGSAVE_RANGE: rows from which a document is "big" / rows per range (tests push every shape
// of document through the ranges with 1 / 64).
static bool save_big_prepare(amh_ctx* ctx, amh_batch* b, BatchDev& B, const std::vector<i32>& status, SaveBigHost& H) {
  B.sv_nj = 0; B.sv_nbc = 0; B.sv_big = nullptr;
#ifndef AMH_EMU
  u64 thr = 65536, range = 16384;
  if (const char* e = getenv("AMH_BIGSAVE_ROWS")) thr = strtoull(e, nullptr, 10);
  if (const char* e = getenv("AMH_BIGSAVE_RANGE")) range = strtoull(e, nullptr, 10);
  range = std::max<u64>(64, range & ~(u64)63);
  if (thr == 0) thr = 1;
  const u32 n = B.n_docs;
  const u64 cthr = std::min<u64>(thr, 8192);   // change-metadata columns: documents with this many changes
  if (n == 0 || ((u64)B.tot_rows < thr && (u64)B.tot_succ < thr && (u64)B.tot_chg < cthr)) return true;
  H.cnt.resize(n);
  if (!rt_d2h(H.cnt.data(), B.cnt, (size_t)n * sizeof(DocCounts), ctx->stream)) return false;
  H.big.assign(n, 0);
  for (u32 d = 0; d < n; d++) {
    if (d < status.size() && status[d]) continue;
    const bool ops_big = (u64)H.cnt[d].n_rows >= thr || (u64)H.cnt[d].n_succ >= thr, chg_big = (u64)H.cnt[d].n_chg >= cthr;
    if (!ops_big && !chg_big) continue;
    H.big[d] = (ops_big ? 1 : 0) | (chg_big ? 2 : 0);
    for (u32 col = chg_big ? 0 : 9; col < (ops_big ? 25u : 9u); col++) {
      const u64 ncol = col < 9 ? (col == 6 ? H.cnt[d].n_deps : H.cnt[d].n_chg) : ((col == 21 || col == 22) ? H.cnt[d].n_succ : H.cnt[d].n_rows);
      const u64 nr = (ncol + range - 1) / range;
      if (nr == 0) continue;
      if (H.j_bc.size() + nr >= 0x7FFFFFFFull) return false;
      H.b_doc.push_back(d); H.b_col.push_back(col); H.b_first.push_back((u32)H.j_bc.size()); H.b_nr.push_back((u32)nr);
      for (u64 k = 0; k < nr; k++) { H.j_bc.push_back((u32)H.b_doc.size() - 1); H.j_k.push_back((u32)k); }
    }
  }
  if (H.j_bc.empty()) return true;
  bool ok = true;
  const size_t nj = H.j_bc.size();
  B.svj_bc = upload(b, H.j_bc, &ok); B.svj_k = upload(b, H.j_k, &ok);
  B.svb_doc = upload(b, H.b_doc, &ok); B.svb_col = upload(b, H.b_col, &ok); B.svb_first = upload(b, H.b_first, &ok); B.svb_nr = upload(b, H.b_nr, &ok);
  B.sv_big = upload(b, H.big, &ok);
  B.svj_sum = balloc<u32>(b, 9 * nj + 1, &ok); B.svj_cin = balloc<u32>(b, 3 * nj + 1, &ok);
  B.svj_total = balloc<u32>(b, nj + 1, &ok); B.svj_any = balloc<u32>(b, nj + 1, &ok); B.svj_tail = balloc<u32>(b, nj + 1, &ok);
  if (!ok) { B.sv_nj = 0; B.sv_big = nullptr; return false; }
  B.sv_nj = (u32)nj; B.sv_nbc = (u32)H.b_doc.size(); B.sv_range = (u32)range;
#else
  (void)ctx; (void)b; (void)status; (void)H;
#endif
  return true;
}
// The same for the change re-encode of a load: a "big column" is one of the 14 op columns of a change with
// AMH_BIGENC_ROWS ops (default 65536) and more; ranges of AMH_BIGENC_RANGE values (default 16384).
static bool enc_big_prepare(amh_ctx* ctx, amh_batch* b, BatchDev& B, SaveBigHost& H) {
  B.sv_nj = 0; B.sv_nbc = 0; B.sv_big = nullptr;
#ifndef AMH_EMU
  u64 thr = 65536, range = 16384;
  if (const char* e = getenv("AMH_BIGENC_ROWS")) thr = strtoull(e, nullptr, 10);
  if (const char* e = getenv("AMH_BIGENC_RANGE")) range = strtoull(e, nullptr, 10);
  range = std::max<u64>(64, range & ~(u64)63);
  if (thr == 0) thr = 1;
  const u32 nc = B.tot_chg;
  if (nc == 0 || (u64)B.tot_ops < thr) return true;
  H.nops.resize(nc); H.opbase.resize(nc);
  if (!rt_d2h(H.nops.data(), B.c_nops, (size_t)nc * sizeof(u32), ctx->stream) || !rt_d2h(H.opbase.data(), B.c_op_base, (size_t)nc * sizeof(u32), ctx->stream)) return false;
  H.big.assign(nc, 0);
  for (u32 c = 0; c < nc; c++) {
    const u64 n = H.nops[c];
    if (n < thr) continue;
    u32 pb[2] = {0, 0};   // the change's pred entries: j_pred_base[first op], j_pred_base[behind its last op]
    if (!rt_d2h(&pb[0], B.j_pred_base + H.opbase[c], sizeof(u32), ctx->stream) || !rt_d2h(&pb[1], B.j_pred_base + H.opbase[c] + n, sizeof(u32), ctx->stream)) return false;
    const u64 pn = pb[1] - pb[0];
    H.big[c] = 1;
    for (u32 col = 0; col < AMH_NCOLS; col++) {
      const u64 ncol = (col == 10 || col == 11) ? pn : n;
      const u64 nr = (ncol + range - 1) / range;
      if (nr == 0) continue;
      if (H.j_bc.size() + nr >= 0x7FFFFFFFull) return false;
      H.b_doc.push_back(c); H.b_col.push_back(col); H.b_first.push_back((u32)H.j_bc.size()); H.b_nr.push_back((u32)nr);
      for (u64 k = 0; k < nr; k++) { H.j_bc.push_back((u32)H.b_doc.size() - 1); H.j_k.push_back((u32)k); }
    }
  }
  if (H.j_bc.empty()) return true;
  bool ok = true;
  const size_t nj = H.j_bc.size();
  B.svj_bc = upload(b, H.j_bc, &ok); B.svj_k = upload(b, H.j_k, &ok);
  B.svb_doc = upload(b, H.b_doc, &ok); B.svb_col = upload(b, H.b_col, &ok); B.svb_first = upload(b, H.b_first, &ok); B.svb_nr = upload(b, H.b_nr, &ok);
  B.sv_big = upload(b, H.big, &ok);
  B.svj_sum = balloc<u32>(b, 9 * nj + 1, &ok); B.svj_cin = balloc<u32>(b, 3 * nj + 1, &ok);
  B.svj_total = balloc<u32>(b, nj + 1, &ok); B.svj_any = balloc<u32>(b, nj + 1, &ok); B.svj_tail = balloc<u32>(b, nj + 1, &ok);
  if (!ok) { B.sv_nj = 0; B.sv_big = nullptr; return false; }
  B.sv_nj = (u32)nj; B.sv_nbc = (u32)H.b_doc.size(); B.sv_range = (u32)range;
#else
  (void)ctx; (void)b; (void)H;
#endif
  return true;
}
static void save_big_launch(amh_ctx* ctx, const BatchDev& B, int phase) {
#ifndef AMH_EMU
  amh_launch_save_big(B, phase, (void*)ctx->stream);
#else
  (void)ctx; (void)B; (void)phase;
#endif
}
struct Hash32 {
  u8 b[32];
  bool operator==(const Hash32& o) const { return memcmp(b, o.b, 32) == 0; }
};
struct Hash32Hasher { size_t operator()(const Hash32& h) const { size_t v; memcpy(&v, h.b, sizeof v); return v; } };

// byte buffer that is not zero-filled on allocation (staging arenas of several hundred MB are written in full anyway)
struct RawBuf {
  u8* p = nullptr;
  size_t n = 0;
  RawBuf() {}
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  ~RawBuf() { free(p); }
  bool alloc(size_t bytes) { free(p); p = (u8*)malloc(bytes ? bytes : 1); n = p ? bytes : 0; return p != nullptr; }
  u8* data() const { return p; }
  size_t size() const { return n; }
};

// per-document host bookkeeping of a batch call on a few threads (documents are independent).
// The threads are a process-wide pool: a wide merge runs a dozen such loops per call, and starting and joining 16
// threads for each cost about as much as the loops themselves.  One region at a time (a second caller, or a loop
// started from inside a region, runs the way it did before the pool: on threads of its own / inline); a forked child
// gets a pool of its own (the parent's threads do not exist there); the pool is never torn down (its threads sleep on a
// condition variable until the process ends).
struct HostPool {
  std::mutex m, region;
  std::condition_variable cv_start, cv_done;
  const std::function<void()>* body = nullptr;
  u64 gen = 0;
  unsigned running = 0, n_workers = 0;
  long pid = 0;
  explicit HostPool(unsigned workers) : n_workers(workers), pid((long)getpid()) {
    for (unsigned t = 0; t < workers; t++)
      std::thread([this]() {
        in_worker() = true;
        u64 seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv_start.wait(lk, [&] { return gen != seen; });
          seen = gen;
          const std::function<void()>* f = body;
          lk.unlock();
          (*f)();
          lk.lock();
          if (--running == 0) cv_done.notify_all();
        }
      }).detach();
  }
  static bool& in_worker() { static thread_local bool w = false; return w; }
  void run(const std::function<void()>& f) {   // f on every worker and on the caller; returns when all are back
    {
      std::lock_guard<std::mutex> lk(m);
      body = &f; running = n_workers; gen++;
    }
    cv_start.notify_all();
    f();
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return running == 0; });
  }
  static HostPool* get(unsigned nt) {
    static std::mutex gm;
    static HostPool* pool = nullptr;
    std::lock_guard<std::mutex> lk(gm);
    if (!pool || pool->pid != (long)getpid()) pool = new HostPool(nt - 1);   // (after a fork: the old one is left alone)
    return pool;
  }
};
template <class F>
static void host_parallel_for(u32 n, F f, u32 min_items, u32 grain) {
  // Threads of the host loops: AMH_HOST_THREADS if set; else at most 16 and at most OMP_NUM_THREADS (on shared machines
  // that is the process's CPU share).  torch.distributed.run sets OMP_NUM_THREADS=1 for every rank when the caller left
  // it unset: under it (LOCAL_WORLD_SIZE > 1) a value of 1 is read as "unset" and the machine's threads are divided
  // among the local ranks, or a rank's host passes would run 16 times slower beside its GPU.
  static const unsigned nt = [] {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned v = std::min<unsigned>(hw, 16);
    if (const char* e = getenv("AMH_HOST_THREADS")) { int o = atoi(e); if (o > 0) return std::min<unsigned>((unsigned)o, 64); }
    if (const char* e = getenv("OMP_NUM_THREADS")) {
      int o = atoi(e);
      const char* lw = getenv("LOCAL_WORLD_SIZE");
      const int ranks = lw ? atoi(lw) : 1;
      if (o == 1 && ranks > 1) v = std::min<unsigned>(v, std::max(1u, hw / (unsigned)ranks));
      else if (o > 0) v = std::min<unsigned>(v, (unsigned)o);
    }
    return v;
  }();
  if (n < min_items || nt < 2 || HostPool::in_worker()) { for (u32 i = 0; i < n; i++) f(i); return; }
  std::atomic<u32> next{0};
  const std::function<void()> body = [&]() {
    for (;;) {
      u32 lo = next.fetch_add(grain);
      if (lo >= n) break;
      u32 hi = std::min(n, lo + grain);
      for (u32 i = lo; i < hi; i++) f(i);
    }
  };
  HostPool* pool = HostPool::get(nt);
  std::unique_lock<std::mutex> one(pool->region, std::try_to_lock);
  if (one.owns_lock()) { pool->run(body); return; }
  std::vector<std::thread> ts;   // the pool is serving another caller
  for (unsigned t = 0; t < nt; t++) ts.emplace_back(body);
  for (auto& th : ts) th.join();
}

// everything the host decides for a merge / apply call (see MergeDev)
struct MgPlan {
  u32 P = 0;
  std::vector<u32> d_pair, s_pair, p_a, p_b, a_len_pad;
  std::vector<i32> p_status;
  std::vector<u32> mapa_base, mapb_base, map_a, map_b, map_b2a;
  std::vector<u8> s_added;
  std::vector<u32> m_actor_base, m_actor_off, m_actor_len, m_actor_rank;
  std::vector<u32> m_chg_base, mc_from, mc_ndeps, mc_dep_base, m_dep_idx, mc_doc;
  std::vector<u8> mc_is_dep;
  std::vector<DocDesc> m_desc;
  std::vector<DocCounts> m_cnt;
  u64 arena_len = 0;
  int n_failed = 0;
};
// apply only: the staged change chunks (batch U, already decoded and verified) and the host's decisions
struct ApplyRun {
  BatchDev U{};
  u32 n_chunks = 0;
  ChgHdr* d_hdr = nullptr;
  u32* d_u_chg = nullptr;
  u32* d_u_fill = nullptr;
  std::vector<u32> c_pair, c_rel, amap_base, amap, ix_base, ix_actor, ix_start, ix_nops, ix_row;
  std::vector<u8> c_applied;
  std::vector<u64> h_start_op;       // [n_chunks] first op counter of the chunk's change
  std::vector<u32> h_nops;           // [n_chunks] its ops (deletes included)
  std::vector<u64> h_chunk_off;      // [n_chunks] where the chunk lies in U's arena
  std::vector<u32> h_chunk_len;
  bool all_dev = false;              // every chunk came out of a loaded batch's device memory (merge): verified and canonical by construction
  std::vector<DocDesc> s_desc;       // per pair: region of the arena holding its chunks
  std::vector<DocCounts> s_cnt;      // per pair: rows of U
  std::vector<u32> s_row_base;       // [P + 1]
};
#define MG_REFS(pl)                                                                                                   \
  auto &d_pair = (pl).d_pair, &s_pair = (pl).s_pair, &p_a = (pl).p_a, &p_b = (pl).p_b, &a_len_pad = (pl).a_len_pad;   \
  auto& p_status = (pl).p_status;                                                                                     \
  auto &mapa_base = (pl).mapa_base, &mapb_base = (pl).mapb_base, &map_a = (pl).map_a, &map_b = (pl).map_b, &map_b2a = (pl).map_b2a; \
  auto& s_added = (pl).s_added;                                                                                       \
  auto &m_actor_base = (pl).m_actor_base, &m_actor_off = (pl).m_actor_off, &m_actor_len = (pl).m_actor_len, &m_actor_rank = (pl).m_actor_rank; \
  auto &m_chg_base = (pl).m_chg_base, &mc_from = (pl).mc_from, &mc_ndeps = (pl).mc_ndeps, &mc_dep_base = (pl).mc_dep_base, &m_dep_idx = (pl).m_dep_idx, &mc_doc = (pl).mc_doc; \
  auto& mc_is_dep = (pl).mc_is_dep;                                                                                   \
  auto& m_desc = (pl).m_desc;                                                                                         \
  auto& m_cnt = (pl).m_cnt;                                                                                           \
  u64& arena_len = (pl).arena_len;                                                                                    \
  int& n_failed = (pl).n_failed;                                                                                      \
  (void)d_pair; (void)s_pair; (void)p_a; (void)p_b; (void)a_len_pad; (void)p_status; (void)mapa_base; (void)mapb_base; (void)map_a; (void)map_b; \
  (void)map_b2a; (void)s_added; (void)m_actor_base; (void)m_actor_off; (void)m_actor_len; (void)m_actor_rank; (void)m_chg_base; (void)mc_from;   \
  (void)mc_ndeps; (void)mc_dep_base; (void)m_dep_idx; (void)mc_doc; (void)mc_is_dep; (void)m_desc; (void)m_cnt; (void)arena_len; (void)n_failed;
static void mg_plan_init(MgPlan& pl, u32 P, u32 nd, u32 ns, u32 s_tot_chg) {
  pl.P = P;
  pl.d_pair.assign(nd, AMH_NULL32); pl.s_pair.assign(ns, AMH_NULL32);
  pl.p_a.assign(P, 0); pl.p_b.assign(P, 0); pl.a_len_pad.assign(P, 0); pl.p_status.assign(P, 0);
  pl.mapa_base.assign(P + 1, 0); pl.mapb_base.assign(P + 1, 0); pl.s_added.assign((size_t)s_tot_chg + 1, 0);
  pl.m_actor_base.assign(P + 1, 0); pl.m_chg_base.assign(P + 1, 0);
  pl.m_desc.resize(P); pl.m_cnt.resize(P);
}
// length of the document chunk an input starts with: what a reload of an unchanged document feeds the
// pipeline (the chunks behind it have been applied already).  0 when the input starts with a change chunk.
static size_t first_doc_chunk_len(const u8* p, size_t avail) {
  if (avail < 10 || p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83) return avail;
  if (p[8] == 1 || p[8] == 2) return 0;
  if (p[8] != 0) return avail;
  u64 len = 0; u32 shift = 0; size_t pos = 9;
  for (;;) {
    if (pos >= avail || shift > 63) return avail;
    u8 b = p[pos++];
    len |= (u64)(b & 0x7f) << shift;
    if (!(b & 0x80)) break;
    shift += 7;
  }
  if (len > avail - pos) return avail;
  return pos + (size_t)len;
}
static amh_batch* load_staged_impl(amh_ctx* ctx, const amh_staged* st_c, const amh_load_opts* opts, bool trailing);
// The merged documents of a merge / apply call as a batch of their own, built from the rows the merge kernels put in
// document order (M) instead of from saved bytes: M's arrays are what the decode would have produced, the rest of the
// load pipeline (load_tail) derives everything else and re-verifies every change (re-encode, SHA-256, against the hash
// the change came with).  M's arrays are borrowed from the merge scratch until adopt_merged_arrays hands them over.
static amh_batch* rows_to_batch(amh_ctx* ctx, amh_batch* mb, const amh_batch* dst, const MergeDev& G, const MgPlan& pl, const ApplyRun* ap, u8* m_arena, const char* who) {
  const BatchDev& M = G.M;
  const u32 P = pl.P;
  auto* st = new amh_staged();
  st->ctx = ctx; st->n_docs = P; st->host_ok = false; st->single_chunk = true; st->internal = true;
  st->arena_bytes = (size_t)pl.arena_len + 256;
  st->h_doc_off.assign(2 * (size_t)P + 2, 0);
  for (u32 p = 0; p < P; p++) { st->h_doc_off[2 * (size_t)p] = pl.m_desc[p].chunk_off; st->h_doc_off[2 * (size_t)p + 1] = pl.m_desc[p].chunk_off + pl.m_desc[p].chunk_len; }
  st->d_arena.p = m_arena; st->d_arena.cap = 0;   // borrowed
  st->d_doc_off.p = ctx->pool.alloc(st->h_doc_off.size() * sizeof(u64), &st->d_doc_off.cap);
  std::unique_ptr<amh_batch, void (*)(amh_batch*)> nbh(new amh_batch(), amh_batch_free);
  amh_batch* nb = nbh.get();
  nb->ctx = ctx;
  nb->staged = st;
  auto oom = [&]() -> amh_batch* { set_error(ctx, std::string(who) + ": device out of memory"); return nullptr; };
  if (!st->d_doc_off.p) return oom();
  BatchDev& B = nb->B;
  B = M;
  B.arena = m_arena;
  B.doc_off = (const u64*)st->d_doc_off.p;
  B.check_heads = 0;   // there are no stored heads to compare with: every change hash is checked instead (k_verify_hash)
  B.max_doc_rows = ctx->max_doc_rows;
  B.skip_doc_sha = 0; B.sha_host_min = 0; B.first_len = nullptr; B.sha_gate = nullptr; B.sha_bad = nullptr;
  B.sv_status = nullptr; B.sv_col_len = nullptr; B.sv_wide = nullptr; B.sv_body_len = nullptr; B.sv_off = nullptr; B.sv_out = nullptr;
  // per document: dep / actor-map bases, highest op counter, dep counts (the host made the merged history)
  std::vector<u32> dep_base(P + 2, 0), map_base(P + 2, 0), max_op(P + 1, 0);
  std::vector<DocCounts> cnt(pl.m_cnt);
  {
    std::vector<u32> dst_max(P);
    if (!rt_d2h(dst_max.data(), G.D.doc_max_op, (size_t)P * sizeof(u32), ctx->stream) || !rt_sync(ctx->stream)) { set_error(ctx, std::string(who) + ": D2H copy failed"); return nullptr; }
    u64 deps = 0, maps = 0;
    for (u32 p = 0; p < P; p++) {
      dep_base[p] = (u32)deps; map_base[p] = (u32)maps;
      u32 nd_p = 0;
      for (u32 c = pl.m_chg_base[p]; c < pl.m_chg_base[p + 1]; c++) nd_p += pl.mc_ndeps[c];
      cnt[p].n_chg = pl.m_chg_base[p + 1] - pl.m_chg_base[p];
      cnt[p].n_deps = nd_p;
      deps += nd_p;
      maps += (u64)cnt[p].n_chg * pl.m_desc[p].n_actors;
      max_op[p] = dst_max[p];
      if (!ap) {   // row merge: the source document's own highest counter
        u32 sm = 0;
        if (!rt_d2h(&sm, G.S.doc_max_op + pl.p_b[p], sizeof(u32), ctx->stream) || !rt_sync(ctx->stream)) { set_error(ctx, std::string(who) + ": D2H copy failed"); return nullptr; }
        if (sm > max_op[p]) max_op[p] = sm;
      }
    }
    if (deps >= 0xFFFFFFFFull || maps >= 0xFFFFFFFFull) { set_error(ctx, std::string(who) + ": batch too large"); return nullptr; }
    dep_base[P] = (u32)deps; map_base[P] = (u32)maps;
    nb->tot_map = (u32)maps;
    if (ap)   // the applied changes: ids start_op .. start_op + ops - 1 (deletes included: they are rows of U)
      for (u32 u = 0; u < ap->n_chunks; u++)
        if (ap->c_applied[u]) { u32 p = ap->c_pair[u]; u64 hi = ap->h_start_op[u] + ap->h_nops[u]; if (hi) hi--; if (hi > max_op[p]) max_op[p] = (u32)std::min<u64>(hi, 0xFFFFFFFEull); }
  }
  bool ok = true;
  const size_t NC = (size_t)B.tot_chg + 2, NR = (size_t)B.tot_rows + 2, NS = (size_t)B.tot_succ + 2, NA = (size_t)B.tot_actors + 1;
  // row n_rows / n_succ of the counts come from the device (k_mg_doc_succ); the change counts from the host
  std::vector<DocCounts> dev_cnt(P);
  if (!rt_d2h(dev_cnt.data(), M.cnt, (size_t)P * sizeof(DocCounts), ctx->stream) || !rt_sync(ctx->stream)) { set_error(ctx, std::string(who) + ": D2H copy failed"); return nullptr; }
  for (u32 p = 0; p < P; p++) { cnt[p].n_rows = dev_cnt[p].n_rows; cnt[p].n_succ = dev_cnt[p].n_succ; }
  if (!rt_h2d(M.cnt, cnt.data(), (size_t)P * sizeof(DocCounts), ctx->stream)) { set_error(ctx, std::string(who) + ": H2D copy failed"); return nullptr; }
  B.dep_base = upload(nb, dep_base, &ok); B.map_base = upload(nb, map_base, &ok); B.doc_max_op = upload(nb, max_op, &ok);
  B.ht_base = balloc<u32>(nb, P + 2, &ok); B.dt_base = balloc<u32>(nb, P + 2, &ok);
  B.doc_hash = balloc<u8>(nb, (size_t)P * 32 + 32, &ok);
  B.dn_base = balloc<u32>(nb, P + 2, &ok); B.dn_stride = balloc<u32>(nb, P + 2, &ok);
  B.heads_ok = balloc<u32>(nb, P + 1, &ok); B.doc_level = balloc<u32>(nb, P + 1, &ok); B.max_level = balloc<u32>(nb, 4, &ok);
  B.a_first_chg = balloc<u32>(nb, NA, &ok); B.a_last_chg = balloc<u32>(nb, NA, &ok);
  B.c_nops = balloc<u32>(nb, NC, &ok); B.c_start_op = balloc<u64>(nb, NC, &ok); B.c_op_base = balloc<u32>(nb, NC, &ok);
  B.c_npred = nullptr;   // a merged batch's changes are adopted or verified, not counted again: as a later merge source it is counted by k_ap_count
  B.c_next_same_actor = balloc<u32>(nb, NC, &ok); B.c_level = balloc<u32>(nb, NC, &ok);
  // merged from loaded batches: the documents adopt their sources' change chunks and hashes
  const bool adopt = ap && ap->all_dev && getenv("AMH_VERIFY_MERGE") == nullptr && dst->hv.chg_ok;
  if (!adopt) B.c_hash = balloc<u8>(nb, NC * 32, &ok);   // else: M's (the hashes the changes came with)
  B.c_out_off = balloc<u32>(nb, NC, &ok); B.c_out_len = balloc<u32>(nb, NC, &ok); B.c_body_len = balloc<u32>(nb, NC, &ok);
  B.c_cols_off = balloc<u32>(nb, NC, &ok); B.c_col_len = balloc<u32>(nb, NC * AMH_NCOLS, &ok); B.c_wide = balloc<u32>(nb, NC, &ok);
  B.c_map_base = balloc<u32>(nb, NC, &ok); B.c_n_other = balloc<u32>(nb, NC, &ok);
  B.chg_actor_map = balloc<u32>(nb, (size_t)nb->tot_map + 1, &ok);
  B.r_inc_cnt = balloc<u32>(nb, NR, &ok); B.r_inc_sum = balloc<i64>(nb, NR, &ok);
  B.r_visible = balloc<u8>(nb, NR, &ok); B.r_winner = balloc<u8>(nb, NR, &ok);
  B.s_row = balloc<u32>(nb, NS, &ok); B.s_target = balloc<u32>(nb, NS, &ok); B.pred_ids = balloc<u64>(nb, NS, &ok);
  if (!ok) return oom();
  if (!rt_h2d(st->d_doc_off.p, st->h_doc_off.data(), st->h_doc_off.size() * sizeof(u64), ctx->stream)) { set_error(ctx, std::string(who) + ": H2D copy failed"); return nullptr; }
  rt_memset(B.doc_hash, 0, (size_t)P * 32 + 32, ctx->stream);
  rt_memset(B.max_level, 0, 16, ctx->stream);
  rt_memset(B.s_row, 0xFF, NS * 4, ctx->stream);
  rt_memset(B.chg_actor_map, 0, ((size_t)nb->tot_map + 1) * 4, ctx->stream);
  rt_memset(B.ht_base, 0, (P + 2) * sizeof(u32), ctx->stream); rt_memset(B.dt_base, 0, (P + 2) * sizeof(u32), ctx->stream);
  launch(ctx, AK_M_INIT, B, P);
  BigSha big;   // nothing of a merged document is hashed on the host
  AdoptHistory ah;
  if (adopt) {
    const HostView& ha = dst->hv;
    const u32 nc = B.tot_chg;
    ah.src.resize(nc); ah.len.resize(nc);
    for (u32 c = 0; c < nc; c++) {
      const u32 from = pl.mc_from[c], x = from & 0x7FFFFFFFu;
      if (from >> 31) { ah.src[c] = (u64)(uintptr_t)(G.U.arena + ap->h_chunk_off[x]); ah.len[c] = ap->h_chunk_len[x]; }
      else { ah.src[c] = (u64)(uintptr_t)(G.D.out_arena + ha.c_out_off[x]); ah.len[c] = ha.c_out_len[x]; }
    }
  }
  amh_batch* out = load_tail(ctx, nbh.get(), big, nullptr, false, true, adopt ? nullptr : M.c_hash, adopt ? &ah : nullptr);
  if (!out) { nbh.release(); return nullptr; }   // load_tail freed it
  return nbh.release();
}
// the merged rows now belong to the new batch: its part of the merge scratch moves over, the rest goes back to the pool
static void adopt_merged_arrays(amh_batch* mb, amh_batch* nb, u8* m_arena) {
  const BatchDev& B = nb->B;
  const void* keep[] = {B.status, B.fail_line, B.desc, B.cnt, B.row_base, B.succ_base, B.chg_base, B.actor_base, B.actor_off, B.actor_len, B.actor_rank,
                        B.actor_by_rank, B.c_hash, B.c_actor, B.c_seq, B.c_max_op, B.c_time, B.c_msg_off, B.c_msg_len, B.c_extra_off, B.c_extra_len, B.c_ndeps,
                        B.c_dep_base, B.dep_idx, B.c_doc, B.c_is_dep, B.r_id_actor, B.r_id_ctr, B.r_obj_actor, B.r_obj_ctr, B.r_key_actor, B.r_key_ctr,
                        B.r_key_off, B.r_key_len, B.r_vraw_off, B.r_mark_off, B.r_mark_len, B.r_succ_cnt, B.r_succ_base, B.r_doc, B.r_insert, B.r_action,
                        B.r_expand, B.r_vmeta, B.s_actor, B.s_ctr};
  std::vector<DevAlloc> rest;
  for (auto& a : mb->allocs) {
    if (a.p == (void*)m_arena) { nb->staged->d_arena.p = a.p; nb->staged->d_arena.cap = a.cap; continue; }
    bool k = false;
    for (const void* q : keep) k = k || (q && q == a.p);
    if (k) nb->allocs.push_back(a); else rest.push_back(a);
  }
  mb->allocs.swap(rest);
}
static int mg_execute(amh_ctx* ctx, amh_batch* dst, const BatchDev& SB, MgPlan& pl, ApplyRun* ap, const char* who) {
  MG_REFS(pl)
  const BatchDev& DB = dst->B;
  u32 P = pl.P;
  PhaseTimer pt(who);
  // ---- device ---------------------------------------------------------------------------------
  std::unique_ptr<amh_batch, void (*)(amh_batch*)> mbh(new amh_batch(), amh_batch_free);
  amh_batch* mb = mbh.get();
  mb->ctx = ctx;
  MergeDev G;
  memset(&G, 0, sizeof G);
  G.D = DB; G.S = SB;
  if (ap) G.U = ap->U;
  BatchDev& M = G.M;
  bool ok = true;
  auto failm = [&](const char* m) { set_error(ctx, std::string(who) + ": " + m); collect_profile(ctx); return AMH_ERR_DEVICE; };
  G.n_pairs = P;
  G.p_a = upload(mb, p_a, &ok); G.p_b = upload(mb, p_b, &ok); G.p_status = upload(mb, p_status, &ok);
  G.d_pair = upload(mb, d_pair, &ok); G.s_pair = upload(mb, s_pair, &ok);
  G.mapa_base = upload(mb, mapa_base, &ok); G.mapb_base = upload(mb, mapb_base, &ok);
  G.map_a = upload(mb, map_a, &ok); G.map_b = upload(mb, map_b, &ok); G.map_b2a = upload(mb, map_b2a, &ok);
  G.s_added = upload(mb, s_added, &ok); G.a_len_pad = upload(mb, a_len_pad, &ok); G.mc_from = upload(mb, mc_from, &ok);
  M.n_docs = P;
  M.tot_chg = (u32)mc_from.size(); M.tot_actors = (u32)m_actor_off.size(); M.tot_deps = (u32)m_dep_idx.size();
  M.status = balloc<i32>(mb, P + 1, &ok); M.fail_line = balloc<u32>(mb, P + 1, &ok);
  M.desc = upload(mb, m_desc, &ok); M.cnt = upload(mb, m_cnt, &ok);
  M.row_base = balloc<u32>(mb, P + 2, &ok); M.succ_base = balloc<u32>(mb, P + 2, &ok);
  M.chg_base = upload(mb, m_chg_base, &ok); M.actor_base = upload(mb, m_actor_base, &ok);
  M.actor_off = upload(mb, m_actor_off, &ok); M.actor_len = upload(mb, m_actor_len, &ok);
  M.actor_rank = upload(mb, m_actor_rank, &ok); M.actor_by_rank = upload(mb, m_actor_rank, &ok);
  size_t NC = M.tot_chg + 2;
  M.c_actor = balloc<u32>(mb, NC, &ok); M.c_seq = balloc<u64>(mb, NC, &ok); M.c_max_op = balloc<u64>(mb, NC, &ok); M.c_time = balloc<i64>(mb, NC, &ok);
  M.c_msg_off = balloc<u32>(mb, NC, &ok); M.c_msg_len = balloc<u32>(mb, NC, &ok); M.c_extra_off = balloc<u32>(mb, NC, &ok); M.c_extra_len = balloc<u32>(mb, NC, &ok);
  M.c_hash = balloc<u8>(mb, NC * 32, &ok);
  M.c_ndeps = upload(mb, mc_ndeps, &ok); M.c_dep_base = upload(mb, mc_dep_base, &ok); M.dep_idx = upload(mb, m_dep_idx, &ok);
  M.c_doc = upload(mb, mc_doc, &ok); M.c_is_dep = upload(mb, mc_is_dep, &ok);
  u8* m_arena = balloc<u8>(mb, (size_t)arena_len + 256, &ok);
  M.arena = m_arena;
  size_t NRD = DB.tot_rows + 2, NRS = SB.tot_rows + 2;
  G.new_idx = balloc<u32>(mb, NRS, &ok); G.gap_base = balloc<u32>(mb, P + 2, &ok);
  G.sub_end = balloc<u32>(mb, NRD, &ok); G.upd_end = balloc<u32>(mb, NRD, &ok); G.stack = balloc<u32>(mb, NRD, &ok);
  G.x_cnt = balloc<u32>(mb, NRD, &ok); G.x_fill = balloc<u32>(mb, NRD, &ok); G.d_to_m = balloc<u32>(mb, NRD, &ok);
  G.sub_min1 = balloc<u32>(mb, NRD / 64 + 2, &ok); G.sub_min2 = balloc<u32>(mb, NRD / 4096 + 2, &ok);
  MergeDev* Gdev = (MergeDev*)balloc<u8>(mb, sizeof(MergeDev), &ok);
  {   // lists too long for one thread (amh_types.h, big_*): a list has more than AMH_BIG_LIST entries, so there are few
    u64 entries = (u64)DB.tot_rows + DB.tot_succ + 2 * ((u64)SB.tot_rows + SB.tot_succ) + (ap ? 2 * ((u64)ap->U.tot_rows + ap->U.tot_succ) : 0);
    G.big_cap = (u32)(entries / AMH_BIG_LIST + 64);
    G.big_n = balloc<u32>(mb, 8, &ok); G.big_items = balloc<u32>(mb, 4 * 2 * (size_t)G.big_cap, &ok);
  }
  if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
  rt_memset(G.big_n, 0, 8 * sizeof(u32), ctx->stream);
  {
    std::vector<i32> mst(P);
    for (u32 p = 0; p < P; p++) mst[p] = p_status[p];   // documents of pairs that are not merged are skipped by the save kernels
    if (!rt_h2d(M.status, mst.data(), P * sizeof(i32), ctx->stream)) return failm("H2D copy failed");
  }
  rt_memset(M.fail_line, 0, (P + 1) * 4, ctx->stream);
  rt_memset(G.x_cnt, 0, NRD * 4, ctx->stream); rt_memset(G.x_fill, 0, NRD * 4, ctx->stream);
  if (ok) rt_memset(G.sub_min2, 0xFF, (NRD / 4096 + 2) * 4, ctx->stream);
  // the block is sent again only when the host changed it since the last launch (it is ~6 KB: 45 launches of a call
  // used to make 45 small copies)
  std::unique_ptr<MergeDev> G_sent(new MergeDev);
  bool G_any = false;
  auto mg = [&](int k, u32 n) {
    if (n == 0) return true;
    if (!G_any || memcmp(G_sent.get(), &G, sizeof(MergeDev)) != 0) {
      if (!rt_h2d(Gdev, &G, sizeof(MergeDev), ctx->stream)) return false;
      memcpy(G_sent.get(), &G, sizeof(MergeDev));
      G_any = true;
    }
#ifdef AMH_EMU
    launch_mg_timed(ctx, k, &G, n);
#else
    launch_mg_timed(ctx, k, Gdev, n);
#endif
    return true;
  };
  if (ap) {
    BatchDev &S = G.S, &U = G.U;
    ApplyDev& A = G.A;
    size_t NU = (size_t)U.tot_rows + 2, NC = (size_t)ap->n_chunks + 2;
    S.desc = upload(mb, ap->s_desc, &ok); S.cnt = upload(mb, ap->s_cnt, &ok); S.row_base = upload(mb, ap->s_row_base, &ok);
    S.r_succ_cnt = balloc<u32>(mb, NU, &ok); S.r_succ_base = balloc<u32>(mb, NU, &ok);
    S.c_actor = balloc<u32>(mb, NC, &ok); S.c_seq = balloc<u64>(mb, NC, &ok); S.c_max_op = balloc<u64>(mb, NC, &ok); S.c_time = balloc<i64>(mb, NC, &ok);
    S.c_msg_off = balloc<u32>(mb, NC, &ok); S.c_msg_len = balloc<u32>(mb, NC, &ok); S.c_extra_off = balloc<u32>(mb, NC, &ok); S.c_extra_len = balloc<u32>(mb, NC, &ok);
    S.c_hash = balloc<u8>(mb, NC * 32, &ok);
    A.hdr = ap->d_hdr; A.u_chg = ap->d_u_chg; A.u_fill = ap->d_u_fill;
    A.c_pair = upload(mb, ap->c_pair, &ok); A.c_rel = upload(mb, ap->c_rel, &ok); A.c_applied = upload(mb, ap->c_applied, &ok);
    A.amap_base = upload(mb, ap->amap_base, &ok); A.amap = upload(mb, ap->amap, &ok);
    A.ix_base = upload(mb, ap->ix_base, &ok); A.ix_actor = upload(mb, ap->ix_actor, &ok); A.ix_start = upload(mb, ap->ix_start, &ok);
    A.ix_nops = upload(mb, ap->ix_nops, &ok); A.ix_row = upload(mb, ap->ix_row, &ok);
    u32** scr[] = {&A.gap, &A.anchor, &A.parent, &A.o_next, &A.o_head_upd, &A.o_head_ins, &A.o_top, &A.o_tmp, &A.o_stack};
    for (auto ps : scr) *ps = balloc<u32>(mb, NU, &ok);
    A.o_up = balloc<u32>(mb, NU, &ok); A.o_rank = balloc<u64>(mb, NU, &ok);
    A.p_tgt = balloc<u32>(mb, (size_t)U.tot_succ + 2, &ok);
    if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    rt_memset(A.p_tgt, 0xFF, ((size_t)U.tot_succ + 2) * 4, ctx->stream);
    mg(MK_AP_MAP, U.tot_rows);   // also flags the new rows (k_ap_new_flag's rule) into G.new_idx
    mg(MK_AP_CHG_META, ap->n_chunks);
  } else
  mg(MK_NEW_FLAG, SB.tot_rows);
  u32 tot_new = 0;
  if (!scan_inplace(mb, G.new_idx, SB.tot_rows, &tot_new)) return failm("scan failed");
  G.tot_new = tot_new;
  mg(MK_PAIR_COUNTS, P);
  u32 m_rows = 0, gap_tot = 0;
  if (!scan_inplace(mb, M.row_base, P, &m_rows) || !scan_inplace(mb, G.gap_base, P, &gap_tot)) return failm("scan failed");
  if (m_rows >= 0x7FFFFFF0u) { set_error(ctx, std::string(who) + ": batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
  M.tot_rows = m_rows;
  G.new_src = balloc<u32>(mb, (size_t)tot_new + 1, &ok); G.new_gap = balloc<u32>(mb, (size_t)tot_new + 1, &ok);
  G.gap_cnt = balloc<u32>(mb, (size_t)gap_tot + 2, &ok);
  size_t NR = (size_t)m_rows + 2;
  u32** rcols[] = {&M.r_id_actor, &M.r_id_ctr, &M.r_obj_actor, &M.r_obj_ctr, &M.r_key_actor, &M.r_key_ctr, &M.r_key_off, &M.r_key_len,
                   &M.r_vraw_off, &M.r_mark_off, &M.r_mark_len, &M.r_succ_cnt, &M.r_succ_base, &M.r_doc};
  for (auto pc : rcols) *pc = balloc<u32>(mb, NR, &ok);
  M.r_insert = balloc<u8>(mb, NR, &ok); M.r_action = balloc<u8>(mb, NR, &ok); M.r_expand = balloc<u8>(mb, NR, &ok);
  M.r_vmeta = balloc<u64>(mb, NR, &ok);
  if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
  rt_memset(G.gap_cnt, 0, ((size_t)gap_tot + 2) * 4, ctx->stream);
  rt_memset(G.new_src, 0xFF, ((size_t)tot_new + 1) * 4, ctx->stream);   // pairs given up by a kernel leave their slots empty
  rt_memset(G.new_gap, 0xFF, ((size_t)tot_new + 1) * 4, ctx->stream);
  rt_memset(M.r_succ_cnt, 0, NR * 4, ctx->stream);
#ifdef AMH_EMU
  mg(MK_SUBTREE, P);            // per-document stack walk
#else
  mg(MK_SUBTREE, DB.tot_rows);  // thread per row (g_mg_sub_*)
#endif
  pt.mark("exec: setup + subtree", &ctx->stream);
  if (ap) {
    u32 nr = G.U.tot_rows;
    G.A.g_head = balloc<u32>(mb, (size_t)gap_tot + 2, &ok); G.A.flag = balloc<u32>(mb, 4, &ok);
    if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    rt_memset(G.A.g_head, 0xFF, ((size_t)gap_tot + 2) * 4, ctx->stream);
    mg(MK_AP_LINK, nr);
    mg(MK_AP_CHILDREN, nr);
    mg(MK_AP_SORT_CHILDREN, nr);
    auto jump_until_still = [&](int k) {   // pointer jumping: distances halve per launch
      for (int round = 0; round < 40; round++) {
        u32 moved = 0;
        rt_memset(G.A.flag, 0, 4, ctx->stream);
        mg(k, nr);
        if (!rt_d2h(&moved, G.A.flag, 4, ctx->stream) || !rt_sync(ctx->stream)) return false;
        if (!moved) break;
      }
      return true;
    };
    mg(MK_AP_SUCC_INIT, nr);
    if (!jump_until_still(MK_AP_UP_JUMP)) return failm("D2H copy failed");
    mg(MK_AP_SUCC_BUILD, nr);
    if (!jump_until_still(MK_AP_RANK_JUMP)) return failm("D2H copy failed");
    mg(MK_AP_WEIGHT, nr);
    mg(MK_AP_TOP_PLACE, nr);
    mg(MK_AP_TIES, nr);
    pt.mark("exec: ap link .. top_place", &ctx->stream);
  } else {
    mg(MK_GAP, SB.tot_rows);
    mg(MK_INHERIT, P);
  }
  if (!scan_inplace(mb, G.gap_cnt, gap_tot, nullptr)) return failm("scan failed");
  if (ap) { mg(MK_AP_TOP_K, G.U.tot_rows); mg(MK_AP_EMIT, G.U.tot_rows); pt.mark("exec: ap_emit", &ctx->stream); }
  if (ap) {   // succ lists of the new rows themselves (a later op of the same call overwrote / deleted them)
    BatchDev& S = G.S;
    mg(MK_AP_PRED_CNT, G.U.tot_rows);
#ifdef AMH_EMU
    memcpy(S.r_succ_base, S.r_succ_cnt, (size_t)G.U.tot_rows * 4);
#else
    (void)hipMemcpyAsync(S.r_succ_base, S.r_succ_cnt, (size_t)G.U.tot_rows * 4, hipMemcpyDeviceToDevice, ctx->stream);
#endif
    u32 s_succ = 0;
    if (!scan_inplace(mb, S.r_succ_base, G.U.tot_rows, &s_succ)) return failm("scan failed");
    S.tot_succ = s_succ;
    S.s_actor = balloc<u32>(mb, (size_t)s_succ + 1, &ok); S.s_ctr = balloc<u32>(mb, (size_t)s_succ + 1, &ok);
    if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    mg(MK_AP_SUCC_FILL_U, G.U.tot_succ);
    mg(MK_AP_SUCC_SORT_U, G.U.tot_rows);
  } else
  mg(MK_EXTRA_CNT, SB.tot_succ);
  mg(MK_ROWS_DST, DB.tot_rows);
  mg(MK_ROWS_NEW, tot_new);
#ifdef AMH_EMU
  memcpy(M.r_succ_base, M.r_succ_cnt, (size_t)m_rows * 4);
#else
  (void)hipMemcpyAsync(M.r_succ_base, M.r_succ_cnt, (size_t)m_rows * 4, hipMemcpyDeviceToDevice, ctx->stream);
#endif
  u32 m_succ = 0;
  if (!scan_inplace(mb, M.r_succ_base, m_rows, &m_succ)) return failm("scan failed");
  M.tot_succ = m_succ;
  M.s_actor = balloc<u32>(mb, (size_t)m_succ + 1, &ok); M.s_ctr = balloc<u32>(mb, (size_t)m_succ + 1, &ok);
  if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
  mg(MK_SUCC_DST, DB.tot_rows);
  mg(MK_SUCC_NEW, tot_new);
  if (ap) mg(MK_AP_SUCC_FILL_D, G.U.tot_succ);
  else mg(MK_SUCC_EXTRA, SB.tot_succ);
  mg(MK_SUCC_SORT, DB.tot_rows);
  mg(MK_DOC_SUCC, P);
  mg(MK_CHANGES, M.tot_chg);
  mg(MK_COPY_CHUNKS, P);
  // pairs the kernels rejected
  if (!rt_d2h(p_status.data(), G.p_status, P * sizeof(i32), ctx->stream)) return failm("D2H copy failed");
  if (!rt_h2d(M.status, p_status.data(), P * sizeof(i32), ctx->stream)) return failm("H2D copy failed");
  pt.mark("exec: rows + succ + chunks", &ctx->stream);
  u32 nd = DB.n_docs;
  amh_batch* nb = nullptr;
  int rc = AMH_SUCCESS;
  // ---- the merged rows go straight into the rest of the load pipeline (id tables, linking, op -> change, the change
  //      re-encode and hashing): nothing is saved and decoded in between.  Taken when every document of dst is merged
  //      by this call; otherwise (documents outside the call, pairs with nothing to do, failed pairs) the merged
  //      documents go through their saved form below, next to the untouched ones.  AMH_MERGE_RELOAD=1 forces that. ---
  bool direct = getenv("AMH_MERGE_RELOAD") == nullptr && P == nd;
  for (u32 p = 0; p < P && direct; p++) direct = p_status[p] == 0 && p_a[p] == p && dst->status[p] == 0;
  if (direct) {
    nb = rows_to_batch(ctx, mb, dst, G, pl, ap, m_arena, who);
    if (!nb) return AMH_ERR_DEVICE;
    bool all_ok = true;
    for (u32 p = 0; p < P; p++) all_ok = all_ok && nb->status[p] == 0;
    if (all_ok) {
      adopt_merged_arrays(mb, nb, m_arena);
    } else {
      // A document merged from loaded documents verifies whenever its changes are re-derived (amh_merge_batch,
      // AMH_VERIFY_MERGE=1): anything else is an engine bug, never a silent result.  (amh_merge_many adopts the sources'
      // chunks by default and re-derives nothing: there all_ok only says that linking found the rows well-formed; the
      // row placement is then covered by the test suites, which run both modes against the oracle's saved bytes.)
      // A change handed to apply may not survive the document format (a pred naming no op, a non-canonical
      // encoding: the reference's own save + load rejects such a document too): those documents are left as they
      // were, and the rest goes through the saved form.
      if (!ap) { set_error(ctx, std::string(who) + ": a merged document failed verification"); amh_batch_free(nb); return AMH_ERR_DEVICE; }
      for (u32 p = 0; p < P; p++) if (nb->status[p] != 0) p_status[p] = AMH_E_UNSUPPORTED;
      amh_batch_free(nb);
      nb = nullptr;
      if (!rt_h2d(M.status, p_status.data(), P * sizeof(i32), ctx->stream)) return failm("H2D copy failed");
    }
    pt.mark("exec: rows -> pipeline", &ctx->stream);
  }
  if (!nb) {
  // ---- merged rows -> document chunks (the save kernels over M) ---------------------------------
  M.sv_status = balloc<i32>(mb, P + 1, &ok); M.sv_col_len = balloc<u32>(mb, (size_t)P * 25 + 1, &ok); M.sv_wide = balloc<u32>(mb, P + 1, &ok);
  M.sv_body_len = balloc<u32>(mb, P + 1, &ok); M.sv_off = balloc<u32>(mb, P + 2, &ok);
  if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
  SaveBigHost sbh;
  if (!save_big_prepare(ctx, mb, M, std::vector<i32>(), sbh)) return failm("save: job tables failed");
  launch(ctx, AK_SAVE_CHECK, M, P);
  launch(ctx, AK_SAVE_SIZE, M, P * 25);
  save_big_launch(ctx, M, 0);
  launch(ctx, AK_SAVE_LAYOUT, M, P);
  u32 sv_total = 0;
  if (!scan_inplace(mb, M.sv_off, P, &sv_total)) return failm("scan failed");
  if (sv_total == 0xFFFFFFFFu) { set_error(ctx, std::string(who) + ": merged documents exceed 4 GiB"); return AMH_ERR_BATCH_TOO_LARGE; }
  M.sv_out = balloc<u8>(mb, (size_t)sv_total + 256, &ok);
  if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
  launch(ctx, AK_SAVE_STORE, M, P * 25);
  save_big_launch(ctx, M, 1);
  M.skip_doc_sha = 1;   // these chunks only feed the reload below (a later save re-encodes from the rows)
  launch(ctx, AK_SAVE_FINISH, M, P);
  pt.mark("exec: save kernels", &ctx->stream);
  std::vector<u32> sv_off(P + 1), sv_body(P);
  std::vector<i32> sv_st(P);
  if (!rt_d2h(sv_off.data(), M.sv_off, (P + 1) * 4, ctx->stream) || !rt_d2h(sv_body.data(), M.sv_body_len, P * 4, ctx->stream) ||
      !rt_d2h(sv_st.data(), M.sv_status, P * 4, ctx->stream))
    return failm("D2H copy failed");
  collect_profile(ctx);
  pt.mark("exec: save D2H", &ctx->stream);
  // ---- reload: every document of dst, merged ones from their new bytes.  The arena of the new batch is put
  //      together in device memory (merged chunks from the save buffer, the others from the old arena): the
  //      bytes do not visit the host; its host copy is fetched only when an accessor asks for bytes. ---------
  amh_staged* old_st = dst->staged;
  if (!old_st->single_chunk && !ensure_host_arena(old_st)) return failm("D2H copy failed");   // where the first chunk of an input ends is read from its bytes
  for (int attempt = 0; attempt < 2; attempt++) {
  std::vector<u64> seg_src(nd), seg_dst(nd), se(2 * (size_t)nd + 2, 0);
  std::vector<u32> seg_len(nd);
  u64 off = 0;
  for (u32 d = 0; d < nd; d++) {
    u64 o0 = old_st->h_doc_off[2 * (size_t)d], o1 = old_st->h_doc_off[2 * (size_t)d + 1];
    size_t len = (size_t)(o1 - o0);
    if (!old_st->single_chunk) len = first_doc_chunk_len(old_st->h_arena.data() + o0, len);
    seg_src[d] = (u64)(uintptr_t)((const u8*)old_st->d_arena.p + o0);
    u32 p = d_pair[d];
    if (p != AMH_NULL32 && p_status[p] == 0) {
      if (sv_st[p] != 0) p_status[p] = sv_st[p];
      else {
        u32 bl = sv_body[p];
        seg_src[d] = (u64)(uintptr_t)(M.sv_out + sv_off[p]);
        len = 9 + bl + (bl < 0x80 ? 1 : bl < 0x4000 ? 2 : bl < 0x200000 ? 3 : bl < 0x10000000 ? 4 : 5);
      }
    }
    seg_len[d] = (u32)len;
    seg_dst[d] = off;
    se[2 * (size_t)d] = off; se[2 * (size_t)d + 1] = off + len;
    off = (off + len + 15) & ~(u64)15;
  }
  amh_load_opts lo;
  lo.verify_heads = 1; lo.on_partial_load = 0;
  {
    auto* st = new amh_staged();
    st->ctx = ctx; st->n_docs = nd; st->host_ok = false; st->single_chunk = true;
    st->internal = true;
    st->arena_bytes = (size_t)off + 256;
    st->h_doc_off = se;
    st->d_arena.p = ctx->pool.alloc(st->arena_bytes, &st->d_arena.cap);
    st->d_doc_off.p = ctx->pool.alloc(se.size() * sizeof(u64), &st->d_doc_off.cap);
    u64* d_src = balloc<u64>(mb, nd + 1, &ok); u64* d_dst = balloc<u64>(mb, nd + 1, &ok); u32* d_len = balloc<u32>(mb, nd + 1, &ok);
    if (!st->d_arena.p || !st->d_doc_off.p || !ok) { staged_unref(st); set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    // padding between and behind the chunks must read as zero only where the windowed decoders look past a column:
    // clearing the tail is enough, the gaps are below 16 bytes and belong to no column
    rt_memset((u8*)st->d_arena.p + off, 0, 256, ctx->stream);
    bool cok = rt_h2d(st->d_doc_off.p, se.data(), se.size() * sizeof(u64), ctx->stream) && rt_h2d(d_src, seg_src.data(), nd * sizeof(u64), ctx->stream) &&
               rt_h2d(d_dst, seg_dst.data(), nd * sizeof(u64), ctx->stream) && rt_h2d(d_len, seg_len.data(), nd * sizeof(u32), ctx->stream);
    if (!cok) { staged_unref(st); return failm("H2D copy failed"); }
    amh_copy_segments(d_src, d_len, d_dst, (u8*)st->d_arena.p, nd, (void*)ctx->stream);
    pt.mark("exec: reload arena (device)", &ctx->stream);
    nb = load_staged_impl(ctx, st, &lo, false);
    staged_unref(st);
  }
  if (!nb) return AMH_ERR_DEVICE;
  rc = AMH_SUCCESS;
  bool again = false;
  for (u32 p = 0; p < P; p++) {
    if (p_status[p] > 0) { rc = AMH_ERR_DOC_FAILED; continue; }
    if (p_status[p] == 0 && nb->status[p_a[p]] != 0) {
      // Documents merged from loaded documents always verify: anything else is an engine bug, never a silent result.
      // A change handed to apply may not survive the trip through the document format (its re-derived chunk
      // hashes differently: a pred naming no op, a non-canonical encoding — the reference's own save + load
      // rejects such a document too): that document is left as it was and reported.
      if (!ap || attempt) rc = AMH_ERR_DEVICE;
      else { p_status[p] = AMH_E_UNSUPPORTED; again = true; }
    }
  }
  if (rc == AMH_ERR_DEVICE) { set_error(ctx, std::string(who) + ": a merged document failed verification on reload"); amh_batch_free(nb); return rc; }
  if (!again) break;
  amh_batch_free(nb);
  nb = nullptr;
  }
  pt.mark("exec: reload pipeline", &ctx->stream);
  }
  for (u32 p = 0; p < P; p++) if (p_status[p] > 0) rc = AMH_ERR_DOC_FAILED;
  // documents of dst that failed earlier keep their status (a reload of their first chunk alone may succeed)
  bool st_dirty = false;
  for (u32 d = 0; d < nd; d++)
    if (dst->status[d] != 0 && nb->status[d] == 0) { nb->status[d] = dst->status[d]; nb->fail_line[d] = dst->fail_line[d]; st_dirty = true; }
  if (st_dirty && (!rt_h2d(nb->B.status, nb->status.data(), nd * sizeof(i32), ctx->stream) || !rt_sync(ctx->stream))) { amh_batch_free(nb); return AMH_ERR_DEVICE; }
  nb->queue = std::move(dst->queue);
  for (u32 p = 0; p < P; p++) pl.p_status[p] = p_status[p];
  std::swap(*dst, *nb);
  amh_batch_free(nb);
  if (rc == AMH_ERR_DOC_FAILED) set_error(ctx, std::string(who) + ": some pairs were not merged (failed documents, duplicate seq numbers or unsupported shapes); their dst documents are unchanged");
  return rc;
}
extern "C" int amh_merge_batch(amh_ctx* ctx, amh_batch* dst, const amh_batch* src, const uint32_t* dst_idx, const uint32_t* src_idx, size_t n_pairs) {
  if (!ctx || !dst || !src || dst == src || (n_pairs && (!dst_idx || !src_idx)) || n_pairs >= 0x7FFFFFFFull) { set_error(ctx, "amh_merge_batch: invalid argument"); return AMH_ERR_INVALID_ARG; }
  if (n_pairs == 0) return AMH_SUCCESS;
  AMH_NEEDS_OPS(dst, "amh_merge_batch")
  AMH_NEEDS_OPS(src, "amh_merge_batch")
  // A document with queued changes (orphans waiting for their dependencies) goes through the apply path, which offers
  // the queue again once the source's changes are in, as the reference's merge does (automerge.rs:821-825, 938).
  if (!dst->queue.empty()) {
    std::vector<u32> qd, qs, rd, rs;
    for (size_t p = 0; p < n_pairs; p++) {
      bool queued = dst->queue.count(dst_idx[p]) != 0;
      (queued ? qd : rd).push_back(dst_idx[p]);
      (queued ? qs : rs).push_back(src_idx[p]);
    }
    if (!qd.empty()) {
      int rc_q = amh_merge_many(ctx, dst, &src, 1, qd.data(), qs.data(), qd.size());
      if (rc_q != AMH_SUCCESS && rc_q != AMH_ERR_DOC_FAILED) return rc_q;
      std::string msg_q = rc_q ? ctx->last_error : std::string();
      int rc_r = rd.empty() ? AMH_SUCCESS : amh_merge_batch(ctx, dst, src, rd.data(), rs.data(), rd.size());
      if (rc_r == AMH_SUCCESS && rc_q != AMH_SUCCESS) set_error(ctx, msg_q);
      return rc_r != AMH_SUCCESS ? rc_r : rc_q;
    }
  }
  if (!ensure_chg(dst) || !ensure_chg(src) || !ensure_host_arena(dst->staged) || !ensure_host_arena(src->staged)) { set_error(ctx, "amh_merge_batch: D2H copy failed"); return AMH_ERR_DEVICE; }
  const HostView &ha = dst->hv, &hb = src->hv;
  const BatchDev &DB = dst->B, &SB = src->B;
  u32 P = (u32)n_pairs;
  MgPlan pl;
  mg_plan_init(pl, P, DB.n_docs, SB.n_docs, SB.tot_chg);
  MG_REFS(pl)
  for (u32 p = 0; p < P; p++) {
    u32 a = dst_idx[p], b = src_idx[p];
    if (a >= DB.n_docs || b >= SB.n_docs || d_pair[a] != AMH_NULL32 || s_pair[b] != AMH_NULL32) { set_error(ctx, "amh_merge_batch: document index out of range or used twice"); return AMH_ERR_INVALID_ARG; }
    d_pair[a] = p; s_pair[b] = p; p_a[p] = a; p_b[p] = b;
  }
  // ---- host: actors and changes -----------------------------------------------------------------
  for (u32 p = 0; p < P; p++) {
    u32 a = p_a[p], b = p_b[p];
    mapa_base[p] = (u32)map_a.size(); mapb_base[p] = (u32)map_b.size();
    m_actor_base[p] = (u32)m_actor_off.size(); m_chg_base[p] = (u32)mc_from.size();
    memset(&m_desc[p], 0, sizeof(DocDesc)); memset(&m_cnt[p], 0, sizeof(DocCounts));
    if (dst->status[a] || src->status[b]) { p_status[p] = dst->status[a] ? dst->status[a] : src->status[b]; n_failed++; continue; }
    const u8* ca = dst->staged->h_arena.data() + dst->staged->h_doc_off[2 * (size_t)a];
    const u8* cb = src->staged->h_arena.data() + src->staged->h_doc_off[2 * (size_t)b];
    u32 na = ha.desc[a].n_actors, nb = hb.desc[b].n_actors, aba = ha.actor_base[a], abb = hb.actor_base[b];
    auto abytes = [&](u32 i) { return std::make_pair(ca + ha.actor_off[aba + i], ha.actor_len[aba + i]); };
    auto bbytes = [&](u32 i) { return std::make_pair(cb + hb.actor_off[abb + i], hb.actor_len[abb + i]); };
    auto cmp = [](std::pair<const u8*, u32> x, std::pair<const u8*, u32> y) { int c = memcmp(x.first, y.first, std::min(x.second, y.second)); return c ? c : (x.second < y.second ? -1 : (x.second > y.second ? 1 : 0)); };
    bool sorted = true;
    for (u32 i = 1; i < na && sorted; i++) sorted = cmp(abytes(i - 1), abytes(i)) < 0;
    for (u32 i = 1; i < nb && sorted; i++) sorted = cmp(bbytes(i - 1), bbytes(i)) < 0;
    if (!sorted) { p_status[p] = AMH_E_UNSUPPORTED; n_failed++; continue; }
    u32 lenA = ha.desc[a].chunk_len, lenApad = (lenA + 15u) & ~15u, lenB = hb.desc[b].chunk_len;
    a_len_pad[p] = lenApad;
    // union of the two sorted tables
    std::vector<u32> la(na), lb(nb), lb2a(nb, AMH_NULL32);
    {
      u32 i = 0, j = 0, k = 0;
      while (i < na || j < nb) {
        int c = i >= na ? 1 : (j >= nb ? -1 : cmp(abytes(i), bbytes(j)));
        if (c <= 0) { m_actor_off.push_back(ha.actor_off[aba + i]); m_actor_len.push_back(ha.actor_len[aba + i]); }
        else { m_actor_off.push_back(lenApad + hb.actor_off[abb + j]); m_actor_len.push_back(hb.actor_len[abb + j]); }
        m_actor_rank.push_back(k);
        if (c == 0) lb2a[j] = i;
        if (c <= 0) la[i++] = k;
        if (c >= 0) lb[j++] = k;
        k++;
      }
    }
    map_a.insert(map_a.end(), la.begin(), la.end());
    map_b.insert(map_b.end(), lb.begin(), lb.end());
    map_b2a.insert(map_b2a.end(), lb2a.begin(), lb2a.end());
    // changes: which of src's are new, and in which order they are applied
    u32 nca = ha.cnt[a].n_chg, ncb = hb.cnt[b].n_chg, cba = ha.chg_base[a], cbb = hb.chg_base[b];
    std::unordered_map<Hash32, u32, Hash32Hasher> in_a;
    in_a.reserve(nca * 2 + 1);
    for (u32 i = 0; i < nca; i++) { Hash32 h; memcpy(h.b, ha.c_hash.data() + (size_t)(cba + i) * 32, 32); in_a.emplace(h, i); }
    std::vector<u32> b2a(ncb, AMH_NULL32);
    for (u32 j = 0; j < ncb; j++) { Hash32 h; memcpy(h.b, hb.c_hash.data() + (size_t)(cbb + j) * 32, 32); auto it = in_a.find(h); if (it != in_a.end()) b2a[j] = it->second; }
    auto bhash = [&](u32 j) { return hb.c_hash.data() + (size_t)(cbb + j) * 32; };
    auto bdeps_sorted = [&](u32 j) {   // deps in the order of their hashes (Change::deps is sorted)
      std::vector<u32> d;
      for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) d.push_back(hb.dep_idx[hb.c_dep_base[cbb + j] + k]);
      std::sort(d.begin(), d.end(), [&](u32 x, u32 y) { return memcmp(bhash(x), bhash(y), 32) < 0; });
      return d;
    };
    // get_changes_added: DFS from src's heads (sorted), popping from the end; result reversed
    std::vector<u8> is_dep_b(ncb, 0);
    for (u32 j = 0; j < ncb; j++) for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) is_dep_b[hb.dep_idx[hb.c_dep_base[cbb + j] + k]] = 1;
    std::vector<u32> stack;
    for (u32 j = 0; j < ncb; j++) if (!is_dep_b[j]) stack.push_back(j);
    std::sort(stack.begin(), stack.end(), [&](u32 x, u32 y) { return memcmp(bhash(x), bhash(y), 32) < 0; });
    std::vector<u8> seen(ncb, 0);
    std::vector<u32> added;
    while (!stack.empty()) {
      u32 j = stack.back(); stack.pop_back();
      if (seen[j] || b2a[j] != AMH_NULL32) continue;
      seen[j] = 1; added.push_back(j);
      for (u32 dj : bdeps_sorted(j)) stack.push_back(dj);
    }
    std::reverse(added.begin(), added.end());
    if (added.empty()) { p_status[p] = -1; continue; }   // nothing to do: dst stays as it is
    // apply_changes: duplicate-seq check, causal gate, queue drained with swap_remove
    std::vector<u32> per_actor(m_actor_off.size() - m_actor_base[p], 0);
    for (u32 i = 0; i < nca; i++) per_actor[la[ha.c_actor[cba + i]]]++;
    std::vector<u8> applied(ncb, 0);
    std::vector<u32> order, queue;
    auto ready = [&](u32 j) { for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) { u32 dj = hb.dep_idx[hb.c_dep_base[cbb + j] + k]; if (b2a[dj] == AMH_NULL32 && !applied[dj]) return false; } return true; };
    auto apply = [&](u32 j) { applied[j] = 1; order.push_back(j); per_actor[lb[hb.c_actor[cbb + j]]]++; };
    bool dup = false;
    for (u32 j : added) {
      if (applied[j]) continue;
      if (per_actor[lb[hb.c_actor[cbb + j]]] >= hb.c_seq[cbb + j]) { dup = true; break; }
      if (ready(j)) apply(j); else queue.push_back(j);
    }
    while (!dup) {
      bool got = false;
      for (size_t q = 0; q < queue.size(); q++)
        if (ready(queue[q])) { u32 j = queue[q]; queue[q] = queue.back(); queue.pop_back(); if (!applied[j]) apply(j); got = true; break; }
      if (!got) break;
    }
    if (dup) { p_status[p] = AMH_E_DUPLICATE_SEQ; n_failed++; continue; }
    if (!queue.empty()) { p_status[p] = AMH_E_MISSING_DEPS; n_failed++; continue; }
    // merged history = dst's history, then the applied changes
    std::vector<u32> b2m(ncb, AMH_NULL32);
    for (u32 j = 0; j < ncb; j++) b2m[j] = b2a[j];
    for (size_t k = 0; k < order.size(); k++) { b2m[order[k]] = nca + (u32)k; s_added[cbb + order[k]] = 1; }
    size_t first_c = mc_from.size();
    for (u32 i = 0; i < nca; i++) {
      mc_from.push_back(cba + i); mc_doc.push_back(p); mc_is_dep.push_back(0);
      mc_ndeps.push_back(ha.c_ndeps[cba + i]); mc_dep_base.push_back((u32)m_dep_idx.size());
      for (u32 k = 0; k < ha.c_ndeps[cba + i]; k++) m_dep_idx.push_back(ha.dep_idx[ha.c_dep_base[cba + i] + k]);
    }
    for (u32 j : order) {
      mc_from.push_back(0x80000000u | (cbb + j)); mc_doc.push_back(p); mc_is_dep.push_back(0);
      mc_ndeps.push_back(hb.c_ndeps[cbb + j]); mc_dep_base.push_back((u32)m_dep_idx.size());
      for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) m_dep_idx.push_back(b2m[hb.dep_idx[hb.c_dep_base[cbb + j] + k]]);
    }
    for (size_t c = first_c; c < mc_from.size(); c++)
      for (u32 k = 0; k < mc_ndeps[c]; k++) mc_is_dep[first_c + m_dep_idx[mc_dep_base[c] + k]] = 1;
    m_cnt[p].n_chg = nca + (u32)order.size();
    m_desc[p].chunk_off = arena_len;
    m_desc[p].chunk_len = lenApad + lenB;
    m_desc[p].n_actors = (u32)(m_actor_off.size() - m_actor_base[p]);
    m_desc[p].ocol[OC_MARK_NAME].len = ha.desc[a].ocol[OC_MARK_NAME].len | hb.desc[b].ocol[OC_MARK_NAME].len;
    arena_len += (lenApad + lenB + 15u) & ~15ull;
  }
  mapa_base[P] = (u32)map_a.size(); mapb_base[P] = (u32)map_b.size();
  m_actor_base[P] = (u32)m_actor_off.size(); m_chg_base[P] = (u32)mc_from.size();
  bool any_work = false;
  for (u32 p = 0; p < P; p++) any_work = any_work || p_status[p] == 0;
  if (!any_work) {
    if (n_failed) { set_error(ctx, "amh_merge_batch: no pair could be merged (failed documents / duplicate seq numbers)"); return AMH_ERR_DOC_FAILED; }
    return AMH_SUCCESS;
  }
  if (arena_len >= 0xFFFFFFF0ull || mc_from.size() >= 0x7FFFFFFFull) { set_error(ctx, "amh_merge_batch: batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
  return mg_execute(ctx, dst, SB, pl, nullptr, "amh_merge_batch");
}
// ---------------------------------------------------------------------------------------------
// apply_changes / load_incremental / chunks behind the first one of a loaded input
// ---------------------------------------------------------------------------------------------
enum { APM_STRICT = 0, APM_PREFIX = 1, APM_LOAD_ERROR = 2, APM_LOAD_IGNORE = 3 };
struct ApplyItem {
  u32 doc = 0;
  int mode = APM_STRICT;
  bool first_was_doc = true;
  bool one_call = false;       // all buffers together are one apply_changes call (else: one call per buffer)
  std::vector<std::pair<const u8*, size_t>> bufs;   // each: one or more concatenated chunks
  // change chunks that are already in device memory (merge: the re-derived chunks of the source batches): each entry is
  // one whole uncompressed change chunk; entries with the same `buf` form one apply_changes call.  They are gathered
  // into the staged arena by a device copy and never cross to the host as a batch's change bytes.
  struct DevChunk { const u8* d; u32 len; u32 buf; const BatchDev* sb; u32 chg; };   // sb / chg: the loaded batch and the change (global index) the chunk is: its hash and op / pred counts are there   // hash: the chunk's SHA-256 (= change hash) where the source batch keeps it, device memory
  std::vector<DevChunk> dev;
};
// a table filled in full by its owner: sized without the zero fill (and the page faults on one thread) of std::vector::resize
template <class T> struct RawVec {
  std::unique_ptr<T[]> p; T* q = nullptr; size_t n = 0;
  void resize(size_t k) { p.reset(k ? new T[k] : nullptr); q = p.get(); n = k; }
  void attach(T* ext, size_t k) { p.reset(); q = ext; n = k; }   // storage owned by someone else (the context's pinned table area)
  T* data() const { return q; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T& operator[](size_t i) const { return q[i]; }
};
template <class T> struct PinSpan { T* p = nullptr; T& operator[](size_t i) const { return p[i]; } T* data() const { return p; } };
struct HostChunk {
  u32 pair = 0;
  const u8* p = nullptr;       // whole type-1 chunk (caller memory, the queue, or `owned`)
  size_t len = 0;
  std::vector<u8> owned;
  i32 bad = 0;                 // host-level failure of this position of the stream (nothing staged)
  bool from_queue = false;
  bool is_doc = false;         // a document chunk: replaced by the changes it holds before staging
  u32 buf = 0;                 // which of the item's buffers it came from (one buffer = one apply_changes call)
  const u8* dev = nullptr;     // the chunk lies in device memory (p is set once the staged arena is back on the host)
  const BatchDev* dev_sb = nullptr;   // ... as change dev_chg of that loaded batch, which holds its hash and its op / pred counts
  u32 dev_chg = 0;
  u32 staged = AMH_NULL32;     // index in U
};
static bool host_uleb(const u8* p, size_t& pos, size_t end, u64& out) {
  out = 0;
  for (u32 shift = 0; shift < 64; shift += 7) {
    if (pos >= end) return false;
    u8 b = p[pos++];
    out |= (u64)(b & 0x7f) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
static i32 change_err_kind(i32 dev_status) {   // storage/load.rs:15-30 as seen through load_changes
  if (dev_status == 0 || dev_status == AMH_E_PARSE || dev_status == AMH_E_BAD_CHECKSUM || dev_status >= AMH_E_TOO_LARGE) return dev_status;
  return AMH_E_INVALID_CHANGE_COLUMNS;
}
// split a byte stream into chunks (storage/load.rs:56-114); stops at the first position that is not a chunk
// `sink` takes every chunk in stream order
template <class Sink>
static void split_chunks(u32 pair, const u8* p, size_t len, Sink&& sink) {
  size_t off = 0;
  while (off < len) {
    HostChunk hc;
    hc.pair = pair;
    const u8* q = p + off;
    size_t avail = len - off;
    size_t pos = 9;
    u64 dlen = 0;
    if (avail < 10 || q[0] != 0x85 || q[1] != 0x6f || q[2] != 0x4a || q[3] != 0x83 || q[8] > 2 || !host_uleb(q, pos, avail, dlen) || dlen > avail - pos) {
      hc.bad = AMH_E_PARSE;
      sink(std::move(hc));
      return;
    }
    size_t total = pos + (size_t)dlen;
    if (q[8] == 0) {   // a document chunk inside a change stream stands for its changes (load.rs:84-100): expanded by apply_core
      hc.p = q; hc.len = total; hc.is_doc = true;
      sink(std::move(hc));
      off += total;
      continue;
    }
    if (q[8] == 2) {
      if (!amh_ingest_inflate_change(q, total, hc.owned)) { hc.bad = AMH_E_PARSE; sink(std::move(hc)); return; }
      hc.p = nullptr; hc.len = hc.owned.size();
    } else {
      hc.p = q; hc.len = total;
    }
    sink(std::move(hc));
    off += total;
  }
}

// results[i] = outcome for items[i] (0, or the AutomergeError kind); documents with outcome != 0 are left as
// they were, except under APM_LOAD_ERROR where the document's status becomes the outcome.
static int apply_core(amh_ctx* ctx, amh_batch* dst, std::vector<ApplyItem>& items, std::vector<i32>& results, const char* who) {
  u32 P = (u32)items.size();
  results.assign(P, 0);
  if (P == 0) return AMH_SUCCESS;
  AMH_NEEDS_OPS(dst, who)
  PhaseTimer pt(who);
  // of the documents' own bytes the decisions further down read the actor tables: packed on the device when the bytes
  // are not on the host already
  const bool use_pack = !dst->staged->host_ok;
  if (use_pack && !ensure_actor_pack(dst)) { set_error(ctx, std::string(who) + ": D2H copy failed"); return AMH_ERR_DEVICE; }
  pt.mark("apply: actor tables of dst (device pack)");
  // The host view of dst's history (hashes, deps, actors: ~100 bytes per change) is fetched here, before anything else
  // runs.  (It used to come back on a helper thread beside the staging below; once that staging ran on all host threads
  // the helper's copies took 33 ms instead of 6.5 and the decisions waited for them.)
  if (!ensure_chg(dst)) { set_error(ctx, std::string(who) + ": D2H copy failed"); return AMH_ERR_DEVICE; }
  pt.mark("apply: host view of dst");
  const HostView& ha = dst->hv;
  const BatchDev& DB = dst->B;
  // ---- chunks of every pair: the new ones, then the document's queue -------------------------------
  // The chunk list of a wide merge is 700,000 entries = 60 MB: a fresh block of that size is an mmap whose pages fault in
  // one by one while the entries are constructed, and is unmapped again at the end.  The calling thread keeps the storage
  // between calls (up to 256 MB; a nested call takes a list of its own).
  static thread_local std::vector<HostChunk> t_chunks;
  static thread_local bool t_chunks_busy = false;
  std::vector<HostChunk> own_chunks;
  const bool cached = !t_chunks_busy;
  std::vector<HostChunk>& chunks = cached ? t_chunks : own_chunks;
  // A list of device-resident chunks owns no memory: its entries are left in place for the next call, which then
  // constructs nothing and overwrites every field (in parallel); any other list is cleared.
  struct ChunkListGuard {
    std::vector<HostChunk>& v; bool cached; bool keep = false;
    ~ChunkListGuard() {
      if (!cached) return;
      if (!keep) v.clear();
      if (v.capacity() * sizeof(HostChunk) > ((size_t)256 << 20)) std::vector<HostChunk>().swap(v);
      t_chunks_busy = false;
    }
  } chunk_guard{chunks, cached};
  if (cached) t_chunks_busy = true;
  std::vector<u32> pc_base(P + 1, 0);
  bool any_doc = false;   // a document chunk among the incoming chunks
  bool dev_only = dst->queue.empty();   // every item names device-resident chunks only (amh_merge_many of loaded batches): no stream to split
  for (u32 p = 0; p < P && dev_only; p++) dev_only = items[p].bufs.empty();
  if (dev_only) {
    size_t total = 0;
    for (u32 p = 0; p < P; p++) { pc_base[p] = (u32)total; total += items[p].dev.size(); }
    chunks.resize(total);   // entries left by the last such call stay as they are
    chunk_guard.keep = true;
    pt.mark("apply: chunk list sized");
    host_parallel_for(P, [&](u32 p) {
      HostChunk* to = chunks.data() + pc_base[p];
      for (auto& dc : items[p].dev) {
        HostChunk& hc = *to++;
        hc.pair = p; hc.p = nullptr; hc.len = dc.len; hc.bad = 0; hc.from_queue = false; hc.is_doc = false; hc.buf = dc.buf;
        hc.dev = dc.d; hc.dev_sb = dc.sb; hc.dev_chg = dc.chg; hc.staged = AMH_NULL32;   // (`owned` is never used on this path)
      }
    });
  } else {
    // two walks over the streams' chunk headers: count per pair, then (places by prefix sums) every pair writes its
    // stretch of the list itself — no list per pair, nothing constructed that is not used; entries left by the last call
    // are overwritten whole (their `owned` buffers go with the assignment)
    std::atomic<bool> saw_owned{false}, saw_doc{false};
    auto walk_pair = [&](u32 p, HostChunk* to) -> u32 {   // to = nullptr: count
      u32 n = 0, bi = 0;
      auto emit = [&](HostChunk&& hc) {
        if (to) {
          HostChunk& d = to[n];
          d = std::move(hc);
          if (!d.owned.empty()) { saw_owned = true; if (!d.p && !d.bad) d.p = d.owned.data(); }
          if (d.is_doc) saw_doc = true;
        }
        n++;
      };
      for (auto& bf : items[p].bufs) {
        const u32 before = n;
        bool bad = false;
        split_chunks(p, bf.first, bf.second, [&](HostChunk&& hc) { hc.buf = bi; bad = hc.bad != 0; emit(std::move(hc)); });   // (the count walk inflates a deflated chunk as well: the same decisions in both walks)
        if (!items[p].one_call) bi++;
        if (n > before && bad) break;   // nothing behind a bad position is looked at
      }
      for (auto& dc : items[p].dev) {
        HostChunk hc;
        hc.pair = p; hc.dev = dc.d; hc.dev_sb = dc.sb; hc.dev_chg = dc.chg; hc.len = dc.len; hc.buf = bi + dc.buf;
        emit(std::move(hc));
      }
      auto qit = dst->queue.find(items[p].doc);
      if (qit != dst->queue.end())
        for (auto& qb : qit->second) {
          HostChunk hc;
          hc.pair = p; hc.p = qb.data(); hc.len = qb.size(); hc.from_queue = true;
          emit(std::move(hc));
        }
      return n;
    };
    std::vector<u32> p_cnt(P);
    host_parallel_for(P, [&](u32 p) { p_cnt[p] = walk_pair(p, nullptr); });
    size_t total = 0;
    for (u32 p = 0; p < P; p++) { pc_base[p] = (u32)total; total += p_cnt[p]; }
    chunks.resize(total);
    pt.mark("apply: chunk list sized");
    host_parallel_for(P, [&](u32 p) { walk_pair(p, chunks.data() + pc_base[p]); });
    any_doc = saw_doc;
    chunk_guard.keep = !saw_owned && !saw_doc;   // a list that owns memory (inflated chunks, expanded documents) is cleared at the end
  }
  pt.mark("apply: split");
  // document chunks inside the streams (load_incremental of a whole saved document): loaded as a batch of their own
  // without the heads check (load.rs:90-93), each then stands for its re-derived changes
  if (any_doc) {   // (known from the split: the list is not walked to find that out)
    std::vector<u32> docs_at;
    for (u32 k = 0; k < chunks.size(); k++) if (chunks[k].is_doc) docs_at.push_back(k);
    if (!docs_at.empty()) {
      std::vector<const uint8_t*> dp;
      std::vector<size_t> dl;
      for (u32 k : docs_at) { dp.push_back(chunks[k].p); dl.push_back(chunks[k].len); }
      amh_load_opts lo;
      lo.verify_heads = 0; lo.on_partial_load = 0;
      amh_staged* est = amh_stage_batch(ctx, dp.data(), dl.data(), dp.size());
      if (!est) return AMH_ERR_DEVICE;
      amh_batch* eb = load_staged_impl(ctx, est, &lo, false);
      staged_unref(est);
      if (!eb) return AMH_ERR_DEVICE;
      std::unique_ptr<amh_batch, void (*)(amh_batch*)> ebh(eb, amh_batch_free);
      if (!ensure_chg(eb)) return AMH_ERR_DEVICE;
      HostView& ev = eb->hv;
      if (!ev.out_ok) { if (!fetch(eb, ev.out_arena, eb->B.out_arena, (size_t)eb->tot_out) || !rt_sync(ctx->stream)) return AMH_ERR_DEVICE; ev.out_ok = true; }
      std::vector<HostChunk> expanded;
      expanded.reserve(chunks.size());
      u32 di = 0;
      for (u32 k = 0; k < chunks.size(); k++) {
        HostChunk& hc = chunks[k];
        if (!hc.is_doc) { expanded.push_back(std::move(hc)); continue; }
        u32 e = di++;
        if (eb->status[e]) {   // the stream stops here with this document's error (load.rs:94-99)
          HostChunk bad;
          bad.pair = hc.pair; bad.buf = hc.buf; bad.bad = eb->status[e] == AMH_E_BAD_CHECKSUM || eb->status[e] == AMH_E_PARSE ? eb->status[e] : AMH_E_INVALID_CHANGE_COLUMNS;
          expanded.push_back(std::move(bad));
          // everything of the same buffer behind it is dropped, as split_chunks does at a bad position
          u32 k2 = k + 1;
          while (k2 < chunks.size() && chunks[k2].pair == hc.pair && !chunks[k2].from_queue) k2++;
          k = k2 - 1;
          continue;
        }
        u32 cb = ev.chg_base[e], nc = ev.cnt[e].n_chg;
        for (u32 i = 0; i < nc; i++) {
          HostChunk c;
          c.pair = hc.pair; c.buf = hc.buf;
          const u8* q = ev.out_arena.data() + ev.c_out_off[cb + i];
          c.owned.assign(q, q + ev.c_out_len[cb + i]);
          c.len = c.owned.size();
          expanded.push_back(std::move(c));
        }
      }
      chunks.swap(expanded);
      std::fill(pc_base.begin(), pc_base.end(), 0);
      for (auto& hc : chunks) pc_base[hc.pair + 1]++;
      for (u32 p = 0; p < P; p++) pc_base[p + 1] += pc_base[p];
    }
  }
  pc_base[P] = (u32)chunks.size();
  if (any_doc) for (auto& hc : chunks) if (!hc.p && !hc.bad && !hc.dev) hc.p = hc.owned.data();   // the changes of expanded documents
  // ---- stage the good ones -----------------------------------------------------------------------
  std::unique_ptr<amh_batch, void (*)(amh_batch*)> holder(new amh_batch(), amh_batch_free);
  amh_batch* hb = holder.get();
  hb->ctx = ctx;
  ApplyRun ap;
  HVec<u64> doc_off;
  struct StageView { u8* p = nullptr; size_t n = 0; u8* data() const { return p; } size_t size() const { return n; } } h_arena;
  ap.s_desc.resize(P); ap.s_cnt.resize(P); ap.s_row_base.assign(P + 1, 0);
  HVec<u32> staged_of;   // U index -> chunks index (every entry written by the parallel fill below: no zero fill)
  RawVec<u64> seg_nops, seg_npred;   // where the op / pred counts of a device-resident chunk lie (0 = not known)
  RawVec<u64> seg_src, seg_dst, seg_hash, seg_hdst;   // seg_hash / seg_hdst: where the hash of a device-resident chunk lies / goes in U.doc_hash
  RawVec<u32> seg_len;
  size_t n_host_staged = 0;
  std::vector<u32> p_n(P + 1, 0);   // per pair: its staged chunks, then (prefix sums) the U index of its first one
  {
    // per pair: staged chunks, device-resident ones among them, bytes of its region (parallel); places by prefix sums;
    // then every pair fills its own stretch of the tables (parallel)
    std::vector<u32> p_dev(P + 1, 0);
    std::vector<u64> p_bytes(P + 1, 0);
    host_parallel_for(P, [&](u32 p) {
      u32 n = 0, nd = 0; u64 by = 0;
      for (u32 k = pc_base[p]; k < pc_base[p + 1]; k++) {
        const HostChunk& hc = chunks[k];
        if (hc.bad) continue;
        n++; nd += hc.dev != nullptr; by += ((u64)hc.len + 15) & ~(u64)15;
      }
      p_n[p] = n; p_dev[p] = nd; p_bytes[p] = by;
    });
    u64 off = 0;
    u32 nu = 0, ndev = 0;
    for (u32 p = 0; p < P; p++) {
      if (p_bytes[p] >= 0xFFFFFFF0ull) { set_error(ctx, std::string(who) + ": changes of one document exceed 4 GiB"); return AMH_ERR_BATCH_TOO_LARGE; }
      if ((u64)nu + p_n[p] >= 0x7FFFFFF0ull) { set_error(ctx, std::string(who) + ": batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
      const u64 by = p_bytes[p];
      const u32 n = p_n[p], nd = p_dev[p];
      ap.s_desc[p].chunk_off = off; ap.s_desc[p].chunk_len = (u32)by;
      p_n[p] = nu; p_dev[p] = ndev; p_bytes[p] = off;
      nu += n; ndev += nd; off += by;
    }
    p_n[P] = nu; p_dev[P] = ndev;
    n_host_staged = nu - ndev;
    staged_of.resize(nu); ap.c_pair.resize(nu); ap.c_rel.resize(nu); doc_off.resize(2 * (size_t)nu);
    // 52 bytes per device-resident chunk, 36 MB for the 700,000 chunks of the headline step: written straight into pinned
    // memory (kept by the context, grow only), so that their uploads run at the link's rate and do not hold the host
    {
      const size_t per = ((size_t)ndev * 8 + 255) & ~(size_t)255, need = 7 * per + 256;
      if (ndev >= 4096 && need > ctx->tabs_cap) {
        rt_sync(ctx->stream);
        if (ctx->tabs) rt_host_free(ctx->tabs);
        ctx->tabs_cap = need + need / 4;
        ctx->tabs = (u8*)rt_host_alloc(ctx->tabs_cap);
        if (!ctx->tabs) ctx->tabs_cap = 0;
      }
      if (ndev >= 4096 && ctx->tabs) {
        u8* t = ctx->tabs;
        seg_src.attach((u64*)t, ndev); seg_dst.attach((u64*)(t + per), ndev); seg_hash.attach((u64*)(t + 2 * per), ndev); seg_hdst.attach((u64*)(t + 3 * per), ndev);
        seg_nops.attach((u64*)(t + 4 * per), ndev); seg_npred.attach((u64*)(t + 5 * per), ndev); seg_len.attach((u32*)(t + 6 * per), ndev);
      } else {
        seg_src.resize(ndev); seg_len.resize(ndev); seg_dst.resize(ndev); seg_hash.resize(ndev); seg_hdst.resize(ndev); seg_nops.resize(ndev); seg_npred.resize(ndev);
      }
    }
    pt.mark("apply: stage layout");
    if ((size_t)off + 256 > ctx->stage_cap) {
      rt_sync(ctx->stream);   // nothing may still be copying out of the old area
      if (ctx->stage) rt_host_free(ctx->stage);
      ctx->stage_cap = ((size_t)off + 256) + ((size_t)off + 256) / 4;
      ctx->stage = (u8*)rt_host_alloc(ctx->stage_cap);
      if (!ctx->stage) { ctx->stage_cap = 0; set_error(ctx, std::string(who) + ": out of host memory"); return AMH_ERR_OOM; }
    }
    h_arena.p = ctx->stage; h_arena.n = (size_t)off + 256;   // read until the decisions below are made
    memset(h_arena.data() + off, 0, 256);
    host_parallel_for(P, [&](u32 p) {
      u32 u = p_n[p], sd = p_dev[p];
      const u64 region = p_bytes[p];
      u64 o = region;
      for (u32 k = pc_base[p]; k < pc_base[p + 1]; k++) {
        HostChunk& hc = chunks[k];
        if (hc.bad) continue;
        hc.staged = u;
        staged_of[u] = k; ap.c_pair[u] = p; ap.c_rel[u] = (u32)(o - region);
        doc_off[2 * (size_t)u] = o; doc_off[2 * (size_t)u + 1] = o + hc.len;
        const u64 next = (o + hc.len + 15) & ~(u64)15;
        if (hc.dev) {   // gathered on the device; the host's picture of the arena receives its head only
          hc.p = h_arena.data() + o;
          seg_src[sd] = (u64)(uintptr_t)hc.dev; seg_len[sd] = (u32)hc.len; seg_dst[sd] = o;
          seg_hdst[sd] = (u64)u * 32;
          if (hc.dev_sb) {
            seg_hash[sd] = (u64)(uintptr_t)(hc.dev_sb->c_hash + (size_t)hc.dev_chg * 32);
            seg_nops[sd] = (u64)(uintptr_t)(hc.dev_sb->c_nops + hc.dev_chg);
            seg_npred[sd] = hc.dev_sb->c_npred ? (u64)(uintptr_t)(hc.dev_sb->c_npred + hc.dev_chg) : 0;
          } else { seg_hash[sd] = 0; seg_nops[sd] = 0; seg_npred[sd] = 0; }
          sd++;
        } else {        // chunk bytes + the alignment gap behind them
          memcpy(h_arena.data() + o, hc.p, hc.len);
          memset(h_arena.data() + o + hc.len, 0, (size_t)(next - (o + hc.len)));
        }
        o = next; u++;
      }
    });
  }
  u32 NCH = (u32)staged_of.size();
  ap.n_chunks = NCH;
  pt.mark("apply: stage copy");
  // what comes back per chunk lands in pinned memory (pageable copies run at a fraction of the link's rate)
  PinSpan<i32> u_status, u_status2; PinSpan<ChgHdr> hdr; PinSpan<DocCounts> u_cnt; PinSpan<u8> u_hash; PinSpan<u32> u_row_base, h_head_len, h_head_off;
  {
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t need = 2 * al((size_t)NCH * sizeof(i32)) + al((size_t)NCH * sizeof(ChgHdr)) + al((size_t)NCH * sizeof(DocCounts)) + al((size_t)NCH * 32 + 32) + 3 * al(((size_t)NCH + 1) * sizeof(u32)) + 64;
    if (need > ctx->res_cap) {
      rt_sync(ctx->stream);
      if (ctx->res) rt_host_free(ctx->res);
      ctx->res_cap = need + need / 4;
      ctx->res = (u8*)rt_host_alloc(ctx->res_cap);
      if (!ctx->res) { ctx->res_cap = 0; set_error(ctx, std::string(who) + ": out of host memory"); return AMH_ERR_OOM; }
    }
    u8* q = ctx->res;
    u_status.p = (i32*)q; q += al((size_t)NCH * sizeof(i32));
    u_status2.p = (i32*)q; q += al((size_t)NCH * sizeof(i32));
    hdr.p = (ChgHdr*)q; q += al((size_t)NCH * sizeof(ChgHdr));
    u_cnt.p = (DocCounts*)q; q += al((size_t)NCH * sizeof(DocCounts));
    u_hash.p = q; q += al((size_t)NCH * 32 + 32);
    u_row_base.p = (u32*)q; q += al(((size_t)NCH + 1) * sizeof(u32));
    h_head_len.p = (u32*)q; q += al(((size_t)NCH + 1) * sizeof(u32));
    h_head_off.p = (u32*)q;
    u_row_base[0] = 0;
  }
  BatchDev& U = ap.U;
  bool ok = true;
  auto faild = [&](const char* m) { set_error(ctx, std::string(who) + ": " + m); collect_profile(ctx); return AMH_ERR_DEVICE; };
  MergeDev G0;
  memset(&G0, 0, sizeof G0);
  MergeDev* G0dev = nullptr;
  if (NCH) {
    U.n_docs = NCH;
    U.no_checksum = items[0].mode == APM_STRICT;
    U.max_doc_rows = ctx->max_doc_rows;
    u8* d_arena = balloc<u8>(hb, h_arena.size(), &ok);
    u64* d_off = balloc<u64>(hb, doc_off.size() + 2, &ok);
    U.status = balloc<i32>(hb, NCH + 1, &ok); U.fail_line = balloc<u32>(hb, NCH + 1, &ok);
    U.desc = balloc<DocDesc>(hb, NCH + 1, &ok); U.cnt = balloc<DocCounts>(hb, NCH + 1, &ok);
    U.row_base = balloc<u32>(hb, NCH + 2, &ok); U.succ_base = balloc<u32>(hb, NCH + 2, &ok);
    U.doc_hash = balloc<u8>(hb, (size_t)NCH * 32 + 32, &ok); U.doc_max_op = balloc<u32>(hb, NCH + 1, &ok);
    ap.d_hdr = balloc<ChgHdr>(hb, NCH + 1, &ok);
    G0dev = (MergeDev*)balloc<u8>(hb, sizeof(MergeDev), &ok);
    if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    U.arena = d_arena; U.doc_off = d_off;
    if (!rt_h2d(d_off, doc_off.data(), doc_off.size() * sizeof(u64), ctx->stream)) return faild("H2D copy failed");
    if (n_host_staged || seg_src.empty()) { if (!rt_h2d(d_arena, h_arena.data(), h_arena.size(), ctx->stream)) return faild("H2D copy failed"); }
    else rt_memset(d_arena, 0, h_arena.size(), ctx->stream);   // alignment gaps and the tail read as zero
    if (!seg_src.empty()) {
      // device-resident chunks: gathered into the arena on the device; the host's copy of the arena (actor ids and
      // dependency hashes are read from it when the decisions are made) comes back on the side stream while the
      // chunks are being decoded
      u32 ns = (u32)seg_src.size();
      u64* d_src = balloc<u64>(hb, ns + 1, &ok); u64* d_dst = balloc<u64>(hb, ns + 1, &ok); u32* d_len = balloc<u32>(hb, ns + 1, &ok);
      if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
      if (!rt_h2d(d_src, seg_src.data(), ns * sizeof(u64), ctx->stream) || !rt_h2d(d_dst, seg_dst.data(), ns * sizeof(u64), ctx->stream) ||
          !rt_h2d(d_len, seg_len.data(), ns * sizeof(u32), ctx->stream))
        return faild("H2D copy failed");
      // the sources are re-derived chunks where they lie in a loaded batch's out_arena (k_chg_layout packs them back to
      // back: any alignment), so the copy is the byte-exact one: nothing behind a chunk's last byte is read, and the
      // alignment gap behind its copy keeps the zeros of the memset above (g_ap_head_copy relies on them)
      amh_copy_bytes64(d_src, d_len, d_dst, d_arena, ns, (void*)ctx->stream);
    }
    rt_memset(U.status, 0, (NCH + 1) * sizeof(i32), ctx->stream);
    rt_memset(U.fail_line, 0, (NCH + 1) * sizeof(u32), ctx->stream);
    rt_memset(U.doc_max_op, 0, (NCH + 1) * sizeof(u32), ctx->stream);
    std::unique_ptr<MergeDev> G0_sent(new MergeDev);
    bool G0_any = false;
    auto mg0 = [&](int k, u32 n) {
      if (n == 0) return true;
      G0.U = U; G0.A.hdr = ap.d_hdr; G0.A.u_chg = ap.d_u_chg; G0.A.u_fill = ap.d_u_fill;
      if (!G0_any || memcmp(G0_sent.get(), &G0, sizeof(MergeDev)) != 0) {
        if (!rt_h2d(G0dev, &G0, sizeof(MergeDev), ctx->stream)) return false;
        memcpy(G0_sent.get(), &G0, sizeof(MergeDev));
        G0_any = true;
      }
#ifdef AMH_EMU
      launch_mg_timed(ctx, k, &G0, n);
#else
      launch_mg_timed(ctx, k, G0dev, n);
#endif
      return true;
    };
    mg0(MK_AP_PARSE, NCH);
    // chunk hashes (= change hashes).  When every chunk comes out of a loaded batch (amh_merge_many) the hashes were
    // computed, and the checksums written, when those batches were loaded: they are copied, not taken again (1.3 GB of
    // SHA-256 per headline step)
    bool hashes_given = !seg_src.empty() && n_host_staged == 0 && seg_src.size() == NCH;
    for (size_t k = 0; hashes_given && k < seg_hash.size(); k++) hashes_given = seg_hash[k] != 0;
    if (hashes_given) {
      const u32 ns = (u32)seg_hash.size();
      u64* d_hs = balloc<u64>(hb, ns + 1, &ok); u64* d_hd = balloc<u64>(hb, ns + 1, &ok); u32* d_hn = balloc<u32>(hb, ns + 1, &ok);
      if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
      std::vector<u32> len32(ns, 32);
      if (!rt_h2d(d_hs, seg_hash.data(), ns * sizeof(u64), ctx->stream) || !rt_h2d(d_hd, seg_hdst.data(), ns * sizeof(u64), ctx->stream) ||
          !rt_h2d(d_hn, len32.data(), ns * sizeof(u32), ctx->stream) || !rt_sync(ctx->stream))
        return faild("H2D copy failed");
      amh_copy_bytes64(d_hs, d_hn, d_hd, U.doc_hash, ns, (void*)ctx->stream);
    } else {
      launch(ctx, AK_SHA_DOC, U, NCH);
    }
    u32 *d_hl = nullptr, *d_ho = nullptr;
    u8* d_heads = nullptr;
    u32 heads_total = 0;
    if (!seg_src.empty()) {
      // Of the device-resident chunks the host reads the heads only (author and other actors, dependency hashes): those
      // are packed on the device and come back with the other results of this pass; the rest of a chunk is fetched only
      // if it ends up in the document's queue (commit_queues).
      d_hl = balloc<u32>(hb, (size_t)NCH + 2, &ok); d_ho = balloc<u32>(hb, (size_t)NCH + 2, &ok);
      if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
      amh_ap_head_len(ap.d_hdr, U.status, d_hl, NCH, (void*)ctx->stream);
      if (!rt_d2d(d_ho, d_hl, (size_t)NCH * sizeof(u32), ctx->stream)) return faild("device copy failed");
      if (!scan_inplace(hb, d_ho, NCH, &heads_total)) return faild("scan failed");
      if (heads_total == 0xFFFFFFFFu) { set_error(ctx, std::string(who) + ": batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
      d_heads = balloc<u8>(hb, (size_t)heads_total + 16, &ok);
      if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
      if ((size_t)heads_total + 16 > ctx->heads_cap) {
        if (ctx->heads) rt_host_free(ctx->heads);
        ctx->heads_cap = ((size_t)heads_total + 16) + ((size_t)heads_total + 16) / 4;
        ctx->heads = (u8*)rt_host_alloc(ctx->heads_cap);
        if (!ctx->heads) { ctx->heads_cap = 0; set_error(ctx, std::string(who) + ": out of host memory"); return AMH_ERR_OOM; }
      }
      amh_ap_head_copy(d_arena, d_off, d_hl, d_ho, d_heads, NCH, (void*)ctx->stream);
    }
    // op and pred counts per chunk: known where the chunks come out of loaded batches that counted them (k_chg_layout)
    bool counts_given = hashes_given;
    for (size_t k = 0; counts_given && k < seg_nops.size(); k++) counts_given = seg_nops[k] != 0 && seg_npred[k] != 0;
    if (counts_given) {
      const u32 ns = (u32)seg_nops.size();
      u64* d_a = balloc<u64>(hb, 3 * (size_t)ns + 3, &ok);
      if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
      if (!rt_h2d(d_a, seg_nops.data(), ns * sizeof(u64), ctx->stream) || !rt_h2d(d_a + ns, seg_npred.data(), ns * sizeof(u64), ctx->stream) ||
          !rt_h2d(d_a + 2 * (size_t)ns, seg_hdst.data(), ns * sizeof(u64), ctx->stream) || !rt_sync(ctx->stream))
        return faild("H2D copy failed");
      amh_ap_counts_given(U, d_a, d_a + ns, d_a + 2 * (size_t)ns, ns, (void*)ctx->stream);
    } else {
      mg0(MK_AP_COUNT, NCH);
    }
    u32 tot_rows = 0, tot_preds = 0;
    for (int attempt = 0;; attempt++) {
      if (!scan_inplace(hb, U.row_base, NCH, &tot_rows) || !scan_inplace(hb, U.succ_base, NCH, &tot_preds)) return faild("scan failed");
      const u64 budget = batch_byte_budget(ctx);
      if (tot_rows < 0x7FFFFFF0u && tot_preds < 0x7FFFFFF0u && 2 * derived_bytes(tot_rows, tot_preds, 0, 0, 0) <= budget) break;
      if (attempt) { set_error(ctx, std::string(who) + ": batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
      // the largest chunks are failed (TooLarge) until the rest fits; the counts are taken again (k_ap_count skips them)
      std::vector<DocCounts> cn(NCH);
      std::vector<i32> st(NCH);
      if (!rt_d2h(cn.data(), U.cnt, (size_t)NCH * sizeof(DocCounts), ctx->stream) || !rt_d2h(st.data(), U.status, (size_t)NCH * sizeof(i32), ctx->stream) || !rt_sync(ctx->stream))
        return faild("D2H copy failed");
      std::vector<std::pair<u64, u32>> weight(NCH);
      u64 total = 0;
      for (u32 c = 0; c < NCH; c++) { weight[c] = {st[c] ? 0 : 2 * derived_bytes(cn[c].n_rows, cn[c].n_succ, 0, 0, 0), c}; total += weight[c].first; }
      std::sort(weight.begin(), weight.end(), [](const std::pair<u64, u32>& x, const std::pair<u64, u32>& y) { return x.first > y.first; });
      for (u32 k = 0; k < NCH && (total > budget || total / 448 >= 0x7FFFFFF0ull); k++) { total -= weight[k].first; st[weight[k].second] = AMH_E_TOO_LARGE; }
      if (!rt_h2d(U.status, st.data(), (size_t)NCH * sizeof(i32), ctx->stream) || !rt_sync(ctx->stream)) return faild("H2D copy failed");
      mg0(MK_AP_COUNT, NCH);
    }
    U.tot_rows = tot_rows; U.tot_succ = tot_preds;
    size_t NR = (size_t)tot_rows + 2, NS = (size_t)tot_preds + 2;
    u32** rcols[] = {&U.r_id_actor, &U.r_id_ctr, &U.r_obj_actor, &U.r_obj_ctr, &U.r_key_actor, &U.r_key_ctr, &U.r_key_off, &U.r_key_len,
                     &U.r_vraw_off, &U.r_mark_off, &U.r_mark_len, &U.r_succ_cnt, &U.r_succ_base, &U.r_doc};
    for (auto pc : rcols) *pc = balloc<u32>(hb, NR, &ok);
    U.r_insert = balloc<u8>(hb, NR, &ok); U.r_action = balloc<u8>(hb, NR, &ok); U.r_expand = balloc<u8>(hb, NR, &ok);
    U.r_vmeta = balloc<u64>(hb, NR, &ok);
    U.s_actor = balloc<u32>(hb, NS, &ok); U.s_ctr = balloc<u32>(hb, NS, &ok); U.s_row = balloc<u32>(hb, NS, &ok);
    ap.d_u_chg = balloc<u32>(hb, NR, &ok); ap.d_u_fill = balloc<u32>(hb, NR, &ok);
    if (!ok) { set_error(ctx, std::string(who) + ": device out of memory"); return AMH_ERR_OOM; }
    // What the decisions need is there once the chunks are parsed, hashed and counted: it comes back now, and the ops
    // are decoded and verified (k_ap_verify can still fail a chunk) WHILE the host decides; the decisions are checked
    // against the final statuses afterwards and made again in the rare case that a chunk failed late.
    if (!seg_src.empty() && (!rt_d2h(ctx->heads, d_heads, (size_t)heads_total, ctx->stream) || !rt_d2h(h_head_len.data(), d_hl, (size_t)NCH * sizeof(u32), ctx->stream) ||
                             !rt_d2h(h_head_off.data(), d_ho, (size_t)NCH * sizeof(u32), ctx->stream)))
      return faild("D2H copy failed");
    if (!rt_d2h(u_status.data(), U.status, NCH * sizeof(i32), ctx->stream) || !rt_d2h(hdr.data(), ap.d_hdr, NCH * sizeof(ChgHdr), ctx->stream) ||
        !rt_d2h(u_cnt.data(), U.cnt, NCH * sizeof(DocCounts), ctx->stream) ||
        !rt_d2h(u_hash.data(), U.doc_hash, (size_t)NCH * 32, ctx->stream) || !rt_d2h(u_row_base.data(), U.row_base, (NCH + 1) * sizeof(u32), ctx->stream) ||
        !rt_sync(ctx->stream))
      return faild("device pipeline failed");
    rt_memset(U.r_doc, 0xFF, NR * 4, ctx->stream);
    rt_memset(U.r_key_off, 0, NR * 4, ctx->stream); rt_memset(U.r_key_len, 0xFF, NR * 4, ctx->stream);
    rt_memset(U.r_mark_off, 0, NR * 4, ctx->stream); rt_memset(U.r_mark_len, 0xFF, NR * 4, ctx->stream);
#ifdef AMH_EMU
    rt_memset(U.r_obj_actor, 0xFF, NR * 4, ctx->stream); rt_memset(U.r_obj_ctr, 0xFF, NR * 4, ctx->stream);
    rt_memset(U.r_key_actor, 0xFF, NR * 4, ctx->stream); rt_memset(U.r_key_ctr, 0xFF, NR * 4, ctx->stream);
    rt_memset(U.r_expand, 0, NR, ctx->stream);
#endif
    rt_memset(U.s_row, 0xFF, NS * 4, ctx->stream);
    launch(ctx, AK_DECODE_OPS, U, NCH * OC_COUNT);
    mg0(MK_AP_VERIFY, tot_rows);
    if (!seg_src.empty())   // the heads go to their places in the host's picture of the arena
      host_parallel_for(NCH, [&](u32 u) {
        if (chunks[staged_of[u]].dev && h_head_len[u]) memcpy(h_arena.data() + doc_off[2 * (size_t)u], ctx->heads + h_head_off[u], h_head_len[u]);
      });
  }
  pt.mark("apply: U parse + count (decode + verify launched)");
  // ---- host: which changes apply, in which order (automerge.rs:797-858); actors; merged history -----
  MgPlan pl;
  MG_REFS(pl)
  std::vector<std::vector<u32>> new_queue;
  std::vector<u8> queue_dirty;
  bool verify_pending = NCH != 0;
  for (int pass = 0;; pass++) {
  pl = MgPlan();
  mg_plan_init(pl, P, DB.n_docs, P, NCH);
  results.assign(P, 0);
  ap.ix_actor.clear(); ap.ix_start.clear(); ap.ix_nops.clear(); ap.ix_row.clear(); ap.amap.clear();
  ap.c_applied.assign((size_t)NCH + 1, 0);
  ap.h_start_op.resize(NCH); ap.h_nops.resize(NCH);
  ap.h_chunk_off.resize(NCH); ap.h_chunk_len.resize(NCH);
  host_parallel_for(NCH, [&](u32 u) {
    ap.h_start_op[u] = hdr[u].start_op; ap.h_nops[u] = u_cnt[u].n_rows;
    ap.h_chunk_off[u] = doc_off[2 * (size_t)u]; ap.h_chunk_len[u] = (u32)(doc_off[2 * (size_t)u + 1] - doc_off[2 * (size_t)u]);
  }, 4096, 1024);
  ap.all_dev = NCH != 0 && n_host_staged == 0 && seg_src.size() == NCH;
  ap.amap_base.assign((size_t)NCH + 1, 0);
  ap.ix_base.assign(P + 1, 0);
  new_queue.assign(P, std::vector<u32>());
  queue_dirty.assign(P, 0);
  typedef std::pair<const u8*, u32> Bytes;
  auto bcmp = [](Bytes x, Bytes y) { int c = memcmp(x.first, y.first, std::min(x.second, y.second)); return c ? c : (x.second < y.second ? -1 : (x.second > y.second ? 1 : 0)); };
  // per staged chunk: author + other actors as byte ranges of h_arena
  auto chunk_actors = [&](u32 u, std::vector<Bytes>& out) {
    const u8* q = h_arena.data() + doc_off[2 * (size_t)u];
    out.clear();
    out.push_back(Bytes(q + hdr[u].actor_off, hdr[u].actor_len));
    size_t pos = hdr[u].others_off, end = (size_t)(doc_off[2 * (size_t)u + 1] - doc_off[2 * (size_t)u]);
    for (u32 i = 0; i < hdr[u].n_other; i++) {
      u64 l = 0;
      host_uleb(q, pos, end, l);
      out.push_back(Bytes(q + pos, (u32)l));
      pos += (size_t)l;
    }
  };
  host_parallel_for(P, [&](u32 p) {
    u32 a = items[p].doc;
    d_pair[a] = p; s_pair[p] = p; p_a[p] = a; p_b[p] = p;
    memset(&m_desc[p], 0, sizeof(DocDesc)); memset(&m_cnt[p], 0, sizeof(DocCounts));
    // rows of U per pair (the chunks of a pair are staged next to each other, pair after pair)
    const u32 rb0 = u_row_base[p_n[p]], rb1 = u_row_base[p_n[p + 1]];
    ap.s_row_base[p] = rb0;
    if (p + 1 == P) ap.s_row_base[P] = rb1;
    ap.s_cnt[p].n_rows = rb1 - rb0;
  }, 1024, 256);
  // Pairs are decided in blocks of DEC_BLK by a few host threads; a block appends into one set of vectors (with the
  // offsets at which each of its pairs starts) and reuses one set of scratch containers, so the join below is a few
  // bulk appends per block and a pair costs no allocations of its own.
  struct HashIndex {   // change hash -> position in the merged history; keys are pointers to 32 stable bytes
    std::vector<u32> slot, val;
    std::vector<const u8*> key;
    u32 mask = 0;
    void reset(size_t n) { size_t cap = 16; while (cap < 2 * n + 2) cap <<= 1; slot.assign(cap, 0); key.clear(); val.clear(); mask = (u32)cap - 1; }
    const u32* find(const u8* h) const {
      u64 k; memcpy(&k, h, 8);
      for (u32 i = (u32)((k * 0x9E3779B97F4A7C15ull) >> 40) & mask; slot[i]; i = (i + 1) & mask)
        if (memcmp(key[slot[i] - 1], h, 32) == 0) return &val[slot[i] - 1];
      return nullptr;
    }
    void emplace(const u8* h, u32 v) {
      u64 k; memcpy(&k, h, 8);
      u32 i = (u32)((k * 0x9E3779B97F4A7C15ull) >> 40) & mask;
      for (; slot[i]; i = (i + 1) & mask) if (memcmp(key[slot[i] - 1], h, 32) == 0) return;
      key.push_back(h); val.push_back(v); slot[i] = (u32)key.size();
    }
  };
  typedef std::pair<const u8*, u32> BytesT;
  struct Ix { u32 actor, start, nops, row; };
  struct PairOut {
    std::vector<u32> map_a, map_b, map_b2a, m_actor_off, m_actor_len, m_actor_rank, mc_from, mc_ndeps, mc_dep_base, m_dep_idx, mc_doc;
    std::vector<u8> mc_is_dep;
    std::vector<u32> ix_actor, ix_start, ix_nops, ix_row;
    std::vector<u32> amap, amap_u, amap_at;                  // actor maps of the applied chunks, flat; chunk u's starts at amap_at
    std::vector<u32> p_mapa, p_mapb, p_actor, p_chg, p_ix;   // per pair of the block: where its entries start in the vectors above
    std::vector<u64> p_arena;                                // per pair: bytes of the merged arena it needs (0 = none)
    int failed = 0;
    // scratch, cleared per pair
    HashIndex history;
    std::vector<std::pair<BytesT, u64>> per_actor;
    std::vector<u32> fresh, oldq, order, queue, la, lb, lb2a;
    std::vector<BytesT> sact, tmp;
    std::vector<Ix> ix;
  };
  const u32 DEC_BLK = 64;
  const u32 n_blk = (P + DEC_BLK - 1) / DEC_BLK;
  std::vector<PairOut> outs(n_blk);
  auto decide = [&](u32 p, PairOut& o) {
    o.p_mapa.push_back((u32)o.map_a.size()); o.p_mapb.push_back((u32)o.map_b.size()); o.p_actor.push_back((u32)o.m_actor_off.size());
    o.p_chg.push_back((u32)o.mc_from.size()); o.p_ix.push_back((u32)o.ix_actor.size()); o.p_arena.push_back(0);
    u32 a = items[p].doc;
    int mode = items[p].mode;
    auto &map_a = o.map_a, &map_b = o.map_b, &map_b2a = o.map_b2a, &m_actor_off = o.m_actor_off, &m_actor_len = o.m_actor_len, &m_actor_rank = o.m_actor_rank;
    auto &mc_from = o.mc_from, &mc_ndeps = o.mc_ndeps, &mc_dep_base = o.mc_dep_base, &m_dep_idx = o.m_dep_idx, &mc_doc = o.mc_doc;
    auto& mc_is_dep = o.mc_is_dep;
    if (dst->status[a]) { p_status[p] = dst->status[a]; results[p] = dst->status[a]; o.failed = 1; return; }
    // usable new chunks under the call's policy
    auto &fresh = o.fresh, &oldq = o.oldq;
    fresh.clear(); oldq.clear();
    i32 first_err = 0;
    for (u32 k = pc_base[p]; k < pc_base[p + 1]; k++) {
      HostChunk& hc = chunks[k];
      i32 e = hc.bad ? hc.bad : change_err_kind(u_status[hc.staged]);
      if (hc.from_queue) { if (!e) oldq.push_back(k); continue; }
      if (e) { first_err = e; break; }
      fresh.push_back(k);
    }
    if (first_err) {
      if (mode == APM_STRICT && first_err < AMH_E_TOO_LARGE) first_err = AMH_E_PARSE;   // change::LoadError::Parse covers header and column errors alike
      if (mode == APM_STRICT || mode == APM_LOAD_ERROR) { p_status[p] = first_err; results[p] = first_err; o.failed = 1; return; }
      if (mode == APM_LOAD_IGNORE) { p_status[p] = -1; return; }   // nothing behind the first chunk is applied (automerge.rs:690-694)
    }
    u32 na = ha.desc[a].n_actors, aba = ha.actor_base[a];
    // actor i of the document: in the packed tables (which start at the document's first actor id) or in the host's copy of its bytes
    const u8* ca = use_pack ? ha.actor_pack.data() + ha.actor_pack_off[a] - (na ? ha.actor_off[aba] : 0) : dst->staged->h_arena.data() + dst->staged->h_doc_off[2 * (size_t)a];
    auto abytes = [&](u32 i) { return Bytes(ca + ha.actor_off[aba + i], ha.actor_len[aba + i]); };
    bool sorted = true;
    for (u32 i = 1; i < na && sorted; i++) sorted = bcmp(abytes(i - 1), abytes(i)) < 0;
    if (!sorted) { p_status[p] = AMH_E_UNSUPPORTED; results[p] = AMH_E_UNSUPPORTED; o.failed = 1; return; }
    u32 nca = ha.cnt[a].n_chg, cba = ha.chg_base[a];
    HashIndex& history = o.history;
    history.reset((size_t)nca + (pc_base[p + 1] - pc_base[p]));
    for (u32 i = 0; i < nca; i++) history.emplace(ha.c_hash.data() + (size_t)(cba + i) * 32, i);
    auto& per_actor = o.per_actor;   // actor -> number of its changes in the history (states, automerge.rs:771-779); a handful per document
    per_actor.clear();
    auto actor_slot = [&](Bytes ab) -> u64& {
      for (auto& e : per_actor) if (bcmp(e.first, ab) == 0) return e.second;
      per_actor.push_back({ab, 0});
      return per_actor.back().second;
    };
    for (u32 i = 0; i < nca; i++) actor_slot(abytes(ha.c_actor[cba + i]))++;
    auto hash_of = [&](u32 k) { return (const u8*)u_hash.data() + (size_t)chunks[k].staged * 32; };
    auto actor_of = [&](u32 k) { u32 u = chunks[k].staged; const u8* q = h_arena.data() + doc_off[2 * (size_t)u]; return Bytes(q + hdr[u].actor_off, hdr[u].actor_len); };
    auto ready = [&](u32 k) {
      u32 u = chunks[k].staged;
      const u8* q = h_arena.data() + doc_off[2 * (size_t)u] + hdr[u].deps_off;
      for (u32 i = 0; i < hdr[u].n_deps; i++) if (!history.find(q + 32 * (size_t)i)) return false;
      return true;
    };
    auto &order = o.order, &queue = o.queue;
    order.clear();
    queue = oldq;
    auto apply = [&](u32 k) { history.emplace(hash_of(k), nca + (u32)order.size()); order.push_back(k); actor_slot(actor_of(k))++; };
    bool dup = false;
    auto drain = [&]() {   // pop_next_causally_ready_change (automerge.rs:849-858): linear scan + swap_remove
      for (;;) {
        bool got = false;
        for (size_t q = 0; q < queue.size(); q++)
          if (ready(queue[q])) { u32 k = queue[q]; queue[q] = queue.back(); queue.pop_back(); if (!history.find(hash_of(k))) apply(k); got = true; break; }
        if (!got) break;
      }
    };
    for (size_t fi = 0; fi < fresh.size() && !dup; fi++) {
      u32 k = fresh[fi];
      if (!history.find(hash_of(k))) {
        if (actor_slot(actor_of(k)) >= hdr[chunks[k].staged].seq) { dup = true; break; }
        if (ready(k)) apply(k); else queue.push_back(k);
      }
      // every buffer is one apply_changes call: the queue is drained at its end
      if (fi + 1 == fresh.size() || chunks[fresh[fi + 1]].buf != chunks[k].buf) drain();
    }
    if (fresh.empty() && !dup) drain();
    if (dup) {
      // apply_changes returns from the middle of its loop (automerge.rs:804-812): the changes in front of the offender
      // stay applied, those it pushed stay queued, the queue is not drained and nothing behind the offender is looked
      // at.  A document load whose trailing changes do this fails as a whole (load_with_options returns the error).
      results[p] = AMH_E_DUPLICATE_SEQ;
      if (mode == APM_LOAD_ERROR || mode == APM_LOAD_IGNORE) { p_status[p] = AMH_E_DUPLICATE_SEQ; o.failed = 1; return; }
    }
    if (mode == APM_LOAD_ERROR && !items[p].first_was_doc && !queue.empty()) { p_status[p] = AMH_E_MISSING_DEPS; results[p] = AMH_E_MISSING_DEPS; o.failed = 1; return; }
    new_queue[p] = queue; queue_dirty[p] = 1;
    if (order.empty()) { p_status[p] = -1; return; }
    // actor table of S for this pair: the actors of the applied changes, bytewise sorted
    auto &sact = o.sact, &tmp = o.tmp;
    sact.clear();
    for (u32 k : order) { chunk_actors(chunks[k].staged, tmp); sact.insert(sact.end(), tmp.begin(), tmp.end()); }
    std::sort(sact.begin(), sact.end(), [&](Bytes x, Bytes y) { return bcmp(x, y) < 0; });
    sact.erase(std::unique(sact.begin(), sact.end(), [&](Bytes x, Bytes y) { return bcmp(x, y) == 0; }), sact.end());
    u32 nb = (u32)sact.size();
    const u8* region = h_arena.data() + ap.s_desc[p].chunk_off;
    u32 lenA = ha.desc[a].chunk_len, lenApad = (lenA + 15u) & ~15u, lenB = ap.s_desc[p].chunk_len;
    a_len_pad[p] = lenApad;
    auto &la = o.la, &lb = o.lb, &lb2a = o.lb2a;
    la.assign(na, 0); lb.assign(nb, 0); lb2a.assign(nb, AMH_NULL32);
    {
      u32 i = 0, j = 0, k = 0;
      while (i < na || j < nb) {
        int c = i >= na ? 1 : (j >= nb ? -1 : bcmp(abytes(i), sact[j]));
        if (c <= 0) { m_actor_off.push_back(ha.actor_off[aba + i]); m_actor_len.push_back(ha.actor_len[aba + i]); }
        else { m_actor_off.push_back(lenApad + (u32)(sact[j].first - region)); m_actor_len.push_back(sact[j].second); }
        m_actor_rank.push_back(k);
        if (c == 0) lb2a[j] = i;
        if (c <= 0) la[i++] = k;
        if (c >= 0) lb[j++] = k;
        k++;
      }
    }
    map_a.insert(map_a.end(), la.begin(), la.end());
    map_b.insert(map_b.end(), lb.begin(), lb.end());
    map_b2a.insert(map_b2a.end(), lb2a.begin(), lb2a.end());
    auto s_index = [&](Bytes x) { return (u32)(std::lower_bound(sact.begin(), sact.end(), x, [&](Bytes y, Bytes z) { return bcmp(y, z) < 0; }) - sact.begin()); };
    // merged history = dst's history, then the applied changes
    size_t first_c = mc_from.size();
    for (u32 i = 0; i < nca; i++) {
      mc_from.push_back(cba + i); mc_doc.push_back(p); mc_is_dep.push_back(0);
      mc_ndeps.push_back(ha.c_ndeps[cba + i]); mc_dep_base.push_back((u32)m_dep_idx.size());
      for (u32 k = 0; k < ha.c_ndeps[cba + i]; k++) m_dep_idx.push_back(ha.dep_idx[ha.c_dep_base[cba + i] + k]);
    }
    u32 mark_len = ha.desc[a].ocol[OC_MARK_NAME].len;
    auto& ix = o.ix;
    ix.clear();
    for (u32 k : order) {
      u32 u = chunks[k].staged;
      ap.c_applied[u] = 1;
      mc_from.push_back(0x80000000u | u); mc_doc.push_back(p); mc_is_dep.push_back(0);
      mc_ndeps.push_back(hdr[u].n_deps); mc_dep_base.push_back((u32)m_dep_idx.size());
      const u8* q = h_arena.data() + doc_off[2 * (size_t)u] + hdr[u].deps_off;
      for (u32 i = 0; i < hdr[u].n_deps; i++) m_dep_idx.push_back(*history.find(q + 32 * (size_t)i));
      mark_len |= hdr[u].mark_len;
      chunk_actors(u, tmp);
      o.amap_u.push_back(u); o.amap_at.push_back((u32)o.amap.size());
      for (auto& x : tmp) o.amap.push_back(s_index(x));
      ix.push_back(Ix{s_index(tmp[0]), (u32)hdr[u].start_op, u_cnt[u].n_rows, u_row_base[u]});
    }
    std::sort(ix.begin(), ix.end(), [](const Ix& x, const Ix& y) { return x.actor != y.actor ? x.actor < y.actor : x.start < y.start; });
    for (auto& e : ix) { o.ix_actor.push_back(e.actor); o.ix_start.push_back(e.start); o.ix_nops.push_back(e.nops); o.ix_row.push_back(e.row); }
    for (size_t c = first_c; c < mc_from.size(); c++)
      for (u32 k = 0; k < mc_ndeps[c]; k++) mc_is_dep[first_c + m_dep_idx[mc_dep_base[c] + k]] = 1;
    m_cnt[p].n_chg = nca + (u32)order.size();
    m_desc[p].chunk_len = lenApad + lenB;
    m_desc[p].n_actors = (u32)m_actor_off.size() - o.p_actor.back();
    m_desc[p].ocol[OC_MARK_NAME].len = mark_len;
    o.p_arena.back() = (lenApad + lenB + 15u) & ~15ull;
    };
  pt.mark("apply: decisions: setup");
  host_parallel_for(n_blk, [&](u32 blk) {
    for (u32 p = blk * DEC_BLK, e = std::min(P, p + DEC_BLK); p < e; p++) decide(p, outs[blk]);
  }, 4, 1);
  pt.mark("apply: decisions: per pair (threads)");
  {   // join: every block's vectors go to their places in the plan (places by prefix sums, copies in parallel)
    struct Base { size_t mapa, mapb, actor, chg, dep, ix, amap; };
    std::vector<Base> base(n_blk + 1);
    Base run{map_a.size(), map_b.size(), m_actor_off.size(), mc_from.size(), m_dep_idx.size(), ap.ix_actor.size(), ap.amap.size()};
    for (u32 blk = 0; blk < n_blk; blk++) {
      PairOut& o = outs[blk];
      base[blk] = run;
      for (u32 p = blk * DEC_BLK, e = std::min(P, p + DEC_BLK), i = 0; p < e; p++, i++)
        if (o.p_arena[i]) { m_desc[p].chunk_off = arena_len; arena_len += o.p_arena[i]; }
      n_failed += o.failed;
      run.mapa += o.map_a.size(); run.mapb += o.map_b.size(); run.actor += o.m_actor_off.size(); run.chg += o.mc_from.size();
      run.dep += o.m_dep_idx.size(); run.ix += o.ix_actor.size(); run.amap += o.amap.size();
    }
    base[n_blk] = run;
    map_a.resize(run.mapa); map_b.resize(run.mapb); map_b2a.resize(run.mapb);
    m_actor_off.resize(run.actor); m_actor_len.resize(run.actor); m_actor_rank.resize(run.actor);
    mc_from.resize(run.chg); mc_ndeps.resize(run.chg); mc_dep_base.resize(run.chg); mc_doc.resize(run.chg); mc_is_dep.resize(run.chg);
    m_dep_idx.resize(run.dep);
    ap.ix_actor.resize(run.ix); ap.ix_start.resize(run.ix); ap.ix_nops.resize(run.ix); ap.ix_row.resize(run.ix);
    ap.amap.resize(run.amap);
    host_parallel_for(n_blk, [&](u32 blk) {
      PairOut& o = outs[blk];
      const Base& bs = base[blk];
      for (u32 p = blk * DEC_BLK, e = std::min(P, p + DEC_BLK), i = 0; p < e; p++, i++) {
        mapa_base[p] = (u32)bs.mapa + o.p_mapa[i]; mapb_base[p] = (u32)bs.mapb + o.p_mapb[i];
        m_actor_base[p] = (u32)bs.actor + o.p_actor[i]; m_chg_base[p] = (u32)bs.chg + o.p_chg[i];
        ap.ix_base[p] = (u32)bs.ix + o.p_ix[i];
      }
      for (auto& v : o.mc_dep_base) v += (u32)bs.dep;
      auto put = [](auto& dst, size_t at, const auto& src) { std::copy(src.begin(), src.end(), dst.begin() + at); };
      put(map_a, bs.mapa, o.map_a); put(map_b, bs.mapb, o.map_b); put(map_b2a, bs.mapb, o.map_b2a);
      put(m_actor_off, bs.actor, o.m_actor_off); put(m_actor_len, bs.actor, o.m_actor_len); put(m_actor_rank, bs.actor, o.m_actor_rank);
      put(mc_from, bs.chg, o.mc_from); put(mc_ndeps, bs.chg, o.mc_ndeps); put(mc_dep_base, bs.chg, o.mc_dep_base);
      put(mc_doc, bs.chg, o.mc_doc); put(mc_is_dep, bs.chg, o.mc_is_dep); put(m_dep_idx, bs.dep, o.m_dep_idx);
      put(ap.ix_actor, bs.ix, o.ix_actor); put(ap.ix_start, bs.ix, o.ix_start); put(ap.ix_nops, bs.ix, o.ix_nops); put(ap.ix_row, bs.ix, o.ix_row);
      for (size_t i = 0; i < o.amap_u.size(); i++) ap.amap_base[o.amap_u[i]] = (u32)bs.amap + o.amap_at[i];
      put(ap.amap, bs.amap, o.amap);
      o = PairOut();
    }, 2, 1);
  }
  ap.amap_base[NCH] = (u32)ap.amap.size();
  mapa_base[P] = (u32)map_a.size(); mapb_base[P] = (u32)map_b.size();
  m_actor_base[P] = (u32)m_actor_off.size(); m_chg_base[P] = (u32)mc_from.size();
  ap.ix_base[P] = (u32)ap.ix_actor.size();
  if (!verify_pending) break;
  // the ops are decoded and verified by now: the statuses the decisions were made with against the final ones
  verify_pending = false;
  if (!rt_d2h(u_status2.data(), ap.U.status, (size_t)NCH * sizeof(i32), ctx->stream) || !rt_sync(ctx->stream)) { set_error(ctx, std::string(who) + ": device pipeline failed"); collect_profile(ctx); return AMH_ERR_DEVICE; }
  collect_profile(ctx);
  if (memcmp(u_status.data(), u_status2.data(), (size_t)NCH * sizeof(i32)) == 0) break;
  memcpy(u_status.data(), u_status2.data(), (size_t)NCH * sizeof(i32));
  }
  bool queue_fetch_failed = false;
  auto commit_queues = [&]() {
    for (u32 p = 0; p < P; p++) {
      if (!queue_dirty[p] || (p_status[p] > 0)) continue;
      std::vector<std::vector<u8>> nq;
      for (u32 k : new_queue[p]) {
        if (chunks[k].dev) {   // only its head is on the host
          nq.emplace_back(chunks[k].len);
          if (!rt_d2h(nq.back().data(), ap.U.arena + doc_off[2 * (size_t)chunks[k].staged], chunks[k].len, ctx->stream)) queue_fetch_failed = true;
        } else
        nq.emplace_back(chunks[k].p, chunks[k].p + chunks[k].len);
        memcpy(nq.back().data() + 4, u_hash.data() + (size_t)chunks[k].staged * 32, 4);   // a change accepted with a stale checksum keeps its content hash
      }
      if (nq.empty()) dst->queue.erase(items[p].doc); else dst->queue[items[p].doc] = std::move(nq);
    }
  };
  bool any_work = false;
  for (u32 p = 0; p < P; p++) any_work = any_work || p_status[p] == 0;
  int rc = AMH_SUCCESS;
  pt.mark("apply: decisions: join");
  if (any_work) {
    if (arena_len >= 0xFFFFFFF0ull || mc_from.size() >= 0x7FFFFFFFull) { set_error(ctx, std::string(who) + ": batch too large"); return AMH_ERR_BATCH_TOO_LARGE; }
    BatchDev SB;
    memset(&SB, 0, sizeof SB);
    SB = U;   // S shares U's row arrays; its per-pair tables and succ lists are set up by mg_execute
    SB.n_docs = P; SB.tot_chg = NCH;
    rc = mg_execute(ctx, dst, SB, pl, &ap, who);
    if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) return rc;
    for (u32 p = 0; p < P; p++) if (p_status[p] > 0 && results[p] == 0) results[p] = p_status[p];
  }
  commit_queues();
  if (queue_fetch_failed) { set_error(ctx, std::string(who) + ": D2H copy failed"); return AMH_ERR_DEVICE; }
  for (u32 p = 0; p < P; p++) if (results[p]) rc = AMH_ERR_DOC_FAILED;
  if (rc == AMH_ERR_DOC_FAILED) set_error(ctx, std::string(who) + ": some documents were left unchanged, or (duplicate seq number) changed only by the changes in front of the offender; see the per-document results");
  return rc;
}

static int load_trailing_chunks(amh_ctx* ctx, amh_batch* b, const amh_load_opts* opts) {
  const amh_staged* st = b->staged;
  std::vector<ApplyItem> items;
  // where every input's first chunk ends comes from the parse kernel (4 bytes per document): the host only needs the
  // input bytes of the few documents that have something behind it
  std::vector<u32> first_len(b->B.n_docs);
  if (b->B.n_docs && (!b->B.first_len || !rt_d2h(first_len.data(), b->B.first_len, (size_t)b->B.n_docs * sizeof(u32), ctx->stream) || !rt_sync(ctx->stream))) return AMH_ERR_DEVICE;
  bool any = false;
  for (u32 d = 0; d < b->B.n_docs && !any; d++)
    any = !b->status[d] && first_len[d] != (u32)(st->h_doc_off[2 * (size_t)d + 1] - st->h_doc_off[2 * (size_t)d]);
  if (!any) return AMH_SUCCESS;
  if (!ensure_host_arena(st)) return AMH_ERR_DEVICE;
  for (u32 d = 0; d < b->B.n_docs; d++) {
    if (b->status[d]) continue;
    const u8* ptr = st->h_arena.data() + st->h_doc_off[2 * (size_t)d];
    size_t avail = (size_t)(st->h_doc_off[2 * (size_t)d + 1] - st->h_doc_off[2 * (size_t)d]);
    size_t fl = first_len[d];
    if (fl == avail) continue;
    ApplyItem it;
    it.doc = d;
    it.mode = (opts && opts->on_partial_load == 1) ? APM_LOAD_IGNORE : APM_LOAD_ERROR;
    it.first_was_doc = fl != 0;
    it.bufs.push_back({ptr + fl, avail - fl});
    items.push_back(std::move(it));
  }
  if (items.empty()) return AMH_SUCCESS;
  // the inputs are read from b's staging arena, which apply_core replaces: keep it alive for the call
  amh_staged* keep = b->staged;
  keep->refs++;
  std::vector<i32> results;
  int rc = apply_core(ctx, b, items, results, "amh_load_batch");
  staged_unref(keep);
  if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) return rc;
  bool dirty = false;
  for (size_t i = 0; i < items.size(); i++)
    if (results[i]) { b->status[items[i].doc] = results[i]; b->fail_line[items[i].doc] = 0; dirty = true; }
  if (dirty && (!rt_h2d(b->B.status, b->status.data(), b->status.size() * sizeof(i32), ctx->stream) || !rt_sync(ctx->stream))) return AMH_ERR_DEVICE;
  return AMH_SUCCESS;
}

extern "C" int amh_apply_batch(amh_ctx* ctx, amh_batch* b, const uint32_t* doc_idx, const uint8_t* const* bufs, const size_t* lens, size_t n, int incremental, int* results) {
  if (!ctx || !b || (n && (!doc_idx || !bufs || !lens)) || n >= 0x7FFFFFFFull) { set_error(ctx, "amh_apply_batch: invalid argument"); return AMH_ERR_INVALID_ARG; }
  std::vector<ApplyItem> items;
  std::unordered_map<u32, u32> item_of;
  std::vector<u32> buf_item(n);
  for (size_t i = 0; i < n; i++) {
    if (doc_idx[i] >= b->B.n_docs) { set_error(ctx, "amh_apply_batch: document index out of range"); return AMH_ERR_INVALID_ARG; }
    auto it = item_of.find(doc_idx[i]);
    if (it == item_of.end()) {
      it = item_of.emplace(doc_idx[i], (u32)items.size()).first;
      ApplyItem ai;
      ai.doc = doc_idx[i];
      ai.mode = incremental ? APM_PREFIX : APM_STRICT;
      items.push_back(std::move(ai));
    }
    buf_item[i] = it->second;
    if (lens[i]) items[it->second].bufs.push_back({bufs[i], lens[i]});
  }
  std::vector<i32> res;
  int rc = apply_core(ctx, b, items, res, "amh_apply_batch");
  if (results && res.size() == items.size()) for (size_t i = 0; i < n; i++) results[i] = res[buf_item[i]];
  return rc;
}
extern "C" int amh_apply_changes(amh_ctx* ctx, amh_batch* b, uint32_t doc, const uint8_t* const* changes, const size_t* lens, size_t n) {
  if (!ctx || !b || doc >= b->B.n_docs || (n && (!changes || !lens))) { set_error(ctx, "amh_apply_changes: invalid argument"); return AMH_ERR_INVALID_ARG; }
  std::vector<ApplyItem> items(1);
  items[0].doc = doc;
  items[0].mode = APM_STRICT;
  items[0].one_call = true;
  for (size_t i = 0; i < n; i++) if (lens[i]) items[0].bufs.push_back({changes[i], lens[i]});
  std::vector<i32> res;
  return apply_core(ctx, b, items, res, "amh_apply_changes");
}
// dst[dst_idx[p]].merge(srcs[0][..]); .merge(srcs[1][..]); ... in one pass: the changes each source adds
// (get_changes_added, automerge.rs:1311-1334, against the history as it stands after the sources before it) are
// handed to the apply path as one apply_changes call per source, so the result — history order included — is that
// of the sequential merges, while rows are placed, saved and re-verified once instead of once per source.
extern "C" int amh_merge_many(amh_ctx* ctx, amh_batch* dst, const amh_batch* const* srcs, size_t n_srcs, const uint32_t* dst_idx, const uint32_t* src_idx, size_t n_pairs) {
  if (!ctx || !dst || (n_srcs && !srcs) || (n_pairs && (!dst_idx || (n_srcs && !src_idx))) || n_pairs >= 0x7FFFFFFFull || n_srcs > 4096) { set_error(ctx, "amh_merge_many: invalid argument"); return AMH_ERR_INVALID_ARG; }
  if (n_pairs == 0 || n_srcs == 0) return AMH_SUCCESS;
  if (!ensure_chg(dst)) { set_error(ctx, "amh_merge_many: D2H copy failed"); return AMH_ERR_DEVICE; }
  for (size_t s = 0; s < n_srcs; s++) {
    const amh_batch* sb = srcs[s];
    if (!sb || sb == dst) { set_error(ctx, "amh_merge_many: invalid source batch"); return AMH_ERR_INVALID_ARG; }
    if (!ensure_chg(sb)) { set_error(ctx, "amh_merge_many: D2H copy failed"); return AMH_ERR_DEVICE; }
  }
  if (!rt_sync(ctx->stream)) return AMH_ERR_DEVICE;
  const HostView& ha = dst->hv;
  u32 P = (u32)n_pairs;
  PhaseTimer pt("amh_merge_many");
  std::vector<ApplyItem> items(P);
  std::vector<i32> pre_fail(P, 0);
  std::vector<u8> seen_doc(dst->B.n_docs, 0);
  for (u32 p = 0; p < P; p++) {
    u32 a = dst_idx[p];
    if (a >= dst->B.n_docs || seen_doc[a]) { set_error(ctx, "amh_merge_many: document index out of range or used twice"); return AMH_ERR_INVALID_ARG; }
    seen_doc[a] = 1;
    for (size_t s = 0; s < n_srcs; s++) {
      u32 b = src_idx[s * n_pairs + p];
      if (b != AMH_NULL32 && b >= srcs[s]->B.n_docs) { set_error(ctx, "amh_merge_many: source document index out of range"); return AMH_ERR_INVALID_ARG; }
    }
  }
  host_parallel_for(P, [&](u32 p) {
    u32 a = dst_idx[p];
    items[p].doc = a;
    items[p].mode = APM_STRICT;
    if (dst->status[a]) return;   // apply_core reports it
    u32 nca = ha.cnt[a].n_chg, cba = ha.chg_base[a];
    // the hashes the document holds so far: an open-addressing set of pointers into the host views, reused by the thread
    // from pair to pair (a node-based map cost one allocation per hash: 800,000 per call of the headline step)
    static thread_local std::vector<const u8*> known;
    size_t kcap = 16, kneed = nca;
    for (size_t s = 0; s < n_srcs; s++) { u32 b = src_idx[s * n_pairs + p]; if (b != AMH_NULL32 && !srcs[s]->status[b]) kneed += srcs[s]->hv.cnt[b].n_chg; }
    while (kcap < 2 * kneed) kcap <<= 1;
    known.assign(kcap, nullptr);
    const size_t kmask = kcap - 1;
    auto kslot = [&](const u8* h) { u64 k; memcpy(&k, h, 8); size_t i = (size_t)(k ^ (k >> 29)) & kmask; while (known[i] && memcmp(known[i], h, 32) != 0) i = (i + 1) & kmask; return i; };
    auto known_has = [&](const u8* h) { return known[kslot(h)] != nullptr; };
    auto known_add = [&](const u8* h) { known[kslot(h)] = h; };
    for (u32 i = 0; i < nca; i++) known_add(ha.c_hash.data() + (size_t)(cba + i) * 32);
    u32 n_calls = 0;
    for (size_t s = 0; s < n_srcs; s++) {
      u32 b = src_idx[s * n_pairs + p];
      if (b == AMH_NULL32) continue;
      const amh_batch* sb = srcs[s];
      if (sb->status[b]) { pre_fail[p] = sb->status[b]; break; }
      const HostView& hb = sb->hv;
      u32 ncb = hb.cnt[b].n_chg, cbb = hb.chg_base[b];
      auto bhash = [&](u32 j) { return hb.c_hash.data() + (size_t)(cbb + j) * 32; };
      static thread_local std::vector<u8> is_dep, seen;   // reused across the pairs a host thread serves: no allocation per (pair, source)
      static thread_local std::vector<u32> stack, added, deps;
      is_dep.assign(ncb, 0); seen.assign(ncb, 0);
      stack.clear(); added.clear(); deps.clear();
      for (u32 j = 0; j < ncb; j++) for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) is_dep[hb.dep_idx[hb.c_dep_base[cbb + j] + k]] = 1;
      for (u32 j = 0; j < ncb; j++) if (!is_dep[j]) stack.push_back(j);
      std::sort(stack.begin(), stack.end(), [&](u32 x, u32 y) { return memcmp(bhash(x), bhash(y), 32) < 0; });
      while (!stack.empty()) {
        u32 j = stack.back(); stack.pop_back();
        if (seen[j]) continue;
        if (known_has(bhash(j))) continue;
        seen[j] = 1; added.push_back(j);
        deps.clear();
        for (u32 k = 0; k < hb.c_ndeps[cbb + j]; k++) deps.push_back(hb.dep_idx[hb.c_dep_base[cbb + j] + k]);
        std::sort(deps.begin(), deps.end(), [&](u32 x, u32 y) { return memcmp(bhash(x), bhash(y), 32) < 0; });
        for (u32 dj : deps) stack.push_back(dj);
      }
      std::reverse(added.begin(), added.end());
      if (added.empty()) continue;
      // the added changes of one source = one apply_changes call; their re-derived chunks stay where they are, in the
      // source batch's device memory
      for (u32 j : added) {
        items[p].dev.push_back({sb->B.out_arena + hb.c_out_off[cbb + j], hb.c_out_len[cbb + j], n_calls, &sb->B, cbb + j});
        known_add(bhash(j));   // every added change applies (the source's history is causally closed)
      }
      n_calls++;
    }
    if (pre_fail[p]) items[p].dev.clear();
    });
  pt.mark("added changes (host)");
  std::vector<i32> results;
  int rc = apply_core(ctx, dst, items, results, "amh_merge_many");
  if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) return rc;
  for (u32 p = 0; p < P; p++) if (pre_fail[p]) rc = AMH_ERR_DOC_FAILED;
  if (rc == AMH_ERR_DOC_FAILED) set_error(ctx, "amh_merge_many: some documents were left unmerged (failed documents, duplicate seq numbers or unsupported shapes)");
  return rc;
}
// == Automerge::save_after (automerge.rs:993-1000) = get_changes(heads) (:1306, get_changes_clock :1022-1047) as raw
// change chunks back to back.  O(changes) graph walk over the batch's host mirrors; the chunks are the re-derived ones.
extern "C" int amh_save_after(const amh_batch* b, uint32_t doc, const uint8_t* heads32, size_t n_heads, amh_bytes_out* out) {
  if (!b || !out || doc >= b->B.n_docs || (n_heads && !heads32)) return AMH_ERR_INVALID_ARG;
  out->data = nullptr; out->len = 0;
  if (b->status[doc]) return AMH_ERR_DOC_FAILED;
  if (!ensure_chg(b)) return AMH_ERR_DEVICE;
  HostView& hv = b->hv;
  if (!hv.out_ok) {
    if (!fetch(b, hv.out_arena, b->B.out_arena, (size_t)b->tot_out) || !rt_sync(b->ctx->stream)) return AMH_ERR_DEVICE;
    hv.out_ok = true;
  }
  u32 cb = hv.chg_base[doc], nc = hv.cnt[doc].n_chg, na = hv.desc[doc].n_actors;
  std::vector<u64> clock(na, 0);
  std::vector<u8> visited(nc, 0);
  std::vector<u32> to_visit;
  for (size_t h = 0; h < n_heads; h++)
    for (u32 i = 0; i < nc; i++)
      if (memcmp(hv.c_hash.data() + (size_t)(cb + i) * 32, heads32 + 32 * h, 32) == 0) { to_visit.push_back(i); break; }
  while (!to_visit.empty()) {
    u32 i = to_visit.back();
    to_visit.pop_back();
    if (visited[i]) continue;
    visited[i] = 1;
    u32 a = hv.c_actor[cb + i];
    if (hv.c_seq[cb + i] > clock[a]) clock[a] = hv.c_seq[cb + i];
    for (u32 k = 0; k < hv.c_ndeps[cb + i]; k++) to_visit.push_back(hv.dep_idx[hv.c_dep_base[cb + i] + k]);
  }
  size_t total = 0;
  for (u32 i = 0; i < nc; i++) if (hv.c_seq[cb + i] > clock[hv.c_actor[cb + i]]) total += hv.c_out_len[cb + i];
  out->data = (uint8_t*)malloc(total ? total : 1);
  if (!out->data) return AMH_ERR_OOM;
  size_t o = 0;
  for (u32 i = 0; i < nc; i++)
    if (hv.c_seq[cb + i] > clock[hv.c_actor[cb + i]]) { memcpy(out->data + o, hv.out_arena.data() + hv.c_out_off[cb + i], hv.c_out_len[cb + i]); o += hv.c_out_len[cb + i]; }
  out->len = total;
  return AMH_SUCCESS;
}
extern "C" int amh_doc_queue_len(const amh_batch* b, uint32_t doc, size_t* n) {
  if (!b || !n || doc >= b->B.n_docs) return AMH_ERR_INVALID_ARG;
  auto it = b->queue.find(doc);
  *n = it == b->queue.end() ? 0 : it->second.size();
  return AMH_SUCCESS;
}
extern "C" int amh_save_batch(amh_ctx* ctx, const amh_batch* bc, const amh_save_opts* opts, amh_bytes_out* out) {
  if (!ctx || !bc || !out) { set_error(ctx, "invalid argument"); return AMH_ERR_INVALID_ARG; }
  AMH_NEEDS_OPS(bc, "amh_save_batch")
  amh_batch* b = const_cast<amh_batch*>(bc);
  BatchDev B = b->B;
  u32 n = B.n_docs;
  for (u32 d = 0; d < n; d++) { out[d].data = nullptr; out[d].len = 0; }
  if (n == 0) return AMH_SUCCESS;
  bool ok = true;
  CallScratch scratch(b);   // everything this call allocates goes back to the pool when it returns
  B.sv_status = balloc<i32>(b, n + 1, &ok);
  B.sv_col_len = balloc<u32>(b, (size_t)n * 25 + 1, &ok);
  B.sv_wide = balloc<u32>(b, n + 1, &ok);
  B.sv_body_len = balloc<u32>(b, n + 1, &ok);
  B.sv_off = balloc<u32>(b, n + 2, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
#ifndef AMH_EMU
  B.sha_host_min = AMH_HOST_SHA_MIN;   // chunks of that size and more: checksum by the host below, which receives the bytes anyway
#endif
  SaveBigHost sbh;
  if (!save_big_prepare(ctx, b, B, b->status, sbh)) { set_error(ctx, "save: job tables failed (device out of memory?)"); return AMH_ERR_DEVICE; }
  launch(ctx, AK_SAVE_CHECK, B, n);
  launch(ctx, AK_SAVE_SIZE, B, n * 25);
  save_big_launch(ctx, B, 0);
  launch(ctx, AK_SAVE_LAYOUT, B, n);
  u32 total = 0;
  if (!scan_inplace(b, B.sv_off, n, &total)) { set_error(ctx, "scan failed"); return AMH_ERR_DEVICE; }
  if (total == 0xFFFFFFFFu) { set_error(ctx, "saved documents exceed 4 GiB: save the batch in parts"); return AMH_ERR_BATCH_TOO_LARGE; }
  B.sv_out = balloc<u8>(b, (size_t)total + 256, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  launch(ctx, AK_SAVE_STORE, B, n * 25);
  save_big_launch(ctx, B, 1);
  launch(ctx, AK_SAVE_FINISH, B, n);
  // the saved bytes come back through the context's pinned area when they fit a modest one (pageable copies run at a
  // fraction of the link's rate), else into pageable memory
  std::vector<u8> host_pageable;
  struct { u8* p; u8* data() const { return p; } } host{nullptr};
  if ((size_t)total + 1 <= ((size_t)256 << 20)) {
    if ((size_t)total + 1 > ctx->bigc_cap) {
      if (ctx->bigc) rt_host_free(ctx->bigc);
      ctx->bigc_cap = ((size_t)total + 1) + ((size_t)total + 1) / 4;
      ctx->bigc = (u8*)rt_host_alloc(ctx->bigc_cap);
      if (!ctx->bigc) ctx->bigc_cap = 0;
    }
    host.p = ctx->bigc;
  }
  if (!host.p) { host_pageable.resize((size_t)total + 1); host.p = host_pageable.data(); }
  std::vector<u32> off(n + 1), body(n);
  std::vector<i32> st(n);
  if (!rt_d2h(host.data(), B.sv_out, total, ctx->stream) || !rt_d2h(off.data(), B.sv_off, (n + 1) * sizeof(u32), ctx->stream) ||
      !rt_d2h(body.data(), B.sv_body_len, n * sizeof(u32), ctx->stream) ||
      !rt_d2h(st.data(), B.sv_status, n * sizeof(i32), ctx->stream)) {
    set_error(ctx, "D2H copy failed");
    return AMH_ERR_DEVICE;
  }
  collect_profile(ctx);
  bool deflate = opts ? opts->deflate != 0 : true;
  for (u32 d = 0; d < n; d++) {
    if (st[d] != 0) continue;   // failed to load, or a shape save does not handle: no bytes (amh_doc_status / amh_save_status)
    size_t len = 9 + body[d] + (body[d] < 0x80 ? 1 : body[d] < 0x4000 ? 2 : body[d] < 0x200000 ? 3 : body[d] < 0x10000000 ? 4 : 5);
    if (B.sha_host_min && len >= B.sha_host_min) {   // k_save_finish left the checksum of a large chunk to the host
      u8 h[32];
      amh_host_sha256(host.data() + off[d] + 8, len - 8, nullptr, 0, h);
      memcpy(host.data() + off[d] + 4, h, 4);
    }
    const u8* src = host.data() + off[d];
    std::vector<u8> z;
    if (deflate && amh_host_deflate_document(src, len, z)) { src = z.data(); len = z.size(); }
    // retain_orphans: the queued changes follow the document chunk as they are (automerge.rs:959-963)
    size_t orphans = 0;
    auto qit = b->queue.find(d);
    bool keep = (!opts || opts->retain_orphans) && qit != b->queue.end();
    if (keep) for (auto& q : qit->second) orphans += q.size();
    out[d].data = (uint8_t*)malloc(len + orphans ? len + orphans : 1);
    if (!out[d].data) return AMH_ERR_OOM;
    memcpy(out[d].data, src, len);
    if (keep) { size_t o = len; for (auto& q : qit->second) { memcpy(out[d].data + o, q.data(), q.size()); o += q.size(); } }
    out[d].len = len + orphans;
  }
  return AMH_SUCCESS;
}
extern "C" int amh_sync_summaries(amh_ctx* ctx, const amh_batch* bc, amh_bytes_out* out) {
  if (!ctx || !bc || !out) { set_error(ctx, "invalid argument"); return AMH_ERR_INVALID_ARG; }
  amh_batch* b = const_cast<amh_batch*>(bc);
  BatchDev B = b->B;
  u32 n = B.n_docs;
  out->data = nullptr;
  out->len = 0;
  if (n == 0) return AMH_SUCCESS;
  bool ok = true;
  CallScratch scratch(b);
  B.sync_off = balloc<u32>(b, n + 2, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  launch(ctx, AK_SYNC_SIZE, B, n);
  u32 total = 0;
  if (!scan_inplace(b, B.sync_off, n, &total)) { set_error(ctx, "scan failed"); return AMH_ERR_DEVICE; }
  if (total == 0xFFFFFFFFu) { set_error(ctx, "sync summaries exceed 4 GiB"); return AMH_ERR_BATCH_TOO_LARGE; }
  B.sync_out = balloc<u8>(b, (size_t)total + 64, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  launch(ctx, AK_SYNC_FILL, B, n);
  out->data = (uint8_t*)malloc(total ? total : 1);
  if (!out->data) return AMH_ERR_OOM;
  if (!rt_d2h(out->data, B.sync_out, total, ctx->stream)) { free(out->data); out->data = nullptr; set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE; }
  out->len = total;
  collect_profile(ctx);
  return AMH_SUCCESS;
}

// == the exchange step of the sync path across GPUs: every rank's summaries, gathered on every rank.
// The collective runs on device buffers (RCCL over xGMI): sizes by one all-gather of a u64 per rank, the records by an
// all-gather(v) into one device buffer, which then comes to the host once.  rank_off[world + 1] = where each rank's
// records start in out.  Without a communicator (single GPU) this is amh_sync_summaries with rank_off = {0, len}.
extern "C" int amh_sync_summaries_all(amh_ctx* ctx, const amh_batch* bc, amh_bytes_out* out, size_t* rank_off) {
  if (!ctx || !bc || !out || !rank_off) { set_error(ctx, "invalid argument"); return AMH_ERR_INVALID_ARG; }
#ifdef AMH_EMU
  int rc0 = amh_sync_summaries(ctx, bc, out);
  rank_off[0] = 0; rank_off[1] = out->len;
  return rc0;
#else
  if (!ctx->comm) {
    int rc0 = amh_sync_summaries(ctx, bc, out);
    rank_off[0] = 0; rank_off[1] = out->len;
    return rc0;
  }
  amh_batch* b = const_cast<amh_batch*>(bc);
  BatchDev B = b->B;
  const u32 n = B.n_docs;
  const int world = amh_comm_world(ctx->comm), rank = amh_comm_rank(ctx->comm);
  out->data = nullptr; out->len = 0;
  bool ok = true;
  CallScratch scratch(b);
  u32 total = 0;
  B.sync_off = balloc<u32>(b, (size_t)n + 2, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  if (n) {
    launch(ctx, AK_SYNC_SIZE, B, n);
    if (!scan_inplace(b, B.sync_off, n, &total)) { set_error(ctx, "scan failed"); return AMH_ERR_DEVICE; }
    if (total == 0xFFFFFFFFu) { set_error(ctx, "sync summaries exceed 4 GiB"); return AMH_ERR_BATCH_TOO_LARGE; }
  }
  B.sync_out = balloc<u8>(b, (size_t)total + 64, &ok);
  u64* d_sizes = balloc<u64>(b, (size_t)world + 1, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  if (n) launch(ctx, AK_SYNC_FILL, B, n);
  std::vector<u64> sizes(world, 0), offs(world + 1, 0);
  u64 mine = total;
  if (!rt_h2d(d_sizes + world, &mine, sizeof(u64), ctx->stream)) { set_error(ctx, "H2D copy failed"); return AMH_ERR_DEVICE; }
  const char* e = amh_comm_allgather_u64(ctx->comm, d_sizes + world, d_sizes, (void*)ctx->stream);
  if (e) { set_error(ctx, std::string("amh_sync_summaries_all: all-gather of the sizes failed: ") + e); return AMH_ERR_DEVICE; }
  if (!rt_d2h(sizes.data(), d_sizes, (size_t)world * sizeof(u64), ctx->stream) || !rt_sync(ctx->stream)) { set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE; }
  if (sizes[rank] != total) { set_error(ctx, "amh_sync_summaries_all: the communicator's rank does not match this context"); return AMH_ERR_DEVICE; }
  for (int r = 0; r < world; r++) offs[r + 1] = offs[r] + ((sizes[r] + 15) & ~(u64)15);   // 16-byte aligned slots in the device buffer
  u8* d_all = balloc<u8>(b, (size_t)offs[world] + 64, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  e = amh_comm_allgatherv(ctx->comm, B.sync_out, d_all, sizes.data(), offs.data(), (void*)ctx->stream);
  if (e) { set_error(ctx, std::string("amh_sync_summaries_all: all-gather failed: ") + e); return AMH_ERR_DEVICE; }
  u64 len = 0;
  for (int r = 0; r < world; r++) { rank_off[r] = (size_t)len; len += sizes[r]; }
  rank_off[world] = (size_t)len;
  out->data = (uint8_t*)malloc(len ? (size_t)len : 1);
  if (!out->data) return AMH_ERR_OOM;
  for (int r = 0; r < world; r++)
    if (sizes[r] && !rt_d2h(out->data + rank_off[r], d_all + offs[r], (size_t)sizes[r], ctx->stream)) { free(out->data); out->data = nullptr; set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE; }
  if (!rt_sync(ctx->stream)) { free(out->data); out->data = nullptr; set_error(ctx, "device pipeline failed"); return AMH_ERR_DEVICE; }
  out->len = (size_t)len;
  collect_profile(ctx);
  return AMH_SUCCESS;
#endif
}
// a sync message kernel, timed like the others when the context is profiling
static const char* const SK_KERNEL_NAMES[SK_NUM] = {"sm_lookup", "sm_plan", "sm_fill", "sm_copy", "sm_bloom", "sm_jobs"};
static void launch_sm_timed(amh_ctx* ctx, int k, const SyncDev* G, u32 n) {
  if (n == 0) return;
  if (ctx->profiling) {
    amh_ctx::Pending p;
    p.k = AK_NUM + MK_NUM + k;
    rt_event_create(&p.a);
    rt_event_create(&p.b);
    rt_event_record(&p.a, ctx->stream);
    amh_launch_sm(k, G, n, (void*)ctx->stream);
    rt_event_record(&p.b, ctx->stream);
    ctx->pending.push_back(p);
  } else {
    amh_launch_sm(k, G, n, (void*)ctx->stream);
  }
}
// BloomFilter::parse (sync/bloom.rs:49-67) of a Have's wire bytes: parameters and where the bit array starts
static bool sm_parse_bloom(const u8* p, size_t n, u32& entries, u32& probes, size_t& bits_at, u32& nbytes) {
  entries = 0; probes = 7; bits_at = 0; nbytes = 0;
  if (n == 0) return true;
  size_t pos = 0;
  u64 e = 0, bpe = 0, pr = 0;
  if (!host_uleb(p, pos, n, e) || !host_uleb(p, pos, n, bpe) || !host_uleb(p, pos, n, pr)) return false;
  if (e > 0xFFFFFFFFull || bpe > 0xFFFFFFFFull || pr > 0xFFFFFFFFull) return false;
  const double cap = std::ceil(((double)e * (double)bpe) / 8.0);
  if (cap > (double)(n - pos) || cap >= (double)(1u << 28)) return false;
  entries = (u32)e; probes = (u32)pr; bits_at = pos; nbytes = (u32)cap;
  return !(entries != 0 && nbytes == 0);
}
// the messages of one call into device memory owned by the batch (sm_msg: message r at sm_off[r] .. sm_off[r + 1];
// sm_sent: 32-byte hashes of the changes in message r at 32 * sm_job[r] ..), kinds to the host
static void sm_release(amh_ctx* ctx, amh_batch* b) {
  for (DevAlloc* a : {&b->sm_msg, &b->sm_sent, &b->sm_off, &b->sm_job}) if (a->p) { ctx->pool.release(a->p, a->cap); *a = DevAlloc(); }
  b->sm_total = b->sm_jobs = 0; b->sm_n = 0;
}
static int sm_to_device(amh_ctx* ctx, amh_batch* b, const amh_sync_state* states, size_t n, uint32_t* kinds, bool raw = false) {
  PhaseTimer pt(raw ? "amh_save_after_batch" : "amh_generate_sync_messages");
  sm_release(ctx, b);
  for (size_t i = 0; i < n; i++) kinds[i] = AMH_SYNC_NONE;
  if (n == 0) return AMH_SUCCESS;
  const u32 R = (u32)n;
  bool ok = true;
  CallScratch scratch(b);
  // changes and actors of every record's document: all the host needs of the batch to lay out the per-record scratch
  std::vector<u32> doc_sizes((size_t)R * 2);
  {
    std::vector<u32> docs(R);
    for (u32 r = 0; r < R; r++) {
      if (states[r].doc >= b->B.n_docs) { set_error(ctx, "amh_generate_sync_messages: invalid sync state"); return AMH_ERR_INVALID_ARG; }
      docs[r] = states[r].doc;
    }
    u32* d_docs = balloc<u32>(b, (size_t)R * 3 + 1, &ok);
    if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
    if (!rt_h2d(d_docs, docs.data(), (size_t)R * 4, ctx->stream)) { set_error(ctx, "H2D copy failed"); return AMH_ERR_DEVICE; }
    amh_sm_doc_sizes(b->B, d_docs, d_docs + R, R, (void*)ctx->stream);
    if (!rt_d2h(doc_sizes.data(), d_docs + R, (size_t)R * 8, ctx->stream)) { set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE; }
  }
  // ---- the States, flattened: one hash pool, per-record list bounds, the Haves and their filters ----
  // Three passes, the first and the last over all records in parallel: what every record contributes (hashes, Haves,
  // filter bytes; everything a record can be refused for), places by prefix sums, the copies.
  HVec<u32> r_doc(R), r_flags(R), r_list((size_t)R * SM_LISTS + 1), r_have(R + 1), pool_rec, h_ls, h_entries, h_probes, h_nbytes;
  HVec<u64> h_bits, r_flag_base(R), r_clock_base(R), r_word_base(R);
  HVec<u8> pool, blooms;
  u64 n_flag = 0, n_clock = 0, n_word = 0;
  // deps of queued changes that are not queued themselves, and the hashes of the queued changes (get_missing_deps)
  struct QInfo { std::vector<u8> deps, hashes; };
  std::unordered_map<u32, QInfo> qinfo;
  auto queue_of = [&](u32 d) -> const QInfo* {
    auto qit = b->queue.find(d);
    if (qit == b->queue.end() || qit->second.empty()) return nullptr;
    auto it = qinfo.find(d);
    if (it != qinfo.end()) return &it->second;
    QInfo qi;
    std::set<std::array<u8, 32>> in_queue, deps;
    for (auto& q : qit->second) {
      std::array<u8, 32> h;
      if (q.size() < 10) continue;
      amh_host_sha256(q.data() + 8, q.size() - 8, nullptr, 0, h.data());
      in_queue.insert(h);
    }
    for (auto& q : qit->second) {
      if (q.size() < 10) continue;
      size_t pos = 9;
      u64 blen = 0, nd = 0;
      if (!host_uleb(q.data(), pos, q.size(), blen) || !host_uleb(q.data(), pos, q.size(), nd) || nd > (q.size() - pos) / 32) continue;
      for (u64 k = 0; k < nd; k++) { std::array<u8, 32> h; memcpy(h.data(), q.data() + pos + 32 * k, 32); if (!in_queue.count(h)) deps.insert(h); }
    }
    for (auto& h : deps) qi.deps.insert(qi.deps.end(), h.begin(), h.end());
    for (auto& h : in_queue) qi.hashes.insert(qi.hashes.end(), h.begin(), h.end());
    return &qinfo.emplace(d, std::move(qi)).first->second;
  };
  // the queues of the documents named (rare; the map is only read from here on)
  if (!b->queue.empty()) for (u32 r = 0; r < R; r++) if (!b->status[states[r].doc]) (void)queue_of(states[r].doc);
  auto queue_info = [&](u32 d) -> const QInfo* { if (qinfo.empty()) return nullptr; auto it = qinfo.find(d); return it == qinfo.end() ? nullptr : &it->second; };
  struct RecSize { u32 hashes, haves; u64 bloom_bytes; };
  std::vector<RecSize> rs(R);
  std::atomic<int> flat_err{0};   // 1 = invalid state, 2 = malformed filter
  host_parallel_for(R, [&](u32 r) {
    const amh_sync_state& s = states[r];
    if ((s.n_shared_heads && !s.shared_heads32) || (s.n_last_sent_heads && !s.last_sent_heads32) || (s.n_their_heads && !s.their_heads32) ||
        (s.n_their_need && !s.their_need32) || (s.n_their_have && !s.their_have) || (s.n_sent_hashes && !s.sent_hashes32)) { flat_err = 1; rs[r] = {0, 0, 0}; return; }
    const bool th = s.flags & AMH_SYNC_THEIR_HEADS, tn = s.flags & AMH_SYNC_THEIR_NEED, tv = s.flags & AMH_SYNC_THEIR_HAVE;
    const QInfo* qi = b->status[s.doc] ? nullptr : queue_info(s.doc);
    u64 nh = (u64)s.n_shared_heads + s.n_last_sent_heads + (th ? s.n_their_heads : 0) + (tn ? s.n_their_need : 0) + s.n_sent_hashes +
             (qi ? qi->deps.size() / 32 + qi->hashes.size() / 32 : 0);
    u64 bl = 0;
    const u32 nv = tv ? (u32)std::min<size_t>(s.n_their_have, 0x7FFFFFFFu) : 0;
    for (u32 k = 0; k < nv; k++) {
      const amh_sync_have& h = s.their_have[k];
      if ((h.n_last_sync && !h.last_sync32) || (h.bloom_len && !h.bloom)) { flat_err = 1; rs[r] = {0, 0, 0}; return; }
      u32 e, pr, nb; size_t at;
      if (!sm_parse_bloom(h.bloom, h.bloom_len, e, pr, at, nb)) { flat_err = 2; rs[r] = {0, 0, 0}; return; }
      nh += h.n_last_sync; bl += nb;
    }
    rs[r] = {(u32)std::min<u64>(nh, 0xFFFFFFFFull), nv, bl};
  }, 4096, 1024);
  if (flat_err == 1) { set_error(ctx, "amh_generate_sync_messages: invalid sync state"); return AMH_ERR_INVALID_ARG; }
  if (flat_err == 2) { set_error(ctx, "amh_generate_sync_messages: malformed Bloom filter in a Have"); return AMH_ERR_INVALID_ARG; }
  std::vector<u64> pool_at(R + 1), bloom_at(R + 1);
  {
    u64 np = 0, nv = 0, nbl = 0;
    for (u32 r = 0; r < R; r++) {
      pool_at[r] = np; r_have[r] = (u32)nv; bloom_at[r] = nbl;
      np += rs[r].hashes; nv += rs[r].haves; nbl += rs[r].bloom_bytes;
      if (np >= 0x7FFFFFFFull || nv >= 0x7FFFFFFFull) { set_error(ctx, "amh_generate_sync_messages: too many hashes in one call"); return AMH_ERR_BATCH_TOO_LARGE; }
      const amh_sync_state& s = states[r];
      const bool th = s.flags & AMH_SYNC_THEIR_HEADS, tn = s.flags & AMH_SYNC_THEIR_NEED;
      const QInfo* qi = b->status[s.doc] ? nullptr : queue_info(s.doc);
      const u32 nc = doc_sizes[2 * (size_t)r], na = doc_sizes[2 * (size_t)r + 1];
      r_flag_base[r] = n_flag; n_flag += nc;
      r_clock_base[r] = n_clock; n_clock += na;
      r_word_base[r] = n_word;
      n_word += (u64)nc * 3 + (qi ? qi->deps.size() / 32 : 0) + (th ? s.n_their_heads : 0) + 2 * (u64)(tn ? s.n_their_need : 0);
    }
    pool_at[R] = np; r_have[R] = (u32)nv; bloom_at[R] = nbl;
    pool.resize((size_t)np * 32); pool_rec.resize((size_t)np);
    h_ls.resize(2 * (size_t)nv); h_entries.resize((size_t)nv); h_probes.resize((size_t)nv); h_nbytes.resize((size_t)nv); h_bits.resize((size_t)nv);
    blooms.resize((size_t)nbl);
  }
  host_parallel_for(R, [&](u32 r) {
    const amh_sync_state& s = states[r];
    r_doc[r] = s.doc;
    r_flags[r] = s.flags & 15u;
    const bool th = s.flags & AMH_SYNC_THEIR_HEADS, tn = s.flags & AMH_SYNC_THEIR_NEED;
    const QInfo* qi = b->status[s.doc] ? nullptr : queue_info(s.doc);
    u32 at = (u32)pool_at[r];
    auto add_list = [&](const u8* h, size_t k) {
      if (!k) return;
      memcpy(pool.data() + (size_t)at * 32, h, 32 * k);
      for (size_t i = 0; i < k; i++) pool_rec[at + i] = r;
      at += (u32)k;
    };
    u32* L = r_list.data() + (size_t)r * SM_LISTS;
    L[SM_SHARED] = at; add_list(s.shared_heads32, s.n_shared_heads);
    L[SM_LAST_SENT] = at; add_list(s.last_sent_heads32, s.n_last_sent_heads);
    L[SM_THEIR_HEADS] = at; if (th) add_list(s.their_heads32, s.n_their_heads);
    L[SM_THEIR_NEED] = at; if (tn) add_list(s.their_need32, s.n_their_need);
    L[SM_SENT] = at; add_list(s.sent_hashes32, s.n_sent_hashes);
    L[SM_QDEPS] = at; if (qi) add_list(qi->deps.data(), qi->deps.size() / 32);
    L[SM_QHASH] = at; if (qi) add_list(qi->hashes.data(), qi->hashes.size() / 32);
    L[SM_LAST_SYNC] = at;
    u64 bits_at = bloom_at[r];
    for (u32 k = 0, hv = r_have[r]; k < rs[r].haves; k++, hv++) {
      const amh_sync_have& h = s.their_have[k];
      h_ls[2 * (size_t)hv] = at;
      add_list(h.last_sync32, h.n_last_sync);
      h_ls[2 * (size_t)hv + 1] = at;
      u32 e, pr, nb; size_t off;
      sm_parse_bloom(h.bloom, h.bloom_len, e, pr, off, nb);   // (parsed in the first pass already: cannot fail here)
      h_entries[hv] = e; h_probes[hv] = pr; h_nbytes[hv] = nb; h_bits[hv] = bits_at;
      if (nb) memcpy(blooms.data() + bits_at, h.bloom + off, nb);
      bits_at += nb;
    }
  }, 4096, 1024);
  r_list[(size_t)R * SM_LISTS] = (u32)pool_at[R];
  const u32 n_pool = (u32)pool_at[R];
  pt.mark("states flattened (host)");
  // ---- one upload: the tables side by side in one device block, put together in pinned memory when they fit it ----
  struct Part { const void* p; size_t bytes; size_t at; };
  Part parts[] = {{r_doc.data(), (size_t)R * 4, 0}, {r_flags.data(), (size_t)R * 4, 0}, {pool.data(), pool.size(), 0}, {pool_rec.data(), (size_t)n_pool * 4, 0},
                  {r_list.data(), r_list.size() * 4, 0}, {r_have.data(), r_have.size() * 4, 0}, {h_ls.data(), h_ls.size() * 4, 0}, {h_entries.data(), h_entries.size() * 4, 0},
                  {h_probes.data(), h_probes.size() * 4, 0}, {h_bits.data(), h_bits.size() * 8, 0}, {h_nbytes.data(), h_nbytes.size() * 4, 0}, {blooms.data(), blooms.size(), 0},
                  {r_flag_base.data(), (size_t)R * 8, 0}, {r_clock_base.data(), (size_t)R * 8, 0}, {r_word_base.data(), (size_t)R * 8, 0}};
  size_t blob_len = 0;
  for (Part& q : parts) { q.at = (blob_len + 255) & ~(size_t)255; blob_len = q.at + q.bytes; }
  const size_t o_doc = parts[0].at, o_flags = parts[1].at, o_pool = parts[2].at, o_prec = parts[3].at, o_list = parts[4].at, o_have = parts[5].at, o_ls = parts[6].at,
               o_ent = parts[7].at, o_pr = parts[8].at, o_bits = parts[9].at, o_nb = parts[10].at, o_bl = parts[11].at, o_fb = parts[12].at, o_cb = parts[13].at, o_wb = parts[14].at;
  const size_t o_G = (blob_len + 255) & ~(size_t)255;
  blob_len = o_G + sizeof(SyncDev);
  std::vector<u8> blob_mem;
  std::unique_lock<std::mutex> pin_lock(ctx->pin_mu, std::defer_lock);
  u8* blob = nullptr;
  if (blob_len <= ((size_t)32 << 20)) { pin_lock.lock(); if (ensure_pins(ctx)) blob = ctx->pin[0]; else pin_lock.unlock(); }
  if (!blob) { blob_mem.resize(blob_len); blob = blob_mem.data(); }
  for (const Part& q : parts) if (q.bytes) memcpy(blob + q.at, q.p, q.bytes);
  u8* d_blob = balloc<u8>(b, blob_len + 256, &ok);
  SyncDev G;
  memset(&G, 0, sizeof G);
  G.B = b->B;
  G.n_rec = R; G.n_pool = n_pool; G.raw = raw ? 1 : 0;
  G.pool_idx = balloc<u32>(b, (size_t)n_pool + 1, &ok);
  G.c_flag = balloc<u8>(b, n_flag + 16, &ok);
  G.clock = balloc<u64>(b, n_clock + 1, &ok);
  G.words = balloc<u32>(b, n_word + 1, &ok);
  G.r_kind = balloc<u32>(b, (size_t)R * 6 + 8, &ok);
  b->sm_off.p = ctx->pool.alloc(((size_t)R + 2) * 4, &b->sm_off.cap);
  b->sm_job.p = ctx->pool.alloc(((size_t)R + 2) * 4, &b->sm_job.cap);
  if (!ok || !b->sm_off.p || !b->sm_job.p) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  G.msg_off = (u32*)b->sm_off.p;
  G.job_base = (u32*)b->sm_job.p;
  G.r_nheads = G.r_kind + R; G.r_nneed = G.r_kind + 2 * (size_t)R; G.r_nsend = G.r_kind + 3 * (size_t)R; G.r_bloom_n = G.r_kind + 4 * (size_t)R; G.r_bits_at = G.r_kind + 5 * (size_t)R;
  G.r_doc = (const u32*)(d_blob + o_doc); G.r_flags = (const u32*)(d_blob + o_flags); G.pool = d_blob + o_pool;
  G.pool_rec = (const u32*)(d_blob + o_prec); G.r_list = (const u32*)(d_blob + o_list); G.r_have = (const u32*)(d_blob + o_have);
  G.h_ls = (const u32*)(d_blob + o_ls); G.h_entries = (const u32*)(d_blob + o_ent); G.h_probes = (const u32*)(d_blob + o_pr);
  G.h_bits = (const u64*)(d_blob + o_bits); G.h_nbytes = (const u32*)(d_blob + o_nb); G.blooms = d_blob + o_bl;
  G.r_flag_base = (const u64*)(d_blob + o_fb); G.r_clock_base = (const u64*)(d_blob + o_cb); G.r_word_base = (const u64*)(d_blob + o_wb);
  memcpy(blob + o_G, &G, sizeof G);
  SyncDev* dG = (SyncDev*)(d_blob + o_G);
  if (!rt_h2d(d_blob, blob, blob_len, ctx->stream)) { set_error(ctx, "H2D copy failed"); return AMH_ERR_DEVICE; }
  if (pin_lock.owns_lock()) { if (!rt_sync(ctx->stream)) { set_error(ctx, "H2D copy failed"); return AMH_ERR_DEVICE; } pin_lock.unlock(); }   // the pinned buffer is free again
  launch_sm_timed(ctx, SK_LOOKUP, dG, n_pool);
  launch_sm_timed(ctx, SK_PLAN, dG, R);
  u32 total = 0, jobs = 0;
  if (!scan_inplace(b, G.msg_off, R, &total) || !scan_inplace(b, G.job_base, R, &jobs)) { set_error(ctx, "scan failed"); return AMH_ERR_DEVICE; }
  if (total == 0xFFFFFFFFu || jobs == 0xFFFFFFFFu) { set_error(ctx, "sync messages exceed 4 GiB: generate them in parts"); return AMH_ERR_BATCH_TOO_LARGE; }
  pt.mark("upload + lookup + plan + scans");
  b->sm_msg.p = ctx->pool.alloc((size_t)total + 64, &b->sm_msg.cap);
  b->sm_sent.p = ctx->pool.alloc((size_t)jobs * 32 + 32, &b->sm_sent.cap);
  ok = ok && b->sm_msg.p && b->sm_sent.p;
  G.msg_out = (u8*)b->sm_msg.p;
  G.sent_out = (u8*)b->sm_sent.p;
  G.job_src = balloc<u64>(b, (size_t)jobs + 1, &ok);
  G.job_dst = balloc<u64>(b, (size_t)jobs + 1, &ok);
  G.job_len = balloc<u32>(b, (size_t)jobs + 1, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  if (!rt_h2d(dG, &G, sizeof G, ctx->stream)) { set_error(ctx, "H2D copy failed"); return AMH_ERR_DEVICE; }
  if (n_flag >= 0xFFFFFFFFull) { set_error(ctx, "amh_generate_sync_messages: too many (record, change) pairs in one call"); return AMH_ERR_BATCH_TOO_LARGE; }
  rt_memset(G.msg_out, 0, (size_t)total + 64, ctx->stream);   // the Bloom filters' bits are ORed in
  launch_sm_timed(ctx, SK_FILL, dG, R);
  if (!raw) launch_sm_timed(ctx, SK_BLOOM, dG, (u32)n_flag);
  launch_sm_timed(ctx, SK_JOBS, dG, jobs);
  if (jobs) {
    amh_ctx::Pending p;
    p.k = AK_NUM + MK_NUM + SK_COPY;
    if (ctx->profiling) { rt_event_create(&p.a); rt_event_create(&p.b); rt_event_record(&p.a, ctx->stream); }
    amh_copy_bytes64(G.job_src, G.job_len, G.job_dst, G.msg_out, jobs, (void*)ctx->stream);
    if (ctx->profiling) { rt_event_record(&p.b, ctx->stream); ctx->pending.push_back(p); }
  }
  std::vector<u32> h_kind(R);
  if (!rt_d2h(h_kind.data(), G.r_kind, (size_t)R * 4, ctx->stream)) { set_error(ctx, std::string("device pipeline failed: ") + rt_last_error()); return AMH_ERR_DEVICE; }
  pt.mark("fill + copy", &ctx->stream);
  collect_profile(ctx);
  b->sm_total = total; b->sm_jobs = jobs; b->sm_n = R;
  int rc = AMH_SUCCESS;
  for (u32 r = 0; r < R; r++) { kinds[r] = h_kind[r]; if (h_kind[r] == SMK_FAILED) rc = AMH_ERR_DOC_FAILED; }
  if (rc == AMH_ERR_DOC_FAILED) set_error(ctx, "amh_generate_sync_messages: some records name documents that failed to load");
  return rc;
}
extern "C" int amh_generate_sync_messages_device(amh_ctx* ctx, const amh_batch* bc, const amh_sync_state* states, size_t n, uint32_t* kinds,
                                                 const uint8_t** d_msgs, const uint32_t** d_msg_off, uint64_t* total,
                                                 const uint8_t** d_sent, const uint32_t** d_sent_idx, uint64_t* n_sent) {
  if (!ctx || !bc || !d_msgs || !d_msg_off || !total || (n && (!states || !kinds)) || n >= 0x7FFFFFFFull) { set_error(ctx, "amh_generate_sync_messages_device: invalid argument"); return AMH_ERR_INVALID_ARG; }
  amh_batch* b = const_cast<amh_batch*>(bc);
  *d_msgs = nullptr; *d_msg_off = nullptr; *total = 0;
  if (d_sent) *d_sent = nullptr;
  if (d_sent_idx) *d_sent_idx = nullptr;
  if (n_sent) *n_sent = 0;
  int rc = sm_to_device(ctx, b, states, n, kinds);
  if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) { sm_release(ctx, b); return rc; }
  *d_msgs = (const uint8_t*)b->sm_msg.p; *d_msg_off = (const uint32_t*)b->sm_off.p; *total = b->sm_total;
  if (d_sent) *d_sent = (const uint8_t*)b->sm_sent.p;
  if (d_sent_idx) *d_sent_idx = (const uint32_t*)b->sm_job.p;
  if (n_sent) *n_sent = b->sm_jobs;
  return rc;
}
extern "C" int amh_generate_sync_messages(amh_ctx* ctx, const amh_batch* bc, const amh_sync_state* states, size_t n, uint32_t* kinds,
                                          amh_bytes_out* msgs, uint64_t** msg_off, amh_bytes_out* sent, uint64_t** sent_off) {
  if (!ctx || !bc || !msgs || !msg_off || (sent && !sent_off) || (n && (!states || !kinds)) || n >= 0x7FFFFFFFull) { set_error(ctx, "amh_generate_sync_messages: invalid argument"); return AMH_ERR_INVALID_ARG; }
  amh_batch* b = const_cast<amh_batch*>(bc);
  msgs->data = nullptr; msgs->len = 0; *msg_off = nullptr;
  if (sent) { sent->data = nullptr; sent->len = 0; *sent_off = nullptr; }
  struct OutGuard {   // a failed call hands nothing out
    amh_ctx* ctx; amh_batch* b; amh_bytes_out *m, *s; uint64_t **mo, **so; bool keep = false;
    ~OutGuard() {
      sm_release(ctx, b);   // the host copy is the result of this entry point: nothing stays on the device
      if (keep) return;
      amh_bytes_free(m); free(*mo); *mo = nullptr;
      if (s) { amh_bytes_free(s); free(*so); *so = nullptr; }
    }
  } guard{ctx, b, msgs, sent, msg_off, sent_off};
  *msg_off = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
  if (sent) *sent_off = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
  if (!*msg_off || (sent && !*sent_off)) return AMH_ERR_OOM;
  int rc = sm_to_device(ctx, b, states, n, kinds);
  if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) return rc;
  if (n == 0) { guard.keep = true; return rc; }
  PhaseTimer pt("amh_generate_sync_messages (host copy)");
  const u32 R = (u32)n;
  const size_t total = (size_t)b->sm_total, jobs = (size_t)b->sm_jobs;
  std::vector<u32> h_off(R + 1), h_job(R + 1);
  msgs->data = (uint8_t*)malloc(total ? total : 1);
  if (sent) sent->data = (uint8_t*)malloc(jobs ? jobs * 32 : 1);
  if (!msgs->data || (sent && !sent->data)) return AMH_ERR_OOM;
  if (!d2h_large(ctx, msgs->data, b->sm_msg.p, total) || (sent && !d2h_large(ctx, sent->data, b->sm_sent.p, jobs * 32)) ||
      !rt_d2h(h_off.data(), b->sm_off.p, ((size_t)R + 1) * 4, ctx->stream) || !rt_d2h(h_job.data(), b->sm_job.p, ((size_t)R + 1) * 4, ctx->stream)) {
    set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE;
  }
  msgs->len = total;
  if (sent) sent->len = jobs * 32;
  for (u32 r = 0; r <= R; r++) {
    (*msg_off)[r] = h_off[r];
    if (sent) (*sent_off)[r] = (uint64_t)h_job[r] * 32;
  }
  pt.mark("to the host");
  guard.keep = true;
  return rc;
}
// == Automerge::save_after(heads) (automerge.rs:993-1000) for many (document, heads) pairs in one pass on the device: the
// clock of the heads' ancestors, the changes above it in history order (get_changes_clock, :1022-1047), their chunks
// gathered in HBM and copied out once.  The sync message kernels in their raw mode (amh_syncmsg.inl).
extern "C" int amh_save_after_batch(amh_ctx* ctx, const amh_batch* bc, const uint32_t* doc_idx, const uint8_t* const* heads32, const size_t* n_heads,
                                    size_t n, amh_bytes_out* out, uint64_t** off) {
  if (!ctx || !bc || !out || !off || (n && (!doc_idx || !heads32 || !n_heads)) || n >= 0x7FFFFFFFull) { set_error(ctx, "amh_save_after_batch: invalid argument"); return AMH_ERR_INVALID_ARG; }
  amh_batch* b = const_cast<amh_batch*>(bc);
  out->data = nullptr; out->len = 0; *off = nullptr;
  std::vector<amh_sync_state> st(n);
  std::vector<amh_sync_have> hv(n);
  std::vector<uint32_t> kinds(n ? n : 1);
  for (size_t i = 0; i < n; i++) {
    if (n_heads[i] && !heads32[i]) { set_error(ctx, "amh_save_after_batch: invalid argument"); return AMH_ERR_INVALID_ARG; }
    memset(&st[i], 0, sizeof st[i]);
    hv[i] = amh_sync_have{heads32[i], n_heads[i], nullptr, 0};
    st[i].doc = doc_idx[i];
    st[i].flags = AMH_SYNC_THEIR_HAVE | AMH_SYNC_THEIR_NEED;
    st[i].their_have = &hv[i];
    st[i].n_their_have = 1;
  }
  uint64_t* offs = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
  if (!offs) return AMH_ERR_OOM;
  int rc = sm_to_device(ctx, b, st.data(), n, kinds.data(), true);
  if ((rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) || n == 0) { sm_release(ctx, b); if (rc == AMH_SUCCESS) *off = offs; else free(offs); return rc; }
  const size_t total = (size_t)b->sm_total;
  std::vector<u32> h_off(n + 1);
  out->data = (uint8_t*)malloc(total ? total : 1);
  bool ok = out->data && d2h_large(ctx, out->data, b->sm_msg.p, total) && rt_d2h(h_off.data(), b->sm_off.p, (n + 1) * 4, ctx->stream);
  sm_release(ctx, b);
  if (!ok) {
    const int e = out->data ? AMH_ERR_DEVICE : AMH_ERR_OOM;
    free(out->data); out->data = nullptr; free(offs);
    set_error(ctx, "amh_save_after_batch: copy to the host failed");
    return e;
  }
  for (size_t i = 0; i <= n; i++) offs[i] = h_off[i];
  out->len = total;
  *off = offs;
  return rc;
}
extern "C" int amh_comm_unique_id(uint8_t* id128) {
  if (!id128) return AMH_ERR_INVALID_ARG;
#ifdef AMH_EMU
  memset(id128, 0, 128);
  return AMH_SUCCESS;
#else
  const char* e = amh_comm_make_id(id128);
  if (e) { set_error(nullptr, std::string("amh_comm_unique_id: ") + e); return AMH_ERR_DEVICE; }
  return AMH_SUCCESS;
#endif
}
extern "C" int amh_comm_init(amh_ctx* ctx, const uint8_t* id128, int rank, int world) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { set_error(ctx, "amh_comm_init: invalid argument"); return AMH_ERR_INVALID_ARG; }
#ifdef AMH_EMU
  if (world != 1) { set_error(ctx, "amh_comm_init: the emulation build has no collective"); return AMH_ERR_DEVICE; }
  return AMH_SUCCESS;
#else
  if (ctx->comm) { amh_comm_close(ctx->comm); ctx->comm = nullptr; }
  const char* e = nullptr;
  ctx->comm = amh_comm_open(id128, rank, world, ctx->device, &e);
  if (!ctx->comm) { set_error(ctx, std::string("amh_comm_init: ") + (e ? e : "failed")); return AMH_ERR_DEVICE; }
  return AMH_SUCCESS;
#endif
}
// == the byte-balanced prefix split of SURVEY.md 8e: documents are independent, so a rank takes a contiguous range; the
// ranges are cut where the running byte count crosses k / world of the total (load time follows the encoded bytes, not
// the document count).  first[world + 1]: rank k loads documents [first[k], first[k + 1]).
extern "C" int amh_partition_docs(const size_t* lens, size_t n, int world, size_t* first) {
  if ((n && !lens) || world < 1 || !first) return AMH_ERR_INVALID_ARG;
  unsigned __int128 total = 0;
  for (size_t i = 0; i < n; i++) total += lens[i];
  size_t i = 0;
  unsigned __int128 run = 0;
  first[0] = 0;
  for (int k = 1; k < world; k++) {
    unsigned __int128 want = total * (unsigned)k / (unsigned)world;
    // a document goes to the rank in whose share its middle byte falls
    while (i < n && run + lens[i] / 2 < want) run += lens[i++];
    first[k] = i;
  }
  first[world] = n;
  return AMH_SUCCESS;
}
// == ReadDoc::text for every document of the batch at once (read.rs; OpSet::text op_set.rs:307-311, OpType::to_str
// types.rs:278-286): per document the text of its text objects (in document order, back to back: a document with one text
// object — the shape of BASELINE.json configs[2] / [3] — gives exactly text(obj)).  Winners come from the load's
// visibility pass; per row the bytes it contributes, one scan, one fill.  out = the texts back to back (UTF-8),
// offsets[n_docs + 1] = where each document's text starts (freed with amh_bytes_free / amh_u64_free); a failed document
// has an empty text.
// the texts into device memory owned by the batch (b->text_out / text_doc); total bytes in b->text_total
static int text_to_device(amh_ctx* ctx, amh_batch* b) {
  BatchDev B = b->B;
  const u32 n = B.n_docs;
  if (b->text_out.p) { ctx->pool.release(b->text_out.p, b->text_out.cap); b->text_out = DevAlloc(); }
  if (b->text_doc.p) { ctx->pool.release(b->text_doc.p, b->text_doc.cap); b->text_doc = DevAlloc(); }
  b->text_total = 0;
  b->text_doc.p = ctx->pool.alloc(((size_t)n + 2) * sizeof(u32), &b->text_doc.cap);
  if (!b->text_doc.p) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  B.tx_doc = (u32*)b->text_doc.p;
  if (n == 0 || B.tot_rows == 0) { rt_memset(B.tx_doc, 0, ((size_t)n + 2) * sizeof(u32), ctx->stream); return rt_sync(ctx->stream) ? AMH_SUCCESS : AMH_ERR_DEVICE; }
  bool ok = true;
  PhaseTimer pt("amh_text_batch");
  CallScratch scratch(b);
  B.tx_off = balloc<u32>(b, (size_t)B.tot_rows + 2, &ok);
  if (!ok) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  launch(ctx, AK_TEXT_LEN, B, B.tot_rows);
  u32 total = 0;
  if (!scan_inplace(b, B.tx_off, B.tot_rows, &total) || !rt_sync(ctx->stream)) { set_error(ctx, "scan failed"); return AMH_ERR_DEVICE; }
  if (total == 0xFFFFFFFFu) { set_error(ctx, "amh_text_batch: the texts of the batch exceed 4 GiB"); return AMH_ERR_BATCH_TOO_LARGE; }
  pt.mark("len + scan", &ctx->stream);
  b->text_out.p = ctx->pool.alloc((size_t)total + 64, &b->text_out.cap);
  if (!b->text_out.p) { set_error(ctx, "device out of memory"); return AMH_ERR_OOM; }
  B.tx_out = (u8*)b->text_out.p;
  launch(ctx, AK_TEXT_FILL, B, B.tot_rows);
  launch(ctx, AK_TEXT_DOC, B, n + 1);
  if (!rt_sync(ctx->stream)) { set_error(ctx, std::string("device pipeline failed: ") + rt_last_error()); return AMH_ERR_DEVICE; }
  pt.mark("fill", &ctx->stream);
  b->text_total = total;
  collect_profile(ctx);
  return AMH_SUCCESS;
}
extern "C" int amh_text_batch_device(amh_ctx* ctx, const amh_batch* bc, const uint8_t** d_text, const uint32_t** d_doc_off, uint64_t* total) {
  if (!ctx || !bc || !d_text || !d_doc_off || !total) { set_error(ctx, "amh_text_batch_device: invalid argument"); return AMH_ERR_INVALID_ARG; }
  AMH_NEEDS_OPS(bc, "amh_text_batch_device")
  amh_batch* b = const_cast<amh_batch*>(bc);
  int rc = text_to_device(ctx, b);
  if (rc != AMH_SUCCESS) return rc;
  *d_text = (const uint8_t*)b->text_out.p; *d_doc_off = (const uint32_t*)b->text_doc.p; *total = b->text_total;
  return AMH_SUCCESS;
}
extern "C" int amh_text_batch(amh_ctx* ctx, const amh_batch* bc, amh_bytes_out* out, uint64_t** offsets) {
  if (!ctx || !bc || !out || !offsets) { set_error(ctx, "amh_text_batch: invalid argument"); return AMH_ERR_INVALID_ARG; }
  AMH_NEEDS_OPS(bc, "amh_text_batch")
  amh_batch* b = const_cast<amh_batch*>(bc);
  const u32 n = b->B.n_docs;
  out->data = nullptr; out->len = 0; *offsets = nullptr;
  int rc = text_to_device(ctx, b);
  if (rc != AMH_SUCCESS) return rc;
  const size_t total = (size_t)b->text_total;
  uint64_t* offs = (uint64_t*)calloc((size_t)n + 1, sizeof(uint64_t));
  out->data = (uint8_t*)malloc(total ? total : 1);
  if (!offs || !out->data) { free(offs); free(out->data); out->data = nullptr; return AMH_ERR_OOM; }
  std::vector<u32> doc32((size_t)n + 1);
  if (!rt_d2h(doc32.data(), b->text_doc.p, ((size_t)n + 1) * sizeof(u32), ctx->stream) || (total && !d2h_large(ctx, out->data, b->text_out.p, total)) || !rt_sync(ctx->stream)) {
    free(out->data); out->data = nullptr; free(offs); set_error(ctx, "D2H copy failed"); return AMH_ERR_DEVICE;
  }
  for (u32 d = 0; d <= n; d++) offs[d] = doc32[d];
  out->len = total;
  *offsets = offs;
  return AMH_SUCCESS;
}

// The records of amh_sync_summaries written into the caller's device buffer, with nothing allocated on the host or the
// device: no pool, no event, no copy to the host (include/automerge_hip.h).  Queues the kernels on the context's own
// stream and waits for them (hipStreamSynchronize allocates nothing), so d_out / d_meta are complete on return; because
// of that wait and the private stream the call cannot itself be captured into a HIP graph.
extern "C" int amh_sync_summaries_noalloc(amh_ctx* ctx, const amh_batch* b, uint8_t* d_out, size_t cap, uint32_t* d_meta) {
  if (!ctx || !b || !d_meta || (cap && !d_out)) { set_error(ctx, "invalid argument"); return AMH_ERR_INVALID_ARG; }
#ifdef AMH_EMU
  return AMH_ERR_UNIMPLEMENTED;
#else
  if (b->B.n_docs > AMH_NOALLOC_MAX_DOCS) { set_error(ctx, "amh_sync_summaries_noalloc: more than 1,048,576 documents in the batch (static scratch limit)"); return AMH_ERR_BATCH_TOO_LARGE; }
  if (cap > 0xFFFFFFFFull) cap = 0xFFFFFFFFull;   // record offsets are 32-bit
  rt_set_device(ctx->device);
  const char* err = nullptr;
  int rc = amh_sync_noalloc_run(b->B, ctx->device, d_out, (u32)cap, d_meta, (void*)ctx->stream, &err);
  if (rc == 2) { set_error(ctx, err ? err : "over the static limit"); return AMH_ERR_BATCH_TOO_LARGE; }
  if (rc != 0) { set_error(ctx, err ? err : "device error"); return AMH_ERR_DEVICE; }
  if (!rt_sync(ctx->stream)) { set_error(ctx, std::string("device pipeline failed: ") + rt_last_error()); return AMH_ERR_DEVICE; }
  return AMH_SUCCESS;
#endif
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" amh_ctx* amh_ctx_create(const int* gpu_ids, int n_gpus) {
  if (n_gpus != 1 || !gpu_ids) { set_error(nullptr, "amh_ctx_create: exactly one GPU per context (one process per GPU)"); return nullptr; }
  auto* ctx = new amh_ctx();
  ctx->device = gpu_ids[0];
  std::string err;
  if (!rt_init(ctx->device, &ctx->stream, err)) {
    set_error(nullptr, err);
    delete ctx;
    return nullptr;
  }
  if (!rt_stream_create(&ctx->side)) {
    set_error(nullptr, "stream creation failed");
    rt_shutdown(ctx->stream);
    delete ctx;
    return nullptr;
  }
  rt_event_create(&ctx->ev_fork);
  rt_event_create(&ctx->ev_fork2);
  rt_event_create(&ctx->ev_join);
  return ctx;
}
extern "C" void amh_ctx_destroy(amh_ctx* ctx) {
  if (!ctx) return;
  rt_sync(ctx->stream);
  rt_sync(ctx->side);
  ctx->pool.trim();
  if (ctx->stage) rt_host_free(ctx->stage);
  if (ctx->heads) rt_host_free(ctx->heads);
  if (ctx->tabs) rt_host_free(ctx->tabs);
  if (ctx->res) rt_host_free(ctx->res);
  if (ctx->bigc) rt_host_free(ctx->bigc);
#ifndef AMH_EMU
  if (ctx->comm) { amh_comm_close(ctx->comm); ctx->comm = nullptr; }
#endif
  for (int k = 0; k < 2; k++) if (ctx->pin[k]) { rt_host_free(ctx->pin[k]); rt_event_destroy(&ctx->ev_pin[k]); }
  rt_event_destroy(&ctx->ev_fork);
  rt_event_destroy(&ctx->ev_fork2);
  rt_event_destroy(&ctx->ev_join);
  rt_shutdown(ctx->side);
  rt_shutdown(ctx->stream);
  delete ctx;
}
extern "C" const char* amh_last_error(const amh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }
extern "C" int amh_ctx_set_limits(amh_ctx* ctx, uint64_t max_doc_rows, uint64_t max_batch_bytes) {
  if (!ctx || max_doc_rows > AMH_MAX_ROWS_PER_DOC) return AMH_ERR_INVALID_ARG;
  ctx->max_doc_rows = max_doc_rows ? (u32)max_doc_rows : AMH_MAX_ROWS_PER_DOC;
  ctx->max_batch_bytes = max_batch_bytes;
  return AMH_SUCCESS;
}
extern "C" int amh_ctx_set_profiling(amh_ctx* ctx, int enable) {
  if (!ctx) return AMH_ERR_INVALID_ARG;
  ctx->profiling = enable != 0;
  return AMH_SUCCESS;
}
extern "C" int amh_ctx_num_kernels(void) { return AK_NUM + MK_NUM + SK_NUM; }
extern "C" const char* amh_ctx_kernel_name(int k) { return (k >= 0 && k < AK_NUM) ? AMH_KERNEL_NAMES[k] : (k >= AK_NUM && k < AK_NUM + MK_NUM) ? MK_KERNEL_NAMES[k - AK_NUM] : (k >= AK_NUM + MK_NUM && k < AK_NUM + MK_NUM + SK_NUM) ? SK_KERNEL_NAMES[k - AK_NUM - MK_NUM] : nullptr; }
extern "C" int amh_ctx_kernel_time(amh_ctx* ctx, int k, double* ms, uint64_t* launches) {
  if (!ctx || k < 0 || k >= AK_NUM + MK_NUM + SK_NUM) return AMH_ERR_INVALID_ARG;
  if (ms) *ms = ctx->kernel_ms[k];
  if (launches) *launches = ctx->kernel_calls[k];
  return AMH_SUCCESS;
}
extern "C" int amh_ctx_pool_bytes(const amh_ctx* ctx, uint64_t* live, uint64_t* pooled) {
  if (!ctx) return AMH_ERR_INVALID_ARG;
  if (live) *live = ctx->pool.live_bytes;
  if (pooled) *pooled = ctx->pool.pooled_bytes;
  return AMH_SUCCESS;
}
extern "C" int amh_ctx_reset_profile(amh_ctx* ctx) {
  if (!ctx) return AMH_ERR_INVALID_ARG;
  for (int k = 0; k < AK_NUM + MK_NUM + SK_NUM; k++) { ctx->kernel_ms[k] = 0; ctx->kernel_calls[k] = 0; }
  return AMH_SUCCESS;
}
