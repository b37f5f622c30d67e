// ThreadSanitizer driver for the host loops of the library (TEST INFRASTRUCTURE, tests/emu; built with -fsanitize=thread
// by the Makefile target pool_tsan and run by tests/test_emu_fuzz.py when the sanitizer can start in this environment).
// Batches wide enough for every host pass to run on the process-wide worker pool: amh_merge_many of loaded batches,
// amh_apply_batch of host buffers, amh_generate_sync_messages over thousands of records — once from one thread, once
// from two threads with a context each (the second finds the pool busy and runs its loops on threads of its own).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/automerge_hip.h"

typedef std::vector<uint8_t> Bytes;

static int run(unsigned seed) {
  int gpu = 0;
  amh_ctx* ctx = amh_ctx_create(&gpu, 1);
  if (!ctx) return 2;
  const uint32_t N = 320;
  amh_bytes_out blob;
  uint64_t* offs = nullptr;
  std::vector<Bytes> docs[2];
  for (int k = 0; k < 2; k++) {
    if (amh_gen_map_docs(seed + 7 * k, N, 3, 4, 12, 8, 1, &blob, &offs) != 0) return 2;
    for (uint32_t i = 0; i < N; i++) docs[k].emplace_back(blob.data + offs[i], blob.data + offs[i + 1]);
    amh_bytes_free(&blob); amh_u64_free(offs);
  }
  auto load = [&](std::vector<Bytes>& d) {
    std::vector<const uint8_t*> p; std::vector<size_t> l;
    for (auto& x : d) { p.push_back(x.data()); l.push_back(x.size()); }
    return amh_load_batch(ctx, p.data(), l.data(), p.size(), nullptr);
  };
  amh_batch* a = load(docs[0]);
  amh_batch* b = load(docs[1]);
  if (!a || !b) return 2;
  // the changes of b's documents as host buffers for amh_apply_batch on a second copy of a
  std::vector<Bytes> bufs(N);
  for (uint32_t i = 0; i < N; i++) {
    size_t n = 0;
    amh_changes(b, i, nullptr, &n);
    std::vector<amh_change_view> v(n ? n : 1);
    size_t cap = n;
    if (amh_changes(b, i, v.data(), &cap) != 0) return 2;
    for (size_t k = 0; k < n; k++) bufs[i].insert(bufs[i].end(), v[k].raw_bytes, v[k].raw_bytes + v[k].raw_len);
  }
  std::vector<uint32_t> idx(N);
  for (uint32_t i = 0; i < N; i++) idx[i] = i;
  const amh_batch* srcs[1] = {b};
  int rc = amh_merge_many(ctx, a, srcs, 1, idx.data(), idx.data(), N);
  if (rc != 0) { fprintf(stderr, "merge_many: %d %s\n", rc, amh_last_error(ctx)); return 3; }
  amh_batch* a2 = load(docs[0]);
  if (!a2) return 2;
  std::vector<const uint8_t*> bp; std::vector<size_t> bl;
  for (auto& x : bufs) { bp.push_back(x.data()); bl.push_back(x.size()); }
  std::vector<int> results(N);
  rc = amh_apply_batch(ctx, a2, idx.data(), bp.data(), bl.data(), N, 0, results.data());
  if (rc != 0) { fprintf(stderr, "apply_batch: %d %s\n", rc, amh_last_error(ctx)); return 3; }
  // both routes end in the same documents
  for (uint32_t i = 0; i < N; i += 37) {
    const uint8_t *h1 = nullptr, *h2 = nullptr; size_t n1 = 0, n2 = 0;
    if (amh_heads(a, i, &h1, &n1) != 0 || amh_heads(a2, i, &h2, &n2) != 0 || n1 != n2 || memcmp(h1, h2, 32 * n1) != 0) { fprintf(stderr, "heads differ at %u\n", i); return 4; }
  }
  // one sync record per (document, 16 peers that hold nothing): 5,120 records
  const size_t R = (size_t)N * 16;
  std::vector<amh_sync_have> have(1);
  have[0].last_sync32 = nullptr; have[0].n_last_sync = 0; have[0].bloom = nullptr; have[0].bloom_len = 0;
  std::vector<amh_sync_state> st(R);
  for (size_t r = 0; r < R; r++) {
    memset(&st[r], 0, sizeof st[r]);
    st[r].doc = (uint32_t)(r % N);
    st[r].flags = AMH_SYNC_THEIR_HEADS | AMH_SYNC_THEIR_NEED | AMH_SYNC_THEIR_HAVE;
    st[r].their_have = have.data(); st[r].n_their_have = 1;
  }
  std::vector<uint32_t> kinds(R);
  amh_bytes_out msgs, sent;
  uint64_t *moff = nullptr, *soff = nullptr;
  rc = amh_generate_sync_messages(ctx, a, st.data(), R, kinds.data(), &msgs, &moff, &sent, &soff);
  if (rc != 0) { fprintf(stderr, "sync messages: %d %s\n", rc, amh_last_error(ctx)); return 3; }
  size_t with = 0;
  for (size_t r = 0; r < R; r++) with += kinds[r] == AMH_SYNC_MESSAGE;
  if (with != R) { fprintf(stderr, "messages: %zu of %zu\n", with, R); return 4; }
  amh_bytes_free(&msgs); amh_bytes_free(&sent); amh_u64_free(moff); amh_u64_free(soff);
  amh_batch_free(a); amh_batch_free(a2); amh_batch_free(b);
  amh_ctx_destroy(ctx);
  return 0;
}

int main() {
  int r0 = run(101);
  int r1 = 0, r2 = 0;
  std::thread t1([&] { r1 = run(202); }), t2([&] { r2 = run(303); });
  t1.join(); t2.join();
  printf("pool_tsan: %d %d %d\n", r0, r1, r2);
  return r0 | r1 | r2;
}
