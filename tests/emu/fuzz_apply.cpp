// Sanitizer fuzz driver for the host build of the kernel bodies (TEST INFRASTRUCTURE, tests/emu): malformed input must
// never make the pipeline read or write out of bounds — the reference's "no panics on malformed input" invariant
// (rust/automerge/tests/test.rs:1405-1415) restated for the load and apply paths.  Built with
// -fsanitize=address,undefined by tests/emu/Makefile (target fuzz_apply) and run by tests/test_emu_fuzz.py.
// Documents and changes are generated, mutated (bit flips, byte writes, truncations, splices) and pushed through
// amh_load_batch / amh_apply_batch (both modes) / amh_save_batch / amh_hydrate_json / amh_text_batch, and whatever state
// results through amh_generate_sync_messages and amh_save_after_batch with States made of known hashes, random hashes,
// random flags and damaged Bloom filters; the run passes when the sanitizers stay silent and every call returns.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/automerge_hip.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
typedef std::vector<uint8_t> Bytes;

static void mutate(Bytes& b) {
  if (b.empty()) return;
  int n = 1 + (int)(rnd() % 4);
  for (int i = 0; i < n; i++) {
    switch (rnd() % 6) {
      case 0: b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
      case 1: b[rnd() % b.size()] = (uint8_t)rnd(); break;
      case 2: b.resize(rnd() % b.size() + 1); break;
      case 3: { size_t at = rnd() % b.size(); b.insert(b.begin() + at, (uint8_t)rnd()); break; }
      case 4: { size_t at = rnd() % b.size(); b.erase(b.begin() + at); if (b.empty()) b.push_back(0); break; }
      case 5: { size_t at = rnd() % b.size(); b[at] = (rnd() & 1) ? 0xFF : 0x80; break; }
    }
  }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 200;
  if (argc > 2) rng_state ^= strtoull(argv[2], nullptr, 0);
  int gpu = 0;
  amh_ctx* ctx = amh_ctx_create(&gpu, 1);
  if (!ctx) { fprintf(stderr, "no context\n"); return 2; }
  amh_bytes_out blob;
  uint64_t* offs = nullptr;
  const uint32_t ND = 6;
  if (amh_gen_map_docs(11, 3, 3, 2, 8, 6, 1, &blob, &offs) != 0) return 2;
  std::vector<Bytes> docs;
  for (uint32_t i = 0; i < 3; i++) docs.emplace_back(blob.data + offs[i], blob.data + offs[i + 1]);
  amh_bytes_free(&blob); amh_u64_free(offs);
  if (amh_gen_text_docs(12, 1, 3, 12, 40, 10, 1, &blob, &offs) != 0) return 2;
  for (uint32_t i = 0; i < 3; i++) docs.emplace_back(blob.data + offs[i], blob.data + offs[i + 1]);
  amh_bytes_free(&blob); amh_u64_free(offs);
  // the changes of every document
  std::vector<std::vector<Bytes>> changes(ND);
  {
    std::vector<const uint8_t*> p; std::vector<size_t> l;
    for (auto& d : docs) { p.push_back(d.data()); l.push_back(d.size()); }
    amh_batch* b = amh_load_batch(ctx, p.data(), l.data(), ND, nullptr);
    if (!b) return 2;
    for (uint32_t i = 0; i < ND; i++) {
      size_t n = 0;
      amh_changes(b, i, nullptr, &n);
      std::vector<amh_change_view> v(n ? n : 1);
      size_t cap = n;
      if (amh_changes(b, i, v.data(), &cap) != 0) return 2;
      for (size_t k = 0; k < n; k++) changes[i].emplace_back(v[k].raw_bytes, v[k].raw_bytes + v[k].raw_len);
    }
    amh_batch_free(b);
  }
  long ok_docs = 0, bad_docs = 0, applied = 0, refused = 0;
  for (int it = 0; it < iters; it++) {
    // 1. load: some documents mutated, some followed by (mutated) change chunks, some change-only
    std::vector<Bytes> in;
    for (uint32_t i = 0; i < ND; i++) {
      Bytes b;
      switch (rnd() % 4) {
        case 0: b = docs[i]; mutate(b); break;
        case 1: b = docs[i]; for (auto& c : changes[(i + 1) % ND]) { Bytes m = c; if (rnd() % 3 == 0) mutate(m); b.insert(b.end(), m.begin(), m.end()); } break;
        case 2: for (auto& c : changes[i]) { Bytes m = c; if (rnd() % 4 == 0) mutate(m); b.insert(b.end(), m.begin(), m.end()); } break;
        default: b = docs[i]; break;
      }
      in.push_back(std::move(b));
    }
    std::vector<const uint8_t*> p; std::vector<size_t> l;
    for (auto& d : in) { p.push_back(d.data()); l.push_back(d.size()); }
    amh_load_opts lo; lo.verify_heads = (int)(rnd() & 1); lo.on_partial_load = (int)(rnd() & 1);
    amh_batch* b = amh_load_batch(ctx, p.data(), l.data(), ND, &lo);
    if (!b) { fprintf(stderr, "load failed as a call: %s\n", amh_last_error(ctx)); return 3; }
    for (uint32_t i = 0; i < ND; i++) { int k = 0; const char* m = nullptr; amh_doc_status(b, i, &k, &m); if (k) bad_docs++; else ok_docs++; }
    // 2. apply mutated changes in both modes
    std::vector<uint32_t> idx; std::vector<Bytes> bufs;
    for (uint32_t i = 0; i < ND; i++) {
      const auto& src = changes[rnd() % ND];
      Bytes buf;
      for (auto& c : src) { Bytes m = c; if (rnd() % 3 == 0) mutate(m); buf.insert(buf.end(), m.begin(), m.end()); }
      idx.push_back(i); bufs.push_back(std::move(buf));
    }
    std::vector<const uint8_t*> bp; std::vector<size_t> bl; std::vector<int> res(ND);
    for (auto& x : bufs) { bp.push_back(x.data()); bl.push_back(x.size()); }
    int rc = amh_apply_batch(ctx, b, idx.data(), bp.data(), bl.data(), ND, (int)(rnd() & 1), res.data());
    if (rc != AMH_SUCCESS && rc != AMH_ERR_DOC_FAILED) { fprintf(stderr, "apply failed as a call: %d %s\n", rc, amh_last_error(ctx)); return 3; }
    for (int r : res) { if (r) refused++; else applied++; }
    // 3. whatever state resulted must save and materialize
    std::vector<amh_bytes_out> outs(ND);
    amh_save_opts so; so.deflate = (int)(rnd() & 1); so.retain_orphans = 1;
    if (amh_save_batch(ctx, b, &so, outs.data()) != 0) { fprintf(stderr, "save failed: %s\n", amh_last_error(ctx)); return 3; }
    for (auto& o : outs) amh_bytes_free(&o);
    for (uint32_t i = 0; i < ND; i++) { amh_bytes_out j; if (amh_hydrate_json(b, i, &j) == 0) amh_bytes_free(&j); }
    amh_bytes_out sy;
    if (amh_sync_summaries(ctx, b, &sy) == 0) amh_bytes_free(&sy);
    // 4. the text of every document, and the sender side of the sync protocol over States put together from the hashes the
    //    batch knows, random ones, random flags and (sometimes damaged) Bloom filters: calls may fail as calls on a malformed
    //    filter (INVALID_ARG), nothing may read or write out of bounds
    { amh_bytes_out tx; uint64_t* to = nullptr; if (amh_text_batch(ctx, b, &tx, &to) == 0) { amh_bytes_free(&tx); amh_u64_free(to); } }
    {
      std::vector<Bytes> pools(ND), blooms(ND);
      std::vector<amh_sync_state> st(ND);
      std::vector<amh_sync_have> hv(ND);
      for (uint32_t i = 0; i < ND; i++) {
        const uint8_t* hs = nullptr; size_t nh = 0;
        Bytes& pool = pools[i];
        if (amh_heads(b, i, &hs, &nh) == 0) pool.insert(pool.end(), hs, hs + 32 * nh);
        size_t nc = 0;
        if (amh_changes(b, i, nullptr, &nc) == 0 && nc) {
          std::vector<amh_change_view> v(nc); size_t cap = nc;
          if (amh_changes(b, i, v.data(), &cap) == 0) for (size_t k = 0; k < nc; k++) if (rnd() % 2) pool.insert(pool.end(), v[k].hash, v[k].hash + 32);
        }
        for (int k = 0; k < 3; k++) { for (int j = 0; j < 32; j++) pool.push_back((uint8_t)rnd()); }
        const size_t np = pool.size() / 32;
        auto pick = [&](const uint8_t*& ptr, size_t& n) { size_t a = rnd() % np, c = rnd() % 4; if (a + c > np) c = np - a; ptr = pool.data() + 32 * a; n = c; };
        memset(&st[i], 0, sizeof st[i]);
        st[i].doc = i; st[i].flags = (uint32_t)(rnd() & 15);
        pick(st[i].shared_heads32, st[i].n_shared_heads); pick(st[i].last_sent_heads32, st[i].n_last_sent_heads);
        pick(st[i].their_heads32, st[i].n_their_heads); pick(st[i].their_need32, st[i].n_their_need);
        pick(st[i].sent_hashes32, st[i].n_sent_hashes);
        Bytes& bl = blooms[i];
        const uint32_t ne = (uint32_t)(rnd() % 5);
        if (ne) { bl.push_back((uint8_t)ne); bl.push_back(10); bl.push_back((uint8_t)(rnd() % 9)); for (uint32_t k = 0; k < (ne * 10 + 7) / 8; k++) bl.push_back((uint8_t)rnd()); }
        if (rnd() % 5 == 0) mutate(bl);
        pick(hv[i].last_sync32, hv[i].n_last_sync);
        hv[i].bloom = bl.data(); hv[i].bloom_len = bl.size();
        st[i].their_have = &hv[i]; st[i].n_their_have = rnd() % 2;
      }
      std::vector<uint32_t> kinds(ND);
      amh_bytes_out msgs, sent; uint64_t *mo = nullptr, *so = nullptr;
      int rc2 = amh_generate_sync_messages(ctx, b, st.data(), ND, kinds.data(), &msgs, &mo, &sent, &so);
      if (rc2 == AMH_SUCCESS || rc2 == AMH_ERR_DOC_FAILED) {
        for (uint32_t i = 0; i < ND; i++) if (mo[i + 1] < mo[i] || mo[i + 1] > msgs.len) { fprintf(stderr, "sync message offsets out of order\n"); return 3; }
        amh_bytes_free(&msgs); amh_bytes_free(&sent); amh_u64_free(mo); amh_u64_free(so);
      } else if (rc2 != AMH_ERR_INVALID_ARG) { fprintf(stderr, "sync messages failed as a call: %d %s\n", rc2, amh_last_error(ctx)); return 3; }
      std::vector<uint32_t> di(ND); std::vector<const uint8_t*> hp(ND); std::vector<size_t> hn(ND);
      for (uint32_t i = 0; i < ND; i++) { di[i] = i; hp[i] = st[i].their_heads32; hn[i] = st[i].n_their_heads; }
      amh_bytes_out sa; uint64_t* sao = nullptr;
      int rc3 = amh_save_after_batch(ctx, b, di.data(), hp.data(), hn.data(), ND, &sa, &sao);
      if (rc3 == AMH_SUCCESS || rc3 == AMH_ERR_DOC_FAILED) { amh_bytes_free(&sa); amh_u64_free(sao); }
      else { fprintf(stderr, "save_after_batch failed as a call: %d %s\n", rc3, amh_last_error(ctx)); return 3; }
    }
    amh_batch_free(b);
  }
  printf("fuzz ok: %d iterations, documents ok/bad %ld/%ld, apply outcomes ok/refused %ld/%ld\n", iters, ok_docs, bad_docs, applied, refused);
  amh_ctx_destroy(ctx);
  return 0;
}
