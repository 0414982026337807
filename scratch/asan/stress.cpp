// Host-side memory check of libsw_hip.so: the library's host code built with AddressSanitizer (device code is not
// instrumented) driven through the paths that juggle memory -- result slabs and their cache, the pooled page-locked
// arenas, sub-batches in flight, the work-package fast path, databases, the node handle, bulk and mixed frees,
// several caller threads.  Prints "asan stress ok" if every call succeeded; ASan aborts on the first bad access.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sw_worker.h"

static std::vector<std::string> synth(uint64_t seed, uint64_t n, uint32_t lo, uint32_t hi) {
  std::vector<uint32_t> lens(n);
  if (sw_synth_lengths(seed, 0, n, lo, hi, lens.data())) exit(2);
  std::vector<uint64_t> off(n + 1, 0);
  for (uint64_t i = 0; i < n; i++) off[i + 1] = off[i] + lens[i];
  std::string flat(off[n], '\0');
  if (sw_synth_fill(seed, nullptr, n, lo, hi, &flat[0], off.data())) exit(2);
  std::vector<std::string> out(n);
  for (uint64_t i = 0; i < n; i++) out[i] = flat.substr(off[i], lens[i]);
  return out;
}

static void package(uint64_t seed, size_t nq, size_t nt, uint32_t qlo, uint32_t qhi, uint32_t tlo, uint32_t thi, int free_mode) {
  std::vector<std::string> qs = synth(seed, nq, qlo, qhi), ts = synth(seed + 1, nt, tlo, thi);
  std::vector<const char*> seqs;
  for (auto& s : qs) seqs.push_back(s.c_str());
  for (auto& s : ts) seqs.push_back(s.c_str());
  std::vector<sw_work_pair> pairs;
  for (size_t a = 0; a < nq; a++)
    for (size_t b = 0; b < nt; b++) pairs.push_back(sw_work_pair{(uint32_t)a, (uint32_t)(nq + b)});
  sw_work_package pkg{seqs.data(), seqs.size(), pairs.data(), pairs.size(), 2, 1, 1};
  std::vector<sw_alignment_detail> out(pairs.size());
  const int rc = sw_execute_work(&pkg, out.data());
  if (rc < 0) {
    fprintf(stderr, "sw_execute_work: %s\n", sw_hip_last_error());
    exit(3);
  }
  size_t sum = 0;
  for (auto& d : out) sum += strlen(d.query_alignment) + (size_t)d.score;
  if (free_mode == 0) {
    sw_alignment_details_free(out.data(), out.size());
  } else {  // the strings first for some, the rest in bulk
    for (size_t i = 0; i < out.size(); i += 3) sw_alignment_details_free(&out[i], 1);
    sw_alignment_details_free(out.data(), out.size());
  }
  fprintf(stderr, "package %zux%zu: rc %d, checksum %zu\n", nq, nt, rc, sum);
}

static void batch_and_frees(uint64_t seed) {
  std::vector<std::string> qs = synth(seed, 4, 30, 60), ts = synth(seed + 1, 9000, 20, 80);
  std::vector<const char*> q, t;
  for (size_t i = 0; i < 36000; i++) q.push_back(qs[i % 4].c_str()), t.push_back(ts[i % 9000].c_str());
  AlignmentScores sc{(uint16_t)-1, 2, (uint16_t)-1};
  for (int round = 0; round < 3; round++) {
    std::vector<Result*> out(q.size(), nullptr);
    if (sw_align_batch(q.data(), t.data(), q.size(), sc, SW_PATH_LOW_MEMORY, out.data())) exit(4);
    for (size_t i = 0; i < out.size(); i += 5) {
      if (i % 2) free_c_string(out[i]->target_ptr);
      free_alignment_result(out[i]);
      out[i] = nullptr;
    }
    sw_free_alignment_results(out.data(), out.size());
  }
}

static void database(uint64_t seed) {
  std::vector<std::string> ts = synth(seed, 3000, 50, 900), qs = synth(seed + 5, 3, 100, 400);
  std::string flat;
  std::vector<uint64_t> off{0};
  for (auto& s : ts) flat += s, off.push_back(flat.size());
  sw_db* db = sw_db_create(flat.data(), off.data(), ts.size());
  if (!db) exit(5);
  AlignmentScores sc{(uint16_t)-1, 2, (uint16_t)-1};
  std::vector<const char*> q;
  for (auto& s : qs) q.push_back(s.c_str());
  std::vector<sw_hit> hits(3 * 5);
  int nh[3];
  if (sw_db_search_topk_batch(db, q.data(), 3, sc, 5, hits.data(), nh)) exit(6);
  for (int i = 0; i < 3; i++) sw_hits_free(&hits[(size_t)i * 5], nh[i]);
  std::vector<int32_t> scores(ts.size());
  if (sw_db_scores(db, q[0], sc, scores.data())) exit(7);
  if (sw_db_scores_noalloc(db, q[0], sc, scores.data())) exit(8);
  int dev[2] = {0, 0};
  sw_node* node = sw_node_create(dev, 2);
  if (!node) exit(9);
  sw_node_db* ndb = sw_node_db_create(node, flat.data(), off.data(), ts.size());
  if (!ndb) exit(10);
  if (sw_node_search_topk_batch(ndb, q.data(), 3, sc, 5, hits.data(), nh)) exit(11);
  for (int i = 0; i < 3; i++) sw_hits_free(&hits[(size_t)i * 5], nh[i]);
  sw_node_db_free(ndb);
  // a work package over the node handle
  std::vector<const char*> seqs;
  for (size_t i = 0; i < 3; i++) seqs.push_back(qs[i].c_str());
  for (size_t i = 0; i < 500; i++) seqs.push_back(ts[i].c_str());
  std::vector<sw_work_pair> pairs;
  for (uint32_t a = 0; a < 3; a++)
    for (uint32_t b = 0; b < 500; b++) pairs.push_back(sw_work_pair{a, 3 + b});
  sw_work_package pkg{seqs.data(), seqs.size(), pairs.data(), pairs.size(), 2, 1, 1};
  std::vector<sw_alignment_detail> out(pairs.size());
  if (sw_node_execute_work(node, &pkg, out.data()) < 0) exit(12);
  sw_alignment_details_free(out.data(), out.size());
  sw_node_free(node);
  sw_db_free(db);
}

int main() {
  if (sw_hip_init(0)) {
    fprintf(stderr, "no GPU: %s\n", sw_hip_last_error());
    return 1;
  }
  package(1, 4, 10000, 100, 150, 100, 150, 0);    // 40 000 short pairs: sub-batches in flight, slab per sub-batch
  package(2, 4, 10000, 100, 150, 100, 150, 1);    // again (recycled slabs, pooled arenas), mixed frees
  package(3, 3, 40, 200, 600, 20000, 60000, 0);   // long targets: the fast path (chunks, windows, continuations)
  package(4, 6, 700, 800, 1100, 800, 1100, 1);    // mid-size pairs: two sub-batches
  batch_and_frees(5);
  database(6);
  std::thread a([] { package(7, 4, 9000, 60, 90, 60, 120, 0); });
  std::thread b([] { batch_and_frees(8); });
  std::thread c([] { database(9); });
  a.join();
  b.join();
  c.join();
  printf("asan stress ok\n");
  return 0;
}
