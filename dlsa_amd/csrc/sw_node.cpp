// dlsa_amd/csrc/sw_node.cpp -- several GPUs behind one handle: ONE worker process shards the target database over
// the GPUs of its node, every GPU keeps its own top-k, and the only cross-GPU traffic is a host merge of
// n_queries x k small records.
//
// What it replaces: the reference distributes work by (query, target) pair over worker PROCESSES
// (master/work_package/_scheduler/proportional_work_scheduler.py:30-55: a share of the missing pairs per worker in
// proportion to its benchmark figure) and sorts every pair's score client-side (cli/__main__.py:153-173).  Here the
// split is by TARGET, balanced by residues, inside one process; each shard is an ordinary resident database
// (sw_db.cpp) on its own device context with its own host thread, stream and scratch, so the per-GPU path is
// exactly the single-GPU path and nothing is shared on the data path.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "sw_host.h"

using namespace swhip;

struct sw_node {
  std::vector<int> devices;    // one entry per shard; a device may appear more than once (logical shards)
  std::vector<Context*> ctx;
};
struct sw_node_db {
  std::vector<Context*> ctx;
  std::vector<int> devices;
  std::vector<sw_db*> shards;
  std::vector<std::vector<uint32_t>> global_index;  // per shard: shard-local index -> index as passed to create
  uint64_t n = 0, residues = 0;
};

namespace {

constexpr uint64_t kNodeMagic = 0x0145444f4e505753ull;  // "SWPNODE" + version 1

// ---- the sharder ---------------------------------------------------------------------------------------------
// Targets are sorted by length (descending, stable on the index) -- the order in which a database packs them into
// groups of 128 lanes -- and cut into tiles of consecutive targets; tiles are dealt longest-first to the shard with
// the fewest residues so far (LPT).  A tile is what one wave works on (similar lengths), so every shard receives
// whole, well-filled groups from every length bucket, and the residue totals -- which is what the score kernel's
// time is proportional to -- differ by at most one tile.  Deterministic: ties go to the lowest shard.
void shard_plan(const uint64_t* offsets, uint64_t n, int n_shards, uint32_t* shard_of) {
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return (offsets[a + 1] - offsets[a]) > (offsets[b + 1] - offsets[b]);
  });
  // a full lane group when the database is large; smaller tiles for a small one so that every shard gets work
  const uint64_t tile = std::min<uint64_t>(kGroupTargets, std::max<uint64_t>(1, n / (4ull * (uint64_t)n_shards)));
  std::vector<uint64_t> load(n_shards, 0);
  for (uint64_t at = 0; at < n; at += tile) {
    const uint64_t end = std::min(n, at + tile);
    uint64_t w = 0;
    for (uint64_t k = at; k < end; k++) w += offsets[order[k] + 1] - offsets[order[k]];
    int best = 0;
    for (int s = 1; s < n_shards; s++)
      if (load[s] < load[best]) best = s;
    load[best] += w + 1;  // (+1: empty sequences still spread)
    for (uint64_t k = at; k < end; k++) shard_of[order[k]] = (uint32_t)best;
  }
}

// runs fn(shard) for every shard on its own host thread (the calling thread takes shard 0); a failing shard's
// error message is carried back to the caller's thread
template <class F>
int on_every_shard(size_t n_shards, F&& fn) {
  std::vector<int> rcs(n_shards, SW_OK);
  std::vector<std::string> errs(n_shards);
  auto run = [&](size_t s) noexcept {
    rcs[s] = guarded<int>(SW_E_NOMEM, [&] { return fn(s); });
    if (rcs[s] != SW_OK) {
      try {
        errs[s] = last_error_string();
      } catch (...) {
      }
    }
  };
  std::vector<std::thread> th;
  th.reserve(n_shards);
  for (size_t s = 1; s < n_shards; s++) {
    try {
      th.emplace_back(run, s);
    } catch (...) {
      run(s);
    }
  }
  if (n_shards) run(0);
  for (auto& t : th) t.join();
  for (size_t s = 0; s < n_shards; s++)
    if (rcs[s] != SW_OK) return fail(rcs[s], "shard " + std::to_string(s) + ": " + errs[s]);
  return SW_OK;
}

void node_db_release(sw_node_db* ndb) {
  if (!ndb) return;
  for (sw_db* d : ndb->shards) sw_db_free(d);
  delete ndb;
}

// ---- a batch of pairs over the GPUs of the node -------------------------------------------------------------------
// The reference's unit of distribution is the (query, target) pair (proportional_work_scheduler.py:30-55); inside a
// node the same holds: pairs are independent, so the batch is cut into one part per shard and every part is an
// ordinary single-GPU batch (align_many_on) on its own host thread.  Pairs of one query stay together where the
// balance allows it -- the work-package fast path (sw_package.cpp) works per query -- and parts are balanced by
// cells: a query's group heavier than half a shard's share is cut into pieces of consecutive pairs, and pieces are
// dealt heaviest-first to the lightest shard (LPT; ties go to the lowest shard, so the split is deterministic).
void split_pairs(const char* const* queries, const char* const* targets, size_t n, size_t S,
                 std::vector<std::vector<size_t>>& part) {
  std::unordered_map<const char*, uint64_t> len;
  auto length_of = [&](const char* s) -> uint64_t {
    if (!s) return 0;
    auto it = len.find(s);
    if (it != len.end()) return it->second;
    const uint64_t l = strlen(s);
    len.emplace(s, l);
    return l;
  };
  std::unordered_map<const char*, uint32_t> gid;
  std::vector<std::vector<size_t>> groups;
  std::vector<uint64_t> w(n);
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    // cells, plus a per-pair term for the host work around short pairs (decode, strings)
    w[i] = (length_of(queries[i]) + 1) * (length_of(targets[i]) + 1) + 4096;
    total += w[i];
    auto it = gid.find(queries[i]);
    if (it == gid.end()) {
      it = gid.emplace(queries[i], (uint32_t)groups.size()).first;
      groups.emplace_back();
    }
    groups[it->second].push_back(i);
  }
  struct Piece {
    uint64_t w;
    uint32_t group;
    size_t lo, hi;
  };
  std::vector<Piece> pieces;
  const uint64_t cap = std::max<uint64_t>(1, total / (2 * S));
  for (uint32_t g = 0; g < groups.size(); g++) {
    size_t lo = 0;
    uint64_t acc = 0;
    for (size_t k = 0; k < groups[g].size(); k++) {
      acc += w[groups[g][k]];
      if (acc >= cap || k + 1 == groups[g].size()) {
        pieces.push_back(Piece{acc, g, lo, k + 1});
        lo = k + 1;
        acc = 0;
      }
    }
  }
  std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.w > b.w; });
  std::vector<uint64_t> load(S, 0);
  part.assign(S, {});
  for (const Piece& pc : pieces) {
    size_t best = 0;
    for (size_t s = 1; s < S; s++)
      if (load[s] < load[best]) best = s;
    load[best] += pc.w;
    for (size_t k = pc.lo; k < pc.hi; k++) part[best].push_back(groups[pc.group][k]);
  }
  for (auto& v : part) std::sort(v.begin(), v.end());  // the caller's order inside a part
}

int node_align_batch(const sw_node* node, const char* const* queries, const char* const* targets, size_t n,
                     AlignmentScores as, int path, Result** out) {
  if (!node || node->devices.empty() || (n && (!queries || !targets || !out))) return fail(SW_E_ARG, "bad argument");
  if (path < 0 || path > 2) return fail(SW_E_ARG, "bad path");
  for (size_t i = 0; i < n; i++) out[i] = nullptr;
  const size_t S = node->devices.size();
  if (S == 1 || n < 2) return align_many_on(node->devices[0], queries, targets, n, as, path, out);
  std::vector<std::vector<size_t>> part;
  split_pairs(queries, targets, n, S, part);
  std::vector<PackageStats> st(S);
  const int rc = on_every_shard(S, [&](size_t s) -> int {
    const std::vector<size_t>& mine = part[s];
    if (mine.empty()) return SW_OK;
    std::vector<const char*> qs(mine.size()), ts(mine.size());
    for (size_t k = 0; k < mine.size(); k++) qs[k] = queries[mine[k]], ts[k] = targets[mine[k]];
    std::vector<Result*> res(mine.size(), nullptr);
    g_package_stats = PackageStats{};
    const int r = align_many_on(node->devices[s], qs.data(), ts.data(), mine.size(), as, path, res.data());
    st[s] = g_package_stats;
    if (r) return r;
    for (size_t k = 0; k < mine.size(); k++) out[mine[k]] = res[k];
    return SW_OK;
  });
  if (rc) {  // nothing is handed over
    const std::string msg = last_error_string();
    for (size_t i = 0; i < n; i++) {
      if (out[i]) free_alignment_result(out[i]);
      out[i] = nullptr;
    }
    return fail(rc, msg);
  }
  PackageStats sum;
  for (const PackageStats& p : st) {
    sum.pairs += p.pairs, sum.queries += p.queries, sum.windows += p.windows, sum.rounds += p.rounds;
    sum.window_cells += p.window_cells, sum.chunks += p.chunks, sum.chunk_escapes += p.chunk_escapes;
  }
  g_package_stats = sum;
  return SW_OK;
}

int node_search(sw_node_db* ndb, const char* const* queries, size_t n_queries, AlignmentScores as, int k, sw_hit* hits,
                int* n_hits) {
  if (!ndb || (!queries && n_queries) || !hits || !n_hits || k < 0) return fail(SW_E_ARG, "bad argument");
  for (size_t q = 0; q < n_queries; q++) n_hits[q] = 0;
  if (n_queries == 0 || k == 0) return SW_OK;
  std::vector<QueryPrep> qps(n_queries);
  for (size_t q = 0; q < n_queries; q++)
    if (!queries[q] || !prep_query(queries[q], qps[q]))
      return fail(SW_E_ARG, "query " + std::to_string(q) + " is not valid UTF-8");
  const Scoring sc = to_scoring(as);
  const size_t S = ndb->shards.size();
  // phase A, every GPU at once: score pass + on-device selection per query -> the shard's own k best per query
  std::vector<std::vector<std::vector<TopEntry>>> tops(S);
  int rc = on_every_shard(S, [&](size_t s) { return db_topk_phase(ndb->shards[s], qps, sc, (size_t)k, tops[s]); });
  if (rc) return rc;
  // the one exchange: S x n_queries x k records of (score, global index) merged on the host by
  // (score descending, global index ascending)
  struct Rec {
    int32_t score;
    uint32_t global, shard, local;
  };
  std::vector<std::vector<HitRef>> refs(S);
  std::vector<std::vector<size_t>> dest(S);  // where each aligned hit goes in `hits`
  std::vector<Rec> pool;
  for (size_t q = 0; q < n_queries; q++) {
    pool.clear();
    for (size_t s = 0; s < S; s++)
      for (const TopEntry& e : tops[s][q]) pool.push_back(Rec{e.score, ndb->global_index[s][e.target], (uint32_t)s, e.target});
    std::sort(pool.begin(), pool.end(), [](const Rec& a, const Rec& b) {
      return a.score != b.score ? a.score > b.score : a.global < b.global;
    });
    const size_t kk = std::min<size_t>((size_t)k, pool.size());
    for (size_t i = 0; i < kk; i++) {
      refs[pool[i].shard].push_back(HitRef{(uint32_t)q, pool[i].local, pool[i].score});
      dest[pool[i].shard].push_back(q * (size_t)k + i);
    }
    n_hits[q] = (int)kk;
  }
  // phase B: exact end cell + traceback of the merged hits only, each on the GPU that owns the target
  std::vector<std::vector<sw_hit>> aligned(S);
  for (size_t s = 0; s < S; s++) aligned[s].resize(refs[s].size());
  rc = on_every_shard(S, [&](size_t s) {
    return refs[s].empty() ? SW_OK : db_align_phase(ndb->shards[s], qps, sc, refs[s], aligned[s].data());
  });
  if (rc) {
    for (size_t s = 0; s < S; s++) sw_hits_free(aligned[s].data(), (int)aligned[s].size());  // (zeroed entries are fine)
    for (size_t q = 0; q < n_queries; q++) n_hits[q] = 0;
    return rc;
  }
  for (size_t s = 0; s < S; s++)
    for (size_t i = 0; i < refs[s].size(); i++) {
      sw_hit h = aligned[s][i];
      h.target_index = ndb->global_index[s][refs[s][i].target];
      hits[dest[s][i]] = h;
    }
  return SW_OK;
}

}  // namespace

extern "C" {

int sw_shard_plan(const uint64_t* offsets, uint64_t n, int n_shards, uint32_t* shard_of) {
  return guarded<int>(SW_E_NOMEM, [&]() -> int {
    if (!offsets || (!shard_of && n) || n_shards < 1 || n >= 0xffffffffull) return fail(SW_E_ARG, "bad argument");
    for (uint64_t i = 0; i < n; i++)
      if (offsets[i + 1] < offsets[i]) return fail(SW_E_ARG, "offsets must be non-decreasing");
    shard_plan(offsets, n, n_shards, shard_of);
    return SW_OK;
  });
}

sw_node* sw_node_create(const int* devices, int n_devices) {
  return guarded<sw_node*>(nullptr, [&]() -> sw_node* {
    std::vector<int> devs;
    if (!devices) {
      int n = 0;
      if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fail(SW_E_NOGPU, "no HIP device available; libsw_hip has no CPU fallback");
        return nullptr;
      }
      if (n_devices > 0) n = std::min(n, n_devices);
      for (int d = 0; d < n; d++) devs.push_back(d);
    } else {
      if (n_devices < 1 || n_devices > 1024) {
        fail(SW_E_ARG, "sw_node_create: bad device count");
        return nullptr;
      }
      devs.assign(devices, devices + n_devices);
    }
    std::unique_ptr<sw_node> node(new sw_node);
    for (int d : devs) {
      Context* c = nullptr;
      if (d < 0 || ctx_get(d, &c) != SW_OK) {
        if (d < 0) fail(SW_E_ARG, "sw_node_create: negative device index");
        return nullptr;
      }
      node->devices.push_back(d);
      node->ctx.push_back(c);
    }
    return node.release();
  });
}
void sw_node_free(sw_node* node) { delete node; }
int sw_node_shards(const sw_node* node) { return node ? (int)node->devices.size() : 0; }

sw_node_db* sw_node_db_create(sw_node* node, const char* flat, const uint64_t* offsets, uint64_t n) {
  return guarded<sw_node_db*>(nullptr, [&]() -> sw_node_db* {
    if (!node || (!flat && n) || !offsets || n >= 0xffffffffull) {
      fail(SW_E_ARG, "bad argument");
      return nullptr;
    }
    for (uint64_t i = 0; i < n; i++)
      if (offsets[i + 1] < offsets[i]) {
        fail(SW_E_ARG, "offsets must be non-decreasing");
        return nullptr;
      }
    const size_t S = node->devices.size();
    std::vector<uint32_t> shard_of(n);
    shard_plan(offsets, n, (int)S, shard_of.data());
    std::unique_ptr<sw_node_db, void (*)(sw_node_db*)> ndb(new sw_node_db, node_db_release);
    ndb->ctx = node->ctx;
    ndb->devices = node->devices;
    ndb->shards.assign(S, nullptr);
    ndb->global_index.assign(S, {});
    ndb->n = n;
    ndb->residues = n ? offsets[n] - offsets[0] : 0;
    std::vector<uint64_t> bytes(S, 0);
    for (uint64_t i = 0; i < n; i++) {
      ndb->global_index[shard_of[i]].push_back((uint32_t)i);
      bytes[shard_of[i]] += offsets[i + 1] - offsets[i];
    }
    // every shard gathers its sequences, packs and uploads them on its own thread
    const int rc = on_every_shard(S, [&](size_t s) -> int {
      const std::vector<uint32_t>& gi = ndb->global_index[s];
      std::string f;
      f.resize(bytes[s]);
      std::vector<uint64_t> off(gi.size() + 1, 0);
      size_t at = 0;
      for (size_t k = 0; k < gi.size(); k++) {
        const uint64_t len = offsets[gi[k] + 1] - offsets[gi[k]];
        if (len) memcpy(&f[at], flat + offsets[gi[k]], len);
        at += len;
        off[k + 1] = at;
      }
      ndb->shards[s] = db_create_owned(ndb->ctx[s], std::move(f), std::move(off));
      return ndb->shards[s] ? SW_OK : SW_E_HIP;
    });
    if (rc) return nullptr;
    return ndb.release();
  });
}
int sw_node_db_save(sw_node_db* ndb, const char* dir) {
  return guarded<int>(SW_E_NOMEM, [&]() -> int {
    if (!ndb || !dir) return fail(SW_E_ARG, "null argument");
    if (mkdir(dir, 0777) != 0 && errno != EEXIST)
      return fail(SW_E_ARG, std::string("sw_node_db_save: cannot create ") + dir + ": " + strerror(errno));
    const std::string base = std::string(dir) + "/";
    const size_t S = ndb->shards.size();
    int rc = on_every_shard(S, [&](size_t s) {
      char name[32];
      snprintf(name, sizeof name, "shard-%03zu.swdb", s);
      return sw_db_save(ndb->shards[s], (base + name).c_str());
    });
    if (rc) return rc;
    FILE* fp = fopen((base + "node.swnode").c_str(), "wb");
    if (!fp) return fail(SW_E_ARG, std::string("sw_node_db_save: cannot write the manifest in ") + dir);
    const uint64_t head[4] = {kNodeMagic, S, ndb->n, ndb->residues};
    bool ok = fwrite(head, 8, 4, fp) == 4;
    for (size_t s = 0; s < S && ok; s++) {
      const uint64_t cnt = ndb->global_index[s].size();
      ok = fwrite(&cnt, 8, 1, fp) == 1 && (cnt == 0 || fwrite(ndb->global_index[s].data(), 4, cnt, fp) == cnt);
    }
    if (fclose(fp) != 0) ok = false;
    if (!ok) return fail(SW_E_ARG, std::string("sw_node_db_save: write error in ") + dir);
    return SW_OK;
  });
}

sw_node_db* sw_node_db_open(sw_node* node, const char* dir) {
  return guarded<sw_node_db*>(nullptr, [&]() -> sw_node_db* {
    if (!node || !dir) {
      fail(SW_E_ARG, "null argument");
      return nullptr;
    }
    const std::string base = std::string(dir) + "/";
    FILE* fp = fopen((base + "node.swnode").c_str(), "rb");
    if (!fp) {
      fail(SW_E_ARG, std::string("sw_node_db_open: no manifest in ") + dir);
      return nullptr;
    }
    std::unique_ptr<sw_node_db, void (*)(sw_node_db*)> ndb(new sw_node_db, node_db_release);
    uint64_t head[4] = {0, 0, 0, 0};
    bool ok = fread(head, 8, 4, fp) == 4 && head[0] == kNodeMagic && head[1] == node->devices.size() && head[2] < 0xffffffffull;
    const size_t S = ok ? (size_t)head[1] : 0;
    ndb->global_index.assign(S, {});
    std::vector<uint8_t> seen(ok ? head[2] : 0, 0);
    for (size_t s = 0; s < S && ok; s++) {
      uint64_t cnt = 0;
      ok = fread(&cnt, 8, 1, fp) == 1 && cnt <= head[2];
      if (!ok) break;
      ndb->global_index[s].resize(cnt);
      ok = cnt == 0 || fread(ndb->global_index[s].data(), 4, cnt, fp) == cnt;
      for (size_t k = 0; k < cnt && ok; k++) {  // every database index exactly once over all shards
        const uint32_t g = ndb->global_index[s][k];
        ok = g < head[2] && !seen[g];
        if (ok) seen[g] = 1;
      }
    }
    fclose(fp);
    if (ok)
      for (uint8_t v : seen) ok = ok && v;
    if (!ok) {
      fail(SW_E_ARG, std::string("sw_node_db_open: ") + dir + ": bad manifest, or written for a different number of shards (the node has " +
                         std::to_string(node->devices.size()) + ")");
      return nullptr;
    }
    ndb->ctx = node->ctx;
    ndb->devices = node->devices;
    ndb->shards.assign(S, nullptr);
    ndb->n = head[2];
    ndb->residues = head[3];
    const int rc = on_every_shard(S, [&](size_t s) -> int {
      char name[32];
      snprintf(name, sizeof name, "shard-%03zu.swdb", s);
      ndb->shards[s] = db_open_on(ndb->ctx[s], (base + name).c_str());
      if (!ndb->shards[s]) return SW_E_ARG;
      if (sw_db_size(ndb->shards[s]) != ndb->global_index[s].size()) return fail(SW_E_ARG, "shard file and manifest disagree");
      return SW_OK;
    });
    if (rc) return nullptr;
    return ndb.release();
  });
}
void sw_node_db_free(sw_node_db* ndb) { node_db_release(ndb); }
uint64_t sw_node_db_size(const sw_node_db* ndb) { return ndb ? ndb->n : 0; }
int sw_node_db_shards(const sw_node_db* ndb) { return ndb ? (int)ndb->shards.size() : 0; }

int sw_node_db_shard_info(const sw_node_db* ndb, int shard, int* device, uint64_t* n_targets, uint64_t* residues) {
  if (!ndb || shard < 0 || (size_t)shard >= ndb->shards.size()) return fail(SW_E_ARG, "bad argument");
  if (device) *device = ndb->devices[shard];
  if (n_targets) *n_targets = sw_db_size(ndb->shards[shard]);
  if (residues) *residues = sw_db_residues(ndb->shards[shard]);
  return SW_OK;
}

int sw_node_search_topk_batch(sw_node_db* ndb, const char* const* queries, size_t n_queries, AlignmentScores as, int k,
                              sw_hit* hits, int* n_hits) {
  return guarded<int>(SW_E_NOMEM, [&] { return node_search(ndb, queries, n_queries, as, k, hits, n_hits); });
}

int sw_node_last_stats(const sw_node_db* ndb, int shard, sw_kernel_stats* out) {
  if (!ndb || !out || shard < 0 || (size_t)shard >= ndb->shards.size()) return fail(SW_E_ARG, "bad argument");
  CtxLock L(ndb->ctx[shard]);  // (no search of this shard is running while its counters are read)
  if (L.rc) return L.rc;
  *out = db_last_stats(ndb->shards[shard]);
  return SW_OK;
}

int sw_node_align_batch(const sw_node* node, const char* const* queries, const char* const* targets, size_t n_pairs,
                        AlignmentScores scores, int path, Result** out) {
  return guarded<int>(SW_E_NOMEM, [&] { return node_align_batch(node, queries, targets, n_pairs, scores, path, out); });
}

float sw_node_benchmark_cups(sw_node* node) {
  return guarded<float>(0.0f, [&]() -> float {
    if (!node || node->ctx.empty()) {
      fail(SW_E_ARG, "bad argument");
      return 0.0f;
    }
    // every DISTINCT device of the node at the same time; the node's capacity is the sum
    std::vector<Context*> distinct;
    for (Context* c : node->ctx)
      if (std::find(distinct.begin(), distinct.end(), c) == distinct.end()) distinct.push_back(c);
    std::vector<double> cups(distinct.size(), 0.0);
    const int rc = on_every_shard(distinct.size(), [&](size_t s) -> int {
      cups[s] = benchmark_cups_on(distinct[s]);
      return cups[s] > 0 ? SW_OK : SW_E_HIP;
    });
    if (rc) return 0.0f;
    double total = 0;
    for (double v : cups) total += v;
    // integer-valued: the master's field is a pydantic int (master/api_models/worker.py:7-16)
    return (float)(double)(uint64_t)total;
  });
}

}  // extern "C"
