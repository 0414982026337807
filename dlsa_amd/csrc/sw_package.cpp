// dlsa_amd/csrc/sw_package.cpp -- the work-package fast path: every pair of a package still gets its exact end
// cell and its full traceback (the unmodified master wants one record per pair,
// internal/worker/worker_impl.go:108-168), but the Q x T matrix of a pair is filled by the packed-int16 inter-task
// score kernel, and the int32 exact kernel with direction bits only re-enters the matrix in the WINDOW the
// alignment occupies.
//
// How: the pairs of a package are grouped by query; the distinct targets of a group become a temporary resident
// database.  The score pass runs in "persist" mode: every work unit keeps its own slot of the boundary ring, so
// after the launch HBM holds the H column of every target at every unit boundary (every 256 target columns), and
// every 16-column strip reports its own maximum.  For a pair with score s the end cell lies in a strip whose maximum
// is s; the exact kernel fills the rows 1..Q of the columns from the preceding unit boundary to that strip -- a window
// of at most 256 columns, entered through the stored boundary column -- with direction bits and the reference's
// discovery-order key, and the walker follows the path from the end cell.  A path that reaches the window's left
// edge with H > 0 continues in the window to the left (rows 1..x only), as often as it takes.  Every cell the exact
// kernel computes has exactly the value of the full matrix (the boundary column is exact: the fast path is only
// taken where int16 cannot saturate), so end cell, tie-breaking and traceback are those of the full computation.
//
// Eligible: find_alignment_low_memory semantics on its ring path without quirks (T >= 2Q, T >= Qp: rust/src/
// algorithm/mod.rs:213,251), ACGT-only pair, match > 0 / mismatch < 0 / gap < 0, Q * match < 32767, T >= 2048.
// That is the reference's own synthetic distribution (utils/generate_synthetic_data.py: queries 200-1000 nt against
// targets of 10-200 kb).  Everything else takes the whole-matrix path of sw_lib.cpp.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <memory>
#include <numeric>
#include <string>
#include <thread>

#include "sw_host.h"

using namespace swhip;

namespace {

bool acgt_only(const Decoded& d) {
  if (!d.ascii) return false;
  // over the source bytes (ASCII: one per symbol), a block at a time without an exit inside the block, so that the
  // compiler makes 32 bytes per instruction of it: a package's targets are ~100 MB
  const unsigned char* s = (const unsigned char*)d.src;
  const size_t n = d.sym.size();
  for (size_t i = 0; i < n; i += 4096) {
    const size_t e = std::min(n, i + 4096);
    unsigned bad = 0;
    for (size_t k = i; k < e; k++) {
      const unsigned char ch = s[k];
      bad |= !((ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T'));
    }
    if (bad) return false;
  }
  return true;
}

// The byte buffer of a temporary database (the chunks back to back, > 100 MB for a package of long targets) is
// handed from one package to the next: a fresh std::string of that size is zero-filled and page-faulted in, which
// cost as much as two score passes.  One buffer of at most 1 GB is kept.
std::mutex g_flat_mu;
std::string g_flat_cache;
std::string flat_cache_take() {
  std::lock_guard<std::mutex> lk(g_flat_mu);
  std::string out;
  out.swap(g_flat_cache);
  return out;
}
void flat_cache_put(std::string&& s) {
  if (s.capacity() == 0 || s.capacity() > ((size_t)1 << 30)) return;
  std::lock_guard<std::mutex> lk(g_flat_mu);
  if (s.capacity() > g_flat_cache.capacity()) g_flat_cache.swap(s);
}
// a target set of at most this many residues is a job that leaves the GPU mostly idle (less than ~15 lane groups of
// chunks): several of them run side by side
constexpr uint64_t kSmallSetResidues = 4000000;
// scoring << 1 | "the first windows of the last set mostly had to be continued" (run_job: wide_first)
std::atomic<uint64_t> g_wide_hint{0};
std::atomic<uint32_t> g_wide_probe{0};
struct DbHolder {  // (of a temporary database: its maker holds the context's lock for the holder's whole life)
  sw_db* db = nullptr;
  ~DbHolder() {
    if (!db) return;
    try {
      flat_cache_put(db_take_flat(db));
    } catch (...) {
    }
    db_release_locked(db);
  }
};

}  // namespace

namespace swhip {
// counters of the last package through the fast path (test hook / profiling)
thread_local PackageStats g_package_stats;  // of the calling thread's last batch (sw_node_align_batch sums its threads')
}

int swhip::align_windows(const std::vector<Decoded>& dec, const std::vector<uint32_t>& pair_q,
                         const std::vector<uint32_t>& pair_t, Scoring sc, std::vector<PairOut>& po,
                         OpsArena& arena, std::vector<uint8_t>& handled, int device) {
  const size_t n = pair_q.size();
  handled.assign(n, 0);
  g_package_stats = PackageStats{};
  StageTimer tm0(getenv("SW_HIP_TRACE_LANES") ? 0 : 100000);
  static const bool disabled = getenv("SW_HIP_NO_WINDOWS") != nullptr;
  if (disabled || n == 0) return SW_OK;
  if (!(sc.match > 0 && sc.miss < 0 && sc.gap < 0 && sc.match <= 16384 && sc.miss >= -16384 && sc.gap >= -16384)) return SW_OK;
  // ---- eligibility, grouped by query
  auto shape_ok = [&](size_t k) {
    const uint64_t Q = dec[pair_q[k]].sym.size(), T = dec[pair_t[k]].sym.size();
    return !(Q < 32 || T < kWindowsMinTarget || T < 2 * Q || T < roundup64((uint32_t)Q) || Q * (uint64_t)sc.match >= 32767 || T >= 0x7fff0000ull);
  };
  // which sequences are pure ACGT: those of the pairs that qualify by shape, scanned once each on the host threads
  std::vector<int8_t> acgt(dec.size(), -1);
  size_t scan_bytes = 0;
  for (size_t k = 0; k < n; k++) {
    if (!shape_ok(k)) continue;
    for (uint32_t d : {pair_q[k], pair_t[k]})
      if (acgt[d] < 0) acgt[d] = 2, scan_bytes += dec[d].sym.size();
  }
  parallel_ranges(dec.size(), scan_bytes, [&](size_t lo, size_t hi) {
    for (size_t d = lo; d < hi; d++)
      if (acgt[d] == 2) acgt[d] = acgt_only(dec[d]) ? 1 : 0;
  });
  auto is_acgt = [&](uint32_t d) { return acgt[d] == 1; };
  std::map<uint32_t, std::vector<size_t>> by_query;
  double cells = 0;
  for (size_t k = 0; k < n; k++) {
    const uint64_t Q = dec[pair_q[k]].sym.size(), T = dec[pair_t[k]].sym.size();
    if (!shape_ok(k)) continue;
    if (!is_acgt(pair_q[k]) || !is_acgt(pair_t[k])) continue;
    by_query[pair_q[k]].push_back(k);
    cells += (double)Q * (double)T;
  }
  if (cells < 1e7) return SW_OK;  // a handful of small matrices: the whole-matrix path is just as fast
  tm0.lap("  windows: eligibility");
  Context* ctx = nullptr;
  int rc = ctx_get(device, &ctx);
  if (rc) return rc;
  PackageStats st{};
  // ---- one temporary database per distinct target set (a package is normally a few queries x one target set)
  std::map<std::vector<uint32_t>, std::vector<uint32_t>> by_targets;  // sorted target ids -> queries
  for (auto& kv : by_query) {
    std::vector<uint32_t> tids;
    for (size_t k : kv.second) tids.push_back(pair_t[k]);
    std::sort(tids.begin(), tids.end());
    tids.erase(std::unique(tids.begin(), tids.end()), tids.end());
    by_targets[tids].push_back(kv.first);
  }
  // (4 context lanes: the hardware queues a process has by default -- streams beyond that share a queue and take
  // turns; measured, 15 sets of one target: 1 lane 30.9 ms, 2 19.0, 3 15.2, 4 14.1, 8 15.5.  Merging the small sets
  // into one database per lane, every query passing over all of its targets, was tried and is slower: 15.4 ms.)
  static const unsigned max_lanes = [] {
    const char* e = getenv("SW_HIP_PACKAGE_LANES");
    return e ? (unsigned)std::min<long>(kMaxCtxLanes, std::max<long>(1, atol(e))) : 4u;
  }();
  tm0.lap("  windows: target sets");
  // A set whose boundary ring (one slot per work unit: residues of the chunks x Q / 128 bytes) would not fit twice the
  // scratch budget is processed in SLICES of its targets, each a temporary database of its own, instead of being
  // left to the whole-matrix path: the same passes, just more of them.
  struct SetJob {
    std::vector<uint32_t> tids;             // sorted
    const std::vector<uint32_t>* queries;   // of the whole set
    bool partial;                           // a slice: only the pairs whose target is in tids
  };
  std::vector<SetJob> jobs;
  for (auto& set : by_targets) {
    uint64_t residues = 0;
    uint32_t qmax = 0;
    for (uint32_t t : set.first) residues += dec[t].sym.size();
    for (uint32_t qid : set.second) qmax = std::max<uint32_t>(qmax, (uint32_t)dec[qid].sym.size());
    // (chunks overlap: 1.5 x the residues at the usual stride of 2 O; a slice that still comes out too large is
    // caught by the exact check against the built database below)
    const double qpad = (double)roundup64(qmax);
    const double ring = (double)residues * 1.6 * qpad / 128.0;
    // (the ring is sized by a bound on the number of units that includes three per persistent workgroup)
    const double fixed = (3.0 * ctx->cus * score_workgroups_per_cu(16) + 4.0) * qpad * 256.0;
    const double room = 2.0 * (double)ctx->scratch_budget * 0.85 - fixed;
    size_t parts = room > 0 ? (size_t)std::min<double>((double)set.first.size(), std::ceil(ring / room)) : 1;
    if (parts <= 1) {
      jobs.push_back(SetJob{set.first, &set.second, false});
      continue;
    }
    const uint64_t per = residues / parts + 1;
    std::vector<uint32_t> cur;
    uint64_t have = 0;
    for (uint32_t t : set.first) {
      cur.push_back(t);
      have += dec[t].sym.size();
      if (have >= per) {
        jobs.push_back(SetJob{std::move(cur), &set.second, true});
        cur.clear();
        have = 0;
      }
    }
    if (!cur.empty()) jobs.push_back(SetJob{std::move(cur), &set.second, true});
  }
  // One job: a temporary database of the set's chunks, a score pass per query, window rounds -- on context `ctx`
  // (every buffer it uses belongs to that context), counters into `st`.  Jobs touch distinct pairs; the caller's
  // arena is the one thing they share (arena_mu).
  std::mutex arena_mu;
  auto run_job = [&](SetJob& job, Context* ctx, PackageStats& st, StageTimer& tm0) -> int {
    int rc = SW_OK;
    const std::vector<uint32_t>& tids = job.tids;
    const std::vector<uint32_t>& set_queries = *job.queries;
    struct LapAtExit {  // (declared before the database holder: reports the time of its release too)
      StageTimer& t;
      ~LapAtExit() { t.lap("  windows: set done"); }
    } lap_at_exit{tm0};
    // ---- chunks.  A long target is a long chain of dependent strips for the score kernel (T / 16 hops, each one
    // block of rows long): few long targets leave the GPU idle.  A target is therefore cut into overlapping CHUNKS
    // that enter the temporary database as sequences of their own.  An alignment spans at most
    // O = Q + floor(Q * match / |gap|) target columns (it has at most Q query symbols, and every target-only column
    // costs |gap| of at most Q * match gained -- the bound behind the reference's ring buffer,
    // rust/src/algorithm/mod.rs:232-242), so with O columns of overlap every alignment that ENDS in a chunk's own
    // region lies inside the chunk: there the chunk's H equals the full matrix's H on the whole traceback path, and
    // every decision of the traceback falls the same way (the predecessor that wins is on the path, hence exact;
    // the losers can only be smaller).  Own regions tile the target, so score, first end cell and path are exact.
    uint32_t Qmax = 0;
    for (uint32_t qid : set_queries) Qmax = std::max<uint32_t>(Qmax, (uint32_t)dec[qid].sym.size());
    const uint64_t O = (uint64_t)Qmax + (uint64_t)Qmax * (uint64_t)sc.match / (uint64_t)(-sc.gap) + 1;
    // Chunk stride S: 2 O (half as much overlap as payload) when the set is large; a set with fewer residues gets
    // shorter chunks, down to O / 2, so that there are still about a hundred groups of 128 chunks -- every group is
    // one chain of strips, and the GPU wants that many chains in flight more than it minds the extra overlap (swept:
    // 10 x 800 targets, 8.4e7 residues: 2 O is best, 1 O +30 %, 4 O +12 %; 10 x 100 targets, 1e7 residues: 2 O 64 ms,
    // 1 O 52 ms, O / 2 35 ms)
    uint64_t set_residues = 0;
    for (uint32_t t : tids) set_residues += dec[t].sym.size();
    static const double chunk_mult = getenv("SW_HIP_CHUNK_MULT") ? atof(getenv("SW_HIP_CHUNK_MULT")) : 2.0;  // experiments
    uint64_t S = (uint64_t)(chunk_mult * (double)O);
    const uint64_t S_floor = std::max<uint64_t>(O / 2, 1024);
    const uint64_t S_many = std::max<uint64_t>(std::min<uint64_t>(S, set_residues / (100 * kGroupTargets)), S_floor);
    // ... but only where that gets somewhere: a single pair (the capacity benchmark's 4 102 x 4.2e6) has 5 groups at
    // O / 2 and 3 at the stride below, and loses a third of its speed to the extra overlap
    if (set_residues / S_many >= 32 * kGroupTargets) S = S_many;
    else S = std::max<uint64_t>(std::min<uint64_t>(S, set_residues / 256), S_floor);
    struct Chunk {
      uint32_t tid;     // index into dec
      uint32_t start;   // first target column of the chunk (0-based)
      uint32_t len;
      uint32_t own_lo;  // chunk-local columns [own_lo, len) are its own region (0-based)
    };
    std::vector<Chunk> chunks;
    std::vector<uint32_t> first_chunk(tids.size() + 1, 0);
    for (size_t i = 0; i < tids.size(); i++) {
      const uint64_t T = dec[tids[i]].sym.size();
      first_chunk[i] = (uint32_t)chunks.size();
      for (uint64_t c = 0;; c++) {
        const uint64_t start = c * S;
        if (c > 0 && start + O >= T) break;  // nothing left that the previous chunk does not own
        chunks.push_back(Chunk{tids[i], (uint32_t)start, (uint32_t)std::min<uint64_t>(S + O, T - start), (uint32_t)(c ? O : 0)});
      }
    }
    first_chunk[tids.size()] = (uint32_t)chunks.size();
    std::string flat = flat_cache_take();  // (resize below keeps what it has: no fill, pages already mapped)
    std::vector<uint64_t> off(chunks.size() + 1, 0);
    size_t total = 0;
    for (const Chunk& c : chunks) total += c.len;
    StageTimer tm(tm0.on ? 0 : 100000);
    flat.resize(total);
    for (size_t i = 0; i < chunks.size(); i++) off[i + 1] = off[i] + chunks[i].len;
    parallel_ranges(chunks.size(), total, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++)  // (ASCII: one byte per symbol)
        memcpy(&flat[off[i]], dec[chunks[i].tid].src + chunks[i].start, chunks[i].len);
    });
    tm.lap("  windows: chunks");
    // The context stays locked for the whole set: its temporary database and both workspaces are in use.
    CtxLock L(ctx);
    if (L.rc) return L.rc;
    DbHolder holder;
    holder.db = db_create_temp_locked(*L.c, std::move(flat), std::move(off));  // (the eligibility scan above vouches for ACGT)
    tm.lap("  windows: temp db");
    if (!holder.db) return SW_E_HIP;
    sw_db* db = holder.db;
    st.chunks += chunks.size();
    // Alignments that run on to the top of the matrix (cheap gaps: random DNA at match 2 / mismatch 1 / gap 1 is in
    // the linear regime) leave their first window almost always.  Once half of a query's windows needed a second
    // submission, the following queries of the set enter theirs far enough to the left right away (Q rows reach
    // ~1.5 Q columns): one submission per query instead of two, for about the same number of cells.
    // ... and what the last set with the same scoring saw carries over to the next one (a combined batch of per-pair
    // callers is many sets of one pair each: nothing to learn from inside a set).  A hint only: window sizes never
    // change a result.
    static const bool no_wide = getenv("SW_HIP_NO_WIDE_WINDOWS") != nullptr;  // experiments
    const uint64_t hint_key = ((uint64_t)(uint16_t)sc.match << 32) | ((uint64_t)(uint16_t)sc.miss << 16) | (uint16_t)sc.gap;
    // (only for sets too small to learn from themselves; every 32nd of them starts narrow again and renews the hint)
    size_t job_pairs = 0;
    for (uint32_t qid : set_queries) job_pairs += by_query.at(qid).size();
    bool wide_first = !no_wide && job_pairs < 8 && g_wide_hint.load(std::memory_order_relaxed) == (hint_key << 1 | 1) &&
                      g_wide_probe.fetch_add(1, std::memory_order_relaxed) % 32 != 31;
    // TWO score passes in flight, on the context's two workspaces: while the host picks and submits the windows of
    // query i (small, latency-bound submissions), the GPU already runs the pass of query i + 1, queued before them.
    // A slot -- boundary ring, slot scores, per-strip maxima, their host copies -- is reused as soon as its query's
    // windows are done.
    struct Flight {
      PersistPass pp;  // (its host buffers serve every other query of the set)
      QueryPrep qp;
      uint32_t qid = 0;
      bool active = false;
    };
    Flight flights[2];
    StreamDrain drain(L.c->stream);  // (declared after the flights: an early return waits for the passes in flight first)
    size_t next_query = 0;
    static const bool no_overlap = getenv("SW_HIP_NO_PASS_OVERLAP") != nullptr;  // experiments: one pass at a time
    auto start = [&](int slot) -> int {
      Flight& F = flights[slot];
      F.active = false;
      while (next_query < set_queries.size()) {
        const uint32_t qid = set_queries[next_query++];
        const Decoded& q = dec[qid];
        const uint32_t Q = (uint32_t)q.sym.size();
        if (db_persist_ring_bytes(db, Q, L.c->cus) > 2 * L.c->scratch_budget) continue;  // whole-matrix path instead
        F.qp.d.src = q.src;
        F.qp.d.ascii = true;
        F.qp.d.sym = q.sym;
        F.qp.acgt = true;
        F.qp.qwords.assign((Q + kScoreB - 1) / kScoreB + 1, 0);
        F.qp.qwords_small.assign((Q + kScoreBSmall - 1) / kScoreBSmall + 1, 0);
        for (uint32_t i = 0; i < Q; i++) {
          const uint32_t code = q.sym[i] == 'A' ? 0 : q.sym[i] == 'C' ? 1 : q.sym[i] == 'G' ? 2 : 3;
          F.qp.qwords[i / kScoreB] |= code << (2 * (i % kScoreB));
          F.qp.qwords_small[i / kScoreBSmall] |= code << (2 * (i % kScoreBSmall));
        }
        const int rc2 = db_score_persist_begin_locked(*L.c, slot ? L.c->ws2 : L.c->ws, db, F.qp, sc, F.pp);
        if (rc2) return rc2;
        F.qid = qid;
        F.active = true;
        break;
      }
      return SW_OK;
    };
    if ((rc = start(0))) return rc;
    if (!no_overlap && (rc = start(1))) return rc;
    for (int cur = 0; flights[cur].active; cur = no_overlap ? 0 : cur ^ 1) {
      PersistPass& pp = flights[cur].pp;
      const uint32_t qid = flights[cur].qid;
      const Decoded& q = dec[qid];
      const uint32_t Q = (uint32_t)q.sym.size();
      std::vector<size_t> pairs_of_slice;
      if (job.partial)
        for (size_t k : by_query.at(qid))
          if (std::binary_search(tids.begin(), tids.end(), pair_t[k])) pairs_of_slice.push_back(k);
      const std::vector<size_t>& pairs = job.partial ? pairs_of_slice : by_query.at(qid);
      if ((rc = db_score_persist_end_locked(*L.c, db, pp))) return rc;
      st.queries++;
      if (tm.on) fprintf(stderr, "[sw_hip]   windows: score kernel %.3f ms (Q = %u)\n", L.c->stats.score_kernel_ms, Q);
      tm.lap("  windows: score pass");
      const std::vector<ScoreUnit>& units = *pp.units;
      const std::vector<ScoreGroup>& groups = *pp.groups;
      std::vector<uint32_t> unit_of_elem(pp.n_elems, UINT32_MAX);
      for (uint32_t u = 0; u < units.size(); u++)
        for (uint32_t e = 0; e < units[u].count; e++) unit_of_elem[units[u].first_elem + e] = u;
      std::vector<uint32_t> slot_of(chunks.size(), UINT32_MAX);  // chunk -> lane slot
      for (size_t sl = 0; sl < pp.slot_target->size(); sl++)
        if ((*pp.slot_target)[sl] != UINT32_MAX) slot_of[(*pp.slot_target)[sl]] = (uint32_t)sl;
      // ---- round 1: the windows that can hold the end cell
      constexpr uint32_t kNearStrips = 4;
      struct Win {
        size_t pair;     // index into `pairs`
        uint32_t chunk;  // database sequence
        uint32_t unit;   // ticket of the work unit the window lies in
        uint32_t col0;   // chunk-local columns left of the window
        uint32_t cols;   // window width
      };
      std::vector<Win> wins;
      std::vector<int32_t> want(pairs.size(), 0);
      for (size_t pi = 0; pi < pairs.size(); pi++) {
        const size_t k = pairs[pi];
        const uint32_t tl = (uint32_t)(std::lower_bound(tids.begin(), tids.end(), pair_t[k]) - tids.begin());
        int32_t s = 0;
        bool usable = true;
        for (uint32_t c = first_chunk[tl]; c < first_chunk[tl + 1]; c++) {
          if (slot_of[c] == UINT32_MAX || pp.slot[slot_of[c]] >= 32767) usable = false;  // (cannot happen: Q * match < 32767)
          else s = std::max(s, pp.slot[slot_of[c]]);
        }
        if (!usable) continue;
        want[pi] = s;
        if (s <= 0) {  // no hit: nothing to trace
          po[k] = PairOut{};
          handled[k] = 1;
          st.pairs++;
          continue;
        }
        for (uint32_t c = first_chunk[tl]; c < first_chunk[tl + 1]; c++) {
          const uint32_t slot = slot_of[c];
          if (pp.slot[slot] != s) continue;
          const Chunk& C = chunks[c];
          const ScoreGroup& G = groups[slot / kGroupTargets];
          const uint32_t lane = slot % 64, shift = 16 * ((slot % kGroupTargets) / 64);
          const uint32_t nst = (C.len + kScoreR - 1) / kScoreR;
          uint32_t last_unit = UINT32_MAX;
          for (uint32_t st_ = C.own_lo / kScoreR; st_ < nst; st_++) {  // strips that reach into the own region
            const uint32_t e = G.strip_base + st_;
            if ((int32_t)(int16_t)((pp.strip_best[(size_t)e * 64 + lane] >> shift) & 0xffffu) != s) continue;
            const uint32_t u = unit_of_elem[e];
            uint32_t entry = u;  // the unit at whose left boundary the window is entered
            // a maximum within a few strips of its unit's left boundary: the alignment very likely crosses it, and a
            // continuation round costs a whole submission -- enter one unit earlier right away
            if (e - std::max(units[u].first_elem, G.strip_base) < kNearStrips && units[u].first_elem > G.strip_base &&
                units[u].dep != UINT32_MAX)
              entry = units[u].dep;
            if (wide_first) {  // (in strips, not units: a unit is as many strips as the pass has waves per workgroup)
              const uint32_t need = std::min<uint32_t>((Q + Q / 2) / kScoreR + 1, 64 * 16);
              for (uint32_t have = 0; have < need && units[entry].first_elem > G.strip_base && units[entry].dep != UINT32_MAX;) {
                entry = units[entry].dep;
                have += units[entry].count;
              }
            }
            const uint32_t first = std::max(units[entry].first_elem, G.strip_base);
            const uint32_t col0 = (first - G.strip_base) * kScoreR;
            const uint32_t cols = std::min(C.len, (st_ + 1) * kScoreR) - col0;
            if (u == last_unit) wins.back().cols = C.len < (st_ + 1) * kScoreR ? C.len - wins.back().col0 : (st_ + 1) * kScoreR - wins.back().col0;  // a later strip of the same unit: widen
            else wins.push_back(Win{pi, c, entry, col0, cols});
            last_unit = u;
          }
        }
      }
      // ---- rounds: fill + walk the windows; a path that leaves a window on its left continues in the next round
      struct Trace {
        uint32_t bx = 0, bj = 0;        // the end cell (global, 1-based)
        uint32_t res_x = 0;             // where an unfinished traceback resumes: row, and its H there
        int32_t res_H = 0;
        uint64_t key = ~0ull;
        std::vector<uint8_t> ops;
        bool done = false, failed = false;
      };
      std::vector<Trace> tr(pairs.size());
      bool first_round = true;
      uint32_t round_no = 0;
      while (!wins.empty()) {
        std::vector<PairSpec> specs(wins.size());
        for (size_t w = 0; w < wins.size(); w++) {
          const Win& W = wins[w];
          const Chunk& C = chunks[W.chunk];
          const Decoded& t = dec[C.tid];
          const uint32_t slot = slot_of[W.chunk];
          const ScoreGroup& G = groups[slot / kGroupTargets];
          PairSpec& sp = specs[w];
          sp.q = q.sym.data();
          sp.Q = first_round ? Q : tr[W.pair].res_x;  // (a continuation needs the rows above the resume point only)
          sp.t = t.sym.data() + C.start + W.col0;
          sp.T = W.cols;
          sp.path = SW_PATH_LOW_MEMORY;
          sp.col0 = C.start + W.col0;  // global: the discovery-order key and the column limits count from the target's start
          sp.Qfull = Q;
          sp.Tfull = (uint32_t)t.sym.size();
          if (W.col0 > 0) {
            if (units[W.unit].dep == UINT32_MAX || std::max(units[W.unit].first_elem, G.strip_base) != units[W.unit].first_elem)
              return fail(SW_E_HIP, "internal: window without a boundary column");
            sp.left = pp.d_ring + (size_t)units[W.unit].dep * pp.ring_stride + slot % 64;
            sp.left_shift = 16 * ((slot % kGroupTargets) / 64);
          }
          if (!first_round) {
            sp.walk_x = tr[W.pair].res_x;
            sp.walk_j = W.cols;
            sp.walk_H = tr[W.pair].res_H;
          }
        }
        std::vector<PairOut> outs(wins.size());
        ArenaLease ops_lease;
        OpsArena& ops_arena = ops_lease.a;
        if ((rc = run_exact(*L.c, L.c->ws, specs.data(), specs.size(), sc, true, outs.data(), &ops_arena))) return rc;
        st.windows += wins.size();
        for (size_t w = 0; w < wins.size(); w++) st.window_cells += (uint64_t)specs[w].Q * specs[w].T;
        // choose, per pair, the window whose best cell comes first in the reference's discovery order
        std::vector<size_t> chosen(pairs.size(), SIZE_MAX);
        for (size_t w = 0; w < wins.size(); w++) {
          const size_t pi = wins[w].pair;
          if (first_round) {
            if (outs[w].score != want[pi]) continue;  // this window's own maximum is lower: the strip's lay in other rows
            // a maximal cell in the overlap part of a chunk belongs to the chunk before (which reports it too)
            if (outs[w].bj + wins[w].col0 <= chunks[wins[w].chunk].own_lo) continue;
            if (chosen[pi] == SIZE_MAX || outs[w].key < outs[chosen[pi]].key) chosen[pi] = w;
          } else {
            chosen[pi] = w;
          }
        }
        std::vector<Win> next;
        for (size_t pi = 0; pi < pairs.size(); pi++) {
          const size_t w = chosen[pi];
          Trace& T_ = tr[pi];
          if (w == SIZE_MAX) {
            if (first_round && want[pi] > 0 && !handled[pairs[pi]]) T_.failed = true;
            continue;
          }
          const PairOut& o = outs[w];
          if (first_round) {
            T_.bx = o.bx;
            T_.bj = o.bj + wins[w].col0 + chunks[wins[w].chunk].start;
            T_.key = o.key;
          }
          T_.ops.insert(T_.ops.end(), ops_arena.begin() + (ptrdiff_t)o.ops_off, ops_arena.begin() + (ptrdiff_t)(o.ops_off + o.nops));
          if (o.end_H > 0 && o.end_j == 0 && o.end_x >= 1 && (wins[w].col0 > 0 || chunks[wins[w].chunk].start > 0)) {
            if (wins[w].col0 == 0) {  // the path would leave its chunk: impossible by the overlap bound -- do not guess
              T_.failed = true;
              st.chunk_escapes++;
              continue;
            }
            // the path crossed the window's left edge: resume at (end_x, last column of the unit to the left).  The
            // next window spans 2, 4, 8, ... units (every unit boundary has its column stored, so any unit can be
            // the entry): a long alignment costs a logarithmic number of rounds, a short one few cells.
            const ScoreGroup& G = groups[slot_of[wins[w].chunk] / kGroupTargets];
            // ... and never fewer than the rows still above the resume point suggest: with the usual scorings a
            // path that has left its first window runs on diagonally to the top of the matrix (x rows left ~ 1.5 x
            // columns at most), so the second submission is normally the last.
            uint32_t up = units[wins[w].unit].dep;
            const uint32_t by_rows = (o.end_x + o.end_x / 2) / kScoreR + 1;  // strips
            const uint32_t need = std::max<uint32_t>(((1u << std::min<uint32_t>(round_no + 1, 4)) - 1) * 16, std::min<uint32_t>(by_rows, 64 * 16));
            for (uint32_t have = units[up].count; have < need && units[up].first_elem > G.strip_base && units[up].dep != UINT32_MAX;) {
              up = units[up].dep;
              have += units[up].count;
            }
            const uint32_t first = std::max(units[up].first_elem, G.strip_base);
            const uint32_t col0 = (first - G.strip_base) * kScoreR;
            next.push_back(Win{pi, wins[w].chunk, up, col0, wins[w].col0 - col0});
            T_.done = false;
            T_.res_H = o.end_H;
            T_.res_x = o.end_x;
          } else {
            T_.done = true;
          }
        }
        if (tm.on)
          fprintf(stderr, "[sw_hip]   windows: round %u: %zu windows (first: %u rows x %u columns%s), %zu continue\n", round_no,
                  wins.size(), specs[0].Q, specs[0].T, wide_first ? ", wide" : "", next.size());
        if (first_round && !no_wide && 2 * next.size() >= wins.size() && wins.size() >= 8) wide_first = true;
        if (first_round && !no_wide && !wide_first)
          g_wide_hint.store(hint_key << 1 | (2 * next.size() >= wins.size() ? 1 : 0), std::memory_order_relaxed);
        wins.swap(next);
        first_round = false;
        round_no++;
        st.rounds++;
        tm.lap("  windows: round");
      }
      for (size_t pi = 0; pi < pairs.size(); pi++) {
        const size_t k = pairs[pi];
        Trace& T_ = tr[pi];
        if (handled[k] || !T_.done || T_.failed) continue;
        PairOut o;
        o.score = want[pi];
        o.bx = T_.bx;
        o.bj = T_.bj;
        o.nops = (uint32_t)T_.ops.size();
        {
          std::lock_guard<std::mutex> lk(arena_mu);
          o.ops_off = arena.size();
          arena.insert(arena.end(), T_.ops.begin(), T_.ops.end());
        }
        po[k] = o;
        handled[k] = 1;
        st.pairs++;
      }
      if ((rc = start(cur))) return rc;  // the slot is free: the next query's pass goes behind whatever is queued
    }
    return SW_OK;
  };
  // A combined batch of per-pair callers (15 goroutines, each with a query and a target of its own:
  // worker_impl.go:204-238) is as many sets of ONE target with ONE query: every job is ~2 ms of pure latency -- a
  // database build, one score pass that is a single chain of strips, a few window submissions -- with the GPU nearly
  // idle.  Such jobs run side by side, each on a context lane of its own (stream + workspaces + temporary database);
  // a set that fills the GPU by itself (a package's queries x hundreds of targets) runs alone on the device's
  // context, as before.
  size_t small_jobs = 0;
  for (const SetJob& job : jobs) {
    uint64_t residues = 0;
    for (uint32_t t : job.tids) residues += dec[t].sym.size();
    small_jobs += residues <= kSmallSetResidues;
  }
  const unsigned lanes = small_jobs == jobs.size() ? (unsigned)std::min<size_t>(max_lanes, jobs.size()) : 1;
  if (lanes <= 1) {
    for (SetJob& job : jobs)
      if ((rc = run_job(job, ctx, st, tm0))) return rc;
    g_package_stats = st;
    return SW_OK;
  }
  std::atomic<size_t> next_job{0};
  const auto t_call = std::chrono::steady_clock::now();
  std::mutex res_mu;
  int first_rc = SW_OK;
  std::string first_err;
  auto lane_main = [&](unsigned lane) noexcept {
    PackageStats mine{};
    StageTimer tm_lane(getenv("SW_HIP_TRACE_LANES") ? 0 : SIZE_MAX);  // (off: the laps of concurrent jobs interleave)
    int lrc = SW_OK;
    std::string lerr;
    try {
      Context* lctx = nullptr;
      lrc = ctx_get_lane(ctx->device, lane, &lctx);
      while (!lrc) {
        {
          std::lock_guard<std::mutex> lk(res_mu);
          if (first_rc) break;  // another lane failed: the call fails, nothing more to do
        }
        const size_t j = next_job.fetch_add(1);
        if (j >= jobs.size()) break;
        const auto t_a = std::chrono::steady_clock::now();
        lrc = run_job(jobs[j], lctx, mine, tm_lane);
        if (tm0.on) {
          const auto t_b = std::chrono::steady_clock::now();
          fprintf(stderr, "[sw_hip]   windows: lane %u job %zu: %.3f .. %.3f ms\n", lane, j,
                  std::chrono::duration<double, std::milli>(t_a - t_call).count(),
                  std::chrono::duration<double, std::milli>(t_b - t_call).count());
        }
      }
      if (lrc) lerr = last_error_string();
    } catch (...) {
      lrc = SW_E_NOMEM;
      lerr = "out of host memory";
    }
    std::lock_guard<std::mutex> lk(res_mu);
    if (lrc && !first_rc) first_rc = lrc, first_err.swap(lerr);
    st.pairs += mine.pairs, st.queries += mine.queries, st.windows += mine.windows, st.rounds += mine.rounds;
    st.window_cells += mine.window_cells, st.chunks += mine.chunks, st.chunk_escapes += mine.chunk_escapes;
  };
  {
    std::vector<std::thread> helpers;
    for (unsigned lane = 1; lane < lanes; lane++) {
      try {
        helpers.emplace_back(lane_main, lane);
      } catch (...) {
        break;  // no thread to be had: the lanes that exist take the jobs
      }
    }
    lane_main(0);
    for (std::thread& t : helpers) t.join();
  }
  if (first_rc) return fail(first_rc, first_err);
  g_package_stats = st;
  return SW_OK;
}

extern "C" void sw_hip_package_stats(uint64_t out[7]) {
  if (!out) return;
  const PackageStats s = g_package_stats;
  out[0] = s.pairs;
  out[1] = s.queries;
  out[2] = s.windows;
  out[3] = s.rounds;
  out[4] = s.window_cells;
  out[5] = s.chunks;
  out[6] = s.chunk_escapes;
}
