// dlsa_amd/csrc/sw_worker.cpp -- the worker-side callers of the hot path (include/sw_worker.h): ExecuteWork,
// the capacity benchmark, FASTA input and the CLI's result files.  Host code over the C ABI of include/sw.h
// and include/sw_hip.h; nothing here computes an alignment.
//
// Mirrors internal/worker/worker_impl.go:108-202 (ExecuteWork + findAlignmentWithFallback),
// internal/worker/worker_rust_benchmark.go:9-68 (runOnce, BenchmarkRust), cli/__main__.py:19-34 (parse_fasta)
// and cli/__main__.py:175-194 (result files).
#include <sys/stat.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "sw_host.h"

using swhip::set_last_error;

namespace {

char* dup_cstr(const char* s) {
  const size_t n = strlen(s);
  char* p = (char*)malloc(n + 1);
  if (p) memcpy(p, s, n + 1);
  return p;
}

// convertResultToGoResult (worker_rust_bridge.go:131-139) + the packing of worker_impl.go:136-155
// The detail takes the Result over: its strings are used in place (a package returns 10^5 of them; copying each
// twice and freeing the originals was a third of the call).
bool fill_detail(sw_alignment_detail& d, Result* r, int entry) {
  d.query_alignment = r->query_ptr;
  d.target_alignment = r->target_ptr;
  d.owner_ = r;
  if (!d.query_alignment || !d.target_alignment) return false;
  d.length = (int64_t)strlen(d.query_alignment);  // Go: len(qRes), bytes
  d.score = (int64_t)r->score;                    // int(uint16)
  d.max_x = (int64_t)r->max_x;                    // int(uint64): 2^64-1 -> -1
  d.max_y = (int64_t)r->max_y;
  d.score32 = sw_result_score32(r);
  d.entry = entry;
  return true;
}

}  // namespace

extern "C" {

static int execute_work(const sw_node* node, const sw_work_package* work, sw_alignment_detail* out,
                        bool& out_initialised) {
  if (!work || (work->n_pairs && (!work->pairs || !out)) || (work->n_sequences && !work->sequences)) {
    set_last_error("sw_execute_work: null argument");
    return SW_E_ARG;
  }
  const uint64_t n = work->n_pairs;
  // (on the host threads: a caller's fresh array of 10^5 records is megabytes of first-touch pages)
  std::atomic<uint64_t> first_bad{UINT64_MAX};
  swhip::parallel_ranges(n, n * 96, [&](size_t lo, size_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      // worker_impl.go:116-122: both sequences must be in the package (the reference aborts the process)
      if (work->pairs[i].query >= work->n_sequences || work->pairs[i].target >= work->n_sequences ||
          !work->sequences[work->pairs[i].query] || !work->sequences[work->pairs[i].target]) {
        uint64_t seen = first_bad.load();
        while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {
        }
        return;
      }
      out[i] = sw_alignment_detail{};
      out[i].query = work->pairs[i].query;
      out[i].target = work->pairs[i].target;
      out[i].entry = -1;
    }
  });
  if (first_bad.load() != UINT64_MAX) {
    set_last_error("sw_execute_work: target and query not found in work package (pair " + std::to_string(first_bad.load()) + ")");
    return SW_E_ARG;
  }
  // worker_impl.go:126-127: {MatchScore, -MismatchPenalty, -GapPenalty}; worker_rust_bridge.go:108-112: C.ushort(...)
  out_initialised = true;
  AlignmentScores sc;
  sc.match = (uint16_t)work->match_score;
  sc.miss = (uint16_t)(-work->mismatch_penalty);
  sc.gap = (uint16_t)(-work->gap_penalty);
  // findAlignmentWithFallback (worker_impl.go:170-202), one GPU submission per level instead of one call per pair
  std::vector<uint64_t> todo(n);
  for (uint64_t i = 0; i < n; i++) todo[i] = i;
  const int chain[3] = {SW_PATH_LOW_MEMORY, SW_PATH_SIMD, SW_PATH_STRAIGHT};
  for (int level = 0; level < 3 && !todo.empty(); level++) {
    std::vector<Result*> res(todo.size(), nullptr);
    int rc;
    if (node) {
      std::vector<const char*> qs(todo.size()), ts(todo.size());
      for (size_t k = 0; k < todo.size(); k++) {
        qs[k] = work->sequences[work->pairs[todo[k]].query];
        ts[k] = work->sequences[work->pairs[todo[k]].target];
      }
      rc = sw_node_align_batch(node, qs.data(), ts.data(), todo.size(), sc, chain[level], res.data());
    } else {
      // the package already names its sequences by index: nothing to re-discover from pointers
      std::vector<uint32_t> qi(todo.size()), ti(todo.size());
      for (size_t k = 0; k < todo.size(); k++) qi[k] = work->pairs[todo[k]].query, ti[k] = work->pairs[todo[k]].target;
      rc = swhip::align_interned(-1, work->sequences, work->n_sequences, qi.data(), ti.data(), todo.size(), sc,
                                 chain[level], res.data());
      if (rc != SW_OK)  // nothing is handed over on failure
        for (Result*& r : res) {
          free_alignment_result(r);
          r = nullptr;
        }
    }
    if (rc != SW_OK) {
      sw_alignment_details_free(out, n);
      return rc;  // no GPU / HIP failure: nothing can be computed (message already recorded)
    }
    std::vector<uint64_t> failed;
    std::atomic<bool> nomem_flag{false};
    // (the detail's Length is strlen of a string of the pair's size: 10^5 of them are worth the host threads)
    swhip::parallel_ranges(todo.size(), todo.size() * 512, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++)
        if (res[k] && !fill_detail(out[todo[k]], res[k], chain[level])) nomem_flag.store(true);
    });
    const bool nomem = nomem_flag.load();
    for (size_t k = 0; k < todo.size(); k++) {
      if (!res[k]) failed.push_back(todo[k]);
      else if (nomem && !out[todo[k]].owner_) free_alignment_result(res[k]);
    }
    if (nomem) {
      sw_alignment_details_free(out, n);
      set_last_error("sw_execute_work: out of memory");
      return SW_E_NOMEM;
    }
    todo.swap(failed);
  }
  for (uint64_t i : todo) {  // "All alignment functions failed to align a sequence" (worker_impl.go:200)
    out[i].query_alignment = dup_cstr("");
    out[i].target_alignment = dup_cstr("");
  }
  // every pair has a record; the count of pairs nothing could align is the (positive) return value
  return SW_W_FAILED_PAIRS + (int)std::min<size_t>(todo.size(), 0x7fffffff);
}

static int execute_work_guarded(const sw_node* node, const sw_work_package* work, sw_alignment_detail* out) {
  bool out_initialised = false, returned = false;
  const int rc = swhip::guarded<int>(SW_E_NOMEM, [&] {
    const int r = execute_work(node, work, out, out_initialised);
    returned = true;
    return r;
  });
  if (!returned && out_initialised) sw_alignment_details_free(out, work->n_pairs);  // left by an exception
  return rc;
}

int sw_execute_work(const sw_work_package* work, sw_alignment_detail* out) {
  return execute_work_guarded(nullptr, work, out);
}

int sw_node_execute_work(const sw_node* node, const sw_work_package* work, sw_alignment_detail* out) {
  if (!node) {
    set_last_error("sw_node_execute_work: null node");
    return SW_E_ARG;
  }
  return execute_work_guarded(node, work, out);
}

void sw_alignment_details_free(sw_alignment_detail* d, uint64_t n) {
  if (!d) return;
  // The strings are the Results': freed in runs (one reference-count update per run and slab), on the host threads
  // when there are 10^5 of them (every result touches three cache lines of a slab of tens of megabytes).
  auto free_range = [d](size_t lo, size_t hi) {
    Result* run[256];
    size_t nrun = 0;
    for (size_t i = lo; i < hi; i++) {
      if (d[i].owner_) {
        run[nrun++] = (Result*)d[i].owner_;
        if (nrun == 256) {
          sw_free_alignment_results(run, nrun);
          nrun = 0;
        }
      } else {
        free(d[i].query_alignment);
        free(d[i].target_alignment);
      }
      d[i].query_alignment = d[i].target_alignment = nullptr;
      d[i].owner_ = nullptr;
    }
    sw_free_alignment_results(run, nrun);
  };
  try {
    swhip::parallel_ranges((size_t)n, (size_t)n * 256, free_range);
  } catch (...) {  // (no threads to be had and the range already partly done: entries already freed are NULL)
    free_range(0, (size_t)n);
  }
}

// ---- capacity benchmark ---------------------------------------------------------------------------------------
namespace {
// runOnce (worker_rust_benchmark.go:9-36)
bool run_once(size_t nQ, size_t nT, double& elapsed_s, float& cups) {
  std::string query = "TTTTCA" + std::string(nQ, 'A');
  std::string target = "TTTTCA" + std::string(nT, 'T');
  AlignmentScores sc;
  sc.match = 1;
  sc.miss = (uint16_t)-1;
  sc.gap = (uint16_t)-2;
  const auto t0 = std::chrono::steady_clock::now();
  Result* r = find_alignment_low_memory(&query[0], &target[0], sc);
  if (!r) r = find_alignment_simd(&query[0], &target[0], sc);
  if (!r) r = find_alignment_sequential_straight(&query[0], &target[0], sc);
  const auto t1 = std::chrono::steady_clock::now();
  if (!r) return false;
  free_alignment_result(r);
  const double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
  elapsed_s = ns * 1e-9;
  cups = (float)(nQ * nT) / (float)ns * 1e9f;  // float32 arithmetic as worker_rust_benchmark.go:31
  return true;
}
}  // namespace

float sw_benchmark_rust(double threshold_seconds, int q_steps, int t_steps) {
  return swhip::guarded<float>(0.0f, [&]() -> float {
  if (q_steps < 1 || t_steps < 1 || q_steps + t_steps > 20) {
    set_last_error("sw_benchmark_rust: bad step counts");
    return 0.0f;
  }
  size_t nQ = (size_t)1 << 12, nT = (size_t)1 << 18;
  double el = 0;
  float cups = 0;
  if (!run_once(64, 256, el, cups)) return 0.0f;  // device + code object warm-up, not part of the protocol
  for (;;) {
    // every size twice, the second run counts: the library sizes its scratch buffers (device ring, pinned staging,
    // host buffers) to the largest submission seen, and a worker that has just started would otherwise stop
    // doubling at the first size whose ALLOCATION takes 10 ms and register a latency-bound figure (first call of
    // a process 167 GCUPS, every later one 770)
    if (!run_once(nQ, nT, el, cups)) return 0.0f;
    if (!run_once(nQ, nT, el, cups)) return 0.0f;
    if (el > threshold_seconds) break;
    if (nT >= ((size_t)1 << 27)) break;  // bound the scratch memory of one pair (the reference has no bound)
    nT *= 2;
  }
  nT /= (size_t)1 << (q_steps + t_steps - 2);
  if (nT == 0) nT = 1;
  float sum = 0;
  for (int iT = 0; iT < t_steps; iT++)
    for (int iQ = 0; iQ < q_steps; iQ++) {
      if (!run_once(nQ << iQ, nT << iT, el, cups)) return 0.0f;  // (untimed, as above)
      if (!run_once(nQ << iQ, nT << iT, el, cups)) return 0.0f;
      sum += cups;
    }
  return sum / (float)(q_steps * t_steps);
  });
}

}  // extern "C"

// ---- FASTA ---------------------------------------------------------------------------------------------------
struct sw_fasta {
  std::vector<std::string> ids, seqs;
};

namespace {
// Python's str.strip() / str.split() whitespace for the ASCII range (the file is read as text)
inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
}  // namespace

extern "C" {

sw_fasta* sw_fasta_parse(const char* path) {
  return swhip::guarded<sw_fasta*>(nullptr, [&]() -> sw_fasta* {
  if (!path) {
    set_last_error("sw_fasta_parse: null path");
    return nullptr;
  }
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    set_last_error(std::string("sw_fasta_parse: cannot open ") + path + ": " + strerror(errno));
    return nullptr;
  }
  std::string raw;
  {
    struct Closer {
      FILE* f;
      ~Closer() { fclose(f); }
    } closer{fp};
    std::vector<char> buf(1 << 16);
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) raw.append(buf.data(), got);
    if (ferror(fp)) {
      set_last_error(std::string("sw_fasta_parse: read error on ") + path);
      return nullptr;
    }
  }
  // open(..., "r"): universal newlines -- "\r\n" and lone "\r" become "\n"
  std::string data;
  data.reserve(raw.size());
  for (size_t i = 0; i < raw.size(); i++) {
    if (raw[i] == '\r') {
      data.push_back('\n');
      if (i + 1 < raw.size() && raw[i + 1] == '\n') i++;
    } else {
      data.push_back(raw[i]);
    }
  }
  std::unique_ptr<sw_fasta> f(new sw_fasta);
  // fasta_data.split(">")[1:]
  size_t pos = data.find('>');
  while (pos != std::string::npos) {
    const size_t next = data.find('>', pos + 1);
    size_t b = pos + 1, e = next == std::string::npos ? data.size() : next;
    while (b < e && py_space((unsigned char)data[b])) b++;       // entry.strip()
    while (e > b && py_space((unsigned char)data[e - 1])) e--;
    // lines = entry.split("\n"); seq_id = lines[0].split()[0]; seq_data = "".join(lines[1:])
    size_t eol = data.find('\n', b);
    if (eol == std::string::npos || eol > e) eol = e;
    size_t ib = b;
    while (ib < eol && py_space((unsigned char)data[ib])) ib++;
    size_t ie = ib;
    while (ie < eol && !py_space((unsigned char)data[ie])) ie++;
    if (ie == ib) {  // lines[0].split() is empty: IndexError in the reference
      set_last_error("sw_fasta_parse: record without an id in " + std::string(path));
      return nullptr;
    }
    std::string seq;
    for (size_t i = eol; i < e; i++)
      if (data[i] != '\n') seq.push_back(data[i]);
    if (seq.find('\0') != std::string::npos) {
      set_last_error("sw_fasta_parse: NUL byte inside a sequence");
      return nullptr;
    }
    f->ids.emplace_back(data, ib, ie - ib);
    f->seqs.push_back(std::move(seq));
    pos = next;
  }
  return f.release();
  });
}

uint64_t sw_fasta_count(const sw_fasta* f) { return f ? f->ids.size() : 0; }
const char* sw_fasta_id(const sw_fasta* f, uint64_t i) { return (f && i < f->ids.size()) ? f->ids[i].c_str() : nullptr; }
const char* sw_fasta_sequence(const sw_fasta* f, uint64_t i) {
  return (f && i < f->seqs.size()) ? f->seqs[i].c_str() : nullptr;
}
void sw_fasta_free(sw_fasta* f) { delete f; }

sw_db* sw_db_from_fasta(const sw_fasta* f) {
  return swhip::guarded<sw_db*>(nullptr, [&]() -> sw_db* {
  if (!f) {
    set_last_error("sw_db_from_fasta: null argument");
    return nullptr;
  }
  std::string flat;
  std::vector<uint64_t> off(f->seqs.size() + 1, 0);
  for (size_t i = 0; i < f->seqs.size(); i++) {
    flat += f->seqs[i];
    off[i + 1] = flat.size();
  }
  return sw_db_create(flat.data(), off.data(), f->seqs.size());
  });
}

int sw_write_query_results(const char* results_dir, const char* query_id, const char* const* target_ids,
                           const sw_hit* hits, int n_hits) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
  if (!results_dir || !query_id || n_hits < 0 || (n_hits && (!target_ids || !hits))) {
    set_last_error("sw_write_query_results: bad argument");
    return SW_E_ARG;
  }
  // os.makedirs(results_dir, exist_ok=True)
  std::string dir(results_dir);
  for (size_t i = 1; i <= dir.size(); i++)
    if (i == dir.size() || dir[i] == '/') {
      const std::string part = dir.substr(0, i);
      if (!part.empty() && mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) {
        set_last_error("sw_write_query_results: cannot create " + part + ": " + strerror(errno));
        return SW_E_ARG;
      }
    }
  std::string path = dir;
  if (!path.empty() && path.back() != '/') path.push_back('/');
  path += std::string(query_id) + ".txt";
  FILE* fp = fopen(path.c_str(), "ab");  // "a" if the file exists, else "w": the same thing
  if (!fp) {
    set_last_error("sw_write_query_results: cannot open " + path + ": " + strerror(errno));
    return SW_E_ARG;
  }
  for (int i = 0; i < n_hits; i++)
    fprintf(fp, ">%s\nAlignment: %s\nLength: %u\nScore: %d\n\n", target_ids[i] ? target_ids[i] : "",
            hits[i].query_aln ? hits[i].query_aln : "", hits[i].length, hits[i].score);
  const bool bad = ferror(fp) != 0;
  if (fclose(fp) != 0 || bad) {
    set_last_error("sw_write_query_results: write error on " + path + ": " + strerror(errno));
    return SW_E_ARG;
  }
  return SW_OK;
  });
}

}  // extern "C"

// ---- synthetic sequences (utils/generate_synthetic_data.py:7-12 with a fixed PRNG) ---------------------------------
namespace {
struct SynthRng {  // xoshiro256** seeded by splitmix64 of (seed, sequence index)
  uint64_t s[4];
  static uint64_t splitmix(uint64_t& x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  SynthRng(uint64_t seed, uint64_t index) {
    uint64_t x = seed ^ (index * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull);
    for (auto& v : s) v = splitmix(x);
  }
  static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
};
// random.randint(min_size, max_size): the first draw of the sequence's generator
inline uint32_t synth_length(SynthRng& g, uint32_t min_len, uint32_t max_len) {
  const uint64_t range = (uint64_t)max_len - min_len + 1;
  return min_len + (uint32_t)(((g.next() >> 32) * range) >> 32);
}
// random.choices(["A", "T", "C", "G"], k=seq_length): 32 residues per 64-bit draw
inline void synth_residues(SynthRng& g, char* dst, uint32_t len) {
  static const char kAlpha[4] = {'A', 'T', 'C', 'G'};
  uint32_t i = 0;
  while (i < len) {
    uint64_t v = g.next();
    for (int k = 0; k < 32 && i < len; k++, v >>= 2) dst[i++] = kAlpha[v & 3];
  }
}
bool synth_args_ok(uint32_t min_len, uint32_t max_len) { return min_len <= max_len && max_len < 0x7fffff00u; }
}  // namespace

extern "C" {

int sw_synth_lengths(uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len, uint32_t max_len, uint32_t* lengths) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
    if ((!lengths && n) || !synth_args_ok(min_len, max_len)) return swhip::fail(SW_E_ARG, "sw_synth_lengths: bad argument");
    swhip::parallel_ranges(n, n * 64, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        SynthRng g(seed, first + k);
        lengths[k] = synth_length(g, min_len, max_len);
      }
    });
    return SW_OK;
  });
}

int sw_synth_fill(uint64_t seed, const uint64_t* indices, uint64_t n, uint32_t min_len, uint32_t max_len, char* flat,
                  const uint64_t* offsets) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
    if ((!flat && n) || !offsets || !synth_args_ok(min_len, max_len)) return swhip::fail(SW_E_ARG, "sw_synth_fill: bad argument");
    std::atomic<bool> bad{false};
    swhip::parallel_ranges(n, n ? (size_t)(offsets[n] - offsets[0]) + n * 64 : 0, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        SynthRng g(seed, indices ? indices[k] : k);
        const uint32_t len = synth_length(g, min_len, max_len);
        if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] != len) {
          bad = true;
          return;
        }
        synth_residues(g, flat + offsets[k], len);
      }
    });
    if (bad) return swhip::fail(SW_E_ARG, "sw_synth_fill: offsets do not match the generated lengths");
    return SW_OK;
  });
}

int sw_synth_write_fasta(const char* path, uint64_t seed, uint64_t n, uint32_t min_len, uint32_t max_len) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
    if (!path || !synth_args_ok(min_len, max_len)) return swhip::fail(SW_E_ARG, "sw_synth_write_fasta: bad argument");
    FILE* fp = fopen(path, "wb");
    if (!fp) return swhip::fail(SW_E_ARG, std::string("sw_synth_write_fasta: cannot open ") + path + ": " + strerror(errno));
    std::string seq;
    for (uint64_t i = 0; i < n; i++) {
      SynthRng g(seed, i);
      const uint32_t len = synth_length(g, min_len, max_len);
      seq.resize(len);
      synth_residues(g, &seq[0], len);
      fprintf(fp, ">seq%llu\n", (unsigned long long)(i + 1));
      fwrite(seq.data(), 1, seq.size(), fp);
      fputc('\n', fp);
    }
    const bool bad = ferror(fp) != 0;
    if (fclose(fp) != 0 || bad) return swhip::fail(SW_E_ARG, std::string("sw_synth_write_fasta: write error on ") + path);
    return SW_OK;
  });
}

}  // extern "C"
