// dlsa_amd/csrc/sw_lib.cpp -- host side of libsw_hip.so: the C ABI of include/sw.h and
// include/sw_hip.h over the HIP kernels in sw_exact.hip / sw_score.hip.
//
// Mirrors, on the callee side of the cgo boundary, what rust/src/bindings.rs does around the
// algorithm: C string -> UTF-8 check -> symbols (bindings.rs:122-129), run, box a Result with two
// heap strings (bindings.rs:143-151), NULL on any failure (bindings.rs:153-156).  There is no CPU
// implementation of the alignment in this file: every cell is computed by a HIP kernel and a missing
// GPU makes every entry point fail loudly (NULL / SW_E_NOGPU + sw_hip_last_error()).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/sw_worker.h"
#include "sw_internal.h"

using namespace swhip;

namespace {

// ---- errors --------------------------------------------------------------------------------------------
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace
void swhip::set_last_error(const std::string& msg) { g_err = msg; }
namespace {
#define HIPCK(expr)                                                                                  \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return fail(SW_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                     \
  } while (0)

// ---- device context --------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return SW_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      e = hipMalloc(&p, bytes);
      want = bytes;
    }
    if (e != hipSuccess) return fail(SW_E_NOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    cap = want;
    return SW_OK;
  }
};

// pinned host staging memory (async copies from pageable memory are staged by the runtime, synchronously)
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return SW_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess)
      return fail(SW_E_NOMEM, std::string("hipHostMalloc(") + std::to_string(want) + ") failed");
    cap = want;
    return SW_OK;
  }
};

struct Context {
  std::mutex mu;  // one GPU submission at a time; the ABI is entered from many threads
  bool ready = false;
  int device = 0;
  int cus = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  DevBuf xin, xzero, xout, dirs, xrecs;                // exact path scratch (run_exact_chunk)
  HostBuf h_in, h_out;                                 // ... and its pinned staging buffers
  DevBuf sel_hist, sel_ctr, sel_rerun, sel_cand, sel_tie;  // device-side top-k selection
  DevBuf qwords, slot_scores, gbound, ticket, prog;    // score path scratch
  sw_kernel_stats stats{};
  size_t scratch_budget = (size_t)8 << 30;
};
Context g_ctx;

int ctx_init_locked(int device) {
  Context& c = g_ctx;
  if (c.ready) {
    if (device >= 0 && device != c.device) return fail(SW_E_ARG, "sw_hip_init: already initialised on another device");
    HIPCK(hipSetDevice(c.device));  // HIP's current device is per thread; the ABI is entered from many threads
    return SW_OK;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SW_E_NOGPU, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                                "); libsw_hip has no CPU fallback");
  if (device < 0) device = 0;
  if (device >= n) return fail(SW_E_ARG, "sw_hip_init: device index out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(SW_E_NOGPU, std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only");
  c.cus = prop.multiProcessorCount;
  HIPCK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  HIPCK(hipEventCreate(&c.ev0));
  HIPCK(hipEventCreate(&c.ev1));
  HIPCK(hipEventCreate(&c.ev2));
  HIPCK(hipEventCreate(&c.ev3));
  if (const char* mb = getenv("SW_HIP_SCRATCH_MB")) c.scratch_budget = (size_t)atoll(mb) << 20;
  c.device = device;
  c.ready = true;
  return SW_OK;
}

// ---- UTF-8 -> symbols --------------------------------------------------------------------------------------
// rust/src/bindings.rs:125-129: String::from_utf8(...).unwrap().chars().collect(); invalid UTF-8 panics
// -> NULL.  cps receives the Unicode scalar values (for re-encoding the alignment), sym their u16
// truncation (`*x as u16`, rust/src/algorithm/mod.rs:78,224) which is what scores are computed on.
bool decode_utf8(const char* s, std::vector<uint32_t>& cps) {
  const unsigned char* p = (const unsigned char*)s;
  cps.clear();
  while (*p) {
    uint32_t c = *p;
    if (c < 0x80) {
      cps.push_back(c);
      p++;
      continue;
    }
    int n;
    uint32_t minv;
    if ((c & 0xE0) == 0xC0) { n = 1; c &= 0x1F; minv = 0x80; }
    else if ((c & 0xF0) == 0xE0) { n = 2; c &= 0x0F; minv = 0x800; }
    else if ((c & 0xF8) == 0xF0) { n = 3; c &= 0x07; minv = 0x10000; }
    else return false;
    for (int i = 1; i <= n; i++) {
      if ((p[i] & 0xC0) != 0x80) return false;
      c = (c << 6) | (p[i] & 0x3F);
    }
    if (c < minv || c > 0x10FFFF || (c >= 0xD800 && c <= 0xDFFF)) return false;
    cps.push_back(c);
    p += n + 1;
  }
  return true;
}
void encode_utf8(uint32_t c, std::string& out) {
  if (c < 0x80) out.push_back((char)c);
  else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
  else if (c < 0x10000) { out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
  else { out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
}

inline uint32_t roundup64(uint32_t v) { return (v + 63u) / 64u * 64u; }

// ---- the four integers that make one plain SW reproduce a reference entry point ---------------------------
// DESIGN.md "Effective semantics"; proven equal to the restatement by tests/test_oracle_clean.py.
struct PathParams {
  uint32_t Tlim, T0, phaseQp, straight;
};
PathParams resolve_path(int path, uint32_t Q, uint32_t T) {
  PathParams pp{T, T, 0, 0};
  if (path == SW_PATH_STRAIGHT) {
    pp.straight = 1;
    return pp;
  }
  const uint32_t Qp = roundup64(Q), Tp = roundup64(T);
  const bool full = (path == SW_PATH_SIMD) || ((uint64_t)T < 2ull * Q);  // rust/src/algorithm/mod.rs:213
  if (full) {
    if (T > 0 && T % 64 == 0) {        // algorithm/mod.rs:98,178: column Tp is skipped ...
      pp.Tlim = T - 1;
      pp.T0 = (Tp >= Qp) ? T : T - 1;  // ... except H(1,T) in the lane phase, algorithm/mod.rs:134
    }
  } else if (T < Qp) {                 // algorithm/mod.rs:251 with an empty lane phase (:313)
    pp.T0 = T - 1;
    pp.phaseQp = Qp;
  }
  return pp;
}

// ---- exact path driver ------------------------------------------------------------------------------------
struct PairSpec {
  const uint16_t* q;
  uint32_t Q;
  const uint16_t* t;
  uint32_t T;
  int path;
  // optional: small dense ids of the two sequences within the submission (equal id <=> same sequence); lets the
  // driver upload every distinct sequence once without hashing 10^5 pointers.  UINT32_MAX: look the pointer up.
  uint32_t qid = UINT32_MAX, tid = UINT32_MAX;
};
struct PairOut {
  int32_t score = 0;
  uint32_t bx = 0, bj = 0;
  size_t ops_off = 0;  // traceback ops (last alignment column first) at arena[ops_off .. ops_off + nops)
  uint32_t nops = 0;
};

// SW_HIP_TRACE=1: host stage times of big submissions on stderr (development aid)
struct StageTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(size_t n) : on(n >= 1000 && getenv("SW_HIP_TRACE")), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sw_hip] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// Runs pairs [begin, end) in one submission.  Caller holds the context mutex.
int run_exact_chunk(const PairSpec* specs, size_t begin, size_t end, Scoring sc, bool want_trace, PairOut* outs,
                    std::vector<uint8_t>* arena, int rpl) {
  Context& c = g_ctx;
  const size_t n = end - begin;
  if (n == 0) return SW_OK;
  StageTimer tm(n);
  // symbol pool: every distinct sequence once (a database search shares one query)
  std::unordered_map<const uint16_t*, size_t> seen;
  std::vector<size_t> off_by_id;
  struct SeqCopy {
    const uint16_t* p;
    uint32_t len;
    size_t off;
  };
  std::vector<SeqCopy> copies;  // distinct sequences, copied straight into the pinned staging buffer below
  size_t pool_syms = 0;
  auto add_seq = [&](const uint16_t* p, uint32_t len, uint32_t id) -> size_t {
    if (id != UINT32_MAX) {
      if (id >= off_by_id.size()) off_by_id.resize((size_t)id + 1 + off_by_id.size() / 2, SIZE_MAX);
      if (off_by_id[id] != SIZE_MAX) return off_by_id[id];
    } else {
      auto it = seen.find(p);
      if (it != seen.end()) return it->second;
    }
    const size_t off = pool_syms;
    copies.push_back(SeqCopy{p, len, off});
    pool_syms += (len + 1u) & ~(size_t)1;
    if (id != UINT32_MAX) off_by_id[id] = off;
    else seen.emplace(p, off);
    return off;
  };
  std::vector<ExactPair> hp(n);
  std::vector<size_t> qoff(n), toff(n);
  size_t dir_words = 0, bnd_ints = 0, ops_bytes = 0;
  std::vector<uint32_t> unit_pair;
  unit_pair.reserve(n);
  copies.reserve(n + 1);
  for (size_t i = 0; i < n; i++) {
    const PairSpec& s = specs[begin + i];
    qoff[i] = add_seq(s.q, s.Q, s.qid);
    toff[i] = add_seq(s.t, s.T, s.tid);
    PathParams pp = resolve_path(s.path, s.Q, s.T);
    ExactPair& e = hp[i];
    e.Q = s.Q;
    e.T = s.T;
    e.Tlim = pp.Tlim;
    e.T0 = pp.T0;
    e.phaseQp = pp.phaseQp;
    e.straight = pp.straight;
    const uint32_t maxcol = std::max(pp.Tlim, pp.T0);
    e.bnd_pitch = maxcol + 2;
    e.nkb = exact_nsteps(maxcol) / 16;
    e.n_units = std::max<uint32_t>(exact_nstrips(s.Q, rpl), 1);
    if (unit_pair.size() + e.n_units > 0x7fffff00ull) return fail(SW_E_ARG, "too many strips in one submission");
    e.first_unit = (uint32_t)unit_pair.size();
    unit_pair.insert(unit_pair.end(), e.n_units, (uint32_t)i);
    // temporarily store offsets in the pointer fields
    e.bnd = (int32_t*)(uintptr_t)bnd_ints;
    bnd_ints += (size_t)e.n_units * e.bnd_pitch;
    if (want_trace) {
      e.dirs = (uint32_t*)(uintptr_t)(dir_words + 1);  // +1 so that offset 0 is not nullptr
      dir_words += exact_dir_words(s.Q, maxcol, rpl);
      e.ops = (uint8_t*)(uintptr_t)ops_bytes;
      ops_bytes += ((size_t)s.Q + s.T + 15) / 16 * 16;
    } else {
      e.dirs = nullptr;
      e.ops = nullptr;
    }
  }
  const size_t n_units = unit_pair.size();
  // Ticket order: strip level by level (all pairs' strip 0, then every strip 1, ...; longest pairs first within a
  // level), so that in a big submission the strip above has long finished when a strip starts, instead of the
  // strips of one pair trailing each other through HBM with adjacent tickets.  The strip above always has an
  // EARLIER ticket, which is all the kernel needs.  unit_order[ticket] = record index (first_unit + strip).
  std::vector<uint32_t> unit_order;
  unit_order.reserve(n_units);
  {
    uint32_t max_units = 0;
    for (size_t i = 0; i < n; i++) max_units = std::max(max_units, hp[i].n_units);
    std::vector<uint32_t> by_len(n);
    std::iota(by_len.begin(), by_len.end(), 0u);
    if (max_units > 1)
      std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) { return hp[a].n_units > hp[b].n_units; });
    size_t live = n;  // pairs that still have a strip at this level: a prefix of by_len
    for (uint32_t s = 0; s < max_units; s++) {
      while (live > 0 && hp[by_len[live - 1]].n_units <= s) live--;
      for (size_t k = 0; k < live; k++) unit_order.push_back(hp[by_len[k]].first_unit + s);
    }
  }
  tm.lap("  chunk: tables");
  // One blob each way (a per-pair call is dominated by the number of HIP calls, not by bytes):
  //   xin   (H2D):   [ExactPair x n | unit_pair x n_units | unit_order x n_units | symbols]   staged in pinned host memory
  //   xzero (memset): [control words: [0] ticket, [32] err, [64 ..) per-pair done counters | boundary rows]
  //   xout  (D2H):   [ExactOut x n | traceback ops]
  auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t in_pairs = 0, in_units = align16(n * sizeof(ExactPair)), in_order = in_units + align16(n_units * 4),
               in_sym = in_order + align16(n_units * 4);
  const size_t in_bytes = in_sym + align16(pool_syms * 2 + 16);
  const size_t ctl_words = 64 + n;
  const size_t z_bnd = align16(ctl_words * 4), z_bytes = z_bnd + align16(std::max<size_t>(bnd_ints * 4, 16));
  const size_t out_ops = align16(n * sizeof(ExactOut)), out_bytes = out_ops + (want_trace ? align16(ops_bytes) : 0);
  int rc;
  if ((rc = c.xin.ensure(in_bytes))) return rc;
  if ((rc = c.xzero.ensure(z_bytes))) return rc;
  if ((rc = c.xout.ensure(out_bytes))) return rc;
  if ((rc = c.xrecs.ensure(n_units * sizeof(ExactRec)))) return rc;
  if (want_trace && (rc = c.dirs.ensure(std::max<size_t>(dir_words * 4, 16)))) return rc;
  if ((rc = c.h_in.ensure(in_bytes))) return rc;
  uint8_t* const d_in = (uint8_t*)c.xin.p;
  uint8_t* const d_zero = (uint8_t*)c.xzero.p;
  uint8_t* const d_out = (uint8_t*)c.xout.p;
  for (size_t i = 0; i < n; i++) {
    ExactPair& e = hp[i];
    e.q = (const uint16_t*)(d_in + in_sym) + qoff[i];
    e.t = (const uint16_t*)(d_in + in_sym) + toff[i];
    e.bnd = (int32_t*)(d_zero + z_bnd) + (uintptr_t)e.bnd;
    if (want_trace) {
      e.dirs = (uint32_t*)c.dirs.p + ((uintptr_t)e.dirs - 1);
      e.ops = d_out + out_ops + (uintptr_t)e.ops;
    }
  }
  uint8_t* const h_in = (uint8_t*)c.h_in.p;
  memcpy(h_in + in_pairs, hp.data(), n * sizeof(ExactPair));
  memcpy(h_in + in_units, unit_pair.data(), n_units * 4);
  memcpy(h_in + in_order, unit_order.data(), n_units * 4);
  for (const SeqCopy& sc_ : copies) memcpy(h_in + in_sym + sc_.off * 2, sc_.p, (size_t)sc_.len * 2);
  tm.lap("  chunk: stage");
  HIPCK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, c.stream));
  // (strip boundary rows: a word is "published" when its bit 31 is set (sw_exact.hip), so they start at zero)
  HIPCK(hipMemsetAsync(d_zero, 0, z_bytes, c.stream));
  HIPCK(hipEventRecord(c.ev2, c.stream));
  ExactWork w{};
  w.pairs = (const ExactPair*)(d_in + in_pairs);
  w.outs = (ExactOut*)d_out;
  w.unit_pair = (const uint32_t*)(d_in + in_units);
  w.unit_order = (const uint32_t*)(d_in + in_order);
  w.n_units = (uint32_t)n_units;
  w.n_pairs = (uint32_t)n;
  w.rpl = rpl;
  w.ticket = (uint32_t*)d_zero;
  w.err = (uint32_t*)d_zero + 32;
  w.pair_done = (uint32_t*)d_zero + 64;
  w.recs = (ExactRec*)c.xrecs.p;
  HIPCK(launch_exact(w, sc, c.stream));
  if (want_trace) HIPCK(launch_walk(w.pairs, w.outs, (uint32_t)n, rpl, sc, c.stream));
  HIPCK(hipEventRecord(c.ev3, c.stream));
  // small submissions fetch results and ops in ONE copy (the ops capacity is Q + T per pair); big ones fetch the
  // results first and then only the op bytes the walkers wrote
  const bool one_copy = want_trace && out_bytes <= ((size_t)4 << 20);
  const size_t first_bytes = one_copy ? out_bytes : n * sizeof(ExactOut);
  if ((rc = c.h_out.ensure(first_bytes))) return rc;
  HIPCK(hipMemcpyAsync(c.h_out.p, d_out, first_bytes, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  const ExactOut* ho = (const ExactOut*)c.h_out.p;
  tm.lap("  chunk: gpu");
  for (size_t i = 0; i < n; i++)
    if (ho[i].nops == 0xffffffffu) return fail(SW_E_HIP, "exact kernel: strip pipeline wait timed out");
  float ms = 0;
  HIPCK(hipEventElapsedTime(&ms, c.ev2, c.ev3));
  c.stats.exact_kernel_ms += ms;
  size_t arena_base = 0;
  if (want_trace && ops_bytes) {
    arena_base = arena->size();
    arena->resize(arena_base + ops_bytes);
    uint8_t* hops = arena->data() + arena_base;
    if (one_copy) {
      for (size_t i = 0; i < n; i++) {
        const size_t s0 = (size_t)(hp[i].ops - (d_out + out_ops));
        if (ho[i].nops) memcpy(hops + s0, (const uint8_t*)c.h_out.p + out_ops + s0, ho[i].nops);
      }
    } else {
      size_t lo = 0;
      while (lo < n) {  // coalesce into a few large copies: ops buffers are contiguous per pair
        size_t hi = lo;
        size_t start = (size_t)(hp[lo].ops - (d_out + out_ops));
        size_t endb = start;
        while (hi < n) {
          size_t s0 = (size_t)(hp[hi].ops - (d_out + out_ops));
          if (s0 - endb > (1u << 16)) break;
          endb = s0 + ho[hi].nops;
          hi++;
        }
        if (endb > start) HIPCK(hipMemcpyAsync(hops + start, d_out + out_ops + start, endb - start, hipMemcpyDeviceToHost, c.stream));
        lo = hi;
      }
      HIPCK(hipStreamSynchronize(c.stream));
    }
  }
  for (size_t i = 0; i < n; i++) {
    PairOut& o = outs[begin + i];
    o.score = ho[i].score;
    o.bx = ho[i].bx;
    o.bj = ho[i].bj;
    c.stats.exact_cells += (uint64_t)specs[begin + i].Q * specs[begin + i].T;
    if (want_trace && ho[i].nops) {
      o.ops_off = arena_base + (size_t)(hp[i].ops - (d_out + out_ops));
      o.nops = ho[i].nops;
    }
  }
  tm.lap("  chunk: results");
  return SW_OK;
}

// Rows per lane for a submission.  4 does the least work per cell and is used whenever the submission fills the
// GPU anyway (>= one wave per SIMD).  For a few long pairs the strips of a pair form a pipeline in which strip
// s + 1 trails strip s by a fixed lag, so the time is about steps * step time + strips * lag; smaller strips
// shorten the step (fewer rows per lane) at the price of more strips.  Step times and lags (ns) as measured on
// MI355X with T = 100 000.
int exact_choose_rpl(const PairSpec* specs, size_t n, int simds) {
  static const int forced = [] {
    const char* e = getenv("SW_HIP_EXACT_RPL");
    const int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2 || v == 4) ? v : 0;
  }();
  if (forced) return forced;
  const int rpls[3] = {4, 2, 1};
  const double tstep[3] = {154, 104, 78}, lag[3] = {16800, 11000, 9400};
  int best = 4;
  double best_t = 0;
  for (int i = 0; i < 3; i++) {
    uint64_t waves = 0;
    double t = 0;
    for (size_t k = 0; k < n; k++) {
      const uint32_t st = std::max<uint32_t>(exact_nstrips(specs[k].Q, rpls[i]), 1);
      waves += st;
      t = std::max(t, (double)exact_nsteps(specs[k].T) * tstep[i] + st * lag[i]);
      if (waves > (uint64_t)simds) break;
    }
    if (i > 0 && waves > (uint64_t)simds) break;  // the GPU is full: stay with the larger strips
    if (i == 0 || t < best_t) best = rpls[i], best_t = t;
  }
  return best;
}

int run_exact(const PairSpec* specs, size_t n, Scoring sc, bool want_trace, PairOut* outs,
              std::vector<uint8_t>* arena = nullptr) {
  Context& c = g_ctx;
  const int rpl = exact_choose_rpl(specs, n, c.cus * 4);
  size_t begin = 0;
  while (begin < n) {
    size_t end = begin, bytes = 0;
    while (end < n) {
      PathParams pp = resolve_path(specs[end].path, specs[end].Q, specs[end].T);
      size_t b = (size_t)specs[end].Q * 2 + (size_t)specs[end].T * 2 +
                 (size_t)std::max<uint32_t>(exact_nstrips(specs[end].Q, rpl), 1) * (specs[end].T + 2) * 4 + 256;
      if (want_trace) b += exact_dir_words(specs[end].Q, std::max(pp.Tlim, pp.T0), rpl) * 4 + specs[end].Q + specs[end].T;
      if (end > begin && (bytes + b > c.scratch_budget || end - begin >= (1u << 20))) break;
      bytes += b;
      end++;
    }
    int rc = run_exact_chunk(specs, begin, end, sc, want_trace, outs, arena, rpl);
    if (rc) return rc;
    begin = end;
  }
  return SW_OK;
}

// ---- Result boxes ------------------------------------------------------------------------------------------
struct ResultBox {
  Result r;
  int32_t score32;
  uint32_t magic;
};
constexpr uint32_t kMagic = 0x53574850u;  // "SWHP"
// Result strings carry a 16-byte header in front of the characters: which Result (if any) still owns them.  That
// is what lets free_c_string() be called on either string of a Result before free_alignment_result()
// (SURVEY.md 8b "Ownership") without a global registry -- a work package returns 10^5 results at a time.
struct StrHdr {
  ResultBox* owner;  // nullptr: a stand-alone string (sw_hit)
  uint64_t magic;
};
constexpr uint64_t kStrMagic = 0x5357485053545231ull;  // "SWHPSTR1"
char* str_alloc(size_t n) {
  StrHdr* h = (StrHdr*)malloc(sizeof(StrHdr) + n + 1);
  if (!h) return nullptr;
  h->owner = nullptr;
  h->magic = kStrMagic;
  char* p = (char*)(h + 1);
  p[n] = 0;
  return p;
}
void str_free(char* p) {
  if (!p) return;
  StrHdr* h = (StrHdr*)p - 1;
  h->magic = 0;
  free(h);
}

char* dup_string(const std::string& s) {
  char* p = str_alloc(s.size());
  if (p) memcpy(p, s.data(), s.size());
  return p;
}

// A sequence as the algorithm sees it: Unicode scalars truncated to u16 (`*x as u16`,
// rust/src/algorithm/mod.rs:78,224).  Pure-ASCII input (the normal case: DNA) never materialises code points.
struct Decoded {
  const char* src = nullptr;   // borrowed
  bool ascii = true;
  std::vector<uint32_t> cps;   // only when !ascii: for re-encoding the alignment
  std::vector<uint16_t> sym;
  uint32_t cp(size_t i) const { return ascii ? (uint32_t)(unsigned char)src[i] : cps[i]; }
};
bool decode(const char* s, size_t n, Decoded& d) {
  d.src = s;
  unsigned char any = 0;
  for (size_t i = 0; i < n; i++) any |= (unsigned char)s[i];
  d.sym.resize(n);
  if (!(any & 0x80)) {
    d.ascii = true;
    for (size_t i = 0; i < n; i++) d.sym[i] = (unsigned char)s[i];
    return true;
  }
  d.ascii = false;
  std::string tmp(s, n);  // decode_utf8 wants a terminator
  if (!decode_utf8(tmp.c_str(), d.cps)) return false;
  d.sym.resize(d.cps.size());
  for (size_t i = 0; i < d.cps.size(); i++) d.sym[i] = (uint16_t)d.cps[i];
  return true;
}
bool decode(const char* s, Decoded& d) { return s && decode(s, strlen(s), d); }

// Builds the two aligned strings (malloc'ed, NUL-terminated) from the walker's ops (rust/src/algorithm/mod.rs:
// 677-706: the recursion pushes after returning, so the first cell visited is the LAST column).
bool build_alignment(const Decoded& q, const Decoded& t, const PairOut& o, const uint8_t* ops, char** qa_out,
                     char** ta_out, size_t* qa_len) {
  uint32_t nx = 0, nj = 0;
  for (uint32_t k = 0; k < o.nops; k++) {
    nx += ops[k] != 2;
    nj += ops[k] != 1;
  }
  uint32_t cx = o.bx - nx, cj = o.bj - nj;  // 0-based index of the first aligned symbol
  if (q.ascii && t.ascii) {
    char* qa = str_alloc(o.nops);
    char* ta = str_alloc(o.nops);
    if (!qa || !ta) {
      str_free(qa);
      str_free(ta);
      return false;
    }
    size_t w = 0;
    for (uint32_t k = o.nops; k-- > 0; w++) {
      const uint8_t op = ops[k];
      qa[w] = op == 2 ? '-' : q.src[cx++];
      ta[w] = op == 1 ? '-' : t.src[cj++];
    }
    qa[w] = ta[w] = 0;
    *qa_out = qa;
    *ta_out = ta;
    *qa_len = w;
    return true;
  }
  std::string qa, ta;
  for (uint32_t k = o.nops; k-- > 0;) {
    const uint8_t op = ops[k];
    if (op == 2) qa.push_back('-');
    else encode_utf8(q.cp(cx++), qa);
    if (op == 1) ta.push_back('-');
    else encode_utf8(t.cp(cj++), ta);
  }
  *qa_out = dup_string(qa);
  *ta_out = dup_string(ta);
  *qa_len = qa.size();
  if (!*qa_out || !*ta_out) {
    str_free(*qa_out);
    str_free(*ta_out);
    return false;
  }
  return true;
}

// takes ownership of qa / ta (from str_alloc)
ResultBox* box_result(char* qa, char* ta, int path, const PairOut& o) {
  ResultBox* b = (ResultBox*)calloc(1, sizeof(ResultBox));
  if (!b) {
    str_free(qa);
    str_free(ta);
    return nullptr;
  }
  b->r.query_ptr = qa;
  b->r.target_ptr = ta;
  ((StrHdr*)qa - 1)->owner = b;
  ((StrHdr*)ta - 1)->owner = b;
  b->magic = kMagic;
  if (path == SW_PATH_STRAIGHT) {  // rust/src/lib.rs:70: (.., 0, x, y)
    b->score32 = 0;
    b->r.score = 0;
    b->r.max_x = o.bx;
    b->r.max_y = o.bj;
  } else {                         // rust/src/lib.rs:135, algorithm/mod.rs:444: (.., max, x-1, y-x-1), wrapping
    b->score32 = o.score;
    b->r.score = (uint16_t)o.score;
    b->r.max_x = (uint64_t)o.bx - 1;
    b->r.max_y = (uint64_t)o.bj - 1;
  }
  return b;
}
Scoring to_scoring(AlignmentScores s) {
  return Scoring{(int)(int16_t)s.match, (int)(int16_t)s.miss, (int)(int16_t)s.gap};
}

// The reference panics (-> NULL) on the ring path when gap == 0 (rust/src/algorithm/mod.rs:236-238).
bool ring_div_by_zero(int path, uint32_t Q, uint32_t T, Scoring sc) {
  return path == SW_PATH_LOW_MEMORY && !((uint64_t)T < 2ull * Q) && sc.gap == 0;
}

// host threads for the byte loops around a big submission (decoding, string assembly)
template <class F>
void parallel_ranges(size_t n, size_t work, F&& f) {
  unsigned nt = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  if (work < (1u << 20) || n < 2 * nt) nt = 1;
  // an exception inside a helper thread would end the process: it is caught there and re-raised by the caller
  std::atomic<bool> failed{false};
  auto run = [&](size_t lo, size_t hi) noexcept {
    try {
      f(lo, hi);
    } catch (...) {
      failed.store(true);
    }
  };
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; t++) {
    const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
    if (t + 1 == nt) {
      run(lo, hi);
    } else {
      try {
        th.emplace_back(run, lo, hi);
      } catch (...) {  // no thread to be had: do the range here
        run(lo, hi);
      }
    }
  }
  for (auto& x : th) x.join();
  if (failed.load()) throw std::bad_alloc();
}

int align_many(const char* const* queries, const char* const* targets, size_t n, AlignmentScores as, int path,
               Result** out) {
  if (n == 0) return SW_OK;
  StageTimer tm(n);
  if (!queries || !targets || !out) return fail(SW_E_ARG, "null argument");
  if (path < 0 || path > 2) return fail(SW_E_ARG, "bad path");
  const Scoring sc = to_scoring(as);
  // A work package names the same sequences over and over (worker_impl.go:115-117): decode each distinct input
  // pointer once.
  std::unordered_map<const char*, uint32_t> seen;
  std::vector<const char*> distinct;
  std::vector<uint32_t> qi(n), ti(n);
  auto intern = [&](const char* s) -> uint32_t {
    if (!s) return UINT32_MAX;
    auto it = seen.find(s);
    if (it != seen.end()) return it->second;
    const uint32_t id = (uint32_t)distinct.size();
    distinct.push_back(s);
    seen.emplace(s, id);
    return id;
  };
  for (size_t i = 0; i < n; i++) {
    out[i] = nullptr;
    qi[i] = intern(queries[i]);
    ti[i] = intern(targets[i]);
  }
  tm.lap("intern");
  std::vector<Decoded> dec(distinct.size());
  std::vector<uint8_t> ok(distinct.size(), 0);
  std::vector<size_t> lens(distinct.size());
  size_t total_bytes = 0;
  for (size_t d = 0; d < distinct.size(); d++) total_bytes += (lens[d] = strlen(distinct[d]));
  parallel_ranges(distinct.size(), total_bytes, [&](size_t lo, size_t hi) {
    for (size_t d = lo; d < hi; d++)  // bindings.rs:125 unwrap: invalid UTF-8 -> NULL
      ok[d] = decode(distinct[d], lens[d], dec[d]) && dec[d].sym.size() < 0x7fffff00u;
  });
  tm.lap("decode");
  std::vector<PairSpec> specs;
  std::vector<size_t> idx;
  specs.reserve(n);
  idx.reserve(n);
  for (size_t i = 0; i < n; i++) {
    if (qi[i] == UINT32_MAX || ti[i] == UINT32_MAX || !ok[qi[i]] || !ok[ti[i]]) continue;
    const Decoded &q = dec[qi[i]], &t = dec[ti[i]];
    if (ring_div_by_zero(path, (uint32_t)q.sym.size(), (uint32_t)t.sym.size(), sc)) continue;
    specs.push_back(PairSpec{q.sym.data(), (uint32_t)q.sym.size(), t.sym.data(), (uint32_t)t.sym.size(), path, qi[i], ti[i]});
    idx.push_back(i);
  }
  std::vector<PairOut> po(specs.size());
  std::vector<uint8_t> arena;
  tm.lap("specs");
  {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    int rc = ctx_init_locked(-1);
    if (rc) return rc;
    rc = run_exact(specs.data(), specs.size(), sc, true, po.data(), &arena);
    if (rc) return rc;
  }
  tm.lap("run_exact");
  parallel_ranges(specs.size(), arena.size() * 4, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++) {
      const size_t i = idx[k];
      char *qa = nullptr, *ta = nullptr;
      size_t qlen = 0;
      if (!build_alignment(dec[qi[i]], dec[ti[i]], po[k], arena.data() + po[k].ops_off, &qa, &ta, &qlen)) continue;
      ResultBox* b = box_result(qa, ta, path, po[k]);
      out[i] = b ? &b->r : nullptr;
    }
  });
  tm.lap("strings");
  return SW_OK;
}

// ---- one pair per call, many callers: combining ---------------------------------------------------------------
// The Go worker enters the drop-in ABI from NumCPU-1 goroutines, one pair per call (worker_impl.go:204-238).  A GPU
// round trip costs a few hundred microseconds whatever the batch size, so concurrent callers are COMBINED: a caller
// that finds no submission in flight becomes the leader and submits everything that has queued up (its own request
// included) as one batch; the others sleep until their result is there.  A leader steps down once its own request
// is done, and a waiter whose request is still queued takes over.  Results are identical to one call per pair
// (align_many computes every pair independently); with a single caller this is exactly the old path.
struct OneReq {
  const char* q;
  const char* t;
  AlignmentScores as;
  int path;
  Result* out = nullptr;
  int rc = SW_OK;
  std::string err;
  bool done = false;
};
std::mutex g_comb_mu;
std::condition_variable g_comb_cv;
std::vector<OneReq*> g_comb_pending;
bool g_comb_leader = false;
size_t g_comb_expect = 1;  // how many concurrent callers recent batches have seen (decays)

void combine_run_batch(std::vector<OneReq*>& batch) {
  // group by (scores, path): one submission per group (a worker uses one scoring per work package)
  std::vector<char> taken(batch.size(), 0);
  for (size_t i = 0; i < batch.size(); i++) {
    if (taken[i]) continue;
    std::vector<size_t> grp;
    for (size_t k = i; k < batch.size(); k++)
      if (!taken[k] && batch[k]->path == batch[i]->path && memcmp(&batch[k]->as, &batch[i]->as, sizeof(AlignmentScores)) == 0) {
        taken[k] = 1;
        grp.push_back(k);
      }
    std::vector<const char*> qs(grp.size()), ts(grp.size());
    std::vector<Result*> out(grp.size(), nullptr);
    for (size_t k = 0; k < grp.size(); k++) {
      qs[k] = batch[grp[k]]->q;
      ts[k] = batch[grp[k]]->t;
    }
    g_err.clear();
    const int rc = swhip::guarded<int>(SW_E_NOMEM, [&] {
      return align_many(qs.data(), ts.data(), grp.size(), batch[i]->as, batch[i]->path, out.data());
    });
    for (size_t k = 0; k < grp.size(); k++) {
      OneReq* r = batch[grp[k]];
      r->rc = rc;
      r->out = rc == SW_OK ? out[k] : nullptr;
      if (!r->out) r->err = rc != SW_OK ? g_err : std::string("invalid input (UTF-8) or reference panic condition");
    }
  }
}

Result* align_one(char* query, char* target, AlignmentScores as, int path) {
  OneReq me;
  me.q = query;
  me.t = target;
  me.as = as;
  me.path = path;
  std::unique_lock<std::mutex> lk(g_comb_mu);
  g_comb_pending.push_back(&me);
  if (g_comb_leader) g_comb_cv.notify_all();  // a leader may be gathering
  while (!me.done) {
    if (g_comb_leader) {
      g_comb_cv.wait(lk);
      continue;
    }
    g_comb_leader = true;  // lead until my own request has been served
    while (!me.done) {
      // Callers released by the previous batch are on their way back with their next pair: give them a moment
      // (bounded, a fraction of a GPU round trip) instead of submitting a batch of two.
      if (g_comb_pending.size() < g_comb_expect) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(150);
        while (g_comb_pending.size() < g_comb_expect && std::chrono::steady_clock::now() < deadline)
          g_comb_cv.wait_for(lk, std::chrono::microseconds(25));
      }
      std::vector<OneReq*> batch;
      batch.swap(g_comb_pending);
      g_comb_expect = std::max(batch.size(), g_comb_expect - (g_comb_expect + 3) / 4);
      lk.unlock();
      try {
        combine_run_batch(batch);
      } catch (...) {  // out of host memory while grouping: the batch fails, the protocol goes on
        for (OneReq* r : batch)
          if (!r->out) r->rc = SW_E_NOMEM;
      }
      lk.lock();
      for (OneReq* r : batch) r->done = true;
      g_comb_cv.notify_all();
    }
    g_comb_leader = false;
    g_comb_cv.notify_all();  // a waiter whose request is still queued takes over
  }
  lk.unlock();
  if (!me.out) g_err = me.err.empty() && me.rc == SW_E_NOMEM ? std::string("out of host memory") : me.err;
  return me.out;
}
// the three drop-in entry points: any failure, a C++ exception included, is a NULL result
Result* align_one_guarded(char* query, char* target, AlignmentScores as, int path) {
  return swhip::guarded<Result*>(nullptr, [&] { return align_one(query, target, as, path); });
}

}  // namespace

// ================================================================================================
// include/sw.h
// ================================================================================================
extern "C" {

Result* find_alignment_low_memory(char* query, char* target, AlignmentScores s) {
  return align_one_guarded(query, target, s, SW_PATH_LOW_MEMORY);
}
Result* find_alignment_simd(char* query, char* target, AlignmentScores s) {
  return align_one_guarded(query, target, s, SW_PATH_SIMD);
}
Result* find_alignment_sequential_straight(char* query, char* target, AlignmentScores s) {
  return align_one_guarded(query, target, s, SW_PATH_STRAIGHT);
}
Result* find_alignment_parallel(char*, char*, size_t, AlignmentScores) {
  g_err = "find_alignment_parallel is declared by rust/header/sw.h but not defined by the reference";
  return nullptr;
}
Result* find_alignment_sequential(char*, char*, AlignmentScores) {
  g_err = "find_alignment_sequential is declared by rust/header/sw.h but not defined by the reference";
  return nullptr;
}

void free_alignment_result(Result* result) {
  if (!result) return;
  ResultBox* b = (ResultBox*)result;
  if (b->magic != kMagic) return;  // not ours (or already freed): leave it alone
  str_free(b->r.query_ptr);        // (a string handed to free_c_string earlier has been detached: nullptr)
  str_free(b->r.target_ptr);
  b->magic = 0;
  free(b);
}

void free_c_string(char* str) {
  if (!str) return;
  StrHdr* h = (StrHdr*)str - 1;
  if (h->magic != kStrMagic) {  // not one of ours: the reference would hand it to CString::from_raw all the same
    free(str);
    return;
  }
  if (ResultBox* b = h->owner) {  // detach from its Result so free_alignment_result will not free it again
    if (b->r.query_ptr == str) b->r.query_ptr = nullptr;
    if (b->r.target_ptr == str) b->r.target_ptr = nullptr;
  }
  str_free(str);
}

// ================================================================================================
// include/sw_hip.h
// ================================================================================================
int sw_hip_init(int device) {
  return swhip::guarded<int>(SW_E_NOMEM, [&] {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    return ctx_init_locked(device);
  });
}
const char* sw_hip_last_error(void) { return g_err.c_str(); }
const char* sw_hip_version(void) { return "libsw_hip 0.1 (gfx950; exact int32 wavefront + packed-int16 inter-task kernels)"; }

int32_t sw_result_score32(const Result* result) {
  if (!result) return 0;
  const ResultBox* b = (const ResultBox*)result;
  return b->magic == kMagic ? b->score32 : (int32_t)(int16_t)result->score;
}

int sw_align_batch(const char* const* queries, const char* const* targets, size_t n_pairs, AlignmentScores scores,
                   int path, Result** out) {
  if (out)
    for (size_t i = 0; i < n_pairs; i++) out[i] = nullptr;
  const int rc = swhip::guarded<int>(SW_E_NOMEM, [&] { return align_many(queries, targets, n_pairs, scores, path, out); });
  if (rc != SW_OK && out)  // nothing is handed over on failure
    for (size_t i = 0; i < n_pairs; i++) {
      free_alignment_result(out[i]);
      out[i] = nullptr;
    }
  return rc;
}

int sw_hip_path_params(int path, uint64_t q_len, uint64_t t_len, uint32_t out[4]) {
  if (!out || path < 0 || path > 2 || q_len >= 0x7fffff00ull || t_len >= 0x7fffff00ull) return fail(SW_E_ARG, "bad argument");
  const PathParams pp = resolve_path(path, (uint32_t)q_len, (uint32_t)t_len);
  out[0] = pp.Tlim;
  out[1] = pp.T0;
  out[2] = pp.phaseQp;
  out[3] = pp.straight;
  return SW_OK;
}

int sw_hip_last_stats(sw_kernel_stats* out) {
  if (!out) return fail(SW_E_ARG, "null argument");
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  *out = g_ctx.stats;
  return SW_OK;
}

}  // extern "C"

// ================================================================================================
// resident database + top-k
// ================================================================================================
struct sw_db {
  uint64_t n = 0, residues = 0;
  std::vector<uint64_t> off;       // n + 1
  std::string flat;                // original bytes
  std::vector<uint8_t> packed_ok;  // per target: only A, C, G, T
  // packed part
  std::vector<uint32_t> slot_target;  // per slot: database index or UINT32_MAX
  std::vector<ScoreGroup> groups;
  uint32_t* d_words = nullptr;
  uint32_t* d_lens = nullptr;
  uint32_t* d_slot_target = nullptr;  // per slot: database index (UINT32_MAX: padding)
  uint64_t packed_residues = 0;       // residues of the packed targets
  ScoreGroup* d_groups = nullptr;
  uint32_t* d_elem_group = nullptr;  // per strip of the stream: its group
  uint32_t n_elems = 0;              // strips in the stream
  // ticket-ordered unit table, rebuilt when the lane count / ring geometry of a search differs (build_units)
  ScoreUnit* d_units = nullptr;
  uint32_t n_units = 0, units_S = 0, units_R = 0, units_W = 0;
  std::vector<uint32_t> generic;  // targets scored by the exact kernel only
  uint32_t max_len = 0;
};

namespace {

inline int nt_code(unsigned char ch) {
  switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

void db_release(sw_db* db) {
  if (!db) return;
  if (db->d_words) (void)hipFree(db->d_words);
  if (db->d_lens) (void)hipFree(db->d_lens);
  if (db->d_slot_target) (void)hipFree(db->d_slot_target);
  if (db->d_groups) (void)hipFree(db->d_groups);
  if (db->d_elem_group) (void)hipFree(db->d_elem_group);
  if (db->d_units) (void)hipFree(db->d_units);
  delete db;
}

// host threads for building a database (validation and 2-bit packing are byte loops over GBs)
template <class F>
void parallel_ranges(size_t n, F&& f) {
  unsigned nt = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  if (n < 4096) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) {
    const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
    if (t + 1 == nt) f(lo, hi);
    else th.emplace_back([=, &f] { f(lo, hi); });
  }
  for (auto& x : th) x.join();
}

sw_db* db_build(std::string&& flat, std::vector<uint64_t>&& off) {
  sw_db* db = new sw_db;
  db->n = off.size() - 1;
  db->flat = std::move(flat);
  db->off = std::move(off);
  db->residues = db->off.back() - db->off.front();
  db->packed_ok.assign(db->n, 1);
  for (uint64_t i = 0; i < db->n; i++) {
    const uint64_t len = db->off[i + 1] - db->off[i];
    if (len >= 0x7fffff00ull) {
      db_release(db);
      fail(SW_E_ARG, "sequence too long");
      return nullptr;
    }
    db->max_len = std::max<uint32_t>(db->max_len, (uint32_t)len);
  }
  // per target: 1 = only A/C/G/T (packable), 0 = something else (exact kernel), 2 = NUL inside
  static const std::array<int8_t, 256> kCode = [] {
    std::array<int8_t, 256> t;
    t.fill(-1);
    t['A'] = 0, t['C'] = 1, t['G'] = 2, t['T'] = 3;
    return t;
  }();
  std::atomic<bool> has_nul{false};
  parallel_ranges(db->n, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      const unsigned char* s = (const unsigned char*)db->flat.data() + db->off[i];
      const uint64_t len = db->off[i + 1] - db->off[i];
      int bad = 0, nul = 0;
      for (uint64_t k = 0; k < len; k++) {
        bad |= kCode[s[k]];  // any -1 sets the sign bit
        nul |= (s[k] == 0);
      }
      if (nul) has_nul = true;
      db->packed_ok[i] = bad >= 0;
    }
  });
  if (has_nul) {
    db_release(db);
    fail(SW_E_ARG, "NUL byte inside a sequence");
    return nullptr;
  }
  // sequences are text like the per-pair inputs (bindings.rs:125-129): anything non-ASCII must be valid UTF-8
  for (uint64_t i = 0; i < db->n; i++) {
    if (db->packed_ok[i]) continue;
    Decoded d;
    if (!decode(db->flat.data() + db->off[i], (size_t)(db->off[i + 1] - db->off[i]), d)) {
      db_release(db);
      fail(SW_E_ARG, "sequence " + std::to_string(i) + " is not valid UTF-8");
      return nullptr;
    }
  }
  std::vector<uint32_t> packable;
  for (uint64_t i = 0; i < db->n; i++) {
    if (db->packed_ok[i] && db->off[i + 1] > db->off[i]) packable.push_back((uint32_t)i);
    else db->generic.push_back((uint32_t)i);
  }
  // length-sorted (descending, stable on index) groups of 128
  std::stable_sort(packable.begin(), packable.end(), [&](uint32_t a, uint32_t b) {
    return (db->off[a + 1] - db->off[a]) > (db->off[b + 1] - db->off[b]);
  });
  const size_t ngroups = (packable.size() + kGroupTargets - 1) / kGroupTargets;
  db->groups.resize(ngroups);
  db->slot_target.assign(ngroups * kGroupTargets, UINT32_MAX);
  std::vector<uint32_t> lens(ngroups * kGroupTargets, 0);
  size_t total_strips = 0;
  for (size_t g = 0; g < ngroups; g++) {
    uint32_t maxlen = 0;
    for (int s = 0; s < kGroupTargets; s++) {
      size_t k = g * kGroupTargets + s;
      if (k >= packable.size()) break;
      uint32_t ti = packable[k];
      uint32_t len = (uint32_t)(db->off[ti + 1] - db->off[ti]);
      db->slot_target[k] = ti;
      lens[k] = len;
      db->packed_residues += len;
      maxlen = std::max(maxlen, len);
    }
    if (total_strips > 0xfff00000ull) {
      db_release(db);
      fail(SW_E_ARG, "database too large for one device (more than 2^32 strips)");
      return nullptr;
    }
    db->groups[g].strip_base = (uint32_t)total_strips;
    db->groups[g].nstrips = (maxlen + kScoreR - 1) / kScoreR;
    db->groups[g].first_slot = (uint32_t)(g * kGroupTargets);
    db->groups[g].pad_ = 0;
    total_strips += db->groups[g].nstrips;
  }
  db->n_elems = (uint32_t)total_strips;
  // the strip stream: strip e = dwords [e*128, e*128+128), [half][lane], 16 residues (2 bits each) per dword
  std::vector<uint32_t> words(std::max<size_t>(total_strips * kGroupTargets, 1), 0);
  std::vector<uint32_t> elem_group(std::max<size_t>(total_strips, 1), 0);
  parallel_ranges(ngroups, [&](size_t glo, size_t ghi) {
    for (size_t g = glo; g < ghi; g++) {
      const ScoreGroup& G = db->groups[g];
      for (uint32_t st = 0; st < G.nstrips; st++) elem_group[G.strip_base + st] = (uint32_t)g;
      for (int s = 0; s < kGroupTargets; s++) {
        size_t k = g * kGroupTargets + s;
        uint32_t ti = db->slot_target[k];
        if (ti == UINT32_MAX) continue;
        const unsigned char* seq = (const unsigned char*)db->flat.data() + db->off[ti];
        const uint32_t len = lens[k];
        uint32_t* dst = words.data() + (size_t)G.strip_base * kGroupTargets + (size_t)(s / 64) * 64 + (s % 64);
        uint32_t col = 0;
        for (; col + kScoreR <= len; col += kScoreR, dst += kGroupTargets) {
          uint32_t w = 0;
          for (int r = 0; r < kScoreR; r++) w |= (uint32_t)kCode[seq[col + r]] << (2 * r);
          *dst = w;
        }
        if (col < len) {
          uint32_t w = 0;
          for (int r = 0; col + r < len; r++) w |= (uint32_t)kCode[seq[col + r]] << (2 * r);
          *dst = w;
        }
      }
    }
  });
  if (ngroups) {
    if (hipMalloc(&db->d_words, words.size() * 4) != hipSuccess || hipMalloc(&db->d_lens, lens.size() * 4) != hipSuccess ||
        hipMalloc(&db->d_slot_target, lens.size() * 4) != hipSuccess ||
        hipMalloc(&db->d_groups, ngroups * sizeof(ScoreGroup)) != hipSuccess ||
        hipMalloc(&db->d_elem_group, elem_group.size() * 4) != hipSuccess) {
      db_release(db);
      fail(SW_E_NOMEM, "hipMalloc failed for the database");
      return nullptr;
    }
    if (hipMemcpy(db->d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db->d_lens, lens.data(), lens.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db->d_slot_target, db->slot_target.data(), lens.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db->d_groups, db->groups.data(), ngroups * sizeof(ScoreGroup), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db->d_elem_group, elem_group.data(), elem_group.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      db_release(db);
      fail(SW_E_HIP, "hipMemcpy failed for the database");
      return nullptr;
    }
  }
  return db;
}

// Ticket order of the score kernel's work units.  The strip stream is cut into lanes (runs of whole groups,
// cut where the last unit is nearly full); S lanes are active at a time and tickets visit them round-robin, so
// consecutive units of one lane -- the only ones that depend on each other -- are S tickets apart.
// R = slots of the boundary ring (slot of ticket t: t % R); `reuse` = the latest earlier reader of that slot.
//
// The tail: n_units is rarely a multiple of the number of persistent workgroups, and the last, partial round would
// keep a handful of CUs busy for a whole unit time while the rest idle (bench shape: 3079 units on 256 CUs = 12
// rounds + 7 units, i.e. 8 % of the kernel for 0.2 % of the work).  The units of that partial round are cut into
// up to four short units (fewer strips): they land on more CUs with one wave per SIMD instead of four, which runs
// a strip more than twice as fast.
void build_units(const sw_db* db, uint32_t W, uint32_t S, uint32_t R, uint32_t max_wg, std::vector<ScoreUnit>& units) {
  struct Lane { uint32_t pos, end, last_ticket; };
  std::vector<Lane> lanes;
  const size_t ng = db->groups.size();
  const uint64_t approx_units = ((uint64_t)db->n_elems + W - 1) / W;
  const uint64_t Lmin = std::min<uint64_t>(32, std::max<uint64_t>(2, approx_units / (4ull * S)));
  const uint64_t Lmax = 4 * Lmin + 16;
  for (size_t g = 0; g < ng;) {
    uint64_t cum = 0;
    size_t best = g;
    double best_fill = -1;
    for (size_t h = g; h < ng; h++) {
      cum += db->groups[h].nstrips;
      const uint64_t u = (cum + W - 1) / W;
      if (u < Lmin && h + 1 < ng) continue;
      const double fill = (double)cum / (double)(u * W);
      if (fill > best_fill) best_fill = fill, best = h;
      if (fill >= 0.995 || u >= Lmax) break;
    }
    lanes.push_back(Lane{db->groups[g].strip_base, db->groups[best].strip_base + db->groups[best].nstrips, UINT32_MAX});
    g = best + 1;
  }
  const std::vector<Lane> lanes0 = lanes;
  // pass 0 counts the units; pass 1 generates them, with short units from ticket `full` on
  uint32_t full = UINT32_MAX, Wtail = W;
  for (int pass = 0; pass < 2; pass++) {
    lanes = lanes0;
    units.clear();
    units.reserve((size_t)approx_units + lanes.size() + 4 * max_wg);
    std::vector<uint32_t> slot_reader(R, UINT32_MAX);
    std::vector<size_t> active;
    size_t next_lane = 0;
    while (active.size() < S && next_lane < lanes.size()) active.push_back(next_lane++);
    while (!active.empty()) {
      for (size_t i = 0; i < active.size();) {
        Lane& L = lanes[active[i]];
        const uint32_t t = (uint32_t)units.size();
        ScoreUnit u;
        u.first_elem = L.pos;
        u.count = std::min<uint32_t>(t >= full ? Wtail : W, L.end - L.pos);
        u.dep = L.last_ticket;
        u.reuse = slot_reader[t % R];
        if (u.dep != UINT32_MAX) slot_reader[u.dep % R] = t;
        units.push_back(u);
        L.pos += u.count;
        L.last_ticket = t;
        if (L.pos >= L.end) {
          if (next_lane < lanes.size()) active[i++] = next_lane++;
          else active.erase(active.begin() + (ptrdiff_t)i);
        } else {
          i++;
        }
      }
    }
    if (pass == 0) {
      const uint32_t n = (uint32_t)units.size();
      const uint32_t r = max_wg ? n % max_wg : 0;
      const uint32_t f = r ? std::min<uint32_t>(4, max_wg / r) : 0;
      if (n <= max_wg || f < 2 || W < 8) break;  // no partial round worth cutting
      full = n - r;
      Wtail = W / f;
    }
  }
}

}  // namespace
// Test hook (host arithmetic only, no GPU): builds the ticket-ordered unit table for a database whose groups have
// the given numbers of strips and checks the invariants the score kernel relies on -- every strip is covered
// exactly once, units are 1..W strips, a unit that continues a group depends on the EARLIER ticket that ends right
// before it, the ring slot it writes was last read by an earlier ticket.  Returns the number of units, or a
// negative number naming the violated invariant.  out[0] = units of the short (tail) kind, out[1] = their size.
extern "C" int64_t sw_hip_test_units(const uint32_t* group_strips, uint32_t n_groups, uint32_t W, uint32_t S, uint32_t R,
                                     uint32_t max_wg, uint32_t* out) {
  if (!group_strips || !W || !S || !R) return -100;
  return swhip::guarded<int64_t>(-101, [&]() -> int64_t {
  sw_db db;
  uint32_t total = 0;
  std::vector<uint32_t> elem_group;
  for (uint32_t g = 0; g < n_groups; g++) {
    ScoreGroup G{};
    G.strip_base = total;
    G.nstrips = group_strips[g];
    G.first_slot = g * kGroupTargets;
    db.groups.push_back(G);
    for (uint32_t k = 0; k < group_strips[g]; k++) elem_group.push_back(g);
    total += group_strips[g];
  }
  db.n_elems = total;
  std::vector<ScoreUnit> units;
  build_units(&db, W, S, R, max_wg, units);
  std::vector<uint8_t> covered(total, 0);
  std::vector<uint32_t> last_reader(R, UINT32_MAX);
  uint32_t n_short = 0, short_size = 0;
  for (uint32_t t = 0; t < units.size(); t++) {
    const ScoreUnit& u = units[t];
    if (u.count < 1 || u.count > W) return -1;
    if ((uint64_t)u.first_elem + u.count > total) return -2;
    for (uint32_t k = 0; k < u.count; k++) {
      if (covered[u.first_elem + k]) return -3;
      covered[u.first_elem + k] = 1;
    }
    const uint32_t g0 = elem_group[u.first_elem];
    const bool continues = u.first_elem > db.groups[g0].strip_base;  // wave 0 needs a left column from another unit
    if (continues) {
      if (u.dep == UINT32_MAX || u.dep >= t) return -4;
      const ScoreUnit& d = units[u.dep];
      if (d.first_elem + d.count != u.first_elem) return -5;
    }
    if (u.reuse != UINT32_MAX && u.reuse >= t) return -6;
    if (u.reuse != last_reader[t % R]) return -7;
    if (u.dep != UINT32_MAX) {
      if (u.dep >= t) return -8;
      last_reader[u.dep % R] = t;
    }
    if (u.count < W && u.first_elem + u.count < total &&
        elem_group[u.first_elem + u.count] == elem_group[u.first_elem + u.count - 1]) {  // short, and the group goes on
      n_short++;
      short_size = std::max(short_size, u.count);
    }
  }
  for (uint32_t e = 0; e < total; e++)
    if (!covered[e]) return -9;
  if (out) {
    out[0] = n_short;
    out[1] = short_size;
  }
  return (int64_t)units.size();
  });
}

namespace {

struct QueryPrep {
  Decoded d;
  bool acgt = true;
  std::vector<uint32_t> qwords;
};

bool fast_domain(Scoring sc) {
  return sc.match > 0 && sc.miss < 0 && sc.gap < 0 && sc.match <= 16384 && sc.miss >= -16384 && sc.gap >= -16384;
}

// Launches the score kernel for one query over the packed part of the database (asynchronous, on the library's
// stream; ev0/ev1 bracket the kernel).  The per-slot maxima are left in c.slot_scores.
int score_pass_locked(sw_db* db, const QueryPrep& qp, Scoring sc) {
  Context& c = g_ctx;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  const uint32_t nblk = (Q + kScoreB - 1) / kScoreB;
  const size_t Qpad = (size_t)nblk * kScoreB;
  const uint32_t W = (uint32_t)score_waves_per_group(db->n_elems, c.cus);
  const uint32_t max_wg = (uint32_t)c.cus * (uint32_t)score_workgroups_per_cu((int)W);
  // Active lanes S of the ticket order (build_units): the unit that continues a group is handed out S tickets
  // after its producer.  With S beyond the number of resident workgroups the producer has FINISHED by then and the
  // consumer never waits; a consumer that trails a running producer (small S) costs 5-9 % whatever the lag
  // (swept 1..8192 on every shape of scratch/sweep.py: flat from ~512 up).  So: as many lanes as the database has
  // groups, up to 1024, as long as the boundary ring (one slot of Qpad x 256 B per unit in flight or waiting) stays
  // within twice the scratch budget; a database with few groups or a very long query falls back to chains.
  const uint32_t nblk16 = (Q + 15) / 16;
  // boundary ring: a workgroup has at most 1 + queue depth units in flight, fewer when a unit is long
  const uint32_t ahead = std::min<uint32_t>(score_queue_depth(), 48 / nblk16 + 1);
  const uint32_t ring_min = max_wg * (1 + ahead) + 8;
  const uint64_t ring_fit = 2 * (uint64_t)c.scratch_budget / ((uint64_t)Qpad * 256);
  uint32_t S = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(8, db->groups.size()));
  if (ring_fit < (uint64_t)ring_min + S) S = (uint32_t)std::max<uint64_t>(8, ring_fit > ring_min ? ring_fit - ring_min : 0);
  if (const char* e = getenv("SW_HIP_SCORE_LANES")) S = std::min(16384, std::max(1, atoi(e)));  // experiments
  const uint32_t ring_slots = S + max_wg * (1 + ahead) + 8;
  if (db->units_S != S || db->units_R != ring_slots || db->units_W != W || !db->d_units) {
    std::vector<ScoreUnit> units;
    build_units(db, W, S, ring_slots, max_wg, units);
    if (db->d_units) (void)hipFree(db->d_units);
    db->d_units = nullptr;
    if (hipMalloc(&db->d_units, std::max<size_t>(units.size(), 1) * sizeof(ScoreUnit)) != hipSuccess)
      return fail(SW_E_NOMEM, "hipMalloc failed for the unit table");
    HIPCK(hipMemcpy(db->d_units, units.data(), units.size() * sizeof(ScoreUnit), hipMemcpyHostToDevice));
    db->n_units = (uint32_t)units.size();
    db->units_S = S;
    db->units_R = ring_slots;
    db->units_W = W;
  }
  const int nwg = (int)std::min<uint32_t>(db->n_units, max_wg);
  const size_t gb_stride = Qpad * 64;
  int rc;
  if ((rc = c.qwords.ensure(qp.qwords.size() * 4))) return rc;
  if ((rc = c.slot_scores.ensure(db->slot_target.size() * 4))) return rc;
  if ((rc = c.gbound.ensure((size_t)ring_slots * gb_stride * 4))) return rc;
  if ((rc = c.ticket.ensure(256))) return rc;
  if ((rc = c.prog.ensure(std::max<size_t>((size_t)db->n_units * 8, 16)))) return rc;
  HIPCK(hipMemcpyAsync(c.qwords.p, qp.qwords.data(), qp.qwords.size() * 4, hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemsetAsync(c.ticket.p, 0, 256, c.stream));
  HIPCK(hipMemsetAsync(c.prog.p, 0, std::max<size_t>((size_t)db->n_units * 8, 16), c.stream));
  HIPCK(hipMemsetAsync(c.slot_scores.p, 0, db->slot_target.size() * 4, c.stream));
  ScoreParams p{};
  p.words = db->d_words;
  p.groups = db->d_groups;
  p.lens = db->d_lens;
  p.qwords = (const uint32_t*)c.qwords.p;
  p.Q = Q;
  p.elem_group = db->d_elem_group;
  p.units = db->d_units;
  p.n_units = db->n_units;
  p.n_elems = db->n_elems;
  p.n_groups = (uint32_t)db->groups.size();
  p.n_slots = (uint32_t)db->slot_target.size();
  p.slot_scores = (int32_t*)c.slot_scores.p;
  p.gbound = (uint32_t*)c.gbound.p;
  p.gbound_stride = gb_stride;
  p.ring_slots = ring_slots;
  p.prog = (uint32_t*)c.prog.p;
  p.ticket = (uint32_t*)c.ticket.p;
  p.err = (uint32_t*)c.ticket.p + 32;
  p.match = sc.match;
  p.miss = sc.miss;
  p.gap = sc.gap;
  // a small database is a few chains of units: publish their boundary columns often (latency), a big one rarely
  // (in blocks; swept: 10 kb queries like 10 blocks, 1 kb queries 5 -- a short unit is over before a coarse
  // publication lets its consumer start)
  p.pub = W == 4 ? 2 : nblk >= 400 ? 10 : 5;
  if (const char* e = getenv("SW_HIP_SCORE_PUB")) p.pub = std::max(1, atoi(e));  // experiments
  HIPCK(hipEventRecord(c.ev0, c.stream));
  HIPCK(launch_score(p, nwg, (int)W, c.stream));
  HIPCK(hipEventRecord(c.ev1, c.stream));
  return SW_OK;
}

// after the stream has been synchronised: error word + kernel time of the last score pass
int score_pass_finish_locked(sw_db* db, uint32_t Q, uint32_t kerr) {
  Context& c = g_ctx;
  if (kerr) {
    static const char* const what[] = {"", "a wait for another workgroup's boundary column ran out of time",
                                       "a wait for a neighbouring wave ran out of time",
                                       "the unit queue of a workgroup stalled", "inconsistent unit table entry",
                                       "boundary ring slot smaller than the query", "strip with a bad group index",
                                       "strip outside its group / chained unit without a producer"};
    return fail(SW_E_HIP, std::string("score kernel: ") + (kerr < 8 ? what[kerr] : "unknown error") + " (code " +
                              std::to_string(kerr) + ")");
  }
  float ms = 0;
  HIPCK(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  c.stats.score_kernel_ms += ms;
  c.stats.score_launches += 1;
  c.stats.score_cells += (uint64_t)Q * db->packed_residues;
  return SW_OK;
}

// A packed target's int16 score is final unless one of these holds (the same rules run on the
// device in sw_topk.hip):
bool needs_exact(uint32_t T, int32_t s, uint32_t Q, uint32_t Qp, Scoring sc) {
  const bool full = (uint64_t)T < 2ull * Q;
  if (s >= 32767) return true;                                                    // int16 saturation -> int32 re-run
  if (!full && T < Qp) return true;                                               // ring path with T < Qp (quirk Q2)
  if (full && T % 64 == 0 && roundup64(T) >= Qp && s < sc.match) return true;     // only H(1,T) can score
  return false;
}

// The database path answers what the WORKER would have got for the pair, i.e. the fallback chain of
// worker_impl.go:170-202: find_alignment_low_memory, or find_alignment_simd where that one panics (gap == 0 on
// the ring path, rust/src/algorithm/mod.rs:236-238).
int db_pair_path(uint32_t Q, uint32_t T, Scoring sc) {
  return ring_div_by_zero(SW_PATH_LOW_MEMORY, Q, T, sc) ? SW_PATH_SIMD : SW_PATH_LOW_MEMORY;
}

// exact (int32, score only) re-run of a list of database targets
int rescore_locked(sw_db* db, const QueryPrep& qp, Scoring sc, const std::vector<uint32_t>& rerun,
                   std::vector<int32_t>& out) {
  out.assign(rerun.size(), 0);
  if (rerun.empty()) return SW_OK;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  std::vector<PairSpec> specs(rerun.size());
  std::vector<Decoded> tdec(rerun.size());
  for (size_t k = 0; k < rerun.size(); k++) {
    uint32_t ti = rerun[k];
    // (valid UTF-8: checked when the database was built)
    decode(db->flat.data() + db->off[ti], (size_t)(db->off[ti + 1] - db->off[ti]), tdec[k]);
    specs[k] = PairSpec{qp.d.sym.data(), Q, tdec[k].sym.data(), (uint32_t)tdec[k].sym.size(),
                        db_pair_path(Q, (uint32_t)tdec[k].sym.size(), sc)};
  }
  std::vector<PairOut> po(rerun.size());
  int rc = run_exact(specs.data(), specs.size(), sc, false, po.data());
  if (rc) return rc;
  for (size_t k = 0; k < rerun.size(); k++) out[k] = po[k].score;
  g_ctx.stats.rerun_pairs += (uint32_t)rerun.size();
  return SW_OK;
}

bool fast_path(const sw_db* db, const QueryPrep& qp, Scoring sc) {
  return qp.acgt && fast_domain(sc) && !qp.d.sym.empty() && !db->groups.empty();
}

// scores of every database sequence against one query, reference-exact (low_memory semantics)
int db_scores_locked(sw_db* db, const QueryPrep& qp, Scoring sc, std::vector<int32_t>& scores) {
  Context& c = g_ctx;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  scores.assign(db->n, 0);
  std::vector<uint32_t> rerun(db->generic);
  if (!fast_path(db, qp, sc)) {
    rerun.resize(db->n);
    std::iota(rerun.begin(), rerun.end(), 0u);
  } else {
    int rc = score_pass_locked(db, qp, sc);
    if (rc) return rc;
    std::vector<int32_t> slot(db->slot_target.size());
    HIPCK(hipMemcpyAsync(slot.data(), c.slot_scores.p, slot.size() * 4, hipMemcpyDeviceToHost, c.stream));
    uint32_t kerr = 0;
    HIPCK(hipMemcpyAsync(&kerr, (uint32_t*)c.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipStreamSynchronize(c.stream));
    if ((rc = score_pass_finish_locked(db, Q, kerr))) return rc;
    const uint32_t Qp = roundup64(Q);
    for (size_t k = 0; k < slot.size(); k++) {
      uint32_t ti = db->slot_target[k];
      if (ti == UINT32_MAX) continue;
      const uint32_t T = (uint32_t)(db->off[ti + 1] - db->off[ti]);
      if (needs_exact(T, slot[k], Q, Qp, sc)) rerun.push_back(ti);
      else scores[ti] = slot[k];
    }
  }
  std::vector<int32_t> rs;
  int rc = rescore_locked(db, qp, sc, rerun, rs);
  if (rc) return rc;
  for (size_t k = 0; k < rerun.size(); k++) scores[rerun[k]] = rs[k];
  return SW_OK;
}

struct TopEntry {
  int32_t score;
  uint32_t target;
};
inline bool top_before(const TopEntry& a, const TopEntry& b) {
  // score descending, then database index ascending (declared replacement of the reference's arrival-order
  // ties, cli/__main__.py:171)
  return a.score != b.score ? a.score > b.score : a.target < b.target;
}

bool g_host_topk = [] {
  const char* e = getenv("SW_HIP_TOPK");
  return e && std::string(e) == "host";
}();
constexpr uint32_t kTieCap = 65536;

// The k best (score, target) of the database for one query.  Selection runs on the device (sw_topk.hip): only
// the candidates and the slots that need an exact re-run are downloaded.  Falls back to the full download when
// the tie list overflows (e.g. a database where every target scores the same) or with SW_HIP_TOPK=host.
int db_topk_locked(sw_db* db, const QueryPrep& qp, Scoring sc, size_t k, std::vector<TopEntry>& top) {
  Context& c = g_ctx;
  top.clear();
  k = std::min<size_t>(k, db->n);
  if (k == 0) return SW_OK;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  bool device_done = false;
  std::vector<TopEntry> pool;
  if (fast_path(db, qp, sc) && !g_host_topk && k <= 4096) {
    int rc = score_pass_locked(db, qp, sc);
    if (rc) return rc;
    const uint32_t n_slots = (uint32_t)db->slot_target.size();
    if ((rc = c.sel_hist.ensure(32768 * 4))) return rc;
    if ((rc = c.sel_ctr.ensure(kSelWords * 4))) return rc;
    if ((rc = c.sel_rerun.ensure((size_t)n_slots * 4))) return rc;
    if ((rc = c.sel_cand.ensure(std::max<size_t>(k, 16) * 8))) return rc;
    if ((rc = c.sel_tie.ensure((size_t)kTieCap * 8))) return rc;
    HIPCK(hipMemsetAsync(c.sel_hist.p, 0, 32768 * 4, c.stream));
    HIPCK(hipMemsetAsync(c.sel_ctr.p, 0, kSelWords * 4, c.stream));
    SelectParams sp{};
    sp.slot_scores = (const int32_t*)c.slot_scores.p;
    sp.lens = db->d_lens;
    sp.slot_target = db->d_slot_target;
    sp.n_slots = n_slots;
    sp.Q = Q;
    sp.match = sc.match;
    sp.k = (uint32_t)k;
    sp.hist = (uint32_t*)c.sel_hist.p;
    sp.counters = (uint32_t*)c.sel_ctr.p;
    sp.rerun = (uint32_t*)c.sel_rerun.p;
    sp.rerun_cap = n_slots;
    sp.cand = (uint32_t*)c.sel_cand.p;
    sp.cand_cap = (uint32_t)std::max<size_t>(k, 16);
    sp.tie = (uint32_t*)c.sel_tie.p;
    sp.tie_cap = kTieCap;
    HIPCK(launch_select(sp, c.stream));
    uint32_t ctr[kSelWords] = {0};
    uint32_t kerr = 0;
    HIPCK(hipMemcpyAsync(ctr, c.sel_ctr.p, sizeof(ctr), hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipMemcpyAsync(&kerr, (uint32_t*)c.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipStreamSynchronize(c.stream));
    if ((rc = score_pass_finish_locked(db, Q, kerr))) return rc;
    if (ctr[kSelTie] <= kTieCap && ctr[kSelCand] <= sp.cand_cap && ctr[kSelRerun] <= n_slots) {
      std::vector<uint32_t> cand(2 * (size_t)ctr[kSelCand]), tie(2 * (size_t)ctr[kSelTie]), rr(ctr[kSelRerun]);
      if (!cand.empty()) HIPCK(hipMemcpyAsync(cand.data(), c.sel_cand.p, cand.size() * 4, hipMemcpyDeviceToHost, c.stream));
      if (!tie.empty()) HIPCK(hipMemcpyAsync(tie.data(), c.sel_tie.p, tie.size() * 4, hipMemcpyDeviceToHost, c.stream));
      if (!rr.empty()) HIPCK(hipMemcpyAsync(rr.data(), c.sel_rerun.p, rr.size() * 4, hipMemcpyDeviceToHost, c.stream));
      HIPCK(hipStreamSynchronize(c.stream));
      for (size_t i = 0; i < cand.size(); i += 2) pool.push_back(TopEntry{(int32_t)cand[i], cand[i + 1]});
      for (size_t i = 0; i < tie.size(); i += 2) pool.push_back(TopEntry{(int32_t)tie[i], tie[i + 1]});
      std::vector<uint32_t> rerun(db->generic);
      for (uint32_t slot : rr) rerun.push_back(db->slot_target[slot]);
      std::vector<int32_t> rs;
      if ((rc = rescore_locked(db, qp, sc, rerun, rs))) return rc;
      for (size_t i = 0; i < rerun.size(); i++) pool.push_back(TopEntry{rs[i], rerun[i]});
      device_done = true;
    } else {
      // undo the accounting of the pass we are about to repeat through the host path
      c.stats.score_launches -= 1;
      c.stats.score_cells -= (uint64_t)Q * db->packed_residues;
    }
  }
  if (!device_done) {
    std::vector<int32_t> scores;
    int rc = db_scores_locked(db, qp, sc, scores);
    if (rc) return rc;
    pool.resize(db->n);
    for (uint32_t i = 0; i < db->n; i++) pool[i] = TopEntry{scores[i], i};
  }
  const size_t kk = std::min(k, pool.size());
  std::partial_sort(pool.begin(), pool.begin() + kk, pool.end(), top_before);
  top.assign(pool.begin(), pool.begin() + kk);
  return SW_OK;
}

bool prep_query(const char* query, QueryPrep& qp) {
  if (!decode(query, qp.d)) return false;
  const size_t Q = qp.d.sym.size();
  qp.qwords.assign((Q + kScoreB - 1) / kScoreB + 1, 0);  // one word per block of rows, 2 bits per row
  for (size_t i = 0; i < Q; i++) {
    int code = qp.d.cp(i) < 128 ? nt_code((unsigned char)qp.d.cp(i)) : -1;
    if (code < 0) {
      qp.acgt = false;
      code = 0;
    }
    qp.qwords[i / kScoreB] |= (uint32_t)code << (2 * (i % kScoreB));
  }
  return true;
}

}  // namespace

extern "C" {

sw_db* sw_db_create(const char* flat, const uint64_t* offsets, uint64_t n) {
  return swhip::guarded<sw_db*>(nullptr, [&]() -> sw_db* {
  if ((!flat && n) || !offsets) {
    fail(SW_E_ARG, "null argument");
    return nullptr;
  }
  for (uint64_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) {
      fail(SW_E_ARG, "offsets must be non-decreasing");
      return nullptr;
    }
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  if (ctx_init_locked(-1)) return nullptr;
  std::vector<uint64_t> off(n + 1);
  for (uint64_t i = 0; i <= n; i++) off[i] = offsets[i] - offsets[0];
  std::string f(flat + offsets[0], flat + offsets[n]);
  return db_build(std::move(f), std::move(off));
  });
}

sw_db* sw_db_load_fasta(const char* path) {
  // cli/__main__.py:19-34 through the one parser of sw_worker.cpp
  sw_fasta* f = sw_fasta_parse(path);
  if (!f) return nullptr;
  if (sw_fasta_count(f) == 0) {
    sw_fasta_free(f);
    fail(SW_E_ARG, "no FASTA records");
    return nullptr;
  }
  sw_db* db = sw_db_from_fasta(f);
  sw_fasta_free(f);
  return db;
}

void sw_db_free(sw_db* db) {
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  db_release(db);
}
uint64_t sw_db_size(const sw_db* db) { return db ? db->n : 0; }
uint64_t sw_db_residues(const sw_db* db) { return db ? db->residues : 0; }

int sw_db_scores(sw_db* db, const char* query, AlignmentScores as, int32_t* scores_out) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
  if (!db || !query || !scores_out) return fail(SW_E_ARG, "null argument");
  QueryPrep qp;
  if (!prep_query(query, qp)) return fail(SW_E_ARG, "query is not valid UTF-8");
  const Scoring sc = to_scoring(as);
  std::vector<int32_t> scores;
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  int rc = ctx_init_locked(-1);
  if (rc) return rc;
  g_ctx.stats = sw_kernel_stats{};
  rc = db_scores_locked(db, qp, sc, scores);
  if (rc) return rc;
  memcpy(scores_out, scores.data(), scores.size() * sizeof(int32_t));
  return SW_OK;
  });
}

static int db_search_topk_batch(sw_db* db, const char* const* queries, size_t n_queries, AlignmentScores as, int k,
                                sw_hit* hits, int* n_hits) {
  if (!db || (!queries && n_queries) || !hits || !n_hits || k < 0) return fail(SW_E_ARG, "bad argument");
  for (size_t q = 0; q < n_queries; q++) n_hits[q] = 0;
  if (n_queries == 0 || k == 0) return SW_OK;
  std::vector<QueryPrep> qps(n_queries);
  for (size_t q = 0; q < n_queries; q++)
    if (!queries[q] || !prep_query(queries[q], qps[q]))
      return fail(SW_E_ARG, "query " + std::to_string(q) + " is not valid UTF-8");
  const Scoring sc = to_scoring(as);
  std::vector<std::vector<TopEntry>> tops(n_queries);
  std::vector<PairSpec> specs;
  std::vector<Decoded> tdec;
  std::vector<PairOut> po;
  std::vector<uint8_t> arena;
  {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    int rc = ctx_init_locked(-1);
    if (rc) return rc;
    g_ctx.stats = sw_kernel_stats{};
    // score pass + selection per query, then ONE exact submission (end cell + traceback) for all reported hits
    size_t total = 0;
    for (size_t q = 0; q < n_queries; q++) {
      rc = db_topk_locked(db, qps[q], sc, (size_t)k, tops[q]);
      if (rc) return rc;
      total += tops[q].size();
    }
    specs.reserve(total);
    tdec.resize(total);
    size_t at = 0;
    for (size_t q = 0; q < n_queries; q++)
      for (const TopEntry& e : tops[q]) {
        decode(db->flat.data() + db->off[e.target], (size_t)(db->off[e.target + 1] - db->off[e.target]), tdec[at]);
        specs.push_back(PairSpec{qps[q].d.sym.data(), (uint32_t)qps[q].d.sym.size(), tdec[at].sym.data(),
                                 (uint32_t)tdec[at].sym.size(),
                                 db_pair_path((uint32_t)qps[q].d.sym.size(), (uint32_t)tdec[at].sym.size(), sc)});
        at++;
      }
    po.resize(total);
    rc = run_exact(specs.data(), total, sc, true, po.data(), &arena);
    if (rc) return rc;
  }
  size_t at = 0;
  for (size_t q = 0; q < n_queries; q++) {
    for (size_t i = 0; i < tops[q].size(); i++, at++) {
      const TopEntry& e = tops[q][i];
      char *qa = nullptr, *ta = nullptr;
      size_t qlen = 0;
      const bool built = build_alignment(qps[q].d, tdec[at], po[at], arena.data() + po[at].ops_off, &qa, &ta, &qlen);
      if (!built || po[at].score != e.score) {
        str_free(qa);
        str_free(ta);
        for (size_t r = 0; r < n_queries; r++) {
          sw_hits_free(hits + r * (size_t)k, n_hits[r]);
          n_hits[r] = 0;
        }
        if (!built) return fail(SW_E_NOMEM, "out of memory");
        return fail(SW_E_HIP, "internal: score kernel and exact kernel disagree on target " + std::to_string(e.target) +
                                  " (" + std::to_string(e.score) + " vs " + std::to_string(po[at].score) + ")");
      }
      sw_hit& h = hits[q * (size_t)k + i];
      h.target_index = e.target;
      h.score = po[at].score;
      h.length = (uint32_t)qlen;
      h.max_x = (uint64_t)po[at].bx - 1;
      h.max_y = (uint64_t)po[at].bj - 1;
      h.query_aln = qa;
      h.target_aln = ta;
      n_hits[q] = (int)i + 1;
    }
  }
  return SW_OK;
}

int sw_db_search_topk_batch(sw_db* db, const char* const* queries, size_t n_queries, AlignmentScores as, int k,
                            sw_hit* hits, int* n_hits) {
  const int rc = swhip::guarded<int>(SW_E_NOMEM, [&] { return db_search_topk_batch(db, queries, n_queries, as, k, hits, n_hits); });
  if (rc == SW_E_NOMEM && hits && n_hits && k > 0)  // an exception may have left some hits filled: hand over nothing
    for (size_t q = 0; q < n_queries; q++) {
      sw_hits_free(hits + q * (size_t)k, n_hits[q]);
      n_hits[q] = 0;
    }
  return rc;
}

int sw_db_search_topk(sw_db* db, const char* query, AlignmentScores as, int k, sw_hit* hits, int* n_hits) {
  if (!query) return fail(SW_E_ARG, "bad argument");
  return sw_db_search_topk_batch(db, &query, 1, as, k, hits, n_hits);
}

void sw_hits_free(sw_hit* hits, int n_hits) {
  if (!hits) return;
  for (int i = 0; i < n_hits; i++) {
    str_free(hits[i].query_aln);
    str_free(hits[i].target_aln);
    hits[i].query_aln = hits[i].target_aln = nullptr;
  }
}

float sw_hip_benchmark_cups(void) {
  return swhip::guarded<float>(0.0f, [&]() -> float {
  // internal/worker/worker_rust_benchmark.go:38-68 measures one CPU thread on one synthetic pair; a GPU
  // worker runs work packages through the batch path, so that is what is timed here: 1 query (4096 nt) x
  // 16384 targets (1024 nt) of seeded uniform DNA, match 1 / mismatch -1 / gap -2 (the reference's
  // benchmark scores, worker_rust_benchmark.go:10-18).  Returned as an integer-valued float32 because
  // the master's field is a pydantic int (master/api_models/worker.py:7-16).
  const uint32_t nQ = 4096, nT = 1024, n = 16384;
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  const char nt[4] = {'A', 'C', 'G', 'T'};
  std::string flat((size_t)n * nT, 'A'), query(nQ, 'A');
  for (auto& ch : flat) ch = nt[rnd() & 3];
  for (auto& ch : query) ch = nt[rnd() & 3];
  std::vector<uint64_t> off(n + 1);
  for (uint32_t i = 0; i <= n; i++) off[i] = (uint64_t)i * nT;
  sw_db* db = sw_db_create(flat.data(), off.data(), n);
  if (!db) return 0.0f;
  std::vector<int32_t> sc(n);
  AlignmentScores as{(uint16_t)-2, 1, (uint16_t)-1};
  float best = 0.0f;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    if (sw_db_scores(db, query.c_str(), as, sc.data()) != SW_OK) {
      sw_db_free(db);
      return 0.0f;
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float cups = (float)((double)nQ * nT * n / secs);
    best = std::max(best, cups);
  }
  sw_db_free(db);
  return (float)(double)(uint64_t)best;
  });
}

}  // extern "C"
