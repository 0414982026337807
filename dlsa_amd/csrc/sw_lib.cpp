// dlsa_amd/csrc/sw_lib.cpp -- host side of libsw_hip.so: device contexts, the exact-path driver and the
// drop-in C ABI of include/sw.h (+ the batch entry of include/sw_hip.h) over the HIP kernels in sw_exact.hip.
// The resident database lives in sw_db.cpp, the multi-GPU handle in sw_node.cpp.
//
// Mirrors, on the callee side of the cgo boundary, what rust/src/bindings.rs does around the
// algorithm: C string -> UTF-8 check -> symbols (bindings.rs:122-129), run, box a Result with two
// heap strings (bindings.rs:143-151), NULL on any failure (bindings.rs:153-156).  There is no CPU
// implementation of the alignment in this file: every cell is computed by a HIP kernel and a missing
// GPU makes every entry point fail loudly (NULL / SW_E_NOGPU + sw_hip_last_error()).
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "sw_host.h"

using namespace swhip;

// ---- errors --------------------------------------------------------------------------------------------
namespace {
thread_local std::string g_err;
}
int swhip::fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
void swhip::set_last_error(const std::string& msg) { g_err = msg; }
const std::string& swhip::last_error_string() { return g_err; }

// ---- device memory ---------------------------------------------------------------------------------------
int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return SW_OK;
  if (fixed)
    return fail(SW_E_LIMIT, "the allocation-free entry point needs " + std::to_string(bytes) + " bytes of one scratch window, " +
                                std::to_string(cap) + " are reserved: use the allocating entry point for this size");
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
void DevBuf::release() {
  if (p && !fixed) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
int HostBuf::ensure(size_t bytes) {
  if (bytes <= cap) return SW_OK;
  if (fixed)
    return fail(SW_E_LIMIT, "the allocation-free entry point needs " + std::to_string(bytes) + " bytes of host staging, " +
                                std::to_string(cap) + " are reserved");
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess)
    return fail(SW_E_NOMEM, std::string("hipHostMalloc(") + std::to_string(want) + ") failed");
  cap = want;
  return SW_OK;
}
void HostBuf::release() {
  if (p && !fixed) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

// ---- device contexts -------------------------------------------------------------------------------------
// One context per HIP device, created on first use.  The drop-in ABI and the single-database API work on the
// process's DEFAULT device (sw_hip_init, else 0: one process per GPU, as bench.py under torchrun); the node API
// (sw_node.cpp) reaches every device of the machine from one process, each through its own context, lock, stream
// and scratch, so searches on different GPUs never serialise on each other.
// ---- the scratch of the allocation-free entry point ---------------------------------------------------------
// sw_db_scores_noalloc (include/sw_hip.h) may not call hipMalloc: everything its kernels need beyond inputs and
// outputs lives in this file-scope device array.  Every GPU the library runs on holds its own copy at its own
// address until the process ends; `scratch` itself is only a host-side stand-in for the symbol, so the device
// address is looked up once per device when its context is created (ctx_create, that device current) and only that
// pointer is ever used -- as kernel argument, for hipMemsetAsync and for copies.
constexpr size_t kNaQwords = (size_t)64 << 10, kNaTicket = (size_t)4 << 10, kNaProg = (size_t)1 << 20,
                 kNaUnits = (size_t)2 << 20, kNaSlotScores = (size_t)8 << 20, kNaXin = (size_t)4 << 20,
                 kNaXzero = (size_t)6 << 20, kNaXout = (size_t)1 << 20, kNaXrecs = (size_t)256 << 10,
                 kNaRing = (size_t)72 << 20;
constexpr size_t kNaScratchBytes =
    kNaQwords + kNaTicket + kNaProg + kNaUnits + kNaSlotScores + kNaXin + kNaXzero + kNaXout + kNaXrecs + kNaRing;
static __device__ __align__(256) unsigned char scratch[kNaScratchBytes];
constexpr size_t kNaHostBytes = (size_t)24 << 20;  // host arena: slot scores, unit table, symbols, staging

namespace {
std::mutex g_ctx_mu;  // creation of contexts and the default device
Context* g_ctx[kMaxDevices] = {};
int g_default_device = -1;

// aux: an auxiliary context of the device (ctx_get_lane) -- a stream and workspaces of its own, no share of the static
// scratch array (the allocation-free entry point runs on the device's first context only)
int ctx_create(int device, Context** out, bool aux = false) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SW_E_NOGPU, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                                "); libsw_hip has no CPU fallback");
  if (device >= n || device >= kMaxDevices) return fail(SW_E_ARG, "device index out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(SW_E_NOGPU, std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only");
  std::unique_ptr<Context> c(new Context);
  // a failure below must not leave the stream, the events or the host arena behind (a caller may retry)
  struct Undo {
    Context* c;
    ~Undo() {
      if (!c) return;
      for (hipEvent_t e : {c->ev0, c->ev1, c->ev2, c->ev3})
        if (e) (void)hipEventDestroy(e);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      if (c->stream2) (void)hipStreamDestroy(c->stream2);
      if (c->na_host.base) (void)hipHostFree(c->na_host.base);
    }
  } undo{c.get()};
  c->device = device;
  c->cus = prop.multiProcessorCount;
  HIPCK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCK(hipEventCreate(&c->ev0));
  HIPCK(hipEventCreate(&c->ev1));
  HIPCK(hipEventCreate(&c->ev2));
  HIPCK(hipEventCreate(&c->ev3));
  if (const char* mb = getenv("SW_HIP_SCRATCH_MB")) c->scratch_budget = (size_t)atoll(mb) << 20;
  if (aux) {
    for (DevBuf* b : {&c->na.qwords, &c->na.ticket, &c->na.prog, &c->na.units, &c->na.slot_scores, &c->na.xin, &c->na.xzero,
                      &c->na.xout, &c->na.xrecs, &c->na.gbound, &c->na.dirs, &c->na.sel_hist, &c->na.sel_ctr, &c->na.sel_rerun,
                      &c->na.sel_cand, &c->na.sel_tie})
      b->bind_fixed(nullptr, 0);
    c->na.h_in.bind_fixed(nullptr, 0);
    c->na.h_out.bind_fixed(nullptr, 0);
    reset_stats(*c);
    undo.c = nullptr;
    *out = c.release();
    return SW_OK;
  }
  // this device's copy of the scratch array (the device is current: hipSetDevice above)
  void* ptr = nullptr;
  HIPCK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return fail(SW_E_HIP, "hipGetSymbolAddress(scratch) returned a null address");
  c->na_scratch = ptr;
  {
    unsigned char* at = (unsigned char*)ptr;
    auto window = [&](DevBuf& b, size_t bytes) {
      b.bind_fixed(at, bytes);
      at += bytes;
    };
    window(c->na.qwords, kNaQwords);
    window(c->na.ticket, kNaTicket);
    window(c->na.prog, kNaProg);
    window(c->na.units, kNaUnits);
    window(c->na.slot_scores, kNaSlotScores);
    window(c->na.xin, kNaXin);
    window(c->na.xzero, kNaXzero);
    window(c->na.xout, kNaXout);
    window(c->na.xrecs, kNaXrecs);
    window(c->na.gbound, kNaRing);
    // (the traceback and selection buffers stay empty windows: the entry point uses neither)
    for (DevBuf* b : {&c->na.dirs, &c->na.sel_hist, &c->na.sel_ctr, &c->na.sel_rerun, &c->na.sel_cand, &c->na.sel_tie})
      b->bind_fixed(nullptr, 0);
  }
  // page-locked, so that the entry point's copies are truly asynchronous and the runtime has nothing to stage
  void* arena = nullptr;
  if (hipHostMalloc(&arena, kNaHostBytes, hipHostMallocPortable) != hipSuccess || !arena)
    return fail(SW_E_NOMEM, "out of page-locked host memory");
  c->na_host.base = (unsigned char*)arena;
  c->na_host.cap = kNaHostBytes;
  c->na.h_in.bind_fixed(nullptr, 0);
  c->na.h_out.bind_fixed(nullptr, 0);
  reset_stats(*c);
  undo.c = nullptr;
  *out = c.release();
  return SW_OK;
}
}  // namespace

int swhip::ctx_get(int device, Context** out) {
  *out = nullptr;
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (device < 0) {
    if (g_default_device < 0) g_default_device = 0;
    device = g_default_device;
  }
  if (device >= kMaxDevices) return fail(SW_E_ARG, "device index out of range");
  if (!g_ctx[device]) {
    int rc = ctx_create(device, &g_ctx[device]);
    if (rc) return rc;
  }
  *out = g_ctx[device];
  return SW_OK;
}
namespace {
Context* g_ctx_aux[kMaxDevices][kMaxCtxLanes - 1] = {};
}
int swhip::ctx_get_lane(int device, unsigned lane, Context** out) {
  if (lane == 0) return ctx_get(device, out);
  *out = nullptr;
  if (lane >= kMaxCtxLanes) return fail(SW_E_ARG, "context lane out of range");
  Context* first = nullptr;
  const int rc0 = ctx_get(device, &first);  // (resolves a negative device, and makes sure the device is usable)
  if (rc0) return rc0;
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  Context*& slot = g_ctx_aux[first->device][lane - 1];
  if (!slot) {
    const int rc = ctx_create(first->device, &slot, true);
    if (rc) return rc;
  }
  slot->scratch_budget = first->scratch_budget;  // (follows sw_hip_set_scratch_mb; read when a job starts)
  *out = slot;
  return SW_OK;
}
void swhip::reset_stats(Context& c) {
  c.stats = sw_kernel_stats{};
  score_loop_counts(&c.stats.score_packed_per_row, &c.stats.score_aux_per_2rows);
}
size_t swhip::noalloc_scratch_bytes() { return kNaScratchBytes; }
swhip::CtxLock::CtxLock(int device) {
  rc = ctx_get(device, &c);
  if (rc) return;
  lk = std::unique_lock<std::mutex>(c->mu);
  if (hipSetDevice(c->device) != hipSuccess) rc = fail(SW_E_HIP, "hipSetDevice failed");
}
swhip::CtxLock::CtxLock(Context* ctx) : c(ctx) {
  if (!c) {
    rc = fail(SW_E_ARG, "no device context");
    return;
  }
  lk = std::unique_lock<std::mutex>(c->mu);
  if (hipSetDevice(c->device) != hipSuccess) rc = fail(SW_E_HIP, "hipSetDevice failed");
}
void swhip::Workspace::release() {
  for (DevBuf* b : {&xin, &xzero, &xout, &dirs, &xrecs, &sel_hist, &sel_ctr, &sel_rerun, &sel_cand, &sel_tie, &qwords,
                    &slot_scores, &gbound, &ticket, &prog, &units, &strip_best})
    b->release();
  h_in.release();
  h_out.release();
}

namespace {
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

}  // namespace

// ---- the four integers that make one plain SW reproduce a reference entry point ---------------------------
// DESIGN.md "Effective semantics"; proven equal to the restatement by tests/test_oracle_clean.py.
// budget of a hand-over wait of the exact kernels in s_memrealtime ticks (100 MHz): 4 s, or SW_HIP_EXACT_WAIT_MS
// (read at every submission, so that a test can shorten it)
unsigned long long swhip::exact_wait_ticks() {
  const char* ms = getenv("SW_HIP_EXACT_WAIT_MS");
  return (unsigned long long)(ms && atoll(ms) > 0 ? atoll(ms) : 4000) * 100000ull;
}

PathParams swhip::resolve_path(int path, uint32_t Q, uint32_t T) {
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


namespace {
// One submission of pairs [begin, end), in two halves so that a driver can keep two in flight (a work package:
// the next sub-batch is staged and queued while this one is on the GPU).  Caller holds the context mutex.
struct ChunkJob {
  size_t begin = 0, n = 0, out_ops = 0, ops_bytes = 0, arena_base = 0, err_at = 0;
  bool want_trace = false, one_copy = false, full_copy = false;
  uint8_t* d_out = nullptr;
  const PairSpec* specs = nullptr;
  OpsArena* arena = nullptr;
};
int ws_events(Workspace& ws) {
  for (hipEvent_t* e : {&ws.ev_k0, &ws.ev_k1, &ws.ev_done})
    if (!*e) HIPCK(hipEventCreate(e));
  return SW_OK;
}
// Stages and queues everything, including the copies back; returns without waiting.  full_copy: the whole ops
// region (capacity Q + T per pair) is copied straight into the arena instead of only the bytes the walkers wrote,
// which needs no second round trip -- right for many short pairs.  On success the work is in flight and the caller
// must call exact_chunk_end (or drain the stream) before touching ws or the arena again.
int exact_chunk_begin(Context& c, Workspace& ws, const PairSpec* specs, size_t begin, size_t end, Scoring sc,
                      bool want_trace, OpsArena* arena, int rpl, bool full_copy, ChunkJob& job) {
  const size_t n = end - begin;
  job = ChunkJob{};
  job.begin = begin;
  job.n = n;
  if (n == 0) return SW_OK;
  StageTimer tm(n);
  int erc = ws_events(ws);
  if (erc) return erc;
  // symbol pool: every distinct sequence once (a database search shares one query)
  std::unordered_map<const uint16_t*, size_t> seen;  // pointer -> index into copies
  std::vector<uint64_t>& id_copy = ws.xh.id_copy;    // id -> (epoch, index into copies), kept between submissions
  if (++ws.xh.epoch == 0) {                          // (wrapped: every stamp is stale and could look current)
    std::fill(id_copy.begin(), id_copy.end(), 0);
    ws.xh.epoch = 1;
  }
  const uint64_t epoch_tag = (uint64_t)ws.xh.epoch << 32;
  struct SeqCopy {
    const uint16_t* p;
    uint32_t len;
    size_t off;
  };
  std::vector<SeqCopy> copies;  // distinct sequences, copied straight into the pinned staging buffer below
  size_t pool_syms = 0;
  // (a cached copy is reused only for the SAME length: windows of one pair name prefixes of one query and
  // different stretches of one target)
  auto add_seq = [&](const uint16_t* p, uint32_t len, uint32_t id) -> size_t {
    if (id != UINT32_MAX) {
      if (id >= id_copy.size()) id_copy.resize((size_t)id + 1 + id_copy.size() / 2, 0);
      const uint64_t e = id_copy[id];
      if ((e >> 32) == ws.xh.epoch) {
        const SeqCopy& sc_ = copies[(uint32_t)e];
        if (sc_.len == len && sc_.p == p) return sc_.off;
      }
    } else {
      auto it = seen.find(p);
      if (it != seen.end() && copies[it->second].len == len) return copies[it->second].off;
    }
    const size_t off = pool_syms;
    copies.push_back(SeqCopy{p, len, off});
    pool_syms += (len + 1u) & ~(size_t)1;
    if (id != UINT32_MAX) id_copy[id] = epoch_tag | (uint64_t)(copies.size() - 1);
    else seen[p] = copies.size() - 1;
    return off;
  };
  struct Trim {
    ExactHost& h;
    ~Trim() { h.trim(); }
  } trim{ws.xh};
  auto& hp = ws.xh.hp;
  auto& qoff = ws.xh.qoff;
  auto& toff = ws.xh.toff;
  auto& dirw = ws.xh.dirw;
  auto& unit_pair = ws.xh.unit_pair;
  auto& unit_order = ws.xh.unit_order;
  hp.resize(n);
  qoff.resize(n);
  toff.resize(n);
  dirw.resize(want_trace ? n : 0);
  unit_pair.clear();
  unit_order.clear();
  size_t dir_words = 0, bnd_ints = 0, ops_bytes = 0;
  unit_pair.reserve(n);
  copies.reserve(n + 1);
  tm.lap("  chunk: alloc");
  // per pair, independent of every other pair: on the host threads when the submission is a whole work package
  parallel_ranges(n, n * 256, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      const PairSpec& s = specs[begin + i];
      PathParams pp = resolve_path(s.path, s.Qfull ? s.Qfull : s.Q, s.Tfull ? s.Tfull : s.T);
      if (s.Tfull) {  // a window: the column limits are global, the kernel counts columns from col0
        pp.Tlim = pp.Tlim > s.col0 ? std::min(pp.Tlim - s.col0, s.T) : 0;
        pp.T0 = pp.T0 > s.col0 ? std::min(pp.T0 - s.col0, s.T) : 0;
      }
      ExactPair& e = hp[i];
      e.Q = s.Q;
      e.T = s.T;
      e.left = s.left;
      e.left_shift = s.left_shift;
      e.col0 = s.col0;
      e.Tfull = s.Tfull;
      e.walk_x = s.walk_x;
      e.walk_j = s.walk_j;
      e.walk_H = s.walk_H;
      e.Tlim = pp.Tlim;
      e.T0 = pp.T0;
      e.phaseQp = pp.phaseQp;
      e.straight = pp.straight;
      const uint32_t maxcol = std::max(pp.Tlim, pp.T0);
      e.bnd_pitch = exact_bnd_pitch(maxcol);  // (rows of whole 64-byte lines, sw_internal.h)
      e.nkb = exact_nsteps(maxcol, s.Q, rpl) / 16;
      e.n_units = std::max<uint32_t>(exact_nstrips(s.Q, rpl), 1);
      if (want_trace) dirw[i] = exact_dir_words(s.Q, maxcol, rpl);
    }
  });
  tm.lap("  chunk: per pair");
  // what depends on the pairs before: the symbol pool, the unit numbering and the offsets into the shared buffers
  for (size_t i = 0; i < n; i++) {
    const PairSpec& s = specs[begin + i];
    qoff[i] = add_seq(s.q, s.Q, s.qid);
    toff[i] = add_seq(s.t, s.T, s.tid);
    ExactPair& e = hp[i];
    if (unit_pair.size() + e.n_units > 0x7fffff00ull) return fail(SW_E_ARG, "too many strips in one submission");
    // the exact kernel keeps 4 * H in an int32 whose top bit is its hand-over tag (sw_exact.hip); a window starts
    // from int16 boundary values
    const uint64_t step_up = (uint64_t)std::max(std::max((int)sc.match, (int)sc.miss), std::max((int)sc.gap, 1));
    if (((uint64_t)s.Q + s.T) * step_up + 65536 >= (1ull << 29)) return fail(SW_E_ARG, "pair too long for this scoring: (Q + T) * largest score must stay below 2^29");
    e.first_unit = (uint32_t)unit_pair.size();
    unit_pair.insert(unit_pair.end(), e.n_units, (uint32_t)i);
    // temporarily store offsets in the pointer fields
    e.bnd = (int32_t*)(uintptr_t)bnd_ints;
    bnd_ints += (size_t)e.n_units * e.bnd_pitch;
    if (want_trace) {
      e.dirs = (uint32_t*)(uintptr_t)(dir_words + 1);  // +1 so that offset 0 is not nullptr
      dir_words += dirw[i];
      e.ops = (uint8_t*)(uintptr_t)ops_bytes;
      ops_bytes += ((size_t)s.Q + s.T + 15) / 16 * 16;
    } else {
      e.dirs = nullptr;
      e.ops = nullptr;
    }
  }
  const size_t n_units = unit_pair.size();
  tm.lap("  chunk: offsets");
  // Ticket order: strip level by level (all pairs' strip 0, then every strip 1, ...; longest pairs first within a
  // level), so that in a big submission the strip above has long finished when a strip starts, instead of the
  // strips of one pair trailing each other through HBM with adjacent tickets.  The strip above always has an
  // EARLIER ticket, which is all the kernel needs.  unit_order[ticket] = record index (first_unit + strip).
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
  const size_t z_bnd = (ctl_words * 4 + 63) & ~(size_t)63, z_bytes = z_bnd + align16(std::max<size_t>(bnd_ints * 4, 16));
  const size_t out_ops = align16(n * sizeof(ExactOut)), out_bytes = out_ops + (want_trace ? align16(ops_bytes) : 0);
  int rc;
  if ((rc = ws.xin.ensure(in_bytes))) return rc;
  if ((rc = ws.xzero.ensure(z_bytes))) return rc;
  if ((rc = ws.xout.ensure(out_bytes))) return rc;
  if ((rc = ws.xrecs.ensure(n_units * sizeof(ExactRec)))) return rc;
  if (want_trace && (rc = ws.dirs.ensure(std::max<size_t>(dir_words * 4, 16)))) return rc;
  if ((rc = ws.h_in.ensure(in_bytes))) return rc;
  uint8_t* const d_in = (uint8_t*)ws.xin.p;
  uint8_t* const d_zero = (uint8_t*)ws.xzero.p;
  uint8_t* const d_out = (uint8_t*)ws.xout.p;
  for (size_t i = 0; i < n; i++) {
    ExactPair& e = hp[i];
    e.q = (const uint16_t*)(d_in + in_sym) + qoff[i];
    e.t = (const uint16_t*)(d_in + in_sym) + toff[i];
    e.bnd = (int32_t*)(d_zero + z_bnd) + (uintptr_t)e.bnd;
    if (want_trace) {
      e.dirs = (uint32_t*)ws.dirs.p + ((uintptr_t)e.dirs - 1);
      e.ops = d_out + out_ops + (uintptr_t)e.ops;
    }
  }
  uint8_t* const h_in = (uint8_t*)ws.h_in.p;
  memcpy(h_in + in_pairs, hp.data(), n * sizeof(ExactPair));
  memcpy(h_in + in_units, unit_pair.data(), n_units * 4);
  memcpy(h_in + in_order, unit_order.data(), n_units * 4);
  parallel_ranges(copies.size(), pool_syms * 2, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++)
      memcpy(h_in + in_sym + copies[k].off * 2, copies[k].p, (size_t)copies[k].len * 2);
  });
  tm.lap("  chunk: stage");
  // any error return from here on leaves work queued that reads / writes the staging buffers: drain it first
  StreamDrain drain(c.stream);
  HIPCK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, c.stream));
  // (strip boundary rows: a word is "published" when its bit 31 is set (sw_exact.hip), so they start at zero)
  HIPCK(hipMemsetAsync(d_zero, 0, z_bytes, c.stream));
  HIPCK(hipEventRecord(ws.ev_k0, c.stream));
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
  w.recs = (ExactRec*)ws.xrecs.p;
  w.wait_ticks = exact_wait_ticks();
  {  // (test hook, read at every submission)
    const char* drop = getenv("SW_HIP_DEBUG_DROP_HANDOVER");
    w.debug_drop = drop && atoi(drop) ? 1u : 0u;
  }
  HIPCK(launch_exact(w, sc, c.stream));
  if (want_trace) HIPCK(launch_walk(w.pairs, w.outs, (uint32_t)n, rpl, sc, c.stream));
  HIPCK(hipEventRecord(ws.ev_k1, c.stream));
  // small submissions fetch results and ops in ONE copy (the ops capacity is Q + T per pair); big ones fetch the
  // results first and then only the op bytes the walkers wrote (or, full_copy, the whole region right away)
  full_copy = full_copy && want_trace && ops_bytes;
  const bool one_copy = !full_copy && want_trace && out_bytes <= ((size_t)4 << 20);
  const size_t first_bytes = one_copy ? out_bytes : n * sizeof(ExactOut);
  const size_t err_at = (first_bytes + 15) & ~(size_t)15;  // the kernels' error word travels with the results
  if ((rc = ws.h_out.ensure(err_at + 16))) return rc;
  HIPCK(hipMemcpyAsync(ws.h_out.p, d_out, first_bytes, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipMemcpyAsync((uint8_t*)ws.h_out.p + err_at, w.err, 4, hipMemcpyDeviceToHost, c.stream));
  job.err_at = err_at;
  if (full_copy) {  // (the arena is page-locked memory: the copy lands in place)
    job.arena_base = arena->size();
    arena->resize(job.arena_base + ops_bytes);
    HIPCK(hipMemcpyAsync(arena->data() + job.arena_base, d_out + out_ops, ops_bytes, hipMemcpyDeviceToHost, c.stream));
  }
  HIPCK(hipEventRecord(ws.ev_done, c.stream));
  drain.dismiss();
  job.out_ops = out_ops;
  job.ops_bytes = ops_bytes;
  job.want_trace = want_trace;
  job.one_copy = one_copy;
  job.full_copy = full_copy;
  job.d_out = d_out;
  job.specs = specs;
  job.arena = arena;
  tm.lap("  chunk: queue");
  return SW_OK;
}

// Waits for the submission and turns its records into PairOuts.
int exact_chunk_end(Context& c, Workspace& ws, ChunkJob& job, PairOut* outs) {
  const size_t n = job.n, begin = job.begin;
  if (n == 0) return SW_OK;
  StageTimer tm(n);
  StreamDrain drain(c.stream);  // (an error below must not leave this submission's copies in flight)
  HIPCK(hipEventSynchronize(ws.ev_done));
  const bool want_trace = job.want_trace;
  const size_t out_ops = job.out_ops, ops_bytes = job.ops_bytes;
  uint8_t* const d_out = job.d_out;
  OpsArena* const arena = job.arena;
  const PairSpec* const specs = job.specs;
  auto& hp = ws.xh.hp;
  const ExactOut* ho = (const ExactOut*)ws.h_out.p;
  tm.lap("  chunk: gpu");
  // a hand-over wait ran out: the waiting waves ended without results (sw_exact.hip), nothing of this submission
  // may be used
  if (*(const uint32_t*)((const uint8_t*)ws.h_out.p + job.err_at) != 0)
    return fail(SW_E_HIP, "exact kernel: strip pipeline wait timed out");
  for (size_t i = 0; i < n; i++)
    if (ho[i].nops == 0xffffffffu) return fail(SW_E_HIP, "exact kernel: strip pipeline wait timed out");
  float ms = 0;
  HIPCK(hipEventElapsedTime(&ms, ws.ev_k0, ws.ev_k1));
  c.stats.exact_kernel_ms += ms;
  size_t arena_base = job.arena_base;
  if (want_trace && ops_bytes && !job.full_copy) {
    arena_base = arena->size();
    arena->resize(arena_base + ops_bytes);
    uint8_t* hops = arena->data() + arena_base;
    if (job.one_copy) {
      for (size_t i = 0; i < n; i++) {
        const size_t s0 = (size_t)(hp[i].ops - (d_out + out_ops));
        if (ho[i].nops) memcpy(hops + s0, (const uint8_t*)ws.h_out.p + out_ops + s0, ho[i].nops);
      }
    } else {
      // the op bytes the walkers wrote: ONE copy of the span they cover (the arena is page-locked memory: the copy
      // lands in place)
      size_t span_lo = SIZE_MAX, span_hi = 0;
      for (size_t i = 0; i < n; i++)
        if (ho[i].nops) {
          const size_t s0 = (size_t)(hp[i].ops - (d_out + out_ops));
          span_lo = std::min(span_lo, s0);
          span_hi = std::max(span_hi, s0 + ho[i].nops);
        }
      if (span_hi > span_lo) {
        HIPCK(hipMemcpyAsync(hops + span_lo, d_out + out_ops + span_lo, span_hi - span_lo, hipMemcpyDeviceToHost, c.stream));
        HIPCK(hipStreamSynchronize(c.stream));
      }
    }
  }
  for (size_t i = 0; i < n; i++) {
    PairOut& o = outs[begin + i];
    o.score = ho[i].score;
    o.bx = ho[i].bx;
    o.bj = ho[i].bj;
    o.end_x = ho[i].end_x;
    o.end_j = ho[i].end_j;
    o.end_H = ho[i].end_H;
    o.key = ((uint64_t)ho[i].key_hi << 32) | ho[i].key_lo;
    c.stats.exact_cells += (uint64_t)specs[begin + i].Q * specs[begin + i].T;
    if (want_trace && ho[i].nops) {
      o.ops_off = arena_base + (size_t)(hp[i].ops - (d_out + out_ops));
      o.nops = ho[i].nops;
    }
  }
  drain.dismiss();  // nothing of this submission is queued any more (another one may be: not ours to wait for)
  tm.lap("  chunk: results");
  return SW_OK;
}

int run_exact_chunk(Context& c, Workspace& ws, const PairSpec* specs, size_t begin, size_t end, Scoring sc,
                    bool want_trace, PairOut* outs, OpsArena* arena, int rpl) {
  ChunkJob job;
  const int rc = exact_chunk_begin(c, ws, specs, begin, end, sc, want_trace, arena, rpl, false, job);
  if (rc) return rc;
  return exact_chunk_end(c, ws, job, outs);
}

// bytes of device scratch one pair needs (the split of a submission into chunks goes by it)
size_t exact_pair_bytes(const PairSpec& sp, int rpl, bool want_trace) {
  const PathParams pp = resolve_path(sp.path, sp.Q, sp.T);
  size_t b = (size_t)sp.Q * 2 + (size_t)sp.T * 2 + (size_t)std::max<uint32_t>(exact_nstrips(sp.Q, rpl), 1) * exact_bnd_pitch(sp.T + 1) * 4 + 256;
  if (want_trace) b += exact_dir_words(sp.Q, std::max(pp.Tlim, pp.T0), rpl) * 4 + sp.Q + sp.T;
  return b;
}

// Rows per lane for a submission.  4 does the least work per cell and is used whenever the submission fills the
// GPU anyway (>= one wave per SIMD).  For a few long pairs the strips of a pair form a pipeline in which strip
// s + 1 trails strip s by a fixed lag, so the time is about steps * step time + (strips - 1) * lag; smaller strips
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
  const double tstep[3] = {129, 88, 65}, lag[3] = {14400, 9800, 8800};
  int best = 4;
  double best_t = 0;
  for (int i = 0; i < 3; i++) {
    uint64_t waves = 0;
    double t = 0;
    for (size_t k = 0; k < n; k++) {
      const uint32_t st = std::max<uint32_t>(exact_nstrips(specs[k].Q, rpls[i]), 1);
      waves += st;
      t = std::max(t, (double)exact_nsteps(specs[k].T, specs[k].Q, rpls[i]) * tstep[i] + (st - 1) * lag[i]);
      if (waves > (uint64_t)simds) break;
    }
    if (i > 0 && waves > (uint64_t)simds) break;  // the GPU is full: stay with the larger strips
    if (i == 0 || t < best_t) best = rpls[i], best_t = t;
  }
  return best;
}

}  // namespace

int swhip::run_exact(Context& c, Workspace& ws, const PairSpec* specs, size_t n, Scoring sc, bool want_trace,
                     PairOut* outs, OpsArena* arena) {
  const int rpl = exact_choose_rpl(specs, n, c.cus * 4);
  size_t begin = 0;
  while (begin < n) {
    size_t end = begin, bytes = 0;
    while (end < n) {
      const size_t b = exact_pair_bytes(specs[end], rpl, want_trace);
      if (end > begin && (bytes + b > c.scratch_budget || end - begin >= (1u << 20))) break;
      bytes += b;
      end++;
    }
    int rc = run_exact_chunk(c, ws, specs, begin, end, sc, want_trace, outs, arena, rpl);
    if (rc) return rc;
    begin = end;
  }
  return SW_OK;
}


namespace {
// Sub-batches [cut[b], cut[b + 1]) of one submission, two in flight: while sub-batch b is on the GPU, b + 1 is
// staged and queued behind it (second workspace), and ready(b, rc) -- called in order, once per sub-batch, also
// after a failure -- lets the caller work on b's results while b + 1 runs.  Tracebacks land in arenas[b].  Caller
// holds the context mutex for the whole call.
template <class Ready>
int run_exact_pipelined(Context& c, const PairSpec* specs, const std::vector<size_t>& cut, Scoring sc, PairOut* outs,
                        const std::vector<OpsArena*>& arenas, Ready&& ready) {
  const size_t B = cut.size() - 1;
  const int rpl = exact_choose_rpl(specs, cut[B], c.cus * 4);
  Workspace* wss[2] = {&c.ws, &c.ws2};
  ChunkJob jobs[2];
  StreamDrain drain(c.stream);
  int rc = exact_chunk_begin(c, *wss[0], specs, cut[0], cut[1], sc, true, arenas[0], rpl, true, jobs[0]);
  size_t reported = 0;
  for (size_t b = 0; b < B && !rc; b++) {
    int rc_next = SW_OK;
    if (b + 1 < B)
      rc_next = exact_chunk_begin(c, *wss[(b + 1) & 1], specs, cut[b + 1], cut[b + 2], sc, true, arenas[b + 1], rpl, true,
                                  jobs[(b + 1) & 1]);
    rc = exact_chunk_end(c, *wss[b & 1], jobs[b & 1], outs);
    ready(b, rc);
    reported = b + 1;
    if (!rc) rc = rc_next;
  }
  for (size_t bb = reported; bb < B; bb++) ready(bb, rc);  // (only after a failure)
  return rc;  // (the drain above waits for whatever a failure left queued)
}
}  // namespace

// ---- the pool behind ArenaLease -------------------------------------------------------------------------------
namespace {
constexpr size_t kArenaPoolSlots = 4, kArenaKeepMax = (size_t)256 << 20;
std::mutex g_arena_mu;
// (never destroyed: its blocks are page-locked memory of the HIP runtime, which may be gone at static destruction)
std::vector<OpsArena>& arena_pool() {
  static std::vector<OpsArena>* pool = new std::vector<OpsArena>();
  return *pool;
}
}  // namespace
void swhip::ExactHost::trim() {
  constexpr size_t kKeep = (size_t)1 << 18;  // pairs (~30 MB of tables)
  if (hp.capacity() <= kKeep && unit_pair.capacity() <= 4 * kKeep) return;
  decltype(hp)().swap(hp);
  decltype(qoff)().swap(qoff);
  decltype(toff)().swap(toff);
  decltype(dirw)().swap(dirw);
  std::vector<uint32_t>().swap(unit_pair);
  std::vector<uint32_t>().swap(unit_order);
  std::vector<uint64_t>().swap(id_copy);
}
swhip::ArenaLease::ArenaLease(size_t want) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  std::vector<OpsArena>& pool = arena_pool();
  if (!pool.empty()) {  // the smallest one that is large enough, else the largest: a worker's packages are of a kind
    size_t best = 0;
    auto fits = [&](size_t i) { return want != 0 && pool[i].capacity() >= want; };
    for (size_t i = 1; i < pool.size(); i++) {
      const bool better = fits(best) ? (fits(i) && pool[i].capacity() < pool[best].capacity())
                                     : (fits(i) || pool[i].capacity() > pool[best].capacity());
      if (better) best = i;
    }
    a.swap(pool[best]);
    pool.erase(pool.begin() + (std::ptrdiff_t)best);
  }
}
swhip::ArenaLease::~ArenaLease() {
  a.clear();
  if (a.capacity() == 0 || a.capacity() > kArenaKeepMax) return;
  try {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    std::vector<OpsArena>& pool = arena_pool();
    if (pool.size() < kArenaPoolSlots) pool.emplace_back(std::move(a));
  } catch (...) {
  }
}

namespace {
// ---- Result boxes ------------------------------------------------------------------------------------------
// A batch (sw_align_batch / sw_execute_work: 10^5 results at a time) allocates ONE block, a slab, for all its boxes
// and strings instead of three heap blocks per result; the slab counts its live parts (a box and each of its two
// strings) and is freed with the last one, so free_alignment_result / free_c_string keep their one-call-per-object
// contract.  A result of the one-pair entry points has no slab.
struct Slab {
  std::atomic<uint64_t> refs;
  size_t capacity;  // bytes of the block
};
struct ResultBox {
  Result r;
  int32_t score32;
  uint32_t magic;
  Slab* slab;  // nullptr: box and strings are heap blocks of their own
};
constexpr uint32_t kMagic = 0x53574850u;  // "SWHP"
// A worker produces one slab per work package and frees it before the next one arrives; a block of tens of
// megabytes goes back to the kernel on free() and comes back as fresh zero pages (one page fault per 4 KiB while the
// strings are written).  The last released slabs are therefore kept for the next batch of a similar size: at most
// kSlabCacheSlots blocks of at most kSlabCacheMax bytes each and kSlabCacheTotal bytes together stay allocated
// (SW_HIP_NO_SLAB_CACHE=1: none).  A package of a million read pairs is ~30 slabs of ~6 MB, one per slice of the lane
// path (sw_lanes.cpp): with four slots most of them were page-faulted in again on every package.
constexpr size_t kSlabCacheSlots = 32, kSlabCacheMin = (size_t)1 << 20, kSlabCacheMax = (size_t)256 << 20,
                 kSlabCacheTotal = (size_t)1 << 30;
std::atomic<Slab*> g_slab_cache[kSlabCacheSlots];
std::atomic<size_t> g_slab_cache_bytes{0};
bool slab_cache_enabled() {
  static const bool on = getenv("SW_HIP_NO_SLAB_CACHE") == nullptr;
  return on;
}
Slab* slab_alloc(size_t bytes) {
  if (bytes >= kSlabCacheMin && bytes <= kSlabCacheMax && slab_cache_enabled()) {
    for (auto& slot : g_slab_cache) {
      Slab* s = slot.exchange(nullptr, std::memory_order_acq_rel);
      if (!s) continue;
      if (s->capacity >= bytes && s->capacity / 2 <= bytes) {
        g_slab_cache_bytes.fetch_sub(s->capacity, std::memory_order_relaxed);
        return s;
      }
      Slab* expected = nullptr;  // not a fit: put it back (or drop it if the slot was taken meanwhile)
      if (!slot.compare_exchange_strong(expected, s, std::memory_order_acq_rel)) {
        g_slab_cache_bytes.fetch_sub(s->capacity, std::memory_order_relaxed);
        free(s);
      }
    }
  }
  const size_t cap = bytes + bytes / 8;  // (a package of the same shape next time fits although its strings differ)
  Slab* s = (Slab*)malloc(cap);
  if (s) s->capacity = cap;
  return s;
}
void slab_free(Slab* s) {
  if (s->capacity >= kSlabCacheMin && s->capacity <= kSlabCacheMax + kSlabCacheMax / 8 && slab_cache_enabled() &&
      g_slab_cache_bytes.load(std::memory_order_relaxed) + s->capacity <= kSlabCacheTotal) {
    g_slab_cache_bytes.fetch_add(s->capacity, std::memory_order_relaxed);  // (the budget may be overshot by a racing free: by one slab)
    for (auto& slot : g_slab_cache) {
      Slab* expected = nullptr;
      if (slot.compare_exchange_strong(expected, s, std::memory_order_acq_rel)) return;
    }
    // all slots taken: replace the first (the oldest habits of the caller change slowly; any policy does)
    Slab* old = g_slab_cache[0].exchange(s, std::memory_order_acq_rel);
    if (old) {
      g_slab_cache_bytes.fetch_sub(old->capacity, std::memory_order_relaxed);
      free(old);
    }
    return;
  }
  free(s);
}
void slab_release(Slab* s, uint64_t n) {
  if (s->refs.fetch_sub(n, std::memory_order_acq_rel) == n) slab_free(s);
}
// Result strings carry a 16-byte header in front of the characters: which Result (if any) still owns them.  That
// is what lets free_c_string() be called on either string of a Result before free_alignment_result()
// (SURVEY.md 8b "Ownership") without a global registry -- a work package returns 10^5 results at a time.
struct StrHdr {
  ResultBox* owner;  // nullptr: a stand-alone string (sw_hit)
  uint64_t magic;
};
constexpr uint64_t kStrMagic = 0x5357485053545231ull;      // "SWHPSTR1": a heap block of its own
constexpr uint64_t kStrMagicSlab = 0x5357485053545232ull;  // "SWHPSTR2": lives in its owner's slab
}  // namespace
char* swhip::str_alloc(size_t n) {
  StrHdr* h = (StrHdr*)malloc(sizeof(StrHdr) + n + 1);
  if (!h) return nullptr;
  h->owner = nullptr;
  h->magic = kStrMagic;
  char* p = (char*)(h + 1);
  p[n] = 0;
  return p;
}
void swhip::str_free(char* p) {
  if (!p) return;
  StrHdr* h = (StrHdr*)p - 1;
  h->magic = 0;
  free(h);
}

namespace {
char* dup_string(const std::string& s) {
  char* p = str_alloc(s.size());
  if (p) memcpy(p, s.data(), s.size());
  return p;
}

}  // namespace
bool swhip::decode(const char* s, size_t n, Decoded& d) {
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
bool swhip::decode(const char* s, Decoded& d) { return s && decode(s, strlen(s), d); }

// Builds the two aligned strings (malloc'ed, NUL-terminated) from the walker's ops (rust/src/algorithm/mod.rs:
// 677-706: the recursion pushes after returning, so the first cell visited is the LAST column).
bool swhip::build_alignment(const Decoded& q, const Decoded& t, const PairOut& o, const uint8_t* ops, char** qa_out,
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

namespace {
// takes ownership of qa / ta (from str_alloc)
void box_fill(ResultBox* b, char* qa, char* ta, int path, const PairOut& o);
ResultBox* box_result(char* qa, char* ta, int path, const PairOut& o) {
  ResultBox* b = (ResultBox*)calloc(1, sizeof(ResultBox));
  if (!b) {
    str_free(qa);
    str_free(ta);
    return nullptr;
  }
  box_fill(b, qa, ta, path, o);
  return b;
}
void box_fill(ResultBox* b, char* qa, char* ta, int path, const PairOut& o) {
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
}
}  // namespace
Scoring swhip::to_scoring(AlignmentScores s) {
  return Scoring{(int)(int16_t)s.match, (int)(int16_t)s.miss, (int)(int16_t)s.gap};
}

// The reference panics (-> NULL) on the ring path when gap == 0 (rust/src/algorithm/mod.rs:236-238).
bool swhip::ring_div_by_zero(int path, uint32_t Q, uint32_t T, Scoring sc) {
  return path == SW_PATH_LOW_MEMORY && !((uint64_t)T < 2ull * Q) && sc.gap == 0;
}


// ---- host threads ---------------------------------------------------------------------------------------------
// ---- host threads ---------------------------------------------------------------------------------------------
// A work package calls parallel_ranges a dozen times (decode, tables, staging, strings, frees); creating 15 threads
// each time cost more than some of the loops.  The helpers are therefore a pool, started on first use and kept for
// the life of the process (never joined: they sleep on a condition variable).  Several callers may be in here at
// once (the two host threads of a pipelined package, the shards of a node); a caller runs tasks itself -- its own
// or anybody's -- until its job is done, so it never depends on a helper being there (e.g. in a forked child).
namespace {
struct PoolJob {
  void (*fn)(void*, size_t, size_t);
  void* arg;
  std::atomic<size_t> remaining{0};
  std::atomic<bool> failed{false};
};
struct PoolTask {
  PoolJob* job;
  size_t lo, hi;
};
struct Pool {
  std::mutex mu;
  std::condition_variable work_cv, done_cv;
  std::vector<PoolTask> tasks;  // (a stack: order does not matter)
  unsigned helpers = 0;
};
Pool& pool() {
  static Pool* p = new Pool();  // never destroyed: helpers may be asleep in it at exit
  return *p;
}
void pool_run(PoolTask t) noexcept {
  try {
    t.job->fn(t.job->arg, t.lo, t.hi);
  } catch (...) {  // an exception inside a helper would end the process: flagged, re-raised by the job's owner
    t.job->failed.store(true);
  }
  if (t.job->remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    std::lock_guard<std::mutex> lk(pool().mu);  // (the owner checks `remaining` under the lock before it sleeps)
    pool().done_cv.notify_all();
  }
}
void pool_helper() noexcept {
  Pool& P = pool();
  std::unique_lock<std::mutex> lk(P.mu);
  for (;;) {
    P.work_cv.wait(lk, [&] { return !P.tasks.empty(); });
    const PoolTask t = P.tasks.back();
    P.tasks.pop_back();
    lk.unlock();
    pool_run(t);
    lk.lock();
  }
}
}  // namespace

void swhip::parallel_ranges_impl(size_t n, unsigned nt, void (*fn)(void*, size_t, size_t), void* arg) {
  nt = std::min<unsigned>(nt, std::max(1u, std::thread::hardware_concurrency()));
  if (n < 2 * (size_t)nt) nt = 1;
  if (nt <= 1) {
    fn(arg, 0, n);
    return;
  }
  Pool& P = pool();
  PoolJob job;
  job.fn = fn;
  job.arg = arg;
  job.remaining.store(nt);
  {
    std::unique_lock<std::mutex> lk(P.mu);
    while (P.helpers + 1 < nt) {  // grow the pool to what this call wants (at most 15 helpers)
      try {
        std::thread(pool_helper).detach();
      } catch (...) {
        break;  // no thread to be had: the caller does the work
      }
      P.helpers++;
    }
    // (room first: a push_back that throws half-way would unwind `job` while queued tasks still point at it)
    P.tasks.reserve(P.tasks.size() + nt);
    for (unsigned t = 1; t < nt; t++) P.tasks.push_back(PoolTask{&job, n * t / nt, n * (t + 1) / nt});
  }
  P.work_cv.notify_all();
  pool_run(PoolTask{&job, 0, n / nt});
  std::unique_lock<std::mutex> lk(P.mu);
  while (job.remaining.load(std::memory_order_acquire) != 0) {
    if (!P.tasks.empty()) {  // help: mine or somebody else's
      const PoolTask t = P.tasks.back();
      P.tasks.pop_back();
      lk.unlock();
      pool_run(t);
      lk.lock();
    } else {
      P.done_cv.wait(lk);
    }
  }
  lk.unlock();
  if (job.failed.load()) throw std::bad_alloc();
}

int swhip::align_many_on(int device, const char* const* queries, const char* const* targets, size_t n,
                         AlignmentScores as, int path, Result** out) {
  if (n == 0) return SW_OK;
  StageTimer tm(n);
  if (!queries || !targets || !out) return fail(SW_E_ARG, "null argument");
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
    qi[i] = intern(queries[i]);
    ti[i] = intern(targets[i]);
  }
  tm.lap("intern");
  return align_interned(device, distinct.data(), distinct.size(), qi.data(), ti.data(), n, as, path, out);
}

namespace {
// The per-pair host tables of one part of a package.  The parts of a package of millions of pairs use the same ones
// one after the other: 80 MB of fresh vectors per part were 10 ms of page faults and constructors each time.
struct PartTables {
  std::vector<PairSpec> specs;
  std::vector<size_t> idx;
  std::vector<uint8_t> valid;
  std::vector<PairOut> po;
};
int align_interned_part(int device, const char* const* distinct_p, size_t n_distinct, const uint32_t* qi, const uint32_t* ti,
                        size_t n, AlignmentScores as, int path, Result** out, PartTables& tables);
std::mutex g_tables_mu;
PartTables g_tables_kept;
}  // namespace

int swhip::align_interned(int device, const char* const* distinct_p, size_t n_distinct, const uint32_t* qi,
                          const uint32_t* ti, size_t n, AlignmentScores as, int path, Result** out) {
  if (n == 0) return SW_OK;
  if (!distinct_p || !qi || !ti || !out) return fail(SW_E_ARG, "null argument");
  if (path < 0 || path > 2) return fail(SW_E_ARG, "bad path");
  // A package of millions of pairs (C5's 1 000 x 10 000 reads) goes through in parts of 2^19 pairs: host tables,
  // the page-locked traceback arena and the device scratch stay what half a million pairs need, whatever n is.
  constexpr size_t kPart = (size_t)1 << 19;
  // (... and a worker's next package takes the tables of the last one: one set is kept, like the arenas)
  struct TablesLease {
    PartTables t;
    bool big;
    explicit TablesLease(bool big_) : big(big_) {
      if (!big) return;
      std::lock_guard<std::mutex> lk(g_tables_mu);
      std::swap(t, g_tables_kept);
    }
    ~TablesLease() {
      if (!big || t.specs.capacity() > kPart) return;
      std::lock_guard<std::mutex> lk(g_tables_mu);
      if (t.specs.capacity() > g_tables_kept.specs.capacity()) std::swap(t, g_tables_kept);
    }
  } lease(n >= 32768);
  for (size_t lo = 0; lo < n; lo += kPart) {
    const int rc = align_interned_part(device, distinct_p, n_distinct, qi + lo, ti + lo, std::min(kPart, n - lo), as, path,
                                       out + lo, lease.t);
    if (rc) return rc;  // (the caller frees what earlier parts handed over)
  }
  return SW_OK;
}

namespace {
int align_interned_part(int device, const char* const* distinct_p, size_t n_distinct, const uint32_t* qi, const uint32_t* ti,
                        size_t n, AlignmentScores as, int path, Result** out, PartTables& tables) {
  StageTimer tm(n);
  const Scoring sc = to_scoring(as);
  // only the sequences some pair names are decoded (a package may carry more)
  std::vector<uint8_t> used(n_distinct, 0);
  for (size_t i = 0; i < n; i++) {
    out[i] = nullptr;
    if (qi[i] != UINT32_MAX) {
      if (qi[i] >= n_distinct) return fail(SW_E_ARG, "sequence index out of range");
      used[qi[i]] = 1;
    }
    if (ti[i] != UINT32_MAX) {
      if (ti[i] >= n_distinct) return fail(SW_E_ARG, "sequence index out of range");
      used[ti[i]] = 1;
    }
  }
  tm.lap("  decode: used");
  std::vector<Decoded> dec(n_distinct);
  std::vector<uint8_t> ok(n_distinct, 0);
  std::vector<size_t> lens(n_distinct, 0);
  size_t total_bytes = 0;
  std::vector<size_t> sym_at(n_distinct, 0);
  tm.lap("  decode: tables");
  parallel_ranges(n_distinct, n_distinct * 256, [&](size_t lo, size_t hi) {  // (a package of reads: 10^5 strlen calls)
    for (size_t d = lo; d < hi; d++)
      if (used[d] && distinct_p[d]) lens[d] = strlen(distinct_p[d]);
  });
  size_t longest = 0;
  for (size_t d = 0; d < n_distinct; d++)
    if (used[d] && distinct_p[d]) {
      sym_at[d] = total_bytes;
      total_bytes += (lens[d] + 7) & ~(size_t)7;  // (every sequence at a multiple of 8 symbols: SymPool)
      longest = std::max(longest, lens[d]);
    }
  tm.lap("  decode: strlen");
  // One pool for the symbols of all sequences (a sequence has at most as many symbols as bytes).  A batch big enough
  // for the lane kernel decodes into page-locked memory, which that path uploads as it is.
  std::vector<uint16_t, NoInitAlloc<uint16_t>> sym_plain;
  std::unique_ptr<ArenaLease> sym_lease;
  uint16_t* sym_base = nullptr;
  // (only a batch the lane kernel can take as a whole: its pairs have Q + T < 65 000 and the block must stay a
  // block that ArenaLease keeps -- the long targets of a package for the fast path would be 170 MB of it)
  if (n >= 1024 && path != SW_PATH_STRAIGHT && longest < 65000 && total_bytes <= ((size_t)64 << 20)) {
    sym_lease.reset(new ArenaLease((total_bytes + 8) * 2));
    sym_lease->a.resize((total_bytes + 8) * 2);
    sym_base = (uint16_t*)sym_lease->a.data();
  } else {
    sym_plain.resize(total_bytes + 8);
    sym_base = sym_plain.data();
  }
  const SymPool sym_pool_desc{sym_lease ? sym_base : nullptr, total_bytes};
  for (size_t d = 0; d < n_distinct; d++) dec[d].sym.bind(sym_base + sym_at[d]);
  tm.lap("  decode: pool");
  parallel_ranges(n_distinct, total_bytes, [&](size_t lo, size_t hi) {
    for (size_t d = lo; d < hi; d++)  // bindings.rs:125 unwrap: invalid UTF-8 -> NULL
      ok[d] = used[d] && distinct_p[d] && decode(distinct_p[d], lens[d], dec[d]) && dec[d].sym.size() < 0x7fffff00u;
  });
  tm.lap("decode");
  // (filled in place on the host threads; pairs the reference would fail on are squeezed out afterwards -- rare)
  // (a table left by the previous part of the same package is reused as it is: every entry that counts is rewritten)
  std::vector<PairSpec>& specs = tables.specs;
  std::vector<size_t>& idx = tables.idx;
  std::vector<uint8_t>& valid = tables.valid;  // 0: the pair fails as the reference's would; 1: valid; 3: valid and both ASCII
  specs.resize(n);
  idx.resize(n);
  valid.resize(n);
  parallel_ranges(n, n * 64, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      idx[i] = i;
      valid[i] = 0;
      if (qi[i] == UINT32_MAX || ti[i] == UINT32_MAX || !ok[qi[i]] || !ok[ti[i]]) continue;
      const Decoded &q = dec[qi[i]], &t = dec[ti[i]];
      if (ring_div_by_zero(path, (uint32_t)q.sym.size(), (uint32_t)t.sym.size(), sc)) continue;
      specs[i] = PairSpec{q.sym.data(), (uint32_t)q.sym.size(), t.sym.data(), (uint32_t)t.sym.size(), path, qi[i], ti[i]};
      valid[i] = (q.ascii && t.ascii) ? 3 : 1;
    }
  });
  {
    size_t w = 0;
    for (size_t i = 0; i < n; i++)
      if (valid[i]) {
        if (w != i) specs[w] = specs[i], idx[w] = i, valid[w] = valid[i];
        w++;
      }
    specs.resize(w);
    idx.resize(w);
    valid.resize(w);
  }
  std::vector<PairOut>& po = tables.po;
  po.resize(specs.size());
  parallel_ranges(po.size(), po.size() * 64, [&](size_t lo, size_t hi) { std::fill(po.begin() + (ptrdiff_t)lo, po.begin() + (ptrdiff_t)hi, PairOut{}); });
  ArenaLease lease;
  OpsArena& arena = lease.a;
  tm.lap("specs");
  // The work-package fast path (sw_package.cpp) takes the pairs it is eligible for: long ACGT targets on the ring
  // path, filled by the packed score kernel and traced inside windows.  Same results, by construction and by test.
  std::vector<uint8_t> handled(specs.size(), 0);
  // (its shortest eligible target has kWindowsMinTarget symbols: a package of reads does not even build the tables)
  if (path == SW_PATH_LOW_MEMORY && !specs.empty() && longest >= kWindowsMinTarget) {
    std::vector<uint32_t> pq(specs.size()), pt(specs.size());
    for (size_t k = 0; k < specs.size(); k++) pq[k] = qi[idx[k]], pt[k] = ti[idx[k]];
    const int rc = align_windows(dec, pq, pt, sc, po, arena, handled, device);
    if (rc) return rc;
    tm.lap("windows");
  } else {
    g_package_stats = PackageStats{};
  }
  // Results of pairs [k_lo, k_hi) as ONE slab: [Slab | ResultBox x pairs | per pair: StrHdr + query alignment + NUL,
  // StrHdr + target alignment + NUL].  ASCII pairs (the normal case) know their string lengths up front (= the
  // number of traceback ops); a pair with non-ASCII symbols re-encodes to UTF-8 and keeps heap strings of its own.
  // (ks: the pairs as a list of indices into specs; nullptr: the range [k_lo, k_hi))
  std::vector<uint8_t> ascii_of(valid.size());  // per pair: both sequences ASCII (valid[k] bit 1)
  for (size_t k = 0; k < valid.size(); k++) ascii_of[k] = valid[k] >> 1;
  auto pair_ascii = [&](size_t k) -> bool { return ascii_of[k] != 0; };
  auto emit_list = [&](const uint32_t* ks, size_t k_lo, size_t m, const OpsArena& ar) -> int {
    if (m == 0) return SW_OK;
    auto k_of = [&](size_t kk) -> size_t { return ks ? ks[kk] : k_lo + kk; };
    auto round8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    std::vector<size_t> str_off(m + 1, 0);
    const size_t boxes_off = round8(sizeof(Slab));
    size_t bytes = boxes_off + m * sizeof(ResultBox);
    for (size_t kk = 0; kk < m; kk++) {
      const size_t k = k_of(kk);
      str_off[kk] = bytes;
      // (valid[k] bit 1: both sequences ASCII -- set with the pair's spec; not the three lookups through dec[])
      if (pair_ascii(k)) bytes += 2 * (sizeof(StrHdr) + round8((size_t)po[k].nops + 1));
    }
    str_off[m] = bytes;
    Slab* slab = slab_alloc(bytes);
    if (!slab) return fail(SW_E_NOMEM, "out of host memory");
    new (&slab->refs) std::atomic<uint64_t>(0);
    ResultBox* boxes = (ResultBox*)((char*)slab + boxes_off);
    std::atomic<uint64_t> live{0};
    parallel_ranges(m, ar.size() * 4 + m * 64, [&](size_t lo, size_t hi) {
      uint64_t mine = 0;
      for (size_t kk = lo; kk < hi; kk++) {
        const size_t k = k_of(kk), i = idx[k];
        ResultBox* b = boxes + kk;
        memset(b, 0, sizeof(ResultBox));
        const PairOut& o = po[k];
        const uint8_t* ops = ar.data() + o.ops_off;
        if (o.strings) {  // the lane kernel wrote the strings themselves (sw_lanes.hip; ASCII pairs only)
          const size_t each = sizeof(StrHdr) + round8((size_t)o.nops + 1);
          StrHdr* hq = (StrHdr*)((char*)slab + str_off[kk]);
          StrHdr* ht = (StrHdr*)((char*)slab + str_off[kk] + each);
          hq->magic = ht->magic = kStrMagicSlab;
          char *qa = (char*)(hq + 1), *ta = (char*)(ht + 1);
          memcpy(qa, ops, o.nops);
          memcpy(ta, ops + (((size_t)o.nops + 3) & ~(size_t)3), o.nops);
          qa[o.nops] = ta[o.nops] = 0;
          b->slab = slab;
          box_fill(b, qa, ta, path, o);
          mine += 3;
          out[i] = &b->r;
          continue;
        }
        const Decoded &q = dec[qi[i]], &t = dec[ti[i]];
        if (q.ascii && t.ascii) {
          const size_t each = sizeof(StrHdr) + round8((size_t)o.nops + 1);
          StrHdr* hq = (StrHdr*)((char*)slab + str_off[kk]);
          StrHdr* ht = (StrHdr*)((char*)slab + str_off[kk] + each);
          hq->magic = ht->magic = kStrMagicSlab;
          char *qa = (char*)(hq + 1), *ta = (char*)(ht + 1);
          {
            uint32_t nx = 0, nj = 0;
            for (uint32_t c = 0; c < o.nops; c++) {
              nx += ops[c] != 2;
              nj += ops[c] != 1;
            }
            uint32_t cx = o.bx - nx, cj = o.bj - nj;  // 0-based index of the first aligned symbol
            size_t w = 0;
            for (uint32_t c = o.nops; c-- > 0; w++) {
              const uint8_t op = ops[c];
              qa[w] = op == 2 ? '-' : q.src[cx++];
              ta[w] = op == 1 ? '-' : t.src[cj++];
            }
          }
          qa[o.nops] = ta[o.nops] = 0;
          b->slab = slab;
          box_fill(b, qa, ta, path, o);
          mine += 3;
        } else {
          char *qa = nullptr, *ta = nullptr;
          size_t qlen = 0;
          if (!build_alignment(q, t, o, ops, &qa, &ta, &qlen)) continue;  // (out of memory: the pair reads as failed)
          b->slab = slab;
          box_fill(b, qa, ta, path, o);
          mine += 1;  // the box only: its strings are heap blocks
        }
        out[i] = &b->r;
      }
      live.fetch_add(mine);
    });
    if (live.load() == 0) slab_free(slab);
    else slab->refs.store(live.load(), std::memory_order_release);
    return SW_OK;
  };
  auto emit = [&](size_t k_lo, size_t k_hi, const OpsArena& ar) -> int {
    return k_hi > k_lo ? emit_list(nullptr, k_lo, k_hi - k_lo, ar) : SW_OK;
  };

  // Big batches of short pairs: one pair per lane (sw_lanes.hip) instead of one 256-row strip per wave.  The batch
  // goes to the GPU in slices; the result strings of a finished slice are assembled on a second thread while the
  // next slice runs (one slab per slice).
  std::vector<uint8_t> emitted;
  if (path != SW_PATH_STRAIGHT && !specs.empty()) {
    const std::vector<uint8_t>& small = ascii_of;
    struct {
      std::mutex mu;
      std::condition_variable cv;
      std::deque<std::vector<uint32_t>> q;
      bool done = false;
      int rc = SW_OK;
      std::string err;
    } lq;
    auto emit_slice = [&](const std::vector<uint32_t>& ks) {
      if (emitted.empty()) emitted.assign(specs.size(), 0);
      const int erc = guarded<int>(SW_E_NOMEM, [&] { return emit_list(ks.data(), 0, ks.size(), arena); });
      if (erc) {
        std::lock_guard<std::mutex> lk(lq.mu);
        if (!lq.rc) lq.rc = erc, lq.err = last_error_string();
      }
      for (uint32_t k : ks) emitted[k] = 1;
    };
    std::thread consumer;
    bool threaded = false;
    if (specs.size() >= 32768) {
      emitted.assign(specs.size(), 0);
      try {
        consumer = std::thread([&]() noexcept {
          for (;;) {
            std::vector<uint32_t> ks;
            {
              std::unique_lock<std::mutex> lk(lq.mu);
              lq.cv.wait(lk, [&] { return lq.done || !lq.q.empty(); });
              if (lq.q.empty()) return;
              ks.swap(lq.q.front());
              lq.q.pop_front();
            }
            emit_slice(ks);
          }
        });
        threaded = true;
      } catch (...) {
      }
    }
    const LanesReady ready = [&](const uint32_t* ks, size_t m) {
      if (!threaded) return;  // (the strings are assembled with everything else, below)
      std::vector<uint32_t> v(ks, ks + m);
      {
        std::lock_guard<std::mutex> lk(lq.mu);
        lq.q.emplace_back(std::move(v));
      }
      lq.cv.notify_one();
    };
    const int rc = guarded<int>(SW_E_NOMEM, [&] {
      return align_lanes(device, specs.data(), specs.size(), small.data(), sc, po.data(), arena, handled.data(), ready,
                         sym_pool_desc.base ? &sym_pool_desc : nullptr);
    });
    const std::string rc_err = rc ? last_error_string() : std::string();
    if (threaded) {
      {
        std::lock_guard<std::mutex> lk(lq.mu);
        lq.done = true;
      }
      lq.cv.notify_one();
      consumer.join();
    }
    if (rc) return fail(rc, rc_err);
    if (lq.rc) return fail(lq.rc, lq.err);
    tm.lap("lanes (+ strings of its slices)");
  }

  size_t n_handled = 0;
  for (uint8_t h_ : handled) n_handled += h_ != 0;
  // A whole work package of short pairs: cut into a few sub-batches so that the result strings of one are assembled
  // (here) while the next is on the GPU (a second host thread drives it).  One slab per sub-batch.
  size_t scratch_est = 0;  // generous bound of the device scratch the pairs need together (direction bits dominate)
  uint64_t cells_est = 0;
  if (n_handled == 0)
    for (const PairSpec& sp : specs) {
      scratch_est += (size_t)sp.Q * sp.T / 3 + 16 * ((size_t)sp.Q + sp.T) + 4096;
      cells_est += (uint64_t)sp.Q * sp.T;
    }
  // (many pairs, or a few thousand with milliseconds of GPU work: below that one submission is quicker)
  static const size_t pipe_min_pairs = [] {
    const char* e = getenv("SW_HIP_PIPE_MIN_PAIRS");
    return e ? (size_t)std::max(2, atoi(e)) : (size_t)2048;
  }();
  const bool worth = specs.size() >= 32768 || (specs.size() >= pipe_min_pairs && cells_est >= 2000000000ull);
  if (n_handled == 0 && worth && scratch_est <= ((size_t)3 << 30)) {
    size_t B = std::min<size_t>(8, std::max<size_t>(2, std::max<size_t>(specs.size() / 25000, (size_t)(cells_est / 4000000000ull))));
    if (const char* e = getenv("SW_HIP_PIPE_PARTS")) B = std::min<size_t>(16, std::max(2, atoi(e)));  // experiments
    struct Part {
      ArenaLease lease;
      int rc = SW_OK;
      bool done = false;
      std::string err;
    };
    std::vector<Part> parts(B);
    std::mutex mu;
    std::condition_variable cv;
    auto lo_of = [&](size_t b) { return specs.size() * b / B; };
    auto produce = [&]() noexcept {
      std::vector<size_t> cut(B + 1);
      std::vector<OpsArena*> arenas(B);
      for (size_t b = 0; b <= B; b++) cut[b] = lo_of(b);
      for (size_t b = 0; b < B; b++) arenas[b] = &parts[b].lease.a;
      auto ready = [&](size_t b, int rc) {
        std::string err;
        if (rc) {
          try {
            err = last_error_string();
          } catch (...) {
          }
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          parts[b].rc = rc;
          parts[b].err.swap(err);
          parts[b].done = true;
        }
        cv.notify_all();
      };
      size_t told = 0;
      const int rc = guarded<int>(SW_E_NOMEM, [&]() -> int {
        CtxLock L(device);
        if (L.rc) return L.rc;
        return run_exact_pipelined(*L.c, specs.data(), cut, sc, po.data(), arenas, [&](size_t b, int r) {
          ready(b, r);
          told = b + 1;
        });
      });
      for (size_t b = told; b < B; b++) ready(b, rc ? rc : SW_E_HIP);  // (lock failure or an exception: nothing ran)
    };
    std::thread producer;
    bool threaded = true;
    try {
      producer = std::thread(produce);
    } catch (...) {
      threaded = false;
    }
    if (threaded) {
      int rc = SW_OK;
      std::string err;
      for (size_t b = 0; b < B; b++) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return parts[b].done; });
        }
        if (rc) continue;  // (drain: the producer marks the remaining parts done)
        if (parts[b].rc) {
          rc = parts[b].rc;
          err = parts[b].err;
          continue;
        }
        const int erc = guarded<int>(SW_E_NOMEM, [&] { return emit(lo_of(b), lo_of(b + 1), parts[b].lease.a); });
        if (erc) {
          rc = erc;
          err = last_error_string();
        }
      }
      producer.join();
      tm.lap("run_exact + strings (pipelined)");
      return rc ? fail(rc, err) : SW_OK;
    }
    // (no thread to be had: the plain path below)
  }
  if (n_handled == 0 && !specs.empty()) {  // no copies
    CtxLock L(device);
    if (L.rc) return L.rc;
    const int rc = run_exact(*L.c, L.c->ws, specs.data(), specs.size(), sc, true, po.data(), &arena);
    if (rc) return rc;
  } else if (n_handled < specs.size()) {
    std::vector<PairSpec> rest;
    std::vector<size_t> rest_k;
    rest.reserve(specs.size() - n_handled);
    rest_k.reserve(specs.size() - n_handled);
    for (size_t k = 0; k < specs.size(); k++)
      if (!handled[k]) {
        rest.push_back(specs[k]);
        rest_k.push_back(k);
      }
    {
      std::vector<PairOut> ro(rest.size());
      CtxLock L(device);
      if (L.rc) return L.rc;
      const int rc = run_exact(*L.c, L.c->ws, rest.data(), rest.size(), sc, true, ro.data(), &arena);
      if (rc) return rc;
      for (size_t r = 0; r < rest.size(); r++) po[rest_k[r]] = ro[r];
    }
  }
  tm.lap("run_exact");
  int erc;
  if (emitted.empty()) {
    erc = emit(0, specs.size(), arena);
  } else {  // (the lane kernel's slices have their strings already)
    std::vector<uint32_t> rest_ks;
    for (size_t k = 0; k < specs.size(); k++)
      if (!emitted[k]) rest_ks.push_back((uint32_t)k);
    erc = emit_list(rest_ks.data(), 0, rest_ks.size(), arena);
  }
  tm.lap("strings");
  return erc;
}
}  // namespace

namespace {
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
      return align_many_on(-1, qs.data(), ts.data(), grp.size(), batch[i]->as, batch[i]->path, out.data());
    });
    for (size_t k = 0; k < grp.size(); k++) {
      OneReq* r = batch[grp[k]];
      r->rc = rc;
      if (rc != SW_OK) free_alignment_result(out[k]);  // boxed before the failure: nothing is handed over
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
  b->magic = 0;
  // (a string handed to free_c_string earlier has been detached: nullptr)
  uint64_t parts = 1;
  for (char* sp : {b->r.query_ptr, b->r.target_ptr}) {
    if (!sp) continue;
    StrHdr* h = (StrHdr*)sp - 1;
    if (h->magic == kStrMagicSlab) {
      h->magic = 0;
      parts++;
    } else {
      str_free(sp);
    }
  }
  if (b->slab) slab_release(b->slab, parts);
  else free(b);
}

// Many results at once (additive, include/sw_hip.h): what free_alignment_result does for each, with one reference
// count update per run of results that share a slab instead of one per result.
void sw_free_alignment_results(Result** results, size_t n) {
  if (!results) return;
  Slab* cur = nullptr;
  uint64_t cur_parts = 0;
  for (size_t i = 0; i < n; i++) {
    Result* r = results[i];
    results[i] = nullptr;
    if (!r) continue;
    ResultBox* b = (ResultBox*)r;
    if (b->magic != kMagic) continue;
    b->magic = 0;
    uint64_t parts = 1;
    for (char* sp : {b->r.query_ptr, b->r.target_ptr}) {
      if (!sp) continue;
      StrHdr* h = (StrHdr*)sp - 1;
      if (h->magic == kStrMagicSlab) {
        h->magic = 0;
        parts++;
      } else {
        str_free(sp);
      }
    }
    if (!b->slab) {
      free(b);
      continue;
    }
    if (b->slab != cur) {
      if (cur) slab_release(cur, cur_parts);
      cur = b->slab;
      cur_parts = 0;
    }
    cur_parts += parts;
  }
  if (cur) slab_release(cur, cur_parts);
}

void free_c_string(char* str) {
  if (!str) return;
  StrHdr* h = (StrHdr*)str - 1;
  // Only strings this library returned may be passed (include/sw.h).  Anything else is left alone: looking in front
  // of a foreign pointer is already out of bounds, freeing it would be worse.
  if (h->magic != kStrMagic && h->magic != kStrMagicSlab) return;
  ResultBox* b = h->owner;
  if (b) {  // detach from its Result so free_alignment_result will not free it again
    if (b->r.query_ptr == str) b->r.query_ptr = nullptr;
    if (b->r.target_ptr == str) b->r.target_ptr = nullptr;
  }
  if (h->magic == kStrMagicSlab) {
    h->magic = 0;
    if (b && b->slab) slab_release(b->slab, 1);
  } else {
    str_free(str);
  }
}

// ================================================================================================
// include/sw_hip.h
// ================================================================================================
int sw_hip_init(int device) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
    {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      if (device < 0) device = g_default_device < 0 ? 0 : g_default_device;
      if (g_default_device >= 0 && device != g_default_device)
        return fail(SW_E_ARG, "sw_hip_init: already initialised on another device (one default device per process; "
                              "use sw_node_create to drive several GPUs from one process)");
      g_default_device = device;
    }
    Context* c = nullptr;
    const int rc = ctx_get(device, &c);
    if (rc) {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      if (!g_ctx[device]) g_default_device = -1;  // nothing was created: the next call may choose again
    }
    return rc;
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
  const int rc = swhip::guarded<int>(SW_E_NOMEM, [&] { return align_many_on(-1, queries, targets, n_pairs, scores, path, out); });
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

int sw_hip_set_scratch_mb(int device, uint64_t mb) {
  if (mb == 0) return fail(SW_E_ARG, "scratch budget must be positive");
  CtxLock L(device);
  if (L.rc) return L.rc;
  L.c->scratch_budget = (size_t)mb << 20;
  return SW_OK;
}

uint32_t sw_hip_struct_size(int which) {
  switch (which) {
    case 0: return (uint32_t)sizeof(sw_kernel_stats);
    case 1: return (uint32_t)sizeof(sw_hit);
    case 2: return (uint32_t)sizeof(sw_alignment_detail);
    case 3: return (uint32_t)sizeof(sw_work_package);
    case 4: return (uint32_t)sizeof(sw_work_pair);
    default: return 0;
  }
}

int sw_hip_last_stats(sw_kernel_stats* out) {
  if (!out) return fail(SW_E_ARG, "null argument");
  CtxLock L(-1);
  if (L.rc) return L.rc;
  *out = L.c->stats;
  return SW_OK;
}

}  // extern "C"


