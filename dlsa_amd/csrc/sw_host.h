// dlsa_amd/csrc/sw_host.h -- host-side internals shared by the translation units of libsw_hip.so
// (sw_lib.cpp: contexts, exact driver, per-pair ABI; sw_db.cpp: resident database, score pass, top-k;
// sw_node.cpp: several GPUs behind one handle; sw_worker.cpp: the worker-side callers).
// Not installed; the public contract is include/*.h.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/sw_worker.h"
#include "sw_internal.h"

namespace swhip {

// ---- errors --------------------------------------------------------------------------------------------
int fail(int code, const std::string& msg);  // records the calling thread's message, returns code
const std::string& last_error_string();
#define HIPCK(expr)                                                                                  \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return ::swhip::fail(SW_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));            \
  } while (0)

// ---- device memory that grows on demand, or a fixed window of the no-allocation scratch ------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool fixed = false;  // a window of the static scratch array: never allocates, never frees (sw_db_scores_noalloc)
  int ensure(size_t bytes);
  void bind_fixed(void* ptr, size_t bytes) {
    p = ptr;
    cap = bytes;
    fixed = true;
  }
  void release();
};
// pinned host staging memory (async copies from pageable memory are staged by the runtime, synchronously);
// in fixed mode a window of plain host memory reserved when the context was set up
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool fixed = false;
  int ensure(size_t bytes);
  void bind_fixed(void* ptr, size_t bytes) {
    p = ptr;
    cap = bytes;
    fixed = true;
  }
  void release();
};

// std::allocator without value-initialisation: resize() of a big host table does not write it first
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
// Host tables of one exact-path submission (run_exact_chunk), kept between calls: a work package of 10^5 pairs needs
// ~12 MB of them, and fresh vectors cost more in page faults than filling them does.
struct ExactHost {
  std::vector<ExactPair, NoInitAlloc<ExactPair>> hp;
  std::vector<size_t, NoInitAlloc<size_t>> qoff, toff, dirw;
  std::vector<uint32_t> unit_pair, unit_order;
  // sequence id -> index of its copy in the current submission; an entry counts only if its stamp is the current
  // epoch, so nothing is cleared between submissions
  std::vector<uint64_t> id_copy;  // (epoch << 32) | index
  uint32_t epoch = 0;
  void trim();  // gives the memory back when a submission was unusually large
};

// Everything a submission needs beyond its inputs and outputs.  A context owns two: `ws` grows on demand
// (hipMalloc), `na` is carved out of the per-device static scratch array once, at set-up.
struct Workspace {
  DevBuf xin, xzero, xout, dirs, xrecs;                    // exact path (run_exact_chunk)
  HostBuf h_in, h_out;                                     // ... its host staging buffers
  ExactHost xh;                                            // ... and host tables
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr;  // ... kernel bracket and completion (made on first use)
  DevBuf sel_hist, sel_ctr, sel_rerun, sel_cand, sel_tie;  // device-side top-k selection
  DevBuf qwords, slot_scores, gbound, ticket, prog, units; // score path
  DevBuf strip_best;                                       // ... per-strip maxima (work-package fast path)
  DevBuf blkinfo;                                          // ... row blocks of a batch of queries (ScoreParams::blkinfo)
  void release();
};

constexpr int kMaxDevices = 64;

// A block of plain host memory reserved when a context is set up; the allocation-free entry point carves its
// temporaries out of it with a bump pointer (reset at the start of every call, under the context's lock).
struct HostArena {
  unsigned char* base = nullptr;
  size_t cap = 0, used = 0;
  void reset() { used = 0; }
  void* take(size_t bytes) {
    const size_t at = (used + 63) & ~(size_t)63;
    if (at + bytes > cap || at + bytes < at) return nullptr;
    used = at + bytes;
    return base + at;
  }
};

struct Context {
  std::mutex mu;  // one GPU submission at a time per device; the ABI is entered from many threads
  int device = 0;
  int cus = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // the lane path's second slice in flight (sw_lanes.cpp)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  Workspace ws;
  Workspace ws2;  // second set for submissions in flight while the first is being read back (work packages)
  struct {  // device arrays of the context's temporary database (sw_db.cpp: sw_db::borrowed)
    DevBuf words, lens, slot_target, groups, elem_group, units;
    bool busy = false;
  } tdb;
  DevBuf sympool;  // the lane path's copy of a batch's symbol block (SymPool)
  hipEvent_t ev_pool = nullptr;  // ... uploaded (made on first use)
  // the allocation-free entry point (sw_db_scores_noalloc): windows of this device's copy of the static scratch
  // array (sw_lib.cpp) and a host arena, both bound once when the context is created
  Workspace na;
  HostArena na_host;
  void* na_scratch = nullptr;  // device address of the scratch array on THIS device (hipGetSymbolAddress)
  sw_kernel_stats stats{};
  size_t scratch_budget = (size_t)8 << 30;
};

// The context of a device (created on first use; device < 0: the process's default device, which sw_hip_init
// selects and which is 0 otherwise).  On failure *out is null and the error is recorded.
int ctx_get(int device, Context** out);
// Lanes of a device: lane 0 is the device's context above; lanes 1 .. kMaxCtxLanes - 1 are auxiliary contexts (made on
// first use) with a stream and workspaces of their own, so that independent latency-bound jobs of one call -- the
// target sets of a combined batch of per-pair callers, sw_package.cpp -- run side by side on one GPU.
constexpr unsigned kMaxCtxLanes = 8;
int ctx_get_lane(int device, unsigned lane, Context** out);
void reset_stats(Context& c);
size_t noalloc_scratch_bytes();
// Locks a context for one submission and makes its device current for the calling thread (HIP's current device
// is per thread; the ABI is entered from many threads).
struct CtxLock {
  Context* c = nullptr;
  int rc = SW_OK;
  std::unique_lock<std::mutex> lk;
  explicit CtxLock(int device);
  explicit CtxLock(Context* ctx);
};

// Error returns between queued asynchronous work and its synchronisation must not leave copies in flight that read
// or write buffers the caller is about to release (stack variables, vectors, the staging buffers of the next
// submission): every function that queues work holds one of these; on the success path the stream is idle already.
struct StreamDrain {
  hipStream_t s;
  bool armed = true;
  explicit StreamDrain(hipStream_t st) : s(st) {}
  void dismiss() { armed = false; }  // the work stays queued on purpose: the caller holds another drain further out
  ~StreamDrain() {
    if (armed) (void)hipStreamSynchronize(s);
  }
  StreamDrain(const StreamDrain&) = delete;
  StreamDrain& operator=(const StreamDrain&) = delete;
};

// ---- text -> symbols ----------------------------------------------------------------------------------------
// A sequence as the algorithm sees it: Unicode scalars truncated to u16 (`*x as u16`,
// rust/src/algorithm/mod.rs:78,224).  Pure-ASCII input (the normal case: DNA) never materialises code points.
// The symbols of one sequence: a vector of its own, or -- bind() -- a stretch of a pool that the caller owns and that
// has room for every later resize() (a work package decodes 10^5 sequences: one pool instead of 10^5 heap blocks).
class SymBuf {
  std::vector<uint16_t, NoInitAlloc<uint16_t>> own_;
  uint16_t* p_ = nullptr;
  size_t n_ = 0;
  bool ext_ = false;

 public:
  SymBuf() = default;
  SymBuf(const SymBuf& o) { *this = o; }
  SymBuf& operator=(const SymBuf& o) {
    if (this != &o) {
      own_.assign(o.data(), o.data() + o.size());
      p_ = own_.data();
      n_ = own_.size();
      ext_ = false;
    }
    return *this;
  }
  SymBuf(SymBuf&&) = default;  // (a moved vector keeps its block, so p_ stays right)
  SymBuf& operator=(SymBuf&&) = default;
  void bind(uint16_t* pool) {
    p_ = pool;
    n_ = 0;
    ext_ = true;
  }
  void resize(size_t n) {
    if (!ext_) {
      own_.resize(n);
      p_ = own_.data();
    }
    n_ = n;
  }
  uint16_t* data() { return p_; }
  const uint16_t* data() const { return p_; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  uint16_t& operator[](size_t i) { return p_[i]; }
  const uint16_t& operator[](size_t i) const { return p_[i]; }
  const uint16_t* begin() const { return p_; }
  const uint16_t* end() const { return p_ + n_; }
};
struct Decoded {
  const char* src = nullptr;  // borrowed
  bool ascii = true;
  std::vector<uint32_t> cps;  // only when !ascii: for re-encoding the alignment
  SymBuf sym;
  uint32_t cp(size_t i) const { return ascii ? (uint32_t)(unsigned char)src[i] : cps[i]; }
};
bool decode(const char* s, size_t n, Decoded& d);
bool decode(const char* s, Decoded& d);
inline uint32_t roundup64(uint32_t v) { return (v + 63u) / 64u * 64u; }
Scoring to_scoring(AlignmentScores s);

// ---- exact path driver ----------------------------------------------------------------------------------------
struct PairSpec {
  const uint16_t* q;
  uint32_t Q;
  const uint16_t* t;
  uint32_t T;
  int path;
  // optional: small dense ids of the two sequences within the submission (equal id <=> same sequence); lets the
  // driver upload every distinct sequence once without hashing 10^5 pointers.  UINT32_MAX: look the pointer up.
  uint32_t qid = UINT32_MAX, tid = UINT32_MAX;
  // a WINDOW of the pair's matrix (ExactPair in sw_internal.h): rows 1..Q, columns col0+1 .. col0+T; t points at the
  // symbol of column col0+1; Qfull / Tfull are the lengths that decide the reference's path quirks and tie order
  uint32_t col0 = 0, Qfull = 0, Tfull = 0;
  const uint32_t* left = nullptr;  // device pointer into the score kernel's boundary ring (nullptr: zeros)
  uint32_t left_shift = 0;
  uint32_t walk_x = 0, walk_j = 0;  // != 0: the walker resumes here with H = walk_H
  int32_t walk_H = 0;
};
struct PairOut {
  int32_t score = 0;
  uint32_t bx = 0, bj = 0;
  size_t ops_off = 0;  // traceback ops (last alignment column first) at arena[ops_off .. ops_off + nops)
  uint32_t nops = 0;
  // != 0 (the lane kernel, ASCII pairs only): the arena holds the two ALIGNED STRINGS instead of ops, in reading
  // order -- the query alignment at arena[ops_off, ops_off + nops), the target alignment nops bytes from the next
  // multiple of 4 after it
  uint32_t strings = 0;
  uint32_t end_x = 0, end_j = 0;  // where the walker stopped (window-local column; 0 = the boundary column)
  int32_t end_H = 0;
  uint64_t key = ~0ull;           // discovery-order key of the best cell (global column)
};
struct PathParams {
  uint32_t Tlim, T0, phaseQp, straight;
};
// Traceback ops of a submission (one byte per alignment column, PairOut::ops_off / nops index it).  Tens of
// megabytes for a work package of 10^5 pairs, so: no value-initialisation on resize (the bytes are overwritten by a
// device-to-host copy), and the storage is leased from a small pool and given back with its pages already mapped
// and pinned (ArenaLease) instead of being allocated afresh on every call.  Always use it through a lease: a
// stand-alone OpsArena pays a hipHostMalloc per growth step.
// ... in page-locked host memory, so that the device-to-host copy of the ops lands in it directly at link speed
template <class T>
struct PinnedNoInitAlloc {
  using value_type = T;
  template <class U> struct rebind { using other = PinnedNoInitAlloc<U>; };
  PinnedNoInitAlloc() = default;
  template <class U> PinnedNoInitAlloc(const PinnedNoInitAlloc<U>&) {}
  T* allocate(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocPortable) != hipSuccess || !p) throw std::bad_alloc();
    return (T*)p;
  }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <class U> bool operator==(const PinnedNoInitAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedNoInitAlloc<U>&) const { return false; }
};
using OpsArena = std::vector<uint8_t, PinnedNoInitAlloc<uint8_t>>;
struct ArenaLease {
  OpsArena a;
  // want: bytes the caller is going to need (the smallest pooled block that holds them; 0: the largest block)
  explicit ArenaLease(size_t want = 0);
  ~ArenaLease();
  ArenaLease(const ArenaLease&) = delete;
  ArenaLease& operator=(const ArenaLease&) = delete;
};
PathParams resolve_path(int path, uint32_t Q, uint32_t T);
unsigned long long exact_wait_ticks();
bool ring_div_by_zero(int path, uint32_t Q, uint32_t T, Scoring sc);
// Caller holds the context's lock.
int run_exact(Context& c, Workspace& ws, const PairSpec* specs, size_t n, Scoring sc, bool want_trace, PairOut* outs,
              OpsArena* arena = nullptr);

// ---- the work-package fast path (sw_package.cpp) ------------------------------------------------------------------
// For the pairs it is eligible for, fills po[k] (global end cell, score, ops appended to `arena`) and sets
// handled[k]; everything else is left to the whole-matrix path.  dec: the distinct decoded sequences of the batch;
// pair_q / pair_t: per pair, indices into dec.  Takes the lock of `device`'s context (-1: the default device) itself.
constexpr size_t kWindowsMinTarget = 2048;  // symbols of the shortest target the fast path takes
int align_windows(const std::vector<Decoded>& dec, const std::vector<uint32_t>& pair_q, const std::vector<uint32_t>& pair_t,
                  Scoring sc, std::vector<PairOut>& po, OpsArena& arena, std::vector<uint8_t>& handled,
                  int device = -1);
// The one-pair-per-lane exact kernel (sw_lanes.cpp / sw_lanes.hip) for big batches of short pairs: answers the pairs
// it is eligible for and that its cost model prefers (po[k], ops appended to `arena`, handled[k] = 1), leaves the
// rest alone.  small_syms[k]: every symbol of both sequences is below 0xfffe (the kernel's sentinels).  Takes the
// lock of `device`'s context itself.
// `ready(ks, m)` is called once per finished slice of the batch with the indices of the pairs it answered, while
// the next slice is on the GPU (the arena does not move until the call returns: the ops of every slice have their
// place in it from the start).
using LanesReady = std::function<void(const uint32_t* ks, size_t m)>;
// Optional: the symbols of every sequence of the batch as ONE page-locked block in which each sequence starts at a
// multiple of 8 symbols (align_interned decodes into such a block).  The lane path then uploads the block once and
// addresses it by offsets, instead of gathering the distinct sequences of every slice into a staging buffer.
struct SymPool {
  const uint16_t* base = nullptr;
  size_t syms = 0;
};
int align_lanes(int device, const PairSpec* specs, size_t n, const uint8_t* small_syms, Scoring sc, PairOut* po,
                OpsArena& arena, uint8_t* handled, const LanesReady& ready = nullptr, const SymPool* syms = nullptr);
// A batch of pairs on one device (sw_lib.cpp): the body of sw_align_batch.  out[i] = NULL for a pair the reference
// would have failed on; a non-zero return means nothing was handed over.
int align_many_on(int device, const char* const* queries, const char* const* targets, size_t n, AlignmentScores as,
                  int path, Result** out);
// The same for pairs given as indices into a table of distinct sequences (UINT32_MAX or a NULL entry: the pair
// fails) -- what a work package holds (sw_worker.cpp), so that nothing has to be re-discovered from pointers.
int align_interned(int device, const char* const* distinct, size_t n_distinct, const uint32_t* qi, const uint32_t* ti,
                   size_t n, AlignmentScores as, int path, Result** out);
struct PackageStats {
  uint64_t pairs = 0, queries = 0, windows = 0, rounds = 0, window_cells = 0, chunks = 0, chunk_escapes = 0;
};
extern thread_local PackageStats g_package_stats;

// ---- result strings ---------------------------------------------------------------------------------------------
char* str_alloc(size_t n);
void str_free(char* p);
// Builds the two aligned strings (str_alloc'ed, NUL-terminated) from the walker's ops.
bool build_alignment(const Decoded& q, const Decoded& t, const PairOut& o, const uint8_t* ops, char** qa_out,
                     char** ta_out, size_t* qa_len);

// host threads for the byte loops around a big submission (decoding, string assembly, packing); an exception
// inside a helper thread is caught there and re-raised (as bad_alloc) by the caller
void parallel_ranges_impl(size_t n, unsigned nt, void (*fn)(void*, size_t, size_t), void* arg);
template <class F>
void parallel_ranges(size_t n, size_t work, F&& f) {
  unsigned nt = 16;
  if (work < (1u << 20) || n < 32) nt = 1;
  using Fn = typename std::remove_reference<F>::type;
  parallel_ranges_impl(n, nt, [](void* a, size_t lo, size_t hi) { (*(Fn*)a)(lo, hi); }, (void*)&f);
}

// SW_HIP_TRACE=1: host stage times of big submissions on stderr (development aid)
struct StageTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  // (submissions of at least 1000 pairs, or SW_HIP_TRACE_MIN)
  static size_t min_pairs() {
    static const size_t v = getenv("SW_HIP_TRACE_MIN") ? (size_t)atoll(getenv("SW_HIP_TRACE_MIN")) : 1000;
    return v;
  }
  explicit StageTimer(size_t n) : on((n >= min_pairs() || n == 0) && getenv("SW_HIP_TRACE")), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sw_hip] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// ---- resident database (sw_db.cpp) ------------------------------------------------------------------------------
struct QueryPrep {
  Decoded d;
  bool acgt = true;
  std::vector<uint32_t> qwords;        // 2 bits per row, kScoreB rows per word (+ a padding word)
  std::vector<uint32_t> qwords_small;  // ... kScoreBSmall rows per word: passes over small databases (score_block_rows)
};
bool prep_query(const char* query, QueryPrep& qp);
struct TopEntry {
  int32_t score;
  uint32_t target;
};
// score descending, then database index ascending (declared replacement of the reference's arrival-order
// ties, cli/__main__.py:171)
inline bool top_before(const TopEntry& a, const TopEntry& b) {
  return a.score != b.score ? a.score > b.score : a.target < b.target;
}
// one reported hit to be aligned: query index, database index, the score the selection saw
struct HitRef {
  uint32_t query;
  uint32_t target;
  int32_t score;
};
// Phase A of a search: per query the k best (score, target) of this database.  Takes the database's context lock.
int db_topk_phase(sw_db* db, const std::vector<QueryPrep>& qps, Scoring sc, size_t k, std::vector<std::vector<TopEntry>>& tops);
// Phase B: end cell + traceback of the listed hits in one exact submission; hits_out[i] describes refs[i]
// (target_index = the database-local index; strings owned by the caller).  Takes the context lock.
int db_align_phase(sw_db* db, const std::vector<QueryPrep>& qps, Scoring sc, const std::vector<HitRef>& refs, sw_hit* hits_out);
Context* db_context(sw_db* db);
const sw_kernel_stats& db_last_stats(const sw_db* db);  // counters of the database's own last search
// A score pass that leaves every inter-unit boundary column resident (ring slot = ticket) and reports, per strip
// of the stream, the packed maxima of that strip alone.  Pointers stay valid until the next pass on the context.
struct PersistPass {
  // Host copies of the pass's two outputs, in a page-locked arena leased from the pool (14 MB per pass on the
  // reference's synthetic data: pageable and freshly allocated they cost a millisecond per query).
  ArenaLease host;
  const int32_t* slot = nullptr;         // per slot: best score (32767 = saturated)
  const uint32_t* strip_best = nullptr;  // per strip 64 dwords (lo half: slot lane, hi half: slot 64 + lane)
  std::shared_ptr<const std::vector<ScoreUnit>> units_keep;  // (a later pass may replace the database's table)
  const std::vector<ScoreUnit>* units = nullptr;   // ticket order
  const std::vector<ScoreGroup>* groups = nullptr;
  const std::vector<uint32_t>* slot_target = nullptr;
  const uint32_t* d_ring = nullptr;     // device: boundary column of ticket u at d_ring + u * ring_stride (+ lane)
  size_t ring_stride = 0;               // dwords
  uint32_t n_elems = 0;
  // a pass in flight (db_score_persist_begin_locked .. _end_locked): its own events, its error word in `host`
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr;
  const uint32_t* kerr = nullptr;
  uint32_t Q = 0;
  PersistPass() = default;
  PersistPass(const PersistPass&) = delete;
  PersistPass& operator=(const PersistPass&) = delete;
  ~PersistPass();
};
int db_score_persist_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, PersistPass& out);
int db_score_persist_begin_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, PersistPass& out);
int db_score_persist_end_locked(Context& c, sw_db* db, PersistPass& out);
size_t db_persist_ring_bytes(const sw_db* db, uint32_t Q, int cus);
sw_db* db_create_on(Context* ctx, const char* flat, const uint64_t* offsets, uint64_t n);
sw_db* db_create_owned(Context* ctx, std::string&& flat, std::vector<uint64_t>&& off, bool known_acgt = false);
// A temporary database of ACGT-only sequences on the device arrays of `c` (no hipMalloc / hipFree per database);
// the caller holds c's lock from here until db_release_locked.
sw_db* db_create_temp_locked(Context& c, std::string&& flat, std::vector<uint64_t>&& off);
void db_release_locked(sw_db* db);
std::string db_take_flat(sw_db* db);
sw_db* db_open_on(Context* ctx, const char* path);  // a file written by sw_db_save
const std::string& db_flat(const sw_db* db);
const std::vector<uint64_t>& db_offsets(const sw_db* db);
double benchmark_cups_on(Context* ctx);  // cells per second of the batch path on one device

}  // namespace swhip
