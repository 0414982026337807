// dlsa_amd/csrc/sw_lanes.cpp -- host side of the one-pair-per-lane exact kernel (sw_lanes.hip): which pairs of a
// batch it may answer, whether that pays, the length-sorted groups of 64, one submission per slice of groups.
//
// Caller: align_interned (sw_lib.cpp), i.e. sw_align_batch / sw_execute_work / the combined per-pair callers -- the
// record-per-pair protocol of internal/worker/worker_impl.go:108-168.  Results are those of the whole-matrix path
// (sw_exact.hip) bit for bit: same score, same first maximal cell in the reference's discovery order, same
// traceback (tests/test_gpu_lanes.py compares both with the oracle).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "sw_host.h"

namespace swhip {

namespace {

// SW_HIP_LANES: "0" never, "1" whenever a pair is eligible (tests), unset / "auto": when the cost model says so
// (read at every call, so that a test can switch it)
int lanes_mode() {
  const char* e = getenv("SW_HIP_LANES");
  if (!e || !*e || !strcmp(e, "auto")) return 2;
  return atoi(e) ? 1 : 0;
}

struct Elig {
  uint32_t k;     // index into specs
  uint32_t Q, T;  // rows, columns computed (Tlim)
  uint32_t Tfull; // the target's length (statistics count the pair's whole matrix)
  bool row1_extra;  // the path also computes H(1, Tlim + 1) (T % 64 == 0 quirk): decided after the kernel
};
struct LanesScratch {
  std::vector<Elig> cand, el;
  std::vector<uint8_t> is_el;
};
std::mutex g_scratch_mu;
LanesScratch g_scratch_kept;

}  // namespace

int align_lanes(int device, const PairSpec* specs, size_t n, const uint8_t* small_syms, Scoring sc, PairOut* po,
                OpsArena& arena, uint8_t* handled, const LanesReady& ready, const SymPool* syms) {
  const int mode = lanes_mode();
  if (mode == 0 || n == 0) return SW_OK;
  // the kernel's domain: sentinel cells lose only if every step off a match costs something
  if (!(sc.match > 0 && sc.miss < 0 && sc.gap < 0)) return SW_OK;
  if (mode == 2 && n < 1024) return SW_OK;  // (the cost model never takes fewer than 1 024 eligible pairs)
  StageTimer tm(n);
  CtxLock L(device);
  if (L.rc) return L.rc;
  Context& c = *L.c;
  Workspace* wss[2] = {&c.ws, &c.ws2};
  for (Workspace* w : wss)
    for (hipEvent_t* e : {&w->ev_k0, &w->ev_k1, &w->ev_done})
      if (!*e && hipEventCreate(e) != hipSuccess) return fail(SW_E_HIP, "hipEventCreate");
  if (!c.stream2 && hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking) != hipSuccess)
    return fail(SW_E_HIP, "hipStreamCreate");
  if (!c.ev_pool && hipEventCreate(&c.ev_pool) != hipSuccess) return fail(SW_E_HIP, "hipEventCreate");
  // A big batch's symbol block (SymPool, below) goes to the device right away: the copy -- 30 MB for 100 000 distinct
  // reads -- runs while the host threads sort out which pairs are eligible.  (Should the batch not take the block
  // after all, the copy was for nothing; the drain makes sure it is over before the caller's block goes away.)
  StreamDrain drain_upload(c.stream);
  bool uploaded = false;
  if (syms && syms->base && syms->syms && n >= 32768 && syms->syms < 0x7fffff00u && syms->syms * 2 <= c.scratch_budget / 4) {
    int erc;
    if ((erc = c.sympool.ensure(syms->syms * 2 + 256))) return erc;
    HIPCK(hipMemcpyAsync(c.sympool.p, syms->base, syms->syms * 2, hipMemcpyHostToDevice, c.stream));
    HIPCK(hipEventRecord(c.ev_pool, c.stream));
    uploaded = true;
  }
  // (on the host threads: a package of reads is 10^5 pairs)
  // (the tables of a big batch are handed from call to call -- one set is kept -- instead of being page-faulted in
  // afresh: 20 MB for a part of a package of reads)
  struct ScratchLease {
    LanesScratch t;
    bool big;
    explicit ScratchLease(bool big_) : big(big_) {
      if (!big) return;
      std::lock_guard<std::mutex> lk(g_scratch_mu);
      std::swap(t, g_scratch_kept);
    }
    ~ScratchLease() {
      if (!big || t.cand.capacity() > ((size_t)1 << 19)) return;
      std::lock_guard<std::mutex> lk(g_scratch_mu);
      if (t.cand.capacity() > g_scratch_kept.cand.capacity()) std::swap(t, g_scratch_kept);
    }
  } scratch(n >= 32768);
  std::vector<Elig>& cand = scratch.t.cand;
  std::vector<uint8_t>& is_el = scratch.t.is_el;
  cand.resize(n);
  is_el.assign(n, 0);
  // (for the choice between one upload of the caller's symbol block and per-slice staging, below)
  std::atomic<uint64_t> syms_unshared{0};
  std::atomic<bool> all_in_block{syms && syms->base && syms->syms};
  parallel_ranges(n, n * 64, [&](size_t lo, size_t hi) {
    uint64_t mine = 0;
    bool inside = true;
    for (size_t k = lo; k < hi; k++) {
      if (handled[k] || !small_syms[k]) continue;
      const PairSpec& s = specs[k];
      if (s.path == SW_PATH_STRAIGHT || s.left || s.col0 || s.Tfull || s.Qfull || s.walk_x) continue;
      if (s.Q == 0 || s.T == 0 || (uint64_t)s.Q + s.T > 65000u) continue;
      const PathParams pp = resolve_path(s.path, s.Q, s.T);
      if (pp.phaseQp || pp.straight || pp.Tlim == 0 || pp.T0 < pp.Tlim) continue;
      // V = H << 16 | tag must stay a non-negative int32
      if ((uint64_t)std::min(s.Q, pp.Tlim) * (uint64_t)sc.match > 32767u) continue;
      cand[k] = Elig{(uint32_t)k, s.Q, pp.Tlim, s.T, pp.T0 > pp.Tlim};
      is_el[k] = 1;
      mine += (uint64_t)s.Q + s.T;
      if (syms && inside)
        inside = s.q >= syms->base && s.q + s.Q <= syms->base + syms->syms && s.t >= syms->base &&
                 s.t + s.T <= syms->base + syms->syms && ((s.q - syms->base) & 7) == 0 && ((s.t - syms->base) & 7) == 0;
    }
    syms_unshared.fetch_add(mine);
    if (!inside) all_in_block.store(false);
  });
  std::vector<Elig>& el = scratch.t.el;
  el.clear();
  el.reserve(n);
  for (size_t k = 0; k < n; k++)
    if (is_el[k]) el.push_back(cand[k]);
  if (el.empty()) return SW_OK;
  // length-sorted (rows, then columns, descending): the lanes of a wave run max rows x max strips of their group
  const bool sorted = std::is_sorted(el.begin(), el.end(), [](const Elig& a, const Elig& b) {
    return a.Q != b.Q ? a.Q > b.Q : a.T > b.T;
  });
  if (!sorted)
    std::stable_sort(el.begin(), el.end(), [](const Elig& a, const Elig& b) { return a.Q != b.Q ? a.Q > b.Q : a.T > b.T; });
  const size_t n_groups = (el.size() + 63) / 64;
  std::vector<LaneGroup> groups(n_groups);
  double cyc_sum = 0, cyc_max = 0, wave_ns = 0;
  uint64_t padded_cells = 0;
  for (size_t g = 0; g < n_groups; g++) {
    const size_t lo = g * 64, hi = std::min(el.size(), lo + 64);
    uint32_t qmax = 0, tmax = 0;
    for (size_t i = lo; i < hi; i++) {
      qmax = std::max(qmax, el[i].Q);
      tmax = std::max(tmax, el[i].T);
      // what the row-strip wavefront kernel needs for the pair (measured: reads 0.48 TCUPS, 1 kb pairs 1.85)
      const double strips = (el[i].Q + 255) / 256;
      wave_ns += strips * 256.0 * (el[i].T + 63.0) / (el[i].Q < 256 ? 1.2e3 : 2.0e3);
    }
    groups[g].Qmax = (qmax + 7) / 8 * 8;
    groups[g].nstrips = (tmax + 15) / 16;
    const double cyc = (double)groups[g].Qmax * groups[g].nstrips * 16 * 30.0;  // issue cycles of one wave
    cyc_sum += cyc;
    cyc_max = std::max(cyc_max, cyc);
    padded_cells += (uint64_t)groups[g].Qmax * groups[g].nstrips * 16 * 64;
  }
  if (mode == 2) {
    // Every strip of a group is a wave of its own: the waves share 1024 SIMDs (about 0.7 of the issue rate is
    // reached), and a group is done no sooner than its pipeline of strips -- the rows of one strip plus about 24 rows
    // of lag per further strip, on waves that may be alone on their SIMD (half rate).
    double crit = 0;
    for (const LaneGroup& g : groups) crit = std::max(crit, ((double)g.Qmax + 24.0 * g.nstrips) * 16 * 30.0);
    const double lanes_ns = std::max(crit / 0.5, cyc_sum / 1024 / 0.7) / 2.4;
    if (el.size() < 1024 || lanes_ns > 0.8 * wave_ns) return SW_OK;
  }
  tm.lap("lanes: eligible + sort");

  // The symbols: when the caller decoded the batch into one page-locked block (SymPool) that is no bigger than what
  // the slices would gather anyway, the block is uploaded ONCE, as it is, and a pair names its sequences by offset --
  // no per-slice gathering of distinct sequences, no staging copy.  Otherwise every slice stages its own pool.
  // (the block must not be much more than staging without any sharing would copy -- its sequences are padded to 8
  // symbols, hence the quarter of slack; every eligible pair's sequences lie in it at multiples of 8 symbols by
  // construction -- a spec from elsewhere turns the choice off)
  const bool whole_pool = all_in_block.load() && syms->syms < 0x7fffff00u &&
                          syms->syms <= syms_unshared.load() + syms_unshared.load() / 4 + 4096 &&
                          syms->syms * 2 <= c.scratch_budget / 4;
  if (whole_pool && !uploaded) {
    int erc;
    if ((erc = c.sympool.ensure(syms->syms * 2 + 256))) return erc;
  }

  // Slices of groups: each fits the scratch budget (direction words dominate: rows x strips x 256 bytes per group),
  // and a big batch is cut into a few slices anyway so that TWO are in flight -- while slice i is on the GPU, slice
  // i + 1 is staged and queued behind it on the second workspace, and the caller's `ready` (string assembly, on
  // another thread) works on slice i - 1.
  struct Slice {
    size_t g0 = 0, g1 = 0, dir_words = 0, hb_words = 0, n_units = 0;
    size_t dense_cap = 0;   // most bytes the packed strings of the slice can take ...
    size_t dense_base = 0;  // ... and where they go in the arena (relative to arena_base)
    uint64_t cells = 0;
  };
  std::vector<Slice> slices;
  {
    // (two slices at least for a batch worth pipelining; beyond that a slice keeps >= 2048 waves so that it still
    // fills the GPU's 1024 SIMDs twice over)
    const size_t want = el.size() >= 32768 ? std::min<size_t>(16, std::max<size_t>(2, el.size() / 32768)) : 1;
    const size_t per = (n_groups + want - 1) / want;
    size_t g0 = 0;
    while (g0 < n_groups) {
      Slice sl;
      sl.g0 = g0;
      size_t g1 = g0;
      while (g1 < n_groups && g1 - g0 < per && g1 - g0 < 0xfffffu && sl.n_units < 0x7ff00000u) {
        const size_t dw = (size_t)groups[g1].Qmax * groups[g1].nstrips * 64, hw = (size_t)groups[g1].Qmax * 64;
        if (g1 > g0 && (sl.dir_words + dw + sl.hb_words + hw) * 4 > c.scratch_budget / 2) break;
        // (the packed strings of a slice are addressed by a 32-bit byte counter on the device)
        size_t dense = 0;
        for (size_t i = g1 * 64; i < std::min(el.size(), g1 * 64 + 64); i++) dense += lanes_dense_bytes(el[i].Q + el[i].T);
        if (g1 > g0 && sl.dense_cap + dense > 0xc0000000u) break;
        sl.dense_cap += dense;
        groups[g1].dirs_off = sl.dir_words;
        groups[g1].hb_off = sl.hb_words;
        groups[g1].rec_off = (uint32_t)sl.n_units;
        sl.n_units += groups[g1].nstrips;
        sl.dir_words += dw;
        sl.hb_words += hw;
        g1++;
      }
      sl.g1 = g1;
      slices.push_back(std::move(sl));
      g0 = g1;
    }
  }
  // the aligned strings of ALL slices: one stretch of the arena, sized once for the longest paths possible, so that
  // nothing moves while `ready` reads the finished slices (only what a slice really used is copied)
  size_t total_dense = 0;
  for (Slice& sl : slices) {
    sl.dense_base = total_dense;
    total_dense += sl.dense_cap;
  }
  const size_t arena_base = arena.size();
  arena.resize(arena_base + total_dense);  // (page-locked: the copies land in place)
  tm.lap("lanes: slices");

  struct InFlight {
    size_t np = 0, ctl_at = 0;
    const uint8_t* d_dense = nullptr;
#ifdef SW_LANES_TIMELINE
    const unsigned long long* d_timeline = nullptr;
    size_t n_units = 0;
#endif
  } fl[2];
  // a slice has its stream: the copies of one overlap the kernels of the other
  hipStream_t streams[2] = {c.stream, c.stream2};
  auto align64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
  double t_stage[5] = {0, 0, 0, 0, 0};  // (SW_HIP_TRACE) lanes table, buffers, unit table, symbol pool, HIP calls
  auto now_ms = [&]() { return tm.on ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0; };
  // stages slice `si` and queues it (copies in, kernels, copies out) on workspace `ws`; does not wait
  auto begin = [&](size_t si, Workspace& ws, InFlight& f) -> int {
    Slice& sl = slices[si];
    const hipStream_t stream = streams[si & 1];
    const size_t p0 = sl.g0 * 64, p1 = std::min(el.size(), sl.g1 * 64), np = p1 - p0, ng = sl.g1 - sl.g0;
    f.np = np;
    double t_a = now_ms();
    auto lapse = [&](int k) {
      const double t_b = now_ms();
      t_stage[k] += t_b - t_a;
      t_a = t_b;
    };
    // where a pair's walker stages its strings (sw_lanes.hip): a running sum, the one serial step of the table
    std::vector<uint64_t> stage_off(np + 1);
    {
      uint64_t at = 0;
      for (size_t i = p0; i < p1; i++) {  // (el only: sequential memory)
        stage_off[i - p0] = at;
        at += (2 * ((uint64_t)el[i].Q + el[i].T) + 15) & ~(uint64_t)15;
        sl.cells += (uint64_t)el[i].Q * el[i].Tfull;
      }
      stage_off[np] = at;
    }
    const size_t ops_bytes = stage_off[np];
    // symbol pool of the slice (unless the whole batch's block is on the device): every distinct sequence once,
    // 16-byte aligned
    std::unordered_map<const uint16_t*, uint32_t> seen;
    std::vector<uint32_t> by_id;
    struct SeqCopy {
      const uint16_t* p;
      uint32_t len, off;
    };
    std::vector<SeqCopy> copies;
    std::vector<uint32_t> qoffs, toffs;
    size_t pool_syms = 0;
    if (!whole_pool) {
      auto add_seq = [&](const uint16_t* p, uint32_t len, uint32_t id) -> uint32_t {
        if (id != UINT32_MAX) {
          if (id >= by_id.size()) by_id.resize((size_t)id + 1 + by_id.size() / 2, UINT32_MAX);
          if (by_id[id] != UINT32_MAX && copies[by_id[id]].p == p && copies[by_id[id]].len == len) return copies[by_id[id]].off;
        } else {
          auto it = seen.find(p);
          if (it != seen.end() && copies[it->second].len == len) return copies[it->second].off;
        }
        const uint32_t off = (uint32_t)pool_syms;
        copies.push_back(SeqCopy{p, len, off});
        pool_syms += ((size_t)len + 7) & ~(size_t)7;
        if (id != UINT32_MAX) by_id[id] = (uint32_t)copies.size() - 1;
        else seen[p] = (uint32_t)copies.size() - 1;
        return off;
      };
      qoffs.resize(np);
      toffs.resize(np);
      for (size_t i = p0; i < p1; i++) {
        const PairSpec& s = specs[el[i].k];
        qoffs[i - p0] = add_seq(s.q, s.Q, s.qid);
        toffs[i - p0] = add_seq(s.t, s.T, s.tid);
        if (pool_syms > 0x7fffff00u) return fail(SW_E_ARG, "too many symbols in one submission");
      }
      pool_syms += 64;  // (the kernel loads 32 bytes at a sequence's last 16-symbol boundary)
    }
    lapse(0);
    const size_t n_lanes = ng * 64;
    const size_t in_groups = 0, in_lanes = align64(ng * sizeof(LaneGroup)), in_units = in_lanes + align64(n_lanes * sizeof(LanePair)),
                 in_pool = in_units + align64(sl.n_units * 4);
    const size_t in_bytes = in_pool + align64(pool_syms * 2);
    // control words in front of the boundary columns: [0] ticket, [32] error word, [33] bytes of packed strings
    constexpr size_t kCtl = 256;
    const size_t out_ops = align64(np * sizeof(LaneOut)), out_dense = out_ops + align64(ops_bytes),
                 out_bytes = out_dense + align64(sl.dense_cap);
    int erc;
    if ((erc = ws.xin.ensure(in_bytes))) return erc;
    if ((erc = ws.h_in.ensure(in_bytes))) return erc;
    if ((erc = ws.dirs.ensure(std::max<size_t>(sl.dir_words * 4, 16)))) return erc;
    if ((erc = ws.xzero.ensure(kCtl + sl.hb_words * 4))) return erc;
    if ((erc = ws.xrecs.ensure(std::max<size_t>(sl.n_units * 64 * sizeof(LaneRec), 16)))) return erc;
    if ((erc = ws.xout.ensure(out_bytes))) return erc;
    if ((erc = ws.h_out.ensure(out_ops + 16))) return erc;
    f.ctl_at = out_ops;
    uint8_t* const h_in = (uint8_t*)ws.h_in.p;
    lapse(1);
    {
      // Tickets strip level by level: a strip's left neighbour always holds an earlier ticket (no deadlock).  Measured
      // against it on 8 000 x 1 kb (`scratch/lanes_timeline.sh`, `profiles/r03_lanes_timeline.txt`): the groups
      // staggered (key = strip + group x strips / groups, so that the launch ends with the deep strips of the last
      // groups only) 4.2 instead of 3.9 ms; issue priority (s_setprio) rising with the strip's depth 4.6, falling
      // 3.9, high for a strip's first 64 rows 3.9.
      uint32_t* units = (uint32_t*)(h_in + in_units);
      uint32_t max_strips = 0;
      for (size_t g = sl.g0; g < sl.g1; g++) max_strips = std::max(max_strips, groups[g].nstrips);
      size_t u = 0;
      for (uint32_t st = 0; st < max_strips; st++)
        for (size_t g = sl.g0; g < sl.g1; g++)
          if (groups[g].nstrips > st) units[u++] = (st << 20) | (uint32_t)(g - sl.g0);
    }
    memcpy(h_in + in_groups, groups.data() + sl.g0, ng * sizeof(LaneGroup));
    lapse(2);
    // the lanes' table, straight into the staging buffer
    LanePair* const hl = (LanePair*)(h_in + in_lanes);
    parallel_ranges(n_lanes, n_lanes * 64, [&](size_t lo, size_t hi) {
      for (size_t l = lo; l < hi; l++) {
        LanePair lp{};
        if (l < np) {
          const Elig& e = el[p0 + l];
          const PairSpec& sp = specs[e.k];
          lp.qoff = whole_pool ? (uint32_t)(sp.q - syms->base) : qoffs[l];
          lp.toff = whole_pool ? (uint32_t)(sp.t - syms->base) : toffs[l];
          lp.Q = e.Q;
          lp.T = e.T;
          lp.pair = (uint32_t)l;
          lp.ops_off = stage_off[l];
        }
        hl[l] = lp;
      }
    });
    lapse(0);
    if (!whole_pool) {
      uint16_t* const hpool = (uint16_t*)(h_in + in_pool);
      parallel_ranges(copies.size(), pool_syms * 2, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) memcpy(hpool + copies[k].off, copies[k].p, (size_t)copies[k].len * 2);
      });  // (what lies between the sequences is never looked at: the kernel masks by the lengths)
    }
    lapse(3);
    HIPCK(hipMemcpyAsync(ws.xin.p, h_in, in_bytes, hipMemcpyHostToDevice, stream));
    // (a word of a boundary column is "published" by the tag in its low half, so the columns start at zero)
    HIPCK(hipMemsetAsync(ws.xzero.p, 0, kCtl + sl.hb_words * 4, stream));
    HIPCK(hipEventRecord(ws.ev_k0, stream));
    LanesWork w{};
    w.groups = (const LaneGroup*)((uint8_t*)ws.xin.p + in_groups);
    w.lanes = (const LanePair*)((uint8_t*)ws.xin.p + in_lanes);
    w.pool = whole_pool ? (const uint16_t*)c.sympool.p : (const uint16_t*)((uint8_t*)ws.xin.p + in_pool);
    w.dirs = (uint32_t*)ws.dirs.p;
    w.hb = (uint32_t*)((uint8_t*)ws.xzero.p + kCtl);
    w.recs = (LaneRec*)ws.xrecs.p;
    w.units = (const uint32_t*)((uint8_t*)ws.xin.p + in_units);
    w.ticket = (uint32_t*)ws.xzero.p;
    w.err = (uint32_t*)ws.xzero.p + 32;
    w.dense_used = (uint32_t*)ws.xzero.p + 33;
    w.wait_ticks = exact_wait_ticks();
    w.n_units = (uint32_t)sl.n_units;
    w.ops = (uint8_t*)ws.xout.p + out_ops;
    w.dense = (uint8_t*)ws.xout.p + out_dense;
    w.outs = (LaneOut*)ws.xout.p;
    w.n_groups = (uint32_t)ng;
    f.d_dense = w.dense;
#ifdef SW_LANES_TIMELINE
    static DevBuf tl_buf[2];
    if ((erc = tl_buf[si & 1].ensure(sl.n_units * 32 + 64))) return erc;
    HIPCK(hipMemsetAsync(tl_buf[si & 1].p, 0, sl.n_units * 32, stream));
    w.timeline = (unsigned long long*)tl_buf[si & 1].p;
    f.d_timeline = w.timeline;
    f.n_units = sl.n_units;
#endif
    HIPCK(launch_lanes(w, sc, true, stream));
    HIPCK(hipEventRecord(ws.ev_k1, stream));
    HIPCK(hipMemcpyAsync(ws.h_out.p, ws.xout.p, np * sizeof(LaneOut), hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync((uint8_t*)ws.h_out.p + out_ops, w.err, 8, hipMemcpyDeviceToHost, stream));
    HIPCK(hipEventRecord(ws.ev_done, stream));
    lapse(4);
    return SW_OK;
  };
  // waits for slice `si` and hands its pairs to the caller
  std::vector<uint32_t> ks;
  auto end = [&](size_t si, Workspace& ws, InFlight& f) -> int {
    Slice& sl = slices[si];
    const size_t p0 = sl.g0 * 64, p1 = p0 + f.np;
    const hipStream_t stream = streams[si & 1];
    HIPCK(hipEventSynchronize(ws.ev_done));
    const uint32_t* ctl = (const uint32_t*)((const uint8_t*)ws.h_out.p + f.ctl_at);
    if (ctl[0] != 0) return fail(SW_E_HIP, "lane kernel: strip pipeline wait timed out");
    // the strings the slice really has (the other slice's kernels run meanwhile, on the other stream)
    const size_t dense_used = ctl[1];
    if (dense_used > sl.dense_cap) return fail(SW_E_HIP, "lane kernel: packed strings beyond their bound");
    if (dense_used) {
      HIPCK(hipMemcpyAsync(arena.data() + arena_base + sl.dense_base, f.d_dense, dense_used, hipMemcpyDeviceToHost, stream));
      HIPCK(hipStreamSynchronize(stream));
    }
#ifdef SW_LANES_TIMELINE
    if (const char* path = getenv("SW_LANES_TIMELINE_FILE")) {
      std::vector<unsigned long long> tl(f.n_units * 4);
      HIPCK(hipMemcpy(tl.data(), f.d_timeline, tl.size() * 8, hipMemcpyDeviceToHost));
      if (FILE* fp = fopen(path, "ab")) {
        const unsigned long long hdr[4] = {0x54494d454c494e45ull, f.n_units, si, 0};
        fwrite(hdr, 8, 4, fp);
        fwrite(tl.data(), 8, tl.size(), fp);
        fclose(fp);
      }
    }
#endif
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, ws.ev_k0, ws.ev_k1));
    c.stats.exact_kernel_ms += ms;
    c.stats.lanes_kernel_ms += ms;
    const LaneOut* ho = (const LaneOut*)ws.h_out.p;
    // (on the host threads: the pairs of a slice are scattered over the caller's tables)
    ks.resize(p1 - p0);
    std::atomic<size_t> handed_back{0};
    parallel_ranges(p1 - p0, (p1 - p0) * 128, [&](size_t lo, size_t hi) {
      size_t back = 0;
      for (size_t i = p0 + lo; i < p0 + hi; i++) {
        const LaneOut& o = ho[i - p0];
        // The path also computes H(1, Tlim + 1) <= match, a cell this kernel leaves out: it can only be the best
        // cell if nothing scored above match.  Such a pair goes to the whole-matrix path.
        if (el[i].row1_extra && o.score <= sc.match) {
          ks[i - p0] = UINT32_MAX;
          back++;
          continue;
        }
        PairOut& r = po[el[i].k];
        r = PairOut{};
        r.score = o.score;
        r.bx = o.bxbj & 0xffffu;
        r.bj = o.bxbj >> 16;
        r.nops = o.nops;
        r.ops_off = arena_base + sl.dense_base + (size_t)o.off16 * 16;
        r.strings = 1;
        r.key = o.score > 0 ? (((uint64_t)(r.bx + r.bj) << 31) | r.bx) : ~0ull;
        handled[el[i].k] = 1;
        ks[i - p0] = el[i].k;
      }
      if (back) handed_back.fetch_add(back);
    });
    if (handed_back.load()) ks.erase(std::remove(ks.begin(), ks.end(), UINT32_MAX), ks.end());
    c.stats.exact_cells += sl.cells;
    c.stats.lanes_cells += sl.cells;
    c.stats.lanes_pairs += ks.size();
    for (size_t g = sl.g0; g < sl.g1; g++) c.stats.lanes_padded_cells += (uint64_t)groups[g].Qmax * groups[g].nstrips * 16 * 64;
    if (ready && !ks.empty()) ready(ks.data(), ks.size());
    return SW_OK;
  };
  // (an error return must not leave copies in flight that write the arena or read h_in)
  StreamDrain drain(c.stream), drain2(c.stream2);
  if (whole_pool) {  // (both streams wait for it; the slices staged meanwhile queue behind it)
    if (!uploaded) {
      HIPCK(hipMemcpyAsync(c.sympool.p, syms->base, syms->syms * 2, hipMemcpyHostToDevice, c.stream));
      HIPCK(hipEventRecord(c.ev_pool, c.stream));
    }
    HIPCK(hipStreamWaitEvent(c.stream2, c.ev_pool, 0));
  }
  double t_begin = 0, t_end = 0;  // (SW_HIP_TRACE: host time staging slices / waiting for and handing over slices)
  auto timed = [&](double& acc, auto&& fn) -> int {
    if (!tm.on) return fn();
    const auto a = std::chrono::steady_clock::now();
    const int r = fn();
    acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    return r;
  };
  int rc = timed(t_begin, [&] { return begin(0, *wss[0], fl[0]); });
  for (size_t si = 0; si < slices.size() && !rc; si++) {
    int rc_next = SW_OK;
    if (si + 1 < slices.size()) rc_next = timed(t_begin, [&] { return begin(si + 1, *wss[(si + 1) & 1], fl[(si + 1) & 1]); });
    rc = timed(t_end, [&] { return end(si, *wss[si & 1], fl[si & 1]); });
    if (!rc) rc = rc_next;
  }
  if (tm.on)
    fprintf(stderr, "[sw_hip]   lanes: %zu slices, staging %.3f ms (table %.3f, buffers %.3f, units %.3f, symbols %.3f, HIP calls %.3f), waits + hand-over %.3f ms\n",
            slices.size(), t_begin, t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], t_end);
  if (rc) return rc;
  drain.dismiss();
  drain2.dismiss();
  drain_upload.dismiss();
  tm.lap("lanes: submissions");
  (void)padded_cells;
  return SW_OK;
}

}  // namespace swhip
