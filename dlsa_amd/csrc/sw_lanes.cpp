// dlsa_amd/csrc/sw_lanes.cpp -- host side of the one-pair-per-lane exact kernel (sw_lanes.hip): which pairs of a
// batch it may answer, whether that pays, the length-sorted groups of 64, one submission per slice of groups.
//
// Caller: align_interned (sw_lib.cpp), i.e. sw_align_batch / sw_execute_work / the combined per-pair callers -- the
// record-per-pair protocol of internal/worker/worker_impl.go:108-168.  Results are those of the whole-matrix path
// (sw_exact.hip) bit for bit: same score, same first maximal cell in the reference's discovery order, same
// traceback (tests/test_gpu_lanes.py compares both with the oracle).
#include <algorithm>
#include <cstring>
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
  bool row1_extra;  // the path also computes H(1, Tlim + 1) (T % 64 == 0 quirk): decided after the kernel
};

}  // namespace

int align_lanes(int device, const PairSpec* specs, size_t n, const uint8_t* small_syms, Scoring sc, PairOut* po,
                OpsArena& arena, uint8_t* handled, const LanesReady& ready) {
  const int mode = lanes_mode();
  if (mode == 0 || n == 0) return SW_OK;
  // the kernel's domain: sentinel cells lose only if every step off a match costs something
  if (!(sc.match > 0 && sc.miss < 0 && sc.gap < 0)) return SW_OK;
  StageTimer tm(n);
  std::vector<Elig> el;
  el.reserve(n);
  for (size_t k = 0; k < n; k++) {
    if (handled[k] || !small_syms[k]) continue;
    const PairSpec& s = specs[k];
    if (s.path == SW_PATH_STRAIGHT || s.left || s.col0 || s.Tfull || s.Qfull || s.walk_x) continue;
    if (s.Q == 0 || s.T == 0 || (uint64_t)s.Q + s.T > 65000u) continue;
    const PathParams pp = resolve_path(s.path, s.Q, s.T);
    if (pp.phaseQp || pp.straight || pp.Tlim == 0 || pp.T0 < pp.Tlim) continue;
    // V = H << 16 | tag must stay a non-negative int32
    if ((uint64_t)std::min(s.Q, pp.Tlim) * (uint64_t)sc.match > 32767u) continue;
    el.push_back(Elig{(uint32_t)k, s.Q, pp.Tlim, pp.T0 > pp.Tlim});
  }
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

  CtxLock L(device);
  if (L.rc) return L.rc;
  Context& c = *L.c;
  Workspace* wss[2] = {&c.ws, &c.ws2};
  for (Workspace* w : wss)
    for (hipEvent_t* e : {&w->ev_k0, &w->ev_k1, &w->ev_done})
      if (!*e && hipEventCreate(e) != hipSuccess) return fail(SW_E_HIP, "hipEventCreate");

  // Slices of groups: each fits the scratch budget (direction words dominate: rows x strips x 256 bytes per group),
  // and a big batch is cut into a few slices anyway so that TWO are in flight -- while slice i is on the GPU, slice
  // i + 1 is staged and queued behind it on the second workspace, and the caller's `ready` (string assembly, on
  // another thread) works on slice i - 1.
  struct Slice {
    size_t g0 = 0, g1 = 0, dir_words = 0, hb_words = 0, ops_base = 0, ops_bytes = 0, n_units = 0;
    std::vector<LanePair> lanes;
    uint64_t cells = 0;
  };
  std::vector<Slice> slices;
  {
    // (two slices at least for a batch worth pipelining; beyond that a slice keeps >= 2048 waves so that it still
    // fills the GPU's 1024 SIMDs twice over)
    const size_t want = el.size() >= 32768 ? std::min<size_t>(16, std::max<size_t>(2, el.size() / 131072)) : 1;
    const size_t per = (n_groups + want - 1) / want;
    size_t g0 = 0;
    while (g0 < n_groups) {
      Slice sl;
      sl.g0 = g0;
      size_t g1 = g0;
      while (g1 < n_groups && g1 - g0 < per && g1 - g0 < 0xfffffu && sl.n_units < 0x7ff00000u) {
        const size_t dw = (size_t)groups[g1].Qmax * groups[g1].nstrips * 64, hw = (size_t)groups[g1].Qmax * 64;
        if (g1 > g0 && (sl.dir_words + dw + sl.hb_words + hw) * 4 > c.scratch_budget / 2) break;
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
  // the traceback ops of ALL slices: one stretch of the arena, sized once, so that nothing moves while `ready` reads
  // the finished slices
  size_t total_ops = 0;
  for (Slice& sl : slices) {
    sl.ops_base = total_ops;
    const size_t p0 = sl.g0 * 64, p1 = std::min(el.size(), sl.g1 * 64);
    for (size_t i = p0; i < p1; i++) total_ops += ((size_t)el[i].Q + specs[el[i].k].T + 15) & ~(size_t)15;
  }
  const size_t arena_base = arena.size();
  arena.resize(arena_base + total_ops);  // (page-locked: the copies land in place)
  tm.lap("lanes: slices");

  struct InFlight {
    size_t np = 0, err_at = 0;
  } fl[2];
  auto align64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
  // stages slice `si` and queues it (copies in, kernels, copies out) on workspace `ws`; does not wait
  auto begin = [&](size_t si, Workspace& ws, InFlight& f) -> int {
    Slice& sl = slices[si];
    const size_t p0 = sl.g0 * 64, p1 = std::min(el.size(), sl.g1 * 64), np = p1 - p0, ng = sl.g1 - sl.g0;
    f.np = np;
    // symbol pool of the slice: every distinct sequence once, 16-byte aligned
    std::unordered_map<const uint16_t*, uint32_t> seen;
    std::vector<uint32_t> by_id;
    struct SeqCopy {
      const uint16_t* p;
      uint32_t len, off;
    };
    std::vector<SeqCopy> copies;
    size_t pool_syms = 0;
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
    sl.lanes.assign(ng * 64, LanePair{});
    size_t ops_bytes = 0;
    for (size_t i = p0; i < p1; i++) {
      const PairSpec& s = specs[el[i].k];
      LanePair& lp = sl.lanes[i - p0];
      lp.qoff = add_seq(s.q, s.Q, s.qid);
      lp.toff = add_seq(s.t, s.T, s.tid);
      if (pool_syms > 0x7fffff00u) return fail(SW_E_ARG, "too many symbols in one submission");
      lp.Q = el[i].Q;
      lp.T = el[i].T;
      lp.pair = (uint32_t)(i - p0);
      lp.ops_off = ops_bytes;
      ops_bytes += ((size_t)s.Q + s.T + 15) & ~(size_t)15;
      sl.cells += (uint64_t)s.Q * s.T;
    }
    sl.ops_bytes = ops_bytes;
    pool_syms += 64;  // (the kernel loads 32 bytes at a sequence's last 16-symbol boundary)
    const size_t in_groups = 0, in_lanes = align64(ng * sizeof(LaneGroup)), in_units = in_lanes + align64(sl.lanes.size() * sizeof(LanePair)),
                 in_pool = in_units + align64(sl.n_units * 4);
    const size_t in_bytes = in_pool + align64(pool_syms * 2);
    constexpr size_t kCtl = 256;  // control words in front of the boundary columns: [0] ticket, [32] error word
    const size_t out_ops = align64(np * sizeof(LaneOut)), out_bytes = out_ops + align64(ops_bytes);
    int erc;
    if ((erc = ws.xin.ensure(in_bytes))) return erc;
    if ((erc = ws.h_in.ensure(in_bytes))) return erc;
    if ((erc = ws.dirs.ensure(std::max<size_t>(sl.dir_words * 4, 16)))) return erc;
    if ((erc = ws.xzero.ensure(kCtl + sl.hb_words * 4))) return erc;
    if ((erc = ws.xrecs.ensure(std::max<size_t>(sl.n_units * 64 * sizeof(LaneRec), 16)))) return erc;
    if ((erc = ws.xout.ensure(out_bytes))) return erc;
    if ((erc = ws.h_out.ensure(out_ops + 16))) return erc;
    uint8_t* const h_in = (uint8_t*)ws.h_in.p;
    {  // tickets strip level by level: a strip's left neighbour always holds an earlier ticket
      uint32_t* units = (uint32_t*)(h_in + in_units);
      uint32_t max_strips = 0;
      for (size_t g = sl.g0; g < sl.g1; g++) max_strips = std::max(max_strips, groups[g].nstrips);
      size_t u = 0;
      for (uint32_t st = 0; st < max_strips; st++)
        for (size_t g = sl.g0; g < sl.g1; g++)
          if (groups[g].nstrips > st) units[u++] = (st << 20) | (uint32_t)(g - sl.g0);
    }
    memcpy(h_in + in_groups, groups.data() + sl.g0, ng * sizeof(LaneGroup));
    memcpy(h_in + in_lanes, sl.lanes.data(), sl.lanes.size() * sizeof(LanePair));
    uint16_t* const hpool = (uint16_t*)(h_in + in_pool);
    parallel_ranges(copies.size(), pool_syms * 2, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        memcpy(hpool + copies[k].off, copies[k].p, (size_t)copies[k].len * 2);
        const size_t pad = (((size_t)copies[k].len + 7) & ~(size_t)7) - copies[k].len;
        if (pad) memset(hpool + copies[k].off + copies[k].len, 0xff, pad * 2);
      }
    });
    memset(hpool + pool_syms - 64, 0xff, 128);
    HIPCK(hipMemcpyAsync(ws.xin.p, h_in, in_bytes, hipMemcpyHostToDevice, c.stream));
    // (a word of a boundary column is "published" by the tag in its low half, so the columns start at zero)
    HIPCK(hipMemsetAsync(ws.xzero.p, 0, kCtl + sl.hb_words * 4, c.stream));
    HIPCK(hipEventRecord(ws.ev_k0, c.stream));
    LanesWork w{};
    w.groups = (const LaneGroup*)((uint8_t*)ws.xin.p + in_groups);
    w.lanes = (const LanePair*)((uint8_t*)ws.xin.p + in_lanes);
    w.pool = (const uint16_t*)((uint8_t*)ws.xin.p + in_pool);
    w.dirs = (uint32_t*)ws.dirs.p;
    w.hb = (uint32_t*)((uint8_t*)ws.xzero.p + kCtl);
    w.recs = (LaneRec*)ws.xrecs.p;
    w.units = (const uint32_t*)((uint8_t*)ws.xin.p + in_units);
    w.ticket = (uint32_t*)ws.xzero.p;
    w.err = (uint32_t*)ws.xzero.p + 32;
    w.wait_ticks = exact_wait_ticks();
    w.n_units = (uint32_t)sl.n_units;
    w.ops = (uint8_t*)ws.xout.p + out_ops;
    w.outs = (LaneOut*)ws.xout.p;
    w.n_groups = (uint32_t)ng;
    HIPCK(launch_lanes(w, sc, true, c.stream));
    HIPCK(hipEventRecord(ws.ev_k1, c.stream));
    HIPCK(hipMemcpyAsync(ws.h_out.p, ws.xout.p, np * sizeof(LaneOut), hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipMemcpyAsync((uint8_t*)ws.h_out.p + out_ops, w.err, 4, hipMemcpyDeviceToHost, c.stream));
    f.err_at = out_ops;
    if (ops_bytes)
      HIPCK(hipMemcpyAsync(arena.data() + arena_base + sl.ops_base, (uint8_t*)ws.xout.p + out_ops, ops_bytes, hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipEventRecord(ws.ev_done, c.stream));
    return SW_OK;
  };
  // waits for slice `si` and hands its pairs to the caller
  std::vector<uint32_t> ks;
  auto end = [&](size_t si, Workspace& ws, InFlight& f) -> int {
    Slice& sl = slices[si];
    const size_t p0 = sl.g0 * 64, p1 = p0 + f.np;
    HIPCK(hipEventSynchronize(ws.ev_done));
    if (*(const uint32_t*)((const uint8_t*)ws.h_out.p + f.err_at) != 0)
      return fail(SW_E_HIP, "lane kernel: strip pipeline wait timed out");
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, ws.ev_k0, ws.ev_k1));
    c.stats.exact_kernel_ms += ms;
    c.stats.lanes_kernel_ms += ms;
    const LaneOut* ho = (const LaneOut*)ws.h_out.p;
    ks.clear();
    for (size_t i = p0; i < p1; i++) {
      const LaneOut& o = ho[i - p0];
      // The path also computes H(1, Tlim + 1) <= match, a cell this kernel leaves out: it can only be the best
      // cell if nothing scored above match.  Such a pair goes to the whole-matrix path.
      if (el[i].row1_extra && o.score <= sc.match) continue;
      PairOut& r = po[el[i].k];
      r = PairOut{};
      r.score = o.score;
      r.bx = o.bx;
      r.bj = o.bj;
      r.nops = o.nops;
      r.ops_off = arena_base + sl.ops_base + sl.lanes[i - p0].ops_off;
      r.key = o.score > 0 ? (((uint64_t)(o.bx + o.bj) << 31) | o.bx) : ~0ull;
      handled[el[i].k] = 1;
      ks.push_back(el[i].k);
    }
    c.stats.exact_cells += sl.cells;
    c.stats.lanes_cells += sl.cells;
    c.stats.lanes_pairs += ks.size();
    for (size_t g = sl.g0; g < sl.g1; g++) c.stats.lanes_padded_cells += (uint64_t)groups[g].Qmax * groups[g].nstrips * 16 * 64;
    std::vector<LanePair>().swap(sl.lanes);
    if (ready && !ks.empty()) ready(ks.data(), ks.size());
    return SW_OK;
  };
  StreamDrain drain(c.stream);  // (an error return must not leave copies in flight that write the arena or read h_in)
  int rc = begin(0, *wss[0], fl[0]);
  for (size_t si = 0; si < slices.size() && !rc; si++) {
    int rc_next = SW_OK;
    if (si + 1 < slices.size()) rc_next = begin(si + 1, *wss[(si + 1) & 1], fl[(si + 1) & 1]);
    rc = end(si, *wss[si & 1], fl[si & 1]);
    if (!rc) rc = rc_next;
  }
  if (rc) return rc;
  drain.dismiss();
  tm.lap("lanes: submissions");
  (void)padded_cells;
  return SW_OK;
}

}  // namespace swhip
