// dlsa_amd/csrc/sw_lanes.hip -- exact Smith-Waterman for BIG BATCHES of short pairs on gfx950: one pair per LANE.
//
// What it replaces: the same reference functions as sw_exact.hip (matrix fill rust/src/algorithm/mod.rs:66-202 /
// :204-445, best cell rust/src/lib.rs:122-135, traceback algorithm/mod.rs:637-714), for the record-per-pair
// protocol of the worker (internal/worker/worker_impl.go:108-168) when a work package holds tens of thousands of
// pairs of reads: there the row-strip wavefront of sw_exact.hip wastes lanes (a 150-row read fills 38 of 64) and
// steps (the 64-step skew), and needs a DPP shift, a hand-over and a compare + select per cell.
//
// Formulation (inter-task, like the score kernel, but exact and with a traceback): a wave holds 64 PAIRS, one per
// lane, length-sorted by the host so that the lanes of a wave finish together.  A lane keeps a strip of 16 target
// columns of its own matrix in registers and walks down its query.  No lane ever talks to another lane: no DPP, no
// skew.  The 16-column STRIPS of a group are waves of their own (one wave per group and strip, tickets strip level
// by level), so that a few thousand pairs already fill the GPU: strip s + 1 takes the right-hand H column of strip
// s through one dword per row in HBM ([row][lane], coalesced, rewritten in place by every strip).  The low half of
// such a word is free and carries the number of the strip that wrote it, so a word is its own "published" flag
// (a dword store is atomic; the column is zeroed before the launch): the consumer prefetches the 8 rows of its next
// block while it computes the current one and re-polls only rows whose tag is still the previous strip's.  With
// more groups than resident waves the producer finished long ago and nothing ever waits.
//
// Arithmetic (int32, exact on the domain the host admits: min(Q, T) x match <= 32767, Q + T < 65535): a value is
//   V = H << 16 | tag.  The three candidates of a cell carry tag 3 (diagonal), 2 (up), 1 (left) in their low bits,
// so ONE v_max3_i32 yields the new H and, in its low bits, which candidate won with the reference's priority among
// equal values (diagonal, then up, then left: algorithm/mod.rs:665-706); the saturating subtract that applies the
// gap is also the recurrence's max(., 0).  The two direction bits enter the row's direction word with one
// v_alignbit_b32 (16 columns x 2 bits = exactly one dword per row and strip).  The low half of a cleaned value is
// free, so the first maximal cell of every COLUMN is kept with one OR and one unsigned max: the key
// H << 16 | (0xffff - row) is largest for the largest H and, among equal H, the smallest row.  Of the cells that
// reach the pair's maximum the reference takes the one with the smallest x + j, then the smallest x (first flat
// index of the skewed matrix, lib.rs:122; ring path: strict row maxima in row order, mod.rs:358-362); within a
// column that is the smallest row, so the column records are enough and the strip's 16 records are folded into the
// lane's best (value, x + j, x) once per strip.
// Instruction mix per cell (profiles/r03_valu_issue.txt: add / sub / and / or / xor issue at 2 cycles per wave,
// max / cmp / cndmask / alignbit / every packed op at 4): v_cmp + v_cndmask + v_add (diag + s), two v_sub_u32 clamp,
// v_max3_i32, v_alignbit, v_and, v_or, v_max_u32 = 10 instructions, 30 issue cycles -- against 12 instructions / 38
// cycles plus per-step overhead in the wavefront kernel, with every lane busy.
//
// Cells outside a lane's matrix (rows >= Q, columns >= T of a wave whose other lanes are longer) are computed with
// sentinel symbols that never match; with miss < 0 and gap < 0 such a cell is strictly smaller than some real cell
// (or 0), so it never wins the maximum, and the walker never visits it.
#include <type_traits>

#include "sw_internal.h"

namespace swhip {

namespace {

constexpr uint32_t kSentQ = 0xffffu, kSentT = 0xfffeu;  // (the host admits only symbols below 0xfffe)

template <class T>
using gptr = __attribute__((address_space(1))) T*;

__device__ __forceinline__ uint32_t sat_sub(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
// symbol i (0..7) of eight u16 held in four dwords
__device__ __forceinline__ uint32_t sym8(const u4& w, int i) {
  const uint32_t d = i < 2 ? w.x : (i < 4 ? w.y : (i < 6 ? w.z : w.w));
  return (i & 1) ? (d >> 16) : (d & 0xffffu);
}

// Rows per hand-over block: the strip to the right may start a block once the whole block has been published, so
// the block is what a strip trails its left neighbour by (plus one block of prefetch and the memory latency);
// 4 rows instead of 8 took the 63-strip pipelines of 1 kb pairs from 3.0 to 2.x ms (profiles/r03_lanes_*).
#ifndef SW_LANES_RB
#define SW_LANES_RB 4
#endif
constexpr int RB = SW_LANES_RB;
// back-off of a wave that waits for its left neighbour, in units of 64 clocks: a polling wave takes issue slots from
// the working waves of its SIMD (one poll is ~60 instructions), so it sleeps several hundred clocks between polls
#ifndef SW_LANES_SLEEP
#define SW_LANES_SLEEP 16
#endif
static_assert(RB == 4 || RB == 8, "a block is 4 or 8 rows: 8 or 16 bytes of query symbols");
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef std::conditional<RB == 8, u4, u2v>::type qv_t;
__device__ __forceinline__ uint32_t symq(const u4& w, int i) { return sym8(w, i); }
__device__ __forceinline__ uint32_t symq(const u2v& w, int i) {
  const uint32_t d = i < 2 ? w.x : w.y;
  return (i & 1) ? (d >> 16) : (d & 0xffffu);
}

struct CellK {
  int scM, scX;          // (match << 16) | 3, (miss << 16) | 3
  uint32_t lsub, usub;   // (-gap << 16) - 1, (-gap << 16) - 2
};

// One row of the strip: 16 cells, left to right.  hl = H(r, first column - 1), hd = H(r - 1, first column - 1).
__device__ __forceinline__ void lane_row(uint32_t q, const uint32_t (&t)[16], uint32_t (&H)[16], uint32_t (&colkey)[16],
                                         uint32_t hl, uint32_t hd, uint32_t stamp, const CellK& K, uint32_t& acc,
                                         uint32_t& hright) {
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const uint32_t hu = H[c];
    const int sc = (q == t[c]) ? K.scM : K.scX;
    const int d = (int)hd + sc;
    const int lf = (int)sat_sub(hl, K.lsub);
    const int up = (int)sat_sub(hu, K.usub);
    const int ht = max(max(d, lf), up);
    acc = __builtin_amdgcn_alignbit((uint32_t)ht, acc, 2);  // column c ends up in bits [2c, 2c + 2)
    const uint32_t hn = (uint32_t)ht & 0xffff0000u;
    colkey[c] = max(colkey[c], hn | stamp);
    hd = hu;
    hl = hn;
    H[c] = hn;
  }
  hright = hl;
}

}  // namespace

// One wave per (group of 64 pairs, 16-column strip).  Tickets are drawn when a wave starts, strip level by level
// (LanesWork::units), so the wave of the strip to my left holds an earlier ticket: it is resident or has finished,
// whatever the dispatch order, and the pipeline of strips cannot deadlock.  A wait that never ends (a bug, not a legal
// state) is bounded in time, stores the error word and ends the wave; every other waiting wave sees the word and
// ends too (same containment as sw_exact.hip).
__global__ void __launch_bounds__(64) sw_lanes_kernel(LanesWork W, int match, int miss, int gap) {
  const uint32_t lane = threadIdx.x;
  uint32_t tk = 0;
  if (lane == 0) tk = atomicAdd(W.ticket, 1u);
  const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
  if (ticket >= W.n_units) return;
  const uint32_t unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.units[ticket]);
  const uint32_t g = unit & 0xfffffu, s = unit >> 20;
  const LaneGroup G = W.groups[g];
  const LanePair P = W.lanes[(size_t)g * 64 + lane];
  const gptr<const uint16_t> pool = (gptr<const uint16_t>)W.pool;
  const gptr<uint32_t> dirs = (gptr<uint32_t>)W.dirs + G.dirs_off + lane;
  const gptr<uint32_t> hb = (gptr<uint32_t>)W.hb + G.hb_off + lane;
  const uint32_t Q = P.Q, T = P.T;  // 0: idle lane (every symbol a sentinel)
  CellK K;
  K.scM = (int)(((uint32_t)match << 16) | 3u);
  K.scX = (int)(((uint32_t)miss << 16) | 3u);
  K.lsub = ((uint32_t)(-gap) << 16) - 1u;
  K.usub = ((uint32_t)(-gap) << 16) - 2u;
  const uint32_t nblk = G.Qmax / RB;  // (Qmax is a multiple of 8)
#ifndef SW_LANES_TIMING_NO_WAIT
  const uint32_t tag_in = s, tag_out = s + 1;
#else  // (timing experiment: every tag reads as published -- results are garbage)
  const uint32_t tag_in = 0, tag_out = 0;
#endif  // a word of the boundary column carries the strip that wrote it, + 1
  const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + W.wait_ticks;
#ifdef SW_LANES_TIMELINE
  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl_wait = 0, tl_first = 0;
#endif

  // the strip's 16 target symbols
  uint32_t t[16];
  {
    u4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (s * 16 < T) {
      const gptr<const u4> tp = (gptr<const u4>)(pool + P.toff + s * 16);
      a = tp[0];
      b = tp[1];
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const uint32_t v = c < 8 ? sym8(a, c) : sym8(b, c - 8);
      t[c] = (s * 16 + (uint32_t)c < T) ? v : kSentT;
    }
  }
  uint32_t H[16], colkey[16];
#pragma unroll
  for (int c = 0; c < 16; c++) H[c] = 0, colkey[c] = 0;
  uint32_t hdiag = 0;  // H(r - 1, first column - 1)
  // device-scope accesses: the producer may sit on another XCD (no coherent L2 between them)
  auto load_hb = [&](uint32_t r) -> uint32_t {
    return __hip_atomic_load(hb + (size_t)r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // boundary column and query symbols of the first block
  uint32_t hbn[RB];
  qv_t qn = {};
#pragma unroll
  for (int i = 0; i < RB; i++) hbn[i] = s ? load_hb((uint32_t)i) : 0u;
  if (0 < Q) qn = *(gptr<const qv_t>)(pool + P.qoff);
  for (uint32_t rb = 0; rb < nblk; rb++) {
    if (s) {
      // Every row of this block must come from the strip to my left; normally it does (fetched a block ago).  The
      // re-poll is a slow path of its own that ends with nothing of its own in flight, so that the fast path carries
      // no wait for it: a wait after the merge would also wait for the previous block's stores.
      auto stale = [&]() -> bool {
        uint32_t bad = 0;
#pragma unroll
        for (int i = 0; i < RB; i++) bad |= hbn[i] ^ tag_in;
        return __builtin_amdgcn_ballot_w64((bad & 0xffffu) != 0) != 0;
      };
      if (stale()) {
#ifdef SW_LANES_TIMELINE
        const unsigned long long tl_w0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (uint32_t spin = 0;; spin++) {
          if ((spin & 63u) == 63u) {  // (both conditions wave-uniform: s_endpgm must sit behind a scalar branch)
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(W.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const bool late = __builtin_amdgcn_s_memrealtime() >= deadline;
            if (e != 0 || late) {
              if (e == 0) __hip_atomic_store(W.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              asm volatile("s_endpgm" ::: "memory");
            }
          }
          __builtin_amdgcn_s_sleep(SW_LANES_SLEEP);
#pragma unroll
          for (int i = 0; i < RB; i++)
            if ((hbn[i] ^ tag_in) & 0xffffu) hbn[i] = load_hb(rb * RB + (uint32_t)i);
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads above (and everything older) have landed
          if (!stale()) break;
        }
#ifdef SW_LANES_TIMELINE
        tl_wait += __builtin_amdgcn_s_memrealtime() - tl_w0;
#endif
      }
    }
#ifdef SW_LANES_TIMELINE
    if (rb == 0) tl_first = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t hbc[RB];
#pragma unroll
    for (int i = 0; i < RB; i++) hbc[i] = hbn[i] & 0xffff0000u;
    const qv_t qc = qn;
    if (rb + 1 < nblk) {  // next block's inputs, in flight while this one is computed
      if (s) {
#pragma unroll
        for (int i = 0; i < RB; i++) hbn[i] = load_hb((rb + 1) * RB + (uint32_t)i);
      }
      if ((rb + 1) * RB < Q) qn = *(gptr<const qv_t>)(pool + P.qoff + (rb + 1) * RB);
    }
#pragma unroll
    for (int i = 0; i < RB; i++) {
      const uint32_t r = rb * RB + (uint32_t)i;
      const uint32_t q = r < Q ? symq(qc, i) : kSentQ;
      uint32_t acc = 0, hright;
      lane_row(q, t, H, colkey, hbc[i], hdiag, 0xffffu - r, K, acc, hright);
      hdiag = hbc[i];
#ifndef SW_LANES_TIMING_NO_DIRS  // (timing experiments only: scratch/lanes_variants.sh)
      dirs[((size_t)s * G.Qmax + r) * 64] = acc;
#else
      if (acc == 0x12345678u && hright == 77u) dirs[0] = acc;
#endif
      // (the last strip stores too: a test here makes the compiler split the row loop and costs 80 registers)
      __hip_atomic_store(hb + (size_t)r * 64, hright | tag_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // fold the strip's column records into the lane's best cell of this strip; the walker combines the strips
  uint32_t bestH = 0, bestkey = 0xffffffffu;  // key = (x + j) << 16 | x, 1-based
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const uint32_t v = colkey[c] >> 16;
    const uint32_t x = 0x10000u - (colkey[c] & 0xffffu);  // row + 1
    const uint32_t j = s * 16 + (uint32_t)c + 1;
    const uint32_t key = ((x + j) << 16) | x;
    const bool better = v > bestH || (v == bestH && v > 0 && key < bestkey);
    bestH = better ? v : bestH;
    bestkey = better ? key : bestkey;
  }
  LaneRec rec;
  rec.score = bestH;
  rec.key = bestkey;
  W.recs[((size_t)G.rec_off + s) * 64 + lane] = rec;
#ifdef SW_LANES_TIMELINE
  if (lane == 0 && W.timeline) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = W.timeline + (size_t)ticket * 4;
    t[0] = tl_start;
    t[1] = __builtin_amdgcn_s_memrealtime();
    t[2] = (tl_wait & 0xffffffffull) | ((tl_first - tl_start) << 32);  // (both well below 2^32 ticks of 10 ns)
    t[3] = ((unsigned long long)((xcc & 0xfu) << 28 | (hw & 0x0fffffffu)) << 32) | unit;
  }
#endif
}

// Traceback: one LANE per pair follows the direction bits from the best cell until H reaches 0
// (rust/src/algorithm/mod.rs:663-664) or an edge (:650).  It writes the two ALIGNED STRINGS themselves, not move
// codes: a pair's region holds `cap` bytes for the query alignment followed by `cap` for the target alignment, and
// the walk -- which meets the last alignment column first -- fills both from their ends backwards, so that the
// finished strings are the last nops bytes of each half, in reading order: column (q[x-1], t[j-1]) for a diagonal
// move, (q[x-1], '-') for a move up, ('-', t[j-1]) for a move left (algorithm/mod.rs:677-706).  Symbols are ASCII
// on this path (the host admits nothing else), so a symbol is its byte.  The staging regions are sized for the
// longest possible path (Q + T columns) and stay on the device: the wave then reserves, with one atomic, room for
// the strings its 64 pairs really have in LanesWork::dense, and sw_lanes_pack_kernel moves them there -- the host
// copies back what was used (a tenth of the staging for unrelated reads) and memcpy's two strings per pair.
// The walk is a chain of dependent loads, tens of thousands of chains in flight.
__global__ void __launch_bounds__(64) sw_lanes_walk_kernel(LanesWork W, int match, int miss, int gap) {
  const uint32_t lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  if (g >= W.n_groups) return;
  const LaneGroup G = W.groups[g];
  const LanePair P = W.lanes[(size_t)g * 64 + lane];  // (an idle lane, Q == 0, stays for the wave-wide reservation)
  const gptr<const uint16_t> pool = (gptr<const uint16_t>)W.pool;
  const gptr<const uint32_t> dirs = (gptr<const uint32_t>)W.dirs + G.dirs_off + lane;
  const uint32_t cap = P.Q + P.T;
  const gptr<uint8_t> qa_end = (gptr<uint8_t>)W.ops + P.ops_off + cap;       // one past the query alignment's half
  const gptr<uint8_t> ta_end = qa_end + cap;                                 // ... the target alignment's
  // the pair's best cell: largest H, then the reference's discovery order (smallest x + j, then smallest x)
  uint32_t bestH = 0, bestkey = 0xffffffffu;
  for (uint32_t s = 0; s < G.nstrips; s++) {
    const LaneRec r = W.recs[((size_t)G.rec_off + s) * 64 + lane];
    const bool better = r.score > bestH || (r.score == bestH && r.score > 0 && r.key < bestkey);
    bestH = better ? r.score : bestH;
    bestkey = better ? r.key : bestkey;
  }
#ifdef SW_LANES_TIMING_NO_WAIT  // (timing experiment: the fill computed garbage, nothing may be walked)
  bestH = 0;
#endif
  if (!P.Q) bestH = 0;
  LaneOut o;
  o.score = (int32_t)bestH;
  const uint32_t bx = bestH ? (bestkey & 0xffffu) : 0, bj = bestH ? ((bestkey >> 16) - (bestkey & 0xffffu)) : 0;
  o.bxbj = bx | (bj << 16);
  int H = o.score;
  uint32_t x = bx, j = bj, n = 0;
  bool broken = false;
  // A BLOCK of the path per round of loads: the direction words and query symbols of 4 rows up from the current
  // cell (same strip) and 5 target symbols to its left are fetched together -- 13 independent loads, one memory
  // latency -- and the walk then runs inside them until it leaves the 4 x 5 block or the strip: about four moves per
  // latency instead of one.
  while (H > 0 && x > 0 && j > 0 && n < cap && !broken) {
    const uint32_t r0 = x - 1, c0 = j - 1, s = c0 >> 4;
    uint32_t w[4], qs[4], ts[5];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const uint32_t row = r0 >= i ? r0 - i : 0;
      w[i] = dirs[((size_t)s * G.Qmax + row) * 64];
      qs[i] = pool[P.qoff + row];
    }
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) ts[k] = pool[P.toff + (c0 >= k ? c0 - k : 0)];
    uint32_t i = 0, k = 0;  // the current cell is (r0 - i, c0 - k)
    while (i < 4 && k < 5 && ((c0 - k) >> 4) == s) {
      const uint32_t wv = i == 0 ? w[0] : (i == 1 ? w[1] : (i == 2 ? w[2] : w[3]));
      const uint32_t code = (wv >> (((c0 - k) & 15u) * 2)) & 3u;
      const uint32_t qv = i == 0 ? qs[0] : (i == 1 ? qs[1] : (i == 2 ? qs[2] : qs[3]));
      const uint32_t tv = k == 0 ? ts[0] : (k == 1 ? ts[1] : (k == 2 ? ts[2] : (k == 3 ? ts[3] : ts[4])));
      uint32_t qc, tc;  // this alignment column
      if (code == 3u) {
        H -= (qv == tv) ? match : miss;
        i++;
        k++;
        qc = qv;
        tc = tv;
      } else if (code == 2u) {
        H -= gap;
        i++;
        qc = qv;
        tc = '-';
      } else if (code == 1u) {
        H -= gap;
        k++;
        qc = '-';
        tc = tv;
      } else {
        broken = true;  // (code 0 with H > 0 cannot happen: every cell on the path had a winning candidate)
        break;
      }
      n++;
      qa_end[-(ptrdiff_t)n] = (uint8_t)qc;
      ta_end[-(ptrdiff_t)n] = (uint8_t)tc;
      if (H <= 0 || n >= cap || i > r0 || k > c0) break;  // done, or off the top / left edge
    }
    x = r0 + 1 - i;
    j = c0 + 1 - k;
  }
  o.nops = n;
  // room in the packed output: an inclusive scan of the lanes' needs, one atomic per wave
  const uint32_t need = P.Q ? lanes_dense_bytes(n) : 0;
  uint32_t incl = need;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
    if ((int)lane >= d) incl += up;
  }
  uint32_t base = 0;
  if (lane == 63) base = atomicAdd(W.dense_used, incl);
  base = (uint32_t)__shfl((int)base, 63, 64);
  o.off16 = (base + incl - need) >> 4;
  if (P.Q) W.outs[P.pair] = o;
}

// Moves the aligned strings from the walker's staging into their packed place: one wave per pair, four bytes per
// lane and step (the target alignment starts at a multiple of 4, so that every store is an aligned dword).
__global__ void __launch_bounds__(256) sw_lanes_pack_kernel(LanesWork W) {
  const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slot >= W.n_groups * 64) return;
  const LanePair P = W.lanes[slot];
  if (!P.Q) return;
  const LaneOut o = W.outs[P.pair];
  const uint32_t n = o.nops, cap = P.Q + P.T, n4 = (n + 3u) & ~3u;
  const gptr<const uint8_t> src_q = (gptr<const uint8_t>)W.ops + P.ops_off + cap - n;
  const gptr<const uint8_t> src_t = src_q + cap;
  const gptr<uint32_t> dst = (gptr<uint32_t>)(W.dense + (size_t)o.off16 * 16);
  for (uint32_t b = lane * 4; b < 2 * n4; b += 256) {
    const bool second = b >= n4;
    const gptr<const uint8_t> src = second ? src_t : src_q;
    const uint32_t at = second ? b - n4 : b;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (at + k < n) v |= (uint32_t)src[at + k] << (8 * k);
    dst[b >> 2] = v;
  }
}

hipError_t launch_lanes(const LanesWork& w, Scoring sc, bool want_trace, hipStream_t stream) {
  if (w.n_groups == 0 || w.n_units == 0) return hipSuccess;
  hipLaunchKernelGGL(sw_lanes_kernel, dim3(w.n_units), dim3(64), 0, stream, w, sc.match, sc.miss, sc.gap);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  (void)want_trace;  // (the walker also combines the strips' best cells: it always runs)
  hipLaunchKernelGGL(sw_lanes_walk_kernel, dim3(w.n_groups), dim3(64), 0, stream, w, sc.match, sc.miss, sc.gap);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sw_lanes_pack_kernel, dim3(w.n_groups * 16), dim3(256), 0, stream, w);
  return hipGetLastError();
}

}  // namespace swhip
