// dlsa_amd/csrc/sw_exact.hip -- the exact (intra-task) Smith-Waterman kernels for gfx950.
//
// What it replaces: the matrix fill + best-cell selection + traceback of ONE pair, i.e.
// rust/src/algorithm/mod.rs:66-202 + rust/src/lib.rs:122-135 (full path), algorithm/mod.rs:204-445
// (ring path) and algorithm/mod.rs:518-549 + lib.rs:55-70 (straight), with the traceback priority of
// algorithm/mod.rs:665-706.  It is not a translation: the reference fills a skewed i16 matrix row by
// row; here a 64-lane wavefront sweeps a strip of 256 query rows (4 per lane) along the target as an
// anti-diagonal wavefront and passes H across lanes with a DPP wave shift; every strip of a pair has its own
// wave and the strips pipeline along the target across the whole GPU (strip boundary rows in HBM at device
// scope, each word carrying its own "published" tag in bit 31 so no separate progress flag is polled; tickets
// in (pair, strip) order); only 2 direction bits per cell are kept for the traceback walker.
//
// Arithmetic: int32 (the "int32 re-run" of the north star; equal to the reference's wrapping i16 on its
// safe domain max H <= 32767).
//
// Direction code, per cell, two bits: 3 = H came from the diagonal, 2 = from (x-1, j), 1 = from (x, j-1), with the
// reference's if/else-if priority among equal candidates (diagonal first, then up, then left); it falls out of the
// cell's v_max3 (see "arithmetic of one cell").  The walker knows H along the path (it subtracts what each step
// added), so no "zero" code is needed: 0 appears only where H = 0, and the walk stops there.
// Layout: word[((strip * nkb + k/16) * RPL + r) * 64 + lane], step k of that lane in bits
// [2*(k%16), +2); lane's row i = strip*256 + lane*RPL + r, its column at step k is j = k - lane + 1.
#include "sw_internal.h"

// back-off of the boundary poll (units of 64 clocks)
#ifndef SW_EXACT_SLEEP
#define SW_EXACT_SLEEP 4
#endif

namespace swhip {

namespace {

// rows per lane (RPL): 4, 2 or 1, chosen per launch (ExactWork::rpl)

// The pointers of an ExactPair are loaded from memory, so the compiler takes them for generic (flat) addresses; a flat
// access counts on lgkmcnt as well as vmcnt, which makes every wait for LDS or scalar data also wait for the stores
// in flight.  They all point to device memory: say so.
template <class T>
using gptr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ gptr<T> glb(T* p) {
  return (gptr<T>)p;
}

__device__ __forceinline__ int dpp_wave_shr1(int fill_lane0, int v) {
  // lane i <- lane i-1; lane 0 keeps `fill_lane0` (bound_ctrl = 0)
  return __builtin_amdgcn_update_dpp(fill_lane0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// lane i <- lane (i + K) mod 16 of its row of 16 lanes (row_ror:(16 - K)); K = 0 is the value itself
template <int K>
__device__ __forceinline__ int row_rol(int v) {
  if constexpr (K == 0) return v;
  else return __builtin_amdgcn_update_dpp(0, v, 0x120 + (16 - K), 0xf, 0xf, true);  // (bound_ctrl: no "old" value to set up)
}
__device__ __forceinline__ int row_elem_to_lane0(int v, int k) {  // k is a constant after unrolling
  switch (k & 15) {
    case 0: return row_rol<0>(v);
    case 1: return row_rol<1>(v);
    case 2: return row_rol<2>(v);
    case 3: return row_rol<3>(v);
    case 4: return row_rol<4>(v);
    case 5: return row_rol<5>(v);
    case 6: return row_rol<6>(v);
    case 7: return row_rol<7>(v);
    case 8: return row_rol<8>(v);
    case 9: return row_rol<9>(v);
    case 10: return row_rol<10>(v);
    case 11: return row_rol<11>(v);
    case 12: return row_rol<12>(v);
    case 13: return row_rol<13>(v);
    case 14: return row_rol<14>(v);
    default: return row_rol<15>(v);
  }
}

// Discovery order of the reference as one integer: the cell with the SMALLER key is found first among
// cells of equal H.  x and j are 1-based and < 2^31.
__device__ __forceinline__ uint64_t order_key(uint32_t x, uint32_t j, uint32_t T, uint32_t phaseQp,
                                              uint32_t straight) {
  if (straight) return ((uint64_t)j << 31) | x;  // first flat index j*width + x, rust/src/lib.rs:55
  uint64_t p = 0;
  // rust/src/algorithm/mod.rs:249-298 runs before :385-442 when the lane phase (:313) is empty
  if (phaseQp) p = !((j < T) && (x + j <= phaseQp));
  // skewed row y = x + j; first flat index y*width + x, rust/src/lib.rs:122
  return (p << 63) | ((uint64_t)(x + j) << 31) | x;
}

// ---- strip-to-strip hand-over ---------------------------------------------------------------------------
// The bottom row of strip s is the top boundary of strip s+1.  It travels through one HBM row per strip; the
// waves may sit on different CUs and XCDs, so stores and loads are device-scope.  There are no flags: every
// boundary word carries its own "published" tag in bit 31 (H is a non-negative int32, so the bit is free; the
// rows are zeroed before the launch).  The producer stores 16 tagged values every 16 steps and never waits for
// them; the consumer fetches the 16 values of its NEXT 16-step block while it computes the current one and
// re-polls only if a tag is still missing.  A dword store is atomic, so a tagged word is a complete value.
constexpr uint32_t kTag = 0x80000000u;

// ---- arithmetic of one cell ---------------------------------------------------------------------------------
// H is kept times four with a two-bit TAG in the low bits of each of the three candidates: 3 = diagonal, 2 = up
// (x-1, j), 1 = left (x, j-1).  One v_max3 then yields the new H and, in its low bits, WHICH candidate won with the
// reference's priority among equal values (diagonal first, then up, then left: rust/src/algorithm/mod.rs:665-706) --
// the direction code costs no comparison.  Per cell: select + add (diag + s, constants pre-tagged), one saturating
// subtract each for left and up (4 |gap| minus the tag; saturation at 0 is the max(., 0) of the recurrence, so the
// result needs no clamp), v_max3, one AND to clear the tag, shift + v_bfi to append the code to the lane's
// direction word.  Code 0 appears only where H = 0, where the walker never looks (it stops when H reaches 0).
// H * 4 must stay below 2^31 (bit 31 is the hand-over tag): the host refuses pairs whose score could reach 2^29.
struct CellConsts {
  int match4, miss4;     // 4 * score + 3
  int left4, up4;        // 4 * gap + 1, 4 * gap + 2 (signed form, masked variant)
  uint32_t lsub, usub;   // -4 * gap - 1, -4 * gap - 2 (saturating form, gap < 0)
};
__device__ __forceinline__ CellConsts cell_consts(int match, int miss, int gap) {
  CellConsts c;
  c.match4 = 4 * match + 3;
  c.miss4 = 4 * miss + 3;
  c.left4 = 4 * gap + 1;
  c.up4 = 4 * gap + 2;
  c.lsub = (uint32_t)(-4 * gap - 1);
  c.usub = (uint32_t)(-4 * gap - 2);
  return c;
}
__device__ __forceinline__ int sat_sub(int a, uint32_t b) {  // max(a - b, 0) for a >= 0: v_sub_u32 clamp
  return (int)__builtin_elementwise_sub_sat((uint32_t)a, b);
}

// One 256-row strip of one pair per WAVE.  Tickets are drawn strip level by level (every pair's strip 0, then
// every strip 1, ...: W.unit_order), so the strip above mine always belongs to an earlier ticket: its wave is
// resident or has finished, whatever the dispatch order, and the chain of strips cannot deadlock.  The strips of
// ONE long pair run concurrently as a pipeline along the target, spread over the whole GPU; strip s trails strip
// s-1 by about 100 steps (64 lanes of skew + one 16-column hand-over block + the prefetch distance).  In a big
// submission the level order puts thousands of tickets between a strip and the one above: no trailing, no polls.
//
// 16 steps of the wavefront.  MASK = false is the interior of the matrix (every column any lane touches is
// <= min(Tlim, T0), miss < 0 and gap < 0): cells outside the matrix (columns < 1 of lanes that have not
// started, rows >= Q of the last strip) need no masking there, because with sentinel symbols that never match
// they evaluate to 0 from zero inputs, and phantom rows below the matrix only see values strictly smaller than
// a real cell's, so they never win the maximum.  MASK = true handles the right edge and odd scorings.
// WIN = true: the pair is a window with a known boundary column on its left (P.left); compiled separately so that
// the whole-matrix path keeps exactly its registers and instructions.
// All H values in here (h, top, boundary, leftv, bestv) are times four, tag bits clear.
template <int RPL, bool DIRS, bool MASK, bool WIN>
__device__ __forceinline__ void exact_steps16(const uint32_t (&qs)[RPL], const bool (&valid)[RPL], int (&h)[RPL],
                                              int (&bestv)[RPL], uint32_t (&bestj)[RPL], uint32_t (&acc)[RPL],
                                              int& top_prev, int& last_out, int& tcur, int& pub, int& jv,
                                              int bcur, int tblk, const CellConsts& C, uint32_t lim0,
                                              uint32_t Tlim, const int (&leftv)[RPL]) {
#pragma unroll
  for (int k4 = 0; k4 < 16; k4++) {
    // lane 0 takes element k4 of the block's boundary values / target symbols (held by lanes 0..15): a row rotate
    // puts it there, the wave shift then overwrites every lane but lane 0
    const int s_b = row_elem_to_lane0(bcur, k4);
    const int s_t = row_elem_to_lane0(tblk, k4);
    const int top = dpp_wave_shr1(s_b, last_out);  // H(row above this lane, column j)
    tcur = dpp_wave_shr1(s_t, tcur);               // target symbol of column j
    // everything that does not depend on this step's cell above: diag + s, left + gap
    int d[RPL], lfg[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) {
      const int sc = ((int)qs[r] == tcur) ? C.match4 : C.miss4;
      d[r] = (r == 0 ? top_prev : h[r - 1]) + sc;
      lfg[r] = MASK ? h[r] + C.left4 : sat_sub(h[r], C.lsub);
    }
    // the dependent chain down the lane's rows
    int upg = MASK ? top + C.up4 : sat_sub(top, C.usub);
#pragma unroll
    for (int r = 0; r < RPL; r++) {
      int ht = max(max(d[r], lfg[r]), upg);  // tagged
      if (MASK) ht = max(ht, 0);
      int hn = ht & ~3;
      int hb = hn;  // what competes for the best cell
      if (MASK) {
        const uint32_t lim = (r == 0) ? lim0 : Tlim;
        const bool act = valid[r] && jv >= 1 && (uint32_t)jv <= lim;
        // column 0 of a window holds the known boundary column H(x, col0) (zero for a whole matrix): every lane
        // passes through jv == 0 one step before its first real column, which is what puts the boundary into
        // h[] (the next step's `left`), into the row above's diagonal and into the lane below's top_prev
        if (WIN) {
          hn = act ? hn : (jv == 0 ? leftv[r] : 0);
          hb = act ? hn : 0;
        } else {
          hn = act ? hn : 0;
          hb = hn;
        }
      }
      // one v_alignbit: the word moves down two bits and the cell's code enters at the top, so after the 16 steps of
      // a block step k sits in bits [2 * (k % 16), +2)  (a masked cell's code is never read)
      if (DIRS) acc[r] = __builtin_amdgcn_alignbit((uint32_t)ht, acc[r], 2);
      if (hb > bestv[r]) bestj[r] = (uint32_t)jv;  // strict: the first (smallest j) maximal cell of the row
      bestv[r] = max(bestv[r], hb);
      h[r] = hn;
      upg = MASK ? hn + C.up4 : sat_sub(hn, C.usub);
    }
    top_prev = top;
    last_out = h[RPL - 1];
    // The bottom row of the strip leaves through lane 63: column k - 62 at step k.  `pub` is a shift register ACROSS
    // the lanes: every step it moves one lane down (lane i takes lane i + 1's word) and lane 63, which has no source,
    // keeps its own new last_out -- after the 16 steps of a block lanes 48..63 hold the block's 16 values in column
    // order, and ONE store instruction of 64 contiguous bytes publishes them (four 16-byte stores from lane 63 cost a
    // chained strip 15 ns per step: it is the number of store instructions that counts, profiles/r03_exact_step.txt).
    pub = __builtin_amdgcn_update_dpp(last_out, pub, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    jv++;
  }
}

template <int RPL, bool DIRS, bool WIN>
__device__ __forceinline__ void exact_strip(const ExactPair& P, const ExactWork& W, uint32_t strip, int match,
                                            int miss, int gap, int& gbv, uint64_t& gkey) {
  const int lane = threadIdx.x & 63;
  const uint32_t Q = P.Q, T = P.T;
  const uint32_t maxcol = P.Tlim > P.T0 ? P.Tlim : P.T0;
  const uint32_t mincol = P.Tlim < P.T0 ? P.Tlim : P.T0;
  const uint32_t NG = exact_nsteps(maxcol, Q, RPL) / 16;  // 16-step blocks (== P.nkb)
  constexpr uint32_t SH = 64 * RPL;
  const uint32_t nstrips = exact_nstrips(Q, RPL);

  // (column c of a row sits at word c + kBndShift: the 16 columns 16 g - 62 .. 16 g - 47 a block publishes then start on
  // a 64-byte line -- rows are whole lines, exact_bnd_pitch -- and the one store that carries them is one line)
  constexpr int kBndShift = (int)kExactBndShift;
  const gptr<const int32_t> bin = glb((const int32_t*)P.bnd) + (size_t)(strip > 0 ? strip - 1 : 0) * P.bnd_pitch + kBndShift;  // bottom row of strip-1
  const gptr<int32_t> bout = glb(P.bnd) + (size_t)strip * P.bnd_pitch + kBndShift;
  const auto gq = glb(P.q);
  const auto gt = glb(P.t);
  const auto gleft = glb(P.left);
  const bool chained = strip > 0;
  const bool feeds = strip + 1 < nstrips;
  const bool plain = miss < 0 && gap < 0;          // MASK = false is legal in the interior
  const CellConsts C = cell_consts(match, miss, gap);
  const uint32_t ibase = strip * SH + lane * RPL;  // 0-based first row of this lane
  uint32_t qs[RPL];
  bool valid[RPL];
  int h[RPL], bestv[RPL], leftv[RPL];
  uint32_t bestj[RPL], acc[RPL];
#pragma unroll
  for (int r = 0; r < RPL; r++) {
    uint32_t i = ibase + r;
    valid[r] = i < Q;
    qs[r] = valid[r] ? gq[i] : 0x10000u;  // never equals a u16 symbol
    leftv[r] = (WIN && valid[r]) ? 4 * (int)((gleft[(size_t)i * 64] >> P.left_shift) & 0xffffu) : 0;  // (times four, like every H here)
    h[r] = leftv[r];  // lane 0 starts at column 1 and never passes through column 0: its `left` is the boundary now
    bestv[r] = 0;
    bestj[r] = 0;
    acc[r] = 0;
  }
  const bool row1_lane = (strip == 0) && (lane == 0);
  const uint32_t lim0 = row1_lane ? P.T0 : P.Tlim;  // column limit of this lane's row r == 0
  // ... and so is its first diagonal: H(row above lane 0's first row, col0).  (The other lanes get both through the
  // wavefront: every lane above them is forced to the boundary column when it passes column 0.)
  int top_prev = (WIN && lane == 0 && ibase > 0) ? 4 * (int)((gleft[(size_t)(ibase - 1) * 64] >> P.left_shift) & 0xffffu) : 0;
  int last_out = h[RPL - 1];  // lane 1's first diagonal is lane 0's last row at column 0: the boundary again
  int pub = 0;
  int tcur = 0x20000;
  int jv = 1 - lane;  // the column this lane works on

  // lane l (mod 16) fetches what lane 0 needs at step 16 g + l: the boundary and the symbol of column 16 g + l + 1
  auto fetch_bnd = [&](uint32_t g) -> int {
    const uint32_t col = g * 16 + (lane & 15) + 1;
    if (!chained || col > maxcol) return (int)kTag;
    return __hip_atomic_load(bin + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto fetch_t = [&](uint32_t g) -> int {
    const uint32_t col = g * 16 + (lane & 15) + 1;
    return (col <= T) ? (int)gt[col - 1] : 0x20000;
  };
  int bnext = fetch_bnd(0), tnext = fetch_t(0);
  const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + W.wait_ticks;
  // (test hook, ExactWork::debug_drop: the first strip of the first pair never publishes its bottom row)
  const bool publish = feeds && !(W.debug_drop && strip == 0 && P.first_unit == 0);
  for (uint32_t g = 0; g < NG; g++) {
    // The boundary of this block must be complete; normally it is (fetched 16 steps ago).  The wait is bounded in
    // TIME (s_memrealtime, 100 MHz), not in iterations; when it runs out -- a bug, not a legal state: the strip above
    // holds an earlier ticket -- the wave stores the error word and ENDS on the spot (it never computes on with an
    // incomplete boundary), and every other waiting wave sees the word and ends too: the grid drains, the host
    // reads the word with the results (exact_chunk_end) and reports SW_E_HIP.  Both conditions are wave-uniform
    // (scalar branch: s_endpgm must not sit behind an EXEC mask).
    for (uint32_t spin = 0; __builtin_amdgcn_ballot_w64(((uint32_t)bnext & kTag) == 0) != 0; spin++) {
      if ((spin & 63u) == 63u) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(W.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const bool late = __builtin_amdgcn_s_memrealtime() >= deadline;
        if (e != 0 || late) {
          if (e == 0) __hip_atomic_store(W.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_endpgm" ::: "memory");
        }
      }
      __builtin_amdgcn_s_sleep(SW_EXACT_SLEEP);
      bnext = fetch_bnd(g);
    }
    const int bcur = (int)((uint32_t)bnext & ~kTag), tblk = tnext;
    if (g + 1 < NG) {
      bnext = fetch_bnd(g + 1);
      tnext = fetch_t(g + 1);
    }
    // (a window with a boundary column runs masked while any lane is still at or left of column 0: lane 63
    // reaches column 1 at step 63)
    if (plain && g * 16 + 16 <= mincol && !(WIN && g < 4))
      exact_steps16<RPL, DIRS, false, WIN>(qs, valid, h, bestv, bestj, acc, top_prev, last_out, tcur, pub, jv, bcur, tblk,
                                 C, lim0, P.Tlim, leftv);
    else
      exact_steps16<RPL, DIRS, true, WIN>(qs, valid, h, bestv, bestj, acc, top_prev, last_out, tcur, pub, jv, bcur, tblk,
                                C, lim0, P.Tlim, leftv);
    if (DIRS) {
      const gptr<uint32_t> w = glb(P.dirs) + ((size_t)(strip * P.nkb + g) * RPL) * 64 + lane;
#pragma unroll
      for (int r = 0; r < RPL; r++) w[r * 64] = acc[r];
    }
    if (publish && lane >= 48) {
      // lane 63 finished columns 16 g - 62 .. 16 g - 47 of the strip's bottom row during this block; lane 48 + i holds
      // column 16 g - 62 + i (see exact_steps16).  One device-scope store, each word with its "published" tag.
      const int col = (int)(g * 16) - 62 + (lane - 48);
      if (col >= 1 && (uint32_t)col <= maxcol)
        __hip_atomic_store(bout + col, (int)((uint32_t)pub | kTag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int r = 0; r < RPL; r++) {
    const uint64_t key = order_key(ibase + r + 1, bestj[r] + P.col0, P.Tfull ? P.Tfull : T, P.phaseQp, P.straight);
    const int bv = bestv[r] >> 2;
    if (bv > gbv || (bv == gbv && bv > 0 && key < gkey)) {
      gbv = bv;
      gkey = key;
    }
  }
}

// (score, key) with the reference's tie order: larger score, then smaller key
__device__ __forceinline__ void wave_best(int& gbv, uint64_t& gkey) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int ov = __shfl_xor(gbv, off);
    const uint32_t klo = __shfl_xor((uint32_t)gkey, off), khi = __shfl_xor((uint32_t)(gkey >> 32), off);
    const uint64_t okey = ((uint64_t)khi << 32) | klo;
    if (ov > gbv || (ov == gbv && okey < gkey)) {
      gbv = ov;
      gkey = okey;
    }
  }
}

// (211 VGPRs at 4 rows per lane = two waves per SIMD.  Forcing three or four -- amdgpu_waves_per_eu -- spills 51 / 129
// registers and changes a batch of 1 kb pairs by -1 / -2 % and a batch of reads by +3 / +6 %: left alone.)
template <int RPL>
__global__ void __launch_bounds__(64) sw_exact_kernel(ExactWork W, int match, int miss, int gap) {
  const int lane = threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(W.ticket, 1u);
  const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (ticket >= W.n_units) return;
  const uint32_t rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.unit_order[ticket]);
  const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.unit_pair[rec]);
  const ExactPair P = W.pairs[p];
  const uint32_t strip = rec - P.first_unit;
  const uint32_t maxcol = P.Tlim > P.T0 ? P.Tlim : P.T0;
  int gbv = 0;
  uint64_t gkey = ~0ull;
  if (strip < exact_nstrips(P.Q, RPL) && maxcol > 0) {
    if (P.left) {  // a window entered through a stored boundary column (always with direction bits)
      if (P.dirs) exact_strip<RPL, true, true>(P, W, strip, match, miss, gap, gbv, gkey);
      else exact_strip<RPL, false, true>(P, W, strip, match, miss, gap, gbv, gkey);
    } else if (P.dirs) {
      exact_strip<RPL, true, false>(P, W, strip, match, miss, gap, gbv, gkey);
    } else {
      exact_strip<RPL, false, false>(P, W, strip, match, miss, gap, gbv, gkey);
    }
  }
  wave_best(gbv, gkey);
  // The last strip of a pair to finish combines the per-strip records (threadfence reduction).
  if (lane == 0) {
    __hip_atomic_store(&W.recs[rec].score, gbv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&W.recs[rec].key_lo, (uint32_t)gkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&W.recs[rec].key_hi, (uint32_t)(gkey >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t prev = 0;
  if (lane == 0) prev = __hip_atomic_fetch_add(W.pair_done + p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
  if (prev + 1 != P.n_units) return;
  gbv = 0;
  gkey = ~0ull;
  for (uint32_t u = lane; u < P.n_units; u += 64) {
    const ExactRec* r = W.recs + P.first_unit + u;
    const int ov = __hip_atomic_load(&r->score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t okey = ((uint64_t)__hip_atomic_load(&r->key_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) |
                          __hip_atomic_load(&r->key_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ov > gbv || (ov == gbv && okey < gkey)) {
      gbv = ov;
      gkey = okey;
    }
  }
  wave_best(gbv, gkey);
  if (lane == 0) {
    uint32_t bx = 0, bj = 0;
    if (gbv > 0) {
      bx = (uint32_t)(gkey & 0x7fffffffu);
      const uint32_t hi = (uint32_t)((gkey >> 31) & 0xffffffffu);
      bj = (P.straight ? hi : hi - bx) - P.col0;  // window-local
    }
    ExactOut* out = W.outs + p;
    out->score = gbv;
    out->bx = bx;
    out->bj = bj;
    out->key_lo = (uint32_t)gkey;
    out->key_hi = (uint32_t)(gkey >> 32);
    out->end_x = out->end_j = 0;
    out->end_H = 0;
    out->nops = __hip_atomic_load(W.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0xffffffffu : 0;
  }
}

// Traceback walker: one WAVE per pair follows the direction bits from the best cell until H reaches 0
// (rust/src/algorithm/mod.rs:663-664 `if score == 0 return`) or an edge (:650).  ops[] receives, last
// alignment column first: 0 = (q[x-1], t[j-1]), 1 = (q[x-1], '-'), 2 = ('-', t[j-1]).
// The walk itself is serial, so what matters is the latency of one step: the wave stages a window of the
// direction words (8 blocks of 16 steps x 256 rows of the current strip = 8 KB, coalesced), the strip's query
// symbols and the matching target range in LDS and walks inside it with wave-uniform (scalar) state; within a
// strip the step index k = j - 1 + lane(row) never increases along the walk, so a window is left only downwards
// or when the path crosses into the strip above.
#ifndef SW_WALK_WKB
#define SW_WALK_WKB 16
#endif
constexpr uint32_t WKB = SW_WALK_WKB;  // 16-step blocks per window
template <int RPL>
__global__ void __launch_bounds__(64) sw_walk_kernel(const ExactPair* pairs, ExactOut* outs, uint32_t n,
                                                     int match, int miss, int gap) {
  constexpr int SH = 64 * RPL;
  constexpr int LOG_RPL = RPL == 4 ? 2 : (RPL == 2 ? 1 : 0);
  __shared__ uint32_t wdir[WKB * SH];  // [block][row of the strip]
  __shared__ uint32_t wq[SH];
  __shared__ uint32_t wt[WKB * 16 + 64];
  const uint32_t p = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  if (p >= n) return;
  const ExactPair P = pairs[p];
  if (!P.dirs || !P.ops) return;
  const auto gq = glb(P.q);
  const auto gt = glb(P.t);
  const auto gops = glb(P.ops);
  if (outs[p].nops == 0xffffffffu) return;  // the fill kernel reported a hand-over time-out
  const bool resumed = P.walk_x != 0;  // continue a traceback that left the window to the right through its column 0
  int H = __builtin_amdgcn_readfirstlane(resumed ? P.walk_H : outs[p].score);
  const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(resumed ? P.walk_x : outs[p].bx));
  const uint32_t bj = (uint32_t)__builtin_amdgcn_readfirstlane((int)(resumed ? P.walk_j : outs[p].bj));
  const uint32_t cap = P.Q + P.T;
  uint32_t nops = 0;
  uint32_t end_x = bx, end_j = bj;
  int end_H = H;
  if (H > 0 && bx > 0 && bj > 0) {
    // wave-uniform walk state: strip, row inside the strip, target index j - 1
    uint32_t strip = (bx - 1) / SH;
    int within = (int)((bx - 1) % SH);
    int jm1 = (int)bj - 1;
    bool go = true;
    bool tile_mode = false;    // how the path is followed inside a window: diagonal runs (below) or 7 x 7 tiles
    const int pos_match = match > 0 ? match : 0, pos_miss = miss > 0 ? miss : 0, pos_gap = gap > 0 ? gap : 0;
    uint32_t diag_tiles = 0;
#ifdef SW_WALK_COUNT
    uint32_t c_stage = 0, c_tile = 0, c_run = 0, c_tmoves = 0, c_rmoves = 0;
#define WCOUNT(x) x
#else
#define WCOUNT(x)
#endif
    while (go) {
      WCOUNT(c_stage++);
      // ---- stage a window: blocks [wbase, wtop] of this strip; all loads in flight before the first LDS write
      const uint32_t ktop = (uint32_t)jm1 + ((uint32_t)within >> LOG_RPL);
      const uint32_t wtop = ktop >> 4;
      const uint32_t wbase = wtop >= WKB - 1 ? wtop - (WKB - 1) : 0;
      const uint32_t nb = wtop - wbase + 1;
      const int tbase = (int)(wbase * 16) - 63;  // wt[i] = t[tbase + i]
      __syncthreads();  // (one wave: orders the LDS reads of the old window before the overwrite)
      {
        const gptr<const uint32_t> src = glb((const uint32_t*)P.dirs) + ((size_t)(strip * P.nkb + wbase) * RPL) * 64 + lane;
        uint32_t v[WKB * RPL];
#pragma unroll
        for (uint32_t u = 0; u < WKB * RPL; u++) {  // blocks past the window top are clamped (never read back)
          const uint32_t blk = u / RPL < nb ? u / RPL : nb - 1;
          v[u] = src[((size_t)blk * RPL + u % RPL) * 64];
        }
        uint32_t qv[SH / 64], tv[(WKB * 16 + 64) / 64];
#pragma unroll
        for (uint32_t u = 0; u < SH / 64; u++) {
          const uint32_t row = strip * SH + u * 64 + lane;
          qv[u] = row < P.Q ? gq[row] : 0x10000u;
        }
#pragma unroll
        for (uint32_t u = 0; u < (WKB * 16 + 64) / 64; u++) {
          const int tj = tbase + (int)(u * 64 + lane);
          tv[u] = (tj >= 0 && (uint32_t)tj < P.T) ? gt[tj] : 0x20000u;
        }
#pragma unroll
        for (uint32_t u = 0; u < WKB * RPL; u++)  // word u = (block u / RPL, row lane * RPL + u % RPL)
          wdir[(u / RPL) * SH + lane * RPL + (u % RPL)] = v[u];
#pragma unroll
        for (uint32_t u = 0; u < SH / 64; u++) wq[u * 64 + lane] = qv[u];
#pragma unroll
        for (uint32_t u = 0; u < (WKB * 16 + 64) / 64; u++) wt[u * 64 + lane] = tv[u];
      }
      __syncthreads();
      // ---- walk inside the window: one diagonal RUN plus the gap move that ends it per iteration.  Lane t
      // looks at the cell t diagonal moves further on; a ballot gives the length of the run, its H bookkeeping
      // is two popcounts (v_mbcnt) because a diagonal step subtracts either match or miss, and the first
      // non-diagonal cell's code is already in its lane.  Alignments of related sequences are long diagonal
      // runs: tens of steps per LDS round trip instead of one.
      const int wlo = (int)(wbase * 16);  // smallest step index covered
      for (;;) {
        const int k0 = jm1 + (within >> LOG_RPL);
        if (k0 < wlo) break;  // window exhausted (same strip)
        if (tile_mode) {
          // ---- a TILE of cells up-left of the current one, a band of 7 rows along the diagonal: lane (dr, e) =
          // (lane / 8, lane % 8) holds the code of cell (row - dr, col - dc) with dc = dr + e - 3, i.e. e - 3 is how
          // far the path has drifted off the diagonal (more lefts than ups); the start is lane 3.  Row 7 and the
          // columns e = 0 and e = 7 of the lane grid are sentinels (never valid).  The path through the tile is
          // followed on the scalar unit over the ballots of the codes -- a move adds 8 (diagonal), 7 (up) or 1 (left)
          // to the lane index, so path order is lane order -- and the H bookkeeping of all its moves is done at once
          // afterwards.  A path leaves the band after 7 row moves or when it drifts 3 columns off the diagonal:
          // alignments of unrelated sequences (a gap every third move) cross it in 7.4 moves per LDS round trip
          // (a 7 x 7 square: 7.1; strip and window edges cut some tiles short), the diagonal-run iteration below ~3.
          const int dr = (int)(lane >> 3), e = (int)(lane & 7u);
          const int dc = dr + e - 3;
          const int wr = within - dr, jt = jm1 - dc;
          const int kt = jt + (wr >> LOG_RPL);
          const bool valid = dr < 7 && e >= 1 && e < 7 && dc >= 0 && wr >= 0 && jt >= 0 && kt >= wlo;
          // (defaults of an invalid lane: code 0 and two symbols that differ, so the three masks below need no
          // `valid &&` -- a bool carried out of the branch costs a v_cndmask + v_cmp pair per ballot)
          uint32_t code = 0, qsym = 0x30000u, tsym = 0x40000u;
          if (valid) {
            const uint32_t w = wdir[(((uint32_t)kt >> 4) - wbase) * SH + (uint32_t)wr];
            code = (w >> (((uint32_t)kt & 15u) * 2)) & 3u;
            qsym = wq[wr];
            tsym = wt[jt - tbase];
          }
          const uint64_t V = __builtin_amdgcn_ballot_w64(valid);
          const uint64_t Dm = __builtin_amdgcn_ballot_w64(code == 3u);
          const uint64_t Um = __builtin_amdgcn_ballot_w64(code == 2u);
          const uint64_t EQ = __builtin_amdgcn_ballot_w64(qsym == tsym);
          // The path: a move adds 7 to the lane index if the cell's code is diagonal or up, and 1 if it is diagonal
          // or left.  At most 16 cells of the band lie on one path (6 row moves and at most 9 lefts keep it inside),
          // so the loop is unrolled with not-taken exits: 7 scalar instructions per move, no taken branch before the
          // end.
          const uint64_t M7 = Dm | Um, M1 = Dm | ~Um;
          uint64_t PM;
          uint32_t pidx, ptmp;
#define SW_WALK_MOVE                    \
  "s_cbranch_scc0 .Ltile_end_%=\n\t"    \
  "s_bitset1_b64 %[pm], %[idx]\n\t"     \
  "s_bitcmp1_b64 %[m7], %[idx]\n\t"     \
  "s_cselect_b32 %[tmp], 7, 0\n\t"      \
  "s_bitcmp1_b64 %[m1], %[idx]\n\t"     \
  "s_addc_u32 %[idx], %[idx], %[tmp]\n\t" \
  "s_bitcmp1_b64 %[v], %[idx]\n\t"
          asm volatile(
              "s_mov_b64 %[pm], 0\n\t"
              "s_mov_b32 %[idx], 3\n\t"
              "s_bitcmp1_b64 %[v], 3\n\t" SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE
                  SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE SW_WALK_MOVE
                      SW_WALK_MOVE SW_WALK_MOVE
              ".Ltile_end_%=:"
              : [pm] "=&s"(PM), [idx] "=&s"(pidx), [tmp] "=&s"(ptmp)
              : [v] "s"(V), [m7] "s"(M7), [m1] "s"(M1)
              : "scc");
#undef SW_WALK_MOVE
          const bool on_path = __builtin_amdgcn_inverse_ballot_w64(PM);
          const bool isd = __builtin_amdgcn_inverse_ballot_w64(Dm);
          const bool isu = !isd && __builtin_amdgcn_inverse_ballot_w64(Um);
          const uint64_t PD = PM & Dm, PE = PD & EQ, PG = PM & ~Dm;
          auto below = [&](uint64_t m) -> int {  // bits of m in lanes < mine
            return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          };
          const int cE = __builtin_popcountll(PE), cD = __builtin_popcountll(PD), cG = __builtin_popcountll(PG);
          uint32_t n = (uint32_t)(cD + cG);  // cells on the path
          uint64_t T = PM;                   // the moves taken
          // Walking back, only a move with a positive score lowers H.  If H stays positive even after all of them
          // the walk cannot end inside this tile and everything but the op bytes is scalar: counts instead of prefix
          // counts, and the position after the last move is the lane index the path loop stopped at.
          if (H - pos_match * cE - pos_miss * (cD - cE) - pos_gap * cG > 0 && nops + n <= cap) {
            H -= match * cE + miss * (cD - cE) + gap * cG;
            if (on_path) gops[nops + (uint32_t)below(PM)] = (uint8_t)(isd ? 0u : (isu ? 1u : 2u));
            within -= (int)(pidx >> 3);
            jm1 -= (int)(pidx >> 3) + (int)(pidx & 7u) - 3;
          } else {
            // H after this lane's move (meaningful in path lanes): the moves before it by prefix counts, its own by
            // type (scores are 16-bit and counts <= 16: 24-bit multiplies)
            const int own = isd ? (qsym == tsym ? match : miss) : gap;
            const int h_after = H - own - __mul24(miss, below(PD)) - __mul24(match - miss, below(PE)) - __mul24(gap, below(PG));
            const uint64_t stop = __builtin_amdgcn_ballot_w64(on_path && h_after <= 0);
            if (stop) T = PM & ((2ull << __builtin_ctzll(stop)) - 1ull);
            const uint32_t lt = 63u - (uint32_t)__builtin_clzll(T);  // lane of the last one (T != 0)
            n = (uint32_t)__builtin_popcountll(T);
            const uint32_t rank = (uint32_t)below(T);
            if (nops + n > cap) n = cap - nops;
            if (__builtin_amdgcn_inverse_ballot_w64(T) && rank < n) gops[nops + rank] = (uint8_t)(isd ? 0u : (isu ? 1u : 2u));
            const int moved = (dr + ((isd || isu) ? 1 : 0)) | ((dc + ((isd || !isu) ? 1 : 0)) << 8);
            const int mv = __builtin_amdgcn_readlane(moved, (int)lt);
            H = __builtin_amdgcn_readlane(h_after, (int)lt);
            within -= mv & 0xff;
            jm1 -= mv >> 8;
          }
          nops += n;
          WCOUNT(c_tile++; c_tmoves += n);
          // back to diagonal runs once two tiles in a row were crossed corner to corner
          diag_tiles = ((T & ~Dm) == 0 && n >= 7) ? diag_tiles + 1 : 0;
          if (diag_tiles >= 2) tile_mode = false;
          if (H <= 0 || jm1 < 0 || nops >= cap) {
            go = false;
            break;
          }
          if (within < 0) {  // into the strip above (or off the top edge)
            if (strip == 0) {
              go = false;
              within = -1;  // row 0
              break;
            }
            strip--;
            within += SH;
            break;
          }
          continue;
        }
        const int wr = within - (int)lane, jt = jm1 - (int)lane;
        const int kt = jt + (wr >> LOG_RPL);
        const bool valid = wr >= 0 && jt >= 0 && kt >= wlo;
        uint32_t code = 0;
        bool eq = false;
        if (valid) {
          const uint32_t w = wdir[(((uint32_t)kt >> 4) - wbase) * SH + (uint32_t)wr];
          code = (w >> (((uint32_t)kt & 15u) * 2)) & 3u;
          eq = wq[wr] == wt[jt - tbase];
        }
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        const uint64_t run = __builtin_amdgcn_ballot_w64(valid && code == 3u);
        const uint32_t L = ~run ? (uint32_t)__builtin_ctzll(~run) : 64u;  // diagonal moves from here
        const uint64_t eqm = __builtin_amdgcn_ballot_w64(eq);
        const int neq = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(eqm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)eqm, 0u));
        const int h_before = H - __mul24(match, neq) - __mul24(miss, (int)lane - neq);  // H at lane t's cell if all before were diagonal
        const int h_after = h_before - (eq ? match : miss);
        const uint64_t stop = __builtin_amdgcn_ballot_w64(lane < L && h_after <= 0);  // the walk ends where H reaches 0
        // branch-free bookkeeping (every taken branch costs a lone wave several issue slots)
        const uint32_t nd = stop ? min(L, (uint32_t)__builtin_ctzll(stop) + 1u) : L;  // diagonal moves taken
        const bool take_gap = !stop && L < 64 && ((vmask >> (L & 63u)) & 1ull);  // the cell that ended the run is in reach
        const int h_run = __builtin_amdgcn_readlane(h_after, (int)(nd ? nd - 1 : 0));
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)code, (int)(L & 63u));
        const bool up = take_gap && c == 2u;
        const bool left = take_gap && c != 2u;
        H = (nd ? h_run : H) - (take_gap ? gap : 0);
        within -= (int)nd + (up ? 1 : 0);
        jm1 -= (int)nd + (left ? 1 : 0);
        uint32_t n = nd + (take_gap ? 1u : 0u);
        const uint32_t op = (take_gap && lane == n - 1) ? (up ? 1u : 2u) : 0u;
        if (nops + n > cap) n = cap - nops;
        if (lane < n) gops[nops + lane] = (uint8_t)op;
        nops += n;
        WCOUNT(c_run++; c_rmoves += n);
        if (nd < 4 && take_gap) {  // direction changes every few cells: follow the path tile by tile
          tile_mode = true;
          diag_tiles = 0;
        }
        if (H <= 0 || jm1 < 0 || nops >= cap) {
          go = false;
          break;
        }
        if (within < 0) {  // into the strip above (or off the top edge)
          if (strip == 0) {
            go = false;
            within = -1;  // row 0
            break;
          }
          strip--;
          within = SH - 1;
          break;
        }
      }
    }
#ifdef SW_WALK_COUNT
    if (lane == 0 && nops > 1000) printf("walk: %u stages, %u tiles (%u moves), %u runs (%u moves)\n", c_stage, c_tile, c_tmoves, c_run, c_rmoves);
#endif
    // where the walk stopped (row 1-based, window-local column; column 0 = the window's boundary column)
    end_x = (uint32_t)((int)(strip * SH) + within + 1);
    end_j = (uint32_t)(jm1 + 1);
    end_H = H;
  }
  if (lane == 0) {
    outs[p].nops = nops;
    outs[p].end_x = end_x;
    outs[p].end_j = end_j;
    outs[p].end_H = end_H;
  }
}

}  // namespace

hipError_t launch_exact(const ExactWork& w, Scoring sc, hipStream_t stream) {
  if (w.n_units == 0) return hipSuccess;
  if (w.rpl == 4)
    hipLaunchKernelGGL(sw_exact_kernel<4>, dim3(w.n_units), dim3(64), 0, stream, w, sc.match, sc.miss, sc.gap);
  else if (w.rpl == 2)
    hipLaunchKernelGGL(sw_exact_kernel<2>, dim3(w.n_units), dim3(64), 0, stream, w, sc.match, sc.miss, sc.gap);
  else if (w.rpl == 1)
    hipLaunchKernelGGL(sw_exact_kernel<1>, dim3(w.n_units), dim3(64), 0, stream, w, sc.match, sc.miss, sc.gap);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_walk(const ExactPair* d_pairs, ExactOut* d_outs, uint32_t n_pairs, int rpl, Scoring sc,
                       hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  if (rpl == 4)
    hipLaunchKernelGGL(sw_walk_kernel<4>, dim3(n_pairs), dim3(64), 0, stream, d_pairs, d_outs, n_pairs, sc.match,
                       sc.miss, sc.gap);
  else if (rpl == 2)
    hipLaunchKernelGGL(sw_walk_kernel<2>, dim3(n_pairs), dim3(64), 0, stream, d_pairs, d_outs, n_pairs, sc.match,
                       sc.miss, sc.gap);
  else if (rpl == 1)
    hipLaunchKernelGGL(sw_walk_kernel<1>, dim3(n_pairs), dim3(64), 0, stream, d_pairs, d_outs, n_pairs, sc.match,
                       sc.miss, sc.gap);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace swhip
