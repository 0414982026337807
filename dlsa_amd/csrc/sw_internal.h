// dlsa_amd/csrc/sw_internal.h -- shared between the translation units of libsw_hip.so.
// Not installed; the public contract is include/sw.h + include/sw_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <exception>
#include <new>
#include <string>
#include <vector>

namespace swhip {

// records the calling thread's last error message (sw_hip_last_error)
void set_last_error(const std::string& msg);

// No C++ exception may cross the C ABI: the reference turns every internal failure into NULL
// (rust/src/bindings.rs:109-112, 153-156) and a cgo caller cannot unwind.  Every extern "C" entry point that
// can allocate runs its body through this; `on_error` is what the caller sees, the message goes to
// sw_hip_last_error().
template <class R, class F>
R guarded(R on_error, F&& body) noexcept {
  const char* what = "internal error";
  try {
    return body();
  } catch (const std::bad_alloc&) {
    what = "out of host memory";
  } catch (const std::exception&) {
    what = "internal error (C++ exception)";
  } catch (...) {
  }
  try {
    set_last_error(what);
  } catch (...) {
  }
  return on_error;
}

// ---- scoring ------------------------------------------------------------------------------------------
struct Scoring {
  int match, miss, gap;  // as the reference sees them: match > 0, miss < 0, gap < 0 on the fast paths
};

// ---- exact (intra-task) kernel ------------------------------------------------------------------------
// Which reference entry point a pair reproduces decides four small integers (DESIGN.md "Effective
// semantics"); the kernel itself is one plain linear-gap SW over the cells
//   { (x, j) : 1 <= x <= Q, 1 <= j <= (x == 1 ? T0 : Tlim) }.
struct ExactPair {
  const uint16_t* q;  // query symbols (Unicode scalar truncated to u16, as rust/src/algorithm/mod.rs:78,224)
  const uint16_t* t;  // target symbols
  uint32_t Q, T;      // real lengths
  uint32_t Tlim;      // last column computed for rows x >= 2
  uint32_t T0;        // last column computed for row x == 1
  uint32_t phaseQp;   // != 0: ring path with T < Qp, two-phase discovery order, value = Qp
  uint32_t straight;  // != 0: tie order of find_alignment_sequential_straight (smallest j, then x)
  int32_t* bnd;       // scratch: nstrips rows of bnd_pitch ints (row s = bottom row of strip s), zeroed:
                      // bit 31 of a word = published (sw_exact.hip)
  uint32_t bnd_pitch;
  uint32_t* dirs;     // nullptr: score + best cell only.  Else 2 bits per cell (layout in sw_exact.hip)
  uint32_t nkb;       // 16-step blocks per strip
  uint8_t* ops;       // traceback ops written by the walker (capacity Q + T)
  uint32_t first_unit; // record index of this pair's strip 0 (records are in (pair, strip) order; tickets: ExactWork::unit_order)
  uint32_t n_units;    // max(number of strips, 1)
  // ---- a WINDOW of the matrix (the work-package fast path, sw_lib.cpp: align_windows): the pair above is rows
  // 1..Q x columns col0+1 .. col0+T of a larger matrix whose column col0 is known.  t points at the symbol of
  // column col0+1; T, Tlim, T0 are local (already reduced by col0).
  const uint32_t* left;  // nullptr: H(x, col0) = 0 (col0 == 0).  Else the boundary column as the score kernel left
                         // it in its ring: row x (1-based) is the int16 half `left_shift` of left[(x - 1) * 64]
  uint32_t left_shift;   // 0 or 16
  uint32_t col0;         // columns left of the window (added to j in the discovery-order key)
  uint32_t Tfull;        // length of the whole target (the key's phase rule compares j with it); 0: T
  // the walker starts at (walk_x, walk_j, walk_H) instead of the pair's best cell when walk_x != 0 (a traceback
  // that continues from the window to its right)
  uint32_t walk_x, walk_j;
  int32_t walk_H;
  uint32_t pad_;
};
struct ExactOut {
  int32_t score;  // H of the best cell (0: no hit)
  uint32_t bx;    // best cell, 1-based query index (0: none)
  uint32_t bj;    // best cell, 1-based target index (window-local)
  uint32_t nops;  // number of traceback ops (walker)
  // where and with which H the walker stopped: end_H > 0 with end_j == 0 means it reached the window's left edge
  // (column col0) at row end_x and the traceback continues in the window to the left
  uint32_t end_x, end_j;
  int32_t end_H;
  uint32_t key_lo, key_hi;  // discovery-order key of the best cell (order_key, global j): lets the host choose
  uint32_t pad_;            // between windows of one pair
};
// Rows per lane (rpl) of the exact kernel: 4, 2 or 1, i.e. strips of 256, 128 or 64 query rows -- chosen per
// submission (exact_choose_rpl): 4 does the least work per cell (batches of many pairs), smaller strips give a
// single long pair more waves and a shorter step.

// steps per strip (multiple of 16) for a pair of Q rows whose widest row has `maxcol` columns: lane l reaches column
// maxcol at step maxcol + l - 1.  A pair of several strips needs all 64 lanes of every strip (the last lane hands the
// bottom row to the next strip); a pair that fits ONE strip stops with its last lane that holds a row -- a 150-row
// read on 4 rows per lane uses 38 lanes and 192 steps instead of 224.
__host__ __device__ inline uint32_t exact_nstrips(uint32_t Q, int rpl) { return (Q + 64 * rpl - 1) / (64 * rpl); }
__host__ __device__ inline uint32_t exact_nsteps(uint32_t maxcol, uint32_t Q, int rpl) {
  if (maxcol == 0) return 0;
  const uint32_t lanes = Q > 64u * (uint32_t)rpl ? 64u : (Q + (uint32_t)rpl - 1) / (uint32_t)rpl;
  return ((maxcol + (lanes ? lanes : 1u) - 1 + 15) / 16) * 16;
}
// Words per boundary row (ExactPair::bnd_pitch): whole 64-byte lines, with room for the kernel's 14-word shift of the
// columns (sw_exact.hip, kBndShift: the 16 columns a block publishes then start on a line).  Every builder of an
// ExactPair sizes and strides its rows with this; the region itself must start on a 64-byte boundary.
constexpr uint32_t kExactBndShift = 14;
inline uint32_t exact_bnd_pitch(uint32_t maxcol) { return (maxcol + 2 + kExactBndShift + 15) & ~15u; }
inline size_t exact_dir_words(uint32_t Q, uint32_t maxcol, int rpl) {
  return (size_t)exact_nstrips(Q, rpl) * (exact_nsteps(maxcol, Q, rpl) / 16) * rpl * 64;
}

// best cell of one strip (the last strip of a pair to finish combines them)
struct ExactRec {
  int32_t score;
  uint32_t key_lo, key_hi;  // order_key of sw_exact.hip
  uint32_t pad_;
};
// One launch = n_units waves, one 256-row strip each, spread over the whole GPU.
struct ExactWork {
  const ExactPair* pairs;
  ExactOut* outs;
  const uint32_t* unit_pair;  // per record (first_unit + strip): index of its pair
  const uint32_t* unit_order; // per ticket: the record it works on (strip level by level, sw_lib.cpp)
  uint32_t n_units;
  uint32_t n_pairs;
  int rpl;              // rows per lane of this launch: 4, 2 or 1
  ExactRec* recs;       // per ticket
  uint32_t* pair_done;  // per pair: strips finished (zeroed)
  uint32_t* ticket;     // work counter (zeroed)
  uint32_t* err;        // != 0: a hand-over wait timed out (zeroed); the waiting waves have ended, results are void
  unsigned long long wait_ticks;  // budget of a hand-over wait in s_memrealtime ticks (100 MHz)
  uint32_t debug_drop;  // test hook: strip 0 of pair 0 never publishes its bottom row (SW_HIP_DEBUG_DROP_HANDOVER)
};
hipError_t launch_exact(const ExactWork& w, Scoring sc, hipStream_t stream);
hipError_t launch_walk(const ExactPair* d_pairs, ExactOut* d_outs, uint32_t n_pairs, int rpl, Scoring sc,
                       hipStream_t stream);

// ---- exact kernel for big batches of short pairs: one pair per lane (sw_lanes.hip) ---------------------------
struct LanePair {       // lane l of group g: lanes[g * 64 + l]
  uint32_t qoff, toff;  // first symbol in the pool (u16 units, multiples of 8: the kernel loads 16 bytes at a time)
  uint32_t Q, T;        // rows; columns computed (Tlim of the path).  Q == 0: idle lane
  uint32_t pair;        // output slot
  uint32_t pad_;
  uint64_t ops_off;     // this pair's region in LanesWork::ops: Q + T bytes for the query alignment, then Q + T for
                        // the target alignment; the walker fills both from their ends (sw_lanes.hip)
};
struct LaneGroup {      // 64 pairs: one wave per 16-column strip
  uint32_t Qmax;        // rows a wave runs: max Q of the group's lanes, rounded up to 8
  uint32_t nstrips;     // 16-column strips: ceil(max T / 16)
  uint64_t dirs_off;    // dwords: direction words of the group, [strip][row][lane]
  uint64_t hb_off;      // dwords: boundary column between strips, [row][lane], rewritten in place by every strip;
                        // a word is H << 16 | (strip that wrote it + 1); zeroed before the launch
  uint32_t rec_off;     // first of the group's nstrips x 64 best-cell records (LanesWork::recs)
  uint32_t pad_;
};
struct LaneRec {        // best cell of one lane within one strip
  uint32_t score;
  uint32_t key;         // (x + j) << 16 | x, 1-based (0xffffffff: none)
};
struct LaneOut {
  int32_t score;        // H of the best cell (0: no hit)
  uint32_t bxbj;        // best cell, 1-based: bx | bj << 16 (0: none; Q + T < 65000 on this path)
  uint32_t nops;        // alignment columns found by the walker
  uint32_t off16;       // where the packer put the pair's two aligned strings in LanesWork::dense, in 16-byte units:
                        // the query alignment (nops bytes), then at the next multiple of 4 the target alignment
};
// bytes a pair of `nops` alignment columns takes in LanesWork::dense
__host__ __device__ inline uint32_t lanes_dense_bytes(uint32_t nops) { return (2u * ((nops + 3u) & ~3u) + 15u) & ~15u; }
struct LanesWork {
  const LaneGroup* groups;
  const LanePair* lanes;
  const uint16_t* pool;  // symbols; padded so that 32-byte loads at any sequence start + multiple of 16 stay inside
  uint32_t* dirs;
  uint32_t* hb;
  uint8_t* ops;          // staging of the walker (LanePair::ops_off); never leaves the device
  uint8_t* dense;        // the aligned strings of all pairs, packed (LaneOut::off16): what the host copies back
  uint32_t* dense_used;  // bytes of `dense` handed out so far (zeroed)
  LaneOut* outs;
  LaneRec* recs;
  const uint32_t* units;  // per ticket: strip << 20 | group, strip level by level (a strip's left neighbour first)
  uint32_t* ticket;       // work counter (zeroed)
  uint32_t* err;          // != 0: a hand-over wait timed out (zeroed); the waiting waves have ended, results are void
  unsigned long long wait_ticks;  // budget of a hand-over wait in s_memrealtime ticks (100 MHz)
  uint32_t n_groups;
  uint32_t n_units;       // sum of the groups' strips
#ifdef SW_LANES_TIMELINE  // (development build, scratch/lanes_timeline.sh: when which strip ran, and how long it waited)
  unsigned long long* timeline;  // per ticket: start, end, ticks spent waiting (s_memrealtime), hw id << 32 | unit
#endif
};
hipError_t launch_lanes(const LanesWork& w, Scoring sc, bool want_trace, hipStream_t stream);

// ---- inter-task score kernel ---------------------------------------------------------------------------
constexpr int kScoreR = 16;       // target columns per wave (one 2-bit packed dword per target)
#ifndef SW_SCORE_B
#define SW_SCORE_B 12
#endif
constexpr int kScoreB = SW_SCORE_B;       // query rows per hand-over block (one 2-bit packed dword of the query)
// ... of a pass with one query (score_block_rows; the ring is then twice as many blocks deep): finer hand-over, first
// for passes over small databases, which are a few chains of strips bound by the block's latency (one pair of the
// reference's synthetic default through the work-package fast path: 1.94 -> 1.61 ms), and since the waves set their
// issue priority by their lead also for the throughput shapes (sw_score.hip)
#ifndef SW_SCORE_B_SMALL
#define SW_SCORE_B_SMALL 6
#endif
constexpr int kScoreBSmall = SW_SCORE_B_SMALL;
constexpr int kGroupTargets = 128; // 64 lanes x 2 packed int16 halves

// The packed database is one stream of STRIPS: all strips of group 0, then group 1, ...; strip e occupies
// dwords [e*128, e*128+128) of `words` ([half][lane], 16 residues per dword).  A work unit is W consecutive
// strips of that stream (W = waves per workgroup), whatever groups they belong to.  The stream is cut into
// LANES (runs of whole groups that fill their last unit well); tickets visit S lanes round-robin, so the unit
// that continues a group is handed out S tickets after the unit it depends on -- with S beyond the number of
// resident workgroups (the normal case, sw_lib.cpp: score_pass_locked) after that one has finished.
struct ScoreGroup {       // one group of up to 128 length-sorted targets
  uint32_t strip_base;    // index of this group's first strip in the stream
  uint32_t nstrips;       // ceil(max length / 16)
  uint32_t first_slot;    // index of lane 0 / half 0 in the per-slot arrays (lens, scores); 128 slots
  uint32_t last_cols;     // columns of the last strip that any target of the group reaches: max length - 16 * (nstrips
                          // - 1), 1..16; <= 8: the strip runs the half-width row loop.  0 (unknown) = 16.
};

struct ScoreUnit {        // ticket t -> what to do (built per search on the host, see build_units)
  uint32_t first_elem;    // first strip of the stream
  uint32_t count;         // strips (1..W); wave w takes strip first_elem + w
  uint32_t dep;           // ticket of the unit that produces my wave 0's left column (~0u: none)
  uint32_t reuse;         // ticket whose wave 0 must have released the boundary-ring slot I write (~0u: none)
};

struct ScoreParams {
  const uint32_t* words;      // the strip stream
  const ScoreUnit* units;     // n_units, in ticket order
  const uint32_t* elem_group; // per strip of the stream: its group
  const ScoreGroup* groups;   // n_groups
  const uint32_t* lens;       // per slot: target length (0 for padding slots)
  const uint32_t* qwords;     // query, 2 bits per residue, 16 per dword (row-block aligned)
  uint32_t Q;
  uint32_t n_units;
  uint32_t n_elems;           // strips in the stream   } bounds the kernel checks the tables against
  uint32_t n_groups;          //                        } (an inconsistent table is an error code, not a fault)
  uint32_t n_slots;           // 128 * n_groups         }
  int32_t* slot_scores;       // per slot: best H over the effective domain (32767 = saturated); zeroed
  uint32_t* gbound;           // ring of inter-unit boundary columns: ring_slots * gbound_stride dwords
  uint64_t gbound_stride;     // dwords per ring slot (Q_pad * 64)
  uint32_t ring_slots;
  uint32_t* strip_best;       // nullptr, or per strip of the stream 64 dwords: the packed maxima of that strip alone
                              // (which 16 columns of which target reach its score: the work-package fast path)
  uint32_t* prog;             // 2 per unit: boundary blocks published / input slot released (zeroed)
  uint32_t* ticket;           // work counter (zeroed before launch)
  uint32_t* err;              // set != 0 if a dependency wait timed out (zeroed before launch)
  uint32_t pub;               // blocks per publication of an inter-unit boundary column
  int match, miss, gap;
  // A batch of queries in one pass (nullptr: one query of Q rows).  One word per row block of the stacked queries,
  // + one padding word: bits 0..3 rows of the block (1..12), kBlkFirst / kBlkLast = first / last block of its
  // query, bits kBlkQueryShift.. = index of the query.  qwords has one word per block accordingly; slot_scores holds
  // one array of n_slots per query; Q is the length of any one of them (they agree on which targets take the
  // full-matrix path, see db_topk_phase).
  const uint32_t* blkinfo;
  uint32_t n_blocks;
};
enum : uint32_t { kBlkFirst = 16u, kBlkLast = 32u, kBlkQueryShift = 8u };
// block_rows: kScoreB, or kScoreBSmall (4-wave workgroups, one query only); the query words and the ring geometry
// of the launch must have been made for it
hipError_t launch_score(const ScoreParams& p, int n_workgroups, int waves, int block_rows, hipStream_t stream);
int score_block_rows(int waves, bool multi, uint32_t n_groups);  // which of the two a pass of that shape runs with
int score_waves_per_group(uint32_t total_strips, int cus, uint32_t n_groups);  // waves per workgroup for a database of that many strips in that many lane groups
int score_workgroups_per_cu(int waves);
uint32_t score_queue_depth();  // units a workgroup may have in flight
// instruction counts of the generated row loop (sw_score_asm.inc): packed VALU per row, address increments per 2 rows
void score_loop_counts(uint32_t* packed_per_row, uint32_t* aux_per_2rows);


// ---- device-side top-k selection over the per-slot scores (sw_topk.hip) --------------------------------------
enum { kSelRerun = 0, kSelCand = 1, kSelTie = 2, kSelThresh = 3, kSelAbove = 4, kSelWords = 8 };
struct SelectParams {
  const int32_t* slot_scores;   // per slot, as the score kernel left them
  const uint32_t* lens;         // per slot: target length (0: padding)
  const uint32_t* slot_target;  // per slot: database index
  uint32_t n_slots;
  uint32_t Q;
  int match;
  uint32_t k;
  uint32_t* hist;      // 32768 bins (zeroed)
  uint32_t* counters;  // kSelWords words (zeroed)
  uint32_t* rerun;     // slots the exact kernel must re-score
  uint32_t rerun_cap;
  uint32_t* cand;      // (score, database index) pairs above the threshold score
  uint32_t cand_cap;
  uint32_t* tie;       // ... at the threshold score
  uint32_t tie_cap;
};
hipError_t launch_select(const SelectParams& p, hipStream_t stream);

}  // namespace swhip
