// dlsa_amd/csrc/sw_score.hip -- the inter-task (many targets, one query) score kernel for gfx950.
//
// What it replaces: for every (query, target) pair of a database search, the score-matrix fill and the
// maximum of rust/src/algorithm/mod.rs:66-202 / :204-445 (one cgo call per pair in the reference,
// internal/worker/worker_impl.go:108-168).  Only the best score per target leaves the kernel; end
// coordinates and alignments of the reported hits come from the exact kernel (sw_exact.hip).
//
// Design (MI355X-first, nothing like the reference's anti-diagonal CPU SIMD):
//  * one 32-bit lane = TWO targets (packed int16 halves), one wavefront = 128 length-sorted targets;
//  * a wave keeps R = 16 target COLUMNS of its 128 targets in registers and walks down the query, so the
//    query residue of a row is wave-uniform: the substitution scores of a row are pre-built register
//    tables, v(64 + 4*column + nucleotide), and the row's nucleotide selects its table through the VGPR
//    index mode (s_set_gpr_idx_idx on the third source of the add) -- no branch, no LDS profile lookup and
//    no VALU compare per cell;
//  * per cell 4.5 packed VALU instructions: v_pk_max_i16(left, up), v_pk_sub_u16 clamp (the unsigned
//    saturating subtract of |gap| is also the max(.,0)), v_pk_mad_i16 clamp (diag*1 + s, saturation = the
//    overflow detector; the mad form because only a third source can be indexed on its own),
//    v_pk_max_i16, and the running maximum on every second column only (max(left,up) of an odd column
//    covers the new even-column cell of this row and the odd-column cell of the previous row);
//  * the W waves of a workgroup own consecutive 16-column strips and run as a software pipeline over
//    blocks of B query rows; the H column between neighbouring strips travels through an LDS ring of D blocks per
//    wave with ready/done flags in LDS -- no workgroup barrier in the main loop, so a wave that the issue arbiter
//    favours runs ahead instead of parking the SIMD at a barrier.  B x D = 6 x 6 for one query, 12 x 3 for a stacked
//    batch of queries (36 rows x 17 rings is what 160 KB of LDS holds).  Before the waves set their own issue
//    priority (see the block loop) 12 x 3 was the faster one by 4 %; with it the finer hand-over wins: bench shape
//    67.3 -> 66.2 ms, 150-nt reads 16.5 -> 16.2 ms, 8 x 4 the same, 4 x 9 slower by 3-4 %;
//  * a work unit is up to W consecutive strips of the database's strip stream (all strips of group 0, then
//    group 1, ...) against the whole query.  Persistent workgroups (one per CU) pull units from an atomic
//    ticket; when a group continues in the next unit the two run concurrently on different CUs as a chain:
//    the producer streams its last H column into a ring slot in HBM (write-through stores, published every
//    ~128 rows through a progress counter) and the consumer trails it.  Dependencies always point to an
//    earlier ticket, whose workgroup is resident, so the chain cannot deadlock (sw_lib.cpp: build_units).
//
// Saturation: diag + s uses the clamp bit, so a lane whose true maximum exceeds 32766 reports 32767 and
// the host re-runs that pair with the int32 exact kernel.
#include <cstdlib>

#include "sw_internal.h"
#include "sw_score_asm.inc"

// back-off of the two spin loops (units of 64 clocks): global progress counters / LDS flags
#ifndef SW_SCORE_GSLEEP
#define SW_SCORE_GSLEEP 16
#endif
#ifndef SW_SCORE_LSLEEP
#define SW_SCORE_LSLEEP 16
#endif
// ... per wait: for the left neighbour's block (ready) / for room in the right neighbour's ring (done)
#ifndef SW_SCORE_RSLEEP
#define SW_SCORE_RSLEEP SW_SCORE_LSLEEP
#endif
#ifndef SW_SCORE_DSLEEP
#define SW_SCORE_DSLEEP SW_SCORE_LSLEEP
#endif
#ifndef SW_SCORE_QSLEEP
#define SW_SCORE_QSLEEP 4
#endif

namespace swhip {

namespace {

typedef short s2 __attribute__((ext_vector_type(2)));

constexpr int R = kScoreR;
// B, query rows per hand-over block, is a template parameter of the kernel: kScoreB or kScoreBSmall
static_assert(SW_SCORE_ASM_B == kScoreB && SW_SCORE_ASM_B_SMALL == kScoreBSmall, "sw_score_asm.inc was generated for other block sizes");
#ifndef SW_SCORE_D
#define SW_SCORE_D SW_SCORE_ASM_D
#endif
constexpr int D = SW_SCORE_D;  // blocks of slack between neighbouring waves (ring depth)
#ifndef SW_SCORE_D_SMALL
#define SW_SCORE_D_SMALL (SW_SCORE_D * SW_SCORE_ASM_B / SW_SCORE_ASM_B_SMALL)  // the same rows of slack
#endif

__device__ __forceinline__ s2 pk_max(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }

// One query row over the wave's R columns (H[r] holds H(x-1, col r) on entry and H(x, col r) on exit):
//   hl = left_in; hd = diag_in;
//   for r: hu = H[r]; m = max(hl, hu); if (r odd) best = max(best, m);
//          t = sat_sub_u16(m, |gap|); d = sat_add_i16(hd, tab[r]); h = max(t, d); hd = hu; hl = h; H[r] = h;
// The row loop of a block is hand-scheduled gfx950 assembly, generated by gen_score_asm.py into
// sw_score_asm.inc (register map and scheduling rules are documented there).
static_assert(D >= 2 && kScoreB <= 16 && D * kScoreB <= 36 && kScoreBSmall <= kScoreB,
              "a wave's ring is D * B rows of 256 bytes; 17 rings share 160 KB of LDS; one query word = B rows");
typedef int i4 __attribute__((ext_vector_type(4)));

// Every wait of this kernel is bounded in TIME (s_memrealtime, 100 MHz), not in iterations: a sleep that returns
// early must not shorten the budget.  A wait that runs out (a bug, not a legal state: dependencies always point to
// earlier tickets, whose workgroups are resident) stores a code in *err and ENDS THE WAVE on the spot (s_endpgm
// waits for the wave's outstanding memory operations by itself): it never touches memory again with unsynchronised
// state.  Every other wave notices *err in its own wait loops and ends too, so the grid drains within microseconds.
// The deadline lives in LDS, written once at kernel start, so a wait loop carries no state; the wave ends through an
// asm statement the compiler sees as an ordinary instruction, so the bail-out costs the hot path no control flow
// (a C++ `return` from inside the loops tripled the kernel's SGPR spills and cost 8 % on the bench shape).
constexpr unsigned long long kWaitTicks = 400000000ull;  // 4 s
__device__ __forceinline__ void die(uint32_t code, uint32_t* err) {  // wave-uniform callers only
  if (code) __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_endpgm" ::: "memory");
}
__device__ __forceinline__ void spin_check(uint32_t spin, uint32_t code, uint32_t* err, const unsigned long long* deadline) {
  if ((spin & 255u) != 255u) return;
  // (conditions made wave-uniform explicitly: the bail-out must sit behind a SCALAR branch, never behind an
  // EXEC mask -- s_endpgm is a scalar instruction and would run with EXEC = 0)
  const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  if (e != 0) die(0, err);  // somebody else gave up
  // (read through an LDS-typed pointer: a volatile load through the generic pointer becomes a flat access whose
  // aperture check has tripped the backend -- "V_CMP_NE_U32_e32 0, $src_shared_base" -- on some builds)
  typedef __attribute__((address_space(3))) const volatile unsigned long long* lds_u64_t;
  const unsigned long long dl = *(lds_u64_t)deadline;
  const uint32_t dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dl);
  const uint32_t dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dl >> 32));
  if (__builtin_amdgcn_s_memrealtime() >= (((unsigned long long)dhi << 32) | dlo)) die(code, err);
}

#ifdef SW_SCORE_TIMELINE  // (development build, scratch/score_timeline.sh: where a wave's time goes, in 10 ns ticks)
__device__ uint32_t g_score_tl[256 * 16 * 8];
__device__ __forceinline__ uint32_t tl_now() { return (uint32_t)__builtin_amdgcn_s_memrealtime(); }
struct TlAcc { uint32_t queue = 0, setup = 0, ready = 0, done = 0, glb = 0, strips = 0, tail = 0; };
#define TL(...) __VA_ARGS__
#else
#define TL(...)
#endif

// Spin on a progress counter written by another workgroup (sc1 load = L1 bypass).
__device__ __forceinline__ void wait_ge(const uint32_t* ctr, uint32_t need, uint32_t* err,
                                        const unsigned long long* deadline TL(, uint32_t* acc = nullptr)) {
#ifdef SW_SCORE_TIMELINE
  if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return;
  const uint32_t t0 = tl_now();
#endif
  for (uint32_t spin = 0;; spin++) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) {
      TL(if (acc) *acc += tl_now() - t0);
      return;
    }
    spin_check(spin, 1u, err, deadline);
    __builtin_amdgcn_s_sleep(SW_SCORE_GSLEEP);
  }
}

// Consumer side of the inter-pass hand-off: ONE relaxed poll -> ONE agent-scope acquire (invalidates this
// CU's L1) -> s_waitcnt vmcnt(0) -> loads.  The producer stores the column with write-through (sc1) stores,
// drains them (vmcnt(0)) and only then adds to the counter, so no release fence is needed on its side.
// Returns the number of published blocks.  Runs once per PUB blocks, not per block.
__device__ __forceinline__ uint32_t acquire_blocks(const uint32_t* ctr, uint32_t need, uint32_t* err,
                                                   const unsigned long long* deadline TL(, uint32_t* acc = nullptr)) {
  wait_ge(ctr, need, err, deadline TL(, acc));
  const uint32_t v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return v;
}

// B rows of 64 dwords, HBM -> LDS without passing through VGPRs (global_load_lds_dword: the LDS address is
// the wave-uniform base + 4 * lane, the global address is per lane).  Completion: the issuing wave's vmcnt.
template <int B>
__device__ __forceinline__ void dma_block(const uint32_t* g_lane, uint32_t* lds_base) {
  typedef __attribute__((address_space(1))) const uint32_t* gptr_t;
  typedef __attribute__((address_space(3))) uint32_t* lptr_t;
#pragma unroll
  for (int i = 0; i < B; i++)
    __builtin_amdgcn_global_load_lds((gptr_t)(g_lane + (size_t)i * 64), (lptr_t)(lds_base + i * 64), 4, 0, 0);
}

// Intra-workgroup flags live in LDS.  Relaxed workgroup-scope atomics keep the compiler from caching or
// fusing them; ordering against the data comes from the in-order LDS pipe plus the explicit waits below.
__device__ __forceinline__ uint32_t lds_flag_load(const uint32_t* f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_store(uint32_t* f, uint32_t v) {
  __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spin on an LDS flag, bounded like wait_ge
template <int SLEEP = SW_SCORE_LSLEEP>
__device__ __forceinline__ void lds_wait_ge(const uint32_t* f, uint32_t need, uint32_t* err,
                                            const unsigned long long* deadline TL(, uint32_t* acc = nullptr)) {
#ifdef SW_SCORE_TIMELINE
  if (lds_flag_load(f) >= need) {
    asm volatile("" ::: "memory");
    return;
  }
  const uint32_t t0 = tl_now();
#endif
  for (uint32_t spin = 0;; spin++) {
    if (lds_flag_load(f) >= need) {
      asm volatile("" ::: "memory");
      TL(if (acc) *acc += tl_now() - t0);
      return;
    }
    spin_check(spin, 2u, err, deadline);
    __builtin_amdgcn_s_sleep(SLEEP);
  }
}

__device__ __forceinline__ uint32_t lds_addr(const uint32_t* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)p;
}
__device__ __forceinline__ int pk2(short lo, short hi) { return (int)(((uint32_t)(uint16_t)hi << 16) | (uint16_t)lo); }

#ifndef SW_SCORE_PRIO_W4
#define SW_SCORE_PRIO_W4 0  // (experiments: the issue priorities in 4-wave workgroups too)
#endif
__device__ __forceinline__ void set_prio(int p) {  // wave-uniform
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// units a workgroup's first wave may run ahead of its slowest wave (entries of the LDS unit queue)
constexpr uint32_t QN = 8;

// MULTI: several queries in one pass, stacked into one column of row blocks (ScoreParams::blkinfo): a strip's
// substitution tables, its unit bookkeeping and the pipeline start are paid once per batch instead of once per query
// -- what a 150-nt query alone amortises badly.  Every query starts on a block boundary; at its first block H, the
// running maximum and the diagonal start from zero, after its last block the maxima go to that query's slot_scores.
template <int W, bool MULTI, int B>
__global__ void __launch_bounds__(W * 64, W == 16 ? 4 : W == 8 ? 2 : 3) sw_score_kernel(ScoreParams p) {
  static_assert(B == kScoreB || B == kScoreBSmall, "the unrolled row blocks exist for these two sizes");
  // H column entering each wave: [wave][slot][row][lane], D slots of B rows, written by the wave on its
  // left (wave 0: the previous unit's column, DMA'd from HBM).  Two rows of padding: the row loop
  // prefetches one row past the block it works on.  Ring W is written by the last wave and never read.
  constexpr int DD = B == kScoreB ? D : SW_SCORE_D_SMALL;  // ring depth in blocks: the same rows of slack for either block size
  static_assert(DD >= 2 && DD * B <= 36, "a wave's ring is DD * B rows of 256 bytes; 17 rings share 160 KB of LDS");
  constexpr int RING = DD * B * 64;  // dwords per wave
  __shared__ uint32_t lds[(W + 1) * RING + 2 * 64];
  __shared__ uint32_t zeros[(B + 2) * 64];  // the H column left of a group's first strip: H(x, 0) = 0
  __shared__ uint32_t ready[W + 1];   // ready[w]: blocks written into wave w's input ring (cumulative over units)
  __shared__ uint32_t done[W + 1];    // done[w]:  blocks of its input ring wave w has released (cumulative)
  __shared__ uint32_t started[64];    // started[w]: units wave w has fetched from the queue
  __shared__ uint32_t uq[QN];         // unit queue: wave 0 draws the tickets, the other waves follow
  __shared__ uint32_t uq_cnt;
  __shared__ unsigned long long deadline;  // s_memrealtime value after which a wait gives up (spin_expired)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t ring_lds = lds_addr(lds), zeros_lds = lds_addr(zeros);
  const uint32_t Q = p.Q;
  const uint32_t nblk = MULTI ? p.n_blocks : (Q + B - 1) / B;
  const uint32_t gapabs = __builtin_amdgcn_readfirstlane(((uint32_t)(-p.gap) & 0xffffu) * 0x10001u);
  const short mt = (short)p.match, ms = (short)p.miss;
  const uint32_t PUB = p.pub;  // blocks per publication of the inter-unit boundary (8; 2 for small databases)

  for (int i = threadIdx.x; i < (B + 2) * 64; i += W * 64) zeros[i] = 0;
  if (threadIdx.x < 64) started[threadIdx.x] = 0;
  if (threadIdx.x <= W) {
    ready[threadIdx.x] = 0;
    done[threadIdx.x] = 0;
  }
  if (threadIdx.x < QN) uq[threadIdx.x] = 0xffffffffu;  // "no unit": a stale entry can only end a wave
  if (threadIdx.x == 0) {
    uq_cnt = 0;
    deadline = __builtin_amdgcn_s_memrealtime() + kWaitTicks;
  }
  __syncthreads();  // the only workgroup barrier: from here on the waves are coupled by LDS flags alone
  // Operand shapes the addressing below relies on (the host computes them; a mismatch must be an error code, not
  // a memory fault): a boundary column of nblk * B rows fits a ring slot, and there is a ring.
  if ((uint64_t)nblk * B * 64 > p.gbound_stride || p.ring_slots == 0) {
    if (threadIdx.x == 0) __hip_atomic_store(p.err, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;  // (uniform over the whole grid, before any wave waits for another)
  }

  // `base`: blocks of all earlier units of this workgroup; flags and ring slots count blocks cumulatively, so
  // a wave that finishes its strip goes straight on to the next unit while its right-hand neighbours drain.
  uint32_t base = 0;
  TL(TlAcc tl; const uint32_t tl_begin = tl_now(); uint32_t tl_mark = tl_begin);
  for (uint32_t k = 0;; k++, base += nblk) {
    uint32_t unit;
    TL(tl_mark = tl_now());
    if (wave == 0) {
      if (k >= QN) {  // the queue entry about to be overwritten has been read by every wave
        for (uint32_t spin = 0;; spin++) {
          const uint32_t v = (lane > 0 && lane < W) ? lds_flag_load(&started[lane]) : 0xffffffffu;
          if (__builtin_amdgcn_ballot_w64(v < k - QN + 1) == 0) break;
          spin_check(spin, 3u, p.err, &deadline);
          __builtin_amdgcn_s_sleep(SW_SCORE_QSLEEP);
        }
      }
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(p.ticket, 1u);
      unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      lds_flag_store(&uq[k % QN], unit);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      lds_flag_store(&uq_cnt, k + 1);
    } else {
      lds_wait_ge(&uq_cnt, k + 1, p.err, &deadline TL(, &tl.queue));
      unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_flag_load(&uq[k % QN]));
      lds_flag_store(&started[wave], k + 1);
    }
    if (unit >= p.n_units) break;
    TL(tl.tail = tl_now() - tl_begin);  // (left at its last value: when this wave took its last real unit)

    // a unit is up to W consecutive strips of the database's strip stream
    const ScoreUnit U = p.units[unit];
    // the unit table is host-built (sw_lib.cpp: build_units, checked by sw_hip_test_units); what the addressing
    // below assumes about an entry is checked here once per unit, so a bad table is an error code, not a fault
    if (U.count == 0 || U.count > (uint32_t)W || (uint64_t)U.first_elem + U.count > p.n_elems ||
        (U.dep != 0xffffffffu && U.dep >= unit) || (U.reuse != 0xffffffffu && U.reuse >= unit)) {
      die(4u, p.err);
    }
    if ((uint32_t)wave >= U.count) {  // only in the last unit of a lane
      lds_flag_store(&done[wave], base + nblk);
      continue;
    }
    const uint32_t elem = U.first_elem + wave;
    const uint32_t gi = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.elem_group[elem]);
    if (gi >= p.n_groups) {
      die(6u, p.err);
    }
    const ScoreGroup G = p.groups[gi];
    const uint32_t strip = elem - G.strip_base;
    if (elem < G.strip_base || strip >= G.nstrips || (uint64_t)G.first_slot + 128 > p.n_slots ||
        (strip != 0 && wave == 0 && U.dep == 0xffffffffu)) {  // a chained unit always has a producer
      die(7u, p.err);
    }
    const bool first = strip == 0;                // H(x, 0) = 0 on my left
    const bool last = strip + 1 == G.nstrips;     // nobody needs my last column
    const bool ring_in = !first && wave > 0;      // left column comes from the wave on my left
    const bool glb_in = !first && wave == 0;      // ... from the previous unit (another workgroup), via HBM
    const bool lds_out = wave + 1 < W;            // handshake with the right-hand wave (even if it ignores the data)
    const bool glb_out = !last && (uint32_t)wave + 1 == U.count;  // my last column feeds the next unit
    // every target of the group ends within the first 8 columns of this (last) strip: the 8-column row loop
#ifdef SW_SCORE_NO_HALF  // (experiments: scratch/variants)
    const bool half = false;
#else
    const bool half = last && G.last_cols >= 1 && G.last_cols <= (uint32_t)R / 2;
#endif
    // nobody has to wait for me before overwriting my input ring during this unit
    if (!ring_in) lds_flag_store(&done[wave], base + nblk);

    const uint32_t Tlo = p.lens[G.first_slot + lane], Thi = p.lens[G.first_slot + 64 + lane];
    // full-matrix path (T < 2Q) skips column Tp: the last real column when T % 64 == 0
    // (rust/src/algorithm/mod.rs:98,178); H(1,T) is handled by the host's re-run rule.
    const uint32_t Elo = (Tlo < 2 * Q && Tlo > 0 && (Tlo & 63) == 0) ? Tlo - 1 : Tlo;
    const uint32_t Ehi = (Thi < 2 * Q && Thi > 0 && (Thi & 63) == 0) ? Thi - 1 : Thi;
    const uint32_t wlo = p.words[(size_t)elem * 128 + lane];
    const uint32_t whi = p.words[(size_t)elem * 128 + 64 + lane];
    // substitution scores and H.  TQ[r] = the scores of column r against A, C, G, T (one register quad per column:
    // the row loop picks the entry of its row's nucleotide through VGPR index mode, sw_score_asm.inc); HV = 16
    // packed H values as 4 quads
    i4 TQ[R], HV[4];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t col = strip * R + r;
      const uint32_t clo = (wlo >> (2 * r)) & 3u, chi = (whi >> (2 * r)) & 3u;
      const bool vlo = col < Elo, vhi = col < Ehi;
#pragma unroll
      for (int c = 0; c < 4; c++) TQ[r][c] = pk2((vlo && clo == (uint32_t)c) ? mt : ms, (vhi && chi == (uint32_t)c) ? mt : ms);
      HV[r >> 2][r & 3] = 0;
    }
    // Inter-unit boundary: ticket U.dep produced the H column this unit's wave 0 consumes; it lives in ring
    // slot U.dep % ring_slots, this unit's own output in slot unit % ring_slots.  prog[2t]: blocks published by
    // ticket t's last wave; prog[2t+1]: ticket t's wave 0 has finished reading its input slot.
    const uint32_t dep = glb_in ? U.dep : unit;  // (a chained unit always has a producer: build_units)
    const uint32_t* gin = p.gbound + (size_t)(dep % p.ring_slots) * p.gbound_stride;
    uint32_t* gout = p.gbound + (size_t)(unit % p.ring_slots) * p.gbound_stride;
    const uint32_t* prog_in = p.prog + 2 * (size_t)dep;
    uint32_t* prog_out = p.prog + 2 * (size_t)unit;
    uint32_t* const my_in_base = lds + (size_t)wave * RING;
    uint32_t* const my_out_base = lds + (size_t)(wave + 1) * RING;
    int best = 0;

    uint32_t avail = 0;  // wave 0 of a chained unit: blocks of the producer's column known to be published
    if (glb_out && U.reuse != 0xffffffffu) {
      // the previous content of my output slot was read by ticket U.reuse: its wave 0 must be done with it
      wait_ge(p.prog + 2 * (size_t)U.reuse + 1, 1, p.err, &deadline TL(, &tl.glb));
    }
    if (glb_in) {
      avail = acquire_blocks(prog_in, 1, p.err, &deadline TL(, &tl.glb));
      dma_block<B>(gin + lane, my_in_base + (base % DD) * (B * 64));
    }
    int diag = 0;
    uint32_t goff = (uint32_t)lane * 4u;
    uint32_t qw_next = p.qwords[0];  // fetched one block ahead (the array has a padding entry)
    uint32_t binfo_next = MULTI ? p.blkinfo[0] : 0u;
    // readfirstlane returns a (signed) int: widen through uint32_t, never sign-extend an address half
    const uint64_t gout_u = (uint64_t)(uintptr_t)gout;
    const uint32_t gout_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gout_u);
    const uint32_t gout_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gout_u >> 32));
    const uint64_t gout_s = ((uint64_t)gout_hi << 32) | (uint64_t)gout_lo;
#ifdef SW_SCORE_TIMELINE
    {  // set-up ends when the tables are built: make the loads land before the clock is read
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int sink = TQ[0][0] ^ TQ[R - 1][3];
      asm volatile("" ::"v"(sink));
      tl.setup += tl_now() - tl_mark;
      tl.strips++;
    }
#endif
    for (uint32_t blk = 0; blk < nblk; blk++) {
      const uint32_t cb = base + blk;  // cumulative block number
      const uint32_t slot = cb % DD;
      if (glb_in) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA of this block has landed
        if (blk + 1 < nblk) {
          if (avail < blk + 2) avail = acquire_blocks(prog_in, blk + 2, p.err, &deadline TL(, &tl.glb));
          dma_block<B>(gin + (size_t)(blk + 1) * B * 64 + lane, my_in_base + ((cb + 1) % DD) * (B * 64));
        }
      }
      // ---- wait for my input block and for room in my neighbour's ring
      if (ring_in) lds_wait_ge<SW_SCORE_RSLEEP>(&ready[wave], cb + 1, p.err, &deadline TL(, &tl.ready));
      if (lds_out && cb >= (uint32_t)DD) lds_wait_ge<SW_SCORE_DSLEEP>(&done[wave + 1], cb + 1 - DD, p.err, &deadline TL(, &tl.done));

      // ---- issue priority for this block.  The SIMD serves its resident waves oldest first, so without a hint the
      // low-numbered waves of a workgroup sprint until their output ring is full, stall, and leave their SIMD with
      // three or two runnable waves -- and a SIMD issues the dependent packed chain of ONE wave at a third of its
      // rate.  The wave whose right-hand neighbour has caught up with it (nothing pending in the ring: the neighbour
      // is about to starve) is the one the pipeline waits for: it gets priority 3, a wave that is a full ring ahead
      // priority 0, the last wave of the pipeline always 3.  (Measured with the per-wave clocks of
      // scratch/score_timeline.sh, 150-nt reads: waiting for ring room 39 % of all wave time -> 14 %, for the left
      // neighbour 5 % -> 21 %, kernel 17.8 -> 16.5 ms; bench shape 68.7 -> 67.1 ms.  A static priority rising with
      // the wave index loses 2 %, variants that also weigh the input side gain nothing.)
      if constexpr (W == 16 || SW_SCORE_PRIO_W4) {
        if (lds_out) {
          const int lead = (int)(cb - (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_flag_load(&done[wave + 1])));
#if !defined(SW_SCORE_PRIO_MAP) || SW_SCORE_PRIO_MAP == 0
          set_prio(3 - (lead > 0 ? (lead * 3 + DD - 1) / DD : 0));
#elif SW_SCORE_PRIO_MAP == 1  // (experiments) steeper
          set_prio(3 - (lead > 0 ? lead : 0));
#elif SW_SCORE_PRIO_MAP == 2  // flatter at the top
          set_prio(lead <= 1 ? 3 : lead <= 3 ? 2 : lead <= 4 ? 1 : 0);
#elif SW_SCORE_PRIO_MAP == 3  // a strip nobody waits for is not urgent
          set_prio(last ? 1 : 3 - (lead > 0 ? (lead * 3 + DD - 1) / DD : 0));
#elif SW_SCORE_PRIO_MAP == 4  // two levels
          set_prio(lead <= 2 ? 3 : 0);
#endif
        } else {
          set_prio(3);
        }
      }
      uint32_t qw = __builtin_amdgcn_readfirstlane(qw_next);
      qw_next = p.qwords[blk + 1];
      uint32_t nrows, binfo = 0;
      if (MULTI) {
        binfo = __builtin_amdgcn_readfirstlane(binfo_next);
        binfo_next = p.blkinfo[blk + 1];  // (the table has a padding entry)
        nrows = binfo & 15u;
        if (binfo & kBlkFirst) {  // a new query starts here
#pragma unroll
          for (int q4 = 0; q4 < 4; q4++) HV[q4] = i4{0, 0, 0, 0};
          best = 0;
          diag = 0;
        }
        goff = (uint32_t)lane * 4u + blk * (uint32_t)(B * 256);  // a shorter block before this one left it behind
      } else {
        const uint32_t x0 = blk * B;
        nrows = (Q - x0) < (uint32_t)B ? (Q - x0) : (uint32_t)B;
      }
      const uint32_t n_rows = __builtin_amdgcn_readfirstlane(nrows);
      // (LDS byte addresses by integer arithmetic from the two array bases: a select between two pointers followed
      // by the cast to the LDS address space does not fold on every build and then trips the backend)
      uint32_t addr = first ? zeros_lds + (uint32_t)lane * 4u
                            : ring_lds + ((uint32_t)wave * RING + slot * (B * 64) + (uint32_t)lane) * 4u;
      uint32_t waddr = ring_lds + ((uint32_t)(wave + 1) * RING + slot * (B * 64) + (uint32_t)lane) * 4u;
      uint32_t tmp_i, tmp_t, tmp_m0;
      // the row loop: nrows rows x 16 columns x 2 targets per lane (sw_score_asm.inc), in the form with or without
      // the store of my last column to the next unit's boundary slot (decided here, once per block, not per row)
#define SW_SCORE_ROWS(TEXT)                                                                                             \
  asm volatile(TEXT                                                                                                     \
               : "+{v[48:51]}"(HV[0]), "+{v[52:55]}"(HV[1]), "+{v[56:59]}"(HV[2]), "+{v[60:63]}"(HV[3]),                \
                 [best] "+v"(best), [diag] "+v"(diag), [addr] "+v"(addr), [waddr] "+v"(waddr), [goff] "+v"(goff),       \
                 [qw] "+s"(qw), [i] "=&s"(tmp_i), [t] "=&s"(tmp_t), [m0s] "=&s"(tmp_m0)                                 \
               : "{v[64:67]}"(TQ[0]), "{v[68:71]}"(TQ[1]), "{v[72:75]}"(TQ[2]), "{v[76:79]}"(TQ[3]),                    \
                 "{v[80:83]}"(TQ[4]), "{v[84:87]}"(TQ[5]), "{v[88:91]}"(TQ[6]), "{v[92:95]}"(TQ[7]),                    \
                 "{v[96:99]}"(TQ[8]), "{v[100:103]}"(TQ[9]), "{v[104:107]}"(TQ[10]), "{v[108:111]}"(TQ[11]),            \
                 "{v[112:115]}"(TQ[12]), "{v[116:119]}"(TQ[13]), "{v[120:123]}"(TQ[14]), "{v[124:127]}"(TQ[15]),        \
                 [n] "s"(n_rows), [g] "s"(gapabs), [gb] "s"(gout_s), [one] "s"(0x00010001u)                             \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "scc", "memory")
      if (nrows == (uint32_t)B) {  // a full block: the unrolled form (no loop counter, no branch, immediate offsets)
        if constexpr (B == kScoreB) {
          if (glb_out) SW_SCORE_ROWS(SW_SCORE_BLOCK_STORE_ASM);
          else if (half) SW_SCORE_ROWS(SW_SCORE_BLOCK_HALF_ASM);  // (a last strip never stores: half excludes glb_out)
          else SW_SCORE_ROWS(SW_SCORE_BLOCK_ASM);
        } else {
          if (glb_out) SW_SCORE_ROWS(SW_SCORE_BLOCK_STORE_ASM_S);
          else if (half) SW_SCORE_ROWS(SW_SCORE_BLOCK_HALF_ASM_S);
          else SW_SCORE_ROWS(SW_SCORE_BLOCK_ASM_S);
        }
      } else {                     // the last block of a query
        if (glb_out) SW_SCORE_ROWS(SW_SCORE_ROWS_STORE_ASM);
        else if (half) SW_SCORE_ROWS(SW_SCORE_ROWS_HALF_ASM);
        else SW_SCORE_ROWS(SW_SCORE_ROWS_ASM);
      }
#undef SW_SCORE_ROWS
      // ---- hand the block over: my LDS writes are complete before the flags move
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lds_out) lds_flag_store(&ready[wave + 1], cb + 1);
      if (ring_in) lds_flag_store(&done[wave], cb + 1);
      if (glb_out && ((blk + 1) % PUB == 0 || blk + 1 == nblk)) {
        // publish: this wave is the only one storing the boundary; drain its stores, then count
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          const uint32_t prev = (blk + 1 == nblk) ? ((nblk - 1) / PUB) * PUB : blk + 1 - PUB;
          __hip_atomic_fetch_add(prog_out, blk + 1 - prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (MULTI && (binfo & kBlkLast)) {  // this query is done: its maxima go to its own score array
        s2 bq = as_s2((uint32_t)best);
#pragma unroll
        for (int r = 1; r < R; r += 2) bq = pk_max(bq, as_s2((uint32_t)HV[r >> 2][r & 3]));
        int32_t* const sq = p.slot_scores + (size_t)(binfo >> kBlkQueryShift) * p.n_slots + G.first_slot;
        atomicMax(sq + lane, (int)bq.x);
        atomicMax(sq + 64 + lane, (int)bq.y);
      }
    }
    if (!MULTI) {
      // the running maximum skipped the odd columns of the last row
      s2 bb = as_s2((uint32_t)best);
#pragma unroll
      for (int r = 1; r < R; r += 2) bb = pk_max(bb, as_s2((uint32_t)HV[r >> 2][r & 3]));
      // every wave covered different columns of its 128 targets: combine in memory (zeroed before the launch)
      atomicMax(p.slot_scores + G.first_slot + lane, (int)bb.x);
      atomicMax(p.slot_scores + G.first_slot + 64 + lane, (int)bb.y);
      if (p.strip_best) p.strip_best[(size_t)elem * 64 + lane] = __builtin_bit_cast(uint32_t, bb);
    }
    if (wave == 0) {
      // my input slot of the HBM ring may be overwritten (all my DMA reads have landed: vmcnt(0) above)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(prog_out + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef SW_SCORE_TIMELINE
  if (lane == 0 && blockIdx.x < 256 && wave < 16) {
    uint32_t* o = g_score_tl + ((size_t)blockIdx.x * 16 + wave) * 8;
    o[0] = tl_now() - tl_begin;
    o[1] = tl.queue;
    o[2] = tl.setup;
    o[3] = tl.ready;
    o[4] = tl.done;
    o[5] = tl.glb;
    o[6] = tl.strips;
    o[7] = tl.tail;
  }
#endif
}

}  // namespace

// Waves per workgroup.  16 (one 1024-thread workgroup per CU, four waves per SIMD) is the throughput
// configuration.  A SMALL database has fewer strips than the GPU has SIMDs: with 4-wave workgroups its units
// spread over up to 256 CUs at one wave per SIMD instead of packing 16 waves onto a few CUs.
// SW_HIP_SCORE_WAVES=4|8|16 forces one value (experiments).
int score_waves_per_group(uint32_t total_strips, int cus, uint32_t n_groups) {
  static const int forced = [] {
    const char* e = getenv("SW_HIP_SCORE_WAVES");
    const int v = e ? atoi(e) : 0;
    return (v == 4 || v == 8 || v == 16) ? v : 0;
  }();
  if (forced) return forced;
  // ... and so has a database of a handful of groups however long its targets are: its few chains advance one
  // unit at a time, and small units hand over sooner (the capacity benchmark's one pair of 4 102 x 4.2e6, 3 groups of
  // ~1 500 strips: 1 280-1 430 GCUPS with 16 waves, 2 070-2 370 with 4)
  static const uint32_t few = getenv("SW_HIP_SCORE_W4_GROUPS") ? (uint32_t)atoi(getenv("SW_HIP_SCORE_W4_GROUPS")) : 16;  // (experiments)
  return (total_strips <= (uint32_t)cus * 8u || n_groups <= few) ? 4 : 16;
}
// what 160 KB of LDS holds: 17 / 9 / 5 rings of 9 KB + the zero column
int score_workgroups_per_cu(int waves) { return waves == 4 ? 3 : 1; }
uint32_t score_queue_depth() { return QN; }
void score_loop_counts(uint32_t* packed_per_row, uint32_t* aux_per_2rows) {
  *packed_per_row = SW_SCORE_ASM_PACKED_PER_ROW;
  *aux_per_2rows = SW_SCORE_ASM_AUX_PER_2ROWS;
}

// Rows per hand-over block of a pass: the small block (ring of twice as many blocks) for every pass of one query; a
// stacked batch of queries keeps the large one (its block table is built for it), and so does the allocation-free
// entry point.  History: the small block came in for passes that are a few chains of strips (the capacity
// benchmark's one pair of 4 102 x 4.2e6, 3 groups of ~1 500 strips: 1 210 -> 1 405 GCUPS; a package of 10 queries x
// 100 targets of 10-200 kb, ~60 groups: 33.6 -> 27.0 ms) and cost the throughput shapes 4-5 % until the waves began
// to set their issue priority by their lead (sw_score_kernel); since then it is the faster one everywhere.
// SW_HIP_SCORE_BLOCK=12 keeps the large block everywhere (experiments).
int score_block_rows(int waves, bool multi, uint32_t n_groups) {
  static const bool large_only = [] {
    const char* e = getenv("SW_HIP_SCORE_BLOCK");
    return e && atoi(e) == kScoreB;
  }();
  (void)n_groups;
  // (the small block is instantiated for 4 and 16 waves: 8 is an experiment's choice, SW_HIP_SCORE_WAVES, never the library's)
  return (!multi && !large_only && (waves == 4 || waves == 16)) ? kScoreBSmall : kScoreB;
}

hipError_t launch_score(const ScoreParams& p, int n_workgroups, int waves, int block_rows, hipStream_t stream) {
  if (p.n_units == 0 || n_workgroups <= 0) return hipSuccess;
  if (p.blkinfo && (p.n_blocks == 0 || p.strip_best)) return hipErrorInvalidValue;  // (a batch keeps no per-strip maxima)
  const bool multi = p.blkinfo != nullptr;
  if (block_rows == kScoreBSmall) {
    if (multi) return hipErrorInvalidValue;
    if (waves == 4) hipLaunchKernelGGL((sw_score_kernel<4, false, kScoreBSmall>), dim3(n_workgroups), dim3(4 * 64), 0, stream, p);
    else if (waves == 16) hipLaunchKernelGGL((sw_score_kernel<16, false, kScoreBSmall>), dim3(n_workgroups), dim3(16 * 64), 0, stream, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (block_rows != kScoreB) return hipErrorInvalidValue;
#define SW_LAUNCH_SCORE(WAVES)                                                                                    \
  do {                                                                                                            \
    if (multi) hipLaunchKernelGGL((sw_score_kernel<WAVES, true, kScoreB>), dim3(n_workgroups), dim3(WAVES * 64), 0, stream, p);  \
    else hipLaunchKernelGGL((sw_score_kernel<WAVES, false, kScoreB>), dim3(n_workgroups), dim3(WAVES * 64), 0, stream, p); \
  } while (0)
  if (waves == 16) SW_LAUNCH_SCORE(16);
  else if (waves == 8) SW_LAUNCH_SCORE(8);
  else if (waves == 4) SW_LAUNCH_SCORE(4);
  else return hipErrorInvalidValue;
#undef SW_LAUNCH_SCORE
  return hipGetLastError();
}

}  // namespace swhip

#ifdef SW_SCORE_TIMELINE
extern "C" int sw_hip_debug_score_timeline(uint32_t* out, size_t n_words) {
  if (n_words > sizeof(swhip::g_score_tl) / 4) n_words = sizeof(swhip::g_score_tl) / 4;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(swhip::g_score_tl), n_words * 4);
}
#endif
