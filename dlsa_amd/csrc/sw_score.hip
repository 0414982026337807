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
//    query residue of a row is wave-uniform: the substitution scores of a row are four pre-built
//    register tables (one per query nucleotide) selected by a scalar branch -- no LDS profile lookup and
//    no VALU compare per cell;
//  * per cell 4.5 packed VALU instructions: v_pk_max_i16(left, up), v_pk_sub_u16 clamp (the unsigned
//    saturating subtract of |gap| is also the max(.,0)), v_pk_add_i16 clamp (diag + s, saturation = the
//    overflow detector), v_pk_max_i16, and the running maximum on every second column only (max(left,up)
//    of an odd column covers the new even-column cell of this row and the odd-column cell of the previous
//    row);
//  * the W waves of a workgroup own consecutive 16-column strips and run as a software pipeline over
//    16-row blocks; the H column between neighbouring strips travels through LDS (double-buffered),
//    one s_barrier per block;
//  * a work unit is (group, pass): W*16 columns of 128 targets against the whole query.  Persistent
//    workgroups (one per CU) pull units from an atomic ticket in (group, pass) order, so the passes of a
//    group run concurrently on different CUs as a chain: pass p streams its last H column into a ring
//    slot in HBM (write-through stores, published every 8 blocks through a progress counter) and pass
//    p+1 trails it by a few blocks.  Dependencies always point to an earlier ticket, whose workgroup is
//    resident, so the chain cannot deadlock; fine-grained units keep all 256 CUs busy to the end.
//
// Saturation: diag + s uses the clamp bit, so a lane whose true maximum exceeds 32766 reports 32767 and
// the host re-runs that pair with the int32 exact kernel.
#include "sw_internal.h"

namespace swhip {

namespace {

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

constexpr int R = kScoreR;
constexpr int B = kScoreB;

__device__ __forceinline__ s2 pk_max(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pk_add_sat(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s2 pk_subu_sat(s2 a, s2 b) {
  return (s2)__builtin_elementwise_sub_sat((u2)a, (u2)b);
}
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }

// One query row over the wave's R columns.  H[r] holds H(x-1, col r) on entry and H(x, col r) on exit.
// Reference form (kept for documentation and as the source of truth for the asm below):
//   hl = left_in; hd = diag_in;
//   for r: hu = H[r]; m = max(hl, hu); if (r odd) best = max(best, m);
//          t = sat_sub_u16(m, |gap|); d = sat_add_i16(hd, tab[r]); h = max(t, d); hd = hu; hl = h; H[r] = h;
//
// Hand-scheduled in blocks of four cells so that H[] is updated IN PLACE (the compiler's version copied
// the 16 H registers back at the bottom of every row) with the next cell's diag + s computed one cell
// ahead, before the old H value is overwritten.  gfx950 needs one wait state between a packed op and a
// dependent packed op (hipcc pads `s_nop 0` itself but pads nothing inside asm): every dependent pair
// below is separated by an independent instruction or an explicit s_nop.
//   d  : in  = diag + s of the block's first cell, out = diag + s of the next block's first cell
//   tn : tab of the cell after the block (any register for the last block; that result is unused)
__device__ __forceinline__ void sw_cells4(s2& h0, s2& h1, s2& h2, s2& h3, s2 t1, s2 t2, s2 t3, s2 tn, s2 hl,
                                          s2& d, s2& best, uint32_t gap2) {
  s2 m, n, a;
  asm volatile(
      "v_pk_max_i16 %[m], %[hl], %[h0]\n\t"
      "v_pk_add_i16 %[a], %[h0], %[t1] clamp\n\t"
      "v_pk_sub_u16 %[m], %[m], %[g] clamp\n\t"
      "s_nop 0\n\t"
      "v_pk_max_i16 %[h0], %[m], %[d]\n\t"
      "s_nop 0\n\t"
      "v_pk_max_i16 %[m], %[h0], %[h1]\n\t"
      "v_pk_add_i16 %[d], %[h1], %[t2] clamp\n\t"
      "v_pk_sub_u16 %[n], %[m], %[g] clamp\n\t"
      "v_pk_max_i16 %[best], %[best], %[m]\n\t"
      "v_pk_max_i16 %[h1], %[n], %[a]\n\t"
      "s_nop 0\n\t"
      "v_pk_max_i16 %[m], %[h1], %[h2]\n\t"
      "v_pk_add_i16 %[a], %[h2], %[t3] clamp\n\t"
      "v_pk_sub_u16 %[m], %[m], %[g] clamp\n\t"
      "s_nop 0\n\t"
      "v_pk_max_i16 %[h2], %[m], %[d]\n\t"
      "s_nop 0\n\t"
      "v_pk_max_i16 %[m], %[h2], %[h3]\n\t"
      "v_pk_add_i16 %[d], %[h3], %[tn] clamp\n\t"
      "v_pk_sub_u16 %[n], %[m], %[g] clamp\n\t"
      "v_pk_max_i16 %[best], %[best], %[m]\n\t"
      "v_pk_max_i16 %[h3], %[n], %[a]\n\t"
      "s_nop 0"
      : [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3), [d] "+v"(d), [best] "+v"(best),
        [m] "=&v"(m), [n] "=&v"(n), [a] "=&v"(a)
      : [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [tn] "v"(tn), [hl] "v"(hl), [g] "s"(gap2));
}

__device__ __forceinline__ void sw_row(s2 (&H)[R], const s2 (&tab)[R], s2& best, s2 left_in, s2 diag_in,
                                       uint32_t gap2) {
  static_assert(R % 4 == 0, "the row is scheduled in blocks of four cells");
  s2 d = pk_add_sat(diag_in, tab[0]);
  sw_cells4(H[0], H[1], H[2], H[3], tab[1], tab[2], tab[3], tab[4], left_in, d, best, gap2);
#pragma unroll
  for (int r = 4; r < R; r += 4)
    sw_cells4(H[r], H[r + 1], H[r + 2], H[r + 3], tab[r + 1], tab[r + 2], tab[r + 3], tab[r + 4 < R ? r + 4 : r + 3],
              H[r - 1], d, best, gap2);
}

// Bounded spin on a progress counter written by another workgroup (sc1 load = L1 bypass).  A counter that
// never arrives (a bug, not a legal state: dependencies always point to earlier tickets, whose workgroups
// are resident) sets *err and lets the kernel drain instead of hanging the GPU.
__device__ __forceinline__ void wait_ge(const uint32_t* ctr, uint32_t need, uint32_t* err) {
  for (uint32_t spin = 0; spin < (1u << 24); spin++) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return;
    __builtin_amdgcn_s_sleep(16);
  }
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int W>
__global__ void __launch_bounds__(W * 64) sw_score_kernel(ScoreParams p) {
  __shared__ uint32_t lds[W * 2 * B * 64];  // [wave][block parity][row][lane]: H column entering a wave
  __shared__ uint32_t s_unit;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t Q = p.Q;
  const uint32_t nblk = (Q + B - 1) / B;
  const uint32_t gapabs = __builtin_amdgcn_readfirstlane(((uint32_t)(-p.gap) & 0xffffu) * 0x10001u);
  const short mt = (short)p.match, ms = (short)p.miss;
  constexpr uint32_t PUB = 8;  // blocks per publication of the inter-pass boundary

  for (;;) {
    if (threadIdx.x == 0) s_unit = atomicAdd(p.ticket, 1u);
    __syncthreads();
    const uint32_t unit = s_unit;
    __syncthreads();
    if (unit >= p.n_units) break;
    const ScoreUnit U = p.units[unit];
    const ScoreGroup G = p.groups[U.group];
    const uint32_t pass = U.pass;
    const uint32_t npass = (G.nstrips + W - 1) / W;
    const uint32_t Tlo = p.lens[G.first_slot + lane], Thi = p.lens[G.first_slot + 64 + lane];
    // full-matrix path (T < 2Q) skips column Tp: the last real column when T % 64 == 0
    // (rust/src/algorithm/mod.rs:98,178); H(1,T) is handled by the host's re-run rule.
    const uint32_t Elo = (Tlo < 2 * Q && Tlo > 0 && (Tlo & 63) == 0) ? Tlo - 1 : Tlo;
    const uint32_t Ehi = (Thi < 2 * Q && Thi > 0 && (Thi & 63) == 0) ? Thi - 1 : Thi;
    s2 best = {0, 0};

    const uint32_t strip = pass * W + wave;
    const bool active = strip < G.nstrips;
    uint32_t wlo = 0, whi = 0;
    if (active) {
      wlo = p.words[G.word_off + ((size_t)strip * 2 + 0) * 64 + lane];
      whi = p.words[G.word_off + ((size_t)strip * 2 + 1) * 64 + lane];
    }
    s2 tabA[R], tabC[R], tabG[R], tabT[R], H[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t col = strip * R + r;
      const uint32_t clo = (wlo >> (2 * r)) & 3u, chi = (whi >> (2 * r)) & 3u;
      const bool vlo = col < Elo, vhi = col < Ehi;
      tabA[r] = s2{(vlo && clo == 0) ? mt : ms, (vhi && chi == 0) ? mt : ms};
      tabC[r] = s2{(vlo && clo == 1) ? mt : ms, (vhi && chi == 1) ? mt : ms};
      tabG[r] = s2{(vlo && clo == 2) ? mt : ms, (vhi && chi == 2) ? mt : ms};
      tabT[r] = s2{(vlo && clo == 3) ? mt : ms, (vhi && chi == 3) ? mt : ms};
      H[r] = s2{0, 0};
    }
    // Inter-pass boundary: unit u-1 = (group, pass-1) produces the H column this unit consumes; it lives in
    // ring slot (u-1) % ring_slots, this unit's own output in slot u % ring_slots.
    const bool from_global = pass > 0;
    const bool to_global = pass + 1 < npass;
    const uint32_t* gin = p.gbound + (size_t)((unit + p.ring_slots - 1) % p.ring_slots) * p.gbound_stride;
    uint32_t* gout = p.gbound + (size_t)(unit % p.ring_slots) * p.gbound_stride;
    const uint32_t* prog_in = p.prog + (unit - (from_global ? 1u : 0u));
    uint32_t* prog_out = p.prog + unit;
    s2 diag_in = {0, 0};

    if (threadIdx.x == 0) {
      // the previous user of my output slot was read by unit (unit - ring_slots + 1): it must be finished
      if (to_global && unit + 1 >= p.ring_slots + 1) wait_ge(p.prog + (unit - p.ring_slots + 1), nblk, p.err);
      if (from_global) wait_ge(prog_in, nblk < 2 ? nblk : 2, p.err);
    }
    __syncthreads();
    // stage block 0 of the inter-pass boundary for wave 0 (every wave copies rows wave, wave+W, ..)
    if (from_global) {
      for (int i = wave; i < B; i += W)
        lds[((0 * 2 + 0) * B + i) * 64 + lane] =
            __hip_atomic_load(gin + (size_t)i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    for (uint32_t s = 0; s < nblk + W - 1; s++) {
      // prefetch the next block of the inter-pass boundary (used by wave 0 in superstep s + 1)
      uint32_t pre[(B + W - 1) / W];
      const bool do_pre = from_global && (s + 1 < nblk);
      if (do_pre) {
#pragma unroll
        for (int i = 0; i < (B + W - 1) / W; i++) {
          const int row = wave + i * W;
          if (row < B)
            pre[i] = __hip_atomic_load(gin + ((size_t)(s + 1) * B + row) * 64 + lane, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // block s + 2 is loaded in superstep s + 1: its publication must be seen before this superstep's
      // barrier (poll -> workgroup barrier -> sc1 loads)
      if (from_global && threadIdx.x == 0 && s + 2 < nblk) wait_ge(prog_in, s + 3, p.err);
      const int blk = (int)s - wave;
      if (active && blk >= 0 && (uint32_t)blk < nblk) {
        const uint32_t qw = p.qwords[blk];
        const uint32_t* in = lds + ((wave * 2 + (blk & 1)) * B) * 64 + lane;
        uint32_t* out = lds + (((wave + 1) * 2 + (blk & 1)) * B) * 64 + lane;
        const bool has_in = (wave > 0) || from_global;
        const bool lds_out = wave + 1 < W;
        const uint32_t x0 = (uint32_t)blk * B;
        const int nrows = (Q - x0) < (uint32_t)B ? (int)(Q - x0) : B;
        s2 left_next = has_in ? as_s2(in[0]) : s2{0, 0};
#pragma unroll 1
        for (int i = 0; i < nrows; i++) {
          const s2 left_in = left_next;
          if (has_in && i + 1 < nrows) left_next = as_s2(in[(i + 1) * 64]);
          const uint32_t code = (qw >> (2 * i)) & 3u;
          switch (code) {
            // Four bodies, one per query nucleotide, selected by a scalar branch.  The trailing asm comments
            // differ per case so that the optimiser cannot merge the bodies; the library is built with
            // -mllvm -structurizecfg-skip-uniform-regions=true so that this uniform switch stays a real
            // branch (structurised, old and new H would both be live across the sibling bodies and every
            // row would pay 16-32 register copies).
            case 0: sw_row(H, tabA, best, left_in, diag_in, gapabs); asm volatile("; row A"); break;
            case 1: sw_row(H, tabC, best, left_in, diag_in, gapabs); asm volatile("; row C"); break;
            case 2: sw_row(H, tabG, best, left_in, diag_in, gapabs); asm volatile("; row G"); break;
            default: sw_row(H, tabT, best, left_in, diag_in, gapabs); asm volatile("; row T"); break;
          }
          diag_in = left_in;
          if (lds_out)
            out[i * 64] = as_u32(H[R - 1]);
          else if (to_global)  // write-through store: visible to the consumer workgroup without a release
            __hip_atomic_store(gout + ((size_t)x0 + i) * 64 + lane, as_u32(H[R - 1]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!lds_out && to_global && (((uint32_t)blk + 1) % PUB == 0 || (uint32_t)blk + 1 == nblk)) {
          // publish: this wave is the only one storing the boundary; drain its stores, then count
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            const uint32_t done = (uint32_t)blk + 1;
            const uint32_t prev = ((uint32_t)blk + 1 == nblk) ? ((nblk - 1) / PUB) * PUB : done - PUB;
            __hip_atomic_fetch_add(prog_out, done - prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (do_pre) {
#pragma unroll
        for (int i = 0; i < (B + W - 1) / W; i++) {
          const int row = wave + i * W;
          if (row < B) lds[((0 * 2 + ((s + 1) & 1)) * B + row) * 64 + lane] = pre[i];
        }
      }
      __syncthreads();
    }
    // the running maximum skipped the odd columns of the last row
#pragma unroll
    for (int r = 1; r < R; r += 2) best = pk_max(best, H[r]);
    // reduce the W waves' maxima (each wave covered different columns of the same 128 targets); other
    // passes of the group add theirs with atomicMax (slot_scores is zeroed before the launch)
    lds[wave * 64 + lane] = as_u32(best);
    __syncthreads();
    if (wave == 0) {
      s2 b = best;
      for (int w = 1; w < W; w++) b = pk_max(b, as_s2(lds[w * 64 + lane]));
      atomicMax(p.slot_scores + G.first_slot + lane, (int)b.x);
      atomicMax(p.slot_scores + G.first_slot + 64 + lane, (int)b.y);
      // a unit that publishes no boundary still reports completion (ring slot reuse waits on it)
      if (!to_global && lane == 0)
        __hip_atomic_store(prog_out, nblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

constexpr int kW = 16;

}  // namespace

int score_waves_per_group() { return kW; }

hipError_t launch_score(const ScoreParams& p, int n_workgroups, hipStream_t stream) {
  if (p.n_units == 0 || n_workgroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(sw_score_kernel<kW>, dim3(n_workgroups), dim3(kW * 64), 0, stream, p);
  return hipGetLastError();
}

}  // namespace swhip
