// dlsa_amd/csrc/sw_exact.hip -- the exact (intra-task) Smith-Waterman kernels for gfx950.
//
// What it replaces: the matrix fill + best-cell selection + traceback of ONE pair, i.e.
// rust/src/algorithm/mod.rs:66-202 + rust/src/lib.rs:122-135 (full path), algorithm/mod.rs:204-445
// (ring path) and algorithm/mod.rs:518-549 + lib.rs:55-70 (straight), with the traceback priority of
// algorithm/mod.rs:665-706.  It is not a translation: the reference fills a skewed i16 matrix row by
// row; here a 64-lane wavefront sweeps a strip of 256 query rows (4 per lane) along the target as an
// anti-diagonal wavefront and passes H across lanes with a DPP wave shift; the up-to-16 waves of a
// workgroup take consecutive strips of one pair and pipeline along the target (strip boundary rows in HBM,
// progress counters in LDS); only 2 direction bits per cell are kept for the traceback walker.
//
// Arithmetic: int32 (the "int32 re-run" of the north star; equal to the reference's wrapping i16 on its
// safe domain max H <= 32767).
//
// Direction bits, per cell: bit0 = (diag + s >= max(up, left) + gap), bit1 = (up >= left).  With H > 0
// this is exactly the reference's if/else-if chain: diagonal first, then (x-1, j), then (x, j-1).  The
// walker knows H along the path (it subtracts what each step added), so no "zero" code is needed.
// Layout: word[((strip * nkb + k/16) * RPL + r) * 64 + lane], step k of that lane in bits
// [2*(15 - k%16), +2); lane's row i = strip*256 + lane*RPL + r, its column at step k is j = k - lane + 1.
#include "sw_internal.h"

namespace swhip {

namespace {

constexpr int RPL = kExactRPL;
constexpr int SH = kExactSH;

__device__ __forceinline__ int dpp_wave_shr1(int fill_lane0, int v) {
  // lane i <- lane i-1; lane 0 keeps `fill_lane0` (bound_ctrl = 0)
  return __builtin_amdgcn_update_dpp(fill_lane0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
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

// bounded spin on an LDS progress word of the wave that owns the strip above mine
__device__ __forceinline__ void wait_strip(const uint32_t* f, uint32_t need, uint32_t* err) {
  for (uint32_t spin = 0; spin < (1u << 26); spin++) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) {
      asm volatile("" ::: "memory");
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  *err = 1;
}

// One pair per workgroup.  The NW waves of the workgroup own consecutive 256-row strips (wave w: strips w,
// w + NW, ...) and run as a pipeline along the target: a strip may enter column block c once the strip above
// it has left it, i.e. 128 steps later (64 lanes of skew + 64-column hand-over blocks).  The bottom row of
// every strip goes through one HBM row per strip (same CU, same L2: plain stores, L1-bypassing loads);
// progress counters live in LDS.
template <bool DIRS>
__device__ __forceinline__ void exact_pair(const ExactPair& P, ExactOut* out, int match, int miss, int gap,
                                           uint32_t* prog, uint32_t* red, uint32_t* err) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t NW = blockDim.x >> 6;
  const uint32_t Q = P.Q, T = P.T;
  const uint32_t maxcol = P.Tlim > P.T0 ? P.Tlim : P.T0;
  const uint32_t nsteps = maxcol == 0 ? 0 : ((maxcol + 63 + 63) / 64) * 64;
  const uint32_t NC = nsteps / 64;  // hand-over events per strip
  const uint32_t nstrips = (Q + SH - 1) / SH;

  int gbv = 0;
  uint64_t gkey = ~0ull;

  if (threadIdx.x < 16) prog[threadIdx.x] = 0;
  __syncthreads();

  for (uint32_t pass = 0, strip = wave; strip < nstrips && nsteps; pass++, strip += NW) {
    const int32_t* bin = P.bnd + (size_t)(strip > 0 ? strip - 1 : 0) * P.bnd_pitch;  // bottom row of strip-1
    int32_t* bout = P.bnd + (size_t)strip * P.bnd_pitch;
    // the strip above me belongs to wave-1 in this pass, or to the last wave in the previous pass
    const uint32_t* pflag = prog + (wave > 0 ? wave - 1 : NW - 1);
    const uint32_t pbase = (wave > 0 ? pass : pass - 1) * NC;
    const uint32_t ibase = strip * SH + lane * RPL;  // 0-based first row of this lane
    uint32_t qs[RPL];
    bool valid[RPL];
    int h[RPL], bestv[RPL];
    uint32_t bestj[RPL], acc[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) {
      uint32_t i = ibase + r;
      valid[r] = i < Q;
      qs[r] = valid[r] ? P.q[i] : 0x10000u;  // never equals a u16 symbol
      h[r] = 0;
      bestv[r] = 0;
      bestj[r] = 0;
      acc[r] = 0;
    }
    const bool row1_lane = (strip == 0) && (lane == 0);
    const uint32_t lim0 = row1_lane ? P.T0 : P.Tlim;  // column limit of this lane's row r == 0
    int top_prev = 0, last_out = 0, bndreg = 0, outacc = 0;
    int tcur = 0x20000, tblock = 0;

    for (uint32_t k = 0; k < nsteps; k++) {
      if ((k & 63) == 0) {
        uint32_t jj = k + lane + 1;  // the column lane 0 reaches at step k + lane
        if (strip > 0) {
          // columns k+1 .. k+64 of the row above left its wave at step k + 127: event (k / 64) + 2
          const uint32_t ev = (k >> 6) + 2;
          wait_strip(pflag, pbase + (ev < NC ? ev : NC), err);
          bndreg = (jj <= maxcol) ? __hip_atomic_load(bin + jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        } else {
          bndreg = 0;
        }
        tblock = (jj <= T) ? (int)P.t[jj - 1] : 0x20000;
      }
      const int s_b = __builtin_amdgcn_readlane(bndreg, k & 63);
      const int s_t = __builtin_amdgcn_readlane(tblock, k & 63);
      const int top = dpp_wave_shr1(s_b, last_out);  // H(row above this lane, column j)
      tcur = dpp_wave_shr1(s_t, tcur);               // target symbol of column j
      const int j = (int)k - lane + 1;
      int up = top, dg = top_prev;
#pragma unroll
      for (int r = 0; r < RPL; r++) {
        const int lf = h[r];
        const int sc = ((int)qs[r] == tcur) ? match : miss;
        const int d = dg + sc;
        const int t2 = max(up, lf) + gap;
        int hn = max(max(d, t2), 0);
        const uint32_t lim = (r == 0) ? lim0 : P.Tlim;
        const bool act = valid[r] && j >= 1 && (uint32_t)j <= lim;
        hn = act ? hn : 0;
        if (DIRS) {
          const uint32_t code = (d >= t2 ? 1u : 0u) | (up >= lf ? 2u : 0u);
          acc[r] = (acc[r] << 2) | code;
        }
        if (hn > bestv[r]) {  // strict: keeps the first (smallest j) maximal cell of the row
          bestv[r] = hn;
          bestj[r] = (uint32_t)j;
        }
        dg = lf;
        up = hn;
        h[r] = hn;
      }
      top_prev = top;
      last_out = h[RPL - 1];
      // bottom row of the strip leaves through lane 63: column k - 62 at step k
      const int s_o = __builtin_amdgcn_readlane(last_out, 63);
      outacc = (lane == (int)(k & 63)) ? s_o : outacc;
      if ((k & 63) == 63) {
        if (strip + 1 < nstrips) {
          const int col = (int)k - 125 + lane;
          if (col >= 1 && (uint32_t)col <= maxcol) bout[col] = outacc;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row segment is in L2 before the counter moves
        }
        if (lane == 0)
          __hip_atomic_store(prog + wave, pass * NC + (k >> 6) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (DIRS && (k & 15) == 15) {
        uint32_t* w = P.dirs + ((size_t)(strip * P.nkb + (k >> 4)) * RPL) * 64 + lane;
#pragma unroll
        for (int r = 0; r < RPL; r++) w[r * 64] = acc[r];
      }
    }
#pragma unroll
    for (int r = 0; r < RPL; r++) {
      const uint64_t key = order_key(ibase + r + 1, bestj[r], T, P.phaseQp, P.straight);
      if (bestv[r] > gbv || (bestv[r] == gbv && bestv[r] > 0 && key < gkey)) {
        gbv = bestv[r];
        gkey = key;
      }
    }
  }
  // wave reduction with the reference's tie order
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
  // ... then across the workgroup's waves
  if (lane == 0) {
    red[wave * 3 + 0] = (uint32_t)gbv;
    red[wave * 3 + 1] = (uint32_t)gkey;
    red[wave * 3 + 2] = (uint32_t)(gkey >> 32);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < NW; w++) {
      const int ov = (int)red[w * 3];
      const uint64_t okey = ((uint64_t)red[w * 3 + 2] << 32) | red[w * 3 + 1];
      if (ov > gbv || (ov == gbv && okey < gkey)) {
        gbv = ov;
        gkey = okey;
      }
    }
    uint32_t bx = 0, bj = 0;
    if (gbv > 0) {
      bx = (uint32_t)(gkey & 0x7fffffffu);
      const uint32_t hi = (uint32_t)((gkey >> 31) & 0xffffffffu);
      bj = P.straight ? hi : hi - bx;
    }
    out->score = gbv;
    out->bx = bx;
    out->bj = bj;
    out->nops = *err ? 0xffffffffu : 0;
  }
}

__global__ void __launch_bounds__(1024) sw_exact_kernel(const ExactPair* pairs, ExactOut* outs, uint32_t n,
                                                        int match, int miss, int gap) {
  __shared__ uint32_t prog[16];
  __shared__ uint32_t red[16 * 3];
  __shared__ uint32_t err;
  const uint32_t p = blockIdx.x;
  if (p >= n) return;
  if (threadIdx.x == 0) err = 0;
  const ExactPair P = pairs[p];
  if (P.dirs)
    exact_pair<true>(P, outs + p, match, miss, gap, prog, red, &err);
  else
    exact_pair<false>(P, outs + p, match, miss, gap, prog, red, &err);
}

// Traceback walker: one lane per pair follows the direction bits from the best cell until H reaches 0
// (rust/src/algorithm/mod.rs:663-664 `if score == 0 return`) or an edge (:650).  ops[] receives, last
// alignment column first: 0 = (q[x-1], t[j-1]), 1 = (q[x-1], '-'), 2 = ('-', t[j-1]).
__global__ void __launch_bounds__(64) sw_walk_kernel(const ExactPair* pairs, ExactOut* outs, uint32_t n,
                                                     int match, int miss, int gap) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const ExactPair P = pairs[p];
  if (!P.dirs || !P.ops) return;
  if (outs[p].nops == 0xffffffffu) return;  // the fill kernel reported a pipeline time-out
  int H = outs[p].score;
  uint32_t x = outs[p].bx, j = outs[p].bj;
  uint32_t nops = 0;
  const uint32_t cap = P.Q + P.T;
  while (H > 0 && x > 0 && j > 0 && nops < cap) {
    const uint32_t i = x - 1;
    const uint32_t strip = i / SH, within = i % SH, lane = within / RPL, r = within % RPL;
    const uint32_t k = (j - 1) + lane;
    const uint32_t w = P.dirs[((size_t)(strip * P.nkb + (k >> 4)) * RPL + r) * 64 + lane];
    const uint32_t code = (w >> (2 * (15 - (k & 15)))) & 3u;
    uint8_t op;
    if (code & 1u) {
      H -= (P.q[x - 1] == P.t[j - 1]) ? match : miss;
      x--;
      j--;
      op = 0;
    } else if (code & 2u) {
      H -= gap;
      x--;
      op = 1;
    } else {
      H -= gap;
      j--;
      op = 2;
    }
    P.ops[nops++] = op;
  }
  outs[p].nops = nops;
}

}  // namespace

hipError_t launch_exact(const ExactPair* d_pairs, ExactOut* d_outs, uint32_t n_pairs, int waves, Scoring sc,
                        hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  if (waves < 1) waves = 1;
  if (waves > 16) waves = 16;
  hipLaunchKernelGGL(sw_exact_kernel, dim3(n_pairs), dim3(64 * waves), 0, stream, d_pairs, d_outs, n_pairs,
                     sc.match, sc.miss, sc.gap);
  return hipGetLastError();
}

hipError_t launch_walk(const ExactPair* d_pairs, ExactOut* d_outs, uint32_t n_pairs, Scoring sc,
                       hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(sw_walk_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, stream, d_pairs, d_outs, n_pairs,
                     sc.match, sc.miss, sc.gap);
  return hipGetLastError();
}

}  // namespace swhip
