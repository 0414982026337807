// dlsa_amd/csrc/sw_topk.hip -- per-GPU top-k selection over the score kernel's output, on the device.
//
// What it replaces: cli/__main__.py:153-173 sorts every pair's score client-side after all results have
// travelled worker -> master -> CLI.  Here only the candidates for the k best targets (plus the few targets the
// int32 exact kernel has to re-score) leave the GPU: for a database of 10^7 reads that is a few hundred bytes
// instead of 40 MB of scores and a host pass over them.
//
// Three small kernels over the per-slot scores (all HBM-bound integer work, one coalesced pass each):
//   1. sw_hist_kernel    -- classifies every slot (padding / needs an exact re-run / plain) with the same rules
//                           as the host path (sw_lib.cpp), appends re-run slots to a list and
//                           histograms the plain scores (0..32767) in LDS, one flush per workgroup;
//   2. sw_thresh_kernel  -- finds the score s* with  count(score > s*) < k <= count(score >= s*);
//   3. sw_compact_kernel -- appends (score, target index) of the slots above s* to the candidate list and of
//                           the slots at s* to the tie list (bounded; overflow is reported and the host falls
//                           back to the full download, so the result is exact in every case).
// The host orders the candidates by (score desc, target index asc) -- the library's declared tie rule.
#include "sw_internal.h"

namespace swhip {

namespace {

constexpr uint32_t kBins = 32768;

__device__ __forceinline__ uint32_t roundup64(uint32_t v) { return (v + 63u) & ~63u; }

// 0 = padding slot, 1 = plain (the packed score is final), 2 = must be re-scored by the exact kernel
__device__ __forceinline__ int classify(uint32_t T, int32_t s, uint32_t Q, uint32_t Qp, int match) {
  if (T == 0) return 0;
  if (s >= 32767) return 2;  // int16 saturation
  const bool full = (uint64_t)T < 2ull * Q;  // rust/src/algorithm/mod.rs:213
  if (!full && T < Qp) return 2;             // ring path with T < Qp (two-phase discovery, H(1,T) skipped)
  if (full && (T & 63u) == 0 && roundup64(T) >= Qp && s < match) return 2;  // only H(1,T) could still score
  return 1;
}

__global__ void __launch_bounds__(1024) sw_hist_kernel(SelectParams p) {
  __shared__ uint32_t h[kBins];
  for (uint32_t i = threadIdx.x; i < kBins; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const uint32_t Qp = roundup64(p.Q);
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < p.n_slots; s += gridDim.x * blockDim.x) {
    const int32_t v = p.slot_scores[s];
    const int c = classify(p.lens[s], v, p.Q, Qp, p.match);
    if (c == 1) {
      atomicAdd(&h[v < 0 ? 0 : v], 1u);
    } else if (c == 2) {
      const uint32_t at = atomicAdd(p.counters + kSelRerun, 1u);
      if (at < p.rerun_cap) p.rerun[at] = s;
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kBins; i += blockDim.x)
    if (h[i]) atomicAdd(p.hist + i, h[i]);
}

// one workgroup of 1024 threads; thread t owns bins [32 t, 32 t + 32).  Suffix sums over the 1024 partial counts
// (log-step scan in LDS) locate the one thread whose bins contain the k-th best score.
__global__ void __launch_bounds__(1024) sw_thresh_kernel(SelectParams p) {
  __shared__ uint32_t part[2][1024];
  const uint32_t t = threadIdx.x;
  uint32_t bins[32];
  uint32_t mine = 0;
  for (uint32_t i = 0; i < 32; i++) {
    bins[i] = p.hist[t * 32 + i];
    mine += bins[i];
  }
  part[0][t] = mine;
  __syncthreads();
  int cur = 0;
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // part[cur][t] = sum of mine over threads t .. min(t + 2d - 1, 1023)
    part[cur ^ 1][t] = part[cur][t] + (t + d < 1024 ? part[cur][t + d] : 0u);
    cur ^= 1;
    __syncthreads();
  }
  const uint32_t incl = part[cur][t];  // slots with a score in my bins or above
  uint32_t above = incl - mine;        // ... strictly above my bins
  if (above < p.k && p.k <= incl) {
    for (int i = 31; i >= 0; i--) {
      if (above + bins[i] >= p.k) {
        p.counters[kSelThresh] = t * 32 + (uint32_t)i;
        p.counters[kSelAbove] = above;
        break;
      }
      above += bins[i];
    }
  }
  if (t == 0 && incl < p.k) {  // fewer than k plain slots: everything is a candidate
    p.counters[kSelThresh] = 0;
    p.counters[kSelAbove] = incl - bins[0];
  }
}

__global__ void __launch_bounds__(256) sw_compact_kernel(SelectParams p) {
  const uint32_t thr = p.counters[kSelThresh];
  const uint32_t Qp = roundup64(p.Q);
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < p.n_slots; s += gridDim.x * blockDim.x) {
    const int32_t v0 = p.slot_scores[s];
    if (classify(p.lens[s], v0, p.Q, Qp, p.match) != 1) continue;
    const uint32_t v = v0 < 0 ? 0u : (uint32_t)v0;
    if (v < thr) continue;
    const uint32_t ti = p.slot_target[s];
    if (v > thr) {
      const uint32_t at = atomicAdd(p.counters + kSelCand, 1u);
      if (at < p.cand_cap) {
        p.cand[2 * at] = v;
        p.cand[2 * at + 1] = ti;
      }
    } else {
      const uint32_t at = atomicAdd(p.counters + kSelTie, 1u);
      if (at < p.tie_cap) {
        p.tie[2 * at] = v;
        p.tie[2 * at + 1] = ti;
      }
    }
  }
}

}  // namespace

hipError_t launch_select(const SelectParams& p, hipStream_t stream) {
  if (p.n_slots == 0) return hipSuccess;
  uint32_t nwg = (p.n_slots + 16383) / 16384;
  if (nwg > 256) nwg = 256;
  hipLaunchKernelGGL(sw_hist_kernel, dim3(nwg), dim3(1024), 0, stream, p);
  hipLaunchKernelGGL(sw_thresh_kernel, dim3(1), dim3(1024), 0, stream, p);
  uint32_t nwg2 = (p.n_slots + 1023) / 1024;
  if (nwg2 > 2048) nwg2 = 2048;
  hipLaunchKernelGGL(sw_compact_kernel, dim3(nwg2), dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace swhip
