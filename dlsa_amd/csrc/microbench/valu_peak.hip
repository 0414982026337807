// valu_peak.hip -- issue-rate microbenchmark for the packed-int16 VALU roofline (SURVEY.md 8d:
// "confirm ... with a v_pk_max_i16 issue-rate microbenchmark").  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// NCH independent chains of {v_pk_max_i16, v_pk_add_i16}: 2*NCH packed ops per iteration
template <int NCH>
__global__ void __launch_bounds__(256) pk_chain(const int* in, int* out, int iters) {
  s2 a[NCH], b[NCH];
  for (int i = 0; i < NCH; i++) { a[i] = __builtin_bit_cast(s2, in[(threadIdx.x + i) & 1023]); b[i] = __builtin_bit_cast(s2, in[(threadIdx.x + 7 * i + 3) & 1023]); }
  s2 g = __builtin_bit_cast(s2, in[5]);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      a[i] = __builtin_elementwise_max(a[i], b[i]);
      a[i] = a[i] + g;
    }
  }
  int r = 0;
  for (int i = 0; i < NCH; i++) r ^= __builtin_bit_cast(int, a[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// same with 32-bit ops
template <int NCH>
__global__ void __launch_bounds__(256) i32_chain(const int* in, int* out, int iters) {
  int a[NCH], b[NCH];
  for (int i = 0; i < NCH; i++) { a[i] = in[(threadIdx.x + i) & 1023]; b[i] = in[(threadIdx.x + 7 * i + 3) & 1023]; }
  int g = in[5];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      a[i] = max(a[i], b[i]);
      a[i] = a[i] + g;
    }
  }
  int r = 0;
  for (int i = 0; i < NCH; i++) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// the SW cell recurrence as the score kernel runs it: 16 columns, dependent along the row
__global__ void __launch_bounds__(256) sw_cells(const int* in, int* out, int iters) {
  constexpr int R = 16;
  s2 H[R], tab[R];
  for (int i = 0; i < R; i++) { H[i] = __builtin_bit_cast(s2, in[(threadIdx.x + i) & 1023] & 0x00ff00ff); tab[i] = __builtin_bit_cast(s2, in[(threadIdx.x * 3 + i) & 1023] & 0x00010001); }
  s2 gap = {1, 1};
  s2 best = {0, 0}, diag = {0, 0}, left = {0, 0};
  for (int x = 0; x < iters; x++) {
    s2 hl = left, hd = diag;
#pragma unroll
    for (int r = 0; r < R; r++) {
      s2 hu = H[r];
      s2 m = __builtin_elementwise_max(hl, hu);
      if (r & 1) best = __builtin_elementwise_max(best, m);
      s2 t = (s2)__builtin_elementwise_sub_sat((u2)m, (u2)gap);
      s2 d = __builtin_elementwise_add_sat(hd, tab[r]);
      s2 h = __builtin_elementwise_max(t, d);
      hd = hu; hl = h; H[r] = h;
    }
    left = H[3]; diag = H[7];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(int, best) ^ __builtin_bit_cast(int, H[R - 1]);
}

// The shader clock while a VALU-bound kernel runs: one wave per workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (a constant 100 MHz) around the same chain, so clock = d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, profiling recipe step 6).  The device property `clockRate` is only the nominal 2.4 GHz.
__global__ void __launch_bounds__(256) i32_chain_clock(const int* in, int* out, int iters, unsigned long long* stamps) {
  int a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = in[(threadIdx.x + i) & 1023]; b[i] = in[(threadIdx.x + 7 * i + 3) & 1023]; }
  int g = in[5];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a[i] = max(a[i], b[i]);
      a[i] = a[i] + g;
    }
  }
  int r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  // (r feeds the stamps' address so that the loop cannot sink below them)
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x + (r & 0)] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main(int argc, char** argv) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz wave %d\n", p.name, p.multiProcessorCount, p.clockRate, p.warpSize);
  int *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 4096 * 4));
  std::vector<int> h(1024); for (int i = 0; i < 1024; i++) h[i] = (i * 2654435761u) & 0x7fff7fff; h[5] = 0x00010001;
  CK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int cus = p.multiProcessorCount;
  double ghz = p.clockRate / 1e6;
  {  // measured shader clock under an all-SIMDs VALU load (after ~1 s of it, so that the clock has settled)
    const int blocks = cus * 8;
    unsigned long long* stamps; CK(hipMalloc(&stamps, blocks * 16));
    for (int rep = 0; rep < 200; rep++) i32_chain_clock<<<blocks, 256>>>(in, out, iters, stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(2 * blocks); CK(hipMemcpy(hs.data(), stamps, blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> f; for (int i = 0; i < blocks; i++) if (hs[2 * i + 1]) f.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 0.1);
    std::sort(f.begin(), f.end());
    ghz = f[f.size() / 2];
    printf("measured shader clock under VALU load: %.3f GHz (median of %zu workgroups; nominal %.3f)\n", ghz, f.size(), p.clockRate / 1e6);
    printf("issue peak at the measured clock: %.2f T i32-op/s, %.2f T packed i16-op/s (256 CUs x 64 lanes/clk)\n",
           cus * 64 * ghz / 1e3, 2 * cus * 64 * ghz / 1e3);
  }
  for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    int blocks = cus * wps;
    double ms = time_ms([&] { pk_chain<8><<<blocks, 256>>>(in, out, iters); }, 3);
    double ops = (double)blocks * 256 * iters * 16 * 2;  // 16 packed instr per iter per lane, 2 i16 ops each
    printf("pk_chain<8>   waves/SIMD %d: %.3f ms  %.2f T i16-op/s  (%.2f T lane-instr/s)\n", wps, ms, ops / ms / 1e9, ops / 2 / ms / 1e9);
    ms = time_ms([&] { pk_chain<2><<<blocks, 256>>>(in, out, iters); }, 3);
    ops = (double)blocks * 256 * iters * 4 * 2;
    printf("pk_chain<2>   waves/SIMD %d: %.3f ms  %.2f T i16-op/s\n", wps, ms, ops / ms / 1e9);
    ms = time_ms([&] { i32_chain<8><<<blocks, 256>>>(in, out, iters); }, 3);
    ops = (double)blocks * 256 * iters * 16;
    printf("i32_chain<8>  waves/SIMD %d: %.3f ms  %.2f T i32-op/s  (%.3f of the peak at the measured clock)\n", wps, ms, ops / ms / 1e9,
           ops / ms / 1e9 / (cus * 64 * ghz / 1e3));
    ms = time_ms([&] { sw_cells<<<blocks, 256>>>(in, out, iters / 4); }, 3);
    double cells = (double)blocks * 256 * (iters / 4) * 16 * 2;
    printf("sw_cells R=16 waves/SIMD %d: %.3f ms  %.2f TCUPS  (4.5 packed instr / 2 cells)\n", wps, ms, cells / ms / 1e9);
  }
  return 0;
}
