// dep_latency.hip -- latency of DEPENDENT VALU instructions on a lone wave (one wave per CU, as the strips of the exact
// kernel run): cycles (s_memtime) and nanoseconds (s_memrealtime, 100 MHz) per instruction of a chain in which every
// instruction reads the result of the one before it.  Tells whether the exact kernel's step (DPP wave shift -> saturating
// subtract -> max3 -> and) is bound by the instructions' latency or by their number.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 dep_latency.hip -o dep_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define R8(S) S S S S S S S S
#define R64(S) R8(R8(S))

#define DEFINE_KERNEL(NAME, BODY, PER)                                                                        \
  __global__ void __launch_bounds__(64) NAME(const uint32_t* in, uint32_t* out, int iters, unsigned long long* stamps) { \
    uint32_t a = in[threadIdx.x], b = in[64 + threadIdx.x];                                                   \
    const uint32_t g = in[5], h = in[7];                                                                      \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                           \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < iters; it++) {                                                                      \
      asm volatile(R64(BODY) : "+v"(a), "+v"(b) : "v"(g), "v"(h) : "vcc", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24");                                    \
    }                                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                           \
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b;                                                               \
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }         \
  }                                                                                                           \
  static const int NAME##_per = PER;

DEFINE_KERNEL(k_add, "v_add_u32 %0, %0, %2\n\t", 1)
DEFINE_KERNEL(k_and, "v_and_b32 %0, %0, %2\n\t", 1)
DEFINE_KERNEL(k_subc, "v_sub_u32 %0, %0, %2 clamp\n\t", 1)
DEFINE_KERNEL(k_max3, "v_max3_i32 %0, %0, %2, %3\n\t", 1)
DEFINE_KERNEL(k_dpp_wave, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t", 1)
DEFINE_KERNEL(k_dpp_row, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t", 1)
DEFINE_KERNEL(k_dpp_ror, "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t", 1)
// the exact kernel's chain of one row: shift in the cell above, subtract the gap, maximum of three, clear the tag
DEFINE_KERNEL(k_step, "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_sub_u32 %1, %1, %2 clamp\n\tv_max3_i32 %0, %3, %2, %1\n\tv_and_b32 %0, -4, %0\n\t", 4)
// ... the same with 13 independent instructions between the links, as in the real step of 17
DEFINE_KERNEL(k_step_fill, "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_sub_u32 %1, %1, %2 clamp\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_max3_i32 %0, %3, %2, %1\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_and_b32 %0, -4, %0\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\t", 17)

// compare -> select through VCC, both on the same value
DEFINE_KERNEL(k_cmpcnd, "v_cmp_gt_i32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\t", 2)
// ... the select on another value: only VCC links the two
DEFINE_KERNEL(k_cmpcnd_vcc, "v_cmp_gt_i32 vcc, %0, %2\n\tv_cndmask_b32 %1, %1, %3, vcc\n\t", 2)
// one step of the exact kernel at one row per lane as the compiler emits it (17 instructions, registers renamed)
DEFINE_KERNEL(k_realstep,
  "v_and_b32 v12, -4, %0\n\t"
  "v_mov_b32_dpp v16, %3 row_ror:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  "v_alignbit_b32 v19, %0, v19, 2\n\t"
  "v_add_u32 v20, 5, %2\n\t"
  "v_cmp_gt_i32 vcc, v12, v17\n\t"
  "v_mov_b32_dpp v16, v15 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
  "v_mov_b32_dpp v14, %2 row_ror:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  "v_cndmask_b32 v18, v18, v20, vcc\n\t"
  "v_cmp_eq_u32 vcc, %1, v16\n\t"
  "v_mov_b32_dpp v14, v12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
  "v_max_i32 v17, v17, v12\n\t"
  "v_cndmask_b32 v15, %2, %3, vcc\n\t"
  "v_or_b32 v21, 0x80000000, v12\n\t"
  "v_add_u32 v13, v15, v13\n\t"
  "v_sub_u32 v12, v12, %2 clamp\n\t"
  "v_sub_u32 v22, v14, %3 clamp\n\t"
  "v_max3_i32 %0, v13, v12, v22\n\t", 17)
// ... the same without the two compare/select pairs (4 instructions replaced by plain ones)
DEFINE_KERNEL(k_realstep_novcc,
  "v_and_b32 v12, -4, %0\n\t"
  "v_mov_b32_dpp v16, %3 row_ror:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  "v_alignbit_b32 v19, %0, v19, 2\n\t"
  "v_add_u32 v20, 5, %2\n\t"
  "v_max_i32 v23, v12, v17\n\t"
  "v_mov_b32_dpp v16, v15 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
  "v_mov_b32_dpp v14, %2 row_ror:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  "v_add_u32 v18, v18, v20\n\t"
  "v_xor_b32 v24, %1, v16\n\t"
  "v_mov_b32_dpp v14, v12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
  "v_max_i32 v17, v17, v12\n\t"
  "v_min_u32 v15, v24, %3\n\t"
  "v_or_b32 v21, 0x80000000, v12\n\t"
  "v_add_u32 v13, v15, v13\n\t"
  "v_sub_u32 v12, v12, %2 clamp\n\t"
  "v_sub_u32 v22, v14, %3 clamp\n\t"
  "v_max3_i32 %0, v13, v12, v22\n\t", 17)

typedef void (*kern_t)(const uint32_t*, uint32_t*, int, unsigned long long*);
struct Case { const char* name; kern_t k; int per; };

int main(int argc, char** argv) {
  const int iters = 2000; const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  uint32_t *in, *out;
  unsigned long long* stamps;
  CK(hipMalloc(&in, 4096));
  CK(hipMalloc(&out, blocks * 64 * 4));
  CK(hipMalloc(&stamps, blocks * 16));
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = 1000 + 7 * i;
  CK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
  const Case cases[] = {{"v_add_u32", k_add, 1}, {"v_and_b32", k_and, 1}, {"v_sub_u32 clamp", k_subc, 1}, {"v_max3_i32", k_max3, 1},
                        {"v_mov_dpp wave_shr:1", k_dpp_wave, 1}, {"v_mov_dpp row_shr:1", k_dpp_row, 1}, {"v_mov_dpp row_ror:1", k_dpp_ror, 1},
                        {"step chain (dpp, sub, max3, and)", k_step, 4}, {"step of 17 (chain + 13 independent adds)", k_step_fill, 17},
                        {"v_cmp -> v_cndmask (same value)", k_cmpcnd, 2}, {"v_cmp -> v_cndmask (VCC only)", k_cmpcnd_vcc, 2},
                        {"the kernel's step at 1 row per lane (17)", k_realstep, 17}, {"... without the compare/select pairs", k_realstep_novcc, 17}};
  printf("one wave per CU (%d workgroups of 64), every instruction reads the result of the one before\n", blocks);
  printf("%-44s %10s %10s %8s\n", "chain", "cycles/ins", "ns/ins", "GHz");
  for (const Case& c : cases) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, in, out, iters, stamps);
      CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> s(blocks * 2);
    CK(hipMemcpy(s.data(), stamps, blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ns;
    for (int b = 0; b < blocks; b++) { cyc.push_back((double)s[2 * b]); ns.push_back((double)s[2 * b + 1] * 10.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
    const double n_ins = (double)iters * 64 * c.per;
    printf("%-44s %10.2f %10.2f %8.2f\n", c.name, cyc[blocks / 2] / n_ins, ns[blocks / 2] / n_ins, cyc[blocks / 2] / ns[blocks / 2]);
  }
  return 0;
}
