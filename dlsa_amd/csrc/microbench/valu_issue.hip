// valu_issue.hip -- cycles per wave-instruction per SIMD for the VALU operations the Smith-Waterman kernels are
// made of, at 1 / 2 / 4 / 8 waves per SIMD, 16 independent chains per wave (no dependency stall): settles which
// denominators the int32 and the packed-int16 rooflines may use.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue ; run on the GPU box.
//
// Method: every wave executes ITERS x 64 copies of ONE instruction (inline asm, 16 accumulators round-robin, so an
// instruction depends on the one 16 slots earlier) between two s_memtime stamps; all waves of a SIMD run the loop at
// the same time (one workgroup of 256 threads = one wave per SIMD; W workgroups per CU).  cycles per instruction per
// SIMD = median wave's stamp difference / (ITERS x 64 x W).  The wall-clock figure (HIP events, whole grid) is printed
// beside it as T lane-op/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define R4(X, a, b, c, d) X(a) X(b) X(c) X(d)
#define R16(X) R4(X, 0, 1, 2, 3) R4(X, 4, 5, 6, 7) R4(X, 8, 9, 10, 11) R4(X, 12, 13, 14, 15)
#define OPS16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
              "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

// one kernel per instruction form; T = register type (uint32_t, or uint64_t for the 64-bit packed-f32 form)
#define DEFINE_KERNEL(NAME, T, LINE)                                                                         \
  __global__ void __launch_bounds__(256) NAME(const T* in, T* out, int iters, unsigned long long* stamps) { \
    T a[16];                                                                                                 \
    for (int i = 0; i < 16; i++) a[i] = in[(threadIdx.x + 5 * i) & 1023];                                    \
    const T g = in[5], h = in[7];                                                                            \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
    for (int it = 0; it < iters; it++) {                                                                     \
      asm volatile(R16(LINE) R16(LINE) R16(LINE) R16(LINE) : OPS16 : "v"(g), "v"(h) : "vcc", "s10", "s11");                        \
    }                                                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
    T r = 0;                                                                                                 \
    for (int i = 0; i < 16; i++) r ^= a[i];                                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                          \
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                      \
  }

#define L_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
#define L_MAX(i) "v_max_i32 %" #i ", %" #i ", %16\n\t"
#define L_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %16, %17\n\t"
#define L_SUBC(i) "v_sub_u32 %" #i ", %" #i ", %16 clamp\n\t"
#define L_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n\t"
#define L_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 2, %17\n\t"
#define L_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n\t"
#define L_CMP(i) "v_cmp_gt_i32 vcc, %" #i ", %16\n\t"
#define L_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n\t"
#define L_BFI(i) "v_bfi_b32 %" #i ", %16, %17, %" #i "\n\t"
#define L_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %16\n\t"
#define L_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %16\n\t"
#define L_PKSUBC(i) "v_pk_sub_u16 %" #i ", %" #i ", %16 clamp\n\t"
#define L_PKMAD(i) "v_pk_mad_i16 %" #i ", %" #i ", %16, %17 clamp\n\t"
#define L_PKMADU(i) "v_pk_mad_u16 %" #i ", %" #i ", %16, %17\n\t"
#define L_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %16\n\t"
#define L_PKLSHL(i) "v_pk_lshlrev_b16 %" #i ", 2, %" #i "\n\t"
#define L_FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define L_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define L_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define L_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n\t"

#define L_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n\t"
#define L_OR(i) "v_or_b32 %" #i ", %" #i ", %16\n\t"
#define L_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define L_MOV(i) "v_mov_b32 %" #i ", %16\n\t"
#define L_SUB(i) "v_sub_u32 %" #i ", %" #i ", %16\n\t"
#define L_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n\t"
#define L_MINU(i) "v_min_u32 %" #i ", %" #i ", %16\n\t"
#define L_MAXU(i) "v_max_u32 %" #i ", %" #i ", %16\n\t"
#define L_MAXF(i) "v_max_f32 %" #i ", %" #i ", %16\n\t"
#define L_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n\t"
#define L_ADDF(i) "v_add_f32 %" #i ", %" #i ", %16\n\t"
#define L_MULF(i) "v_mul_f32 %" #i ", %" #i ", %16\n\t"
#define L_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n\t"
#define L_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n\t"
#define L_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %16, s[10:11]\n\t"
#define L_CMPS(i) "v_cmp_eq_u32 s[10:11], %" #i ", %16\n\t"
#define L_CMPF(i) "v_cmp_gt_f32 vcc, %" #i ", %16\n\t"
#define L_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 2, 9\n\t"
#define L_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n\t"
#define L_PKADDF(i) "v_pk_add_f32 %" #i ", %" #i ", %16\n\t"
#define L_PKMULF(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n\t"
#define L_MAXI16(i) "v_max_i16 %" #i ", %" #i ", %16\n\t"
#define L_ADDU16(i) "v_add_u16 %" #i ", %" #i ", %16\n\t"
#define L_MOVDPPR(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define L_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define L_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
#define L_MAXSDWA(i) "v_max_i16_sdwa %" #i ", %" #i ", %16 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"

#define L_CMPCND(i) "v_cmp_gt_i32 vcc, %" #i ", %16\n\tv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n\t"
#define L_CMPCNDS(i) "v_cmp_gt_i32 s[10:11], %" #i ", %16\n\tv_cndmask_b32 %" #i ", %" #i ", %17, s[10:11]\n\t"
#define L_CNDVCCADD(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n\tv_add_u32 %" #i ", %" #i ", %17\n\t"
#define L_MINU16(i) "v_min_u16 %" #i ", %" #i ", %16\n\t"
#define L_SUBU16C(i) "v_sub_u16 %" #i ", %" #i ", %16 clamp\n\t"
#define L_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %16, %" #i ", 2\n\t"
#define L_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n\t"
#define L_XAD(i) "v_xad_u32 %" #i ", %" #i ", %16, %17\n\t"
#define L_MAX3U(i) "v_max3_u32 %" #i ", %" #i ", %16, %17\n\t"
DEFINE_KERNEL(k_cmpcnd_vcc, uint32_t, L_CMPCND)
DEFINE_KERNEL(k_cmpcnd_sgpr, uint32_t, L_CMPCNDS)
DEFINE_KERNEL(k_cndvcc_add, uint32_t, L_CNDVCCADD)
DEFINE_KERNEL(k_min_u16, uint32_t, L_MINU16)
DEFINE_KERNEL(k_sub_u16_clamp, uint32_t, L_SUBU16C)
DEFINE_KERNEL(k_alignbit, uint32_t, L_ALIGNBIT)
DEFINE_KERNEL(k_lshl_add, uint32_t, L_LSHLADD)
DEFINE_KERNEL(k_xad, uint32_t, L_XAD)
DEFINE_KERNEL(k_max3_u32, uint32_t, L_MAX3U)
DEFINE_KERNEL(k_add_u32, uint32_t, L_ADD)
DEFINE_KERNEL(k_max_i32, uint32_t, L_MAX)
DEFINE_KERNEL(k_max3_i32, uint32_t, L_MAX3)
DEFINE_KERNEL(k_sub_u32_clamp, uint32_t, L_SUBC)
DEFINE_KERNEL(k_and_or_b32, uint32_t, L_ANDOR)
DEFINE_KERNEL(k_lshl_or_b32, uint32_t, L_LSHLOR)
DEFINE_KERNEL(k_cndmask_b32, uint32_t, L_CNDMASK)
DEFINE_KERNEL(k_cmp_gt_i32, uint32_t, L_CMP)
DEFINE_KERNEL(k_xor_b32, uint32_t, L_XOR)
DEFINE_KERNEL(k_bfi_b32, uint32_t, L_BFI)
DEFINE_KERNEL(k_pk_max_i16, uint32_t, L_PKMAX)
DEFINE_KERNEL(k_pk_add_u16, uint32_t, L_PKADD)
DEFINE_KERNEL(k_pk_sub_u16_clamp, uint32_t, L_PKSUBC)
DEFINE_KERNEL(k_pk_mad_i16_clamp, uint32_t, L_PKMAD)
DEFINE_KERNEL(k_pk_mad_u16, uint32_t, L_PKMADU)
DEFINE_KERNEL(k_pk_min_u16, uint32_t, L_PKMIN)
DEFINE_KERNEL(k_pk_lshlrev_b16, uint32_t, L_PKLSHL)
DEFINE_KERNEL(k_fma_f32, uint32_t, L_FMA)
DEFINE_KERNEL(k_pk_fma_f32, uint64_t, L_PKFMA)
DEFINE_KERNEL(k_mov_dpp_wave_shr, uint32_t, L_MOVDPP)
DEFINE_KERNEL(k_perm_b32, uint32_t, L_PERM)

DEFINE_KERNEL(k_and_b32, uint32_t, L_AND)
DEFINE_KERNEL(k_or_b32, uint32_t, L_OR)
DEFINE_KERNEL(k_lshlrev_b32, uint32_t, L_LSHL)
DEFINE_KERNEL(k_mov_b32, uint32_t, L_MOV)
DEFINE_KERNEL(k_sub_u32, uint32_t, L_SUB)
DEFINE_KERNEL(k_add3_u32, uint32_t, L_ADD3)
DEFINE_KERNEL(k_min_u32, uint32_t, L_MINU)
DEFINE_KERNEL(k_max_u32, uint32_t, L_MAXU)
DEFINE_KERNEL(k_max_f32, uint32_t, L_MAXF)
DEFINE_KERNEL(k_max3_f32, uint32_t, L_MAX3F)
DEFINE_KERNEL(k_add_f32, uint32_t, L_ADDF)
DEFINE_KERNEL(k_mul_f32, uint32_t, L_MULF)
DEFINE_KERNEL(k_mad_u32_u24, uint32_t, L_MAD24)
DEFINE_KERNEL(k_mul_u32_u24, uint32_t, L_MUL24)
DEFINE_KERNEL(k_cndmask_sgpr, uint32_t, L_CNDS)
DEFINE_KERNEL(k_cmp_eq_sgpr, uint32_t, L_CMPS)
DEFINE_KERNEL(k_cmp_gt_f32, uint32_t, L_CMPF)
DEFINE_KERNEL(k_bfe_u32, uint32_t, L_BFE)
DEFINE_KERNEL(k_add_co_u32, uint32_t, L_ADDCO)
DEFINE_KERNEL(k_max_i16, uint32_t, L_MAXI16)
DEFINE_KERNEL(k_add_u16, uint32_t, L_ADDU16)
DEFINE_KERNEL(k_mov_dpp_row_shr, uint32_t, L_MOVDPPR)
DEFINE_KERNEL(k_add_dpp, uint32_t, L_ADDDPP)
DEFINE_KERNEL(k_add_sdwa, uint32_t, L_ADDSDWA)
DEFINE_KERNEL(k_max_i16_sdwa, uint32_t, L_MAXSDWA)
DEFINE_KERNEL(k_pk_add_f32, uint64_t, L_PKADDF)
DEFINE_KERNEL(k_pk_mul_f32, uint64_t, L_PKMULF)

struct Case {
  const char* name;
  int regs64;  // 1: 64-bit registers
  void (*k32)(const uint32_t*, uint32_t*, int, unsigned long long*);
  void (*k64)(const uint64_t*, uint64_t*, int, unsigned long long*);
  int lane_ops;  // values per lane a line produces (2 for packed forms and for two-instruction lines: the rate then counts instructions)
};

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  printf("device %s CUs %d nominal clock %d kHz; %d x 64 instructions per wave, 16 independent chains\n", p.name, cus,
         p.clockRate, iters);
  void *in, *out;
  unsigned long long* stamps;
  CK(hipMalloc(&in, 1024 * 8));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8));
  CK(hipMalloc(&stamps, (size_t)cus * 8 * 4 * 8));
  std::vector<uint64_t> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = ((uint64_t)(i * 2654435761u) & 0x00ff00ffu) | ((uint64_t)0x3f800000u << 32);
  CK(hipMemcpy(in, h.data(), 1024 * 8, hipMemcpyHostToDevice));
  const Case cases[] = {
      {"v_add_u32", 0, k_add_u32, nullptr, 1},
      {"v_max_i32", 0, k_max_i32, nullptr, 1},
      {"v_max3_i32", 0, k_max3_i32, nullptr, 1},
      {"v_sub_u32 clamp", 0, k_sub_u32_clamp, nullptr, 1},
      {"v_and_or_b32", 0, k_and_or_b32, nullptr, 1},
      {"v_lshl_or_b32", 0, k_lshl_or_b32, nullptr, 1},
      {"v_cndmask_b32 (vcc)", 0, k_cndmask_b32, nullptr, 1},
      {"v_cmp_gt_i32 -> vcc", 0, k_cmp_gt_i32, nullptr, 1},
      {"v_xor_b32", 0, k_xor_b32, nullptr, 1},
      {"v_bfi_b32", 0, k_bfi_b32, nullptr, 1},
      {"v_mov_b32 dpp wave_shr:1", 0, k_mov_dpp_wave_shr, nullptr, 1},
      {"v_perm_b32", 0, k_perm_b32, nullptr, 1},
      {"v_pk_max_i16", 0, k_pk_max_i16, nullptr, 2},
      {"v_pk_add_u16", 0, k_pk_add_u16, nullptr, 2},
      {"v_pk_sub_u16 clamp", 0, k_pk_sub_u16_clamp, nullptr, 2},
      {"v_pk_mad_i16 clamp", 0, k_pk_mad_i16_clamp, nullptr, 2},
      {"v_pk_mad_u16", 0, k_pk_mad_u16, nullptr, 2},
      {"v_pk_min_u16", 0, k_pk_min_u16, nullptr, 2},
      {"v_pk_lshlrev_b16", 0, k_pk_lshlrev_b16, nullptr, 2},
      {"v_fma_f32", 0, k_fma_f32, nullptr, 1},
      {"v_pk_fma_f32", 1, nullptr, k_pk_fma_f32, 2},
      {"v_and_b32", 0, k_and_b32, nullptr, 1},
      {"v_or_b32", 0, k_or_b32, nullptr, 1},
      {"v_lshlrev_b32", 0, k_lshlrev_b32, nullptr, 1},
      {"v_mov_b32", 0, k_mov_b32, nullptr, 1},
      {"v_sub_u32", 0, k_sub_u32, nullptr, 1},
      {"v_add3_u32", 0, k_add3_u32, nullptr, 1},
      {"v_min_u32", 0, k_min_u32, nullptr, 1},
      {"v_max_u32", 0, k_max_u32, nullptr, 1},
      {"v_max_f32", 0, k_max_f32, nullptr, 1},
      {"v_max3_f32", 0, k_max3_f32, nullptr, 1},
      {"v_add_f32", 0, k_add_f32, nullptr, 1},
      {"v_mul_f32", 0, k_mul_f32, nullptr, 1},
      {"v_mad_u32_u24", 0, k_mad_u32_u24, nullptr, 1},
      {"v_mul_u32_u24", 0, k_mul_u32_u24, nullptr, 1},
      {"v_cndmask_b32 (sgpr pair)", 0, k_cndmask_sgpr, nullptr, 1},
      {"v_cmp_eq_u32 -> sgpr pair", 0, k_cmp_eq_sgpr, nullptr, 1},
      {"v_cmp_gt_f32 -> vcc", 0, k_cmp_gt_f32, nullptr, 1},
      {"v_bfe_u32", 0, k_bfe_u32, nullptr, 1},
      {"v_add_co_u32", 0, k_add_co_u32, nullptr, 1},
      {"v_max_i16", 0, k_max_i16, nullptr, 1},
      {"v_add_u16", 0, k_add_u16, nullptr, 1},
      {"v_mov_b32 dpp row_shr:1", 0, k_mov_dpp_row_shr, nullptr, 1},
      {"v_add_u32 dpp row_shr:1", 0, k_add_dpp, nullptr, 1},
      {"v_add_u32 sdwa", 0, k_add_sdwa, nullptr, 1},
      {"v_max_i16 sdwa", 0, k_max_i16_sdwa, nullptr, 1},
      {"cmp->vcc + cndmask vcc (x2)", 0, k_cmpcnd_vcc, nullptr, 2},
      {"cmp->sgpr + cndmask sgpr (x2)", 0, k_cmpcnd_sgpr, nullptr, 2},
      {"cndmask vcc + v_add_u32 (x2)", 0, k_cndvcc_add, nullptr, 2},
      {"v_min_u16", 0, k_min_u16, nullptr, 1},
      {"v_sub_u16 clamp", 0, k_sub_u16_clamp, nullptr, 1},
      {"v_alignbit_b32", 0, k_alignbit, nullptr, 1},
      {"v_lshl_add_u32", 0, k_lshl_add, nullptr, 1},
      {"v_xad_u32", 0, k_xad, nullptr, 1},
      {"v_max3_u32", 0, k_max3_u32, nullptr, 1},
      {"v_pk_add_f32", 1, nullptr, k_pk_add_f32, 2},
      {"v_pk_mul_f32", 1, nullptr, k_pk_mul_f32, 2},
  };
  // warm the clock up
  for (int rep = 0; rep < 50; rep++) k_add_u32<<<cus * 8, 256>>>((const uint32_t*)in, (uint32_t*)out, iters, stamps);
  CK(hipDeviceSynchronize());
  printf("%-26s", "instruction");
  for (int w = 1; w <= 8; w *= 2) printf("  w/SIMD=%d: cyc/instr  T lane-op/s", w);
  printf("\n");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Case& c : cases) {
    printf("%-26s", c.name);
    for (int w = 1; w <= 8; w *= 2) {
      const int blocks = cus * w;
      auto launch = [&] {
        if (c.regs64) c.k64<<<blocks, 256>>>((const uint64_t*)in, (uint64_t*)out, iters, stamps);
        else c.k32<<<blocks, 256>>>((const uint32_t*)in, (uint32_t*)out, iters, stamps);
      };
      launch();
      CK(hipDeviceSynchronize());
      const int reps = 3;
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; r++) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      std::vector<unsigned long long> hs((size_t)blocks * 4);
      CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
      std::sort(hs.begin(), hs.end());
      const double cyc = (double)hs[hs.size() / 2] / ((double)iters * 64 * w);
      const double tops = (double)blocks * 256 * iters * 64 * c.lane_ops / (ms * 1e-3) / 1e12;
      printf("  %18.2f  %13.2f", cyc, tops);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
