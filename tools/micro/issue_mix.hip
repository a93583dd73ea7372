// What a SIMD of gfx950 can issue beside its vector instructions: a loop of independent v_mul_f64
// (eight per block) with other instruction classes mixed in -- scalar ALU, branches (taken / not),
// a vector compare whose mask a scalar instruction reads and a select uses, LDS adds -- at one to
// four waves per SIMD.  Printed: SIMD cycles per block of 8 v_mul_f64 (32 = the vector unit alone).
// The stream kernel's facet trip is 55 vector + 31 scalar instructions with six branches and one
// ds_add_f64; this is the experiment that says what of that the vector instructions wait for.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_mix.hip -o tools/micro/build/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 8192;
constexpr int kUnroll = 8;

#define VALU8                                                  \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(m));    \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(m));

/* the same eight, each followed by the extra (so that the classes alternate as in real code) */
#define VALU8_WITH(EXTRA)                                             \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(m)); EXTRA

#define VALU8_WITH_EVERY_OTHER(EXTRA)                                 \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(m));

#define VALU8_WITH_ONE(EXTRA)                                         \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(m)); EXTRA     \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(m));           \
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(m));

#define SALU asm volatile("s_and_b64 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
#define SALU_CHAIN asm volatile("s_xor_b64 %0, %0, %1\n s_and_b64 %0, %0, %1\n s_or_b64 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
#define BR_UNTAKEN asm volatile("s_cmp_eq_u32 %0, 0x12345\n s_cbranch_scc1 L_%=\n L_%=:" : : "s"(s2) : "scc");
#define BR_TAKEN asm volatile("s_branch L_%=\n s_nop 0\n L_%=:" : : : "scc");
/* vector compare -> mask in scalar registers -> scalar instruction -> select that uses the mask */
#define CMP_SALU_SELECT                                                                      \
  asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(mk) : "v"(a0), "v"(a4));                 \
  asm volatile("s_and_b64 %0, %0, %1" : "+s"(mk) : "s"(s1) : "scc");                                 \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(mk));
#define CMP_ONLY asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(mk) : "v"(a0), "v"(a4));
/* the same select without the scalar instruction in between */
#define CMP_SELECT                                                                           \
  asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(mk) : "v"(a0), "v"(a4));                 \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(mk));
/* ... through vcc, as VOP2 */
#define CMP_SELECT_VCC                                                                       \
  asm volatile("v_cmp_lt_f64_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(x) : "v"(a0), "v"(a4), "v"(y) : "vcc");
/* a wave-uniform branch on "any lane set" of a mask: by scc, and the way the compiler lowers a
 * uniform i1 that has other uses (s_cselect + s_and vcc, exec + s_cbranch_vccnz) */
#define BR_MASK_SCC asm volatile("s_cmp_lg_u64 %0, 0\n s_cbranch_scc0 L_%=\n L_%=:" : : "s"(s1) : "scc");
#define BR_MASK_VCC asm volatile("s_cmp_lg_u64 %1, 0\n s_cselect_b64 %0, -1, 0\n s_and_b64 vcc, exec, %0\n s_cbranch_vccz L_%=\n L_%=:" : "=s"(mk) : "s"(s1) : "scc", "vcc");
/* exec narrowed and restored around one vector instruction */
#define EXEC_TOGGLE asm volatile("s_and_saveexec_b64 %0, %1\n v_add_u32_e32 %2, 1, %2\n s_mov_b64 exec, %0" : "=s"(mk), "+s"(s1), "+v"(x) : : "scc");
#define LDS_ADD asm volatile("ds_add_f64 %0, %1" : : "v"(lds_addr), "v"(a7) : "memory");
/* a quarter-rate instruction whose result nothing in the block waits for: does it issue beside the others? */
#define RCP asm volatile("v_rcp_f64 %0, %1" : "=v"(rc) : "v"(a7));
#define RSQ asm volatile("v_rsq_f64 %0, %1" : "=v"(rc) : "v"(a7));
#define SNOP asm volatile("s_nop 0");
#define WAITCNT asm volatile("s_waitcnt lgkmcnt(15)");

#define KERNEL(NAME, BODY)                                                                          \
  __global__ __launch_bounds__(1024) void NAME(double* out, double seed, unsigned long long k1,     \
                                               unsigned k2) {                                       \
    extern __shared__ double lds[];                                                                 \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,             \
           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                                   \
    const double m = 1.0000001;                                                                     \
    unsigned long long s0 = k1 | 3ull, s1 = k1 | 5ull, mk = 0;                                      \
    unsigned s2 = k2;                                                                               \
    unsigned x = threadIdx.x, y = threadIdx.x * 3u;                                                 \
    double rc = 0.0;                                                                                \
    /* every lane its own cell, consecutive: no bank conflict, no address conflict */               \
    const unsigned lds_addr = threadIdx.x * 8u;                                                     \
    lds[threadIdx.x] = 0.0;                                                                         \
    (void)lds_addr; (void)s2; (void)mk; (void)x; (void)y; (void)rc;                                 \
    _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {                                          \
      _Pragma("unroll") for (int j = 0; j < kUnroll; ++j) { BODY }                                  \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] =                                                    \
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(s0 & 1) + (double)x + (double)(mk & 1) +   \
        lds[threadIdx.x] + rc;                                                                      \
  }

KERNEL(k_valu, VALU8)
KERNEL(k_salu2, VALU8 SALU SALU)
KERNEL(k_salu4, VALU8_WITH_EVERY_OTHER(SALU))
KERNEL(k_salu8, VALU8_WITH(SALU))
KERNEL(k_salu16, VALU8_WITH(SALU SALU))
KERNEL(k_salu_chain12, VALU8_WITH_EVERY_OTHER(SALU_CHAIN))
KERNEL(k_br_untaken1, VALU8_WITH_ONE(BR_UNTAKEN))
KERNEL(k_br_untaken4, VALU8_WITH_EVERY_OTHER(BR_UNTAKEN))
KERNEL(k_br_taken1, VALU8_WITH_ONE(BR_TAKEN))
KERNEL(k_br_taken4, VALU8_WITH_EVERY_OTHER(BR_TAKEN))
KERNEL(k_cmp1, VALU8_WITH_ONE(CMP_ONLY))
KERNEL(k_cmp_salu_select1, VALU8_WITH_ONE(CMP_SALU_SELECT))
KERNEL(k_cmp_salu_select4, VALU8_WITH_EVERY_OTHER(CMP_SALU_SELECT))
KERNEL(k_cmp_select4, VALU8_WITH_EVERY_OTHER(CMP_SELECT))
KERNEL(k_cmp_select_vcc4, VALU8_WITH_EVERY_OTHER(CMP_SELECT_VCC))
KERNEL(k_br_mask_scc4, VALU8_WITH_EVERY_OTHER(BR_MASK_SCC))
KERNEL(k_br_mask_vcc4, VALU8_WITH_EVERY_OTHER(BR_MASK_VCC))
KERNEL(k_exec_toggle4, VALU8_WITH_EVERY_OTHER(EXEC_TOGGLE))
KERNEL(k_lds1, VALU8_WITH_ONE(LDS_ADD))
KERNEL(k_lds4, VALU8_WITH_EVERY_OTHER(LDS_ADD))
KERNEL(k_snop8, VALU8_WITH(SNOP))
KERNEL(k_rcp1, VALU8_WITH_ONE(RCP))
KERNEL(k_rcp4, VALU8_WITH_EVERY_OTHER(RCP))
KERNEL(k_rsq1, VALU8_WITH_ONE(RSQ))
KERNEL(k_rcp_only, RCP RCP RCP RCP RCP RCP RCP RCP)
KERNEL(k_facet_like, VALU8_WITH_EVERY_OTHER(SALU) BR_UNTAKEN CMP_SALU_SELECT)

typedef void (*kernel_t)(double*, double, unsigned long long, unsigned);

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const double ghz = 2.4;
  double* out = nullptr;
  CHECK(hipMalloc(&out, sizeof(double) * (size_t)cus * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct { const char* name; kernel_t k; int valu_extra; } ks[] = {
      {"8 v_mul_f64", k_valu, 0},
      {"+2 s_and_b64", k_salu2, 0},
      {"+4 s_and_b64", k_salu4, 0},
      {"+8 s_and_b64", k_salu8, 0},
      {"+16 s_and_b64", k_salu16, 0},
      {"+12 scalar, chains of 3", k_salu_chain12, 0},
      {"+1 branch not taken (s_cmp+s_cbranch)", k_br_untaken1, 0},
      {"+4 branches not taken", k_br_untaken4, 0},
      {"+1 branch taken", k_br_taken1, 0},
      {"+4 branches taken", k_br_taken4, 0},
      {"+1 v_cmp_lt_f64 -> sgpr", k_cmp1, 1},
      {"+1 (v_cmp -> s_and -> v_cndmask)", k_cmp_salu_select1, 2},
      {"+4 (v_cmp -> s_and -> v_cndmask)", k_cmp_salu_select4, 8},
      {"+4 (v_cmp -> v_cndmask), mask in sgprs", k_cmp_select4, 8},
      {"+4 (v_cmp -> v_cndmask), mask in vcc", k_cmp_select_vcc4, 8},
      {"+4 uniform branches on a mask, by scc (2 scalar)", k_br_mask_scc4, 0},
      {"+4 uniform branches on a mask, by vcc (4 scalar)", k_br_mask_vcc4, 0},
      {"+4 (saveexec, v_add_u32, restore exec)", k_exec_toggle4, 4},
      {"+1 ds_add_f64", k_lds1, 0},
      {"+4 ds_add_f64", k_lds4, 0},
      {"+8 s_nop 0", k_snop8, 0},
      {"+1 v_rcp_f64 (result unused by the block)", k_rcp1, 1},
      {"+4 v_rcp_f64", k_rcp4, 4},
      {"+1 v_rsq_f64", k_rsq1, 1},
      {"8 v_rcp_f64 alone (no v_mul_f64)", k_rcp_only, 8},
      {"+4 s_and, 1 branch not taken, 1 cmp-s_and-select", k_facet_like, 2},
  };
  printf("%d CUs; SIMD cycles per block (8 v_mul_f64 + what the row names; 32 = the vector unit alone; %.1f GHz assumed)\n", cus, ghz);
  printf("%-52s %8s %8s %8s %8s\n", "", "1 wave", "2 waves", "3 waves", "4 waves");
  for (auto& k : ks) {
    printf("%-52s", k.name);
    for (int waves = 1; waves <= 4; ++waves) {
      const int block = 256 * waves;
      const size_t lds = 100 * 1024; /* one workgroup per CU */
      CHECK(hipFuncSetAttribute((const void*)k.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k.k, dim3(cus), dim3(block), lds, 0, out, 1.0, 0xF0F0F0F0F0F0F0F0ull, 7u);
      CHECK(hipDeviceSynchronize());
      float ms = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k.k, dim3(cus), dim3(block), lds, 0, out, 1.0, 0xF0F0F0F0F0F0F0F0ull, 7u);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float one = 0;
        CHECK(hipEventElapsedTime(&one, e0, e1));
        ms = one < ms ? one : ms;
      }
      /* a SIMD ran `waves` waves, each kIters * kUnroll blocks */
      const double cycles = (double)ms * 1e-3 * ghz * 1e9 / ((double)kIters * kUnroll * waves);
      printf(" %8.2f", cycles);
    }
    printf("\n");
  }
  return 0;
}
