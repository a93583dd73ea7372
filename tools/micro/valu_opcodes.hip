// Issue cost of every vector opcode that appears in the hot loops of the history kernels
// (tools/isa_histogram.py lists them), measured on the box: wave64 instructions per SIMD
// per second at 4 waves per SIMD, reported relative to v_mul_f64 = 4 cycles (the clock under this load is not the nominal one).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_opcodes.hip -o gpurun_out/valu_opcodes
//   gpurun_out/valu_opcodes > gpurun_out/valu_cycles.json     (-> profiles/valu_cycles.json)
// Every kernel issues kIters * 32 copies of ONE opcode per wave over four independent
// register chains (a dependent chain issues at the same rate: profiles/r02/valu_peak.log);
// loop overhead is 3 scalar instructions per 32.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 2048;

#define LOOP(BODY)                                            \
  _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {      \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) { BODY }    \
  }

// 32-bit destination, two or three 32-bit sources
#define K_U32(NAME, A0, A1, A2, A3)                                                        \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint64_t seed) {              \
    unsigned a0 = (unsigned)seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;  \
    const uint64_t mask = seed * 0x9E3779B97F4A7C15ull; (void)mask;                        \
    LOOP(A0; A1; A2; A3;)                                                                  \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;                               \
  }
#define U2(OP) K_U32(k_##OP, \
  asm volatile(#OP " %0, %0, %1" : "+v"(a0) : "v"(a1)), asm volatile(#OP " %0, %0, %1" : "+v"(a1) : "v"(a2)), \
  asm volatile(#OP " %0, %0, %1" : "+v"(a2) : "v"(a3)), asm volatile(#OP " %0, %0, %1" : "+v"(a3) : "v"(a0)))
#define U3(OP) K_U32(k_##OP, \
  asm volatile(#OP " %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)), asm volatile(#OP " %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3)), \
  asm volatile(#OP " %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a0)), asm volatile(#OP " %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "v"(a1)))
// carry-producing / carry-consuming adds (VCC)
#define UCO(OP) K_U32(k_##OP, \
  asm volatile(#OP " %0, vcc, %0, %1" : "+v"(a0) : "v"(a1) : "vcc"), asm volatile(#OP " %0, vcc, %0, %1" : "+v"(a1) : "v"(a2) : "vcc"), \
  asm volatile(#OP " %0, vcc, %0, %1" : "+v"(a2) : "v"(a3) : "vcc"), asm volatile(#OP " %0, vcc, %0, %1" : "+v"(a3) : "v"(a0) : "vcc"))
#define UCI(OP) K_U32(k_##OP, \
  asm volatile(#OP " %0, vcc, %0, %1, vcc" : "+v"(a0) : "v"(a1) : "vcc"), asm volatile(#OP " %0, vcc, %0, %1, vcc" : "+v"(a1) : "v"(a2) : "vcc"), \
  asm volatile(#OP " %0, vcc, %0, %1, vcc" : "+v"(a2) : "v"(a3) : "vcc"), asm volatile(#OP " %0, vcc, %0, %1, vcc" : "+v"(a3) : "v"(a0) : "vcc"))
// shifts by a constant
#define USH(OP) K_U32(k_##OP, \
  asm volatile(#OP " %0, 7, %1" : "+v"(a0) : "v"(a1)), asm volatile(#OP " %0, 7, %1" : "+v"(a1) : "v"(a2)), \
  asm volatile(#OP " %0, 7, %1" : "+v"(a2) : "v"(a3)), asm volatile(#OP " %0, 7, %1" : "+v"(a3) : "v"(a0)))
// 32-bit compares: into VCC (e32) or an SGPR pair (e64)
#define UCMP32(OP) K_U32(k_##OP##_e32, \
  asm volatile(#OP " vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"), asm volatile(#OP " vcc, %0, %1" : : "v"(a1), "v"(a2) : "vcc"), \
  asm volatile(#OP " vcc, %0, %1" : : "v"(a2), "v"(a3) : "vcc"), asm volatile(#OP " vcc, %0, %1" : : "v"(a3), "v"(a0) : "vcc"))
#define UCMP64(OP) K_U32(k_##OP##_e64, \
  uint64_t m0; asm volatile(#OP " %0, %1, %2" : "=s"(m0) : "v"(a0), "v"(a1)), uint64_t m1; asm volatile(#OP " %0, %1, %2" : "=s"(m1) : "v"(a1), "v"(a2)), \
  uint64_t m2; asm volatile(#OP " %0, %1, %2" : "=s"(m2) : "v"(a2), "v"(a3)), uint64_t m3; asm volatile(#OP " %0, %1, %2" : "=s"(m3) : "v"(a3), "v"(a0)))

U2(v_xor_b32) U2(v_add_u32) U2(v_sub_u32) U2(v_subrev_u32) U2(v_and_b32) U2(v_or_b32) U2(v_min_i32)
U2(v_mul_lo_u32)
U3(v_alignbit_b32) U3(v_lshl_add_u32) U3(v_add_lshl_u32) U3(v_add3_u32) U3(v_and_or_b32) U3(v_lshl_or_b32)
UCO(v_add_co_u32) UCO(v_subrev_co_u32) UCI(v_addc_co_u32) UCI(v_subb_co_u32) UCI(v_subbrev_co_u32)
USH(v_lshlrev_b32) USH(v_ashrrev_i32) USH(v_lshrrev_b32)
UCMP32(v_cmp_lt_i32) UCMP32(v_cmp_eq_u32) UCMP64(v_cmp_lt_i32) UCMP64(v_cmp_ne_u32)
K_U32(k_v_mov_b32,
  asm volatile("v_mov_b32 %0, %1" : "+v"(a0) : "v"(a1)), asm volatile("v_mov_b32 %0, %1" : "+v"(a1) : "v"(a2)),
  asm volatile("v_mov_b32 %0, %1" : "+v"(a2) : "v"(a3)), asm volatile("v_mov_b32 %0, %1" : "+v"(a3) : "v"(a0)))
// selects: mask in VCC (e32; VCC written once, outside the loop) or in an SGPR pair (e64)
// (VCC is set by the first instruction of every group of four and read by all four: the
// compiler is not told, and has no reason to touch VCC inside a loop of asm statements)
K_U32(k_v_cndmask_b32_e32,
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1)), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a2)),
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3)), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a0)))
K_U32(k_v_cndmask_b32_e64,
  asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "s"(mask)), asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "s"(mask)),
  asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "s"(mask)), asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "s"(mask)))

// 64-bit destination (f64 or integer pair)
#define K_F64(NAME, A0, A1, A2, A3)                                                        \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint64_t seed) {              \
    double a0 = 1.0 + 1e-3 * (double)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; \
    unsigned u0 = threadIdx.x + 1, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;                  \
    (void)u0; (void)u1; (void)u2; (void)u3;                                                \
    LOOP(A0; A1; A2; A3;)                                                                  \
    out[blockIdx.x * 256 + threadIdx.x] = (uint64_t)__double_as_longlong(a0 + a1 + a2 + a3) + u0 + u1 + u2 + u3; \
  }
#define F2(OP) K_F64(k_##OP, \
  asm volatile(#OP " %0, %0, %1" : "+v"(a0) : "v"(a1)), asm volatile(#OP " %0, %0, %1" : "+v"(a1) : "v"(a2)), \
  asm volatile(#OP " %0, %0, %1" : "+v"(a2) : "v"(a3)), asm volatile(#OP " %0, %0, %1" : "+v"(a3) : "v"(a0)))
#define F3(OP) K_F64(k_##OP, \
  asm volatile(#OP " %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)), asm volatile(#OP " %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3)), \
  asm volatile(#OP " %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a0)), asm volatile(#OP " %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "v"(a1)))
#define F1(OP) K_F64(k_##OP, \
  asm volatile(#OP " %0, %1" : "+v"(a0) : "v"(a1)), asm volatile(#OP " %0, %1" : "+v"(a1) : "v"(a2)), \
  asm volatile(#OP " %0, %1" : "+v"(a2) : "v"(a3)), asm volatile(#OP " %0, %1" : "+v"(a3) : "v"(a0)))
#define FCMP32(OP) K_F64(k_##OP##_e32, \
  asm volatile(#OP " vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"), asm volatile(#OP " vcc, %0, %1" : : "v"(a1), "v"(a2) : "vcc"), \
  asm volatile(#OP " vcc, %0, %1" : : "v"(a2), "v"(a3) : "vcc"), asm volatile(#OP " vcc, %0, %1" : : "v"(a3), "v"(a0) : "vcc"))
#define FCMP64(OP) K_F64(k_##OP##_e64, \
  uint64_t m0; asm volatile(#OP " %0, %1, %2" : "=s"(m0) : "v"(a0), "v"(a1)), uint64_t m1; asm volatile(#OP " %0, %1, %2" : "=s"(m1) : "v"(a1), "v"(a2)), \
  uint64_t m2; asm volatile(#OP " %0, %1, %2" : "=s"(m2) : "v"(a2), "v"(a3)), uint64_t m3; asm volatile(#OP " %0, %1, %2" : "=s"(m3) : "v"(a3), "v"(a0)))
// f64 <- 32-bit source / 32-bit <- f64 source
#define FCVT_TO(OP) K_F64(k_##OP, \
  asm volatile(#OP " %0, %1" : "=v"(a0) : "v"(u0)), asm volatile(#OP " %0, %1" : "=v"(a1) : "v"(u1)), \
  asm volatile(#OP " %0, %1" : "=v"(a2) : "v"(u2)), asm volatile(#OP " %0, %1" : "=v"(a3) : "v"(u3)))
#define FCVT_FROM(OP) K_F64(k_##OP, \
  asm volatile(#OP " %0, %1" : "=v"(u0) : "v"(a0)), asm volatile(#OP " %0, %1" : "=v"(u1) : "v"(a1)), \
  asm volatile(#OP " %0, %1" : "=v"(u2) : "v"(a2)), asm volatile(#OP " %0, %1" : "=v"(u3) : "v"(a3)))

F2(v_mul_f64) F2(v_add_f64) F2(v_fmac_f64) F3(v_fma_f64) F3(v_div_fmas_f64) F3(v_div_fixup_f64)
F1(v_rcp_f64) F1(v_rsq_f64) F1(v_mov_b64) F1(v_frexp_mant_f64)
FCMP32(v_cmp_lt_f64) FCMP64(v_cmp_lt_f64) FCMP64(v_cmp_nle_f64) FCMP32(v_cmp_gt_i64) FCMP32(v_cmp_ne_u64)
FCVT_TO(v_cvt_f64_u32) FCVT_TO(v_cvt_f64_i32) FCVT_FROM(v_frexp_exp_i32_f64)
K_F64(k_v_ldexp_f64,
  asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a0) : "v"(u0)), asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a1) : "v"(u1)),
  asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a2) : "v"(u2)), asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a3) : "v"(u3)))
K_F64(k_v_div_scale_f64,
  asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a0) : "v"(a1) : "vcc"), asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a1) : "v"(a2) : "vcc"),
  asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a2) : "v"(a3) : "vcc"), asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a3) : "v"(a0) : "vcc"))
K_F64(k_v_lshl_add_u64,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a0) : "v"(a1)), asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a1) : "v"(a2)),
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a2) : "v"(a3)), asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a3) : "v"(a0)))
K_F64(k_v_lshlrev_b64,
  asm volatile("v_lshlrev_b64 %0, 3, %1" : "+v"(a0) : "v"(a1)), asm volatile("v_lshlrev_b64 %0, 3, %1" : "+v"(a1) : "v"(a2)),
  asm volatile("v_lshlrev_b64 %0, 3, %1" : "+v"(a2) : "v"(a3)), asm volatile("v_lshlrev_b64 %0, 3, %1" : "+v"(a3) : "v"(a0)))
K_F64(k_v_ashrrev_i64,
  asm volatile("v_ashrrev_i64 %0, 3, %1" : "+v"(a0) : "v"(a1)), asm volatile("v_ashrrev_i64 %0, 3, %1" : "+v"(a1) : "v"(a2)),
  asm volatile("v_ashrrev_i64 %0, 3, %1" : "+v"(a2) : "v"(a3)), asm volatile("v_ashrrev_i64 %0, 3, %1" : "+v"(a3) : "v"(a0)))
K_F64(k_v_mad_u64_u32,
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(u0), "v"(u1) : "vcc"), asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(u1), "v"(u2) : "vcc"),
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(u2), "v"(u3) : "vcc"), asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(u3), "v"(u0) : "vcc"))
K_F64(k_v_mad_i64_i32,
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(u0), "v"(u1) : "vcc"), asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(u1), "v"(u2) : "vcc"),
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(u2), "v"(u3) : "vcc"), asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(u3), "v"(u0) : "vcc"))

// What a select costs in context: an f64 compare and the two v_cndmask_b32 of an f64 select,
// with the mask in VCC (VOP2 form) or in an SGPR pair (VOP3 form); three instructions per group
K_F64(k_select_f64_via_vcc,
  asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1) : "vcc"),
  asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(u1), "+v"(u2) : "v"(a1), "v"(a2) : "vcc"),
  asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(u2), "+v"(u3) : "v"(a2), "v"(a3) : "vcc"),
  asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(u3), "+v"(u0) : "v"(a3), "v"(a0) : "vcc"))
K_F64(k_select_f64_via_sgpr,
  uint64_t m0; asm volatile("v_cmp_lt_f64 %2, %3, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %1, %2\n v_cndmask_b32 %1, %1, %0, %2" : "+v"(u0), "+v"(u1), "=&s"(m0) : "v"(a0), "v"(a1)),
  uint64_t m1; asm volatile("v_cmp_lt_f64 %2, %3, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %1, %2\n v_cndmask_b32 %1, %1, %0, %2" : "+v"(u1), "+v"(u2), "=&s"(m1) : "v"(a1), "v"(a2)),
  uint64_t m2; asm volatile("v_cmp_lt_f64 %2, %3, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %1, %2\n v_cndmask_b32 %1, %1, %0, %2" : "+v"(u2), "+v"(u3), "=&s"(m2) : "v"(a2), "v"(a3)),
  uint64_t m3; asm volatile("v_cmp_lt_f64 %2, %3, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %1, %2\n v_cndmask_b32 %1, %1, %0, %2" : "+v"(u3), "+v"(u0), "=&s"(m3) : "v"(a3), "v"(a0)))
// v_cndmask_b32 (VOP2, mask in VCC) with VCC written once, by a scalar move, before the loop
__global__ __launch_bounds__(256) void k_v_cndmask_b32_e32_vcc_set(uint64_t* out, uint64_t seed) {
  unsigned a0 = (unsigned)seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  const uint64_t mask = seed * 0x9E3779B97F4A7C15ull;
  asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  LOOP(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a2));
       asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a0));)
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
// the same with the VOP3 encoding reading VCC (v_cndmask_b32_e64 ..., vcc)
__global__ __launch_bounds__(256) void k_v_cndmask_b32_e64_vcc(uint64_t* out, uint64_t seed) {
  unsigned a0 = (unsigned)seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  const uint64_t mask = seed * 0x9E3779B97F4A7C15ull;
  asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  LOOP(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a1) : "v"(a2));
       asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a3) : "v"(a0));)
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
// four independent chains that do NOT read each other (is it the operand pattern?)
__global__ __launch_bounds__(256) void k_v_cndmask_b32_e32_indep(uint64_t* out, uint64_t seed) {
  unsigned a0 = (unsigned)seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  unsigned b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;
  const uint64_t mask = seed * 0x9E3779B97F4A7C15ull;
  asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  LOOP(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b0)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b1));
       asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b2)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b3));)
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

struct Entry { const char* op; void (*kernel)(uint64_t*, uint64_t); };
#define E(OP) {#OP, k_##OP}
#define E2(OP, SUF) {#OP "_" #SUF, k_##OP##_##SUF}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int blocks = cus * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
  uint64_t* buf; CHECK(hipMalloc((void**)&buf, 8ull * 256 * blocks));
  std::vector<Entry> entries = {
    E(v_fma_f64), E(v_fmac_f64), E(v_mul_f64), E(v_add_f64), E(v_rcp_f64), E(v_rsq_f64), E(v_mov_b64),
    E(v_frexp_mant_f64), E(v_frexp_exp_i32_f64), E(v_ldexp_f64), E(v_div_scale_f64), E(v_div_fmas_f64),
    E(v_div_fixup_f64), E(v_cvt_f64_u32), E(v_cvt_f64_i32),
    E2(v_cmp_lt_f64, e32), E2(v_cmp_lt_f64, e64), E2(v_cmp_nle_f64, e64), E2(v_cmp_gt_i64, e32), E2(v_cmp_ne_u64, e32),
    E(v_lshl_add_u64), E(v_lshlrev_b64), E(v_ashrrev_i64), E(v_mad_u64_u32), E(v_mad_i64_i32),
    E(v_xor_b32), E(v_add_u32), E(v_sub_u32), E(v_subrev_u32), E(v_and_b32), E(v_or_b32), E(v_min_i32), E(v_mul_lo_u32),
    E(v_mov_b32), E(v_alignbit_b32), E(v_lshl_add_u32), E(v_add_lshl_u32), E(v_add3_u32), E(v_and_or_b32), E(v_lshl_or_b32),
    E(v_add_co_u32), E(v_subrev_co_u32), E(v_addc_co_u32), E(v_subb_co_u32), E(v_subbrev_co_u32),
    E(v_lshlrev_b32), E(v_ashrrev_i32), E(v_lshrrev_b32),
    E2(v_cmp_lt_i32, e32), E2(v_cmp_eq_u32, e32), E2(v_cmp_lt_i32, e64), E2(v_cmp_ne_u32, e64),
    E(v_cndmask_b32_e32), E(v_cndmask_b32_e64),
    {"v_cndmask_b32_e32 (vcc set once)", k_v_cndmask_b32_e32_vcc_set},
    {"v_cndmask_b32_e64 reading vcc", k_v_cndmask_b32_e64_vcc},
    {"v_cndmask_b32_e32 (independent chains)", k_v_cndmask_b32_e32_indep},
    {"(v_cmp_lt_f64 + 2 v_cndmask via vcc) / 3", k_select_f64_via_vcc},
    {"(v_cmp_lt_f64 + 2 v_cndmask via sgpr) / 3", k_select_f64_via_sgpr},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<double> ms(entries.size());
  auto per_instruction = [&](size_t k) { return entries[k].op[0] == '(' ? 3.0 : 1.0; };
  for (size_t k = 0; k < entries.size(); ++k) {
    hipLaunchKernelGGL(entries[k].kernel, dim3(blocks), dim3(256), 0, 0, buf, (uint64_t)3);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(entries[k].kernel, dim3(blocks), dim3(256), 0, 0, buf, (uint64_t)3);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float t; CHECK(hipEventElapsedTime(&t, e0, e1));
      if (t < best) best = t;
    }
    ms[k] = best / per_instruction(k);
  }
  // wave64 instructions one SIMD issued: 4 waves x kIters x 32; cycles at the nominal clock,
  // and relative to v_mul_f64 = 4 (what bench.py prices with; entries[2])
  const double per_simd = 4.0 * kIters * 32.0;
  printf("{\n \"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": 4, \"nominal_clock_mhz\": %d,\n",
         p.name, cus, p.clockRate / 1000);
  printf(" \"note\": \"cycles = 4 * t(op) / t(v_mul_f64): issue cycles one wave64 instruction holds its SIMD for, four "
         "waves per SIMD; cycles_at_nominal_clock uses %d MHz\",\n", p.clockRate / 1000);
  printf(" \"ms\": {");
  for (size_t k = 0; k < entries.size(); ++k) printf("%s\"%s\": %.4f", k ? ", " : "", entries[k].op, ms[k]);
  printf("},\n \"cycles_at_nominal_clock\": {");
  for (size_t k = 0; k < entries.size(); ++k)
    printf("%s\"%s\": %.3f", k ? ", " : "", entries[k].op, ms[k] * 1e-3 * (p.clockRate * 1e3) / per_simd);
  printf("},\n \"cycles\": {");
  for (size_t k = 0; k < entries.size(); ++k) printf("%s\"%s\": %.3f", k ? ", " : "", entries[k].op, 4.0 * ms[k] / ms[2]);
  printf("}\n}\n");
  return 0;
}
