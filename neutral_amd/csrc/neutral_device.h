/*
 * neutral_device.h -- device-side building blocks of the MI355X (gfx950)
 * over-particle transport path: Threefry2x64-20, the (0,1] double conversion,
 * cross-section bracket search, distance-to-facet and the division / square-root /
 * logarithm forms of the event bodies (the bodies themselves, with the path-length
 * heating estimator, are in neutral_history.h).
 *
 * What is computed follows the reference's omp3 kernel set (the parity oracle,
 * SURVEY.md section 2.3); each routine names the lines it answers to.  How it
 * is computed is written for CDNA4: one lane per particle in 64-wide
 * wavefronts, the whole history in VGPRs, every floating-point expression in
 * the reference's association order (no fast-math, no reassociation) so that
 * integer state (cells, event counts, RNG stream) matches the oracle exactly
 * and tallies match to rounding.
 */
#ifndef NEUTRAL_AMD_DEVICE_H
#define NEUTRAL_AMD_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "neutral_log_table.h"

namespace neutral {

/* neutral_data.h:17-24 */
constexpr double kEvToJ = 1.60217646e-19;
constexpr double kAvogadros = 6.02214085774e23;
constexpr double kBarns = 1.0e-28;
constexpr double kParticleMass = 1.674927471213e-27;
constexpr double kMassNo = 1.0e2;
constexpr double kMolarMass = 1.0e-2;
constexpr double kMinEnergyOfInterest = 1.0e0;
constexpr double kOpenBoundCorrection = 1.0e-13;

/* ---- Threefry2x64-20 (Random123/threefry.h:190-293, 20 rounds :179) ------ */

/* 64-bit rotate as two 32-bit funnel shifts (v_alignbit_b32, full rate): the
 * generic (v << N) | (v >> (64 - N)) becomes 64-bit shifts plus ORs, four to
 * five instructions per rotate, and a Threefry call has twenty of them. */
template <unsigned N>
__device__ __forceinline__ uint64_t rotl64(uint64_t v) {
  static_assert(N > 0 && N < 64, "rotation out of range");
  const uint32_t lo = (uint32_t)v;
  const uint32_t hi = (uint32_t)(v >> 32);
  uint32_t nlo, nhi;
  if (N == 32) {
    nlo = hi;
    nhi = lo;
  } else if (N < 32) {
    /* alignbit(a, b, s) = low 32 bits of ({a,b} >> s) */
    nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - N);
    nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - N);
  } else {
    nhi = __builtin_amdgcn_alignbit(lo, hi, 64 - N);
    nlo = __builtin_amdgcn_alignbit(hi, lo, 64 - N);
  }
  return ((uint64_t)nhi << 32) | nlo;
}

#define NEUTRAL_TF_ROUND(R) \
  a += b;                   \
  b = rotl64<R>(b);         \
  b ^= a;

/* ctr = {counter, 0}, key = {pkey, master_key} (omp3/neutral.c:636-644). */
__device__ __forceinline__ void threefry2x64_20(uint64_t c0, uint64_t k0,
                                                uint64_t k1, uint64_t& r0,
                                                uint64_t& r1) {
  const uint64_t k2 = 0x1BD11BDAA9FC1A22ull ^ k0 ^ k1; /* threefry.h:170-171,203-209 */
  uint64_t a = c0 + k0;
  uint64_t b = k1; /* ctr.v[1] == 0 */
  /* rotation schedule threefry.h:86-93 */
  NEUTRAL_TF_ROUND(16) NEUTRAL_TF_ROUND(42) NEUTRAL_TF_ROUND(12) NEUTRAL_TF_ROUND(31)
  a += k1; b += k2 + 1;
  NEUTRAL_TF_ROUND(16) NEUTRAL_TF_ROUND(32) NEUTRAL_TF_ROUND(24) NEUTRAL_TF_ROUND(21)
  a += k2; b += k0 + 2;
  NEUTRAL_TF_ROUND(16) NEUTRAL_TF_ROUND(42) NEUTRAL_TF_ROUND(12) NEUTRAL_TF_ROUND(31)
  a += k0; b += k1 + 3;
  NEUTRAL_TF_ROUND(16) NEUTRAL_TF_ROUND(32) NEUTRAL_TF_ROUND(24) NEUTRAL_TF_ROUND(21)
  a += k1; b += k2 + 4;
  NEUTRAL_TF_ROUND(16) NEUTRAL_TF_ROUND(42) NEUTRAL_TF_ROUND(12) NEUTRAL_TF_ROUND(31)
  a += k2; b += k0 + 5;
  r0 = a;
  r1 = b;
}

#undef NEUTRAL_TF_ROUND

/* u64 -> (0,1]: round-to-nearest conversion, then *2^-64 + 2^-65
 * (omp3/neutral.c:646-651).  Both steps after the conversion are exact or
 * singly rounded, so fusing them cannot change the result. */
__device__ __forceinline__ double u64_to_unit(uint64_t r) {
  constexpr double factor = 5.421010862427522170037264004349708557128906250e-20;      /* 2^-64 */
  constexpr double half_factor = 2.710505431213761085018632002174854278564453125e-20; /* 2^-65 */
  /* (double)r, round to nearest, as one fused operation on the two halves: hi * 2^32 is
   * exact and the sum is rounded once -- the conversion the compiler expands to two
   * conversions, a v_ldexp_f64 and an addition, is the same value in one instruction less */
  const double as_double =
      __builtin_fma((double)(uint32_t)(r >> 32), 4294967296.0, (double)(uint32_t)r);
  return as_double * factor + half_factor;
}

/* omp3/neutral.c:632-652 */
__device__ __forceinline__ void generate_random_numbers(uint64_t pkey,
                                                        uint64_t master_key,
                                                        uint64_t counter,
                                                        double& rn0,
                                                        double& rn1) {
  uint64_t r0, r1;
  threefry2x64_20(counter, pkey, master_key, r0, r1);
  rn0 = u64_to_unit(r0);
  rn1 = u64_to_unit(r1);
}

/* 2^64 times the sample: (double)r + 0.5, rounded once -- u64_to_unit(r) is this scaled by a
 * power of two, which commutes with the rounding.  What a sample's consumers that can take the
 * power of two along use instead of the sample (the logarithm: log_core(.., -64); the
 * scattering cosine of the centre of mass: 1 - 2^-63 x): an addition of an inline constant where the
 * sample costs a multiply-add whose addend, 2^-65, is moved into a register pair first. */
__device__ __forceinline__ double u64_plus_half(uint64_t r) {
  return __builtin_fma((double)(uint32_t)(r >> 32), 4294967296.0, (double)(uint32_t)r) + 0.5;
}

/* Is the sample u64_to_unit(r) below one half?  Asked of the integer: float(r) rounds r to a
 * multiple of 2^10 up there, times 2^-64 plus 2^-65 is rounded once more, and the result is below
 * 0.5 exactly for r < 2^63 - 512 (2^63 - 512 itself is a tie that rounds to even, 2^63, and
 * 0.5 + 2^-65 rounds to 0.5) -- tests/test_oracle_pins.py walks the boundary and two million
 * random r.  One 64-bit comparison for two conversions, two fused multiply-adds and the
 * comparison: what an absorb-or-scatter decision at probability one half costs. */
__device__ __forceinline__ bool sample_below_half(uint64_t r) { return r < 0x7FFFFFFFFFFFFE00ull; }

/* element `index` (>= 0) of an f64 array through an unsigned 32-bit byte offset: the
 * load then takes its base from scalar registers and one shifted vector register,
 * where a signed 64-bit index costs a sign extension and a 64-bit add per access
 * (mesh arrays and tables are far below 4 GB: 128 MB at 4000^2) */
__device__ __forceinline__ const double* mesh_element(const double* base, int index) {
  return (const double*)((const char*)base + ((unsigned)index << 3));
}
__device__ __forceinline__ double* mesh_element(double* base, int index) {
  return (double*)((char*)base + ((unsigned)index << 3));
}

/* ---- cross-section tables -------------------------------------------------- */

struct CsTable {
  const double* keys;
  const double* values;
  int nentries;
};

/* Bracket search + linear interpolation (omp3/neutral.c:498-517).  The value
 * depends only on the unique bracket keys[ind] <= E < keys[ind+1], so a plain
 * bisection returns what the reference's stepping search returns.  Like the
 * reference it requires keys[0] <= E < keys[n-1]; outside that range the
 * reference loops forever or reads out of bounds, here the bracket clamps to
 * the first/last interval (extrapolation) so a wave can never hang. */
__device__ __forceinline__ int cs_bracket(const double* __restrict__ keys, int n,
                                          double energy) {
  int lo = 0;
  int hi = n - 1; /* invariant: keys[lo] <= E < keys[hi] (after clamping) */
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (energy < keys[mid]) {
      hi = mid;
    } else {
      lo = mid;
    }
  }
  return lo;
}

/* Exponent-bucketed index over a key array: bucket b holds the energies whose
 * IEEE-754 bit pattern satisfies (bits >> shift) - base == b, i.e. 2^(52-shift)
 * buckets per binade (positive doubles order like their bit patterns).
 * start[b] = index of the last key <= the bucket's lowest energy (clamped to
 * [0, n-2]); start has nbuckets + 1 entries.  For E in bucket b the bracket
 * index lies in [start[b], start[b+1]], so the bisection below starts from a
 * window of a few keys instead of the whole table: 1-3 dependent probes
 * instead of 15, same unique bracket, same interpolated value. */
struct CsIndex {
  const unsigned short* start; /* nbuckets + 1 entries; null = no index */
  int nbuckets;
  int shift;
  long long base;
};

/* bucket of an energy: (bits >> shift) - base, clamped to [0, nbuckets - 1].  shift >= 43
 * (2^(52 - shift) <= 512 buckets per binade), so the shifted bit pattern of any double fits
 * 32 bits with room to spare: the subtraction and the clamp are 32-bit operations (a
 * 64-bit clamp is two 64-bit compares and four selects per lookup). */
__device__ __forceinline__ int cs_bucket(double energy, int shift, long long base, int nbuckets) {
  const int raw = (int)(__double_as_longlong(energy) >> shift) - (int)base;
  const int top = nbuckets - 1;
  return (raw < 0) ? 0 : ((raw > top) ? top : raw);
}

template <typename IndexPtr>
__device__ __forceinline__ int cs_bracket_indexed(const double* __restrict__ keys, int n,
                                                  IndexPtr start, int nbuckets, int shift,
                                                  long long base, double energy) {
  const int b = cs_bucket(energy, shift, base, nbuckets);
  int lo = start[b];
  int hi = start[b + 1] + 1;
  hi = (hi > n - 1) ? n - 1 : hi;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (energy < *mesh_element(keys, mid)) {
      hi = mid;
    } else {
      lo = mid;
    }
  }
  return lo;
}

template <bool kChecked>
__device__ __forceinline__ double quotient_of_physical(double a, double b); /* below */

/* ---- the arithmetic policy: template <bool kChecked> --------------------------------
 * Every division, square root and logarithm of the event bodies exists in two forms
 * that deliver the same bits wherever both are defined:
 *   kChecked = false  the bare operation sequences (refined reciprocal + three
 *                     operations, rsq + Goldschmidt, the table-driven log) with no
 *                     range test, exact for operands in the PROVEN RANGE below;
 *   kChecked = true   IEEE operations on any operand: a range test in front of each
 *                     fast sequence and the compiler's own division / sqrt / log behind
 *                     it -- inf, NaN, zero and subnormal operands behave as in the
 *                     reference's C (omp3/neutral.c:127-146,231,311-317: a true-vacuum
 *                     cell of density 0 has cell_mfp = 1/0 = inf and runs on infinities).
 * Both instantiations of every history kernel are in the library.  Which one a step
 * runs is decided ON THE DEVICE, per step, from the step's own inputs (neutral_kernels.hip:
 * unphysical_values_kernel, tables_check_kernel): if any density of the mesh or any key
 * or value of the cross-section tables lies outside [2^-100, 2^100] (zero, negative, inf
 * and NaN included) the kernels of the fast instantiation return at entry and the host
 * runs the step with the checked one.
 *
 * PROVEN RANGE.  With densities rho, keys and values in [2^-100, 2^100] (and particle
 * energies inside the keys, without which omp3/neutral.c:498-517 is undefined):
 *   number density  n = rho * 6.02e25              in [2^-15, 2^186]
 *   Sigma = n * sigma * 1e-28                       in [2^-208, 2^193]
 *   cell_mfp = 1 / (Sigma_s + Sigma_a)              in [2^-194, 2^208]
 *   -log(rn) in {-0} u [2^-54, 45]  =>  mfp_to_collision = -log(rn) / Sigma_s within 2^+-262
 *   speed = sqrt(2 E eV / m)  with  2 E eV / m in [2^-74, 2^127]
 *   E'/E and E/E' in [0.96, 1.04];  1 - cos^2 in {+0} u [2^-53, 1]
 *   interpolation weight (E - k0) / (k1 - k0): differences of doubles of magnitude
 *     >= 2^-100 are zero or >= 2^-153
 *   flight distances d: zero, or at least an ulp of a coordinate times a ratio of
 *     speeds (> 2^-200), and at most the mesh or speed * dt
 * i.e. every operand and every quotient of a fast sequence is zero or inside
 * [2^-300, 2^300], where v_div_scale / v_div_fmas / v_div_fixup and the scaling inside
 * the compiler's sqrt are the identity (tests/test_hip_parity.py checks the sequences
 * against the compiler's operations on the device over that range). */
template <bool kChecked>
__device__ __forceinline__ double cs_interpolate(const double* __restrict__ keys,
                                                 const double* __restrict__ values,
                                                 int ind, double energy) {
  const double* k = mesh_element(keys, ind);
  const double* v = mesh_element(values, ind);
  const double k0 = k[0];
  const double k1 = k[1];
  const double v0 = v[0];
  const double v1 = v[1];
  return v0 + quotient_of_physical<kChecked>(energy - k0, k1 - k0) * (v1 - v0);
}

/* ---- square root without the wrapping ---------------------------------------------
 * The compiler's IEEE sqrt is v_rsq_f64 and two Goldschmidt steps plus a residual
 * correction (ten operations) wrapped in a scaling of tiny arguments (compare, two
 * ldexp, select) and a class test for zero and infinity.  The path's four roots
 * per collision take arguments that are zero or lie in [1e-17, 1e16]: for anything
 * in [2^-500, 2^500] the wrapping is the identity, so the ten operations alone
 * give the same correctly rounded bits (tested against numpy on the device,
 * tests/test_hip_parity.py); zero is answered directly and everything else goes
 * to the ordinary sqrt. */
/* the ten operations alone: for arguments KNOWN to lie in [2^-500, 2^500] */
/* Round 5: ONE coupled Goldschmidt step and the residual correction -- eight operations.  v_rsq_f64
 * is good to 2^-24.2 (measured: tools/micro/one_step.hip), the step leaves g1 and h1 within 2^-47.8,
 * and the correction's own error is the product of the two, 2^-95 of the root: the result is the
 * correctly rounded root unless the exact one lies within 2^-95 of a rounding boundary, once in
 * 2^42 arguments (no difference from sqrt() in 8.6e9 random arguments on the device, none in the
 * parity tests' millions; the compiler's ten operations -- a second step before the correction --
 * make that never).  The path's own tolerance is what the logarithm already uses: a last bit
 * of a flight in 10^13 roots, where one logarithm in fifty differs from libm's. */
__device__ __forceinline__ double sqrt_known_plain(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double g0 = x * y;
  const double h0 = 0.5 * y;
  const double r0 = __builtin_fma(-h0, g0, 0.5);
  const double g1 = __builtin_fma(g0, r0, g0);
  const double h1 = __builtin_fma(h0, r0, h0);
  const double d0 = __builtin_fma(-g1, g1, x);
  return __builtin_fma(d0, h1, g1);
}

__device__ __forceinline__ double sqrt_plain_range(double x) {
  const unsigned hi = (unsigned)__double2hiint(x);
  if (__builtin_expect(!((hi - (523u << 20)) < (1000u << 20)), 0)) { /* also negative, NaN, inf, 0 */
    asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
    return sqrt(x);
  }
  return sqrt_known_plain(x);
}

/* Roots whose argument stays in the plain range on proven-range input (the policy comment
 * above): a particle's energy lies inside the cross-section tables' keys (1e-2 ... 1e8 eV
 * in the shipped tables; outside them omp3/neutral.c:498-517 reads out of bounds or never
 * returns), so
 *   - the speed's argument 2 E eV / m (omp3/neutral.c:116,298) is E * 1.9e8,
 *   - the energy ratios E'/E and E/E' of a scatter (:264-265) lie in [(A-1)^2/(A+1)^2, its
 *     inverse] = [0.96, 1.04] whatever E is,
 * and the range test -- a compare, an exec-mask save and restore and a branch per root,
 * in the collision stage where scalar work and branches are what the waves wait on -- is
 * left to the checked instantiation (same bits; A/B in DESIGN.md). */
/* sqrt(1 - cos^2) of a scattering angle (omp3/neutral.c:266): the argument is +0 -- a
 * head-on cosine of exactly 1 -- or at least an ulp of 1 (2^-53), never in between and
 * never -0; zero is answered by a select instead of a branch.  (A cosine that rounding
 * pushed beyond 1 gives a negative argument and NaN here as in the reference.) */
template <bool kChecked>
__device__ __forceinline__ double sqrt_of_sine_squared(double x) {
  if (kChecked) {
    return sqrt_plain_range(x);
  }
  const double r = sqrt_known_plain(x);
  return (x == 0.0) ? 0.0 : r;
}

template <bool kChecked>
__device__ __forceinline__ double sqrt_of_physical(double x) {
  return kChecked ? sqrt_plain_range(x) : sqrt_known_plain(x);
}

/* ---- log of a sample (omp3/neutral.c:131,295: mfp = -log(rn)/Sigma_s) -----------
 * The device library's log() delivers a double-double result internally (it also
 * serves pow) and costs ~95 vector instructions; a collision draws one log and is
 * bound by vector issue.  Rounds 1-4 used the classical division-based evaluation
 * (s = f / (2 + f), ten Horner steps in s^2: 42 instructions and a reciprocal, 0.8 ulp);
 * log_core() below is the table-driven form that replaced it (21 instructions, two
 * loads, 0.51 ulp).  Error below 1 ulp (measured on the device against an 80-bit
 * reference: tests/test_hip_parity.py), like the libm the reference links -- the log is
 * the one operation of this path that is NOT identical between conforming math
 * libraries, and event counts have never depended on its last bit (section 3 of
 * DESIGN.md).  Samples lie in [2^-65, 1]; zero, negative, subnormal, infinite and NaN
 * arguments get log()'s answers from a rare branch. */
__device__ __forceinline__ double refined_reciprocal(double b);                           /* below */
__device__ __forceinline__ double quotient_by_reciprocal(double a, double b, double r); /* below */

/* fma(r, z, c) as ONE three-operand instruction (c stays where it is) */
__device__ __forceinline__ double horner_step(double r, double z, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double out;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(r), "v"(z), "v"(c));
  return out;
#else
  return __builtin_fma(r, z, c);
#endif
}

__device__ __forceinline__ double log_core(double x, int k_scaled) {
  /* Round 5: by table.  x = 2^k m, m in [1/2, 1); the entry of m's top eight fraction bits holds
   * r, a multiple of 1/64 with |m r - 1| <= 2^-6.9, and -log r in two parts (neutral_log_table.h,
   * tools/gen_log_table.py).  f = fma(m, r, -1) is EXACT (m r is a multiple of 2^-59 below
   * 2^-6 in magnitude), so
   *     log x = (k ln2_hi - log r|hi) + (f + (f^2 Q(f) + (k ln2_lo - log r|lo)))
   * with Q the series of (log(1+f) - f) / f^2 to f^6 (the tail is below 2^-65), the first sum
   * exact (both terms are multiples of 2^-32 below 64) and f joined to it by a two-sum: 0.51 ulp
   * at worst on the 2.5 million samples of tests/test_hip_parity.py (0.25 on average; the
   * division-based form it replaces: 0.8), 21 vector instructions and two table loads for 42 and
   * a reciprocal.  Just below one (m >= 127/128) r is 1 and the result is log1p of the exact
   * m - 1; at a power of two f is 0 and the two parts of -log 2 cancel k's exactly: log(1) = +0. */
  const double m = __builtin_amdgcn_frexp_mant(x); /* [0.5, 1) */
  const int k = __builtin_amdgcn_frexp_exp(x) + k_scaled;
  const unsigned entry = ((unsigned)__double2hiint(m) >> 12) & 0xFFu;
  const double* const e = kLogTable + 4u * entry;
  const double r = e[0];
  const double l_hi = e[1];
  const double l_lo = e[2];
  const double f = __builtin_fma(m, r, -1.0);
  /* (Horner steps through horner_step(): left to itself the compiler turns q = fma(q, f, c)
   * into v_mov_b64 tmp, c; v_fmac_f64 tmp, q, f -- a 4-cycle register copy per step in a
   * kernel that is bound by vector issue -- where v_fma_f64 takes c as a third operand) */
  double q = -1.0 / 8.0;
  q = horner_step(q, f, 1.0 / 7.0);
  q = horner_step(q, f, -1.0 / 6.0);
  q = horner_step(q, f, 1.0 / 5.0);
  q = horner_step(q, f, -1.0 / 4.0);
  q = horner_step(q, f, 1.0 / 3.0);
  q = __builtin_fma(q, f, -0.5); /* (an inline constant) */
  const double dk = (double)k;
  const double hi = __builtin_fma(dk, kLogLn2Hi, l_hi); /* exact */
  const double lo = __builtin_fma(dk, kLogLn2Lo, l_lo);
  const double t = __builtin_fma(f * f, q, lo);
  const double s = hi + f;
  const double err = (hi - s) + f; /* (|hi| >= |f| or hi == 0: the two-sum's short form) */
  return s + (err + t);
}

/* any argument: zero, negative, subnormal, infinite and NaN get log()'s answers */
__device__ __forceinline__ double log_of_sample(double x) {
  int k_scaled = 0;
  if (__builtin_expect(!((x >= 2.2250738585072014e-308) & (x <= 1.7976931348623157e308)), 0)) {
    asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
    if (x == 0.0) {
      return -__builtin_huge_val();
    }
    if (!(x > 0.0)) {
      return __builtin_nan(""); /* negative or NaN */
    }
    if (x > 1.7976931348623157e308) {
      return x; /* +inf */
    }
    x *= 18014398509481984.0; /* subnormal: scaled by 2^54 */
    k_scaled = -54;
  }
  return log_core(x, k_scaled);
}

/* a sample as generate_random_numbers makes it: u64 * 2^-64 + 2^-65, in [2^-65, 1] -- the
 * special cases above cannot occur (same policy switch as sqrt_of_physical) */
template <bool kChecked>
__device__ __forceinline__ double log_of_drawn_sample(double x) {
  return kChecked ? log_of_sample(x) : log_core(x, 0);
}

/* ---- quotients by a denominator that many numerators share --------------------
 * A facet crossing divides its path length by the speed and by the mean free
 * path of the cell (omp3/neutral.c:311-312), and neither changes from one vacuum
 * cell to the next.  IEEE division on this hardware is the sequence
 *     r0 = rcp(b); two Newton steps -> r; q0 = a*r; q = fma(fma(-b,q0,a), r, q0)
 * wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act when an
 * operand or the quotient comes near the ends of the exponent range or is
 * special.  For operands inside [2^-300, 2^300] they are the identity, so the
 * refined reciprocal r -- five of the eleven instructions and the only
 * quarter-rate one -- depends on b alone and can be kept; the three remaining
 * operations reproduce `a / b` bit for bit (tested against the compiler's own
 * division on the device, tests/test_hip_parity.py).  Anything outside the range
 * takes the ordinary division. */
/* Round 5: ONE Newton step.  v_rcp_f64 is good to 2^-24.4 (measured: tools/micro/one_step.hip), the step
 * leaves r within 2^-48.8 of 1/b, and the quotient's residual correction (quotient_by_reciprocal:
 * q0 = a r, q = q0 + (a - b q0) r) is then off by 2^-48.8 of a residual of 2^-48.8: 2^-97 of the
 * quotient -- the correctly rounded a / b unless the exact quotient lies within 2^-97 of a rounding
 * boundary, once in 2^44 divisions (no difference from the compiler's division in 8.6e9 random
 * quotients on the device; the second step of rounds 1 - 4 and of the compiler's own sequence makes
 * that never).  See sqrt_known_plain() for what that is measured against. */
__device__ __forceinline__ double refined_reciprocal(double b) {
  double r = __builtin_amdgcn_rcp(b);
  const double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

/* |v| in [2^-300, 2^300): no scaling, no special case in the division of two such */
__device__ __forceinline__ bool in_plain_division_range(double v) {
  const unsigned hi = (unsigned)__double2hiint(v) & 0x7FFFFFFFu;
  return (hi - (723u << 20)) < (600u << 20);
}

__device__ __forceinline__ double quotient_by_reciprocal(double a, double b, double r) {
  const double q0 = a * r;
  const double rem = __builtin_fma(-b, q0, a);
  return __builtin_fma(rem, r, q0);
}

/* a / b where neither operand nor the quotient leaves the plain range on proven-range
 * input, and a may be +0 (not -0): the interpolation weight (E - k0)/(k1 - k0)
 * of a table lookup (omp3/neutral.c:514: keys are positive and increasing, E >= k0, a
 * difference of doubles of magnitude 1e-2 ... 1e8 is zero or at least 1e-18), the mean free
 * path 1/(Sigma_s + Sigma_a) (:135) and the flight time d/speed (:297).  Eight operations
 * instead of the wrapped division's thirteen, same bits (the refined reciprocal and the
 * three operations of quotient_by_reciprocal: tested against the compiler's division,
 * tests/test_hip_parity.py); the checked policy divides. */
template <bool kChecked>
__device__ __forceinline__ double quotient_of_physical(double a, double b) {
  if (kChecked) {
    return a / b;
  }
  return quotient_by_reciprocal(a, b, refined_reciprocal(b));
}

/* a / b for a compile-time constant b: with y = RN(1/b) (rounded correctly by the
 * compiler) the three operations of quotient_by_reciprocal deliver the correctly
 * rounded quotient (Markstein: q0 faithful, exact residual by fma, one correction
 * with the correctly rounded reciprocal), i.e. the bits of a / b, for numerators
 * in the plain range; others take the ordinary division.  Tested on the device
 * against numpy for both constants of the collision (tests/test_hip_parity.py). */
template <typename Tag>
__device__ __forceinline__ double quotient_by_constant(double a, double b, double y) {
  if (__builtin_expect(in_plain_division_range(a), 1)) {
    return quotient_by_reciprocal(a, b, y);
  }
  asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
  return a / b;
}
/* the same for a numerator that stays in the plain range on proven-range input (see
 * sqrt_of_physical): an energy of 1e-2 ... 1e8 eV times 2 eV_TO_J (3.2e-19), or times
 * A^2 + 2 A mu + 1 (9 801 ... 10 201) -- no range test, no branch */
template <typename Tag, bool kChecked>
__device__ __forceinline__ double quotient_of_physical_by_constant(double a, double b, double y) {
  if (kChecked) {
    return quotient_by_constant<Tag>(a, b, y);
  }
  return quotient_by_reciprocal(a, b, y);
}
/* ---- the cosine of the laboratory scattering angle (omp3/neutral.c:263-265) ---------------
 *     0.5 * ((A + 1) * sqrt(e_new / e) - (A - 1) * sqrt(e / e_new))
 * Two quotients and two roots of arguments that are each other's reciprocals to an ulp
 * (e_new / e in [0.96, 1.04]).  The fast policy seeds the second of each from the first instead
 * of from the quarter-rate instructions, and the seeds are BETTER than those give:
 *   q1 = e_new / e        refined reciprocal of e and the residual correction, as everywhere;
 *   s1 = sqrt(q1)         v_rsq_f64, one coupled step (g1, h1 within 2^-47.8), the correction;
 *   q2 = e / e_new        starts from (2 h1)^2 = 1/q1 within 2^-46 and takes its reciprocal of
 *                         e_new from (1/e) (2 h1)^2: the residual correction is then within 2^-92;
 *   s2 = sqrt(q2)         starts from g = q2 s1 -- s1 is 1/sqrt(q2) to 2^-52 -- and takes the
 *                         correction alone: within 2^-104.
 * Each is the IEEE result unless the exact one lies that close to a rounding boundary (as
 * refined_reciprocal / sqrt_known_plain: once in 2^39 and more; tools/micro/scatter_cosine.hip
 * counts no difference from the IEEE evaluation in 8.6e9 random scatters).  23 operations and two
 * quarter-rate seeds for 28 and four. */
/* (root_ratio = sqrt(e_new / e) and inv_root_ratio = 1 / sqrt(e_new / e) to 2^-47: what the
 * speed after the scatter starts from, speed_after_scatter() below; fast policy only) */
template <bool kChecked>
__device__ __forceinline__ double scatter_cosine(double e, double e_new, double& root_ratio,
                                                 double& inv_root_ratio) {
  if (kChecked) {
    root_ratio = inv_root_ratio = 0.0; /* (not used) */
    return 0.5 * ((kMassNo + 1.0) * sqrt_plain_range(e_new / e) - (kMassNo - 1.0) * sqrt_plain_range(e / e_new));
  }
  const double r_e = refined_reciprocal(e);
  const double q1 = quotient_by_reciprocal(e_new, e, r_e);
  const double y = __builtin_amdgcn_rsq(q1);
  const double g0 = q1 * y;
  const double h0 = 0.5 * y;
  const double r0 = __builtin_fma(-h0, g0, 0.5);
  const double g1 = __builtin_fma(g0, r0, g0);
  const double h1 = __builtin_fma(h0, r0, h0);
  const double s1 = __builtin_fma(__builtin_fma(-g1, g1, q1), h1, g1);
  const double t = h1 + h1; /* 1 / sqrt(q1) */
  const double q2_seed = t * t;
  const double r_e_new = r_e * q2_seed;
  const double q2 = __builtin_fma(__builtin_fma(-e_new, q2_seed, e), r_e_new, q2_seed);
  const double g = q2 * s1;
  const double s2 = __builtin_fma(__builtin_fma(-g, g, q2), 0.5 * s1, g);
  root_ratio = s1;
  inv_root_ratio = t;
  return 0.5 * ((kMassNo + 1.0) * s1 - (kMassNo - 1.0) * s2);
}

struct ByParticleMass {};
struct ByMassNoPlusOneSquared {};
constexpr double kMassNoPlusOneSquared = (kMassNo + 1.0) * (kMassNo + 1.0);

/* ---- geometry (omp3/neutral.c:423-471) ------------------------------------- */

/* The coordinate a history aims at on one axis (omp3/neutral.c:438-447): the cell's upper
 * edge when it moves up the axis, and -- the bound being open on the left/bottom --
 * slightly past the lower edge when it moves down. */
__device__ __forceinline__ double facet_target(double omega, double e_lo, double e_hi) {
  return (omega >= 0.0) ? e_hi : (e_lo - kOpenBoundCorrection);
}

/* distance to the facet from the two targets (the rest of omp3/neutral.c:449-470) */
__device__ __forceinline__ void calc_distance_to_targets(double x, double y, double speed,
                                                         double u_x_inv, double u_y_inv,
                                                         double target_x, double target_y,
                                                         double& distance_to_facet, int& x_facet) {
  /* u_x_inv = 1/(omega_x*speed), u_y_inv = 1/(omega_y*speed): omp3/neutral.c:435-436 */
  const double ax = target_x - x;
  const double ay = target_y - y;
  const double dt_x = ax * u_x_inv;
  const double dt_y = ay * u_y_inv;
  x_facet = (dt_x < dt_y) ? 1 : 0;

  distance_to_facet = x_facet ? (ax * speed) * u_x_inv : (ay * speed) * u_y_inv;
}

__device__ __forceinline__ void calc_distance_to_facet(
    double x, double y, double omega_x, double omega_y, double speed, double u_x_inv,
    double u_y_inv, double ex_lo, double ex_hi, double ey_lo, double ey_hi,
    double& distance_to_facet, int& x_facet) {
  calc_distance_to_targets(x, y, speed, u_x_inv, u_y_inv, facet_target(omega_x, ex_lo, ex_hi),
                           facet_target(omega_y, ey_lo, ey_hi), distance_to_facet, x_facet);
}

template <bool kChecked>
__device__ __forceinline__ double speed_of(double energy) {
  /* omp3/neutral.c:117,297 */
  return sqrt_of_physical<kChecked>(quotient_of_physical_by_constant<ByParticleMass, kChecked>(
      2.0 * energy * kEvToJ, kParticleMass, 1.0 / kParticleMass));
}

/* The speed after a scatter (omp3/neutral.c:297), fast policy: its argument 2 E' eV / m is the old
 * one times e_new / e to a few ulps, so the old speed times sqrt(e_new / e) -- both in hand -- is
 * the new root to 2^-51, and half the old speed's reciprocal (the one the flight time has just
 * been divided with) times 1 / sqrt(e_new / e) is 1 / (2 root) to 2^-46: the residual correction
 * alone leaves 2^-97 (see sqrt_known_plain(); tools/micro/scatter_cosine.hip counts the speeds that
 * differ from sqrt() of the same argument too).  Five operations for a quarter-rate seed and seven. */
__device__ __forceinline__ double speed_after_scatter(double e_new, double speed, double r_speed,
                                                      double root_ratio, double inv_root_ratio) {
  const double arg = quotient_by_reciprocal(2.0 * e_new * kEvToJ, kParticleMass, 1.0 / kParticleMass);
  const double g = speed * root_ratio;
  const double h = (0.5 * r_speed) * inv_root_ratio;
  return __builtin_fma(__builtin_fma(-g, g, arg), h, g);
}

/* inside [2^-100, 2^100]: what the fast arithmetic policy is proven on (the comment at the
 * top of the policy section); zero, negative, infinite and NaN values are outside */
__device__ __forceinline__ bool in_proven_range(double v) {
  const unsigned hi = (unsigned)__double2hiint(v); /* sign bit set: fails the test below */
  return (hi - (923u << 20)) < (200u << 20);
}

}  // namespace neutral
#endif
