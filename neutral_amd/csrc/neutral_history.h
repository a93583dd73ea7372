/*
 * neutral_history.h -- one particle history held in registers, and the event
 * bodies that advance it.  Shared by every history kernel variant so that all
 * of them execute the same floating-point operations in the same order:
 * variants differ only in WHICH lane runs WHICH event WHEN, never in what an
 * event computes.
 *
 * Mapping to the reference (omp3/neutral.c):
 *   load/store        the Particle fields (neutral_data.h:45-61)
 *   prologue          handle_particles :103-131
 *   decide            the loop head :134-150,170 (which event is next)
 *   collide           collision_event :209-300
 *   cross_facet       facet_event :303-380
 *   census            census_event :383-405
 */
#ifndef NEUTRAL_AMD_HISTORY_H
#define NEUTRAL_AMD_HISTORY_H

#include "neutral_device.h"
#include "neutral_kernels.h"

namespace neutral {

/* No history of the reference's decks comes within three orders of magnitude of
 * this many events per timestep.  A degenerate input that would make the
 * reference's loop spin (e.g. a zero-length step that never advances) must not
 * hang a GPU shared with others: such a history is ended as if its time had run
 * out, and counted in StepCounters::aborted. */
constexpr unsigned kMaxEventsPerHistory = 1u << 27;

enum Event : int {
  kEvEnd = 0,       /* dt_to_census <= 0: the while loop at :134 exits */
  kEvFacet = 1,
  kEvCollision = 2,
  kEvCensus = 3,
};

struct History {
  /* particle state */
  double x, y, omega_x, omega_y, energy, weight, dt_to_census, mfp_to_collision;
  int cellx, celly, dead;
  /* locals of handle_particles that live across events */
  double local_density, micro_s, micro_a, number_density, macro_s, macro_a, speed;
  double energy_deposition;
  double track_length; /* weight * path length not yet tallied: scalar-flux tally only */
  /* Values the reference recomputes at every event although their inputs only
   * change at a collision, a reflection or a density change; they are
   * recomputed here exactly when an input changes (same operations on the same
   * operands, hence the same bits):
   *   u_x_inv, u_y_inv   1/(omega*speed)                  omp3/neutral.c:435-436
   *   cell_mfp           1/(macro_s+macro_a)              :135
   *   dep_sigma/dep_heat factors of the heating estimator :481-494 */
  double u_x_inv, u_y_inv, dep_sigma, dep_heat;
  /* stream kernel only (see refined_reciprocal): reciprocals of speed and cell_mfp
   * for the two quotients of a facet crossing, and whether each may be used */
  double r_speed, r_cell_mfp;
  /* stream kernel only: what a unit of path length deposits, ready for the tally -- weight,
   * the two heating factors, the number density and 1/N in one factor (deposit_rate below) */
  double dep_rate;
  int plain_div; /* bit 0: speed, bit 1: cell_mfp inside the plain division range */
  /* stream kernel only: the two coordinates the history aims at (facet_target of its cell's
   * edges and its direction).  A streaming history keeps its direction from facet to facet,
   * so a crossing replaces the edges of the axis it crossed -- loaded together with the
   * new cell's density, before the arithmetic of the crossing -- instead of reading all
   * four edges again at the next loop head (omp3/neutral.c:438-447 reads them per event) */
  double target_x, target_y;
  /* ... and how they follow from the mesh: target = edge[target_ix] + target_adj with
   * target_ix = cell + 1, target_adj = -0.0 for a history that moves up the axis (the cell's
   * upper edge: e + -0.0 is e, bit for bit, whatever e is) and target_ix = cell, target_adj
   * = -OPEN_BOUND_CORRECTION for one that moves down (omp3/neutral.c:442-447).  Both are
   * fixed between two reflections, so a crossing is one 8-byte load and one addition per
   * axis -- no direction test, no select. */
  double target_adj_x, target_adj_y;
  int target_ix, target_iy; /* edge numbers, counted over the whole mesh */
  /* ... and where a crossing leads (omp3/neutral.c:333-369), also fixed between two
   * reflections: the cell index moves by step (+1, -1; 0 for a direction cosine of exactly
   * zero, which crosses nothing on that axis) unless the history is in the wall cell (the
   * last one going up, cell 0 going down; -1, no cell, for step 0), where it reflects.
   * Cells of a history lie inside the global mesh, so "is in the wall cell" is what the
   * reference's cell >= last / cell <= 0 come to. */
  int step_x, step_y, wall_x, wall_y;
  unsigned id; /* particle index in the SoA store; RNG key = pid_base + id (omp3/neutral.c:89) */
  unsigned counter;
  unsigned nevents; /* events of this history so far: watchdog only */
  /* the decision taken at the loop head */
  double cell_mfp, distance;
  int x_facet;
  /* stream kernel: the lanes whose next facet is an x facet, as the comparison left them
   * (wave-uniform: lives in scalar registers; lanes outside the facet loop do not look) */
  unsigned long long m_x_facet;
  int ev;
};

/* ---- tally policies: WHERE update_tallies (omp3/neutral.c:408-420) adds -------- */

/* Every policy also says whether the scalar-flux tally (neutral_data.h:95; the
 * path-length estimator sum(weight * segment length) / N per cell, flushed where
 * the energy deposition is -- defined in oracle/neutral_oracle.c, the reference only
 * declares the array) is kept: kFlux is a compile-time property, so the event bodies
 * of the default build carry no trace of it. */

/* straight to the mesh in HBM: one global_atomic_add_f64 per tally */
template <bool kWithFlux>
struct GlobalTallyT {
  static constexpr bool kFlux = kWithFlux;
  static constexpr bool kUniformDensity = false; /* (see WindowCellTallyT) */
  __device__ __forceinline__ bool inside() const { return false; }
  __device__ __forceinline__ void operator()(const SolveArgs& a, int pcellx, int pcelly,
                                             double energy_deposition) const {
    const int cellx = pcellx - a.x_off;
    const int celly = pcelly - a.y_off;
    unsafeAtomicAdd(mesh_element(a.tally, celly * a.nx + cellx), energy_deposition * a.inv_ntotal_particles);
  }
  /* (a contribution that carries its 1/N already: deposit_rate) */
  __device__ __forceinline__ void scaled(const SolveArgs& a, int pcellx, int pcelly, double v) const {
    unsafeAtomicAdd(mesh_element(a.tally, (pcelly - a.y_off) * a.nx + (pcellx - a.x_off)), v);
  }
  __device__ __forceinline__ void flux(const SolveArgs& a, int pcellx, int pcelly,
                                       double track_length) const {
    unsafeAtomicAdd(mesh_element(a.flux_tally, (pcelly - a.y_off) * a.nx + (pcellx - a.x_off)),
                    track_length * a.inv_ntotal_particles);
  }
};
typedef GlobalTallyT<false> GlobalTally;

/* into a W x W window of the mesh held in LDS (ds_add_f64) when the cell lies
 * inside it, to HBM otherwise; the owner flushes the window to the mesh.  With the
 * scalar-flux tally a second window of the same geometry follows the first. */
typedef __attribute__((address_space(3))) double lds_double;

/* window edge in cells: one 128 x 128 window (128 KB) fills the LDS next to the cs
 * index; two windows of 88 x 88 (121 KB) take its place when the flux is kept -- of 100 x 100
 * (158 KB) in the stream kernel's instantiations that stage no index (histories start from
 * carried values: neutral_tiled.hip, kCarried), where the LDS is the windows' alone */
constexpr int kWindowCells = 128;
constexpr int kWindowCellsWithFlux = 88;
#ifndef NEUTRAL_FLUX_WINDOW_NO_INDEX
#define NEUTRAL_FLUX_WINDOW_NO_INDEX 100 /* (A/B: 88 is the window beside an index) */
#endif
constexpr int kWindowCellsWithFluxNoIndex = NEUTRAL_FLUX_WINDOW_NO_INDEX;
__host__ __device__ constexpr int window_cells(bool with_flux, bool no_index) {
  return with_flux ? (no_index ? kWindowCellsWithFluxNoIndex : kWindowCellsWithFlux) : kWindowCells;
}
/* A row of the window in LDS is this many cells longer than the window is wide: with rows of
 * exactly 128 cells (1 KB) the cells of one COLUMN share an LDS bank, and the histories a wave
 * streams together sit in a patch of neighbouring cells -- rows apart as often as columns
 * (profiles/r04/experiments/window_row_pad.log) */
#ifndef NEUTRAL_WINDOW_ROW_PAD
#define NEUTRAL_WINDOW_ROW_PAD 1
#endif
constexpr int kWindowRowPad = NEUTRAL_WINDOW_ROW_PAD;

template <bool kWithFlux, bool kNoIndex = false>
struct WindowTallyT {
  static constexpr bool kFlux = kWithFlux;
  static constexpr bool kUniformDensity = false;
  __device__ __forceinline__ bool inside() const { return false; }
  static constexpr int W = window_cells(kWithFlux, kNoIndex);
  static constexpr int S = W + kWindowRowPad; /* cells per row in LDS */
  lds_double* window; /* LDS, W rows of S, row-major (flux: the next W rows) */
  int ox;         /* local cell coordinates of window element (0,0) */
  int oy;
  __device__ __forceinline__ void add(const SolveArgs& a, int pcellx, int pcelly, double v,
                                      unsigned which, double* mesh) const {
    const int cellx = pcellx - a.x_off;
    const int celly = pcelly - a.y_off;
    const unsigned lx = (unsigned)(cellx - ox);
    const unsigned ly = (unsigned)(celly - oy);
    if (lx < (unsigned)W && ly < (unsigned)W) {
      /* ds_add_f64, no return value */
      (void)__hip_atomic_fetch_add(&window[which * (unsigned)(W * S) + ly * (unsigned)S + lx], v,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      unsafeAtomicAdd(mesh_element(mesh, celly * a.nx + cellx), v);
    }
  }
  __device__ __forceinline__ void operator()(const SolveArgs& a, int pcellx, int pcelly,
                                             double energy_deposition) const {
    add(a, pcellx, pcelly, energy_deposition * a.inv_ntotal_particles, 0u, a.tally);
  }
  /* (a contribution that carries its 1/N already: deposit_rate) */
  __device__ __forceinline__ void scaled(const SolveArgs& a, int pcellx, int pcelly, double v) const {
    add(a, pcellx, pcelly, v, 0u, a.tally);
  }
  __device__ __forceinline__ void flux(const SolveArgs& a, int pcellx, int pcelly,
                                       double track_length) const {
    add(a, pcellx, pcelly, track_length * a.inv_ntotal_particles, 1u, a.flux_tally);
  }
};

/* The same destination for a cell whose window coordinates the caller has already
 * worked out (the stream kernel needs them anyway, to decide whether a particle
 * that left the window should wait for the next pass). */
template <bool kWithFlux, bool kUniform = false, bool kNoIndex = false>
struct WindowCellTallyT {
  static constexpr bool kFlux = kWithFlux;
  /* kUniform: the density of every cell of the window, and of the cells around it, is the
   * same bits (TiledArgs::tile_uniform: checked on the device every step).  A history that
   * leaves a cell INSIDE such a window enters a cell of the density it already has: its
   * crossing needs neither the load of the new cell's density nor the compare that
   * follows it -- the one load of the facet loop whose result the next trip waits for.
   * The stream kernel compiles its facet loop for both kinds of window. */
  static constexpr bool kUniformDensity = kUniform;
  static constexpr int W = window_cells(kWithFlux, kNoIndex);
  static constexpr int S = W + kWindowRowPad; /* cells per row in LDS */
  lds_double* window;
  unsigned lx, ly; /* cell - window origin; >= W outside the window */
  /* the lanes of the wave whose cell is outside the window (the caller's ballot of
   * outside()): zero on most trips, and then the add below is the LDS add alone, without
   * the exec-mask bookkeeping of a two-sided branch */
  unsigned long long m_outside;
  /* (from the one comparison the mask was made of: asked again as lx < W & ly < W the
   * compiler compares again, once per use) */
  __device__ __forceinline__ bool inside() const { return __builtin_amdgcn_inverse_ballot_w64(~m_outside); }
  /* (its own comparison, not !inside(): a wave-wide "any lane outside?" on the negation of
   * a comparison goes through a vector register; on a comparison it is its lane mask) */
  __device__ __forceinline__ bool outside() const { return (lx >= (unsigned)W) | (ly >= (unsigned)W); }
  __device__ __forceinline__ void add(const SolveArgs& a, int pcellx, int pcelly, double v,
                                      unsigned which, double* mesh) const {
    /* (a power-of-two row: the row and the column share no bit, and saying so -- `|` -- makes
     * the address a shift-or and a shift; any other row length: a multiply-add) */
    const unsigned cell = ((S & (S - 1)) == 0) ? ((ly * (unsigned)S) | lx) : (ly * (unsigned)S + lx);
    lds_double* const slot = &window[which * (unsigned)(W * S) + cell];
    /* (this question is asked twice per trip -- here and at the end of cross_facet() -- and a
     * condition with two uses is kept as a lane mask: s_cselect, s_and with exec,
     * s_cbranch_vcc, 7 cycles of a SIMD's issue (tools/micro/issue_mix.hip) where a comparison
     * that feeds ONE branch is s_cmp + s_cbranch_scc, 2.  So the other place asks it of a
     * copy of the mask that went through an empty asm statement) */
    if (__builtin_expect(m_outside == 0, 1)) {
      /* every lane inside (most trips): the LDS add and nothing else */
      (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      asm volatile(""); /* (a branch of its own: merged with the one above, every trip pays
                         * the exec-mask bookkeeping of the two-sided form) */
      if (inside()) {
        (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        unsafeAtomicAdd(mesh_element(mesh, (pcelly - a.y_off) * a.nx + (pcellx - a.x_off)), v);
      }
    }
  }
  __device__ __forceinline__ void operator()(const SolveArgs& a, int pcellx, int pcelly,
                                             double energy_deposition) const {
    add(a, pcellx, pcelly, energy_deposition * a.inv_ntotal_particles, 0u, a.tally);
  }
  __device__ __forceinline__ void scaled(const SolveArgs& a, int pcellx, int pcelly, double v) const {
    add(a, pcellx, pcelly, v, 0u, a.tally);
  }
  __device__ __forceinline__ void flux(const SolveArgs& a, int pcellx, int pcelly,
                                       double track_length) const {
    add(a, pcellx, pcelly, track_length * a.inv_ntotal_particles, 1u, a.flux_tally);
  }
};

/* Where a kernel variant keeps the bucketed cs indexes: K1 reads them from
 * global memory (L1/L2 hits), the persistent K2 stages them in LDS once per
 * workgroup.  A null pointer means "no index": plain bisection. */
template <typename IndexPtr>
struct CsLookup {
  IndexPtr scatter_index;
  IndexPtr absorb_index;
};

template <typename IndexPtr>
__device__ __forceinline__ int bracket_of(const double* keys, int n, IndexPtr index, int index_n,
                                          int shift, long long base, double energy) {
  if (index) {
    return cs_bracket_indexed(keys, n, index, index_n, shift, base, energy);
  }
  return cs_bracket(keys, n, energy);
}

/* both microscopic cross sections for one energy */
template <bool kSameTables, bool kChecked, typename IndexPtr>
__device__ __forceinline__ void lookup_cs(const SolveArgs& a, const CsLookup<IndexPtr>& ix,
                                          double energy, double& micro_scatter,
                                          double& micro_absorb) {
  const int is = bracket_of(a.scatter_keys, a.scatter_n, ix.scatter_index, a.scatter_index_n,
                            a.index_shift, a.scatter_index_base, energy);
  micro_scatter = cs_interpolate<kChecked>(a.scatter_keys, a.scatter_values, is, energy);
  if (kSameTables) {
    micro_absorb = micro_scatter;
  } else {
    const int ia = bracket_of(a.absorb_keys, a.absorb_n, ix.absorb_index, a.absorb_index_n,
                              a.index_shift, a.absorb_index_base, energy);
    micro_absorb = cs_interpolate<kChecked>(a.absorb_keys, a.absorb_values, ia, energy);
  }
}

/* ---- the same lookup in two stages -------------------------------------------
 * begin: the search window (bucketed index or the whole table) and the first
 * bisection probe, ISSUED but not waited for; finish: the rest of the bisection
 * and the interpolation.  Whatever the caller puts between the two runs while the
 * probe is in flight.  Same comparisons in the same order as cs_bracket(_indexed),
 * hence the same bracket and the same value. */
struct BracketSearch {
  int lo, hi, mid;
  double kmid;
};

template <typename IndexPtr>
__device__ __forceinline__ BracketSearch bracket_begin(const double* __restrict__ keys, int n,
                                                       IndexPtr index, int index_n, int shift,
                                                       long long base, double energy) {
  BracketSearch s;
  if (index) {
    const int b = cs_bucket(energy, shift, base, index_n);
    s.lo = index[b];
    s.hi = index[b + 1] + 1;
    s.hi = (s.hi > n - 1) ? n - 1 : s.hi;
  } else {
    s.lo = 0;
    s.hi = n - 1;
  }
  s.mid = (s.lo + s.hi) >> 1;
  s.kmid = *mesh_element(keys, s.mid); /* lo <= mid <= hi: valid even when no probe is needed */
  return s;
}

__device__ __forceinline__ int bracket_finish(const double* __restrict__ keys,
                                              const BracketSearch& s, double energy) {
  int lo = s.lo;
  int hi = s.hi;
  if (hi - lo > 1) {
    if (energy < s.kmid) {
      hi = s.mid;
    } else {
      lo = s.mid;
    }
  }
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

struct CsSearch {
  BracketSearch scatter, absorb;
};

template <bool kSameTables, typename IndexPtr>
__device__ __forceinline__ CsSearch lookup_cs_begin(const SolveArgs& a,
                                                    const CsLookup<IndexPtr>& ix, double energy) {
  CsSearch s;
  s.scatter = bracket_begin(a.scatter_keys, a.scatter_n, ix.scatter_index, a.scatter_index_n,
                            a.index_shift, a.scatter_index_base, energy);
  if (!kSameTables) {
    s.absorb = bracket_begin(a.absorb_keys, a.absorb_n, ix.absorb_index, a.absorb_index_n,
                             a.index_shift, a.absorb_index_base, energy);
  }
  return s;
}

template <bool kSameTables, bool kChecked>
__device__ __forceinline__ void lookup_cs_finish(const SolveArgs& a, const CsSearch& s,
                                                 double energy, double& micro_scatter,
                                                 double& micro_absorb) {
  const int is = bracket_finish(a.scatter_keys, s.scatter, energy);
  micro_scatter = cs_interpolate<kChecked>(a.scatter_keys, a.scatter_values, is, energy);
  if (kSameTables) {
    micro_absorb = micro_scatter;
  } else {
    const int ia = bracket_finish(a.absorb_keys, s.absorb, energy);
    micro_absorb = cs_interpolate<kChecked>(a.absorb_keys, a.absorb_values, ia, energy);
  }
}

/* macroscopic cross sections from number_density and the microscopic ones
 * (omp3/neutral.c:114-116, :290-291, :376-377) and the loop head's :135 */
template <bool kChecked>
__device__ __forceinline__ void macroscopic_from_micro(History& h) {
  h.macro_s = h.number_density * h.micro_s * kBarns;
  h.macro_a = h.number_density * h.micro_a * kBarns;
  h.cell_mfp = quotient_of_physical<kChecked>(1.0, h.macro_s + h.macro_a);
}

template <bool kChecked>
__device__ __forceinline__ void macroscopic_from_density(History& h) {
  /* omp3/neutral.c:112-113, :289, :375.  A collision re-evaluates this quotient
   * from an unchanged density (:289): same operands, same bits, so collide()
   * keeps the value and calls macroscopic_from_micro() alone. */
  h.number_density = (h.local_density * kAvogadros / kMolarMass);
  macroscopic_from_micro<kChecked>(h);
}

/* x / (x + x) for two bit-identical operands, as p_absorb (omp3/neutral.c:224)
 * and the absorbed fraction (:481-482) become when both cs tables hold the same
 * data: x + x is exact and the quotient is exactly one half unless x is zero,
 * infinite or NaN, or the sum overflows -- those go through the division. */
template <bool kChecked>
__device__ __forceinline__ double half_or_quotient(double x, double sum) {
  if (kChecked) {
    if (__builtin_expect(sum != 0.0 && fabs(sum) < __builtin_huge_val(), 1)) {
      return 0.5;
    }
    asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
    return x / sum;
  }
  /* One half, without the test: on proven-range input (neutral_device.h: the arithmetic
   * policy) a cross section is neither zero nor infinite nor NaN and the sum cannot
   * overflow. */
  return 0.5;
}

template <bool kChecked>
__device__ __forceinline__ void refresh_speed_reciprocal(History& h) {
  h.r_speed = refined_reciprocal(h.speed);
  if (kChecked) {
    h.plain_div = (h.plain_div & ~1) | (in_plain_division_range(h.speed) ? 1 : 0);
  }
}

template <bool kChecked>
__device__ __forceinline__ void refresh_mfp_reciprocal(History& h) {
  h.r_cell_mfp = refined_reciprocal(h.cell_mfp);
  if (kChecked) {
    h.plain_div = (h.plain_div & ~2) | (in_plain_division_range(h.cell_mfp) ? 2 : 0);
  }
}

/* omp3/neutral.c:435-436 */
__device__ __forceinline__ void refresh_direction(History& h) {
  h.u_x_inv = 1.0 / (h.omega_x * h.speed); /* (a cosine can be exactly zero: the wrapped */
  h.u_y_inv = 1.0 / (h.omega_y * h.speed); /*  division; a select around the plain one: +-0) */
}

/* The same two reciprocals after a collision in the fast arithmetic policy: where no lane's
 * cosine is (nearly) zero -- the product of the two cosines, each at most 1 in magnitude, is at
 * least 2^-150, so omega * speed lies in [2^-187, 2^101] for every speed of the proven range --
 * the division's wrapping (v_div_scale / v_div_fmas / v_div_fixup) is the identity and the
 * refined reciprocal with its one correction delivers the bits of 1.0 / (omega * speed)
 * (neutral_device.h: quotient_by_reciprocal; 11 operations and one seed for 24 and two).  Asked of the wave: a
 * cosine of exactly zero, a NaN (a scattering cosine that rounding pushed past 1: :266) or
 * anything else outside sends every lane through the wrapped divisions above. */
__device__ __forceinline__ void refresh_direction_plain_or_wrapped(History& h) {
  const double both = h.omega_x * h.omega_y;
  const unsigned hi = (unsigned)__double2hiint(both) & 0x7FFFFFFFu;
  const bool plain = (hi - ((1023u - 150u) << 20)) < (200u << 20); /* 2^-150 <= |both| < 2^50 */
  if (__builtin_expect(__ballot(!plain) == 0ull, 1)) {
    /* ... and both off ONE quarter-rate seed: the refined reciprocal of ux uy (within 2^-48.8)
     * times uy is 1 / ux to 2^-48, and the residual correction with that same value as the
     * reciprocal leaves 2^-96 (tools/micro/scatter_cosine.hip counts the pairs that differ from
     * the two divisions) */
    const double ux = h.omega_x * h.speed;
    const double uy = h.omega_y * h.speed;
    const double r_both = refined_reciprocal(ux * uy);
    const double qx = r_both * uy;
    const double qy = r_both * ux;
    h.u_x_inv = __builtin_fma(__builtin_fma(-ux, qx, 1.0), qx, qx);
    h.u_y_inv = __builtin_fma(__builtin_fma(-uy, qy, 1.0), qy, qy);
  } else {
    asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
    refresh_direction(h);
  }
}

/* the energy- and table-dependent factors of calculate_energy_deposition
 * (omp3/neutral.c:481-494); deposit() below finishes the product */
template <bool kSameTables, bool kChecked>
__device__ __forceinline__ void refresh_deposition_terms(History& h) {
  const double microscopic_cs_total = h.micro_s + h.micro_a;
  constexpr double average_exit_energy_absorb = 0.0;
  /* identical tables: micro_s and micro_a are the same bits (lookup_cs) */
  const double absorbed_fraction = kSameTables
                                       ? half_or_quotient<kChecked>(h.micro_a, microscopic_cs_total)
                                       : (h.micro_a / microscopic_cs_total);
  const double absorption_heating = absorbed_fraction * average_exit_energy_absorb;
  const double average_exit_energy_scatter =
      h.energy * ((kMassNo * kMassNo + kMassNo + 1) / ((kMassNo + 1) * (kMassNo + 1)));
  const double scattering_heating = (1.0 - absorbed_fraction) * average_exit_energy_scatter;
  h.dep_heat = (h.energy - scattering_heating - absorption_heating);
  h.dep_sigma = (microscopic_cs_total * kBarns);
}

/* omp3/neutral.c:493-494 */
__device__ __forceinline__ double deposit(const History& h, double path_length) {
  return h.weight * path_length * h.dep_sigma * h.dep_heat * h.number_density;
}

/* The stream kernel's form of the same estimator.  Between two collisions only the path length
 * of :493-494's product changes (and the number density, where a facet leads into another
 * density), so the product of everything else -- and of the 1/N update_tallies multiplies by
 * (:414-416) -- is formed once, when the history starts and where its density changes, and a
 * facet deposits path_length * dep_rate: one multiplication per facet for five.  This is the ONE
 * place where the path departs from the reference's association order: the TALLY's value moves
 * in its last bits (a relative 1e-16 per contribution, on a sum whose order the atomics do not
 * fix either; bar: 1e-9 per cell against the oracle, 1e-12 against the stream deck's closed form),
 * no particle's state and no event count can -- nothing reads the tally back. */
__device__ __forceinline__ double deposit_rate(const History& h, const SolveArgs& a) {
  return h.weight * h.dep_sigma * h.dep_heat * h.number_density * a.inv_ntotal_particles;
}

__device__ __forceinline__ void load_particle(History& h, const SolveArgs& a, int pid) {
  h.x = a.p.x[pid];
  h.y = a.p.y[pid];
  h.omega_x = a.p.omega_x[pid];
  h.omega_y = a.p.omega_y[pid];
  h.energy = a.p.energy[pid];
  h.weight = a.p.weight[pid];
  h.cellx = a.p.cellx[pid];
  h.celly = a.p.celly[pid];
  h.dead = 0;
  h.id = (unsigned)pid;
}

__device__ __forceinline__ void store_particle_view(const History& h, const ParticleView& p,
                                                    int pid) {
  p.x[pid] = h.x;
  p.y[pid] = h.y;
  p.omega_x[pid] = h.omega_x;
  p.omega_y[pid] = h.omega_y;
  p.energy[pid] = h.energy;
  p.weight[pid] = h.weight;
  p.dt_to_census[pid] = h.dt_to_census;
  p.mfp_to_collision[pid] = h.mfp_to_collision;
  p.cellx[pid] = h.cellx;
  p.celly[pid] = h.celly;
  p.dead[pid] = h.dead;
}

__device__ __forceinline__ void store_particle(const History& h, const SolveArgs& a, int pid) {
  store_particle_view(h, a.p, pid);
}

constexpr int kRecStateBits = 3;
/* The record's last word: its state in the low bits and, above them, the history's
 * RNG counter -- what a history interrupted between two events needs beyond its
 * particle fields to go on elsewhere (another pass, the collision stage, another
 * rank) with the stream it would have had: omp3/neutral.c:131,235,294 count up from 0
 * within a timestep. */
/* ... and, in its top bit, whether the history changed direction, energy, weight or died in
 * this timestep (a collision, a reflection): kRecChanged.  The write-back of a record without it
 * leaves those fields of the interface's arrays alone -- they hold them already when the arrays
 * were current as the step began -- and reads only the record's first 48 bytes.  A history
 * carries the flag in the top bit of History::id while it is in the stream kernel's registers
 * (kIdChanged: set by the reflection's rare branch, no register of its own in the facet loop). */
constexpr int kRecChanged = (int)0x80000000u;
constexpr unsigned kIdChanged = 0x80000000u;
__device__ __forceinline__ int record_word(int state, unsigned counter) {
  return state | (int)((counter << kRecStateBits) & 0x7FFFFFFFu);
}
__device__ __forceinline__ int record_state(int word) { return word & ((1 << kRecStateBits) - 1); }
__device__ __forceinline__ unsigned record_counter(int word) {
  return ((unsigned)word & 0x7FFFFFFFu) >> kRecStateBits;
}

/* keep_changed: the record's kRecChanged flag goes on with the history (a migrant of the stream
 * kernel: the same timestep); otherwise it starts clean */
__device__ __forceinline__ void load_record(History& h, const SolveArgs& a, const ParticleRec& r,
                                            bool keep_changed = false) {
  h.x = r.x;
  h.y = r.y;
  h.omega_x = r.omega_x;
  h.omega_y = r.omega_y;
  h.energy = r.energy;
  h.weight = r.weight;
  h.dt_to_census = r.dt_to_census;
  h.mfp_to_collision = r.mfp_to_collision;
  h.cellx = r.cellx;
  h.celly = r.celly;
  h.dead = 0;
  h.id = r.id | ((keep_changed && (r.dead & kRecChanged)) ? kIdChanged : 0u);
  h.counter = record_counter(r.dead); /* (prologue starts from 0 whatever this says) */
}

/* ParticleRec::dead doubles as the record's state inside a timestep */
enum RecState : int {
  kRecIdle = 0,      /* alive, this step's history is complete */
  kRecDead = 1,      /* omp3/neutral.c:91,245 */
  kRecCollide = 2,   /* suspended at a collision, waits for the collision kernel */
  kRecMigrate = 3,   /* left its tally window, waits for the next streaming pass */
  /* spatial domain decomposition only (a rank that owns part of the mesh): */
  kRecEmigrate = 4,  /* crossed into another rank's part of the mesh: waits to be sent */
  kRecGone = 5,      /* sent: the slot is empty (dropped at the next sort) */
};
constexpr int kReachBits = 2;
constexpr unsigned kSummaryTileMask = (1u << (32 - kRecStateBits - kReachBits)) - 1u;

/* record summary: state in the top three bits, then the history's reach class (below),
 * then the tile of the cell (tiles of 1 << tile_shift cells per edge: the tiled variant
 * picks 16..128 per problem).  An emigrant's cell lies outside the local mesh: its tile
 * field is not used. */
__device__ __forceinline__ unsigned slot_summary(int state, int cellx, int celly, int tiles_x,
                                                 int tile_shift, unsigned reach = 0u) {
  return ((unsigned)state << (32 - kRecStateBits)) |
         (reach << (32 - kRecStateBits - kReachBits)) |
         ((unsigned)((celly >> tile_shift) * tiles_x + (cellx >> tile_shift)) & kSummaryTileMask);
}
__device__ __forceinline__ unsigned summary_reach(unsigned summary) {
  return (summary >> (32 - kRecStateBits - kReachBits)) & ((1u << kReachBits) - 1u);
}

/* How many facets a history will cross under the tally window of the tile it is in, in
 * four classes (quarters of twice the window edge).  Sparse problems sort by it INSIDE a
 * tile, so that the 64 histories a wave streams together are about equally long: a wave
 * lasts as long as its longest history, and where a workgroup gets one particle per lane
 * (the reference's decks as shipped: a thousand particles per tile and pass) nothing
 * refills the lanes that finish early.  An estimate (straight flight to the window's
 * edge); it orders work and changes no result. */
__device__ __forceinline__ unsigned reach_class(double omega_x, double omega_y, int local_cellx,
                                                int local_celly, int tile_shift, int window_cells,
                                                double cells_per_x, double cells_per_y) {
  const int tile = 1 << tile_shift;
  const int margin = (window_cells - tile) >> 1;
  const int fx = local_cellx & (tile - 1);
  const int fy = local_celly & (tile - 1);
  const double to_edge_x = (double)((omega_x >= 0.0) ? (tile - fx + margin) : (fx + 1 + margin));
  const double to_edge_y = (double)((omega_y >= 0.0) ? (tile - fy + margin) : (fy + 1 + margin));
  const double rate_x = fabs(omega_x) * cells_per_x; /* cells crossed per unit of path, per axis */
  const double rate_y = fabs(omega_y) * cells_per_y;
  const double path = fmin(to_edge_x / rate_x, to_edge_y / rate_y); /* (x / 0 = inf: the other axis) */
  const double facets = path * (rate_x + rate_y);
  const int cls = (int)(facets * (2.0 / (double)window_cells));
  return (unsigned)((cls < 0) ? 0 : ((cls > 3) ? 3 : cls));
}
__device__ __forceinline__ int summary_state(unsigned summary) {
  return (int)(summary >> (32 - kRecStateBits));
}
__device__ __forceinline__ unsigned summary_tile(unsigned summary) {
  return summary & kSummaryTileMask;
}

/* spatial domain decomposition: is the history's cell outside this rank's mesh? */
__device__ __forceinline__ bool outside_domain(const History& h, const SolveArgs& a) {
  return ((unsigned)(h.cellx - a.x_off) >= (unsigned)a.nx) |
         ((unsigned)(h.celly - a.y_off) >= (unsigned)a.ny);
}

typedef unsigned rec_quad __attribute__((ext_vector_type(4)));
constexpr int kRecQuads = kParticleRecBytes / 16;

/* the 80 bytes a record holds as five 16-byte quads (not the padding of an over-aligned one) */
/* changed: the history's direction, energy or weight differ from what it began the timestep with,
 * or it died (the collision stage: always; the stream kernel: History::id says) */
__device__ __forceinline__ void record_quads(const History& h, int state, rec_quad (&q)[kRecQuads],
                                             bool changed = false) {
  ParticleRec o;
  o.x = h.x;
  o.y = h.y;
  o.omega_x = h.omega_x;
  o.omega_y = h.omega_y;
  o.energy = h.energy;
  o.weight = h.weight;
  o.dt_to_census = h.dt_to_census;
  o.mfp_to_collision = h.mfp_to_collision;
  o.cellx = h.cellx;
  o.celly = h.celly;
  o.id = h.id & ~kIdChanged;
  o.dead = record_word(state, h.counter) | ((changed || (h.id & kIdChanged)) ? kRecChanged : 0);
  __builtin_memcpy(q, &o, kParticleRecBytes); /* (registers to registers: no type punning) */
}

__device__ __forceinline__ void store_record(const History& h, const SolveArgs& a,
                                             ParticleRec& r, int state, bool changed = false) {
  rec_quad q[kRecQuads];
  record_quads(h, state, q, changed);
  rec_quad* dst = (rec_quad*)__builtin_assume_aligned(&r, 16);
#pragma unroll
  for (int k = 0; k < kRecQuads; ++k) {
    dst[k] = q[k];
  }
}

/* ---- records that change hands INSIDE a launch --------------------------------------
 * A record that ANOTHER WORKGROUP ON ANY XCD may pick up before the kernel ends (a migrant queued
 * for another tile's window in the stream kernel) is stored WRITE-THROUGH (sc1: the bytes
 * leave this XCD's L2 for the memory side) and the storing wave waits for its stores
 * (drain_stores) BEFORE it publishes the record's number with an agent-scope atomic; whoever
 * picks the number up loads the record with sc1 loads, which bypass its CU's L1 (never
 * refreshed by other CUs' stores).  tools/micro/handoff_litmus.hip is this protocol on records
 * that share 128-byte lines with records other workgroups are rewriting at the same time, across
 * XCDs: 0 stale of 6.3e6 hops, where plain accesses see 1.6e6 (profiles/r04/experiments/).
 * (The collision stage's ring hand-backs are NOT this protocol: a ring is only taken from by
 * waves of the same XCD -- the CU key carries the XCC id -- so the record is plain stores, drained
 * before the ring's control word says it waits, and the thief's take ends in an agent-scope
 * acquire before its plain loads: neutral_kernels.hip, try_steal.) */
__device__ __forceinline__ void store_record_through(const History& h, const SolveArgs& a,
                                                     ParticleRec& r, int state) {
  rec_quad q[kRecQuads];
  record_quads(h, state, q);
  static_assert(kRecQuads == 5, "five quads");
  asm volatile(
      "global_store_dwordx4 %0, %1, off sc1\n"
      "global_store_dwordx4 %0, %2, off offset:16 sc1\n"
      "global_store_dwordx4 %0, %3, off offset:32 sc1\n"
      "global_store_dwordx4 %0, %4, off offset:48 sc1\n"
      "global_store_dwordx4 %0, %5, off offset:64 sc1\n"
      /* (a store of more than 8 bytes reads its data registers up to two wait states after it
       * issues, and the compiler pads nothing around an asm statement: without this its next
       * instruction may overwrite them first -- the stream kernel's did, with the -1 of an
       * initialiser, in one record of eight; cdna_hip_programming.md 5.7 item 1) */
      "s_nop 1"
      :
      : "v"(&r), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4])
      : "memory");
}

/* every vector-memory operation of this wave has completed (gfx9: loads and stores share
 * vmcnt): what it stored write-through is where others can see it */
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void load_record_through(History& h, const SolveArgs& a,
                                                    const ParticleRec& r, bool keep_changed = false) {
  rec_quad q[kRecQuads];
  asm volatile(
      "global_load_dwordx4 %0, %5, off sc1\n"
      "global_load_dwordx4 %1, %5, off offset:16 sc1\n"
      "global_load_dwordx4 %2, %5, off offset:32 sc1\n"
      "global_load_dwordx4 %3, %5, off offset:48 sc1\n"
      "global_load_dwordx4 %4, %5, off offset:64 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4])
      : "v"(&r)
      : "memory");
  ParticleRec o;
  __builtin_memcpy(&o, q, kParticleRecBytes);
  load_record(h, a, o, keep_changed);
}

/* ... together with the 16-byte side record that belongs to it (SuspendExtra): six loads, one wait */
template <typename T>
__device__ __forceinline__ void load_record_through(History& h, const SolveArgs& a,
                                                    const ParticleRec& r, const T* side, T& side_out) {
  static_assert(sizeof(T) == 16, "one quad");
  rec_quad q[kRecQuads];
  rec_quad s;
  asm volatile(
      "global_load_dwordx4 %0, %6, off sc1\n"
      "global_load_dwordx4 %1, %6, off offset:16 sc1\n"
      "global_load_dwordx4 %2, %6, off offset:32 sc1\n"
      "global_load_dwordx4 %3, %6, off offset:48 sc1\n"
      "global_load_dwordx4 %4, %6, off offset:64 sc1\n"
      "global_load_dwordx4 %5, %7, off sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(s)
      : "v"(&r), "v"(side)
      : "memory");
  ParticleRec o;
  __builtin_memcpy(&o, q, kParticleRecBytes);
  load_record(h, a, o);
  __builtin_memcpy(&side_out, &s, 16);
}

/* a 16-byte side record (SuspendExtra) the same way */
template <typename T>
__device__ __forceinline__ void store_through16(T* dst, const T& v) {
  static_assert(sizeof(T) == 16, "one quad");
  rec_quad q;
  __builtin_memcpy(&q, &v, 16);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\ns_nop 1" : : "v"(dst), "v"(q) : "memory");
}
template <typename T>
__device__ __forceinline__ T load_through16(const T* src) {
  static_assert(sizeof(T) == 16, "one quad");
  rec_quad q;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\ns_waitcnt vmcnt(0)" : "=&v"(q) : "v"(src) : "memory");
  T v;
  __builtin_memcpy(&v, &q, 16);
  return v;
}

/* omp3/neutral.c:103-131 (initial == 1 always: :35-36) */
template <bool kSameTables, bool kChecked, typename IndexPtr>
__device__ __forceinline__ void prologue(History& h, const SolveArgs& a,
                                         const CsLookup<IndexPtr>& ix) {
  h.local_density = a.density[(h.celly - a.y_off + a.pad) * (a.nx + 2 * a.pad) +
                              (h.cellx - a.x_off + a.pad)];
  lookup_cs<kSameTables, kChecked>(a, ix, h.energy, h.micro_s, h.micro_a);
  macroscopic_from_density<kChecked>(h);
  h.speed = speed_of<kChecked>(h.energy);
  h.energy_deposition = 0.0;
  h.track_length = 0.0;
  h.counter = 0;
  h.nevents = 0;
  h.dt_to_census = a.dt;
  double rn0, rn1;
  generate_random_numbers(a.pid_base + (uint64_t)h.id, a.master_key, h.counter++, rn0, rn1);
  h.mfp_to_collision = -log_of_drawn_sample<kChecked>(rn0) / h.macro_s;
  refresh_direction(h);
  refresh_deposition_terms<kSameTables, kChecked>(h);
}

/* Re-derives the locals of a history that another kernel suspended at a loop
 * head after its streaming phase (no collision yet): the particle record holds
 * x, y, omega, energy, weight, cell AND the live dt_to_census / mfp_to_collision;
 * density, cross sections and speed are pure functions of those; the RNG counter
 * travels in the record's last word (one draw, the prologue's, for a history that
 * has only streamed) and nothing is pending in the deposition accumulator (every
 * facet flushes it, omp3/neutral.c:325-327). */
template <bool kSameTables, bool kChecked, typename IndexPtr>
__device__ __forceinline__ void resume(History& h, const SolveArgs& a,
                                       const CsLookup<IndexPtr>& ix) {
  /* h.dt_to_census and h.mfp_to_collision come from the record (load_record) */
  h.local_density = a.density[(h.celly - a.y_off + a.pad) * (a.nx + 2 * a.pad) +
                              (h.cellx - a.x_off + a.pad)];
  lookup_cs<kSameTables, kChecked>(a, ix, h.energy, h.micro_s, h.micro_a);
  macroscopic_from_density<kChecked>(h);
  h.speed = speed_of<kChecked>(h.energy);
  h.energy_deposition = 0.0;
  h.track_length = 0.0;
  /* h.counter comes from the record as well: 1 for a history that has only streamed */
  h.nevents = 0;
  refresh_direction(h);
  refresh_deposition_terms<kSameTables, kChecked>(h);
}

/* ---- the same two starts with what a history's start does NOT depend on taken out ----------
 * (stream kernel, identical tables, one rank's whole mesh).  Of the prologue's inputs
 * (omp3/neutral.c:103-131) two are known before the history's lane is: the microscopic cross
 * section -- a function of the energy, which a history keeps from the end of one timestep to
 * the start of the next and through every facet of its flight -- and the first sample of the
 * step, -log(rn0), a function of the particle id and the timestep alone.  The first travels with
 * the record (TiledArgs::carried_in / carried_out: written by whoever last changed the energy), the
 * second is worked out by the counting sort's placement pass, which touches every live record
 * once per step and waits on atomics while it does (CarriedStart::minus_log_rn0).  What is left of a
 * start is a chain of three dependent loads (sorted order -> record -> density) instead of
 * seven (+ index, 1-3 table probes, the bracket's keys and values), and a third of the vector
 * instructions: the refill was a quarter of the stream stage's wave time
 * (profiles/r04/experiments/fast_facets.log).  Same operations on the same operands as
 * prologue() / resume(): same bits.
 * density_known (wave-uniform): the history starts under a tally window whose cells all hold
 * one density (TiledArgs::tile_uniform, checked on the device this step) -- the caller has it in
 * a scalar register, and the chain is order -> record. */
template <bool kChecked>
__device__ __forceinline__ void start_carried(History& h, const SolveArgs& a, double micro,
                                              bool density_known, double known_density) {
  if (density_known) {
    h.local_density = known_density;
  } else {
    h.local_density = a.density[(h.celly - a.y_off + a.pad) * (a.nx + 2 * a.pad) +
                                (h.cellx - a.x_off + a.pad)];
  }
  h.micro_s = micro;
  h.micro_a = micro; /* identical tables: the same bits (lookup_cs) */
  macroscopic_from_density<kChecked>(h);
  h.speed = speed_of<kChecked>(h.energy);
  h.energy_deposition = 0.0;
  h.track_length = 0.0;
  h.nevents = 0;
  if (kChecked) {
    refresh_direction(h);
  } else {
    refresh_direction_plain_or_wrapped(h); /* (same bits: asked of the wave's starting lanes) */
  }
  refresh_deposition_terms<true, kChecked>(h);
}

/* omp3/neutral.c:103-131 with the lookup and the draw carried in: minus_log_rn0 = -log(rn0) of
 * generate_random_numbers(pid_base + id, master_key, 0) (draw_first_flight below) */
template <bool kChecked>
__device__ __forceinline__ void prologue_carried(History& h, const SolveArgs& a, double micro,
                                                 double minus_log_rn0, bool density_known,
                                                 double known_density) {
  start_carried<kChecked>(h, a, micro, density_known, known_density);
  h.counter = 1; /* (the draw of :127-131 has been made) */
  h.dt_to_census = a.dt;
  h.mfp_to_collision = minus_log_rn0 / h.macro_s;
}

/* resume() with the lookup carried in (dt_to_census, mfp_to_collision and the counter come
 * from the record) */
template <bool kChecked>
__device__ __forceinline__ void resume_carried(History& h, const SolveArgs& a, double micro,
                                               bool density_known, double known_density) {
  start_carried<kChecked>(h, a, micro, density_known, known_density);
}

/* -log(rn0) of a history's first draw of the timestep (omp3/neutral.c:127-131, counter 0).  A
 * sample lies in [2^-65, 1]: log_core() is what both arithmetic policies evaluate there. */
__device__ __forceinline__ double draw_first_flight(uint64_t pkey, uint64_t master_key) {
  double rn0, rn1;
  generate_random_numbers(pkey, master_key, 0, rn0, rn1);
  return -log_core(rn0, 0);
}

/* the four mesh edges around a cell (omp3/neutral.c:438-447 reads them per event) */
struct CellEdges {
  double x_lo, x_hi, y_lo, y_hi;
};


__device__ __forceinline__ CellEdges load_edges(const SolveArgs& a, int cellx, int celly) {
  const int ex = cellx - a.x_off + a.pad;
  const int ey = celly - a.y_off + a.pad;
  const double* px = mesh_element(a.edgex, ex);
  const double* py = mesh_element(a.edgey, ey);
  return CellEdges{px[0], px[1], py[0], py[1]};
}

/* loop head, omp3/neutral.c:134-150,170: which event comes next, and how far.
 * `e` holds the edges of the history's cell: a collision leaves the cell alone, so
 * the collision kernel loads them once per cell instead of once per event.
 * kWatchdog = false: the caller counts events and applies the watchdog itself (the
 * stream kernel does so once per pass of up to 64 facets instead of once per facet). */
template <bool kWatchdog = true>
__device__ __forceinline__ void decide(History& h, const SolveArgs& a, const CellEdges& e) {
  /* the loop condition of :134.  Without the watchdog the whole function is a pure
   * computation, so it runs for every lane and the condition becomes a select at
   * the end (no exec-mask region in the stream kernel's facet loop) */
  const bool running = (h.dt_to_census > 0.0);
  if (kWatchdog) {
    if (!running) {
      h.ev = kEvEnd;
      return;
    }
    if (++h.nevents > kMaxEventsPerHistory) {
      atomicAdd(&a.counters->aborted, 1u);
      h.ev = kEvEnd;
      return;
    }
  }
  double distance_to_facet;
  calc_distance_to_facet(h.x, h.y, h.omega_x, h.omega_y, h.speed, h.u_x_inv, h.u_y_inv,
                         e.x_lo, e.x_hi, e.y_lo, e.y_hi, distance_to_facet, h.x_facet);
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  /* omp3/neutral.c:150,170 as selects (& on purpose: no exec-mask regions) */
  const bool collides = (distance_to_collision < distance_to_facet) &
                        (distance_to_collision < distance_to_census);
  const bool crosses = (distance_to_facet < distance_to_census);
  h.ev = collides ? kEvCollision : (crosses ? kEvFacet : kEvCensus);
  h.distance = collides ? distance_to_collision
                        : (crosses ? distance_to_facet : distance_to_census);
  if (!kWatchdog) {
    h.ev = running ? h.ev : (int)kEvEnd;
  }
}

/* The collision stage's own loop head after a collision: is the next event another
 * collision?  The comparisons of decide() (omp3/neutral.c:150) without the selects that
 * name the event and its distance: a chain of collisions only asks this, and the history
 * whose answer is no gets its event named by decide() then, once, from the same state.
 * h.distance is the distance to the collision (what collide() moves by). */
__device__ __forceinline__ bool next_is_collision(History& h, const CellEdges& e,
                                                  double x_lo_open, double y_lo_open) {
  /* (x_lo_open = e.x_lo - OPEN_BOUND_CORRECTION, worked out when the cell was entered: a
   * collision changes the direction, not the cell) */
  double distance_to_facet;
  calc_distance_to_targets(h.x, h.y, h.speed, h.u_x_inv, h.u_y_inv,
                           (h.omega_x >= 0.0) ? e.x_hi : x_lo_open,
                           (h.omega_y >= 0.0) ? e.y_hi : y_lo_open, distance_to_facet, h.x_facet);
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  h.distance = distance_to_collision;
  return (h.dt_to_census > 0.0) & (distance_to_collision < distance_to_facet) &
         (distance_to_collision < distance_to_census);
}

/* Surely yes?  The same answer without the distance to the facet.  The way to an edge of the
 * cell ALONG the history's direction is at least the perpendicular distance to the nearest of the
 * four edges (a direction cosine is at most 1 in magnitude; the open lower bound lies beyond the
 * lower edge), so a collision closer than HALF of that -- the half covers the few ulps of the
 * computed quotients and of a cosine that rounding pushed past 1 -- comes before any facet: where
 * this says yes, next_is_collision() says yes, for every input of the fast arithmetic policy
 * (no NaN, no infinity among the operands: a history outside its cell by rounding has a negative
 * distance to an edge and is not sure).  Four differences, three minima and a sum where the
 * facet's distance is two direction tests, eight selects, two differences and six products --
 * and in a dense block the collision is ten million times closer than the edge.  h.distance is
 * the distance to the collision, as next_is_collision() leaves it. */
__device__ __forceinline__ bool surely_next_is_collision(History& h, const CellEdges& e) {
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  const double nearest = __builtin_fmin(__builtin_fmin(h.x - e.x_lo, e.x_hi - h.x),
                                        __builtin_fmin(h.y - e.y_lo, e.y_hi - h.y));
  h.distance = distance_to_collision;
  return (h.dt_to_census > 0.0) & (distance_to_collision + distance_to_collision < nearest) &
         (distance_to_collision < distance_to_census);
}

/* loop head for a history that carries its targets (stream kernel): no edge loads here */
__device__ __forceinline__ void decide_carried(History& h) {
  const bool running = (h.dt_to_census > 0.0);
  double distance_to_facet;
  calc_distance_to_targets(h.x, h.y, h.speed, h.u_x_inv, h.u_y_inv, h.target_x, h.target_y,
                           distance_to_facet, h.x_facet);
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  const bool collides = (distance_to_collision < distance_to_facet) &
                        (distance_to_collision < distance_to_census);
  const bool crosses = (distance_to_facet < distance_to_census);
  h.ev = collides ? kEvCollision : (crosses ? kEvFacet : kEvCensus);
  h.distance = collides ? distance_to_collision
                        : (crosses ? distance_to_facet : distance_to_census);
  h.ev = running ? h.ev : (int)kEvEnd;
}

/* the facet loop's own loop head: does the history cross another facet?  The same
 * comparisons as decide_carried() without the selects that name the event and its
 * distance -- a history that does NOT go on leaves the loop, and decide_carried() names
 * its event there, once, from the same state (same operands, same bits). */
/* (h.distance / h.x_facet: the distance to the next facet, worked out by cross_facet while
 * the new cell's density was still on its way) */
__device__ __forceinline__ bool next_is_facet(History& h) {
  const double distance_to_facet = h.distance;
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  const bool collides = (distance_to_collision < distance_to_facet) &
                        (distance_to_collision < distance_to_census);
  const bool crosses = (distance_to_facet < distance_to_census);
  return (h.dt_to_census > 0.0) & crosses & !collides;
}

/* The same question answered with one comparison: |d_facet| < min(d_census, d_collision).  Never
 * yes where next_is_facet() says no (dt <= 0 makes d_census <= 0; d_facet below both means
 * neither comparison of `collides` holds), and no where that says yes only if d_facet equals
 * d_collision exactly or is negative -- a history this stops is named by decide_carried() from
 * the same state and crosses on the wave's next pass.  Four comparisons and the scalar logic
 * that joins their masks become a minimum and a comparison (tools/micro/issue_mix.hip: scalar
 * instructions are not free beside vector ones). */
__device__ __forceinline__ bool surely_next_is_facet(const History& h) {
  const double distance_to_collision = h.mfp_to_collision * h.cell_mfp;
  const double distance_to_census = h.speed * h.dt_to_census;
  return __builtin_fabs(h.distance) < __builtin_fmin(distance_to_census, distance_to_collision);
}

/* which edges the history aims at on each axis (History::target_ix ...), from its direction */
__device__ __forceinline__ void aim_targets(History& h, const SolveArgs& a) {
  const bool up_x = (h.omega_x >= 0.0); /* omp3/neutral.c:438-447 */
  const bool up_y = (h.omega_y >= 0.0);
  h.target_ix = h.cellx + (up_x ? 1 : 0);
  h.target_iy = h.celly + (up_y ? 1 : 0);
  h.target_adj_x = up_x ? -0.0 : -kOpenBoundCorrection;
  h.target_adj_y = up_y ? -0.0 : -kOpenBoundCorrection;
  h.step_x = (h.omega_x > 0.0) ? 1 : ((h.omega_x < 0.0) ? -1 : 0);
  h.step_y = (h.omega_y > 0.0) ? 1 : ((h.omega_y < 0.0) ? -1 : 0);
  h.wall_x = (h.omega_x > 0.0) ? a.global_nx - 1 : ((h.omega_x < 0.0) ? 0 : -1);
  h.wall_y = (h.omega_y > 0.0) ? a.global_ny - 1 : ((h.omega_y < 0.0) ? 0 : -1);
}

/* Edge number `index` of an axis (counted over the whole mesh, as cells are: the offset into
 * the array this rank sees goes into the array's base, on the scalar unit): loaded, or
 * -- kComputed: the device has checked this step that the formula reproduces the array --
 * worked out: one conversion and one multiplication, rounded as the host's were (never
 * contracted into the addition that follows), and no load for the trip to wait for. */
/* (the product is the host's product, rounded by itself: contraction is switched off for
 * it, or the compiler fuses it with the addition of the open-bound correction that follows
 * and the last bit of a target -- hence of every position -- changes.  __dmul_rn does not
 * prevent that here: it is a plain `*`.) */
__device__ __forceinline__ double edge_from_formula(double spacing, int number) {
#pragma clang fp contract(off)
  const double product = spacing * (double)number;
  return product;
}
template <bool kComputed>
__device__ __forceinline__ double edge_x(const SolveArgs& a, int index) {
  return kComputed ? edge_from_formula(a.edge_dx, index)
                   : *mesh_element(a.edgex + (a.pad - a.x_off), index);
}
template <bool kComputed>
__device__ __forceinline__ double edge_y(const SolveArgs& a, int index) {
  return kComputed ? edge_from_formula(a.edge_dy, index)
                   : *mesh_element(a.edgey + (a.pad - a.y_off), index);
}

/* the targets of the history's cell and direction, from the edge arrays */
template <bool kComputedEdges = false>
__device__ __forceinline__ void load_targets(History& h, const SolveArgs& a) {
  aim_targets(h, a);
  h.target_x = __dadd_rn(edge_x<kComputedEdges>(a, h.target_ix), h.target_adj_x);
  h.target_y = __dadd_rn(edge_y<kComputedEdges>(a, h.target_iy), h.target_adj_y);
}

template <bool kWatchdog = true>
__device__ __forceinline__ void decide(History& h, const SolveArgs& a) {
  decide<kWatchdog>(h, a, load_edges(a, h.cellx, h.celly));
}

/* collision_event, omp3/neutral.c:209-300.  Returns true when the particle died.
 *
 * on_death(h) is called for a history that dies, AT the point of death (:243-252), with its
 * final state: the caller stores it there.  Nothing returns early: the lanes of the dead go
 * on through the rest of the body with everybody else (their instructions issue anyway, as
 * long as one lane of the wave lives) and what they compute is never looked at again.
 * With an early return every field the rest of the body updates was a conditional update,
 * and the compiler kept old and new copies of it apart with register moves around the
 * branch: 20 v_mov_b64 per pass of the collision stage, 4 % of its issue cycles. */
template <bool kSameTables, bool kChecked, typename IndexPtr, typename Tally, typename OnDeath>
__device__ __forceinline__ bool collide(History& h, const SolveArgs& a,
                                        const CsLookup<IndexPtr>& ix, const Tally& tally,
                                        const OnDeath& on_death) {
  const double distance_to_collision = h.distance;
  h.energy_deposition += deposit(h, distance_to_collision);
  if (Tally::kFlux) {
    h.track_length += h.weight * distance_to_collision; /* (the weight it travelled with) */
  }
  h.x += distance_to_collision * h.omega_x;
  h.y += distance_to_collision * h.omega_y;

  /* identical tables: macro_s and macro_a are the same bits (macroscopic_from_micro) */
  const double p_absorb = kSameTables ? half_or_quotient<kChecked>(h.macro_a, h.macro_s + h.macro_a)
                                      : h.macro_a / (h.macro_s + h.macro_a);
  double mu_cm; /* (:254: 1 - 2 x the second sample) */
  bool absorbed;
  if (kSameTables && !kChecked) {
    /* p_absorb is exactly one half: the first sample is compared as the integer it is made of
     * (sample_below_half: the same answer) and never converted; the second enters the cosine
     * as 2^64 times itself (u64_plus_half: 1 - 2 (2^-64 x) is one fused operation either way) */
    uint64_t r0, r1;
    threefry2x64_20(h.counter++, a.pid_base + (uint64_t)h.id, a.master_key, r0, r1);
    absorbed = sample_below_half(r0);
    mu_cm = __builtin_fma(u64_plus_half(r1), -0x1p-63, 1.0);
  } else {
    double rc0, rc1;
    generate_random_numbers(a.pid_base + (uint64_t)h.id, a.master_key, h.counter++, rc0, rc1);
    absorbed = (rc0 < p_absorb);
    mu_cm = 1.0 - 2.0 * rc1;
  }
  /* absorption: the weight drops; below 1 eV the history ends here.  The short pieces
   * are selects: a divergent region costs the collision stage more in exec-mask
   * bookkeeping and a branch than the few vector instructions it would skip. */
  const double absorbed_weight = h.weight * (1.0 - p_absorb);
  h.weight = absorbed ? absorbed_weight : h.weight;
  const bool died = absorbed & (h.energy < kMinEnergyOfInterest);
  if (__builtin_expect(died, 0)) {
    h.dead = 1;
    tally(a, h.cellx, h.celly, h.energy_deposition);
    h.energy_deposition = 0.0;
    if (Tally::kFlux) {
      tally.flux(a, h.cellx, h.celly, h.track_length);
      h.track_length = 0.0;
    }
    on_death(h);
  }
  /* The energy after the collision is known before the scattering angle is: the
   * table search for it (:281-286) starts here, and its first probe is in flight
   * while the scattered lanes work out their direction (:254-272). */
  const double e_scattered = quotient_of_physical_by_constant<ByMassNoPlusOneSquared, kChecked>(
      h.energy * (kMassNo * kMassNo + 2.0 * kMassNo * mu_cm + 1.0), kMassNoPlusOneSquared,
      1.0 / kMassNoPlusOneSquared);
  const double e_new = absorbed ? h.energy : e_scattered;
  const CsSearch search = lookup_cs_begin<kSameTables>(a, ix, e_new);
  /* the flight to the collision took distance / speed of the speed it was flown at (:296):
   * taken off here, ahead of the scatter that changes the speed -- and starts the new one from
   * this reciprocal */
  double r_speed_before = 0.0;
  if (kChecked) {
    h.dt_to_census -= distance_to_collision / h.speed;
  } else {
    r_speed_before = refined_reciprocal(h.speed);
    h.dt_to_census -= quotient_by_reciprocal(distance_to_collision, h.speed, r_speed_before);
  }
  if (!absorbed) {
    double root_ratio, inv_root_ratio;
    const double cos_theta = scatter_cosine<kChecked>(h.energy, e_new, root_ratio, inv_root_ratio);
    const double sin_theta = sqrt_of_sine_squared<kChecked>(1.0 - cos_theta * cos_theta);
    const double omega_x_new = (h.omega_x * cos_theta - h.omega_y * sin_theta);
    const double omega_y_new = (h.omega_x * sin_theta + h.omega_y * cos_theta);
    h.omega_x = omega_x_new;
    h.omega_y = omega_y_new;
    h.energy = e_new;
    /* (:297; an absorbed history keeps its energy, hence its speed: the same bits) */
    h.speed = kChecked ? speed_of<true>(e_new)
                       : speed_after_scatter(e_new, h.speed, r_speed_before, root_ratio, inv_root_ratio);
  }

  /* the draw for the next free flight (:293-295) needs nothing from the tables: it
   * and its logarithm are worked out while the search is still in flight */
  double minus_log_rn0;
  if (kChecked) {
    double rn0, rn1;
    generate_random_numbers(a.pid_base + (uint64_t)h.id, a.master_key, h.counter++, rn0, rn1);
    minus_log_rn0 = -log_of_sample(rn0);
  } else {
    /* (log(2^-64 x) = log_core(x, -64): the same mantissa, the power of two in the exponent) */
    uint64_t r0, r1;
    threefry2x64_20(h.counter++, a.pid_base + (uint64_t)h.id, a.master_key, r0, r1);
    minus_log_rn0 = -log_core(u64_plus_half(r0), -64);
  }

  lookup_cs_finish<kSameTables, kChecked>(a, search, h.energy, h.micro_s, h.micro_a);
  if (kChecked) {
    macroscopic_from_micro<true>(h); /* the density, hence number_density, has not changed (:289) */
    h.mfp_to_collision = minus_log_rn0 / h.macro_s; /* (-0.0 for a sample of exactly 1) */
  } else if (kSameTables) {
    /* Identical tables: Sigma_a is Sigma_s bit for bit, so the mean free path 1/(Sigma_s +
     * Sigma_a) (:135) is exactly half of 1/Sigma_s -- x + x and the halving are exact --
     * and it shares the refined reciprocal of Sigma_s with -log(rn)/Sigma_s (:295): two
     * plain quotients (neutral_device.h) off one reciprocal, 15 operations instead of the
     * 22 of a plain and a wrapped division.  The numerator of the second is -0.0 when the
     * sample is exactly 1, and so is the quotient then (by a select: the three
     * operations would give +0.0). */
    h.macro_s = h.number_density * h.micro_s * kBarns;
    h.macro_a = h.macro_s;
    const double r = refined_reciprocal(h.macro_s);
    h.cell_mfp = 0.5 * quotient_by_reciprocal(1.0, h.macro_s, r);
    const double q = quotient_by_reciprocal(minus_log_rn0, h.macro_s, r);
    h.mfp_to_collision = (minus_log_rn0 == 0.0) ? minus_log_rn0 : q;
  } else {
    macroscopic_from_micro<false>(h); /* the density, hence number_density, has not changed (:289) */
    h.mfp_to_collision = minus_log_rn0 / h.macro_s; /* (-0.0 for a sample of exactly 1) */
  }
  if (kChecked) {
    refresh_direction(h);
  } else {
    refresh_direction_plain_or_wrapped(h);
  }
  refresh_deposition_terms<kSameTables, kChecked>(h);
  return died;
}

/* facet_event, omp3/neutral.c:303-380.
 *
 * Written load-first: the neighbour cell (:333-369, integer work) is worked out
 * before the floating-point updates (:311-330) so that the density load of the
 * new cell (:372) is in flight while they execute -- issued after them it is a
 * dependent L2 round trip per facet (14 % of the stream kernel,
 * profiles/r01g/ablate_facetloads.log).  Every value is computed from the same
 * operands as in the reference's order: the position update uses the direction
 * BEFORE a reflection, as :329-330 precede :333. */
/* kDomain: may the neighbour cell belong to another rank (spatial decomposition)?
 * 0 never, 1 always, 2 ask a.decomposed (wave-uniform; free in the collision stage,
 * but six percent of the stream kernel, which therefore compiles both answers). */
/* kCarryTargets: the history keeps target_x / target_y (stream kernel): the edges of the
 * cell it enters are loaded here, next to the density, and the reflection -- rare -- sits
 * behind one wave-uniform branch instead of eight selects per facet. */
template <bool kChecked, bool kCachedReciprocals = false, int kDomain = 2, bool kCarryTargets = false,
          bool kComputedEdges = false, typename Tally>
__device__ __forceinline__ void cross_facet(History& h, const SolveArgs& a, const Tally& tally) {
  /* step to the neighbour cell, or reflect at the outer boundary (:333-369), as
   * selects: the branch ladder of the reference costs ~35 scalar instructions of
   * exec-mask bookkeeping per facet, and the stream kernel issues 0.7 scalar
   * instructions per vector instruction as it is */
  const bool xf = kCarryTargets ? __builtin_amdgcn_inverse_ballot_w64(h.m_x_facet) : (h.x_facet != 0);
  bool reflect;
  unsigned long long m_reflect = 0; /* kCarryTargets: the lanes that reflect */
  int ncellx, ncelly;
  /* Under a window of one density on a mesh whose edges are a formula nothing is LOADED for
   * the cell a crossing enters, so the step can be made first and the question "was that the
   * mesh's edge?" asked of its result: the edge number of the axis that moved has left [0, n]
   * (the other axis' cannot have).  Two comparisons and an OR of their masks, where asking
   * before the step is two comparisons and six scalar instructions that cut the step's
   * selects to size (tools/micro/issue_mix.hip: 2-3 cycles of the SIMD's issue each); the
   * rare lane that did reflect steps back in the branch that turns it round. */
  constexpr bool kStepFirst = kCarryTargets && kComputedEdges && Tally::kUniformDensity && kDomain == 0;
  if (kStepFirst) {
    ncellx = h.cellx + (xf ? h.step_x : 0);
    ncelly = h.celly + (xf ? 0 : h.step_y);
    /* (two masks joined on the scalar unit: the ballot of an OR of two comparisons comes out
     * as a 0/1 vector register and a third comparison) */
    m_reflect = __builtin_amdgcn_ballot_w64((unsigned)(h.target_ix + (ncellx - h.cellx)) >
                                            (unsigned)a.global_nx) |
                __builtin_amdgcn_ballot_w64((unsigned)(h.target_iy + (ncelly - h.celly)) >
                                            (unsigned)a.global_ny);
    reflect = __builtin_amdgcn_inverse_ballot_w64(m_reflect);
  } else if (kCarryTargets) {
    /* (History::step_x ...: the direction tests were made when the history last turned)
     * The axis that moves is picked first, with selects on 32-bit values: written as
     * logic on the two axes' booleans it comes out as 0/1 values in vector registers
     * combined by and / compare / select again -- 25 vector instructions where these
     * are 9. */
    /* ... or the logic is done where it is free: the comparisons' lane masks are combined
     * on the scalar unit and come back as select conditions (two compares and two selects
     * in the vector unit) */
    const unsigned long long m_xf = h.m_x_facet;
    const unsigned long long m_wall_x = __builtin_amdgcn_ballot_w64(h.cellx == h.wall_x);
    const unsigned long long m_wall_y = __builtin_amdgcn_ballot_w64(h.celly == h.wall_y);
    m_reflect = (m_xf & m_wall_x) | (~m_xf & m_wall_y);
    reflect = __builtin_amdgcn_inverse_ballot_w64(m_reflect);
    ncellx = h.cellx + (__builtin_amdgcn_inverse_ballot_w64(m_xf & ~m_wall_x) ? h.step_x : 0);
    ncelly = h.celly + (__builtin_amdgcn_inverse_ballot_w64(~m_xf & ~m_wall_y) ? h.step_y : 0);
  } else {
    const double omega = xf ? h.omega_x : h.omega_y;
    const int cell = xf ? h.cellx : h.celly;
    const int last = (xf ? a.global_nx : a.global_ny) - 1;
    const bool forward = (omega > 0.0);
    const bool backward = (omega < 0.0);
    /* (& and | on purpose: && / || come out as exec-mask regions) */
    reflect = (forward & (cell >= last)) | (backward & (cell <= 0));
    const int step = reflect ? 0 : (forward ? 1 : (backward ? -1 : 0));
    ncellx = h.cellx + (xf ? step : 0);
    ncelly = h.celly + (xf ? 0 : step);
  }
  /* (Loads come back in the order they were issued, so what is needed first is asked for
   * first: the edges, then the density.)
   * The edges the history aims at from the cell it will be in (two 8-byte loads issued
   * here, consumed after the arithmetic below): the target indices move with the cell; a
   * reflection leaves the cell where it is and turns the history round, which re-aims
   * the targets below */
  double edge_ahead_x = 0.0, edge_ahead_y = 0.0;
  if (kCarryTargets) {
    h.target_ix += ncellx - h.cellx;
    h.target_iy += ncelly - h.celly;
    int ex = h.target_ix;
    int ey = h.target_iy;
    if (kDomain == 1 || (kDomain == 2 && a.decomposed)) {
      /* (the neighbour may be another rank's cell: any edge of ours; the history stops) */
      const int ex_lo = a.x_off - a.pad, ex_hi = a.x_off + a.nx + a.pad;
      const int ey_lo = a.y_off - a.pad, ey_hi = a.y_off + a.ny + a.pad;
      ex = (ex < ex_lo) ? ex_lo : ((ex > ex_hi) ? ex_hi : ex);
      ey = (ey < ey_lo) ? ey_lo : ((ey > ey_hi) ? ey_hi : ey);
    }
    edge_ahead_x = edge_x<kComputedEdges>(a, ex);
    edge_ahead_y = edge_y<kComputedEdges>(a, ey);
  }
  if (kCarryTargets) {
    __builtin_amdgcn_sched_barrier(0); /* (the edge loads are issued by here) */
  }
  int dens_x = ncellx - a.x_off;
  int dens_y = ncelly - a.y_off;
  if (kDomain == 1 || (kDomain == 2 && a.decomposed)) {
    /* a rank that owns part of the mesh: the neighbour may be another rank's cell;
     * read a cell of ours (the history stops here and is sent on: its caller checks
     * outside_domain() before anything looks at the density) */
    dens_x = (dens_x < 0) ? 0 : ((dens_x >= a.nx) ? a.nx - 1 : dens_x);
    dens_y = (dens_y < 0) ? 0 : ((dens_y >= a.ny) ? a.ny - 1 : dens_y);
  }
  /* (a facet loop compiled for windows of one density, Tally::kUniformDensity, has no load
   * here: only a history that leaves a cell outside the window loads, below) */
  double new_density = h.local_density;
  if (!Tally::kUniformDensity) {
    new_density = *mesh_element(a.density, dens_y * a.nx + dens_x);
  }

  const double distance_to_facet = h.distance;
  if (kCachedReciprocals) {
    /* both quotients of :311-312 through the kept reciprocals */
    if (kChecked) {
      if (__builtin_expect((h.plain_div == 3) & in_plain_division_range(distance_to_facet), 1)) {
        h.mfp_to_collision -= quotient_by_reciprocal(distance_to_facet, h.cell_mfp, h.r_cell_mfp);
        h.dt_to_census -= quotient_by_reciprocal(distance_to_facet, h.speed, h.r_speed);
      } else {
        asm volatile("" ::: "memory"); /* keep the rare path a branch, not a select */
        h.mfp_to_collision -= (distance_to_facet / h.cell_mfp);
        h.dt_to_census -= (distance_to_facet / h.speed);
      }
    } else {
      /* No range test on proven-range input (neutral_device.h: the arithmetic policy).  A
       * distance to a facet is +0 (the particle sits on the edge it is heading for) or at
       * least an ulp of a coordinate times a speed/speed ratio, and at most the mesh; the
       * shipped decks have speeds of 1e3 ... 1e8 and mean free paths of 1e-6 ... 1e29
       * (densities of 1e-30 ... 1e4; a true vacuum of density 0 runs checked). */
      h.mfp_to_collision -= quotient_by_reciprocal(distance_to_facet, h.cell_mfp, h.r_cell_mfp);
      h.dt_to_census -= quotient_by_reciprocal(distance_to_facet, h.speed, h.r_speed);
    }
  } else {
    h.mfp_to_collision -= quotient_of_physical<kChecked>(distance_to_facet, h.cell_mfp);
    h.dt_to_census -= quotient_of_physical<kChecked>(distance_to_facet, h.speed);
  }
  if (kCarryTargets) {
    /* the stream kernel: nothing is pending in the accumulator (every facet flushes it and
     * collisions happen elsewhere; prologue() and resume() start it at zero), and 0 + d is d:
     * the accumulator stays out of the facet loop's registers */
    tally.scaled(a, h.cellx, h.celly, distance_to_facet * h.dep_rate); /* (deposit_rate) */
  } else {
    h.energy_deposition += deposit(h, distance_to_facet);
    tally(a, h.cellx, h.celly, h.energy_deposition);
    h.energy_deposition = 0.0;
  }
  if (Tally::kFlux) {
    tally.flux(a, h.cellx, h.celly, h.track_length + h.weight * distance_to_facet);
    h.track_length = 0.0;
  }

  h.x += distance_to_facet * h.omega_x;
  h.y += distance_to_facet * h.omega_y;
  if (kCarryTargets) {
    /* the position is final before the (rare) reflection turns the direction round: sunk
     * below that branch, old and new direction are both live across it and every trip of
     * the facet loop pays four 64-bit register copies for the one in a thousand that turns */
    asm volatile("" : "+v"(h.x), "+v"(h.y));
  }

  /* 1/((-omega)*speed) = -(1/(omega*speed)) bit for bit (IEEE multiplication and
   * division are sign-symmetric), so omp3/neutral.c:435-436 needs no divide here */
  const bool flip_x = reflect & xf;
  const bool flip_y = reflect & !xf;
  if (kCarryTargets) {
    h.cellx = ncellx;
    h.celly = ncelly;
    if (__builtin_expect(m_reflect != 0, 0)) { /* wave-uniform: most trips of the facet loop skip it */
      if (kStepFirst) {
        /* (the step that was the mesh's edge, taken back: the history stays in its cell) */
        ncellx -= (reflect & xf) ? h.step_x : 0;
        ncelly -= (reflect & !xf) ? h.step_y : 0;
        h.cellx = ncellx;
        h.celly = ncelly;
      }
      h.omega_x = flip_x ? -h.omega_x : h.omega_x;
      h.u_x_inv = flip_x ? -h.u_x_inv : h.u_x_inv;
      h.omega_y = flip_y ? -h.omega_y : h.omega_y;
      h.u_y_inv = flip_y ? -h.u_y_inv : h.u_y_inv;
      if (reflect) {
        h.id |= kIdChanged; /* (the direction is no longer what the arrays hold: kRecChanged) */
        /* turned round in the same cell: the other edge of it (a dependent load, here only) */
        aim_targets(h, a);
        double ex = edge_x<kComputedEdges>(a, h.target_ix);
        double ey = edge_y<kComputedEdges>(a, h.target_iy);
        /* (waited for here, inside the rare branch: the common path then knows how many
         * loads are in flight where the paths join, and waits for the edges alone) */
        asm volatile("" : "+v"(ex), "+v"(ey));
        edge_ahead_x = ex;
        edge_ahead_y = ey;
      }
    }
  } else {
    h.omega_x = flip_x ? -h.omega_x : h.omega_x;
    h.u_x_inv = flip_x ? -h.u_x_inv : h.u_x_inv;
    h.omega_y = flip_y ? -h.omega_y : h.omega_y;
    h.u_y_inv = flip_y ? -h.u_y_inv : h.u_y_inv;
  }
  h.cellx = ncellx;
  h.celly = ncelly;
  if (kCarryTargets) {
    h.target_x = __dadd_rn(edge_ahead_x, h.target_adj_x);
    h.target_y = __dadd_rn(edge_ahead_y, h.target_adj_y);
    /* the next loop head's distance to the facet needs the edges, not the density: it is
     * worked out here, ahead of the wait for the density below (the loads were issued in
     * that order), and pinned there */
    calc_distance_to_targets(h.x, h.y, h.speed, h.u_x_inv, h.u_y_inv, h.target_x, h.target_y,
                             h.distance, h.x_facet);
    h.m_x_facet = __builtin_amdgcn_ballot_w64(h.x_facet != 0);
    /* (x_facet is not pinned with it: the comparison's lane mask serves the next trip's
     * selects as it is -- pinned, it is a 0/1 vector register and a compare per trip) */
    asm volatile("" : "+v"(h.distance));
  }

  /* pin the two quotients above the wait for the density: left alone, the compiler
   * sinks both divides below the branch that consumes the load */
  asm volatile("" : "+v"(h.mfp_to_collision), "+v"(h.dt_to_census));

  if constexpr (Tally::kUniformDensity) {
    /* (of a copy the compiler cannot merge with the tally's question: see WindowCellTallyT::add) */
    unsigned long long m_any_outside = tally.m_outside;
    asm volatile("" : "+s"(m_any_outside));
    if (__builtin_expect(m_any_outside == 0, 1)) {
      return; /* (no lane outside the window: asked of the wave first, it is a scalar test) */
    }
    if (tally.inside()) {
      return; /* the cell entered has the density this history carries */
    }
    /* (of the cell the history IS in: a step taken back at the mesh's edge is back by here) */
    new_density = kStepFirst ? *mesh_element(a.density, (h.celly - a.y_off) * a.nx + (h.cellx - a.x_off))
                             : *mesh_element(a.density, dens_y * a.nx + dens_x); /* (rare, dependent) */
  }
  if (__double_as_longlong(new_density) != __double_as_longlong(h.local_density)) {
    h.local_density = new_density;
    macroscopic_from_density<kChecked>(h);
    if (kCachedReciprocals) {
      refresh_mfp_reciprocal<kChecked>(h);
    }
    if (kCarryTargets) {
      h.dep_rate = deposit_rate(h, a); /* (the number density is one of its factors) */
    }
  }
}

/* census_event, omp3/neutral.c:383-405 */
template <bool kChecked, typename Tally>
__device__ __forceinline__ void census(History& h, const SolveArgs& a, const Tally& tally) {
  const double distance_to_census = h.distance;
  h.x += distance_to_census * h.omega_x;
  h.y += distance_to_census * h.omega_y;
  h.mfp_to_collision -= quotient_of_physical<kChecked>(distance_to_census, h.cell_mfp);
  h.energy_deposition += deposit(h, distance_to_census);
  tally(a, h.cellx, h.celly, h.energy_deposition);
  if (Tally::kFlux) {
    tally.flux(a, h.cellx, h.celly, h.track_length + h.weight * distance_to_census);
    h.track_length = 0.0;
  }
  h.dt_to_census = 0.0;
}

/* census_event for the stream kernel: nothing is pending in the accumulator (every facet
 * flushes it, collisions happen elsewhere), and the deposit is path length x dep_rate */
template <bool kChecked, typename Tally>
__device__ __forceinline__ void census_streamed(History& h, const SolveArgs& a, const Tally& tally) {
  const double distance_to_census = h.distance;
  h.x += distance_to_census * h.omega_x;
  h.y += distance_to_census * h.omega_y;
  h.mfp_to_collision -= quotient_of_physical<kChecked>(distance_to_census, h.cell_mfp);
  tally.scaled(a, h.cellx, h.celly, distance_to_census * h.dep_rate);
  if (Tally::kFlux) {
    tally.flux(a, h.cellx, h.celly, h.track_length + h.weight * distance_to_census);
    h.track_length = 0.0;
  }
  h.dt_to_census = 0.0;
}

}  // namespace neutral
#endif
