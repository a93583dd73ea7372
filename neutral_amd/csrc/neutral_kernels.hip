/*
 * neutral_kernels.hip -- hand-written gfx950 kernels of the over-particle
 * transport path.
 *
 *   K0 inject_kernel       initial particle state        (omp3/neutral.c:560-630)
 *   K1 history_kernel      one lane = one history         (omp3/neutral.c:43-206)
 *   tables_equal_kernel    are the two cs tables the same data?
 *
 * Execution model: one work-item per particle, 64-wide wavefronts, 256-thread
 * workgroups; the grid has nparticles/256 workgroups (>> 256 CUs for every
 * BASELINE configuration), which the dispatcher balances dynamically across
 * the 8 XCDs -- histories vary in length by orders of magnitude, so a static
 * blockIdx->tile map would only hurt here.  Particle state is read once
 * (coalesced SoA), lives in VGPRs for the whole timestep and is written once.
 * The tally mesh, density, edges and cross-section tables (<= ~11 MB together)
 * are shared by all workgroups and stay L2/Infinity-Cache resident; tallies go
 * to the mesh with native f64 atomics (global_atomic_add_f64).  No MFMA: there
 * is no dense contraction anywhere on this path.
 */
#include "neutral_kernels.h"

#include "neutral_device.h"

namespace neutral {

constexpr int kBlock = 256;

/* ---- small wave utilities -------------------------------------------------- */

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += __shfl_down(v, off, 64);
  }
  return v; /* valid in lane 0 */
}

/* ---- K0: injection --------------------------------------------------------- */

/* Cell of coordinate c in a monotone edge array: the first ii in [0, n) with
 * edge[ii] <= c < edge[ii+1], or 0 when there is none -- what the linear scan
 * at omp3/neutral.c:590-603 returns, found by bisection. */
__device__ __forceinline__ int find_cell(const double* __restrict__ edge, int n, double c) {
  if (!(c >= edge[0]) || !(c < edge[n])) {
    return 0;
  }
  int lo = 0;
  int hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (c < edge[mid]) {
      hi = mid;
    } else {
      lo = mid;
    }
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void inject_kernel(InjectArgs a) {
  const int kk = blockIdx.x * kBlock + threadIdx.x;
  if (kk >= a.nparticles) {
    return;
  }
  const uint64_t pkey = a.pid_base + (uint64_t)kk;

  double rn0, rn1;
  generate_random_numbers(pkey, 0, 0, rn0, rn1); /* omp3/neutral.c:581 */
  const double px = a.left_off + rn0 * a.width;
  const double py = a.bottom_off + rn1 * a.height;

  const int cellx = a.x_off + find_cell(a.edgex + a.pad, a.local_nx, px);
  const int celly = a.y_off + find_cell(a.edgey + a.pad, a.local_ny, py);

  generate_random_numbers(pkey, 0, 1, rn0, rn1); /* omp3/neutral.c:611 */
  const double theta = 2.0 * M_PI * rn0;
  double s, c;
  sincos(theta, &s, &c);

  a.p.x[kk] = px;
  a.p.y[kk] = py;
  a.p.cellx[kk] = cellx;
  a.p.celly[kk] = celly;
  a.p.omega_x[kk] = c;
  a.p.omega_y[kk] = s;
  a.p.energy[kk] = a.initial_energy;
  a.p.weight[kk] = 1.0;
  a.p.dt_to_census[kk] = a.dt;
  a.p.mfp_to_collision[kk] = 0.0;
  a.p.dead[kk] = 0;
}

/* ---- K1: over-particle history kernel -------------------------------------- */

/* both microscopic cross sections for one energy */
template <bool kSameTables>
__device__ __forceinline__ void lookup_cs(const SolveArgs& a, double energy,
                                          double& micro_scatter, double& micro_absorb) {
  const int is = cs_bracket(a.scatter_keys, a.scatter_n, energy);
  micro_scatter = cs_interpolate(a.scatter_keys, a.scatter_values, is, energy);
  if (kSameTables) {
    micro_absorb = micro_scatter;
  } else {
    const int ia = cs_bracket(a.absorb_keys, a.absorb_n, energy);
    micro_absorb = cs_interpolate(a.absorb_keys, a.absorb_values, ia, energy);
  }
}

/* omp3/neutral.c:408-420 */
__device__ __forceinline__ void update_tallies(const SolveArgs& a, int pcellx, int pcelly,
                                               double energy_deposition) {
  const int cellx = pcellx - a.x_off;
  const int celly = pcelly - a.y_off;
  unsafeAtomicAdd(&a.tally[celly * a.nx + cellx],
                  energy_deposition * a.inv_ntotal_particles);
}

template <bool kSameTables>
__global__ __launch_bounds__(kBlock) void history_kernel(SolveArgs a) {
  const int pid = blockIdx.x * kBlock + threadIdx.x;

  unsigned nfacets = 0;
  unsigned ncollisions = 0;
  unsigned nprocessed = 0;
  unsigned ncensus = 0;

  if (pid < a.nparticles && !a.p.dead[pid]) { /* omp3/neutral.c:91-93 */
    nprocessed = 1;
    const uint64_t pkey = a.pid_base + (uint64_t)pid; /* omp3/neutral.c:89 */

    double px = a.p.x[pid];
    double py = a.p.y[pid];
    double omega_x = a.p.omega_x[pid];
    double omega_y = a.p.omega_y[pid];
    double energy = a.p.energy[pid];
    double weight = a.p.weight[pid];
    int pcellx = a.p.cellx[pid];
    int pcelly = a.p.celly[pid];
    int dead = 0;

    /* prologue, omp3/neutral.c:103-131 */
    double local_density =
        a.density[(pcelly - a.y_off + a.pad) * (a.nx + 2 * a.pad) + (pcellx - a.x_off + a.pad)];
    double micro_s, micro_a;
    lookup_cs<kSameTables>(a, energy, micro_s, micro_a);
    double number_density = (local_density * kAvogadros / kMolarMass);
    double macro_s = number_density * micro_s * kBarns;
    double macro_a = number_density * micro_a * kBarns;
    double speed = speed_of(energy);
    double energy_deposition = 0.0;

    uint64_t counter = 0;
    double rn0, rn1;
    double dt_to_census = a.dt; /* initial == 1: omp3/neutral.c:35-36,127-128 */
    generate_random_numbers(pkey, a.master_key, counter++, rn0, rn1);
    double mfp_to_collision = -log(rn0) / macro_s;

    /* event loop, omp3/neutral.c:134-197 */
    while (dt_to_census > 0.0) {
      const double cell_mfp = 1.0 / (macro_s + macro_a);

      const int ex = pcellx - a.x_off + a.pad;
      const int ey = pcelly - a.y_off + a.pad;
      double distance_to_facet;
      int x_facet;
      calc_distance_to_facet(px, py, omega_x, omega_y, speed, a.edgex[ex], a.edgex[ex + 1],
                             a.edgey[ey], a.edgey[ey + 1], distance_to_facet, x_facet);

      const double distance_to_collision = mfp_to_collision * cell_mfp;
      const double distance_to_census = speed * dt_to_census;

      if (distance_to_collision < distance_to_facet &&
          distance_to_collision < distance_to_census) {
        /* ---- collision, omp3/neutral.c:209-300 ---- */
        ncollisions++;
        energy_deposition += calculate_energy_deposition(
            energy, weight, distance_to_collision, number_density, micro_a, micro_s + micro_a);
        px += distance_to_collision * omega_x;
        py += distance_to_collision * omega_y;

        const double p_absorb = macro_a / (macro_s + macro_a);
        double rc0, rc1;
        generate_random_numbers(pkey, a.master_key, counter++, rc0, rc1);

        if (rc0 < p_absorb) {
          weight *= (1.0 - p_absorb);
          if (energy < kMinEnergyOfInterest) {
            dead = 1;
            update_tallies(a, pcellx, pcelly, energy_deposition);
            energy_deposition = 0.0;
            break;
          }
        } else {
          const double mu_cm = 1.0 - 2.0 * rc1;
          const double e_new = energy * (kMassNo * kMassNo + 2.0 * kMassNo * mu_cm + 1.0) /
                               ((kMassNo + 1.0) * (kMassNo + 1.0));
          const double cos_theta = 0.5 * ((kMassNo + 1.0) * sqrt(e_new / energy) -
                                          (kMassNo - 1.0) * sqrt(energy / e_new));
          const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
          const double omega_x_new = (omega_x * cos_theta - omega_y * sin_theta);
          const double omega_y_new = (omega_x * sin_theta + omega_y * cos_theta);
          omega_x = omega_x_new;
          omega_y = omega_y_new;
          energy = e_new;
        }

        lookup_cs<kSameTables>(a, energy, micro_s, micro_a);
        number_density = (local_density * kAvogadros / kMolarMass);
        macro_s = number_density * micro_s * kBarns;
        macro_a = number_density * micro_a * kBarns;

        generate_random_numbers(pkey, a.master_key, counter++, rn0, rn1);
        mfp_to_collision = -log(rn0) / macro_s;
        dt_to_census -= distance_to_collision / speed;
        speed = speed_of(energy);
      } else if (distance_to_facet < distance_to_census) {
        /* ---- facet, omp3/neutral.c:303-380 ---- */
        nfacets++;
        mfp_to_collision -= (distance_to_facet / cell_mfp);
        dt_to_census -= (distance_to_facet / speed);
        energy_deposition += calculate_energy_deposition(
            energy, weight, distance_to_facet, number_density, micro_a, micro_s + micro_a);
        update_tallies(a, pcellx, pcelly, energy_deposition);
        energy_deposition = 0.0;

        px += distance_to_facet * omega_x;
        py += distance_to_facet * omega_y;

        if (x_facet) {
          if (omega_x > 0.0) {
            if (pcellx >= (a.global_nx - 1)) {
              omega_x = -omega_x;
            } else {
              pcellx++;
            }
          } else if (omega_x < 0.0) {
            if (pcellx <= 0) {
              omega_x = -omega_x;
            } else {
              pcellx--;
            }
          }
        } else {
          if (omega_y > 0.0) {
            if (pcelly >= (a.global_ny - 1)) {
              omega_y = -omega_y;
            } else {
              pcelly++;
            }
          } else if (omega_y < 0.0) {
            if (pcelly <= 0) {
              omega_y = -omega_y;
            } else {
              pcelly--;
            }
          }
        }

        local_density = a.density[(pcelly - a.y_off) * a.nx + (pcellx - a.x_off)];
        number_density = (local_density * kAvogadros / kMolarMass);
        macro_s = number_density * micro_s * kBarns;
        macro_a = number_density * micro_a * kBarns;
      } else {
        /* ---- census, omp3/neutral.c:383-405 ---- */
        px += distance_to_census * omega_x;
        py += distance_to_census * omega_y;
        mfp_to_collision -= (distance_to_census / cell_mfp);
        energy_deposition += calculate_energy_deposition(
            energy, weight, distance_to_census, number_density, micro_a, micro_s + micro_a);
        update_tallies(a, pcellx, pcelly, energy_deposition);
        dt_to_census = 0.0;
        ncensus = 1;
        break;
      }
    }

    a.p.x[pid] = px;
    a.p.y[pid] = py;
    a.p.omega_x[pid] = omega_x;
    a.p.omega_y[pid] = omega_y;
    a.p.energy[pid] = energy;
    a.p.weight[pid] = weight;
    a.p.dt_to_census[pid] = dt_to_census;
    a.p.mfp_to_collision[pid] = mfp_to_collision;
    a.p.cellx[pid] = pcellx;
    a.p.celly[pid] = pcelly;
    a.p.dead[pid] = dead;
  }

  /* event counters: wave reduction, one atomic per wave and counter
   * (the cuda analog's block tree reduction + host finish, cuda/neutral.k:475-493) */
  const unsigned wf = wave_sum_u32(nfacets);
  const unsigned wc = wave_sum_u32(ncollisions);
  const unsigned wp = wave_sum_u32(nprocessed);
  const unsigned wz = wave_sum_u32(ncensus);
  if ((threadIdx.x & 63) == 0) {
    if (wz) atomicAdd(&a.counters->ncensus, (unsigned long long)wz);
    if (wp) atomicAdd(&a.counters->nprocessed, (unsigned long long)wp);
    if (wf) atomicAdd(&a.counters->nfacets, (unsigned long long)wf);
    if (wc) atomicAdd(&a.counters->ncollisions, (unsigned long long)wc);
  }
}

/* ---- table comparison ------------------------------------------------------ */

__global__ __launch_bounds__(kBlock) void tables_equal_kernel(const double* ka, const double* va,
                                                              const double* kb, const double* vb,
                                                              int n, int* flag) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* bit comparison: NaNs and signed zeros must not compare "equal enough" */
    const bool same = (__double_as_longlong(ka[i]) == __double_as_longlong(kb[i])) &&
                      (__double_as_longlong(va[i]) == __double_as_longlong(vb[i]));
    if (!same) {
      atomicAnd(flag, 0);
    }
  }
}

/* ---- probes: unit-level access to the device building blocks (for KATs) ----- */

__global__ __launch_bounds__(kBlock) void probe_threefry_kernel(const uint64_t* in, uint64_t* out,
                                                                 double* rn, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {counter, pkey, master_key} per row */
    uint64_t r0, r1;
    threefry2x64_20(in[3 * i], in[3 * i + 1], in[3 * i + 2], r0, r1);
    out[2 * i] = r0;
    out[2 * i + 1] = r1;
    double a, b;
    generate_random_numbers(in[3 * i + 1], in[3 * i + 2], in[3 * i], a, b);
    rn[2 * i] = a;
    rn[2 * i + 1] = b;
  }
}

__global__ __launch_bounds__(kBlock) void probe_cs_kernel(const double* keys, const double* values,
                                                           int nentries, const double* energy,
                                                           double* value, int* index, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const int ind = cs_bracket(keys, nentries, energy[i]);
    index[i] = ind;
    value[i] = cs_interpolate(keys, values, ind, energy[i]);
  }
}

__global__ __launch_bounds__(kBlock) void probe_facet_kernel(const double* in, double* dist,
                                                              int* x_facet, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {x, y, omega_x, omega_y, speed, ex_lo, ex_hi, ey_lo, ey_hi} per row */
    const double* r = in + 9 * i;
    double d;
    int xf;
    calc_distance_to_facet(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], d, xf);
    dist[i] = d;
    x_facet[i] = xf;
  }
}

/* ---- launchers -------------------------------------------------------------- */

hipError_t launch_probe_threefry(const uint64_t* in, uint64_t* out, double* rn, int n,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(probe_threefry_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     stream, in, out, rn, n);
  return hipGetLastError();
}

hipError_t launch_probe_cs(const double* keys, const double* values, int nentries,
                           const double* energy, double* value, int* index, int n,
                           hipStream_t stream) {
  hipLaunchKernelGGL(probe_cs_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     keys, values, nentries, energy, value, index, n);
  return hipGetLastError();
}

hipError_t launch_probe_facet(const double* in, double* dist, int* x_facet, int n,
                              hipStream_t stream) {
  hipLaunchKernelGGL(probe_facet_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     in, dist, x_facet, n);
  return hipGetLastError();
}

hipError_t launch_inject(const InjectArgs& a, hipStream_t stream) {
  if (a.nparticles <= 0) {
    return hipSuccess;
  }
  const int grid = (a.nparticles + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(inject_kernel, dim3(grid), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_solve(const SolveArgs& a, int variant, hipStream_t stream) {
  if (a.nparticles <= 0) {
    return hipSuccess;
  }
  (void)variant; /* K2 (event-sorted) joins here */
  const int grid = (a.nparticles + kBlock - 1) / kBlock;
  if (a.same_tables) {
    hipLaunchKernelGGL(history_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, a);
  } else {
    hipLaunchKernelGGL(history_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_tables_equal(const double* ka, const double* va, const double* kb,
                               const double* vb, int n, int* d_flag, hipStream_t stream) {
  const int grid = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(tables_equal_kernel, dim3(grid), dim3(kBlock), 0, stream, ka, va, kb, vb, n,
                     d_flag);
  return hipGetLastError();
}

}  // namespace neutral
