"""ctypes mirror of ``include/neutral_hip.h`` (libneutral_hip.so).

The three functions keep the reference's names, argument order and meaning
(``neutral_interface.h:11-36``); ``Simulation`` is a thin convenience that keeps
the device buffers of one problem together, the way ``main.c`` keeps them in
``Mesh``/``SharedData``/``NeutralData``.

There is no CPU fallback: if the HIP library is missing, importing this module
raises.  Device buffers are torch CUDA tensors (PyTorch is used for device
memory, streams and ``torch.distributed`` only); particle arrays are allocated
by the library's own ``inject_particles`` as in the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
# torch must be imported before libneutral_hip.so is loaded: the PyTorch-ROCm
# wheel bundles its own libamdhip64, and one process must hold exactly one HIP
# runtime (loading the system one first makes every later device call fail).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEUTRAL_HIP_LIB selects another build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("NEUTRAL_HIP_LIB") or os.path.join(_HERE, "libneutral_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} has not been built (run __graft_entry__.build() or "
        "`make -C neutral_amd`); neutral_amd has no CPU fallback")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)

VARIANT_OVER_PARTICLE = 0
VARIANT_EVENT_SORTED = 1
VARIANT_TILED = 2

F64_FIELDS = ("x", "y", "omega_x", "omega_y", "energy", "weight", "dt_to_census",
              "mfp_to_collision")
I32_FIELDS = ("cellx", "celly", "dead")


class CrossSection(C.Structure):
    """neutral_data.h:38-43"""
    _fields_ = [("keys", C.c_void_p), ("values", C.c_void_p), ("nentries", C.c_int)]


class Particle(C.Structure):
    """neutral_data.h:45-61 (-DSoA): host struct of device arrays"""
    _fields_ = [(n, C.c_void_p) for n in F64_FIELDS + I32_FIELDS]


class StepStats(C.Structure):
    _fields_ = [("nprocessed", C.c_uint64), ("facets", C.c_uint64),
                ("collisions", C.c_uint64), ("census", C.c_uint64),
                ("kernel_ms", C.c_double),
                ("same_tables", C.c_int), ("variant", C.c_int),
                ("sort_ms", C.c_double), ("stream_ms", C.c_double),
                ("collide_ms", C.c_double), ("stream_facets", C.c_uint64),
                ("stream_census", C.c_uint64), ("suspended", C.c_uint64),
                ("aborted", C.c_uint64), ("stream_passes", C.c_int),
                ("requeued", C.c_uint64), ("collide_passes", C.c_uint64),
                ("host_syncs", C.c_int), ("stream_passes_enqueued", C.c_int),
                ("tile_cells", C.c_int), ("export_ms", C.c_double),
                ("checked_arithmetic", C.c_int), ("attempts", C.c_int),
                ("host_collectives", C.c_int), ("exchange_ranks", C.c_int),
                ("steals", C.c_uint64), ("steals_refused", C.c_uint64),
                ("stream_hops", C.c_uint64), ("stream_overflows", C.c_uint64),
                ("stream_batches", C.c_uint64), ("stream_idle_polls", C.c_uint64),
                ("local_nprocessed", C.c_uint64), ("exchange_ms", C.c_double),
                ("exchange_rounds", C.c_int), ("emigrants", C.c_uint64),
                ("weighted_waves", C.c_uint64),
                ("stream_clock_ghz", C.c_double), ("collide_clock_ghz", C.c_double)]


# every symbol include/neutral_hip.h declares
ABI_SYMBOLS = (
    "solve_transport_2d", "inject_particles", "validate",
    "allocate_data", "allocate_float_data", "allocate_int_data", "allocate_uint64_data",
    "allocate_host_data", "allocate_host_int_data", "deallocate_data",
    "deallocate_int_data", "deallocate_uint64_data", "deallocate_host_data",
    "copy_buffer", "copy_int_buffer", "move_host_buffer_to_device",
    "neutral_hip_device_count", "neutral_hip_set_device", "neutral_hip_set_stream",
    "neutral_hip_set_pid_base", "neutral_hip_get_pid_base", "neutral_hip_set_variant",
    "neutral_hip_set_quiet", "neutral_hip_set_tests_file", "neutral_hip_last_step",
    "neutral_hip_set_arithmetic",
    "neutral_hip_reinject_particles", "neutral_hip_free_particles",
    "neutral_hip_set_lazy_export", "neutral_hip_set_stream_queues", "neutral_hip_sync_particles",
    "neutral_hip_invalidate_particles", "neutral_hip_set_scalar_flux_tally",
    "neutral_hip_comm_start", "neutral_hip_comm_stop", "neutral_hip_comm_rank",
    "neutral_hip_comm_nranks", "neutral_hip_comm_transport", "neutral_hip_comm_rccl_version",
    "neutral_hip_set_auto_shard",
    "neutral_hip_store_count", "neutral_hip_set_decomposition",
    "neutral_hip_clear_decomposition", "neutral_hip_set_source_box", "neutral_hip_store_keys",
    "neutral_hip_comm_allreduce_f64", "neutral_hip_comm_max",
    "neutral_hip_comm_barrier", "neutral_hip_bind_rank_device",
    "neutral_hip_comm_barrier_device", "neutral_hip_comm_selftest",
    "neutral_hip_memcpy_d2h", "neutral_hip_memcpy_h2d", "neutral_hip_memset",
    "neutral_hip_synchronize", "neutral_hip_abi_version",
    "neutral_hip_probe_threefry", "neutral_hip_probe_cs_lookup",
    "neutral_hip_probe_distance_to_facet", "neutral_hip_probe_division", "neutral_hip_probe_scatter",
    "neutral_hip_probe_log",
)

_lib = C.CDLL(LIB_PATH)

_lib.solve_transport_2d.restype = None
_lib.solve_transport_2d.argtypes = [
    C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
    C.c_double, C.c_int, _ip, C.c_void_p, C.POINTER(Particle), C.c_void_p,
    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CrossSection),
    C.POINTER(CrossSection), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
    _u64p, _u64p]
_lib.inject_particles.restype = C.c_size_t
_lib.inject_particles.argtypes = [
    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
    C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_double,
    C.POINTER(C.POINTER(Particle))]
_lib.validate.restype = None
_lib.validate.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p]
_lib.neutral_hip_device_count.restype = C.c_int
_lib.neutral_hip_set_device.restype = C.c_int
_lib.neutral_hip_set_device.argtypes = [C.c_int]
_lib.neutral_hip_set_stream.argtypes = [C.c_void_p]
_lib.neutral_hip_set_pid_base.argtypes = [C.c_uint64]
_lib.neutral_hip_get_pid_base.restype = C.c_uint64
_lib.neutral_hip_set_variant.restype = C.c_int
_lib.neutral_hip_set_variant.argtypes = [C.c_int]
_lib.neutral_hip_set_quiet.argtypes = [C.c_int]
_lib.neutral_hip_set_arithmetic.restype = C.c_int
_lib.neutral_hip_set_arithmetic.argtypes = [C.c_int]
_lib.neutral_hip_set_tests_file.argtypes = [C.c_char_p]
_lib.neutral_hip_last_step.argtypes = [C.POINTER(StepStats)]
_lib.neutral_hip_reinject_particles.restype = None
_lib.neutral_hip_reinject_particles.argtypes = [
    C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
    C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(Particle)]
_lib.neutral_hip_free_particles.argtypes = [C.POINTER(Particle)]
_lib.neutral_hip_set_lazy_export.argtypes = [C.c_int]
_lib.neutral_hip_sync_particles.argtypes = [C.POINTER(Particle)]
_lib.neutral_hip_invalidate_particles.argtypes = [C.POINTER(Particle)]
_lib.neutral_hip_set_scalar_flux_tally.argtypes = [C.c_void_p]
_lib.neutral_hip_comm_start.restype = C.c_int
_lib.neutral_hip_comm_rank.restype = C.c_int
_lib.neutral_hip_comm_nranks.restype = C.c_int
_lib.neutral_hip_comm_transport.restype = C.c_int
_lib.neutral_hip_set_auto_shard.argtypes = [C.c_int]
_lib.neutral_hip_store_count.restype = C.c_int
_lib.neutral_hip_store_count.argtypes = [C.POINTER(Particle)]
_lib.neutral_hip_set_decomposition.restype = C.c_int
_lib.neutral_hip_set_decomposition.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip]
_lib.neutral_hip_set_source_box.argtypes = [C.c_double] * 4
_lib.neutral_hip_store_keys.restype = C.c_void_p
_lib.neutral_hip_store_keys.argtypes = [C.POINTER(Particle)]
_lib.neutral_hip_comm_allreduce_f64.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
if hasattr(_lib, "neutral_hip_comm_rccl_version"):   # (absent from older builds: same-box A/B runs)
    _lib.neutral_hip_comm_rccl_version.restype = C.c_int
_lib.neutral_hip_comm_max.restype = C.c_double
_lib.neutral_hip_comm_max.argtypes = [C.c_double]
_lib.neutral_hip_bind_rank_device.argtypes = [C.c_int]
_lib.neutral_hip_comm_selftest.restype = C.c_int
_lib.neutral_hip_comm_selftest.argtypes = [C.c_int]

COMM_NONE, COMM_RCCL, COMM_HOST = 0, 1, 2
_lib.neutral_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
_lib.neutral_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
_lib.neutral_hip_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
_lib.neutral_hip_abi_version.restype = C.c_int
_lib.neutral_hip_probe_threefry.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
_lib.neutral_hip_probe_cs_lookup.argtypes = [C.POINTER(CrossSection), C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int]
_lib.neutral_hip_probe_division.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
_lib.neutral_hip_probe_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
_lib.neutral_hip_probe_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
_lib.neutral_hip_probe_distance_to_facet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int]


def library() -> C.CDLL:
    return _lib


# ---- the three interface functions, reference names and argument order ---------

def solve_transport_2d(nx, ny, global_nx, global_ny, master_key, pad, x_off, y_off, dt,
                       ntotal_particles, nlocal_particles, neighbours, particles, density,
                       edgex, edgey, edgedx, edgedy, cs_scatter_table, cs_absorb_table,
                       energy_deposition_tally, reduce_array0, reduce_array1,
                       reduce_array2, facet_events, collision_events):
    """neutral_interface.h:11-20.  Pointer arguments are device addresses (ints)
    or ctypes objects; `nlocal_particles`, `facet_events`, `collision_events` are
    ctypes scalars passed by reference like the C int*/uint64_t*."""
    _lib.solve_transport_2d(nx, ny, global_nx, global_ny, master_key, pad, x_off, y_off,
                            dt, ntotal_particles, C.byref(nlocal_particles), neighbours,
                            particles, density, edgex, edgey, edgedx, edgedy,
                            C.byref(cs_scatter_table), C.byref(cs_absorb_table),
                            energy_deposition_tally, reduce_array0, reduce_array1,
                            reduce_array2, C.byref(facet_events), C.byref(collision_events))


def inject_particles(nparticles, global_nx, local_nx, local_ny, pad,
                     local_particle_left_off, local_particle_bottom_off,
                     local_particle_width, local_particle_height, x_off, y_off, dt,
                     edgex, edgey, initial_energy):
    """neutral_interface.h:23-31.  Returns (particles, bytes_allocated), where
    `particles` is the C `Particle*` the library allocated."""
    pp = C.POINTER(Particle)()
    nbytes = _lib.inject_particles(nparticles, global_nx, local_nx, local_ny, pad,
                                   local_particle_left_off, local_particle_bottom_off,
                                   local_particle_width, local_particle_height, x_off,
                                   y_off, dt, edgex, edgey, initial_energy, C.byref(pp))
    return pp, nbytes


def validate(nx, ny, params_filename, rank, energy_tally):
    """neutral_interface.h:35-36 (prints; returns nothing, like the reference)."""
    _lib.validate(nx, ny, params_filename.encode(), rank, energy_tally)


# ---- extensions -------------------------------------------------------------------

def device_count() -> int:
    return _lib.neutral_hip_device_count()


def set_device(device: int) -> None:
    if _lib.neutral_hip_set_device(device) != 0:
        raise RuntimeError(f"hipSetDevice({device}) failed")


def set_stream(stream_handle: int) -> None:
    _lib.neutral_hip_set_stream(C.c_void_p(stream_handle))


def set_pid_base(pid_base: int) -> None:
    _lib.neutral_hip_set_pid_base(pid_base)


def set_variant(variant: int) -> None:
    if _lib.neutral_hip_set_variant(variant) != 0:
        raise ValueError(f"unknown kernel variant {variant}")


ARITH_AUTO, ARITH_CHECKED = 0, 1


def set_arithmetic(mode: int) -> None:
    """ARITH_AUTO: the device picks the fast or the IEEE-checked kernels per step from the
    step's density mesh and tables; ARITH_CHECKED: always the checked ones."""
    if _lib.neutral_hip_set_arithmetic(mode) != 0:
        raise ValueError(f"unknown arithmetic mode {mode}")


def set_quiet(quiet: bool) -> None:
    _lib.neutral_hip_set_quiet(1 if quiet else 0)


def set_tests_file(path: str) -> None:
    _lib.neutral_hip_set_tests_file(path.encode())


def set_lazy_export(lazy: bool) -> None:
    _lib.neutral_hip_set_lazy_export(1 if lazy else 0)


def set_stream_queues(on: bool) -> None:
    """Tiled variant: migrants change tiles inside the stream kernel (include/neutral_hip.h)."""
    if hasattr(_lib, "neutral_hip_set_stream_queues"):
        _lib.neutral_hip_set_stream_queues(1 if on else 0)


def comm_start() -> int:
    """Joins the ranks named by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* and brings up
    the tally exchange on the current device; returns COMM_NONE / COMM_RCCL / COMM_HOST."""
    return _lib.neutral_hip_comm_start()


def last_step() -> StepStats:
    s = StepStats()
    _lib.neutral_hip_last_step(C.byref(s))
    return s


def to_host(device_ptr: int, n: int, dtype) -> np.ndarray:
    out = np.empty(n, dtype=dtype)
    if n:
        _lib.neutral_hip_memcpy_d2h(out.ctypes.data, C.c_void_p(device_ptr), out.nbytes)
    return out


def probe_threefry(counter_pkey_mkey: np.ndarray):
    """rows {counter, pkey, master_key} -> (words [n,2] uint64, rn [n,2] float64)"""
    a = np.ascontiguousarray(counter_pkey_mkey, dtype=np.uint64).reshape(-1, 3)
    n = a.shape[0]
    words = np.zeros((n, 2), dtype=np.uint64)
    rn = np.zeros((n, 2), dtype=np.float64)
    _lib.neutral_hip_probe_threefry(a.ctypes.data, words.ctypes.data, rn.ctypes.data, n)
    return words, rn


def probe_cs_lookup(cs: CrossSection, energies: np.ndarray, use_index: bool = True):
    e = np.ascontiguousarray(energies, dtype=np.float64)
    value = np.zeros(e.size, dtype=np.float64)
    index = np.zeros(e.size, dtype=np.int32)
    _lib.neutral_hip_probe_cs_lookup(C.byref(cs), e.ctypes.data, value.ctypes.data,
                                     index.ctypes.data, e.size, 1 if use_index else 0)
    return value, index


def probe_division(rows: np.ndarray):
    """rows {a, b} -> (a / b, quotient through the kept reciprocal, in-range flag)"""
    a = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((a.shape[0], 2), dtype=np.float64)
    plain = np.zeros(a.shape[0], dtype=np.int32)
    _lib.neutral_hip_probe_division(a.ctypes.data, out.ctypes.data, plain.ctypes.data,
                                    a.shape[0])
    return out[:, 0], out[:, 1], plain.astype(bool)


def probe_scatter(rows: np.ndarray):
    """rows {energy, mu_cm, omega_x, omega_y} -> dict of the scatter's kinematics the fast kernels' way and
    with IEEE divisions and roots (include/neutral_hip.h: neutral_hip_probe_scatter)"""
    a = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((a.shape[0], 10), dtype=np.float64)
    _lib.neutral_hip_probe_scatter(a.ctypes.data, out.ctypes.data, a.shape[0])
    names = ("e_new", "cos_fast", "cos_ieee", "speed_fast", "speed_ieee", "u_x_inv_fast", "u_y_inv_fast",
             "u_x_inv_ieee", "u_y_inv_ieee")
    return {k: out[:, j] for j, k in enumerate(names)}


def probe_log(x: np.ndarray):
    """x -> (the kernels' log of a sample, the device library's log)"""
    a = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros((2 * a.size, 4), dtype=np.float64)
    _lib.neutral_hip_probe_log(a.ctypes.data, out.ctypes.data, a.size)
    return out[:a.size, 0], out[:a.size, 1]


def probe_sqrt(x: np.ndarray):
    """x -> (the kernels' square root, the compiler's sqrt)"""
    a = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros((2 * a.size, 4), dtype=np.float64)
    _lib.neutral_hip_probe_log(a.ctypes.data, out.ctypes.data, a.size)
    return out[:a.size, 2], out[:a.size, 3]


def probe_constant_quotients(x: np.ndarray):
    """x -> (x / PARTICLE_MASS kernels' way, compiler's, x / (MASS_NO+1)^2 kernels' way, compiler's)"""
    a = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros((2 * a.size, 4), dtype=np.float64)
    _lib.neutral_hip_probe_log(a.ctypes.data, out.ctypes.data, a.size)
    q = out[a.size:]
    return q[:, 0], q[:, 1], q[:, 2], q[:, 3]


def probe_distance_to_facet(rows: np.ndarray):
    """rows {x, y, omega_x, omega_y, speed, ex_lo, ex_hi, ey_lo, ey_hi}"""
    a = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 9)
    dist = np.zeros(a.shape[0], dtype=np.float64)
    xf = np.zeros(a.shape[0], dtype=np.int32)
    _lib.neutral_hip_probe_distance_to_facet(a.ctypes.data, dist.ctypes.data,
                                             xf.ctypes.data, a.shape[0])
    return dist, xf


@dataclass
class StepResult:
    nprocessed: int
    facets: int
    collisions: int
    kernel_ms: float
    census: int = 0
    stats: Optional["StepStats"] = None

    @property
    def particle_steps(self) -> int:
        """trips of the event loop (omp3/neutral.c:134): facets + collisions + census"""
        return self.facets + self.collisions + self.census


class Simulation:
    """Device-side state of one problem (or one particle shard of it).

    `shard = (first, count)` makes this process own global particles
    [first, first+count); the RNG key of local particle i is first + i, so any
    partition reproduces the single-GPU histories.
    """

    def __init__(self, problem, cs_keys, cs_values, device: int = 0, shard=None,
                 cs_absorb=None, variant: Optional[int] = None, scalar_flux: bool = False,
                 domain=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("neutral_amd.interface.Simulation needs a GPU")
        self.torch = torch
        self.p = problem
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        set_device(device)
        set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.variant = variant
        if variant is not None:
            set_variant(variant)
        # shard = (first, count): this process owns those global ids (the caller shards);
        # None: all of them -- or, when the rank layer is up (comm_start) with several
        # ranks, the share inject_particles cuts for this rank
        first, count = shard if shard is not None else (0, problem.nparticles)
        self.pid_base, self.n = int(first), int(count)
        self.explicit_shard = shard is not None

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

        # domain = (ranks_x, ranks_y): spatial decomposition over the ranks of the rank
        # layer (comm_start first).  This rank then holds one block of the mesh -- its
        # edges, density and tally -- and whatever particles are inside it.
        self.domain = domain
        self.x_off, self.y_off, self.lnx, self.lny = problem.x_off, problem.y_off, problem.nx, \
            problem.ny
        if domain is not None:
            xo, yo, lx, ly = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            if _lib.neutral_hip_set_decomposition(domain[0], domain[1], problem.nx, problem.ny,
                                                  C.byref(xo), C.byref(yo), C.byref(lx),
                                                  C.byref(ly)) != 0:
                raise ValueError(f"decomposition {domain} does not fit the ranks or the mesh")
            self.x_off, self.y_off, self.lnx, self.lny = xo.value, yo.value, lx.value, ly.value
            _lib.neutral_hip_set_source_box(problem.local_particle_left_off,
                                            problem.local_particle_bottom_off,
                                            problem.local_particle_width,
                                            problem.local_particle_height)
        bx = slice(self.x_off, self.x_off + self.lnx + 1)
        by = slice(self.y_off, self.y_off + self.lny + 1)
        self.edgex, self.edgey = dev(problem.edgex[bx]), dev(problem.edgey[by])
        self.edgedx, self.edgedy = dev(problem.edgedx[bx]), dev(problem.edgedy[by])
        block = np.asarray(problem.density).reshape(problem.ny, problem.nx)[
            self.y_off:self.y_off + self.lny, self.x_off:self.x_off + self.lnx]
        self.density = dev(block.ravel())
        self.tally = torch.zeros(self.lnx * self.lny, dtype=torch.float64, device=self.device)
        # scalar-flux tally (include/neutral_hip.h): optional second mesh
        self.flux = torch.zeros(self.lnx * self.lny, dtype=torch.float64,
                                device=self.device) if scalar_flux else None
        self._sk, self._sv = dev(cs_keys), dev(cs_values)
        if cs_absorb is None:
            # two separate device copies, as neutral_data.c:176-177 reads both files
            self._ak, self._av = dev(cs_keys), dev(cs_values)
        else:
            self._ak, self._av = dev(cs_absorb[0]), dev(cs_absorb[1])
        self.cs_scatter = CrossSection(self._sk.data_ptr(), self._sv.data_ptr(), len(cs_keys))
        self.cs_absorb = CrossSection(self._ak.data_ptr(), self._av.data_ptr(),
                                      self._ak.numel())
        self.particles = None
        self.nlocal = C.c_int(self.n)
        self.bytes_allocated = 0

    def _inject_args(self):
        p = self.p
        return (self.lnx, self.lny, p.pad, p.local_particle_left_off, p.local_particle_bottom_off,
                p.local_particle_width, p.local_particle_height, self.x_off, self.y_off, p.dt,
                self.edgex.data_ptr(), self.edgey.data_ptr(), p.initial_energy)

    def inject(self):
        """inject_particles on first use, a state reset (no allocation) afterwards."""
        set_pid_base(self.pid_base)
        _lib.neutral_hip_set_auto_shard(0 if self.explicit_shard else 1)
        if self.particles is None:
            p = self.p
            self.particles, self.bytes_allocated = inject_particles(
                self.n, p.nx, *self._inject_args())
            local = _lib.neutral_hip_store_count(self.particles)
            if local >= 0:  # the library cut this rank's share (or block)
                self.n = local
                self.pid_base = int(_lib.neutral_hip_get_pid_base())
                self.nlocal = C.c_int(self.n)
        else:
            _lib.neutral_hip_reinject_particles(self.n, *self._inject_args(), self.particles)
            if self.domain is not None:
                self.n = _lib.neutral_hip_store_count(self.particles)
                self.nlocal = C.c_int(self.n)

    def step(self, master_key: int) -> StepResult:
        p = self.p
        # variant and pid base are process-global in the library: re-apply this
        # simulation's own before every call (several Simulations may be alive)
        set_pid_base(self.pid_base)
        if self.variant is not None:
            set_variant(self.variant)
        facets, collisions = C.c_uint64(0), C.c_uint64(0)
        _lib.neutral_hip_set_scalar_flux_tally(
            C.c_void_p(self.flux.data_ptr()) if self.flux is not None else None)
        solve_transport_2d(
            self.lnx - 2 * p.pad, self.lny - 2 * p.pad, p.nx, p.ny, master_key, p.pad, self.x_off,
            self.y_off, p.dt, p.nparticles, self.nlocal, None, self.particles,
            self.density.data_ptr(), self.edgex.data_ptr(), self.edgey.data_ptr(),
            self.edgedx.data_ptr(), self.edgedy.data_ptr(), self.cs_scatter,
            self.cs_absorb, self.tally.data_ptr(), None, None, None, facets, collisions)
        s = last_step()
        if self.domain is not None:
            self.n = self.nlocal.value  # histories crossed between the ranks' blocks
        return StepResult(int(s.nprocessed), facets.value, collisions.value, s.kernel_ms,
                          int(s.census), s)

    def particle_keys(self) -> np.ndarray:
        """Global ids of the particles of a decomposed store, in array order."""
        ptr = _lib.neutral_hip_store_keys(self.particles)
        return to_host(ptr, self.n, np.uint32)

    def particle_arrays(self):
        """Host copies of the SoA particle store."""
        _lib.neutral_hip_sync_particles(self.particles)
        pc = self.particles.contents
        out = {}
        for f in F64_FIELDS:
            out[f] = to_host(getattr(pc, f), self.n, np.float64)
        for f in I32_FIELDS:
            out[f] = to_host(getattr(pc, f), self.n, np.int32)
        return out

    def tally_host(self) -> np.ndarray:
        return self.tally.cpu().numpy()

    def zero_tally(self):
        self.tally.zero_()

    def validate(self, params_filename: Optional[str] = None):
        validate(self.lnx, self.lny, params_filename or self.p.deck, _lib.neutral_hip_comm_rank(),
                 self.tally.data_ptr())

    def close(self):
        if self.particles is not None:
            _lib.neutral_hip_free_particles(self.particles)
            self.particles = None
