"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the reference's
own Threefry (oracle/_ref/libref_threefry.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (neutral_amd/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REF_THREEFRY_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_threefry.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)


class OrcCrossSection(C.Structure):
    _fields_ = [("keys", _dp), ("values", _dp), ("nentries", C.c_int)]


class OrcParticles(C.Structure):
    _fields_ = [(n, _dp) for n in ("x", "y", "omega_x", "omega_y", "energy", "weight",
                                   "dt_to_census", "mfp_to_collision")] + \
               [(n, _ip) for n in ("cellx", "celly", "dead")]


F64_FIELDS = ("x", "y", "omega_x", "omega_y", "energy", "weight", "dt_to_census",
              "mfp_to_collision")
I32_FIELDS = ("cellx", "celly", "dead")

_lib: Optional[C.CDLL] = None
_ref: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_LIB)
        L.orc_threefry2x64_20.argtypes = [C.c_uint64] * 4 + [_u64p, _u64p]
        L.orc_generate_random_numbers.argtypes = [C.c_uint64] * 3 + [_dp, _dp]
        L.orc_microscopic_cs_for_energy.restype = C.c_double
        L.orc_microscopic_cs_for_energy.argtypes = [C.POINTER(OrcCrossSection), C.c_double, _ip]
        L.orc_calc_distance_to_facet.argtypes = [
            C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
            C.c_double, C.c_int, C.c_int, _dp, _ip, _dp, _dp]
        L.orc_calculate_energy_deposition.restype = C.c_double
        L.orc_calculate_energy_deposition.argtypes = [C.c_double] * 6
        L.orc_inject_particles.argtypes = [
            C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
            C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_double,
            C.POINTER(OrcParticles)]
        L.orc_set_scalar_flux_tally.argtypes = [_dp]
        L.orc_solve_transport_2d.restype = C.c_uint64
        L.orc_solve_transport_2d.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
            C.c_double, C.c_int, C.c_int, C.c_uint64, C.POINTER(OrcParticles), _dp, _dp,
            _dp, C.POINTER(OrcCrossSection), C.POINTER(OrcCrossSection), _dp, _u64p, _u64p]
        L.orc_last_census.restype = C.c_uint64
        L.orc_sum_tally.restype = C.c_double
        L.orc_sum_tally.argtypes = [C.c_int, C.c_int, _dp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def have_ref_threefry() -> bool:
    return os.path.exists(REF_THREEFRY_LIB)


def ref_lib() -> C.CDLL:
    global _ref
    if _ref is None:
        R = C.CDLL(REF_THREEFRY_LIB)
        R.ref_threefry2x64.argtypes = [C.c_uint64] * 4 + [_u64p, _u64p]
        R.ref_threefry2x64_stream.argtypes = [C.c_uint64] * 3 + [C.c_int, _u64p]
        _ref = R
    return _ref


def threefry(c0: int, c1: int, k0: int, k1: int):
    a, b = C.c_uint64(), C.c_uint64()
    lib().orc_threefry2x64_20(c0, c1, k0, k1, C.byref(a), C.byref(b))
    return a.value, b.value


def ref_threefry(c0: int, c1: int, k0: int, k1: int):
    a, b = C.c_uint64(), C.c_uint64()
    ref_lib().ref_threefry2x64(c0, c1, k0, k1, C.byref(a), C.byref(b))
    return a.value, b.value


def generate_random_numbers(pkey: int, master_key: int, counter: int):
    a, b = C.c_double(), C.c_double()
    lib().orc_generate_random_numbers(pkey, master_key, counter, C.byref(a), C.byref(b))
    return a.value, b.value


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


class CsTable:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, keys: np.ndarray, values: np.ndarray):
        self.keys = np.ascontiguousarray(keys, dtype=np.float64)
        self.values = np.ascontiguousarray(values, dtype=np.float64)
        self.c = OrcCrossSection(_ptr(self.keys, _dp), _ptr(self.values, _dp), len(self.keys))

    def lookup(self, energy: float):
        idx = C.c_int(-1)
        v = lib().orc_microscopic_cs_for_energy(C.byref(self.c), energy, C.byref(idx))
        return v, idx.value


class ParticleStore:
    """SoA particle arrays in host memory (neutral_data.h:48-61)."""

    def __init__(self, n: int):
        self.n = n
        for f in F64_FIELDS:
            setattr(self, f, np.zeros(n, dtype=np.float64))
        for f in I32_FIELDS:
            setattr(self, f, np.zeros(n, dtype=np.int32))
        self.c = OrcParticles(*[_ptr(getattr(self, f), _dp) for f in F64_FIELDS],
                              *[_ptr(getattr(self, f), _ip) for f in I32_FIELDS])

    def as_dict(self):
        return {f: getattr(self, f) for f in F64_FIELDS + I32_FIELDS}


@dataclass
class StepResult:
    nprocessed: int
    facets: int
    collisions: int
    census: int = 0

    @property
    def particle_steps(self) -> int:
        return self.facets + self.collisions + self.census


class OracleRun:
    """Drives the oracle over a neutral_amd.host.Problem.

    `shard = (first, count)` restricts the run to global particle ids
    [first, first+count): the particle-shard extension.
    """

    def __init__(self, problem, cs_keys, cs_values, shard=None, cs_absorb=None,
                 scalar_flux=False):
        self.p = problem
        # scalar-flux tally (path-length estimator, oracle/neutral_oracle.c): optional
        self.flux = np.zeros(problem.nx * problem.ny, dtype=np.float64) if scalar_flux else None
        self.cs_scatter = CsTable(cs_keys, cs_values)
        self.cs_absorb = CsTable(*cs_absorb) if cs_absorb is not None else \
            CsTable(cs_keys, cs_values)
        first, count = shard if shard is not None else (0, problem.nparticles)
        self.pid_base = int(first)
        self.n = int(count)
        self.particles = ParticleStore(self.n)
        self.tally = np.zeros(problem.nx * problem.ny, dtype=np.float64)
        self.edgex = np.ascontiguousarray(problem.edgex)
        self.edgey = np.ascontiguousarray(problem.edgey)
        self.density = np.ascontiguousarray(problem.density)

    def inject(self):
        p = self.p
        lib().orc_inject_particles(
            self.n, self.pid_base, p.nx, p.ny, p.pad, p.local_particle_left_off,
            p.local_particle_bottom_off, p.local_particle_width, p.local_particle_height,
            p.x_off, p.y_off, p.dt, _ptr(self.edgex, _dp), _ptr(self.edgey, _dp),
            p.initial_energy, C.byref(self.particles.c))

    def step(self, master_key: int) -> StepResult:
        p = self.p
        facets, collisions = C.c_uint64(0), C.c_uint64(0)
        lib().orc_set_scalar_flux_tally(_ptr(self.flux, _dp) if self.flux is not None else None)
        nproc = lib().orc_solve_transport_2d(
            p.nx - 2 * p.pad, p.ny - 2 * p.pad, p.nx, p.ny, master_key, p.pad, p.x_off,
            p.y_off, p.dt, p.nparticles, self.n, self.pid_base, C.byref(self.particles.c),
            _ptr(self.density, _dp), _ptr(self.edgex, _dp), _ptr(self.edgey, _dp),
            C.byref(self.cs_scatter.c), C.byref(self.cs_absorb.c), _ptr(self.tally, _dp),
            C.byref(facets), C.byref(collisions))
        lib().orc_set_scalar_flux_tally(None)
        return StepResult(int(nproc), facets.value, collisions.value,
                          int(lib().orc_last_census()))

    def tally_sum(self) -> float:
        p = self.p
        return lib().orc_sum_tally(p.nx, p.ny, _ptr(self.tally, _dp))
