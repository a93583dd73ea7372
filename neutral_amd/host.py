"""ctypes mirror of the plain-C host layer (``neutral_amd/host/*.c``).

The host layer is the product's equivalent of the parent ``arch`` project's
services that the reference driver uses (deck reader, mesh, per-cell density)
plus ``neutral_problem.c``, the restatement of what ``neutral_data.c`` computes
before the first ``solve_transport_2d`` call.  Python only *binds* it: all
parsing and set-up arithmetic runs in C, so bench.py, the tests and the C driver
see identical inputs.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libneutral_host.so")

NNEIGHBOURS = 6
MAX_KEYS = 40
MAX_STR_LEN = 1024

_dp = C.POINTER(C.c_double)


class Mesh(C.Structure):
    """host/mesh.h"""
    _fields_ = [
        ("global_nx", C.c_int), ("global_ny", C.c_int),
        ("local_nx", C.c_int), ("local_ny", C.c_int),
        ("pad", C.c_int), ("x_off", C.c_int), ("y_off", C.c_int),
        ("width", C.c_double), ("height", C.c_double),
        ("dt", C.c_double), ("sim_end", C.c_double), ("niters", C.c_int),
        ("rank", C.c_int), ("nranks", C.c_int), ("ndims", C.c_int),
        ("neighbours", C.c_int * NNEIGHBOURS),
        ("edgex", _dp), ("edgey", _dp), ("edgedx", _dp), ("edgedy", _dp),
        ("celldx", _dp), ("celldy", _dp),
    ]


class SharedData(C.Structure):
    """host/shared_data.h"""
    _fields_ = [("density", _dp), ("energy", _dp)]


class NeutralSource(C.Structure):
    """host/neutral_problem.h"""
    _fields_ = [
        ("nparticles", C.c_int), ("initial_energy", C.c_double),
        ("source_xpos", C.c_double), ("source_ypos", C.c_double),
        ("source_width", C.c_double), ("source_height", C.c_double),
        ("local_particle_left_off", C.c_double),
        ("local_particle_bottom_off", C.c_double),
        ("local_particle_width", C.c_double),
        ("local_particle_height", C.c_double),
        ("nlocal_particles", C.c_int),
    ]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C neutral_amd`) first")
        L = C.CDLL(LIB_PATH)
        L.get_int_parameter.restype = C.c_int
        L.get_int_parameter.argtypes = [C.c_char_p, C.c_char_p]
        L.get_double_parameter.restype = C.c_double
        L.get_double_parameter.argtypes = [C.c_char_p, C.c_char_p]
        L.get_key_value_parameter.restype = C.c_int
        L.get_key_value_parameter.argtypes = [
            C.c_char_p, C.c_char_p, C.c_char_p, _dp, C.POINTER(C.c_int)]
        L.within_tolerance.restype = C.c_int
        L.within_tolerance.argtypes = [C.c_double, C.c_double, C.c_double]
        L.initialise_comms.argtypes = [C.POINTER(Mesh)]
        L.initialise_mesh_2d.argtypes = [C.POINTER(Mesh)]
        L.initialise_shared_data_2d.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_char_p,
            _dp, _dp, C.POINTER(SharedData)]
        L.neutral_source_from_deck.argtypes = [
            C.c_char_p, C.c_double, C.c_double, C.c_double, C.c_double,
            C.c_double, C.c_double, C.POINTER(NeutralSource)]
        L.deallocate_data.argtypes = [_dp]
        L.neutral_cs_file_entries.restype = C.c_int
        L.neutral_cs_file_entries.argtypes = [C.c_char_p]
        L.neutral_read_cs_file.restype = C.c_int
        L.neutral_read_cs_file.argtypes = [C.c_char_p, C.c_int, _dp, _dp]
        _lib = L
    return _lib


def get_int_parameter(name: str, filename: str) -> int:
    return lib().get_int_parameter(name.encode(), filename.encode())


def get_double_parameter(name: str, filename: str) -> float:
    return lib().get_double_parameter(name.encode(), filename.encode())


def get_key_value_parameter(specifier: str, filename: str):
    """Returns {key: value} (in file order) or None when the entry is absent."""
    keys = C.create_string_buffer(MAX_KEYS * MAX_STR_LEN)
    values = (C.c_double * MAX_KEYS)()
    n = C.c_int(0)
    ok = lib().get_key_value_parameter(specifier.encode(), filename.encode(),
                                       keys, values, C.byref(n))
    if not ok:
        return None
    out = {}
    for i in range(n.value):
        k = keys.raw[i * MAX_STR_LEN:(i + 1) * MAX_STR_LEN].split(b"\0", 1)[0]
        out[k.decode()] = values[i]
    return out


def within_tolerance(expected: float, result: float, tol: float) -> bool:
    return bool(lib().within_tolerance(expected, result, tol))


def read_cs_file(filename: str):
    """(keys, values) float64 arrays parsed by the C reader."""
    n = lib().neutral_cs_file_entries(filename.encode())
    if n < 0:
        raise FileNotFoundError(filename)
    keys = np.zeros(n, dtype=np.float64)
    values = np.zeros(n, dtype=np.float64)
    m = lib().neutral_read_cs_file(filename.encode(), n,
                                   keys.ctypes.data_as(_dp), values.ctypes.data_as(_dp))
    return keys[:m].copy(), values[:m].copy()


@dataclass
class Problem:
    """Everything the three interface functions need, in host memory."""
    deck: str
    nx: int
    ny: int
    pad: int
    x_off: int
    y_off: int
    width: float
    height: float
    dt: float
    niters: int
    edgex: np.ndarray
    edgey: np.ndarray
    edgedx: np.ndarray
    edgedy: np.ndarray
    density: np.ndarray            # ny * nx, row-major (celly * nx + cellx)
    nparticles: int                # global particle count (deck value)
    nlocal_particles: int
    initial_energy: float
    local_particle_left_off: float
    local_particle_bottom_off: float
    local_particle_width: float
    local_particle_height: float
    neighbours: np.ndarray = field(default_factory=lambda: np.full(6, -1, np.int32))


def _take(ptr, n: int) -> np.ndarray:
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def setup_problem(deck_path: str, width: float = 1.0, height: float = 1.0) -> Problem:
    """main.c:26-72 up to (not including) particle injection, on host memory."""
    L = lib()
    f = deck_path.encode()
    mesh = Mesh()
    mesh.global_nx = L.get_int_parameter(b"nx", f)
    mesh.global_ny = L.get_int_parameter(b"ny", f)
    mesh.pad = 0
    mesh.local_nx = mesh.global_nx + 2 * mesh.pad
    mesh.local_ny = mesh.global_ny + 2 * mesh.pad
    mesh.width = width
    mesh.height = height
    mesh.dt = L.get_double_parameter(b"dt", f)
    mesh.niters = L.get_int_parameter(b"iterations", f)
    mesh.rank = 0
    mesh.nranks = 1
    mesh.ndims = 2
    L.initialise_comms(C.byref(mesh))
    L.initialise_mesh_2d(C.byref(mesh))
    nx, ny = mesh.local_nx, mesh.local_ny

    shared = SharedData()
    L.initialise_shared_data_2d(nx, ny, mesh.pad, mesh.width, mesh.height, f,
                                mesh.edgex, mesh.edgey, C.byref(shared))

    edgex = _take(mesh.edgex, nx + 1)
    edgey = _take(mesh.edgey, ny + 1)
    src = NeutralSource()
    L.neutral_source_from_deck(f, mesh.width, mesh.height,
                               edgex[mesh.x_off + mesh.pad], edgey[mesh.y_off + mesh.pad],
                               edgex[nx - 2 * mesh.pad + mesh.x_off + mesh.pad],
                               edgey[ny - 2 * mesh.pad + mesh.y_off + mesh.pad],
                               C.byref(src))
    density = _take(shared.density, nx * ny)
    edgedx = _take(mesh.edgedx, nx + 1)
    edgedy = _take(mesh.edgedy, ny + 1)
    for buf in (mesh.edgex, mesh.edgey, mesh.edgedx, mesh.edgedy, mesh.celldx,
                mesh.celldy, shared.density, shared.energy):
        L.deallocate_data(buf)
    return Problem(
        deck=deck_path, nx=nx, ny=ny, pad=mesh.pad, x_off=mesh.x_off, y_off=mesh.y_off,
        width=width, height=height, dt=mesh.dt, niters=mesh.niters,
        edgex=edgex, edgey=edgey,
        edgedx=edgedx, edgedy=edgedy, density=density,
        nparticles=src.nparticles, nlocal_particles=src.nlocal_particles,
        initial_energy=src.initial_energy,
        local_particle_left_off=src.local_particle_left_off,
        local_particle_bottom_off=src.local_particle_bottom_off,
        local_particle_width=src.local_particle_width,
        local_particle_height=src.local_particle_height)
