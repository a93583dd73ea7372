"""CPU-side checks: the C-ABI library loads and exports every symbol that
include/neutral_hip.h declares (no compute without a GPU), and the plain-C host
layer (deck reader, mesh, density boxes, source box, cs reader) behaves as the
reference's call sites need."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from neutral_amd import cs_table, decks, host

HEADER = os.path.join(ROOT, "include", "neutral_hip.h")
HIPLIB = os.path.join(ROOT, "neutral_amd", "libneutral_hip.so")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.mark.skipif(not os.path.exists(HIPLIB), reason="libneutral_hip.so not built")
def test_library_exports_every_declared_symbol():
    declared = _declared_functions()
    assert {"solve_transport_2d", "inject_particles", "validate"} <= set(declared)
    assert len(declared) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", HIPLIB], capture_output=True, text=True)
    exported = {ln.split()[-1] for ln in out.stdout.splitlines() if ln.strip()}
    missing = [n for n in declared if n not in exported]
    assert not missing, f"declared in neutral_hip.h but not exported: {missing}"
    # loads without a GPU, and the python mirror lists exactly the declared surface
    lib = ctypes.CDLL(HIPLIB)
    for n in declared:
        assert hasattr(lib, n)
    import neutral_amd.interface as iface
    assert sorted(iface.ABI_SYMBOLS) == declared
    assert lib.neutral_hip_abi_version() >= 1


def test_struct_layouts_match_the_reference_types():
    import neutral_amd.interface as iface
    # Particle (-DSoA, neutral_data.h:45-61): 11 pointers; CrossSection (:38-43): 2 pointers + int
    assert ctypes.sizeof(iface.Particle) == 11 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(iface.CrossSection) == 24
    assert [f[0] for f in iface.Particle._fields_] == [
        "x", "y", "omega_x", "omega_y", "energy", "weight", "dt_to_census", "mfp_to_collision",
        "cellx", "celly", "dead"]


def test_step_stats_mirror_matches_the_c_struct(tmp_path):
    """interface.StepStats (ctypes) against NeutralHipStepStats as a C compiler lays it out:
    same size, same offset for every field (a field added to one and not the other would
    silently shift everything behind it)."""
    import neutral_amd.interface as iface
    fields = [f[0] for f in iface.StepStats._fields_]
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "neutral_hip.h"\n'
        'int main(void) {\n  printf("%zu\\n", sizeof(NeutralHipStepStats));\n' +
        "".join(f'  printf("{f} %zu\\n", offsetof(NeutralHipStepStats, {f}));\n' for f in fields) +
        "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe)])
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert int(lines[0]) == ctypes.sizeof(iface.StepStats)
    for line in lines[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(iface.StepStats, name).offset == int(off), name


def test_deck_reader(tmp_path):
    p = tmp_path / "d.params"
    p.write_text("# comment line\n"
                 "source xpos=0.1 ypos=0.2 width=0.3 height=0.4\n"
                 "problem_0 density=1.0e-30 energy=0.0 xpos=0.0 ypos=0.0 width=1.0 height=1.0\n"
                 "nparticles   1000   # trailing comment\n"
                 "dt\t1.0e-7\n"
                 "nxx 5\nnx 7\n")
    f = str(p)
    assert host.get_int_parameter("nparticles", f) == 1000
    assert host.get_int_parameter("nx", f) == 7            # whole-token match, not prefix
    assert host.get_double_parameter("dt", f) == 1.0e-7
    kv = host.get_key_value_parameter("source", f)
    assert list(kv.items()) == [("xpos", 0.1), ("ypos", 0.2), ("width", 0.3), ("height", 0.4)]
    assert host.get_key_value_parameter("problem_1", f) is None
    assert host.get_key_value_parameter("source", str(tmp_path / "absent")) is None
    # neutral.tests-style lookup: the specifier is a path (omp3/neutral.c:541)
    t = tmp_path / "neutral.tests"
    t.write_text("problems/csp.params result=1.121870290714e+07\n")
    assert host.get_key_value_parameter("problems/csp.params", str(t)) == {"result": 1.121870290714e+07}


def test_within_tolerance_is_relative():
    assert host.within_tolerance(1.121870290714e+07, 1.121829757714269e+07, 1e-3)
    assert not host.within_tolerance(1.0, 1.002, 1e-3)
    assert host.within_tolerance(5.760064605960129e-24, 5.760059926484882e-24, 1e-3)


def test_mesh_and_density_boxes(make_problem):
    prob = make_problem("csp", nx=400, nparticles=1000, iterations=1)
    assert prob.edgex.shape == (401,) and prob.edgex[0] == 0.0
    assert np.array_equal(prob.edgex, (1.0 / 400) * np.arange(401))
    assert np.all(np.diff(prob.edgex) > 0)
    d = prob.density.reshape(400, 400)
    # problem_1: density 1e4 where the cell's lower-left corner lies in [0.4,0.6)^2
    inside = (prob.edgex[:-1] >= 0.4) & (prob.edgex[:-1] < 0.4 + 0.2)
    assert np.all(d[np.ix_(inside, inside)] == 1.0e4)
    assert np.all(d[~inside, :] == 1.0e-30) and np.all(d[:, ~inside] == 1.0e-30)
    assert 78 <= inside.sum() <= 82
    split = make_problem("split", nx=800, nparticles=1000, iterations=1)
    ds = split.density.reshape(800, 800)
    assert np.all(ds[:400, :] == 1.0e-30) and np.all(ds[400:, :] == 1.0e3)


def test_source_box_keeps_the_reference_arithmetic(make_problem):
    # neutral_data.c:65-76: scatter's box comes out 0.6000000000000001 wide, not 0.6
    sc = make_problem("scatter", nx=100, nparticles=100000, iterations=1)
    assert sc.local_particle_left_off == 0.2
    assert sc.local_particle_width == 1.0 - ((1.0 - (0.2 + 0.6)) + 0.2)
    assert sc.local_particle_width != 0.6
    assert sc.nlocal_particles == 100000 and sc.initial_energy == 1.0e3
    st = make_problem("stream", nx=400, nparticles=12345, iterations=1)
    assert st.nlocal_particles == 12345
    assert st.local_particle_left_off == 0.45 and st.dt == 1.0e-7 and st.niters == 1


def test_cs_reader_round_trip(tmp_path, cs):
    keys, values = cs
    scatter, capture = cs_table.write_files(str(tmp_path))
    k, v = host.read_cs_file(scatter)
    assert len(k) == 29999                          # neutral_data.c:129-136 counts newlines
    assert np.array_equal(k, keys) and np.array_equal(v, values)
    assert open(scatter, "rb").read() == open(capture, "rb").read()
    with pytest.raises(FileNotFoundError):
        host.read_cs_file(str(tmp_path / "missing.cs"))


def test_deck_writer_overrides():
    text = decks.deck_text("csp", nx=400, ny=400, nparticles=100000000, iterations=10)
    assert "nx                400" in text and "nparticles        100000000" in text
    assert text.count("problem_") == 2
    with pytest.raises(KeyError):
        decks.deck_text("csp", source=(0, 0, 1, 1))
