"""Drop-in boundary: the reference's unmodified main.c + neutral_data.c (-DSoA)
compile against the host-layer headers and link against libneutral_hip.so
alone; on a GPU the resulting driver passes the reference's own validation."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT, gpu_available

REFERENCE = os.environ.get("NEUTRAL_REFERENCE", "/root/reference")
DROPIN = os.path.join(ROOT, "integration", "_dropin", "neutral.hip_dropin")
HIPLIB = os.path.join(ROOT, "neutral_amd", "libneutral_hip.so")


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "main.c")),
                    reason="reference tree not present")
@pytest.mark.skipif(not os.path.exists(HIPLIB), reason="libneutral_hip.so not built")
def test_reference_driver_links_against_the_library_alone():
    out = subprocess.run(["bash", os.path.join(ROOT, "integration", "build_dropin.sh")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert os.path.exists(DROPIN)
    # every interface symbol the driver imports is resolved by libneutral_hip.so
    undefined = subprocess.run(["nm", "-D", "--undefined-only", DROPIN], capture_output=True,
                               text=True).stdout
    exported = subprocess.run(["nm", "-D", "--defined-only", HIPLIB], capture_output=True,
                              text=True).stdout
    exported = {ln.split()[-1] for ln in exported.splitlines() if ln.strip()}
    for sym in ("solve_transport_2d", "inject_particles", "validate", "allocate_data",
                "allocate_uint64_data", "allocate_host_data", "copy_buffer",
                "move_host_buffer_to_device", "get_int_parameter", "get_double_parameter",
                "get_key_value_parameter", "initialise_mesh_2d", "initialise_shared_data_2d",
                "initialise_comms", "handle_boundary_2d", "barrier", "profiler_start_timer",
                "profiler_end_timer"):
        assert sym in undefined, f"driver does not import {sym}?"
        assert sym in exported, f"libneutral_hip.so does not export {sym}"


@pytest.mark.gpu
@pytest.mark.skipif(not gpu_available(), reason="needs a GPU")
@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in driver not prebuilt")
@pytest.mark.parametrize("name", ["stream", "csp"])
def test_reference_driver_passes_its_own_validation_on_the_gpu(tmp_path, name):
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    deck = decks.write_deck(name, str(run / "problems" / f"{name}.params"))
    rel = os.path.join("problems", f"{name}.params")
    decks.write_tests_file(str(run / "problems" / "neutral.tests"), {name: rel})
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    # the binary links the system ROCm runtime; keep torch's bundled one out of its way
    out = subprocess.run([DROPIN, rel], cwd=str(run), capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED validation." in out.stdout, out.stdout[-2000:]
    assert "File elastic_scatter.cs contains 29999 entries" in out.stdout
    shutil.rmtree(run, ignore_errors=True)


OWN_DRIVER = os.path.join(ROOT, "neutral_amd", "host", "neutral.hip")


@pytest.mark.gpu
@pytest.mark.skipif(not gpu_available(), reason="needs a GPU")
@pytest.mark.skipif(not os.path.exists(OWN_DRIVER), reason="neutral.hip not built")
def test_own_driver_runs_decks_like_the_reference_driver(tmp_path):
    """neutral.hip (neutral_amd/host/neutral_driver.c): csp at its default size must
    pass the reference's known answer; --set overrides give the BASELINE shapes."""
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    rel = os.path.join("problems", "csp.params")
    decks.write_deck("csp", str(run / rel))
    decks.write_tests_file(str(run / "problems" / "neutral.tests"), {"csp": rel})
    out = subprocess.run([OWN_DRIVER, rel], cwd=str(run), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED validation." in out.stdout, out.stdout[-2000:]
    assert out.stdout.count("Iteration") == 10
    # BASELINE config 1 shape through --set: exact collision count of the omp3 run
    # recorded in BASELINE.md (69 884 072 collisions, 11 facets)
    rel2 = os.path.join("problems", "scatter.params")
    decks.write_deck("scatter", str(run / rel2))
    out = subprocess.run([OWN_DRIVER, rel2, "--set", "nx=100", "--set", "ny=100", "--set",
                          "nparticles=100000", "--set", "iterations=1"], cwd=str(run),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Collisions 69884072" in out.stdout and "Facets     11" in out.stdout, out.stdout
    assert "Particles  100000" in out.stdout
