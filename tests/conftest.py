import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line(
        "markers", "fullkat: reference known answers at the decks' default sizes "
        "(minutes of CPU; enabled with NEUTRAL_FULL_KATS=1)")


def _ensure_built():
    """Artefacts missing from a fresh checkout are built on demand (the normal
    route is __graft_entry__.build()): the oracle, the host layer and -- where
    hipcc is installed, it cross-compiles without a GPU -- the HIP library."""
    import shutil
    import subprocess
    oracle = os.path.join(ROOT, "oracle", "liboracle.so")
    hostlib = os.path.join(ROOT, "neutral_amd", "host", "libneutral_host.so")
    hiplib = os.path.join(ROOT, "neutral_amd", "libneutral_hip.so")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(hostlib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "neutral_amd"),
                               "host/libneutral_host.so"])
    if not os.path.exists(hiplib) and (shutil.which("hipcc") or
                                       os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "neutral_amd"), "all"])


_ensure_built()


@pytest.fixture(scope="session")
def cs():
    from neutral_amd import cs_table
    return cs_table.load()


@pytest.fixture(scope="session")
def pins():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")) as f:
        return json.load(f)


@pytest.fixture()
def make_problem(tmp_path):
    """make_problem('csp', nx=64, ny=64, nparticles=4096, iterations=2) -> Problem"""
    from neutral_amd import decks, host

    def _make(name, **overrides):
        if "nx" in overrides and "ny" not in overrides:
            overrides["ny"] = overrides["nx"]
        path = decks.write_deck(name, str(tmp_path / f"{name}.params"), **overrides)
        return host.setup_problem(path, decks.ARCH_WIDTH, decks.ARCH_HEIGHT)

    return _make


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
