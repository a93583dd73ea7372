"""The plain-C host layer and the CPU oracle under AddressSanitizer +
UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available on
this pool).  The selftest walks the deck reader, mesh, density boxes, source
box, cs reader, profiler and a full oracle run."""
import os
import subprocess

import pytest

from conftest import ROOT
from neutral_amd import cs_table, decks

SRC = [os.path.join(ROOT, "tests", "c", "host_selftest.c"),
       os.path.join(ROOT, "neutral_amd", "host", "host.c"),
       os.path.join(ROOT, "neutral_amd", "host", "comms_ranks.c"),
       os.path.join(ROOT, "neutral_amd", "host", "alloc_host.c"),
       os.path.join(ROOT, "neutral_amd", "host", "neutral_problem.c"),
       os.path.join(ROOT, "oracle", "neutral_oracle.c")]


@pytest.mark.parametrize("deck", ["csp", "split"])
def test_host_layer_and_oracle_under_asan_ubsan(tmp_path, deck):
    exe = str(tmp_path / "host_selftest")
    cmd = ["gcc", "-std=gnu99", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", exe] + SRC + ["-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    scatter, _ = cs_table.write_files(str(tmp_path))
    d = decks.write_deck(deck, str(tmp_path / f"{deck}.params"), nx=48, ny=48, nparticles=3000,
                         iterations=2, dt=2.0e-6)
    tests_file = str(tmp_path / "neutral.tests")
    with open(tests_file, "w") as f:
        f.write(f"{d} result=1.0\n")
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    env["ASAN_OPTIONS"] = "detect_leaks=1"
    run = subprocess.run([exe, d, scatter, tests_file], capture_output=True, text=True, env=env,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "selftest ok" in run.stdout


def test_rank_layer_under_asan_ubsan(tmp_path):
    """Three ranks of tests/c/comms_selftest.c built with the sanitizers: rendezvous,
    reductions, broadcast, array all-reduce, shard ranges."""
    import socket
    host = os.path.join(ROOT, "neutral_amd", "host")
    exe = str(tmp_path / "comms_selftest")
    cmd = ["gcc", "-std=gnu99", "-O1", "-g", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I", host, "-o", exe,
           os.path.join(ROOT, "tests", "c", "comms_selftest.c"),
           os.path.join(host, "host.c"), os.path.join(host, "comms_ranks.c"),
           os.path.join(host, "alloc_host.c"), os.path.join(host, "neutral_problem.c"), "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUTRAL_COMM_PORT=str(port),
                   NEUTRAL_COMM_TIMEOUT="60", ASAN_OPTIONS="detect_leaks=1")
        procs.append(subprocess.Popen([exe], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, (r, out, err[-3000:])
        assert out.strip() == f"rank {r} of 3 ok"
