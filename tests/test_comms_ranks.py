"""The rank layer through the C path, on the CPU: N processes of a plain-C program
(tests/c/comms_selftest.c, linked against the host library) meet over TCP the way
`neutral.hip --gpus N` and a multi-rank main.c do, and check barrier, the
reference's reduce_all_* hooks, the id broadcast, the tally-sized array all-reduce
and the particle shards.  (The device half -- RCCL -- needs GPUs: test_ranks_gpu.py.)"""
import os
import socket
import subprocess

import pytest

from conftest import ROOT

HOST_DIR = os.path.join(ROOT, "neutral_amd", "host")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("comms") / "comms_selftest")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-I", HOST_DIR,
                           os.path.join(ROOT, "tests", "c", "comms_selftest.c"),
                           "-L", HOST_DIR, "-lneutral_host", f"-Wl,-rpath,{HOST_DIR}", "-lm",
                           "-o", exe])
    return exe


def launch(exe, nranks, extra_env=None):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUTRAL_COMM_PORT=str(port),
                   NEUTRAL_COMM_TIMEOUT="30")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    return [(p,) + p.communicate(timeout=120) for p in procs]


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_ranks_meet_and_reduce(selftest, nranks):
    for r, (p, out, err) in enumerate(launch(selftest, nranks)):
        assert p.returncode == 0, (r, out, err)
        assert out.strip() == f"rank {r} of {nranks} ok"


def test_without_a_launcher_there_is_one_rank(selftest):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([selftest], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "rank 0 of 1 ok"


def test_a_missing_rank_is_an_error_not_a_hang(selftest):
    port = free_port()
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), NEUTRAL_COMM_PORT=str(port), NEUTRAL_COMM_TIMEOUT="2")
    out = subprocess.run([selftest], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode != 0
    assert "only 1 of 2 ranks arrived" in out.stderr


def test_strays_do_not_end_a_launch(selftest):
    """A hundred connections that are not ranks of the launch (garbage hellos, and silent ones
    that wait out the handshake timeout) knock on rank 0's port while it waits for rank 1: each is
    turned away, none of them ends the launch (an absolute count of 64 used to: round-4 advisor
    finding), and rank 1, arriving last behind them, is still let in and told so."""
    import threading
    import time
    port = free_port()
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               NEUTRAL_COMM_PORT=str(port), NEUTRAL_COMM_TIMEOUT="60")
    rank0 = subprocess.Popen([selftest], env=dict(env, RANK="0", LOCAL_RANK="0"),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    comm_port = port              # (NEUTRAL_COMM_PORT: the port that was found free)

    def knock(payload):
        for _ in range(200):
            try:
                s = socket.create_connection(("127.0.0.1", comm_port), timeout=2)
                break
            except OSError:
                time.sleep(0.05)
        else:
            return
        try:
            if payload:
                s.sendall(payload)
                s.settimeout(5)
                try:
                    s.recv(8)
                except OSError:
                    pass
            else:
                time.sleep(1.5)   # says nothing: rank 0's handshake timeout turns it away
        finally:
            s.close()

    strays = [threading.Thread(target=knock, args=(os.urandom(16),)) for _ in range(97)]
    strays += [threading.Thread(target=knock, args=(b"",)) for _ in range(3)]
    for t in strays:
        t.start()
    for t in strays:
        t.join()
    rank1 = subprocess.Popen([selftest], env=dict(env, RANK="1", LOCAL_RANK="1"),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out1, err1 = rank1.communicate(timeout=120)
    out0, err0 = rank0.communicate(timeout=120)
    assert rank0.returncode == 0, (out0, err0)
    assert rank1.returncode == 0, (out1, err1)
    assert out0.strip() == "rank 0 of 2 ok" and out1.strip() == "rank 1 of 2 ok"
