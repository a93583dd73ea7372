"""bench.py's N > 1 path on real hardware: two ranks (sharing the one GPU of the
test box, tally exchange staged through the host: RCCL refuses two ranks on one
device) each run the HIP path on the shard the library cuts for them, and
solve_transport_2d all-reduces the step tally; totals must equal the one-rank run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]


def _bench(extra, nproc, record=None):
    env = dict(os.environ)
    if record:
        env["NEUTRAL_ONE_RANK_RECORD"] = record
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    base = [sys.executable]
    if nproc > 1:
        base += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                 "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200)]
    cmd = base + [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3",
                  "--warmup", "1", "--nparticles", "3000001", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    return json.loads(lines[0])


def test_two_rank_bench_equals_one_rank(tmp_path):
    record = str(tmp_path / "one_rank.json")
    one = _bench(["--record-one-rank"], 1, record)
    two = _bench(["--comm", "host", "--share-device"], 2, record)
    # the N > 1 line checks itself against the one-rank record and says what its exchange did
    assert two["parity_vs_one_rank"]["recorded"] and two["parity_vs_one_rank"]["event_counts_equal"]
    assert two["parity_vs_one_rank"]["global_tally_rel"] <= 1e-12
    assert two["exchange"] == {"ranks_summed_over": 2, "host_collectives_per_step": 0}
    assert one["exchange"]["ranks_summed_over"] == 1
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    # what a first multi-GPU record needs to be read: every rank's own view of the run
    rk = two["ranks"]
    assert "ranks" not in one
    assert rk["transport_asked"] == "host" and rk["transport_per_rank"] == ["host", "host"]
    assert isinstance(rk["rccl_version"], int) and rk["rccl_version"] >= 0
    for key in ("device_ms_per_step", "stream_ms_per_step", "collide_ms_per_step",
                "exchange_ms_per_step", "particles_alive_last_step", "particles_in_shard",
                "first_call_wall_ms", "first_timed_step_wall_ms", "steady_step_wall_ms"):
        assert len(rk[key]["per_rank"]) == 2 and rk[key]["min"] <= rk[key]["max"], key
    # the first call of the process (lazy set-up) is told apart from the steady state
    assert rk["first_call_wall_ms"]["min"] > 0 and rk["steady_step_wall_ms"]["min"] > 0
    assert rk["first_call_wall_ms"]["max"] >= rk["steady_step_wall_ms"]["min"]
    # the exchange is judged against its bar in the line itself (the host route of this test
    # is allowed to miss it: what is asserted is that the line says which)
    chk = rk["exchange_check"]
    assert chk["bar_ms"] == 0.5 and chk["worst_step_ms_any_rank"] >= chk["mean_ms_per_step_max_over_ranks"] > 0
    assert chk["ok"] == (chk["mean_ms_per_step_max_over_ranks"] < chk["bar_ms"])
    assert rk["device_ms_per_step"]["min"] > 0 and rk["exchange_ms_per_step"]["min"] > 0
    assert sum(rk["particles_in_shard"]["per_rank"]) == 3000001
    assert 0 < sum(rk["particles_alive_last_step"]["per_rank"]) <= 3000001
    assert "host" in two["config"]["tally_exchange"]
    assert two["events"] == one["events"]                      # exact event totals
    assert two["global_tally"] == pytest.approx(one["global_tally"], rel=1e-12)
    for d in (one, two):
        assert d["metric"] == "particle-steps/sec" and d["unit"] == "particle-steps/s"
        assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
        assert d["roofline"]["bound"] == "valu_issue"
        assert d["roofline"]["frac"] is None or 0 < d["roofline"]["frac"] <= 1.0
        assert d["roofline"]["hbm"]["b_hbm_min_per_step"] > 0
        assert d["lazy_export"]["events_equal"] and d["lazy_export"]["value"] > 0
        assert d["value"] == pytest.approx(
            (d["events"]["facets"] + d["events"]["collisions"] + d["events"]["census"])
            / (d["ms_per_step"] * 1e-3 * d["steps"]), rel=1e-6)
