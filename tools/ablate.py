#!/usr/bin/env python3
"""Kernel experiments (not part of the product or the tests).

One timing run of the library selected by NEUTRAL_HIP_LIB:

  NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_x.so \
      python tools/ablate.py csp 400 10000000 10 [variant]

An experiment = a few builds of the same ABI with different -D knobs, each timed
on a few workloads (what the one-off tools/ablate_r*.sh scripts of round 1 did):

  python tools/ablate.py build  rf24=-DNEUTRAL_REFILL_MIN=24 rf48=-DNEUTRAL_REFILL_MIN=48
        (in the build container: `make -C neutral_amd variant TAG=.. EXTRA=..` per tag)
  python tools/ablate.py matrix --libs default,rf24,rf48 \
        --run "scatter 400 20000000 1 2" --run "csp 400 100000000 10 2"
        (on the GPU box: one child process per library and workload; a child that
         fails or exceeds --timeout ends the matrix, nothing is retried)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lib_path(tag):
    return None if tag in ("", "default") else \
        os.path.join(ROOT, "neutral_amd", "build", f"libneutral_hip_{tag}.so")


def build(specs):
    for spec in specs:
        tag, _, extra = spec.partition("=")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "neutral_amd"), "variant",
                               f"TAG={tag}", f"EXTRA={extra}"])


def matrix(argv):
    import argparse
    ap = argparse.ArgumentParser(prog="ablate.py matrix")
    ap.add_argument("--libs", default="default", help="comma-separated build tags")
    ap.add_argument("--run", action="append", required=True,
                    help='"deck nx nparticles iterations [variant]"')
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for every child")
    a = ap.parse_args(argv)
    extra_env = dict(e.split("=", 1) for e in a.env)
    for tag in a.libs.split(","):
        env = dict(os.environ, **extra_env)
        env.pop("NEUTRAL_HIP_LIB", None)
        path = lib_path(tag)
        if path:
            env["NEUTRAL_HIP_LIB"] = path
        for run in a.run:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + run.split(),
                               env=env, timeout=a.timeout, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            lines = [l for l in p.stdout.splitlines() if "amdgpu.ids" not in l]
            print(lines[-1] if lines else f"{tag}: no output", flush=True)
            if p.returncode != 0:
                print("\n".join(lines[-15:]))
                sys.exit(p.returncode)


def single(argv):
    from neutral_amd import cs_table, decks, host
    from neutral_amd import interface as iface

    deck, nx, n, its = argv[0], int(argv[1]), int(float(argv[2])), int(argv[3])
    variant = int(argv[4]) if len(argv) > 4 else None
    iface.set_quiet(True)
    iface.set_lazy_export(os.environ.get("NEUTRAL_EAGER_EXPORT") != "1")
    keys, values = cs_table.load()
    with tempfile.TemporaryDirectory() as tmp:
        path = decks.write_deck(deck, os.path.join(tmp, "d.params"), nx=nx, ny=nx, nparticles=n,
                                iterations=its)
        prob = host.setup_problem(path)
        sim = iface.Simulation(prob, keys, values, variant=variant)
        sim.inject()
        sim.step(1)            # warm-up
        sim.inject()
        sim.zero_tally()
        import time
        import torch
        tot_ms = 0.0
        export_ms = 0.0
        tot_steps = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per = []
        stages = [0.0, 0.0, 0.0]
        for tt in range(1, its + 1):
            r = sim.step(tt)
            tot_ms += r.kernel_ms
            export_ms += r.stats.export_ms
            tot_steps += r.particle_steps
            stages[0] += r.stats.sort_ms
            stages[1] += r.stats.stream_ms
            stages[2] += r.stats.collide_ms
            per.append(f"{r.kernel_ms:.1f}" + (f"/p{r.stats.stream_passes}" if r.stats.stream_passes > 1 else ""))
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        tag = os.path.basename(os.environ.get("NEUTRAL_HIP_LIB", "default"))
        vname = {0: "K1 over-particle", 1: "K2 event-regrouped", 2: "K3 tiled", None: "K3 tiled"}[variant]
        print(f"{tag:24s} {vname:18s} {deck} nx={nx} n={n}: {tot_ms:9.1f} ms  {tot_steps / tot_ms / 1e6:8.3f} Gsteps/s"
              f"  wall {wall_ms:.1f} ms (write-back pass {export_ms:.1f})"
              f"  tally={float(sim.tally.sum()):.6e}  sort/stream/collide "
              f"{stages[0]:.1f}/{stages[1]:.1f}/{stages[2]:.1f}  per-step ms: {' '.join(per)}",
              flush=True)
        sim.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "matrix":
        matrix(sys.argv[2:])
    else:
        single(sys.argv[1:])
