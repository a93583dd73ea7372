#!/usr/bin/env python3
"""Static opcode histogram of a kernel's hot loop, from the ISA listing `make -C neutral_amd asm`
leaves in neutral_amd/build/*.s, priced with the issue cost per opcode measured on the box
(profiles/valu_cycles.json, tools/micro/valu_opcodes.hip).

The hot loop is found, not assumed: the innermost loop (a label and a backward branch to
it) that contains at least --min-marker instructions matching --marker (v_alignbit_b32 for
the collision pass: Threefry's rotations; v_mul_f64 for the facet loop: the smallest loop
with a dozen of them is the facet trip).

  python tools/isa_histogram.py collide     # history_regroup_kernel<true,true,false,false>
  python tools/isa_histogram.py facet       # stream_kernel<true,false,false,false,false>
  python tools/isa_histogram.py collide --json
"""
import argparse
import collections
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.environ.get("NEUTRAL_ISA_BUILD") or os.path.join(ROOT, "neutral_amd", "build")  # (another directory: flag sweeps)

TARGETS = {
    "collide": ("neutral_kernels-hip-amdgcn-amd-amdhsa-gfx950.s",
                "_ZN7neutral22history_regroup_kernelILb1ELb1ELb0ELb0EEEvNS_9SolveArgsE",
                r"v_alignbit_b32", 60, None),
    # the facet loop is compiled four times (neutral_tiled.hip: run_facets).  Two are priced: for
    # windows of one density (no density load in the trip) and for any other, both with the
    # edges computed.  A loop is recognised by the cross_facet instantiation inlined into it
    # (the compiler names it in the block comments): <kChecked=0, kCachedReciprocals=1,
    # kDomain=0, kCarryTargets=1, kComputedEdges=1, WindowCellTallyT<flux=0, uniform=0|1>>
    "facet": ("neutral_tiled-hip-amdgcn-amd-amdhsa-gfx950.s",
              "_ZN7neutral13stream_kernelILb1ELb0ELb0ELb0ELb0EEEvNS_9SolveArgsENS_9TiledArgsE",
              r"v_mul_f64", 12, r"cross_facetILb0ELb1ELi0ELb1ELb1ENS_16WindowCellTallyTILb0ELb0ELb1EEE"),
    "facet_uniform": ("neutral_tiled-hip-amdgcn-amd-amdhsa-gfx950.s",
                      "_ZN7neutral13stream_kernelILb1ELb0ELb0ELb0ELb0EEEvNS_9SolveArgsENS_9TiledArgsE",
                      r"v_mul_f64", 12, r"cross_facetILb0ELb1ELi0ELb1ELb1ENS_16WindowCellTallyTILb0ELb1ELb1EEE"),
}

# Issue cycles one wave64 instruction holds its SIMD for, by opcode, measured with
# tools/micro/valu_opcodes.hip at 4 waves per SIMD, relative to v_mul_f64 = 4.  Opcodes not
# listed fall back by family (see cost_of).
MEASURED = os.path.join(ROOT, "profiles", "valu_cycles.json")


def source_sha16():
    """sha256 (first 16 hex digits) of the device sources the listing was compiled from: what
    bench.py compares with the sources it finds, so that a stale listing is flagged, not priced
    with silently."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "neutral_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load_costs():
    try:
        with open(MEASURED) as f:
            return json.load(f)["cycles"]
    except OSError:
        return {}


def base_op(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def cost_of(op, costs):
    """(cycles, how) for a VALU opcode: measured, or the family rule."""
    b = base_op(op)
    if b == "v_cndmask_b32":
        # back to back the VOP2 form measures ~20 cycles; in the kernels it costs what the
        # VOP3 form costs (profiles/r03/experiments/retune_isa_ab.log)
        return costs.get("v_cndmask_b32_e64", 4.0), "measured"
    if b in costs:
        return costs[b], "measured"
    for enc in ("_e64", "_e32"):  # compares are measured per encoding
        if b + enc in costs:
            return costs[b + enc], "measured"
    if re.search(r"_(rcp|rsq|sqrt)_f64", b):
        return costs.get("v_rcp_f64", 16.0), "family"
    if re.search(r"(_f64|_u64|_i64|_b64)\b", b) or b.startswith("v_cvt_"):
        return 4.0, "family"
    return costs.get("v_xor_b32", 2.0), "family"


def function_body(path, symbol):
    src = open(path).read()
    i = src.index(symbol + ":")
    j = src.index(".Lfunc_end", i)
    return src[i:j].splitlines()


def parse(body):
    """[(label or None, opcode or None, text)] per line"""
    out = []
    for line in body:
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            out.append((m.group(1), None, t))
            continue
        m = re.match(r"^([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|flat_[a-z0-9_]+|"
                     r"buffer_[a-z0-9_]+|scratch_[a-z0-9_]+)\b", t)
        if m:
            out.append((None, m.group(1), t))
    return out


def loops(items):
    """(start, end) index pairs: a label and a later branch back to it"""
    pos = {lab: k for k, (lab, _, _) in enumerate(items) if lab}
    out = []
    for k, (_, op, text) in enumerate(items):
        if op and op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(\.LBB\d+_\d+)", text)
            if m and m.group(1) in pos and pos[m.group(1)] < k:
                out.append((pos[m.group(1)], k))
    return out


def hot_loop(items, marker, min_marker, named=None):
    """the smallest loop with min_marker `marker` opcodes; `named`: one of its block labels
    must carry that text in its comment (the inlined function the block came from)"""
    best = None
    for a, b in loops(items):
        n = sum(1 for _, op, _ in items[a:b + 1] if op and re.match(marker, op))
        if named is not None and not any(lab and named in text for lab, _, text in items[a:b + 1]):
            continue
        if n >= min_marker and (best is None or (b - a) < (best[1] - best[0])):
            best = (a, b)
    if best is None:
        raise SystemExit("no loop with the marker found")
    return best


def hot_path(span, rare_min=8):
    """the trip without its rare sides: a run of instructions a conditional branch jumps over
    (forward, inside the loop) that holds more than rare_min vector instructions is a path the
    common trip skips -- in the facet loop: the reflection, the change of density, the
    question whether a history outside the window waits for the next pass, the tally of a
    cell outside the window.  (The common trip's own conditional runs are a handful of
    instructions: the LDS add.)"""
    pos = {lab: k for k, (lab, _, _) in enumerate(span) if lab}
    drop = [False] * len(span)
    for k, (_, op, text) in enumerate(span):
        if op and op.startswith("s_cbranch"):
            m = re.search(r"(\.LBB\d+_\d+)", text)
            if m and m.group(1) in pos and pos[m.group(1)] > k:
                j = pos[m.group(1)]
                if sum(1 for _, o, _ in span[k + 1:j] if o and o.startswith("v_")) > rare_min:
                    for i in range(k + 1, j):
                        drop[i] = True
    return [it for it, d in zip(span, drop) if not d]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target", choices=sorted(TARGETS) + ["mix"])
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--dump", action="store_true", help="print the loop's instructions")
    args = ap.parse_args()
    if args.target == "mix":
        # profiles/isa_mix.json: what bench.py prices the issue roofline with
        import subprocess
        import sys
        mix = {t: json.loads(subprocess.check_output([sys.executable, os.path.abspath(__file__), t,
                                                      "--json"])) for t in sorted(TARGETS)}
        out_mix = dict(mix)
        out_mix["source_sha16"] = source_sha16()  # (of neutral_amd/csrc as `make asm` compiled it)
        with open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w") as f:
            json.dump(out_mix, f, indent=1)
        for t, m in mix.items():
            print(f"{t}: {m['valu_instructions']} VALU per trip, {m['issue_cycles_per_trip']:.0f} cycles, "
                  f"mean {m['mean_cycles_per_valu']:.3f} ({m['mean_cycles_per_valu_low']:.3f}-"
                  f"{m['mean_cycles_per_valu_high']:.3f})")
        return
    fname, symbol, marker, min_marker, named = TARGETS[args.target]
    items = parse(function_body(os.path.join(BUILD, fname), symbol))
    a, b = hot_loop(items, marker, min_marker, named)
    span = items[a:b + 1]
    whole_valu = sum(1 for _, op, _ in span if op and op.startswith("v_"))
    if args.target.startswith("facet"):
        span = hot_path(span)
    if args.dump:
        for lab, op, text in span:
            print(text)
        return
    costs = load_costs()
    valu = collections.Counter(op for _, op, _ in span if op and op.startswith("v_")
                               and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
    salu = sum(1 for _, op, _ in span if op and op.startswith("s_"))
    mem = collections.Counter(op for _, op, _ in span
                              if op and op.startswith(("ds_", "global_", "flat_", "scratch_", "buffer_")))
    rows, cycles, by_how = [], 0.0, collections.Counter()
    for op, n in valu.most_common():
        c, how = cost_of(op, costs)
        rows.append({"op": op, "count": n, "cycles_each": c, "priced": how})
        cycles += n * c
        by_how[how] += n * c
    nvalu = sum(valu.values())
    # what the family-priced opcodes could move: all of them at the cheapest cost measured
    # (a two-operand 32-bit op) or at the dearest full-rate one
    fam = sum(r["count"] for r in rows if r["priced"] == "family")
    fam_cycles = sum(r["count"] * r["cycles_each"] for r in rows if r["priced"] == "family")
    lo_cycles = cycles - fam_cycles + fam * min(costs.get("v_xor_b32", 2.0), 2.0)
    hi_cycles = cycles - fam_cycles + fam * max(costs.get("v_mad_u64_u32", 4.4), 4.4)
    out = {"target": args.target, "symbol": symbol,
           "loop": {"first_label": span[0][0], "instructions": sum(1 for _, op, _ in span if op),
                    "valu_with_rare_paths": whole_valu},
           "valu_instructions": nvalu, "salu_instructions": salu, "memory_instructions": dict(mem),
           "issue_cycles_per_trip": cycles, "mean_cycles_per_valu": cycles / max(nvalu, 1),
           "mean_cycles_per_valu_low": lo_cycles / max(nvalu, 1),
           "mean_cycles_per_valu_high": hi_cycles / max(nvalu, 1),
           "cycles_priced_by_measurement": by_how["measured"] / max(cycles, 1e-9),
           "opcodes": rows}
    if args.json:
        print(json.dumps(out, indent=1))
        return
    print(f"{args.target}: loop at {span[0][0]}, {out['loop']['instructions']} instructions: "
          f"{nvalu} VALU, {salu} SALU, {sum(mem.values())} memory")
    for r in rows:
        print(f"  {r['op']:28s} {r['count']:4d} x {r['cycles_each']:5.2f}  ({r['priced']})")
    print(f"  issue cycles per trip {cycles:.0f}; mean {cycles / max(nvalu, 1):.2f} per VALU instruction; "
          f"{100 * out['cycles_priced_by_measurement']:.0f} % of the cycles priced by a measured opcode")


if __name__ == "__main__":
    main()
