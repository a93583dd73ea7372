"""The reference's problem decks as data, plus a writer for the deck format.

The values are those of ``problems/{scatter,stream,csp,split}.params`` and the
known answers of ``problems/neutral.tests`` in the reference tree; they are
configuration data, kept here as dictionaries so tests and bench.py can emit a
deck with ``nx``/``ny``/``nparticles``/``iterations`` overridden (the BASELINE
configurations are the shipped decks at other sizes).
"""
from __future__ import annotations

import os
from typing import Dict, List

# box = (xpos, ypos, width, height) as fractions of the mesh extent
STANDARD_DECKS: Dict[str, dict] = {
    "scatter": dict(
        source=(0.2, 0.2, 0.6, 0.6),
        problems=[dict(density=1.0e4, energy=0.0, box=(0.0, 0.0, 1.0, 1.0))],
        nparticles=10000000, initial_energy=1.0e3, dt=1.0e-7,
        nx=4000, ny=4000, iterations=2, visit_dump=0),
    "stream": dict(
        source=(0.45, 0.45, 0.1, 0.1),
        problems=[dict(density=1.0e-30, energy=1.0, box=(0.0, 0.0, 1.0, 1.0))],
        nparticles=1000000, initial_energy=1.0e6, dt=1.0e-7,
        nx=4000, ny=4000, iterations=1, visit_dump=0),
    "csp": dict(
        source=(0.1, 0.1, 0.2, 0.2),
        problems=[dict(density=1.0e-30, energy=0.0, box=(0.0, 0.0, 1.0, 1.0)),
                  dict(density=1.0e4, energy=1.0, box=(0.4, 0.4, 0.2, 0.2))],
        nparticles=1000000, initial_energy=1.0e4, dt=1.0e-7,
        nx=4000, ny=4000, iterations=10, visit_dump=0),
    "split": dict(
        source=(0.4, 0.4, 0.2, 0.2),
        problems=[dict(density=1.0e-30, energy=0.0, box=(0.0, 0.0, 1.0, 0.5)),
                  dict(density=1.0e3, energy=1.0, box=(0.0, 0.5, 1.0, 0.5))],
        nparticles=1000000, initial_energy=2.5e4, dt=1.0e-7,
        nx=4000, ny=4000, iterations=1, visit_dump=0),
}

# problems/neutral.tests:1-3 -- sum of the energy deposition tally at the decks'
# default sizes, checked by validate() at VALIDATE_TOLERANCE = 1e-3
# (neutral_data.h:27, omp3/neutral.c:549).  split has no entry.
KNOWN_ANSWERS: Dict[str, float] = {
    "scatter": 3.411662060900e-02,
    "stream": 5.760064605960129e-24,
    "csp": 1.121870290714e+07,
}
VALIDATE_TOLERANCE = 1.0e-3

# ../arch.params of the parent project: mesh extent.  width = height = 1.0 is
# what makes the csp known answer come out.
ARCH_WIDTH = 1.0
ARCH_HEIGHT = 1.0


def _fmt(v: float) -> str:
    return repr(float(v))


def deck_text(name: str, **overrides) -> str:
    """Returns the text of deck `name` with scalar entries overridden."""
    d = dict(STANDARD_DECKS[name])
    for k, v in overrides.items():
        if v is None:
            continue
        if k not in d or k in ("source", "problems"):
            raise KeyError(f"cannot override deck entry {k!r}")
        d[k] = v
    sx, sy, sw, sh = d["source"]
    lines: List[str] = [
        f"source xpos={_fmt(sx)} ypos={_fmt(sy)} width={_fmt(sw)} height={_fmt(sh)}"
    ]
    for i, p in enumerate(d["problems"]):
        bx, by, bw, bh = p["box"]
        lines.append(
            f"problem_{i} density={_fmt(p['density'])} energy={_fmt(p['energy'])} "
            f"xpos={_fmt(bx)} ypos={_fmt(by)} width={_fmt(bw)} height={_fmt(bh)}")
    lines.append(f"nparticles        {int(d['nparticles'])}  # particles per source injection")
    lines.append(f"initial_energy    {_fmt(d['initial_energy'])}  # eV")
    lines.append(f"dt                {_fmt(d['dt'])}")
    lines.append(f"nx                {int(d['nx'])}")
    lines.append(f"ny                {int(d['ny'])}")
    lines.append(f"iterations        {int(d['iterations'])}")
    lines.append(f"visit_dump        {int(d['visit_dump'])}")
    return "\n".join(lines) + "\n"


def write_deck(name: str, path: str, **overrides) -> str:
    """Writes deck `name` (with overrides) to `path`; returns `path`."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write(deck_text(name, **overrides))
    return path


def write_tests_file(path: str, deck_paths: Dict[str, str]) -> str:
    """Writes a neutral.tests-style file: `<deck path> result=<known answer>`."""
    with open(path, "w") as f:
        for name, p in deck_paths.items():
            if name in KNOWN_ANSWERS:
                f.write(f"{p} result={KNOWN_ANSWERS[name]!r}\n")
    return path
