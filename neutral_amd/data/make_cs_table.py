#!/usr/bin/env python3
"""Regenerates neutral_amd/data/cs_table.npz from the reference's data files.

The reference ships two byte-identical 29 999-row text tables
(elastic_scatter.cs, capture.cs; "%.12e %.12e\\n" rows, MIT licence).  They are
INPUT DATA of the hot path, needed on machines where the reference tree is
absent, so their parsed float64 columns are kept as one compressed array file.
Printing the columns back with "%.12e %.12e\\n" reproduces the text files byte
for byte (checked below).

Run in the build container:  python neutral_amd/data/make_cs_table.py
"""
import hashlib
import os
import sys

import numpy as np

REF = os.environ.get("NEUTRAL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    a = open(os.path.join(REF, "elastic_scatter.cs"), "rb").read()
    b = open(os.path.join(REF, "capture.cs"), "rb").read()
    if a != b:
        print("the two tables differ: keep them separately", file=sys.stderr)
        return 1
    rows = [ln.split() for ln in a.decode().splitlines()]
    keys = np.array([float(r[0]) for r in rows], dtype=np.float64)
    values = np.array([float(r[1]) for r in rows], dtype=np.float64)
    text = "".join("%.12e %.12e\n" % (k, v) for k, v in zip(keys, values)).encode()
    assert text == a, "round trip through float64 is not byte exact"
    assert np.all(np.diff(keys) > 0)
    np.savez_compressed(os.path.join(HERE, "cs_table.npz"), keys=keys, values=values,
                        md5=np.frombuffer(hashlib.md5(a).digest(), dtype=np.uint8))
    print(len(keys), "rows, md5", hashlib.md5(a).hexdigest())
    return 0


if __name__ == "__main__":
    sys.exit(main())
