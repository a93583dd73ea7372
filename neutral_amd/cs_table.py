"""The dummy resonance cross-section table shipped with the reference.

``elastic_scatter.cs`` and ``capture.cs`` of the reference are byte-identical
(md5 6deb6261687eb2c528e9f4f1bff12793), so one table serves both roles; see
``data/make_cs_table.py`` for how ``data/cs_table.npz`` was made.
"""
from __future__ import annotations

import hashlib
import os
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_NPZ = os.path.join(_HERE, "data", "cs_table.npz")
REFERENCE_MD5 = "6deb6261687eb2c528e9f4f1bff12793"
CS_SCATTER_FILENAME = "elastic_scatter.cs"   # neutral_data.h:30
CS_CAPTURE_FILENAME = "capture.cs"           # neutral_data.h:31


def load() -> Tuple[np.ndarray, np.ndarray]:
    """Returns (keys [eV], values [barns]) as float64 arrays of 29 999 rows."""
    with np.load(_NPZ) as z:
        return np.ascontiguousarray(z["keys"]), np.ascontiguousarray(z["values"])


def text() -> bytes:
    """The table in the reference's on-disk format ("%.12e %.12e\\n" rows)."""
    keys, values = load()
    t = "".join("%.12e %.12e\n" % (k, v) for k, v in zip(keys, values)).encode()
    if hashlib.md5(t).hexdigest() != REFERENCE_MD5:
        raise RuntimeError("cross-section table does not reproduce the reference file")
    return t


def write_files(directory: str) -> Tuple[str, str]:
    """Writes elastic_scatter.cs and capture.cs into `directory`."""
    os.makedirs(directory, exist_ok=True)
    t = text()
    paths = (os.path.join(directory, CS_SCATTER_FILENAME),
             os.path.join(directory, CS_CAPTURE_FILENAME))
    for p in paths:
        with open(p, "wb") as f:
            f.write(t)
    return paths
