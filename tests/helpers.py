"""Shared helpers for the parity tests (load golden fixtures into oracle state-dicts)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name))
    rec = {k: torch.from_numpy(z[k]) for k in z.files}
    src = rec
    if not any(k.startswith("sdf.") for k in rec):      # neus_full_128.npz: the nets of neus_full.npz, not stored twice
        w = np.load(os.path.join(GOLDEN, "neus_full.npz"))
        src = {k: torch.from_numpy(w[k]) for k in w.files if k[:4] in ("sdf.", "col.", "var.")}
    sd = {p: {k[len(p) + 1:]: v for k, v in src.items() if k.startswith(p + ".")} for p in ("sdf", "col", "var")}
    return rec, sd["sdf"], sd["col"], sd["var"]["variance"]


def relerr(a, b):
    a = a.double()
    b = b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
