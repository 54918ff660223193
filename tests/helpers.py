"""Shared helpers for the parity tests (load golden fixtures into oracle state-dicts)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name))
    rec = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = {p: {k[len(p) + 1:]: v for k, v in rec.items() if k.startswith(p + ".")} for p in ("sdf", "col", "var")}
    return rec, sd["sdf"], sd["col"], sd["var"]["variance"]


def relerr(a, b):
    a = a.double()
    b = b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
