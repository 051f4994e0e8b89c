"""Generates tests/golden/oracle_films.npz: two small films of the Cornell box rendered by the CPU oracle (oracle/rt_oracle.c),
64 x 48 pixels, 8 samples, 30 bands, seed 42, NEE+MIS and NEE.  Run from the repository root:

    python tests/golden/make_golden_films.py

These are outputs of the oracle, not of the reference (its renderer cannot be built here: Boost is absent): they pin the oracle
against its own past, so that an edit which moves the oracle and the kernels together is noticed.  Stored as float64, one row of 30
bins per pixel in the film's column-major pixel order."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402

NX, NY, NS, BANDS, SEED = 64, 48, 8, 30, 42


def films():
    mesh, records, cam, oc = oracle.cornell_setup(os.path.join(HERE, "cornell"), NX, NY)
    sc = oracle.Scene(mesh)
    assert sc.set_materials(records) == 1
    out = {}
    for name, alg in (("mis", oracle.ALG_MIS), ("nee", oracle.ALG_NEE)):
        film, st = sc.render(oc, NX, NY, NS, BANDS, alg, SEED)
        assert st["samples"] == NX * NY * NS
        out[name] = film
    return out


if __name__ == "__main__":
    out = films()
    np.savez_compressed(os.path.join(HERE, "oracle_films.npz"), **out)
    print({k: (v.shape, float(v.sum())) for k, v in out.items()})
