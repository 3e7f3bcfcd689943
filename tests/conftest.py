import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_fit_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "fit_*.npz")))


def load_golden(kind, name):
    z = np.load(os.path.join(GOLDEN, "%s_%s.npz" % (kind, name)), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (test infrastructure)"""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def c2_problem():
    """BASELINE.json config 2 / 3 data at full size (seeded): 100 k users x 50 k items x 5 M interactions"""
    from rankfm_amd import synthetic
    cfg = synthetic.CONFIGS["C2"]
    U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
    pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
    return U, I, N, F, pairs, csr
