import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


FLOOR_FRAC = 0.1


def rel_err(a, b, floor_frac=FLOOR_FRAC):
    """Parity metric: max |a-b| / max(|b|, floor), floor = floor_frac * max|b|.

    SURVEY proposes floor ~ 1e-3 of the tensor's max-abs; measured here, the reference's OWN fp32
    roundoff (reference fp32 vs the same modules run in fp64) is already 1.3e-4 under that floor on a
    single encoder cell and ~1.4e-4 (floor 1e-2) through the whole network, i.e. above the 1e-4 bar
    before any port exists.  The tests therefore use floor = 0.1 * max-abs (the torch.testing
    convention: rtol 1e-4 plus atol 1e-5 on O(1) tensors) for the 1e-4 bar, and additionally check
    against the float64 rollout of the reference ("*64" golden entries) that the HIP path is no
    further from the exact result than the fp32 reference itself (DESIGN.md "Parity metric")."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    floor = floor_frac * max(float(np.abs(b).max()), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def assert_close(a, b, tol, what=""):
    err = rel_err(a, b)
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol:.1e}"


def masked_parity(masked, ref_masked, ref_cls, ref_raw, tol, thred=0.5, band=1e-5):
    """Flip-tolerant comparison of the thresholded output: compare only where the
    reference's cls is not within ``band`` of the threshold; returns the number of excluded pixels."""
    sure = np.abs(np.asarray(ref_cls, np.float64) - thred) > band
    floor = FLOOR_FRAC * max(float(np.abs(ref_raw).max()), 1e-30)
    err = np.abs(np.asarray(masked, np.float64) - ref_masked) / np.maximum(np.abs(ref_masked), floor)
    bad = float(err[sure].max()) if sure.any() else 0.0
    assert bad <= tol, f"masked output rel err {bad:.3e} > {tol:.1e}"
    return int((~sure).sum())


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
