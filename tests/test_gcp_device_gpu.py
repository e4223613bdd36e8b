"""-m gpu: the device form of the generalized-Cauchy-point search (lbfgsx_b_cauchy_scan, csrc/gcp_scan.cuh;
reference loop Cauchy.h:183-256).  By default the host keeps the reference's sequential form for the first 65536
crossings of a search, which covers everything the 1e-10 parity cases do; here LBFGSX_GCP_DEVICE_MIN=0 sends the
search to the device from the first crossing.  Single searches must agree with the oracle at the same tight
tolerances as the sequential form.  Whole trajectories are held to 1e-8 instead of 1e-10: f' is a cancelling sum
whose last bits depend on the summation order (tree vs left-to-right), and 20 L-BFGS-B iterations amplify that
~1e-13 difference to ~1.6e-10 (see Cauchy.h, device_switch)."""
import numpy as np
import pytest

import oracle_lib as O

import test_lbfgsb_gpu as T
from test_lbfgsb_gpu import A, boracle  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def device_search(monkeypatch):
    monkeypatch.setenv("LBFGSX_GCP_DEVICE_MIN", "0")


@pytest.mark.parametrize("n,m,npairs,mode", [(3000, 6, 0, "hard"), (3000, 6, 4, "hard"), (5000, 6, 9, "edge"),
                                             (4096, 8, 8, "gentle"), (2500, 5, 5, "edge"), (64, 3, 2, "hard"),
                                             (70000, 10, 10, "hard"), (300000, 7, 7, "edge")])
def test_device_search_matches_oracle(A, boracle, device_search, n, m, npairs, mode):
    T.test_cauchy_and_subspace_match_oracle(A, boracle, O.F64, n, m, npairs, mode)


@pytest.mark.parametrize("inst", T.GOLD["instances"], ids=["seed%d" % i["seed"] for i in T.GOLD["instances"]])
def test_device_search_golden_instances(A, device_search, inst):
    T.test_cauchy_subspace_golden(A, inst)


@pytest.mark.parametrize("case", T.GOLD["trajectories"], ids=[c["name"] for c in T.GOLD["trajectories"]])
def test_device_search_golden_trajectories(A, device_search, case):
    T.test_lbfgsb_trajectory_golden(A, case, tol=1e-8)


@pytest.mark.parametrize("n,m,iters", [(2000, 6, 15), (20000, 10, 25)])
def test_device_search_trajectory(A, boracle, device_search, n, m, iters):
    T.test_trajectory_box_quadratic_f64(A, boracle, n, m, iters, tol=1e-8)


@pytest.mark.parametrize("n,m,npairs,mode", [(50000, 6, 6, "hard"), (200000, 10, 10, "edge"), (4096, 8, 0, "hard")])
def test_device_and_host_search_agree(A, monkeypatch, n, m, npairs, mode):
    """same instance through both forms: identical crossing count and sets; xcp / c agree up to the summation-order
    noise of the cancelling f' sum, which grows with the number of crossings (~1e5 here)"""
    rng = np.random.default_rng(7 + n)
    S, Y, x0, g, lb, ub = T._instance(rng, n, npairs, O.F64, mode)
    monkeypatch.setenv("LBFGSX_GCP_DEVICE_MIN", "-1")
    host = T._device_cauchy_subspace(A, O.F64, m, S, Y, x0, g, lb, ub, subspace=False)
    monkeypatch.setenv("LBFGSX_GCP_DEVICE_MIN", "0")
    dev = T._device_cauchy_subspace(A, O.F64, m, S, Y, x0, g, lb, ub, subspace=False)
    assert dev["crossings"] == host["crossings"] and dev["crossings"] > 0
    assert np.array_equal(dev["state"], host["state"])
    assert np.abs(dev["xcp"] - host["xcp"]).max() <= 1e-10 * max(1.0, np.abs(host["xcp"]).max())
    if host["vecc"].size:
        assert np.abs(dev["vecc"] - host["vecc"]).max() <= 1e-9 * max(1.0, np.abs(host["vecc"]).max())


def test_switch_mid_search(A, monkeypatch):
    """host form for the first 300 crossings, then the device: same result as either alone"""
    n, m, npairs = 30000, 8, 8
    rng = np.random.default_rng(11)
    S, Y, x0, g, lb, ub = T._instance(rng, n, npairs, O.F64, "hard")
    monkeypatch.setenv("LBFGSX_GCP_DEVICE_MIN", "-1")
    host = T._device_cauchy_subspace(A, O.F64, m, S, Y, x0, g, lb, ub, subspace=False)
    assert host["crossings"] > 300
    monkeypatch.setenv("LBFGSX_GCP_DEVICE_MIN", "300")
    mix = T._device_cauchy_subspace(A, O.F64, m, S, Y, x0, g, lb, ub, subspace=False)
    assert mix["crossings"] == host["crossings"]
    assert np.array_equal(mix["state"], host["state"])
    assert np.abs(mix["xcp"] - host["xcp"]).max() <= 1e-10 * max(1.0, np.abs(host["xcp"]).max())
