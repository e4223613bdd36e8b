"""-m gpu: the reference's own example programs, compiled UNMODIFIED against the drop-in headers (tests/ref_examples.py,
built by __graft_entry__.build() in the container that holds /root/reference), run on the MI355X next to the same
sources compiled against the reference's headers: same iteration counts, same x and f(x) to 1e-10 (double) / 1e-4
(float), the self-checking loops pass."""
import os
import re
import subprocess

import numpy as np
import pytest

import ref_examples as RX

pytestmark = pytest.mark.gpu
NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|nan|inf)"


def _have(name):
    gpu, ref = RX.paths(name)
    return os.path.exists(gpu) and os.path.exists(ref)


def _run(path, timeout=600, args=()):
    env = {k: v for k, v in os.environ.items() if k != "LBFGSX_PERSIST_MIN_N"}   # the product's defaults (conftest.py)
    try:
        r = subprocess.run([path] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=timeout, env=env)
        return r.returncode, r.stdout, False
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        return None, out, True


def _fields(text):
    """{label: numbers} of an example's printout ("x =" is followed by the vector on the next line)."""
    d = {}
    m = re.search(r"^(\d+) iterations", text, re.M)
    d["niter"] = int(m.group(1))
    m = re.search(r"^x = \n(.*)$", text, re.M)
    d["x"] = np.array([float(v) for v in re.findall(NUM, m.group(1))])
    m = re.search(r"^f\(x\) = (.*)$", text, re.M)
    d["fx"] = float(m.group(1))
    m = re.search(r"^grad = (.*)$", text, re.M)
    if m:
        d["grad"] = np.array([float(v) for v in re.findall(NUM, m.group(1))])
    m = re.search(r"^(?:\|\|grad\|\||projected grad norm) = (.*)$", text, re.M)
    if m:
        d["gnorm"] = float(m.group(1))
    for key in ("approx_hess", "approx_inv_hess"):
        m = re.search(r"^%s = \n((?:[-+0-9 .eEnaif]+\n?)*)" % key, text, re.M)
        if m:
            d[key] = np.array([float(v) for v in re.findall(NUM, m.group(1))])
    return d


@pytest.mark.parametrize("name,tol", [("example-quadratic", 1e-10), ("example-rosenbrock", 1e-4), ("example-rosenbrock-box", 1e-10)])
def test_single_solve_examples_print_what_the_reference_prints(name, tol):
    if not _have(name):
        pytest.skip("tests/cpp/bin/%s.* not built (needs /root/reference at build time)" % name)
    gpu, ref = RX.paths(name)
    rc_g, out_g, _ = _run(gpu)
    rc_r, out_r, _ = _run(ref)
    assert rc_g == 0 and rc_r == 0, out_g[-2000:]
    g, r = _fields(out_g), _fields(out_r)
    assert g["niter"] == r["niter"]
    assert g["x"].shape == r["x"].shape and np.abs(g["x"] - r["x"]).max() <= tol
    assert abs(g["fx"] - r["fx"]) <= tol * max(1.0, abs(r["fx"]))
    for key in ("grad", "approx_hess", "approx_inv_hess"):
        if key in r:
            assert key in g and g[key].shape == r[key].shape
            scale = max(1.0, np.abs(r[key]).max())
            assert np.abs(g[key] - r[key]).max() <= (tol if key == "grad" else 1e-3 if tol > 1e-6 else 1e-7) * scale
    if "gnorm" in r:
        assert abs(g["gnorm"] - r["gnorm"]) <= tol * max(1.0, r["gnorm"])


def test_bracketing_example_self_check_passes():
    """examples/example-rosenbrock-bracketing.cpp: 8 dimensions x 1024 random starts, each solution within 1e-4 of the
    minimiser or the program throws; every block ends with "Test passed!".  Through the host-functor path every
    evaluation crosses PCIe (~200 us): the whole program takes over a minute on the GPU box, so by default it gets a time
    budget and the dimensions it completed are compared with the reference's output (at least two);
    LBFGSX_SLOW_TESTS=1 lets it finish (profiles/r3_reference_examples.txt holds such a run)."""
    name = "example-rosenbrock-bracketing"
    if not _have(name):
        pytest.skip("not built")
    gpu, ref = RX.paths(name)
    slow = os.environ.get("LBFGSX_SLOW_TESTS") == "1"
    rc_g, out_g, timed_out = _run(gpu, timeout=1800 if slow else 30)
    rc_r, out_r, _ = _run(ref)
    assert rc_r == 0 and out_r.count("Test passed!") == 8
    assert "Error is larger" not in out_g and "terminate" not in out_g, out_g[-2000:]
    if slow or not timed_out:
        assert rc_g == 0 and out_g == out_r, out_g[-2000:]
    else:
        done = out_g.count("Test passed!")
        assert done >= 2, out_g[-2000:]
        keep = out_g[:out_g.rindex("Test passed!") + len("Test passed!")]
        assert out_r.startswith(keep)


def _blocks(text):
    """the comparison example's per-dimension blocks that are complete: {n: [(calls, iterations) x 4]}"""
    out = {}
    for m in re.finditer(r"^n = (\d+)\n  Average #calls:\n((?:  LineSearch.*\n){4})", text, re.M):
        out[int(m.group(1))] = [tuple(int(v) for v in re.findall(r"(\d+) calls, (\d+) iterations", ln)[0])
                                for ln in m.group(2).strip().split("\n")]
    return out


def test_comparison_example_counts_match_the_reference():
    """examples/example-rosenbrock-comparison.cpp: four line searches x 12 dimensions x 1024 random starts through the
    host-functor path (every evaluation crosses PCIe: ~7e6 round trips in all).  The program validates every solution
    itself (it throws otherwise); its per-dimension averages of calls and iterations must equal the reference's.  The
    whole run takes ten minutes on the GPU box, so by default the test gives it a time budget and compares the dimensions
    it completed (at least the first); LBFGSX_SLOW_TESTS=1 lets it finish."""
    name = "example-rosenbrock-comparison"
    if not _have(name):
        pytest.skip("not built")
    gpu, ref = RX.paths(name)
    slow = os.environ.get("LBFGSX_SLOW_TESTS") == "1"
    rc_r, out_r, _ = _run(ref)
    assert rc_r == 0
    want = _blocks(out_r)
    assert sorted(want) == list(range(2, 25, 2))
    rc_g, out_g, timed_out = _run(gpu, timeout=3600 if slow else 40)
    assert "Error is larger" not in out_g and (timed_out or rc_g == 0), out_g[-2000:]
    got = _blocks(out_g)
    assert len(got) >= (12 if slow else 1), "only %d dimensions completed: %r" % (len(got), sorted(got))
    for n, rows in got.items():
        assert rows == want[n], "n = %d: %r vs the reference's %r" % (n, rows, want[n])


@pytest.mark.parametrize("policy", [0, 1, 2, 3])
def test_comparison_loop_solve_by_solve(policy):
    """tests/cpp/cmp_probe.cpp: the loop of the comparison example for n = 2 and one line-search policy (0 backtracking,
    1 bracketing, 2 Nocedal-Wright, 3 More-Thuente), printing every one of the 1024 solves -- start point, iteration
    count, call count, f and x with 17 digits.  The build against the drop-in headers (on the GPU, host-functor path)
    must print what the build against the reference's headers prints, character for character.  (Both draw their start
    points from the Eigen stand-in's own generator: std::rand() is shared with the GPU runtime's threads.)"""
    gpu, ref = RX.paths("cmp_probe")
    if not (os.path.exists(gpu) and os.path.exists(ref)):
        pytest.skip("tests/cpp/bin/cmp_probe.* not built")
    rc_g, out_g, _ = _run(gpu, timeout=300, args=(2, policy, 1024))
    rc_r, out_r, _ = _run(ref, args=(2, policy, 1024))
    assert rc_g == 0 and rc_r == 0, out_g[-2000:]
    lg, lr = out_g.strip().split("\n"), out_r.strip().split("\n")
    assert len(lr) == 1024 and len(lg) == 1024
    bad = [k for k in range(1024) if lg[k] != lr[k]]
    assert not bad, "%d solves differ, first: %s | %s" % (len(bad), lg[bad[0]], lr[bad[0]])
