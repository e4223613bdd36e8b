"""-m gpu: the one JSON line bench.py prints is what the driver parses; check its schema on a shortened run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "12"] + list(extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout: %r" % lines
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _run("--cpu-n", "2000000", "--cpu-steps", "4", "--cpu-n-all", "2000000")
    assert d["metric"] == "L-BFGS iterations/sec at n=10^8, m=10; achieved HBM GB/s vs peak"   # BASELINE.json's metric
    assert d["unit"] == "iterations/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 12
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 10 and abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.7 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 1.1 * r["algorithmic_bytes_per_launch"]
    assert d["config"]["apply_Hv_persistent_launches"] > 0
    c = d["cpu_baseline"]
    assert c["unit"] == "iterations/s" and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0 and c["sample"]
    a = d["cpu_baseline_all_cores"]
    assert a is None or (a["cores"] >= 1 and a["value"] > 0)


def test_opt_in_modes_are_labelled_as_such():
    for flag, word in (("gram", "Gram-space"), ("gram-f32h", "f32 history")):
        d = _run("--recursion", flag, "--no-cpu")
        assert word in d["metric"] and "not the bit-parity path" in d["metric"]
        assert d["config"]["recursion"].startswith("gram-space") and d["roofline"]["kernel"].startswith("k_gs_post")
    d = _run("--workload", "sharded", "--no-cpu")
    assert d["scaling"] == "strong" and "row-sharded" in d["metric"] and d["config"]["rows_per_gpu"] == 100000000
