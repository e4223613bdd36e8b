"""-m gpu: the one JSON line bench.py prints is what the driver parses; check its schema on a shortened run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, gpus=1, warmup=12, env=None, verbose=True):
    # verbose: the full line (every key the schema tests read); the default, compact line is checked by the first test
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", str(warmup)] + list(extra)
    if verbose:
        cmd.append("--verbose")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout: %r" % lines
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys(tmp_path):
    full = str(tmp_path / "full.json")
    line = _run("--cpu-n", "2000000", "--cpu-steps", "4", "--cpu-n-all", "2000000", "--cpu-full", "off", "--full-json", full,
                verbose=False)
    # The default line is the compact one: under 7 KB (the driver keeps the last 8 KB of it), the contract keys and the two
    # required objects at the top, every leg reduced to its figures, and `legs_digest` as the LAST key so that whatever tail
    # survives holds every leg's value and fractions.  The full object (--verbose prints it, --full-json writes it) is what
    # the rest of this test reads.
    raw = json.dumps(line)
    assert len(raw) < 7168, "the default line has grown to %d bytes" % len(raw)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert list(line)[-1] == "legs_digest"
    # `traffic` comes from committed PMC passes of the same command, not from this run: the default line says so itself
    assert line["roofline"]["traffic"] is None or line["roofline"]["traffic_static"] is True
    dig = line["legs_digest"]
    assert sorted(dig) == sorted(["north_star", "cfg2", "cfg3", "cfg4_lbfgsb", "cfg4_m20", "cfg5_batched"])
    for name, e in dig.items():
        assert e["value"] > 0 and e["ms_per_step"] > 0 and 0.0 < e["frac"] <= 1.0, (name, e)   # a fraction is at most 1
    assert dig["cfg4_lbfgsb"]["from_x0"] < dig["cfg4_lbfgsb"]["value"] and 0.0 < dig["cfg4_lbfgsb"]["frac_from_x0"] <= 1.0
    assert abs(dig["north_star"]["value"] - line["value"]) < 1e-3 * line["value"]
    for leg in ("cfg2", "cfg3", "cfg4_lbfgsb", "cfg4_m20", "cfg5_batched"):
        assert line[leg]["value"] > 0 and 0.0 < line[leg]["roofline"]["frac"] <= 1.0 and "workload" in line[leg]["config"]
    d = json.load(open(full))
    assert abs(d["value"] - line["value"]) < 1e-12 * d["value"] and d["ms_per_step"] == line["ms_per_step"]
    assert d["metric"] == "L-BFGS iterations/sec at n=10^8, m=10; achieved HBM GB/s vs peak"   # BASELINE.json's metric
    assert d["unit"] == "iterations/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 12
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 10 and abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.7 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 1.1 * r["algorithmic_bytes_per_launch"]
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert d["config"]["apply_Hv_persistent_launches"] > 0
    # the timed window holds full-history products only: 3 iterations x (2m+1) steps
    assert d["config"]["history_full"] is True and r["launches_timed"] == 3 * 21
    # the cfg4 legs' `frac` is the library's own byte model over time; the PMC-measured traffic of the same leg rides beside
    # it (`traffic_frac`) and the two must agree: a kernel change that moves bytes shows up as a model / traffic mismatch
    for leg in ("cfg4_lbfgsb", "cfg4_m20"):
        rr = d[leg]["roofline"]
        if rr.get("model_over_traffic") is not None and d[leg]["steps"] >= 40:
            assert 0.9 < rr["model_over_traffic"] < 1.1, (leg, rr["model_over_traffic"])
    b = d["cfg5_batched"]   # the batched mode of BASELINE.json's cfg5 rides in the same line
    assert b["unit"] == "problem-iterations/s" and b["n_gpus"] == 1 and b["value"] > 1e4 and b["config"]["failed"] == 0
    assert 0.1 < b["roofline"]["frac"] < 1.0
    # the leg times the solve alone (resident batch, setup outside the window) and shows the kernels' share of a step
    bc = b["config"]
    assert bc["one_launch_per_iteration"] is True and bc["setup_seconds"] > 0 and bc["wait_timeouts"] == 0
    assert 0.5 * bc["wall_ms_per_step"] < bc["kernel_ms_per_step"] <= 1.10 * bc["wall_ms_per_step"]
    assert line["cfg5_batched"]["kernel_ms_per_step"] == pytest.approx(bc["kernel_ms_per_step"], rel=1e-5)
    assert line["legs_digest"]["cfg5_batched"]["kernel_ms_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["unit"] == "iterations/s" and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0 and c["sample"]
    assert c["extrapolated"] is True and c["measured_n"] == 2000000
    a = d["cpu_baseline_all_cores"]
    assert a is None or (a["cores"] >= 1 and a["value"] > 0)
    # every other single-GPU configuration of BASELINE.json rides in the same line, each with its own roofline object
    for key, n, m in (("cfg2", 10000000, 10), ("cfg3", 100000000, 20)):
        g = d[key]
        assert g["unit"] == "iterations/s" and g["value"] > 10 and g["config"]["n"] == n and g["config"]["m"] == m
        assert g["config"]["history_full"] is True and "workload" in g["config"]
        assert abs(g["ms_per_step"] * g["value"] - 1e3) < 1e-3
        gr = g["roofline"]
        assert gr["bound"] == "hbm" and 0.2 < gr["frac"] < 1.0 and gr["kernel"] and gr["algorithmic_bytes_per_launch"] > 0
        assert gr["launches_timed"] == g["steps"] * (2 * m + 1)
    assert d["cfg2"]["roofline"]["q_resident_elems"] == 10000000      # cfg2: q never leaves the CUs
    c4 = d["cfg4_lbfgsb"]
    assert c4["unit"] == "iterations/s" and c4["value"] > 50 and c4["from_x0"]["value"] > 30 and c4["steps"] == 40
    assert c4["from_x0"]["value"] < c4["value"] and c4["from_x0"]["first_iteration_ms"] > c4["ms_per_step"]
    cc = c4["config"]
    assert cc["n"] == 10000000 and cc["m"] == 10 and cc["iterations"] == 40 and "workload" in cc
    assert 1.0 <= cc["q"] <= 10.0 and cc["n_ord"] > 1e5 and cc["n_sorted"] <= cc["n_ord"] and cc["gcp_crossings"] > 0
    assert 10 < cc["launches_per_iteration"] < 500 and 1 < cc["host_syncs_per_iteration"] < 200 and cc["copies_per_iteration"] >= 0
    r4 = c4["roofline"]
    assert r4["bound"] == "hbm" and r4["kernel"] and abs(r4["frac"] - r4["achieved"] / 8000.0) < 1e-12 and 0.1 < r4["frac"] < 1.0
    # the numerator is the byte model of the path as built (lbfgsx_counters_ex): what the launches of the window had to move
    assert abs(r4["achieved"] - r4["model_bytes"] * c4["value"] / 1e9) < 1e-6 * r4["achieved"]
    assert r4["frac_from_x0"] < r4["frac"] <= 1.0
    assert 2.0 <= cc["compact_passes_per_iteration"] <= 12.0 and 0.2 * cc["n"] < cc["n_free"] < 0.8 * cc["n"]
    # ... at least the columns of those passes, and within 10 % of what the counters saw where a PMC summary of this very
    # leg is committed (profiles/*_legs_pmc_summary.json)
    assert r4["model_bytes"] > cc["compact_passes_per_iteration"] * cc["n_free"] * 2 * cc["m"] * 8
    if r4["traffic"] is not None:
        assert abs(r4["model_over_traffic"] - r4["model_bytes"] / r4["traffic"]) < 1e-12
        assert 0.90 <= r4["model_over_traffic"] <= 1.10, "byte model %.3g vs counters %.3g" % (r4["model_bytes"], r4["traffic"])
    # both sides of the steady fraction come from the SAME iterations: the second half of the run (the library's counters and
    # the solver's statistics snapshotted at the iteration hook), value = 1 / mean, the median beside it
    w = cc["window"]
    assert w["first_iteration"] == 20 and w["last_iteration"] == 39 and w["iterations"] == 20 and w["history_full"] is True
    per = cc["per_iteration_ms"]
    assert len(per) == 39 and abs(sum(per[19:]) / 20 - c4["ms_per_step"]) < 2e-3
    assert abs(c4["ms_per_step"] * c4["value"] - 1e3) < 1e-6 * 1e3 and abs(c4["ms_per_step_median"] * c4["value_median"] - 1e3) < 1e-3
    assert cc["submin_sweeps"] <= cc["submin_sweeps_total"] and cc["submin_calls"] == 20
    assert abs(cc["q"] - cc["submin_sweeps"] / cc["submin_calls"]) < 1e-12
    m_, n_ = cc["m"], cc["n"]
    want = ((4 * m_ + 19) + (cc["q"] + 1.0) * (4 * m_ + 1)) * n_ * 8 + 96.0 * cc["n_sorted"]
    assert abs(r4["reference_statement_bytes"] - want) < 1e-9 * want   # SURVEY 8(d)'s statement count, for comparison only
    assert r4["model_bytes"] < r4["reference_statement_bytes"]
    fx0 = c4["from_x0"]
    assert fx0["q"] >= 1.0 and fx0["launches_per_iteration"] > cc["launches_per_iteration"] * 0.5
    assert (r4["traffic"] is None) == (r4["traffic_source"] is None)
    # the same problem with the largest history a BASELINE configuration uses: the L-BFGS-B path is generic in m
    c20 = d["cfg4_m20"]
    assert c20["config"]["m"] == 20 and c20["steps"] == 60 and c20["config"]["window"]["history_full"] is True
    assert c20["config"]["window"]["first_iteration"] == 30 and c20["config"]["gram_carried"] >= 15
    assert c20["value"] > 0.45 * c4["value"], "m = 20 moves 1.9x the bytes of m = 10: %.1f vs %.1f it/s" % (c20["value"], c4["value"])
    assert 0.1 < c20["roofline"]["frac"] <= 1.0 and 1.15 * r4["model_bytes"] < c20["roofline"]["model_bytes"] < 2.2 * r4["model_bytes"]
    for leg in ("cfg2", "cfg3", "cfg5_batched"):
        rr = d[leg]["roofline"]
        assert (rr["traffic"] is None) == (rr.get("traffic_source") is None)


def test_cpu_baseline_at_the_metric_size_is_measured_not_scaled():
    """--cpu-full on (at a size this test can afford): `cpu_baseline` is the un-extrapolated figure of ONE run whose
    iteration boundaries come from the functor-call clock; the scaled small-n sample rides along."""
    d = _run("--n", "4000000", "--cpu-full", "on", "--cpu-n", "1000000", "--cpu-steps", "3", "--cpu-n-all", "1000000",
             "--no-batched", "--no-legs")
    c = d["cpu_baseline"]
    assert c["extrapolated"] is False and c["measured_n"] == 4000000 and c["cores"] == 1 and c["timed_iterations"] == 3
    assert "boundaries from the functor-call clock" in c["sample"] and c["value"] > 0
    s = d["cpu_baseline_sample"]
    assert s["extrapolated"] is True and s["measured_n"] == 1000000
    assert 0.3 < c["value"] / s["value"] < 3.0   # the scaled sample and the measured point agree in magnitude


def test_opt_in_modes_are_labelled_as_such():
    for flag, word in (("gram", "Gram-space"), ("gram-f32h", "f32 history")):
        d = _run("--recursion", flag, "--no-cpu")
        assert word in d["metric"] and "not the bit-parity path" in d["metric"]
        assert d["config"]["recursion"].startswith("gram-space") and d["roofline"]["kernel"].startswith("k_gs_post")
    d = _run("--workload", "sharded", "--no-cpu")
    assert d["scaling"] == "strong" and "row-sharded" in d["metric"] and d["config"]["rows_per_gpu"] == 100000000
    # the sums cross the shards through the library's own RCCL all-reduce (here a communicator of one rank)
    ar = d["config"]["allreduce"]
    assert ar["transport"].startswith("RCCL") and ar["ranks"] == 1 and ar["rccl_version_code"] > 20000
    assert "all-reduces of <=" in d["config"]["workload"] and " 0 all-reduces" not in d["config"]["workload"]


def test_sharded_workload_from_one_process_over_a_device_list():
    """--workload sharded --single-process --gpus 2: ONE problem, two row blocks, two host threads of one process, the
    sums through lbfgsx_comm_allreduce_sum.  With two GPUs the transport is RCCL; the one-GPU box lists device 0 twice and
    the communicator adds the bundles in host memory."""
    sys.path.insert(0, ROOT)
    import lbfgspp_amd as A
    ndev = A.load()[0].lbfgsx_device_count()
    env = {} if ndev >= 2 else {"LBFGSX_BENCH_DEVICES": "0,0"}
    d = _run("--workload", "sharded", "--single-process", "--no-cpu", "--n", "40000000", gpus=2, env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["n"] == 40000000
    assert d["config"]["rows_per_gpu"] == 20000000 and d["config"]["allreduce"]["ranks"] == 2
    assert d["config"]["allreduce"]["transport"].startswith("RCCL" if ndev >= 2 else "host memory")
    assert [p["rows"] for p in d["per_rank"]] == [20000000, 20000000]


def test_history_is_full_whatever_the_warmup():
    """timed iterations run with c = m.  --warmup 0 and --warmup 12 time the same work."""
    a = _run("--no-cpu", "--no-batched", "--no-legs", warmup=0)
    b = _run("--no-cpu", "--no-batched", "--no-legs", warmup=12)
    for d in (a, b):
        assert d["config"]["history_full"] is True and d["roofline"]["launches_timed"] == 3 * 21
    assert a["config"]["warmup_run"] == 10 and b["config"]["warmup_run"] == 12 and a["warmup"] == 0
    assert abs(a["value"] / b["value"] - 1.0) < 0.1
    assert abs(a["roofline"]["apply_Hv_GBs"] / a["roofline"]["algorithmic_GBs"] - 1.0) < 0.02


def test_small_problem_reports_the_hbm_model_fraction():
    """cfg2 size: q lives on the CUs, the algorithmic figure exceeds the HBM peak; `frac` stays a roofline fraction."""
    d = _run("--objective", "quadratic", "--n", "10000000", "--no-cpu", "--no-batched", "--no-legs")
    r = d["roofline"]
    assert r["frac"] < 1.0 and r["q_resident_elems"] == 10000000 and r["hbm_model_GBs"] < r["algorithmic_GBs"]
    assert (r["traffic"] is None) == (r["traffic_source"] is None)   # quoted only from a committed PMC summary of this (n, m)


def test_gpus_2_starts_two_ranks_or_refuses():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks.  On a one-GPU box they share device 0 over
    gloo (protocol test); without that override the script must refuse rather than report one GPU."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import lbfgspp_amd as A
    ndev = A.load()[0].lbfgsx_device_count()
    env = {} if ndev >= 2 else {"LBFGSX_BENCH_FORCE_DEVICE": "0"}
    d = _run("--no-cpu", "--n", "20000000", "--problems-per-gpu", "256", gpus=2, env=env)
    assert d["n_gpus"] == 2 and d["cfg5_batched"]["n_gpus"] == 2 and d["cfg5_batched"]["config"]["problems_total"] == 512
    assert d["config"]["history_full"] is True
    # the N > 1 line explains itself: what every rank saw, which collective library carried the (non data-path) exchanges
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["value"] > 0 and p["stream_copy_GBs"] > 0 for p in d["per_rank"])
    assert d["collective"]["world_size"] == 2 and d["collective"]["backend"] in ("nccl", "gloo")
    assert d["collective"]["data_path_collectives"] == 0
    # the rank count is the collective library's own (a sum of ones over the timing communicator), not the launcher's claim
    assert d["collective"]["ranks_counted_by_allreduce"] == 2 and len(d["collective"]["devices"]) == 2
    pr = d["cfg5_batched"]["config"]["per_rank"]
    assert [(p["first_problem"], p["problems"]) for p in pr] == [(0, 256), (256, 256)]
    if ndev < 2:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu"]
        env0 = {k: v for k, v in os.environ.items() if k not in ("LBFGSX_BENCH_FORCE_DEVICE", "WORLD_SIZE", "RANK")}
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env0)
        assert r.returncode != 0 and not r.stdout.strip()


def test_single_process_drives_every_listed_device():
    """--single-process --gpus 2: ONE process, one host thread + context per listed device, the batch through
    one resident lock-step batch per listed device and its records gathered natively over RCCL.  On the one-GPU box the device list
    is {0, 0} (LBFGSX_BENCH_DEVICES)."""
    sys.path.insert(0, ROOT)
    import lbfgspp_amd as A
    ndev = A.load()[0].lbfgsx_device_count()
    env = {} if ndev >= 2 else {"LBFGSX_BENCH_DEVICES": "0,0"}
    d = _run("--single-process", "--no-cpu", "--n", "20000000", "--problems-per-gpu", "128", gpus=2, env=env)
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2 and [p["rank"] for p in d["per_rank"]] == [0, 1]
    assert all(p["value"] > 0 and p["stream_copy_GBs"] > 0 for p in d["per_rank"])
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    assert "one process" in d["config"]["process_model"]
    b = d["cfg5_batched"]
    assert b["n_gpus"] == 2 and b["config"]["problems_total"] == 256 and b["config"]["failed"] == 0
    assert b["config"]["rccl_ranks"] == min(2, ndev) and b["config"]["rccl_allgather_seconds"] > 0
    if ndev < 2:   # asking for two devices on a one-GPU box without the override is refused
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--steps", "2", "--no-cpu"]
        env0 = {k: v for k, v in os.environ.items() if k not in ("LBFGSX_BENCH_DEVICES", "WORLD_SIZE", "RANK")}
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env0)
        assert r.returncode != 0 and not r.stdout.strip()


def test_north_star_parity_object_at_a_size_this_test_can_afford():
    """`parity` of the default line: the reference's own run (native accumulators, the same instance, --cpu-full) against the
    GPU solver, all n coordinates.  Here at n = 1e7 (the default line does it at n = 1e8); --cpu-full-dd adds the comparison
    with the reference built on the parity contract's extended sums, where nothing may differ."""
    d = _run("--n", "10000000", "--cpu-full", "on", "--cpu-full-dd", "--cpu-n", "1000000", "--cpu-steps", "3", "--cpu-n-all", "1000000",
             "--no-batched", "--no-legs")
    p = d["parity"]
    assert p["n"] == 10000000 and p["m"] == 10 and p["iterations"] == 15 and p["iterations_equal"] and p["nfev_equal"]
    assert p["max_abs_dx"] <= 1e-10 and p["within_tolerance"] is True and p["fx_rel_diff"] <= 1e-11
    assert "native accumulators" in p["oracle"] and p["max_abs_dx_strided_sample"] <= p["max_abs_dx"]
    q = d["parity_dd"]
    assert q["iterations_equal"] and q["nfev_equal"] and q["max_abs_dx"] == 0.0 and q["fx"] == q["fx_oracle"]
