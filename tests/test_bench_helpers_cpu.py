"""CPU: bench.py's static `traffic` lookups (committed PMC summaries under profiles/) and the summary scripts' parsing."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_leg_of_the_default_line_finds_its_pmc_summary():
    """`roofline.traffic` is only quoted when a committed summary was taken at exactly the leg's (n, m): the round's
    summaries serve the headline, cfg2, cfg3, cfg4 (m = 10, 20) and cfg5; anything else gets null, not a guess."""
    b = _bench()
    head = b.pmc_traffic(100_000_000, 10, fused=True)
    assert head and 2.9e9 < head["bytes_per_launch"] < 3.3e9 and head["source"].startswith("profiles/")
    c2, c3 = b.pmc_traffic(10_000_000, 10, fused=True), b.pmc_traffic(100_000_000, 20, fused=True)
    assert c2 and 1.6e8 <= c2["bytes_per_launch"] < 2.2e8          # 2 x 80 MB of history per step, q resident
    assert c3 and 2.9e9 < c3["bytes_per_launch"] < 3.3e9
    assert b.pmc_traffic(12345, 10, fused=True) is None and b.pmc_traffic(100_000_000, 7, fused=True) is None
    for m in (10, 20):
        t = b.leg_traffic("cfg4", 10_000_000, m)
        assert t and t["per"].startswith("L-BFGS-B iteration") and 5e9 < t["hbm_bytes"] < 2e10
        f = b.traffic_fields(t)
        assert f["traffic"] == t["hbm_bytes"] and f["traffic_static"] is True and "profiles/" in f["traffic_source"]
    t5 = b.leg_traffic("cfg5", 100_000, 10)
    assert t5 and 1.9e7 < t5["hbm_bytes"] < 2.6e7                   # (4m + 14) n 4 B = 2.16e7 by the model
    assert b.leg_traffic("cfg4", 10_000_000, 11) is None
    assert b.traffic_fields(None) == {"traffic": None, "traffic_static": None, "traffic_source": None}


def test_pmc_legs_matches_the_two_counter_passes_dispatch_by_dispatch(tmp_path):
    """scripts/r4/pmc_legs.py on a hand-made pair of passes: 12 warm-up + 4 iterations of a two-kernel 'solver'; FETCH_SIZE is
    doubled (gfx950), counters are KiB, the window is the second half of the iterations, delimited by the post kernel."""
    src = tmp_path / "prof"
    for sub, counter, val in (("cfg4_m10_pmc_fetch", "FETCH_SIZE", 100.0), ("cfg4_m10_pmc_write", "WRITE_SIZE", 50.0)):
        d = src / sub
        d.mkdir(parents=True)
        with open(d / "b_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            did = 0
            for it in range(12 + 40):
                for k in ("void lbfgsx::kx_rows<double, 10, 2, 1, false>(int)", "void lbfgsx::k_b_post_build<double>(int)"):
                    did += 1
                    scale = 1.0 if it < 12 + 20 else 2.0     # the window's iterations move twice the bytes
                    w.writerow([did, k, counter, val * scale])
    prof = tmp_path / "profiles"
    prof.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r4", "pmc_legs.py"), str(src), "rt"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    legs = json.load(open(prof / "rt_legs_pmc_summary.json"))["legs"]
    assert len(legs) == 1 and legs[0]["leg"] == "cfg4" and legs[0]["m"] == 10
    per_kernel = (2 * 100.0 + 50.0) * 1024
    assert abs(legs[0]["hbm_bytes"] - 2 * 2 * per_kernel) < 1e-6          # window: 2 kernels per iteration at scale 2
    assert abs(legs[0]["hbm_bytes_from_x0"] - 2 * per_kernel * 1.5) < 1e-6  # 20 iterations at 1, 20 at 2


def test_the_default_line_is_compact_and_ends_with_the_digest_of_every_leg():
    """The driver keeps the last 8 KB of bench.py's one line.  The default line is the compact form of the full object: notes and
    per-iteration lists dropped, every leg reduced to its figures, `legs_digest` appended as the last key.  Checked here on the
    full line of an earlier round (profiles/r4_bench_default.json, 15 KB)."""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r4_bench_default.json")))
    line = dict(full)
    for name in b.LEGS:
        line[name] = b.compact_leg(line[name])
    line = b.compact_line(line)
    line["legs_digest"] = b.legs_digest(full)
    raw = json.dumps(line)
    assert len(raw) < 7168 and list(line)[-1] == "legs_digest"
    tail = raw[-1500:]                       # what survives a cut well inside the driver's 8 KB
    for name in ("north_star", "cfg2", "cfg3", "cfg4_lbfgsb", "cfg4_m20", "cfg5_batched"):
        assert '"%s": {"value"' % name in tail
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]          # digit for digit
    assert line["cfg4_lbfgsb"]["config"]["fx"] == full["cfg4_lbfgsb"]["config"]["fx"]
    assert "note" not in line["cfg4_lbfgsb"]["roofline"] and "per_iteration_ms" not in line["cfg4_lbfgsb"]["config"]
    assert line["roofline"]["frac"] == float("%.7g" % full["roofline"]["frac"]) and "sample" in line["cpu_baseline"]
