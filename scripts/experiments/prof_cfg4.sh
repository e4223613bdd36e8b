cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 40 > /tmp/pg.json 2>/dev/null
python3 - <<PY
import csv, json
d = json.loads(open("/tmp/pg.json").read().strip().splitlines()[-1])
print("it/s from x0 %.1f  steady %.1f  sweeps %d" % (d["it_per_s"], d["steady_it_per_s"], d["stats"]["submin_sweeps"]))
rows = list(csv.DictReader(open("/tmp/pg/b_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    print("%5.1f%% %5s calls %8.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:70]))
PY
