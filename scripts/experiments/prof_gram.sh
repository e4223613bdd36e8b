cd /tmp; export TMPDIR=/tmp
for g in dd i8; do
rm -rf /tmp/pg; LBFGSX_GRAM=$g rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 40 > /tmp/pg.json 2>/dev/null
echo "LBFGSX_GRAM=$g"; python3 - <<PY
import csv, json
d = json.loads(open("/tmp/pg.json").read().strip().splitlines()[-1])
print("  it/s from x0 %.1f  steady %.1f" % (d["it_per_s"], d["steady_it_per_s"]))
for r in csv.DictReader(open("/tmp/pg/b_kernel_stats.csv")):
    if "gram" in r["Name"] or "k_b_post" in r["Name"]:
        print("  %-44s calls %4s avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
