# full-Gram passes at several history lengths: the double-double VALU kernel against the exact integer-MFMA kernel
cd /tmp; export TMPDIR=/tmp
for m in ${MS:-10 12 14 15}; do
for g in ${GS:-dd i8}; do
rm -rf /tmp/pg; env $( [ "$g" = default ] || echo LBFGSX_GRAM=$g ) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --m $m --iters 50 > /tmp/pg.json 2>/dev/null
echo "m=$m LBFGSX_GRAM=$g"; python3 - <<PY
import csv, json
d = json.loads(open("/tmp/pg.json").read().strip().splitlines()[-1])
print("  it/s from x0 %.1f  steady %.1f  fx %.17g" % (d["it_per_s"], d["steady_it_per_s"], d["fx"]))
for r in csv.DictReader(open("/tmp/pg/b_kernel_stats.csv")):
    if ("gram" in r["Name"] and "finish" not in r["Name"]) and float(r["AverageNs"]) > 100e3:
        print("  %-52s calls %4s avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
done
