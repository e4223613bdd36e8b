# per-call durations (us) of the kernels whose name contains $1 during the cfg4 run, in launch order
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 40 > /tmp/pg.json 2>/dev/null
python3 - "$@" <<PY
import csv, sys
rows = list(csv.DictReader(open("/tmp/pg/b_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for pat in sys.argv[1:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    g = [r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in rows if pat in r["Kernel_Name"]]
    print(pat, len(d), "calls; us:", " ".join("%.0f" % x for x in d))
    print("   grid:", " ".join(g[:80]))
PY
