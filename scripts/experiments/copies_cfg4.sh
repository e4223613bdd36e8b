# which copies does a cfg4 iteration issue?  duration histogram of __amd_rocclr_copyBuffer and the kernel before each
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 40 > /tmp/pg.json 2>/dev/null
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pg/b_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = collections.Counter(); dur = collections.defaultdict(list)
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        p = rows[i-1]["Kernel_Name"][:60] if i else "-"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev[p] += 1; dur[p].append(d)
for p, c in prev.most_common(25):
    v = dur[p]; print("%4d  avg %7.1f us  max %7.1f  after %s" % (c, sum(v)/len(v), max(v), p))
PY
