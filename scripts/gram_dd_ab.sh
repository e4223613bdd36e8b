#!/bin/bash
# k_gram_dd on cfg4 under rocprofv3: the kernel's average duration, steady-state it/s and the final objective
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_lbfgsb_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -1
O=gpurun_out/gram_ab/cur; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/bench.json 2> /dev/null
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/b_kernel_stats.csv")))
d = json.load(open("$O/bench.json"))
for r in rows:
    if "k_gram_dd<double" in r["Name"]:
        print("%s avg %.1f us over %s calls" % (r["Name"][:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("steady %.1f it/s; overall %.1f; fx %.10g" % (d["steady_it_per_s"], d["it_per_s"], d["fx"]))
PY
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled: steady %.1f overall %.1f' % (d['steady_it_per_s'], d['it_per_s']))"
