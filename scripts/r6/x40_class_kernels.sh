#!/bin/bash
# per-kernel averages of cfg4 at m = 20 with each split (rocprofv3 --kernel-trace --stats), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for v in base x202; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  rm -rf /tmp/px_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$v -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m 20 --iters 60 > /dev/null 2>&1
  echo "== $v"; f=$(find /tmp/px_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-70s calls %4s avg %8.1f us  %5s %%" % (r["Name"][:70].replace("void lbfgsx::xl::",""), r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
